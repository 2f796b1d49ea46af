import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
workload = sys.argv[1]
model = bench.build_scene(dev, 199, 128, True)
if workload == "cfg2":
    model.nvfi.use_vel = False
step = bench.Step(model, dev, 2048, 262144, 1, 0, workload)
if len(sys.argv) > 2 and sys.argv[2] == "nostreams":
    step.streams = None
for _ in range(4): step()
torch.cuda.synchronize(); print("eager ok", flush=True)
gs = bench.GraphedStep(step)
gs.host_record(); torch.cuda.synchronize(); print("record ok", flush=True)
cap = torch.cuda.Stream(device=dev); cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    gs.body()
torch.cuda.current_stream().wait_stream(cap); torch.cuda.synchronize(); print("eager body ok", float(gs.loss), flush=True)
gs.host_record(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); g.register_generator_state(step.gen)
with torch.cuda.graph(g, stream=cap):
    gs.body()
torch.cuda.synchronize(); print("capture ok", flush=True)
g.replay(); torch.cuda.synchronize(); print("replay 1 ok", float(gs.loss), flush=True)
gs.graph = g
for i in range(5):
    gs(); torch.cuda.synchronize(); print("replay ok", i, float(gs.loss), flush=True)
