"""Host-side profile of the bench step (cProfile) - where does the Python/launch time go?"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
model = bench.build_scene(torch.device("cuda", 0), 199, 128, True)
step = bench.Step(model, torch.device("cuda", 0), 2048, 262144, 1, 0, "cfg3")
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(10): step()
pr.disable()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 100)
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(38)
