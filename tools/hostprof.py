"""Host-side profile of the bench step (cProfile) - where does the Python/launch time go?
    python tools/hostprof.py [fused|dropin] [cfg3|cfg2]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
workload = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
dev = torch.device("cuda", 0)
model = bench.build_scene(dev, 199, 128, True)
if workload == "cfg2":
    model.nvfi.use_vel = False
cls = bench.DropinStep if mode == "dropin" else bench.Step
step = cls(model, dev, 2048, 262144, 1, 0, workload)
for _ in range(8): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print(mode, workload, "ms/step (no profiler)", (time.perf_counter() - t0) * 50)
if mode == "dropin" and getattr(step, "_tm", None):
    import numpy as np
    rows = np.array([r for r in step._tm[-20:] if len(r) == len(step._tm[-1])])
    names = ["render1 issued", "sync1 (.item)", "render2 + regularisers issued", "get_vel_loss returned (sync2)", "loss arithmetic", "backward() returned",
             "optimizer.step() returned", "sync3 (.item)"]
    d = np.diff(rows, axis=1).mean(0) * 1e3
    print("host stations (ms, mean of 20): " + " | ".join(f"{n} {v:.3f}" for n, v in zip(names, d)) + f" | between iterations {np.diff(rows[:, 0]).mean() * 1e3 - d.sum():.3f}")
# host issue time alone: how long does the host need to queue a step when it never waits for the device?
if mode != "dropin":
    t0 = time.perf_counter()
    for _ in range(20): step()
    th = (time.perf_counter() - t0) * 50
    torch.cuda.synchronize()
    print("host issue ms/step (queue-ahead, includes back-pressure)", th)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(10): step()
pr.disable()
torch.cuda.synchronize()
print("ms/step under cProfile", (time.perf_counter() - t0) * 100)
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
