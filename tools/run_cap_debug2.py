"""debug: the captured two-rank step (tests/test_gpu_dist.py::_worker_graph) with a synchronise + print behind every graph replay"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, port, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      NVFI_BENCH_BACKEND="gloo", NVFI_OVERLAP="1")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    dev = torch.device("cuda", 0)
    torch.manual_seed(7); torch.cuda.manual_seed(7)
    model = bench.build_scene(dev, 64, 64, True)
    step = bench.Step(model, dev, 512, 8192, world, rank, "cfg3", live=True)
    step()
    gs = bench.GraphedStep(step)
    s = gs.s
    variant = os.environ.get("CAPV", "")
    if variant:
        def body_pde_v():
            m = s.m
            if "nostage" not in variant:
                s.pde_stage.zero()
                m.vel_grad_targets = s.pde_stage.views
            else:
                m.vel_grad_targets = None
            m.vel_loss_weight = 1.0 if "floatw" in variant else gs.rec[5:6]
            if "nopde" not in variant:
                s.last_lv = m.get_vel_loss(s.n_pts, points=s.d_pts, t=s.d_t)
                gs.pde_out = m.nvfi.last_pde_out
        gs.body_pde = body_pde_v
    def P(msg):
        torch.cuda.synchronize(); print(f"[rank {rank}] {msg}", flush=True)
    if not s.stepped: s.bucket.zero()
    gs.host_record(); torch.cuda.synchronize()
    gs._multi_iteration(gs.body_head, gs.body_pde, gs.body_renders); P("eager iteration done")
    gs.flags.clear(); gs.host_record(); torch.cuda.synchronize()
    gs.gH, gs.gP, gs.gR = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    for name, g, body in (("H", gs.gH, gs.body_head), ("P", gs.gP, gs.body_pde), ("R", gs.gR, gs.body_renders)):
        cap = torch.cuda.Stream(device=s.dev)
        cap.wait_stream(torch.cuda.current_stream())
        s.m.nvfi.frags_packed_by_earlier_graph(g is not gs.gH)
        try:
            with torch.cuda.graph(g, stream=cap):
                body()
        finally:
            s.m.nvfi.frags_packed_by_earlier_graph(False)
        torch.cuda.current_stream().wait_stream(cap)
        P(f"captured {name}")
    for it in range(3):
        gs.host_record(); torch.cuda.synchronize()
        if mode in ("all", "H"): gs.gH.replay(); P(f"it {it}: replayed H")
        if mode in ("all", "P"): gs.gP.replay(); P(f"it {it}: replayed P")
        if mode in ("all", "R"): gs.gR.replay(); P(f"it {it}: replayed R")
    P("done")

if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    mp.spawn(worker, args=(2, 29871, mode), nprocs=2, join=True)
