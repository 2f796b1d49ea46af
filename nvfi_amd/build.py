"""Builds libnvfi_hip.so (gfx950) in-tree with hipcc.  `python -m nvfi_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["engine.hip", "wgrad_ring.hip", "vel.hip", "render.hip", "scatter.hip", "mask.hip", "pde.hip", "pde_jet.hip", "pre16.hip", "vel_split.hip", "vel_fuse.hip", "pde_fuse.hip", "regs.hip", "optim.hip", "abi.hip", "comm.hip", "frags.hip", "vel_x6.hip", "vel_x6w.hip", "pde_jet6.hip"]
# every header under csrc/ (engine16.h, ... - a header that is not listed here would leave stale objects behind) + the public ABI
HEADERS = sorted(h for h in os.listdir(CSRC) if h.endswith(".h")) + [os.path.join("..", "..", "include", "nvfi_hip.h")]
SO = os.environ.get("NVFI_BUILD_SO", os.path.join(CSRC, "libnvfi_hip.so"))   # experiments build a second library elsewhere
OBJDIR = os.environ.get("NVFI_BUILD_OBJDIR", CSRC)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"] + os.environ.get("NVFI_EXTRA_FLAGS", "").split()
# per-file flags.  vel_x6.hip: no SLP vectorisation, i.e. no packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) in the
# kernels that run two workgroups per CU beside 16-bit MFMAs - with them one encoder input of one wave came out wrong in lanes 48..63
# once per few hundred tiles (delta debugging in DESIGN.md 4.8; tests/test_gpu_x6.py repeats 4 M points bit for bit)
FILE_FLAGS = {"vel_x6.hip": ["-fno-slp-vectorize"], "vel_x6w.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"],      # (accumulators in VGPRs: the drain reads them without 48 v_accvgpr_read per tile)
              # round 6: the same fence for the other units that issue 16-bit MFMAs with more than one wave per SIMD (pre16.hip: eight waves per
              # workgroup; mask.hip: the fp16 MaskField kernels) - the mechanism of the x6 glitch is not understood, so the recipe (packed fp32
              # beside 16-bit MFMAs) is kept out of every such unit, and check_no_packed_f32() below fails the build if it comes back
              "pre16.hip": ["-fno-slp-vectorize"], "mask.hip": ["-fno-slp-vectorize"],
              # round 6: the fused RK2 adjoint runs its dgrad on bf16 MFMAs (x6) with three waves per SIMD
              "vel_fuse.hip": ["-fno-slp-vectorize"], "pde_jet6.hip": ["-fno-slp-vectorize"]}
NO_PACKED_F32 = ["vel_x6.hip", "vel_x6w.hip", "pre16.hip", "mask.hip", "vel_fuse.hip", "pde_jet6.hip"]
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def packed_f32_count(obj):
    """number of v_pk_*_f32 instructions in the gfx950 code object bundled in `obj` (None if llvm-objdump is not there)"""
    import re
    import shutil
    import tempfile
    if not os.path.exists(OBJDUMP):
        return None
    with tempfile.TemporaryDirectory() as td:
        o = os.path.join(td, os.path.basename(obj))
        shutil.copy(obj, o)
        subprocess.run([OBJDUMP, "--offloading", o], cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [f for f in os.listdir(td) if "amdgcn" in f]
        if not cos:
            return None
        dis = subprocess.run([OBJDUMP, "-d", os.path.join(td, cos[0])], capture_output=True, text=True).stdout
    return len(re.findall(r"\bv_pk_(?:mul|add|fma)_f32\b", dis))


def check_no_packed_f32(objs):
    """a compiler change or an edit that re-introduces packed-fp32 VALU code beside the 16-bit MFMAs must not pass silently"""
    bad = {}
    for s in NO_PACKED_F32:
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        if obj in objs and os.path.exists(obj):
            n = packed_f32_count(obj)
            if n:
                bad[s] = n
    if bad:
        raise RuntimeError(f"packed-fp32 VALU instructions in units that must not have them (DESIGN.md 4.8.3): {bad}")


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = [os.path.join(CSRC, h) for h in HEADERS if os.path.exists(os.path.join(CSRC, h))]
    objs = []
    todo = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(d, obj) for d in deps):
            todo.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(cc, todo))
        check_no_packed_f32([obj for _, obj in todo])
    if todo or not os.path.exists(SO):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
