"""Round 6: the velocity net's BACKWARD kernels on the 16-bit matrix pipe (x6: fp32 products formed exactly from three bfloat16 terms per
operand, nvfi_amd/csrc/x6.h) against the fp32 MFMA kernels they replace.

* k_rk2_fuse_bwd<true> (vel_fuse.hip): the RK2 adjoint's dgrad from the transposed weight images, the hidden layers' weight gradients from
  operands transposed by selection-matrix MFMAs - NVFI_FUSE_X6=0 keeps round 4's fp32 kernel;
* k_pde_jet6_fwd (pde_jet6.hip): the Jacobian program's forward - NVFI_PDE_JET_X6=0 keeps k_pde_jet_fwd.
Same-results rule of round 5: the x6 kernels must agree with the fp32 ones to the rounding of a summation order (2e-6 max-norm / L2 here, 250 x
under the suite's 5e-4 gradient tolerance), be bit-repeatable from run to run at the occupancy they are launched with (three waves per SIMD
beside bf16 MFMAs - the configuration class of round 5's glitch), and both must sit on the reference goldens
(tests/test_gpu_parity.py runs with the defaults, i.e. with x6)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, relerr

pytestmark = pytest.mark.gpu


def _run(tmp_path, **env):
    out = str(tmp_path / ("x6bwd_" + "_".join(f"{k}{v}" for k, v in env.items()) + ".npz"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "x6_bwd_check.py"), out], env=dict(os.environ, **{k: str(v) for k, v in env.items()}),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    print(r.stdout.strip().splitlines()[-1])
    return np.load(out)


def test_x6_backward_kernels_match_the_fp32_kernels_and_repeat(tmp_path):
    a = _run(tmp_path, NVFI_FUSE_X6=0, NVFI_PDE_JET_X6=0)
    b = _run(tmp_path, NVFI_FUSE_X6=1, NVFI_PDE_JET_X6=1)
    assert int(a["render:evals"][3]) > 200000 and int(a["pde:kept"]) > 4000 and int(a["pde:kept"]) == int(b["pde:kept"])
    # the hidden layers' gradients come out of per-workgroup slabs summed in a fixed order: identical bits from run to run, for both kernels
    assert bool(a["render:repeat_identical"]) and bool(b["render:repeat_identical"])
    np.testing.assert_allclose(float(b["render:loss"]), float(a["render:loss"]), rtol=1e-6)
    np.testing.assert_allclose(float(b["pde:loss"]), float(a["pde:loss"]), rtol=2e-6)
    worst = {}
    for k in a.files:
        if ":vel_net." in k:
            e = relerr(b[k], a[k])
            worst[k.split(":")[0]] = max(worst.get(k.split(":")[0], 0.0), e)
            assert e < 2e-6, (k, e)
    print("x6 vs fp32 backward kernels, worst max(max-norm, L2) relative difference per call:", {k: f"{v:.2e}" for k, v in worst.items()})
