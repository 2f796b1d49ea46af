import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold():
    return np.load(os.path.join(GOLD, "hotpath.npz"))


@pytest.fixture(scope="session")
def fields():
    from oracle import oracle as orc
    fa = orc.FieldSpec.from_npz(os.path.join(GOLD, "field_A.npz"))
    fb = orc.FieldSpec.from_npz(os.path.join(GOLD, "field_B.npz"), shared=fa)
    # shared nets only: planes of B are its own
    return {"A": fa, "B": fb}


def maxrel(a, b):
    """max-norm error relative to the max-norm of the reference (what `relerr` was until round 3)"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30)) if b.size else 0.0


def rel_l2(a, b):
    """relative L2 error of the whole tensor: sensitive to the many small entries of a sparse gradient that the max-norm ignores"""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)) if b.size else 0.0


def outfrac(a, b, rtol, floor=1e-3):
    """fraction of the elements outside |a - b| <= rtol * (|b| + floor * max|b|): an element-wise relative check with an absolute
    floor of `floor` x the peak (fp32 sums of a different order cannot be relative-exact below that)"""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    if not b.size:
        return 0.0
    return float(np.mean(np.abs(a - b) > rtol * (np.abs(b) + floor * np.max(np.abs(b)))))


GRAD_STATS = []     # (max-norm rel, rel L2, out-of-tolerance fraction at 20 x the bound, size): printed at the end of a GPU run
MIXED_STATS = []    # the same for comparisons of a REDUCED-PRECISION mode with the fp32 golden (a direction check, bound 0.2): reported on
                    # its own line so that it cannot hide a regression of an fp32 comparison in the summary


def relerr_mixed(a, b):
    """relerr for an fp16-input mode against the fp32 reference: recorded under its own label (round-3 verdict)"""
    m, l2 = maxrel(a, b), rel_l2(a, b)
    MIXED_STATS.append((m, l2, outfrac(a, b, 1e-2), int(np.asarray(b).size)))
    return max(m, l2)


def relerr(a, b):
    """The gradient metric of the suite: the LARGER of the max-norm relative error and the relative L2 error (round-2 verdict:
    the max-norm alone leaves everything below ~1e-3 of the peak of a sparse plane gradient unchecked).  Call sites assert it
    under their existing bound; `outfrac` (element-wise, with a floor at 1e-3 of the peak) is recorded alongside and asserted
    by the main gradient tests through `assert_grad`."""
    m, l2 = maxrel(a, b), rel_l2(a, b)
    GRAD_STATS.append((m, l2, outfrac(a, b, 1e-2), int(np.asarray(b).size)))
    return max(m, l2)


OUT_STATS = []     # (fraction outside, size, label) of every assert_grad call


def assert_grad(a, b, tol, label="", out_rtol=None, out_max=None):
    """max-norm and L2 relative error under `tol`, and at most `out_max` of the elements outside an element-wise relative
    tolerance of `out_rtol` (default 20 x tol) with the 1e-3-of-peak floor"""
    e = relerr(a, b)
    assert e < tol, (label, "max(maxrel, rel_l2)", e)
    of = outfrac(a, b, out_rtol if out_rtol is not None else 20 * tol)
    n = int(np.asarray(b).size)
    OUT_STATS.append((of, n, label))
    if out_max is None:
        out_max = max(float(os.environ.get("NVFI_TEST_OUT_MAX", "5e-3")), 4.0 / max(n, 1))    # (a handful of elements of a small tensor)
    assert of <= out_max, (label, "fraction outside the element-wise tolerance", of)
    return e


def pytest_terminal_summary(terminalreporter):
    if GRAD_STATS:
        s = np.array([(m, l, o) for m, l, o, _ in GRAD_STATS])
        terminalreporter.write_line(f"gradient comparisons: {len(GRAD_STATS)} tensors; worst max-norm rel {s[:, 0].max():.2e}, worst rel L2 {s[:, 1].max():.2e}, "
                                    f"worst fraction outside 1e-2 element-wise (floor 1e-3 of the peak) {s[:, 2].max():.2e}")
    if MIXED_STATS:
        s = np.array([(m, l, o) for m, l, o, _ in MIXED_STATS])
        terminalreporter.write_line(f"fp16-mode vs fp32-golden direction checks (separate from the line above): {len(MIXED_STATS)} tensors; worst max-norm rel "
                                    f"{s[:, 0].max():.2e}, worst rel L2 {s[:, 1].max():.2e}")
    if OUT_STATS:
        w = max(OUT_STATS)
        terminalreporter.write_line(f"assert_grad: {len(OUT_STATS)} tensors; worst fraction outside 20 x tol element-wise: {w[0]:.2e} ({w[2]}, {w[1]} elements)")
