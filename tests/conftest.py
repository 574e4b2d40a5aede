import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(params=[3, 0], ids=["bf16x3", "policy"])
def arith_bound(request):
    """For tests that hold a module to the reference's outputs at fp32 level (golden vectors written by
    the reference itself): runs the test twice -- with the six-product arithmetic pinned (factor 1.0: the
    bound as written) and under the library's built-in policy (two pieces for forward / gradient
    products, include/s2t_mi355.h s2t_gemm_arith; factor 10.0: that mode's own bound, ~2^-17 per term
    instead of fp32 rounding).  Which kernel serves a small Linear is the plan cache's timed choice, so
    without the pin the tighter bound would hold or not from run to run."""
    from speech2text_amd import _native as N
    assert N.lib().s2t_gemm_arith_set(request.param) == 0
    yield 1.0 if request.param == 3 else 10.0
    N.lib().s2t_gemm_arith_set(0)
