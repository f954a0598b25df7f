import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fast_exp: GPU test of the library's DEFAULT forward (hardware exp outside the "
                                       "decision bands); GPU tests without it and without the exp_mode fixture run "
                                       "inside bloomscene_amd.numerics(exact_exp=True)")


@pytest.fixture(autouse=True)
def _exp_mode(request):
    """Numerics are per call (include/bloomscene_rast.h: BSR_FLAG_*); the python host takes a call's mode from the calling
    thread's `bloomscene_amd.numerics(...)` context.  A GPU test runs inside numerics(exact_exp=True) -- the forward is
    then compared BIT FOR BIT with the oracle -- unless it is marked `fast_exp` (the product default, compared with the
    tolerances the mode documents) or takes the `exp_mode` fixture below (both modes, one after the other).  Child
    processes of a test run the default unless they enter a context themselves."""
    if request.node.get_closest_marker("gpu") is None or "exp_mode" in request.fixturenames:
        yield
        return
    from bloomscene_amd import numerics
    with numerics(exact_exp=request.node.get_closest_marker("fast_exp") is None):
        yield


@pytest.fixture(params=["exact", "default"])
def exp_mode(request):
    """Runs the test once per forward mode: "exact" (BSR_FLAG_EXACT_EXP: bit-equal to the oracle) and "default" (what
    bench.py times; SURVEY.md 8(d)'s elementwise metric, helpers.assert_forward_parity)."""
    from bloomscene_amd import numerics
    with numerics(exact_exp=request.param == "exact"):
        yield request.param


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as O
    O.build()
    return O
