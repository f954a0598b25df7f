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
                                       "decision bands); every other GPU test runs bsr_set_option('exact_exp', 1)")


@pytest.fixture(autouse=True)
def _exp_mode(request):
    """The forward is compared BIT FOR BIT with the oracle, which holds in the library's exact_exp mode (the pinned exp
    on every evaluation).  Tests marked `fast_exp` run the default mode instead and compare with the tolerances the
    mode documents (include/bloomscene_rast.h: bsr_set_option).  Child processes of a test run the default."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from bloomscene_amd import _capi
    _capi.set_option("exact_exp", 0 if request.node.get_closest_marker("fast_exp") else 1)
    yield
    _capi.set_option("exact_exp", 0)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as O
    O.build()
    return O
