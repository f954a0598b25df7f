"""Per-call numerics of the rasterizer (the ``flags`` word of ``bsr_forward_ex`` / ``bsr_backward_ex``,
``include/bloomscene_rast.h``).

The C library has no numerics state: every native call carries its mode.  On the python side a mode can be given
explicitly (``GaussianRasterizer(..., exact_exp=True, strict_gradients=True)``) or for a block of code through the
``numerics`` context manager, which is THREAD-LOCAL (two host threads can hold different modes) and is read when a
forward call is issued -- its backward runs with the flags recorded on the autograd node, whatever thread executes it.
"""
from __future__ import annotations

import threading

FLAG_EXACT_EXP = 1    # BSR_FLAG_EXACT_EXP: pinned exp on every evaluation of the forward blend (bit-equal to the oracle)
FLAG_EXACT_GRAD = 2   # BSR_FLAG_EXACT_GRAD: the reference's per-pair operations in the backward tile walk
# test-only flags (include/bloomscene_rast.h BSR_FLAG_TEST_*): no result changes, they steer a call through rare code paths
FLAG_TEST_SORT_INT = 0x100
FLAG_TEST_SMALL_GRIDS = 0x200
FLAG_TEST_NO_HALF_MASKS = 0x400
FLAG_TEST_SORT_NETWORK = 0x800
FLAG_TEST_MASK = 0xf00

_tls = threading.local()


def _current():
    return getattr(_tls, "flags", 0)


def resolve_flags(exact_exp=None, strict_gradients=None) -> int:
    """Flags word of a call: explicit arguments win, ``None`` falls back to the calling thread's context."""
    f = _current()
    if exact_exp is not None:
        f = (f | FLAG_EXACT_EXP) if exact_exp else (f & ~FLAG_EXACT_EXP)
    if strict_gradients is not None:
        f = (f | FLAG_EXACT_GRAD) if strict_gradients else (f & ~FLAG_EXACT_GRAD)
    return f


class numerics:
    """``with numerics(exact_exp=True, strict_gradients=True): ...`` -- the default mode of every rasterizer call the
    calling THREAD issues inside the block (nestable; ``None`` keeps the enclosing block's choice)."""

    def __init__(self, exact_exp=None, strict_gradients=None, test_flags=None):
        self._args = (exact_exp, strict_gradients)
        self._test_flags = test_flags   # tests only: FLAG_TEST_* bits, replacing the enclosing block's (None keeps them)
        self._saved = 0

    def __enter__(self):
        self._saved = _current()
        f = resolve_flags(*self._args)
        if self._test_flags is not None:
            f = (f & ~FLAG_TEST_MASK) | (int(self._test_flags) & FLAG_TEST_MASK)
        _tls.flags = f
        return self

    def __exit__(self, *exc):
        _tls.flags = self._saved
        return False
