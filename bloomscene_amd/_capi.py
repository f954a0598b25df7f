"""ctypes binding of ``libbloomscene_rast.so`` (C ABI declared in ``include/bloomscene_rast.h`` and
``include/bloomscene_anchors.h``).

The library is built in-tree (``bloomscene_amd/csrc/Makefile``, hipcc --offload-arch=gfx950).
There is deliberately NO fallback: if the shared object is missing or a call fails, this module
raises -- the product path never routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbloomscene_rast.so")
CSRC = os.path.join(_HERE, "csrc")

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

_F = C.c_void_p  # device pointer
ABI_VERSION = 4  # BSR_VERSION of include/bloomscene_rast.h this binding was written against
# use_library(..., allow_older_abi=True): 2 -> 3 only added entry points; 3 -> 4 dropped bsr_set_option / bsr_get_option (not
# bound here any more) and added flag bits -- every entry point bound below has the same signature in all three
OLDER_ABI_ACCEPTED = frozenset({2, 3})


class StageProfile(C.Structure):
    _fields_ = [("name", C.c_char_p), ("total_ms", C.c_double), ("launches", C.c_int)]


# name -> (restype, argtypes); every symbol include/*.h declares
SIGNATURES = {
    "bsr_version": (C.c_int, []),
    "bsr_last_error": (C.c_char_p, []),
    "bsr_mark_visible": (C.c_int, [C.c_int, _F, _F, _F, _F, C.c_void_p]),
    "bsr_forward": (C.c_int, [ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p,
                              C.c_int, C.c_int, C.c_int, _F, C.c_int, C.c_int, _F, _F, _F, _F, _F, C.c_float, _F, _F,
                              _F, _F, _F, C.c_float, C.c_float, C.c_int, _F, _F, _F, C.c_int, C.c_void_p,
                              C.POINTER(C.c_int)]),
    "bsr_forward_ex": (C.c_int, [ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p,
                                 C.c_int, C.c_int, C.c_int, _F, C.c_int, C.c_int, _F, _F, _F, _F, _F, C.c_float, _F, _F,
                                 _F, _F, _F, C.c_float, C.c_float, C.c_int, _F, _F, _F, C.c_int, C.c_void_p,
                                 C.POINTER(C.c_int)] + [C.c_uint]),
    "bsr_forward_views": (C.c_int, [ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, _F, C.c_int, C.c_int, _F, _F, _F, _F, _F,
                                    C.c_float, _F, _F, _F, _F, _F, C.c_float, C.c_float, C.c_int, _F, _F, _F, C.c_int,
                                    C.c_void_p, C.POINTER(C.c_int), C.c_uint]),
    "bsr_visible_filter": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _F, _F, C.c_float, _F, _F, _F, _F,
                                     C.c_float, C.c_float, C.c_int, _F, C.c_int, C.c_void_p]),
    "bsr_visible_scratch_bytes": (C.c_size_t, [C.c_int]),
    "bsr_visible_filter_indices": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _F, _F, C.c_float, _F, _F, _F, _F,
                                             C.c_float, C.c_float, C.c_int, _F, _F, _F, C.POINTER(C.c_int), C.c_int,
                                             C.c_void_p]),
    "bsr_visible_filter_views": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _F, _F, C.c_float, _F, _F, _F, _F,
                                           C.c_float, C.c_float, _F, C.c_int, C.c_void_p]),
    "bsr_visible_groups_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "bsr_visible_filter_groups": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _F, _F, C.c_float, _F, _F, _F, _F,
                                            C.c_float, C.c_float, _F, _F, _F, _F, C.c_int, C.c_void_p]),
    "bsr_pack_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), _F, C.c_int, _F, C.c_int,
                                C.c_void_p]),
    "bsr_gather_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), _F, C.c_int,
                                  C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    "bsr_backward": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _F, C.c_int, C.c_int, _F, _F, _F, _F, C.c_float,
                               _F, _F, _F, _F, _F, C.c_float, C.c_float, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F,
                               _F, _F, _F, _F, C.c_int, C.c_void_p]),
    "bsr_backward_depth": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _F, C.c_int, C.c_int, _F, _F, _F, _F, C.c_float,
                                     _F, _F, _F, _F, _F, C.c_float, C.c_float, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F,
                                     _F, _F, _F, _F, _F, _F, C.c_int, C.c_void_p]),
    "bsr_backward_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _F, C.c_int, C.c_int, _F, _F, _F, _F, C.c_float,
                                  _F, _F, _F, _F, _F, C.c_float, C.c_float, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F,
                                  _F, _F, _F, _F, _F, _F, C.c_int, C.c_void_p, C.c_uint]),
    "bsr_read_counts": (C.c_int, [_F, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bsr_check_deferred": (C.c_int, []),
    "bsr_geometry_bytes": (C.c_size_t, [C.c_int]),
    "bsr_binning_bytes": (C.c_size_t, [C.c_int]),
    "bsr_image_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "bsr_transmittance_offset": (C.c_size_t, [C.c_void_p]),
    "bsr_profile_enable": (C.c_int, [C.c_int]),
    "bsr_profile_only": (C.c_int, [C.c_char_p]),
    "bsr_profile_reset": (C.c_int, []),
    "bsr_profile_read": (C.c_int, [C.POINTER(StageProfile), C.c_int]),
    # include/bloomscene_anchors.h
    "bsr_anchor_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "bsr_anchor_select": (C.c_int, [C.c_int, C.c_int, _F, _F, _F, C.POINTER(C.c_int), C.c_void_p]),
    "bsr_anchor_expand": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_F] * 12 + [C.c_void_p]),
    "bsr_anchor_expand_backward": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_F] * 16 + [C.c_void_p]),
    "bsr_anchor_gaussian_bytes": (C.c_size_t, [C.c_int]),
    "bsr_anchor_gradient_bytes": (C.c_size_t, [C.c_int]),
    "bsr_anchor_render_forward": (C.c_int, [C.c_int, C.c_int] + [_F] * 8 + [ALLOC_FN, C.c_void_p] * 4
                                  + [_F, C.c_int, C.c_int, C.c_float, _F, _F, _F, C.c_float, C.c_float, _F, _F, C.c_int,
                                     C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_uint]),
    "bsr_anchor_render_backward": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int] + [_F] * 6 + [_F, _F, _F]
                                   + [_F, C.c_int, C.c_int, C.c_float, _F, _F, _F, C.c_float, C.c_float]
                                   + [_F] * 3 + [_F] * 5 + [_F] + [_F] * 6 + [C.c_int, C.c_void_p, C.c_uint]),
}

_lib = None
_allow_older_abi = False   # use_library(..., allow_older_abi=True): A/B runs against a library of an earlier round


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    """Load the library; raises (never falls back) if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
                "bloomscene_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        # a stale build of another ABI version would shift arguments silently (several entry points changed their
        # parameter lists in place between versions): refuse it before binding anything
        handle.bsr_version.restype = C.c_int
        # (only 2 -> 3 was additive; a version-1 library bound with these signatures would shift arguments silently)
        older = handle.bsr_version() in OLDER_ABI_ACCEPTED and _allow_older_abi
        if handle.bsr_version() != ABI_VERSION and not older:
            raise RuntimeError(f"{LIB_PATH} is ABI version {handle.bsr_version()}, this binding needs {ABI_VERSION}: "
                               f"rebuild it (make -C {CSRC})")
        for name, (res, args) in SIGNATURES.items():
            if older and not hasattr(handle, name):
                continue   # (an entry point the older library does not have: calling it raises AttributeError)
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def use_library(path: str, allow_older_abi: bool = False):
    """Measurement tools only (A/B runs of two builds on one GPU box, the diagnostic builds of csrc/Makefile): load
    another build of the SAME C ABI instead of the in-tree product library.  Must precede the first native call of
    the process; the product never calls this and reads no environment variable to find its library.
    ``allow_older_abi``: accept a library of an EARLIER ABI version whose entry points kept their parameter lists
    (version 2 -> 3 added symbols and flags only: INTEGRATION.md "ABI versions") -- for A/B runs against an earlier
    round's build; entry points it lacks are simply not bound."""
    global LIB_PATH, _allow_older_abi
    if _lib is not None:
        raise RuntimeError("use_library() must be called before the library is first used")
    LIB_PATH = os.path.abspath(path)
    _allow_older_abi = bool(allow_older_abi)


def last_error() -> str:
    return lib().bsr_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {last_error()}")


def profile_enable(on):
    """False/0: off; True/1: bracket every stage with events; N > 1: only every Nth forward (+ its backward)."""
    lib().bsr_profile_enable(int(on))


def profile_only(stage=None):
    """Bracket only this stage ("render_bwd", ...); None: all stages again."""
    lib().bsr_profile_only(stage.encode() if stage else None)


def profile_reset():
    lib().bsr_profile_reset()


def profile_read():
    """-> {stage name: (total_ms, launches)} accumulated since the last reset."""
    arr = (StageProfile * 16)()
    n = lib().bsr_profile_read(arr, 16)
    return {arr[i].name.decode(): (arr[i].total_ms, arr[i].launches) for i in range(n)}
