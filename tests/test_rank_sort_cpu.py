"""The bucket-and-rank sort of a tile's keys (bloomscene_amd/csrc/binning.hip: rank_sort; the map: common.h:
rank_sort_shift), restated on the host.

Keys are (depth bits << 32 | Gaussian id), unique within a tile.  The kernel deals them into nb buckets by
b = min(int((z - z_min) * scale), nb - 1), scale = (nb - 0.5) / (z_max - z_min) in binary32 (common.h:
rank_sort_bucket), lays the buckets out one behind the other (arrival order inside a bucket: whatever the LDS atomics
give) and places every key at its bucket's start + the number of smaller keys in the bucket.  What this file pins down:
the map is monotone and stays below nb, so the result is the ascending order for ANY arrival order; a segment is
declined exactly when a bucket holds more than BSR_RANK_CAP keys or a depth word is not a positive finite float; the
buckets the product computes are the buckets restated here (tests/native/libbsr_pure_functions.so, a host-side call)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAP = 32


def buckets_of(z, zmin, zmax, nb):
    """common.h: rank_sort_scale / rank_sort_bucket, in binary32 (no contraction: the library is built without it)."""
    z = np.asarray(z, dtype=np.float32)
    zmin, zmax = np.float32(zmin), np.float32(zmax)
    with np.errstate(over="ignore"):
        scale = np.float32(np.float32(nb - 0.5) / np.float32(zmax - zmin)) if zmax > zmin else np.float32(0.0)
    if not scale < np.float32(3.0e38):
        scale = np.float32(0.0)
    prod = (z - zmin).astype(np.float32) * scale
    b = np.minimum(prod.astype(np.float32).astype(np.int64), nb - 1)     # products are >= 0 and finite here
    return b


def rank_sort_model(keys, nblog, rng=None):
    """ids in ascending key order, or None where the kernel declines.  `rng` permutes the arrival order inside the
    scatter (the kernel's LDS atomics hand out slots in no particular order)."""
    keys = np.asarray(keys, dtype=np.uint64)
    n = len(keys)
    nb = 1 << nblog
    depth = (keys >> np.uint64(32)).astype(np.uint32)
    lo, hi = int(depth.min()), int(depth.max())
    if lo == 0 or hi >= 0x7f800000:
        return None                       # not all positive finite floats
    z = depth.view(np.float32)
    b = buckets_of(z, np.uint32(lo).view(np.float32), np.uint32(hi).view(np.float32), nb)
    assert b.min() >= 0 and b.max() < nb
    order_by_depth = np.argsort(depth, kind="stable")
    assert (np.diff(b[order_by_depth]) >= 0).all()      # monotone in the depth word
    cnt = np.bincount(b, minlength=nb)
    if cnt.max() > CAP:
        return None
    first = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    order = np.arange(n) if rng is None else rng.permutation(n)
    cursor = first.copy()
    out = np.zeros(n, dtype=np.uint64)
    for i in order:                       # scatter: bucket start + the number of keys that arrived before
        out[cursor[b[i]]] = keys[i]
        cursor[b[i]] += 1
    ids = np.zeros(n, dtype=np.uint32)
    bo = buckets_of((out >> np.uint64(32)).astype(np.uint32).view(np.float32), np.uint32(lo).view(np.float32),
                    np.uint32(hi).view(np.float32), nb)
    for p in range(n):                    # read-out: start of the bucket + smaller fellow members
        k = out[p]
        beg = first[bo[p]]
        end = first[bo[p] + 1] if bo[p] + 1 < nb else n
        ids[beg + int((out[beg:end] < k).sum())] = np.uint32(k & np.uint64(0xffffffff))
    return ids


def _keys(rng, n, depth_bits):
    ids = rng.permutation(1 << 24)[:n].astype(np.uint64)
    return (depth_bits.astype(np.uint64) << np.uint64(32)) | ids


@pytest.mark.parametrize("n,nblog", [(65, 9), (366, 9), (512, 9), (1024, 9), (1213, 11), (4096, 11)])
def test_model_sorts_spread_depths_in_any_arrival_order(n, nblog):
    rng = np.random.default_rng(n)
    z = rng.uniform(0.3, 40.0, n).astype(np.float32)
    keys = _keys(rng, n, z.view(np.uint32))
    want = (np.sort(keys) & np.uint64(0xffffffff)).astype(np.uint32)
    for trial in range(3):
        got = rank_sort_model(keys, nblog, rng)
        assert got is not None and np.array_equal(got, want)


def test_map_is_monotone_and_bounded_on_every_range():
    rng = np.random.default_rng(1)
    for nb in (512, 1024, 2048, 4096):
        for _ in range(200):
            lo = np.float32(10.0 ** rng.uniform(-3, 4))
            hi = np.float32(lo * (1.0 + 10.0 ** rng.uniform(-7, 3)))
            z = np.sort(rng.uniform(lo, hi, 64).astype(np.float32))
            z[0], z[-1] = lo, hi
            z = np.clip(z, lo, hi)
            b = buckets_of(z, lo, hi, nb)
            assert (np.diff(b) >= 0).all() and b[0] == 0 and 0 <= b.max() < nb
            if hi > lo:
                assert b[-1] >= nb - 2          # the range is used up to its end
    # neighbouring floats, a range of one ulp, a range so small that the scale overflows to infinity
    lo = np.float32(3.0)
    hi = np.nextafter(lo, np.float32(4.0))
    assert list(buckets_of([lo, hi], lo, hi, 512)) in ([0, 510], [0, 511])
    assert list(buckets_of([lo, lo], lo, lo, 512)) == [0, 0]
    tiny_lo = np.float32(1e-38)
    tiny_hi = np.nextafter(tiny_lo, np.float32(1.0))
    assert list(buckets_of([tiny_lo, tiny_hi], tiny_lo, tiny_hi, 512)) == [0, 0]     # the quotient overflows: one bucket


def test_piled_up_depths_are_declined_and_ties_inside_the_cap_are_sorted_by_id():
    rng = np.random.default_rng(2)
    # one depth value for all: a single bucket of n > CAP keys
    keys = _keys(rng, 200, np.full(200, np.float32(3.5).view(np.uint32)))
    assert rank_sort_model(keys, 9) is None
    # ties of up to CAP keys on a few depth values among spread ones: sorted by id inside the tie
    z = rng.uniform(0.3, 40.0, 400).astype(np.float32)
    z[:CAP] = 7.25
    z[100:100 + CAP // 2] = 9.5
    keys = _keys(rng, 400, z.view(np.uint32))
    got = rank_sort_model(keys, 9, rng)
    if got is not None:      # (the tie's bucket may hold a neighbour or two and go over the cap: then declined)
        assert np.array_equal(got, (np.sort(keys) & np.uint64(0xffffffff)).astype(np.uint32))
    z[:CAP + 1] = 7.25
    assert rank_sort_model(_keys(rng, 400, z.view(np.uint32)), 9) is None


def test_source_still_sorts_the_way_this_test_restates_it():
    b = open(os.path.join(ROOT, "bloomscene_amd", "csrc", "binning.hip")).read()
    h = open(os.path.join(ROOT, "bloomscene_amd", "csrc", "common.h")).read()
    assert re.search(r"#define BSR_RANK_CAP %d\b" % CAP, b)
    assert "if (lo == 0u || hi >= 0x7f800000u) return false;" in b
    assert "if (cmax > (uint32_t)BSR_RANK_CAP) return false;" in b
    assert "beg[q] = (uint32_t)first[b];" in b
    assert "const float s = zmax > zmin ? ((float)nb - 0.5f) / (zmax - zmin) : 0.0f;" in h
    assert "return s < 3.0e38f ? s : 0.0f;" in h
    assert "const uint32_t b = (uint32_t)((z - zmin) * scale);" in h


def test_compiled_buckets_equal_the_restated_ones():
    """common.h: rank_sort_scale / rank_sort_bucket as compiled into tests/native/libbsr_pure_functions.so (host code)."""
    import ctypes as C
    path = os.path.join(ROOT, "tests", "native", "libbsr_pure_functions.so")
    if not os.path.exists(path):
        pytest.skip("tests/native/libbsr_pure_functions.so not built (run __graft_entry__.build())")
    try:
        L = C.CDLL(path)
    except OSError as e:                 # (no HIP runtime on this host)
        pytest.skip(str(e))
    L.pt_rank_sort_buckets.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p]
    L.pt_rank_sort_buckets.restype = None
    rng = np.random.default_rng(3)
    for nb in (512, 1024, 2048, 4096):
        for trial in range(40):
            lo = np.float32(10.0 ** rng.uniform(-2, 3))
            hi = np.float32(lo * (1.0 + 10.0 ** rng.uniform(-6, 2)))
            z = np.clip(rng.uniform(lo, hi, 5000).astype(np.float32), lo, hi)
            z[:2] = [lo, hi]
            out = np.zeros(len(z), dtype=np.uint32)
            L.pt_rank_sort_buckets(len(z), z.ctypes.data, C.c_float(float(lo)), C.c_float(float(hi)), nb, out.ctypes.data)
            assert np.array_equal(out.astype(np.int64), buckets_of(z, lo, hi, nb)), (nb, lo, hi)
