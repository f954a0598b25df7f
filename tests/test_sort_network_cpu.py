"""Host-side model of the tile sort's round-based network (bloomscene_amd/csrc/binning.hip): the index arithmetic
of every round is replayed in Python to check that (1) the rounds, as scheduled, sort every input (the network is a
correct bitonic sort with +infinity pads), (2) the slot swizzle is a bijection of [0, n2), and (3) under the LDS bank
model of the MI355X guide (ds_read_b64: two groups of 32 lanes, slot mod 32; ds_write_b64: four groups of 16 lanes,
slot mod 16) every access of every round, the load and the read-out is conflict-free, while the unswizzled layout is not.
No GPU, no library call: this pins the design the kernels implement."""
import random

import pytest

M = 3
K = 1 << M
PAD = (1 << 64) - 1


@pytest.fixture(params=[3, 4], autouse=True)
def keys_per_thread(request):
    """M = 3: 8 keys per lane and round (k_sort_tiles_small, the wide classes, k_bucket_sort<1024, 1>); M = 4: 16 keys per
    lane, two tiles side by side in the two 32-lane halves of a wave (k_bucket_sort<512, 2>)."""
    global M, K
    old = M
    M = request.param
    K = 1 << M
    yield M
    M = old
    K = 1 << M


def swz(i):
    return i ^ ((i >> M) & 31)


def local_delta(d, L):
    x = 0
    for b in range(M):
        if L & (1 << b):
            x ^= d[b]
    return x


def rounds(n2):
    """Yields (kind, per-thread function i -> list of K key indices in register order, steps) for one segment of n2
    keys, exactly as lds_sort_rounds / lds_stride_rounds schedule them."""
    size = 2 << M
    while size <= n2:
        k = size.bit_length() - 1
        lo = k - M
        tlow = 1 << lo
        d = [tlow << b for b in range(M - 1)]
        low_all = 0
        for x in d:
            low_all ^= x
        d.append((size - 1) ^ low_all)

        def idx_m(i, lo=lo, k=k, tlow=tlow, d=tuple(d)):
            i0 = ((i >> lo) << k) | (i & (tlow - 1))
            return [i0 ^ local_delta(d, L) for L in range(K)]
        yield ("mirror", idx_m, M)
        rem = k - M
        while rem > 0:
            ns = min(M, rem)
            lo2 = max(rem - M, 0)
            d2 = tuple(1 << (lo2 + b) for b in range(M))

            def idx_s(i, lo2=lo2, d2=d2):
                i0 = ((i >> lo2) << (lo2 + M)) | (i & ((1 << lo2) - 1))
                return [i0 ^ local_delta(d2, L) for L in range(K)]
            yield ("stride", idx_s, ns)
            rem -= ns
        size <<= 1


def cx(e, a, b):
    if e[a] > e[b]:
        e[a], e[b] = e[b], e[a]


def run_registers(kind, e, ns):
    if kind == "mirror":
        H = K >> 1
        for c in range(H):
            cx(e, c, H + (H - 1 - c))
        bits = range(M - 2, -1, -1)
    else:
        bits = range(ns - 1, -1, -1)
    for bit in bits:
        for c in range(K):
            if not c & (1 << bit):
                cx(e, c, c | (1 << bit))


def sort_segment(keys, n2):
    a = list(keys) + [PAD] * (n2 - len(keys))
    for i in range(0, n2, K):          # runs of K sorted on the way in
        a[i:i + K] = sorted(a[i:i + K])
    for kind, idx, ns in rounds(n2):
        seen = set()
        for t in range(n2 >> M):
            ids = idx(t)
            assert not seen & set(ids)  # every key belongs to exactly one thread of the round
            seen |= set(ids)
            e = [a[j] for j in ids]
            run_registers(kind, e, ns)
            for j, v in zip(ids, e):
                a[j] = v
        assert len(seen) == n2
    return a


@pytest.mark.parametrize("n2", [8, 16, 32, 64, 128, 256, 512, 1024, 2048])
def test_rounds_sort_every_input(n2):
    if n2 < K:
        pytest.skip("a segment holds at least one run")
    rng = random.Random(n2)
    for trial in range(6):
        n = n2 if trial == 0 else rng.randint(max(1, n2 // 2 + 1), n2)
        keys = [rng.getrandbits(40) for _ in range(n)]
        if trial == 1:
            keys = sorted(keys, reverse=True)
        if trial == 2:
            keys = [rng.choice(keys[:3]) for _ in range(n)]   # many ties
        out = sort_segment(keys, n2)
        assert out[:n] == sorted(keys)
        assert all(v == PAD for v in out[n:])


@pytest.mark.parametrize("n2", [8, 64, 512, 1024, 4096, 8192])
def test_swizzle_is_a_bijection(n2):
    assert sorted(swz(i) for i in range(n2)) == list(range(n2))


def _read_cycles(slots):
    c = 0
    for g in range(0, 64, 32):
        lanes = [s for s in slots[g:g + 32] if s is not None]
        if lanes:
            banks = {}
            for s in set(lanes):
                banks.setdefault(s % 32, set()).add(s)
            c += max(len(v) for v in banks.values())
    return c


def _write_cycles(slots):
    c = 0
    for g in range(0, 64, 16):
        lanes = [s for s in slots[g:g + 16] if s is not None]
        if lanes:
            banks = {}
            for s in set(lanes):
                banks.setdefault(s % 16, set()).add(s)
            c += max(len(v) for v in banks.values())
    return c


def _conflict_cycles(n2, nt, layout):
    extra = 0
    accesses = []
    for _, idx, _ in rounds(n2):
        accesses.append(idx)
    accesses.append(lambda t: [t * K + j for j in range(K)])           # the load: K consecutive keys per thread
    for idx in accesses:
        for it in range(0, n2 >> M, nt):
            for w0 in range(it, min(it + nt, n2 >> M), 64):
                lanes = list(range(w0, min(w0 + 64, n2 >> M, it + nt)))
                for L in range(K):
                    slots = [layout(idx(t)[L]) for t in lanes] + [None] * (64 - len(lanes))
                    ideal_r = sum(1 for g in range(0, 64, 32) if any(s is not None for s in slots[g:g + 32]))
                    ideal_w = sum(1 for g in range(0, 64, 16) if any(s is not None for s in slots[g:g + 16]))
                    extra += _read_cycles(slots) - ideal_r + _write_cycles(slots) - ideal_w
    for w0 in range(0, n2, 64):                                        # the read-out: consecutive keys
        slots = [layout(i) for i in range(w0, min(w0 + 64, n2))]
        slots += [None] * (64 - len(slots))
        extra += _read_cycles(slots) - sum(1 for g in range(0, 64, 32) if any(s is not None for s in slots[g:g + 32]))
    return extra


@pytest.mark.parametrize("n2,nt", [(64, 64), (512, 64), (1024, 64), (4096, 512), (8192, 1024), (128, 32), (512, 32)])
def test_swizzled_slots_are_bank_conflict_free(n2, nt):
    if (M == 4) != (nt == 32):
        pytest.skip("M = 4 runs with 32 lanes per tile (one half of a wave), M = 3 with whole waves")
    assert _conflict_cycles(n2, nt, swz) == 0
    if n2 >= 512:
        assert _conflict_cycles(n2, nt, lambda i: i) > 0   # the natural layout is not
