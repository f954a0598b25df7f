"""GPU tests added in round 6.

* the two forms of the second binning pass -- the tile-owned chain and k_bucket_sort (csrc/binning.hip: binning_plan) --
  produce the same bits in every output of forward + backward, on frames with short lists, with lists of every sort
  class, with 1 / 2 / 4 parts per bucket; and the product library picks the bucket form on sparse frames, the chain on
  dense ones,
* an overflowed BSR_FLAG_NO_READBACK frame is NaN in EVERY output, the accumulated-opacity extension included
  (csrc/render_fwd.hip: final_T / n_contrib of the saved image state),
* a forward issued during stream capture while the previous no-readback forward's overflow check is pending fails
  with a message naming bsr_check_deferred, instead of making a blocking host wait inside the capture (csrc/api.hip).
"""
import pytest
import torch

import helpers as Hh
from test_parity_gpu import CASES, _dev

pytestmark = pytest.mark.gpu


def _call(c, dev, **kw):
    from bloomscene_amd import GaussianRasterizer
    rast = GaussianRasterizer(Hh.hip_settings(c, dev), **kw)
    t = {k: getattr(c, k).to(dev) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    return rast, t


def test_overflowed_frame_is_nan_in_the_alpha_output_too():
    from bloomscene_amd import rasterizer as RZ
    dev = _dev()
    c = Hh.make_case(**CASES["sh3"])
    st, _ = Hh.run_oracle(c, backward=False)
    rast, t = _call(c, dev, capacity=max(1, int(st.num_rendered) // 8))
    with torch.no_grad():
        color, radii, depth, alpha = rast(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), opacities=t["opacities"],
                                          shs=t["shs"], scales=t["scales"], rotations=t["rotations"], return_alpha=True)
    torch.cuda.synchronize()
    assert torch.isnan(color).all() and torch.isnan(depth).all()
    assert torch.isnan(alpha).all()          # (until round 6: 1 - whatever the uninitialised scratch held)
    with pytest.raises(RuntimeError, match="was not rendered"):
        RZ.check_deferred()
    # a capacity that fits: alpha = 1 - final_T of the default path, bit for bit
    rast2, _ = _call(c, dev, capacity=int(st.num_rendered) + 1)
    rast3, _ = _call(c, dev)
    with torch.no_grad():
        a2 = rast2(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), opacities=t["opacities"], shs=t["shs"],
                   scales=t["scales"], rotations=t["rotations"], return_alpha=True)[3]
        a3 = rast3(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), opacities=t["opacities"], shs=t["shs"],
                   scales=t["scales"], rotations=t["rotations"], return_alpha=True)[3]
    RZ.check_deferred()
    assert torch.equal(a2.view(torch.int32), a3.view(torch.int32)) and not torch.isnan(a2).any()


def test_forward_during_capture_with_a_pending_overflow_check_fails_loudly():
    """The warm-up call of a capacity-mode step leaves its overflow check pending; capturing WITHOUT bsr_check_deferred()
    first must raise (naming the fix), not wait on an event inside the capture."""
    from bloomscene_amd import rasterizer as RZ
    dev = _dev()
    c = Hh.make_case(P=5000, W=160, H=96, deg=1, seed=2)
    st, _ = Hh.run_oracle(c, backward=False)
    rast, t = _call(c, dev, capacity=2 * int(st.num_rendered) + 4096)

    def fwd():
        with torch.no_grad():
            return rast(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), opacities=t["opacities"], shs=t["shs"],
                        scales=t["scales"], rotations=t["rotations"])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ref = fwd()                      # warm-up: allocator, pinned buffer, events; leaves the check pending
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="bsr_check_deferred"):
        with torch.cuda.graph(g):
            fwd()
    torch.cuda.synchronize()
    # the documented order works: check, then capture, then replay
    RZ.check_deferred()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        out = fwd()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0].view(torch.int32), ref[0].view(torch.int32)) and torch.equal(out[1], ref[1])


def test_both_forms_of_the_second_binning_pass_give_the_same_bits():
    """libbsr_chain_only.so pins the tile-owned chain, libbsr_bucket_always.so k_bucket_sort with 512- / 1024-key areas,
    libbsr_bucket_big.so k_bucket_sort<2048, 1> (wherever they are eligible: up to 8192 tiles); the product library
    chooses by size.  Every output and gradient of eleven frames (two of them with the depths piled on two or three
    values, so that every form's compare-exchange network sorts them): identical
    digests.  (The order of the tile segments inside point_list differs between the forms; nothing visible may.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "gradient_digest.py")

    def digest(extra):
        r = subprocess.run([sys.executable, tool] + extra + ["--plans"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    libs = {}
    for name in ("chain_only", "bucket_always", "bucket_big"):
        path = os.path.join(root, "bloomscene_amd", f"libbsr_{name}.so")
        assert os.path.exists(path), f"{path} missing: run __graft_entry__.build()"
        libs[name] = digest(["--lib", path])
    product = digest([])
    assert len(product) == 11 and product.keys() == libs["chain_only"].keys() == libs["bucket_always"].keys() == libs["bucket_big"].keys()
    for case in product:
        assert libs["chain_only"][case] == libs["bucket_always"][case], case
        assert libs["chain_only"][case] == libs["bucket_big"][case], case
        assert product[case] == libs["chain_only"][case], case


@pytest.mark.parametrize("name,bucket_form", [("sh3", True), ("c2_100k_800x800", True), ("cluster_lists_mixed", True),
                                               ("lists_gt_1024", False), ("lists_gt_8192", False)])
def test_product_library_picks_the_binning_form_by_size(name, bucket_form):
    """Sparse frames (at most 850 kept instances per tile on average, <= 8192 tiles) take k_bucket_sort: their tile
    segments lie bucket by bucket (low tile byte), inside a bucket part by part; dense frames take the chain: segments
    in tile order.  Either way the lists are the oracle's (checked by _assert_forward_bit_exact)."""
    import numpy as np
    from test_parity_gpu import _assert_forward_bit_exact
    from bloomscene_amd import numerics
    c = Hh.make_case(**CASES[name])
    st, _ = Hh.run_oracle(c, backward=False)
    with numerics(exact_exp=True):
        rs, t, R, radii, gb, bb, ib = _assert_forward_bit_exact(c, st)
    b = Hh.decode_buffers(c.P, c.W, c.H, R, gb, bb, ib)
    T = len(b.tile_lo)
    nz = np.nonzero(b.tile_count)[0]
    tile_order = bool((np.diff(b.tile_lo[nz]) > 0).all())
    # bucket form: all segments of bucket d lie before those of bucket d + 1
    by_bucket = nz[np.argsort(nz & 255, kind="stable")]
    first_of_bucket = {}
    last_of_bucket = {}
    for t in by_bucket:
        d = int(t) & 255
        first_of_bucket[d] = min(first_of_bucket.get(d, 1 << 62), int(b.tile_lo[t]))
        last_of_bucket[d] = max(last_of_bucket.get(d, -1), int(b.tile_hi[t]))
    ds = sorted(first_of_bucket)
    bucket_order = all(last_of_bucket[a] <= first_of_bucket[c] for a, c in zip(ds[:-1], ds[1:]))
    assert b.kept <= 850 * T if bucket_form else b.kept > 850 * T
    if bucket_form:
        assert bucket_order and (T <= 256 or not tile_order)
    else:
        assert tile_order


# ------------------------------------------------------------------ the product's pure device functions, as compiled
def _pure():
    """tests/native/libbsr_pure_functions.so: box_may_hit, the element packing and pooled_tile compiled from the
    product's own headers (csrc/common.h, tile_common.h) -- what the python restatements of test_box_test_cpu.py,
    test_bin_elem_cpu.py and test_tile_pool_cpu.py are restatements OF."""
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "libbsr_pure_functions.so")
    assert os.path.exists(path), f"{path} missing: run __graft_entry__.build()"
    return C.CDLL(path)


def _stream():
    import ctypes as C
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("ext,exty", [(15, 15), (7, 7), (7, 3)])
def test_compiled_box_test_never_drops_a_pair_that_can_blend(ext, exty):
    """box_may_hit as the kernels execute it (fp32, v_med3_f32 clamps) against float64: a (splat, box) pair it drops
    reaches `power >= cut` at NO pixel centre of the box; it agrees with the float64 restatement of
    test_box_test_cpu.py wherever that one is not within its slack of the cut; and it does drop a good share."""
    import ctypes as C
    import numpy as np
    from test_box_test_cpu import _edges
    L = _pure()
    dev = _dev()
    rng = np.random.default_rng(ext * 31 + exty)
    n = 400_000
    X, Y = rng.uniform(-40, 60, n), rng.uniform(-40, 60, n)
    s1, s2, th = 10 ** rng.uniform(-0.5, 1.3, n), 10 ** rng.uniform(-0.5, 1.3, n), rng.uniform(0, np.pi, n)
    ca, sa = np.cos(th), np.sin(th)
    a = ca * ca / s1 ** 2 + sa * sa / s2 ** 2
    c = sa * sa / s1 ** 2 + ca * ca / s2 ** 2
    b = ca * sa * (1 / s1 ** 2 - 1 / s2 ** 2)
    cut = -np.log(255.0 * rng.uniform(0.005, 1.0, n)) - 1e-3          # power_cut of k_preprocess
    bx, by = rng.integers(0, 4, n) * 16.0, rng.integers(0, 4, n) * 16.0
    f = [torch.from_numpy(v.astype(np.float32)).to(dev) for v in (X, Y, a, b, c, cut, bx, by)]
    out = torch.zeros(n, dtype=torch.uint8, device=dev)
    rc = L.pt_box_may_hit(ext, exty, n, *[C.c_void_p(t.data_ptr()) for t in f], C.c_void_p(out.data_ptr()), _stream())
    torch.cuda.synchronize()
    assert rc == 0
    kept = out.cpu().numpy().astype(bool)
    # the fp32 inputs, in float64
    Xf, Yf, af, bf, cf, cutf, bxf, byf = [t.cpu().numpy().astype(np.float64) for t in f]
    best = np.full(n, -np.inf)
    for px in range(ext + 1):
        for py in range(exty + 1):
            dx, dy = Xf - (bxf + px), Yf - (byf + py)
            best = np.maximum(best, -(0.5 * (af * dx * dx + cf * dy * dy) + bf * dx * dy))
    assert not (~kept & (best >= cutf)).any()                          # never drops a pair some pixel can blend
    pd = (af > 0) & (cf > 0) & (af * cf - bf * bf > 0)
    _, _, qmin = _edges(Xf, Yf, af, bf, cf, bxf, byf, float(ext), float(exty))
    slack = 1.0e-3 + 1.0e-4 * np.abs(cutf)
    clear_miss = pd & (-qmin < cutf - 3 * slack)
    clear_hit = ~pd | (-qmin > cutf + slack)
    assert not kept[clear_miss].any() and kept[clear_hit].all()
    assert 0.2 < (~kept).mean() < 0.98


@pytest.mark.parametrize("compact", [0, 1])
def test_compiled_element_packing_round_trips(compact):
    """store_elem_m -> load_elem_m / elem_tile_m / elem_key_m on the device, both element forms: what a reader gets back
    is what test_bin_elem_cpu.py restates (8-byte form: the tile's high byte, the id, the depth bits)."""
    import ctypes as C
    import numpy as np
    L = _pure()
    dev = _dev()
    rng = np.random.default_rng(compact)
    n = 100_000
    tile = rng.integers(0, 1 << (16 if compact else 30), n, dtype=np.uint64)
    gid = rng.integers(0, 1 << (24 if compact else 32), n, dtype=np.uint64)
    depth = rng.integers(0, 1 << 32, n, dtype=np.uint64)
    tile[:4] = [0, 65535, 255, 256]
    gid[:4] = [0, (1 << 24) - 1, 1, (1 << 24) - 2]
    ins = [torch.from_numpy(v.astype(np.uint32).view(np.int32)).to(dev) for v in (tile, gid, depth)]
    scratch = torch.zeros(3 * n, dtype=torch.int32, device=dev)
    o = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4)]
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    rc = L.pt_elem_roundtrip(n, compact, *[C.c_void_p(t.data_ptr()) for t in ins], C.c_void_p(scratch.data_ptr()),
                             *[C.c_void_p(t.data_ptr()) for t in o], C.c_void_p(key.data_ptr()), _stream())
    torch.cuda.synchronize()
    assert rc == 0
    o_tile, o_id, o_depth, o_tile_only = [t.cpu().numpy().view(np.uint32).astype(np.uint64) for t in o]
    want_tile = (tile & 0xff00) if compact else tile
    assert np.array_equal(o_tile, want_tile) and np.array_equal(o_tile_only, want_tile)
    assert np.array_equal(o_id, gid) and np.array_equal(o_depth, depth)
    assert np.array_equal(key.cpu().numpy().view(np.uint64), (depth << np.uint64(32)) | gid)


@pytest.mark.parametrize("n_tiles", [1, 9, 1024, 2500, 4095, 4096, 8160, 8167, 65536])
def test_compiled_pooled_tile_hands_out_every_tile_once(n_tiles):
    """pooled_tile / pooled_grid as compiled: a launch of pooled_grid(n_tiles) workgroups renders every tile exactly
    once, the grid and the pool size are the ones test_tile_pool_cpu.py restates, and the draw that empties the pool
    leaves the counter at zero for the next launch (two launches on one counter)."""
    import ctypes as C
    import numpy as np
    from test_tile_pool_cpu import _assignment
    L = _pure()
    dev = _dev()
    _, grid_want, _, _, k_want = _assignment(n_tiles, 0)
    grid = int(L.pt_pooled_grid(n_tiles))
    assert grid == grid_want and int(L.pt_pool_tiles_per_band(n_tiles)) == k_want
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    for _ in range(2):
        got = torch.full((grid,), -7, dtype=torch.int32, device=dev)
        assert L.pt_pooled_tiles(n_tiles, C.c_void_p(ctr.data_ptr()), C.c_void_p(got.data_ptr()), _stream()) == 0
        torch.cuda.synchronize()
        t = got.cpu().numpy()
        assert (t >= -1).all()
        assert sorted(t[t >= 0].tolist()) == list(range(n_tiles))
        assert int(ctr.item()) == 0


# ------------------------------------------------------------------ the per-tile sort: bucket-and-rank first, the network behind it
SORT_CASES = ["sh3", "lists_gt_1024", "lists_gt_8192", "clustered_84k_list", "cluster_lists_1k_4k", "cluster_lists_mixed",
              "c2_100k_800x800", "huge_splats"]


@pytest.mark.parametrize("name", SORT_CASES)
def test_network_sort_behind_the_rank_sort_gives_the_same_lists(name):
    """By default a tile's segment is first offered to the bucket-and-rank sort (csrc/binning.hip: rank_sort), which
    every other forward test therefore exercises; BSR_FLAG_TEST_SORT_NETWORK sends every segment of the call through the
    compare-exchange network it falls back to (binary64 flavour where the depths allow).  Both must produce the oracle's
    lists -- every size class, both forms of the second binning pass -- and the same point_list words."""
    import numpy as np
    from test_parity_gpu import _assert_forward_bit_exact
    from bloomscene_amd import numerics
    from bloomscene_amd.numerics import FLAG_TEST_SORT_NETWORK, resolve_flags
    c = Hh.make_case(**CASES[name])
    st, _ = Hh.run_oracle(c, backward=False)
    with numerics(exact_exp=True, test_flags=FLAG_TEST_SORT_NETWORK):
        assert resolve_flags() & FLAG_TEST_SORT_NETWORK
        rs, t, R, radii, gb, bb, ib = _assert_forward_bit_exact(c, st)
        net = Hh.decode_buffers(c.P, c.W, c.H, R, gb, bb, ib)
    with numerics(exact_exp=True):
        assert not resolve_flags() & FLAG_TEST_SORT_NETWORK
        rs, t, R, radii, gb, bb, ib = _assert_forward_bit_exact(c, st)
        rank = Hh.decode_buffers(c.P, c.W, c.H, R, gb, bb, ib)
    assert np.array_equal(net.tile_lo, rank.tile_lo) and np.array_equal(net.tile_hi, rank.tile_hi)
    assert np.array_equal(net.point_list, rank.point_list)


@pytest.mark.parametrize("name,levels", [("sh3", 1), ("lists_gt_1024", 2), ("cluster_lists_1k_4k", 3),
                                          ("cluster_lists_mixed", 0), ("c2_100k_800x800", 0)])
def test_depths_piled_on_few_values_fall_back_to_the_network(name, levels):
    """The bucket-and-rank sort declines a segment with more than BSR_RANK_CAP keys in one bucket.  levels > 0: every
    Gaussian sits on one of `levels` depth values (scene A's camera looks down +z from the origin: view depth = z), so
    a tile's keys differ in the id alone and every segment of more than 32 x levels keys is declined; levels = 0: half
    of the Gaussians on one depth value, the others spread -- declined and accepted segments side by side.  The lists
    must be the oracle's either way (ties in depth are ordered by id, as the reference's stable sort leaves them)."""
    from test_parity_gpu import _assert_forward_bit_exact
    from bloomscene_amd import numerics
    c = Hh.make_case(**CASES[name])
    z = c.means3D[:, 2]
    if levels > 0:
        vals = torch.linspace(float(z.median()) * 0.8, float(z.median()) * 1.2, levels)
        c.means3D[:, 2] = vals[torch.arange(c.P) % levels]
    else:
        c.means3D[::2, 2] = float(z.median())
    st, _ = Hh.run_oracle(c, backward=False)
    assert st.num_rendered > 0
    with numerics(exact_exp=True):
        _assert_forward_bit_exact(c, st)
