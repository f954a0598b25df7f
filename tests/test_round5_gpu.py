"""GPU tests added in round 5:

* the forward's hand-over of its per-half box tests to the backward (the top byte of the sorted id list,
  csrc/render_fwd.hip -> render_bwd.hip): every (entry, 8 x 4 half) pair in which a pixel can blend has its bit set
  (checked per pixel in float64), and the backward produces the SAME BITS with the hand-over as without it (the test
  flag BSR_FLAG_TEST_NO_HALF_MASKS makes every wave of the backward test the records again, as before round 5).
"""
import numpy as np
import pytest
import torch

import helpers as Hh
from test_parity_gpu import CASES, _dev, _native_forward, _raw_backward

pytestmark = pytest.mark.gpu


def _option(name, value):
    """The test-only per-call flags of the C ABI (include/bloomscene_rast.h BSR_FLAG_TEST_*; until round 6 process-wide
    switches behind bsr_set_option), through the calling thread's numerics context: they change no result."""
    from bloomscene_amd import numerics
    from bloomscene_amd.numerics import FLAG_TEST_NO_HALF_MASKS, FLAG_TEST_SMALL_GRIDS, FLAG_TEST_SORT_INT, resolve_flags
    bit = {"no_half_masks": FLAG_TEST_NO_HALF_MASKS, "sort_small_grids": FLAG_TEST_SMALL_GRIDS,
           "sort_force_int": FLAG_TEST_SORT_INT}[name]
    cur = resolve_flags() & 0xf00
    return numerics(test_flags=(cur | bit) if value else (cur & ~bit))


HANDOVER_CASES = ["sh3", "sh1_near_ragged", "precomp_color", "lists_gt_1024", "clustered_84k_list", "free_camera_sh3",
                  "huge_splats", "c2_100k_800x800", "c1_10k_256x256", "image_8k_130k_tiles"]


@pytest.mark.parametrize("name", HANDOVER_CASES)
def test_backward_is_bit_identical_with_and_without_the_forwards_half_masks(name, exp_mode):
    c = Hh.make_case(**CASES[name])
    rs, t, R, color, depth, radii, gb, bb, ib = _native_forward(c)
    b = Hh.decode_buffers(c.P, c.W, c.H, R, gb, bb, ib)
    out, _ = _raw_backward(c, rs, t, R, radii, gb, bb, ib, c.gC, c.gD)
    with _option("no_half_masks", 1):
        rs2, t2, R2, color2, depth2, radii2, gb2, bb2, ib2 = _native_forward(c)
        b2 = Hh.decode_buffers(c.P, c.W, c.H, R2, gb2, bb2, ib2)
        out2, _ = _raw_backward(c, rs2, t2, R2, radii2, gb2, bb2, ib2, c.gC, c.gD)
    assert R == R2
    assert torch.equal(color.view(torch.int32), color2.view(torch.int32))
    assert torch.equal(depth.view(torch.int32), depth2.view(torch.int32))
    np.testing.assert_array_equal(b.point_list, b2.point_list)
    assert not b2.half_masks.any()                 # hook on: plain ids
    T = ((c.W + 15) // 16) * ((c.H + 15) // 16)
    split = R >= 48 * T                             # (the forward's rule for its split-list instantiation, on the exact size)
    if split and b.kept:
        assert b.half_masks.any(), "the split-list forward left no masks"
    for k in out:
        np.testing.assert_array_equal(out[k].view(np.uint32), out2[k].view(np.uint32), err_msg=k)


@pytest.mark.parametrize("name", ["sh3", "sh1_near_ragged", "lists_gt_1024", "free_camera_sh3", "huge_splats"])
def test_half_masks_cover_every_pixel_that_can_blend(name):
    """Safety of the hand-over: for every entry the forward staged (list positions below the tile's deepest last
    contributor are all staged) and each of the tile's eight 8 x 4 halves, a pixel of the half with power <= 0 and
    alpha >= 1/255 (float64) implies the half's bit.  A missing bit would drop a contribution from the backward."""
    c = Hh.make_case(**CASES[name])
    st, _ = Hh.run_oracle(c, backward=False)
    rs, t, R, color, depth, radii, gb, bb, ib = _native_forward(c)
    b = Hh.decode_buffers(c.P, c.W, c.H, R, gb, bb, ib)
    W, H = c.W, c.H
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    if R < 48 * T:
        pytest.skip("single-list forward: no hand-over")
    ts = b.tile_lo
    n_c = b.n_contrib.reshape(H, W).astype(np.int64)
    xy = st.means2D.astype(np.float64)
    con = st.conic_opacity.astype(np.float64)
    checked = 0
    for tile in range(T):
        ty, tx = divmod(tile, gx)
        blk = (slice(ty * 16, min(H, ty * 16 + 16)), slice(tx * 16, min(W, tx * 16 + 16)))
        deepest = int(n_c[blk].max()) if n_c[blk].size else 0
        ids = b.point_list[ts[tile]:ts[tile] + deepest].astype(np.int64)
        masks = b.half_masks[ts[tile]:ts[tile] + deepest].astype(np.int64)
        if not len(ids):
            continue
        px = tx * 16 + np.arange(16, dtype=np.float64)[None, None, :]
        py = ty * 16 + np.arange(16, dtype=np.float64)[None, :, None]
        dx = xy[ids, 0, None, None] - px
        dy = xy[ids, 1, None, None] - py
        power = -0.5 * (con[ids, 0, None, None] * dx * dx + con[ids, 2, None, None] * dy * dy) - con[ids, 1, None, None] * dx * dy
        alpha = np.minimum(0.99, con[ids, 3, None, None] * np.exp(np.minimum(power, 0.0)))
        live = (power <= 0) & (alpha >= 1.0 / 255.0)          # [n, 16 rows, 16 cols]
        for q in range(4):
            for h in range(2):
                rows = slice((q >> 1) * 8 + 4 * h, (q >> 1) * 8 + 4 * h + 4)
                cols = slice((q & 1) * 8, (q & 1) * 8 + 8)
                need = live[:, rows, cols].any(axis=(1, 2))
                have = ((masks >> (2 * q + h)) & 1).astype(bool)
                assert not (need & ~have).any(), (tile, q, h)
                checked += int(need.sum())
    assert checked > 0


# ------------------------------------------------------------------ BSR_FLAG_NO_READBACK (capacity mode)
def _leaves(c, dev):
    keys = ("means3D", "opacities", "shs", "scales", "rotations")
    return {k: getattr(c, k).to(dev).clone().requires_grad_(True) for k in keys}


def _step(c, dev, capacity=None, leaves=None, gC=None, gD=None, rast=None):
    from bloomscene_amd import GaussianRasterizer
    leaves = leaves or _leaves(c, dev)
    rast = rast or GaussianRasterizer(Hh.hip_settings(c, dev), capacity=capacity)
    m2d = torch.zeros_like(leaves["means3D"], requires_grad=True)
    color, radii, depth = rast(means3D=leaves["means3D"], means2D=m2d, opacities=leaves["opacities"], shs=leaves["shs"],
                               scales=leaves["scales"], rotations=leaves["rotations"])
    for v in leaves.values():
        v.grad = None
    torch.autograd.backward((color, depth), (c.gC.to(dev) if gC is None else gC, c.gD.to(dev) if gD is None else gD))
    return color, depth, radii, {k: v.grad for k, v in leaves.items()}, m2d.grad


@pytest.mark.parametrize("name", ["sh3", "sh1_near_ragged", "lists_gt_1024", "c2_100k_800x800"])
def test_capacity_mode_renders_the_same_bits_without_waiting(name, exp_mode):
    """GaussianRasterizer(capacity=...) = BSR_FLAG_NO_READBACK: scratch sized by the caller, no host wait; colour, depth,
    radii and every gradient bit-identical to the default path, for a capacity just above the kept count and for a
    generous one."""
    from bloomscene_amd.rasterizer import check_deferred
    dev = _dev()
    c = Hh.make_case(**CASES[name])
    st, _ = Hh.run_oracle(c, backward=False)
    ref = _step(c, dev)
    for cap in (int(st.num_rendered) + 1, 3 * int(st.num_rendered) + 5000):
        got = _step(c, dev, capacity=cap)
        check_deferred()   # (fits: no error)
        for a, b in zip(ref[:3], got[:3]):
            assert torch.equal(a.view(torch.int32) if a.is_floating_point() else a, b.view(torch.int32) if b.is_floating_point() else b)
        for k in ref[3]:
            assert torch.equal(ref[3][k].view(torch.int32), got[3][k].view(torch.int32)), k
        assert torch.equal(ref[4].view(torch.int32), got[4].view(torch.int32))


def test_capacity_overflow_is_never_silent():
    """Too small a capacity: the frame comes back as NaN (not stale, not half rendered), bsr_read_counts says how many
    instances were kept, and the thread's NEXT forward raises naming both numbers -- after which the library works on."""
    from bloomscene_amd import rasterizer as RZ
    dev = _dev()
    c = Hh.make_case(**CASES["sh3"])
    rs, t, R, color, depth, radii, gb, bb, ib = _native_forward(c)
    kept, R2 = RZ.read_counts(ib, c.H, c.W)
    assert R2 == R and 0 < kept <= R
    cap = max(1, kept // 2)
    e = torch.Tensor([])
    args = (rs.bg, t["means3D"], e, t["opac"], t["scales"], t["rot"], rs.scale_modifier, e, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, c.H, c.W, t["shs"], c.deg, rs.campos, False, False)
    Rc, color2, depth2, radii2, gb2, bb2, ib2 = RZ._rasterize_gaussians_native(*args, capacity=cap)
    torch.cuda.synchronize()
    assert Rc == cap
    assert torch.isnan(color2).all() and torch.isnan(depth2).all()
    assert torch.equal(radii2, radii)                       # (preprocess ran: radii are those of the frame)
    assert RZ.read_counts(ib2, c.H, c.W) == (kept, R)
    # a caller that has not looked yet runs the backward on the overflowed frame: no list, no slab exists -- the kernels
    # must not touch them (they would read uninitialised ranges); every Gaussian is treated as culled and dL_dmean3D is NaN
    out, _ = _raw_backward(c, rs, t, Rc, radii2, gb2, bb2, ib2, c.gC, c.gD)
    assert np.isnan(out["mean3D"]).all()
    for k in ("mean2D", "conic", "opacity", "cov3D", "sh", "scale", "rot"):
        assert not out[k].any(), k
    with pytest.raises(RuntimeError, match=f"kept {kept} tile instances but was given a capacity of {cap}"):
        RZ._rasterize_gaussians_native(*args)
    R3, color3, depth3, _, _, _, _ = RZ._rasterize_gaussians_native(*args)   # the error was reported once; back to normal
    assert R3 == R and torch.equal(color3.view(torch.int32), color.view(torch.int32))
    # and through bsr_check_deferred
    RZ._rasterize_gaussians_native(*args, capacity=cap)
    with pytest.raises(RuntimeError, match="was not rendered"):
        RZ.check_deferred()
    RZ.check_deferred()


@pytest.mark.fast_exp
def test_capacity_mode_forward_backward_replays_from_a_hip_graph():
    """A warmed-up forward + backward in capacity mode holds no host wait and no illegal call: captured into a HIP graph
    (torch.cuda.graph) and replayed on new input values, it produces the bits of the eager default path."""
    from bloomscene_amd.rasterizer import check_deferred
    dev = _dev()
    c = Hh.make_case(P=20000, W=320, H=200, deg=3, seed=3, scale_mul=2.0)
    st, _ = Hh.run_oracle(c, backward=False)
    cap = 2 * int(st.num_rendered) + 4096
    from bloomscene_amd import GaussianRasterizer
    leaves = _leaves(c, dev)
    gC, gD = c.gC.to(dev), c.gD.to(dev)
    rast = GaussianRasterizer(Hh.hip_settings(c, dev), capacity=cap)   # (camera tensors uploaded before the capture)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):      # warm-up on a side stream: allocator, pinned buffer, events, code objects
        for _ in range(3):
            _step(c, dev, leaves=leaves, gC=gC, gD=gD, rast=rast)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    check_deferred()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        color, depth, radii, grads, g2d = _step(c, dev, leaves=leaves, gC=gC, gD=gD, rast=rast)
    # new values in the captured input tensors (same shapes): move every Gaussian a little, change the upstream gradients
    with torch.no_grad():
        leaves["means3D"].add_(0.01 * torch.randn_like(leaves["means3D"]))
        gC.mul_(0.5)
    g.replay()
    torch.cuda.synchronize()
    c2 = Hh.make_case(P=20000, W=320, H=200, deg=3, seed=3, scale_mul=2.0)
    c2.means3D = leaves["means3D"].detach().cpu()
    c2.gC = gC.cpu()
    ref = _step(c2, dev)
    assert torch.equal(ref[0].view(torch.int32), color.view(torch.int32))
    assert torch.equal(ref[1].view(torch.int32), depth.view(torch.int32))
    assert torch.equal(ref[2], radii)
    for k in ref[3]:
        assert torch.equal(ref[3][k].view(torch.int32), grads[k].view(torch.int32)), k


# ------------------------------------------------------------------ the fused anchor front end without a host wait
def _anchor_step(sc, dev, leaves, gC, gD, capacity=None, settings=None):
    from bloomscene_amd import views
    names = ("anchor", "grid_scaling", "grid_offsets", "neural_opacity", "color", "scale_rot")
    for v in leaves.values():
        v.grad = None
    res = views.render_neural(sc.camera_dev, *[leaves[k] for k in names], sc.bg_dev, capacity=capacity, settings=settings)
    torch.autograd.backward((res["render"], res["depth"]), (gC, gD))
    return res, {k: v.grad for k, v in leaves.items()}


@pytest.mark.fast_exp
def test_static_shape_anchor_render_equals_the_default_and_replays_from_a_hip_graph():
    """render_anchors(capacity=...) = bsr_anchor_render_forward with BSR_FLAG_NO_READBACK: neither the selection count
    nor num_rendered is read back; every per-Gaussian output has N * K rows, the selected ones first, the rest padding
    no camera sees.  Image, depth and the gradients of the six inputs are the default path's bits; radii / viewspace
    gradients agree on the first S rows and are zero behind them.  The step replays from a HIP graph."""
    from bloomscene_amd import views
    from bloomscene_amd.rasterizer import check_deferred
    from bloomscene_amd.synthetic import anchor_scene, upstream_grads
    dev = _dev()
    N, K, W, H = 20000, 10, 256, 192
    sc = anchor_scene(N, K, W, H, seed=3)
    sc.camera_dev = sc.camera.to(dev)
    sc.bg_dev = torch.tensor([0.1, 0.2, 0.3], device=dev)
    names = ("anchor", "grid_scaling", "grid_offsets", "neural_opacity", "color", "scale_rot")
    leaves = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in names}
    gC, gD = [t.to(dev) for t in upstream_grads(W, H, seed=1)]
    ref, ref_g = _anchor_step(sc, dev, leaves, gC, gD)
    ref_view_grad = ref["viewspace_points"].grad.clone()
    S = int(ref["radii"].numel())
    assert 0 < S < N * K
    cap = 40 * S
    got, got_g = _anchor_step(sc, dev, leaves, gC, gD, capacity=cap)
    check_deferred()
    assert got["radii"].numel() == N * K
    assert torch.equal(got["render"].view(torch.int32), ref["render"].view(torch.int32))
    assert torch.equal(got["depth"].view(torch.int32), ref["depth"].view(torch.int32))
    assert torch.equal(got["radii"][:S], ref["radii"]) and not got["radii"][S:].any()
    assert torch.equal(got["selection_mask"], ref["selection_mask"])
    assert torch.equal(got["viewspace_points"].grad[:S], ref_view_grad) and not got["viewspace_points"].grad[S:].any()
    for k in names:
        assert torch.equal(got_g[k].view(torch.int32), ref_g[k].view(torch.int32)), k
    # ---- replay from a graph, on changed input values.  (No output of an earlier iteration may stay alive: it would keep
    # that iteration's AccumulateGrad nodes, bound to the default stream, in the captured backward -- a torch rule for
    # whole-step capture, not a property of this library.)
    del ref, got, ref_g, got_g, ref_view_grad
    settings = views.make_settings(sc.camera_dev, sc.bg_dev, 1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            _anchor_step(sc, dev, leaves, gC, gD, capacity=cap, settings=settings)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    check_deferred()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        res, grads = _anchor_step(sc, dev, leaves, gC, gD, capacity=cap, settings=settings)
    with torch.no_grad():
        leaves["color"].mul_(0.9)
        leaves["neural_opacity"].add_(0.02 * torch.randn_like(leaves["neural_opacity"]))   # (changes the SELECTION too)
    g.replay()
    torch.cuda.synchronize()
    ref2, ref2_g = _anchor_step(sc, dev, leaves, gC, gD)
    assert int(ref2["radii"].numel()) != S
    assert torch.equal(res["render"].view(torch.int32), ref2["render"].view(torch.int32))
    assert torch.equal(res["depth"].view(torch.int32), ref2["depth"].view(torch.int32))
    for k in names:
        assert torch.equal(grads[k].view(torch.int32), ref2_g[k].view(torch.int32)), k


# ------------------------------------------------------------------ wide sort classes: one workgroup, several tiles
STRIDE_CASES = {
    "lists_gt_1024": CASES["lists_gt_1024"],                                              # 9 tiles of (1024, 4096]
    "lists_4097_8192": dict(P=24000, W=32, H=32, deg=0, seed=12, scale_mul=25.0),       # 4 tiles of (4096, 8192]
    "lists_gt_8192": CASES["lists_gt_8192"],                                              # 4 tiles above 8192
    "clustered_84k_list": CASES["clustered_84k_list"],                                    # all classes in one frame
}


@pytest.mark.parametrize("name", list(STRIDE_CASES))
@pytest.mark.parametrize("force_int", [0, 1])
def test_wide_sort_classes_on_capped_grids_keep_the_reference_order(name, force_int):
    """the wide per-tile sort classes run on capped grids that stride over their work lists; with the product
    caps (2560 / 512 / 512 workgroups) no test frame makes a workgroup sort more than one tile.  The "sort_small_grids"
    hook caps them at 2 / 1 / 1: the lists must still be the oracle's (order-preserving sub-sequences, checked by
    _assert_forward_bit_exact), in both compare-exchange flavours, and the image bit-equal."""
    from test_parity_gpu import _assert_forward_bit_exact
    c = Hh.make_case(**STRIDE_CASES[name])
    st, _ = Hh.run_oracle(c, backward=False)
    T = ((c.W + 15) // 16) * ((c.H + 15) // 16)
    sizes = np.diff(st.ranges, axis=1).reshape(-1) if st.ranges.ndim == 2 else None
    with _option("sort_small_grids", 1), _option("sort_force_int", force_int):
        rs, t, R, radii, gb, bb, ib = _assert_forward_bit_exact(c, st)
    b = Hh.decode_buffers(c.P, c.W, c.H, R, gb, bb, ib)
    n = b.tile_count
    if name == "lists_gt_1024":
        assert ((n > 1024) & (n <= 4096)).sum() > 2
    if name == "lists_4097_8192":
        assert ((n > 4096) & (n <= 8192)).sum() > 1, n
    if name == "lists_gt_8192":
        assert (n > 8192).sum() > 1


def test_backward_without_its_trailing_barrier_gives_the_bits_of_the_build_that_keeps_it():
    """k_render_bwd_t has no barrier behind its per-batch epilogue (render_bwd.hip: waves 1-3 stage the next batch while
    wave 0 adds up this one); the invariants that make this safe are written next to the #ifdef.  The debug build
    `make debug_variants` puts the barrier back: every output of forward + backward on four scenes (with and without the
    depth-gradient instantiation) must be the SAME BITS in both builds -- a later edit that breaks an invariant shows up
    here as a race, not as a tolerance."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variant = os.path.join(root, "bloomscene_amd", "libbsr_trailing_barrier.so")
    assert os.path.exists(variant), "bloomscene_amd/libbsr_trailing_barrier.so missing: run __graft_entry__.build()"
    tool = os.path.join(root, "tools", "gradient_digest.py")

    def digest(extra):
        r = subprocess.run([sys.executable, tool] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    a, b = digest([]), digest(["--lib", variant])
    assert a.keys() == b.keys() and len(a) == 8
    for case in a:
        assert a[case] == b[case], case


def test_tail_pool_of_the_tile_walks_changes_no_bit():
    """Frames of 4096 tiles and more hand the last 64 tiles of every XCD band out through a counter (common.h:
    pooled_tile): which workgroup renders a tile then depends on the XCDs' speeds, the tile's result must not.  Every
    output of forward + backward (with and without the depth-gradient instantiation) of a 1024 x 1024 frame: the same
    bits from the product library, from the same call repeated, and from the debug build without the pool."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variant = os.path.join(root, "bloomscene_amd", "libbsr_no_pool.so")
    assert os.path.exists(variant), "bloomscene_amd/libbsr_no_pool.so missing: run __graft_entry__.build()"
    tool = os.path.join(root, "tools", "gradient_digest.py")

    def digest(extra):
        r = subprocess.run([sys.executable, tool] + extra + ["--pooled"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    a, b = digest([]), digest(["--lib", variant])
    assert a.keys() == b.keys() and len(a) == 4
    for case in a:
        assert a[case] == b[case], case
    assert a["pooled_4096_tiles"] == a["pooled_4096_tiles_again"]
    assert a["pooled_4096_tiles+depth"] == a["pooled_4096_tiles_again+depth"]


def test_second_backward_on_one_forward_state_finds_the_pool_counter_reset():
    """The backward's pool counter is zeroed by the forward's k_scans and by the draw that empties a launch's pool; a
    second and third backward through the same graph (retain_graph) must therefore walk every tile again: the same
    gradients, bit for bit."""
    from bloomscene_amd import GaussianRasterizer
    c = Hh.make_case(P=120000, W=1024, H=1024, deg=1, seed=6, scale_mul=1.5)
    dev = torch.device("cuda")
    leaves = {k: getattr(c, k).to(dev).clone().requires_grad_(True)
              for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    rast = GaussianRasterizer(raster_settings=Hh.hip_settings(c, dev))
    m2d = torch.zeros_like(leaves["means3D"], requires_grad=True)
    color, radii, depth = rast(means3D=leaves["means3D"], means2D=m2d, opacities=leaves["opacities"], shs=leaves["shs"],
                               scales=leaves["scales"], rotations=leaves["rotations"])
    gC, gD = c.gC.to(dev), c.gD.to(dev)
    grads = []
    for _ in range(3):
        for v in list(leaves.values()) + [m2d]:
            v.grad = None
        torch.autograd.backward((color, depth), (gC, gD), retain_graph=True)
        torch.cuda.synchronize()
        g = {k: v.grad.clone() for k, v in leaves.items()}
        g["means2D"] = m2d.grad.clone()
        grads.append(g)
    for g in grads[1:]:
        for k in g:
            assert torch.equal(g[k], grads[0][k]), k
    assert float(grads[0]["means3D"].abs().sum()) > 0
