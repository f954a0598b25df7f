"""GPU tests added in round 3:

* the library's DEFAULT forward (flags 0: hardware exp outside the decision bands) against the oracle and against its
  own exact mode (BSR_FLAG_EXACT_EXP) -- every discrete result identical, images by SURVEY.md 8(d)'s elementwise metric;
* the backward's slab inside a binning buffer the forward sized from a GUESS (kept <= guess < R);
* opacities <= 0 / NaN: pairs the forward skipped must be skipped by the backward;
* data-parallel training over views: two ranks, two different views, the packed all-reduce must
  leave oracle-gradient(view 0) + oracle-gradient(view 1) on every rank.
"""
import math
import os
import socket

import numpy as np
import pytest
import torch

import helpers as Hh
from test_parity_gpu import CASES, _assert_forward_bit_exact, _dev, _native_forward, _raw_backward

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ default (fast exp) forward
FAST_CASES = ["sh3", "sh1_near_ragged", "precomp_color", "precomp_cov", "shell_view", "lists_gt_1024", "lists_gt_8192",
              "clustered_84k_list", "free_camera_sh3", "huge_splats", "c2_100k_800x800", "c1_10k_256x256", "one_gaussian"]


def _compare_default_with_exact(c, st, label):
    """Runs the forward in both modes.  Asserted: radii, num_rendered, the per-tile lists are IDENTICAL; n_contrib
    differs on at most a 1e-5 share of the pixels (+2) -- the `T (1 - alpha) < 1e-4` stop sees a T that differs by ulps;
    colour / depth / final_T of the default mode within 2e-5 of the image's scale of the oracle everywhere else.
    On a pixel whose stop decision moved, the entry that triggers the stop is blended in one mode and not in the other
    (forward.cu:433-437: the stopping Gaussian is NOT blended): its weight alpha T is below 1e-4 / (1 - alpha) * alpha,
    i.e. up to 1e-2 at the 0.99 clamp -- the reference's own 2-ulp expf moves the same pixels against any other exp,
    and SURVEY.md 8(d) sets aside a 1e-5 share of elements for such predicate flips.  Colour and final_T move by at most
    that weight; depth = D / acc (acc > 0.5) by at most 2 x 1e-2 x the depth of the farthest visible Gaussian, which
    may be many times the largest value of the depth image (splats crowding the near plane)."""
    from bloomscene_amd import numerics
    from bloomscene_amd.numerics import resolve_flags
    assert resolve_flags() == 0                      # (the fast_exp marker: this thread's default IS the product default)
    rs, t, R, color, depth, radii, gb, bb, ib = _native_forward(c)
    b = Hh.decode_buffers(c.P, c.W, c.H, R, gb, bb, ib)
    with numerics(exact_exp=True):
        rs2, t2, R2, color2, depth2, radii2, gb2, bb2, ib2 = _native_forward(c)
        b2 = Hh.decode_buffers(c.P, c.W, c.H, R2, gb2, bb2, ib2)
    # the exact mode is the oracle, bit for bit
    np.testing.assert_array_equal(color2.cpu().numpy().view(np.uint32), st.color.view(np.uint32))
    np.testing.assert_array_equal(depth2.cpu().numpy().view(np.uint32), st.depth.view(np.uint32))
    # discrete results
    assert R == R2 == st.num_rendered
    np.testing.assert_array_equal(radii.cpu().numpy(), st.radii)
    np.testing.assert_array_equal(b.tile_lo, b2.tile_lo)
    np.testing.assert_array_equal(b.tile_hi, b2.tile_hi)
    np.testing.assert_array_equal(b.point_list, b2.point_list)
    flips = b.n_contrib != b2.n_contrib
    n_flip = int(flips.sum())
    assert n_flip <= 2 + 1e-5 * flips.size, (label, n_flip, flips.size)
    fl = flips.reshape(c.H, c.W)
    col, dep = color.cpu().numpy(), depth.cpu().numpy()
    # the other decision a pixel's ulps can move: the depth output is D / acc only where acc > 0.5 (else 0), and
    # acc = 1 - final_T + 1e-6 -- a pixel whose final_T sits within ulps of 0.5 has a depth in one mode and 0 in the other
    valid_moved = ((dep[0] == 0.0) != (st.depth[0] == 0.0)) & (np.abs(st.final_T.reshape(c.H, c.W) - 0.5) <= 1e-5)
    assert int(valid_moved.sum()) <= 2 + 1e-5 * flips.size, (label, int(valid_moved.sum()))
    worst = {}
    for name, got, ref in (("color", col, st.color), ("depth", dep, st.depth),
                           ("final_T", b.final_T.reshape(1, c.H, c.W), st.final_T.reshape(1, c.H, c.W))):
        scale = max(float(np.abs(ref).max()), 1e-30)
        err = np.abs(got.astype(np.float64) - ref.astype(np.float64)) / scale
        m = np.broadcast_to(fl | valid_moved if name == "depth" else fl, err.shape)
        worst[name] = float(err[~m].max()) if (~m).any() else 0.0
        assert worst[name] <= 2e-5, (label, name, worst[name])
        if m.any() and name == "depth":
            vis = st.radii > 0
            z_far = max(float(st.depths[vis].max()) if vis.any() else 0.0, scale)
            mm = np.broadcast_to(fl & ~valid_moved, err.shape)
            if mm.any():
                assert float(err[mm].max()) * scale <= 2.2e-2 * z_far, (label, name, float(err[mm].max()), z_far)
        elif m.any():
            assert float(err[m].max()) <= 1.1e-2, (label, name, float(err[m].max()))
    print(f"[fast-exp] {label:24s} stop decisions moved on {n_flip}, depth validity on {int(valid_moved.sum())} of "
          f"{flips.size} pixels; max err / scale: " + " ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    # SURVEY.md 8(d)'s own metric on the default-mode images (elementwise 1e-4 relative, counted outliers <= 1e-5 of the
    # elements): the pixels counted above (a moved stop: 3 colour elements + 1 depth; a moved depth validity: 1) are the
    # only ones allowed beyond the 1e-5 share
    Hh.assert_forward_parity("default", col, dep, st, label=label, allow=3 * n_flip + int(valid_moved.sum()))
    return n_flip + int(valid_moved.sum())


@pytest.mark.fast_exp
@pytest.mark.parametrize("name", FAST_CASES)
def test_default_forward_against_oracle_and_exact_mode(name):
    c = Hh.make_case(**CASES[name])
    st, _ = Hh.run_oracle(c, backward=False)
    _compare_default_with_exact(c, st, name)


# cases of round 3's soaks (22 of ~9 800 default-mode cases) whose colour or depth left the 2e-5 band: a pixel whose
# `T (1 - alpha) < 1e-4` stop lands on the other side with a T that differs by ulps
STOP_MOVED = {
    "soak_big_250k_cov_precomp": dict(P=250000, W=1850, H=645, deg=1, seed=201190006, scale_mul=2.186471765599889,
                                      cov_mode="precomp"),
    "soak_hint_6k_near_plane": dict(P=6000, W=233, H=141, deg=3, seed=96867431, scale_mul=19.390615306525284,
                                    near_fraction=0.6, scale_modifier=1.9, cov_mode="precomp"),
    # second soak of the round (5 more in ~2 900 default-mode cases)
    "soak_mixed_2k_large_splats": dict(P=2000, W=394, H=119, deg=3, seed=613639856, scale_mul=11.726571156016597,
                                       scale_modifier=1.9),
    "soak_mixed_40k": dict(P=40000, W=240, H=190, deg=2, seed=951567977, scale_mul=6.782460180023914, near_fraction=0.1,
                           scale_modifier=1.9),
    # third soak: NOT a stop -- a pixel whose accumulated opacity sits within ulps of the 0.5 below which the depth output
    # is 0 (forward: depth = acc > 0.5 ? D / acc : 0): it has a depth in one mode and none in the other
    "soak3_40k_depth_validity_threshold": dict(P=40000, W=371, H=258, deg=1, seed=893405730, scale_mul=3.087722131808848,
                                          near_fraction=0.6, scene="b", view=16),
}


@pytest.mark.fast_exp
@pytest.mark.parametrize("name", list(STOP_MOVED))
def test_default_forward_where_a_stop_decision_moves(name):
    """The default mode's T differs from the exact mode's by ulps, so the reference's early stop (forward.cu:433-437)
    can fall one entry earlier or later on a pixel whose T (1 - alpha) sits within those ulps of 1e-4: the stopping
    entry is blended or not.  Here it does (found by the soaks: 22 of ~9 800 default-mode cases, one to four pixels each); the
    pixel count and the size of the change are bounded (_compare_default_with_exact), everything else stays inside
    2e-5 of scale."""
    c = Hh.make_case(**STOP_MOVED[name])
    st, _ = Hh.run_oracle(c, backward=False)
    n_flip = _compare_default_with_exact(c, st, name)
    print(f"[stop-moved] {name}: {n_flip} pixel(s) with a moved decision")
    assert n_flip >= 1   # (the seed still exercises the case: it depends on the last bits of the default mode's T chain)


@pytest.mark.fast_exp
@pytest.mark.parametrize("name", ["sh3", "precomp_color", "shell_view", "free_camera_sh3", "lists_gt_1024",
                                  "c2_100k_800x800"])
def test_default_mode_gradients_against_oracle(name):
    """End to end through torch.autograd in the library's default mode: same bars as the exact mode
    (helpers.assert_gradient_parity: norm-wise < 1e-5, elementwise share bounded by the reference's own f32-order
    spread)."""
    c = Hh.make_case(**CASES[name])
    st, g = Hh.run_oracle(c)
    out = Hh.run_hip(c)
    np.testing.assert_array_equal(out.radii, st.radii)
    Hh.assert_gradient_parity(c, st, g, out.grads, label="default:" + name)


@pytest.mark.fast_exp
def test_default_mode_full_size_c3():
    """BASELINE config C3 at full size in the default mode (what bench.py times): discrete results identical to the
    exact mode and the oracle, images within 2e-5 of scale AND by SURVEY 8(d)'s elementwise metric, gradients at the
    exact mode's bars."""
    c = Hh.make_case(P=1_000_000, W=1920, H=1080, deg=3, seed=0)
    st, g = Hh.run_oracle(c)
    _compare_default_with_exact(c, st, "c3")
    out = Hh.run_hip(c)
    Hh.assert_gradient_parity(c, st, g, out.grads, label="default:c3")


# ------------------------------------------------------------------ slab inside a guessed binning buffer
class _CanaryScratch:
    """Stand-in for rasterizer._Scratch whose tensor carries 8 MiB of 0xA5 behind the bytes the library asked for."""
    CANARY = 8 << 20
    made = []

    def __init__(self, device):
        from bloomscene_amd import _capi
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.requests = []
        box = self

        def _resize(_user, nbytes):
            box.requests.append(int(nbytes))
            box.tensor = torch.full((int(nbytes) + _CanaryScratch.CANARY,), 0xA5, dtype=torch.uint8, device=device)
            return box.tensor.data_ptr()

        self.callback = _capi.ALLOC_FN(_resize)
        _CanaryScratch.made.append(self)

    def canary_intact(self):
        if not self.requests:
            return True
        return bool((self.tensor[self.requests[-1]:] == 0xA5).all().item())


def _slab_window_case(monkeypatch, P, W, H, base_scale=1.0):
    """A view of R1 instances, then the same shape with all scales multiplied so that kept2 <= 1.25 R1 + 4096 (the guessed
    buffer holds the kept instances: the guessed tail RUNS) while point_list[R2] plus the slab rows end behind the
    guessed buffer (the forward must notice, ask again and re-run the tail).  Canaries behind every scratch buffer;
    image, depth and gradients must equal, bit for bit, those of a call that sized its buffer exactly.
    Returns (R1, R2, kept2, tile counts of the second view)."""
    from bloomscene_amd import rasterizer as RZ
    base = Hh.make_case(P=P, W=W, H=H, deg=0, seed=0, color_mode="precomp", scale_mul=base_scale)

    def variant(scale_mul):
        c = Hh.make_case(P=P, W=W, H=H, deg=0, seed=0, color_mode="precomp")
        c.scales = (base.scales * scale_mul).contiguous()
        return c

    def reset_hint():   # another shape: the next call of (P, W, H) has no guess
        _native_forward(Hh.make_case(P=1000, W=64, H=64, deg=0, seed=1, color_mode="precomp"))

    def measure(c):
        reset_hint()
        rs, t, R, color, depth, radii, gb, bb, ib = _native_forward(c)
        return R, Hh.decode_buffers(c.P, c.W, c.H, R, gb, bb, ib)

    from bloomscene_amd import _capi
    c1 = variant(1.0)
    R1, b1 = measure(c1)
    cap = R1 + R1 // 4 + 4096
    guess_bytes = int(_capi.lib().bsr_binning_bytes(cap))

    def carve_end(R, kept):   # where the backward's R-based carve ends: point_list[R], then 40 B per kept instance + 16
        return (4 * R + 255) // 256 * 256 + 40 * kept + 16

    # the dangerous window: the kept instances still fit the guess (no re-run for THAT reason) but the R-based carve of
    # the backward ends behind the guessed buffer; with kept / R ~ 0.62 that is R2 ~ 1.54 .. 1.61 x the guess
    chosen = None
    for k in range(60):
        s_mul = 1.4 + 0.02 * k
        R2, b2 = measure(variant(s_mul))
        if b2.kept > cap:
            break
        if carve_end(R2, b2.kept) > guess_bytes:
            chosen = (s_mul, R2, b2.kept, b2.tile_count)
            break
    assert chosen is not None, ("no scale puts the carve behind the guess while the kept instances fit", R1, b1.kept, cap)
    s2, R2, kept2, tile_counts = chosen
    c2 = variant(s2)
    # reference result: exact sizing (no guess)
    reset_hint()
    rs, t, R, color, depth, radii, gb, bb, ib = _native_forward(c2)
    want, _ = _raw_backward(c2, rs, t, R, radii, gb, bb, ib, c2.gC, c2.gD)
    want_color, want_depth = color.clone(), depth.clone()
    # guessed sizing, canaries behind every scratch buffer
    monkeypatch.setattr(RZ, "_Scratch", _CanaryScratch)
    _CanaryScratch.made.clear()
    reset_hint()
    _native_forward(c1)                                   # hint = R1
    _CanaryScratch.made.clear()
    rs, t, R, color, depth, radii, gb, bb, ib = _native_forward(c2)
    geom_s, bin_s, img_s = _CanaryScratch.made[:3]
    assert R == R2
    assert bin_s.requests[0] == guess_bytes               # the first request WAS the guess, too small for the carve ...
    assert len(bin_s.requests) == 2 and bin_s.requests[-1] >= carve_end(R2, kept2), (bin_s.requests, carve_end(R2, kept2))
    #                                                       ... so the forward asked again, for what the backward needs
    assert torch.equal(color.view(torch.int32), want_color.view(torch.int32))
    assert torch.equal(depth.view(torch.int32), want_depth.view(torch.int32))
    got, _ = _raw_backward(c2, rs, t, R, radii, gb, bb, ib, c2.gC, c2.gD)
    for sc in (geom_s, bin_s, img_s):
        assert sc.canary_intact(), "forward/backward wrote behind a scratch buffer"
    for k in want:
        np.testing.assert_array_equal(got[k].view(np.uint32), want[k].view(np.uint32), err_msg=k)
    print(f"[slab] R1 {R1} guess {cap}; scales x{s2}: R2 {R2} kept2 {kept2}; binning requests {bin_s.requests}; tiles > "
          f"1024 / 4096 / 8192 entries: {int((tile_counts > 1024).sum())} / {int((tile_counts > 4096).sum())} / "
          f"{int((tile_counts > 8192).sum())} of {tile_counts.size}")
    return R1, R2, kept2, tile_counts


def test_backward_slab_stays_inside_a_guessed_binning_buffer(monkeypatch, exp_mode):
    """the forward may size the binning buffer from a guess `cap` and keep it when the KEPT instances
    fit; the backward carves it for R = num_rendered (it is not told cap) and writes up to 40 B per kept instance behind
    point_list[R].  With kept <= cap and cap + 512 Ki < R that ran past the buffer.  Here: a 1 M-Gaussian 1080p view."""
    _slab_window_case(monkeypatch, 1_000_000, 1920, 1080)


def test_rerun_of_a_tail_that_ran_does_not_file_the_long_tiles_twice(monkeypatch, exp_mode):
    """the slab-fit re-run repeats a tail that has already RUN to completion; the counters of the
    wide sort classes' work lists (flags[1], [4], [5]) are zeroed only by k_scans, so the second pass appended every long
    tile again -- two workgroups sorting one > 8192-entry tile through the same global scratch at once, and with more
    than T/2 tiles in one class the doubled list spilled into the next class's region (for the last class: past
    big_tiles[3T]).  A dense 256-tile view whose tiles nearly all hold more than 8192 entries, through the same window:
    image, depth and every gradient bit-equal to exact sizing."""
    R1, R2, kept2, tile_counts = _slab_window_case(monkeypatch, 600_000, 256, 256, base_scale=8.0)
    T = tile_counts.size
    assert T == 256 and int((tile_counts > 8192).sum()) > T // 2, int((tile_counts > 8192).sum())


# ------------------------------------------------------------------ opacities the forward never blends
@pytest.mark.parametrize("mode", ["exact", "default"])
def test_non_positive_and_nan_opacity(mode):
    """API-level inputs no sigmoid produces: opacity < 0, == 0 and NaN.  power_cut = -ln(255 o) is NaN (or inf) for
    them; the forward skips a pair with alpha < 1/255 (forward.cu:423-428) -- o < 0 gives alpha < 0 -- and the backward
    must skip the same pairs (outside its decision band it took every candidate as blended)."""
    from bloomscene_amd import numerics
    c = Hh.make_case(P=3000, W=160, H=96, deg=1, seed=31, scale_mul=3.0)
    g = torch.Generator().manual_seed(5)
    idx = torch.randperm(c.P, generator=g)
    c.opacities = c.opacities.clone()
    c.opacities[idx[:300]] = -torch.rand(300, 1, generator=g)
    c.opacities[idx[300:400]] = 0.0
    c.opacities[idx[400:420]] = float("nan")
    st, gr = Hh.run_oracle(c)
    with numerics(exact_exp=mode == "exact"):
        out = Hh.run_hip(c)
    Hh.assert_forward_parity(mode, out.color, out.depth, st, label="nan_opacity", radii=out.radii)
    og = Hh.oracle_grads(c, gr)
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
        ref, got = getattr(og, k), getattr(out.grads, k)
        fin = np.isfinite(ref)
        assert (np.isfinite(got) == fin).all(), k
        assert Hh.max_err_over_scale(got[fin], ref[fin]) < 1e-5, k
    # a Gaussian the forward never blends gets no gradient through the blend
    dead = (c.opacities[:, 0] <= 0).numpy()
    assert not out.grads.means2D[dead].any()


# ------------------------------------------------------------------ §8f-3: gradient all-reduce, values
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _f3_worker(rank, world, port, q):
    """One rank of the data-parallel step: its OWN view of the shared Gaussians, forward + backward on the GPU, then
    views.allreduce_gradients (sum).  Both ranks sit on the one GPU of the box, so the process group is gloo (RCCL
    refuses two ranks on one device); the packing / unpacking under test is backend-independent."""
    import torch.distributed as dist
    import helpers as H2
    from bloomscene_amd import GaussianRasterizer, views
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        c = H2.make_case(P=20000, W=320, H=200, deg=2, seed=77, scene="b", view=3 * rank + 1, scale_mul=4.0)
        names = ("means3D", "opacities", "shs", "scales", "rotations")
        leaves = {k: getattr(c, k).to(dev).clone().requires_grad_(True) for k in names}
        rast = GaussianRasterizer(H2.hip_settings(c, dev))
        means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
        color, radii, depth = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                   shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"])
        torch.autograd.backward((color, depth), (c.gC.to(dev), c.gD.to(dev)))
        ms = views.allreduce_gradients(leaves, average=False)
        torch.cuda.synchronize()
        q.put((rank, ms, {k: v.grad.detach().cpu().numpy() for k, v in leaves.items()}))
    finally:
        dist.destroy_process_group()


def test_allreduced_gradients_equal_the_sum_of_the_per_view_oracle_gradients():
    """SURVEY.md §8(f)-3 / bloomscene.py:232-359 batched over views: after allreduce_gradients(sum) EVERY rank holds
    sum_views dL/dtheta.  Oracle: the per-view gradients of the CPU oracle, added in float64."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_f3_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = None
    for rank in range(world):
        c = Hh.make_case(P=20000, W=320, H=200, deg=2, seed=77, scene="b", view=3 * rank + 1, scale_mul=4.0)
        st, g = Hh.run_oracle(c)
        assert (st.radii > 0).sum() > 500                      # both views see a good part of the shell
        og = Hh.oracle_grads(c, g)
        per = {k: np.asarray(getattr(og, k), dtype=np.float64) for k in ("means3D", "opacities", "shs", "scales",
                                                                         "rotations")}
        want = per if want is None else {k: want[k] + per[k] for k in per}
    for rank, ms, grads in results:
        assert ms > 0.0
        for k, ref in want.items():
            assert Hh.max_err_over_scale(grads[k], ref) < 1e-5, (rank, k)   # the §8(d) norm-wise bound
    for k in want:                                              # and the ranks agree bit for bit
        np.testing.assert_array_equal(results[0][2][k].view(np.uint32), results[1][2][k].view(np.uint32))


# ------------------------------------------------------------------ C4: per-rank visible subsets instead of a broadcast
def _scatter_worker(rank, world, port, q, mode="default"):
    import torch.distributed as dist
    import helpers as H2
    from bloomscene_amd import views, numerics
    numerics(exact_exp=mode == "exact").__enter__()   # (this process renders nothing outside the test)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        P, W, H, deg, V = 60000, 320, 180, 2, 12
        sc = H2.scene_b(P, W, H, deg, n_views=V, seed=3)
        sc.scales = sc.scales * 4.0
        cams = [c.to(dev) for c in sc.cameras]
        names = ("means3D", "scales", "rotations", "opacities", "shs")
        full = {k: getattr(sc, k).to(dev) for k in names}          # every rank holds the full scene here: the control
        bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
        pack = views.CameraPack(cams, dev)           # matrices stacked on the device once
        local, mine, info = views.scatter_visible_gaussians(full if rank == 0 else None, pack, src=0,
                                                            assignment="contiguous", device=dev)
        got = views.render_views_sharded(pack, local, bg, deg, rank=rank, world=world, keep_outputs=True, batch=3,
                                         views=mine)
        ok = sorted(got) == mine and info["counts"][rank] == local["means3D"].shape[0] < P
        for i in mine:
            ref = views.render_view(cams[i], full, bg, deg)
            ok = ok and torch.equal(got[i][0], ref["render"]) and torch.equal(got[i][1], ref["depth"])
            ok = ok and int(ref["visibility_filter"].sum()) > 0
        q.put((rank, bool(ok), info["counts"], mine))
    finally:
        dist.destroy_process_group()


def test_frames_from_scattered_visible_subsets_equal_frames_from_all_gaussians(exp_mode):
    """views.scatter_visible_gaussians (the MI355X-native distribution of the rotate360 sweep): rank 0 runs
    the visibility filter for all cameras, every rank receives only the rows its block of neighbouring views can see
    and renders its views from them -- frames and depths bit-identical to rendering from all Gaussians.  Two ranks on
    the box's one GPU over gloo."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scatter_worker, args=(r, world, port, q, exp_mode)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1] for r in results] == [True, True], results
    assert results[0][2] == results[1][2] and results[0][3] == list(range(0, 6)) and results[1][3] == list(range(6, 12))
    assert all(0 < c < 60000 for c in results[0][2])      # each rank got a strict subset


@pytest.mark.parametrize("shape", ["dense_tiles", "sparse_tiles"])
def test_results_do_not_depend_on_the_order_of_the_gaussians(shape, exp_mode):
    """A size-independent property of the whole pipeline: a pixel blends its Gaussians in depth order and every
    per-Gaussian gradient is summed in tile order, so PERMUTING the input rows changes nothing -- image and depth bit
    for bit, radii and gradients the same rows permuted (both forward modes: which exp a pixel uses on a pair depends on
    that pair alone; a pair of equal depths would be ordered by id and could move a last bit, so the depths are checked
    to be distinct).  Numbering, workgroup composition, histogram
    columns, slab runs, the wave-private staging and the counting sort of small tiles all see different data."""
    kw = dict(P=200_000, W=801, H=601, deg=2, seed=9, scale_mul=2.5) if shape == "dense_tiles" else \
        dict(P=30_000, W=1280, H=720, deg=1, seed=10, scale_mul=1.0, scene="b", view=5)
    c = Hh.make_case(**kw)
    gen = torch.Generator().manual_seed(3)
    for _ in range(8):   # separate Gaussians of equal view depth (200 k float32 depths in [1, 10) collide by birthday)
        st, _ = Hh.run_oracle(c, backward=False)
        d = np.where(st.radii > 0, st.depths, -np.arange(c.P, dtype=np.float32) - 1.0)   # culled ones: distinct dummies
        _, first, counts = np.unique(d, return_index=True, return_counts=True)
        if (counts == 1).all():
            break
        tied = np.setdiff1d(np.arange(c.P), first[counts == 1])
        c.means3D[tied] += (torch.rand(len(tied), 3, generator=gen) - 0.5) * 2.0e-3
    perm = torch.randperm(c.P, generator=gen)
    c2 = Hh.make_case(**kw)
    c2.means3D = c.means3D.clone()
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        setattr(c2, k, getattr(c, k)[perm].contiguous())
    a, b = Hh.run_hip(c), Hh.run_hip(c2)
    vis_depths = st.depths[st.radii > 0]
    assert np.unique(vis_depths).size == vis_depths.size
    np.testing.assert_array_equal(a.color.view(np.uint32), b.color.view(np.uint32))
    np.testing.assert_array_equal(a.depth.view(np.uint32), b.depth.view(np.uint32))
    p = perm.numpy()
    np.testing.assert_array_equal(a.radii[p], b.radii)
    assert 0 < int((a.radii > 0).sum()) < c.P or shape == "dense_tiles"
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
        np.testing.assert_array_equal(getattr(a.grads, k)[p].view(np.uint32), getattr(b.grads, k).view(np.uint32), err_msg=k)


def test_sweep_with_per_batch_compaction_renders_the_same_frames(exp_mode):
    """views.render_views_sharded(batch > 1, compact=True): each batch of views is rendered from the rows its own
    visibility filter kept (one filter pass + one gather for the whole path) -- frames and depths bit-identical to the
    sweep over all Gaussians, with a non-black background, a ragged last batch, and a batch that sees nothing at all."""
    from bloomscene_amd import views
    dev = _dev()
    P, W, H, deg, V = 50000, 320, 180, 2, 11
    sc = Hh.scene_b(P, W, H, deg, n_views=V, seed=4)
    sc.scales = sc.scales * 3.0
    cams = [c.to(dev) for c in sc.cameras]
    g = {k: getattr(sc, k).to(dev) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    g["colors_precomp"] = None
    bg = torch.tensor([0.3, 0.1, 0.6], device=dev)
    pack = views.CameraPack(cams, dev)
    want = views.render_views_sharded(pack, g, bg, deg, rank=0, world=1, keep_outputs=True, batch=4)
    got = views.render_views_sharded(pack, g, bg, deg, rank=0, world=1, keep_outputs=True, batch=4, compact=True)
    assert sorted(got) == sorted(want) == list(range(V))
    for i in range(V):
        assert torch.equal(got[i][0], want[i][0]) and torch.equal(got[i][1], want[i][1]), i
    subsets = views.compact_for_view_groups(pack, {k: v for k, v in g.items() if v is not None}, [[0, 1, 2, 3], [4], []])
    assert 0 < subsets[1]["means3D"].shape[0] < subsets[0]["means3D"].shape[0] < P
    assert subsets[2]["means3D"].shape[0] == 1 and subsets[0]["shs"].shape[1:] == g["shs"].shape[1:]
    # everything far above the yawing cameras: every batch sees nothing and must still come out as the background
    far = dict(g, means3D=g["means3D"] + torch.tensor([0.0, 1.0e4, 0.0], device=dev))
    blank = views.render_views_sharded(pack, far, bg, deg, rank=0, world=1, keep_outputs=True, batch=4, compact=True)
    ref = views.render_views_sharded(pack, far, bg, deg, rank=0, world=1, keep_outputs=True, batch=4)
    for i in range(V):
        assert torch.equal(blank[i][0], ref[i][0]) and torch.equal(blank[i][1], ref[i][1])
    assert torch.equal(blank[0][0], bg.view(3, 1, 1).expand(3, H, W))


@pytest.mark.fast_exp
def test_default_mode_split_and_single_list_kernels_render_the_same_image():
    """Which exp a pixel uses on a pair depends on that pair alone (render_fwd.hip), never on the lanes it shares a wave
    with: the default mode's image must be bit-identical between the split-list instantiation (a view rendered alone:
    >= 48 instances per tile) and the single-list one (the same view inside a batch whose other views are empty, which
    pulls the batch's average below 48).  (View batches vs single calls, guessed vs exact scratch and run-to-run
    reproducibility are covered in both modes by the exp_mode-parametrised tests of test_parity_gpu.py.)"""
    from bloomscene_amd import views as V
    from bloomscene_amd.views import yawed_camera
    dev = _dev()
    c = Hh.make_case(P=60000, W=320, H=200, deg=1, seed=5, scale_mul=3.0)
    g = dict(means3D=c.means3D.to(dev), opacities=c.opacities.to(dev), scales=c.scales.to(dev),
             rotations=c.rotations.to(dev), shs=c.shs.to(dev))
    cams = [yawed_camera(c.W, c.H, c.cam.FoVx, yaw_deg=y).to(dev) for y in [0.0] + [180.0] * 23]   # 24 x 260 tiles
    bg = c.bg.to(dev)
    with torch.no_grad():
        single = V.render_view(cams[0], g, bg, 1)
        color, depth, radii = V.render_views_batched(cams, g, bg, 1)
    n_vis = int((single["radii"] > 0).sum())
    assert n_vis > 20000 and not (radii[1:] > 0).any()   # (~4 instances per visible Gaussian: > 48 per tile alone, < 48 stacked)
    assert torch.equal(single["render"].view(torch.int32), color[0].view(torch.int32))
    assert torch.equal(single["depth"].view(torch.int32), depth[0].view(torch.int32))


def test_group_visibility_equals_any_over_the_per_view_filter():
    """bsr_visible_filter_groups: mask[g] == (visible_filter over the views of group g > 0).any(), for uneven groups, an
    empty group, scale/rotation and cov3D inputs."""
    from bloomscene_amd import views
    dev = _dev()
    P, W, H, V = 50001, 320, 180, 11
    sc = Hh.scene_b(P, W, H, 1, n_views=V, seed=8)
    cams = [c.to(dev) for c in sc.cameras]
    means, scales, rots = sc.means3D.to(dev), (sc.scales * 3.0).to(dev), sc.rotations.to(dev)
    groups = [[0, 1, 2, 3], [4], [], [5, 6, 7, 8, 9, 10]]
    got = views.group_visibility(cams, means, scales, rots, groups)
    per_view = views.prefilter_views(cams, means, scales, rots)           # bool [V, P]
    assert got.shape == (4, P) and got.dtype == torch.bool
    for g, members in enumerate(groups):
        want = per_view[members].any(dim=0) if members else torch.zeros(P, dtype=torch.bool, device=dev)
        assert torch.equal(got[g], want), g
    assert 0 < int(got[0].sum()) < P and not got[2].any()
    from bloomscene_amd.experimental import view_distribution as XD
    rows = XD.visible_rows_per_rank(dict(means3D=means, scales=scales, rotations=rots), cams, worlds=(1, 2))
    assert rows[1] == [int(per_view.any(dim=0).sum())] and len(rows[2]) == 2
    # the counts the kernel accumulates beside the masks
    got2, counts = views.group_visibility(cams, means, scales, rots, groups, return_counts=True)
    assert torch.equal(got2, got) and counts.dtype == torch.int32
    assert counts.tolist() == got.sum(dim=1).tolist()


def test_multi_view_filter_bounding_square_cull_never_changes_a_radius():
    """k_visible_filter_views drops (view, Gaussian) pairs with a conservative bounding-square test before the exact
    one: every row must stay bit-equal to the single-view filter (k_preprocess<true>, no such test) on inputs built to
    sit on the cull's edges -- splats of every size from sub-pixel to larger than the scene, centres far outside, on
    the frustum's borders and around the near plane, un-normalised quaternions, indefinite precomputed covariances,
    NaN / Inf entries, a scaled view matrix, 70 views (more than one queue's worth per lane)."""
    from bloomscene_amd import rasterizer as RZ
    from bloomscene_amd.synthetic import scene_b
    dev = _dev()
    gen = torch.Generator().manual_seed(77)
    P, W, H, V = 30011, 200, 120, 70
    sc = scene_b(P, W, H, 1, n_views=V, seed=12)
    means = sc.means3D.clone()
    spread = torch.exp(torch.rand(P, 1, generator=gen) * 6.0 - 3.0)             # 0.05 .. 20 x the scene's extent
    means = means * spread
    scales = torch.exp(torch.rand(P, 3, generator=gen) * 16.0 - 10.0)            # 4.5e-5 .. 400
    rots = torch.randn(P, 4, generator=gen) * torch.exp(torch.randn(P, 1, generator=gen))
    means[5] = float("nan"); scales[6, 1] = float("inf"); rots[7] = 0.0; means[8, 2] = float("inf")
    cams = sc.cameras
    vms = torch.stack([c.world_view_transform for c in cams]).clone()
    pms = torch.stack([c.full_proj_transform for c in cams]).clone()
    vms[3, :, :3] *= 1.7                                                         # a view matrix that also scales
    tanx, tany = math.tan(cams[0].FoVx * 0.5), math.tan(cams[0].FoVy * 0.5)
    sym = torch.randn(P, 6, generator=gen) * torch.exp(torch.rand(P, 1, generator=gen) * 12.0 - 8.0)   # not PSD
    for mode in ("scale_rot", "cov"):
        kw_s = (scales.to(dev), rots.to(dev), torch.Tensor([])) if mode == "scale_rot" else \
               (torch.Tensor([]), torch.Tensor([]), sym.to(dev))
        for mod in (1.0, 2.5):
            radii = RZ._rasterize_gaussians_filter_views_native(means.to(dev), kw_s[0], kw_s[1], mod, kw_s[2],
                                                                vms.to(dev), pms.to(dev), tanx, tany, H, W, False)
            seen = 0
            for v in range(V):
                single = RZ._rasterize_gaussians_filter_native(means.to(dev), kw_s[0], kw_s[1], mod, kw_s[2],
                                                               vms[v].to(dev), pms[v].to(dev), tanx, tany, H, W, False,
                                                               False)
                assert torch.equal(radii[v], single), (mode, mod, v, int((radii[v] != single).sum()))
                seen += int((single > 0).sum())
            assert 0.01 * P * V < seen < 0.9 * P * V, seen       # neither nothing nor everything visible


def test_pack_rows_equals_index_select_and_cat():
    """bsr_pack_rows: one pass == index_select per tensor + cat, for a strided index column, odd widths, an empty
    selection; a row number outside the tensors packs as zeros."""
    from bloomscene_amd import rasterizer as RZ
    dev = _dev()
    gen = torch.Generator().manual_seed(5)
    P = 7001
    ts = [torch.randn(P, 3, generator=gen), torch.randn(P, 16, 3, generator=gen), torch.randn(P, generator=gen),
          torch.randn(P, 4, generator=gen), torch.randn(P, 1, generator=gen), torch.randn(P, 7, generator=gen)]
    ts = [t.to(dev) for t in ts]
    idx = torch.randperm(P, generator=gen)[:3333].sort().values.to(dev)
    want = torch.cat([t.index_select(0, idx).reshape(idx.numel(), -1) for t in ts], dim=1)
    assert torch.equal(RZ._pack_rows_native(ts, idx), want)
    pairs = torch.stack([torch.zeros_like(idx), idx], dim=1).contiguous()
    assert torch.equal(RZ._pack_rows_native(ts, pairs.reshape(-1)[1:], idx_stride=2, rows=idx.numel()), want)
    assert RZ._pack_rows_native(ts, idx[:0]).shape == (0, want.shape[1])
    bad = torch.tensor([0, P, -1, P - 1], device=dev)
    got = RZ._pack_rows_native(ts[:2], bad)
    assert torch.equal(got[0], want.new_tensor(torch.cat([ts[0][0].reshape(-1), ts[1][0].reshape(-1)]).tolist()))
    assert not got[1].any() and not got[2].any() and torch.equal(got[3, :3], ts[0][P - 1])
    wide = [torch.randn(50, 300, generator=gen).to(dev)]
    sel = torch.tensor([49, 0, 7], device=dev)
    assert torch.equal(RZ._pack_rows_native(wide, sel), wide[0][sel])
    with pytest.raises(RuntimeError, match="1..8"):
        RZ._pack_rows_native(ts + ts, idx)


# ------------------------------------------------------------------ the soak's worst seeds, as named cases
SOAK_SEEDS = {
    # rounds 1-2 of tools/stress_gpu.py: the cases whose gradients missed 1e-5 of scale and were judged by conditioning
    "one_gaussian_depth_gradient": dict(P=1, W=218, H=9, deg=0, seed=807008285, scale_mul=7.509433833011944, scene="a",
                                        free_camera=True, scale_modifier=0.7, color_mode="precomp", M_extra=1,
                                        depth_gradient=True),
    "seven_gaussians_near_plane": dict(P=7, W=289, H=272, deg=3, seed=118639763, scale_mul=18.74694816713805,
                                       near_fraction=0.6, scene="b", free_camera=True, scale_modifier=1.9, view=0,
                                       cov_mode="precomp", M_extra=4),
    "narrow_image_40k": dict(P=40000, W=72, H=205, deg=2, seed=842006941, scale_mul=6.492932378324813, near_fraction=0.1,
                             scene="a", scale_modifier=1.9, color_mode="precomp"),
    "one_small_gaussian": dict(P=1, W=226, H=175, deg=1, seed=347919734, scale_mul=0.33112824823942305, scene="a",
                               free_camera=True, scale_modifier=1.9),
    "big_100k_depth_gradient": dict(P=100000, W=1134, H=786, deg=0, seed=974144424, scale_mul=3.5622006417370153,
                                    scene="a", scale_modifier=1.9, depth_gradient=True),
    "shell_squeezed_depth_gradient": dict(P=6000, W=233, H=141, deg=1, seed=792644185, scale_mul=6.936273668730756,
                                          scene="b", scale_modifier=0.7, view=57, squeeze_xy=0.1, depth_gradient=True),
}


@pytest.mark.parametrize("name", list(SOAK_SEEDS))
def test_ill_conditioned_soak_seeds(name, exp_mode):
    """The random soak (tools/stress_gpu.py) lets a gradient tensor that misses 1e-5 of its scale pass when the miss is
    explained by CONDITIONING: within 8x the change that a 64 eps sum|terms| perturbation of the nine pixel sums
    produces in the oracle's own per-Gaussian chain (single splats whose opacity gradient is a 1 % residue of
    cancelling terms, splats at the near plane).  Those seeds, pinned: forward bit-exact, every gradient either within
    1e-5 of scale or within that bound and, whatever the bound says, within 1e-3 of scale (observed: <= 2.4e-4)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from stress_gpu import chain_sensitivity
    kw = dict(SOAK_SEEDS[name])
    dg = kw.pop("depth_gradient", False)
    c = Hh.make_case(**kw)
    st, g = Hh.run_oracle(c, depth_gradient=dg)
    out = Hh.run_hip(c, depth_gradient=dg)
    Hh.assert_forward_parity(exp_mode, out.color, out.depth, st, label=name, radii=out.radii)
    og = Hh.oracle_grads(c, g)
    sens = None
    for k in Hh.GRAD_KEYS:
        ref, got = getattr(og, k), getattr(out.grads, k)
        if ref is None:
            continue
        assert np.isfinite(got).all(), k
        e = Hh.max_err_over_scale(got, ref)
        if e >= 1e-5:
            sens = sens or chain_sensitivity(st, c, dg)
            assert e <= 1e-5 + 8.0 * sens[k], (k, e, sens[k])
            assert e < 1e-3, (k, e)   # (the conditioning bound alone can be vacuous: these chains amplify by 1e3 and more)
            print(f"[soak-seed] {name}: dL_d{k} {e:.2e} of scale, conditioning bound {8 * sens[k]:.2e}")
