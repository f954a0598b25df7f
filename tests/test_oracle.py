"""CPU tests of the oracle itself: golden vectors, pinned exp, invariants that follow from the
reference code, API error behaviour.  No GPU, no HIP library."""
import glob
import math
import os

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

import helpers as Hh
from oracle import oracle as O

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _case_from_fixture(z):
    """Rebuild the oracle call from the inputs stored in a fixture."""
    W, H, deg = int(z["in_scalars"][0]), int(z["in_scalars"][1]), int(z["in_scalars"][2])
    rs = O.make_settings(H, W, z["in_scalars"][3], z["in_scalars"][4], z["in_bg"], z["in_scalars"][5],
                         z["in_viewmatrix"], z["in_projmatrix"], deg, z["in_campos"])

    def opt(k):
        return None if z[k].size == 0 else z[k]
    kw = dict(shs=opt("in_shs"), colors_precomp=opt("in_colors_precomp"), scales=opt("in_scales"),
              rotations=opt("in_rotations"), cov3D_precomp=opt("in_cov3D_precomp"))
    return rs, kw


def test_fixtures_present():
    assert len(GOLDEN) >= 7


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    """The oracle is deterministic (fixed summation order, correctly rounded libm calls only):
    every stored intermediate, output and gradient must be reproduced bit for bit."""
    z = np.load(path)
    rs, kw = _case_from_fixture(z)
    stt = O.forward(rs, z["in_means3D"], z["in_opacities"], **kw)
    g = O.backward(stt, z["in_gC"], z["in_gD"])
    assert stt.num_rendered == int(z["num_rendered"])
    vis = stt.radii > 0
    for name in ("color", "depth", "radii", "final_T", "n_contrib", "tiles_touched", "point_list",
                 "point_list_keys", "ranges"):
        np.testing.assert_array_equal(getattr(stt, name), z[name], err_msg=name)
    for name in ("means2D", "depths", "cov3D", "conic_opacity", "rgb", "clamped"):
        np.testing.assert_array_equal(getattr(stt, name)[vis], z[name][vis], err_msg=name)
    for name in ("dL_dmeans3D", "dL_dmeans2D", "dL_dcolors", "dL_dconic", "dL_dopacity", "dL_dcov3D", "dL_dsh",
                 "dL_dscales", "dL_drotations"):
        np.testing.assert_array_equal(getattr(g, name), z[name], err_msg=name)


@pytest.mark.parametrize("path", GOLDEN[:3], ids=[os.path.basename(p)[:-4] for p in GOLDEN[:3]])
def test_seeded_generators_match_fixture_inputs(path):
    """tests that rebuild cases from seeds (GPU parity tests) see the same inputs as the fixtures."""
    from golden.make_golden import CASES
    z = np.load(path)
    c = Hh.make_case(**CASES[os.path.basename(path)[:-4]])
    np.testing.assert_array_equal(c.means3D.numpy(), z["in_means3D"])
    np.testing.assert_array_equal(c.cam.full_proj_transform.numpy(), z["in_projmatrix"])


def test_pinned_exp_is_within_one_ulp():
    x = np.concatenate([np.linspace(-100, 5, 1000001), -np.logspace(-8, 2, 200001)]).astype(np.float32)
    y = O.expf(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    m = ref > 1e-37   # normal range (denormal results are below every alpha threshold)
    ulp = np.abs(y - ref)[m] / np.spacing(ref[m].astype(np.float32)).astype(np.float64)
    assert ulp.max() <= 1.0
    assert O.expf(np.array([0.0], dtype=np.float32))[0] == 1.0
    assert O.expf(np.array([-200.0], dtype=np.float32))[0] == 0.0
    assert np.isnan(O.expf(np.array([np.nan], dtype=np.float32))[0])


def _invariants(c, stt, g=None):
    W, H = c.W, c.H
    gx, gy = (W + 15) // 16, (H + 15) // 16
    vis = stt.radii > 0
    # radii == 0 <=> culled <=> no tiles
    assert ((stt.tiles_touched > 0) == vis).all()
    assert stt.num_rendered == int(stt.tiles_touched.astype(np.int64).sum())
    # near plane: nothing at z <= 0.2 survives (auxiliary.h:154)
    V = c.cam.world_view_transform.numpy().reshape(4, 4)
    z = c.means3D.numpy() @ V[:3, 2] + V[3, 2]
    assert not vis[z <= 0.2 - 1e-6].any()
    # ranges partition the list, tile-major; keys sorted (tile, depth bits, id) -- stable-sort order
    r = stt.ranges
    nonempty = r[:, 1] > r[:, 0]
    if stt.num_rendered:
        starts = r[nonempty, 0]
        ends = r[nonempty, 1]
        assert starts[0] == 0 and ends[-1] == stt.num_rendered and (starts[1:] == ends[:-1]).all()
        k = stt.point_list_keys
        assert (k[1:] >= k[:-1]).all()
        same = k[1:] == k[:-1]
        assert (stt.point_list[1:][same] > stt.point_list[:-1][same]).all()
        tile_of = (k >> np.uint64(32)).astype(np.int64)
        for t in np.flatnonzero(nonempty)[:50]:
            assert (tile_of[r[t, 0]:r[t, 1]] == t).all()
        dbits = (k & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        assert (dbits == stt.depths[stt.point_list].view(np.uint32)).all()
    # n_contrib <= range length of the pixel's tile; empty tile -> bg colour, depth 0, T = 1
    n = stt.n_contrib.reshape(H, W)
    T = stt.final_T.reshape(H, W)
    for ty in range(gy):
        for tx in range(gx):
            t = ty * gx + tx
            blk = (slice(ty * 16, min(H, ty * 16 + 16)), slice(tx * 16, min(W, tx * 16 + 16)))
            ln = int(r[t, 1]) - int(r[t, 0])
            assert n[blk].max() <= ln
            if ln == 0:
                assert (T[blk] == 1).all() and (stt.depth[0][blk] == 0).all()
                for ch in range(3):
                    assert (stt.color[ch][blk] == c.bg[ch].item()).all()
    assert (T > 0).all() and (T <= 1).all()
    assert (stt.depth >= 0).all()
    if g is not None:   # gradients of culled Gaussians are exactly zero
        for name in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dcov3D",
                     "dL_dcolors", "dL_dsh"):
            a = getattr(g, name)
            assert not a.reshape(a.shape[0], -1)[~vis].any(), name
        assert not g.dL_dmeans2D[:, 2].any()   # z component is never written (backward.cu:574-575)


@pytest.mark.parametrize("kw", [
    dict(P=1500, W=96, H=64, deg=3, seed=11, near_fraction=0.2),
    dict(P=800, W=50, H=37, deg=0, seed=12, color_mode="precomp", scale_mul=5.0),
    dict(P=2000, W=64, H=64, deg=2, seed=13, scene="b", view=17, scale_mul=4.0),
    dict(P=600, W=33, H=17, deg=1, seed=14, cov_mode="precomp", scale_mul=20.0),
])
def test_reference_invariants(kw):
    c = Hh.make_case(**kw)
    stt, g = Hh.run_oracle(c)
    _invariants(c, stt, g)


def test_blend_weights_sum_to_one():
    """colour = sum w_i c_i + T_final * bg with sum w_i + T_final == 1: with c_i == bg == 1 the
    image is 1 everywhere (forward.cu:439-463)."""
    c = Hh.make_case(P=1500, W=64, H=48, deg=0, seed=21, color_mode="precomp", scale_mul=6.0, bg=(1.0, 1.0, 1.0))
    c.colors_precomp = torch.ones_like(c.colors_precomp)
    stt, _ = Hh.run_oracle(c, backward=False)
    assert np.abs(stt.color - 1.0).max() < 5e-6


def test_visible_filter_and_mark_visible_agree_with_forward():
    c = Hh.make_case(P=3000, W=120, H=70, deg=1, seed=31, near_fraction=0.3, scene="b", view=5, scale_mul=3.0)
    rs = Hh.oracle_settings(c)
    stt, _ = Hh.run_oracle(c, backward=False)
    radii = O.visible_filter(rs, c.means3D, c.scales, c.rotations)
    np.testing.assert_array_equal(radii, stt.radii)   # K1 and K2 share the code path, forward.cu:186-251 vs :281-333
    present = O.mark_visible(c.means3D, rs)
    V = c.cam.world_view_transform.numpy().reshape(4, 4)
    m = c.means3D.numpy()
    z = (V[0, 2] * m[:, 0] + V[1, 2] * m[:, 1] + V[2, 2] * m[:, 2] + V[3, 2]).astype(np.float32)
    assert (present[z > 0.21]).all() and not present[z < 0.19].any()
    assert not stt.radii[~present].any()


def test_depth_gradient_is_ignored():
    """dL_ddepth is accepted but unused (backward.cu:457-463,539-554 are commented out)."""
    c = Hh.make_case(P=500, W=48, H=32, deg=1, seed=41, scale_mul=4.0)
    stt, _ = Hh.run_oracle(c, backward=False)
    g1 = O.backward(stt, c.gC, c.gD)
    g2 = O.backward(stt, c.gC, c.gD * 1000.0 + 3.0)
    for name in ("dL_dmeans3D", "dL_dopacity", "dL_dsh", "dL_dscales"):
        np.testing.assert_array_equal(getattr(g1, name), getattr(g2, name))


def test_error_conventions_and_empty_input():
    c = Hh.make_case(P=10, W=32, H=32, deg=0, seed=51)
    rs = Hh.oracle_settings(c)
    with pytest.raises(ValueError):   # rasterize_points.cu:57-59
        O.forward(rs, np.zeros((10, 4), np.float32), c.opacities, shs=c.shs, scales=c.scales, rotations=c.rotations)
    with pytest.raises(Exception, match="SHs or precomputed colors"):   # python wrapper :192-193
        O.forward(rs, c.means3D, c.opacities, shs=c.shs, colors_precomp=np.ones((10, 3), np.float32),
                  scales=c.scales, rotations=c.rotations)
    with pytest.raises(Exception, match="scale/rotation pair"):   # python wrapper :195-196
        O.forward(rs, c.means3D, c.opacities, shs=c.shs, scales=c.scales)
    stt = O.forward(rs, np.zeros((0, 3), np.float32), np.zeros((0, 1), np.float32), shs=np.zeros((0, 1, 3), np.float32),
                    scales=np.zeros((0, 3), np.float32), rotations=np.zeros((0, 4), np.float32))
    assert stt.num_rendered == 0 and not stt.color.any() and not stt.depth.any()   # rasterize_points.cu:68-82
    c2 = Hh.make_case(P=50, W=32, H=32, deg=0, seed=52, near_fraction=0.5)
    with pytest.raises(RuntimeError, match="prefiltered"):   # auxiliary.h:156-160
        O.forward(Hh.oracle_settings(c2, prefiltered=True), c2.means3D, c2.opacities, shs=c2.shs, scales=c2.scales,
                  rotations=c2.rotations)


def test_get_higher_msb():
    L = O.lib()
    for n, want in [(1, 1), (2, 2), (3, 2), (256, 9), (2500, 12), (8160, 13), (8192, 14), (65535, 16)]:
        assert L.bsro_get_higher_msb(n) == want   # rasterizer_impl.cu:35-50; 32+bit = 41/44/45 for C1/C2/C3
    assert 32 + L.bsro_get_higher_msb(256) == 41 and 32 + L.bsro_get_higher_msb(8160) == 45


@settings(max_examples=15, deadline=None)
@given(P=st.integers(1, 400), W=st.integers(1, 70), H=st.integers(1, 70), deg=st.integers(0, 3),
       seed=st.integers(0, 10000), scale_mul=st.sampled_from([0.5, 3.0, 30.0, 300.0]),
       near=st.sampled_from([0.0, 0.5]))
def test_property_random_scenes(P, W, H, deg, seed, scale_mul, near):
    """Ragged sizes (1x1 .. 70x70, not multiples of 16), huge and tiny splats, near-plane culls."""
    c = Hh.make_case(P=P, W=W, H=H, deg=deg, seed=seed, scale_mul=scale_mul, near_fraction=near)
    stt, g = Hh.run_oracle(c)
    assert np.isfinite(stt.color).all() and np.isfinite(stt.depth).all()
    _invariants(c, stt, g)
