"""GPU tests added in round 5:

* the forward's hand-over of its per-half box tests to the backward (the top byte of the sorted id list,
  csrc/render_fwd.hip -> render_bwd.hip): every (entry, 8 x 4 half) pair in which a pixel can blend has its bit set
  (checked per pixel in float64), and the backward produces the SAME BITS with the hand-over as without it (the test
  hook "no_half_masks" makes every wave of the backward test the records again, as before round 5).
"""
import numpy as np
import pytest
import torch

import helpers as Hh
from test_parity_gpu import CASES, _dev, _native_forward, _raw_backward

pytestmark = pytest.mark.gpu


class _option:
    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        from bloomscene_amd import _capi
        self.old = _capi.get_option(self.name)
        _capi.set_option(self.name, self.value)

    def __exit__(self, *exc):
        from bloomscene_amd import _capi
        _capi.set_option(self.name, self.old)


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
    ts = b.tile_start.astype(np.int64)
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
