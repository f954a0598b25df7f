"""Hand-derived micro-cases for the CPU oracle: one to three Gaussians, expected values written out in closed
form (float64 arithmetic on the formulas of the cited reference lines, no oracle code reused).

They pin what the float64-autograd cross-check (test_oracle_crosscheck.py) cannot see: the places where the
reference backward is NOT the derivative of its forward (0.99 clamp, 1/(denom^2 + 1e-7), dL_dscale without
scale_modifier, clamped t in dL_dtz), the integer / threshold logic (near plane 0.2, 1/255, 1e-4, acc > 0.5 and
its 1e-6 seed, tile rectangles) and the constants of SURVEY.md row A17.  tests/test_oracle_mutations.py breaks
each of them in a copy of bsr_oracle.c and requires these tests (or the cross-check) to fail.

Camera: identity view matrix, camera at the origin looking down +z; projection p_hom = (x / tanx, y / tany, ., z)
stored in the reference's flat layout m[4 j + i] = P[i][j] (CR/auxiliary.h:58-77, utils/graphics.py:57-77).
"""
import math

import numpy as np

from oracle import oracle as O

F = np.float32
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199


def camera(W, H, tanx, tany, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, deg=0, znear=0.01, zfar=100.0):
    view = np.eye(4, dtype=np.float32).reshape(-1)
    proj = np.zeros(16, dtype=np.float32)
    proj[0] = 1.0 / tanx                       # P[0][0]
    proj[5] = 1.0 / tany                       # P[1][1]
    proj[10] = zfar / (zfar - znear)           # P[2][2]
    proj[14] = -(zfar * znear) / (zfar - znear)  # P[2][3]
    proj[11] = 1.0                             # P[3][2]: w = z
    return O.make_settings(H, W, tanx, tany, bg, scale_modifier, view, proj, deg, np.zeros(3, np.float32))


def one_pixel(opacity, color, z=2.0, bg=(0.1, 0.2, 0.3), n=1, zs=None, opacities=None, colors=None):
    """n Gaussians on the optical axis of a 1x1 image: each projects exactly onto pixel centre (0, 0)
    (ndc2Pix(0, 1) = ((0 + 1) * 1 - 1) / 2 = 0, CR/auxiliary.h:41-44), so d = 0, power = 0, G = 1."""
    rs = camera(1, 1, 1.0, 1.0, bg=bg)
    zs = [z] if zs is None else zs
    n = len(zs)
    means = np.array([[0.0, 0.0, zz] for zz in zs], dtype=np.float32)
    op = np.array([[opacity]] * n if opacities is None else [[o] for o in opacities], dtype=np.float32)
    col = np.array([color] * n if colors is None else colors, dtype=np.float32)
    scales = np.full((n, 3), 0.01, dtype=np.float32)
    rots = np.tile(np.array([1, 0, 0, 0], dtype=np.float32), (n, 1))
    return O.forward(rs, means, op, colors_precomp=col, scales=scales, rotations=rots)


def test_alpha_clamp_value_and_its_undifferentiated_backward():
    """forward.cu:426 alpha = min(0.99, o G); backward.cu:513,567,583: dL_dopacity += G dL_dalpha with no special
    case for the clamp.  One Gaussian, o = 1, G = 1: alpha = 0.99f, T_final = 1 - 0.99f.
    dL_dalpha = T sum_ch (c - 0) g + (-T_final / (1 - alpha)) sum_ch bg g with T = T_final / (1 - alpha) = 1
    => dL_dopacity = sum_ch (c_ch - bg_ch) g_ch   (the true derivative under an active clamp would be 0),
       dL_dcolor = alpha T g = 0.99 g."""
    c, bg, g = (0.2, 0.5, 0.9), (0.1, 0.2, 0.3), (1.0, -2.0, 0.5)
    st = one_pixel(1.0, c, bg=bg)
    a = float(F(0.99))
    T = float(F(1.0) - F(0.99))
    want = [c[i] * a + T * bg[i] for i in range(3)]
    np.testing.assert_allclose(st.color.reshape(3), want, rtol=2e-7)
    assert st.n_contrib[0] == 1 and abs(float(st.final_T[0]) - T) < 1e-9
    gr = O.backward(st, np.array(g, dtype=np.float32).reshape(3, 1, 1))
    want_op = sum((c[i] - bg[i]) * g[i] for i in range(3))          # = -0.2
    assert abs(float(gr.dL_dopacity[0, 0]) - want_op) < 2e-6
    np.testing.assert_allclose(gr.dL_dcolors[0], [a * gi for gi in g], rtol=1e-6)
    # the same formula below the clamp (o = 0.5): alpha = 0.5, T_final = 0.5, T = 1
    st = one_pixel(0.5, c, bg=bg)
    gr = O.backward(st, np.array(g, dtype=np.float32).reshape(3, 1, 1))
    assert abs(float(gr.dL_dopacity[0, 0]) - want_op) < 2e-6
    np.testing.assert_allclose(gr.dL_dcolors[0], [0.5 * gi for gi in g], rtol=1e-6)


def test_near_plane_is_exactly_0p2():
    """auxiliary.h:154 `p_view.z <= 0.2f` culls; the next float above is rendered."""
    z_cut = F(0.2)
    for z, visible in ((z_cut, False), (np.nextafter(z_cut, F(1)), True), (F(0.25), True), (F(0.19), False), (F(-1), False)):
        st = one_pixel(0.5, (1, 1, 1), zs=[float(z)])
        assert (st.radii[0] > 0) == visible, float(z)
        assert O.mark_visible(np.array([[0, 0, float(z)]], np.float32), camera(1, 1, 1, 1))[0] == visible


def test_alpha_threshold_is_exactly_1_over_255():
    """forward.cu:427 `alpha < 1.0f / 255.0f` skips: with G = 1, alpha = o."""
    t = F(1.0) / F(255.0)
    bg = (0.1, 0.2, 0.3)
    for o, blended in ((t, True), (np.nextafter(t, F(0)), False), (F(1.0 / 256.0), False)):
        st = one_pixel(float(o), (1.0, 1.0, 1.0), bg=bg)
        assert st.radii[0] > 0
        assert (st.n_contrib[0] == 1) == blended, float(o)
        if not blended:
            np.testing.assert_array_equal(st.color.reshape(3), np.array(bg, np.float32))


def test_depth_target_normalisation_gate_and_seed():
    """forward.cu:387 acc = 1e-6 seed; :444-446 D += z alpha T, acc += alpha T; :464-468 depth = acc > 0.5 ? D / acc : 0."""
    z = 3.0
    for o in (0.8, 0.5, 0.99):
        st = one_pixel(o, (1, 1, 1), zs=[z])
        a = min(float(F(0.99)), float(F(o)))
        want = z * a / (1e-6 + a)
        assert abs(float(st.depth[0, 0, 0]) / want - 1.0) < 2e-7, o      # the 1e-6 seed is 1.25e-6 of the value at o = 0.8
    for o in (0.4999, 0.4, 0.26, 0.1):                                   # acc <= 0.5: constant 0
        st = one_pixel(o, (1, 1, 1), zs=[z])
        assert st.n_contrib[0] == 1 and float(st.depth[0, 0, 0]) == 0.0, o


def test_transmittance_stop_at_1e_minus_4():
    """forward.cu:433-437: if T (1 - alpha) < 1e-4 the pixel is done and THAT Gaussian is not blended.
    Three Gaussians front to back with alpha 0.99, 0.98, 0.99: T = 0.01, then 0.01 * 0.02 = 2e-4 (blended),
    then 2e-4 * 0.01 = 2e-6 < 1e-4: stop; n_contrib = 2."""
    c = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)]
    bg = (0.5, 0.5, 0.5)
    st = one_pixel(None, None, bg=bg, zs=[1.0, 2.0, 3.0], opacities=[1.0, 0.98, 1.0], colors=c)
    a1, a2 = float(F(0.99)), float(F(0.98))
    T1 = float(F(1) - F(0.99))
    T2 = float(F(T1) * (F(1) - F(0.98)))
    want = [a1 * c[0][i] + a2 * T1 * c[1][i] + T2 * bg[i] for i in range(3)]
    assert st.n_contrib[0] == 2
    np.testing.assert_allclose(st.color.reshape(3), want, rtol=3e-7)
    assert abs(float(st.final_T[0]) - T2) < 1e-10
    # front-to-back order is by view depth, not by index: swap the first two Gaussians' depths
    st2 = one_pixel(None, None, bg=bg, zs=[2.0, 1.0, 3.0], opacities=[1.0, 0.98, 1.0], colors=c)
    T1b = float(F(1) - F(0.98))
    want2 = [a2 * c[1][i] + a1 * T1b * c[0][i] + float(F(T1b) * (F(1) - F(0.99))) * bg[i] for i in range(3)]
    np.testing.assert_allclose(st2.color.reshape(3), want2, rtol=3e-7)


def _diag_case(x, z, s, W=64, H=32, tanx=1.0, tany=0.5, mod=1.0, opacity=0.5):
    rs = camera(W, H, tanx, tany, scale_modifier=mod)
    means = np.array([[x, 0.0, z]], dtype=np.float32)
    scales = np.array([s], dtype=np.float32)
    rots = np.array([[1, 0, 0, 0]], dtype=np.float32)
    st = O.forward(rs, means, np.array([[opacity]], np.float32), colors_precomp=np.array([[0.3, 0.6, 0.9]], np.float32),
                   scales=scales, rotations=rots)
    return rs, st


def _cov2d_closed_form(x, z, s, W, H, tanx, tany, mod=1.0):
    """forward.cu:74-113 for an identity view matrix, y = 0 and Sigma = diag((mod s)^2):
    A = J = [[fx/z, 0, -fx tx/z^2], [0, fy/z, 0]] with tx = clamp(x/z, +-1.3 tanx) z; cov = A Sigma A^T + 0.3 I."""
    fx, fy = W / (2.0 * tanx), H / (2.0 * tany)
    txc = min(1.3 * tanx, max(-1.3 * tanx, x / z)) * z
    S = [(mod * si) ** 2 for si in s]
    A00, A02, A11 = fx / z, -fx * txc / (z * z), fy / z
    a = A00 * A00 * S[0] + A02 * A02 * S[2] + 0.3
    c = A11 * A11 * S[1] + 0.3
    return fx, fy, txc, S, A00, A02, A11, a, 0.0, c


def test_cov2d_lowpass_conic_radius_and_tile_rect():
    """forward.cu:110-111 (+0.3), :219-223 conic = (c, -b, a) / det, :229-232 radius = ceil(3 sqrt(lambda_max)),
    auxiliary.h:41-56 pixel centre and tile rectangle."""
    W, H, tanx, tany = 64, 32, 1.0, 0.5
    x, z, s = 0.5, 4.0, (0.5, 0.25, 1e-3)
    rs, st = _diag_case(x, z, s, W, H, tanx, tany)
    fx, fy, txc, S, A00, A02, A11, a, b, c = _cov2d_closed_form(x, z, s, W, H, tanx, tany)
    assert abs(a - (16.0 + 0.3)) < 1e-5 and abs(c - 4.3) < 1e-12      # (32/4)^2 * 0.25 + (tiny A02 term) + 0.3
    np.testing.assert_allclose(st.conic_opacity[0, :3], [1.0 / a, 0.0, 1.0 / c], rtol=3e-7, atol=1e-12)
    assert st.radii[0] == math.ceil(3.0 * math.sqrt(a)) == 13
    # pixel centre: ndc = x / (z tanx) = 0.125 -> ((0.125 + 1) 64 - 1) / 2 = 35.5 ; y: ((0 + 1) 32 - 1) / 2 = 15.5
    np.testing.assert_allclose(st.means2D[0], [35.5, 15.5], rtol=0, atol=1e-5)
    # rect: x in [int((35.5 - 13) / 16), int((35.5 + 13 + 15) / 16)) = [1, 3), y in [int(2.5 / 16), min(2, int(43.5 / 16))) = [0, 2)
    assert st.tiles_touched[0] == 2 * 2 and st.num_rendered == 4
    assert float(st.depths[0]) == z
    # truncation toward zero on the low side: a centre left of the image still starts at tile 0, not -1
    rs, st = _diag_case(-4.2, z, s, W, H, tanx, tany)                   # ndc -1.05 -> pix -2.1 ; radius 13
    assert st.radii[0] > 0 and st.tiles_touched[0] == 1 * 2             # x: [0, int((-2.1 + 13 + 15) / 16) = 1)


def test_cov2d_backward_closed_form_on_axis():
    """backward.cu:197-273 with the hand-set dL_dconic = (g0, g1, -, g2), on the optical axis (A02 = A12 = 0, b = 0):
      denom2inv = 1 / ((a c)^2 + 1e-7)                                     (:203 -- NOT 1 / det^2)
      dL_da = -denom2inv c^2 g0,  dL_dc = -denom2inv a^2 g2,  dL_db = -2 denom2inv a c g1   (:210-212)
      dL_dSigma00 = A00^2 dL_da, dL_dSigma11 = A11^2 dL_dc, dL_dSigma01 = A00 A11 dL_db       (:217-227)
      dL_dtz = -(fx / z^2) 2 A00 Sigma00 dL_da - (fy / z^2) 2 A11 Sigma11 dL_dc,  dL_dtx = dL_dty = 0   (:237-264)
    then backward.cu:322-325 dL_dscale_i = 2 (mod s_i) dL_dSigma_ii  -- no further factor mod -- and, for the
    identity rotation, :333-340 dL_drot = (0, 0, 0, 2 dL_dSigma01 ((mod s0)^2 - (mod s1)^2)).
    The scales are tiny so that det = (0.3 + eps)^2 ~ 0.09: the 1e-7 is 1.2e-5 of denom^2 there."""
    W, H, tanx, tany, mod = 64, 32, 1.0, 0.5, 1.7
    z, s = 2.0, (2e-3, 1e-3, 3e-3)
    rs, st = _diag_case(0.0, z, s, W, H, tanx, tany, mod=mod)
    fx, fy, txc, S, A00, A02, A11, a, b, c = _cov2d_closed_form(0.0, z, s, W, H, tanx, tany, mod)
    g0, g1, g2 = 0.7, -0.4, 1.3
    h = O.empty_grads(st)
    h.dL_dconic[0] = [[g0, g1], [0.0, g2]]
    O.backward_chain(st, h)

    def closed(eps):
        d2i = 1.0 / ((a * c) ** 2 + eps)
        da, dc, db = -d2i * c * c * g0, -d2i * a * a * g2, -2.0 * d2i * a * c * g1
        dS00, dS11, dS01 = A00 * A00 * da, A11 * A11 * dc, A00 * A11 * db
        dtz = -(fx / z ** 2) * 2 * A00 * S[0] * da - (fy / z ** 2) * 2 * A11 * S[1] * dc
        return dS00, dS11, dS01, dtz
    dS00, dS11, dS01, dtz = closed(1e-7)
    got = h.dL_dcov3D[0].astype(np.float64)
    np.testing.assert_allclose(got[[0, 3, 1]], [dS00, dS11, dS01], rtol=2e-6)
    assert not got[[2, 4, 5]].any()
    # the 1e-7 is visible: the same formulas without it are 1.2e-5 away, six times the tolerance above
    assert abs(closed(0.0)[0] / dS00 - 1.0) > 1e-5
    np.testing.assert_allclose(h.dL_dmeans3D[0], [0.0, 0.0, dtz], rtol=3e-6, atol=1e-12)
    want_scale = [2.0 * mod * s[0] * dS00, 2.0 * mod * s[1] * dS11, 0.0]
    np.testing.assert_allclose(h.dL_dscales[0], want_scale, rtol=3e-6, atol=1e-14)
    want_rot_z = 2.0 * dS01 * (S[0] - S[1])
    np.testing.assert_allclose(h.dL_drotations[0], [0.0, 0.0, 0.0, want_rot_z], rtol=3e-6, atol=1e-14)


def test_cov2d_backward_with_active_guard_band_clamp():
    """backward.cu:175-176,262-264: where |tx / tz| exceeds 1.3 tanfovx the forward used the clamped tx, so
    dL_dtx is multiplied by 0 -- while dL_dtz keeps its term in the CLAMPED tx.  Gaussian at x / z = 2 tanx,
    y = 0, Sigma diagonal; dL_dconic hand-set.  With b = 0 (A12 = 0):
      dT00 = 2 A00 S00 da, dT02 = 2 A02 S22 da, dT11 = 2 A11 S11 dc, dT12 = A02 S22 db
      dL_dtx = 0,  dL_dty = -(fy / z^2) dT12,
      dL_dtz = -(fx / z^2) dT00 - (fy / z^2) dT11 + (2 fx txc / z^3) dT02 + 0."""
    W, H, tanx, tany = 32, 32, 1.0, 1.0
    z, s = 2.0, (1.0, 0.5, 0.8)
    x = 2.0 * tanx * z
    rs, st = _diag_case(x, z, s, W, H, tanx, tany)
    assert st.radii[0] > 0                                                  # off-screen centre, rect still touches
    fx, fy, txc, S, A00, A02, A11, a, b, c = _cov2d_closed_form(x, z, s, W, H, tanx, tany)
    assert abs(txc - 1.3 * z) < 1e-12
    np.testing.assert_allclose(st.conic_opacity[0, :3], [1.0 / a, 0.0, 1.0 / c], rtol=1e-6, atol=1e-9)
    g0, g1, g2 = 0.9, 0.6, -1.1
    h = O.empty_grads(st)
    h.dL_dconic[0] = [[g0, g1], [0.0, g2]]
    O.backward_chain(st, h)
    d2i = 1.0 / ((a * c) ** 2 + 1e-7)
    da, dc, db = -d2i * c * c * g0, -d2i * a * a * g2, -2.0 * d2i * a * c * g1
    dT00, dT02, dT11, dT12 = 2 * A00 * S[0] * da, 2 * A02 * S[2] * da, 2 * A11 * S[1] * dc, A02 * S[2] * db
    dty = -(fy / z ** 2) * dT12
    dtz = -(fx / z ** 2) * dT00 - (fy / z ** 2) * dT11 + (2 * fx * txc / z ** 3) * dT02
    np.testing.assert_allclose(h.dL_dmeans3D[0], [0.0, dty, dtz], rtol=5e-6, atol=1e-9)
    # had dL_dtz used the unclamped tx (= x), or dL_dtx not been zeroed, the result would be far away:
    assert abs((-(fx / z ** 2) * dT00 - (fy / z ** 2) * dT11 + (2 * fx * x / z ** 3) * dT02) / dtz - 1.0) > 1e-2
    assert abs(-(fx / z ** 2) * dT02) > 1e-3 * abs(dtz)                       # the dL_dtx that the clamp removes
    # covariance gradient entries that involve A02 (Sigma_22, Sigma_02): :217-227
    np.testing.assert_allclose(h.dL_dcov3D[0][[0, 5, 2]], [A00 * A00 * da, A02 * A02 * da, 2 * A00 * A02 * da], rtol=5e-6)


def test_projection_backward_closed_form():
    """backward.cu:373-387 with dL_dmean2D = (gx, gy) hand-set and nothing else: w = z, m_w = 1 / (z + 1e-7),
    dmean = (gx m_w / tanx, gy m_w / tany, -(x / tanx) m_w^2 gx - (y / tany) m_w^2 gy)."""
    W, H, tanx, tany = 64, 32, 1.0, 0.5
    rs = camera(W, H, tanx, tany)
    x, y, z = 0.3, -0.2, 2.5
    st = O.forward(rs, np.array([[x, y, z]], np.float32), np.array([[0.5]], np.float32),
                   colors_precomp=np.array([[0.3, 0.6, 0.9]], np.float32), scales=np.array([[0.05, 0.05, 0.05]], np.float32),
                   rotations=np.array([[1, 0, 0, 0]], np.float32))
    assert st.radii[0] > 0
    gx, gy = 0.8, -1.5
    h = O.empty_grads(st)
    h.dL_dmeans2D[0] = [gx, gy, 0.0]
    O.backward_chain(st, h)
    mw = 1.0 / (z + 1e-7)
    want = [gx * mw / tanx, gy * mw / tany, -(x / tanx) * mw * mw * gx - (y / tany) * mw * mw * gy]
    np.testing.assert_allclose(h.dL_dmeans3D[0], want, rtol=2e-6)


def test_sh_degree_one_colour_and_gradient_signs():
    """forward.cu:20-71: rgb = 0.5 + C0 sh0 - C1 y sh1 + C1 z sh2 - C1 x sh3 with dir = normalize(p - campos),
    clamped at 0 with the clamp recorded; backward.cu:32-34,54-62: dL_dsh_k = basis_k dL_dRGB, zero where clamped."""
    rs = camera(32, 32, 1.0, 1.0, deg=1)
    p = np.array([[0.6, -0.3, 2.0]], np.float32)
    n = np.linalg.norm(p[0].astype(np.float64))
    dx, dy, dz = (p[0].astype(np.float64) / n)
    sh = np.zeros((1, 4, 3), np.float32)
    sh[0, :, 0] = [0.4, 0.3, -0.2, 0.5]     # red
    sh[0, :, 1] = [-0.5, 0.1, 0.2, -0.3]    # green
    sh[0, :, 2] = [-3.0, 0.0, 0.0, 0.0]     # blue: 0.5 + C0 * -3 < 0 -> clamped
    st = O.forward(rs, p, np.array([[0.5]], np.float32), shs=sh, scales=np.array([[0.1, 0.1, 0.1]], np.float32),
                   rotations=np.array([[1, 0, 0, 0]], np.float32))
    want = [0.5 + SH_C0 * sh[0, 0, ch] - SH_C1 * dy * sh[0, 1, ch] + SH_C1 * dz * sh[0, 2, ch] - SH_C1 * dx * sh[0, 3, ch]
            for ch in range(3)]
    assert want[2] < 0
    np.testing.assert_allclose(st.rgb[0], [want[0], want[1], 0.0], rtol=1e-6)
    np.testing.assert_array_equal(st.clamped[0], [0, 0, 1])
    h = O.empty_grads(st)
    h.dL_dcolors[0] = [1.0, 2.0, 4.0]
    O.backward_chain(st, h)
    basis = [SH_C0, -SH_C1 * dy, SH_C1 * dz, -SH_C1 * dx]
    for k in range(4):
        np.testing.assert_allclose(h.dL_dsh[0, k], [basis[k] * 1.0, basis[k] * 2.0, 0.0], rtol=1e-6, atol=1e-12)
