"""The conservative bounding-square cull of k_visible_filter_views (bloomscene_amd/csrc/preprocess.hip, phase A), checked
on the CPU against the oracle's radii: restated here in numpy with the kernel's constants, it must never drop a
(view, Gaussian) pair the reference's visible_filter keeps -- on splats of every size, centres far outside and on the
frustum's borders, un-normalised quaternions, indefinite precomputed covariances and a view matrix that also scales.
(The GPU counterpart, tests/test_round3_gpu.py::test_multi_view_filter_bounding_square_cull_never_changes_a_radius,
checks the kernel's rows bit for bit against the single-view filter; this one checks the inequality itself.)"""
import math

import numpy as np
import torch

from bloomscene_amd.synthetic import scene_b
from oracle import oracle as O

NEAR = 0.2


def cull_outside(means, cov6, vm, pm, W, H, tanx, tany):
    """True where phase A declares the pair outside (float32 arithmetic as in the kernel, rcp / sqrt exact here)."""
    f = np.float32
    fx, fy = f(W / (2.0 * tanx)), f(H / (2.0 * tany))
    limx, limy = f(1.3) * f(tanx), f(1.3) * f(tany)
    K = fx * fx * (f(1) + limx * limx) + fy * fy * (f(1) + limy * limy)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    x, y, z = means[:, 0], means[:, 1], means[:, 2]
    c = cov6
    sF = np.sqrt(c[:, 0] ** 2 + c[:, 3] ** 2 + c[:, 5] ** 2 + f(2) * (c[:, 1] ** 2 + c[:, 2] ** 2 + c[:, 4] ** 2), dtype=f)
    pvz = vm[2] * x + vm[6] * y + vm[10] * z + vm[14]
    hx = pm[0] * x + pm[4] * y + pm[8] * z + pm[12]
    hy = pm[1] * x + pm[5] * y + pm[9] * z + pm[13]
    hw = pm[3] * x + pm[7] * y + pm[11] * z + pm[15]
    with np.errstate(all="ignore"):
        p_w = f(1) / (hw + f(0.0000001))
        pxf = ((hx * p_w + f(1)) * f(W) - f(1)) * f(0.5)
        pyf = ((hy * p_w + f(1)) * f(H) - f(1)) * f(0.5)
        wF2 = sum(f(vm[i]) ** 2 for i in (0, 1, 2, 4, 5, 6, 8, 9, 10))
        itz = f(1) / pvz
        B = (K * wF2) * sF * (itz * itz)
        rb = f(3.01) * np.sqrt(f(1.5015) * B + f(1), dtype=f) + f(2)
        slx, sly = f(2) + f(1e-5) * np.abs(pxf), f(2) + f(1e-5) * np.abs(pyf)
        outside = (pxf + rb < -slx) | (pxf - rb > f(16 * gx) + slx) | (pyf + rb < -sly) | (pyf - rb > f(16 * gy) + sly)
    return outside & ~(pvz <= NEAR), rb


def test_bounding_square_cull_never_drops_a_visible_pair():
    gen = torch.Generator().manual_seed(123)
    P, W, H, V = 20000, 200, 120, 12
    sc = scene_b(P, W, H, 1, n_views=V, seed=5)
    means = (sc.means3D * torch.exp(torch.rand(P, 1, generator=gen) * 6.0 - 3.0)).numpy().astype(np.float32)
    scales = torch.exp(torch.rand(P, 3, generator=gen) * 16.0 - 10.0).numpy().astype(np.float32)
    rots = (torch.randn(P, 4, generator=gen) * torch.exp(torch.randn(P, 1, generator=gen))).numpy().astype(np.float32)
    sym = (torch.randn(P, 6, generator=gen) * torch.exp(torch.rand(P, 1, generator=gen) * 12.0 - 8.0)).numpy().astype(np.float32)
    cams = sc.cameras
    tanx, tany = math.tan(cams[0].FoVx * 0.5), math.tan(cams[0].FoVy * 0.5)
    kept = culled = 0
    for mode in ("scale_rot", "cov"):
        for v, cam in enumerate(cams):
            vm = cam.world_view_transform.numpy().astype(np.float32).copy()
            if v == 3:
                vm[:, :3] *= 1.7          # a view matrix that also scales
            pm = cam.full_proj_transform.numpy().astype(np.float32)
            rs = O.make_settings(H, W, tanx, tany, [0, 0, 0], 1.0, vm, pm, 1, cam.camera_center)
            if mode == "scale_rot":
                st = O._preprocess(rs, means, None, None, None, scales, rots, None, True)
                cov6 = np.asarray(st.cov3D, dtype=np.float32).reshape(P, 6)
            else:
                st = O._preprocess(rs, means, None, None, None, None, None, sym, True)
                cov6 = sym
            radii = np.asarray(st.radii)
            outside, rb = cull_outside(means, cov6, vm.reshape(-1), pm.reshape(-1), W, H, tanx, tany)
            bad = outside & (radii > 0)
            assert not bad.any(), (mode, v, int(bad.sum()), means[bad][:3], radii[bad][:3], rb[bad][:3])
            # the radius bound itself, wherever the oracle computed a radius
            vis = radii > 0
            assert (rb[vis] >= radii[vis]).all(), (mode, v)
            kept += int(vis.sum())
            culled += int(outside.sum())
    assert kept > 1000 and culled > 10 * kept // 100      # the test saw visible pairs, and the cull is not vacuous


def test_radius_bound_holds_for_adversarial_indefinite_covariances():
    """the kernel's comment derives lambda_max <= 1.5 B + 0.62 with a step that reads as if S were
    positive semi-definite.  The bound itself needs no such assumption: the reference's lambda_1 = mid + sqrt(max(0.1,
    mid^2 - det)) is max(largest eigenvalue of cov2D, mid + sqrt(0.1)), cov2D = A S A^T + 0.3 I, and for ANY symmetric
    S the eigenvalues of A S A^T lie in [-B', B'] with B' = |A|_2^2 |S|_2 <= B, so lambda_1 <= B + 0.3 + sqrt(0.1)
    < 1.5 B + 0.62.  Checked where it would break first: rank-2 covariances s (u u^T - w w^T) with u, w orthonormal (one
    eigenvalue +s, one -s: the widest eigenvalue gap a given Frobenius norm allows), u swept over directions including
    the rows of the projection Jacobian, s over ten decades; the oracle's radius must stay below the cull's bound and no
    visible pair may be dropped."""
    gen = torch.Generator().manual_seed(321)
    P, W, H = 60000, 200, 120
    sc = scene_b(P, W, H, 1, n_views=4, seed=9)
    means = (sc.means3D * torch.exp(torch.rand(P, 1, generator=gen) * 4.0 - 2.0)).numpy().astype(np.float32)
    u = torch.randn(P, 3, generator=gen)
    u[: P // 3] = torch.tensor([1.0, 0.0, 0.0])       # along / across the image axes of an axis-aligned camera
    u[P // 3: 2 * P // 3] = torch.tensor([0.0, 1.0, 0.0])
    u = u / u.norm(dim=1, keepdim=True)
    w = torch.randn(P, 3, generator=gen)
    w = w - (w * u).sum(1, keepdim=True) * u
    w = w / w.norm(dim=1, keepdim=True)
    s = torch.exp(torch.rand(P, 1, 1, generator=gen) * 23.0 - 16.0)          # 1e-7 .. 1e3
    S = s * (u[:, :, None] * u[:, None, :] - w[:, :, None] * w[:, None, :])
    sym = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1).numpy().astype(np.float32)
    tanx, tany = math.tan(sc.cameras[0].FoVx * 0.5), math.tan(sc.cameras[0].FoVy * 0.5)
    kept = 0
    for v, cam in enumerate(sc.cameras):
        vm = cam.world_view_transform.numpy().astype(np.float32).copy()
        pm = cam.full_proj_transform.numpy().astype(np.float32)
        rs = O.make_settings(H, W, tanx, tany, [0, 0, 0], 1.0, vm, pm, 1, cam.camera_center)
        st = O._preprocess(rs, means, None, None, None, None, None, sym, True)
        radii = np.asarray(st.radii)
        outside, rb = cull_outside(means, sym, vm.reshape(-1), pm.reshape(-1), W, H, tanx, tany)
        vis = radii > 0
        assert not (outside & vis).any(), (v, int((outside & vis).sum()))
        assert (rb[vis] >= radii[vis]).all(), (v, float((radii[vis] / rb[vis]).max()))
        kept += int(vis.sum())
    assert kept > 2000
