"""Independent dense PyTorch (float64, autograd) point-splat of the same rasterizer maths.

TEST INFRASTRUCTURE ONLY.  A second restatement, written from the maths of SURVEY.md
Appendix A rather than from ``bsr_oracle.c``, used to cross-check the C oracle's forward and
-- through ``torch.autograd`` -- its hand-written backward on inputs where the reference
backward *is* the derivative of its forward (opacity < 0.99 so the alpha clamp is inactive,
Gaussians inside the +-1.3*tanfov guard band; SURVEY.md §8c lists the deviations).  The depth
target is computed but, like the reference (cuda_rasterizer/backward.cu:457-463,539-554), is
detached from the graph.

Dense O(pixels x Gaussians): only for small cases (P <= ~500, <= ~64x64 px).
"""
from __future__ import annotations

import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def sh_to_rgb(deg, shs, dirs):
    """cuda_rasterizer/forward.cu:20-71 in vector form; returns (rgb clamped at 0)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6]
               + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
               + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
               + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
               + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R  # standard rotation matrix Rq (row-major), no normalisation


def render(means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H, bg, scale_modifier=1.0,
           sh_degree=0, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, means2D=None,
           depth_gradient=False):
    """Returns (color [3,H,W], radii [P] int, depth [1,H,W]) as float64 torch tensors.

    ``means2D`` (optional zeros [P,3] leaf) is added to the NDC position so that its autograd
    gradient is the reference's dL_dmean2D (gradient w.r.t. NDC, backward.cu:473-474,574-575).
    The depth output is detached (the reference drops dL_ddepth) unless ``depth_gradient=True``, which
    keeps it in the graph: the pin of the opt-in depth-gradient extension."""
    dt = torch.float64
    P = means3D.shape[0]
    V = viewmatrix.to(dt).reshape(4, 4)   # V[j, i] = column-vector matrix element (i, j)
    Pm = projmatrix.to(dt).reshape(4, 4)
    Rwc = V[:3, :3].T
    tw = V[3, :3]
    t = means3D @ Rwc.T + tw                               # view space
    hom = torch.cat([means3D, torch.ones(P, 1, dtype=dt)], dim=1) @ Pm
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    if means2D is not None:
        ndc = ndc + means2D[:, :2]
    visible = t[:, 2] > 0.2
    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    if cov3D_precomp is None:
        Rq = quat_to_rot(rotations)
        S = scale_modifier * scales
        Mx = Rq * S[:, None, :]
        Sigma = Mx @ Mx.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                            dim=1).reshape(-1, 3, 3)
    tz = t[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txc = torch.clamp(t[:, 0] / tz, -limx, limx) * tz
    tyc = torch.clamp(t[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * txc / (tz * tz), zero, fy / tz, -fy * tyc / (tz * tz)], dim=1).reshape(-1, 2, 3)
    A = J @ Rwc
    cov2 = A @ Sigma @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c2 = cov2[:, 1, 1] + 0.3
    det = a * c2 - b * b
    visible = visible & (det != 0)
    conic_a, conic_b, conic_c = c2 / det, -b / det, a / det
    mid = 0.5 * (a + c2)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16

    def trunc_clamp(v, hi):
        return torch.clamp(torch.trunc(v), 0, hi).to(torch.int64)
    pxd, pyd = px.detach(), py.detach()
    xmin, xmax = trunc_clamp((pxd - radius) / 16, gx), trunc_clamp((pxd + radius + 15) / 16, gx)
    ymin, ymax = trunc_clamp((pyd - radius) / 16, gy), trunc_clamp((pyd + radius + 15) / 16, gy)
    visible = visible & (((xmax - xmin) * (ymax - ymin)) > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is None:
        d = means3D - campos.to(dt)
        d = d / d.norm(dim=1, keepdim=True)
        rgb = sh_to_rgb(sh_degree, shs, d)
    else:
        rgb = colors_precomp

    # global (depth bits, id) order == per-tile order of the stable key sort
    depth32 = tz.detach().to(torch.float32)
    order = sorted(range(P), key=lambda i: (float(depth32[i]), i))
    order = torch.tensor([i for i in order if bool(visible[i])], dtype=torch.int64)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pixx, pixy = xs.reshape(-1), ys.reshape(-1)
    tilex, tiley = (pixx // 16).to(torch.int64), (pixy // 16).to(torch.int64)
    o = order
    n = o.numel()
    color = torch.zeros(3, H * W, dtype=dt)
    depth = torch.zeros(H * W, dtype=dt)
    if n == 0:
        color = color + bg.to(dt)[:, None]
        return color.reshape(3, H, W), radii, depth.reshape(1, H, W)
    in_tile = ((tilex[:, None] >= xmin[o][None]) & (tilex[:, None] < xmax[o][None]) &
               (tiley[:, None] >= ymin[o][None]) & (tiley[:, None] < ymax[o][None]))
    dx = px[o][None, :] - pixx[:, None]
    dy = py[o][None, :] - pixy[:, None]
    power = -0.5 * (conic_a[o][None] * dx * dx + conic_c[o][None] * dy * dy) - conic_b[o][None] * dx * dy
    alpha = torch.clamp_max(opacities.reshape(-1)[o][None] * torch.exp(power), 0.99)
    ok = in_tile & (power <= 0) & (alpha >= 1.0 / 255.0)
    a_eff = torch.where(ok, alpha, torch.zeros_like(alpha))
    test_T = torch.cumprod(1.0 - a_eff, dim=1)              # T after blending entry i
    stop = ok & (test_T.detach() < 1e-4)
    after_stop = torch.cumsum(stop.to(torch.int64), dim=1) > 0  # the stopping entry itself is not blended
    keep = ok & ~after_stop
    a_keep = torch.where(keep, alpha, torch.zeros_like(alpha))
    T_after = torch.cumprod(1.0 - a_keep, dim=1)
    T_before = torch.cat([torch.ones(H * W, 1, dtype=dt), T_after[:, :-1]], dim=1)
    wgt = a_keep * T_before
    T_final = T_after[:, -1]
    color = (wgt @ rgb[o]).T + T_final[None, :] * bg.to(dt)[:, None]
    wd = wgt if depth_gradient else wgt.detach()
    acc = 1e-6 + wd.sum(dim=1)
    Dsum = wd @ (tz if depth_gradient else tz.detach())[o]
    depth = torch.where(acc > 0.5, Dsum / acc, torch.zeros_like(acc))
    return color.reshape(3, H, W), radii, depth.reshape(1, H, W)
