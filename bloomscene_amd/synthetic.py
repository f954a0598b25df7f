"""Seeded synthetic Gaussians for benchmarks and parity tests.

Scene A ("frustum-filled"): camera at the origin looking +z, Gaussians uniform in NDC x/y and
depth 1..10 with screen sigma of roughly 0.5..5 px at 1080p.  Scene B ("shell"): directions
uniform on the sphere around the origin, for the rotate360 view sweep.  Everything is
generated on the CPU with ``torch.Generator().manual_seed(seed)`` in fp32.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch

from .cameras import identity_camera, rotate360_cameras


def _common(P, sh_degree, dist, g):
    lo, hi = math.log(3e-4), math.log(3e-3)
    scales = torch.exp(torch.rand(P, 3, generator=g) * (hi - lo) + lo) * dist[:, None]
    rot = torch.randn(P, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * 1.5)
    M = (sh_degree + 1) ** 2
    shs = torch.randn(P, M, 3, generator=g) * 0.1
    shs[:, 0, :] = torch.randn(P, 3, generator=g) * 0.5
    return scales.contiguous(), rot.contiguous(), opac.contiguous(), shs.contiguous()


def scene_a(P, width, height, sh_degree, seed=0, fovx_deg=60.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    fovx = math.radians(fovx_deg)
    cam = identity_camera(width, height, fovx)
    tanfovx = math.tan(cam.FoVx * 0.5)
    tanfovy = math.tan(cam.FoVy * 0.5)
    z = torch.rand(P, generator=g) * 9.0 + 1.0
    ndc = torch.rand(P, 2, generator=g) * 2.0 - 1.0
    means = torch.stack([ndc[:, 0] * z * tanfovx, ndc[:, 1] * z * tanfovy, z], dim=1).contiguous()
    scales, rot, opac, shs = _common(P, sh_degree, z, g)
    return SimpleNamespace(means3D=means, scales=scales, rotations=rot, opacities=opac, shs=shs,
                           sh_degree=sh_degree, cameras=[cam], width=width, height=height)


def scene_b(P, width, height, sh_degree, n_views=64, seed=0, fovx_deg=60.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    d = torch.randn(P, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    r = torch.rand(P, generator=g) * 9.0 + 1.0
    means = (d * r[:, None]).contiguous()
    scales, rot, opac, shs = _common(P, sh_degree, r, g)
    cams = rotate360_cameras(n_views, width, height, math.radians(fovx_deg))
    return SimpleNamespace(means3D=means, scales=scales, rotations=rot, opacities=opac, shs=shs,
                           sh_degree=sh_degree, cameras=cams, width=width, height=height)


def anchor_scene(n_anchor, n_offsets, width, height, seed=0, fovx_deg=60.0, keep_fraction=0.5):
    """BloomScene-shaped input (Scaffold-GS anchors, gaussian_renderer/__init__.py:165-203): ``n_anchor`` anchors spread
    through the frustum of the scene-A camera with ``n_offsets`` candidate Gaussians each; the MLP-head outputs are
    seeded noise with about ``keep_fraction`` of the candidates selected (positive opacity).  Returns the camera and
    the six tensors ``neural_gaussians.expand_anchors`` / ``views.render_neural`` take."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    cam = identity_camera(width, height, math.radians(fovx_deg))
    tanfovx, tanfovy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    z = torch.rand(n_anchor, generator=g) * 9.0 + 1.0
    ndc = torch.rand(n_anchor, 2, generator=g) * 2.0 - 1.0
    anchor = torch.stack([ndc[:, 0] * z * tanfovx, ndc[:, 1] * z * tanfovy, z], dim=1).contiguous()
    # columns 0-2 scale the offsets (voxel size), 3-5 the Gaussians (screen sigma of roughly 0.5..5 px at 512^2)
    lo, hi = math.log(1e-3), math.log(1e-2)
    voxel = torch.exp(torch.rand(n_anchor, 3, generator=g) * (hi - lo) + lo) * z[:, None] * 2.0
    gs = torch.exp(torch.rand(n_anchor, 3, generator=g) * (hi - lo) + lo) * z[:, None]
    grid_scaling = torch.cat([voxel, gs], dim=1).contiguous()
    n = n_anchor * n_offsets
    grid_offsets = torch.randn(n_anchor, n_offsets, 3, generator=g)
    op = torch.tanh(torch.randn(n, 1, generator=g)).abs() + 1e-3
    neural_opacity = torch.where(torch.rand(n, 1, generator=g) < keep_fraction, op, -op)
    color = torch.rand(n, 3, generator=g)
    scale_rot = torch.randn(n, 7, generator=g) * 1.5
    return SimpleNamespace(camera=cam, anchor=anchor, grid_scaling=grid_scaling, grid_offsets=grid_offsets,
                           neural_opacity=neural_opacity, color=color, scale_rot=scale_rot, width=width, height=height)


def upstream_grads(width, height, seed=1):
    g = torch.Generator(device="cpu").manual_seed(seed)
    gC = torch.randn(3, height, width, generator=g)
    gD = torch.randn(1, height, width, generator=g)
    return gC, gD
