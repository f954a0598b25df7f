"""Cross-check of the C oracle against an INDEPENDENT float64 PyTorch-autograd restatement
(oracle/torch_splat.py).  Because the reference has no tests of its own, this is what stands
between the oracle and a transcription error: forward outputs and every hand-written gradient
path (cov2D/cov3D chain, projection, SH incl. view direction, opacity, colour, means2D) must agree
with autograd wherever the reference backward is the true derivative."""
import math

import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import oracle as O
from oracle import torch_splat as TS

TOL = 5e-5   # fp32 oracle vs fp64 autograd, relative to the largest magnitude of the tensor (a general
             # camera rotation makes every view-matrix entry inexact: up to 2.5e-5 seen; identity: 1e-5)


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("kw", [
    dict(P=300, W=48, H=32, deg=3, seed=0, scale_mul=6.0),
    dict(P=200, W=40, H=40, deg=1, seed=3, scale_mul=6.0, bg=(0.0, 0.0, 0.0)),
    dict(P=150, W=33, H=20, deg=0, seed=5, scale_mul=6.0),
    dict(P=250, W=48, H=48, deg=2, seed=7, scale_mul=6.0, scale_modifier=1.3),
    dict(P=250, W=45, H=30, deg=0, seed=9, scale_mul=6.0, color_mode="precomp"),
    dict(P=250, W=45, H=30, deg=2, seed=10, scale_mul=6.0, cov_mode="precomp"),
    dict(P=300, W=40, H=24, deg=1, seed=12, scale_mul=8.0, scene="b", view=9),
    dict(P=300, W=48, H=32, deg=3, seed=14, scale_mul=6.0, free_camera=True),    # campos != 0, general W2C
    dict(P=250, W=40, H=40, deg=2, seed=15, scale_mul=6.0, free_camera=True, cov_mode="precomp"),
])
def test_oracle_matches_autograd(kw):
    c = Hh.make_case(**kw)
    c.opacities.clamp_(max=0.95)   # keep the (undifferentiated) 0.99 alpha clamp inactive
    st, g = Hh.run_oracle(c)
    dt = torch.float64

    def leaf(t):
        return None if t is None else t.to(dt).clone().requires_grad_(True)
    inp = dict(means3D=leaf(c.means3D), opacities=leaf(c.opacities), shs=leaf(c.shs),
               colors_precomp=leaf(c.colors_precomp), scales=leaf(c.scales), rotations=leaf(c.rotations),
               cov3D_precomp=leaf(c.cov3D_precomp))
    means2D = torch.zeros(c.P, 3, dtype=dt, requires_grad=True)
    color, radii, depth = TS.render(inp["means3D"], inp["opacities"], c.cam.world_view_transform,
                                    c.cam.full_proj_transform, c.cam.camera_center, c.tanfovx, c.tanfovy, c.W, c.H,
                                    c.bg, c.scale_modifier, c.deg, shs=inp["shs"], colors_precomp=inp["colors_precomp"],
                                    scales=inp["scales"], rotations=inp["rotations"],
                                    cov3D_precomp=inp["cov3D_precomp"], means2D=means2D)
    (color * c.gC.to(dt)).sum().backward()

    assert (radii.numpy() == st.radii).all()
    assert st.num_rendered > 0
    assert _rel(st.color, color.detach().numpy()) < TOL
    assert _rel(st.depth, depth.detach().numpy()) < TOL
    og = Hh.oracle_grads(c, g)
    assert _rel(og.means2D[:, :2], means2D.grad.numpy()[:, :2]) < TOL
    for k in ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations"):
        if getattr(og, k) is None:
            continue
        ours = getattr(og, k)
        if k == "scales":
            # reference deviation: dL_dscale = dot(Rt[i], dL_dMt[i]) lacks the scale_modifier factor
            # of S = mod * scale (backward.cu:322-325), so it is the true derivative divided by mod
            ours = ours * c.scale_modifier
        assert _rel(ours, inp[k].grad.numpy()) < TOL, k
    if c.cov3D_precomp is not None:
        # the reference stores off-diagonal gradients "once, doubled" (backward.cu:221-227);
        # autograd w.r.t. the 6 packed entries sees each off-diagonal twice as well
        assert _rel(og.cov3D_precomp, inp["cov3D_precomp"].grad.numpy()) < TOL


@pytest.mark.parametrize("kw", [
    dict(P=300, W=48, H=32, deg=2, seed=20, scale_mul=6.0),
    dict(P=250, W=40, H=40, deg=0, seed=21, scale_mul=8.0, color_mode="precomp", free_camera=True),
    dict(P=300, W=40, H=24, deg=1, seed=22, scale_mul=8.0, scene="b", view=9),
])
def test_depth_gradient_extension_matches_autograd(kw):
    """EXTENSION: with depth_gradient=True the oracle adds the true derivative of
    the normalised depth target; its pin is float64 autograd through the same forward semantics
    (acc > 0.5 gate, D / acc).  The default (reference) behaviour ignores grad_depth: see
    test_oracle.py::test_depth_gradient_is_ignored."""
    c = Hh.make_case(**kw)
    c.opacities.clamp_(max=0.95)
    st = O.forward(Hh.oracle_settings(c), c.means3D, c.opacities, shs=c.shs, colors_precomp=c.colors_precomp,
                   scales=c.scales, rotations=c.rotations, cov3D_precomp=c.cov3D_precomp)
    g = O.backward(st, c.gC, c.gD, depth_gradient=True)
    dt = torch.float64

    def leaf(t):
        return None if t is None else t.to(dt).clone().requires_grad_(True)
    inp = dict(means3D=leaf(c.means3D), opacities=leaf(c.opacities), shs=leaf(c.shs),
               colors_precomp=leaf(c.colors_precomp), scales=leaf(c.scales), rotations=leaf(c.rotations))
    means2D = torch.zeros(c.P, 3, dtype=dt, requires_grad=True)
    color, radii, depth = TS.render(inp["means3D"], inp["opacities"], c.cam.world_view_transform,
                                    c.cam.full_proj_transform, c.cam.camera_center, c.tanfovx, c.tanfovy, c.W, c.H,
                                    c.bg, c.scale_modifier, c.deg, shs=inp["shs"], colors_precomp=inp["colors_precomp"],
                                    scales=inp["scales"], rotations=inp["rotations"], means2D=means2D,
                                    depth_gradient=True)
    assert (depth != 0).any() and (depth == 0).any()        # both sides of the acc > 0.5 gate
    ((color * c.gC.to(dt)).sum() + (depth * c.gD.to(dt)).sum()).backward()
    og = Hh.oracle_grads(c, g)
    assert _rel(og.means2D[:, :2], means2D.grad.numpy()[:, :2]) < TOL
    for k in ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations"):
        if getattr(og, k) is None:
            continue
        assert _rel(getattr(og, k), inp[k].grad.numpy()) < TOL, k
    # and the extension really changed something: the reference-mode gradient differs
    g_ref = O.backward(st, c.gC, c.gD)
    assert _rel(Hh.oracle_grads(c, g_ref).means3D, inp["means3D"].grad.numpy()) > 1e-2
