#!/usr/bin/env python3
"""Generates the golden fixtures tests/golden/*.npz.

The reference (/root/reference) ships no golden vectors for this path and can be neither built nor
imported here, so these vectors come from this repository's own CPU oracle
(oracle/bsr_oracle.c) after it was cross-checked against the independent torch-autograd
restatement (oracle/torch_splat.py; tests/test_oracle_crosscheck.py).  They pin the oracle against
regressions and travel to the GPU box as data.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as Hh  # noqa: E402

CASES = {
    # name: make_case kwargs  (P <= 4k, images <= 160x96 incl. sizes that are no multiple of 16)
    "sh3_160x96": dict(P=1500, W=160, H=96, deg=3, seed=0),
    "sh1_133x75_near": dict(P=2000, W=133, H=75, deg=1, seed=1, scale_mul=4.0, near_fraction=0.1),
    "sh0_precomp_color_64x64": dict(P=1500, W=64, H=64, deg=0, seed=2, color_mode="precomp"),
    "sh2_precomp_cov_100x50": dict(P=1500, W=100, H=50, deg=2, seed=3, cov_mode="precomp", scale_mul=3.0),
    "sh3_dense_48x48": dict(P=2500, W=48, H=48, deg=3, seed=4, scale_mul=10.0),
    "sh1_extraM_scalemod_80x48": dict(P=1000, W=80, H=48, deg=1, seed=5, M_extra=5, scale_modifier=1.7, bg=(1.0, 0.5, 0.0)),
    "shell_view3_96x64": dict(P=2000, W=96, H=64, deg=2, seed=6, scene="b", view=3, scale_mul=5.0),
    "sh3_free_camera_120x80": dict(P=1500, W=120, H=80, deg=3, seed=8, scale_mul=3.0, free_camera=True),
}


def golden_arrays(c):
    st, g = Hh.run_oracle(c, want_abs_sums=True)
    out = dict(color=st.color, depth=st.depth, radii=st.radii, final_T=st.final_T, n_contrib=st.n_contrib,
               means2D=st.means2D, depths=st.depths, cov3D=st.cov3D, conic_opacity=st.conic_opacity, rgb=st.rgb,
               clamped=st.clamped, tiles_touched=st.tiles_touched, point_list=st.point_list,
               point_list_keys=st.point_list_keys, ranges=st.ranges, num_rendered=np.int64(st.num_rendered),
               dL_dmeans3D=g.dL_dmeans3D, dL_dmeans2D=g.dL_dmeans2D, dL_dcolors=g.dL_dcolors, dL_dconic=g.dL_dconic,
               dL_dopacity=g.dL_dopacity, dL_dcov3D=g.dL_dcov3D, dL_dsh=g.dL_dsh, dL_dscales=g.dL_dscales,
               dL_drotations=g.dL_drotations,
               # sum |term| of the nine pair sums per Gaussian (mean2D.x,y conic.x,y,w opacity colour r,g,b): the accuracy
               # to which the reference itself defines them (unordered fp32 atomicAdds) -- the elementwise bound of
               # test_forward_and_backward_vs_golden_fixture
               abs_sums=g.abs_sums)
    # inputs, so that a fixture is self-contained data (absent optional inputs are stored empty)
    e = np.zeros(0, dtype=np.float32)
    cam = c.cam
    out.update(in_means3D=c.means3D.numpy(), in_opacities=c.opacities.numpy(),
               in_shs=e if c.shs is None else c.shs.numpy(),
               in_colors_precomp=e if c.colors_precomp is None else c.colors_precomp.numpy(),
               in_scales=e if c.scales is None else c.scales.numpy(),
               in_rotations=e if c.rotations is None else c.rotations.numpy(),
               in_cov3D_precomp=e if c.cov3D_precomp is None else c.cov3D_precomp.numpy(),
               in_bg=c.bg.numpy(), in_viewmatrix=cam.world_view_transform.numpy(),
               in_projmatrix=cam.full_proj_transform.numpy(), in_campos=cam.camera_center.numpy(),
               in_scalars=np.array([c.W, c.H, c.deg, c.tanfovx, c.tanfovy, c.scale_modifier], dtype=np.float64),
               in_gC=c.gC.numpy(), in_gD=c.gD.numpy())
    return out


if __name__ == "__main__":
    only = set(sys.argv[1:])   # optional: names to (re)generate; default all
    for name, kw in CASES.items():
        if only and name not in only:
            continue
        c = Hh.make_case(**kw)
        arrs = golden_arrays(c)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **arrs)
        print(name, "R =", int(arrs["num_rendered"]), os.path.getsize(path) // 1024, "KiB")
