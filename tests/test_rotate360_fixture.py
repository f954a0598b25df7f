"""The rotate360 render path (bloomscene.py:191-193: per view prefilter_voxel -> render -> frame, depth) against the
committed fixture tests/golden/rotate360/rotate360_6of8_96x64.npz (SURVEY.md §8c last row; generator:
tests/golden/make_rotate360.py): six yaw angles of an 8-view sweep with the exact camera matrices and the oracle's
visible_filter radii, colour, depth and radii.

CPU: the camera builder reproduces the stored matrices exactly and the oracle reproduces the stored outputs bit for
bit.  GPU: the HIP path, driven the way BloomScene's loop drives the reference (views.prefilter -> views.render_view
on the visible Gaussians), reproduces them bit for bit -- one view at a time, through the batched multi-view entry
points, and without the prefilter (a Gaussian the filter drops has radius 0 in the render as well)."""
import math
import os

import numpy as np
import pytest
import torch

from bloomscene_amd import cameras
from oracle import oracle as O

PATH = os.path.join(os.path.dirname(__file__), "golden", "rotate360", "rotate360_6of8_96x64.npz")


def _load():
    z = np.load(PATH)
    W, H, n_views, deg = (int(v) for v in z["in_scalars"])
    return z, W, H, n_views, deg, [int(v) for v in z["views"]]


def test_camera_builder_reproduces_the_stored_matrices():
    z, W, H, n_views, deg, views = _load()
    assert len(views) >= 4
    cams = cameras.rotate360_cameras(n_views, W, H, math.radians(60.0))
    for v in views:
        cam = cams[v]
        np.testing.assert_array_equal(cam.world_view_transform.numpy(), z[f"v{v}_viewmatrix"])
        np.testing.assert_array_equal(cam.full_proj_transform.numpy(), z[f"v{v}_projmatrix"])
        np.testing.assert_array_equal(cam.camera_center.numpy(), z[f"v{v}_campos"])
        assert math.tan(cam.FoVx * 0.5) == z[f"v{v}_tanfov"][0] and math.tan(cam.FoVy * 0.5) == z[f"v{v}_tanfov"][1]
        # yaw = 360 v / n about +Y: the stored (transposed) W2C rotation is [[c, 0, -s], [0, 1, 0], [s, 0, c]]
        th = 2.0 * math.pi * v / n_views
        want = np.array([[math.cos(th), 0, -math.sin(th)], [0, 1, 0], [math.sin(th), 0, math.cos(th)]])
        np.testing.assert_allclose(z[f"v{v}_viewmatrix"][:3, :3], want, atol=1e-7)
        assert not z[f"v{v}_campos"].any()


def test_oracle_reproduces_the_rotate360_fixture():
    z, W, H, n_views, deg, views = _load()
    means, scales3 = z["in_means3D"], np.ascontiguousarray(z["in_scales6"][:, :3])
    seen = set()
    for v in views:
        rs = O.make_settings(H, W, z[f"v{v}_tanfov"][0], z[f"v{v}_tanfov"][1], z["in_bg"], 1.0, z[f"v{v}_viewmatrix"],
                             z[f"v{v}_projmatrix"], deg, z[f"v{v}_campos"])
        fr = O.visible_filter(rs, means, scales=scales3, rotations=z["in_rotations"])
        np.testing.assert_array_equal(fr, z[f"v{v}_filter_radii"])
        m = fr > 0
        st = O.forward(rs, means[m], z["in_opacities"][m], colors_precomp=z["in_colors_precomp"][m], scales=scales3[m],
                       rotations=z["in_rotations"][m])
        np.testing.assert_array_equal(st.color.view(np.uint32), z[f"v{v}_color"].view(np.uint32))
        np.testing.assert_array_equal(st.depth.view(np.uint32), z[f"v{v}_depth"].view(np.uint32))
        np.testing.assert_array_equal(st.radii, z[f"v{v}_radii"][m])
        assert not z[f"v{v}_radii"][~m].any() and st.num_rendered == int(z[f"v{v}_num_rendered"])
        seen.add(z[f"v{v}_color"].tobytes())
    assert len(seen) == len(views)          # the views really differ


@pytest.mark.gpu
def test_hip_rotate360_loop_reproduces_the_fixture():
    from bloomscene_amd import views as V
    z, W, H, n_views, deg, views = _load()
    dev = torch.device("cuda")
    t = {k: torch.from_numpy(z["in_" + k]).to(dev) for k in ("means3D", "scales6", "rotations", "opacities", "colors_precomp")}
    bg = torch.from_numpy(z["in_bg"]).to(dev)
    cams = [c.to(dev) for c in cameras.rotate360_cameras(n_views, W, H, math.radians(60.0))]
    scales3 = t["scales6"][:, :3].contiguous()
    full = dict(means3D=t["means3D"], opacities=t["opacities"], scales=scales3, rotations=t["rotations"],
                colors_precomp=t["colors_precomp"])
    for v in views:
        # the loop body of bloomscene.py:191-193 / gaussian_renderer/__init__.py:342-349,254-262
        mask = V.prefilter(cams[v], t["means3D"], t["scales6"], t["rotations"], bg)
        np.testing.assert_array_equal(mask.cpu().numpy(), z[f"v{v}_filter_radii"] > 0)
        g = {k: x[mask].contiguous() for k, x in full.items()}
        with torch.no_grad():
            res = V.render_view(cams[v], g, bg, sh_degree=deg)
        np.testing.assert_array_equal(res["render"].cpu().numpy().view(np.uint32), z[f"v{v}_color"].view(np.uint32))
        np.testing.assert_array_equal(res["depth"].cpu().numpy().view(np.uint32), z[f"v{v}_depth"].view(np.uint32))
        np.testing.assert_array_equal(res["radii"].cpu().numpy(), z[f"v{v}_radii"][mask.cpu().numpy()])
        # without the prefilter: same image (the filter only drops Gaussians whose render radius is 0 too)
        with torch.no_grad():
            res_all = V.render_view(cams[v], full, bg, sh_degree=deg)
        np.testing.assert_array_equal(res_all["render"].cpu().numpy().view(np.uint32), z[f"v{v}_color"].view(np.uint32))
        np.testing.assert_array_equal(res_all["radii"].cpu().numpy(), z[f"v{v}_radii"])
    # the batched multi-view entry points (bsr_visible_filter_views, bsr_forward_views) on the same views
    sel = [cams[v] for v in views]
    masks = V.prefilter_views(sel, t["means3D"], t["scales6"], t["rotations"])
    color, depth, radii = V.render_views_batched(sel, full, bg, deg)
    for i, v in enumerate(views):
        np.testing.assert_array_equal(masks[i].cpu().numpy(), z[f"v{v}_filter_radii"] > 0)
        np.testing.assert_array_equal(color[i].cpu().numpy().view(np.uint32), z[f"v{v}_color"].view(np.uint32))
        np.testing.assert_array_equal(depth[i].cpu().numpy().view(np.uint32), z[f"v{v}_depth"].view(np.uint32))
        np.testing.assert_array_equal(radii[i].cpu().numpy(), z[f"v{v}_radii"])
    # and through the sharded sweep helper (all views on this one rank)
    out = V.render_views_sharded(sel, full, bg, deg, rank=0, world=1, keep_outputs=True, batch=4)
    for i, v in enumerate(views):
        np.testing.assert_array_equal(out[i][0].cpu().numpy().view(np.uint32), z[f"v{v}_color"].view(np.uint32))
