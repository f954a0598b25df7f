#!/usr/bin/env python3
"""Developer diagnostic: hipMalloc / hipFree calls of the caching allocator per fwd+bwd step."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bloomscene_amd import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from bloomscene_amd.synthetic import scene_a, upstream_grads  # noqa: E402
from bloomscene_amd.views import yawed_camera  # noqa: E402

P, W, H, deg = int(os.environ.get("P", 1000000)), 1920, 1080, 3
dev = torch.device("cuda:0")
sc = scene_a(P, W, H, deg, seed=0)
leaves = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
gC, gD = [t.to(dev) for t in upstream_grads(W, H, seed=1)]
cam = yawed_camera(W, H, math.radians(60.0), yaw_deg=0.0).to(dev)
st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5),
                                   tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.zeros(3, device=dev), scale_modifier=1.0,
                                   viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=deg,
                                   campos=cam.camera_center, prefiltered=False, debug=False)
rast = GaussianRasterizer(st)
import gc
gc.disable()
for i in range(14):
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    color, radii, depth = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=leaves["shs"],
                               colors_precomp=None, scales=leaves["scales"], rotations=leaves["rotations"])
    for v in leaves.values():
        v.grad = None
    torch.autograd.backward((color, depth), (gC, gD))
    torch.cuda.synchronize()
    m = torch.cuda.memory_stats(dev)
    cur = (m["num_device_alloc"], m["num_device_free"], m["reserved_bytes.all.current"] >> 20,
           m["allocated_bytes.all.peak"] >> 20, m["num_alloc_retries"])
    print(i, cur)
if os.environ.get("SNAP") == "1":
    print(torch.cuda.memory_summary(dev))
