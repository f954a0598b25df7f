#!/usr/bin/env python3
"""Host-side cost of one rasterizer step (python + ctypes + torch allocator + autograd), measured where the GPU is
not the limit: the C3 call shape on a SMALL scene (P = 20 000), so that every step's wall time is the host's.
    python3 tools/host_profile.py [--steps 300] [--profile]
Prints ms per step of forward, backward and the whole step; --profile adds a cProfile table (top 35 by cumulative time)."""
import argparse, cProfile, math, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bloomscene_amd import GaussianRasterizationSettings, GaussianRasterizer
from bloomscene_amd.synthetic import scene_a, upstream_grads
from bloomscene_amd.views import yawed_camera

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--gaussians", type=int, default=20000)
ap.add_argument("--profile", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
W, H, deg = 1920, 1080, 3
sc = scene_a(a.gaussians, W, H, deg, seed=0)
leaves = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
gC, gD = [t.to(dev) for t in upstream_grads(W, H, seed=1)]
cam = yawed_camera(W, H, math.radians(60.0), yaw_deg=0.0).to(dev)
st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                                   bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                                   projmatrix=cam.full_proj_transform, sh_degree=deg, campos=cam.camera_center, prefiltered=False, debug=False)
rast = GaussianRasterizer(st)
t_f = t_b = 0.0

def step():
    global t_f, t_b
    t0 = time.perf_counter()
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    color, radii, depth = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=leaves["shs"],
                               scales=leaves["scales"], rotations=leaves["rotations"])
    t1 = time.perf_counter()
    for v in leaves.values():
        v.grad = None
    torch.autograd.backward((color, depth), (gC, gD))
    t2 = time.perf_counter()
    t_f += t1 - t0
    t_b += t2 - t1

for _ in range(20):
    step()
torch.cuda.synchronize()
t_f = t_b = 0.0
pr = cProfile.Profile() if a.profile else None
t0 = time.perf_counter()
if pr:
    pr.enable()
for _ in range(a.steps):
    step()
if pr:
    pr.disable()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"host-bound step (P={a.gaussians}): {dt / a.steps * 1e3:.4f} ms  forward call {t_f / a.steps * 1e3:.4f} ms  "
      f"backward call {t_b / a.steps * 1e3:.4f} ms  ({os.cpu_count()} host threads)")
if pr:
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
