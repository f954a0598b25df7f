#!/usr/bin/env python3
"""sha256 of every output of forward + backward on a few seeded scenes, for comparing two BUILDS of the library bit for bit:
    python tools/gradient_digest.py [--lib other_build.so]   ->  one JSON line {case: {tensor: digest}}"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bloomscene_amd import _capi
if len(sys.argv) > 2 and sys.argv[1] == "--lib":
    _capi.use_library(sys.argv[2])
import helpers as Hh
POOLED = "--pooled" in sys.argv   # only the frame large enough for the tile walks' tail pool (>= 4096 tiles), twice

CASES = {"c3_small": dict(P=60000, W=480, H=270, deg=3, seed=0), "dense": dict(P=40000, W=160, H=96, deg=1, seed=4, scale_mul=6.0),
         "precomp": dict(P=30000, W=200, H=120, deg=0, seed=2, color_mode="precomp", scale_mul=2.0),
         "long_lists": dict(P=20000, W=48, H=48, deg=1, seed=7, scale_mul=12.0)}
if POOLED:
    CASES = {"pooled_4096_tiles": dict(P=150000, W=1024, H=1024, deg=1, seed=5, scale_mul=1.5),
             "pooled_4096_tiles_again": dict(P=150000, W=1024, H=1024, deg=1, seed=5, scale_mul=1.5)}
if "--plans" in sys.argv:
    # frames on which both second binning passes are eligible (up to 8192 tiles): short lists, lists of every sort class
    # inside a sparse frame, 2 and 4 parts per bucket, a frame few Gaussians survive in
    CASES = {"short_lists_k1": dict(P=60000, W=480, H=270, deg=3, seed=0),
             "mixed_long_lists": dict(P=60000, W=640, H=360, deg=0, seed=4, scale_mul=2.0, squeeze_xy=0.15),
             "lists_1k_4k": dict(P=40000, W=640, H=360, deg=0, seed=3, scale_mul=2.0, squeeze_xy=0.25),
             "c2_k2": dict(P=100000, W=800, H=800, deg=1, seed=0),
             "tiles_4096_k2": dict(P=150000, W=1024, H=1024, deg=1, seed=5, scale_mul=1.5),
             "tiles_8160_k4": dict(P=200000, W=1920, H=1080, deg=0, seed=6, scale_mul=1.5),
             "all_long_lists": dict(P=30000, W=20, H=20, deg=0, seed=8, scale_mul=30.0),
             "mostly_culled": dict(P=3000, W=160, H=96, deg=1, seed=9, near_fraction=1.0),
             "few_tiles": dict(P=500, W=40, H=24, deg=2, seed=10, scale_mul=3.0),
             # every Gaussian on one of two depth values: the bucket-and-rank sort declines every segment of more than 64
             # keys, so each form's NETWORK sorts them -- in k_bucket_sort's 512- / 1024- / 2048-key areas (one wave per
             # tile), its long-tile path, and the chain's small / mid / wide classes
             "piled_depths_1k_per_tile": dict(P=60000, W=320, H=192, deg=0, seed=11, scale_mul=3.0, z_levels=2),
             "piled_depths_300_per_tile": dict(P=40000, W=480, H=270, deg=1, seed=12, scale_mul=1.5, z_levels=3)}
out = {}
for name, kw in CASES.items():
    kw = dict(kw)
    z_levels = kw.pop("z_levels", 0)
    c = Hh.make_case(**kw)
    if z_levels:   # (scene A's camera looks down +z from the origin: view depth = z)
        import torch
        zm = float(c.means3D[:, 2].median())
        c.means3D[:, 2] = torch.linspace(0.8 * zm, 1.2 * zm, z_levels)[torch.arange(c.P) % z_levels]
    for dg in ((False,) if "--plans" in sys.argv else (False, True)):
        r = Hh.run_hip(c, depth_gradient=dg)
        d = {"color": r.color, "depth": r.depth, "radii": r.radii}
        d.update({"grad_" + k: getattr(r.grads, k) for k in Hh.GRAD_KEYS if getattr(r.grads, k) is not None})
        out[name + ("+depth" if dg else "")] = {k: hashlib.sha256(v.tobytes()).hexdigest()[:16] for k, v in d.items()}
print(json.dumps(out))
