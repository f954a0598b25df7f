#!/usr/bin/env python3
"""The C4 sweep alone (1 M Gaussians scene B, 64-view rotate360 path, 1920x1080, SH degree 3, forward only), for kernel
traces: `rocprofv3 --kernel-trace --stats -- python3 tools/profile_c4_sweep.py [--compact] [--batch 16] [--reps 10]`."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--compact", action="store_true")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--shift-y", type=float, default=0.0, help="move every Gaussian up by this much (1e4: no view sees "
                                                                 "anything -- the empty-tile floor of the sweep)")
    ap.add_argument("--lib", default="", help="another build of the same C ABI (A/B and knock-out runs)")
    a = ap.parse_args()
    from bloomscene_amd import _capi, views
    if a.lib:
        _capi.use_library(a.lib)
    from bloomscene_amd.synthetic import scene_b
    dev = torch.device("cuda:0")
    sc = scene_b(a.gaussians, 1920, 1080, 3, n_views=a.views, seed=0)
    bufs = {k: getattr(sc, k).to(dev) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    if a.shift_y:
        bufs["means3D"][:, 1] += a.shift_y
    cams = [c.to(dev) for c in sc.cameras]
    pack = views.CameraPack(cams, dev)
    bg = torch.zeros(3, device=dev)
    ts = []
    for _ in range(a.reps + 2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        views.render_views_sharded(cams if a.batch == 1 else pack, bufs, bg, 3, rank=0, world=1, batch=a.batch,
                                   compact=a.compact)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    print("sweep ms (median of %d): %.3f" % (a.reps, sorted(ts[2:])[a.reps // 2]))


if __name__ == "__main__":
    main()
