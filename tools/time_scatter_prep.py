#!/usr/bin/env python3
"""Where the source rank's share of views.scatter_visible_gaussians goes (C4 scene: 1 M Gaussians, 64-view rotate360
path, SH degree 3), piece by piece and for 1 / 2 / 4 / 8 ranks: the group filter (GPU clock, HIP events on torch's
stream, which is the stream the library is handed), the 4-byte-per-rank count read-back, the (rank, id) pair list, the
row packing.  One JSON line per node size; medians of --reps runs.

    python3 tools/time_scatter_prep.py [--gaussians 1000000] [--views 64] [--reps 20]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    from bloomscene_amd import views
    from bloomscene_amd.rasterizer import _pack_rows_native
    from bloomscene_amd.synthetic import scene_b
    dev = torch.device("cuda:0")
    sc = scene_b(args.gaussians, 1920, 1080, 3, n_views=args.views, seed=0)
    keys = ("means3D", "opacities", "rotations", "scales", "shs")
    bufs = {k: getattr(sc, k).to(dev) for k in keys}
    pack = views.CameraPack([c.to(dev) for c in sc.cameras], dev)

    def gpu_ms(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b), r

    def host_ms(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3, r

    med = lambda xs: sorted(xs)[len(xs) // 2]
    for world in (1, 2, 4, 8):
        groups = [views.assign_views(len(pack), r, world, "contiguous") for r in range(world)]
        rec = {k: [] for k in ("filter_gpu_ms", "filter_host_ms", "counts_readback_ms", "pairs_ms", "pack_gpu_ms",
                               "pack_host_ms")}
        for _ in range(args.reps + 2):
            f = lambda: views.group_visibility(pack, bufs["means3D"], bufs["scales"], bufs["rotations"], groups,
                                               return_counts=True)
            t_g, (masks, counts) = gpu_ms(f)
            t_h, _ = host_ms(f)
            t_c, cl = host_ms(lambda: counts.tolist())
            t_p, pairs = host_ms(lambda: torch.nonzero_static(masks, size=int(sum(cl))))
            g = lambda: _pack_rows_native([bufs[k] for k in keys], pairs.reshape(-1)[1:], idx_stride=2, rows=pairs.shape[0])
            t_kg, _ = gpu_ms(g)
            t_kh, _ = host_ms(g)
            for k, v in zip(rec, (t_g, t_h, t_c, t_p, t_kg, t_kh)):
                rec[k].append(v)
        out = {"ranks": world, "rows": int(sum(cl)), "rows_max": int(max(cl))}
        out.update({k: round(med(v[2:]), 4) for k, v in rec.items()})
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
