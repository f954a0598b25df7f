#!/usr/bin/env python3
"""Lane utilisation and load balance of the FORWARD tile walk (k_render_fwd), from the diagnostic build
`make -C bloomscene_amd/csrc stats` (libbsr_rast_stats.so: per-wave counters + a per-workgroup timeline; the product
library carries neither).  (Until round 5 also of the backward's per-visit network walk, which is now the strict-gradient
path only; its tables: profiles/r03_*, r04_*/walk_stats_*.json.)  Run on the GPU box:

    python tools/walk_stats.py [--lib bloomscene_amd/libbsr_rast_stats.so] [--config c3] [--scale-mul 1.0]

Prints one JSON object: visits per staged entry, share of visits passing each vote, live lanes per reducing visit
(mean + histogram), and from the timeline: workgroup duration spread, concurrency over time, tail idle.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIGS = {"c2": (100_000, 800, 800, 1), "c3": (1_000_000, 1920, 1080, 3), "c5": (5_000_000, 1920, 1080, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--scale-mul", type=float, default=1.0, help="scene A with all scales multiplied (denser lists)")
    ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "bloomscene_amd", "libbsr_rast_stats.so"))
    ap.add_argument("--dump", default=None, help="with --timeline: also save the raw stamps as <prefix>_fwd.npy / _bwd.npy")
    ap.add_argument("--timeline", action="store_true",
                    help="`make timeline` build (libbsr_rast_timeline.so): start / end stamps of both walks AS SHIPPED, no "
                         "counters (the stats build's counters slow the forward ~10x and distort its stamps)")
    args = ap.parse_args()
    if args.timeline and args.lib.endswith("libbsr_rast_stats.so"):
        args.lib = args.lib.replace("libbsr_rast_stats.so", "libbsr_rast_timeline.so")
    from bloomscene_amd import GaussianRasterizationSettings, GaussianRasterizer, _capi
    _capi.use_library(args.lib)
    from bloomscene_amd.synthetic import scene_a, upstream_grads
    lib = _capi.lib()
    fnf = lib.bsr_debug_walk_stats_fwd   # AttributeError here = not the diagnostic build
    fnf.restype = C.c_int
    fnf.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
    P, W, H, deg = CONFIGS[args.config]
    dev = torch.device("cuda")
    sc = scene_a(P, W, H, deg, seed=0)
    cam = sc.cameras[0].to(dev)
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=deg, campos=cam.camera_center, prefiltered=False, debug=False)
    rast = GaussianRasterizer(rs)
    leaves = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    with torch.no_grad():
        leaves["scales"].mul_(args.scale_mul)
    gC, gD = upstream_grads(W, H, seed=1)
    gC, gD = gC.to(dev), gD.to(dev)
    fstats = np.zeros(24, dtype=np.uint64)
    for it in range(3):
        if it == 2 and not args.timeline:
            torch.cuda.synchronize()
            assert fnf(2, fstats.ctypes.data, fstats.nbytes) == 0   # clears what the warm-up forwards counted
        m2d = torch.zeros_like(leaves["means3D"], requires_grad=True)
        color, radii, depth = rast(means3D=leaves["means3D"], means2D=m2d, opacities=leaves["opacities"],
                                   shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"])
        torch.autograd.backward((color, depth), (gC, gD))
    torch.cuda.synchronize()
    T = ((W + 15) // 16) * ((H + 15) // 16)
    nblk = (T + 7) // 8 * 8
    out = {"config": args.config, "scale_mul": args.scale_mul, "tiles": T}
    ftl = np.zeros(4 * 70000, dtype=np.uint64)
    assert fnf(3, ftl.ctypes.data, ftl.nbytes) == 0
    ftl = ftl.reshape(-1, 4)[:nblk]
    ftl = ftl[ftl[:, 1] > 0]
    if args.timeline:
        fnb = lib.bsr_debug_walk_timeline_bwd
        fnb.restype = C.c_int
        fnb.argtypes = [C.c_void_p, C.c_size_t]
        btl = np.zeros(4 * 70000, dtype=np.uint64)
        assert fnb(btl.ctypes.data, btl.nbytes) == 0
        btl = btl.reshape(-1, 4)[:nblk]
        btl = btl[btl[:, 1] > 0]
        if args.dump:
            np.save(args.dump + "_fwd.npy", ftl)
            np.save(args.dump + "_bwd.npy", btl)
        out["build"] = "timeline (walks as shipped)"
        out["forward"] = {"timeline": timeline(ftl)}
        out["backward"] = {"timeline": timeline(btl)}
        print(json.dumps(out))
        return
    assert fnf(2, fstats.ctypes.data, fstats.nbytes) == 0
    f = fstats.astype(np.float64)
    out["forward"] = {
        "lists": "one per 8x4 half of a quadrant when tiles average >= 48 instances, else one per quadrant",
        "visits_per_staged_entry": f[0] / max(f[7] / 4, 1),
        "share_of_visits_with_a_candidate": f[1] / max(f[0], 1),
        "share_of_visits_blending": f[3] / max(f[0], 1),
        "blending_lanes_per_blending_visit": f[5] / max(f[3], 1),
        "blending_lane_histogram_1-8_..._57-64": (f[8:16] / max(f[3], 1)).round(4).tolist(),
        "timeline": timeline(ftl),
    }
    print(json.dumps(out))


def timeline(tl):
    """Per-workgroup start / end stamps (100 MHz) -> duration spread, residency over time, tail."""
    out = {}
    t0 = tl[:, 0].min()
    st = (tl[:, 0] - t0).astype(np.float64) / 100.0   # us
    en = (tl[:, 1] - t0).astype(np.float64) / 100.0
    dur = en - st
    total = en.max()
    n_in_tile = (tl[:, 3] >> np.uint64(32)).astype(np.float64)
    # concurrency curve: workgroups resident over time
    grid = np.linspace(0, total, 201)
    conc = [(float(((st <= g) & (en > g)).sum())) for g in grid]
    busy = float(dur.sum())
    slots = max(conc)
    return {
        "kernel_us": round(float(total), 1), "workgroups": int(len(tl)),
        "wg_duration_us": {"p5": round(float(np.percentile(dur, 5)), 1), "median": round(float(np.median(dur)), 1),
                           "p95": round(float(np.percentile(dur, 95)), 1), "max": round(float(dur.max()), 1)},
        "entries_per_tile": {"min": float(n_in_tile.min()), "median": float(np.median(n_in_tile)), "max": float(n_in_tile.max())},
        "peak_resident_workgroups": slots,
        "mean_resident_workgroups": round(busy / float(total), 1),
        "occupancy_of_peak": round(busy / float(total) / slots, 4),
        "time_below_half_peak_us": round(float(sum(1 for c in conc if c < 0.5 * slots)) * float(total) / 200, 1),
        "last_start_us": round(float(st.max()), 1),
        "resident_at_10pct_steps": [conc[i] for i in range(0, 201, 20)],
        "resident_at_5pct_steps_of_last_30pct": [conc[i] for i in range(140, 201, 10)],
        "corr_duration_entries": round(float(np.corrcoef(dur, n_in_tile)[0, 1]), 3) if len(dur) > 2 and n_in_tile.std() > 0 else None,
        "per_xcc_last_end_us": [round(float(en[(tl[:, 2] & np.uint64(0xf)) == np.uint64(x)].max()), 1)
                                for x in sorted(set(int(v & np.uint64(0xf)) for v in tl[:, 2]))],
        "per_xcc_busy_us": [round(float(dur[(tl[:, 2] & np.uint64(0xf)) == np.uint64(x)].sum()), 1)
                            for x in sorted(set(int(v & np.uint64(0xf)) for v in tl[:, 2]))],
        "per_xcc_workgroups": [int(((tl[:, 2] & np.uint64(0xf)) == np.uint64(x)).sum())
                               for x in sorted(set(int(v & np.uint64(0xf)) for v in tl[:, 2]))],
        "per_xcc_median_duration_us": [round(float(np.median(dur[(tl[:, 2] & np.uint64(0xf)) == np.uint64(x)])), 2)
                                       for x in sorted(set(int(v & np.uint64(0xf)) for v in tl[:, 2]))],
        "per_xcc_mean_entries": [round(float(n_in_tile[(tl[:, 2] & np.uint64(0xf)) == np.uint64(x)].mean()), 1)
                                 for x in sorted(set(int(v & np.uint64(0xf)) for v in tl[:, 2]))],
        "per_xcc_first_tile": [int((tl[(tl[:, 2] & np.uint64(0xf)) == np.uint64(x), 3] & np.uint64(0xffffffff)).min())
                               for x in sorted(set(int(v & np.uint64(0xf)) for v in tl[:, 2]))],
        "per_xcc_cus_seen": [int(len(set((tl[(tl[:, 2] & np.uint64(0xf)) == np.uint64(x), 2] >> np.uint64(32)).astype(np.int64) & 0x7f00 )))
                             for x in sorted(set(int(v & np.uint64(0xf)) for v in tl[:, 2]))],
        "duration_by_start_decile_us": [round(float(np.median(dur[(st >= np.percentile(st, 10 * i)) & (st <= np.percentile(st, 10 * i + 10))])), 1)
                                        for i in range(10)],
        "xcc_ids_seen": sorted(set(int(x & np.uint64(0xf)) for x in tl[:, 2])),
    }


if __name__ == "__main__":
    main()
