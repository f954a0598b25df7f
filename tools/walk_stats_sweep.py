#!/usr/bin/env python3
"""Forward-walk statistics of ONE view-batched call of the C4 sweep (16 neighbouring views of the 64-view rotate360 path
over 1 M Gaussians, rendered from the rows their visibility filter kept), from the diagnostic build
(`make -C bloomscene_amd/csrc stats`): entries per tile, list visits, blending lanes, and the per-workgroup timeline.

    python3 tools/walk_stats_sweep.py [--lib bloomscene_amd/libbsr_rast_stats.so] [--views 16]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=16)
    ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "bloomscene_amd", "libbsr_rast_stats.so"))
    a = ap.parse_args()
    from bloomscene_amd import _capi, views
    _capi.use_library(a.lib)
    from bloomscene_amd.synthetic import scene_b
    from walk_stats import timeline
    lib = _capi.lib()
    fnf = lib.bsr_debug_walk_stats_fwd
    fnf.restype = C.c_int
    fnf.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
    dev = torch.device("cuda:0")
    W, H = 1920, 1080
    sc = scene_b(1_000_000, W, H, 3, n_views=64, seed=0)
    bufs = {k: getattr(sc, k).to(dev) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    pack = views.CameraPack([c.to(dev) for c in sc.cameras], dev)
    idx = list(range(a.views))
    sub = views.compact_for_view_groups(pack, bufs, [idx])[0]
    bg = torch.zeros(3, device=dev)
    fstats = np.zeros(24, dtype=np.uint64)
    for it in range(3):
        if it == 2:
            torch.cuda.synchronize()
            assert fnf(2, fstats.ctypes.data, fstats.nbytes) == 0   # clears the warm-up calls' counts
        views.render_views_batched(pack, sub, bg, 3, idx=idx)
    torch.cuda.synchronize()
    assert fnf(2, fstats.ctypes.data, fstats.nbytes) == 0
    T = ((W + 15) // 16) * ((H + 15) // 16) * a.views
    ftl = np.zeros(4 * 70000, dtype=np.uint64)
    assert fnf(3, ftl.ctypes.data, ftl.nbytes) == 0
    ftl = ftl.reshape(-1, 4)
    ftl = ftl[ftl[:, 1] > 0]
    f = fstats.astype(np.float64)
    n_in_tile = (ftl[:, 3] >> np.uint64(32)).astype(np.int64)
    out = {"views": a.views, "tiles": T, "rows": int(sub["means3D"].shape[0]),
           "entries_staged_per_tile": f[7] / 4 / T,
           "trips_of_four_per_wave": f[0] / 4 / (4 * T),
           "visits_per_staged_entry": f[0] / max(f[7] / 4, 1),
           "share_of_visits_with_a_candidate": f[1] / max(f[0], 1),
           "share_of_visits_blending": f[3] / max(f[0], 1),
           "blending_lanes_per_blending_visit": f[5] / max(f[3], 1),
           "entries_per_tile_histogram_0_1-16_17-32_33-64_65-256_more": [
               int((n_in_tile == 0).sum()), int(((n_in_tile > 0) & (n_in_tile <= 16)).sum()),
               int(((n_in_tile > 16) & (n_in_tile <= 32)).sum()), int(((n_in_tile > 32) & (n_in_tile <= 64)).sum()),
               int(((n_in_tile > 64) & (n_in_tile <= 256)).sum()), int((n_in_tile > 256).sum())],
           "timeline_first_70000_workgroups": timeline(ftl)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
