#!/usr/bin/env python3
"""VERDICT r5 item 7, the one bounded experiment: could the backward of a long tile be cut into segments walked in
parallel, each restarted from forward checkpoints (T and the prefix colour at the segment's end)?  accum_rec would then
come from a subtraction, (C_final - C_prefix) / T_prefix, not from the reference's recurrence (backward.cu:527-536).
This measures, on the CPU oracle alone (oracle/bsr_oracle.c: bsro_set_backward_segment), how far the nine per-Gaussian
sums of such a walk land from the reference's, against the stage-A bound of tests/test_parity_gpu.py
(|err| <= 1e-4 |ref| + 256 eps sum|term| + 1e-6 max|ref|).   python tools/segment_backward_experiment.py [case ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    "dense": dict(P=200_000, W=960, H=540, deg=1, seed=4, scale_mul=4.0),
    "lists": dict(P=20_000, W=48, H=48, deg=1, seed=7, scale_mul=12.0),
    "c3_small": dict(P=120_000, W=640, H=360, deg=1, seed=0, scale_mul=3.0),
}
EPS32 = float(np.finfo(np.float32).eps)


def nine(g):
    return [g.dL_dmeans2D[:, 0], g.dL_dmeans2D[:, 1], g.dL_dconic.reshape(-1, 4)[:, 0], g.dL_dconic.reshape(-1, 4)[:, 1],
            g.dL_dconic.reshape(-1, 4)[:, 3], g.dL_dopacity[:, 0], g.dL_dcolors[:, 0], g.dL_dcolors[:, 1], g.dL_dcolors[:, 2]]


def main():
    for name in (sys.argv[1:] or ["lists", "dense"]):
        c = Hh.make_case(**CASES[name])
        t0 = time.time()
        st, g = Hh.run_oracle(c, backward=True, want_abs_sums=True)
        S = g.abs_sums.astype(np.float64)
        ref = [x.astype(np.float64).copy() for x in nine(g)]
        n = np.diff(st.ranges, axis=1).reshape(-1)
        rec = {"case": name, **CASES[name], "num_rendered": int(st.num_rendered), "entries_per_tile_median": int(np.median(n)),
               "entries_per_tile_max": int(n.max()), "oracle_s": round(time.time() - t0, 1), "segments": {}}
        # the two floors beside it: binary32 sums in one fixed order, and the reference's terms with fp contraction
        for label, gg in (("f32_sums_fixed_order", O.backward(st, c.gC, c.gD, f32_sums=True)),):
            worst = 0.0
            for i, (a, b) in enumerate(zip(nine(gg), ref)):
                err = np.abs(a.astype(np.float64) - b)
                bound = 1e-4 * np.abs(b) + 256 * EPS32 * S[:, i] + 1e-6 * np.abs(b).max() + 1e-30
                worst = max(worst, float((err / bound).max()))
            rec[label + "_worst_err_over_bound"] = round(worst, 4)
        for seg in (64, 256, 1024):
            gs = O.backward(st, c.gC, c.gD, segment=seg)
            worst, where, beyond = 0.0, None, 0
            rel = []
            for i, (a, b) in enumerate(zip(nine(gs), ref)):
                err = np.abs(a.astype(np.float64) - b)
                bound = 1e-4 * np.abs(b) + 256 * EPS32 * S[:, i] + 1e-6 * np.abs(b).max() + 1e-30
                r = err / bound
                beyond += int((r > 1).sum())
                if float(r.max()) > worst:
                    worst, where = float(r.max()), i
                rel.append(float(err.max() / max(np.abs(b).max(), 1e-30)))
            rec["segments"][str(seg)] = {"worst_err_over_stage_A_bound": round(worst, 3), "component": where,
                                         "elements_beyond_bound": beyond, "max_err_over_scale": float("%.3g" % max(rel))}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
