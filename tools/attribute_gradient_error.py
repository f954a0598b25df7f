#!/usr/bin/env python3
"""Which arithmetic shortcut of k_render_bwd owns the distance between the HIP gradients and the f64 sums of the oracle?

The backward walk departs from the reference's per-pair operations (backward.cu:512-583) in five places:
  exp        v_exp_f32 outside the decision band instead of the pinned exp            (-DBSR_BWD_EXACT_EXP switches it off)
  div        v_rcp_f32 + one residual correction instead of two IEEE divisions        (-DBSR_BWD_IEEE_DIV)
  nocontract `#pragma clang fp contract(fast)` on the per-pair block                  (-DBSR_BWD_NO_CONTRACT)
  pairs      six moment sums recombined per entry instead of the per-pair products    (-DBSR_BWD_PAIR_PRODUCTS)
  channels   accum_rec projected on dL_dpixel (one scalar recurrence) instead of three channels  (-DBSR_BWD_CHANNEL_ACCUM)
`make -C bloomscene_amd/csrc attrib` builds the walk with each of them switched off alone and with all of them off
("exact").  For every build this tool runs tools/parity_report.py (the SURVEY §8(d) share of elements off by > 1e-4,
next to the reference's own f32-order-vs-f64 floor) and bench.py (ms of the render_bwd stage) on the GPU box and prints
one JSON line per build plus a markdown table:

    python tools/attribute_gradient_error.py [--cases c1 c3 c5] > profiles/<round>/gradient_error_attribution.md
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILDS = ["product", "exp", "div", "nocontract", "pairs", "channels", "exact"]
TENSORS = ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacities", "dL_dshs", "dL_dscales", "dL_drotations"]


def lib_path(build):
    name = "libbloomscene_rast.so" if build == "product" else f"libbsr_attrib_{build}.so"
    return os.path.join(ROOT, "bloomscene_amd", name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", nargs="+", default=["c1", "c3", "c5"])
    ap.add_argument("--builds", nargs="+", default=BUILDS)
    ap.add_argument("--no-timing", action="store_true")
    args = ap.parse_args()
    rows = {}
    for b in args.builds:
        lib = lib_path(b)
        if not os.path.exists(lib):
            print(f"<!-- {b}: {lib} missing (make -C bloomscene_amd/csrc attrib) -->")
            continue
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_report.py"), "--lib", lib, "--exact-exp"]
                           + args.cases, capture_output=True, text=True, cwd=ROOT, timeout=1500)
        recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or len(recs) != len(args.cases):
            print(f"<!-- {b}: parity_report failed: {r.stderr[-400:]} -->")
            continue
        ms = None
        if not args.no_timing:
            r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lib", lib, "--steps", "40", "--warmup",
                                 "10", "--no-cpu-baseline", "--no-c4", "--no-secondary"], capture_output=True, text=True,
                                cwd=ROOT, timeout=600)
            lines = [ln for ln in r2.stdout.splitlines() if ln.startswith("{")]
            if lines:
                ms = json.loads(lines[-1])["stage_ms"].get("render_bwd")
        rows[b] = {"render_bwd_ms_c3": ms, "cases": {rec["case"]: rec["tensors"] for rec in recs}}
        print("<!-- " + json.dumps({"build": b, **rows[b]}) + " -->", flush=True)
    for case in args.cases:
        print(f"\n### {case}: share of elements with relative error > 1e-4 against the oracle's f64 sums "
              f"(in units of the reference's own f32-order floor)\n")
        print("| build | render_bwd ms (C3) | " + " | ".join(t[4:] for t in TENSORS) + " |")
        print("|---|---|" + "---|" * len(TENSORS))
        floor = None
        floor2 = None
        for b, row in rows.items():
            ten = row["cases"].get(case, {})
            cells = []
            for t in TENSORS:
                if t not in ten:
                    cells.append("-")
                    continue
                f, fl = ten[t]["frac_above_1e-4"], ten[t]["reference_f32_order_vs_f64"]["frac_above_1e-4"]
                cells.append(f"{f:.2e} ({f / fl:.1f}x)" if fl > 0 else f"{f:.2e}")
                floor = floor or {}
                floor[t] = fl
                if "reference_contracted_vs_source_order" in ten[t]:
                    floor2 = floor2 or {}
                    floor2[t] = ten[t]["reference_contracted_vs_source_order"]["frac_above_1e-4"]
            ms = row["render_bwd_ms_c3"]
            print(f"| {b} | {ms if ms is not None else '-'} | " + " | ".join(cells) + " |")
        if floor:
            print("| *floor: reference f32 order vs f64* | | " + " | ".join(f"{floor.get(t, 0):.2e}" for t in TENSORS) + " |")
        if floor2:
            print("| *reference with fp contraction (nvcc default) vs without, f64 sums* | | "
                  + " | ".join(f"{floor2.get(t, 0):.2e}" for t in TENSORS) + " |")


if __name__ == "__main__":
    main()
