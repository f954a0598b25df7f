#!/usr/bin/env python3
"""Where does a step's time go that is NOT inside a kernel?  Reads the kernel trace of

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/profile_bloomscene_shape.py

(any workload whose step starts with a fixed kernel works: --first names it) and prints, over the steady-state steps:
the step period, the sum of kernel durations, the idle time, and the idle intervals ranked by what precedes them.

    python tools/trace_gaps.py out [--first k_anchor_select] [--skip 20]
"""
import argparse
import collections
import csv
import glob
import json
import os


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n.replace("bsr::", "").replace("at::native::", "at::")[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--first", default="k_anchor_select", help="substring of the kernel that starts a step")
    ap.add_argument("--skip", type=int, default=20, help="steps to drop at both ends (warm-up, drain)")
    a = ap.parse_args()
    rows = []
    for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if a.first in r[2]]
    steps = [rows[starts[i]:starts[i + 1]] for i in range(len(starts) - 1)]
    steps = steps[a.skip:len(steps) - a.skip] if len(steps) > 2 * a.skip + 4 else steps
    period, busy, gaps = [], [], collections.defaultdict(list)
    for si, st in enumerate(steps):
        nxt_start = None
        # period = from this step's first kernel start to the next step's first kernel start
        idx = rows.index(st[0])
        nxt = rows[idx + len(st)][0] if idx + len(st) < len(rows) else st[-1][1]
        period.append((nxt - st[0][0]) / 1e3)
        busy.append(sum(e - s for s, e, _ in st) / 1e3)
        for k, (s, e, n) in enumerate(st):
            after = st[k + 1][0] if k + 1 < len(st) else nxt
            gaps[(k, n)].append(max(after - e, 0) / 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    out = {"steps_analysed": len(steps), "kernels_per_step": med([len(s) for s in steps]),
           "step_period_us_median": round(med(period), 1), "kernel_time_us_median": round(med(busy), 1),
           "idle_us_median": round(med([p - b for p, b in zip(period, busy)]), 1),
           "idle_share": round(1 - med(busy) / med(period), 4)}
    print(json.dumps(out))
    print("idle interval AFTER each kernel of the step (median us), largest first:")
    for (k, n), v in sorted(gaps.items(), key=lambda kv: -med(kv[1]))[:16]:
        print(f"  #{k:2d} {n:60s} {med(v):8.1f}")
    print("kernels of one step (median duration us):")
    dur = collections.defaultdict(list)
    for st in steps:
        for k, (s, e, n) in enumerate(st):
            dur[(k, n)].append((e - s) / 1e3)
    for (k, n), v in sorted(dur.items()):
        print(f"  #{k:2d} {n:60s} {med(v):8.1f}")


if __name__ == "__main__":
    main()
