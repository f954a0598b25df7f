#!/usr/bin/env python3
"""Config C5 of BASELINE.json: Gaussian-count sweep (1..5 M, SH deg 3, 1920x1080, fwd+bwd) recording
per-stage milliseconds and the algorithmic HBM GB/s of every stage.  Prints one JSON
line per point.  `rocprofv3 --pmc` passes over the same points: tools/pmc_passes.sh <out> --config c5."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

for P in (1_000_000, 2_000_000, 3_000_000, 5_000_000):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c5", "--gaussians", str(P),
                          "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-c4"], capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(json.dumps({"gaussians": P, "error": out.stderr[-300:]}))
        continue
    d = json.loads(line[-1])
    R, N, M = d["config"]["num_rendered"], 1920 * 1080, 16
    alg = bench.algorithmic_bytes(P, M, R, N)
    gbs = {k: round(alg.get(k, 0) / (v * 1e-3) / 1e9, 1) for k, v in d["stage_ms"].items() if v > 0}
    print(json.dumps({"gaussians": P, "num_rendered": R, "Msplats_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                      "stage_ms": d["stage_ms"], "stage_algorithmic_GBps": gbs}), flush=True)
