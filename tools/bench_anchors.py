#!/usr/bin/env python3
"""Measures the fused anchor expansion on the MI355X against the same lines
of the reference (gaussian_renderer/__init__.py:165-203) run as eager torch ops on the same GPU,
and against the CPU oracle on the host cores.

    python tools/bench_anchors.py [--anchors 500000] [--offsets 10] [--steps 20]

Workload: N visible anchors x K offsets (C5-like default: 500 k x 10 = 5 M candidates, ~half of
them selected).  One step = forward + backward with upstream gradients on all five outputs.
ALGORITHMIC bytes per step (every tensor read or written once):
  forward  : candidates * (4 + 12 + 28 + 12) + anchors * 36 + selected * 56 + candidates (mask)
  backward : candidates * (4 + 28 + 12) + anchors * 24 + selected * 56 + candidates * 56 + anchors * 36
Prints one JSON line.
"""
from __future__ import annotations

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
    ap.add_argument("--anchors", type=int, default=500_000)
    ap.add_argument("--offsets", type=int, default=10)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-anchors", type=int, default=100_000)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs the GPU"
    dev = torch.device("cuda")
    from bloomscene_amd.neural_gaussians import expand_anchors
    from oracle import anchors as OA   # measurement tool: the oracle is the timed CPU/eager baseline only

    N, K = args.anchors, args.offsets
    inp = OA.synthetic_anchor_inputs(N, K, seed=0)
    S = int((inp[3] > 0).sum())
    g = torch.Generator().manual_seed(1)
    upstream = [torch.randn(S, w, generator=g).to(dev) for w in (3, 3, 1, 3, 4)]
    leaves = [t.to(dev).requires_grad_(True) for t in inp]

    def step(fn):
        for l in leaves:
            l.grad = None
        out = fn(*leaves)
        torch.autograd.backward(list(out[:5]), upstream)

    def timed(fn, steps):
        for _ in range(args.warmup):
            step(fn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(fn)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    t_fused = timed(expand_anchors, args.steps)
    t_eager = timed(OA.expand_anchors_reference, max(3, args.steps // 4))
    # forward only
    def fwd_only(fn, steps):
        with torch.no_grad():
            for _ in range(args.warmup):
                fn(*leaves)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn(*leaves)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    f_fused = fwd_only(expand_anchors, args.steps)
    f_eager = fwd_only(OA.expand_anchors_reference, max(3, args.steps // 4))

    cand = N * K
    fwd_bytes = cand * 56 + N * 36 + S * 56 + cand
    bwd_bytes = cand * 44 + N * 24 + S * 56 + cand * 56 + N * 36
    # CPU: the oracle (torch CPU ops, all host threads) on a bounded sample
    Nc = min(N, args.cpu_anchors)
    cin = [t.requires_grad_(True) for t in OA.synthetic_anchor_inputs(Nc, K, seed=0)]
    Sc = int((cin[3] > 0).sum())
    cup = [torch.randn(Sc, w, generator=g) for w in (3, 3, 1, 3, 4)]
    t0 = time.perf_counter()
    out = OA.expand_anchors_reference(*cin)
    torch.autograd.backward(list(out[:5]), cup)
    t_cpu = time.perf_counter() - t0
    print(json.dumps({
        "workload": f"anchor expansion fwd+bwd: {N} anchors x {K} offsets = {cand} candidates, {S} selected",
        "fused_ms": round(t_fused * 1e3, 4), "torch_eager_gpu_ms": round(t_eager * 1e3, 4),
        "speedup_vs_eager": round(t_eager / t_fused, 2),
        "fused_fwd_ms": round(f_fused * 1e3, 4), "torch_eager_gpu_fwd_ms": round(f_eager * 1e3, 4),
        "value": round(cand / t_fused / 1e6, 1), "unit": "Mcandidates/s",
        "roofline": {"bound": "hbm", "algorithmic_bytes": fwd_bytes + bwd_bytes,
                     "achieved": round((fwd_bytes + bwd_bytes) / t_fused / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round((fwd_bytes + bwd_bytes) / t_fused / 8e12, 4),
                     "note": "wall clock of python fwd+bwd incl. the 4-byte count read and allocations"},
        "cpu_baseline": {"value": round(Nc * K / t_cpu / 1e6, 2), "unit": "Mcandidates/s",
                         "cores": torch.get_num_threads(), "kind": "port",
                         "sample": f"{Nc} anchors x {K}, torch CPU ops fwd+bwd, {t_cpu:.2f} s"}}), flush=True)


if __name__ == "__main__":
    main()
