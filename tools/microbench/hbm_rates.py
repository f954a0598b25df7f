#!/usr/bin/env python3
"""Practical HBM rates of the box (torch kernels, no library code): fill (write only), reduction (read only), copy
(read + write), and a 2:1 read:write mix like k_preprocess's -- the ceiling the HBM-bound kernels are priced against
beside the nominal 8 TB/s.  One JSON line."""
import json, time, torch
dev = torch.device("cuda")
N = 256 * 1024 * 1024 // 4   # 256 MB of float32
a = torch.empty(N, device=dev); b = torch.empty(N, device=dev); c = torch.empty(2 * N, device=dev)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
out = {}
s = t(lambda: a.fill_(1.0)); out["fill_write_only_TBps"] = round(N * 4 / s / 1e12, 2)
s = t(lambda: b.copy_(a)); out["copy_read_plus_write_TBps"] = round(2 * N * 4 / s / 1e12, 2)
s = t(lambda: torch.sum(a)); out["sum_read_only_TBps"] = round(N * 4 / s / 1e12, 2)
s = t(lambda: torch.add(c[:N], c[N:], out=b)); out["add_2_reads_1_write_TBps"] = round(3 * N * 4 / s / 1e12, 2)
print(json.dumps(out))
