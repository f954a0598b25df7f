#!/usr/bin/env python3
"""Config C4 of BASELINE.json: 1 M Gaussians ("shell" scene B), SH deg 3, 1920x1080, a 64-view
rotate360 sweep rendered forward-only and sharded round-robin over the ranks after ONE RCCL
broadcast of the packed Gaussian buffers.

    python tools/bench_views.py                                   # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29511 tools/bench_views.py                  # 8 GPUs, one process per GPU

Prints one JSON line on rank 0: Msplats/s = views * P / t_wall (max over ranks), broadcast time,
per-rank view counts.  Not the driver's bench (that is ../bench.py); a measurement tool.
"""
from __future__ import annotations

import argparse
import json
import math
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
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1,
                    help="views per native call (bsr_forward_views; 1 = the reference's one call per view)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "needs the GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from bloomscene_amd import views
    from bloomscene_amd.synthetic import scene_b

    P, W, H, deg = args.gaussians, args.width, args.height, args.sh_degree
    M = (deg + 1) ** 2
    names = ("means3D", "scales", "rotations", "opacities", "shs")
    sc = scene_b(P if rank == 0 else 1, W, H, deg, n_views=args.views, seed=0)   # cameras on every rank
    if rank == 0:
        bufs = {k: getattr(sc, k).to(dev) for k in names}
    else:
        shapes = {"means3D": (P, 3), "scales": (P, 3), "rotations": (P, 4), "opacities": (P, 1), "shs": (P, M, 3)}
        bufs = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in names}
    bcast_ms = views.broadcast_gaussians(bufs, src=0)
    bg = torch.zeros(3, device=dev)
    cams = [c.to(dev) for c in sc.cameras]
    mine = views.shard_views(len(cams), rank, world)

    def sweep():
        return views.render_views_sharded(cams, bufs, bg, deg, rank=rank, world=world, batch=args.batch)

    sweep()   # warm-up (allocator, first-touch)
    from bloomscene_amd import _capi
    _capi.profile_enable(True)
    _capi.profile_reset()
    sweep()
    torch.cuda.synchronize()
    # mean per view (a batched call covers args.batch views)
    stage_ms = {k: round(v[0] / max(v[1], 1) / max(args.batch, 1), 4) for k, v in _capi.profile_read().items()}
    _capi.profile_enable(False)
    times = []
    for _ in range(args.repeats):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sweep()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        times.append(dt)
    # how much work a view is: visible Gaussians / instances of this rank's first view
    with torch.no_grad():
        res = views.render_view(cams[mine[0]], bufs, bg, deg)
        visible = int(res["visibility_filter"].sum().item())
    # anchor prefilter of the whole path (§8f rank 2): one call per view (reference shape) vs one batched call
    scales6 = torch.cat([bufs["scales"], bufs["scales"]], dim=1)
    mine_cams = [cams[i] for i in mine]

    def _time(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3 * 1e3
    pf_per_view = _time(lambda: [views.prefilter(c, bufs["means3D"], scales6, bufs["rotations"], bg) for c in mine_cams])
    pf_batched = _time(lambda: views.prefilter_views(mine_cams, bufs["means3D"], scales6, bufs["rotations"]))
    if rank == 0:
        t = sorted(times)[len(times) // 2]
        print(json.dumps({
            "workload": f"c4: {P} Gaussians scene B, SH deg {deg}, {W}x{H}, {args.views}-view rotate360 sweep, fwd only"
                        + (f", {args.batch} views per native call (bsr_forward_views)" if args.batch > 1 else ""),
            "views_per_call": args.batch,
            "n_gpus": world, "views": args.views, "views_rank0": len(mine), "seconds": round(t, 5),
            "ms_per_view": round(t / len(mine) * 1e3, 4), "value": round(args.views * P / t / 1e6, 2),
            "unit": "Msplats/s", "broadcast_ms": round(bcast_ms, 3), "visible_first_view": visible,
            "stage_ms_per_view_rank0": stage_ms,
            "prefilter_ms_rank0": {"per_view_calls": round(pf_per_view, 3), "one_batched_call": round(pf_batched, 3),
                                   "views": len(mine)}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
