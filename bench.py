#!/usr/bin/env python3
"""Benchmark of the rasterizer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: under torch.distributed.run, or plain -- the parent
                                                            then starts the N ranks itself as child processes)

Headline metric (BASELINE.json): Msplats/s forward+backward @ 1 M Gaussians, 1920x1080, SH degree 3
(config C3, synthetic scene A of SURVEY.md §8d), colour + depth targets with upstream gradients.
One "step" = one `GaussianRasterizer` forward + `torch.autograd.backward` through the C ABI, inputs
resident in HBM.  N > 1: view-parallel -- rank 0 broadcasts the Gaussian buffers over RCCL once
(outside the timed region), every rank then renders its own camera view; no data-path collective
(weak scaling).

Rank 0 prints ONE compact JSON line (< 4 KB, the LAST line of stdout: what the driver parses) with the bench contract
keys plus

  "config":       the workload in words and numbers, the hash of the kernel sources, the mode flags, median / first /
                  max step time and the host's slowest step
  "roofline":     dominant kernel, ALGORITHMIC bytes per launch / mean launch time (hipEvents recorded by the
                  library on the launch stream during the timed region) vs 8 TB/s; "traffic" = HBM bytes per
                  launch from the committed rocprofv3 --pmc passes ("traffic_profile"), only when those passes were
                  taken on the same kernel sources as the library being timed, else null
  "roofline_step", "stage_ms": the whole step against 8 TB/s; mean time of every stage (untimed pass, all stages bracketed)
  "cpu_baseline": the CPU oracle (a port of the reference algorithm, OpenMP) on the same workload, N = 1 only
  "c4":           summary of BASELINE config C4 -- the 64-view rotate360 sweep of scene B (1 M Gaussians, forward
                  only), STRONG-scaled over the N ranks: sweep time with the Gaussians resident, and cold (one packed
                  RCCL broadcast first); the visible-subset distribution beside it
  "secondary_Msplats_per_s": N = 1 only, never the headline: a dense scene A (long tile lists), the camera changing
                  every step, BloomScene's real call shape (512^2, anchors x 10 through the fused expansion,
                  colors_precomp, sh_degree 1), and the headline workload in the two bit-for-bit modes
                  (strict gradients, exact exp)
  "detail":       path of the DETAIL record (JSON) holding everything else: per-step arrays, the whole C4 record, every
                  secondary leg with its stage table.  --full adds the long legs (180 / 720-view presets, capacity-mode
                  and HIP-graph legs) and echoes the detail record on stderr; --experimental adds the C4 latency model.
"""
from __future__ import annotations

import argparse
import gc
import hashlib
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec

CONFIGS = {
    # name: (P, W, H, sh_degree, backward)
    "c2": (100_000, 800, 800, 1, False),
    "c3": (1_000_000, 1920, 1080, 3, True),
    "c5": (5_000_000, 1920, 1080, 3, True),
}


def algorithmic_bytes(P, M, R, N):
    """Per-stage ALGORITHMIC bytes of one launch (SURVEY.md §8d derivation)."""
    return {
        "preprocess": (44 + 12 * M + 75) * P,
        "scan_wg": 8 * (P // 256 + 1),
        "binning": 20 * P + 12 * R,
        "sort_tiles": 24 * R,
        "render_fwd": 4 * R + 40 * P + 24 * N,
        "render_bwd": 4 * R + 40 * P + 20 * N + 36 * P,
        "preprocess_bwd": (56 + 36 + 107 + 12 * M + 40 + 12 * M) * P,
    }


def step_bytes(P, M, R, N, backward):
    if backward:
        return (502 + 36 * M) * P + 52 * R + 44 * N
    return (187 + 12 * M) * P + 48 * R + 24 * N


def csrc_sha256():
    """Hash of the kernel sources the timed library is built from (csrc/*.hip, *.h, Makefile, include/*.h): what ties a
    committed counter profile to a build.  (The .so itself is rebuilt on every box and never committed.)"""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "bloomscene_amd", "csrc")
    files = sorted(os.path.join(src, f) for f in os.listdir(src) if f.endswith((".hip", ".h")) or f == "Makefile")
    files += sorted(os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(config, stage):
    """(HBM bytes per launch, VALU instructions per SIMD and cycle, source) of `stage` from the committed rocprofv3 --pmc
    passes (profiles/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate passes, FETCH_SIZE doubled as the
    MI355X guide prescribes for wide coalesced reads on gfx950) -- or Nones when those passes were taken on other
    kernel sources than the ones being timed now."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        # kind "committed": the figure is a constant read from a profile the builder committed, taken on ANOTHER run of the
        # same kernel sources -- NOT a measurement of the run this line reports (rocprofv3 --pmc cannot run inside it)
        src = {"kind": "committed", "profile": d.get("profile"), "csrc_sha256": d.get("csrc_sha256")}
        if d.get("csrc_sha256") != csrc_sha256():
            return None, None, dict(src, matches_timed_build=False)
        e = d[config][stage]
        return e.get("hbm_bytes"), e.get("valu_insts_per_simd_cycle"), dict(src, matches_timed_build=True)
    except (OSError, KeyError, ValueError):
        return None, None, None


def self_launch(gpus):
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as CHILD processes under
    torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and leave with their exit code.  Nothing in this
    parent has touched the GPU (importing torch does not), and it never execs: rank 0 of the children prints the
    JSON line on the inherited stdout."""
    import socket
    import subprocess
    # (counting devices does not initialise the GPU on this image; everything that does happens in the children)
    have = torch.cuda.device_count()
    if have < gpus and os.environ.get("BSR_BENCH_SINGLE_DEVICE") != "1":
        raise SystemExit(f"bench.py --gpus {gpus}: this box shows {have} GPU(s) -- one rank per GPU is needed "
                         f"(the multi-rank control flow on ONE GPU: BSR_BENCH_SINGLE_DEVICE=1 BSR_BENCH_BACKEND=gloo)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode)


class Dist:
    """RANK / WORLD_SIZE plumbing; BSR_BENCH_FORCE_DIST=1 takes the N > 1 code path (RCCL init, broadcast, barrier,
    all-reduce) with one rank."""

    def __init__(self, gpus):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != gpus:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={self.world}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback in bloomscene_amd)")
        # Test hooks (tests/test_bench_gpu.py): BSR_BENCH_SINGLE_DEVICE=1 puts every rank on GPU 0 and
        # BSR_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device), so the N > 1 control flow --
        # rank-0 generation, broadcasts, view sharding, max-over-ranks timing -- can run with 2 ranks on a 1-GPU box.
        index = 0 if os.environ.get("BSR_BENCH_SINGLE_DEVICE") == "1" else self.local_rank
        torch.cuda.set_device(index)
        self.dev = torch.device("cuda", index)
        self.force = os.environ.get("BSR_BENCH_FORCE_DIST") == "1" and "MASTER_PORT" in os.environ
        self.multi = self.world > 1 or self.force
        if self.multi:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("BSR_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm
            # a rank that waits for a peer which has failed gives up after five minutes, not after the default thirty
            import datetime
            patience = datetime.timedelta(seconds=300)
            if backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev,
                                        timeout=patience)
            else:
                dist.init_process_group(backend, rank=self.rank, world_size=self.world, timeout=patience)

    def fence(self):
        if self.multi:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if not self.multi:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_ints(self, x):
        if not self.multi:
            return [int(x)]
        import torch.distributed as dist
        t = torch.tensor([int(x)], dtype=torch.int64, device=self.dev)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return [int(o.item()) for o in out]

    def close(self):
        if self.multi:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


def timed_steps(D, step, steps, warmup, stage_events=True, dominant=None, prewarm_ms=0.0):
    """The contract's timing: W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides, MAX over
    ranks.  Also one torch event per step (on the launch stream) for the median step time.  -> dict.

    Inside the timed region only the DOMINANT stage (`dominant`, e.g. "render_bwd") is bracketed by hipEvents, on every
    4th step: an event pair costs ~35 us of pipeline bubble on MI355X, and bracketing all seven stages of every 4th
    step made the sampled steps 0.26 ms (20 %) longer -- 5 % off the very throughput being measured.  The stage table
    comes from a short untimed pass after the timed region, every stage of every step bracketed."""
    from bloomscene_amd import _capi
    # the collector runs BEFORE the clock ramp, not between it and the timed region: a full collection is tens of
    # milliseconds of idle GPU, after which the first ten timed steps ran up to 30 % slow (1.68 -> 1.25 ms at C3)
    gc.collect()
    gc.disable()   # no collector pauses inside the timed region (the steps create no reference cycles)
    prewarm_steps = 0
    if prewarm_ms > 0:
        # clock ramp: the same number of untimed steps on every rank (a step may hold a collective), sized from four
        # timed ones to fill prewarm_ms
        for _ in range(4):   # cold: first touch, allocator growth, code-object loads
            step()
        torch.cuda.synchronize()
        t_pre = time.perf_counter()
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        est_ms = max((time.perf_counter() - t_pre) * 1e3 / 4, 1e-3)
        n_more = int(D.max_over_ranks(min(math.ceil(prewarm_ms / est_ms), 2000)))
        for i in range(n_more):
            step()
            if i % 8 == 7:
                torch.cuda.synchronize()   # bound the queue: the loop is paced by the GPU, not by enqueue speed
        torch.cuda.synchronize()
        prewarm_steps = 8 + n_more
    events_on = stage_events and os.environ.get("BSR_BENCH_NO_STAGE_EVENTS") != "1"
    sample_every = 4 if steps >= 8 else 1
    _capi.profile_only(dominant)
    if events_on and warmup > 0:
        # the library creates its hipEvents on first use (milliseconds): let the warm-up steps do that, the timed
        # region then only recycles them
        _capi.profile_enable(1)
    for _ in range(warmup):
        step()
    _capi.profile_enable(sample_every if events_on else 0)
    _capi.profile_reset()   # (waits for the warm-up steps' events)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    host_marks = []
    allocs0 = torch.cuda.memory_stats(D.dev).get("num_device_alloc", 0)
    D.fence()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        step()
        marks[i + 1].record()
        host_marks.append(time.perf_counter())
    D.fence()
    dt = time.perf_counter() - t0
    gc.enable()
    device_allocs = torch.cuda.memory_stats(D.dev).get("num_device_alloc", 0) - allocs0
    host_step_ms = [1e3 * (b - a) for a, b in zip([t0] + host_marks[:-1], host_marks)]
    in_order = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    per_step = sorted(in_order)
    prof_timed = _capi.profile_read()
    _capi.profile_enable(False)
    _capi.profile_only(None)
    prof = dict(prof_timed)
    if events_on and dominant is not None:
        # the stage table: an untimed pass with every stage bracketed (its step time is not reported anywhere)
        _capi.profile_enable(1)
        _capi.profile_reset()
        for _ in range(min(12, max(steps, 1))):
            step()
        D.fence()
        table = _capi.profile_read()
        _capi.profile_enable(False)
        prof = dict(table)
        if dominant in prof_timed:
            prof[dominant] = prof_timed[dominant]   # the dominant stage: as measured inside the timed region
    return {"seconds": D.max_over_ranks(dt), "median_ms": per_step[len(per_step) // 2], "prof": prof,
            "step_ms": [round(x, 4) for x in in_order] if steps <= 64 else None,
            "host_step_ms": [round(x, 4) for x in host_step_ms] if steps <= 64 else None,
            "step_ms_first": in_order[0], "step_ms_max": per_step[-1],
            "steps_over_1p15x_median": sum(1 for x in in_order if x > 1.15 * per_step[len(per_step) // 2]),
            "host_median_ms": sorted(host_step_ms)[len(host_step_ms) // 2],
            # time from the fence to the first timed kernel reaching the GPU is inside step_ms[0]; the tail between the last
            # step's event and the closing fence: seconds - sum(step_ms)
            "tail_ms": dt * 1e3 - sum(in_order),
            "device_allocs": int(device_allocs), "max_host_ms": max(host_step_ms), "prewarm_steps": prewarm_steps,
            "timed_region_events": {"stage": dominant, "every_nth_step": sample_every,
                                    "launches": prof_timed.get(dominant, (0.0, 0))[1] if dominant else None}}


def raster_workload(D, args, P, W, H, deg, do_bwd, precomp=False, scale_mul=1.0, cycle_views=1, steps=None, warmup=None,
                    label="c3", allreduce=False, depth_gradient=False, mode="default"):
    """fwd(+bwd) steps of the rasterizer on scene A; returns the measurements of this rank (rank 0's are printed).
    ``mode``: "default" (the reference's call: the forward blocks on the read-back of num_rendered); "capacity"
    (GaussianRasterizer(capacity=1.25 R + 4096): BSR_FLAG_NO_READBACK, the host never waits); "graph" (the capacity-mode
    step captured once into a HIP graph and replayed).  Secondary legs only: the metric is quoted on "default"."""
    from bloomscene_amd import GaussianRasterizationSettings, GaussianRasterizer
    from bloomscene_amd.synthetic import scene_a, upstream_grads
    from bloomscene_amd.views import allreduce_gradients, broadcast_gaussians, yawed_camera
    dev = D.dev
    if precomp:
        deg = 1   # the reference passes sh_degree=1 with shs=None (gaussian_renderer/__init__.py:244,257)
    M = 0 if precomp else (deg + 1) ** 2
    N = W * H
    # ---- inputs: generated on rank 0's host, broadcast over RCCL/xGMI, resident in HBM ----
    names = ("means3D", "scales", "rotations", "opacities", "shs")
    if D.rank == 0:
        sc = scene_a(P, W, H, 0 if precomp else deg, seed=0)
        if precomp:
            sc.shs = torch.rand(P, 3, generator=torch.Generator().manual_seed(13))   # colours in [0, 1)
        sc.scales = sc.scales * scale_mul
        bufs = {k: getattr(sc, k).to(dev) for k in names}
    else:
        shapes = {"means3D": (P, 3), "scales": (P, 3), "rotations": (P, 4), "opacities": (P, 1),
                  "shs": (P, 3) if precomp else (P, M, 3)}
        bufs = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in names}
    bcast_ms = broadcast_gaussians(bufs, src=0, force=D.force) if D.multi else 0.0
    gC, gD = upstream_grads(W, H, seed=1)
    gC, gD = gC.to(dev), gD.to(dev)
    # independent views: rank r looks 0.25*r degrees to the side of the scene-A camera; cycle_views > 1: the camera
    # changes every step (+-1 degree steps), so the forward's scratch-size guess from the previous call can miss
    yaws = [0.25 * D.rank + (1.0 * k if cycle_views > 1 else 0.0) for k in range(cycle_views)]
    rasterizers = []
    for y in yaws:
        cam = yawed_camera(W, H, math.radians(60.0), yaw_deg=y).to(dev)
        st = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
            bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=deg, campos=cam.camera_center, prefiltered=False, debug=False)
        rasterizers.append(GaussianRasterizer(st, depth_gradient=depth_gradient))
    leaves = {k: v.requires_grad_(do_bwd) for k, v in bufs.items()}
    state = {"i": 0}
    if mode != "default":
        from bloomscene_amd.rasterizer import _rasterize_gaussians_native as _fwd0
        e0 = torch.Tensor([])
        s00 = rasterizers[0].raster_settings
        with torch.no_grad():
            R0 = _fwd0(s00.bg, bufs["means3D"], bufs["shs"] if precomp else e0, bufs["opacities"], bufs["scales"],
                       bufs["rotations"], 1.0, e0, s00.viewmatrix, s00.projmatrix, s00.tanfovx, s00.tanfovy, H, W,
                       e0 if precomp else bufs["shs"], deg, s00.campos, False, False)[0]
        for r_ in rasterizers:
            r_.capacity = int(1.25 * R0) + 4096

    def step():
        rasterizer = rasterizers[state["i"] % len(rasterizers)]
        state["i"] += 1
        means2D = torch.zeros_like(leaves["means3D"], requires_grad=do_bwd)
        color, radii, depth = rasterizer(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                         shs=None if precomp else leaves["shs"],
                                         colors_precomp=leaves["shs"] if precomp else None,
                                         scales=leaves["scales"], rotations=leaves["rotations"])
        if do_bwd:
            for v in leaves.values():
                v.grad = None
            torch.autograd.backward((color, depth), (gC, gD))
            if allreduce and D.multi:
                state["allreduce_ms"] = state.get("allreduce_ms", 0.0) + allreduce_gradients(leaves, force=D.force)
        state["radii"] = radii

    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    if mode == "graph":
        # the capacity-mode step holds no host wait and no call that is illegal during capture: warm it up on a side
        # stream (allocator, pinned buffer, events), capture it once, replay it
        from bloomscene_amd.rasterizer import check_deferred
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        check_deferred()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        eager_step, step = step, graph.replay
    tm = timed_steps(D, step, steps, warmup, dominant="render_bwd" if do_bwd else "render_fwd",
                     prewarm_ms=args.prewarm_ms,   # every leg: the GPU idles while the host builds the leg's scene, and a
                                                   # leg timed on clocks still ramping reads 4-5 % slow (docs/EXPERIMENTS.md)
                     stage_events=mode != "graph")
    if mode == "capacity":
        from bloomscene_amd.rasterizer import check_deferred
        check_deferred()   # (raises if a frame of the timed region overflowed its capacity)
    # instances of this rank's first view, for the algorithmic byte count
    from bloomscene_amd.rasterizer import _rasterize_gaussians_native
    e = torch.Tensor([])
    s0 = rasterizers[0].raster_settings
    with torch.no_grad():
        R = _rasterize_gaussians_native(s0.bg, bufs["means3D"], bufs["shs"] if precomp else e, bufs["opacities"],
                                        bufs["scales"], bufs["rotations"], 1.0, e, s0.viewmatrix, s0.projmatrix,
                                        s0.tanfovx, s0.tanfovy, H, W, e if precomp else bufs["shs"], deg, s0.campos,
                                        False, False)[0]
    visible = int((state["radii"] > 0).sum().item())
    ms_per_step = tm["seconds"] / steps * 1e3
    stages = {k: v[0] / max(v[1], 1) for k, v in tm["prof"].items()}  # mean ms per launch
    alg = algorithmic_bytes(P, M, R, N)
    if precomp:   # 12 B/G of colours read by preprocess, 12 B/G of colour gradient passed through
        alg["preprocess"] += 12 * P
        alg["preprocess_bwd"] += 24 * P
    sb = step_bytes(P, M, R, N, do_bwd) + (36 * P if precomp and do_bwd else 12 * P if precomp else 0)
    res = {"value": D.world * P * steps / tm["seconds"] / 1e6, "ms_per_step": ms_per_step,
           "ms_per_step_median": tm["median_ms"], "steps": steps, "warmup": warmup, "stages": stages, "alg": alg,
           "prof": tm["prof"], "R": R, "visible": visible, "step_bytes": sb, "bcast_ms": bcast_ms, "M": M, "deg": deg,
           "device_allocs": tm["device_allocs"], "max_host_ms": tm["max_host_ms"],
           "timed_region_events": tm["timed_region_events"], "prewarm_steps": tm["prewarm_steps"],
           "step_ms": tm["step_ms"], "host_step_ms": tm["host_step_ms"], "step_ms_first": tm["step_ms_first"],
           "step_ms_max": tm["step_ms_max"], "steps_over_1p15x_median": tm["steps_over_1p15x_median"],
           "host_median_ms": tm["host_median_ms"], "tail_ms": tm["tail_ms"],
           "allreduce_ms_per_step": state.get("allreduce_ms", 0.0) / max(steps + warmup, 1),
           "workload": f"{label}: {P} Gaussians, {'precomputed colours' if precomp else f'SH deg {deg}'}, {W}x{H}, "
                       f"{'fwd+bwd colour+depth targets' if do_bwd else 'fwd only'}"
                       f"{' WITH depth gradient (extension)' if depth_gradient and do_bwd else ''}, synthetic scene A seed 0"
                       + (f", all scales x{scale_mul:g}" if scale_mul != 1.0 else "")
                       + (f", camera changes every step ({cycle_views} views, 1 degree apart)" if cycle_views > 1 else "")
                       + ({"default": "", "capacity": "; capacity mode (BSR_FLAG_NO_READBACK: no host wait)",
                           "graph": "; the capacity-mode step replayed from a HIP graph"}[mode])}
    del leaves, bufs, rasterizers
    torch.cuda.empty_cache()
    return res


def secondary_line(r):
    """Compact record of a non-headline workload."""
    whole = r["step_bytes"] / (r["ms_per_step"] * 1e-3) / 1e9
    return {"workload": r["workload"], "value": round(r["value"], 2), "unit": "Msplats/s",
            "ms_per_step": round(r["ms_per_step"], 4), "ms_per_step_median": round(r["ms_per_step_median"], 4),
            "step_ms_first": round(r["step_ms_first"], 4), "step_ms_max": round(r["step_ms_max"], 4),
            "steps_over_1p15x_median": r["steps_over_1p15x_median"], "host_max_ms_per_step": round(r["max_host_ms"], 3),
            "steps": r["steps"], "num_rendered": r["R"], "instances_per_gaussian": round(r["R"] / max(r["visible"], 1), 2),
            "roofline_step_frac": round(whole / HBM_PEAK_GBS, 5),
            "stage_ms": {k: round(v, 4) for k, v in r["stages"].items()},
            "device_allocs_in_timed_region": r["device_allocs"]}


def bloomscene_shape_workload(D, args, n_anchor=100_000, n_offsets=10, W=512, H=512, mode="default"):
    """BloomScene's real call shape (arguments.py:8,102; gaussian_renderer/__init__.py:165-203,244-262): anchors x 10
    candidate Gaussians through the fused anchor expansion into the rasterizer with colors_precomp and sh_degree = 1,
    512 x 512, forward + backward w.r.t. all six head outputs.  Msplats/s counts the SELECTED Gaussians."""
    from bloomscene_amd import views
    from bloomscene_amd.synthetic import anchor_scene, upstream_grads
    dev = D.dev
    sc = anchor_scene(n_anchor, n_offsets, W, H, seed=0)
    cam = sc.camera.to(dev)
    names = ("anchor", "grid_scaling", "grid_offsets", "neural_opacity", "color", "scale_rot")
    leaves = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in names}
    gC, gD = upstream_grads(W, H, seed=1)
    gC, gD = gC.to(dev), gD.to(dev)
    bg = torch.zeros(3, device=dev)
    state = {}
    # "capacity": static shapes, no host wait (neural_gaussians.render_anchors(capacity=...): every per-Gaussian output
    # has n_anchor * n_offsets rows, the selection count stays on the device); "graph": that step replayed from a HIP graph
    settings = views.make_settings(cam, bg, 1)
    capacity = None
    if mode != "default":
        # the capacity a training loop would carry along: 1.25 x the instance count of a previous frame (here: probed once)
        from bloomscene_amd.neural_gaussians import expand_anchors
        from bloomscene_amd.rasterizer import _rasterize_gaussians_native
        with torch.no_grad():
            xyz, rgb, opac, scal, rot, _ = expand_anchors(*[leaves[k].detach() for k in names])
            e = torch.Tensor([])
            R0 = _rasterize_gaussians_native(settings.bg, xyz, rgb, opac, scal, rot, 1.0, e, settings.viewmatrix,
                                             settings.projmatrix, settings.tanfovx, settings.tanfovy, H, W, e, 1, settings.campos,
                                             False, False)[0]
        capacity = int(1.25 * R0) + 4096
        del xyz, rgb, opac, scal, rot

    def step():
        for v in leaves.values():
            v.grad = None
        res = views.render_neural(cam, *[leaves[k] for k in names], bg, capacity=capacity, settings=settings)
        torch.autograd.backward((res["render"], res["depth"]), (gC, gD))
        state["radii"] = res["radii"]
        state["mask"] = res["selection_mask"]

    steps = max(20, args.steps // 2)
    if mode == "graph":
        from bloomscene_amd.rasterizer import check_deferred
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        check_deferred()
        state.clear()          # (no output of an earlier iteration may outlive into the capture: torch's AccumulateGrad rule)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        step = graph.replay
    tm = timed_steps(D, step, steps, max(3, args.warmup // 2), stage_events=mode != "graph", prewarm_ms=args.prewarm_ms)
    if mode == "capacity":
        from bloomscene_amd.rasterizer import check_deferred
        check_deferred()
    S = int(state["mask"].sum().item())
    ms = tm["seconds"] / steps * 1e3
    return {"workload": f"bloomscene-shaped: {n_anchor} anchors x {n_offsets} offsets -> {S} selected Gaussians, fused "
                        f"expansion + rasterizer (colors_precomp, sh_degree 1), {W}x{H}, fwd+bwd to the six head outputs"
                        + {"default": "", "capacity": "; static shapes, no host wait (BSR_FLAG_NO_READBACK)",
                           "graph": "; the static-shape step replayed from a HIP graph"}[mode],
            "value": round(S * steps / tm["seconds"] / 1e6, 2), "unit": "Msplats/s (selected Gaussians)",
            "ms_per_step": round(ms, 4), "ms_per_step_median": round(tm["median_ms"], 4), "steps": steps,
            "visible": int((state["radii"] > 0).sum().item()),
            "stage_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in tm["prof"].items()},
            "device_allocs_in_timed_region": tm["device_allocs"]}


def c4_sweep(D, args, P=1_000_000, W=1920, H=1080, deg=3, n_views=64, repeats=5):
    """BASELINE config C4: the rotate360 sweep (bloomscene.py:191-193) of scene B, forward only, views dealt round-robin
    to the ranks (STRONG scaling: the 64 views are fixed) after ONE packed RCCL broadcast of the Gaussian buffers."""
    from bloomscene_amd import views
    from bloomscene_amd.synthetic import scene_b
    dev = D.dev
    M = (deg + 1) ** 2
    names = ("means3D", "scales", "rotations", "opacities", "shs")
    sc = scene_b(P if D.rank == 0 else 1, W, H, deg, n_views=n_views, seed=0)   # cameras on every rank
    if D.rank == 0:
        bufs = {k: getattr(sc, k).to(dev) for k in names}
    else:
        shapes = {"means3D": (P, 3), "scales": (P, 3), "rotations": (P, 4), "opacities": (P, 1), "shs": (P, M, 3)}
        bufs = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in names}
    D.fence()
    t0 = time.perf_counter()
    views.broadcast_gaussians(bufs, src=0, force=D.force) if D.multi else None
    D.fence()
    bcast_s = D.max_over_ranks(time.perf_counter() - t0) if D.multi else 0.0
    bg = torch.zeros(3, device=dev)
    cams = [c.to(dev) for c in sc.cameras]
    pack = views.CameraPack(cams, dev)   # the camera path is known before the sweep: matrices stacked on the device once
    mine = views.shard_views(len(cams), D.rank, D.world)
    out = {"workload": f"c4: {P} Gaussians scene B, SH deg {deg}, {W}x{H}, {n_views}-view rotate360 sweep, fwd only, "
                       f"views round-robin over {D.world} rank(s) after one packed RCCL broadcast",
           "scaling": "strong", "n_gpus": D.world, "views": n_views, "views_per_rank": D.gather_ints(len(mine)),
           "broadcast_ms": round(bcast_s * 1e3, 3), "broadcast_bytes": int(sum(v.numel() for v in bufs.values()) * 4),
           "unit": "Msplats/s"}
    with torch.no_grad():
        res = views.render_view(cams[mine[0]], bufs, bg, deg)
        out["visible_first_view_rank0"] = int(res["visibility_filter"].sum().item())
        # tile instances (the reference's num_rendered) of every view of this rank, for the algorithmic byte count
        from bloomscene_amd.rasterizer import _rasterize_gaussians_native
        e = torch.Tensor([])
        R_mine = 0
        for i in mine:
            st = views.make_settings(cams[i], bg, deg)
            R_mine += _rasterize_gaussians_native(st.bg, bufs["means3D"], e, bufs["opacities"], bufs["scales"],
                                                  bufs["rotations"], 1.0, e, st.viewmatrix, st.projmatrix, st.tanfovx,
                                                  st.tanfovy, H, W, bufs["shs"], deg, st.campos, False, False)[0]
    R_all = sum(D.gather_ints(R_mine))
    # SURVEY.md §8(d), forward only: (187 + 12 M) P + 48 R + 24 N per view.  The formula charges every view with all
    # P Gaussians' inputs, as the reference's preprocess reads them; on this sweep ~95 % of them are culled after their
    # 12-byte position, so the fraction says how fast the sweep is, not how busy the HBM was.
    alg_bytes = n_views * ((187 + 12 * M) * P + 24 * W * H) + 48 * R_all
    out["num_rendered_all_views"] = int(R_all)
    out["algorithmic_bytes_all_views"] = int(alg_bytes)
    # ... and what a sweep that renders every 16-view batch from the rows its own filter kept actually submits: per batch
    # the filter's read of positions, scales and rotations (40 B x P), the gather (236-byte rows read and written) and the
    # per-view term on the batch's rows only.  (Round 4 charged the compacted sweeps with all P rows per view: a
    # "fraction" above 1.)
    row_bytes_all = (3 + 3 + 4 + 1 + 3 * M) * 4
    batches16 = [mine[b0:b0 + 16] for b0 in range(0, len(mine), 16)]
    rows16 = [int(x) for x in views.group_visibility(pack, bufs["means3D"], bufs["scales"], bufs["rotations"], batches16,
                                                     return_counts=True)[1].tolist()] if batches16 else []
    alg_mine_compacted = sum(40 * P + 2 * row_bytes_all * r + len(b) * (187 + 12 * M) * r for b, r in zip(batches16, rows16)) \
        + len(mine) * 24 * W * H + 48 * R_mine
    alg_bytes_compacted = sum(D.gather_ints(alg_mine_compacted))
    out["rows_submitted_per_16_view_batch_rank0"] = rows16
    out["algorithmic_bytes_all_views_compacted"] = int(alg_bytes_compacted)
    for batch, compact in ((1, False), (16, False), (16, True)):
        def sweep():
            return views.render_views_sharded(cams if batch == 1 else pack, bufs, bg, deg, rank=D.rank, world=D.world,
                                              batch=batch, compact=compact)
        sweep()   # warm-up (allocator, first touch, size hints)
        times = []
        for _ in range(repeats):
            D.fence()
            t0 = time.perf_counter()
            sweep()
            D.fence()
            times.append(D.max_over_ranks(time.perf_counter() - t0))
        t = sorted(times)[len(times) // 2]
        # "_compacted": every 16-view batch rendered from the rows its own visibility filter kept (the reference's
        # prefilter_voxel step, for the whole path in one pass); the filter and the gather are INSIDE the timed sweep
        out[f"views_per_call_{batch}" + ("_compacted" if compact else "")] = {
            "sweep_ms": round(t * 1e3, 3), "ms_per_view_per_rank": round(t / max(len(mine), 1) * 1e3, 4),
            "value": round(n_views * P / t / 1e6, 1),
            "value_including_broadcast": round(n_views * P / (t + bcast_s) / 1e6, 1),
            # bytes of the rows this form of the sweep SUBMITS (all P per view, or each batch's kept rows) / time / peak
            "roofline_frac_algorithmic": round((alg_bytes_compacted if compact else alg_bytes) / t / 1e9
                                               / (HBM_PEAK_GBS * D.world), 5)}
    # ---- the MI355X-native distribution: every rank gets ONLY what its block of neighbouring views can see, over its
    # own xGMI link (views.scatter_visible_gaussians), instead of the 236 B/Gaussian broadcast on every link ----
    def distribute(form):
        if form == "batched":
            def fn():
                return views.scatter_visible_gaussians(bufs if D.rank == 0 else None, pack, src=0, assignment="contiguous",
                                                       device=dev)
        else:   # --experimental --c4-forms all
            from bloomscene_amd.experimental import view_distribution as XD
            fn = {"pipelined": lambda: XD.scatter_visible_gaussians_pipelined(bufs if D.rank == 0 else None, pack, src=0,
                                                                              assignment="contiguous", device=dev,
                                                                              pipelined=True),
                  "blockwise": lambda: XD.scatter_visible_gaussians_blockwise(bufs if D.rank == 0 else None, pack, src=0,
                                                                              device=dev)}[form]
        for _ in range(2):   # first pass: code-object loads, allocator growth, RCCL's peer connections; second: measured
            D.fence()
            t0 = time.perf_counter()
            res = fn()
            D.fence()
            dt = D.max_over_ranks(time.perf_counter() - t0)
        return res, dt
    (local, my_views, info), dist_s = distribute("batched")
    sc_out = {"assignment": "contiguous blocks of views", "form": "batched: one filter, one pack, all sends posted as one group",
              "rows_per_rank": info["counts"],
              "bytes_per_rank": info["bytes"], "filter_ms": round(info["filter_ms"], 3), "pack_ms": round(info["pack_ms"], 3),
              "comm_ms": round(D.max_over_ranks(info["comm_ms"]), 3), "distribution_ms": round(dist_s * 1e3, 3)}
    if D.multi and args.experimental and args.c4_forms == "all":
        # the two pipelined forms beside it (DESIGN.md "Multi-GPU"): which one a node prefers is for such a record to say.
        # Opt-in: their unbatched point-to-point sends have never run on RCCL with more than one rank, and a hang there
        # must not cost the default invocation its line.
        for form in ("pipelined", "blockwise"):
            (_, _, inf2), d2 = distribute(form)
            sc_out["distribution_ms_" + form] = round(d2 * 1e3, 3)

    def sweep_local():
        return views.render_views_sharded(pack, local, bg, deg, rank=D.rank, world=D.world, batch=16, views=my_views)
    sweep_local()
    times = []
    for _ in range(repeats):
        D.fence()
        t0 = time.perf_counter()
        sweep_local()
        D.fence()
        times.append(D.max_over_ranks(time.perf_counter() - t0))
    t = sorted(times)[len(times) // 2]
    rows_local = int(local["means3D"].shape[0])
    alg_local = sum(D.gather_ints(len(my_views) * ((187 + 12 * M) * rows_local + 24 * W * H))) + 48 * R_all
    sc_out["views_per_call_16"] = {"sweep_ms": round(t * 1e3, 3),
                                   "ms_per_view_per_rank": round(t / max(len(my_views), 1) * 1e3, 4),
                                   "value": round(n_views * P / t / 1e6, 1),
                                   "value_including_distribution": round(n_views * P / (t + dist_s) / 1e6, 1),
                                   "roofline_frac_algorithmic": round(alg_local / t / 1e9 / (HBM_PEAK_GBS * D.world), 5)}
    out["scatter_visible"] = sc_out
    # ---- stated model for the first SCALE record to be checked against (views.modelled_scatter_sweep): the source filters,
    # then packs and sends rank by rank (remote ranks first, own block last); every rank renders from its arrival on.
    # Evaluated for even view blocks and for the best UNEVEN split (the source, busy distributing, takes fewer views);
    # broadcast path: 236 B P / link rate + ceil(views / N) views per rank.
    if D.rank == 0 and args.experimental:
        from bloomscene_amd.experimental import view_distribution as XD
        LINK_GBS = 153.0   # xGMI, one link (MI355X guide); point-to-point mesh: the N - 1 sends run on their own links
        row_bytes = (3 + 3 + 4 + 1 + 3 * M) * 4
        per_view_ms = sc_out["views_per_call_16"]["ms_per_view_per_rank"]
        per_view_bcast_ms = out["views_per_call_16"]["ms_per_view_per_rank"]
        total_rows = max(1, sum(info["counts"]))
        pack_per_row = info["pack_ms"] / total_rows
        pred = {"model": "views.modelled_scatter_sweep: t(N) = max over ranks of [filter_ms + packs up to and including the "
                         "rank's own (source: all packs) + rows_r * row_bytes / 153 GB/s + views_r * ms_per_view]; N = 1: "
                         "views * ms_per_view; broadcast path: P * row_bytes / 153 GB/s + ceil(views / N) * "
                         "ms_per_view_broadcast_path; stage times as measured in THIS run on one GPU",
                "row_bytes": row_bytes, "link_GBs": LINK_GBS, "ms_per_view": per_view_ms,
                "ms_per_view_broadcast_path": per_view_bcast_ms, "filter_ms": round(info["filter_ms"], 4),
                "pack_ms_per_million_rows": round(pack_per_row * 1e6, 4)}
        t1 = n_views * per_view_ms

        def filter_block_ms(block):
            """one launch of the visibility filter for ONE view block (what the blockwise source runs per rank), GPU time"""
            if not block:
                return 0.0
            views.group_visibility(pack, bufs["means3D"], bufs["scales"], bufs["rotations"], [block])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                views.group_visibility(pack, bufs["means3D"], bufs["scales"], bufs["rotations"], [block])
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / 3
        for w in (1, 2, 4, 8):
            def model(sizes, pipelined=True):
                rr = XD.visible_rows_per_rank(bufs, pack, worlds=(w,), assignment="contiguous", sizes=sizes)[w]
                fb = [filter_block_ms(views.assign_views(n_views, r, w, "contiguous", sizes=sizes)) for r in range(w)] \
                    if pipelined == "blocks" else None
                return rr, XD.modelled_scatter_sweep(n_views, w, rr, sizes, info["filter_ms"], pack_per_row, row_bytes,
                                                        per_view_ms, LINK_GBS, pipelined=pipelined, filter_block_ms=fb)
            even_sizes = views.staggered_block_sizes(n_views, w)
            rows_even, even = model(even_sizes)
            _, batched = model(even_sizes, pipelined=False)
            _, blocks = model(even_sizes, pipelined="blocks") if w > 1 else (None, even)
            # uneven blocks: a rank whose rows leave late gets fewer views (views.balanced_block_sizes).  A block's rows
            # (hence its pack and wire time) depend on its size, so the split is iterated: leave times from the previous
            # split's rows -> new split, three rounds.
            sizes, rows_b, bal = even_sizes, rows_even, even
            for _ in range(3 if w > 1 else 0):
                pk = [rows_b[r] * pack_per_row for r in range(w)]
                leave = [info["filter_ms"] + sum(pk)] + [info["filter_ms"] + sum(pk[1:r + 1]) + rows_b[r] * row_bytes /
                                                         (LINK_GBS * 1e6) for r in range(1, w)]
                cand = XD.balanced_block_sizes(n_views, w, leave, per_view_ms)
                rows_c, m = model(cand)
                if m["sweep_ms"] < bal["sweep_ms"]:
                    sizes, rows_b, bal = cand, rows_c, m
            tb = (0.0 if w == 1 else P * row_bytes / (LINK_GBS * 1e6)) + math.ceil(n_views / w) * per_view_bcast_ms
            t_render = math.ceil(n_views / w) * per_view_ms
            pred[str(w)] = {
                "rows_max": max(rows_even),
                "sweep_ms_scatter": round(even["sweep_ms"], 3), "speedup_scatter": round(t1 / even["sweep_ms"], 2),
                "critical_rank": even["critical_rank"],
                "sweep_ms_scatter_unpipelined": round(batched["sweep_ms"], 3),
                # the source pipelined block by block (scatter_visible_gaussians_blockwise), even blocks
                "sweep_ms_scatter_blockwise": round(blocks["sweep_ms"], 3),
                "speedup_scatter_blockwise": round(t1 / blocks["sweep_ms"], 2),
                # uneven contiguous blocks (scatter_visible_gaussians(sizes=...)): ranks served late render fewer views
                "balanced": {"views_per_rank": sizes, "rows_per_rank": rows_b, "sweep_ms_scatter": round(bal["sweep_ms"], 3),
                             "speedup_scatter": round(t1 / bal["sweep_ms"], 2), "critical_rank": bal["critical_rank"],
                             # what no split can beat: the filter, one block's wire time and its share of the views
                             "floor_ms": round(info["filter_ms"] + min(rows_b[1:] or [0]) * row_bytes / (LINK_GBS * 1e6)
                                               + n_views / w * per_view_ms, 3) if w > 1 else None},
                "sweep_ms_broadcast": round(tb, 3),
                "Msplats_per_s_scatter": round(n_views * P / (bal["sweep_ms"] * 1e-3) / 1e6, 1),
                # the sweep alone, Gaussians already distributed (a second camera path over the same scene):
                # ceil(views / N) views per rank, no exchange at all
                "sweep_ms_resident": round(t_render, 3),
                "speedup_resident": round(n_views * per_view_ms / t_render, 2)}
        # Is north_star's >= 6x at 8 GPUs reachable COLD (Gaussians on one rank when the clock starts) for this preset?
        # No schedule can beat: one block's filter launch + its pack + its wire time + an eighth of the rendering.
        p8 = pred["8"]
        rows8 = p8["balanced"]["rows_per_rank"]
        fb_min = filter_block_ms(views.assign_views(n_views, 1, 8, "contiguous"))
        floor8 = fb_min + min(rows8[1:]) * (pack_per_row + row_bytes / (LINK_GBS * 1e6)) + n_views / 8 * per_view_ms
        best8 = min(p8["sweep_ms_scatter"], p8["balanced"]["sweep_ms_scatter"], p8["sweep_ms_scatter_blockwise"])
        pred["cold_6x_at_8_gpus"] = {
            "needed_sweep_ms": round(t1 / 6, 3), "modelled_best_ms": round(best8, 3), "modelled_best_speedup": round(t1 / best8, 2),
            "floor_ms": round(floor8, 3), "floor_speedup": round(t1 / floor8, 2), "reachable": bool(t1 / floor8 >= 6.0),
            "why": ("rendering an eighth of the views takes %.3f ms of the %.3f ms a 6x sweep may take; one block's filter + pack "
                    "+ wire alone need %.3f ms" % (n_views / 8 * per_view_ms, t1 / 6, floor8 - n_views / 8 * per_view_ms))}
        out["predicted"] = pred
    del bufs, local
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--gaussians", type=int, default=0, help="override P (experiments only)")
    ap.add_argument("--colors", default="sh", choices=["sh", "precomp"],
                    help="precomp: colors_precomp[P,3] instead of SHs, the call shape of the reference's render() "
                         "(gaussian_renderer/__init__.py:254-262); experiments only, the metric is quoted on sh")
    ap.add_argument("--scale-mul", type=float, default=1.0, help="multiply every scale (denser tile lists); experiments only")
    ap.add_argument("--cycle-views", type=int, default=1, help="change the camera every step among this many views")
    ap.add_argument("--allreduce-grads", action="store_true",
                    help="data-parallel training over views -- sum the per-view gradients with one packed RCCL all-reduce "
                         "inside every step (SURVEY.md §8f rank 3); off by default, the metric's path has no data-path "
                         "collective")
    ap.add_argument("--depth-gradient", action="store_true",
                    help="opt-in extension: also backpropagate the depth target (bsr_backward_depth); the metric "
                         "is quoted without it (the reference ignores grad_depth)")
    ap.add_argument("--lib", default="", help="measurement only: time another build of the same C ABI (A/B runs, the "
                                                "diagnostic builds of csrc/Makefile) instead of the in-tree product library")
    ap.add_argument("--lib-older-abi", action="store_true",
                    help="with --lib: accept a build of an earlier ABI version (A/B runs against an earlier round's library)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c4", action="store_true", help="skip the C4 rotate360 sweep leg")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary (N = 1) workloads")
    ap.add_argument("--c4-forms", default="batched", choices=["batched", "all"],
                    help="N > 1: which forms of the visible-subset distribution the C4 leg times -- 'batched' (one pack, all "
                         "sends posted as one group: the default of views.scatter_visible_gaussians) or 'all' (also the "
                         "rank-by-rank pipeline and the blockwise one)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="Gaussians in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--exact-exp", action="store_true",
                    help="BSR_FLAG_EXACT_EXP on every call: the pinned exp on every evaluation of the forward blend "
                         "(bit-equal to the CPU oracle); the metric is quoted on the library's default")
    ap.add_argument("--strict-gradients", action="store_true",
                    help="BSR_FLAG_EXACT_GRAD on every call: the reference's per-pair operations in the backward tile "
                         "walk (meets SURVEY 8(d)'s elementwise gradient bar); the metric is quoted on the default")
    ap.add_argument("--full", action="store_true",
                    help="also run the long legs (the 180- and 720-view C4 presets, capacity-mode and HIP-graph legs) and "
                         "echo the detail record on stderr; the contract line on stdout stays compact either way")
    ap.add_argument("--experimental", action="store_true",
                    help="C4 leg: also evaluate the latency model of the visible-subset distribution and (with --c4-forms "
                         "all) its pipelined forms (bloomscene_amd.experimental; validated on gloo only)")
    ap.add_argument("--detail", default="", help="where the detail record (JSON) is written")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed steps run for this long BEFORE the W warm-up steps of the headline workload, so that the "
                         "GPU clocks have ramped when the K timed steps start (reported as prewarm_steps)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)   # does not return

    from bloomscene_amd import _capi
    if args.lib:
        _capi.use_library(args.lib, allow_older_abi=args.lib_older_abi)
    if args.exact_exp or args.strict_gradients:
        # per-call flags, taken by every rasterizer call of this (main) thread from its numerics context
        from bloomscene_amd import numerics
        numerics(exact_exp=args.exact_exp, strict_gradients=args.strict_gradients).__enter__()
    D = Dist(args.gpus)
    P, W, H, deg, do_bwd = CONFIGS[args.config]
    if args.gaussians:
        P = args.gaussians
    precomp = args.colors == "precomp"
    headline = (args.config == "c3" and not precomp and not args.gaussians and not args.depth_gradient
                and not args.allreduce_grads and args.scale_mul == 1.0 and args.cycle_views == 1)
    r = raster_workload(D, args, P, W, H, deg, do_bwd, precomp=precomp, scale_mul=args.scale_mul,
                        cycle_views=args.cycle_views, label=args.config, allreduce=args.allreduce_grads,
                        depth_gradient=args.depth_gradient)
    leg_failed = [False]

    def guarded(fn, *a, **kw):
        # The secondary legs must not cost the headline its line: an exception there is reported in the leg's place.
        # With several ranks the legs hold collectives, so the ranks must leave a failed leg TOGETHER: after every leg
        # each rank contributes its error flag to one MAX all-reduce, and once any rank has failed no rank starts another
        # leg.  That covers the failures that happen on every rank at the same point (an exception in code all
        # ranks run on same-sized data); a rank that fails alone INSIDE a leg still leaves its peers in that leg's
        # collective until the process group's 300 s patience runs out.
        if leg_failed[0]:
            return {"error": "skipped: an earlier secondary leg failed on some rank"}
        try:
            res = fn(*a, **kw)
            err = None
        except Exception as exc:   # noqa: BLE001
            res, err = None, {"error": f"{type(exc).__name__}: {exc}"}
        try:
            failed_somewhere = D.max_over_ranks(0.0 if err is None else 1.0) > 0.0
        except Exception as exc:   # noqa: BLE001  (the ranks are out of step: no further collective from this rank)
            failed_somewhere = True
            err = err or {"error": f"{type(exc).__name__}: {exc}"}
        if failed_somewhere:
            leg_failed[0] = True
            return err or {"error": "abandoned: this leg failed on another rank"}
        return res
    c4 = None if args.no_c4 else guarded(c4_sweep, D, args)
    if c4 is not None and headline and args.full and "error" not in c4:
        # the reference's own rotate360 preset: 180 views, 2 degrees apart (utils/trajectory.py:102-126)
        c4_180 = guarded(c4_sweep, D, args, n_views=180, repeats=3)
        keep = ("workload", "views", "views_per_rank", "broadcast_ms", "views_per_call_16", "views_per_call_16_compacted",
                "scatter_visible", "predicted")
        c4["preset_180_views"] = c4_180 if "error" in c4_180 else {k: c4_180[k] for k in keep if k in c4_180}
        # the camera path the reference SHIPS and render_video loads (cameras/rotate360.json via utils/camera.py:23-51,
        # bloomscene.py:181): 720 frames, half a degree apart
        c4_720 = guarded(c4_sweep, D, args, n_views=720, repeats=2)
        c4["preset_720_views"] = c4_720 if "error" in c4_720 else {k: c4_720[k] for k in keep if k in c4_720}
    secondary = None
    if D.world == 1 and headline and not args.no_secondary:
        sec_steps = max(20, args.steps // 2)

        def leg(label, **kw):
            return secondary_line(raster_workload(D, args, P, W, H, deg, True, steps=sec_steps, label=label, **kw))

        def with_flags(**flags):
            # a leg under other per-call numerics flags than the headline's (bloomscene_amd.numerics: thread context)
            from bloomscene_amd import numerics

            def run(label, **kw):
                with numerics(**flags):
                    return leg(label, **kw)
            return run
        secondary = {
            "c3_dense_scales_x3": leg("c3-dense", scale_mul=3.0),
            "c3_camera_changes_every_step": leg("c3-cycling", cycle_views=8),
            "bloomscene_shape": bloomscene_shape_workload(D, args),
        }
        if not (args.exact_exp or args.strict_gradients):
            # the two bit-for-bit modes of the library on the headline workload: BSR_FLAG_EXACT_GRAD
            # (the reference's per-pair operations in the backward walk) and BSR_FLAG_EXACT_EXP (forward bit-equal to
            # the oracle)
            secondary["c3_strict_gradients"] = with_flags(strict_gradients=True)("c3-strict-gradients")
            secondary["c3_exact_exp"] = with_flags(exact_exp=True)("c3-exact-exp")
        if args.full:
            secondary.update({
                "bloomscene_shape_no_host_wait": bloomscene_shape_workload(D, args, mode="capacity"),
                "bloomscene_shape_hip_graph": bloomscene_shape_workload(D, args, mode="graph"),
                # the forward without its host wait (include/bloomscene_rast.h BSR_FLAG_NO_READBACK), and the whole step
                # as a HIP graph: the headline workload, and the rasterizer alone at BloomScene's call shape (500 k
                # selected Gaussians, colors_precomp, sh_degree 1, 512 x 512) where launch gaps and the wait are a larger
                # share (the headline workload once more, in the same place of the process as the two legs below)
                "c3_default_again": leg("c3-again"),
                "c3_capacity_mode": leg("c3-capacity", mode="capacity"),
                "c3_hip_graph_replay": leg("c3-graph", mode="graph"),
                "raster_512_precomp": {m: secondary_line(raster_workload(D, args, 500_000, 512, 512, 1, True, precomp=True,
                                                                         steps=sec_steps, label="512-" + m, mode=m))
                                       for m in ("default", "capacity", "graph")},
            })

    device_ids = D.gather_ints(D.dev.index)   # (a collective: every rank)
    if D.rank == 0:
        stages, alg = r["stages"], r["alg"]
        dom = max(stages, key=lambda k: stages[k]) if stages else None
        roofline, roofline_detail = None, None
        if dom is not None:
            achieved = alg.get(dom, 0) / (stages[dom] * 1e-3) / 1e9
            traffic, valu, src = measured_traffic(args.config, dom) if headline else (None, None, None)
            # roofline.traffic is a CONSTANT read from a committed rocprofv3 --pmc profile of the same kernel sources,
            # never a measurement of this run (counters cannot be collected inside it): traffic_profile names it
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "traffic_profile": (src or {}).get("profile") if (src or {}).get("matches_timed_build") else None,
                        "algorithmic_bytes": alg.get(dom, 0), "launch_ms": round(stages[dom], 4),
                        "launches_timed": int(r["prof"][dom][1])}
            roofline_detail = dict(roofline, traffic_source=src,
                                   # which stage carried hipEvents INSIDE the timed region (the others: untimed pass)
                                   timed_region_events=r["timed_region_events"],
                                   # what actually bounds the tile renderers (same committed PMC passes): VALU
                                   # instructions issued per SIMD and core-clock cycle
                                   valu_insts_per_simd_cycle=valu)
        whole = r["step_bytes"] / (r["ms_per_step"] * 1e-3) / 1e9
        # ---- the contract line: compact (< 4 KB, asserted), the LAST line of stdout.  Everything else -- per-step
        # arrays, the C4 record, the secondary legs -- goes to the detail file named in "detail" (and to stderr).
        line = {
            "metric": "Msplats/s fwd+bwd @1M Gaussians 1920x1080 SH3; fraction of HBM roofline" if headline
            else f"Msplats/s ({args.config})",
            "value": round(r["value"], 3), "unit": "Msplats/s", "n_gpus": D.world, "steps": r["steps"],
            "warmup": r["warmup"], "ms_per_step": round(r["ms_per_step"], 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": r["workload"], "gaussians": P, "width": W, "height": H, "sh_degree": r["deg"],
                       "num_rendered": r["R"], "visible": r["visible"],
                       "parallelism": f"view-parallel x{D.world}" + (" + gradient all-reduce" if args.allreduce_grads
                                                                      and D.multi and do_bwd else ""),
                       "csrc_sha256": csrc_sha256(), "exact_exp": int(args.exact_exp),
                       "strict_gradients": int(args.strict_gradients),
                       "ms_per_step_median": round(r["ms_per_step_median"], 4),
                       "step_ms_first": round(r["step_ms_first"], 4), "step_ms_max": round(r["step_ms_max"], 4),
                       "host_max_ms_per_step": round(r["max_host_ms"], 3), "prewarm_steps": r["prewarm_steps"]},
            "roofline": roofline,
            "roofline_step": {"algorithmic_bytes": r["step_bytes"], "achieved": round(whole, 2),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(whole / HBM_PEAK_GBS, 5)},
            "stage_ms": {k: round(v, 4) for k, v in stages.items()},
        }
        if D.multi:
            line["world_size"] = D.world
            line["config"]["broadcast_ms"] = round(r["bcast_ms"], 3)
            line["devices"] = device_ids
        if args.allreduce_grads and D.multi:
            line["config"]["allreduce_ms_per_step"] = round(r["allreduce_ms_per_step"], 4)
        if c4 is not None:
            line["c4"] = c4_summary(c4)
        if secondary is not None:
            line["secondary_Msplats_per_s"] = {k: (v.get("value") if isinstance(v, dict) and "value" in v else None)
                                               for k, v in secondary.items() if k != "raster_512_precomp"}
        if D.world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, P, W, H, r["deg"], do_bwd, precomp)
        detail = dict(line)
        detail["roofline"] = roofline_detail
        detail["config"] = dict(line["config"], broadcast_ms=round(r["bcast_ms"], 3),
                                steps_over_1p15x_median=r["steps_over_1p15x_median"],
                                host_median_ms_per_step=round(r["host_median_ms"], 3),
                                closing_fence_tail_ms=round(r["tail_ms"], 4),
                                device_allocs_in_timed_region=r["device_allocs"],
                                sum_stage_ms=round(sum(stages.values()), 4), prewarm_ms=args.prewarm_ms)
        detail["step_ms"] = r["step_ms"]
        detail["host_step_ms"] = r["host_step_ms"]
        detail.pop("secondary_Msplats_per_s", None)
        if c4 is not None:
            detail["c4"] = c4
        if secondary is not None:
            detail["secondary"] = secondary
        line["detail"] = write_detail(detail, args)
        text = json.dumps(line, separators=(",", ":"))
        if len(text) >= MAX_LINE_BYTES:   # never print a line the driver cannot parse: drop the optional blocks
            for k in ("secondary_Msplats_per_s", "c4", "roofline_step", "devices"):
                line.pop(k, None)
            line["truncated"] = True
            text = json.dumps(line, separators=(",", ":"))
        sys.stdout.flush()
        print(text, flush=True)
    D.close()


MAX_LINE_BYTES = 4096


def c4_summary(c4):
    """The C4 record in a handful of numbers (the whole record: the detail file).  cold = Gaussians on rank 0 when the
    clock starts (packed RCCL broadcast + sweep); resident = the sweep alone, Gaussians already on every rank."""
    if "error" in c4:
        return {"error": str(c4["error"])[:200]}
    v16 = c4.get("views_per_call_16", {})
    sv = c4.get("scatter_visible", {})
    out = {"views": c4.get("views"), "scaling": "strong", "n_gpus": c4.get("n_gpus"),
           "views_per_rank": c4.get("views_per_rank"), "unit": "Msplats/s",
           "broadcast_ms": c4.get("broadcast_ms"),
           "sweep_ms_resident": v16.get("sweep_ms"), "value_resident": v16.get("value"),
           "sweep_ms_cold": round(v16.get("sweep_ms", 0.0) + c4.get("broadcast_ms", 0.0), 3),
           "value_cold": v16.get("value_including_broadcast"),
           "sweep_ms_1_view_per_call": c4.get("views_per_call_1", {}).get("sweep_ms")}
    if sv:
        loc = sv.get("views_per_call_16", {})
        out["scatter_visible"] = {"distribution_ms": sv.get("distribution_ms"), "sweep_ms": loc.get("sweep_ms"),
                                  "value_cold": loc.get("value_including_distribution")}
    return out


def write_detail(detail, args):
    """Everything the contract line leaves out, as JSON: to the file --detail names (default bench_detail.json beside
    bench.py, or under gpurun_out/ when that exists so that it travels back from a GPU box) and, with --full, to stderr.
    Returns the path written (None when no location was writable)."""
    text = json.dumps(detail)
    if args.full:
        print(text, file=sys.stderr, flush=True)
    path = args.detail
    if not path:
        out_dir = os.path.join(ROOT, "gpurun_out")
        path = os.path.join(out_dir if os.path.isdir(out_dir) else ROOT, "bench_detail.json")
    try:
        with open(path, "w") as fh:
            fh.write(text + "\n")
        return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError:
        return None


def cpu_baseline(args, P, W, H, deg, do_bwd, precomp=False):
    """The CPU oracle (oracle/bsr_oracle.c: a port of the reference algorithm, OpenMP over
    Gaussians/tiles) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import oracle as O
    from bloomscene_amd.synthetic import scene_a, upstream_grads
    cores = os.cpu_count() or 1
    Ps = args.cpu_sample or min(P, max(50_000, 125_000 * cores))   # ~10-30 s of CPU work
    sc = scene_a(Ps, W, H, 0 if precomp else deg, seed=0)
    col = torch.rand(Ps, 3, generator=torch.Generator().manual_seed(13)) if precomp else None
    cam = sc.cameras[0]
    rs = O.make_settings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), [0, 0, 0], 1.0,
                         cam.world_view_transform, cam.full_proj_transform, deg, cam.camera_center)
    gC, gD = upstream_grads(W, H, seed=1)
    O.lib()
    t0 = time.perf_counter()
    st = O.forward(rs, sc.means3D, sc.opacities, shs=None if precomp else sc.shs, colors_precomp=col,
                   scales=sc.scales * args.scale_mul, rotations=sc.rotations)
    if do_bwd:
        O.backward(st, gC, gD)
    dt = time.perf_counter() - t0
    return {"value": round(Ps / dt / 1e6, 4), "unit": "Msplats/s", "cores": cores, "kind": "port",
            "sample": f"1 step of the same workload with {Ps} Gaussians ({'fwd+bwd' if do_bwd else 'fwd'}, "
                      f"{W}x{H}, SH deg {deg}), {dt:.1f} s, OpenMP over {cores} host threads"}


if __name__ == "__main__":
    main()
