#!/usr/bin/env python3
"""Benchmark of the rasterizer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Metric (BASELINE.json): Msplats/s forward+backward @ 1 M Gaussians, 1920x1080, SH degree 3
(config C3, synthetic scene A of SURVEY.md §8d), colour + depth targets with upstream gradients.
One "step" = one `GaussianRasterizer` forward + `torch.autograd.backward` through the C ABI, inputs
resident in HBM.  N > 1: view-parallel -- rank 0 broadcasts the Gaussian buffers over RCCL once
(outside the timed region), every rank then renders its own camera view; no data-path collective
(weak scaling).  Rank 0 prints ONE JSON line with the bench contract keys plus

  "roofline":     dominant kernel, ALGORITHMIC bytes per launch / mean launch time (hipEvents
                  recorded by the library on the launch stream during the timed region) vs 8 TB/s
  "cpu_baseline": the CPU oracle (a port of the reference algorithm, OpenMP) on the same workload,
                  rank 0 at N = 1 only.
"""
from __future__ import annotations

import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--gaussians", type=int, default=0, help="override P (experiments only)")
    ap.add_argument("--colors", default="sh", choices=["sh", "precomp"],
                    help="precomp: colors_precomp[P,3] instead of SHs, the call shape of the reference's render() "
                         "(gaussian_renderer/__init__.py:254-262); experiments only, the metric is quoted on sh")
    ap.add_argument("--allreduce-grads", action="store_true",
                    help="N > 1 only: data-parallel training over views -- sum the per-view gradients with one "
                         "packed RCCL all-reduce inside every step (SURVEY.md §8f rank 3); off by default, the "
                         "metric's path has no data-path collective")
    ap.add_argument("--depth-gradient", action="store_true",
                    help="opt-in extension: also backpropagate the depth target (bsr_backward_depth); the metric "
                         "is quoted without it (the reference ignores grad_depth)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="Gaussians in the CPU-baseline sample (0 = auto)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in bloomscene_amd)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    # BSR_BENCH_FORCE_DIST=1: take the N > 1 code path (RCCL init, broadcast, barrier, all-reduce) with one rank
    force_dist = os.environ.get("BSR_BENCH_FORCE_DIST") == "1" and "MASTER_PORT" in os.environ
    multi = world > 1 or force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from bloomscene_amd import GaussianRasterizationSettings, GaussianRasterizer, _capi
    from bloomscene_amd.synthetic import scene_a, upstream_grads
    from bloomscene_amd.views import allreduce_gradients, broadcast_gaussians, yawed_camera

    P, W, H, deg, do_bwd = CONFIGS[args.config]
    if args.gaussians:
        P = args.gaussians
    precomp = args.colors == "precomp"
    if precomp:
        deg = 1   # the reference passes sh_degree=1 with shs=None (gaussian_renderer/__init__.py:244,257)
    M = 0 if precomp else (deg + 1) ** 2
    N = W * H

    # ---- inputs: generated on rank 0's host, broadcast over RCCL/xGMI, resident in HBM ----
    names = ("means3D", "scales", "rotations", "opacities", "shs")
    if rank == 0:
        sc = scene_a(P, W, H, 0 if precomp else deg, seed=0)
        if precomp:
            sc.shs = torch.rand(P, 3, generator=torch.Generator().manual_seed(13))   # colours in [0, 1)
        bufs = {k: getattr(sc, k).to(dev) for k in names}
    else:
        shapes = {"means3D": (P, 3), "scales": (P, 3), "rotations": (P, 4), "opacities": (P, 1),
                  "shs": (P, 3) if precomp else (P, M, 3)}
        bufs = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in names}
    bcast_ms = broadcast_gaussians(bufs, src=0, force=force_dist) if multi else 0.0
    gC, gD = upstream_grads(W, H, seed=1)
    gC, gD = gC.to(dev), gD.to(dev)
    # independent views: rank r looks 0.25*r degrees to the side of the scene-A camera
    cam = yawed_camera(W, H, math.radians(60.0), yaw_deg=0.25 * rank).to(dev)
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=deg, campos=cam.camera_center, prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(settings, depth_gradient=args.depth_gradient)
    leaves = {k: v.requires_grad_(do_bwd) for k, v in bufs.items()}
    state = {}

    def step():
        means2D = torch.zeros_like(leaves["means3D"], requires_grad=do_bwd)
        color, radii, depth = rasterizer(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                         shs=None if precomp else leaves["shs"],
                                         colors_precomp=leaves["shs"] if precomp else None,
                                         scales=leaves["scales"], rotations=leaves["rotations"])
        if do_bwd:
            for v in leaves.values():
                v.grad = None
            torch.autograd.backward((color, depth), (gC, gD))
            if args.allreduce_grads and multi:
                state["allreduce_ms"] = state.get("allreduce_ms", 0.0) + allreduce_gradients(leaves, force=force_dist)
        state["radii"] = radii

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    import gc
    gc.collect()
    gc.disable()   # no collector pauses inside the timed region (the steps create no reference cycles)
    # stage events on every 4th step of the timed region: each hipEvent costs a few microseconds of pipeline
    # bubble, 14 per step were 3 % of the step; the per-launch means below are over the sampled launches
    sample_every = 4 if args.steps >= 8 else 1
    _capi.profile_enable(0 if os.environ.get("BSR_BENCH_NO_STAGE_EVENTS") == "1" else sample_every)
    _capi.profile_reset()
    allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    host_marks = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        host_marks.append(time.perf_counter())
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    # diagnostics: hipMalloc calls inside the timed region (0 in steady state) and the slowest host-side step
    device_allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0
    host_step_ms = [1e3 * (b - a) for a, b in zip([t0] + host_marks[:-1], host_marks)]
    if os.environ.get("BSR_BENCH_STEP_TIMES") == "1":
        print("host ms per step:", " ".join("%.2f" % v for v in host_step_ms), file=sys.stderr)
    prof = _capi.profile_read()
    _capi.profile_enable(False)

    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # instances of this rank's view, for the algorithmic byte count
    from bloomscene_amd.rasterizer import _rasterize_gaussians_native
    e = torch.Tensor([])
    with torch.no_grad():
        R = _rasterize_gaussians_native(settings.bg, bufs["means3D"], bufs["shs"] if precomp else e,
                                        bufs["opacities"], bufs["scales"], bufs["rotations"], 1.0, e,
                                        settings.viewmatrix, settings.projmatrix, settings.tanfovx, settings.tanfovy,
                                        H, W, e if precomp else bufs["shs"], deg, settings.campos, False, False)[0]
    visible = int((state["radii"] > 0).sum().item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * P * args.steps / dt / 1e6
        stages = {k: v[0] / max(v[1], 1) for k, v in prof.items()}  # mean ms per launch
        alg = algorithmic_bytes(P, M, R, N)
        if precomp:   # 12 B/G of colours read by preprocess, 12 B/G of colour gradient passed through
            alg["preprocess"] += 12 * P
            alg["preprocess_bwd"] += 24 * P
        dom = max(stages, key=lambda k: stages[k]) if stages else None
        roofline = None
        if dom is not None:
            achieved = alg.get(dom, 0) / (stages[dom] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                        "algorithmic_bytes": alg.get(dom, 0), "launch_ms": round(stages[dom], 4),
                        "launches_timed": int(prof[dom][1])}
        if roofline is not None:
            roofline["traffic"] = measured_traffic(args.config, dom)
            # what actually bounds the tile renderers (from the same committed PMC passes): VALU instructions
            # issued per SIMD and core-clock cycle; ~0.29 with this instruction mix is a saturated VALU port
            roofline["valu_insts_per_simd_cycle"] = measured_traffic(args.config, dom, "valu_insts_per_simd_cycle")
        sb = step_bytes(P, M, R, N, do_bwd) + (36 * P if precomp and do_bwd else 12 * P if precomp else 0)
        whole = sb / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "Msplats/s fwd+bwd @1M Gaussians 1920x1080 SH3; fraction of HBM roofline"
            if args.config == "c3" and not precomp and not args.gaussians and not args.depth_gradient
            and not args.allreduce_grads
            else f"Msplats/s ({args.config})",
            "value": round(value, 3), "unit": "Msplats/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {P} Gaussians, {'precomputed colours' if precomp else f'SH deg {deg}'}, {W}x{H}, "
                                   f"{'fwd+bwd colour+depth targets' if do_bwd else 'fwd only'}"
                                   f"{' WITH depth gradient (extension)' if args.depth_gradient and do_bwd else ''}, synthetic scene A seed 0",
                       "gaussians": P, "width": W, "height": H, "sh_degree": deg, "num_rendered": R,
                       "visible": visible,
                       "parallelism": f"view-parallel x{world}" + (" + gradient all-reduce" if args.allreduce_grads
                                                                    and world > 1 and do_bwd else ""),
                       "broadcast_ms": round(bcast_ms, 3)},
            "roofline": roofline,
            "roofline_step": {"algorithmic_bytes": sb, "achieved": round(whole, 2),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(whole / HBM_PEAK_GBS, 5)},
            "stage_ms": {k: round(v, 4) for k, v in stages.items()},
            "host": {"device_allocs_in_timed_region": int(device_allocs),
                     "max_host_ms_per_step": round(max(host_step_ms), 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, P, W, H, deg, do_bwd, precomp)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(config, stage, key="hbm_bytes"):
    """HBM bytes per launch of `stage` from the committed rocprofv3 --pmc passes of this workload
    (profiles/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate passes, FETCH_SIZE
    doubled as the MI355X guide prescribes for wide coalesced reads on gfx950), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh)[config][stage][key]
    except (OSError, KeyError, ValueError):
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
                   scales=sc.scales, rotations=sc.rotations)
    if do_bwd:
        O.backward(st, gC, gD)
    dt = time.perf_counter() - t0
    return {"value": round(Ps / dt / 1e6, 4), "unit": "Msplats/s", "cores": cores, "kind": "port",
            "sample": f"1 step of the same workload with {Ps} Gaussians ({'fwd+bwd' if do_bwd else 'fwd'}, "
                      f"{W}x{H}, SH deg {deg}), {dt:.1f} s, OpenMP over {cores} host threads"}


if __name__ == "__main__":
    main()
