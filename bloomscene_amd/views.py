"""View-parallel rendering across GPUs and the renderer-side call shapes.

The rasterizer shards naturally by camera view: every view depends only on the shared, read-only
Gaussian buffers and its own camera (reference ``bloomscene.py:191-193`` -- nothing is carried
between loop iterations).  One process per GPU; rank 0 broadcasts the Gaussian buffers once over
RCCL/xGMI (``torch.distributed`` backend "nccl" is RCCL on ROCm), after which the ranks never
talk on the data path.  Views are dealt round-robin so yaw-dependent load balances.

``render_view`` / ``prefilter`` reproduce the call shapes of ``gaussian_renderer.render`` and
``prefilter_voxel`` (reference ``gaussian_renderer/__init__.py:211-291,294-349``) for already
decoded Gaussians; ``render_neural`` starts one step earlier, at the outputs of the anchor MLP heads
(the MLPs, the context model and the entropy coder themselves are out of scope).
"""
from __future__ import annotations

import functools
import math
import time

import numpy as np
import torch
import torch.distributed as dist

from .cameras import MiniCam, make_minicam, vertical_fov, yaw_rotation


def shard_views(n_views: int, rank: int, world: int):
    """Round-robin view assignment: rank r renders views {i : i mod world == r}."""
    return list(range(rank, n_views, world))


class CameraPack:
    """A camera path with its matrices stacked ON THE DEVICE once: world_view [V,4,4], full_proj [V,4,4], centers
    [V,3].  The multi-view entry points take a list of MiniCams or a CameraPack; with a list they stack (and upload)
    the matrices on every call, which for the 16 views of one batched native call is a quarter of a millisecond of
    interpreter time -- as much as the kernels of a sparse view.  All cameras share image size and field of view
    (they do in the rotate360 sweep: utils/trajectory.py:110-121)."""

    def __init__(self, cams, device):
        cams = list(cams)
        if not cams:
            raise ValueError("CameraPack needs at least one camera")
        c0 = cams[0]
        for c in cams:
            if (c.image_width, c.image_height) != (c0.image_width, c0.image_height) or c.FoVx != c0.FoVx \
                    or c.FoVy != c0.FoVy:
                raise ValueError("a CameraPack needs cameras of one image size and field of view")
        self.image_width, self.image_height, self.FoVx, self.FoVy = c0.image_width, c0.image_height, c0.FoVx, c0.FoVy
        self.world_view = torch.stack([c.world_view_transform for c in cams]).to(device).contiguous()
        self.full_proj = torch.stack([c.full_proj_transform for c in cams]).to(device).contiguous()
        self.centers = torch.stack([c.camera_center for c in cams]).to(device).contiguous()

    def __len__(self):
        return self.world_view.shape[0]

    def select(self, idx):
        """(world_view, full_proj, centers) of the views ``idx``: a slice when they are consecutive, else a gather."""
        idx = list(idx)
        if idx and idx == list(range(idx[0], idx[0] + len(idx))):
            sl = slice(idx[0], idx[0] + len(idx))
            return self.world_view[sl], self.full_proj[sl], self.centers[sl]
        t = torch.tensor(idx, dtype=torch.long, device=self.world_view.device)
        return self.world_view.index_select(0, t), self.full_proj.index_select(0, t), self.centers.index_select(0, t)


def _camera_stack(cams, idx, dev):
    """(world_view [n,4,4], full_proj, centers, camera 0) of the views ``idx`` of a CameraPack or a MiniCam list."""
    if isinstance(cams, CameraPack):
        vms, pms, cps = cams.select(idx)
        return vms, pms, cps, cams
    sel = [cams[i] for i in idx]
    c0 = sel[0]
    for c in sel:
        if (c.image_width, c.image_height) != (c0.image_width, c0.image_height) or c.FoVx != c0.FoVx or c.FoVy != c0.FoVy:
            raise ValueError("the multi-view entry points need cameras of one image size and field of view")
    return (torch.stack([c.world_view_transform.to(dev) for c in sel]).contiguous(),
            torch.stack([c.full_proj_transform.to(dev) for c in sel]).contiguous(),
            torch.stack([c.camera_center.to(dev) for c in sel]).contiguous(), c0)


def staggered_block_sizes(n_views: int, world: int, src: int = 0, src_fewer: int = 0):
    """Sizes of the ranks' contiguous view blocks when the distributing rank `src` takes `src_fewer` views LESS than an
    even share and the others share them out (the first ones served take one more): `src` spends the start of the sweep
    filtering, packing and sending, its peers wait for their rows for a fraction of that -- an even split leaves `src`
    (or its last-served peer) on the critical path (DESIGN.md "Multi-GPU").  src_fewer = 0: the even split of
    assign_views(..., "contiguous")."""
    per = -(-n_views // world)
    sizes = [max(0, min(per, n_views - r * per)) for r in range(world)]
    move = max(0, min(int(src_fewer), sizes[src]))
    if world > 1 and move:
        sizes[src] -= move
        others = [r for r in range(world) if r != src]
        for i in range(move):
            sizes[others[i % len(others)]] += 1
    return sizes


def assign_views(n_views: int, rank: int, world: int, mode: str = "round_robin", src: int = 0, src_fewer: int = 0,
                 sizes=None):
    """View indices of `rank`.  "round_robin": {i : i mod world == rank} (balances a yaw-dependent load; every rank sees
    the whole sweep, so it needs nearly every Gaussian any view sees).  "contiguous": the rank's block of ceil(n / world)
    neighbouring views -- neighbouring views of a rotate360 sweep overlap (utils/trajectory.py:110-121: 360 / n degrees
    apart against a ~57 degree field of view), so the Gaussians ONE rank needs are a small part of the scene; what
    `scatter_visible_gaussians` sends.  With ``src_fewer`` > 0 the blocks are uneven (staggered_block_sizes); ``sizes``
    gives them outright."""
    if mode == "round_robin":
        return list(range(rank, n_views, world))
    if mode == "contiguous":
        if sizes is None:
            sizes = staggered_block_sizes(n_views, world, src, src_fewer)
        if len(sizes) != world or sum(sizes) != n_views or min(sizes) < 0:
            raise ValueError("sizes must hold one non-negative block size per rank and add up to the number of views")
        start = sum(sizes[:rank])
        return list(range(start, start + sizes[rank]))
    raise ValueError(f"unknown view assignment '{mode}'")


def yawed_camera(width, height, fovx, yaw_deg=0.0, device="cpu") -> MiniCam:
    """Camera at the origin whose axes are the world's turned by ``yaw_deg`` about +Y: its optical axis is
    (sin yaw, 0, cos yaw) (the scene-A camera when 0)."""
    return make_minicam(yaw_rotation(-yaw_deg), np.zeros(3), fovx, vertical_fov(fovx, width, height), width, height,
                        device=device)


def broadcast_gaussians(bufs: dict, src: int = 0, force: bool = False) -> float:
    """Broadcast every tensor of ``bufs`` (already allocated with the right shape on all ranks)
    from ``src``.  Returns the elapsed milliseconds.  The tensors are packed into ONE flat fp32
    buffer so a single large collective crosses the xGMI links (236 MB at 1 M Gaussians, SH 3)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return 0.0   # (force: run the collective even with a single rank -- plumbing checks)
    keys = sorted(bufs)
    dev = bufs[keys[0]].device
    sizes = [bufs[k].numel() for k in keys]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    if dist.get_rank() == src:
        torch.cat([bufs[k].detach().reshape(-1) for k in keys], out=flat)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    dist.broadcast(flat, src=src)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) * 1e3
    off = 0
    for k, n in zip(keys, sizes):
        with torch.no_grad():
            bufs[k].copy_(flat[off:off + n].view_as(bufs[k]))
        off += n
    return ms


def scatter_visible_gaussians(bufs, cams, src: int = 0, assignment: str = "contiguous", masks=None,
                              scaling_modifier: float = 1.0, device=None, src_fewer: int = 0, sizes=None):
    """Hand every rank ONLY the Gaussians its views can see, instead of broadcasting all of them.

    xGMI is a point-to-point mesh: rank `src` reaches each of its peers over a link of its own (7 x ~153 GB/s on an
    8-GPU MI355X node).  A broadcast puts the same 236 B per Gaussian on every link (1.5 ms per million Gaussians at SH
    degree 3, whatever the algorithm: per-link bound); but a rank that renders a block of neighbouring views needs only
    the Gaussians inside that block's frusta -- ~8 % of scene B for 8 of 64 views -- and the subsets of different
    ranks travel on different links at the same time.  So: `src` runs the reference's own visibility test
    (prefilter_voxel's, gaussian_renderer/__init__.py:342-349) once over the Gaussians for ALL cameras, reduced on the
    fly to one mask per rank (`bsr_visible_filter_groups`: any of the rank's views has radii > 0), compacts each rank's
    rows in ascending id order into one packed fp32 matrix, and posts all sends together (RCCL send/recv =
    `dist.batch_isend_irecv`).  A Gaussian a view's own preprocess would cull contributes nothing to that view and
    ascending compaction keeps the (depth, id) tie order, so every frame rendered from the subset is bit-identical to
    the frame rendered from all Gaussians (tests/test_round3_gpu.py, tests/test_multigpu_gloo.py).

    ONE form: one filter pass, one pack of all rows, all sends posted as one group.  (Forms that pipeline the source rank
    by rank or block by block, uneven view blocks sized by a latency model, and that model itself live in
    ``bloomscene_amd.experimental.view_distribution`` until a multi-GPU record says which of them a node prefers.)
    ``src_fewer`` / ``sizes``: uneven contiguous view blocks (staggered_block_sizes; the same on every rank).

    ``bufs``: on `src` the dict of full per-Gaussian tensors ([P, ...] fp32; must hold means3D, scales, rotations for
    the filter); ignored elsewhere (pass None).  ``cams``: the whole camera path, on every rank.  ``masks``
    (optional, `src` only): bool [world, P] to use instead of running the filter (CPU plumbing tests).  ``device``:
    where a receiving rank wants its tensors (default: the device of its cameras).
    Returns (local_bufs, my_views, info): the rank's compacted tensors, its view indices, and
    info = {"counts": rows per rank, "bytes": bytes per rank, "filter_ms", "pack_ms", "comm_ms"} (host-clock, synchronised).
    Without a process group it is the single-rank case: the union over all views, no communication."""
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if multi else 0
    world = dist.get_world_size() if multi else 1
    my_views = assign_views(len(cams), rank, world, assignment, src, src_fewer, sizes)
    info = {"filter_ms": 0.0, "pack_ms": 0.0, "comm_ms": 0.0, "pipelined": False}

    def sync(dev):
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    meta = [None]
    flat_all = None
    offsets = None
    if rank == src:
        keys = sorted(bufs)
        dev = bufs[keys[0]].device
        P = bufs[keys[0]].shape[0]
        sync(dev)
        t0 = time.perf_counter()
        if masks is None:
            # the kernel counts each rank's rows while it writes the masks: 4 bytes per rank to read back
            masks, counts = group_visibility(cams, bufs["means3D"], bufs["scales"], bufs["rotations"],
                                             [assign_views(len(cams), r, world, assignment, src, src_fewer, sizes)
                                              for r in range(world)], scaling_modifier, return_counts=True)
        else:
            masks = masks.to(dev)
            counts = masks.sum(dim=1)
        counts = counts.tolist()                    # the ONE host synchronisation of the distribution
        # (rank, id) pairs, rank-major, ids ascending; the size is known, so no second read-back inside nonzero
        pairs = torch.nonzero_static(masks, size=int(sum(counts)))
        sync(dev)
        t1 = time.perf_counter()
        trail = {k: tuple(bufs[k].shape[1:]) for k in keys}
        offsets = [0]
        for c in counts:
            offsets.append(offsets[-1] + c)
        native_pack = dev.type == "cuda" and len(keys) <= 8 and all(bufs[k].dtype == torch.float32 for k in keys)
        flat_idx = pairs.reshape(-1)

        def pack_rows(lo, hi):
            """[hi - lo, floats per Gaussian]: the packed rows pairs[lo:hi] (rank-major, ids ascending)"""
            if hi == lo:
                return torch.empty((0, sum(int(np.prod(trail[k], dtype=np.int64)) if trail[k] else 1 for k in keys)),
                                   dtype=torch.float32, device=dev)
            if native_pack:   # one pass: every packed row gathered straight from the tensors (bsr_pack_rows)
                from .rasterizer import _pack_rows_native
                return _pack_rows_native([bufs[k] for k in keys], flat_idx[2 * lo + 1:], idx_stride=2, rows=hi - lo)
            idx = pairs[lo:hi, 1].contiguous()              # CPU plumbing tests (gloo)
            return torch.cat([bufs[k].detach().index_select(0, idx).reshape(idx.numel(), -1).float() for k in keys],
                             dim=1).contiguous()

        flat_all = pack_rows(0, offsets[-1])
        sync(dev)
        info["pack_ms"] = (time.perf_counter() - t1) * 1e3
        info["filter_ms"] = (t1 - t0) * 1e3
        meta = [{"keys": keys, "trail": trail, "counts": counts}]
    if multi:
        dist.broadcast_object_list(meta, src=src)
    keys, trail, counts = meta[0]["keys"], meta[0]["trail"], meta[0]["counts"]
    widths = [int(np.prod(trail[k], dtype=np.int64)) if trail[k] else 1 for k in keys]
    row = sum(widths)
    if rank == src:
        mine = flat_all[offsets[rank]:offsets[rank + 1]]
    else:
        dev = torch.device(device) if device is not None else (
            cams.world_view.device if isinstance(cams, CameraPack) else cams[0].world_view_transform.device)
        mine = torch.empty((counts[rank], row), dtype=torch.float32, device=dev)
    if multi:
        sync(dev)
        t0 = time.perf_counter()
        # gloo moves host memory only (CPU test backend; two ranks on one GPU in the -m gpu tests): stage through it
        via_host = dist.get_backend() == "gloo" and dev.type == "cuda"
        landing = None
        ops = []
        if rank == src:
            wire = flat_all.cpu() if via_host else flat_all
            ops = [dist.P2POp(dist.isend, wire[offsets[r]:offsets[r + 1]], r) for r in range(world)
                   if r != src and counts[r] > 0]
        elif counts[rank] > 0:
            landing = torch.empty(mine.shape, dtype=torch.float32) if via_host else mine
            ops = [dist.P2POp(dist.irecv, landing, src)]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if landing is not None and via_host:
            mine.copy_(landing)
        sync(dev)
        info["comm_ms"] = (time.perf_counter() - t0) * 1e3
    local, off = {}, 0
    for k, w in zip(keys, widths):
        local[k] = mine[:, off:off + w].reshape((mine.shape[0],) + tuple(trail[k])).contiguous()
        off += w
    info["counts"] = list(counts)
    info["bytes"] = [c * row * 4 for c in counts]
    return local, my_views, info


@functools.lru_cache(maxsize=32)
def _group_ids(sizes, dev_str):
    """int32 device tensor [sum(sizes)]: group g repeated sizes[g] times (uploaded once per grouping, not per call)."""
    return torch.tensor([g for g, n in enumerate(sizes) for _ in range(n)], dtype=torch.int32).to(dev_str)


def group_visibility(cams, means3D, scales, rotations, groups, scaling_modifier: float = 1.0, debug=False,
                     return_counts=False):
    """bool [len(groups), P]: row g = "some camera of groups[g] (a list of indices into ``cams``) sees the Gaussian",
    with prefilter_voxel's test (GR:342-349), in ONE pass over the Gaussians (bsr_visible_filter_groups) -- P bytes per
    group written, instead of 4 P per view and the radii > 0 / any() passes behind them.  At most 64 groups.
    ``return_counts``: (masks, int32 [len(groups)] ones per row, on the device -- accumulated by the kernel)."""
    from .rasterizer import _rasterize_gaussians_filter_groups_native
    dev = means3D.device
    order = [i for g in groups for i in g]
    if not order:
        z = torch.zeros((len(groups), means3D.shape[0]), dtype=torch.bool, device=dev)
        return (z, torch.zeros((len(groups),), dtype=torch.int32, device=dev)) if return_counts else z
    vms, pms, _, c0 = _camera_stack(cams, order, dev)
    gid = _group_ids(tuple(len(members) for members in groups), str(dev))
    with torch.no_grad():
        return _rasterize_gaussians_filter_groups_native(
            means3D, scales[:, :3], rotations, scaling_modifier, torch.Tensor([]), vms, pms,
            math.tan(c0.FoVx * 0.5), math.tan(c0.FoVy * 0.5), int(c0.image_height), int(c0.image_width), gid,
            len(groups), debug, return_counts)


def allreduce_gradients(params, average: bool = True, force: bool = False) -> float:
    """Data-parallel training over views: every rank has run forward+backward
    on ITS view with the same Gaussian buffers; sum (or average) the ``.grad`` of ``params`` (an
    iterable of leaf tensors, or a dict of them) across the ranks.  All gradients travel as ONE flat
    fp32 buffer -- 248 B per Gaussian at SH degree 3 -- because a ring all-reduce over xGMI is bound
    by the per-link rate, so one large collective beats one per tensor.  A leaf without gradient
    contributes zeros (and receives the other ranks' sum).  Returns the elapsed milliseconds; a
    no-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return 0.0
    leaves = list(params.values()) if isinstance(params, dict) else list(params)
    if not leaves:
        return 0.0
    dev = leaves[0].device
    sizes = [p.numel() for p in leaves]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    off = 0
    for p, n in zip(leaves, sizes):
        if p.grad is not None:
            flat[off:off + n].copy_(p.grad.reshape(-1))
        off += n
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) * 1e3
    if average:
        flat /= dist.get_world_size()
    off = 0
    for p, n in zip(leaves, sizes):
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return ms


def make_settings(cam: MiniCam, bg_color, sh_degree, scaling_modifier=1.0, debug=False):
    """GaussianRasterizationSettings exactly as gaussian_renderer.render builds it (GR:232-248)."""
    from .rasterizer import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=debug)


def render_view(cam: MiniCam, gaussians: dict, bg_color, sh_degree=0, scaling_modifier=1.0, retain_grad=False,
                debug=False):
    """One view through the rasterizer with the result dict of gaussian_renderer.render
    (GR:264-291): keys render, viewspace_points, visibility_filter, radii, depth.
    ``gaussians``: means3D, opacities, scales, rotations and either shs or colors_precomp."""
    from .rasterizer import GaussianRasterizer
    xyz = gaussians["means3D"]
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    if retain_grad:
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    rasterizer = GaussianRasterizer(raster_settings=make_settings(cam, bg_color, sh_degree, scaling_modifier, debug))
    rendered_image, radii, depth = rasterizer(
        means3D=xyz, means2D=screenspace_points, shs=gaussians.get("shs"),
        colors_precomp=gaussians.get("colors_precomp"), opacities=gaussians["opacities"],
        scales=gaussians.get("scales"), rotations=gaussians.get("rotations"),
        cov3D_precomp=gaussians.get("cov3D_precomp"))
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": depth}


def render_neural(cam: MiniCam, anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot, bg_color,
                  scaling_modifier=1.0, retain_grad=False, debug=False, depth_gradient=False, fused=True, capacity=None,
                  settings=None):
    """gaussian_renderer.render for BloomScene's anchor representation (GR:211-291) from the point
    where the MLP heads have produced their outputs: fused anchor expansion (GR:165-203,
    ``neural_gaussians.expand_anchors``) -> rasterizer with ``colors_precomp`` and ``sh_degree=1``
    (GR:235-262).  Returns the training-mode result dict of GR:266-279 minus the entropy-coder rates
    (``bit_per_*`` come from the out-of-scope context model): render, viewspace_points,
    visibility_filter, radii, depth, selection_mask, neural_opacity, scaling.  ``fused`` (default): one native call
    each way (``neural_gaussians.render_anchors``); False: expand_anchors and the rasterizer as separate autograd nodes.
    Either way ``viewspace_points.grad`` holds the screen-space gradient after backward."""
    from .neural_gaussians import expand_anchors, render_anchors
    from .rasterizer import GaussianRasterizer
    if fused and not debug:
        # one native call for selection + expansion + rasterizer (and one for their backward): no interpreter time
        # between the blocking read of the selection count and the rasterizer's first launch
        # capacity (extension): static shapes + no host wait, see neural_gaussians.render_anchors; settings: a prebuilt
        # GaussianRasterizationSettings (a captured step must not build device tensors)
        image, depth, radii, mask, xyz, rgb, opacity, scaling, rot, viewspace = render_anchors(
            anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot,
            settings or make_settings(cam, bg_color, 1, scaling_modifier, debug), depth_gradient, capacity=capacity)
        return {"render": image, "viewspace_points": viewspace, "visibility_filter": radii > 0, "radii": radii,
                "depth": depth, "selection_mask": mask, "neural_opacity": neural_opacity, "scaling": scaling}
    # the same as two autograd nodes (what fused=True is tested against, bit for bit).  Everything that does not depend
    # on the number of selected Gaussians first: expand_anchors blocks the host on that count, and whatever python runs
    # between its return and the rasterizer's first launch is GPU idle time
    rasterizer = GaussianRasterizer(raster_settings=settings or make_settings(cam, bg_color, 1, scaling_modifier, debug),
                                    depth_gradient=depth_gradient, capacity=capacity)   # (capacity: the rasterizer half only)
    xyz, rgb, opacity, scaling, rot, mask = expand_anchors(anchor, grid_scaling, grid_offsets, neural_opacity, color,
                                                           scale_rot)
    # GR:224-229 builds `zeros_like(xyz, requires_grad=True) + 0` and calls retain_grad(): a leaf receives .grad as well,
    # without the second kernel and the extra autograd node
    screenspace_points = torch.zeros_like(xyz, dtype=anchor.dtype, requires_grad=True, device=xyz.device)
    rendered_image, radii, depth = rasterizer(means3D=xyz, means2D=screenspace_points, shs=None, colors_precomp=rgb,
                                              opacities=opacity, scales=scaling, rotations=rot, cov3D_precomp=None)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": depth, "selection_mask": mask, "neural_opacity": neural_opacity,
            "scaling": scaling}


def prefilter(cam: MiniCam, means3D, scales, rotations, bg_color, scaling_modifier=1.0, debug=False):
    """prefilter_voxel's rasterizer call (GR:312-349): visible_filter(...) > 0."""
    from .rasterizer import GaussianRasterizer
    rasterizer = GaussianRasterizer(raster_settings=make_settings(cam, bg_color, 1, scaling_modifier, debug))
    radii_pure = rasterizer.visible_filter(means3D=means3D, scales=scales[:, :3], rotations=rotations,
                                           cov3D_precomp=None)
    return radii_pure > 0


def training_view(cam: MiniCam, anchor, anchor_scaling, anchor_rotation, grid_offsets, heads, bg_color,
                  scaling_modifier=1.0, retain_grad=False, debug=False, depth_gradient=False):
    """One training iteration's pass through the rasterizer for ONE view, as bloomscene.py:240-243 drives it:
    ``prefilter_voxel`` on the anchors (GR:342-349), the per-anchor gathers of ``generate_neural_gaussians``
    (GR:33-43), [the caller's MLP heads on the visible anchors], the fused anchor expansion (GR:165-203) and the
    render (GR:254-262) -- with ONE rasterizer object / settings tuple for filter and render (the camera terms are
    derived once) and ONE native call that returns the filter's radii together with the index list of the visible
    anchors (``bsr_visible_filter_indices``), so the six boolean indexes of GR:33-43 become index gathers without
    their six nonzero() passes and host synchronisations (the per-iteration half).

    ``anchor [N,3]``, ``anchor_scaling [N,6]``, ``anchor_rotation [N,4]``, ``grid_offsets [N,K,3]`` are the full
    per-anchor parameters; ``heads(idx)`` is the caller's (out-of-scope) MLP evaluation and returns
    ``(neural_opacity [n*K,1], color [n*K,3], scale_rot [n*K,7])`` for the ``n`` visible anchors ``idx``.
    Returns render_neural's dict plus ``visible_mask`` (= prefilter_voxel's result), ``visible_idx`` and
    ``anchor_radii``; every output is bit-identical to the separate calls (tests/test_anchors_gpu.py)."""
    from .neural_gaussians import expand_anchors, render_anchors
    from .rasterizer import GaussianRasterizer
    rasterizer = GaussianRasterizer(raster_settings=make_settings(cam, bg_color, 1, scaling_modifier, debug),
                                    depth_gradient=depth_gradient)
    radii_pure, idx = rasterizer.visible_filter_indices(means3D=anchor, scales=anchor_scaling[:, :3],
                                                        rotations=anchor_rotation, cov3D_precomp=None)
    vis_anchor = anchor.index_select(0, idx)
    vis_scaling = anchor_scaling.index_select(0, idx)
    vis_offsets = grid_offsets.index_select(0, idx)
    neural_opacity, color, scale_rot = heads(idx)
    if not debug:   # selection + expansion + rasterizer: one native call each way (render_neural, fused=True)
        rendered_image, depth, radii, mask, xyz, rgb, opacity, scaling, rot, screenspace_points = render_anchors(
            vis_anchor, vis_scaling, vis_offsets, neural_opacity, color, scale_rot, rasterizer.raster_settings,
            depth_gradient)
    else:
        xyz, rgb, opacity, scaling, rot, mask = expand_anchors(vis_anchor, vis_scaling, vis_offsets, neural_opacity,
                                                               color, scale_rot)
        screenspace_points = torch.zeros_like(xyz, dtype=anchor.dtype, requires_grad=True, device=xyz.device)
        rendered_image, radii, depth = rasterizer(means3D=xyz, means2D=screenspace_points, shs=None,
                                                  colors_precomp=rgb, opacities=opacity, scales=scaling, rotations=rot,
                                                  cov3D_precomp=None)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": depth, "selection_mask": mask, "neural_opacity": neural_opacity,
            "scaling": scaling, "visible_mask": radii_pure > 0, "visible_idx": idx, "anchor_radii": radii_pure}


def prefilter_views(cams, means3D, scales, rotations, scaling_modifier=1.0, debug=False):
    """prefilter_voxel for a whole camera path at once: bool [V, P], row v equal
    to ``prefilter(cams[v], ...)``.  The reference filters the anchors once per view of the rotate360
    sweep (bloomscene.py:191-193 -> GR:342-349); here the anchors are read, and their 3-D covariances
    built, once for all V views.  All cameras must share image size and field of view (they do in
    the sweep: utils/trajectory.py:110-121)."""
    from .rasterizer import _rasterize_gaussians_filter_views_native
    if not len(cams):
        return torch.zeros((0, means3D.shape[0]), dtype=torch.bool, device=means3D.device)
    dev = means3D.device
    vms, pms, _, c0 = _camera_stack(cams, range(len(cams)), dev)
    with torch.no_grad():
        radii = _rasterize_gaussians_filter_views_native(
            means3D, scales[:, :3], rotations, scaling_modifier, torch.Tensor([]), vms, pms,
            math.tan(c0.FoVx * 0.5), math.tan(c0.FoVy * 0.5), int(c0.image_height), int(c0.image_width), debug)
    return radii > 0


def render_views_batched(cams, gaussians: dict, bg_color, sh_degree=0, scaling_modifier=1.0, debug=False, idx=None):
    """Inference forward of several cameras in ONE native call (bsr_forward_views): returns
    (frames [V,3,H,W], depths [V,1,H,W], radii [V,P]), view by view bit-identical to ``render_view``.
    The views of a camera sweep see few Gaussians each, so one at a time they are launch/latency bound; batched,
    binning, per-tile sort and render run once over all of them.  ``cams``: a list of MiniCams or a ``CameraPack``
    (matrices already stacked on the device); ``idx``: which of them (default: all).  Cameras must share image size
    and field of view (they do in the rotate360 sweep: utils/trajectory.py:110-121).  No gradients."""
    from .rasterizer import _rasterize_gaussians_views_native
    from .numerics import resolve_flags
    xyz = gaussians["means3D"]
    dev = xyz.device
    idx = list(range(len(cams))) if idx is None else list(idx)
    if not idx:
        raise ValueError("render_views_batched needs at least one camera")
    if (gaussians.get("shs") is None) == (gaussians.get("colors_precomp") is None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    e = torch.Tensor([])

    def opt(k):
        v = gaussians.get(k)
        return e if v is None else v
    vms, pms, cps, c0 = _camera_stack(cams, idx, dev)
    with torch.no_grad():
        _, color, depth, radii = _rasterize_gaussians_views_native(
            bg_color, xyz, opt("colors_precomp"), gaussians["opacities"], opt("scales"), opt("rotations"),
            scaling_modifier, opt("cov3D_precomp"), vms, pms, math.tan(c0.FoVx * 0.5), math.tan(c0.FoVy * 0.5),
            int(c0.image_height), int(c0.image_width), opt("shs"), sh_degree, cps, False, debug,
            flags=resolve_flags())   # (the calling thread's numerics context; default: the fast path)
    return color, depth, radii


def compact_for_view_groups(cams, gaussians: dict, groups, scaling_modifier: float = 1.0):
    """One dict of per-Gaussian tensors per group of views, holding only the rows some view of the group can see
    (ascending ids, so every frame rendered from it is bit-identical to the frame rendered from all rows): ONE pass of
    the group filter over the Gaussians, a 4-byte-per-group read-back, one gather over all tensors.  This is the
    reference's prefilter_voxel step (GR:342-349: render only what the visibility filter kept) for a whole camera
    path at once.  A group that sees nothing gets row 0 alone (a culled Gaussian contributes nothing; P = 0 would take
    the reference's zero-image path, RP:68-82, instead of the background).  At most 64 groups per filter pass."""
    from .rasterizer import _gather_rows_native
    keys = [k for k, v in gaussians.items() if torch.is_tensor(v) and v.numel() > 0]
    P = gaussians["means3D"].shape[0]
    subsets = []
    for g0 in range(0, len(groups), 64):
        chunk = groups[g0:g0 + 64]
        masks, counts = group_visibility(cams, gaussians["means3D"], gaussians["scales"], gaussians["rotations"], chunk,
                                         scaling_modifier, return_counts=True)
        counts = counts.tolist()
        pairs = torch.nonzero_static(masks, size=int(sum(counts)))
        rows = _gather_rows_native([gaussians[k] for k in keys], pairs.reshape(-1)[1:], idx_stride=2,
                                   rows=pairs.shape[0], packed=False) if pairs.shape[0] else None
        off = 0
        for n in counts:
            if n == 0:
                subsets.append({k: gaussians[k][:min(P, 1)] for k in keys})
            else:
                subsets.append({k: t[off:off + n] for k, t in zip(keys, rows)})
            off += n
    return subsets


def render_views_sharded(cams, gaussians: dict, bg_color, sh_degree, rank=None, world=None, keep_outputs=False,
                         batch=1, views=None, compact=False):
    """The rotate360 loop of BloomScene.render_video (reference bloomscene.py:191-211), sharded:
    this rank renders its round-robin share of ``cams`` with torch.no_grad() and returns
    {view index: (frame [3,H,W], depth [1,H,W])} (or only the indices when not keeping outputs).
    ``batch`` > 1 renders that many of the rank's views per native call (``render_views_batched``).
    ``views``: this rank's view indices when they are not the round-robin share (``scatter_visible_gaussians``).
    ``cams`` may be a ``CameraPack`` when ``batch`` > 1 (no per-call stacking and upload of the matrices).
    ``compact`` (with ``batch`` > 1, scales + rotations given): every batch is rendered from the rows its views can see
    (``compact_for_view_groups``) -- pays on sweeps whose views each see a small part of the scene (rotate360: 15 %),
    costs a filter pass and a gather where every view sees everything.  Frames are bit-identical either way."""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    out = {}
    dev = gaussians["means3D"].device
    mine = list(shard_views(len(cams), rank, world)) if views is None else list(views)
    with torch.no_grad():
        if batch > 1:
            batches = [mine[b0:b0 + batch] for b0 in range(0, len(mine), batch)]
            compact = (compact and bool(batches) and dev.type == "cuda" and gaussians["means3D"].shape[0] > 0
                       and gaussians.get("scales") is not None and gaussians.get("rotations") is not None)
            subsets = compact_for_view_groups(cams, gaussians, batches) if compact else None
            for b, idx in enumerate(batches):
                color, depth, _ = render_views_batched(cams, subsets[b] if compact else gaussians, bg_color, sh_degree,
                                                       idx=idx)
                for k, i in enumerate(idx):
                    out[i] = (color[k], depth[k]) if keep_outputs else None
            return out
        for i in mine:
            cam = cams[i].to(dev)
            res = render_view(cam, gaussians, bg_color, sh_degree)
            out[i] = (res["render"], res["depth"]) if keep_outputs else None
    return out



