"""EXPERIMENTAL forms of the view-parallel distribution -- validated on gloo / one GPU only, never on an RCCL group of more
than one rank ("freeze multi-GPU at one path until hardware speaks").  The product path is
``bloomscene_amd.views``: the packed broadcast of SURVEY.md 8(e) (``broadcast_gaussians``) and ONE visible-subset form
(``scatter_visible_gaussians``: one filter, one pack, all sends posted as one group).  Kept here, out of the default
bench and of the package's public names:

* ``scatter_visible_gaussians_pipelined`` -- the same distribution with the source packing and sending rank by rank,
* ``scatter_visible_gaussians_blockwise`` -- the source pipelined block by block, filters included,
* ``balanced_block_sizes`` -- uneven view blocks for ranks that are served late,
* ``visible_rows_per_rank`` / ``modelled_scatter_sweep`` -- the latency model ``bench.py --experimental`` prints for a
  multi-GPU record to falsify.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.distributed as dist

from ..views import CameraPack, assign_views, group_visibility


def balanced_block_sizes(n_views: int, world: int, leave_ms, per_view_ms: float, src: int = 0):
    """Contiguous view-block sizes that let all ranks FINISH together when they cannot START together: rank r can begin
    rendering at leave_ms[r] (for a peer: when its rows have arrived; for the distributing rank: when it has packed the
    last block), and every view costs per_view_ms.  Views are dealt one at a time to the rank that would finish
    earliest (water-filling); ties go to the lower rank.  A rank served late gets fewer views."""
    sizes = [0] * world
    finish = [float(t) for t in leave_ms]
    for _ in range(n_views):
        r = min(range(world), key=lambda k: (finish[k] + per_view_ms, k))
        sizes[r] += 1
        finish[r] += per_view_ms
    return sizes


def scatter_visible_gaussians_pipelined(bufs, cams, src: int = 0, assignment: str = "contiguous", masks=None,
                              scaling_modifier: float = 1.0, device=None, pipelined: bool = True, src_fewer: int = 0,
                              sizes=None):
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

    ``pipelined=False`` (default, ADVICE r4: until a multi-GPU record says otherwise the sends are posted as ONE group):
    one pack of all rows, then all sends together (round 3's form).  ``pipelined=True``: `src` packs ONE rank's rows at a
    time and posts that rank's send as soon as they are packed
    -- remote ranks first, in rank order, its own block last -- so that the first peer's rows are on the wire while the
    next peer's are still being gathered (the pack kernels run on the compute stream, RCCL's sends on its own); every
    peer starts rendering as soon as ITS message has arrived (unbatched point-to-point sends: on RCCL their first use
    builds the peer connections, and whether sends to different peers overlap is for the first multi-GPU record to
    show).  The source pipelined block by block, filters included: scatter_visible_gaussians_blockwise.  ``src_fewer`` / ``sizes``: uneven view blocks (staggered_block_sizes / balanced_block_sizes; the
    same on every rank): a rank whose rows leave late gets fewer views.

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
    info = {"filter_ms": 0.0, "pack_ms": 0.0, "comm_ms": 0.0, "pipelined": bool(pipelined)}

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
                from ..rasterizer import _pack_rows_native
                return _pack_rows_native([bufs[k] for k in keys], flat_idx[2 * lo + 1:], idx_stride=2, rows=hi - lo)
            idx = pairs[lo:hi, 1].contiguous()              # CPU plumbing tests (gloo)
            return torch.cat([bufs[k].detach().index_select(0, idx).reshape(idx.numel(), -1).float() for k in keys],
                             dim=1).contiguous()

        if not (multi and pipelined):
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
        mine = None if (multi and pipelined) else flat_all[offsets[rank]:offsets[rank + 1]]
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
        if rank == src and pipelined:
            # one rank at a time: pack -> post the send -> pack the next one; own block last.  The order in which the
            # sends were posted is returned for the tests (info["send_order"]).
            pending, order, keep = [], [], []
            for r in [q for q in range(world) if q != src] + [src]:
                part = pack_rows(offsets[r], offsets[r + 1])
                if r == src:
                    mine = part
                elif counts[r] > 0:
                    wire = part.cpu() if via_host else part
                    keep.append(wire)                   # (alive until its send has completed)
                    pending.append(dist.isend(wire, r))
                    order.append(r)
            sync(dev)
            info["pack_ms"] = (time.perf_counter() - t0) * 1e3   # packing with the sends already under way
            for w in pending:
                w.wait()
            info["send_order"] = order
        else:
            ops = []
            if rank == src:
                wire = flat_all.cpu() if via_host else flat_all
                ops = [dist.P2POp(dist.isend, wire[offsets[r]:offsets[r + 1]], r) for r in range(world)
                       if r != src and counts[r] > 0]
            elif counts[rank] > 0:
                landing = torch.empty(mine.shape, dtype=torch.float32) if via_host else mine
                if pipelined:
                    dist.irecv(landing, src).wait()
                else:
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


def scatter_visible_gaussians_blockwise(bufs, cams, src: int = 0, masks=None, scaling_modifier: float = 1.0, device=None,
                                        src_fewer: int = 0, sizes=None, layout=None):
    """scatter_visible_gaussians with the SOURCE pipelined block by block (round 5): instead of one visibility pass for
    all cameras followed by the packs, `src` filters ONE rank's view block at a time (bsr_visible_filter_groups with a
    single group), and while block k's rows are counted, packed and sent, the filter of block k + 1 is already enqueued
    behind it (the GPU never waits for the host's count read-back) -- the first peer's message leaves after one
    eighth of the filter work instead of all of it.  Remote blocks first, in rank order; the source's own block last.

    The row count of a block is known only when its filter has run, so every peer first receives an 8-byte header
    (its row count), then -- if the count is not zero -- its rows; the
    tensor layout (names and trailing shapes) is static and travels ahead of everything else in the one object
    broadcast of the call, which the source issues AFTER it has enqueued the first two filters (``layout``: the dict
    {name: trailing shape} if every rank already knows it -- then there is no collective at all in front of the sends).

    Assumptions the stated model (modelled_scatter_sweep(..., pipelined="blocks")) makes about RCCL, for the first
    multi-GPU record to falsify: point-to-point sends to different peers run on different xGMI links at the same time;
    the lazily built peer connections exist (bench.py runs the distribution twice and reports the second);
    a send posted while an earlier one is on the wire starts at once (one communicator stream per peer).

    Same results as scatter_visible_gaussians: (local_bufs, my_views, info); info["counts"] is complete on `src` only
    (a peer knows its own count)."""
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not multi:
        return scatter_visible_gaussians_pipelined(bufs, cams, src, "contiguous", masks, scaling_modifier, device, True, src_fewer,
                                                   sizes)
    rank, world = dist.get_rank(), dist.get_world_size()
    blocks = [assign_views(len(cams), r, world, "contiguous", src, src_fewer, sizes) for r in range(world)]
    my_views = blocks[rank]
    info = {"filter_ms": 0.0, "pack_ms": 0.0, "comm_ms": 0.0, "pipelined": "blocks"}
    order = [q for q in range(world) if q != src] + [src]
    via_host_backend = dist.get_backend() == "gloo"
    t_start = time.perf_counter()
    if rank == src:
        keys = sorted(bufs)
        dev = bufs[keys[0]].device
        via_host = via_host_backend and dev.type == "cuda"
        trail = {k: tuple(bufs[k].shape[1:]) for k in keys}

        def enqueue_filter(r):
            """(mask bool [P], count landing in host memory, event behind the count's copy) of rank r's block; nothing is
            waited for here.  The count travels to PINNED memory behind its own event, so that reading count i waits for
            filter i alone -- a plain .item() would wait for everything enqueued since, the next filter included."""
            if masks is not None:
                m = masks[r].to(dev)
                c = m.sum().reshape(1)
            elif not blocks[r]:
                m = torch.zeros(bufs[keys[0]].shape[0], dtype=torch.bool, device=dev)
                c = torch.zeros(1, dtype=torch.int32, device=dev)
            else:
                mm, c = group_visibility(cams, bufs["means3D"], bufs["scales"], bufs["rotations"], [blocks[r]],
                                         scaling_modifier, return_counts=True)
                m = mm[0]
            if dev.type != "cuda":
                return m, c, None
            landing = torch.empty(1, dtype=c.dtype, pin_memory=True)
            landing.copy_(c, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            return m, landing, ev
        ahead = [enqueue_filter(r) for r in order[:2]]          # two filters in flight before the host looks at anything
        if layout is None:
            dist.broadcast_object_list([{"keys": keys, "trail": trail}], src=src)
        native_pack = dev.type == "cuda" and len(keys) <= 8 and all(bufs[k].dtype == torch.float32 for k in keys)
        counts, keep, pending, sent = [0] * world, [], [], []
        mine = None
        for i, r in enumerate(order):
            m, c, ev = ahead[i]
            if ev is not None:
                ev.synchronize()                               # waits for filter i only; filter i + 1 is already enqueued
            n = int(c.item())
            if i + 2 < len(order):
                ahead.append(enqueue_filter(order[i + 2]))
            counts[r] = n
            idx = torch.nonzero_static(m.reshape(1, -1), size=n)[:, 1].contiguous() if n else None
            if n == 0:
                part = torch.empty((0, sum(int(np.prod(trail[k], dtype=np.int64)) if trail[k] else 1 for k in keys)),
                                   dtype=torch.float32, device=dev)
            elif native_pack:
                from ..rasterizer import _pack_rows_native
                part = _pack_rows_native([bufs[k] for k in keys], idx, idx_stride=1, rows=n)
            else:
                part = torch.cat([bufs[k].detach().index_select(0, idx).reshape(n, -1).float() for k in keys], dim=1).contiguous()
            if r == src:
                mine = part
            else:
                header = torch.tensor([n], dtype=torch.int64)
                header = header if via_host_backend else header.to(dev)
                wire = part.cpu() if via_host else part
                keep += [header, wire]
                pending.append(dist.isend(header, r))           # (two plain sends, mirrored by the peer's two receives)
                if n:
                    pending.append(dist.isend(wire, r))
                sent.append(r)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        info["pack_ms"] = (time.perf_counter() - t_start) * 1e3   # filters, packs and sends interleaved
        for w in pending:
            w.wait()
        info["send_order"] = sent
        info["counts"] = counts
    else:
        if layout is None:
            meta = [None]
            dist.broadcast_object_list(meta, src=src)
            keys, trail = meta[0]["keys"], meta[0]["trail"]
        else:
            keys, trail = sorted(layout), {k: tuple(v) for k, v in layout.items()}
        dev = torch.device(device) if device is not None else (
            cams.world_view.device if isinstance(cams, CameraPack) else cams[0].world_view_transform.device)
        via_host = via_host_backend and dev.type == "cuda"
        header = torch.zeros(1, dtype=torch.int64, device="cpu" if via_host_backend else dev)
        dist.recv(header, src)
        n = int(header.item())
        row = sum(int(np.prod(trail[k], dtype=np.int64)) if trail[k] else 1 for k in keys)
        mine = torch.empty((n, row), dtype=torch.float32, device=dev)
        if n:
            landing = torch.empty(mine.shape, dtype=torch.float32) if via_host else mine
            dist.recv(landing, src)
            if via_host:
                mine.copy_(landing)
        info["counts"] = [n if r == rank else None for r in range(world)]
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    info["comm_ms"] = (time.perf_counter() - t_start) * 1e3
    widths = [int(np.prod(trail[k], dtype=np.int64)) if trail[k] else 1 for k in keys]
    local, off = {}, 0
    for k, w in zip(keys, widths):
        local[k] = mine[:, off:off + w].reshape((mine.shape[0],) + tuple(trail[k])).contiguous()
        off += w
    info["bytes"] = [None if c is None else c * sum(widths) * 4 for c in info["counts"]]
    return local, my_views, info


def visible_rows_per_rank(bufs, cams, worlds=(1, 2, 4, 8), assignment: str = "contiguous", scaling_modifier: float = 1.0,
                          src_fewer: int = 0, sizes=None):
    """{world: [rows rank 0 .. world-1 would receive from scatter_visible_gaussians]} for several node sizes
    (bench.py's scaling prediction): one pass of the per-group visibility filter per node size."""
    out = {}
    for w in worlds:
        groups = [assign_views(len(cams), r, w, assignment, 0, src_fewer, sizes) for r in range(w)]
        _, c = group_visibility(cams, bufs["means3D"], bufs["scales"], bufs["rotations"], groups, scaling_modifier,
                                return_counts=True)
        out[int(w)] = [int(x) for x in c.tolist()]
    return out


def modelled_scatter_sweep(n_views: int, world: int, rows, sizes, filter_ms: float, pack_ms_per_row: float,
                           row_bytes: int, per_view_ms: float, link_GBs: float = 153.0, pipelined=True,
                           filter_block_ms=None):
    """Critical path (ms) of a cold view-parallel sweep through scatter_visible_gaussians on `world` ranks, from measured
    single-GPU stage times -- the model bench.py states for the first multi-GPU record to falsify:

      the source (rank 0) filters for `filter_ms` (group filter + the count read-back + the pair list), then packs rank
      by rank (rows_r x pack_ms_per_row each), remote ranks first and its own block last; pipelined: rank r's message
      leaves when ITS rows are packed (otherwise when all are), needs rows_r x row_bytes / link rate on its own xGMI link
      (point-to-point mesh: the links do not share bandwidth), and rank r then renders its sizes[r] views at
      per_view_ms each; the source renders after its last pack.

    Returns {"sweep_ms", "critical_rank", "finish_ms": [...]}.  world == 1: no distribution at all."""
    if world == 1:
        t = n_views * per_view_ms
        return {"sweep_ms": t, "critical_rank": 0, "finish_ms": [t]}
    pack = [rows[r] * pack_ms_per_row for r in range(world)]
    total_pack = sum(pack)
    finish = [0.0] * world
    if pipelined == "blocks":
        # scatter_visible_gaussians_blockwise: the source's stream runs F(1) F(2) P(1) F(3) P(2) ... F(0) P(N-1) P(0) -- one
        # filter per view block (filter_block_ms[r]: its launch over all Gaussians for block r's views), always one filter
        # ahead of the pack whose count the host is reading; rank r's header + rows leave when P(r) is done
        fb = list(filter_block_ms)
        order = list(range(1, world)) + [0]
        t_gpu, fdone = 0.0, 0
        for i, r in enumerate(order):
            while fdone < min(i + 2, world):          # filters enqueued ahead of pack i
                t_gpu += fb[order[fdone]]
                fdone += 1
            t_gpu += pack[r]
            if r != 0:
                finish[r] = t_gpu + rows[r] * row_bytes / (link_GBs * 1e6) + sizes[r] * per_view_ms
        finish[0] = t_gpu + sizes[0] * per_view_ms
        worst = max(range(world), key=lambda r: finish[r])
        return {"sweep_ms": finish[worst], "critical_rank": worst, "finish_ms": finish}
    done = filter_ms
    for r in range(1, world):
        done += pack[r]
        leave = done if pipelined else filter_ms + total_pack
        finish[r] = leave + rows[r] * row_bytes / (link_GBs * 1e6) + sizes[r] * per_view_ms
    finish[0] = filter_ms + total_pack + sizes[0] * per_view_ms
    worst = max(range(world), key=lambda r: finish[r])
    return {"sweep_ms": finish[worst], "critical_rank": worst, "finish_ms": finish}
