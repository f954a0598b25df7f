"""N > 1 path on CPU: world_size-2 `gloo` processes exercise the view sharding and the packed
broadcast of the Gaussian buffers (the only collective of the inference path; RCCL on the GPU
box) and the packed gradient all-reduce of data-parallel training over views (§8f rank 3)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers  # noqa: F401  (sys.path)
from bloomscene_amd import views
from bloomscene_amd.experimental import view_distribution as XD   # the forms kept out of the product path (gloo-tested only)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, M = 1000, 4
        g = torch.Generator().manual_seed(1234)
        src = {"means3D": torch.randn(P, 3, generator=g), "scales": torch.rand(P, 3, generator=g),
               "rotations": torch.randn(P, 4, generator=g), "opacities": torch.rand(P, 1, generator=g),
               "shs": torch.randn(P, M, 3, generator=g)}
        bufs = {k: (v.clone() if rank == 0 else torch.full_like(v, float("nan"))) for k, v in src.items()}
        ms = views.broadcast_gaussians(bufs, src=0)
        ok = all(torch.equal(bufs[k], src[k]) for k in src) and ms >= 0.0
        mine = views.shard_views(64, rank, world)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        flat = sorted(i for part in gathered for i in part)
        ok = ok and flat == list(range(64)) and mine == list(range(rank, 64, world))
        # data-parallel training step: per-rank gradients -> one packed all-reduce (sum and mean)
        leaves = {k: v.clone().requires_grad_(True) for k, v in src.items()}
        for i, (k, v) in enumerate(sorted(leaves.items())):
            if not (rank == 1 and k == "scales"):      # rank 1 has no gradient for `scales`
                v.grad = torch.full_like(v, float(rank + 1) * (i + 1))
        views.allreduce_gradients(leaves, average=False)
        for i, (k, v) in enumerate(sorted(leaves.items())):
            want = (1.0 if k == "scales" else 3.0) * (i + 1)
            ok = ok and v.grad is not None and bool((v.grad == want).all())
        views.allreduce_gradients(list(leaves.values()), average=True)   # all ranks now hold equal grads
        for i, (k, v) in enumerate(sorted(leaves.items())):
            want = (1.0 if k == "scales" else 3.0) * (i + 1)
            ok = ok and bool((v.grad == want).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_view_sharding_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _scatter_worker(rank, world, port, q, pipelined=False, src_fewer=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, M = 3000, 4
        g = torch.Generator().manual_seed(4321)
        src = {"means3D": torch.randn(P, 3, generator=g), "scales": torch.rand(P, 3, generator=g),
               "rotations": torch.randn(P, 4, generator=g), "opacities": torch.rand(P, 1, generator=g),
               "shs": torch.randn(P, M, 3, generator=g)}
        masks = torch.rand(world, P, generator=g) < torch.tensor([[0.1], [0.3], [0.0], [0.2]])[:world]   # rank 2: nothing visible
        cams = [helpers.scene_b(1, 64, 48, 0, n_views=10).cameras[i] for i in range(10)]
        if pipelined:   # the experimental form: the source packs and sends rank by rank
            local, mine, info = XD.scatter_visible_gaussians_pipelined(src if rank == 0 else None, cams, src=0,
                                                                       assignment="contiguous",
                                                                       masks=masks if rank == 0 else None, pipelined=True,
                                                                       src_fewer=src_fewer)
        else:           # the product form: one pack, all sends posted as one group
            local, mine, info = views.scatter_visible_gaussians(src if rank == 0 else None, cams, src=0,
                                                                assignment="contiguous",
                                                                masks=masks if rank == 0 else None, src_fewer=src_fewer)
        ok = mine == views.assign_views(10, rank, world, "contiguous", 0, src_fewer)
        if rank == 0 and pipelined:   # remote ranks in rank order (a rank that sees nothing gets no message), own block last
            ok = ok and info["send_order"] == [r for r in range(1, world) if int(masks[r].sum()) > 0]
        ok = ok and info["counts"] == [int(m.sum()) for m in masks] and info["bytes"][rank] == info["counts"][rank] * (3 + 3 + 4 + 1 + 3 * M) * 4
        for k, v in src.items():
            ok = ok and torch.equal(local[k], v[masks[rank]]) and local[k].is_contiguous()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_scatter_of_per_rank_visible_subsets_world3():
    """views.scatter_visible_gaussians over gloo with given masks (the native visibility filter needs the GPU): every
    rank ends with exactly its mask's rows, in ascending id order, through ONE packed point-to-point message per rank;
    a rank that sees nothing gets empty tensors and no message."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scatter_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True), (2, True)]


def test_pipelined_scatter_world4_and_the_batched_form():
    """The pipelined distribution (one pack + one send per rank, posted as soon as that rank's rows are packed, the
    source's own block last) over gloo with FOUR ranks, uneven view blocks (the source takes two views fewer), and the
    round-3 form (one pack, all sends together) beside it: same rows on every rank either way."""
    for world, pipelined, src_fewer in ((4, True, 2), (4, False, 0), (4, False, 2), (2, True, 1)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_scatter_worker, args=(r, world, port, q, pipelined, src_fewer)) for r in range(world)]
        for p in procs:
            p.start()
        results = [q.get(timeout=120) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert sorted(results) == [(r, True) for r in range(world)], (world, pipelined, src_fewer, results)


def _blockwise_worker(rank, world, port, q, src_fewer, with_layout):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, M = 3000, 4
        g = torch.Generator().manual_seed(4321)
        src = {"means3D": torch.randn(P, 3, generator=g), "scales": torch.rand(P, 3, generator=g),
               "rotations": torch.randn(P, 4, generator=g), "opacities": torch.rand(P, 1, generator=g),
               "shs": torch.randn(P, M, 3, generator=g)}
        masks = torch.rand(world, P, generator=g) < torch.tensor([[0.1], [0.3], [0.0], [0.2]])[:world]   # rank 2: nothing visible
        cams = [helpers.scene_b(1, 64, 48, 0, n_views=10).cameras[i] for i in range(10)]
        layout = {k: tuple(v.shape[1:]) for k, v in src.items()} if with_layout else None
        local, mine, info = XD.scatter_visible_gaussians_blockwise(
            src if rank == 0 else None, cams, src=0, masks=masks if rank == 0 else None, src_fewer=src_fewer, layout=layout)
        ok = mine == views.assign_views(10, rank, world, "contiguous", 0, src_fewer)
        if rank == 0:   # every peer gets its header (also the one that sees nothing), in rank order; own block last
            ok = ok and info["send_order"] == list(range(1, world)) and info["counts"] == [int(m.sum()) for m in masks]
        else:
            ok = ok and info["counts"][rank] == int(masks[rank].sum())
        ok = ok and info["pipelined"] == "blocks"
        for k, v in src.items():
            ok = ok and torch.equal(local[k], v[masks[rank]]) and local[k].is_contiguous()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_blockwise_scatter_world4():
    """scatter_visible_gaussians_blockwise (the source filters, packs and sends ONE view block at a time, an 8-byte
    header ahead of every peer's rows) over gloo with four ranks: the same rows on every rank as the batched form, with
    and without the static layout handed to the peers (no collective in front of the sends), even and uneven blocks."""
    for world, src_fewer, with_layout in ((4, 0, False), (4, 2, True), (2, 1, True)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_blockwise_worker, args=(r, world, port, q, src_fewer, with_layout)) for r in range(world)]
        for p in procs:
            p.start()
        results = [q.get(timeout=120) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert sorted(results) == [(r, True) for r in range(world)], (world, src_fewer, with_layout, results)


def test_modelled_blockwise_sweep():
    """The stated critical path of the blockwise pipeline: F(1) F(2) P(1) F(3) P(2) ... on the source's stream."""
    rows, sizes = [100, 200, 300], [2, 3, 3]
    m = XD.modelled_scatter_sweep(8, 3, rows, sizes, filter_ms=9.0, pack_ms_per_row=0.01, row_bytes=100, per_view_ms=1.0,
                                     link_GBs=0.1, pipelined="blocks", filter_block_ms=[0.5, 0.25, 0.125])
    # order 1, 2, 0: t = F1 + F2 = 0.375; + P1 = 2.375 -> rank 1 leaves; + F0 = 2.875; + P2 = 5.875 -> rank 2 leaves; + P0 = 6.875
    wire = [r * 100 / (0.1 * 1e6) for r in rows]
    assert abs(m["finish_ms"][1] - (2.375 + wire[1] + 3.0)) < 1e-9
    assert abs(m["finish_ms"][2] - (5.875 + wire[2] + 3.0)) < 1e-9
    assert abs(m["finish_ms"][0] - (6.875 + 2.0)) < 1e-9
    assert m["critical_rank"] == max(range(3), key=lambda r: m["finish_ms"][r])


def test_view_assignments():
    # uneven contiguous blocks: the distributing rank takes fewer views, its peers share them out; still a partition
    assert views.staggered_block_sizes(64, 8, 0, 0) == [8] * 8
    assert views.staggered_block_sizes(64, 8, 0, 3) == [5, 9, 9, 9, 8, 8, 8, 8]
    assert views.staggered_block_sizes(10, 4, 0, 2) == [1, 4, 4, 1]
    assert views.staggered_block_sizes(64, 1, 0, 5) == [64]
    for n, world, fewer in ((64, 8, 3), (10, 4, 2), (7, 3, 9)):
        parts = [views.assign_views(n, r, world, "contiguous", 0, fewer) for r in range(world)]
        assert [i for p in parts for i in p] == list(range(n)), (n, world, fewer)
    for n, world in ((64, 8), (10, 4), (7, 3), (3, 5)):
        for mode in ("round_robin", "contiguous"):
            parts = [views.assign_views(n, r, world, mode) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n)), (n, world, mode)
    assert views.assign_views(64, 3, 8, "contiguous") == list(range(24, 32))
    assert views.assign_views(64, 3, 8, "round_robin") == views.shard_views(64, 3, 8)


def test_broadcast_is_a_noop_without_process_group():
    b = {"means3D": torch.ones(3, 3)}
    assert views.broadcast_gaussians(b) == 0.0 and torch.equal(b["means3D"], torch.ones(3, 3))
    p = torch.ones(3, requires_grad=True)
    p.grad = torch.full((3,), 2.0)
    assert views.allreduce_gradients([p]) == 0.0 and torch.equal(p.grad, torch.full((3,), 2.0))


def test_modelled_scatter_sweep_critical_path():
    """views.modelled_scatter_sweep (the model bench.py states for a multi-GPU record to falsify) on numbers small enough
    to follow by hand: filter 1.0; packs 0.1 per 1000 rows; 1000 rows = 1 MB on a 1 GB/s link = 1.0; 0.5 per view."""
    kw = dict(filter_ms=1.0, pack_ms_per_row=1e-4, row_bytes=1000, per_view_ms=0.5, link_GBs=1.0)
    one = XD.modelled_scatter_sweep(8, 1, [5000], [8], **kw)
    assert one["sweep_ms"] == 4.0
    rows, sizes = [1000, 2000, 1000, 3000], [2, 2, 2, 2]
    m = XD.modelled_scatter_sweep(8, 4, rows, sizes, pipelined=True, **kw)
    # rank 1 leaves at 1.0 + 0.2, rank 2 at + 0.1, rank 3 at + 0.3; wire 2.0 / 1.0 / 3.0; render 1.0 each; the source:
    # 1.0 + 0.7 + 1.0
    assert [round(x, 6) for x in m["finish_ms"]] == [2.7, 4.2, 3.3, 5.6] and m["critical_rank"] == 3
    b = XD.modelled_scatter_sweep(8, 4, rows, sizes, pipelined=False, **kw)
    assert [round(x, 6) for x in b["finish_ms"]] == [2.7, 4.7, 3.7, 5.7]      # every message leaves after ALL packs
    # fewer views for the rank on the critical path shortens it
    m2 = XD.modelled_scatter_sweep(8, 4, rows, [3, 2, 2, 1], pipelined=True, **kw)
    assert round(m2["sweep_ms"], 6) == 5.1


def test_balanced_block_sizes_deal_fewer_views_to_ranks_that_start_late():
    sizes = XD.balanced_block_sizes(64, 8, [0.53, 0.41, 0.45, 0.48, 0.51, 0.55, 0.58, 0.62], 0.05)
    assert sum(sizes) == 64 and sizes[1] == max(sizes) and sizes[7] == min(sizes) and max(sizes) - min(sizes) >= 3
    finish = [t + 0.05 * v for t, v in zip([0.53, 0.41, 0.45, 0.48, 0.51, 0.55, 0.58, 0.62], sizes)]
    assert max(finish) - min(finish) <= 0.05 + 1e-9          # all ranks finish within one view of each other
    assert XD.balanced_block_sizes(10, 2, [0.0, 0.0], 1.0) == [5, 5]
    parts = [views.assign_views(64, r, 8, "contiguous", sizes=sizes) for r in range(8)]
    assert [i for p in parts for i in p] == list(range(64)) and [len(p) for p in parts] == sizes
    import pytest
    with pytest.raises(ValueError):
        views.assign_views(64, 0, 8, "contiguous", sizes=[8] * 7)
