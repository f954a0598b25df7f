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


def _scatter_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, M = 3000, 4
        g = torch.Generator().manual_seed(4321)
        src = {"means3D": torch.randn(P, 3, generator=g), "scales": torch.rand(P, 3, generator=g),
               "rotations": torch.randn(P, 4, generator=g), "opacities": torch.rand(P, 1, generator=g),
               "shs": torch.randn(P, M, 3, generator=g)}
        masks = torch.rand(world, P, generator=g) < torch.tensor([[0.1], [0.3], [0.0]])[:world]   # rank 2: nothing visible
        cams = [helpers.scene_b(1, 64, 48, 0, n_views=10).cameras[i] for i in range(10)]
        local, mine, info = views.scatter_visible_gaussians(src if rank == 0 else None, cams, src=0,
                                                            assignment="contiguous", masks=masks if rank == 0 else None)
        ok = mine == views.assign_views(10, rank, world, "contiguous")
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


def test_view_assignments():
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
