"""N > 1 path on CPU: world_size-2 `gloo` processes exercise the view sharding and the packed
broadcast of the Gaussian buffers (the only collective of the path; RCCL on the GPU box)."""
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


def test_broadcast_is_a_noop_without_process_group():
    b = {"means3D": torch.ones(3, 3)}
    assert views.broadcast_gaussians(b) == 0.0 and torch.equal(b["means3D"], torch.ones(3, 3))
