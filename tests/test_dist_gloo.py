"""The multi-process paths on CPU: world_size 2 over gloo (rendezvous on 127.0.0.1)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dvae_amd import data as DD
from dvae_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    D.init_from_env("gloo")
    assert D.world_info() == (rank, world)
    # purification sharding: each rank "purifies" its index range, then the shards are all-gathered
    full = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, 8, 8, 3), dtype=np.uint8))
    lo, hi = D.shard_range(n, rank, world)
    local_u8 = 255 - full[lo:hi]                       # any per-image function of the input
    local_noise = full[lo:hi].float().mean(dim=(1, 2, 3))
    got = D.all_gather_shards(local_u8, n)
    got_noise = D.all_gather_shards(local_noise, n)
    assert got.dtype == torch.uint8 and torch.equal(got, 255 - full)          # bit-exact reassembly
    assert torch.equal(got_noise, full.float().mean(dim=(1, 2, 3)))
    # training: each rank takes its slice of every global batch; summed "gradients" equal the 1-rank sum
    ds = DD.ArrayDataset(full.numpy(), np.arange(n) % 10)
    mine = DD.DeviceLoader(ds, 16, True, True, "cpu", seed=5, rank=rank, world=world)
    whole = DD.DeviceLoader(ds, 16 * world, True, True, "cpu", seed=5)
    w = torch.linspace(-1, 1, 8 * 8 * 3)
    for (x, _), (xg, _) in zip(mine, whole):
        g_local = (x.reshape(x.shape[0], -1) @ w).sum().reshape(1) * torch.ones(4)
        D.allreduce_sum_(g_local)
        g_ref = (xg.reshape(xg.shape[0], -1) @ w).sum()
        assert torch.allclose(g_local, g_ref * torch.ones(4), rtol=1e-5)
    # perturbations go to rank 0 only; BatchNorm statistics come from rank 0
    only0 = D.gather_shards(local_noise, n, dst=0)
    assert (only0 is None) == (rank != 0)
    if rank == 0:
        assert torch.equal(only0, full.float().mean(dim=(1, 2, 3)))
    stats = torch.full((6,), float(rank + 1))
    D.broadcast_(stats, 0)
    assert torch.equal(stats, torch.ones(6))
    # classifier stage: per-rank gradients are averaged through one flat bucket
    from dvae_amd.classifier import DataParallelSGD
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    xs = torch.randn(8, 6, generator=torch.Generator().manual_seed(1))
    ys = torch.arange(8) % 3
    per = 8 // world
    torch.nn.functional.cross_entropy(net(xs[rank * per:(rank + 1) * per]), ys[rank * per:(rank + 1) * per]).backward()
    DataParallelSGD().sync(net)
    got_g = [q.grad.clone() for q in net.parameters()]
    net.zero_grad()
    torch.nn.functional.cross_entropy(net(xs), ys).backward()
    for a, q in zip(got_g, net.parameters()):
        assert torch.allclose(a, q.grad, atol=1e-6)
    # one node: every rank delivers its shard through shared host memory, no collective carries the data
    assert D.single_node() and D.backend() == "gloo"
    sh = D.SharedHostBuffers({"u8": ((n, 8, 8, 3), torch.uint8), "noise": ((n,), torch.float64)})
    paths = list(sh.paths.values())
    assert all(os.path.exists(q) for q in paths)
    sh["u8"][lo:hi].copy_(local_u8)
    sh["noise"][lo:hi].copy_(local_noise)
    sh.barrier()
    assert torch.equal(sh["u8"], 255 - full)                                   # the other rank's rows are visible
    mine_u8 = sh["u8"].clone()
    kept = sh.detach("noise", owner=0)
    assert (kept is None) == (rank != 0) and not any(os.path.exists(q) for q in paths)
    if rank == 0:
        assert kept.dtype == torch.float64 and torch.equal(kept, full.float().mean(dim=(1, 2, 3)).double())
    assert torch.equal(mine_u8, 255 - full)
    sums = torch.tensor([1.0, 2.0, 3.0, 4.0]) * (rank + 1)
    D.allreduce_sum_(sums)
    assert torch.equal(sums, torch.tensor([1.0, 2.0, 3.0, 4.0]) * sum(range(1, world + 1)))
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def _run(n, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_world2_shards_divisible(tmp_path):
    _run(64, tmp_path)


def test_world2_shards_ragged(tmp_path):
    _run(101, tmp_path)
