"""Process-per-GPU data parallelism for the two shardable paths.

The reference's only parallelism is single-process ``torch.nn.DataParallel``
(/root/reference/main_vae_disentangle.py:426-428).  Here every GPU has its own process and the ranks
meet through ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests):

* D-VAE training shards each global batch by sample; gradients are summed over ranks with ONE
  all-reduce of the flat fp32 bucket per step (plus a 4-float all-reduce of the loss sums ahead of
  the rate-constraint gate, so that the gate sees the global batch as DataParallel's does).
* Purification shards the dataset by contiguous image-index ranges with no collective on the data
  path; the uint8 shards (and perturbations) are all-gathered once per pass.  Reassembly is
  concatenation in rank order, hence bit-identical to a single-GPU pass.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


_SOLO = [False]


def world_info():
    if not _SOLO[0] and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class solo:
    """Inside the block this process sees itself as a job of one rank (``world_info()`` answers (0, 1)) although a
    process group exists: for timing a one-rank pass inside a multi-rank job, while the other ranks wait at a barrier."""

    def __enter__(self):
        self.prev, _SOLO[0] = _SOLO[0], True

    def __exit__(self, *a):
        _SOLO[0] = self.prev


def init_from_env(backend: str | None = None):
    """Initialise from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).  No-op on one process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return world_info()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return world_info()


def backend() -> str | None:
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None


_SINGLE_NODE = {}


def single_node() -> bool:
    """True when every rank of the job runs on this host (they can then meet in shared host memory).  Decided once per
    process group by comparing host names (plus the boot id, so that two containers with one name do not pass)."""
    rank, world = world_info()
    if world == 1:
        return True
    key = id(dist.group.WORLD)
    if key not in _SINGLE_NODE:
        import socket
        try:
            boot = open("/proc/sys/kernel/random/boot_id").read().strip()
        except OSError:
            boot = ""
        names = [None] * world
        dist.all_gather_object(names, (socket.gethostname(), boot, os.path.isdir("/dev/shm")))
        _SINGLE_NODE[key] = all(x == names[0] for x in names) and names[0][2]
    return _SINGLE_NODE[key]


class SharedHostBuffers:
    """Named host tensors that every rank of a one-node job maps from the same /dev/shm files: ``bufs[name]`` is the whole
    tensor on every rank, a write by one rank is visible to all after ``barrier()``.  Rank 0 creates the files (sized, zero
    pages untouched), the others map them after a barrier; ``detach(name, owner)`` ends the sharing: the owner keeps its
    mapping as an ordinary tensor, every file is unlinked.  Used by the purification pass so that each rank delivers its
    own shard of the results without a collective (main_vae_disentangle.py:66-69 writes them from one process)."""

    _count = 0

    def __init__(self, specs: dict):
        rank, world = world_info()
        token = [None]
        if rank == 0:
            SharedHostBuffers._count += 1
            token[0] = f"dvae_{os.getpid()}_{SharedHostBuffers._count}"
        dist.broadcast_object_list(token, src=0)
        self.paths, self.t = {}, {}
        for name, (shape, dtype) in specs.items():
            self.paths[name] = f"/dev/shm/{token[0]}_{name}"
        if rank == 0:
            for name, (shape, dtype) in specs.items():
                self.t[name] = self._map(name, shape, dtype, create=True)
        dist.barrier()
        if rank != 0:
            for name, (shape, dtype) in specs.items():
                self.t[name] = self._map(name, shape, dtype, create=False)
        self._unlinked = False

    def _map(self, name, shape, dtype, create):
        n = 1
        for k in shape:
            n *= int(k)
        nbytes = max(1, n) * torch.empty((), dtype=dtype).element_size()
        if create:
            with open(self.paths[name], "wb") as f:
                f.truncate(nbytes)
        return torch.from_file(self.paths[name], shared=True, size=max(1, n), dtype=dtype)[:n].view(shape)

    def __getitem__(self, name):
        return self.t[name]

    def barrier(self):
        dist.barrier()

    def detach(self, name, owner=0):
        """End the sharing (collective).  Returns the tensor ``name`` on rank ``owner`` -- still backed by the mapping, which
        lives as long as the tensor -- and None elsewhere.  Every rank must have taken what it needs from the other
        buffers before the call."""
        rank, _ = world_info()
        if rank == 0 and not self._unlinked:         # every rank mapped the files in __init__: the names can go now
            for p in self.paths.values():
                try:
                    os.unlink(p)
                except OSError:
                    pass
        self._unlinked = True
        dist.barrier()                               # every rank has taken what it needs; the names are gone
        keep = self.t[name] if rank == owner else None
        self.t = {}
        return keep

    def __del__(self):
        if not getattr(self, "_unlinked", True) and world_info()[0] == 0:
            for p in self.paths.values():
                try:
                    os.unlink(p)
                except OSError:
                    pass


def shard_range(n: int, rank: int, world: int):
    """Contiguous index range [lo, hi) of rank: ceil(n / world) images per rank, the tail rank(s) short."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def rank_batch_slice(global_batch_indices: torch.Tensor, rank: int, world: int):
    """This rank's samples of one (already shuffled, drop_last'ed) global batch: a contiguous slice of
    len(batch) / world samples.  Every rank derives the same permutation from the shared seed."""
    b = global_batch_indices.numel()
    if b % world:
        raise ValueError(f"global batch {b} is not divisible by world size {world}")
    per = b // world
    return global_batch_indices[rank * per:(rank + 1) * per]


def epoch_permutation(n: int, epoch: int, seed: int, device="cpu"):
    """The shuffled sample order of one epoch, identical on every rank."""
    g = torch.Generator().manual_seed(seed * 1_000_003 + epoch)
    return torch.randperm(n, generator=g).to(device)


def _host_staged(t: torch.Tensor) -> bool:
    """gloo moves device tensors through its own staging copies; when the calling process still has kernels queued and
    another rank shares the card (the CPU-backend rehearsal of the multi-rank paths), that stalls for tens of seconds
    (profiles/r02_gloo_stall_probe_*.txt).  Device tensors therefore cross gloo through an explicit host copy, which
    also drains the stream.  RCCL ("nccl") works on the device tensors themselves, ordered with the current stream."""
    return t.is_cuda and dist.get_backend() == "gloo"


def allreduce_sum_(t: torch.Tensor):
    """In-place sum over ranks (identity on one process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if _host_staged(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_gather_shards(local: torch.Tensor, n: int) -> torch.Tensor:
    """Reassemble per-rank shards made with ``shard_range`` into the full [n, ...] tensor on every rank.
    Shards are padded to ceil(n / world) rows for the collective and the pad is cut off afterwards."""
    rank, world = world_info()
    if world == 1:
        assert local.shape[0] == n
        return local
    per = (n + world - 1) // world
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    if _host_staged(local):
        h = local.cpu()
        out = torch.empty((world * per,) + tuple(h.shape[1:]), dtype=h.dtype)
        dist.all_gather_into_tensor(out, h.contiguous())
        return out[:n].to(local.device)
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out[:n]


def gather_shards(local: torch.Tensor, n: int, dst: int = 0):
    """Like ``all_gather_shards`` but only rank ``dst`` receives the [n, ...] tensor (the others get None): for
    results only one rank consumes, such as the perturbations rank 0 saves (main_vae_disentangle.py:91-92).  The
    other ranks neither allocate nor receive the full tensor."""
    rank, world = world_info()
    if world == 1:
        assert local.shape[0] == n
        return local
    per = (n + world - 1) // world
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    local = local.contiguous()
    if _host_staged(local):
        local = local.cpu()                                 # rank dst then holds the result on the host
    if rank == dst:
        out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.gather(local, list(out.split(per, 0)), dst=dst)
        return out[:n]
    dist.gather(local, None, dst=dst)
    return None


def broadcast_(t: torch.Tensor, src: int = 0):
    """In-place broadcast from rank ``src`` (identity on one process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if _host_staged(t):
            h = t.cpu()
            dist.broadcast(h, src)
            t.copy_(h)
        else:
            dist.broadcast(t, src)
    return t
