"""``VAE_Trainer`` and ``Trainer`` with the reference's call signatures, on the HIP path.

Mirrors /root/reference/main_vae_disentangle.py: ``VAE_Trainer`` (:127-250) and ``Trainer`` (:25-125).
The reference steers both through module-level globals (``iters``, ``args.kd``, ``args.exp_name``,
``datasets_generator``); here they are constructor arguments / attributes with the same meaning:

    stage (== iters)   0 or 1      kld_target = 1.0 if stage == 0 else kd           (:181-184)
    kd                 float       --kd, default 2.5 (:283); the published scripts use 3.0
    exp_name           str         directory of perturbation{stage}.pt / model_s{stage}.pth  (:92, :456)

``VAE_Trainer.train_batch`` has two routes.  With a ``FusedAdam`` optimiser the whole step (forward,
losses, rate-constraint gate, backward, clip, Adam) runs as one fixed kernel schedule in the engine and
statistics stay on the device until a log step.  With any other optimiser the model's autograd node is
used and the loss block is the reference's torch code verbatim in meaning; that route exists for API
compatibility, not speed.
"""
from __future__ import annotations

import math
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib as L
from . import dist as D
from .optim import FusedAdam


class AverageMeter:
    """util.py:124-142."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = self.max = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
        self.max = max(self.max, val)


def log_display(epoch, global_step, time_elapse, **kwargs):
    """util.py:52-61."""
    display = "epoch=" + str(epoch) + "\tglobal_step=" + str(global_step)
    for key, value in kwargs.items():
        display += "\t" + str(key) + ("=" + value if isinstance(value, str) else "=%.4f" % float(value))
    display += "\ttime=%.2fit/s" % (1.0 / max(time_elapse, 1e-9))
    return display


def stage_constants(stage: int, kd: float):
    target = 1.0 if stage == 0 else kd
    return target, (0.01 if target > 0.99 else 0.5)


_METERS = ("loss", "loss_rec", "loss_rec_p", "loss_p", "loss_cls", "loss_cls_p", "loss_cls_p2", "loss_cls_purified",
           "kld_loss", "psnr", "psnr_p")


class VAE_Trainer:
    def __init__(self, data_loader, logger, config, global_step=0, target="train_dataset", stage=0, kd=2.5,
                 monitor_every=None, act_dtype=None):
        self.data_loader, self.logger, self.config = data_loader, logger, config
        self.act_dtype = act_dtype          # None: whatever the model was built with; "f32" | "bf16" sets model.act_dtype
        lf = getattr(config, "log_frequency", None)
        self.log_frequency = lf if lf is not None else 100
        self.global_step, self.target = global_step, target
        self.stage, self.kd = stage, kd
        self.monitor_every = self.log_frequency if monitor_every is None else monitor_every
        self._monitors = {k: 0.0 for k in ("loss_cls", "loss_cls_p", "loss_cls_p2", "loss_cls_purified")}
        self._monitor_fresh, self._monitor_n = False, 0
        self._reset_stats()

    def _reset_stats(self):
        for k in _METERS:
            setattr(self, k + "_meters", AverageMeter())
        self._pending = []          # per-step (device scalars, batch size) since the last host sync

    # -- epoch loop (main_vae_disentangle.py:162-178) ------------------------------------------------
    def train(self, epoch, model, model_c, optimizer):
        model.train()
        if model_c is not None:
            model_c.eval()
        for images, labels in self.data_loader[self.target]:
            images, labels = images.to("cuda", non_blocking=True), labels.to("cuda", non_blocking=True)
            start = time.time()
            payload = self.train_batch(images, labels, model, model_c, optimizer, epoch)
            if self.global_step % self.log_frequency == 0 and self.logger is not None:
                payload = self.materialize(payload)
                self.logger.info(log_display(epoch=epoch, global_step=self.global_step,
                                             time_elapse=time.time() - start, **payload))
            self.global_step += 1
        return self.global_step

    # -- one step (main_vae_disentangle.py:180-250) --------------------------------------------------
    def train_batch(self, images, labels, model, model_c, optimizer, epoch):
        if self.act_dtype is not None:
            model.act_dtype = self.act_dtype
        target, weight = stage_constants(self.stage, self.kd)
        grad_clip = getattr(self.config, "grad_clip", 5.0)
        if isinstance(optimizer, FusedAdam):
            return self._train_batch_fused(images, labels, model, model_c, optimizer, epoch, target, weight, grad_clip)
        model.zero_grad()
        optimizer.zero_grad()
        logits, rec, mu, log_var, perturbation = model(images, labels)
        loss_rec = F.mse_loss(rec, images)
        loss_rec_p = F.mse_loss((rec + perturbation).clamp(0, 1), images)
        loss_p = F.mse_loss(perturbation, torch.zeros_like(perturbation))
        kld_loss = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=(0, 1, 2))
        loss = loss_rec + loss_rec_p + weight * max(kld_loss, target)
        if epoch > 10:
            loss = loss + max(loss_p, 0.002)
        loss.backward()
        grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
        optimizer.step()
        # The model notices in-place updates through the tensors' version counters, but writes through ``param.data``
        # (hand-written SGD, EMA swaps, clamps) bump a different counter: after a foreign optimiser's step the packed
        # weights and the folded BatchNorm are always re-made (one launch).  Code that edits ``p.data`` elsewhere must call
        # ``model._engine.invalidate()`` itself.
        if getattr(model, "_engine", None) is not None:
            model._engine.invalidate()
        self._monitor(images, labels, rec.detach(), perturbation.detach(), model_c)
        vals = dict(loss=loss.item(), loss_rec=loss_rec.item(), loss_rec_p=loss_rec_p.item(), loss_p=loss_p.item(),
                    kld_loss=kld_loss.item())
        return self._payload(vals, labels.shape[0], optimizer, float(grad_norm))

    def _train_batch_fused(self, images, labels, model, model_c, optimizer, epoch, target, weight, grad_clip):
        rank, world = D.world_info()
        eng = model._engine_for(images)
        b = images.shape[0]
        optimizer.max_norm = grad_clip
        g = optimizer.param_groups[0]
        key = (id(eng), g["lr"], g["betas"], g["eps"], grad_clip, 1.0 / world)
        if key != optimizer._pushed:
            eng.set_hyper(lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], max_norm=grad_clip,
                          grad_scale=1.0 / world)
            optimizer._pushed = key
        x0 = eng.input_from_nchw(images)
        eps = eng.eps_philox(b, first=(self.global_step * world + rank) * b, stream_id=11 + self.stage)
        eng.train_step(x0, labels, eps, b, epoch, target, weight, D.allreduce_sum_ if world > 1 else None, world)
        model._nbt += 1
        model._accumulate = False
        if model_c is not None and self.monitor_every and self.global_step % self.monitor_every == 0:
            S = eng.S
            rec = eng.to_nchw(eng.buf("rec", (eng.B, S, S, 4), torch.float32), b, 3, S)
            pert = eng.to_nchw(eng.act("decoder_p.d6", S, 3, f32=True), b, 3, S)
            self._monitor(images, labels, rec, pert, model_c, force=True)
        # statistics stay on the device; materialize() syncs once, at log time
        s = eng.scal.clone()
        self._pending.append((s, b))
        if len(self._pending) >= 1024:          # nobody is logging: fold now and then so the list stays short
            self._fold_pending()
        return {"_scal": s, "_n": b, "_lr": g["lr"]}

    def _fold_pending(self):
        """Feed the per-step scalars kept on the device since the last sync into the meters, step by step and
        weighted by each step's batch size, exactly as the reference's per-step ``.item()`` updates do
        (main_vae_disentangle.py:215-225)."""
        if not self._pending:
            return
        rows = torch.stack([s for s, _ in self._pending]).cpu().numpy()
        for r, (_, n) in zip(rows, self._pending):
            vals = {"loss": float(r[L.SCAL["loss"]]), "loss_rec": float(r[L.SCAL["loss_rec"]]),
                    "loss_rec_p": float(r[L.SCAL["loss_rec_p"]]), "loss_p": float(r[L.SCAL["loss_p"]]),
                    "kld_loss": float(r[L.SCAL["kld"]])}
            vals["psnr"] = -10 * math.log10(max(vals["loss_rec"], 1e-30))
            vals["psnr_p"] = -10 * math.log10(max(vals["loss_rec_p"], 1e-30))
            for k, v in vals.items():
                getattr(self, k + "_meters").update(v, n)
        self._pending = []

    @torch.no_grad()
    def _monitor(self, images, labels, rec, pert, model_c, force=False):
        """The four model_c monitor losses (main_vae_disentangle.py:194-198): logged, never in ``loss``.
        Stock torch ops on the frozen classifier, evaluated every ``monitor_every`` steps."""
        if model_c is None or not self.monitor_every:
            return
        if not force and self.global_step % self.monitor_every:
            return
        x4 = torch.cat([rec, (rec + pert).clamp(0, 1), (images - pert).clamp(0, 1)], 0)
        lg = model_c(x4)
        b = images.shape[0]
        self._monitors["loss_cls"] = F.cross_entropy(lg[:b], labels).item()
        self._monitors["loss_cls_p"] = F.cross_entropy(lg[b:2 * b], labels).item()
        self._monitors["loss_cls_p2"] = F.cross_entropy(lg[2 * b:], labels).item()
        uni = torch.ones_like(lg[2 * b:]) / lg.shape[1]
        self._monitors["loss_cls_purified"] = F.kl_div(F.log_softmax(lg[2 * b:], dim=1), uni, reduction="batchmean").item()
        self._monitor_fresh, self._monitor_n = True, b

    def materialize(self, payload):
        """Turn a fused-step payload into the reference's 24-key dict (one device->host copy)."""
        if "_scal" not in payload:
            return payload
        s = payload["_scal"].cpu().numpy()
        vals = dict(loss=float(s[L.SCAL["loss"]]), loss_rec=float(s[L.SCAL["loss_rec"]]), loss_rec_p=float(s[L.SCAL["loss_rec_p"]]),
                    loss_p=float(s[L.SCAL["loss_p"]]), kld_loss=float(s[L.SCAL["kld"]]))
        self._fold_pending()
        return self._payload(vals, 0, None, float(s[L.SCAL["grad_norm"]]), lr=payload["_lr"])

    def _payload(self, vals, n, optimizer, grad_norm, lr=None):
        psnr = -10 * math.log10(max(vals["loss_rec"], 1e-30))
        psnr_p = -10 * math.log10(max(vals["loss_rec_p"], 1e-30))
        vals = dict(vals, psnr=psnr, psnr_p=psnr_p, **self._monitors)
        if n:
            for k in _METERS:
                getattr(self, k + "_meters").update(vals[k], n)
        else:
            # fused route: the loss / psnr meters were fed step by step from the device scalars; the monitor
            # losses are host floats already
            for k in ("loss_cls", "loss_cls_p", "loss_cls_p2", "loss_cls_purified"):
                if self._monitor_fresh:
                    getattr(self, k + "_meters").update(vals[k], self._monitor_n)
            self._monitor_fresh = False
        out = {}
        for k in ("psnr", "psnr_p", "loss", "loss_rec", "loss_rec_p", "loss_p", "loss_cls", "loss_cls_p", "loss_cls_p2",
                  "loss_cls_purified", "kld_loss"):
            out[k] = vals[k]
            out[k + "_avg"] = getattr(self, k + "_meters").avg
        out["lr"] = lr if lr is not None else optimizer.param_groups[0]["lr"]
        out["|gn|"] = grad_norm
        return out


_STAGE = {}          # kind -> (key, pinned buffers): ONE entry per kind, replaced (and its pinned memory freed) when the key changes
_COPY_STREAM = {}    # device index -> the side stream the drains copy on


def _staging(kind, key, make):
    """Pinned staging buffers of one kind ("chunk" | "drain").  A pass keeps re-using them; a different (batch, image
    size, dtype) drops the old ones first, so the page-locked footprint is that of the current pass only:
    3 x batch x (3 S S x 8 + 3 S S) bytes for the drain (85 MB at 1024 images of 32 x 32, 340 MB at 64 x 64)."""
    ent = _STAGE.get(kind)
    if ent is None or ent[0] != key:
        _STAGE.pop(kind, None)
        ent = _STAGE[kind] = (key, make())
    return ent[1]


def release_staging():
    """Free every pinned staging buffer (and side stream, and the copy threads) kept by purify_dataset's host drains."""
    _STAGE.clear()
    _COPY_STREAM.clear()
    while _POOL:
        _POOL.pop().shutdown(wait=True)


def _copy_stream(kind="down"):
    """Side streams kept per device: "down" for the drains' device-to-host copies, "up" for uploads."""
    key = (torch.cuda.current_device(), kind)
    st = _COPY_STREAM.get(key)
    if st is None:
        st = _COPY_STREAM[key] = torch.cuda.Stream()
    return st


def _to_host_chunked(src, dst, chunk_rows=4096):
    """Device tensor -> host tensor of the same shape (``dst``: pageable, any dtype), through two pinned staging buffers
    that are kept for the life of the process: the device converts a chunk to ``dst``'s dtype and copies it out
    asynchronously while the host moves the previous chunk from the staging buffer into ``dst``.  A pageable
    ``tensor.cpu()`` of the 614 MB of perturbations of a 50 000-image pass takes 145 ms and its float64 conversion on
    the host 25-40 more; this takes 35-45."""
    n = src.shape[0]
    if n == 0:
        return dst
    row = tuple(src.shape[1:])
    stage = _staging("chunk:" + str(dst.dtype), (row, dst.dtype, chunk_rows),
                     lambda: [torch.empty((chunk_rows,) + row, dtype=dst.dtype, pin_memory=True) for _ in range(2)])
    pending = []
    for k, lo in enumerate(range(0, n, chunk_rows)):
        hi = min(n, lo + chunk_rows)
        if len(pending) == 2:                       # the buffer about to be reused must have been emptied
            ev, plo, phi, ps = pending.pop(0)
            ev.synchronize()
            dst[plo:phi].copy_(ps[:phi - plo])
        s = stage[k % 2]
        s[:hi - lo].copy_(src[lo:hi], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pending.append((ev, lo, hi, s))
    for ev, plo, phi, ps in pending:
        ev.synchronize()
        dst[plo:phi].copy_(ps[:phi - plo])
    return dst


_POOL = []


def _pool():
    """A few worker threads for host-side copies (tensor copies release the interpreter lock), kept for the process."""
    if not _POOL:
        from concurrent.futures import ThreadPoolExecutor
        _POOL.append(ThreadPoolExecutor(max_workers=max(2, min(8, (os.cpu_count() or 2) // 2))))
    return _POOL[0]


def _copy_rows_parallel(dst, src, ranges, chunk_rows=4096, src_offset=0):
    """dst[a:b] = src[a - src_offset : b - src_offset] for every (a, b) of ``ranges`` (host tensors of one row shape), in
    chunks spread over the worker threads.  One thread moves about 10 GB/s -- and a fresh destination takes a page fault
    per 4 KB on top -- which makes the float64 perturbations (25 MB per 1024 images) the slowest stage of a pass."""
    jobs = [(a0, min(b, a0 + chunk_rows)) for a, b in ranges for a0 in range(a, b, chunk_rows)]
    if not jobs:
        return
    if len(jobs) == 1:
        a, b = jobs[0]
        dst[a:b].copy_(src[a - src_offset:b - src_offset])
        return
    list(_pool().map(lambda ab: dst[ab[0]:ab[1]].copy_(src[ab[0] - src_offset:ab[1] - src_offset]), jobs))


class _HostDrain:
    """Copies finished batches of a purification pass to the host while later batches are still being computed (one
    rank): for rows [lo, hi) the uint8 images go straight into ``dst_u8`` and the perturbations, converted to float64
    on the device, into ``dst_f64``, each through pinned staging buffers on a copy stream of its own."""

    DEPTH = 3

    def __init__(self, dst_u8, dst_f64, rows, row_u8, row_f64, also_u8=None):
        self.dst = (dst_u8, dst_f64)
        self.also_u8 = also_u8      # a second home of the uint8 rows (a multi-rank pass: this rank's own dataset.data)
        self.stage = _staging("drain", (rows, row_u8, row_f64), lambda: [
            (torch.empty((rows,) + row_u8, dtype=torch.uint8, pin_memory=True),
             torch.empty((rows,) + row_f64, dtype=torch.float64, pin_memory=True)) for _ in range(self.DEPTH)])
        self.stream = _copy_stream()
        self.busy = [[] for _ in range(self.DEPTH)]    # per staging buffer: the copies still reading it
        self.k = 0

    def _copy_out(self, ev, lo, hi, st, a, b):
        """worker: rows [a, b) of a pushed batch, staging -> destination (the float64 rows; the first piece also the uint8)"""
        ev.synchronize()
        self.dst[1][a:b].copy_(st[1][a - lo:b - lo])
        if a == lo:
            self.dst[0][lo:hi].copy_(st[0][:hi - lo])
            if self.also_u8 is not None:
                self.also_u8[lo:hi].copy_(st[0][:hi - lo])

    def push(self, lo, hi, src_u8, src_noise):
        """``src_*``: device tensors of the rows [lo, hi), complete once the current stream reaches this point.  The copies
        out of the staging buffers run on the worker threads (tensor copies and event waits release the interpreter lock):
        the calling thread only waits when it comes round to a staging buffer whose copies are still running, so that it
        stays ahead of the GPU with the next batches' launches (one thread moves a 2048-image batch's 50 MB of float64 in
        3-4 ms, about the time the GPU needs for the batch)."""
        slot = self.k % self.DEPTH
        self.k += 1
        for f in self.busy[slot]:
            f.result()
        st = self.stage[slot]
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            st[0][:hi - lo].copy_(src_u8, non_blocking=True)
            st[1][:hi - lo].copy_(src_noise, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        step = max(32, (hi - lo + 5) // 6)
        self.busy[slot] = [_pool().submit(self._copy_out, ev, lo, hi, st, a, min(hi, a + step)) for a in range(lo, hi, step)]

    def finish(self):
        for fs in self.busy:
            for f in fs:
                f.result()
        self.busy = [[] for _ in range(self.DEPTH)]


@torch.no_grad()
def purify_dataset(model_s, dataset, stage: int, batch: int = 2048, seed: int = 0, device="cuda", act_dtype=None,
                   timings: dict | None = None, overlap_tail: bool = True):
    """The epoch-0 branch of ``Trainer.train`` (main_vae_disentangle.py:59-69) for a whole dataset:
    overwrites ``dataset.data`` in place with the purified uint8 images and returns the recorded
    perturbations as a float64 tensor [N,3,H,W] (on rank 0; the other ranks, which do not save it, get None).
    Images are sharded over ranks by contiguous index range.  On one node every rank drains its own shard (uint8 images
    and float64 perturbations) into shared host memory while it computes its later batches; afterwards every rank holds
    the full purified set (it trains the next stage on it) and rank 0 the perturbations.  Over several nodes the uint8
    shards are all-gathered and the perturbations gathered to rank 0.  The
    reparameterisation noise is keyed by global image index, so the result does not depend on the world size or the
    batch size.  BatchNorm running statistics are taken from rank 0 (``DataParallel`` keeps replica 0's,
    main_vae_disentangle.py:426-428), so every shard is purified by the same function.
    ``timings`` (optional dict) receives the wall time of each phase in seconds."""
    rank, world = D.world_info()
    t0 = time.perf_counter()
    n = len(dataset)
    lo, hi = D.shard_range(n, rank, world)
    dev = torch.device(device)
    if act_dtype is not None:
        model_s.act_dtype = act_dtype
    # The shard's uint8 images go to the device batch by batch on the copy stream, a batch ahead of the kernels that read
    # them (a pageable copy of the whole shard up front is 7 ms of a 0.13 s pass in which the GPU does nothing)
    host_u8 = torch.from_numpy(dataset.data[lo:hi])
    labels = torch.from_numpy(np.asarray(dataset.targets[lo:hi], dtype=np.int64)).to(dev)
    S = dataset.data.shape[1]
    was_training = model_s.training
    model_s.eval()
    probe = torch.zeros(min(batch, max(1, hi - lo)), 3, S, S, device=dev)
    eng = model_s._engine_for(probe)
    if world > 1:
        D.broadcast_(eng.R, 0)
        eng.fold_valid = False
    eng.seed = seed
    cuda = dev.type == "cuda"
    data = torch.empty(host_u8.shape, dtype=torch.uint8, device=dev)
    up_stream = _copy_stream() if cuda else None
    up_events = {}

    def upload(first):
        """rows [first, first + eng.B) of the shard -> device (asynchronous to the kernels in flight)"""
        if first >= hi - lo or first in up_events:
            return
        last = min(hi - lo, first + eng.B)
        if cuda:
            with torch.cuda.stream(up_stream):
                data[first:last].copy_(host_u8[first:last])
                ev = torch.cuda.Event()
                ev.record()
            up_events[first] = ev
        else:
            data[first:last].copy_(host_u8[first:last])
            up_events[first] = None

    upload(0)
    out = torch.empty_like(data)
    noise = torch.empty(hi - lo, 3, S, S, device=dev)
    if timings is not None:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    # Finished batches drain to the host under the batches that follow (uint8 images; float64 perturbations converted on
    # the device), every rank its own shard.  One rank: straight into dataset.data (main_vae_disentangle.py:67) and the
    # result tensor.  Several ranks on one node: into two buffers in shared host memory, so that no rank moves another
    # rank's perturbations and no collective carries them; after a barrier every rank copies the other shards' uint8 rows
    # into its own dataset.data and rank 0 owns the float64 buffer.  Ranks spread over nodes: gather, then one chunked copy.
    drain = shared = None
    if world == 1 and cuda and overlap_tail:
        res = torch.empty(n, 3, S, S, dtype=torch.float64)
        drain = _HostDrain(torch.from_numpy(dataset.data), res, eng.B, tuple(data.shape[1:]), (3, S, S))
    elif world > 1 and overlap_tail and D.single_node():
        shared = D.SharedHostBuffers({"u8": ((n,) + tuple(dataset.data.shape[1:]), torch.uint8),
                                      "noise": ((n, 3, S, S), torch.float64)})
        if cuda:
            drain = _HostDrain(shared["u8"][lo:hi], shared["noise"][lo:hi], eng.B, tuple(data.shape[1:]), (3, S, S),
                               also_u8=torch.from_numpy(dataset.data)[lo:hi])
    for first in range(0, hi - lo, eng.B):
        b = min(eng.B, hi - lo - first)
        if up_events.get(first) is not None:
            torch.cuda.current_stream().wait_event(up_events[first])
        eng.purify_batch(data, out, noise, labels, first, b, stage, global_first=lo + first)
        upload(first + eng.B)
        if drain is not None:
            drain.push(first, first + b, out[first:first + b], noise[first:first + b])
    if timings is not None:
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    if shared is not None:
        # RCCL: the uint8 shards are all-gathered on the device as well (19 MB per rank at 8 GPUs, asynchronous to the host)
        # and handed to the device-resident loaders of the next stage, which then skip the 153 MB re-upload
        if cuda and D.backend() == "nccl":
            dataset._device_copy = (getattr(dataset, "version", 0) + 1, D.all_gather_shards(out, n))
        if drain is not None:
            drain.finish()
        else:
            shared["u8"][lo:hi].copy_(out)
            shared["noise"][lo:hi].copy_(noise)
            torch.from_numpy(dataset.data)[lo:hi].copy_(out)
        shared.barrier()                                     # every shard is in place
        if timings is not None:
            t3 = time.perf_counter()
        # the other ranks' rows (this rank's own went to dataset.data with the drain), a few threads side by side: one
        # thread moves the 135 MB of a 50 000-image set at 8 ranks in 12-15 ms, which is most of such a rank's whole pass
        _copy_rows_parallel(torch.from_numpy(dataset.data), shared["u8"], [(0, lo), (hi, n)])
        res = shared.detach("noise", owner=0)                # rank 0 keeps the mapping (the others get None); files unlinked
    elif drain is not None:
        if timings is not None:
            t3 = t2
        drain.finish()
    else:
        full_u8 = D.all_gather_shards(out, n)
        full_noise = D.gather_shards(noise, n, dst=0)
        if timings is not None:
            torch.cuda.synchronize()
            t3 = time.perf_counter()
        if full_u8.is_cuda:
            _to_host_chunked(full_u8, torch.from_numpy(dataset.data))
        else:
            dataset.data[...] = full_u8.numpy()
        res = None
        if full_noise is not None:
            res = torch.empty(full_noise.shape, dtype=torch.float64)
            if full_noise.is_cuda:
                _to_host_chunked(full_noise, res)
            else:
                res.copy_(full_noise)
    dataset.version = getattr(dataset, "version", 0) + 1      # loaders holding a device copy refresh it
    model_s.train(was_training)
    if timings is not None:
        t4 = time.perf_counter()
        timings.update(upload_s=t1 - t0, kernels_s=t2 - t1, gather_s=t3 - t2, host_tail_s=t4 - t3, total_s=t4 - t0)
    return res


class Trainer:
    """main_vae_disentangle.py:25-125.  ``train(epoch=0, ...)`` is the purification pass; later epochs
    train the classifier on the purified set with stock torch ops (not on the accelerated path)."""

    def __init__(self, criterion, data_loader, logger, config, global_step=0, target="train_dataset",
                 datasets_generator=None, exp_name=".", iters=0, purify_batch=2048, seed=0, act_dtype=None):
        self.criterion, self.data_loader, self.logger, self.config = criterion, data_loader, logger, config
        lf = getattr(config, "log_frequency", None)
        self.log_frequency = lf if lf is not None else 100
        self.global_step, self.target = global_step, target
        self.datasets_generator, self.exp_name, self.iters = datasets_generator, exp_name, iters
        self.purify_batch, self.seed, self.act_dtype = purify_batch, seed, act_dtype
        self.train_transform = "crop_flip"   # RandomCrop(32, 4) + HorizontalFlip + ToTensor (:85-87)
        self._reset_stats()

    def _reset_stats(self):
        self.loss_meters, self.acc_meters, self.acc5_meters = AverageMeter(), AverageMeter(), AverageMeter()

    def train(self, epoch, model, model_s, model_c, criterion, optimizer, random_noise=None):
        if model is not None:
            model.train()
        if model_s is not None:
            model_s.eval()
        if epoch == 0:
            if model_s is not None:
                ds = self.datasets_generator.datasets["train_dataset"]
                noise = purify_dataset(model_s, ds, self.iters, self.purify_batch, self.seed, act_dtype=self.act_dtype)
                ds.transform = self.train_transform
                self.data_loader = self.datasets_generator.getDataLoader()
                rank, _ = D.world_info()
                if rank == 0:
                    os.makedirs(self.exp_name, exist_ok=True)
                    torch.save(noise, os.path.join(self.exp_name, "perturbation{}.pt".format(self.iters)))
            return self.global_step
        if self.iters == 0:
            return self.global_step              # stage 0 only purifies (:71-72)
        for images, labels in self.data_loader[self.target]:
            images, labels = images.to("cuda", non_blocking=True), labels.to("cuda", non_blocking=True)
            if getattr(self.datasets_generator.datasets["train_dataset"], "transform", None) == "crop_flip":
                images = random_crop_flip(images, 4)
            start = time.time()
            payload = self.train_batch(images, labels, model, optimizer)
            if self.global_step % self.log_frequency == 0 and self.logger is not None:
                self.logger.info(log_display(epoch=epoch, global_step=self.global_step, time_elapse=time.time() - start, **payload))
            self.global_step += 1
        return self.global_step

    def train_batch(self, images, labels, model, optimizer):
        """:97-125, CrossEntropyLoss criterion (the D-VAE configs' only one).  With more than one process the gradients
        are averaged over ranks through one flat bucket before the clip (the mean of per-rank batch means = the
        global-batch mean for equal per-rank batches)."""
        from .classifier import DataParallelSGD, accuracy
        model.zero_grad()
        optimizer.zero_grad()
        logits = model(images)
        loss = self.criterion(logits, labels)
        loss.backward()
        if getattr(self, "_dp", None) is None:
            self._dp = DataParallelSGD()
        self._dp.sync(model)
        grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), getattr(self.config, "grad_clip", 5.0))
        optimizer.step()
        if logits.shape[1] >= 5:
            acc, acc5 = (a.item() for a in accuracy(logits, labels, topk=(1, 5)))
        else:
            acc, acc5 = accuracy(logits, labels, topk=(1,))[0].item(), 1
        self.loss_meters.update(loss.item(), labels.shape[0])
        self.acc_meters.update(acc, labels.shape[0])
        self.acc5_meters.update(acc5, labels.shape[0])
        return {"acc": acc, "acc_avg": self.acc_meters.avg, "loss": loss.item(), "loss_avg": self.loss_meters.avg,
                "lr": optimizer.param_groups[0]["lr"], "|gn|": float(grad_norm)}


def random_crop_flip(x, pad):
    """RandomCrop(size, padding=pad) + RandomHorizontalFlip on a device batch (per-sample draws)."""
    b, _, h, w = x.shape
    xp = F.pad(x, (pad, pad, pad, pad))
    oy = torch.randint(0, 2 * pad + 1, (b,), device=x.device)
    ox = torch.randint(0, 2 * pad + 1, (b,), device=x.device)
    ar = torch.arange(h, device=x.device)
    yy = (oy[:, None] + ar[None, :])[:, :, None].expand(b, h, w)
    flip = torch.rand(b, device=x.device) < 0.5
    xs = ox[:, None] + ar[None, :]
    xs = torch.where(flip[:, None], xs.flip(1), xs)[:, None, :].expand(b, h, w)
    bi = torch.arange(b, device=x.device)[:, None, None].expand(b, h, w)
    return xp[bi, :, yy, xs].permute(0, 3, 1, 2).contiguous()
