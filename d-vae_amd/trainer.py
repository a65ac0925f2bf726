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
                 monitor_every=None):
        self.data_loader, self.logger, self.config = data_loader, logger, config
        lf = getattr(config, "log_frequency", None)
        self.log_frequency = lf if lf is not None else 100
        self.global_step, self.target = global_step, target
        self.stage, self.kd = stage, kd
        self.monitor_every = self.log_frequency if monitor_every is None else monitor_every
        self._monitors = {k: 0.0 for k in ("loss_cls", "loss_cls_p", "loss_cls_p2", "loss_cls_purified")}
        self._reset_stats()

    def _reset_stats(self):
        for k in _METERS:
            setattr(self, k + "_meters", AverageMeter())
        self._dev_sum = None
        self._dev_n = 0

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
        key = (g["lr"], g["betas"], g["eps"], grad_clip, 1.0 / world)
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
        self._dev_sum = s if self._dev_sum is None else self._dev_sum + s
        self._dev_n += 1
        return {"_scal": s, "_n": b, "_lr": g["lr"]}

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

    def materialize(self, payload):
        """Turn a fused-step payload into the reference's 24-key dict (one device->host copy)."""
        if "_scal" not in payload:
            return payload
        s = payload["_scal"].cpu().numpy()
        vals = dict(loss=float(s[L.SCAL["loss"]]), loss_rec=float(s[L.SCAL["loss_rec"]]), loss_rec_p=float(s[L.SCAL["loss_rec_p"]]),
                    loss_p=float(s[L.SCAL["loss_p"]]), kld_loss=float(s[L.SCAL["kld"]]))
        if self._dev_sum is not None:
            # fold the device-side sums of the steps since the last sync into the meters
            tot = self._dev_sum.cpu().numpy() / max(1, self._dev_n)
            for k, i in (("loss", "loss"), ("loss_rec", "loss_rec"), ("loss_rec_p", "loss_rec_p"), ("loss_p", "loss_p"), ("kld_loss", "kld")):
                getattr(self, k + "_meters").update(float(tot[L.SCAL[i]]), payload["_n"] * self._dev_n)
            self._dev_sum, self._dev_n = None, 0
        return self._payload(vals, 0, None, float(s[L.SCAL["grad_norm"]]), lr=payload["_lr"])

    def _payload(self, vals, n, optimizer, grad_norm, lr=None):
        psnr = -10 * math.log10(max(vals["loss_rec"], 1e-30))
        psnr_p = -10 * math.log10(max(vals["loss_rec_p"], 1e-30))
        vals = dict(vals, psnr=psnr, psnr_p=psnr_p, **self._monitors)
        if n:
            for k in _METERS:
                getattr(self, k + "_meters").update(vals[k], n)
        out = {}
        for k in ("psnr", "psnr_p", "loss", "loss_rec", "loss_rec_p", "loss_p", "loss_cls", "loss_cls_p", "loss_cls_p2",
                  "loss_cls_purified", "kld_loss"):
            out[k] = vals[k]
            out[k + "_avg"] = getattr(self, k + "_meters").avg
        out["lr"] = lr if lr is not None else optimizer.param_groups[0]["lr"]
        out["|gn|"] = grad_norm
        return out


@torch.no_grad()
def purify_dataset(model_s, dataset, stage: int, batch: int = 1024, seed: int = 0, device="cuda"):
    """The epoch-0 branch of ``Trainer.train`` (main_vae_disentangle.py:59-69) for a whole dataset:
    overwrites ``dataset.data`` in place with the purified uint8 images and returns the recorded
    perturbations as a float64 tensor [N,3,H,W].  Images are sharded over ranks by contiguous index
    range and all-gathered at the end; the reparameterisation noise is keyed by global image index, so
    the result does not depend on the world size or the batch size."""
    rank, world = D.world_info()
    n = len(dataset)
    lo, hi = D.shard_range(n, rank, world)
    dev = torch.device(device)
    data = torch.from_numpy(dataset.data[lo:hi]).to(dev)
    labels = torch.from_numpy(np.asarray(dataset.targets[lo:hi], dtype=np.int64)).to(dev)
    S = dataset.data.shape[1]
    was_training = model_s.training
    model_s.eval()
    probe = torch.zeros(min(batch, max(1, hi - lo)), 3, S, S, device=dev)
    eng = model_s._engine_for(probe)
    eng.seed = seed
    out = torch.empty_like(data)
    noise = torch.empty(hi - lo, 3, S, S, device=dev)
    for first in range(0, hi - lo, eng.B):
        b = min(eng.B, hi - lo - first)
        eng.purify_batch(data, out, noise, labels, first, b, stage, global_first=lo + first)
    full_u8 = D.all_gather_shards(out, n)
    full_noise = D.all_gather_shards(noise, n)
    dataset.data[...] = full_u8.cpu().numpy()
    model_s.train(was_training)
    return full_noise.cpu().to(torch.float64)


class Trainer:
    """main_vae_disentangle.py:25-125.  ``train(epoch=0, ...)`` is the purification pass; later epochs
    train the classifier on the purified set with stock torch ops (not on the accelerated path)."""

    def __init__(self, criterion, data_loader, logger, config, global_step=0, target="train_dataset",
                 datasets_generator=None, exp_name=".", iters=0, purify_batch=1024, seed=0):
        self.criterion, self.data_loader, self.logger, self.config = criterion, data_loader, logger, config
        lf = getattr(config, "log_frequency", None)
        self.log_frequency = lf if lf is not None else 100
        self.global_step, self.target = global_step, target
        self.datasets_generator, self.exp_name, self.iters = datasets_generator, exp_name, iters
        self.purify_batch, self.seed = purify_batch, seed
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
                noise = purify_dataset(model_s, ds, self.iters, self.purify_batch, self.seed)
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
        """:97-125, CrossEntropyLoss criterion (the D-VAE configs' only one)."""
        model.zero_grad()
        optimizer.zero_grad()
        logits = model(images)
        loss = self.criterion(logits, labels)
        loss.backward()
        grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), getattr(self.config, "grad_clip", 5.0))
        optimizer.step()
        acc = (logits.argmax(1) == labels).float().mean().item()
        self.loss_meters.update(loss.item(), labels.shape[0])
        self.acc_meters.update(acc, labels.shape[0])
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
