"""The classifier side of ``main_vae_disentangle.py``: the CIFAR ResNet-18 that is retrained on the purified set
(and serves, frozen, as the monitor ``model_c``), its evaluator, the per-epoch train/eval/checkpoint loop and the
checkpoint dictionary.

Reference lines mirrored (same names, arguments and file layouts):

    ResNet18 / ResNet          /root/reference/configs/models/resnet.py:62-122  (state_dict keys identical)
    accuracy                   /root/reference/util.py:64-82
    save_model / load_model    /root/reference/util.py:85-111
    Evaluator                  /root/reference/evaluator.py:15-71
    train                      /root/reference/main_vae_disentangle.py:323-371
    optimiser / scheduler      /root/reference/configs/cifar10/resnet18.yaml:14-23  (SGD 0.1 / 5e-4 / 0.9, cosine)

On the GPU the convolutional trunk (conv1 .. layer4: every conv, BatchNorm, residual add and ReLU, forward and backward)
runs on libdvae_hip through ``trunk.ResNetTrunkEngine`` -- the same kernels as the D-VAE's stem plus the layer3 / layer4
shapes -- in fp32 or with bf16 activations (``act_dtype``; parameters, gradients, BatchNorm statistics and the optimiser
stay fp32).  Pooling, the linear layer, the loss and SGD are stock torch ops.  There is no stock-torch route for device
tensors: without the built library a forward on the GPU raises.  CPU tensors run the plain torch modules (the reference's
own arithmetic; it is what the tests hold the device path to).  The data side is device-resident too: the purified set
stays in HBM as uint8, batches are gathered and augmented on the device (``trainer.random_crop_flip``), the confusion
matrix is a device ``bincount``, and with more than one process gradients are averaged through one flat bucket per step
(``DataParallelSGD``).
"""
from __future__ import annotations

import os
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dist as D
from .model import _ResBlock
from .trainer import AverageMeter, log_display


class _TrunkFunction(torch.autograd.Function):
    """conv1 .. layer4 as one autograd node: forward runs the engine, backward receives the gradient of layer4's output
    and leaves the trunk's parameter gradients in the engine's bucket (``param.grad`` are views of it)."""

    @staticmethod
    def forward(ctx, model, x, anchor):
        eng = model._engine_for(x)
        out = eng.forward_trunk(x.detach(), model.training)
        if model.training:
            torch._foreach_add_(model._nbt_list, 1)
            eng.fold_valid = False                        # the kernels advanced the running statistics in place
        ctx.model, ctx.eng = model, eng
        return out

    @staticmethod
    def backward(ctx, d_out):
        eng, model = ctx.eng, ctx.model
        params = eng.param
        mine = [n for n in eng.names if params[n].grad is not None and params[n].grad.data_ptr() == eng.grad[n].data_ptr()]
        saved = {n: eng.grad[n].clone() for n in mine}    # gradients kept from an earlier backward: accumulate
        eng.backward_trunk(d_out)
        for n in eng.names:
            p = params[n]
            if n in saved:
                eng.grad[n].add_(saved[n])
            elif p.grad is None:
                p.grad = eng.grad[n]
            else:
                p.grad.add_(eng.grad[n])
        return None, None, torch.zeros_like(model._anchor)


class ResNet(nn.Module):
    """CIFAR ResNet with BasicBlocks: 3x3 stem conv, four stages, global average pool, linear head.
    ``act_dtype`` ("f32" | "bf16", not in the reference, which is fp32): arithmetic of the device trunk; the environment
    variable DVAE_ACT_DTYPE overrides the default."""

    def __init__(self, num_blocks=(2, 2, 2, 2), num_classes=10, act_dtype=None):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (c, s) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2))):
            blocks = []
            for j in range(num_blocks[i]):
                blocks.append(_ResBlock(cin, c, s if j == 0 else 1))
                cin = c
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.linear = nn.Linear(512, num_classes)
        self.pooling = nn.AdaptiveAvgPool2d((1, 1))
        self.num_blocks = tuple(num_blocks)
        self.act_dtype = act_dtype if act_dtype is not None else os.environ.get("DVAE_ACT_DTYPE", "f32")
        self._engine = None
        self._versions = None
        object.__setattr__(self, "_anchor", torch.zeros((), requires_grad=True))

    def cam_register(self, layer="layer4"):
        """Kept for call compatibility (main_vae_disentangle.py:439 registers Grad-CAM hooks on model_c that nothing on
        the D-VAE path reads): records the layer name only."""
        self.cam_grad_layer = layer

    # ---- device trunk ----------------------------------------------------------------------------
    def _trunk_tensors(self):
        named = dict(self.named_parameters())
        named.update(dict(self.named_buffers()))
        return {n: t for n, t in named.items() if not n.startswith("linear.")}

    def _engine_for(self, x):
        from . import _lib as L
        from .model import _dtype_code
        from .trunk import ResNetTrunkEngine
        b, S = x.shape[0], x.shape[-1]
        dtype = _dtype_code(self.act_dtype)
        eng = self._engine
        ok = (eng is not None and eng.B >= b and eng.S == S and eng.dev == x.device and eng.dtype == dtype and
              eng.param["conv1.weight"] is self.conv1.weight)
        if not ok:
            t = self._trunk_tensors()
            for n, v in t.items():
                if v.device != x.device or (v.is_floating_point() and (v.dtype != torch.float32 or not v.is_contiguous())):
                    raise RuntimeError(f"{n}: the device trunk needs contiguous fp32 parameters on {x.device}")
            params = {n: v for n, v in t.items() if isinstance(v, nn.Parameter)}
            buffers = {n: v for n, v in t.items() if not isinstance(v, nn.Parameter)}
            eng = self._engine = ResNetTrunkEngine(self.num_blocks, params, buffers, max(b, eng.B if eng is not None else 0),
                                                   x.device, dtype, S)
            self._nbt_list = [v for n, v in buffers.items() if n.endswith("num_batches_tracked")]
            self._versions = None
        # parameters stepped by an optimiser / loaded / edited in place bump their version counters: re-pack, re-fold
        # (writes through ``param.data`` do not: call ``model._engine.invalidate()`` after those)
        v = tuple(t._version for t in eng.param.values()) + tuple(t._version for t in eng.buffers.values())
        if v != self._versions:
            eng.invalidate()
            self._versions = v
        return eng

    def forward(self, x):
        if x.is_cuda:
            out = _TrunkFunction.apply(self, x, self._anchor)
        else:
            out = F.relu(self.bn1(self.conv1(x)))
            out = self.layer4(self.layer3(self.layer2(self.layer1(out))))
        return self.linear(self.pooling(out).flatten(1))


def ResNet18(num_classes=10, act_dtype=None):
    return ResNet((2, 2, 2, 2), num_classes=num_classes, act_dtype=act_dtype)


def accuracy(output, target, topk=(1,)):
    """Fractions of samples whose label is among the k best logits (util.py:64-82), as 0-dim tensors."""
    maxk = min(max(topk), output.shape[1])
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1))
    return [correct[:min(k, maxk)].reshape(-1).float().sum() / target.shape[0] for k in topk]


def save_model(filename, epoch, model, optimizer, scheduler, save_best=False, **kwargs):
    """util.py:85-99: ``<filename>.pth`` = {epoch + 1, model / optimizer / scheduler state dicts, **kwargs (ENV)}."""
    state = {"epoch": epoch + 1, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
             "scheduler_state_dict": scheduler.state_dict() if scheduler is not None else None}
    state.update(kwargs)
    torch.save(state, filename + ".pth")
    if save_best:
        torch.save(state, filename + "_best.pth")


def load_model(filename, model, optimizer, scheduler, map_location="cpu", **kwargs):
    """util.py:102-111 (``strict=False``).  ``weights_only=True``: a checkpoint holds tensors, numbers and lists only."""
    ck = torch.load(filename + ".pth", map_location=map_location, weights_only=True)
    model.load_state_dict(ck["model_state_dict"], strict=False)
    if optimizer is not None and ck.get("optimizer_state_dict") is not None:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    if scheduler is not None and ck.get("scheduler_state_dict") is not None:
        scheduler.load_state_dict(ck["scheduler_state_dict"])
    return ck


class Evaluator:
    """evaluator.py:15-71: cross-entropy / top-1 / top-5 / confusion matrix over ``data_loader["test_dataset"]``."""

    def __init__(self, data_loader, logger, config):
        self.data_loader, self.logger, self.config = data_loader, logger, config
        lf = getattr(config, "log_frequency", None)
        self.log_frequency = lf if lf is not None else 100
        self.criterion = nn.CrossEntropyLoss()
        self.current_acc = self.current_acc_top5 = 0
        self._reset_stats()

    def _reset_stats(self):
        self.loss_meters, self.acc_meters, self.acc5_meters = AverageMeter(), AverageMeter(), AverageMeter()
        self.confusion_matrix = torch.zeros(self.config.num_classes, self.config.num_classes)

    def eval(self, epoch, model):
        model.eval()
        payload, i, used = {}, 0, 1e-9
        for i, (images, labels) in enumerate(self.data_loader["test_dataset"]):
            start = time.time()
            payload = self.eval_batch(images=images, labels=labels, model=model)
            used = time.time() - start
        if self.logger is not None and payload:
            self.logger.info(log_display(epoch=epoch, global_step=i, time_elapse=used, **payload))

    @torch.no_grad()
    def eval_batch(self, images, labels, model):
        dev = next(model.parameters()).device
        images, labels = images.to(dev, non_blocking=True), labels.to(dev, non_blocking=True)
        pred = model(images)
        loss = self.criterion(pred, labels)
        acc, acc5 = accuracy(pred, labels, topk=(1, 5))
        nc = self.config.num_classes
        cm = torch.bincount(labels * nc + pred.argmax(1), minlength=nc * nc).view(nc, nc)
        self.confusion_matrix += cm.cpu().float()
        n = images.size(0)
        self.loss_meters.update(loss.item(), n=n)
        self.acc_meters.update(acc.item(), n=n)
        self.acc5_meters.update(acc5.item(), n=n)
        return {"acc": acc.item(), "acc_avg": self.acc_meters.avg, "acc5": acc5.item(), "acc5_avg": self.acc5_meters.avg,
                "loss": loss.item(), "loss_avg": self.loss_meters.avg}


class DataParallelSGD:
    """Gradient averaging for the classifier stage with one process per GPU: ``sync(model)`` flattens every gradient into
    one bucket, all-reduces it once (RCCL) and scatters the mean back.  The reference wraps the classifier in
    ``torch.nn.DataParallel`` (main_vae_disentangle.py:497-498); per-replica BatchNorm statistics are the same here."""

    def __init__(self):
        self._bucket = None

    def sync(self, model):
        rank, world = D.world_info()
        if world == 1:
            return
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        n = sum(g.numel() for g in grads)
        if self._bucket is None or self._bucket.numel() != n:
            self._bucket = torch.empty(n, device=grads[0].device, dtype=torch.float32)
        torch.cat([g.reshape(-1) for g in grads], out=self._bucket)
        D.allreduce_sum_(self._bucket).div_(world)
        o = 0
        for g in grads:
            g.copy_(self._bucket[o:o + g.numel()].view_as(g))
            o += g.numel()


def new_env():
    """The ENV dictionary main() starts a run with (main_vae_disentangle.py:476-484)."""
    return {"global_step": 0, "best_acc": 0.0, "curren_acc": 0.0, "best_pgd_acc": 0.0, "train_history": [],
            "eval_history": [], "pgd_eval_history": [], "genotype_list": [], "cm_history": []}


def train(starting_epoch, model, model_s, model_c, optimizer, scheduler, criterion, trainer, evaluator, ENV, data_loader,
          config, logger=None, checkpoint_path_file=None, save_frequency=-1):
    """main_vae_disentangle.py:323-371.  Epoch 0 is the purification pass (``trainer.train`` overwrites the train set in
    place); epochs 1..config.epochs train the classifier (stage 1 only: in stage 0 ``trainer.train`` returns at once),
    each followed by a scheduler step, an evaluation on the clean test set and a checkpoint.  The reference reads
    ``config`` / ``logger`` / ``args`` / ``checkpoint_path_file`` from module globals; here they are arguments."""
    info = logger.info if logger is not None else (lambda *a: None)
    for epoch in range(starting_epoch, config.epochs + 1):
        info("")
        info("=" * 20 + "Training Epoch %d" % epoch + "=" * 20)
        ENV["global_step"] = trainer.train(epoch, model, model_s, model_c, criterion, optimizer)
        ENV["train_history"].append(trainer.acc_meters.avg * 100)
        if epoch > 0:
            scheduler.step()
        info("=" * 20 + "Eval Epoch %d" % epoch + "=" * 20)
        evaluator.eval(epoch, model)
        info("Eval Loss:%.4f\tEval acc: %.2f" % (evaluator.loss_meters.avg, evaluator.acc_meters.avg * 100))
        ENV["eval_history"].append(evaluator.acc_meters.avg * 100)
        ENV["curren_acc"] = evaluator.acc_meters.avg * 100
        ENV["cm_history"].append(evaluator.confusion_matrix.cpu().numpy().tolist())
        trainer._reset_stats()
        evaluator._reset_stats()
        rank, _ = D.world_info()
        if checkpoint_path_file is not None and rank == 0:
            os.makedirs(os.path.dirname(checkpoint_path_file) or ".", exist_ok=True)
            save_model(ENV=ENV, epoch=epoch, model=model, optimizer=optimizer, scheduler=scheduler, save_best=False,
                       filename=checkpoint_path_file)
            info("Model Saved at %s" % checkpoint_path_file)
            if save_frequency > 0 and epoch % save_frequency == 0:
                save_model(ENV=ENV, epoch=epoch, model=model, optimizer=optimizer, scheduler=scheduler,
                           filename=checkpoint_path_file + "_epoch%d" % epoch)
