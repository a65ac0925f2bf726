"""Fused clip_grad_norm_ + Adam over the engine's flat parameter bucket.

Replaces the pair ``torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)`` +
``torch.optim.Adam.step()`` of /root/reference/main_vae_disentangle.py:210-211 (Adam lr 5e-3,
weight_decay 0: configs/cifar10/resnet18DVAE.yaml:21-23) with two launches of libdvae_hip.  It is a
``torch.optim.Optimizer`` so the reference's ``config_s.scheduler(optimizer)`` (CosineAnnealingLR) and
``optimizer.param_groups[0]['lr']`` keep working.  Parameters off the hot path never receive a
gradient in the D-VAE stage (as in the reference, where Adam skips ``grad is None``) and are not touched.
"""
import torch


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=5.0):
        if weight_decay != 0.0:
            raise ValueError("the D-VAE configs use weight_decay 0; decay is not implemented in the fused kernel")
        self.model = model
        self.max_norm = max_norm
        self._pushed = None
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def zero_grad(self, set_to_none: bool = True):
        # gradients live in the engine bucket and are overwritten by the next backward
        self.model._accumulate = False

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        eng = self.model._engine
        if eng is None:
            raise RuntimeError("FusedAdam.step() before any forward/backward on the GPU")
        g = self.param_groups[0]
        key = (id(eng), g["lr"], g["betas"], g["eps"], self.max_norm, grad_scale)
        if key != self._pushed:
            eng.set_hyper(lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], max_norm=self.max_norm,
                          grad_scale=grad_scale)
            self._pushed = key
        eng.adam_step()
        self.model._accumulate = False
        return None
