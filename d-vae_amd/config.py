"""A small stand-in for the ``mlconfig`` package the reference loads its YAML experiment files with
(/root/reference/main_vae_disentangle.py:315-317, :402-404; files such as configs/cifar10/resnet18DVAE.yaml).

Same file format and the same access pattern: top-level keys as attributes (``config.epochs``, ``config.grad_clip``,
``config.log_frequency``), blocks with a ``name`` as factories (``config.model()``, ``config.optimizer(params)``,
``config.scheduler(optimizer)``, ``config.criterion()``), ``$key`` substitution from the top level
(``T_max: $epochs``).  Only the names the D-VAE experiments use are registered.
"""
from __future__ import annotations

import yaml
import torch


def _registry():
    from .classifier import ResNet18
    from .model import ResNet18DVAE
    from .optim import FusedAdam
    return {
        "ResNet18DVAE": ResNet18DVAE,
        "ResNet18": ResNet18,
        "SGD": torch.optim.SGD,
        "Adam": torch.optim.Adam,
        "FusedAdam": FusedAdam,
        "CosineAnnealingLR": torch.optim.lr_scheduler.CosineAnnealingLR,
        "MultiStepLR": torch.optim.lr_scheduler.MultiStepLR,
        "StepLR": torch.optim.lr_scheduler.StepLR,
        "ExponentialLR": torch.optim.lr_scheduler.ExponentialLR,
        "CrossEntropyLoss": torch.nn.CrossEntropyLoss,
    }


class _Block(dict):
    def __init__(self, d, root):
        super().__init__(d)
        self._root = root

    def kwargs(self):
        out = {}
        for k, v in self.items():
            if k == "name":
                continue
            if isinstance(v, str) and v.startswith("$"):
                v = self._root[v[1:]]
            out[k] = v
        return out

    def __call__(self, *args, **kw):
        name = self["name"]
        reg = _registry()
        if name not in reg:
            raise KeyError(f"config block names {name!r}, which is not registered ({sorted(reg)})")
        return reg[name](*args, **{**self.kwargs(), **kw})


class Config(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            return None              # mlconfig's Config answers None for absent keys (the trainers rely on it)
        return _Block(v, self) if isinstance(v, dict) else v

    def set_immutable(self):
        return self


def load(path: str) -> Config:
    with open(path) as f:
        return Config(yaml.load(f, Loader=yaml.SafeLoader))


def from_dict(d: dict) -> Config:
    return Config(d)


CIFAR10_DVAE = {
    # configs/cifar10/resnet18DVAE.yaml
    "num_classes": 10, "epochs": 60, "grad_clip": 5.0, "log_frequency": 100,
    "model": {"name": "ResNet18DVAE", "num_classes": 10, "stem_end_block": 7, "latent_dim": 256},
    "criterion": {"name": "CrossEntropyLoss"},
    "optimizer": {"name": "Adam", "lr": 0.005, "weight_decay": 0.0},
    "scheduler": {"name": "CosineAnnealingLR", "T_max": "$epochs", "eta_min": 0.0},
    "dataset": {"name": "DatasetGenerator", "train_batch_size": 128, "eval_batch_size": 128},
}
CIFAR10_RESNET18 = {
    # configs/cifar10/resnet18.yaml
    "num_classes": 10, "epochs": 60, "grad_clip": 5.0, "log_frequency": 100,
    "model": {"name": "ResNet18", "num_classes": 10},
    "criterion": {"name": "CrossEntropyLoss"},
    "optimizer": {"name": "SGD", "lr": 0.1, "weight_decay": 5.0e-4, "momentum": 0.9},
    "scheduler": {"name": "CosineAnnealingLR", "T_max": "$epochs", "eta_min": 0.0},
    "dataset": {"name": "DatasetGenerator", "train_batch_size": 128, "eval_batch_size": 128},
}
