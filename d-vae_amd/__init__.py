"""MI355X-native implementation of the D-VAE hot path (train step + purification pass).

Host code is Python on PyTorch-ROCm; all arithmetic on the path runs in hand-written gfx950 HIP
kernels behind the C ABI of ``include/dvae_hip.h`` (``lib/libdvae_hip.so``, built by ``make``).
"""
from . import _lib, classifier, config, data, dist
from ._lib import BF16, F32
from .classifier import Evaluator, ResNet18
from .engine import DVAEConfig, DVAEEngine
from .model import ResNet18DVAE, ResNetDVAE
from .optim import FusedAdam
from .trainer import Trainer, VAE_Trainer, purify_dataset

__all__ = ["_lib", "F32", "BF16", "DVAEConfig", "DVAEEngine", "ResNet18DVAE", "ResNetDVAE", "FusedAdam", "Trainer",
           "VAE_Trainer", "purify_dataset", "ResNet18", "Evaluator", "classifier", "config", "data", "dist"]
