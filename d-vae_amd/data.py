"""Minimal stand-ins for the reference's dataset layer, as far as the hot path touches it.

The reference builds datasets through ``dataset.DatasetGenerator`` (/root/reference/dataset.py:138-529),
which needs torchvision / mlconfig and downloads CIFAR; none of that is on the accelerated path.  What
the path needs is (a) a dataset object exposing ``.data`` (uint8 [N,H,W,3]), ``.targets`` and
``.transform`` -- the attributes ``Trainer.train`` overwrites in place
(main_vae_disentangle.py:67, :85-87) -- and (b) ``getDataLoader(train_shuffle, train_drop_last)``
with the semantics of dataset.py:451-464.  The on-disk layout of a saved purified set is the one
``CIFAR10_numpy`` reads back: ``cifar10_data.npy`` (uint8 NHWC) + ``cifar10_target.npy``
(dataset.py:1048-1049).
"""
from __future__ import annotations

import os

import numpy as np
import torch


class ArrayDataset:
    """uint8 NHWC image array + integer targets; ``transform`` None means ToTensor (CHW float / 255)."""

    def __init__(self, data: np.ndarray, targets, transform=None):
        assert data.dtype == np.uint8 and data.ndim == 4 and data.shape[-1] == 3
        self.data = data
        self.targets = np.asarray(targets, dtype=np.int64)
        self.transform = transform

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, i):
        img = torch.from_numpy(self.data[i]).permute(2, 0, 1).float().div(255)
        if self.transform is not None and not isinstance(self.transform, str):
            img = self.transform(img)
        return img, int(self.targets[i])


def synthetic_cifar(n=50000, size=32, classes=10, seed=0):
    rng = np.random.default_rng(seed)
    return ArrayDataset(rng.integers(0, 256, (n, size, size, 3), dtype=np.uint8), np.arange(n) % classes)


class DeviceLoader:
    """Iterates (images float32 NCHW in [0,1], labels int64) batches from a dataset resident in HBM.

    Semantics of ``DataLoader(shuffle, drop_last)`` with ``transform = ToTensor()``: one pass over the
    (optionally shuffled) index order, batches of ``batch_size``, the ragged tail kept unless drop_last.
    With world > 1 each rank yields its slice of every *global* batch (``batch_size`` is per rank)."""

    def __init__(self, dataset: ArrayDataset, batch_size: int, shuffle: bool, drop_last: bool, device="cuda", seed=0,
                 rank=0, world=1):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        self.device, self.seed, self.rank, self.world = torch.device(device), seed, rank, world
        self.epoch = 0
        self._dev_data = None
        self._dev_targets = None

    def resident(self):
        """(uint8 [N,H,W,3], int64 [N]) device copies, refreshed if the host array was replaced."""
        if self._dev_data is None or self._src is not self.dataset.data:
            self._src = self.dataset.data
            self._dev_data = torch.from_numpy(self.dataset.data).to(self.device)
            self._dev_targets = torch.from_numpy(self.dataset.targets).to(self.device)
        return self._dev_data, self._dev_targets

    def batches(self):
        """Index tensors (device, int64) of this rank's batches for the current epoch."""
        n = len(self.dataset)
        gb = self.batch_size * self.world
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed * 1_000_003 + self.epoch)
            order = torch.randperm(n, generator=g)
        else:
            order = torch.arange(n)
        stop = n - n % gb if self.drop_last else n
        out = []
        for lo in range(0, stop, gb):
            idx = order[lo:lo + gb]
            if self.world > 1:
                per = (idx.numel() + self.world - 1) // self.world
                idx = idx[self.rank * per:(self.rank + 1) * per]
            if idx.numel():
                out.append(idx.to(self.device))
        return out

    def __len__(self):
        n = len(self.dataset)
        gb = self.batch_size * self.world
        return n // gb if self.drop_last else (n + gb - 1) // gb

    def __iter__(self):
        data, targets = self.resident()
        for idx in self.batches():
            x = data[idx].permute(0, 3, 1, 2).float().div_(255)
            yield x, targets[idx]
        self.epoch += 1


class DatasetGenerator:
    """``datasets`` dict + ``getDataLoader`` with the reference's signature (dataset.py:451-464)."""

    def __init__(self, train_dataset: ArrayDataset, test_dataset: ArrayDataset | None = None, train_batch_size=128,
                 eval_batch_size=256, device="cuda", seed=0, rank=0, world=1):
        self.datasets = {"train_dataset": train_dataset, "test_dataset": test_dataset or train_dataset}
        self.train_batch_size, self.eval_batch_size = train_batch_size, eval_batch_size
        self.device, self.seed, self.rank, self.world = device, seed, rank, world

    def getDataLoader(self, train_shuffle=True, train_drop_last=True):
        return {
            "train_dataset": DeviceLoader(self.datasets["train_dataset"], self.train_batch_size, train_shuffle, train_drop_last,
                                          self.device, self.seed, self.rank, self.world),
            "test_dataset": DeviceLoader(self.datasets["test_dataset"], self.eval_batch_size, False, False, self.device,
                                         self.seed, 0, 1),
        }


def save_purified_numpy(dataset: ArrayDataset, root: str, prefix="cifar10"):
    """Write the purified set in the layout ``CIFAR10_numpy`` loads (dataset.py:1048-1049)."""
    os.makedirs(root, exist_ok=True)
    np.save(os.path.join(root, f"{prefix}_data.npy"), dataset.data)
    np.save(os.path.join(root, f"{prefix}_target.npy"), dataset.targets)
