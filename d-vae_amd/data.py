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
        self.version = 0            # bump after overwriting ``data`` in place: device-resident loaders refresh their copy

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


def _upsample(a, size):
    """[..., g, g, 3] -> [..., size, size, 3] by separable linear interpolation (numpy only)."""
    g = a.shape[-3]
    pos = (np.arange(size) + 0.5) * g / size - 0.5
    i0 = np.clip(np.floor(pos).astype(int), 0, g - 1)
    i1 = np.clip(i0 + 1, 0, g - 1)
    w = np.clip(pos - i0, 0.0, 1.0).astype(np.float32)
    a = a[..., i0, :, :] * (1 - w)[:, None, None] + a[..., i1, :, :] * w[:, None, None]
    return a[..., :, i0, :] * (1 - w)[:, None] + a[..., :, i1, :] * w[:, None]


def synthetic_separable(n=50000, size=32, classes=10, seed=0, template_seed=1234, clutter=0.25, noise=0.05, separation=1.0):
    """A class-separable stand-in for CIFAR (no dataset can be fetched here): every class has a fixed smooth colour
    template (shared by train and test sets through ``template_seed``); an image is its class template at a random
    contrast plus a smooth random clutter field and pixel noise.  A ResNet-18 learns it in a few epochs, and a
    class-wise additive pattern of a few grey levels is an easier shortcut than the templates -- which is what makes an
    unlearnable-example poison work.  ``separation`` < 1 pulls the class templates towards their common mean (1 = as
    drawn), ``clutter`` scales the class-independent field that lives in the same 4 x 4 colour space: together they
    set how much of the set a perfect classifier can get right, i.e. how hard the stand-in is."""
    trng = np.random.default_rng(template_seed)
    t44 = trng.uniform(0.1, 0.9, (classes, 4, 4, 3)).astype(np.float32)
    if separation != 1.0:
        t44 = t44.mean(0, keepdims=True) + np.float32(separation) * (t44 - t44.mean(0, keepdims=True))
    templates = _upsample(t44, size)
    rng = np.random.default_rng(seed)
    y = (np.arange(n) % classes).astype(np.int64)
    out = np.empty((n, size, size, 3), dtype=np.uint8)
    for lo in range(0, n, 4096):
        hi = min(n, lo + 4096)
        m = hi - lo
        contrast = rng.uniform(0.6, 1.0, (m, 1, 1, 1)).astype(np.float32)
        field = _upsample(rng.uniform(-1, 1, (m, 4, 4, 3)).astype(np.float32), size) * clutter
        img = templates[y[lo:hi]] * contrast + field + rng.normal(0, noise, (m, size, size, 3)).astype(np.float32)
        out[lo:hi] = (np.clip(img, 0, 1) * 255).astype(np.uint8)
    return ArrayDataset(out, y)


def synthetic_classwise_poison(classes=10, size=32, eps=8.0 / 255, seed=0, grid=8):
    """A class-wise shortcut perturbation in the ``perturbation.pt`` layout ([classes,3,H,W] float, values +-eps): a
    random +-eps pattern on a ``grid`` x ``grid`` lattice per class (8: blocks of 4 x 4 pixels; ``size``: every pixel its
    own sign), as error-minimising noise is up to its L-inf bound."""
    rng = np.random.default_rng(seed)
    g = rng.choice([-1.0, 1.0], (classes, 3, grid, grid)).astype(np.float32)
    p = np.kron(g, np.ones((size // grid, size // grid), dtype=np.float32))
    return torch.from_numpy(p * eps)


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
        ver = getattr(self.dataset, "version", 0)      # bumped by whoever overwrites dataset.data in place
        if self._dev_data is None or self._src is not self.dataset.data or self._ver != ver:
            self._src, self._ver = self.dataset.data, ver
            # whoever wrote this version of dataset.data may have left its device copy behind (the purification pass
            # does, after its all-gather): same bytes, no upload
            dc = getattr(self.dataset, "_device_copy", None)
            if dc is not None and dc[0] == ver and dc[1].device == self.device and dc[1].shape == self.dataset.data.shape:
                self._dev_data = dc[1]
            else:
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


def poison_dataset(dataset: ArrayDataset, perturbation: torch.Tensor, perturb_type="samplewise", poison_rate=1.0, seed=0,
                   poison_samples_idx=None, device="cuda"):
    """Apply unlearnable-example noise to a uint8 set the way ``PoisonCIFAR10.__init__`` does
    (dataset.py:1231-1289): ``data = uint8(clip(float32(data) + clamp(255 * perturbation, -255, 255), 0, 255))`` for the
    selected samples, with ``perturbation`` the float tensor of ``perturbation.pt`` ([N,3,H,W] sample-wise or
    [classes,3,H,W] class-wise, full-image noise).  The reference forms ``255 * perturbation`` and the clamp in the
    file's own dtype (dataset.py:1235) and the result becomes float32 when it is written into the float32 image-sized
    mask (dataset.py:534, :550).  A float32 file goes to the kernel as it is; for any other dtype (the
    ``perturbation{it}.pt`` files of main_vae_disentangle.py:91-92 are float64) product and clamp are formed in that
    dtype on the device and rounded to float32 once -- casting the file to float32 first can move a pixel by one grey
    level.  The arithmetic runs on the device (libdvae_hip); ``dataset.data`` is overwritten in place.  Returns the sorted indices of the poisoned samples
    (``poison_samples_idx``; with ``poison_rate`` < 1 they are drawn without replacement from a generator seeded with
    ``seed``, not from numpy's global state as the reference does)."""
    from . import _lib as L
    n, H, W, _ = dataset.data.shape
    pert = perturbation.detach()
    if not pert.dtype.is_floating_point:
        raise ValueError(f"perturbation must be a floating-point tensor, got {pert.dtype}")
    if pert.dim() != 4 or tuple(pert.shape[1:]) != (3, H, W):
        raise ValueError(f"perturbation must be [rows,3,{H},{W}] (full-image noise), got {tuple(pert.shape)}")
    rows = pert.shape[0]
    if perturb_type == "samplewise":
        if rows != n:
            raise ValueError("Poison Perturb Tensor size not match for samplewise")
    elif perturb_type == "classwise":
        if rows <= int(dataset.targets.max()):
            raise ValueError("Poison Perturb Tensor size not match for classwise")
    else:
        raise ValueError("perturb_type must be 'samplewise' or 'classwise'")
    if poison_samples_idx is None:
        k = int(n * poison_rate)
        if k >= n:
            poison_samples_idx = np.arange(n)
        else:
            poison_samples_idx = np.sort(np.random.default_rng(seed).choice(n, k, replace=False))
    poison_samples_idx = np.asarray(poison_samples_idx, dtype=np.int64)
    dev = torch.device(device)
    d = torch.from_numpy(dataset.data).to(dev)
    mask = None
    if poison_samples_idx.size != n:
        mask = torch.zeros(n, dtype=torch.uint8, device=dev)
        mask[torch.from_numpy(poison_samples_idx).to(dev)] = 1
    labels = torch.from_numpy(dataset.targets).to(dev) if perturb_type == "classwise" else None
    pd = pert.contiguous().to(dev)
    prescaled = int(pd.dtype != torch.float32)
    if prescaled:
        pd = pd.mul(255).clamp_(-255, 255).to(torch.float32)
    L.check(L.lib().dvae_poison_u8(L.ptr(d), L.ptr(pd), L.ptr(labels), L.ptr(mask), L.ptr(d), n, H * W, prescaled, L.stream()))
    dataset.data[...] = d.cpu().numpy()
    dataset.version = getattr(dataset, "version", 0) + 1
    dataset.poison_samples_idx = poison_samples_idx.tolist()
    return dataset.poison_samples_idx


def load_purified_numpy(root: str, prefix="cifar10", transform=None):
    """Read back a set saved by ``save_purified_numpy`` (the files ``CIFAR10_numpy`` loads, dataset.py:1048-1049)."""
    data = np.load(os.path.join(root, f"{prefix}_data.npy"))            # allow_pickle stays False
    targets = np.load(os.path.join(root, f"{prefix}_target.npy"))
    return ArrayDataset(data, targets, transform)


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
