"""Exploration for the downstream-accuracy test: which (separation, clutter, noise) of the synthetic stand-in puts a
ResNet-18 trained on clean data at 80-92 % clean test accuracy, how far below that the unpurified poisoned set lands, and
what D-VAE purification (fp32 / bf16 trained) recovers.  Prints one line per measurement."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dvae_amd
from dvae_amd import data as DD
from dvae_amd.classifier import Evaluator, new_env, train
from dvae_amd.model import ResNet18DVAE
from dvae_amd.optim import FusedAdam
from dvae_amd.trainer import Trainer, VAE_Trainer, purify_dataset



class Cfg:
    log_frequency = 1000
    grad_clip = 5.0
    epochs = 60
    num_classes = 10


def make_model(params=None, dtype="f32"):
    m = ResNet18DVAE(num_classes=10, stem_end_block=7, latent_dim=256, act_dtype=dtype)
    m.sample_wise, m.use_y, m.spatial_emb = True, False, True
    m.set_spatial_emb()
    if params is not None:
        m.load_state_dict(params, strict=False)
    return m.cuda()


def classifier_acc(arr, targets, test, epochs, seed):
    cfg = dvae_amd.config.from_dict(dict(dvae_amd.config.CIFAR10_RESNET18, epochs=epochs, log_frequency=100000))
    torch.manual_seed(seed)
    tds = DD.ArrayDataset(arr.copy(), targets.copy(), transform="crop_flip")
    g2 = DD.DatasetGenerator(tds, test, train_batch_size=128, eval_batch_size=512, device="cuda", seed=7)
    ld = g2.getDataLoader()
    model = cfg.model().cuda()
    optimizer = cfg.optimizer(model.parameters())
    scheduler = cfg.scheduler(optimizer)
    trainer = Trainer(cfg.criterion(), ld, None, cfg, datasets_generator=g2, iters=1)
    ENV = new_env()
    t0 = time.time()
    train(1, model, None, None, optimizer, scheduler, cfg.criterion(), trainer, Evaluator(ld, None, cfg), ENV, ld, cfg)
    print("   classifier", round(ENV["eval_history"][-1], 2), "in", round(time.time() - t0, 1), "s", flush=True)
    return round(ENV["eval_history"][-1], 2)


def train_dvae(ds, test, dtype, epochs, seed=2):
    gen = DD.DatasetGenerator(ds, test, train_batch_size=128, eval_batch_size=512, device="cuda", seed=seed)
    loaders = gen.getDataLoader(train_shuffle=True, train_drop_last=True)
    torch.manual_seed(0)
    m = make_model(dtype=dtype)
    m.train()
    opt = FusedAdam(m, lr=5e-3)
    tr = VAE_Trainer(loaders, None, Cfg(), stage=1, kd=3.0)
    for epoch in range(epochs):
        for images, labels in loaders["train_dataset"]:
            pay = tr.train_batch(images, labels, m, None, opt, epoch)
            tr.global_step += 1
    p = tr.materialize(pay)
    return m, {k: round(p[k], 4) for k in ("loss_rec", "loss_rec_p", "loss_p", "kld_loss")}


settings = [tuple(float(v) for v in s.split(",")) for s in sys.argv[1:]] or [(1.0, 0.25, 0.05, 8, 8)]
N, NT, CE = 4096, 8192, int(os.environ.get("CE", "8"))
for sep, clutter, noise, eps255, grid in settings:
    tag = f"sep={sep} clutter={clutter} noise={noise} eps={eps255}/255 grid={int(grid)}"
    clean = DD.synthetic_separable(n=N, seed=21, separation=sep, clutter=clutter, noise=noise)
    test = DD.synthetic_separable(n=NT, seed=22, separation=sep, clutter=clutter, noise=noise)
    ds = DD.ArrayDataset(clean.data.copy(), clean.targets.copy())
    DD.poison_dataset(ds, DD.synthetic_classwise_poison(10, 32, eps255 / 255, seed=3, grid=int(grid)), "classwise")
    t0 = time.time()
    print(tag, "clean", [classifier_acc(clean.data, clean.targets, test, CE, s) for s in (1, 2)], flush=True)
    print(tag, "poisoned", [classifier_acc(ds.data, ds.targets, test, CE, s) for s in (1, 2)], "t", round(time.time() - t0, 1), flush=True)
    for dt in (() if os.environ.get("NO_DVAE") else ("f32", "bf16")):
        t0 = time.time()
        m, pay = train_dvae(ds, test, dt, int(os.environ.get("VE", "30")))
        t1 = time.time()
        pur = DD.ArrayDataset(ds.data.copy(), ds.targets.copy())
        purify_dataset(m, pur, 1, batch=512, seed=31)
        print(tag, "dvae", dt, pay, "train_s", round(t1 - t0, 1), "purified",
              [classifier_acc(pur.data, pur.targets, test, CE, s) for s in (1, 2)], flush=True)
