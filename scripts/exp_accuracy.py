"""Exploration: how stable is the downstream-accuracy check?  Trains a D-VAE on the poisoned synthetic set, purifies it
(fp32 and bf16 HIP paths), trains ResNet-18 on each, prints clean test accuracy for a few settings and seeds."""
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
from dvae_amd.trainer import Trainer, VAE_Trainer


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
    return ENV["eval_history"][-1], time.time() - t0


N, NT = 4096, 4096
clean = DD.synthetic_separable(n=N, seed=21)
test = DD.synthetic_separable(n=NT, seed=22)
ds = DD.ArrayDataset(clean.data.copy(), clean.targets.copy())
DD.poison_dataset(ds, DD.synthetic_classwise_poison(10, 32, 8 / 255, seed=3), "classwise")
for ep in (8, 15):
    for seed in (1, 2):
        print("clean data, classifier epochs", ep, "seed", seed, classifier_acc(clean.data, clean.targets, test, ep, seed), flush=True)
        print("poisoned data, classifier epochs", ep, "seed", seed, classifier_acc(ds.data, ds.targets, test, ep, seed), flush=True)
gen = DD.DatasetGenerator(ds, test, train_batch_size=128, eval_batch_size=512, device="cuda", seed=2)
loaders = gen.getDataLoader(train_shuffle=True, train_drop_last=True)
m = make_model()
m.train()
opt = FusedAdam(m, lr=5e-3)
tr = VAE_Trainer(loaders, None, Cfg(), stage=1, kd=3.0)
eps = torch.randn(2, N, 256, 16, 16, generator=torch.Generator().manual_seed(31))
steps = 0
for epochs_s in (10, 30):
    t0 = time.time()
    while steps < epochs_s * 32:
        for images, labels in loaders["train_dataset"]:
            pay = tr.train_batch(images, labels, m, None, opt, steps // 32)
            tr.global_step += 1
            steps += 1
    print("D-VAE steps", steps, "time", time.time() - t0, tr.materialize(pay), flush=True)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for stage in (0, 1):
        pur = {}
        for dt in ("f32", "bf16"):
            md = make_model(sd, dt)
            md.eval()
            eng = md._engine_for(torch.zeros(128, 3, 32, 32, device="cuda"))
            d = torch.from_numpy(ds.data).cuda()
            out = torch.zeros_like(d)
            lab = torch.from_numpy(ds.targets).cuda()
            for lo in range(0, N, 128):
                eng.purify_batch(d, out, None, lab, lo, 128, stage, eps_nchw=(eps[0, lo:lo + 128].cuda(), eps[1, lo:lo + 128].cuda()))
            pur[dt] = out.cpu().numpy()
        dq = np.abs(pur["f32"].astype(int) - pur["bf16"].astype(int))
        print("stage", stage, "f32 vs bf16 purified: mean", dq.mean(), "max", dq.max(), flush=True)
        for ep in (8, 15):
            for seed in (1, 2):
                for dt in ("f32", "bf16"):
                    print("D-VAE steps", steps, "stage", stage, dt, "classifier epochs", ep, "seed", seed,
                          classifier_acc(pur[dt], ds.targets, test, ep, seed), flush=True)
