#!/usr/bin/env python3
"""Two-stage D-VAE purification + classifier retrain: the orchestration of the reference's
``main_vae_disentangle.py`` (main(): /root/reference/main_vae_disentangle.py:374-501, flags :258-288) on the MI355X
path of this repository.

    python run_disentangle.py --config_path configs/cifar10 --version resnet18 --version_s resnet18DVAE \\
        --exp_name experiments/cifar10/em8_train_bn_1.0_VAE_DIS_kd3.0_Z_spatial --kd 3.0 --train \\
        --train_data_type PoisonCIFAR10 --perturb_tensor_filepath experiments/cifar10/em8/perturbation.pt \\
        --perturb_type samplewise --use_y --spatial_emb

For each of the two stages: train a fresh D-VAE on the (current) train set for ``config_s.epochs`` epochs, save
``<exp_name>/model_s{stage}.pth``, purify the train set in place (stage 0: x - p; stage 1: rec(x - p)), save
``<exp_name>/perturbation{stage}.pt``; stage 1 then trains the classifier on the purified set for ``config.epochs``
epochs with an evaluation on the clean test set and a checkpoint after every epoch.  Stage 1's D-VAE therefore trains on
stage 0's output, because the array was overwritten in place.

Flag traps kept from the reference: ``--use_y`` and ``--sample_wise_rp`` are ``store_false`` (passing ``--use_y`` turns
use_y OFF, which is what the published command lines do).

Data.  No dataset can be downloaded here.  ``--train_data_path`` / ``--test_data_path`` name directories that hold
``cifar10_data.npy`` (uint8 NHWC) + ``cifar10_target.npy`` -- the layout the reference's ``CIFAR10_numpy`` reads
(dataset.py:1048-1049) -- and ``--train_data_type Synthetic`` / ``PoisonSynthetic`` builds a class-separable synthetic
set instead.  ``Poison*`` types add ``--perturb_tensor_filepath`` to the train set the way ``PoisonCIFAR10`` does.
With torchrun (WORLD_SIZE > 1) every rank runs this script: D-VAE training shards each global batch, purification
shards by image index, rank 0 writes the files.
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import dvae_amd  # noqa: E402,F401
from dvae_amd import config as CFG  # noqa: E402
from dvae_amd import data as DD  # noqa: E402
from dvae_amd import dist as D  # noqa: E402
from dvae_amd.classifier import Evaluator, load_model, new_env, train  # noqa: E402
from dvae_amd.optim import FusedAdam  # noqa: E402
from dvae_amd.trainer import Trainer, VAE_Trainer  # noqa: E402


def parse(argv=None):
    p = argparse.ArgumentParser(description="ClasswiseNoise")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--version", type=str, default="resnet18")
    p.add_argument("--version_s", type=str, default="resnet18_s")
    p.add_argument("--exp_name", type=str, default="test_exp")
    p.add_argument("--config_path", type=str, default="configs/cifar10")
    p.add_argument("--load_model", action="store_true", default=False)
    p.add_argument("--data_parallel", action="store_true", default=False,
                   help="accepted for compatibility; parallelism here is one process per GPU (launch with torchrun)")
    p.add_argument("--train", action="store_true", default=False)
    p.add_argument("--save_frequency", default=-1, type=int)
    p.add_argument("--train_batch_size", default=128, type=int)
    p.add_argument("--eval_batch_size", default=256, type=int)
    p.add_argument("--num_of_workers", default=8, type=int, help="accepted for compatibility; the set is HBM-resident")
    p.add_argument("--train_data_type", type=str, default="CIFAR10")
    p.add_argument("--test_data_type", type=str, default="CIFAR10")
    p.add_argument("--train_data_path", type=str, default="../datasets")
    p.add_argument("--test_data_path", type=str, default="../datasets")
    p.add_argument("--perturb_type", default="classwise", type=str, choices=["classwise", "samplewise"])
    p.add_argument("--poison_rate", default=1.0, type=float)
    p.add_argument("--perturb_tensor_filepath", default=None, type=str)
    p.add_argument("--kd", default=2.5, type=float, help="target kld loss")
    p.add_argument("--sample_wise_rp", action="store_false")
    p.add_argument("--use_y", action="store_false")
    p.add_argument("--spatial_emb", action="store_true")
    # not in the reference
    p.add_argument("--dtype", default="bf16", choices=["f32", "bf16"], help="arithmetic of the D-VAE path")
    p.add_argument("--purify_batch", default=2048, type=int)
    p.add_argument("--epochs_s", default=None, type=int, help="override config_s.epochs (D-VAE epochs per stage)")
    p.add_argument("--epochs", default=None, type=int, help="override config.epochs (classifier epochs)")
    p.add_argument("--synthetic_n", default=50000, type=int)
    p.add_argument("--synthetic_test_n", default=10000, type=int)
    p.add_argument("--monitor_every", default=None, type=int,
                   help="evaluate the four model_c monitor losses every this many steps (default: log_frequency; 0 = never)")
    p.add_argument("--stock_optimizer", action="store_true",
                   help="drive the D-VAE with the optimiser of the YAML file through autograd instead of the fused step")
    return p.parse_args(argv)


def setup_logger(name, log_file, console=True):
    logger = logging.getLogger(name + str(time.time()))
    logger.setLevel(logging.INFO)
    fmt = logging.Formatter("%(asctime)s %(message)s")
    fh = logging.FileHandler(log_file)
    fh.setFormatter(fmt)
    logger.addHandler(fh)
    if console:
        ch = logging.StreamHandler()
        ch.setFormatter(fmt)
        logger.addHandler(ch)
    return logger


def load_config(config_path, version, builtin):
    f = os.path.join(config_path, version) + ".yaml"
    if os.path.exists(f):
        return CFG.load(f), f
    return CFG.from_dict(dict(builtin)), None


def build_datasets(args, num_classes, rank, world):
    def one(kind, path, n, seed, train):
        if "Synthetic" in kind:
            return DD.synthetic_separable(n=n, classes=num_classes, seed=seed)
        return DD.load_purified_numpy(path, "cifar10")
    train_ds = one(args.train_data_type, args.train_data_path, args.synthetic_n, args.seed, True)
    test_ds = one(args.test_data_type, args.test_data_path, args.synthetic_test_n, args.seed + 1, False)
    gen = DD.DatasetGenerator(train_ds, test_ds, args.train_batch_size, args.eval_batch_size, "cuda", args.seed, rank, world)
    if "Poison" in args.train_data_type:
        if args.perturb_tensor_filepath is None:
            raise SystemExit("Poison* train data needs --perturb_tensor_filepath")
        pert = torch.load(args.perturb_tensor_filepath, map_location="cpu", weights_only=True)
        DD.poison_dataset(train_ds, pert, args.perturb_type, args.poison_rate, args.seed)
    return gen


def main(argv=None):
    args = parse(argv)
    rank, world = D.init_from_env()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    torch.manual_seed(args.seed)
    exp_path = os.path.join(args.exp_name, args.version)
    checkpoint_path = os.path.join(exp_path, "checkpoints")
    checkpoint_path_file = os.path.join(checkpoint_path, args.version)
    os.makedirs(checkpoint_path, exist_ok=True)
    logger = setup_logger(args.version, os.path.join(exp_path, args.version) + (".log" if rank == 0 else f".rank{rank}.log"),
                          console=rank == 0)
    for k, v in vars(args).items():
        logger.info("%s: %s" % (k, v))
    config, cf = load_config(args.config_path, args.version, CFG.CIFAR10_RESNET18)
    config_s, csf = load_config(args.config_path, args.version_s, CFG.CIFAR10_DVAE)
    logger.info("classifier config: %s; D-VAE config: %s" % (cf or "built-in cifar10/resnet18", csf or "built-in cifar10/resnet18DVAE"))
    if args.epochs is not None:
        config["epochs"] = args.epochs
    if args.epochs_s is not None:
        config_s["epochs"] = args.epochs_s

    model = config.model().cuda()
    gen = build_datasets(args, config.num_classes, rank, world)
    train_ds = gen.datasets["train_dataset"]
    logger.info("Training Dataset: %d images %s" % (len(train_ds), train_ds.data.shape[1:]))
    if "Poison" in args.train_data_type and rank == 0:
        np.save(os.path.join(exp_path, "poison_targets.npy"), np.array(train_ds.poison_samples_idx))

    for it in range(2):
        train_ds.transform = None                                  # ToTensor only while the D-VAE trains (:409)
        data_loader = gen.getDataLoader(train_shuffle=True, train_drop_last=True)
        model_s = config_s.model()
        model_s.sample_wise, model_s.use_y, model_s.spatial_emb = args.sample_wise_rp, args.use_y, args.spatial_emb
        model_s.set_spatial_emb()
        model_s = model_s.cuda()
        model_s.act_dtype = args.dtype
        model_c = config.model().cuda()
        ck = "./{}/resnet18/checkpoints/resnet18".format("_".join(args.exp_name.split("_")[:4]))
        if os.path.exists(ck + ".pth"):
            load_model(ck, model_c, None, None, map_location="cuda")
        else:
            logger.info("monitor classifier checkpoint %s.pth not found: model_c keeps its random initialisation "
                        "(it only feeds the four logged monitor losses)" % ck)
        model_c.cam_register(layer="layer2")

        trainer = VAE_Trainer(data_loader, logger, config, target="train_dataset", stage=it, kd=args.kd,
                              monitor_every=args.monitor_every)
        oc = config_s.optimizer
        if args.stock_optimizer or oc["name"] != "Adam" or oc.kwargs().get("weight_decay", 0.0) != 0.0:
            optimizer = oc(model_s.parameters())
        else:                                                       # same update rule, one fused launch over the bucket
            optimizer = FusedAdam(model_s, **{k: v for k, v in oc.kwargs().items() if k != "weight_decay"})
        scheduler = config_s.scheduler(optimizer)
        t0 = time.time()
        for epoch in range(0, config_s.epochs):
            logger.info("")
            logger.info("=" * 20 + "Training AE Epoch %d" % epoch + "=" * 20)
            trainer.train(epoch, model_s, model_c, optimizer)
            scheduler.step()
            trainer._reset_stats()
        torch.cuda.synchronize()
        logger.info("stage %d: D-VAE trained in %.1f s" % (it, time.time() - t0))
        if rank == 0:
            torch.save({"model_state_dict": model_s.state_dict()}, os.path.join(args.exp_name, "model_s{}.pth".format(it)))

        train_ds.transform = None
        data_loader = gen.getDataLoader(train_shuffle=False, train_drop_last=False)
        optimizer = config.optimizer(model.parameters())
        scheduler = config.scheduler(optimizer)
        criterion = config.criterion()
        trainer = Trainer(criterion, data_loader, logger, config, target="train_dataset", datasets_generator=gen,
                          exp_name=args.exp_name, iters=it, purify_batch=args.purify_batch, seed=args.seed)
        evaluator = Evaluator(data_loader, logger, config)
        starting_epoch, ENV = 0, new_env()
        if args.load_model:
            checkpoint = load_model(checkpoint_path_file, model, optimizer, scheduler, map_location="cuda")
            starting_epoch, ENV = checkpoint["epoch"], checkpoint["ENV"]
            trainer.global_step = ENV["global_step"]
        if args.train:
            train(starting_epoch, model, model_s, model_c, optimizer, scheduler, criterion, trainer, evaluator, ENV, data_loader,
                  config, logger, checkpoint_path_file, args.save_frequency)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    start = time.time()
    rc = main()
    print("Running Cost %.2f Days" % ((time.time() - start) / 86400))
    sys.exit(rc)
