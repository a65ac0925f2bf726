"""The CPU oracle against the fixtures generated from the reference's own model code
(tests/golden/make_golden.py).  fp32 CPU on both sides: tolerance 2e-5 relative to the tensor's scale."""
import os

import numpy as np
import pytest
import torch

from oracle import dvae_oracle as O

SUB = 37
VARIANTS = {
    "main": (dict(), 2),
    "use_y": (dict(use_y=True, spatial_emb=False), 2),
    "classwise": (dict(sample_wise=False), 2),
    "seb9": (dict(stem_end_block=9), 2),
    "img64": (dict(image_size=64, spatial_emb=False, num_classes=200), 2),
}
SMALL_KL = {"fc_mu.weight": 0.02, "fc_mu.bias": 0.02, "fc_var.weight": 0.02, "fc_var.bias": 0.02}
SMALL_P = {"decoder_p3.0.weight": 0.01, "decoder_p3.0.bias": 0.01}
SCENARIOS = {
    "open_e0_s0": (0, 0, None),
    "closed_e0_s1": (0, 1, SMALL_KL),
    "open_e11_s1": (11, 1, None),
    "pclosed_e11_s0": (11, 0, {**SMALL_KL, **SMALL_P}),
}


# Biases that feed an InstanceNorm directly: their true gradient is zero (the norm removes any
# per-channel constant), so what autograd returns is rounding noise, which Adam rescales to +-lr.
# Their trajectories are chaotic by construction and carry no information about the model function.
NULL_GRAD_BIASES = ("decoder1.1.bias", "decoder2.0.bias", "decoder_p1.1.bias", "decoder_p2.0.bias")


def close(a, b, tol=2e-5):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(1e-6, float(np.abs(b).max()))
    assert a.shape == b.shape
    assert float(np.abs(a - b).max()) <= tol * scale, (float(np.abs(a - b).max()), scale)


@pytest.mark.parametrize("name", list(VARIANTS))
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_forward_matches_reference(golden_dir, name, mode):
    g = np.load(os.path.join(golden_dir, "g1_forward.npz"))
    kw, B = VARIANTS[name]
    cfg = O.DVAEConfig(**kw)
    p = O.recipe_params(cfg)
    x, y, eps = O.make_inputs(cfg, B)
    stats = {}
    with torch.no_grad():
        out = O.forward(p, cfg, x, y, eps, mode == "train", stats)
    pre = f"{name}.{mode}"
    close(out["rec"].numpy(), g[pre + ".rec"])
    close(out["perturbation"].numpy(), g[pre + ".perturbation"])
    for nm in ("stem_out", "mu", "log_var", "z"):
        close(out[nm].reshape(-1)[::SUB].numpy(), g[f"{pre}.{nm}.sub"])
        close(float(out[nm].double().norm()), g[f"{pre}.{nm}.l2"])
    if mode == "train":
        for k in ("bn1.running_mean", "bn1.running_var", "layer2.0.shortcut.1.running_mean",
                  "layer2.0.bn2.running_var"):
            close(stats[k].numpy(), g[f"{pre}.stat.{k}"])


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_loss_and_grads_match_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "g2_loss_grads.npz"))
    epoch, stage, ov = SCENARIOS[name]
    cfg = O.DVAEConfig()
    p = O.recipe_params(cfg, ov)
    x, y, eps = O.make_inputs(cfg, 4)
    info, grads, _ = O.train_step(p, cfg, x, y, eps, {}, epoch, stage)
    for k in ("loss", "loss_rec", "loss_rec_p", "loss_p", "kld", "grad_norm"):
        close(info[k], g[f"{name}.{k}"], 1e-4)
    for k, v in grads.items():
        ref = float(g[f"{name}.gnorm.{k}"])
        assert abs(float(v.double().norm()) - ref) <= 1e-4 * max(ref, 1e-6) + 1e-9, k
    for k in ("conv1.weight", "layer2.0.shortcut.0.weight", "fc_var.bias", "decoder1.0.weight",
              "decoder_p3.0.weight", "emb_p.weight", "bn1.weight", "layer1.1.bn2.bias"):
        close(grads[k].reshape(-1)[::SUB].numpy(), g[f"{name}.gsub.{k}"], 1e-4)


def test_gate_states_are_the_intended_ones(golden_dir):
    """The four scenarios really cover both states of max(kld, target) and max(loss_p, 0.002)."""
    g = np.load(os.path.join(golden_dir, "g2_loss_grads.npz"))
    assert g["open_e0_s0.kld"] > 1.0
    assert g["closed_e0_s1.kld"] < 3.0
    assert g["open_e11_s1.kld"] > 3.0 and g["open_e11_s1.loss_p"] > 0.002
    assert g["pclosed_e11_s0.loss_p"] < 0.002 and g["pclosed_e11_s0.kld"] < 1.0
    # closed KL gate => no gradient reaches fc_var.bias through the KL term only if decoder grads vanish;
    # what must hold is that the KL contribution is absent: compare with the open scenario's magnitude.
    assert g["closed_e0_s1.gnorm.fc_var.bias"] < g["open_e0_s0.gnorm.fc_var.bias"]


def test_three_step_trajectory_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_trajectory.npz"))
    cfg = O.DVAEConfig()
    p = O.recipe_params(cfg)
    st = {}
    for step in range(3):
        x, y, eps = O.make_inputs(cfg, 4, seed=100 + step)
        info, _, _ = O.train_step(p, cfg, x, y, eps, st, 11, 1)
        for k in ("loss", "loss_rec", "loss_rec_p", "loss_p", "kld", "grad_norm"):
            close(info[k], g[f"step{step}.{k}"], 2e-4)
        for k in O.param_shapes(cfg):
            if k in NULL_GRAD_BIASES:
                continue
            ref = float(g[f"step{step}.pl2.{k}"])
            assert abs(float(p[k].double().norm()) - ref) <= 2e-4 * max(ref, 1e-6), (step, k)


@pytest.mark.parametrize("stage", [0, 1])
def test_purification_matches_reference(golden_dir, stage):
    g = np.load(os.path.join(golden_dir, "g4_purify.npz"))
    cfg = O.DVAEConfig()
    p = O.recipe_params(cfg)
    data, labels = g["data"], g["labels"]
    N = data.shape[0]
    eps_all = torch.randn(2, N, cfg.latent_dim, 16, 16, generator=torch.Generator().manual_seed(99))
    u8, noise = O.purify(p, cfg, data, labels, lambda ps, lo, hi: eps_all[ps, lo:hi], stage, batch=8)
    assert noise.dtype == np.float64 and u8.dtype == np.uint8 and u8.shape == data.shape
    close(noise, g[f"stage{stage}.noise"], 5e-5)
    diff = np.abs(u8.astype(np.int32) - g[f"stage{stage}.u8"].astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() <= 0.005


def test_quantizer_known_answers(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_quant_kat.npz"))
    x = torch.from_numpy(g["x"]).reshape(1, 1, -1).repeat(3, 1, 1)  # CHW with H=1
    q = O.quantize_u8(x)
    assert np.array_equal(q[0, :, 0], g["q"])


@pytest.mark.parametrize("stage", [0, 1])
def test_purification_64x64_matches_reference(golden_dir, stage):
    """Tiny-ImageNet shape (64x64, 200 classes, plain class embedding), N = 12 in ragged batches of 5."""
    g = np.load(os.path.join(golden_dir, "g4_purify64.npz"))
    cfg = O.DVAEConfig(image_size=64, spatial_emb=False, num_classes=200)
    p = O.recipe_params(cfg)
    data, labels = g["data"], g["labels"]
    N = data.shape[0]
    eps_all = torch.randn(2, N, cfg.latent_dim, 32, 32, generator=torch.Generator().manual_seed(199))
    u8, noise = O.purify(p, cfg, data, labels, lambda ps, lo, hi: eps_all[ps, lo:hi], stage, batch=5)
    close(noise, g[f"stage{stage}.noise"], 5e-5)
    diff = np.abs(u8.astype(np.int32) - g[f"stage{stage}.u8"].astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() <= 0.005


SUB224 = 499


def inputs_224(g):
    """Inputs of the 224x224 fixture (tests/golden/make_golden.py::gen_purify224): images and labels from the file, the
    reparameterisation noise of both passes from its seed."""
    cfg = O.DVAEConfig(image_size=224, spatial_emb=False, num_classes=100)
    data, labels = g["data"], g["labels"]
    eps_all = torch.randn(2, data.shape[0], cfg.latent_dim, 112, 112, generator=torch.Generator().manual_seed(299))
    return cfg, data, labels, eps_all


@pytest.mark.parametrize("stage", [0, 1])
def test_purification_224x224_matches_reference(golden_dir, stage):
    """ImageNet-mini shape (224x224, 100 classes) on the module the reference can run at that size (stem_end_block 7, plain
    class embedding): N = 3 in ragged batches of 2.  The snapshot's stem_end_block-15 variant does not build and stays
    unpinned."""
    g = np.load(os.path.join(golden_dir, "g4_purify224.npz"))
    cfg, data, labels, eps_all = inputs_224(g)
    p = O.recipe_params(cfg)
    u8, noise = O.purify(p, cfg, data, labels, lambda ps, lo, hi: eps_all[ps, lo:hi], stage, batch=2)
    close(noise.reshape(-1)[::SUB224], g[f"stage{stage}.noise.sub"], 5e-5)
    close(float(np.linalg.norm(noise.reshape(-1))), g[f"stage{stage}.noise.l2"], 5e-5)
    diff = np.abs(u8.astype(np.int32) - g[f"stage{stage}.u8"].astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() <= 0.005


def test_train_forward_224x224_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_purify224.npz"))
    cfg, data, labels, eps_all = inputs_224(g)
    p = O.recipe_params(cfg)
    x = torch.from_numpy(data[:2]).permute(0, 3, 1, 2).float().div(255)
    stats = {}
    with torch.no_grad():
        out = O.forward(p, cfg, x, torch.as_tensor(labels[:2], dtype=torch.long), eps_all[0, :2], True, stats)
    for nm in ("stem_out", "mu", "log_var", "z", "rec", "perturbation"):
        close(out[nm].reshape(-1)[::SUB224].numpy(), g[f"train.{nm}.sub"])
        close(float(out[nm].double().norm()), g[f"train.{nm}.l2"])
    for k in ("bn1.running_mean", "bn1.running_var", "layer2.0.shortcut.1.running_mean", "layer2.0.bn2.running_var"):
        close(stats[k].numpy(), g[f"train.stat.{k}"])


def test_data_parallel_step_matches_reference(golden_dir):
    """DataParallel arithmetic (per-replica BatchNorm statistics, whole-batch loss, summed gradients, replica 0's
    running statistics kept) against the fixture made from two parameter-sharing instances of the reference module."""
    g = np.load(os.path.join(golden_dir, "g6_dp_step.npz"))
    cfg = O.DVAEConfig()
    p = O.recipe_params(cfg)
    x, y, eps = O.make_inputs(cfg, 8, seed=4321)
    info, grads, stats = O.train_step_data_parallel(p, cfg, x, y, eps, {}, 11, 1, world=2)
    for k in ("loss", "loss_rec", "loss_rec_p", "loss_p", "kld", "grad_norm"):
        close(info[k], g[k], 1e-4)
    for k, v in grads.items():
        ref = float(g["gnorm." + k])
        assert abs(float(v.double().norm()) - ref) <= 1e-4 * max(ref, 1e-6) + 1e-9, k
    for k in O.param_shapes(cfg):
        if k in NULL_GRAD_BIASES:
            continue
        ref = float(g["pl2." + k])
        assert abs(float(p[k].double().norm()) - ref) <= 2e-4 * max(ref, 1e-6), k
    for k in ("bn1.running_mean", "layer1.0.bn2.running_var", "layer2.0.shortcut.1.running_mean"):
        close(p[k].numpy(), g["stat." + k])
        assert not np.allclose(stats[0][k].numpy(), stats[1][k].numpy())      # the replicas saw different chunks
    # the whole-batch step is a different function: BatchNorm over 8 images is not BatchNorm over 4 + 4
    p1 = O.recipe_params(cfg)
    info1, _, _ = O.train_step(p1, cfg, x, y, eps, {}, 11, 1)
    assert abs(info1["loss"] - info["loss"]) > 1e-6


def resnet18_inputs(B=8, seed=777):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, 3, 32, 32, generator=g), (torch.arange(B) * 3 + 1) % 10


def test_resnet18_step_matches_reference(golden_dir):
    """The classifier stage's restatement (ResNet-18 forward, cross-entropy, backward) against the fixture made with
    the reference's own module (tests/golden/make_golden.py::gen_resnet18_step)."""
    g = np.load(os.path.join(golden_dir, "g7_resnet18_step.npz"))
    p = O.resnet18_recipe(10)
    x, y = resnet18_inputs()
    with torch.no_grad():
        lg, _ = O.resnet18_forward(p, x, False)
    close(lg.numpy(), g["eval.logits"])
    logits, loss, grads, stats = O.resnet18_train_grads(p, x, y)
    close(logits.numpy(), g["train.logits"])
    close(loss, g["train.loss"])
    for k, v in grads.items():
        close(float(v.double().norm()), g["gnorm." + k], 1e-4)
    for k in ("conv1.weight", "layer2.0.shortcut.0.weight", "layer3.1.conv2.weight", "layer4.0.conv1.weight", "layer4.1.bn2.weight",
              "linear.weight"):
        close(grads[k].reshape(-1)[::SUB].numpy(), g["gsub." + k], 1e-4)
    for k in ("bn1.running_mean", "layer1.1.bn2.running_var", "layer3.0.shortcut.1.running_mean", "layer4.1.bn2.running_var"):
        close(stats[k].numpy(), g["stat." + k])
