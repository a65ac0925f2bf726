#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own model code.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

What runs: ``/root/reference/configs/models/resnet.py`` is imported by file path (it depends on
torch only) and its ``ResNet18DVAE`` is instantiated; parameters come from the closed-form recipe
``oracle.dvae_oracle.recipe_params`` loaded through ``load_state_dict``; the reparameterisation
noise is injected by patching ``torch.randn_like``.  ``ResNetDVAE.forward`` itself cannot run on a
CPU-only host (it calls ``.cuda()`` at resnet.py:2206), so its body is driven sub-method by
sub-method in the order of resnet.py:2208-2222.  The loss / clip / Adam sequence of
``VAE_Trainer.train_batch`` (main_vae_disentangle.py:187-211) and the purification loop of
``Trainer.train`` (:59-69) are replayed here with the very torch calls the script uses
(``F.mse_loss``, builtin ``max``, ``clip_grad_norm_``, ``torch.optim.Adam``); that script is not
importable (it parses argv and needs mlconfig/torchvision at import time).

Only data (inputs' seeds, expected outputs) is written; no reference source is stored.
"""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import dvae_oracle as O  # noqa: E402  (recipe + input seeds only)

REF = "/root/reference/configs/models/resnet.py"
SUB = 37   # stride of the flattened subsample stored for the large maps
SUB224 = 499   # ... of the 224x224 fixture's maps (49 x the pixels)


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_resnet", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def build_ref_model(ref, cfg: O.DVAEConfig, overrides=None):
    net = ref.ResNet18DVAE(num_classes=cfg.num_classes, stem_end_block=cfg.stem_end_block,
                           latent_dim=cfg.latent_dim)
    net.sample_wise, net.use_y, net.spatial_emb = cfg.sample_wise, cfg.use_y, cfg.spatial_emb
    net.set_spatial_emb()
    if cfg.spatial_emb and cfg.image_size != 32:
        raise ValueError("reference fixes the spatial embedding to 16x16")
    params = O.recipe_params(cfg, overrides)
    missing, unexpected = net.load_state_dict(params, strict=False)
    assert not unexpected, unexpected
    # aliased decoder.* keys are filled through the shared module objects
    return net, params


class inject_eps:
    def __init__(self, eps):
        self.eps = eps

    def __enter__(self):
        self.orig = torch.randn_like
        torch.randn_like = lambda t, *a, **k: self.eps.to(t.dtype).reshape(t.shape)

    def __exit__(self, *a):
        torch.randn_like = self.orig


def ref_forward(net, x, y, eps):
    """Body of ResNetDVAE.forward, resnet.py:2208-2222, without the classifier head."""
    with inject_eps(eps):
        stem_out = net.forward_stem(x)
        rec, mu, log_var, z = net.forward_ae_branch(stem_out)
    if net.spatial_emb:
        emb = net.emb_p(y).reshape(stem_out.shape)
    else:
        emb = net.emb_p(y)[:, :, None, None].repeat(1, 1, stem_out.shape[2], stem_out.shape[3])
    if net.sample_wise:
        perturbation = net.decoder_p(emb + stem_out) if net.use_y else net.decoder_p(emb + z)
    else:
        perturbation = net.decoder_p(emb)
    return stem_out, rec, mu, log_var, z, perturbation


def summarize(t, sub=SUB):
    f = t.detach().reshape(-1)
    return {"sub": f[::sub].numpy().copy(), "mean": np.float64(f.double().mean()),
            "absmax": np.float64(f.abs().max()), "l2": np.float64(f.double().norm())}


def flat(d, prefix, out):
    for k, v in d.items():
        out[prefix + "." + k] = v


VARIANTS = {
    # name: (cfg kwargs, batch)
    "main":      (dict(), 2),
    "use_y":     (dict(use_y=True, spatial_emb=False), 2),
    "classwise": (dict(sample_wise=False), 2),
    "seb9":      (dict(stem_end_block=9), 2),
    "img64":     (dict(image_size=64, spatial_emb=False, num_classes=200), 2),
}


def gen_forward(ref):
    out = {}
    for name, (kw, B) in VARIANTS.items():
        cfg = O.DVAEConfig(**kw)
        x, y, eps = O.make_inputs(cfg, B)
        for mode in ("train", "eval"):
            net, _ = build_ref_model(ref, cfg)
            net.train(mode == "train")
            with torch.no_grad():
                stem_out, rec, mu, lv, z, pert = ref_forward(net, x, y, eps)
            pre = f"{name}.{mode}"
            out[pre + ".rec"] = rec.numpy()
            out[pre + ".perturbation"] = pert.numpy()
            for nm, t in (("stem_out", stem_out), ("mu", mu), ("log_var", lv), ("z", z)):
                flat(summarize(t), pre + "." + nm, out)
            if mode == "train":
                sd = net.state_dict()
                for k in ("bn1.running_mean", "bn1.running_var",
                          "layer2.0.shortcut.1.running_mean", "layer2.0.bn2.running_var"):
                    out[pre + ".stat." + k] = sd[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g1_forward.npz"), **out)
    print("g1_forward.npz", len(out), "arrays")


# gate scenarios: (label, epoch, stage, overrides) -- overrides shrink parameters so that the
# KL term / perturbation energy fall below their thresholds.
SMALL_KL = {"fc_mu.weight": 0.02, "fc_mu.bias": 0.02, "fc_var.weight": 0.02, "fc_var.bias": 0.02}
SMALL_P = {"decoder_p3.0.weight": 0.01, "decoder_p3.0.bias": 0.01}
SCENARIOS = {
    "open_e0_s0":    (0, 0, None),
    "closed_e0_s1":  (0, 1, SMALL_KL),
    "open_e11_s1":   (11, 1, None),
    "pclosed_e11_s0": (11, 0, {**SMALL_KL, **SMALL_P}),
}


def ref_train_step(net, opt, x, y, eps, epoch, stage, kd=3.0, grad_clip=5.0):
    """main_vae_disentangle.py:181-211 (monitors on model_c left out: they do not enter ``loss``)."""
    kld_target = 1.0 if stage == 0 else kd
    net.zero_grad()
    opt.zero_grad()
    stem_out, rec, mu, log_var, z, perturbation = ref_forward(net, x, y, eps)
    images = x
    loss_rec = F.mse_loss(rec, images)
    loss_rec_p = F.mse_loss((rec + perturbation).clamp(0, 1), images)
    loss_p = F.mse_loss(perturbation, torch.zeros_like(perturbation))
    kld_weight = 0.01 if kld_target > 0.99 else 0.5
    kld_loss = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=(0, 1, 2))
    loss = loss_rec + loss_rec_p + kld_weight * max(kld_loss, kld_target)
    if epoch > 10:
        loss = loss + max(loss_p, 0.002)
    loss.backward()
    grads = {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None}
    grad_norm = torch.nn.utils.clip_grad_norm_(net.parameters(), grad_clip)
    opt.step()
    scal = {"loss": loss.item(), "loss_rec": loss_rec.item(), "loss_rec_p": loss_rec_p.item(),
            "loss_p": loss_p.item(), "kld": kld_loss.item(), "grad_norm": float(grad_norm)}
    return scal, grads, rec.detach(), perturbation.detach()


def gen_loss_grads(ref):
    out = {}
    cfg = O.DVAEConfig()
    x, y, eps = O.make_inputs(cfg, 4)
    live = [k for k in O.param_shapes(cfg) if not O.is_buffer(k)]
    for name, (epoch, stage, ov) in SCENARIOS.items():
        net, _ = build_ref_model(ref, cfg, ov)
        net.train()
        opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=0.0)
        scal, grads, rec, pert = ref_train_step(net, opt, x, y, eps, epoch, stage)
        for k, v in scal.items():
            out[f"{name}.{k}"] = np.float64(v)
        assert sorted(grads) == sorted(live), set(grads) ^ set(live)
        for k in live:
            out[f"{name}.gnorm.{k}"] = np.float64(grads[k].double().norm())
        for k in ("conv1.weight", "layer2.0.shortcut.0.weight", "fc_var.bias", "decoder1.0.weight",
                  "decoder_p3.0.weight", "emb_p.weight", "bn1.weight", "layer1.1.bn2.bias"):
            out[f"{name}.gsub.{k}"] = grads[k].reshape(-1)[::SUB].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g2_loss_grads.npz"), **out)
    print("g2_loss_grads.npz", len(out), "arrays")


def gen_trajectory(ref):
    """3 consecutive Adam steps (lr 5e-3, clip 5.0), stage 1, epoch 11; a different batch each step."""
    out = {}
    cfg = O.DVAEConfig()
    net, _ = build_ref_model(ref, cfg)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=0.0)
    live = list(O.param_shapes(cfg))
    for step in range(3):
        x, y, eps = O.make_inputs(cfg, 4, seed=100 + step)
        scal, _, _, _ = ref_train_step(net, opt, x, y, eps, 11, 1)
        for k, v in scal.items():
            out[f"step{step}.{k}"] = np.float64(v)
        sd = net.state_dict()
        for k in live:
            out[f"step{step}.psum.{k}"] = np.float64(sd[k].double().sum())
            out[f"step{step}.pl2.{k}"] = np.float64(sd[k].double().norm())
    np.savez_compressed(os.path.join(HERE, "g3_trajectory.npz"), **out)
    print("g3_trajectory.npz", len(out), "arrays")


def gen_purify(ref):
    """Trainer.train epoch 0 (main_vae_disentangle.py:59-69) over N=20 images, batch 8 (ragged tail)."""
    out = {}
    cfg = O.DVAEConfig()
    N, B = 20, 8
    rng = np.random.default_rng(7)
    data = rng.integers(0, 256, size=(N, 32, 32, 3), dtype=np.uint8)
    labels = (np.arange(N) * 7) % 10
    g = torch.Generator().manual_seed(99)
    eps_all = torch.randn(2, N, cfg.latent_dim, 16, 16, generator=g)
    out["data"] = data
    out["labels"] = labels
    for stage in (0, 1):
        net, _ = build_ref_model(ref, cfg)
        net.eval()
        res = np.empty_like(data)
        noise = np.zeros((N, 3, 32, 32))
        idx = 0
        for lo in range(0, N, B):
            hi = min(N, lo + B)
            images = torch.from_numpy(data[lo:hi]).permute(0, 3, 1, 2).float().div(255)   # ToTensor
            lab = torch.as_tensor(labels[lo:hi], dtype=torch.long)
            _, rec, mu, lv, z, perturbation = ref_forward(net, images, lab, eps_all[0, lo:hi])
            images = (images - perturbation).clamp(0, 1).detach()
            if stage > 0:
                _, rec, mu, lv, z, perturbation = ref_forward(net, images, lab, eps_all[1, lo:hi])
                images = rec.clamp(0, 1).detach()
            for k in range(images.shape[0]):
                res[idx] = np.clip(255 * images[k].cpu().permute(1, 2, 0).numpy(), a_min=0, a_max=255).astype(np.uint8)
                noise[idx] = perturbation[k].detach().cpu().numpy()
                idx += 1
        out[f"stage{stage}.u8"] = res
        out[f"stage{stage}.noise"] = noise.astype(np.float32)   # values are fp32-exact
    np.savez_compressed(os.path.join(HERE, "g4_purify.npz"), **out)
    print("g4_purify.npz", len(out), "arrays")


def gen_purify64(ref):
    """The same loop at 64x64 (Tiny-ImageNet shape: 200 classes, plain class embedding, stem_end_block 7) over N=12
    images in batches of 5 (ragged tail)."""
    out = {}
    cfg = O.DVAEConfig(image_size=64, spatial_emb=False, num_classes=200)
    N, B = 12, 5
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, size=(N, 64, 64, 3), dtype=np.uint8)
    labels = (np.arange(N) * 37) % 200
    g = torch.Generator().manual_seed(199)
    eps_all = torch.randn(2, N, cfg.latent_dim, 32, 32, generator=g)
    out["data"] = data
    out["labels"] = labels
    for stage in (0, 1):
        net, _ = build_ref_model(ref, cfg)
        net.eval()
        res = np.empty_like(data)
        noise = np.zeros((N, 3, 64, 64))
        idx = 0
        for lo in range(0, N, B):
            hi = min(N, lo + B)
            images = torch.from_numpy(data[lo:hi]).permute(0, 3, 1, 2).float().div(255)   # ToTensor
            lab = torch.as_tensor(labels[lo:hi], dtype=torch.long)
            with torch.no_grad():
                _, rec, mu, lv, z, perturbation = ref_forward(net, images, lab, eps_all[0, lo:hi])
                images = (images - perturbation).clamp(0, 1).detach()
                if stage > 0:
                    _, rec, mu, lv, z, perturbation = ref_forward(net, images, lab, eps_all[1, lo:hi])
                    images = rec.clamp(0, 1).detach()
            for k in range(images.shape[0]):
                res[idx] = np.clip(255 * images[k].cpu().permute(1, 2, 0).numpy(), a_min=0, a_max=255).astype(np.uint8)
                noise[idx] = perturbation[k].detach().cpu().numpy()
                idx += 1
        out[f"stage{stage}.u8"] = res
        out[f"stage{stage}.noise"] = noise.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "g4_purify64.npz"), **out)
    print("g4_purify64.npz", len(out), "arrays")


def gen_purify224(ref):
    """ImageNet-mini shape (224x224, 100 classes).  The snapshot's own configuration for it (stem_end_block 15,
    configs/imagenet-mini/resnet18DVAE.yaml:11) does not build a working module (channel mismatch, resnet.py:2105-2107);
    the module the reference does run at any resolution is stem_end_block 7 with the plain class embedding, and that is
    what is pinned here: the purification loop of main_vae_disentangle.py:59-69 over N=3 images in batches of 2 (ragged
    tail), both stages, plus one train-mode forward of 2 images (BatchNorm batch statistics over 100 352 pixels).  The
    uint8 results are stored whole; the float maps as a strided subsample with their norms."""
    out = {}
    cfg = O.DVAEConfig(image_size=224, spatial_emb=False, num_classes=100)
    N, B = 3, 2
    rng = np.random.default_rng(13)
    data = rng.integers(0, 256, size=(N, 224, 224, 3), dtype=np.uint8)
    labels = (np.arange(N) * 41 + 7) % 100
    g = torch.Generator().manual_seed(299)
    eps_all = torch.randn(2, N, cfg.latent_dim, 112, 112, generator=g)
    out["data"] = data
    out["labels"] = labels
    for stage in (0, 1):
        net, _ = build_ref_model(ref, cfg)
        net.eval()
        res = np.empty_like(data)
        noise = np.zeros((N, 3, 224, 224), dtype=np.float32)
        idx = 0
        for lo in range(0, N, B):
            hi = min(N, lo + B)
            images = torch.from_numpy(data[lo:hi]).permute(0, 3, 1, 2).float().div(255)   # ToTensor
            lab = torch.as_tensor(labels[lo:hi], dtype=torch.long)
            with torch.no_grad():
                _, rec, mu, lv, z, perturbation = ref_forward(net, images, lab, eps_all[0, lo:hi])
                images = (images - perturbation).clamp(0, 1).detach()
                if stage > 0:
                    _, rec, mu, lv, z, perturbation = ref_forward(net, images, lab, eps_all[1, lo:hi])
                    images = rec.clamp(0, 1).detach()
            for k in range(images.shape[0]):
                res[idx] = np.clip(255 * images[k].cpu().permute(1, 2, 0).numpy(), a_min=0, a_max=255).astype(np.uint8)
                noise[idx] = perturbation[k].detach().cpu().numpy()
                idx += 1
        out[f"stage{stage}.u8"] = res
        flat(summarize(torch.from_numpy(noise), SUB224), f"stage{stage}.noise", out)
    # one train-mode forward (batch statistics, running-statistics update)
    x = torch.from_numpy(data[:2]).permute(0, 3, 1, 2).float().div(255)
    lab = torch.as_tensor(labels[:2], dtype=torch.long)
    net, _ = build_ref_model(ref, cfg)
    net.train()
    with torch.no_grad():
        stem_out, rec, mu, lv, z, pert = ref_forward(net, x, lab, eps_all[0, :2])
    for nm, t in (("stem_out", stem_out), ("mu", mu), ("log_var", lv), ("z", z), ("rec", rec), ("perturbation", pert)):
        flat(summarize(t, SUB224), "train." + nm, out)
    sd = net.state_dict()
    for k in ("bn1.running_mean", "bn1.running_var", "layer2.0.shortcut.1.running_mean", "layer2.0.bn2.running_var"):
        out["train.stat." + k] = sd[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g4_purify224.npz"), **out)
    print("g4_purify224.npz", len(out), "arrays")


def gen_dp_step(ref):
    """One train step the way torch.nn.DataParallel over 2 devices runs it (main_vae_disentangle.py:426-428): the
    8-image batch goes through the reference module in two contiguous chunks of 4 (train-mode BatchNorm sees one chunk
    at a time), the outputs are concatenated, the loss block of :190-207 sees all 8 and autograd sums both chunks'
    parameter gradients.  nn.DataParallel itself needs two GPUs; this is its arithmetic on the reference's own
    module.  The second chunk runs through a second instance of the reference module whose Parameter objects ARE the
    first one's (what DataParallel's replicas amount to) and whose BatchNorm buffers are its own and dropped
    afterwards: the module keeps the first chunk's BatchNorm update, because replica 0 shares the module's buffers."""
    out = {}
    cfg = O.DVAEConfig()
    x, y, eps = O.make_inputs(cfg, 8, seed=4321)
    net, _ = build_ref_model(ref, cfg)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=0.0)
    net.zero_grad()
    net2, _ = build_ref_model(ref, cfg)
    net2.train()
    for m1, m2 in zip(net.modules(), net2.modules()):
        for name in m1._parameters:
            m2._parameters[name] = m1._parameters[name]
    parts = [ref_forward(n_, x[4 * r:4 * r + 4], y[4 * r:4 * r + 4], eps[4 * r:4 * r + 4]) for r, n_ in enumerate((net, net2))]
    stem_out, rec, mu, log_var, z, perturbation = (torch.cat([a[i] for a in parts], 0) for i in range(6))
    loss_rec = F.mse_loss(rec, x)
    loss_rec_p = F.mse_loss((rec + perturbation).clamp(0, 1), x)
    loss_p = F.mse_loss(perturbation, torch.zeros_like(perturbation))
    kld_loss = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=(0, 1, 2))
    loss = loss_rec + loss_rec_p + 0.01 * max(kld_loss, 3.0) + max(loss_p, 0.002)       # stage 1, epoch 11
    loss.backward()
    grads = {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None}
    grad_norm = torch.nn.utils.clip_grad_norm_(net.parameters(), 5.0)
    opt.step()
    for k, v in (("loss", loss), ("loss_rec", loss_rec), ("loss_rec_p", loss_rec_p), ("loss_p", loss_p), ("kld", kld_loss)):
        out[k] = np.float64(v.item())
    out["grad_norm"] = np.float64(float(grad_norm))
    for k, g_ in grads.items():
        out["gnorm." + k] = np.float64(g_.double().norm())
    sd = net.state_dict()
    for k in O.param_shapes(cfg):
        out["pl2." + k] = np.float64(sd[k].double().norm())
    for k in ("bn1.running_mean", "layer1.0.bn2.running_var", "layer2.0.shortcut.1.running_mean"):
        out["stat." + k] = sd[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g6_dp_step.npz"), **out)
    print("g6_dp_step.npz", len(out), "arrays")


def gen_quant_kat():
    """Known answers of np.clip(255*x,0,255).astype(uint8) at the fp32 neighbours of k/255."""
    xs = []
    for k in range(256):
        c = np.float32(k) / np.float32(255)
        xs += [np.nextafter(c, np.float32(-1)), c, np.nextafter(c, np.float32(2))]
    xs += [np.float32(0), np.float32(1), np.float32(0.999999), np.float32(0.5), np.float32(1e-8)]
    xs = np.array(xs, dtype=np.float32).clip(0, 1)
    q = np.clip(255 * xs, a_min=0, a_max=255).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "g5_quant_kat.npz"), x=xs, q=q)
    print("g5_quant_kat.npz", xs.size)


def gen_state_dict_keys(ref):
    """Names and shapes of ResNet18DVAE.state_dict() (161 entries, aliased decoder.* included)."""
    import json
    out = {}
    for tag, seb, spatial in (("seb7_spatial", 7, True), ("seb9_flat", 9, False)):
        net = ref.ResNet18DVAE(num_classes=10, stem_end_block=seb, latent_dim=256)
        net.spatial_emb = spatial
        net.set_spatial_emb()
        out[tag] = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(out, f)
    print("state_dict_keys.json", {k: len(v) for k, v in out.items()})


def gen_resnet18_keys(ref):
    """Names and shapes of the CIFAR ResNet18 classifier's state_dict (resnet.py:62-122), the layout of the
    ``model_state_dict`` entry of the classifier checkpoints (util.py:85-99)."""
    import json
    net = ref.ResNet18(num_classes=10)
    with open(os.path.join(HERE, "resnet18_state_dict_keys.json"), "w") as f:
        json.dump([[k, list(v.shape)] for k, v in net.state_dict().items()], f)
    print("resnet18_state_dict_keys.json", len(net.state_dict()))


def resnet18_inputs(B=8, seed=777):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, 3, 32, 32, generator=g), (torch.arange(B) * 3 + 1) % 10


def gen_resnet18_step(ref):
    """The classifier stage's arithmetic on the reference's own module (resnet.py:62-122): ResNet18(10) filled from the
    closed-form recipe, one train-mode forward of 8 images + CrossEntropyLoss + backward as Trainer.train_batch runs them
    (main_vae_disentangle.py:100-107; the clip and the SGD step are torch's own and not restated), and an eval-mode
    forward with the recipe's running statistics.  Stored: logits, loss, every gradient's L2 norm, four gradients as
    strided subsamples, four BatchNorm running statistics after the train forward."""
    out = {}
    x, y = resnet18_inputs()
    params = O.resnet18_recipe(10)
    net = ref.ResNet18(num_classes=10)
    missing, unexpected = net.load_state_dict(params, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    net.eval()
    with torch.no_grad():
        out["eval.logits"] = net(x).numpy()
    net.train()
    logits = net(x)
    loss = F.cross_entropy(logits, y)
    loss.backward()
    out["train.logits"] = logits.detach().numpy()
    out["train.loss"] = np.float64(loss.item())
    for k, v in net.named_parameters():
        out["gnorm." + k] = np.float64(v.grad.double().norm())
    for k in ("conv1.weight", "layer2.0.shortcut.0.weight", "layer3.1.conv2.weight", "layer4.0.conv1.weight", "layer4.1.bn2.weight",
              "linear.weight"):
        out["gsub." + k] = dict(net.named_parameters())[k].grad.reshape(-1)[::SUB].numpy().copy()
    sd = net.state_dict()
    for k in ("bn1.running_mean", "layer1.1.bn2.running_var", "layer3.0.shortcut.1.running_mean", "layer4.1.bn2.running_var"):
        out["stat." + k] = sd[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g7_resnet18_step.npz"), **out)
    print("g7_resnet18_step.npz", len(out), "arrays")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = load_reference()
    gens = {"forward": gen_forward, "loss_grads": gen_loss_grads, "trajectory": gen_trajectory, "purify": gen_purify,
            "purify64": gen_purify64, "purify224": gen_purify224, "dp_step": gen_dp_step, "quant_kat": lambda r: gen_quant_kat(),
            "state_dict_keys": gen_state_dict_keys, "resnet18_keys": gen_resnet18_keys, "resnet18_step": gen_resnet18_step}
    for name in (sys.argv[1:] or list(gens)):       # no arguments: regenerate everything
        gens[name](ref)
