"""The drop-in Python API on the GPU: ResNet18DVAE.forward / autograd, VAE_Trainer (both routes),
Trainer.train(epoch=0) purification, its saved layouts, and the size-independent properties of the
purification pass at the full 50 000-image size."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import dvae_oracle as O

if torch.cuda.is_available():
    import dvae_amd
    from dvae_amd import _lib as L
    from dvae_amd import data as DD
    from dvae_amd.model import ResNet18DVAE
    from dvae_amd.optim import FusedAdam
    from dvae_amd.trainer import Trainer, VAE_Trainer, purify_dataset
    from util_gpu import l2_err, rel_err

NULL_GRAD_BIASES = ("decoder1.1.bias", "decoder2.0.bias", "decoder_p1.1.bias", "decoder_p2.0.bias")


class Cfg:
    log_frequency = 2
    grad_clip = 5.0
    epochs = 60


def make_model(params=None, spatial=True, use_y=False):
    m = ResNet18DVAE(num_classes=10, stem_end_block=7, latent_dim=256)
    m.sample_wise, m.use_y, m.spatial_emb = True, use_y, spatial     # what main() sets from argv (:420-422)
    m.set_spatial_emb()
    if params is not None:
        missing, unexpected = m.load_state_dict(params, strict=False)
        assert not unexpected
    return m.cuda()


def test_forward_and_autograd_route_match_oracle():
    ocfg = O.DVAEConfig()
    p = O.recipe_params(ocfg)
    m = make_model(p)
    m.train()
    x, y, eps = O.make_inputs(ocfg, 4)
    xd, yd = x.cuda(), y.cuda()
    logits, rec, mu, log_var, pert = m(xd, yd, eps=eps.cuda())
    assert logits.shape == (4, 10) and rec.shape == (4, 3, 32, 32) and mu.shape == (4, 256, 16, 16)
    # the reference's loss block on the returned tensors (main_vae_disentangle.py:190-207), epoch 11, stage 1
    loss_rec = F.mse_loss(rec, xd)
    loss_rec_p = F.mse_loss((rec + pert).clamp(0, 1), xd)
    loss_p = F.mse_loss(pert, torch.zeros_like(pert))
    kld = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=(0, 1, 2))
    loss = loss_rec + loss_rec_p + 0.01 * max(kld, 3.0) + max(loss_p, 0.002)
    loss.backward()
    eng = m._engine
    from test_gpu_model import engine_relu_masks
    masks, _ = engine_relu_masks(eng, 4, ocfg, p, x, y, eps)
    info, grads, out = O.train_step({k: v.clone() for k, v in p.items()}, ocfg, x, y, eps, {}, 11, 1, masks=masks)
    assert abs(float(loss) - info["loss"]) <= 1e-4 * info["loss"]
    assert rel_err(rec, out["rec"]) < 5e-5 and rel_err(pert, out["perturbation"]) < 5e-5
    named = dict(m.named_parameters())
    for k, g in grads.items():
        if k in NULL_GRAD_BIASES:
            continue
        assert named[k].grad is not None and rel_err(named[k].grad, g) < 1e-4, k
    assert named["layer3.0.conv1.weight"].grad is None          # off-path parameters get no gradient
    assert int(m.bn1.num_batches_tracked) == 1
    # stock optimiser on the bound parameters: clip + Adam as the reference script does
    opt = torch.optim.Adam(m.parameters(), lr=5e-3, weight_decay=0.0)
    gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
    opt.step()
    assert abs(float(gn) - info["grad_norm"]) <= 1e-4 * info["grad_norm"]
    # eval + state_dict round trip into a fresh module gives identical outputs
    m.eval()
    m._engine.invalidate()
    with torch.no_grad():
        ref = m(xd, yd, eps=eps.cuda())
    m2 = make_model()
    m2.load_state_dict(m.state_dict())
    m2.eval()
    with torch.no_grad():
        got = m2(xd, yd, eps=eps.cuda())
    for a, b in zip(ref[1:], got[1:]):
        assert torch.equal(a, b)


def test_vae_trainer_fused_and_generic_routes():
    torch.manual_seed(0)
    ds = DD.synthetic_cifar(n=512, seed=2)
    gen = DD.DatasetGenerator(ds, train_batch_size=128, device="cuda")
    loaders = gen.getDataLoader(train_shuffle=True, train_drop_last=True)
    m = make_model()
    opt = FusedAdam(m, lr=5e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=60, eta_min=0.0)
    tr = VAE_Trainer(loaders, None, Cfg(), stage=1, kd=3.0)
    first = last = None
    for epoch in range(6):
        m.train()
        for images, labels in loaders["train_dataset"]:
            payload = tr.materialize(tr.train_batch(images, labels, m, None, opt, epoch))
            first = payload if first is None else first
            last = payload
            tr.global_step += 1
        sched.step()
    assert len(last) == 24 and set(last) == set(first)
    for k in ("psnr", "psnr_avg", "loss", "loss_rec", "loss_rec_p", "loss_p", "kld_loss", "kld_loss_avg", "lr", "|gn|",
              "loss_cls", "loss_cls_p", "loss_cls_p2", "loss_cls_purified"):
        assert k in last and np.isfinite(last[k]), k
    assert last["loss_rec"] < first["loss_rec"] and last["kld_loss"] < first["kld_loss"]
    assert abs(last["lr"] - 5e-3 * (1 + np.cos(np.pi * 5 / 60)) / 2) < 1e-9
    assert abs(float(m._engine.hyper[0]) - last["lr"]) < 1e-9 and float(m._engine.hyper[6]) == tr.global_step
    assert int(m.bn1.num_batches_tracked) == tr.global_step
    # generic route: stock Adam through the autograd node, one step
    m2 = make_model()
    opt2 = torch.optim.Adam(m2.parameters(), lr=5e-3)
    tr2 = VAE_Trainer(loaders, None, Cfg(), stage=0)
    images, labels = next(iter(loaders["train_dataset"]))
    before = m2.conv1.weight.detach().clone()
    pay = tr2.train_batch(images, labels, m2, None, opt2, 0)
    assert len(pay) == 24 and np.isfinite(pay["loss"]) and pay["|gn|"] > 0
    assert not torch.equal(before, m2.conv1.weight.detach())


def test_trainer_epoch0_purifies_in_place_and_saves_perturbations(tmp_path):
    ocfg = O.DVAEConfig()
    m = make_model(O.recipe_params(ocfg))
    ds = DD.synthetic_cifar(n=300, seed=4)
    arr = ds.data
    orig = arr.copy()
    gen = DD.DatasetGenerator(ds, train_batch_size=128, device="cuda")
    loaders = gen.getDataLoader(train_shuffle=False, train_drop_last=False)
    for it in (0, 1):
        t = Trainer(torch.nn.CrossEntropyLoss(), loaders, None, Cfg(), datasets_generator=gen, exp_name=str(tmp_path), iters=it,
                    purify_batch=128, seed=9)
        t.train(0, None, m, None, None, None)
        noise = torch.load(os.path.join(str(tmp_path), f"perturbation{it}.pt"), weights_only=True)
        assert noise.dtype == torch.float64 and tuple(noise.shape) == (300, 3, 32, 32)
        assert ds.data is arr and ds.data.dtype == np.uint8 and not np.array_equal(arr, orig)
        assert ds.transform == "crop_flip"
        if it == 0:
            # stage 0 writes trunc(255 * clamp(x - p, 0, 1)) for the p it recorded
            x = torch.from_numpy(orig).permute(0, 3, 1, 2).float() / 255
            want = np.clip(255 * (x - noise.float()).clamp(0, 1).permute(0, 2, 3, 1).numpy(), 0, 255).astype(np.uint8)
            d = np.abs(want.astype(int) - arr.astype(int))
            assert d.max() <= 1 and (d != 0).mean() < 0.005
            stage0 = arr.copy()
    assert not np.array_equal(stage0, arr)       # stage 1 purified the stage-0 output again (in place)


@pytest.mark.parametrize("stage", [0, 1])
def test_purification_is_invariant_to_batching_and_sharding_at_full_size(stage):
    """50 000 images (the CIFAR-10 train set size): the pass with 1024-image batches equals, bit for
    bit, (a) the same images purified in 8 index shards (what 8 ranks do before the all-gather) and
    (b) a sample of images recomputed in batches of 7."""
    ocfg = O.DVAEConfig()
    m = make_model(O.recipe_params(ocfg))
    m.eval()
    N = 50000
    ds = DD.synthetic_cifar(n=N, seed=0)
    data = torch.from_numpy(ds.data).cuda()
    labels = torch.from_numpy(ds.targets).cuda()
    eng = m._engine_for(torch.zeros(1024, 3, 32, 32, device="cuda"))
    eng.seed = 3
    out = torch.zeros_like(data)
    noise = torch.zeros(N, 3, 32, 32, device="cuda")
    for lo in range(0, N, 1024):
        eng.purify_batch(data, out, noise, labels, lo, min(1024, N - lo), stage)
    assert int((out != data).sum()) > 0
    from dvae_amd import dist as D
    out8 = torch.zeros_like(data)
    noise8 = torch.zeros_like(noise)
    for r in range(8):
        lo, hi = D.shard_range(N, r, 8)
        sd, sl = data[lo:hi].contiguous(), labels[lo:hi].contiguous()
        so, sn = torch.zeros_like(sd), torch.zeros(hi - lo, 3, 32, 32, device="cuda")
        for f in range(0, hi - lo, 1000):
            eng.purify_batch(sd, so, sn, sl, f, min(1000, hi - lo - f), stage, global_first=lo + f)
        out8[lo:hi], noise8[lo:hi] = so, sn
    assert torch.equal(out8, out) and torch.equal(noise8, noise)
    idx = torch.arange(0, N, 997, device="cuda")[:49]
    sd, sl = data[idx].contiguous(), labels[idx].contiguous()
    so, sn = torch.zeros_like(sd), torch.zeros(49, 3, 32, 32, device="cuda")
    for f in range(0, 49, 7):
        for k in range(7):     # global index of each image differs: one image per call
            eng.purify_batch(sd, so, sn, sl, f + k, 1, stage, global_first=int(idx[f + k]))
    assert torch.equal(so, out[idx]) and torch.equal(sn, noise[idx])
    # purify_dataset (the API entry point) agrees too and hands back float64
    ds2 = DD.ArrayDataset(ds.data[:3000].copy(), ds.targets[:3000])
    n64 = purify_dataset(m, ds2, stage, batch=1024, seed=3)
    assert n64.dtype == torch.float64 and np.array_equal(ds2.data, out[:3000].cpu().numpy())
    assert torch.equal(n64.float(), noise[:3000].cpu())
