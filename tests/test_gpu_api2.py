"""Drop-in API on the GPU, second part: bf16 driven through the module / trainers, the model_c monitors with a real
frozen classifier, parameters changed behind the engine's back, engine rebuilds in the middle of a run, the poison
ingress arithmetic, the classifier stage (epochs >= 1), the two-stage runner and the downstream-accuracy check of the
purified data.

bf16 tolerances against the fp32 path of the same API (bf16 activations and weights, fp32 accumulation / statistics /
losses): reconstruction 3e-2, mu / log_var / perturbation 6e-2 in relative L2; loss scalars 3 %; purified uint8 mean
|diff| < 0.75 LSB with > 98.5 % of pixels within 4 LSB."""
import json
import os
import sys

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
    from dvae_amd.classifier import Evaluator, ResNet18, load_model, new_env, save_model, train
    from dvae_amd.model import ResNet18DVAE
    from dvae_amd.optim import FusedAdam
    from dvae_amd.trainer import Trainer, VAE_Trainer, purify_dataset
    from util_gpu import l2_err, rel_err


class Cfg:
    log_frequency = 2
    grad_clip = 5.0
    epochs = 60
    num_classes = 10


def make_model(params=None, dtype="f32", spatial=True, use_y=False):
    m = ResNet18DVAE(num_classes=10, stem_end_block=7, latent_dim=256, act_dtype=dtype)
    m.sample_wise, m.use_y, m.spatial_emb = True, use_y, spatial
    m.set_spatial_emb()
    if params is not None:
        missing, unexpected = m.load_state_dict(params, strict=False)
        assert not unexpected
    return m.cuda()


# ------------------------------------------------------------------------------------------------
# bf16 through the API
# ------------------------------------------------------------------------------------------------
def test_bf16_forward_through_module_matches_f32_module():
    ocfg = O.DVAEConfig()
    p = O.recipe_params(ocfg)
    x, y, eps = O.make_inputs(ocfg, 6)
    xd, yd = x.cuda(), y.cuda()
    outs = {}
    for dt in ("f32", "bf16"):
        m = make_model(p, dt)
        assert m.act_dtype == (L.F32 if dt == "f32" else L.BF16)
        m.eval()
        with torch.no_grad():
            injected = m(xd, yd, eps=eps.cuda())
            # default route (noise drawn in-kernel): mu / log_var do not depend on the noise and must be the real ones,
            # not a stale or never-written buffer
            m._engine.act("mulv", 16, 512, f32=True).fill_(123.0)
            if dt == "bf16":
                m._engine.act("mulv.act", 16, 512).fill_(123.0)
            drawn = m(xd, yd)
        outs[dt] = (injected, drawn)
        assert m._engine.dtype == m.act_dtype
        for a, b in zip(injected[2:4], drawn[2:4]):           # mu, log_var: same with or without injected noise
            assert l2_err(b, a) < (1e-6 if dt == "f32" else 1e-2), dt
    f, b = outs["f32"][0], outs["bf16"][0]
    assert l2_err(b[1], f[1]) < 3e-2
    for i in (2, 3, 4):
        assert l2_err(b[i], f[i]) < 6e-2, i
    assert float(outs["bf16"][1][2].abs().max()) < 100.0      # never the 123.0 fill


def test_bf16_trainers_through_api_match_f32(tmp_path):
    """VAE_Trainer (fused route) and Trainer.train(0) with act_dtype="bf16" against the same calls in fp32.  The
    purification passes start from the same (fp32-trained) parameters: after a few Adam steps two training runs differ by
    more than the arithmetic does."""
    ocfg = O.DVAEConfig()
    p = O.recipe_params(ocfg)
    ds0 = DD.synthetic_cifar(n=512, seed=5)
    res, trained = {}, None
    for dt in ("f32", "bf16"):
        ds = DD.ArrayDataset(ds0.data.copy(), ds0.targets.copy())
        gen = DD.DatasetGenerator(ds, train_batch_size=128, device="cuda", seed=3)
        loaders = gen.getDataLoader(train_shuffle=True, train_drop_last=True)
        m = make_model(p)                       # built fp32; the trainers switch it
        opt = FusedAdam(m, lr=5e-3)
        tr = VAE_Trainer(loaders, None, Cfg(), stage=1, kd=3.0, act_dtype=dt)
        m.train()
        pays = []
        for images, labels in loaders["train_dataset"]:
            pays.append(tr.materialize(tr.train_batch(images, labels, m, None, opt, 11)))
            tr.global_step += 1
        assert m._engine.dtype == (L.F32 if dt == "f32" else L.BF16)
        if trained is None:
            trained = {k: v.detach().clone() for k, v in m.state_dict().items()}
        m2 = make_model(trained)
        # bf16 engines draw their purification noise from the 16-bit-uniform Philox stream; the fp32 engine is built here,
        # at the pass's batch size, and switched to that stream so that both legs see the same eps
        m2.act_dtype = dt
        m2._engine_for(torch.empty(256, 3, 32, 32, device="cuda")).inference_noise16 = True
        t = Trainer(torch.nn.CrossEntropyLoss(), loaders, None, Cfg(), datasets_generator=gen, exp_name=str(tmp_path / dt),
                    iters=1, purify_batch=256, seed=9, act_dtype=dt)
        t.train(0, None, m2, None, None, None)
        assert m2._engine.dtype == (L.F32 if dt == "f32" else L.BF16)
        noise = torch.load(os.path.join(str(tmp_path / dt), "perturbation1.pt"), weights_only=True)
        res[dt] = (pays, ds.data.copy(), noise, m.state_dict()["conv1.weight"].clone())
    (pf, uf, nf, wf), (pb, ub, nb, wb) = res["f32"], res["bf16"]
    for i, (a, b) in enumerate(zip(pf, pb)):
        for k in ("loss", "loss_rec", "loss_rec_p", "loss_p", "kld_loss", "|gn|", "psnr"):
            # the first step starts from identical parameters; later ones inherit the divergence of Adam's sign-like updates
            tol = 3e-2 if i == 0 else 0.15
            assert abs(a[k] - b[k]) <= tol * abs(a[k]) + 1e-6, (i, k, a[k], b[k])
        assert a["psnr_avg"] > 0 and b["psnr_avg"] > 0          # fed on the fused route too
    d = np.abs(uf.astype(np.int32) - ub.astype(np.int32))
    assert d.mean() < 0.75 and (d <= 4).mean() > 0.985, (d.mean(), (d <= 4).mean(), d.max())
    assert l2_err(nb, nf) < 6e-2


# ------------------------------------------------------------------------------------------------
# monitors with a real model_c
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("route", ["fused", "generic"])
def test_monitor_losses_with_real_model_c(route):
    torch.manual_seed(3)
    model_c = ResNet18(10).cuda()
    # a few training-mode forwards so that BatchNorm running statistics are not the trivial (0, 1)
    with torch.no_grad():
        for _ in range(2):
            model_c(torch.rand(64, 3, 32, 32, device="cuda"))
    model_c.eval()
    for q in model_c.parameters():
        q.requires_grad_(False)
    model_c.cam_register(layer="layer2")
    ds = DD.synthetic_cifar(n=256, seed=8)
    gen = DD.DatasetGenerator(ds, train_batch_size=64, device="cuda")
    loaders = gen.getDataLoader(train_shuffle=True, train_drop_last=True)
    m = make_model(O.recipe_params(O.DVAEConfig()))
    m.train()
    opt = FusedAdam(m, lr=5e-3) if route == "fused" else torch.optim.Adam(m.parameters(), lr=5e-3)
    tr = VAE_Trainer(loaders, None, Cfg(), stage=1, kd=3.0, monitor_every=1)
    before = [q.detach().clone() for q in model_c.parameters()]
    for images, labels in loaders["train_dataset"]:
        pay = tr.materialize(tr.train_batch(images, labels, m, model_c, opt, 0))
        assert len(pay) == 24 and all(np.isfinite(v) for v in pay.values()), pay
        # stock-torch recomputation the way the reference does it: four separate forwards (main_vae_disentangle.py:194-198)
        eng = m._engine
        b = images.shape[0]
        if route == "fused":
            rec = eng.to_nchw(eng.buf("rec", (eng.B, 32, 32, 4), torch.float32), b, 3, 32)
            pert = eng.to_nchw(eng.act("decoder_p.d6", 32, 3, f32=True), b, 3, 32)
            with torch.no_grad():
                want = {
                    "loss_cls": F.cross_entropy(model_c(rec), labels).item(),
                    "loss_cls_p": F.cross_entropy(model_c((rec + pert).clamp(0, 1)), labels).item(),
                    "loss_cls_p2": F.cross_entropy(model_c((images - pert).clamp(0, 1)), labels).item(),
                    "loss_cls_purified": torch.nn.KLDivLoss(reduction="batchmean")(
                        F.log_softmax(model_c((images - pert).clamp(0, 1)), dim=1), torch.ones(b, 10, device="cuda") / 10).item(),
                }
            for k, v in want.items():
                assert abs(pay[k] - v) <= 1e-4 * abs(v) + 1e-6, (k, pay[k], v)
                assert pay[k + "_avg"] > 0
        else:
            assert pay["loss_cls"] > 0 and pay["loss_cls_purified"] >= 0
        tr.global_step += 1
    for a, q in zip(before, model_c.parameters()):
        assert torch.equal(a, q)                                  # the monitor classifier stays frozen


# ------------------------------------------------------------------------------------------------
# parameters changed behind the engine's back; engine rebuilt in the middle of a run
# ------------------------------------------------------------------------------------------------
def test_load_state_dict_and_stock_optimizer_after_forward_are_noticed():
    ocfg = O.DVAEConfig()
    p1 = O.recipe_params(ocfg)
    p2 = {k: (v * 0.5 if v.dtype.is_floating_point and "running_var" not in k else v) for k, v in p1.items()}
    x, y, eps = O.make_inputs(ocfg, 4)
    xd, yd, ed = x.cuda(), y.cuda(), eps.cuda()
    m = make_model(p1)
    m.eval()
    with torch.no_grad():
        a = m(xd, yd, eps=ed)
        m.load_state_dict(p2, strict=False)            # after the first forward: packed weights / folded BN are stale
        b = m(xd, yd, eps=ed)
        fresh = make_model(p2)
        fresh.eval()
        c = fresh(xd, yd, eps=ed)
    assert not torch.equal(a[1], b[1])
    for u, v in zip(b[1:], c[1:]):
        assert torch.equal(u, v)
    # a stock optimiser stepping the bound parameters, no explicit invalidate
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    out = m(xd, yd, eps=ed)
    (out[1].square().mean() + out[4].square().mean()).backward()
    opt.step()
    m.eval()
    with torch.no_grad():
        d = m(xd, yd, eps=ed)
        fresh = make_model(m.state_dict())
        fresh.eval()
        e = fresh(xd, yd, eps=ed)
    assert not torch.equal(d[1], b[1])
    for u, v in zip(d[1:], e[1:]):
        assert torch.equal(u, v)
    # an in-place edit of a BatchNorm buffer is noticed as well
    with torch.no_grad():
        m.bn1.running_mean.add_(0.05)
        f = m(xd, yd, eps=ed)
    assert not torch.equal(f[1], d[1])


def test_engine_rebuild_mid_run_keeps_optimizer_state():
    """Train 2 steps at batch 16, purify at batch 64 (forces a larger engine), train 1 more step: identical to 3
    uninterrupted steps (Adam moments, step count and hyper-parameters survive the rebuild)."""
    ocfg = O.DVAEConfig()
    p = O.recipe_params(ocfg)
    ds0 = DD.synthetic_cifar(n=64, seed=12)
    runs = []
    for interrupt in (False, True):
        ds = DD.ArrayDataset(ds0.data.copy(), ds0.targets.copy())
        gen = DD.DatasetGenerator(ds, train_batch_size=16, device="cuda", seed=1)
        loaders = gen.getDataLoader(train_shuffle=False, train_drop_last=True)
        m = make_model(p)
        m.train()
        opt = FusedAdam(m, lr=5e-3)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=60, eta_min=0.0)
        sched.step()
        tr = VAE_Trainer(loaders, None, Cfg(), stage=1, kd=3.0)
        batches = list(loaders["train_dataset"])[:3]
        for i, (images, labels) in enumerate(batches):
            if interrupt and i == 2:
                eng_before = m._engine
                scratch = DD.ArrayDataset(ds0.data.copy(), ds0.targets.copy())
                purify_dataset(m, scratch, 0, batch=64, seed=0)
                assert m._engine is not eng_before and m._engine.B == 64
                m.train()
            tr.train_batch(images, labels, m, None, opt, 11)
            tr.global_step += 1
        runs.append((m._engine.P.clone(), m._engine.M.clone(), m._engine.V.clone(), m._engine.hyper.clone(),
                     m._engine.R.clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    assert float(runs[1][3][6]) == 3.0


# ------------------------------------------------------------------------------------------------
# poison ingress
# ------------------------------------------------------------------------------------------------
def test_poison_ingress_is_bit_exact_against_numpy_restatement():
    rng = np.random.default_rng(0)
    n = 300
    data = rng.integers(0, 256, (n, 32, 32, 3), dtype=np.uint8)
    data[:4] = np.array([0, 1, 254, 255], dtype=np.uint8)[:, None, None, None]
    targets = np.arange(n) % 10
    # sample-wise noise: the usual +-8/255 range, plus boundary cases: 255*p at integers, at .5, just around them,
    # saturating past +-1, denormal-small
    pert = torch.from_numpy(rng.uniform(-8 / 255, 8 / 255, (n, 3, 32, 32)).astype(np.float32))
    ks = np.arange(-300, 301, dtype=np.float32)
    edge = np.concatenate([ks / np.float32(255), np.nextafter(ks / np.float32(255), np.float32(9)),
                           np.nextafter(ks / np.float32(255), np.float32(-9)), (ks + np.float32(0.5)) / np.float32(255),
                           np.array([1e-40, -1e-40, 2.0, -2.0, 0.0], dtype=np.float32)])
    flat = pert[:4].reshape(4, -1)
    flat[:, :edge.size] = torch.from_numpy(edge)
    ds = DD.ArrayDataset(data.copy(), targets)
    idx = DD.poison_dataset(ds, pert, "samplewise")
    assert idx == list(range(n)) and ds.version == 1
    assert np.array_equal(ds.data, O.poison_u8(data, pert))
    # class-wise, poison_rate 0.5: only the drawn samples change, by their class's pattern
    pc = DD.synthetic_classwise_poison(10, 32, 8 / 255, seed=2)
    ds2 = DD.ArrayDataset(data.copy(), targets)
    idx2 = DD.poison_dataset(ds2, pc, "classwise", poison_rate=0.5, seed=4)
    assert len(idx2) == n // 2 and idx2 == sorted(idx2)
    assert np.array_equal(ds2.data, O.poison_u8(data, pc, targets, idx2))
    untouched = np.setdiff1d(np.arange(n), idx2)
    assert np.array_equal(ds2.data[untouched], data[untouched]) and not np.array_equal(ds2.data[idx2], data[idx2])
    with pytest.raises(ValueError):
        DD.poison_dataset(ds, pert[:10], "samplewise")
    # float64 files (what main_vae_disentangle.py:91-92 writes): 255 * p and the clamp in double, one rounding
    p64 = pert.double() * (1 - 2.0 ** -26)
    ds3 = DD.ArrayDataset(data.copy(), targets)
    DD.poison_dataset(ds3, p64, "samplewise")
    assert np.array_equal(ds3.data, O.poison_u8(data, p64))


def test_poison_ingress_matches_hand_derived_reference_answers():
    """The device ingress against the known answers of tests/poison_kat.py, which follow dataset.py:1235, :534-550,
    :1287-1289 by hand (the reference's dataset module cannot be imported here: torchvision is absent), for float32
    and float64 perturbation files: bit-exact."""
    import poison_kat
    for dt, (x, p, want) in poison_kat.as_arrays().items():
        ds = DD.ArrayDataset(x.copy(), np.zeros(1, dtype=np.int64))
        DD.poison_dataset(ds, p, "samplewise")
        assert np.array_equal(ds.data, want), dt
        dsc = DD.ArrayDataset(x.copy(), np.zeros(1, dtype=np.int64))
        DD.poison_dataset(dsc, p, "classwise")                   # one class: the same noise by label
        assert np.array_equal(dsc.data, want), dt


# ------------------------------------------------------------------------------------------------
# classifier stage and checkpoints
# ------------------------------------------------------------------------------------------------
def test_classifier_stage_trains_evaluates_and_checkpoints(tmp_path):
    torch.manual_seed(0)
    train_ds = DD.synthetic_separable(n=1024, seed=1)
    test_ds = DD.synthetic_separable(n=512, seed=2)
    gen = DD.DatasetGenerator(train_ds, test_ds, train_batch_size=128, eval_batch_size=256, device="cuda")
    cfg = dvae_amd.config.from_dict(dict(dvae_amd.config.CIFAR10_RESNET18, epochs=3, log_frequency=4))
    model = cfg.model().cuda()
    train_ds.transform = "crop_flip"
    loaders = gen.getDataLoader()
    optimizer = cfg.optimizer(model.parameters())
    scheduler = cfg.scheduler(optimizer)
    criterion = cfg.criterion()
    trainer = Trainer(criterion, loaders, None, cfg, datasets_generator=gen, exp_name=str(tmp_path), iters=1)
    evaluator = Evaluator(loaders, None, cfg)
    ENV = new_env()
    ck = str(tmp_path / "checkpoints" / "resnet18")
    train(1, model, None, None, optimizer, scheduler, criterion, trainer, evaluator, ENV, loaders, cfg, None, ck)
    assert len(ENV["eval_history"]) == 3 and ENV["global_step"] == 3 * 8
    # 10 classes: chance is 10 %.  Three short epochs of stock-torch SGD do not repeat to the digit from run to run
    # (atomics in the library convolutions' weight gradients): the bound only says "it learned".
    assert max(ENV["eval_history"]) > 40.0, ENV["eval_history"]
    cm = np.array(ENV["cm_history"][-1])
    assert cm.shape == (10, 10) and cm.sum() == 512 and abs(np.trace(cm) / 512 * 100 - ENV["eval_history"][-1]) < 1e-3
    state = torch.load(ck + ".pth", weights_only=True)
    assert set(state) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "ENV"}
    assert state["epoch"] == 4 and state["ENV"]["global_step"] == 24
    golden = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "resnet18_state_dict_keys.json")))
    assert [[k, list(v.shape)] for k, v in state["model_state_dict"].items()] == golden
    model2 = cfg.model().cuda()
    opt2 = cfg.optimizer(model2.parameters())
    sch2 = cfg.scheduler(opt2)
    got = load_model(ck, model2, opt2, sch2, map_location="cuda")
    assert got["epoch"] == 4 and sch2.last_epoch == scheduler.last_epoch
    evaluator.eval(3, model2)
    assert abs(evaluator.acc_meters.avg * 100 - ENV["eval_history"][-1]) < 1e-3


def test_two_stage_runner_tiny(tmp_path, monkeypatch):
    """run_disentangle.main on a 256-image synthetic poisoned set: both stages leave files with the reference's layouts,
    and stage 1's D-VAE trains on exactly the array stage 0's purification wrote."""
    import run_disentangle as R
    pert_path = str(tmp_path / "perturbation.pt")
    torch.save(DD.synthetic_classwise_poison(10, 32, 8 / 255, seed=1), pert_path)
    exp = str(tmp_path / "exp_a_b_c_VAE")
    seen = []
    orig = VAE_Trainer.train

    def spy(self, epoch, model, model_c, optimizer):
        if epoch == 0:
            seen.append(self.data_loader[self.target].dataset.data.copy())
        return orig(self, epoch, model, model_c, optimizer)

    monkeypatch.setattr(VAE_Trainer, "train", spy)
    monkeypatch.chdir(tmp_path)
    rc = R.main(["--exp_name", exp, "--version", "resnet18", "--version_s", "resnet18DVAE", "--config_path", "/nonexistent",
                 "--train_data_type", "PoisonSynthetic", "--test_data_type", "Synthetic", "--synthetic_n", "256",
                 "--synthetic_test_n", "128", "--perturb_tensor_filepath", pert_path, "--perturb_type", "classwise",
                 "--kd", "3.0", "--use_y", "--spatial_emb", "--train", "--epochs_s", "2", "--epochs", "1",
                 "--train_batch_size", "64", "--purify_batch", "100", "--dtype", "f32", "--seed", "5"])
    assert rc == 0 and len(seen) == 2
    keys = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_keys.json")))["seb7_spatial"]
    for it in (0, 1):
        sd = torch.load(os.path.join(exp, f"model_s{it}.pth"), weights_only=True)
        assert list(sd) == ["model_state_dict"]
        assert [[k, list(v.shape)] for k, v in sd["model_state_dict"].items()] == keys
        noise = torch.load(os.path.join(exp, f"perturbation{it}.pt"), weights_only=True)
        assert noise.dtype == torch.float64 and tuple(noise.shape) == (256, 3, 32, 32)
    ck = torch.load(os.path.join(exp, "resnet18", "checkpoints", "resnet18.pth"), weights_only=True)
    assert ck["epoch"] == 2 and len(ck["ENV"]["eval_history"]) == 2
    assert os.path.exists(os.path.join(exp, "resnet18", "poison_targets.npy"))
    # stage 1 saw stage 0's output: re-purify the array stage 0 trained on with the saved stage-0 model
    m0 = make_model(torch.load(os.path.join(exp, "model_s0.pth"), weights_only=True)["model_state_dict"])
    redo = DD.ArrayDataset(seen[0].copy(), np.arange(256) % 10)
    n0 = purify_dataset(m0, redo, 0, batch=100, seed=5)
    assert np.array_equal(redo.data, seen[1]) and not np.array_equal(seen[0], seen[1])
    assert torch.equal(n0, torch.load(os.path.join(exp, "perturbation0.pt"), weights_only=True))


# The stand-in for CIFAR of the third correctness leg (no dataset can be fetched here), tuned until it can fail: class
# templates pulled 40 % towards their mean under a class-independent clutter field (a ResNet-18 trained 20 epochs on the
# 4096 clean images reaches about 85 % on clean test data, not 99), and a class-wise block pattern of +-16/255 as the
# unlearnable-example shortcut (trained on the poisoned set the same classifier loses close to 30 points).
PROXY = dict(separation=0.6, clutter=0.4, noise=0.1)
POISON_EPS = 16 / 255
CLASSIFIER_EPOCHS = 20


def _classifier_accuracy(arr, targets, test, act_dtype="f32", seeds=(1,)):
    return float(np.mean([_classifier_accuracy_once(arr, targets, test, act_dtype, s) for s in seeds]))


def _classifier_accuracy_once(arr, targets, test, act_dtype, seed):
    """Clean test accuracy of a ResNet-18 trained on (arr, targets) the way the classifier stage does it
    (main_vae_disentangle.py:323-371: SGD 0.1 / 0.9 / 5e-4, cosine, RandomCrop(32, 4) + flip), same initialisation,
    augmentation draws and batch order for every arm (per seed)."""
    cfg = dvae_amd.config.from_dict(dict(dvae_amd.config.CIFAR10_RESNET18, epochs=CLASSIFIER_EPOCHS, log_frequency=100000))
    torch.manual_seed(seed)
    tds = DD.ArrayDataset(arr.copy(), targets.copy(), transform="crop_flip")
    g2 = DD.DatasetGenerator(tds, test, train_batch_size=128, eval_batch_size=512, device="cuda", seed=6 + seed)
    ld = g2.getDataLoader()
    model = cfg.model(act_dtype=act_dtype).cuda()
    optimizer = cfg.optimizer(model.parameters())
    scheduler = cfg.scheduler(optimizer)
    trainer = Trainer(cfg.criterion(), ld, None, cfg, datasets_generator=g2, iters=1)
    ENV = new_env()

    class _LateEvaluator(Evaluator):
        """Only the last three epochs' evaluations are read below: the earlier ones are skipped (their history entries stay 0)."""

        def eval(self, epoch, model):
            if epoch > CLASSIFIER_EPOCHS - 3:
                super().eval(epoch, model)

    train(1, model, None, None, optimizer, scheduler, cfg.criterion(), trainer, _LateEvaluator(ld, None, cfg), ENV, ld, cfg)
    print(".", end="", file=sys.__stderr__, flush=True)      # a sign of life past pytest's capture: this test runs for minutes
    # mean over the last three epochs (small learning rates at the end of the cosine schedule): the +-1 point band is about
    # three standard deviations of one evaluation's sampling noise on 16 384 test images
    return float(np.mean(ENV["eval_history"][-3:]))


def _train_dvae(ds, test, dtype, epochs=30, seed=2):
    """A D-VAE trained briefly on the poisoned set through the drop-in trainer (fused route), stage-1 constants.
    ``seed``: the loader's shuffling (another order of the batches is another trained network)."""
    gen = DD.DatasetGenerator(ds, test, train_batch_size=128, eval_batch_size=512, device="cuda", seed=seed)
    loaders = gen.getDataLoader(train_shuffle=True, train_drop_last=True)
    torch.manual_seed(0)
    m = make_model(dtype=dtype)
    m.train()
    opt = FusedAdam(m, lr=5e-3)
    tr = VAE_Trainer(loaders, None, Cfg(), stage=1, kd=3.0)
    for epoch in range(epochs):                  # 960 steps
        for images, labels in loaders["train_dataset"]:
            tr.train_batch(images, labels, m, None, opt, epoch)
            tr.global_step += 1
    return m


def _purify_hip(sd, ds, eps, dt):
    md = make_model(sd, dt)
    md.eval()
    N = len(ds)
    eng = md._engine_for(torch.zeros(128, 3, 32, 32, device="cuda"))
    d = torch.from_numpy(ds.data).cuda()
    out = torch.zeros_like(d)
    lab = torch.from_numpy(ds.targets).cuda()
    for lo in range(0, N, 128):
        eng.purify_batch(d, out, None, lab, lo, 128, 1, eps_nchw=(eps[0, lo:lo + 128].cuda(), eps[1, lo:lo + 128].cuda()))
    return out.cpu().numpy()


@torch.no_grad()
def _purify_with_sign_error(sd, ds, eps):
    """Stage-1 purification with ONE defect: the perturbation is added instead of subtracted
    (main_vae_disentangle.py:62 with the sign flipped), everything else as the path does it."""
    md = make_model(sd, "f32")
    md.eval()
    out = np.empty_like(ds.data)
    for lo in range(0, len(ds), 128):
        x = torch.from_numpy(ds.data[lo:lo + 128]).cuda().permute(0, 3, 1, 2).float().div(255)
        y = torch.from_numpy(ds.targets[lo:lo + 128]).cuda()
        _, _, _, _, pert = md(x, y, eps[0, lo:lo + 128].cuda())
        x1 = (x + pert).clamp(0, 1)
        _, rec, _, _, _ = md(x1, y, eps[1, lo:lo + 128].cuda())
        out[lo:lo + 128] = (255 * rec.clamp(0, 1)).permute(0, 2, 3, 1).cpu().numpy().astype(np.uint8)
    return out


def test_downstream_accuracy_of_purified_data_oracle_vs_hip():
    """The third correctness leg, as far as it can be checked without CIFAR.  A D-VAE is trained briefly on a class-wise
    poisoned stand-in set and purifies it (stage 1: rec(x - p)); a ResNet-18 is retrained on every version of the set and
    evaluated on clean test data.  Arms, all with the same classifier initialisation and batch order:

      clean        the unpoisoned set (what there is to recover)
      poisoned     the poisoned set, unpurified: the control -- must lose >= 20 points against clean, else the poison is
                   no shortcut and the leg shows nothing
      oracle       purified by the CPU oracle with the fp32-trained D-VAE's parameters and injected noise
      f32, bf16    the same parameters and noise through the fp32 / bf16 HIP purification: each within +-1 point of oracle
                   (+ two standard errors of the difference of two three-run means, from the variance pooled over the arms)
      f32-trained, bf16-trained
                   D-VAEs trained AND run in fp32 / in bf16 (what bench.py times).  Every training run is another network,
                   and a 960-step D-VAE is a noisy purifier: with nothing changed but the order of the batches the clean
                   accuracy behind it moves between 59 and 71 % in either dtype, and now and then a run does not learn to
                   separate the poison at all and lands at the poisoned control (52 % seen, in both dtypes on the same
                   batch order); a summation-order change in one conv kernel re-rolls all of it just the same.  So the claim
                   that matters -- training in bf16 purifies as well as training in fp32 -- is held on the median over
                   three trainings each (one collapsed run does not move it): bf16 >= f32 - 6 points
      sign-error   the fp32 path with the perturbation added instead of subtracted: a purifier with one defect must land
                   outside the +-1 band (below oracle by more than 3 points), else the leg could not see it
    and the purified arms must recover >= 8 points over the poisoned control."""
    torch.manual_seed(0)
    N, NT = 4096, 16384
    clean = DD.synthetic_separable(n=N, seed=21, **PROXY)
    test = DD.synthetic_separable(n=NT, seed=22, **PROXY)
    ds = DD.ArrayDataset(clean.data.copy(), clean.targets.copy())
    DD.poison_dataset(ds, DD.synthetic_classwise_poison(10, 32, POISON_EPS, seed=3), "classwise")
    accs = {"clean": _classifier_accuracy(clean.data, clean.targets, test),
            "poisoned": _classifier_accuracy(ds.data, ds.targets, test)}
    m = _train_dvae(ds, test, "f32")
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ocfg = O.DVAEConfig()
    po = {k: sd[k].float() for k in O.param_shapes(ocfg)}
    eps = torch.randn(2, N, 256, 16, 16, generator=torch.Generator().manual_seed(31))
    purified = {}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    purified["oracle"], _ = O.purify(po, ocfg, ds.data, ds.targets, lambda ps, lo, hi: eps[ps, lo:hi], 1, batch=128)
    for dt in ("f32", "bf16"):
        purified[dt] = _purify_hip(sd, ds, eps, dt)
    dq = np.abs(purified["f32"].astype(int) - purified["oracle"].astype(int))
    assert dq.max() <= 1 and (dq != 0).mean() <= 0.005
    purified["sign-error"] = _purify_with_sign_error(sd, ds, eps)
    runs = {}
    for tag, arr in purified.items():
        # the three arms of the +-1 band: three classifier seeds each.  1-LSB differences between two purified sets send
        # two trainings apart chaotically (the fp32 arm differs from the oracle's set in < 0.5 % of the pixels by one grey
        # level and its three runs still spread over a point), so the band is +-1 point plus two standard errors of the
        # difference of the two means, estimated from the runs themselves (variance pooled over the three arms) and printed.
        seeds = (1, 2, 3) if tag in ("oracle", "f32", "bf16") else (1,)
        runs[tag] = [_classifier_accuracy_once(arr, ds.targets, test, "f32", sd_) for sd_ in seeds]
        accs[tag] = float(np.mean(runs[tag]))
    # (one variance for the three arms, pooled: they are the same experiment up to grey-level flips, and three runs of one
    # arm can agree to 0.03 points by chance -- seen on the bf16 arm -- which would make that arm's own estimate useless)
    var = float(np.mean([np.var(runs[k], ddof=1) for k in ("oracle", "f32", "bf16")]))
    se_diff = float(np.sqrt(2.0 * var / 3.0))
    band = {k: 1.0 + 2.0 * se_diff for k in ("f32", "bf16")}
    print("per-seed accuracies:", {k: [round(a, 2) for a in v] for k, v in runs.items()}, "bands:", {k: round(v, 2) for k, v in band.items()})
    trained = {"f32": [], "bf16": []}
    for dt in ("f32", "bf16"):
        for seed in (2, 3, 4):
            mt = m if (dt, seed) == ("f32", 2) else _train_dvae(ds, test, dt, seed=seed)
            sdt = {k: v.detach().cpu().clone() for k, v in mt.state_dict().items()}
            trained[dt].append(_classifier_accuracy(_purify_hip(sdt, ds, eps, dt), ds.targets, test))
        accs[dt + "-trained"] = float(np.median(trained[dt]))
    print("clean test accuracy (%) behind three D-VAE trainings per dtype:", {k: [round(a, 2) for a in v] for k, v in trained.items()})
    # the classifier stage itself in bf16 on the oracle-purified set
    accs["oracle, bf16 classifier"] = _classifier_accuracy(purified["oracle"], ds.targets, test, act_dtype="bf16")
    print("clean test accuracy (%) of a ResNet-18 trained on each version of the set:",
          {k: round(v, 2) for k, v in accs.items()})
    assert accs["clean"] - accs["poisoned"] >= 20.0, accs
    assert abs(accs["f32"] - accs["oracle"]) <= band["f32"] and abs(accs["bf16"] - accs["oracle"]) <= band["bf16"], (accs, band)
    assert max(band.values()) < 3.0, band                    # (the sign-error arm below must stay outside it)
    assert accs["bf16-trained"] >= accs["f32-trained"] - 6.0, accs
    assert accs["sign-error"] < accs["oracle"] - 3.0, accs
    for k in ("oracle", "f32", "bf16", "f32-trained", "bf16-trained"):
        assert accs[k] >= accs["poisoned"] + 8.0, accs
    assert abs(accs["oracle, bf16 classifier"] - accs["oracle"]) <= 3.0, accs


def test_device_to_host_staging_is_exact():
    """The chunked pinned-buffer copy of purify_dataset's tail (ragged last chunk, dtype conversion on the device, more
    chunks than staging buffers) and the per-batch drain give the bytes of a plain .cpu()."""
    from dvae_amd import trainer as T
    g = torch.Generator(device="cuda").manual_seed(3)
    src = torch.randn(53, 3, 8, 8, generator=g, device="cuda")
    dst = torch.empty(53, 3, 8, 8, dtype=torch.float64)
    T._to_host_chunked(src, dst, chunk_rows=7)
    assert torch.equal(dst, src.cpu().double())
    u8 = torch.randint(0, 256, (53, 8, 8, 3), generator=g, dtype=torch.uint8, device="cuda")
    d8 = torch.zeros(53, 8, 8, 3, dtype=torch.uint8)
    T._to_host_chunked(u8, d8, chunk_rows=16)
    assert torch.equal(d8, u8.cpu())
    d8.zero_(); dst.zero_()
    drain = T._HostDrain(d8, dst, 10, (8, 8, 3), (3, 8, 8))
    for lo in range(0, 53, 10):                                  # six batches through three staging buffers
        hi = min(53, lo + 10)
        drain.push(lo, hi, u8[lo:hi], src[lo:hi])
    drain.finish()
    assert torch.equal(d8, u8.cpu()) and torch.equal(dst, src.cpu().double())
