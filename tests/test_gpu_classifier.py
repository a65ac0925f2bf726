"""The classifier stage's ResNet-18 trunk on libdvae_hip (main_vae_disentangle.py:97-125 trains
/root/reference/configs/models/resnet.py:62-122 on the purified set): the device path against the CPU oracle on the same
seeded inputs and against the fixture made with the reference's own module (tests/golden/g7_resnet18_step.npz).

Tolerances.  fp32: logits and running statistics 5e-5 of scale, loss 1e-4 relative, parameter gradients 1e-4 of the
tensor's scale against the oracle differentiated with the device path's own ReLU masks (a pre-activation within rounding
of zero flips one mask element between two correct fp32 implementations; every disagreement is checked to sit at such a
pre-activation), gradient norms against the reference fixture 1e-4 when no mask flipped (5e-2 otherwise).  bf16 (bf16
activations, gradient maps and packed weights; fp32 accumulation, statistics, parameters, gradients): held to the oracle
run with bf16 rounding at the device path's storage points (see test_train_step_bf16_sits_inside_bf16_rounding)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import dvae_oracle as O

if torch.cuda.is_available():
    import dvae_amd
    from dvae_amd.classifier import ResNet18
    from util_gpu import from_act, l2_err, rel_err

SUB = 37
GSUB = ("conv1.weight", "layer2.0.shortcut.0.weight", "layer3.1.conv2.weight", "layer4.0.conv1.weight", "layer4.1.bn2.weight",
        "linear.weight")
STATS = ("bn1.running_mean", "layer1.1.bn2.running_var", "layer3.0.shortcut.1.running_mean", "layer4.1.bn2.running_var")


def inputs(B=8, seed=777):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, 3, 32, 32, generator=g), (torch.arange(B) * 3 + 1) % 10


def model(dtype, params=None):
    m = ResNet18(10, act_dtype=dtype)
    m.load_state_dict(params if params is not None else O.resnet18_recipe(10), strict=False)
    return m.cuda()


def device_relu_masks(m, p, x, b):
    """(output > 0) of every ReLU as the device trunk computed it; each disagreement with the oracle's own mask must sit
    at a pre-activation within rounding of zero."""
    eng = m._engine
    pre = {}
    with torch.no_grad():
        O.resnet18_forward(p, x, True, {}, None, pre)
    names = [("bn1.out", 64, 32)]
    for blk in eng.blocks:
        sp = eng.specs[blk + ".conv1"]
        names += [(blk + ".bn1.out", sp.cout, sp.hout), (blk + ".out", sp.cout, sp.hout)]
    masks, flips = {}, 0
    for n, c, h in names:
        mk = from_act(eng._bufs[n][:b], c) > 0
        bad = mk != (pre[n] > 0)
        if bad.any():
            assert float(pre[n][bad].abs().max()) <= 2e-5 * float(pre[n].abs().max()), n
        flips += int(bad.sum())
        masks[n] = mk
    assert flips <= 8
    return masks, flips


def test_train_step_fp32_vs_oracle_and_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g7_resnet18_step.npz"))
    p = O.resnet18_recipe(10)
    x, y = inputs()
    m = model("f32", p)
    m.train()
    logits = m(x.cuda())
    loss = F.cross_entropy(logits, y.cuda())
    loss.backward()
    masks, flips = device_relu_masks(m, p, x, 8)
    print(f"ReLU mask flips against the oracle: {flips}")
    ref_logits, ref_loss, grads, stats = O.resnet18_train_grads(p, x, y, masks)
    assert rel_err(logits, ref_logits) < 5e-5 and rel_err(logits, g["train.logits"]) < 5e-5
    assert abs(float(loss) - ref_loss) <= 1e-4 * ref_loss and abs(float(loss) - float(g["train.loss"])) <= 1e-4 * ref_loss
    named = dict(m.named_parameters())
    for k, gref in grads.items():
        assert named[k].grad is not None, k
        assert rel_err(named[k].grad, gref) < 1e-4, k
        ref = float(g["gnorm." + k])
        assert abs(float(named[k].grad.double().norm()) - ref) <= (1e-4 if flips == 0 else 5e-2) * ref, k
    if flips == 0:
        for k in GSUB:
            assert rel_err(named[k].grad.reshape(-1)[::SUB], g["gsub." + k]) < 1e-4, k
    sd = m.state_dict()
    for k in STATS:
        assert rel_err(sd[k], stats[k]) < 5e-5 and rel_err(sd[k], g["stat." + k]) < 5e-5, k
    assert int(sd["layer3.0.bn1.num_batches_tracked"]) == 1
    # eval mode: folded BatchNorm in the conv epilogues, running statistics as loaded
    m2 = model("f32", p)
    m2.eval()
    with torch.no_grad():
        lg = m2(x.cuda())
    assert rel_err(lg, g["eval.logits"]) < 5e-5


def structured_inputs(B, seed):
    """Images of the class-template stand-in for CIFAR (dvae_amd.data.synthetic_separable) as fp32 NCHW + labels."""
    from dvae_amd import data as DD
    ds = DD.synthetic_separable(n=B, seed=seed)
    return torch.from_numpy(ds.data).permute(0, 3, 1, 2).float().div(255), torch.from_numpy(ds.targets)


@pytest.mark.parametrize("B", [32, 128])
def test_train_step_bf16_sits_inside_bf16_rounding(B):
    """One train step of the bf16 trunk.  The bound is derived from bf16 rounding itself: the oracle is run a second time
    with every tensor the device path stores in bf16 (activations, gradient maps, packed weights) rounded at the same
    points (oracle.bf16_storage), fp32 everywhere else.  BatchNorm's backward subtracts the common mode of a gradient map,
    which at initialisation is most of it, so the rounding of the stored maps is amplified layer by layer: the emulation
    itself ends 0.3-0.4 (relative L2) away from the fp32 gradient of the first layers.  The device path must (a) agree
    with the emulation on what is not amplified -- logits 2e-2, loss 1e-3 -- and (b) for every parameter tensor be no
    further from the fp32 gradient than 1.3 x the emulation is (+ 1e-2), and no further from the emulation than the
    emulation is from fp32 (+ 1e-2): it sits inside the ball bf16 storage draws around the exact result."""
    p = O.resnet18_recipe(10)
    x, y = structured_inputs(B, 5)
    _, loss32, g32, _ = O.resnet18_train_grads(p, x, y)
    with O.bf16_storage():
        lg16, loss16, g16, _ = O.resnet18_train_grads(p, x, y)
    m = model("bf16", p)
    m.train()
    logits = m(x.cuda())
    loss = F.cross_entropy(logits, y.cuda())
    loss.backward()
    assert l2_err(logits, lg16) < 2e-2
    assert abs(float(loss.detach()) - loss16) <= 1e-3 * loss16
    named = dict(m.named_parameters())
    worst = (0.0, "")
    for k in g32:
        dev = named[k].grad
        e_dev, e_emu, d = l2_err(dev, g32[k]), l2_err(g16[k], g32[k]), l2_err(dev, g16[k])
        assert e_dev <= 1.3 * e_emu + 1e-2, (k, e_dev, e_emu)
        assert d <= e_emu + 1e-2, (k, d, e_emu)
        worst = max(worst, (e_dev, k))
    a = torch.cat([named[k].grad.flatten().cpu().double() for k in g32])
    b = torch.cat([g16[k].flatten().double() for k in g32])
    c = torch.cat([g32[k].flatten().double() for k in g32])
    cos = lambda u, v: float((u * v).sum() / u.norm() / v.norm())      # noqa: E731
    print(f"bf16 trunk, batch {B}: logits vs emulation {l2_err(logits, lg16):.2e}; whole-gradient cosine device/emulation "
          f"{cos(a, b):.4f}, device/fp32 {cos(a, c):.4f}, emulation/fp32 {cos(b, c):.4f}; worst tensor vs fp32 {worst}")
    assert cos(a, c) >= cos(b, c) - 0.02
    if B == 32:
        m.eval()
        with torch.no_grad():
            big = m(torch.rand(300, 3, 32, 32, generator=torch.Generator().manual_seed(1)).cuda())    # larger batch: the engine is rebuilt
        assert big.shape == (300, 10) and torch.isfinite(big).all()


def test_sgd_steps_follow_the_cpu_modules_and_gradients_accumulate():
    """Three SGD steps (momentum 0.9, weight decay 5e-4, clip 5.0: configs/cifar10/resnet18.yaml:14-19; lr 0.01 instead of
    0.1 -- a clipped step at 0.1 moves the weights by 0.5 in norm and the two trajectories part chaotically) on the device
    trunk against the same module on the CPU (stock torch fp32, the reference's arithmetic): losses within 2e-3 and
    parameters within 5e-3 in L2 after the three steps (a ReLU mask flip or two per step are carried along), and
    ``backward`` twice without zeroing accumulates like autograd does."""
    p = O.resnet18_recipe(10)
    dev, cpu = model("f32", p), ResNet18(10)
    cpu.load_state_dict(p, strict=False)
    dev.train(); cpu.train()
    od = torch.optim.SGD(dev.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    oc = torch.optim.SGD(cpu.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    for step in range(3):
        x, y = inputs(16, 100 + step)
        ld = F.cross_entropy(dev(x.cuda()), y.cuda())
        lc = F.cross_entropy(cpu(x), y)
        assert abs(float(ld) - float(lc)) <= 2e-3 * abs(float(lc)) + 1e-5, (step, float(ld), float(lc))
        for o, l in ((od, ld), (oc, lc)):
            o.zero_grad()
            l.backward()
            torch.nn.utils.clip_grad_norm_(o.param_groups[0]["params"], 5.0)
            o.step()
    for k, v in cpu.state_dict().items():
        if v.is_floating_point():
            assert l2_err(dev.state_dict()[k], v) < 5e-3, k
    # accumulation
    x, y = inputs(8, 9)
    dev.zero_grad()
    F.cross_entropy(dev(x.cuda()), y.cuda()).backward()
    g1 = {k: v.grad.clone() for k, v in dev.named_parameters()}
    F.cross_entropy(dev(x.cuda()), y.cuda()).backward()           # BatchNorm statistics moved, the batch did not
    for k, v in dev.named_parameters():
        assert rel_err(v.grad, 2 * g1[k]) < 1e-5, k
    dev.zero_grad(set_to_none=False)
    assert all(float(v.grad.abs().max()) == 0 for v in dev.parameters())
    F.cross_entropy(dev(x.cuda()), y.cuda()).backward()
    for k, v in dev.named_parameters():
        assert rel_err(v.grad, g1[k]) < 1e-5, k
