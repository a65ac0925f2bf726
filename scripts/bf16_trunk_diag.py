"""bf16 device paths against the oracle with bf16 storage emulated (oracle.bf16_storage) and against the fp32 oracle:
per-tensor relative L2 error of the parameter gradients of one train step, classifier trunk and D-VAE."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, torch.nn.functional as F
from oracle import dvae_oracle as O
import dvae_amd
from dvae_amd import _lib as L
from dvae_amd.classifier import ResNet18
from dvae_amd.engine import DVAEConfig, DVAEEngine


def l2(a, b):
    a, b = a.flatten().cpu().double(), b.flatten().cpu().double()
    return float((a - b).norm() / b.norm())


from dvae_amd import data as DD
STRUCT = len(sys.argv) > 1 and sys.argv[1] == "structured"


def images(B, seed):
    """iid uniform noise (every image alike in distribution: BatchNorm's backward cancels most of the gradient), or the
    class-template stand-in for CIFAR (structured, class-dependent)."""
    if not STRUCT:
        g = torch.Generator().manual_seed(seed)
        return torch.rand(B, 3, 32, 32, generator=g), (torch.arange(B) * 3 + 1) % 10
    ds = DD.synthetic_separable(n=B, seed=seed)
    return torch.from_numpy(ds.data).permute(0, 3, 1, 2).float().div(255), torch.from_numpy(ds.targets)


p = O.resnet18_recipe(10)
for B in (32, 128):
    x, y = images(B, 5)
    _, loss32, g32, _ = O.resnet18_train_grads(p, x, y)
    with O.bf16_storage():
        lg16, loss16, g16, _ = O.resnet18_train_grads(p, x, y)
    m = ResNet18(10, act_dtype="bf16"); m.load_state_dict(p, strict=False); m = m.cuda().train()
    lg = m(x.cuda()); loss = F.cross_entropy(lg, y.cuda()); loss.backward()
    named = dict(m.named_parameters())
    print(f"classifier B={B}: loss {float(loss):.6f} emulated {loss16:.6f} fp32 {loss32:.6f}; logits vs emulated {l2(lg.detach(), lg16):.2e}")
    worst = max((l2(named[k].grad, g16[k]), k) for k in g16)
    print("   worst tensor vs emulated:", worst, " vs fp32:", max((l2(named[k].grad, g32[k]), k) for k in g32))
    for k in ("conv1.weight", "layer1.0.conv1.weight", "layer2.0.conv2.weight", "layer3.1.conv1.weight", "layer4.1.conv2.weight", "bn1.weight", "linear.weight"):
        print(f"   {k:26s} vs emulated {l2(named[k].grad, g16[k]):.3e}   vs fp32 {l2(named[k].grad, g32[k]):.3e}")

ocfg, cfg = O.DVAEConfig(), DVAEConfig()
pp = O.recipe_params(ocfg)
for B in (8, 128):
    eng = DVAEEngine(cfg, B, "cuda", L.BF16)
    eng.load_params(pp)
    x, y, eps = O.make_inputs(ocfg, B)
    if STRUCT:
        x, y = images(B, 6)
    i32, g32, _ = O.train_step({k: v.clone() for k, v in pp.items()}, ocfg, x, y, eps, {}, 11, 1)
    with O.bf16_storage():
        i16, g16, _ = O.train_step({k: v.clone() for k, v in pp.items()}, ocfg, x, y, eps, {}, 11, 1)
    x0 = eng.input_from_nchw(x.cuda()); e = eng.eps_from_nchw(eps.cuda())
    eng.forward(x0, y.cuda(), e, B, True); eng.loss_sums(); eng.loss_gate(11, 3.0, 0.01); eng.backward()
    s = eng.scal.cpu().numpy()
    print(f"D-VAE B={B}: " + ", ".join(f"{k} {s[L.SCAL[k]]:.6f}/{float(i16[k]):.6f}/{float(i32[k]):.6f}" for k in ("loss", "loss_rec", "loss_rec_p", "loss_p", "kld")))
    live = [k for k in g16 if not (k.endswith("1.1.bias") or k.endswith("2.0.bias"))]
    rows = sorted(((l2(eng.grad[k], g16[k]), l2(eng.grad[k], g32[k]), k) for k in live), reverse=True)
    for r in rows[:8]:
        print(f"   {r[2]:26s} vs emulated {r[0]:.3e}   vs fp32 {r[1]:.3e}")
    print("   median vs emulated", rows[len(rows) // 2][0])
