"""The data-parallel D-VAE train step (world size 2) on ONE GPU: two engines stand for the two ranks, each takes its
half of an 8-image global batch, and the two collectives of the step (the 4-float loss-sum all-reduce ahead of the
rate-constraint gate, the gradient-bucket all-reduce ahead of clip + Adam) are performed by hand, exactly where
``DVAEEngine.train_step`` calls ``allreduce``.  Checked against the oracle's DataParallel arithmetic
(main_vae_disentangle.py:426-428: per-replica BatchNorm statistics, whole-batch loss, summed gradients), which is itself
pinned to a fixture made from the reference's module (tests/golden/g6_dp_step.npz).

Tolerances (fp32 both sides): loss scalars 1e-4 relative, gradients 1e-4 of the tensor's scale, parameters after the step
3e-3 in L2 (Adam's first update is sign-like); the two ranks' parameters must be bit-identical."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dvae_oracle as O

if torch.cuda.is_available():
    import dvae_amd
    from dvae_amd import _lib as L
    from dvae_amd.engine import DVAEConfig, DVAEEngine
    from test_gpu_model import NULL_GRAD_BIASES, engine_relu_masks
    from util_gpu import l2_err, rel_err


def _two_rank_step(engs, x, y, eps, epoch, target, weight):
    """One world-2 step on two engines, issued piecewise in the order DVAEEngine.train_step uses."""
    per = x.shape[0] // 2
    for r, eng in enumerate(engs):
        sl = slice(r * per, (r + 1) * per)
        x0 = eng.input_from_nchw(x[sl].cuda())
        e = eng.eps_from_nchw(eps[sl].cuda())
        eng.forward(x0, y[sl].cuda(), e, per, True)
        eng.loss_sums()
    total = engs[0].sums + engs[1].sums                  # all-reduce #1 (4 floats)
    for eng in engs:
        eng.sums.copy_(total)
        eng.loss_gate(epoch, target, weight, world=2)
        eng.backward()
    g = engs[0].G + engs[1].G                            # all-reduce #2 (the gradient bucket)
    for eng in engs:
        eng.G.copy_(g)
        eng.adam_step()


def test_two_rank_train_step_matches_data_parallel_oracle(golden_dir):
    ocfg, cfg = O.DVAEConfig(), DVAEConfig()
    p = O.recipe_params(ocfg)
    x, y, eps = O.make_inputs(ocfg, 8, seed=4321)
    target, weight = O.stage_constants(1)
    engs = []
    for r in range(2):
        eng = DVAEEngine(cfg, 4, "cuda", L.F32)
        eng.load_params(p)
        eng.set_hyper(lr=5e-3, max_norm=5.0, grad_scale=0.5)          # 1 / world
        engs.append(eng)
    # forward once to read the engines' ReLU masks (gradients are differentiated with them, see test_gpu_model.py)
    masks, flips = [], 0
    for r, eng in enumerate(engs):
        sl = slice(4 * r, 4 * r + 4)
        x0 = eng.input_from_nchw(x[sl].cuda())
        eng.forward(x0, y[sl].cuda(), eng.eps_from_nchw(eps[sl].cuda()), 4, True)
        m, f = engine_relu_masks(eng, 4, ocfg, p, x[sl], y[sl], eps[sl])
        masks.append(m)
        flips += f
        eng.load_params(p)                                            # undo the BatchNorm running-stat update
    po = {k: v.clone() for k, v in p.items()}
    info, grads, stats = O.train_step_data_parallel(po, ocfg, x, y, eps, {}, 11, 1, world=2, masks=masks)
    _two_rank_step(engs, x, y, eps, 11, target, weight)
    g = np.load(os.path.join(golden_dir, "g6_dp_step.npz"))
    for eng in engs:
        scal = eng.scal.cpu().numpy()
        for k in ("loss", "loss_rec", "loss_rec_p", "loss_p", "kld", "grad_norm"):
            assert abs(scal[L.SCAL[k]] - info[k]) <= 1e-4 * abs(info[k]) + 1e-7, (k, scal[L.SCAL[k]], info[k])
            ref = float(g[k])
            tol = 1e-4 if (flips == 0 or k != "grad_norm") else 2e-2
            assert abs(scal[L.SCAL[k]] - ref) <= tol * abs(ref) + 1e-7, (k, scal[L.SCAL[k]], ref)
    # summed gradient bucket: engine holds the SUM of the per-rank gradients, each scaled by its local mean
    # (2 / local_n); the oracle's whole-batch gradient is that sum times 1 / world
    for k, gref in grads.items():
        got = engs[0].grad[k].cpu() * 0.5
        if k in NULL_GRAD_BIASES:
            assert float(got.abs().max()) < 1e-6
            continue
        assert rel_err(got, gref) < 1e-4, k
    # both ranks hold bit-identical parameters and Adam state after the step
    assert torch.equal(engs[0].P, engs[1].P) and torch.equal(engs[0].M, engs[1].M) and torch.equal(engs[0].V, engs[1].V)
    for k in ("conv1.weight", "fc_var.bias", "decoder1.0.weight", "emb_p.weight", "layer1.1.bn2.bias", "layer2.0.conv2.weight"):
        assert l2_err(engs[0].param[k].cpu(), po[k]) < 3e-3, k
    # BatchNorm running statistics are per replica; rank 0's are the ones DataParallel keeps
    for r, eng in enumerate(engs):
        for k in ("bn1.running_mean", "layer1.0.bn2.running_var", "layer2.0.shortcut.1.running_mean"):
            assert rel_err(eng.buffers[k].cpu(), stats[r][k]) < 5e-5, (r, k)
    for k in ("bn1.running_mean", "layer1.0.bn2.running_var", "layer2.0.shortcut.1.running_mean"):
        assert rel_err(engs[0].buffers[k].cpu(), g["stat." + k]) < 5e-5, k
    assert not torch.equal(engs[0].R, engs[1].R)


def test_train_step_with_allreduce_callable_equals_piecewise_step():
    """DVAEEngine.train_step(allreduce=..., world=2) issues the same launches in the same order as the piecewise drive
    above: rank 1's contributions are precomputed and injected by the callable, and the result must be bit-identical
    to the piecewise two-rank step (this covers the wiring of ``allreduce`` / ``world`` / ``grad_scale`` inside
    train_step, side stream and all)."""
    ocfg, cfg = O.DVAEConfig(), DVAEConfig()
    p = O.recipe_params(ocfg)
    x, y, eps = O.make_inputs(ocfg, 8, seed=77)
    target, weight = O.stage_constants(1)
    ref = []
    for r in range(2):
        eng = DVAEEngine(cfg, 4, "cuda", L.F32)
        eng.load_params(p)
        eng.set_hyper(lr=5e-3, max_norm=5.0, grad_scale=0.5)
        ref.append(eng)
    # rank 1's partial results, computed ahead: its loss sums, and its gradient bucket given the GLOBAL sums
    for r, eng in enumerate(ref):
        sl = slice(4 * r, 4 * r + 4)
        eng.forward(eng.input_from_nchw(x[sl].cuda()), y[sl].cuda(), eng.eps_from_nchw(eps[sl].cuda()), 4, True)
        eng.loss_sums()
    sums = [e.sums.clone() for e in ref]
    total = sums[0] + sums[1]
    for eng in ref:
        eng.sums.copy_(total)
        eng.loss_gate(11, target, weight, world=2)
        eng.backward()
    G = [e.G.clone() for e in ref]
    for eng in ref:
        eng.G.copy_(G[0] + G[1])
        eng.adam_step()
    torch.cuda.synchronize()
    # now rank 0 alone through train_step, with the collectives played by a callable
    eng = DVAEEngine(cfg, 4, "cuda", L.F32)
    eng.load_params(p)
    eng.set_hyper(lr=5e-3, max_norm=5.0, grad_scale=0.5)
    calls = []

    def allreduce(t):
        calls.append(t.numel())
        if t.numel() == 4:
            t.add_(sums[1])
        else:                                               # a contiguous piece of the gradient bucket
            off = (t.data_ptr() - eng.G.data_ptr()) // 4
            t.add_(G[1][off:off + t.numel()])

    x0 = eng.input_from_nchw(x[:4].cuda())
    eng.train_step(x0, y[:4].cuda(), eng.eps_from_nchw(eps[:4].cuda()), 4, 11, target, weight, allreduce, 2)
    torch.cuda.synchronize()
    # the bucket travels in two pieces: its tail (latent convs, decoders, embedding) while the stem's backward still runs
    assert calls == [4, eng.nparam - eng.early_offset, eng.early_offset]
    assert torch.equal(eng.scal[:8], ref[0].scal[:8])
    assert torch.equal(eng.P, ref[0].P) and torch.equal(eng.M, ref[0].M)
