import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import torch.nn.functional as F
from oracle import dvae_oracle as O
import dvae_amd
from dvae_amd import _lib as L
from dvae_amd.engine import DVAEConfig, DVAEEngine
from util_gpu import rel_err, from_act, to_act
ocfg = O.DVAEConfig(); cfg = DVAEConfig()
eng = DVAEEngine(cfg, 4, "cuda", L.F32)
p = O.recipe_params(ocfg); eng.load_params(p)
x, y, eps = O.make_inputs(ocfg, 4)
acts = {}
def bn(prefix, t):
    return F.batch_norm(t, p[prefix+".running_mean"].clone(), p[prefix+".running_var"].clone(), p[prefix+".weight"], p[prefix+".bias"], True, 0.1, 1e-5)
with torch.no_grad():
    pre = bn("bn1", F.conv2d(x, p["conv1.weight"], None, 1, 1)); acts["bn1.out"] = pre; a = F.relu(pre)
    for blk in ["layer1.0", "layer1.1", "layer2.0"]:
        s = 2 if blk == "layer2.0" else 1
        pre1 = bn(blk+".bn1", F.conv2d(a, p[blk+".conv1.weight"], None, s, 1)); acts[blk+".bn1.out"] = pre1; t = F.relu(pre1)
        o = bn(blk+".bn2", F.conv2d(t, p[blk+".conv2.weight"], None, 1, 1))
        sc = bn(blk+".shortcut.1", F.conv2d(a, p[blk+".shortcut.0.weight"], None, s, 0)) if blk == "layer2.0" else a
        pre2 = o + sc; acts[blk+".out"] = pre2; a = F.relu(pre2)
x0 = eng.input_from_nchw(x.cuda()); e = eng.eps_from_nchw(eps.cuda())
eng.forward(x0, y.cuda(), e, 4, True)
for k, pre in acts.items():
    c = pre.shape[1]
    got = from_act(eng._bufs[k][:4], c)
    mm = (got > 0) != (pre > 0)
    print(k, "elements", pre.numel(), "mask mismatches", int(mm.sum()), "|pre| at mismatches", pre[mm].abs().tolist()[:5], "scale", float(pre.abs().max()))
