"""Where the wall time of a whole-dataset purification pass goes outside the kernels: cProfile of one warm
purify_dataset call (one rank), and the same pass at the shard size of an 8-rank job (6 250 images)."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dvae_amd  # noqa: E402,F401
from dvae_amd import data as DD  # noqa: E402
from dvae_amd.model import ResNet18DVAE  # noqa: E402
from dvae_amd.trainer import purify_dataset  # noqa: E402

torch.manual_seed(0)
m = ResNet18DVAE(num_classes=10, stem_end_block=7, latent_dim=256, act_dtype="bf16")
m.sample_wise, m.use_y, m.spatial_emb = True, False, True
m.set_spatial_emb()
m = m.cuda()
for n in (50000, 6250):
    ds = DD.synthetic_cifar(n=n, seed=0)
    purify_dataset(m, DD.ArrayDataset(ds.data[:2048].copy(), ds.targets[:2048]), 1, batch=1024, seed=0)
    for rep in range(3):
        tm = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        purify_dataset(m, ds, 1, batch=1024, seed=0, timings=tm)
        torch.cuda.synchronize()
        w = time.perf_counter() - t0
        print(n, "wall %.1f ms" % (w * 1e3), {k: round(v * 1e3, 1) for k, v in tm.items()}, flush=True)
    pr = cProfile.Profile()
    pr.enable()
    purify_dataset(m, ds, 1, batch=1024, seed=0)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
    print("\n".join(s.getvalue().splitlines()[:30]))
