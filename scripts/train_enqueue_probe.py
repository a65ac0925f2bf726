import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dvae_amd import _lib as L
from dvae_amd.engine import DVAEConfig, DVAEEngine
cfg = DVAEConfig(**bench.model_kwargs(32)); torch.manual_seed(0)
eng = DVAEEngine(cfg, 2048, "cuda", L.BF16, seed=0); bench.init_params(eng)
n, Bt = 50000, 128
data = torch.randint(0, 256, (n, 32, 32, 3), dtype=torch.uint8, device="cuda"); labels = (torch.arange(n, device="cuda") % 10).long()
eng.set_hyper(lr=5e-3, max_norm=5.0, grad_scale=1.0, steps_done=0)
idx = torch.randperm(n, device="cuda", dtype=torch.int64).to(torch.int32)
def step(i):
    lo = (i * Bt) % (n - Bt); sel = idx[lo:lo + Bt]
    x0 = eng.input_from_u8(data, 0, Bt, index=sel); eps = eng.eps_philox(Bt, index=sel, stream_id=100)
    eng.train_step(x0, labels[sel.long()], eps, Bt, 11, 3.0, 0.01, None, 1)
for i in range(30): step(i)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(100): step(30 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/100:.3f} ms/step, total {1e3*(t2-t0)/100:.3f} ms/step", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(100): step(200 + i)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
ts = []
for i in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(400 + i)
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print(f"enqueue of one step into an empty queue: median {1e3 * sorted(ts)[10]:.3f} ms", flush=True)
