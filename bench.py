#!/usr/bin/env python3
"""Benchmark of the D-VAE hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], synthetic): CIFAR-10-shape uint8 images resident in HBM,
ResNet18DVAE(num_classes=10, stem_end_block=7, latent_dim=256), sample_wise / use_y=False /
spatial_emb (the published script flags), random-init weights.

One "step" = one stage-1 purification batch per GPU (two eval-mode forwards, clamp, uint8 quantise,
main_vae_disentangle.py:59-69).  ``value`` = purified images/s over all ranks.  The D-VAE train step
(main_vae_disentangle.py:180-211 without the model_c monitors) is timed in a second region of the
same run and reported as ``train_images_per_sec`` (per-GPU batch 128, gradient all-reduce over RCCL).
Purification shards by image index (no data-path collective inside the timed region; the all-gather
of the purified shards happens once per pass and is timed separately as ``allgather_ms``).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import dvae_amd  # noqa: E402
from dvae_amd import _lib as L  # noqa: E402
from dvae_amd.engine import DVAEConfig, DVAEEngine  # noqa: E402

FWD_MACS = 418_775_040           # one model_s forward, classifier head excluded (per image, 32x32)
TRAIN_MACS = 1_254_555_648       # forward + dgrad + wgrad, no dgrad for conv1
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}   # dense MFMA peaks, gfx950
# Algorithmic HBM bytes of one bf16 inference forward as it is launched, per 32x32 image: every launch's input (+ residual)
# read once and its output written once, weights and statistics not counted.  (name, bytes read, bytes written)
FWD_BYTES_BF16 = [
    ("uint8 -> 8-channel input", 3 * 1024, 16 * 1024),
    ("conv1 3 -> 64", 16 * 1024, 128 * 1024),
    ("layer1.0.conv1", 128 * 1024, 128 * 1024),
    ("layer1.0.conv2 + residual", 256 * 1024, 128 * 1024),
    ("layer1.1.conv1", 128 * 1024, 128 * 1024),
    ("layer1.1.conv2 + residual", 256 * 1024, 128 * 1024),
    ("layer2.0.conv1 (stride 2)", 128 * 1024, 64 * 1024),
    ("layer2.0.shortcut (every other pixel of every other row)", 32 * 1024, 64 * 1024),
    ("layer2.0.conv2 + residual", 128 * 1024, 64 * 1024),
    ("fc_mu | fc_var + reparameterisation + post_conv, one launch: two outputs (z' and z' + class embedding)", 64 * 1024, 128 * 1024),
] + 2 * [
    ("decoder ConvT 4x4 128 -> 64", 64 * 1024, 128 * 1024),
    ("decoder ConvT 3x3 64 -> 64", 128 * 1024, 128 * 1024),
    ("decoder ConvT 3x3 64 -> 32", 128 * 1024, 64 * 1024),
    ("decoder ConvT 3x3 32 -> 3 (fp32 out, 4 channels stored)", 64 * 1024, 16 * 1024),
]
FWD_HBM_BYTES_BF16 = sum(r + w for _, r, w in FWD_BYTES_BF16)     # 3.7 MB per image and forward
HBM_PEAK_GBS = 8000.0
CONV_PROBE = "layer1.0.conv1"    # 3x3 64->64 @ 32x32: the shape that carries most of the FLOPs
CONV_PROBE_MACS = 64 * 64 * 9 * 32 * 32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--purify-batch", type=int, default=None,
                    help="images per GPU per purification step (default 2048 at 32x32, 1024 at 64x64, 128 at 224x224)")
    ap.add_argument("--train-batch", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"],
                    help="activation / MFMA input type (accumulation, statistics, losses, optimiser are fp32 either way)")
    ap.add_argument("--images", type=int, default=None,
                    help="size of the synthetic set (default 50 000; 8 192 at 224x224, the ImageNet-mini configuration)")
    ap.add_argument("--stage", type=int, default=1, choices=[0, 1],
                    help="purification stage: 1 (two forwards, the headline) or 0 (one forward, x - p)")
    ap.add_argument("--image-size", type=int, default=32, choices=[32, 64, 224],
                    help="32: CIFAR-10 shape (the headline workload); 64: Tiny-ImageNet shape (200 classes, plain class embedding); "
                         "224: ImageNet-mini shape (100 classes, plain class embedding)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="weight gradients on the main stream (clean per-kernel timings)")
    ap.add_argument("--no-batch-reduce", action="store_true", help="reduce every conv's weight-gradient slabs in a launch of its own")
    ap.add_argument("--graph", action="store_true", help="replay the train step from captured HIP graphs (default: launch by launch; "
                    "the step is GPU-bound and the two-stream schedule runs faster outside a graph)")
    ap.add_argument("--no-purify", action="store_true", help="profiling aid: skip the purification region")
    ap.add_argument("--no-f32", action="store_true", help="skip the fp32 (reference precision) leg of a bf16 run")
    ap.add_argument("--no-pass", action="store_true", help="skip the whole-dataset pass through purify_dataset")
    a = ap.parse_args()
    if a.purify_batch is None:
        # (about 2 M pixels per launch at 32x32, 4 M at 64x64, 6 M at 224x224: a launch carries 2-6 us of fixed cost,
        # and the 32x32 pass is fastest with 2048-image batches: 1024 / 2048 / 4096 give 432 / 461 / 479 k images/s per batch
        # and 0.132 / 0.130 / 0.138 s for the 50 000-image pass)
        a.purify_batch = {32: 2048, 64: 1024, 224: 128}[a.image_size]
    if a.images is None:
        a.images = 8192 if a.image_size == 224 else 50000
    return a


def respawn_under_torchrun(n):
    """``python bench.py --gpus N`` outside torchrun: start N ranks as a child job before anything touches the GPU and
    leave with its exit code (replacing this process is not an option once a GPU runtime is loaded)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def barrier(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(t, world):
    if world > 1:
        x = torch.tensor([t], device="cuda", dtype=torch.float64)
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        return float(x.item())
    return t


def model_kwargs(S):
    return {} if S == 32 else dict(image_size=S, spatial_emb=False, num_classes=200 if S == 64 else 100)


def shape_name(S):
    return {32: "CIFAR-10 shape", 64: "64x64 Tiny-ImageNet shape", 224: "224x224 ImageNet-mini shape"}[S]


def cpu_baseline(n_images, stage=1, S=32):
    """The CPU oracle (stock fp32 torch ops restating the reference) purifying a bounded sample."""
    from oracle import dvae_oracle as O
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    cfg = O.DVAEConfig(**model_kwargs(S))
    p = O.recipe_params(cfg)
    g = torch.Generator().manual_seed(0)
    data = torch.randint(0, 256, (n_images, S, S, 3), generator=g, dtype=torch.uint8).numpy()
    labels = (torch.arange(n_images) % cfg.num_classes).numpy()
    eps = lambda ps, lo, hi: torch.randn(hi - lo, 256, S // 2, S // 2, generator=g)   # noqa: E731
    O.purify(p, cfg, data[:128], labels[:128], eps, stage, batch=128)          # warm-up
    t0 = time.perf_counter()
    O.purify(p, cfg, data, labels, eps, stage, batch=128)
    dt = time.perf_counter() - t0
    return {"value": n_images / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{n_images} synthetic {S}x{S}x3 images, stage-1 purification (2 forwards), batch 128, fp32 torch CPU"}


def cpu_baseline_train(steps=5, batch=128):
    """The CPU oracle's D-VAE train step (forward, losses, backward, clip, Adam; no model_c monitors) at the reference
    batch size."""
    from oracle import dvae_oracle as O
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    cfg = O.DVAEConfig()
    p = O.recipe_params(cfg)
    x, y, eps = O.make_inputs(cfg, batch, seed=0)
    st = {}
    O.train_step(p, cfg, x, y, eps, st, 11, 1)              # warm-up
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        O.train_step(p, cfg, x, y, eps, st, 11, 1)
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[len(ts) // 2]
    return {"value": batch / med, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{steps} train steps at batch {batch} (median), synthetic 32x32x3, fp32 torch CPU, no monitor compute"}


def purify_region(eng, a, world, data, out, noise, labels, Bp, steps, probe_name):
    """Warm-up, then ``steps`` timed purification batches (barrier + synchronize on both sides, max over ranks).
    Returns (images/s over all ranks, ms per step, sorted HIP-event times of the probed conv launch in ms)."""
    nb = data.shape[0] // Bp
    for i in range(a.warmup):
        eng.purify_batch(data, out, noise, labels, (i % nb) * Bp, Bp, a.stage)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    eng.probe = None
    barrier(world)
    t0 = time.perf_counter()
    for i in range(steps):
        eng.probe = (probe_name, ev[i])
        eng.purify_batch(data, out, noise, labels, ((a.warmup + i) % nb) * Bp, Bp, a.stage)
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    eng.probe = None
    return world * Bp * steps / dt, dt / steps * 1e3, sorted(s.elapsed_time(e) for s, e in ev)


def probe_roofline(dtype_name, probe_ms, Bp, px, S):
    """Roofline object of the probed launch (layer1.0.conv1: 3x3 conv 64->64).  Algorithmic work per image:
    2 * 37 748 736 FLOP and (read x + write y) = 2 * 64 ch * 1024 px * element size bytes.  fp32: 144 FLOP/B against a
    ridge of 157.3 T / 8 T = 20 -> MFMA-bound.  bf16: 288 FLOP/B against a ridge of 2500 / 8 = 312 -> HBM-bound (just)."""
    avg = sum(probe_ms) / len(probe_ms)
    peak = PEAK_TFLOPS[dtype_name]
    tflops = 2.0 * CONV_PROBE_MACS * px * Bp / (avg * 1e-3) / 1e12
    nbytes = 2.0 * 64 * 1024 * px * (4 if dtype_name == "f32" else 2) * Bp
    gbs = nbytes / (avg * 1e-3) / 1e9
    if dtype_name == "bf16":
        # 288 FLOP/B is under the ridge (312): HBM is the roof that bounds the launch -- 33.5 us at 8 TB/s against 30.9 us
        # of MFMA at the nominal peak -- and the MFMA figure is reported beside it
        roof = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                "mfma_tflops": tflops, "mfma_frac": tflops / peak}
        kname = "conv_wreg_bf16_kernel<64,64,1,false,false,false>"
    else:
        roof = {"bound": "mfma", "achieved": tflops, "peak": peak, "unit": "TFLOP/s", "frac": tflops / peak, "hbm_gbs": gbs}
        kname = "igemm_kernel<float,2,1,2,2>"
    roof.update(kernel=f"{kname} ({CONV_PROBE}: 3x3 conv 64->64 @{S}x{S}, {Bp} images/launch)", avg_launch_ms=avg,
                launches_timed=len(probe_ms), algorithmic_bytes_per_launch=nbytes,
                algorithmic_flops_per_launch=2.0 * CONV_PROBE_MACS * px * Bp,
                traffic=None, traffic_source=None)
    # HBM bytes per launch are a PMC figure (rocprofv3 --pmc, one counter per pass, of this same command with fewer steps):
    # not measurable inside this run, so the figure of the committed passes is reported when they are of this launch
    tpath = os.path.join(ROOT, "profiles", f"r03_traffic_{dtype_name}.json")
    if not os.path.exists(tpath):
        tpath = os.path.join(ROOT, "profiles", f"r02_traffic_{dtype_name}.json")
    if os.path.exists(tpath) and S == 32:
        t = json.load(open(tpath))
        if kname in t.get("kernel", "").replace(", ", ",") and Bp == t.get("images_per_launch", 1024):
            roof["traffic"] = t.get("hbm_bytes")
            roof["traffic_source"] = os.path.relpath(tpath, ROOT) + " (separate rocprofv3 --pmc passes of the same command, not this run)"
    return roof


def dataset_pass(a, world, rank, dtype_name, S, cfg):
    """One stage-1 pass over the whole synthetic set through the drop-in API (``purify_dataset``): upload of this rank's
    shard, the per-batch kernels, the all-gather of the uint8 shards + gather of the perturbations to rank 0, and the host
    tail (device -> host copies, float64 conversion, ``dataset.data`` overwritten in place).  Wall time, max over ranks."""
    from dvae_amd import data as DD
    from dvae_amd.model import ResNet18DVAE
    from dvae_amd.trainer import purify_dataset
    torch.manual_seed(0)
    m = ResNet18DVAE(num_classes=cfg.num_classes, stem_end_block=7, latent_dim=256, act_dtype=dtype_name)
    m.sample_wise, m.use_y, m.spatial_emb = True, False, cfg.spatial_emb
    m.set_spatial_emb()
    m = m.cuda()
    ds = DD.synthetic_cifar(n=a.images, size=S, classes=cfg.num_classes, seed=0)
    small = DD.ArrayDataset(ds.data[:a.purify_batch * world].copy(), ds.targets[:a.purify_batch * world])
    purify_dataset(m, small, a.stage, batch=a.purify_batch, seed=0)              # warm-up: engine build, first launches
    tm = {}
    barrier(world)
    t0 = time.perf_counter()
    res = purify_dataset(m, ds, a.stage, batch=a.purify_batch, seed=0, timings=tm)   # (kept: unmapping 614 MB is not the pass)
    barrier(world)
    wall = max_over_ranks(time.perf_counter() - t0, world)
    del res
    out = {"images": a.images, "stage": a.stage, "wall_s": wall, "images_per_sec": a.images / wall}
    for k, v in tm.items():
        out[k] = max_over_ranks(v, world)
    out["kernel_only_images_per_sec"] = a.images / out["kernels_s"]
    if world > 1:
        # strong scaling of the pass: the same set through ONE rank of this job (the others wait at the barrier), then
        # efficiency = t(1 rank) / (N * t(N ranks))
        from dvae_amd import dist as D
        solo_s = 0.0
        if rank == 0:
            ds1 = DD.synthetic_cifar(n=a.images, size=S, classes=cfg.num_classes, seed=0)
            with D.solo():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = purify_dataset(m, ds1, a.stage, batch=a.purify_batch, seed=0)
                torch.cuda.synchronize()
                solo_s = time.perf_counter() - t0
                del res
        barrier(world)
        solo_s = max_over_ranks(solo_s, world)
        out["one_rank_wall_s"] = solo_s
        out["efficiency_vs_1gpu"] = solo_s / (world * wall)
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(a.gpus))
    if a.gpus != int(os.environ.get("WORLD_SIZE", "1")):
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch one rank per GPU "
                 "(python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N) or drop WORLD_SIZE")
    # DVAE_BENCH_WATCHDOG=<seconds>: a run that has not finished by then prints every thread's Python stack and exits
    # (a hung collective or a kernel that never returns then names the line it sits on)
    wd = float(os.environ.get("DVAE_BENCH_WATCHDOG", "0"))
    if wd > 0:
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # DVAE_BENCH_BACKEND=gloo rehearses the multi-rank code path with every rank on the GPUs that are there
    # (ranks share a card; RCCL wants one card per rank); the numbers of such a run mean nothing
    backend = os.environ.get("DVAE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dtype = L.F32 if a.dtype == "f32" else L.BF16
    S = a.image_size
    cfg = DVAEConfig(**model_kwargs(S))
    px = (S * S) // 1024                # work per image relative to 32x32 (every layer scales with the pixel count)
    torch.manual_seed(0)

    # ---------------- purification: images resident in HBM, sharded by index -------------------
    Bp = a.purify_batch
    eng = DVAEEngine(cfg, max(Bp, a.train_batch), "cuda", dtype, seed=0)
    init_params(eng)
    if a.no_overlap:
        eng._overlap = False
    if a.no_batch_reduce:
        eng._batch_reduce = False
    n_local = max(Bp, (a.images + world - 1) // world)
    g = torch.Generator(device="cuda").manual_seed(rank)
    data = torch.randint(0, 256, (n_local, S, S, 3), generator=g, dtype=torch.uint8, device="cuda")
    labels = (torch.arange(n_local, device="cuda") % cfg.num_classes).long()
    out = torch.empty_like(data)
    noise = torch.empty(n_local, 3, S, S, device="cuda")
    nb = n_local // Bp

    steps_p = 1 if a.no_purify else a.steps
    if a.no_purify:
        a_w, a.warmup = a.warmup, 0
    purify_ips, purify_ms, probe_ms = purify_region(eng, a, world, data, out, noise, labels, Bp, steps_p, CONV_PROBE)
    if a.no_purify:
        a.warmup = a_w
    # extra, never `value`: the same pass without the first forward's reconstruction decoder, whose result the
    # reference drops (main_vae_disentangle.py:61-62); outputs are bit-identical (tests/test_gpu_bf16.py)
    lean_ips = None
    if not a.no_purify and a.stage == 1:
        barrier(world)
        t0 = time.perf_counter()
        for i in range(steps_p):
            eng.purify_batch(data, out, noise, labels, (i % nb) * Bp, Bp, 1, skip_unread=True)
        barrier(world)
        lean_ips = world * Bp * steps_p / max_over_ranks(time.perf_counter() - t0, world)

    # one all-gather of the purified shards (once per pass, outside the per-batch hot loop)
    allgather_ms = None
    if world > 1:
        gathered = torch.empty(world * n_local, S, S, 3, dtype=torch.uint8, device="cuda")
        barrier(world)
        t1 = time.perf_counter()
        dist.all_gather_into_tensor(gathered, out)
        barrier(world)
        allgather_ms = max_over_ranks((time.perf_counter() - t1) * 1e3, world)

    # ---------------- train step -----------------------------------------------------------------
    train_ips = train_ms = None
    if not a.no_train:
        Bt = a.train_batch
        eng.set_hyper(lr=5e-3, max_norm=5.0, grad_scale=1.0 / world, steps_done=0)
        # RCCL reduces the device tensors in place, ordered with the current stream; the gloo rehearsal goes through a
        # host copy (dvae_amd.dist: gloo stalls on device tensors of a process with queued kernels when ranks share a card)
        from dvae_amd import dist as DD_
        allreduce = DD_.allreduce_sum_ if world > 1 else None
        idx = torch.randperm(n_local, device="cuda", dtype=torch.int64).to(torch.int32)

        def train_step(i):
            lo = (i * Bt) % (n_local - Bt)
            sel = idx[lo:lo + Bt]
            x0 = eng.input_from_u8(data, 0, Bt, index=sel)
            eps = eng.eps_philox(Bt, index=sel, stream_id=100)
            step = eng.train_step_graphed if a.graph else eng.train_step
            step(x0, labels[sel.long()], eps, Bt, 11, 3.0, 0.01, allreduce, world)

        for i in range(a.warmup):
            train_step(i)
        barrier(world)
        t0 = time.perf_counter()
        for i in range(a.steps):
            train_step(a.warmup + i)
        barrier(world)
        dtt = max_over_ranks(time.perf_counter() - t0, world)
        train_ips = world * Bt * a.steps / dtt
        train_ms = dtt / a.steps * 1e3

    # ---------------- the reference's own precision, same run -------------------------------------
    f32_leg = None
    if a.dtype == "bf16" and not a.no_f32 and not a.no_purify:
        del eng
        torch.cuda.empty_cache()
        eng32 = DVAEEngine(cfg, Bp, "cuda", L.F32, seed=0)
        init_params(eng32)
        ips32, ms32, probe32 = purify_region(eng32, a, world, data, out, noise, labels, Bp, max(3, steps_p // 4), CONV_PROBE)
        r32 = probe_roofline("f32", probe32, Bp, px, S)
        f32_leg = {"value": ips32, "unit": "images/s", "ms_per_step": ms32, "dtype": "f32",
                   "probe_frac": r32["frac"], "roofline": r32}
        del eng32
        torch.cuda.empty_cache()

    # ---------------- one whole-dataset pass through the drop-in API -------------------------------
    pass_obj = None
    if not a.no_pass and not a.no_purify:
        pass_obj = dataset_pass(a, world, rank, a.dtype, S, cfg)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    res = {
        "metric": "purified images/sec (node) + VAE train imgs/sec, " + shape_name(S),
        "value": purify_ips, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": purify_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": f"synthetic uint8 {S}x{S}x3 images resident in HBM, random-init weights",
        "config": {"workload": shape_name(S) + f" {S}x{S}x3, {cfg.num_classes} classes" +
                               ", ResNet18DVAE(seb=7, latent=256), " +
                               ("stage-1 purification (2 eval forwards + clamp + uint8 quantise)" if a.stage == 1 else
                                "stage-0 purification (1 eval forward + clamp + uint8 quantise)") +
                               f", {Bp} images/GPU/step, sharded by image index",
                   "purify_batch_per_gpu": Bp, "train_batch_per_gpu": a.train_batch, "images_per_gpu": n_local},
        "train_images_per_sec": train_ips, "train_ms_per_step": train_ms, "allgather_ms": allgather_ms,
        "purify_images_per_sec_skipping_unread_decoder": lean_ips,
        "purify_tflops": 2.0 * (1 + a.stage) * FWD_MACS * px * purify_ips / 1e12,
        "train_tflops": (2.0 * TRAIN_MACS * px * train_ips / 1e12) if train_ips else None,
        "roofline": probe_roofline(a.dtype, probe_ms, Bp, px, S),
    }
    # the whole timed regions against the dense MFMA peak of their arithmetic (every launch, small ones and gaps included)
    peak = PEAK_TFLOPS[a.dtype] * world
    res["step_roofline"] = {"bound": "mfma", "peak": peak, "unit": "TFLOP/s",
                            "purify": {"achieved": res["purify_tflops"], "frac": res["purify_tflops"] / peak},
                            "train": ({"achieved": res["train_tflops"], "frac": res["train_tflops"] / peak}
                                      if train_ips else None)}
    if a.dtype == "bf16":
        # ... and against the HBM peak of its activations' traffic: every layer is a launch of its own, so every activation
        # is written once and read once (twice where it is also a residual); 288 FLOP/B for the 64-channel convs against a
        # ridge of 312: the step sits on both roofs at once
        gbs = (1 + a.stage) * FWD_HBM_BYTES_BF16 * px * purify_ips / 1e9
        res["step_roofline"]["purify_hbm"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                              "frac": gbs / (HBM_PEAK_GBS * world),
                                              "algorithmic_bytes_per_image_forward": FWD_HBM_BYTES_BF16 * px}
    if f32_leg is not None:
        res["f32"] = f32_leg
        res["value_f32"] = f32_leg["value"]               # the reference's own precision, same run
        res["step_roofline"]["purify_f32"] = {"achieved": 2.0 * (1 + a.stage) * FWD_MACS * px * f32_leg["value"] / 1e12,
                                              "frac": 2.0 * (1 + a.stage) * FWD_MACS * px * f32_leg["value"] / 1e12 /
                                              (PEAK_TFLOPS["f32"] * world), "peak": PEAK_TFLOPS["f32"] * world}
    if pass_obj is not None:
        res["pass"] = pass_obj
    if not a.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline(1024 // px, S=S)
        if S == 32:
            res["cpu_baseline_train"] = cpu_baseline_train()
    print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def init_params(eng):
    """Random initialisation with torch's default distributions (kaiming_uniform(a=sqrt 5) for conv
    weights and biases, N(0,1) embedding, BatchNorm gamma=1 beta=0), as the reference's modules do."""
    import math
    g = torch.Generator().manual_seed(0)
    t = {}
    for name, shape, _, _ in eng.layout:
        if name == "emb_p.weight":
            t[name] = torch.randn(shape, generator=g)
        elif len(shape) == 4:
            b = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])
            t[name] = (torch.rand(shape, generator=g) * 2 - 1) * b
        elif name.endswith(".bias") and ("bn" not in name and "shortcut.1" not in name):
            wshape = dict((n, s) for n, s, _, _ in eng.layout)[name[:-4] + "weight"]
            b = 1.0 / math.sqrt(wshape[1] * wshape[2] * wshape[3])
            t[name] = (torch.rand(shape, generator=g) * 2 - 1) * b
        elif name.endswith(".weight"):
            t[name] = torch.ones(shape)
        else:
            t[name] = torch.zeros(shape)
    eng.load_params(t, strict=False)


if __name__ == "__main__":
    main()
