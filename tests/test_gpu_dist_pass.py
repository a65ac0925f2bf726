"""The multi-rank purification pass and train step through the drop-in API with TWO processes sharing the one GPU of
the test box (gloo between them; RCCL wants a card per rank): every rank purifies its index shard and delivers it
through shared host memory (main_vae_disentangle.py:59-69 writes the array from one process; here no rank moves
another rank's rows), and the result -- uint8 set on every rank, float64 perturbations on rank 0 -- must be bit-identical
to the one-process pass, in fp32 (stage 1, two forwards) and in bf16 (stage 0), with a ragged last batch per rank."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, BATCH = 600, 128          # ragged: 300 images per rank = 2 full batches + 44


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dtype):
    from oracle import dvae_oracle as O
    from dvae_amd.model import ResNet18DVAE
    m = ResNet18DVAE(num_classes=10, stem_end_block=7, latent_dim=256, act_dtype=dtype)
    m.sample_wise, m.use_y, m.spatial_emb = True, False, True
    m.set_spatial_emb()
    m.load_state_dict(O.recipe_params(O.DVAEConfig()), strict=False)
    return m.cuda()


def _pass(stage, dtype):
    from dvae_amd import data as DD
    from dvae_amd.trainer import purify_dataset
    ds = DD.synthetic_cifar(n=N, seed=11)
    m = _model(dtype)
    tm = {}
    noise = purify_dataset(m, ds, stage, batch=BATCH, seed=4, timings=tm)
    return ds, noise, tm


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    from dvae_amd import data as DD
    from dvae_amd import dist as D
    torch.cuda.set_device(0)
    D.init_from_env("gloo")
    for stage, dtype in ((1, "f32"), (0, "bf16")):
        ds, noise, tm = _pass(stage, dtype)
        assert (noise is None) == (rank != 0)
        np.save(os.path.join(tmp, f"u8_{stage}_{rank}.npy"), ds.data)
        if rank == 0:
            assert noise.dtype == torch.float64 and tuple(noise.shape) == (N, 3, 32, 32)
            np.save(os.path.join(tmp, f"noise_{stage}.npy"), noise.numpy())
        assert set(tm) == {"upload_s", "kernels_s", "gather_s", "host_tail_s", "total_s"}
        assert not [f for f in os.listdir("/dev/shm") if f.startswith(f"dvae_{os.getpid()}_")]
        # the next stage's loader sees the new version of the array
        ld = DD.DeviceLoader(ds, 64, False, False, "cuda")
        x, _ = next(iter(ld))
        assert torch.equal((x[0] * 255).round().byte().permute(1, 2, 0).cpu(), torch.from_numpy(ds.data[0]))
    dist.destroy_process_group()


def test_two_rank_pass_through_shared_memory_equals_one_rank(tmp_path):
    import torch.multiprocessing as mp
    ref = {}
    for stage, dtype in ((1, "f32"), (0, "bf16")):
        ds, noise, _ = _pass(stage, dtype)
        ref[stage] = (ds.data.copy(), noise.numpy().copy())
    torch.cuda.synchronize()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for stage in (1, 0):
        for rank in (0, 1):
            got = np.load(tmp_path / f"u8_{stage}_{rank}.npy")
            assert np.array_equal(got, ref[stage][0]), (stage, rank)
        assert np.array_equal(np.load(tmp_path / f"noise_{stage}.npy"), ref[stage][1]), stage
    assert not np.array_equal(ref[0][0], ref[1][0])
