"""Does a consumer that reads a tensor right after its producer wrote it run faster than HBM allows (memory-side cache)?
y = x + 1 (writes y), then s = y * 2 (reads y, writes s) on bf16 tensors of growing size; prints the consumer's GB/s, and the
same with a 2 GB tensor touched between the two launches (cold)."""
import torch

torch.cuda.init()
big = torch.empty(1 << 30, dtype=torch.bfloat16, device="cuda")
for mb in (32, 64, 128, 192, 256, 384, 512, 1024):
    n = mb * (1 << 20) // 2
    x = torch.randn(n, device="cuda").bfloat16()
    y = torch.empty_like(x)
    s = torch.empty_like(x)
    for cold in (False, True):
        ts = []
        for rep in range(12):
            torch.add(x, 1, out=y)
            if cold:
                big.add_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.mul(y, 2, out=s)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[len(ts) // 2]
        print(f"{mb:5d} MB {'cold' if cold else 'warm'}: {t * 1e3:8.1f} us  {2 * mb / 1024 / (t * 1e-3):7.1f} GB/s (read + write)", flush=True)
