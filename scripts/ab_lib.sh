#!/bin/bash
# A/B of two builds of the library inside one gpurun call, alternating: scripts/ab_old.so and scripts/ab_new.so are copied
# over the in-tree library in turn (both git-ignored), and the train step is timed at batch 128 and 1024.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in old new; do
    cp scripts/ab_$v.so d-vae_amd/lib/libdvae_hip.so
    a=$(timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-f32 --no-pass --no-purify 2>/dev/null | grep -o "train_ms_per_step\": [0-9.]*") || exit 1
    b=$(timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-f32 --no-pass --no-purify --train-batch 1024 --steps 10 2>/dev/null | grep -o "train_ms_per_step\": [0-9.]*") || exit 1
    echo "$v: $a | $b"
  done
done
cp scripts/ab_new.so d-vae_amd/lib/libdvae_hip.so
