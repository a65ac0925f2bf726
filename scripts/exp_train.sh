#!/bin/bash
# train-step throughput with the tree's library ("base") and with scripts/exp_<tag>.so builds, alternating
set -e
cd $GRAFT_REPO_ROOT
cp d-vae_amd/lib/libdvae_hip.so /tmp/exp_base.so
for pass in 1 2; do
for e in base "$@"; do
  if [ $e = base ]; then cp /tmp/exp_base.so d-vae_amd/lib/libdvae_hip.so; else cp scripts/exp_$e.so d-vae_amd/lib/libdvae_hip.so; fi
  echo -n "$e: "; timeout -k 10 200 python bench.py --no-purify --no-cpu-baseline --steps 300 --warmup 20 | grep -o "\"train_images_per_sec\": [0-9.]*"
done
done
cp /tmp/exp_base.so d-vae_amd/lib/libdvae_hip.so
