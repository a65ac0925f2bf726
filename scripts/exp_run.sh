#!/bin/bash
# times one conv shape with the tree's library and with each scripts/exp_N.so (timing-only variants: results are wrong)
set -e
cd $GRAFT_REPO_ROOT
cp d-vae_amd/lib/libdvae_hip.so /tmp/exp_0.so
for e in 0 1 2 3 4 0; do
  if [ $e = 0 ]; then cp /tmp/exp_0.so d-vae_amd/lib/libdvae_hip.so; else cp scripts/exp_$e.so d-vae_amd/lib/libdvae_hip.so; fi
  echo -n "exp $e: "; timeout -k 10 120 python scripts/conv_time.py "$@"
done
cp /tmp/exp_0.so d-vae_amd/lib/libdvae_hip.so
