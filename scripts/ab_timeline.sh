#!/bin/bash
# A/B of two library builds on ONE box: per-kernel forward timeline with the tree's library ("new") and with
# scripts/ab_old.so ("old"), alternating old, new, old, new -- a later run on a warm card is a few per cent slower than
# the first, so a single new-then-old pair flatters "new".  The two libraries must agree on every buffer layout the
# Python host allocates (the host is the tree's): copy the tree's library to scripts/ab_old.so after any layout change.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp d-vae_amd/lib/libdvae_hip.so /tmp/ab_new.so
for v in old new old new; do
  if [ $v = old ]; then cp scripts/ab_old.so d-vae_amd/lib/libdvae_hip.so; else cp /tmp/ab_new.so d-vae_amd/lib/libdvae_hip.so; fi
  rm -rf gpurun_out/ab_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_$v -- python3 bench.py --no-cpu-baseline --no-train --steps 8 --warmup 2 > /dev/null 2>&1
  echo "== $v"; python scripts/forward_timeline.py gpurun_out/ab_$v/*/*_kernel_trace.csv 8 | awk '{print $1, $5, $6}' | tr '\n' ';' | fold -w 220
done
cp /tmp/ab_new.so d-vae_amd/lib/libdvae_hip.so
