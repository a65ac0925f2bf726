#!/bin/bash
# A/B of two library builds on ONE box: per-kernel forward timeline with the tree's library, then with scripts/ab_old.so
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in new old; do
  if [ $v = old ]; then cp scripts/ab_old.so d-vae_amd/lib/libdvae_hip.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_$v -- python3 bench.py --no-cpu-baseline --no-train --steps 8 --warmup 2 > /dev/null 2>&1
  echo "== $v"; python scripts/forward_timeline.py gpurun_out/ab_$v/*/*_kernel_trace.csv 8 | awk '{print $1, $5, $6}' | tr '\n' ';' | fold -w 220
done
