#!/bin/bash
# A/B of two builds of the library on the purification step inside one gpurun call, alternating: scripts/ab_old.so and
# scripts/ab_new.so (both git-ignored) are copied over the in-tree library in turn; prints images/s, ms per step and the
# HIP-event time of the probed conv launch.  Extra arguments go to bench.py.
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for v in old new; do
    cp scripts/ab_$v.so d-vae_amd/lib/libdvae_hip.so
    timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-f32 --no-pass --no-train "$@" 2>/dev/null > /tmp/ab_line.json || exit 1
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
r = d["roofline"]
print(f"{sys.argv[1]}: {d['value'] / 1e3:8.1f} k images/s  {d['ms_per_step']:.3f} ms/step  probe {r['avg_launch_ms'] * 1e3:.1f} us  mfma_frac {r['mfma_frac']:.3f}")
PY
  done
done
cp scripts/ab_old.so d-vae_amd/lib/libdvae_hip.so
