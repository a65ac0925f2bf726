#!/bin/bash
# Secondary measurements next to the headline bench line: purification stage 0 / 1 at batch 128 and 2048, the train step
# at per-GPU batch 128 / 1024 / 4096.  One JSON line per case in gpurun_out/sweep.jsonl.
set -e
cd /root/repo
OUT=gpurun_out/sweep.jsonl; : > $OUT
for st in 0 1; do for pb in 128 2048; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-train --stage $st --purify-batch $pb --steps 20 >> $OUT
done; done
for tb in 128 1024 4096; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-purify --train-batch $tb --steps 10 --images 16384 >> $OUT
done
