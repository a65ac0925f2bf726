#!/bin/bash
# Samples the card's power draw and clocks (rocm-smi) while the purification step runs in a loop: is the step power-limited?
cd "$(dirname "$0")/.."
OUT=gpurun_out/power_probe.log
: > $OUT
rocm-smi --showpower --showclocks --showmaxpower >> $OUT 2>&1
echo "=== idle above, load below" >> $OUT
timeout -k 10 200 python -u scripts/ab_wreg_bench.py > gpurun_out/power_probe_bench.log 2>&1 &
BENCH=$!
sleep 20
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" >> $OUT
    echo "--" >> $OUT
    sleep 0.3
done
wait $BENCH
echo "bench exit $?" >> $OUT
