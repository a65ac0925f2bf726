#!/bin/bash
# Profiles of bench.py for profiles/: kernel trace + stats of the default command, then the HBM-traffic counters of the
# purification region alone, one counter per pass.  Run on the GPU box from the repository root.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py > $OUT/bench_traced.json 2> $OUT/trace.err
echo "trace done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py --steps 3 --warmup 1 --no-train --no-f32 --no-pass --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py --steps 3 --warmup 1 --no-train --no-f32 --no-pass --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "write done"
find $OUT -name "*.csv" | head -20
