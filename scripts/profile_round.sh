#!/bin/bash
# The measurement evidence in one GPU call: the un-profiled bench line, a kernel trace with per-kernel statistics of the
# same command, and the two counter passes (FETCH_SIZE, WRITE_SIZE) the HBM-traffic figure comes from.
# Outputs land in gpurun_out/prof_<tag>/.
set -e
TAG=${1:-bf16}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /root/repo
timeout -k 10 400 python3 bench.py --dtype $TAG > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 /root/repo/bench.py --dtype $TAG --no-cpu-baseline > $OUT/bench_under_rocprof.json
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 /root/repo/bench.py --dtype $TAG --steps 3 --warmup 1 --no-cpu-baseline --no-train > $OUT/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 /root/repo/bench.py --dtype $TAG --steps 3 --warmup 1 --no-cpu-baseline --no-train > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
