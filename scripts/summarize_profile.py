"""Turn the rocprofv3 output of scripts/profile_r03.sh (gpurun_out/prof_r03; r02 likewise) into the files kept under profiles/:
  r02_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the default `python bench.py`
  r02_probe_kernel.json        the probed conv kernel's 1024-image launches in that trace (count, mean, min, max)
  r02_traffic_bf16.json        FETCH_SIZE / WRITE_SIZE of those launches (one counter per pass) and the HBM bytes they
                               give: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE counts a 128-byte
                               request of a 16-byte-per-lane stream as 64 bytes, WRITE_SIZE is exact for such stores
"""
import csv
import json
import os
import shutil
import statistics as st

import sys

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r03"       # usage: python scripts/summarize_profile.py [r02|r03] [images per probed launch]
IMAGES = int(sys.argv[2]) if len(sys.argv) > 2 else 2048  # bench.py's default purification batch at 32x32 (pass 1024 for traces taken with --purify-batch 1024)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + ROUND)
DST = os.path.join(ROOT, "profiles")
K = "void dvae::conv_wreg_bf16_kernel<64, 64, 1, false, false, false>(dvae::HaloArgs)"   # plain: no residual, no input normalisation, no statistics

shutil.copy(os.path.join(SRC, "trace", "bench_kernel_stats.csv"), os.path.join(DST, ROUND + "_bench_kernel_stats.csv"))
rows = list(csv.DictReader(open(os.path.join(SRC, "trace", "bench_kernel_trace.csv"))))
us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if r["Kernel_Name"] == K]
# the purification launches: the train step's 128-image ones are many times shorter, and a launch that pays a one-off cost (the
# first of the process) several times longer
p90 = sorted(us)[int(0.9 * len(us))]
big = [x for x in us if 0.6 * p90 < x < 2.0 * p90]
probe = {"kernel": K, "images_per_launch": IMAGES, "launches_of_that_size": len(big), "mean_us": st.mean(big), "min_us": min(big), "max_us": max(big),
         "launches_all_shapes": len(us), "outliers_above_us": [x for x in us if x >= 2.0 * p90],
         "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py"}
json.dump(probe, open(os.path.join(DST, ROUND + "_probe_kernel.json"), "w"), indent=1)


def counter(name):
    rr = [r for r in csv.DictReader(open(os.path.join(SRC, f"pmc_{name}", "bench_counter_collection.csv")))
          if r["Kernel_Name"] == K]
    vals = [float(r["Counter_Value"]) for r in rr]
    return st.mean(vals), len(vals), min(vals), max(vals)


f, nf, fmin, fmax = counter("fetch")
w, nw, wmin, wmax = counter("write")
traffic = {"kernel": K, "launch": f"3x3 conv 64->64 @32x32, {IMAGES} images (purification region of bench.py)", "images_per_launch": IMAGES,
           "FETCH_SIZE_KB": {"mean": f, "launches": nf, "min": fmin, "max": fmax},
           "WRITE_SIZE_KB": {"mean": w, "launches": nw, "min": wmin, "max": wmax},
           "correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes)",
           "hbm_bytes": (2 * f + w) * 1024, "algorithmic_bytes": 2 * IMAGES * 1024 * 64 * 2,
           "commands": ["rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-train --no-f32 --no-pass --no-cpu-baseline",
                        "rocprofv3 --pmc WRITE_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-train --no-f32 --no-pass --no-cpu-baseline"]}
json.dump(traffic, open(os.path.join(DST, ROUND + "_traffic_bf16.json"), "w"), indent=1)
print(json.dumps(probe), json.dumps(traffic))
