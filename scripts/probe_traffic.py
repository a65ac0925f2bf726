"""HBM traffic of the launch bench.py brackets with HIP events, from two rocprofv3 --pmc passes.

usage: python scripts/probe_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [images_per_launch]

The bracketed launch is the conv dispatch right after conv1 (bf16: the conv_stem kernel; fp32: the gather GEMM launched
with 2048 * images threads).
FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64
bytes, so it is doubled before being compared with a byte count; WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import csv
import json
import sys


def probe_values(path, counter, images):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals, name = [], None
    # bf16 runs: the bracketed launch is a conv_halo kernel (the fused encoder head has conv1's total grid size too,
    # and is followed by a gather GEMM); fp32 runs have no conv_halo launches and no fused head
    halo_run = any("conv_halo" in r["Kernel_Name"] for r in rows)
    for prev, cur in zip(rows, rows[1:]):
        conv = "conv_halo" in cur["Kernel_Name"] if halo_run else "igemm_kernel" in cur["Kernel_Name"]
        first = "conv_stem" in prev["Kernel_Name"] if halo_run else ("igemm_kernel" in prev["Kernel_Name"] and
                                                                      int(prev["Grid_Size"]) == 2048 * images)
        if conv and first:
            vals.append(float(cur["Counter_Value"]))
            name = cur["Kernel_Name"]
    return vals, name


def main():
    images = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    f, name = probe_values(sys.argv[1], "FETCH_SIZE", images)
    w, _ = probe_values(sys.argv[2], "WRITE_SIZE", images)
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    out = {"kernel": name, "launches": [len(f), len(w)], "FETCH_SIZE_KB_raw": fk, "WRITE_SIZE_KB_raw": wk,
           "fetch_bytes_corrected": fk * 1024 * 2, "write_bytes": wk * 1024,
           "hbm_bytes": fk * 1024 * 2 + wk * 1024, "algorithmic_bytes": 2 * 64 * 1024 * 2 * images,
           "note": "separate --pmc passes (FETCH_SIZE; WRITE_SIZE), counters in KB; FETCH_SIZE doubled (gfx950 counts the "
                   "128-byte requests of 16-byte-per-lane streaming reads at 64 bytes)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
