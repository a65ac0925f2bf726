"""Average duration of the launch bench.py brackets with HIP events, taken from a rocprofv3 kernel trace.

usage: python scripts/probe_from_trace.py <..._kernel_trace.csv> [images_per_launch]

The bracketed launch is the first 3x3 64->64 convolution of layer1 in every stem pass of the purification
region: in the trace it is the conv_halo_bf16_kernel<64,64,32> (bf16) or igemm_kernel (fp32) dispatch that directly
follows the conv1 dispatch (bf16: the conv_stem kernel; fp32: the gather GEMM launched with grid 2048*images threads).
"""
import csv
import json
import sys


def main():
    path = sys.argv[1]
    images = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    durs = []
    for prev, cur in zip(rows, rows[1:]):
        conv = "conv_halo" in cur["Kernel_Name"] or "igemm_kernel" in cur["Kernel_Name"]
        first = "conv_stem" in prev["Kernel_Name"] or ("igemm_kernel" in prev["Kernel_Name"] and
                                                        int(prev["Grid_Size_X"]) == 2048 * images)
        if conv and first:
            durs.append(int(cur["End_Timestamp"]) - int(cur["Start_Timestamp"]))
            name = cur["Kernel_Name"]
    out = {"kernel": name, "launches": len(durs), "avg_ns": sum(durs) / len(durs), "min_ns": min(durs),
           "max_ns": max(durs), "images_per_launch": images}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
