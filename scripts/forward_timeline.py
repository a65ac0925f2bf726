"""Per-kernel timeline of one steady-state purification forward from a rocprofv3 kernel trace.

usage: python scripts/forward_timeline.py <..._kernel_trace.csv> [skip_forwards]
A forward starts at the conv1 dispatch (the gather-GEMM launched right after the uint8->activation or quantiser kernel).
"""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "purify_quant" in r["Kernel_Name"]]
    a, b = starts[skip], starts[skip + 1]
    tot = 0
    for r in rows[a + 1:b + 1]:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        tot += d
        n = re.sub(r"dvae::|void ", "", r["Kernel_Name"])[:72]
        print(f"{d/1000:8.1f} us  grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}  {n}")
    span = int(rows[b]["End_Timestamp"]) - int(rows[a + 1]["Start_Timestamp"])
    print(f"sum of kernels {tot/1000:.1f} us, wall span {span/1000:.1f} us")


if __name__ == "__main__":
    main()
