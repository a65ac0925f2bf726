"""Totals per kernel name from a rocprofv3 --kernel-trace CSV: python scripts/kernel_totals.py <trace.csv> [substring of the
kernel that ends a region] -- launches, total and mean time, share."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
tot = 0.0
for r in rows:
    k = r["Kernel_Name"].replace("dvae::", "").replace("void ", "")[:90]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    e = agg.setdefault(k, [0, 0.0])
    e[0] += 1
    e[1] += d
    tot += d
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t / tot * 100:5.1f} %  {n:6d} x {t / n:8.1f} us  {k}")
print(f"total {tot / 1e3:.1f} ms in {len(rows)} launches")
