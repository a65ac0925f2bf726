"""Per-kernel time of ONE train step from a rocprofv3 kernel trace of `bench.py --no-purify --no-graph`.

    rocprofv3 --kernel-trace -d DIR -o tr -- python3 bench.py --no-purify --no-graph
    python scripts/train_breakdown.py DIR/tr_results.db [--timeline]
"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
last = max(i for i, r in enumerate(rows) if "purify_quant" in r[0])
tr = rows[last + 1:]
ends = [i for i, r in enumerate(tr) if "clip_adam" in r[0]]
step = tr[ends[-2] + 1:ends[-1] + 1]           # from just after one Adam launch to the next
agg = collections.OrderedDict()
tot = 0.0
for r in step:
    k = r[0].replace("dvae::", "").replace("_ZN4dvae", "")[:70]
    e = agg.setdefault(k, [0, 0.0])
    e[0] += 1
    e[1] += (r[2] - r[1]) / 1e3
    tot += (r[2] - r[1]) / 1e3
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:70s} {n:4d} {t:8.1f} us")
print(f"kernel sum {tot:.1f} us, wall {(step[-1][2] - step[0][1]) / 1e3:.1f} us, launches {len(step)}")
if "--timeline" in sys.argv:
    t0 = step[0][1]
    for r in step:
        print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f}us grid {r[3] // r[6]:5d}x{r[4]}x{r[5]} {r[0][:90]}")
