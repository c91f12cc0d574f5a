"""Steady-state kernel statistics from two rocprofv3 --kernel-trace --stats runs of the same command: `--steps K` and `--steps 0`
(warm-up only: plan building, packing, calibration and the W warm-up steps).  Per kernel name: (calls_K - calls_0) / K calls per step,
(total_K - total_0) / K ns per step.  usage: steady_stats.py <stats_K.csv> <stats_0.csv> <K> <out.csv>"""
import csv, sys
full, warm, K, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return d
a, b = load(full), load(warm)
rows = []
for k, (c, t) in a.items():
    c0, t0 = b.get(k, (0, 0.0))
    dc, dt = c - c0, t - t0
    if dc <= 0:
        continue
    rows.append((k, dc / K, dt / K, dt / dc))
tot = sum(r[2] for r in rows)
rows.sort(key=lambda r: -r[2])
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "CallsPerStep", "TotalNsPerStep", "AverageNs", "Percentage"])
    for k, cps, tps, avg in rows:
        w.writerow([k, "%.3f" % cps, "%.0f" % tps, "%.0f" % avg, "%.3f" % (100 * tps / tot)])
print("steady state: %d kernel names, %.3f ms of kernel time per step, %.1f launches per step" % (len(rows), tot / 1e6, sum(r[1] for r in rows)))
names = " ".join(r[0] for r in rows)
for pat in ("FillFunctor", "copyBuffer", "absmax_k", "Cijk_", "elementwise_kernel"):
    hits = [(r[0][:70], r[1]) for r in rows if pat in r[0]]
    print("   %-18s %s" % (pat, hits if hits else "none in the steady state"))
