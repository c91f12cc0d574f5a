"""Aggregate a `bench.py --layers` table (stderr) by kernel family and role (forward / data gradient / weight gradient / ...)."""
import collections, re, sys
for path in sys.argv[1:]:
    fam = collections.defaultdict(lambda: [0.0, 0.0])
    for line in open(path):
        m = re.match(r"(\S+)\s+(\S+)\s+([\d.]+) us\s+([\d.]+) GFLOP\s+([\d.]+) TFLOP/s\s+([\d.]+) GB", line)
        if not m:
            continue
        n, k, us, gb = m.group(1), m.group(2), float(m.group(3)), float(m.group(6))
        last = n.split(".")[-1]
        role = "wgrad" if last == "wgrad" else "dgrad" if last.startswith("dgrad") else last if last in ("act_bwd", "absmax", "stats") else \
            "in_bwd" if last.startswith("in_bwd") else "fwd"
        fam[(k, role)][0] += us
        fam[(k, role)][1] += gb
    tot = sum(v[0] for v in fam.values())
    print(path, "total timed us", round(tot))
    for (k, sf), (v, gb) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:24]:
        print("   %-32s %-8s %9.0f us %5.1f%%  %7.2f GB %5.2f TB/s" % (k, sf, v, 100 * v / tot, gb, gb / v * 1e-3 if v else 0))
