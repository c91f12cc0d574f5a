#!/bin/bash
# usage: bash scratch/trace_kernels.sh <regex> <bench args...>: rocprofv3 kernel trace of a bench.py run; prints the longest launches whose kernel name matches
re=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/trace
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $R/bench.py "$@" > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - "$out" "$re" <<'PY'
import csv, glob, sys, re, collections
out, rx = sys.argv[1], re.compile(sys.argv[2])
rows = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if rx.search(r["Kernel_Name"]):
            rows.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")))
agg = collections.defaultdict(lambda: [0.0, 0])
for d, k, *_ in rows:
    agg[k][0] += d; agg[k][1] += 1
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-92s %5d launches  %9.1f us total  %8.1f us avg" % (k, n, t / 1e3, t / n / 1e3))
print("longest:")
for d, k, gx, gy, gz in sorted(rows, reverse=True)[:24]:
    print("  %8.1f us  grid (%s, %s, %s)  %s" % (d / 1e3, gx, gy, gz, k))
PY
rm -rf $out/*/ 2>/dev/null
