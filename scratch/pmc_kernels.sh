#!/bin/bash
# Per-kernel PMC of a bench.py leg (runs on the GPU box via gpurun): separate rocprofv3 --pmc passes (with --kernel-trace only), per-launch
# averages per kernel name: VALU / SALU per MFMA, MFMA-busy share, LDS bank conflicts, wait share, and the sustained shader clock
# (GRBM_GUI_ACTIVE cycles of the launch / its duration from the kernel trace).
# usage: bash scratch/pmc_kernels.sh <tag> <bench.py args...>    -> gpurun_out/<tag>.txt
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $out/p$i -o r -- python3 $R/bench.py "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
  echo "pass $i done"
done
python3 - "$out" "$*" <<'PY' > $out.txt
import csv, glob, sys, collections, re
out, cmd = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
dur = collections.defaultdict(float); dcnt = collections.Counter()
def short(k):
    k = re.sub(r"^void \(anonymous namespace\)::", "", k)
    k = re.sub(r"\(egne_conv_desc.*$", "", k)
    return k[:84]
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for f in glob.glob(out + "/p5/**/*kernel_trace.csv", recursive=True):          # durations of the pass that also counted GRBM_GUI_ACTIVE
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dcnt[k] += 1
print("# rocprofv3 --kernel-trace --pmc <4 counters per pass> (5 passes), python3 bench.py %s; per-launch averages per kernel, sorted by busy cycles" % cmd)
print("# kernel | launches | VALU/MFMA | SALU/MFMA | MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 SQ_BUSY_CYCLES): the counter sums the SIMDs of an XCD (the deep trunk kernel, 1.23 PFLOP/s of f16 MFMA at the 2.0 GHz it sustains, reads 56 % this way) |"
      " LDS bank-conflict cycles / LDS active cycles | waves waiting on an instruction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) | sustained clock = GRBM_GUI_ACTIVE / 8 XCDs / launch duration")
rows = []
for k, d in acc.items():
    g = lambda c: d[c] / max(cnt[k][c], 1)
    mf = g("SQ_INSTS_MFMA")
    if mf <= 0:
        continue
    clk = (g("GRBM_GUI_ACTIVE") / 8 / (dur[k] / max(dcnt[k], 1))) if dur[k] > 0 else 0.0
    rows.append((g("SQ_BUSY_CYCLES"), "%-84s %4d  VALU/MFMA %6.2f  SALU/MFMA %5.2f  MFMA-busy %5.1f %%  bank-conflict %5.1f %%  wait %5.1f %%  clock %.2f GHz  avg %.1f us"
                 % (k, cnt[k]["SQ_INSTS_MFMA"], g("SQ_INSTS_VALU") / mf, g("SQ_INSTS_SALU") / mf,
                    100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / max(32 * g("SQ_BUSY_CYCLES"), 1), 100 * g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1),
                    100 * g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), clk, dur[k] / max(dcnt[k], 1) / 1e3)))
for _, line in sorted(rows, reverse=True):
    print(line)
PY
rm -rf $out/p1 $out/p2 $out/p3 $out/p4 $out/p5
cat $out.txt
