#!/bin/bash
# TransR step with the row GEMMs on the bf16 matrix pipe (three-term split) and on the fp32 MFMA, at the reference's batch
# (nbatches 0 -> B = 2 721) and at B = 34 014: kernel stats of each (rocprofv3 --kernel-trace).   bash tools/ab_transr.sh OUTDIR
set -e
out=$1
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$out
cd $GRAFT_REPO_ROOT
for nb in 0 8; do
  for x3 in 1 0; do
    tag=transr_nb${nb}_bf16x3_${x3}
    export KGE_OPT_TRANSR_BF16X3=$x3
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/kt_$tag -o r -- python3 tools/run_one.py TransR 200 1 SGD $nb 60 > $out/${tag}_kt.log 2>&1
    python3 tools/rocpd_stats.py $out/kt_$tag/r_results.db $out/${tag}_kernel_stats.csv
    rm -rf $out/kt_$tag
    python3 - $out/${tag}_kernel_stats.csv <<'PY'
import csv, sys
print("==", sys.argv[1])
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = int(r["Calls"])
    if n >= 55:
        print("  %-64s %4d %9.1f us %6.2f%%" % (r["Name"].replace("kge::(anonymous namespace)::", "")[:64], n, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        tot += float(r["AverageNs"]) / 1e3 * n / 60
print("  sum per step %.1f us" % tot)
PY
  done
done
