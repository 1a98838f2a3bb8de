#!/bin/bash
# Kernel stats of a TransR step for several values of one engine option:
#   bash tools/sweep_option.sh OUTDIR OPTION "v1 v2 ..." NBATCHES [kernel-name-substring]
set -e
out=$1; opt=$2; vals=$3; nb=$4; pat=${5:-}
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$out
cd $GRAFT_REPO_ROOT
for v in $vals; do
  tag=${opt}_${v}_nb${nb}
  export KGE_OPT_${opt}=$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/kt_$tag -o r -- python3 tools/run_one.py TransR 200 1 SGD $nb 60 > $out/${tag}_kt.log 2>&1
  python3 tools/rocpd_stats.py $out/kt_$tag/r_results.db $out/${tag}_kernel_stats.csv
  rm -rf $out/kt_$tag
  python3 - $out/${tag}_kernel_stats.csv "$pat" "$tag" <<'PY'
import csv, sys
tot = 0
line = ""
for r in csv.DictReader(open(sys.argv[1])):
    n = int(r["Calls"])
    if n >= 55:
        tot += float(r["AverageNs"]) / 1e3 * n / 60
        if sys.argv[2] and sys.argv[2] in r["Name"]:
            line += " %s %.1f us" % (sys.argv[2], float(r["AverageNs"]) / 1e3)
print("%s: step %.1f us%s" % (sys.argv[3], tot, line))
PY
done
