#!/bin/bash
# A/B of an engine option on one training configuration: per value the step time and the per-kernel averages (kernel trace)
#   bash tools/ab_config.sh OUTDIR OPTION "v1 v2 ..." MODEL DIM NEG OPT NBATCHES STEPS [wn]
set -e
out=$1; opt=$2; vals=$3; shift 3
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in $vals; do
  export KGE_OPT_${opt}=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt_$v -o r -- python3 tools/run_one.py "$@" > $out/kt_${opt}_$v.log 2>&1
  python3 tools/rocpd_stats.py $out/kt_$v/r_results.db $out/stats_${opt}_$v.csv
  rm -rf $out/kt_$v
  echo "== $opt=$v"
  python3 - $out/stats_${opt}_$v.csv "$6" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
tot = 0.0
for r in rows:
    per_step = float(r["TotalDurationNs"]) / steps / 1e3
    if int(r["Calls"]) >= steps - 1:
        tot += per_step
        if per_step >= 1.0:
            print("   %-70s %7.1f us" % (r["Name"][:70], float(r["AverageNs"]) / 1e3))
print("   kernels per step: %.1f us" % tot)
PY
done
