#!/bin/bash
# A/B of engine options on the bench workload: bash tools/ab_bench.sh OUTDIR OPTION v1 v2 ...
# per value: the bench line at the driver's arguments and the per-step kernel durations (early steps and steady state)
set -e
out=$1; opt=$2; shift 2
mkdir -p $GRAFT_REPO_ROOT/$out
for v in "$@"; do
  export KGE_OPT_${opt}=$v
  cd $GRAFT_REPO_ROOT
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench20_${opt}_$v.log 2>&1
  echo "== $opt=$v: $(grep -a '^{' $out/bench20_${opt}_$v.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f M/s  %.4f ms/step" % (d["value"]/1e6, d["ms_per_step"]))')"
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/kt_$v -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/kt_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  python3 tools/step_durations.py $out/kt_$v/r_results.db 5 9 | cut -c1-160
  python3 tools/step_durations.py $out/kt_$v/r_results.db 22 24 | tail -2 | cut -c1-160
  python3 tools/step_durations.py $out/kt_$v/r_results.db 150 152 | tail -2 | cut -c1-160
  rm -rf $out/kt_$v
done
