#!/bin/bash
# TransR step, fused relation-tile kernel on / off, at the reference's batch (nbatches 0 -> B = 2 721) and at B = 34 014:
# kernel stats of each (rocprofv3 --kernel-trace).   bash tools/ab_transr.sh OUTDIR
set -e
out=$1
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$out
cd $GRAFT_REPO_ROOT
for nb in 0 8; do
  for fused in 1 0; do
    tag=transr_nb${nb}_fused${fused}
    export KGE_OPT_TRANSR_FUSED=$fused
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/kt_$tag -o r -- python3 tools/run_one.py TransR 200 1 SGD $nb 60 > $out/${tag}_kt.log 2>&1
    python3 tools/rocpd_stats.py $out/kt_$tag/r_results.db $out/${tag}_kernel_stats.csv
    rm -rf $out/kt_$tag
    echo "== $tag"; cut -d, -f1,2,4,5 $out/${tag}_kernel_stats.csv | cut -c1-150 | sed -n '1,14p'
  done
done
