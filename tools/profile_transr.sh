#!/bin/bash
# kernel breakdown of one TransR step: tools/profile_transr.sh OUTDIR [nbatches]
set -e
out=$1; nb=${2:-8}
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$out
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/t -o r -- python3 tools/run_one_config.py TransR fb 200 1 $nb 40 > $out/transr_$nb.log 2>&1
python tools/rocpd_stats.py $out/t/r_results.db $out/transr_$nb.csv
rm -rf $out/t
