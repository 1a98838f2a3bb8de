#!/bin/bash
# per-configuration kernel breakdown of the projecting models: tools/profile_hd.sh OUTDIR
set -e
out=$1
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$out
cd $GRAFT_REPO_ROOT
for cfg in "TransH wn 25 2" "TransD fb 25 8" "TransH wn 1 3" "TransH fb 25 8"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/$tag -o r -- python3 tools/run_hd_once.py $cfg 40 > $out/$tag.log 2>&1
  python tools/rocpd_stats.py $out/$tag/r_results.db $out/$tag.csv
  rm -rf $out/$tag
done
