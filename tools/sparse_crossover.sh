#!/bin/bash
# Dense (count image + full-table sweep) against sparse rows (touched rows only) for TransE + SGD on ONE GPU, table sizes around
# the automatic switch (Config.sparse_threshold_bytes): bash tools/sparse_crossover.sh OUT.jsonl
# dim 512, B = 131072 positives x 1 negative (the per-GPU batch of BASELINE config #5 on 8 GPUs), uniform entity popularity.
set -e
out=$GRAFT_REPO_ROOT/$1
: > $out
for ents in 125000 250000 500000 1000000 2000000 4000000; do
  for mode in "" "--dense"; do
    python3 tools/bench_sparse.py --entities $ents --relations 1000 --triples 4000000 --dim 512 --batch 131072 --neg 1 --steps 30 --ent-exponent 0 $mode 2>/dev/null | grep -a '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps(dict(entities=$ents, table_GB=$ents * 512 * 4 / 1e9, mode='dense' if '$mode' else 'sparse rows', ms_per_step=d['ms_per_step'], positives_per_s=d['positives_per_s'])))" | tee -a $out
  done
done
