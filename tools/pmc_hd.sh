#!/bin/bash
# SQ counters of the pair-count path's two kernels (TransH, WN18RR-shaped, 25 negatives): tools/pmc_hd.sh OUTDIR
set -e
out=$1
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $out/pmc -o r -- python3 tools/run_hd_once.py TransH wn 25 2 6 > $out/pmc.log 2>&1
python3 - <<PY
import sqlite3
db = sqlite3.connect("$out/pmc/r_results.db")
rows = db.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
ker = {}
for kn, cn, v, n in rows:
    ker.setdefault(kn, {})[cn] = (v, n)
with open("$out/pair_path_sq_counters.txt", "w") as f:
    f.write("rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES\n")
    f.write("MI355X, python3 tools/run_hd_once.py TransH wn 25 2 6 (WN18RR-shaped, TransH dim 200, 25 neg/pos, B = 43 417, pair-count path), one pass\n\n")
    for kn, c in ker.items():
        if "pair_emit" not in kn and "segsum_pairs" not in kn:
            continue
        wc = c["SQ_WAVE_CYCLES"][0] / c["SQ_WAVE_CYCLES"][1]
        f.write(kn[:100] + "\n")
        for cn in sorted(c):
            v = c[cn][0] / c[cn][1]
            f.write("   %-22s %12.4g per launch  %6.1f %% of wave cycles\n" % (cn, v, 100 * v / wc))
        f.write("   VALU instructions per wave: %.0f\n\n" % ((c["SQ_INSTS_VALU"][0] / c["SQ_INSTS_VALU"][1]) / (c["SQ_WAVES"][0] / c["SQ_WAVES"][1])))
print(open("$out/pair_path_sq_counters.txt").read())
PY
rm -rf $out/pmc
