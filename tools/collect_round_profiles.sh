#!/bin/bash
# End-of-round evidence for profiles/: bench lines, kernel stats of the bench command, PMC traffic of the emit kernel.
# usage (on the GPU box): bash tools/collect_round_profiles.sh gpurun_out/final
set -e
out=$1
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$out
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench_default.log 2>&1; grep -a "^{" $out/bench_default.log | tail -1 > $out/bench_line_default.json
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_steps50.log 2>&1; grep -a "^{" $out/bench_steps50.log | tail -1 > $out/bench_line_steps50.json
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_steps20.log 2>&1; grep -a "^{" $out/bench_steps20.log | tail -1 > $out/bench_line_steps20.json
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/kt.log 2>&1
python3 tools/rocpd_stats.py $out/kt/r_results.db $out/kernel_stats_bench.csv
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/pmc_write.log 2>&1
python3 - <<PY
import sqlite3, json
res = {}
for name, d in (("FETCH_SIZE", "$out/pmc_fetch/r_results.db"), ("WRITE_SIZE", "$out/pmc_write/r_results.db")):
    db = sqlite3.connect(d)
    # one row per (dispatch, counter instance): sum the instances of a dispatch, then average over the emit kernel's dispatches
    rows = db.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? group by kernel_name, dispatch_id",
                      (name,)).fetchall()
    per = [v for kn, _, v in rows if "transe_emit_vec_kernel" in kn]
    kn = [kn for kn, _, _ in rows if "transe_emit_vec_kernel" in kn][0].split("(")[0]
    res[name] = dict(kernel=kn, launches=len(per), KB_per_launch=sum(per) / len(per))
print(json.dumps(res))
open("$out/pmc_emit_traffic.json", "w").write(json.dumps(res, indent=1))
PY
rm -rf $out/kt $out/pmc_fetch $out/pmc_write
