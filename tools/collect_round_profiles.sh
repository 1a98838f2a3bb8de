#!/bin/bash
# End-of-round evidence for profiles/: bench lines, kernel stats and per-step durations of the bench command, PMC traffic of the
# emit kernel, TransR kernel stats + SQ / MFMA counters at both batch sizes, config #5 kernel stats.
# usage (on the GPU box): bash tools/collect_round_profiles.sh gpurun_out/final
set -e
out=$1
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$out
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench_default.log 2>&1; grep -a "^{" $out/bench_default.log | tail -1 > $out/bench_line_default.json
python3 bench.py --steps 20 --warmup 5 > $out/bench_steps20.log 2>&1; grep -a "^{" $out/bench_steps20.log | tail -1 > $out/bench_line_steps20.json
echo "bench lines done"
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/kt.log 2>&1
python3 tools/rocpd_stats.py $out/kt/r_results.db $out/kernel_stats_bench.csv
python3 tools/step_durations.py $out/kt/r_results.db 0 30 > $out/step_durations_bench.txt
python3 tools/step_durations.py $out/kt/r_results.db 150 156 >> $out/step_durations_bench.txt
rm -rf $out/kt
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
rm -rf $out/pmc_fetch $out/pmc_write
echo "bench profiles done"
bash tools/profile_config.sh $out transr_b34014 TransR 200 1 SGD 8 30 > /dev/null 2>&1
bash tools/profile_config.sh $out transr_b2721 TransR 200 1 SGD 0 60 > /dev/null 2>&1
echo "transr profiles done"
python3 tools/measure_configs.py > $out/other_configs.jsonl 2>/dev/null
echo "other configs done"
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/kt5 -o r -- python3 $GRAFT_REPO_ROOT/tools/bench_sparse.py --entities 50000000 --relations 1000 --triples 20000000 --dim 512 --batch 1050420 --neg 1 --steps 8 --ent-exponent 0 > $GRAFT_REPO_ROOT/$out/config5.log 2>&1
cd $GRAFT_REPO_ROOT
grep -a "^{" $out/config5.log | tail -1 > $out/config5_one_gpu.json
python3 tools/rocpd_stats.py $out/kt5/r_results.db $out/kernel_stats_config5.csv
rm -rf $out/kt5
echo "config5 done"
cd /tmp
timeout -k 10 600 python3 $GRAFT_REPO_ROOT/tools/bench_sparse.py --entities 50000000 --relations 1000 --triples 500000000 --dim 512 --batch 1050420 --neg 1 --steps 8 --ent-exponent 0 > $GRAFT_REPO_ROOT/$out/config5_500M.log 2>&1 || true
cd $GRAFT_REPO_ROOT
grep -a "^{" $out/config5_500M.log | tail -1 > $out/config5_one_gpu_500M_triples.json
echo "config5 500M done"
