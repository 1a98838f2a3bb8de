#!/bin/bash
# Kernel stats + SQ counters of one training configuration, for profiles/:
#   bash tools/profile_config.sh OUTDIR TAG MODEL DIM NEG OPT NBATCHES STEPS [wn]
# writes OUTDIR/TAG_kernel_stats.csv (rocprofv3 --kernel-trace) and OUTDIR/TAG_sq_counters.txt (one --pmc pass: wave cycles,
# wait / issue-stall / active shares, MFMA busy cycles, LDS bank conflicts), kernels that take >= 1 % of the GPU time.
set -e
out=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt_$tag -o r -- python3 tools/run_one.py "$@" > $out/${tag}_kt.log 2>&1
python3 tools/rocpd_stats.py $out/kt_$tag/r_results.db $out/${tag}_kernel_stats.csv
rm -rf $out/kt_$tag
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $out/pmc_$tag -o r -- python3 tools/run_one.py "$@" > $out/${tag}_pmc.log 2>&1
python3 - "$out" "$tag" "$*" <<'PY'
import csv, sqlite3, sys
out, tag, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
db = sqlite3.connect("%s/pmc_%s/r_results.db" % (out, tag))
rows = db.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
ker = {}
for kn, cn, v, n in rows:
    ker.setdefault(kn, {})[cn] = v / max(n, 1)
stats = {r["Name"]: r for r in csv.DictReader(open("%s/%s_kernel_stats.csv" % (out, tag)))}
with open("%s/%s_sq_counters.txt" % (out, tag), "w") as f:
    f.write("rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES\n")
    f.write("MI355X, python3 tools/run_one.py %s ; per launch; shares are of SQ_WAVE_CYCLES (quad-cycles summed over waves); avg_us from the kernel trace of the same command\n\n" % cmd)
    f.write("%-64s %8s %8s %8s %8s %14s %12s %9s\n" % ("kernel", "avg_us", "wait%", "stall%", "active%", "mfma_busy_cyc", "lds_conf%", "waves"))
    for kn, c in sorted(ker.items(), key=lambda kv: -float(stats.get(kv[0], {}).get("TotalDurationNs", 0))):
        st = stats.get(kn)
        if not st or float(st["Percentage"]) < 1.0:
            continue
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        lds = c.get("SQ_LDS_IDX_ACTIVE", 0) or 1
        f.write("%-64s %8.1f %8.1f %8.1f %8.1f %14.4g %12.1f %9.0f\n" % (kn[:64], float(st["AverageNs"]) / 1e3, 100 * c.get("SQ_WAIT_ANY", 0) / wc,
                100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0),
                100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / lds, c.get("SQ_WAVES", 0)))
print(open("%s/%s_sq_counters.txt" % (out, tag)).read())
PY
rm -rf $out/pmc_$tag
