#!/bin/bash
# Per-kernel SQ / TA counters of the bench command (driver arguments): bash tools/pmc_bench.sh OUTDIR [kernel-name-substring ...]
# Two separate --pmc passes (no trace domains beside them), summed over counter instances, averaged over a kernel's launches.
set -e
out=$1; shift
pat="${@:-segapply emit_vec bkt_ apply_counts}"
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVES -d $out/pmcA -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_BUSY_CYCLES -d $out/pmcB -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/pmcB.log 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE -d $out/pmcC -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/pmcC.log 2>&1 || true
python3 - "$out" $pat <<'PY'
import sqlite3, sys, os
out, pats = sys.argv[1], sys.argv[2:]
ker = {}
for sub in ("pmcA", "pmcB", "pmcC"):
    path = os.path.join(out, sub, "r_results.db")
    if not os.path.exists(path):
        continue
    db = sqlite3.connect(path)
    try:
        rows = db.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(sub, "failed:", e); continue
    for kn, cn, v, n in rows:
        ker.setdefault(kn, {})[cn] = v / max(n, 1)
with open(os.path.join(out, "sq_counters.txt"), "w") as f:
    for kn, c in ker.items():
        if not any(p in kn for p in pats):
            continue
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        waves = c.get("SQ_WAVES", 1) or 1
        f.write(kn[:110] + "\n")
        for cn in sorted(c):
            v = c[cn]
            extra = "  %6.1f %% of wave cycles" % (100 * v / wc) if cn.startswith(("SQ_WAIT", "SQ_ACTIVE")) else ("  %8.1f per wave" % (v / waves) if cn.startswith("SQ_INSTS") else "")
            f.write("   %-34s %14.5g per launch%s\n" % (cn, v, extra))
        f.write("\n")
print(open(os.path.join(out, "sq_counters.txt")).read())
PY
rm -rf $out/pmcA $out/pmcB $out/pmcC
