"""rocprofv3 (rocpd sqlite output) -> per-kernel stats CSV, the same columns `--stats` prints.
usage: python tools/rocpd_stats.py results.db out.csv [name-substring-filter ...]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels group by name order by 3 desc").fetchall()
    keep = sys.argv[3:]
    if keep:
        rows = [r for r in rows if any(k in r[0] for k in keep)]
    total = float(sum(r[2] for r in rows)) or 1.0
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, n, tot, avg, lo, hi in rows:
            w.writerow([name, n, int(tot), "%.1f" % avg, "%.3f" % (100.0 * tot / total), int(lo), int(hi)])


if __name__ == "__main__":
    main()
