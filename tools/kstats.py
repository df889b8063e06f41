"""Print a rocprofv3 *kernel_stats.csv as a short table.  usage: python tools/kstats.py <kernel_stats.csv> [rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n_rows]:
    n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Name"])[:60]
    print("%-60s %5d %9.1f us %5.1f%%" % (n, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / tot * 100))
