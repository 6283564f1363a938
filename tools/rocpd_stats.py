"""Per-kernel statistics from a rocprofv3 rocpd database (ROCm 7.2 default output) or
kernel_trace CSV: name, calls, avg/min/max duration (us), share of GPU kernel time.
usage: python tools/rocpd_stats.py <results.db | kernel_trace.csv>"""
import csv
import sqlite3
import sys


def from_db(path):
    cur = sqlite3.connect(path).cursor()
    return cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                       "from kernels group by name order by 6 desc").fetchall()


def from_csv(path):
    acc = {}
    for r in csv.DictReader(open(path)):
        d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        a = acc.setdefault(r["Kernel_Name"], [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    rows = [(k, v[0], v[1] / v[0], v[2], v[3], v[1]) for k, v in acc.items()]
    return sorted(rows, key=lambda r: -r[5])


rows = from_db(sys.argv[1]) if sys.argv[1].endswith(".db") else from_csv(sys.argv[1])
tot = sum(r[5] for r in rows) or 1.0
print(f"{'kernel':92s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share':>6s}")
for r in rows:
    print(f"{r[0][:92]:92s} {r[1]:6d} {r[2] / 1e3:9.2f} {r[3] / 1e3:9.2f} {r[4] / 1e3:9.2f} {100 * r[5] / tot:5.1f}%")
