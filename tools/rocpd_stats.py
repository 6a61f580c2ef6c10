#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average duration, share) of a rocprofv3 --kernel-trace run stored as a rocpd SQLite
database (the default output format of ROCm 7.2):   python tools/rocpd_stats.py results.db [out.csv]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute(f'select {name_col}, start, end from kernels').fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) * 1e-3
    tot = sum(a[1] for a in agg.values())
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    w = csv.writer(open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout)
    w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'percent'])
    for k, (n, t) in out:
        w.writerow([k, n, round(t, 1), round(t / n, 2), round(100 * t / tot, 2)])
    w.writerow(['TOTAL', sum(a[0] for a in agg.values()), round(tot, 1), '', 100.0])


if __name__ == '__main__':
    main()
