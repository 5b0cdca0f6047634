#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or *_kernel_trace.csv) per kernel:
launches, total / average / min / max duration.  Usage: rocprof_summary.py <file> [> profiles/xxx.txt]"""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    for name, start, end in cur.execute(f"select {name_col}, start, end from kernels"):
        yield name, (end - start)


def rows_from_csv(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            yield r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    agg = defaultdict(list)
    for name, ns in rows:
        agg[name.split("(")[0].replace("void ", "").split("<")[0]].append(ns)
    tot = sum(sum(v) for v in agg.values())
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"{'kernel':<34}{'calls':>8}{'total_ms':>11}{'avg_us':>10}{'min_us':>10}{'max_us':>10}{'share':>8}")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name[:33]:<34}{len(v):>8}{sum(v) / 1e6:>11.3f}{sum(v) / len(v) / 1e3:>10.2f}{min(v) / 1e3:>10.2f}"
              f"{max(v) / 1e3:>10.2f}{100.0 * sum(v) / tot:>7.1f}%")
    print(f"{'TOTAL':<34}{sum(len(v) for v in agg.values()):>8}{tot / 1e6:>11.3f}")


if __name__ == "__main__":
    main()
