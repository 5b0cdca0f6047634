#!/usr/bin/env python
"""Timeline analysis of a rocprofv3 --kernel-trace CSV: per scan, GPU busy time (union of kernel intervals over all
streams), idle time, and the largest idle gaps with the kernels around them.
usage: trace_gaps.py <kernel_trace.csv> [first_scan last_scan]"""
import csv
import sys


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0], r.get("Queue_Id", "")))
    rows.sort()
    # scans are delimited by k1_scatter launches
    starts = [i for i, r in enumerate(rows) if r[2] == "k1_scatter"]
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else min(len(starts) - 2, lo + 20)
    t0, t1 = rows[starts[lo]][0], rows[starts[hi]][0]
    sel = [r for r in rows if t0 <= r[0] < t1]
    n = hi - lo
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    for s, e, name, q in sel:
        if cur_e is None:
            cur_s, cur_e, last = s, e, name
            continue
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last, name))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        last = name if e >= cur_e else last
    busy += cur_e - cur_s
    span = t1 - t0
    print(f"scans {lo}..{hi}: {span / n / 1e3:.1f} us per scan, GPU busy (union) {busy / n / 1e3:.1f} us, idle {(span - busy) / n / 1e3:.1f} us")
    per = {}
    for g, a, b in gaps:
        k = f"{a} -> {b}"
        per.setdefault(k, []).append(g)
    print("idle gaps by (kernel before -> kernel after), per scan:")
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print(f"  {k:<50}{len(v) / n:>6.1f} x {sum(v) / len(v) / 1e3:>6.2f} us = {sum(v) / n / 1e3:>6.1f} us")
    dur = {}
    for s, e, name, q in sel:
        dur.setdefault(name, []).append(e - s)
    print("kernel time per scan:")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {k:<28}{len(v) / n:>6.1f} x {sum(v) / len(v) / 1e3:>7.2f} us = {sum(v) / n / 1e3:>6.1f} us")


if __name__ == "__main__":
    main()
