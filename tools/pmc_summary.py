#!/usr/bin/env python
"""Per-kernel means of every counter in a rocprofv3 *_counter_collection.csv.
usage: pmc_summary.py <counter_collection.csv> [out.txt]"""
import collections
import csv
import sys


def main():
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        by[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for k in by.values() for c in k})
    lines = [f"{'kernel':<28}{'n':>6}" + "".join(f"{c[:18]:>20}" for c in names)]
    for k, cs in sorted(by.items()):
        n = max(len(v) for v in cs.values())
        lines.append(f"{k[:27]:<28}{n:>6}" + "".join(f"{(sum(cs[c]) / len(cs[c]) if cs.get(c) else 0):>20.0f}" for c in names))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
