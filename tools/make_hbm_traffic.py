#!/usr/bin/env python
"""profiles/hbm_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE
--pmc runs, as MI355X_MICROARCH.md prescribes: they do not fit one pass).  Units: both counters are in KB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced stream, so it is
doubled; WRITE_SIZE is used as reported (uncalibrated by the guide).
usage: make_hbm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <width> <height> <out.json>"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_suma_amd.buildinfo import kernel_source_sha  # noqa: E402


def per_kernel(path, counter):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            by[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
    return by


def main():
    f, w, width, height, out = sys.argv[1:6]
    F, Wr = per_kernel(f, "FETCH_SIZE"), per_kernel(w, "WRITE_SIZE")
    res = {}
    for k in sorted(set(F) | set(Wr)):
        fv, wv = F.get(k, [0.0]), Wr.get(k, [0.0])
        fetch_kb, write_kb = sum(fv) / len(fv), sum(wv) / len(wv)
        res[k] = {"launches": len(fv), "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb,
                  "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0}
    json.dump({"width": int(width), "height": int(height), "kernel_source_sha": kernel_source_sha(),
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python bench.py`; "
                         "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE half-count correction)",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print(f"{k[:30]:<31}{v['launches']:>6}{v['fetch_size_kb_raw']:>12.0f}{v['write_size_kb']:>12.0f}{v['hbm_bytes_per_launch'] / 1e6:>12.2f} MB")


if __name__ == "__main__":
    main()
