#!/usr/bin/env python
"""SQ counter summary per kernel from rocprofv3 --pmc passes (any number of *_counter_collection.csv files, one pass
each): per-launch means and the ratios that say what a kernel is bound by.
  valu_busy   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   share of wave-resident quad-cycles in which a VALU op of the wave issues
  wait_any    = SQ_WAIT_ANY / SQ_WAVE_CYCLES           wave parked on s_waitcnt / barrier (memory or LDS latency exposed)
  wait_inst   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES      ready to issue but the pipe is taken (issue-bound)
  active_any  = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_conflict= SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
(units: quad-cycles, MI355X_MICROARCH.md; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)
usage: sq_summary.py a.csv [b.csv ...]"""
import collections
import csv
import sys

WANT = ("k_icp_step", "k_icp_finish", "k_render", "k9_update", "k10_generate", "k1_scatter", "k23_normals_labels",
        "k_resolve_compose", "k_resolve", "k12_extract")


def main():
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[1:]:
        try:
            rows = list(csv.DictReader(open(path)))
        except OSError:
            continue
        for r in rows:
            by[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("# per-launch means of SQ counters (rocprofv3 --pmc, separate passes of `python bench.py --steps 30`), quad-cycle units")
    for k in WANT:
        if k not in by:
            continue
        m = {c: sum(v) / len(v) for c, v in by[k].items()}
        n = max(len(v) for v in by[k].values())
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        print(f"\n{k}  ({n} launches)")
        for c in sorted(m):
            print(f"  {c:<26}{m[c]:>16.0f}")
        if wc:
            r = lambda c: m.get(c, 0.0) / wc  # noqa: E731
            print(f"  -> valu_busy {r('SQ_ACTIVE_INST_VALU'):.3f}  active_any {r('SQ_ACTIVE_INST_ANY'):.3f}  "
                  f"wait_any {r('SQ_WAIT_ANY'):.3f}  wait_inst {r('SQ_WAIT_INST_ANY'):.3f}")
        if m.get("SQ_WAVES") and m.get("SQ_INSTS_VALU"):
            print(f"  -> VALU instructions per wave {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f}, "
                  f"SALU {m.get('SQ_INSTS_SALU', 0) / m['SQ_WAVES']:.0f}, LDS {m.get('SQ_INSTS_LDS', 0) / m['SQ_WAVES']:.0f}, "
                  f"VMEM rd {m.get('SQ_INSTS_VMEM_RD', 0) / m['SQ_WAVES']:.0f} wr {m.get('SQ_INSTS_VMEM_WR', 0) / m['SQ_WAVES']:.0f}")
        if m.get("SQ_LDS_IDX_ACTIVE"):
            print(f"  -> lds_conflict {m.get('SQ_LDS_BANK_CONFLICT', 0.0) / m['SQ_LDS_IDX_ACTIVE']:.3f}")


if __name__ == "__main__":
    main()
