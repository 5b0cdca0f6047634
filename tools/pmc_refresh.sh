#!/bin/bash
# re-takes what depends on the exact kernel sources (bench.py reports roofline.traffic only for PMC passes taken on the
# sources it runs): the five --pmc passes and, with the refreshed profiles/hbm_traffic.json, nothing else.
#   gpurun --timeout 1500 -- 'bash tools/pmc_refresh.sh r03'   then   bash tools/collect_profiles.sh r03 (PMC part)
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$TAG; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
B="python bench.py --cpu-scans 0 --no-kernel-events --adapter-scans 0 --no-loop-closure --no-reference-mode"
$B --steps 30 2>/dev/null | tail -1 | cut -c1-100
for c in "pmc_fetch f FETCH_SIZE" "pmc_write w WRITE_SIZE" \
         "pmc_sq1 s SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" \
         "pmc_sq2 s SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" \
         "pmc_sq3 s SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
  set -- $c; d=$1; o=$2; shift 2
  timeout 500 rocprofv3 --pmc "$@" -d $O/$d -o $o --output-format csv -- $B --steps 30 > $O/$d.log 2>&1
  echo "$d: $(wc -l < $O/$d/${o}_counter_collection.csv) rows; $(grep -c Traceback $O/$d.log) errors"
done
