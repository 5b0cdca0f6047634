#!/bin/bash
# round 5, second session, GPU call 3: chunked lists (no atomics), LIST / FULL as two instantiations; tile-size variants
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/b3; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "visibility or pipeline_process_scan or fallback or render_variants or submap_paging or update_variants or loop_closure or model_image_differs or capacity" 2>&1 | tail -8 > "$O/pytest_sel.txt"
cat "$O/pytest_sel.txt"
B="python bench.py --cpu-scans 0 --adapter-scans 0 --no-host-vectors --profile-scans 40"
for round in 1 2; do
for cfg in nolists t128 t64 t256; do
  case $cfg in
    nolists) E="SUMA_NO_VIS_LISTS=1";;
    t128) E="SUMA_X=1";;
    t64) E="SUMA_HIP_LIB=$PWD/gpurun_in/lib_t64.so";;
    t256) E="SUMA_HIP_LIB=$PWD/gpurun_in/lib_t256.so";;
  esac
  env $E $B --kernels-json "$O/k_${cfg}_$round.json" 2>"$O/k_${cfg}_$round.err" | tail -1 > "$O/bench_${cfg}_$round.json"
done
done
python - <<PY
import json
for cfg in ("nolists","t128","t64","t256"):
    for r in (1,2):
        try:
            d=json.load(open("$O/bench_%s_%d.json"%(cfg,r)))
            k={x["name"]:round(x["avg_us"],1) for x in json.load(open("$O/k_%s_%d.json"%(cfg,r)))["kernels"]}
            print(cfg, r, round(d["value"],1), "k4k7", k.get("k4k7_render_indexmap"), "k4", k.get("k4_render_surfels"), "k9", k.get("k9_update_surfels"), "k10", k.get("k10_generate_surfels"), "visc", k.get("k_vis_compact"), d["visibility_lists"]["post_update_render"], d["visibility_lists"]["post_icp_render_and_index_map"], d["visibility_lists"]["render_launches_on_a_list"])
        except Exception as e:
            print(cfg, r, "failed", e)
PY
