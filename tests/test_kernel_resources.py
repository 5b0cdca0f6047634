"""Register budgets of the hot kernels, checked on the build machine (hipcc -S, no GPU): an innocent-looking edit can
push a kernel over the VGPR count that its occupancy depends on, and no parity test notices.  Round 6 did exactly that:
a 16-float LDS hand-over took k_icp_step from 123 to 129 VGPRs -- one chain got 1 % faster, but two 512-thread blocks no
longer shared a CU and the batched chains (BASELINE configs[2]) lost 19 % until the static table showed it.

Budgets (512 VGPRs per SIMD lane, waves per SIMD = blocks per CU x waves per block / 4):
  k_icp_step     512-thread blocks, TWO per CU when chains are batched  -> 4 waves per SIMD -> <= 128
  k9_update      one 1024-thread block per CU                           -> 4 waves per SIMD -> <= 128
  k10 / k12      1024-thread blocks                                     -> <= 128
  k_render       five 256-thread blocks per CU (amdgpu_waves_per_eu 5)  -> 5 waves per SIMD -> <= 96 (512 / 5 = 102, granule 8)
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUDGET = [(r"^k_icp_step$", 128), (r"^k_icp_finish$", 128), (r"k9_update", 128), (r"k10_generate", 128), (r"k12_extract", 128),
          (r"k_render", 96)]


def test_vgpr_budgets_of_the_hot_kernels():
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_stats.py"), "k_icp.hip", "k_render.hip", "k_update.hip"],
                             capture_output=True, text=True, timeout=900, env=dict(os.environ, PATH="/opt/rocm/bin:" + os.environ.get("PATH", "")))
    except FileNotFoundError:
        pytest.skip("no hipcc")
    assert out.returncode == 0, out.stderr[-2000:]
    seen = {}
    for line in out.stdout.splitlines()[1:]:
        m = re.match(r"^(?:void )?(\S+(?:<\w+>)?)\s+(\d+)\s+(\d+)\s+\|", line)
        if m:
            seen[m.group(1)] = int(m.group(2))
    assert len(seen) >= 10, out.stdout
    for pat, limit in BUDGET:
        hits = {k: v for k, v in seen.items() if re.search(pat, k)}
        assert hits, f"no kernel matches {pat}: {sorted(seen)}"
        for k, v in hits.items():
            assert v <= limit, f"{k}: {v} VGPRs, budget {limit} (see the module docstring: occupancy depends on it)"
