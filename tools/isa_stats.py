#!/usr/bin/env python
"""Static ISA statistics of the gfx950 kernels (no GPU needed): per kernel, and for its largest loop body, the number of
VALU instructions, of v_readlane / v_writelane (SGPR spill traffic when the register report says "SGPRs Spill"), hazard
nops, IEEE division sequences (v_div_fixup), scalar and vector memory instructions, and the register report of the
compiler.  This is what found K9's spilled arguments.
    python tools/isa_stats.py [file.hip ...] [-D...]        (default: the four kernel files)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.environ.get("SUMA_CSRC", os.path.join(ROOT, "semantic_suma_amd", "csrc"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mllvm",
         "-amdgpu-kernarg-preload-count=16", "-w"]


def count(pat, lines):
    return sum(1 for l in lines if re.search(pat, l))


def stats(lines):
    return dict(valu=count(r"^\s+v_", lines), readlane=count(r"v_readlane", lines), writelane=count(r"v_writelane", lines),
                nop=count(r"^\s+s_nop", lines), div=count(r"v_div_fixup", lines), smem=count(r"^\s+s_load", lines),
                vmem=count(r"^\s+(global|flat|buffer)_(load|store|atomic)", lines), lds=count(r"^\s+ds_", lines),
                barrier=count(r"s_barrier", lines))


def largest_loop(lines):
    lab = {}
    for n, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab[m.group(1)] = n
    best = None
    for n, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < n:
            span = (lab[m.group(1)], n)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    return lines[best[0]:best[1]] if best else []


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-")]
    files = [a for a in sys.argv[1:] if not a.startswith("-")] or ["k_icp.hip", "k_render.hip", "k_update.hip", "k_preprocess.hip"]
    print(f"{'kernel':44s} {'VGPR':>4s} {'SGPRspill':>9s} | {'VALU':>5s} {'rdlane':>6s} {'wrlane':>6s} {'s_nop':>5s} {'div':>4s} "
          f"{'smem':>4s} {'vmem':>4s} {'lds':>4s} | largest loop: {'VALU':>5s} {'rdlane':>6s} {'s_nop':>5s} {'div':>4s} {'smem':>4s}")
    for f in files:
        src = f if os.path.exists(f) else os.path.join(CSRC, f)
        with tempfile.TemporaryDirectory() as d:
            asm = os.path.join(d, "k.s")
            subprocess.check_call(["hipcc", *FLAGS, *defs, "--cuda-device-only", "-S", src, "-o", asm], cwd=CSRC,
                                  stderr=subprocess.DEVNULL)
            text = open(asm).read()
        for m in re.finditer(r"^(_Z\w+):[^\n]*\n", text, re.M):
            name = m.group(1)
            # the whole function: a kernel with early exits has several s_endpgm (round 4 cut at the first one, which
            # under-counted k_icp_step / k_icp_finish)
            fe = re.search(r"^\.Lfunc_end\d+:", text[m.end():], re.M)
            end = m.end() + fe.start() if fe else text.find("s_endpgm", m.end())
            if end < 0 or ".amdhsa_kernel " + name not in text:
                continue
            body = text[m.end():end].split("\n")
            md = re.search(r"\.name:\s+" + re.escape(name) + r"\n(.*?)\.vgpr_count:\s+(\d+)", text, re.S)
            vg = md.group(2) if md else "?"
            spm = re.search(r"\.sgpr_spill_count:\s+(\d+)", md.group(1)) if md else None
            sp = spm.group(1) if spm else "?"
            a, l = stats(body), stats(largest_loop(body))
            short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0][:44]
            print(f"{short:44s} {vg:>4s} {sp:>9s} | {a['valu']:5d} "
                  f"{a['readlane']:6d} {a['writelane']:6d} {a['nop']:5d} {a['div']:4d} {a['smem']:4d} {a['vmem']:4d} {a['lds']:4d} | "
                  f"{'':13s} {l['valu']:5d} {l['readlane']:6d} {l['nop']:5d} {l['div']:4d} {l['smem']:4d}")


if __name__ == "__main__":
    main()
