import ctypes, subprocess, sys, os
sys.argv = ["bench.py", "--steps", "60", "--warmup", "5", "--cpu-scans", "0", "--no-kernel-events"]
import runpy
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
from semantic_suma_amd import core
core.lib().suma_dbg_dump_render()
