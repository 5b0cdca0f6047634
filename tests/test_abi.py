"""C-ABI checks that need no GPU: the library loads, exports every symbol include/suma_hip.h declares,
the ctypes mirrors have the C layouts, and the product path fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()  # hipcc cross-compiles gfx950 without a GPU
    from semantic_suma_amd import core
    return core


def declared_functions(header="suma_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(suma_[a-z0-9_]+)\s*\(", src))
    return sorted(n for n in names if n not in ("suma_params_default",))


def test_every_declared_symbol_is_exported(built):
    L = C.CDLL(built.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/suma_hip.h but not exported: {missing}"
    assert b"gfx950" in C.c_char_p(C.cast(L.suma_version, C.CFUNCTYPE(C.c_char_p))()).value


def test_runner_header_is_exported_and_layouts_match(built, tmp_path):
    """include/suma_runner.h (native host loops for the replica configurations): symbols in libsuma_hip.so, struct
    layouts equal to the ctypes mirrors"""
    L = C.CDLL(built.LIB_PATH)
    names = declared_functions("suma_runner.h")
    assert {"suma_run_sequences", "suma_run_hypotheses"} <= set(names)
    assert not [n for n in names if not hasattr(L, n)]
    from semantic_suma_amd.core import HypothesisJob, LoopTrack, ScanRef, SequenceJob, SequenceResult
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "suma_runner.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                   "sizeof(suma_scan_ref),sizeof(suma_sequence_job),sizeof(suma_sequence_result),"
                   "sizeof(suma_hypothesis_job),sizeof(suma_loop_track));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(ScanRef), C.sizeof(SequenceJob), C.sizeof(SequenceResult), C.sizeof(HypothesisJob),
                     C.sizeof(LoopTrack)]


def test_dist_library_exports_its_header(built):
    """libsuma_hip_dist.so (the RCCL gather, kept out of libsuma_hip.so) exports what include/suma_hip_dist.h declares;
    libsuma_hip.so itself must not depend on RCCL"""
    path = os.path.join(os.path.dirname(built.LIB_PATH), "libsuma_hip_dist.so")
    L = C.CDLL(path)
    names = declared_functions("suma_hip_dist.h")
    assert {"suma_gather_poses", "suma_gather", "suma_dist_comm_create", "suma_dist_unique_id"} <= set(names)
    assert not [n for n in names if not hasattr(L, n)]
    needed = subprocess.check_output(["readelf", "-d", built.LIB_PATH]).decode()
    assert "rccl" not in needed


def test_ctypes_layouts_match_c(built, tmp_path):
    from semantic_suma_amd.core import IcpObjective, KernelTime, LoopResult
    from semantic_suma_amd.types import SURFEL_DTYPE, IcpStats, SumaParams
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "suma_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(suma_params),sizeof(suma_icp_stats),sizeof(suma_surfel),sizeof(suma_kernel_time),"
                   "offsetof(suma_params,max_surfels),offsetof(suma_params,cache_surfels),sizeof(suma_icp_objective),"
                   "sizeof(suma_loop_result),offsetof(suma_params,filter_sampling));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert sizes[0] == C.sizeof(SumaParams) and sizes[1] == C.sizeof(IcpStats)
    assert sizes[2] == SURFEL_DTYPE.itemsize == 64 and sizes[3] == C.sizeof(KernelTime)
    assert sizes[4] == SumaParams.max_surfels.offset and sizes[5] == SumaParams.cache_surfels.offset
    assert sizes[6] == C.sizeof(IcpObjective) and sizes[7] == C.sizeof(LoopResult)
    assert sizes[8] == SumaParams.filter_sampling.offset == C.sizeof(SumaParams) - 4


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the failure path cannot be exercised")
    from semantic_suma_amd.types import params_with_size
    with pytest.raises(built.SumaError, match="no HIP device|no CPU fallback"):
        built.Context(params_with_size(900))
    with pytest.raises(built.SumaError):
        built.SurfelMapping(params_with_size(900))


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "semantic_suma_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "libsuma_oracle" not in text and "oracle/" not in text, f
    for f in ("suma_hip.h", "suma_types.h", "suma_detmath.h", "suma_runner.h", "suma_adapter.hpp"):
        assert "ora_" not in open(os.path.join(ROOT, "include", f)).read()


def test_cpp_adapter_compiles_and_links(built, tmp_path):
    """include/suma_adapter.hpp (the reference-side binding of INTEGRATION.md) compiles as C++11 and links
    against libsuma_hip.so"""
    src = tmp_path / "a.cpp"
    src.write_text('#include "suma_adapter.hpp"\nint main(){ suma_params p; suma_params_default(&p);\n'
                   ' try { suma_hip::Context c(p, 0); suma_hip::SurfelMap m(c); return (int)m.size(); }\n'
                   ' catch (const std::runtime_error&) { return 42; } }\n')
    exe = tmp_path / "a.out"
    libdir = os.path.dirname(built.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lsuma_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    import torch
    rc = subprocess.call([str(exe)])
    assert rc == (0 if torch.cuda.is_available() else 42)  # throws like the reference when there is no device


def test_c_example_compiles_and_links(built, tmp_path):
    """examples/odometry.c: a plain C99 host program on the C-ABI; without arguments it prints its usage"""
    exe = tmp_path / "odometry"
    libdir = os.path.dirname(built.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-O1", "-Wall", "-Wextra", "-Werror", "-I",
                           os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "odometry.c"), "-o", str(exe),
                           "-L", libdir, "-lsuma_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.call([str(exe)], stderr=subprocess.DEVNULL) == 2


def test_gl_interop_recipe_compiles_links_and_fails_cleanly(built, tmp_path):
    """examples/gl_interop.cpp = the HIP -> GL hand-over of INTEGRATION.md 2a (hipGraphicsGLRegisterBuffer / Image, map,
    device-to-device copy, unmap): compiles against the ROCm headers, links libamdhip64 + libsuma_hip, and -- there
    being no GL context here -- reports the failed registration instead of crashing"""
    exe = tmp_path / "gl_interop"
    libdir = os.path.dirname(built.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I",
                           os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "gl_interop.cpp"), "-o", str(exe),
                           "-L", libdir, "-lsuma_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "without a GL context: hipError" in out.stdout
