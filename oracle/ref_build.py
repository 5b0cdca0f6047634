#!/usr/bin/env python3
"""oracle/ref_build.py -- TEST INFRASTRUCTURE: builds oracle/_ref/libsuma_ref*.so, "the reference compiled here".

The reference (PRBonn/semantic_suma) cannot be linked or run in this container: its host side needs an OpenGL
context, glow, Eigen, gtsam and Qt.  But the arithmetic of the hot path lives in plain-text GLSL
under /root/reference/src/shader/ and in src/core/lie_algebra.cpp.  This script

  1. READS those files where they lie (nothing is copied into the repository; the generated C++ goes to
     oracle/_ref/, which is git-ignored),
  2. turns each shader stage into a C++ namespace: resolves `#include`, drops `#version` / `layout(..)`,
     turns `uniform` / `in` / `out` declarations and interface blocks into plain globals, appends the `f` suffix
     to unsuffixed floating literals (GLSL 3.30 has no double), renames `main` -> `shader_main`,
  3. compiles the result with g++ against oracle/glsl_compat.hpp (the GLSL language + built-ins on the CPU),
     oracle/eigen_shim/ (the few Eigen types lie_algebra.cpp uses) and oracle/ref_driver.cpp (what the GL
     fixed-function pipeline does around the shaders: vertex fetch, point rasterisation, depth test, blending,
     transform-feedback capture), flags -O2 -ffp-contract=off,
  4. into two libraries:  libsuma_ref.so       transcendentals = include/suma_detmath.h (GL leaves them to the driver;
                                               with this choice the compiled shaders must agree with oracle/ bit for bit)
                          libsuma_ref_libm.so  transcendentals = glibc (agreement to a few ulp only).

Run:  python oracle/ref_build.py [--reference /root/reference] [--force]
Without /root/reference (e.g. on the GPU box) the script does nothing: the prebuilt libraries travel with the tree.
"""
import argparse
import hashlib
import itertools
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

# program -> shader stages (the glow programs of Preprocessing.cpp:30-60, Frame2Model.cpp:47-50,
# SurfelMap.cpp:128-156,173-176,230-233,256-260)
STAGES = [
    "gen_vertexmap.vert", "gen_vertexmap.frag", "avg_vertexmap.frag", "bilateral_filter.frag",
    "gen_normalmap.frag", "floodfill.frag",
    "Frame2Model_jacobians.vert", "Frame2Model_jacobians.geom", "Frame2Model_jacobians.frag",
    "render_surfels.vert", "render_surfels.geom", "render_surfels.frag", "render_compose.frag",
    "gen_indexmap.vert", "gen_indexmap.frag",
    "init_radiusConf.vert", "init_radiusConf.frag",
    "update_surfels.vert", "update_surfels.geom", "update_surfels.frag",
    "gen_surfels.vert", "gen_surfels.geom",
    "copy_surfels.vert", "copy_surfels.geom", "extract_surfels.vert",
]

KINDS = {"float": "K_FLOAT", "int": "K_INT", "bool": "K_BOOL", "vec2": "K_VEC2", "vec3": "K_VEC3", "vec4": "K_VEC4",
         "mat4": "K_MAT4", "sampler2DRect": "K_SAMPLER_RECT", "samplerBuffer": "K_SAMPLER_BUFFER"}


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def resolve_includes(src, src_root):
    def inc(m):
        with open(os.path.join(src_root, m.group(1))) as f:
            return strip_comments(f.read())
    return re.sub(r'^[ \t]*#include\s+"([^"]+)"[^\n]*$', inc, src, flags=re.M)


FLOAT_LIT = re.compile(r"(?<![\w.])((?:\d+\.\d+|\d+\.(?![A-Za-z_])|\.\d+)(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)(?![\w.])")


def translate(stage, src_root):
    """GLSL source text of one stage -> (C++ namespace text, uniforms, out-blocks)"""
    with open(os.path.join(src_root, "shader", stage)) as f:
        src = strip_comments(f.read())
    src = resolve_includes(src, src_root)
    src = re.sub(r"^[ \t]*#(version|pragma)[^\n]*$", "", src, flags=re.M)
    # geometry-shader primitive layouts
    src = re.sub(r"^[ \t]*layout\s*\([^)]*\)\s*(in|out)\s*;", "", src, flags=re.M)
    src = re.sub(r"layout\s*\([^)]*\)\s*", "", src)

    blocks = []

    def block(m):
        direction, name, body, inst, arr = m.group(1), m.group(2), m.group(3), m.group(4), m.group(5)
        members = re.findall(r"(\w+)\s+(\w+)\s*;", body)
        blocks.append((direction, name, inst, members))
        return "struct %s {%s};\n%s %s%s;" % (name, body, name, inst, "[1]" if arr else "")
    src = re.sub(r"^[ \t]*(in|out)\s+(\w+)\s*\{([^}]*)\}\s*(\w+)\s*(\[\s*\])?\s*;", block, src, flags=re.M)

    uniforms = []

    def uniform(m):
        uniforms.append((m.group(1), m.group(2)))
        return "%s %s;" % (m.group(1), m.group(2))
    src = re.sub(r"^[ \t]*uniform\s+(\w+)\s+(\w+)\s*;", uniform, src, flags=re.M)
    src = re.sub(r"^[ \t]*(?:flat\s+)?(?:in|out)\s+(\w+)\s+(\w+)\s*;", r"\1 \2;", src, flags=re.M)
    src = FLOAT_LIT.sub(lambda m: m.group(1) + "f", src)
    src = re.sub(r"\bvoid\s+main\s*\(\s*\)", "void shader_main()", src)
    src = re.sub(r"\bdiscard\s*;", "{ gl_discard = true; return; }", src)

    ns = "s_" + stage.replace(".", "_")
    pre = ("namespace %s {\n"
           "vec4 gl_Position; int gl_VertexID; bool gl_discard;\n"
           "struct gl_PerVertex_ { vec4 gl_Position; } gl_in[1];\n"
           "void (*emit_cb)() = nullptr;\n"
           "inline void EmitVertex() { if (emit_cb) emit_cb(); }\n"
           "inline void EndPrimitive() {}\n" % ns)
    post = ""
    for direction, name, inst, members in blocks:
        if direction == "out":
            post += "template <class D> inline void export_%s(D& dst) {\n" % inst
            post += "".join("  dst.%s = %s.%s;\n" % (m, inst, m) for _, m in members) + "}\n"
    return pre + src + "\n" + post + "}  // namespace %s\n" % ns, [(ns, stage, t, n) for t, n in uniforms]


def swizzle_includes():
    comps = {"vec2": 2, "vec3": 3, "vec4": 4}
    for vname, n in comps.items():
        lines = []
        for names in ("xyzw", "rgba"):
            for k in (2, 3, 4):
                for idx in itertools.product(range(n), repeat=k):
                    lines.append("swz%d<%s> %s;" % (k, ",".join(map(str, idx)), "".join(names[i] for i in idx)))
        with open(os.path.join(OUT, "glsl_swz_%s.inc" % vname), "w") as f:
            f.write("\n".join(lines) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("SUMA_REFERENCE_DIR", "/root/reference"))
    ap.add_argument("--force", action="store_true")
    args = ap.parse_args()
    src_root = os.path.join(args.reference, "src")
    if not os.path.isdir(os.path.join(src_root, "shader")):
        print("ref_build: %s not present -- keeping the prebuilt oracle/_ref libraries" % args.reference)
        return 0
    os.makedirs(OUT, exist_ok=True)

    inputs = [os.path.join(src_root, "shader", s) for s in STAGES]
    inputs += [os.path.join(src_root, "shader", g) for g in ("color.glsl", "color_map.glsl")]
    inputs += [os.path.join(src_root, "core", "lie_algebra.cpp"), os.path.join(src_root, "core", "lie_algebra.h")]
    inputs += [os.path.join(HERE, f) for f in ("glsl_compat.hpp", "ref_driver.cpp", "ref_build.py",
                                               os.path.join("eigen_shim", "eigen3", "Eigen", "Dense"))]
    inputs += [os.path.join(HERE, "..", "include", h) for h in ("suma_detmath.h", "suma_types.h")]
    h = hashlib.sha256()
    for p in inputs:
        with open(p, "rb") as f:
            h.update(f.read())
    stamp = os.path.join(OUT, "build.stamp")
    libs = [os.path.join(OUT, "libsuma_ref.so"), os.path.join(OUT, "libsuma_ref_libm.so")]
    if not args.force and all(os.path.exists(p) for p in libs + [stamp]) and open(stamp).read().strip() == h.hexdigest():
        return 0

    swizzle_includes()
    gen, registry = [], []
    for st in STAGES:
        text, uniforms = translate(st, src_root)
        gen.append(text)
        registry += uniforms
    with open(os.path.join(OUT, "shaders_gen.inc"), "w") as f:
        f.write("// GENERATED by oracle/ref_build.py from %s/shader -- not part of the repository\n" % src_root)
        f.write("namespace glsl {\n" + "\n".join(gen))
        f.write("struct uniform_entry { const char* stage; const char* name; int kind; void* ptr; };\n")
        f.write("static const uniform_entry uniform_table[] = {\n")
        for ns, stage, t, n in registry:
            if t not in KINDS:
                raise SystemExit("uniform type %s not handled (%s %s)" % (t, stage, n))
            f.write('  {"%s", "%s", %s, (void*)&%s::%s},\n' % (stage, n, KINDS[t], ns, n))
        f.write("  {nullptr, nullptr, 0, nullptr}};\n}  // namespace glsl\n")

    cxx = os.environ.get("CXX", "g++")
    fma = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []  # explicit fmas inline (the same bits as glibc's fmaf)
    flags = ["-O2", "-std=gnu++14", "-fPIC", "-shared", "-w", "-ffp-contract=off", "-fno-fast-math", "-fno-strict-aliasing", *fma,
             "-I", OUT, "-I", HERE, "-I", os.path.join(HERE, "eigen_shim"), "-I", os.path.join(src_root, "core"),
             "-I", src_root]
    srcs = [os.path.join(HERE, "ref_driver.cpp"), os.path.join(src_root, "core", "lie_algebra.cpp")]
    for lib, extra in ((libs[0], []), (libs[1], ["-DREF_USE_LIBM"])):
        cmd = [cxx] + flags + extra + ["-o", lib] + srcs + ["-lm"]
        print(" ".join(cmd))
        subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(h.hexdigest() + "\n")
    return 0


def build_gl_shim():
    """oracle/_ref/libsuma_glctx.so: the window-system-free GL context of oracle/glref.py (Mesa DRI headers needed;
    without them the GL-backed tests skip)"""
    src = os.path.join(HERE, "glref", "gl_ctx.c")
    lib = os.path.join(OUT, "libsuma_glctx.so")
    if not os.path.exists("/usr/include/GL/internal/dri_interface.h"):
        return
    if os.path.exists(lib) and os.path.getmtime(lib) >= os.path.getmtime(src):
        return
    os.makedirs(OUT, exist_ok=True)
    cmd = [os.environ.get("CC", "gcc"), "-O1", "-fPIC", "-shared", "-o", lib, src, "-ldl"]
    print(" ".join(cmd))
    subprocess.check_call(cmd)


if __name__ == "__main__":
    rc = main()
    build_gl_shim()
    sys.exit(rc)
