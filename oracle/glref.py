"""oracle/glref.py -- TEST INFRASTRUCTURE: the reference's own GLSL executed by a real OpenGL implementation.

Mesa's llvmpipe (swrast_dri.so, OpenGL 4.5 core) is driven without a window system through oracle/glref/gl_ctx.c; this
module loads the shader sources of PRBonn/semantic_suma from /root/reference/src/shader (read where they lie, nothing is
copied) and replays the GL calls the reference's host code makes around them (file:line cited per pass), so that the
FIXED-FUNCTION half of the passes -- point and triangle rasterisation, the depth test on a DEPTH24_STENCIL8
renderbuffer, transform-feedback order, rectangle-texture fetches -- is the behaviour of a conformant GL and not the
builder's reading of the specification (oracle/ref_driver.cpp).  Used by tests/test_gl_reference.py and
tests/golden/make_gl_golden.py only; the product never imports it.

What a GL driver is free to choose (the last ulp of atan / asin / sin, fused multiply-adds, the precision of attribute
interpolation) differs between llvmpipe, the vendor driver the reference was developed on, and include/suma_detmath.h;
comparisons against this module therefore come in two kinds: EXACT where only fixed-function rules are involved (own
pass-through shaders fed with the oracle's vertices), and STATISTICAL (fraction of texels that agree) where the
reference's shaders compute the vertices themselves.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SHADER_DIR = os.environ.get("SUMA_REFERENCE_SHADERS", "/root/reference/src/shader")
LIB = os.path.join(HERE, "_ref", "libsuma_glctx.so")

# --- GL enums used below
GL = dict(
    VERSION=0x1F02, RENDERER=0x1F01, VERTEX_SHADER=0x8B31, GEOMETRY_SHADER=0x8DD9, FRAGMENT_SHADER=0x8B30,
    COMPILE_STATUS=0x8B81, LINK_STATUS=0x8B82, INFO_LOG_LENGTH=0x8B84, ARRAY_BUFFER=0x8892, STATIC_DRAW=0x88E4,
    DYNAMIC_COPY=0x88EA, FLOAT=0x1406, INT=0x1404, UNSIGNED_INT=0x1405, TEXTURE_RECTANGLE=0x84F5, TEXTURE_BUFFER=0x8C2A,
    RGBA32F=0x8814, RGBA=0x1908, R32F=0x822E, RED=0x1903, TEXTURE0=0x84C0, TEXTURE_MIN_FILTER=0x2801,
    TEXTURE_MAG_FILTER=0x2800, TEXTURE_WRAP_S=0x2802, TEXTURE_WRAP_T=0x2803, NEAREST=0x2600, LINEAR=0x2601,
    CLAMP_TO_BORDER=0x812D, FRAMEBUFFER=0x8D40, RENDERBUFFER=0x8D41, COLOR_ATTACHMENT0=0x8CE0,
    DEPTH_STENCIL_ATTACHMENT=0x821A, DEPTH24_STENCIL8=0x88F0, DEPTH_COMPONENT32F=0x8CAC, DEPTH_ATTACHMENT=0x8D00,
    FRAMEBUFFER_COMPLETE=0x8CD5, COLOR_BUFFER_BIT=0x4000, DEPTH_BUFFER_BIT=0x100, DEPTH_TEST=0x0B71, LESS=0x0201,
    LEQUAL=0x0203, POINTS=0x0000, TRIANGLE_STRIP=0x0005, TRANSFORM_FEEDBACK_BUFFER=0x8C8E, INTERLEAVED_ATTRIBS=0x8C8C,
    TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN=0x8C88, QUERY_RESULT=0x8866, RASTERIZER_DISCARD=0x8C89, BLEND=0x0BE2,
    ONE=1, FUNC_ADD=0x8006, SUBPIXEL_BITS=0x0D50, PROGRAM_POINT_SIZE=0x8642, DEPTH_BITS=0x0D56,
)


class GLError(RuntimeError):
    pass


class Context:
    """process-wide GL context + ctypes access to the entry points"""
    _inst = None

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def __init__(self):
        if not os.path.exists(LIB):
            build_shim()
        self.L = C.CDLL(LIB)
        self.L.gl_ctx_create.argtypes = [C.c_char_p]
        self.L.gl_ctx_error.restype = C.c_char_p
        self.L.gl_ctx_proc.restype = C.c_void_p
        self.L.gl_ctx_proc.argtypes = [C.c_char_p]
        rc = self.L.gl_ctx_create(os.environ.get("SUMA_SWRAST_DRI", "").encode())
        if rc != 0:
            raise GLError(f"no software GL context ({rc}): {self.L.gl_ctx_error().decode()}")
        self._fn = {}
        gs = self.fn("glGetString", C.c_char_p, C.c_uint)
        self.version = gs(GL["VERSION"]).decode()
        self.renderer = gs(GL["RENDERER"]).decode()

    def fn(self, name, restype, *argtypes):
        key = (name, restype, argtypes)
        f = self._fn.get(key)
        if f is None:
            p = self.L.gl_ctx_proc(name.encode())
            if not p:
                raise GLError(f"GL entry point {name} not found")
            f = C.CFUNCTYPE(restype, *argtypes)(p)
            self._fn[key] = f
        return f

    def check(self, what=""):
        e = self.fn("glGetError", C.c_uint)()
        if e:
            raise GLError(f"GL error 0x{e:x} after {what}")


def build_shim():
    """compile oracle/glref/gl_ctx.c -> oracle/_ref/libsuma_glctx.so (needs the Mesa DRI headers of this image)"""
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-o", LIB, os.path.join(HERE, "glref", "gl_ctx.c"), "-ldl"])


def available():
    try:
        Context.get()
        return True
    except (GLError, OSError, subprocess.CalledProcessError):
        return False


# --- the CONTROL of tests/test_gl_pipeline.py: the reference's shader text with the driver's transcendentals replaced.
# GLSL leaves the accuracy of asin / acos / atan to the implementation (GLSL 4.50 section 4.7.1: "undefined" for the
# angle functions); llvmpipe's asin is off by up to 3.9e-4 rad.  This prelude -- the Cephes
# kernels of include/suma_detmath.h (sdm_atan, sdm_atan2, sdm_asin, sdm_acos) restated in GLSL, one IEEE operation per
# statement -- a fused multiply-add where the header's SDM_MA is (through double: llvmpipe's own fma() rounds twice),
# `precise` so that the GLSL compiler neither fuses nor reassociates the rest; tests/test_gl_controls.py checks the
# functions' bits on 200 000 arguments -- is put
# between the #version line and the reference's text, followed by three #defines.  Not one character of the reference's
# shader is changed: what changes is the library the GL implementation evaluates its three angle functions with.
DETMATH_PRELUDE = """
#extension GL_ARB_gpu_shader_fp64 : require
// one rounding: the product of two floats is exact in double, the sum is rounded to 53 bits and then to 24 (llvmpipe
// lowers GLSL's own sdm_fma() to a multiply and an add, i.e. two roundings: measured, tests/test_gl_controls.py)
float sdm_fma(float a, float b, float c) { precise double t = double(a) * double(b); t = t + double(c); return float(t); }
float sdm_atan1(float xx) {
  precise float x = abs(xx);
  precise float y = 0.0;
  if (x > 2.414213562373095) { y = 1.57079632679489661923; x = -(1.0 / x); }
  else if (x > 0.4142135623730950) { y = 0.78539816339744830962; precise float a = x - 1.0; precise float b = x + 1.0; x = a / b; }
  precise float z = x * x;
  precise float p = sdm_fma(8.05374449538e-2, z, -1.38776856032e-1);
  p = sdm_fma(p, z, 1.99777106478e-1);
  p = sdm_fma(p, z, -3.33329491539e-1);
  precise float pz = p * z;
  p = sdm_fma(pz, x, x);
  y = y + p;
  return (xx < 0.0) ? -y : y;
}
float sdm_atan_gl(float yx) { return sdm_atan1(yx); }
float sdm_atan_gl(float y, float x) {
  if (isnan(x) || isnan(y)) return x + y;
  if (x == 0.0) { if (y > 0.0) return 1.57079632679489661923; if (y < 0.0) return -1.57079632679489661923; return 0.0; }
  precise float q = y / x;
  precise float z = sdm_atan1(q);
  if (x < 0.0) { if (y < 0.0) z = z - 3.14159265358979323846; else z = z + 3.14159265358979323846; }
  return z;
}
float sdm_asin_gl(float xx) {
  precise float a = abs(xx);
  if (!(a <= 1.0)) return uintBitsToFloat(0x7fc00000u);
  if (a < 1.0e-4) return xx;
  precise float x; precise float z; bool flag = false;
  if (a > 0.5) { z = 1.0 - a; z = 0.5 * z; x = sqrt(z); flag = true; } else { x = a; z = x * x; }
  precise float p = sdm_fma(4.2163199048e-2, z, 2.4181311049e-2);
  p = sdm_fma(p, z, 4.5470025998e-2);
  p = sdm_fma(p, z, 7.4953002686e-2);
  p = sdm_fma(p, z, 1.6666752422e-1);
  precise float pz = p * z;
  z = sdm_fma(pz, x, x);
  if (flag) { z = z + z; z = 1.57079632679489661923 - z; }
  return (xx < 0.0) ? -z : z;
}
float sdm_acos_gl(float x) {
  if (!(abs(x) <= 1.0)) return uintBitsToFloat(0x7fc00000u);
  if (x < -0.5) { precise float t = 1.0 + x; t = 0.5 * t; t = sdm_asin_gl(sqrt(t)); t = 2.0 * t; return 3.14159265358979323846 - t; }
  if (x > 0.5) { precise float t = 1.0 - x; t = 0.5 * t; t = sdm_asin_gl(sqrt(t)); return 2.0 * t; }
  return 1.57079632679489661923 - sdm_asin_gl(x);
}
#define asin sdm_asin_gl
#define acos sdm_acos_gl
#define atan sdm_atan_gl
"""

_prelude = [""]


class transcendentals:
    """`with glref.transcendentals("detmath"):` -- programs BUILT inside the block get DETMATH_PRELUDE (above) in front
    of the reference's text; "driver" (the default) leaves the GL implementation's own functions in place"""

    def __init__(self, which):
        self.text = {"driver": "", "detmath": DETMATH_PRELUDE}[which]

    def __enter__(self):
        self.saved, _prelude[0] = _prelude[0], self.text

    def __exit__(self, *exc):
        _prelude[0] = self.saved


def shader_source(name):
    """a shader of the reference with its #include lines resolved (glow does that at load time)"""
    import re
    with open(os.path.join(SHADER_DIR, name)) as f:
        src = f.read()

    def inc(m):
        with open(os.path.join(os.path.dirname(SHADER_DIR), m.group(1))) as g:
            return g.read()
    src = re.sub(r'#include\s+"([^"]+)"', inc, src)
    if _prelude[0]:
        # behind the #version line, which stays the reference's; `precise` comes with GL_ARB_gpu_shader5
        src = re.sub(r"^(\s*#version[^\n]*\n)", lambda m: m.group(1) + "#extension GL_ARB_gpu_shader5 : require\n" + _prelude[0],
                     src, count=1)
    return src


u32, i32, f32, vp = C.c_uint, C.c_int, C.c_float, C.c_void_p


class Program:
    def __init__(self, stages, tf_varyings=None, from_reference=True):
        """stages: {GL stage enum name: file name under src/shader (from_reference) or GLSL text}"""
        g = Context.get()
        self.g = g
        self.id = g.fn("glCreateProgram", u32)()
        for kind, src in stages.items():
            text = shader_source(src) if from_reference else src
            sh = g.fn("glCreateShader", u32, u32)(GL[kind])
            buf = C.c_char_p(text.encode())
            g.fn("glShaderSource", None, u32, i32, C.POINTER(C.c_char_p), vp)(sh, 1, C.byref(buf), None)
            g.fn("glCompileShader", None, u32)(sh)
            ok = i32(0)
            g.fn("glGetShaderiv", None, u32, u32, C.POINTER(i32))(sh, GL["COMPILE_STATUS"], C.byref(ok))
            if not ok.value:
                log = C.create_string_buffer(8192)
                g.fn("glGetShaderInfoLog", None, u32, i32, vp, vp)(sh, 8192, None, log)
                raise GLError(f"compile {src if from_reference else kind}: {log.value.decode()}")
            g.fn("glAttachShader", None, u32, u32)(self.id, sh)
        if tf_varyings:
            arr = (C.c_char_p * len(tf_varyings))(*[v.encode() for v in tf_varyings])
            g.fn("glTransformFeedbackVaryings", None, u32, i32, vp, u32)(self.id, len(tf_varyings), arr, GL["INTERLEAVED_ATTRIBS"])
        g.fn("glLinkProgram", None, u32)(self.id)
        ok = i32(0)
        g.fn("glGetProgramiv", None, u32, u32, C.POINTER(i32))(self.id, GL["LINK_STATUS"], C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(8192)
            g.fn("glGetProgramInfoLog", None, u32, i32, vp, vp)(self.id, 8192, None, log)
            raise GLError(f"link: {log.value.decode()}")

    def use(self):
        self.g.fn("glUseProgram", None, u32)(self.id)

    def loc(self, name):
        return self.g.fn("glGetUniformLocation", i32, u32, C.c_char_p)(self.id, name.encode())

    def set(self, **uniforms):
        """ints / bools -> glUniform1i, floats -> glUniform1f, 4x4 numpy (row-major) -> glUniformMatrix4fv (transposed),
        length-2/3/4 float sequences -> glUniformNf"""
        self.use()
        g = self.g
        for name, v in uniforms.items():
            l = self.loc(name)
            if l < 0:
                continue  # optimised out / not declared in this program: glow ignores it too
            if isinstance(v, (bool, np.bool_)) or isinstance(v, (int, np.integer)):
                g.fn("glUniform1i", None, i32, i32)(l, int(v))
            elif isinstance(v, (float, np.floating)):
                g.fn("glUniform1f", None, i32, f32)(l, float(v))
            else:
                a = np.asarray(v, dtype=np.float32)
                if a.shape == (4, 4):
                    m = np.ascontiguousarray(a.T)  # column-major, as Eigen hands it over
                    g.fn("glUniformMatrix4fv", None, i32, i32, C.c_ubyte, vp)(l, 1, 0, m.ctypes.data)
                elif a.size == 2:
                    g.fn("glUniform2f", None, i32, f32, f32)(l, *map(float, a))
                elif a.size == 3:
                    g.fn("glUniform3f", None, i32, f32, f32, f32)(l, *map(float, a))
                elif a.size == 4:
                    g.fn("glUniform4f", None, i32, f32, f32, f32, f32)(l, *map(float, a))
                else:
                    raise ValueError(name)


def gen(kind):
    g = Context.get()
    out = u32(0)
    g.fn("glGen" + kind, None, i32, C.POINTER(u32))(1, C.byref(out))
    return out.value


class Buffer:
    def __init__(self, data=None, nbytes=0, target="ARRAY_BUFFER"):
        g = Context.get()
        self.g, self.id, self.target = g, gen("Buffers"), GL[target]
        if data is not None:
            data = np.ascontiguousarray(data)
            nbytes = data.nbytes
        self.nbytes = nbytes
        g.fn("glBindBuffer", None, u32, u32)(self.target, self.id)
        g.fn("glBufferData", None, u32, C.c_ssize_t, vp, u32)(self.target, nbytes, data.ctypes.data if data is not None else None,
                                                                GL["DYNAMIC_COPY"])

    def read(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        g = self.g
        g.fn("glBindBuffer", None, u32, u32)(GL["ARRAY_BUFFER"], self.id)
        g.fn("glGetBufferSubData", None, u32, C.c_ssize_t, C.c_ssize_t, vp)(GL["ARRAY_BUFFER"], 0, out.nbytes, out.ctypes.data)
        return out


class RectTexture:
    """GL_TEXTURE_RECTANGLE, RGBA32F (glow::GlTextureRectangle of Frame.h:63-70); row 0 = bottom row"""

    def __init__(self, w, h, data=None, filter="NEAREST"):
        g = Context.get()
        self.g, self.w, self.h, self.id = g, w, h, gen("Textures")
        T = GL["TEXTURE_RECTANGLE"]
        g.fn("glBindTexture", None, u32, u32)(T, self.id)
        if data is not None:
            data = np.ascontiguousarray(data, dtype=np.float32).reshape(h, w, 4)
        g.fn("glTexImage2D", None, u32, i32, i32, i32, i32, i32, u32, u32, vp)(
            T, 0, GL["RGBA32F"], w, h, 0, GL["RGBA"], GL["FLOAT"], data.ctypes.data if data is not None else None)
        for pname, val in (("TEXTURE_MIN_FILTER", filter), ("TEXTURE_MAG_FILTER", filter), ("TEXTURE_WRAP_S", "CLAMP_TO_BORDER"),
                           ("TEXTURE_WRAP_T", "CLAMP_TO_BORDER")):
            g.fn("glTexParameteri", None, u32, u32, i32)(T, GL[pname], GL[val])

    def bind(self, unit):
        self.g.fn("glActiveTexture", None, u32)(GL["TEXTURE0"] + unit)
        self.g.fn("glBindTexture", None, u32, u32)(GL["TEXTURE_RECTANGLE"], self.id)

    def read(self):
        out = np.empty((self.h, self.w, 4), dtype=np.float32)
        g = self.g
        g.fn("glBindTexture", None, u32, u32)(GL["TEXTURE_RECTANGLE"], self.id)
        g.fn("glGetTexImage", None, u32, i32, u32, u32, vp)(GL["TEXTURE_RECTANGLE"], 0, GL["RGBA"], GL["FLOAT"], out.ctypes.data)
        return out


class BufferTexture:
    """samplerBuffer over a float4 array (glow::GlTextureBuffer poseTexture_, SurfelMap.cpp:25)"""

    def __init__(self, data):
        g = Context.get()
        self.g = g
        self.buf = Buffer(np.ascontiguousarray(data, dtype=np.float32), target="ARRAY_BUFFER")
        self.id = gen("Textures")
        g.fn("glBindTexture", None, u32, u32)(GL["TEXTURE_BUFFER"], self.id)
        g.fn("glTexBuffer", None, u32, u32, u32)(GL["TEXTURE_BUFFER"], GL["RGBA32F"], self.buf.id)

    def bind(self, unit):
        self.g.fn("glActiveTexture", None, u32)(GL["TEXTURE0"] + unit)
        self.g.fn("glBindTexture", None, u32, u32)(GL["TEXTURE_BUFFER"], self.id)


class Framebuffer:
    """glow::GlFramebuffer with a DEPTH24_STENCIL8 renderbuffer (SurfelMap.cpp:103-122, 172-173; Preprocessing.cpp:22-28)"""

    def __init__(self, w, h, depth="DEPTH24_STENCIL8"):
        g = Context.get()
        self.g, self.w, self.h = g, w, h
        self.id = gen("Framebuffers")
        self.rbo = gen("Renderbuffers")
        g.fn("glBindRenderbuffer", None, u32, u32)(GL["RENDERBUFFER"], self.rbo)
        g.fn("glRenderbufferStorage", None, u32, u32, i32, i32)(GL["RENDERBUFFER"], GL[depth], w, h)
        self.bind()
        att = "DEPTH_STENCIL_ATTACHMENT" if depth == "DEPTH24_STENCIL8" else "DEPTH_ATTACHMENT"
        g.fn("glFramebufferRenderbuffer", None, u32, u32, u32, u32)(GL["FRAMEBUFFER"], GL[att], GL["RENDERBUFFER"], self.rbo)

    def bind(self):
        self.g.fn("glBindFramebuffer", None, u32, u32)(GL["FRAMEBUFFER"], self.id)

    def attach(self, textures):
        g = self.g
        self.bind()
        for k, t in enumerate(textures):
            g.fn("glFramebufferTexture2D", None, u32, u32, u32, u32, i32)(GL["FRAMEBUFFER"], GL["COLOR_ATTACHMENT0"] + k,
                                                                          GL["TEXTURE_RECTANGLE"], t.id, 0)
        bufs = (u32 * len(textures))(*[GL["COLOR_ATTACHMENT0"] + k for k in range(len(textures))])
        g.fn("glDrawBuffers", None, i32, vp)(len(textures), bufs)
        st = g.fn("glCheckFramebufferStatus", u32, u32)(GL["FRAMEBUFFER"])
        if st != GL["FRAMEBUFFER_COMPLETE"]:
            raise GLError(f"framebuffer incomplete: 0x{st:x}")


SURFEL_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("radius", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
                         ("confidence", "<f4"), ("timestamp", "<i4"), ("color", "<f4"), ("weight", "<f4"), ("count", "<f4"),
                         ("r", "<f4"), ("g", "<f4"), ("b", "<f4"), ("w", "<f4")])


def surfel_vao(buf):
    """vao_surfels_, SurfelMap.cpp:46-55: the 64-byte Surfel record as five vertex attributes"""
    g = Context.get()
    vao = gen("VertexArrays")
    g.fn("glBindVertexArray", None, u32)(vao)
    g.fn("glBindBuffer", None, u32, u32)(GL["ARRAY_BUFFER"], buf.id)
    vap = g.fn("glVertexAttribPointer", None, u32, i32, u32, C.c_ubyte, i32, vp)
    vaip = g.fn("glVertexAttribIPointer", None, u32, i32, u32, i32, vp)
    en = g.fn("glEnableVertexAttribArray", None, u32)
    vap(0, 4, GL["FLOAT"], 0, 64, 0)
    vap(1, 4, GL["FLOAT"], 0, 64, 16)
    vaip(2, 1, GL["INT"], 64, 32)
    vap(3, 3, GL["FLOAT"], 0, 64, 36)
    vap(4, 4, GL["FLOAT"], 0, 64, 48)
    for k in range(5):
        en(k)
    return vao


def common_state(w, h, depth_func="LESS"):
    g = Context.get()
    g.fn("glEnable", None, u32)(GL["DEPTH_TEST"])
    g.fn("glDepthFunc", None, u32)(GL[depth_func])
    g.fn("glClearColor", None, f32, f32, f32, f32)(0, 0, 0, 0)
    g.fn("glViewport", None, i32, i32, i32, i32)(0, 0, w, h)


def clear():
    Context.get().fn("glClear", None, u32)(GL["COLOR_BUFFER_BIT"] | GL["DEPTH_BUFFER_BIT"])


def draw_points(vao, n):
    g = Context.get()
    g.fn("glBindVertexArray", None, u32)(vao)
    g.fn("glDrawArrays", None, u32, i32, i32)(GL["POINTS"], 0, n)


# =====================================================================================================================
# passes
# =====================================================================================================================
class SurfelRenderer:
    """SurfelMap::render / render_active / render_inactive with the reference's render_surfels.{vert,geom,frag}
    (SurfelMap.cpp:172-186 program + framebuffer, :847-1021 render(), :1023-1114 the single passes)"""

    def __init__(self, params):
        self.p = params
        self.W, self.H = params.model_width, params.model_height
        self.prog = Program({"VERTEX_SHADER": "render_surfels.vert", "GEOMETRY_SHADER": "render_surfels.geom",
                             "FRAGMENT_SHADER": "render_surfels.frag"})
        # SurfelMap.cpp:179-186: the uniforms of the render program
        self.prog.set(fov_up=float(params.model_fov_up), fov_down=float(params.model_fov_down),
                      max_depth=float(params.model_max_depth), min_depth=float(params.model_min_depth),
                      use_stability=bool(params.use_stability), poseBuffer=5)
        self.fbo = Framebuffer(self.W, self.H)

    def render_pass(self, surfels, poses, pose, conf_threshold, timestamp_threshold, render_old, depth_func="LESS",
                    targets=None, clear_first=True):
        """one glDrawArrays(GL_POINTS, 0, surfels_.size()) of render(): returns the three attachments as (H, W, 4)"""
        W, H = self.W, self.H
        surfels = np.ascontiguousarray(surfels)
        vbo = Buffer(surfels.view(np.uint8))
        vao = surfel_vao(vbo)
        ptex = BufferTexture(np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 4))
        if targets is None:
            targets = [RectTexture(W, H) for _ in range(3)]
        common_state(W, H, depth_func)
        ptex.bind(5)
        self.fbo.attach(targets)
        inv_pose = np.linalg.inv(np.asarray(pose, dtype=np.float32).astype(np.float32))  # Eigen: Matrix4f::inverse()
        self.prog.set(conf_threshold=float(conf_threshold), timestamp_threshold=int(timestamp_threshold),
                      inv_pose=inv_pose.astype(np.float32), render_old_surfels=bool(render_old))
        self.prog.use()
        if clear_first:
            clear()
        draw_points(vao, surfels.shape[0])
        Context.get().fn("glFinish", None)()
        Context.get().check("render_surfels draw")
        return targets

    def render(self, surfels, poses, pose, conf_threshold, timestamp_threshold, render_old):
        return [t.read() for t in self.render_pass(surfels, poses, pose, conf_threshold, timestamp_threshold, render_old)]


PASS_VS = """#version 330 core
layout (location = 0) in vec4 pos;      // clip-space position, w = 1
layout (location = 1) in vec2 tc;
layout (location = 2) in float prim;    // primitive id + 1 (float: exact up to 2^24)
out vec2 texCoords;
flat out float id;
void main() { gl_Position = pos; texCoords = tc; id = prim; }
"""
PASS_FS = """#version 330 core
in vec2 texCoords;
flat in float id;
uniform bool disc;
layout (location = 0) out vec4 o;
void main() {
  if (disc && dot(texCoords, texCoords) > 1.0f) discard;
  o = vec4(id, gl_FragCoord.z, texCoords);
}
"""


class QuadRaster:
    """Fixed-function check: the oracle's OWN quad corners (ora_debug_render_quads) go through GL's clipping,
    rasterisation and depth test behind a pass-through vertex shader -- nothing but the rules of the GL pipeline
    decides which primitive owns a pixel.  strips: [n, 4, 3] corners in [0, 1]^3 (x may leave [0, 1] at the seam)."""

    def __init__(self, W, H, depth="DEPTH24_STENCIL8"):
        self.W, self.H = W, H
        self.prog = Program({"VERTEX_SHADER": PASS_VS, "FRAGMENT_SHADER": PASS_FS}, from_reference=False)
        self.fbo = Framebuffer(W, H, depth)

    def run(self, corners, ids, disc=True, depth_func="LESS", flat_z=False, depth_test=True):
        g = Context.get()
        corners = np.asarray(corners, dtype=np.float32)
        n = corners.shape[0]
        tcs = np.array([[-1, -1], [1, -1], [-1, 1], [1, 1]], dtype=np.float32)
        v = np.zeros((n, 4, 7), dtype=np.float32)
        # gl_Position = vec4(2.0 * project2model(...) - 1.0, 1.0f), render_surfels.geom:104-117 -- in fp32, as the shader does
        v[:, :, 0:3] = np.float32(2.0) * corners - np.float32(1.0)
        if flat_z:
            v[:, :, 2] = v[:, :1, 2]
        v[:, :, 3] = 1.0
        v[:, :, 4:6] = tcs[None]
        v[:, :, 6] = (np.asarray(ids, dtype=np.float32) + 1.0)[:, None]
        vbo = Buffer(v.reshape(-1, 7))
        vao = gen("VertexArrays")
        g.fn("glBindVertexArray", None, u32)(vao)
        g.fn("glBindBuffer", None, u32, u32)(GL["ARRAY_BUFFER"], vbo.id)
        vap = g.fn("glVertexAttribPointer", None, u32, i32, u32, C.c_ubyte, i32, vp)
        en = g.fn("glEnableVertexAttribArray", None, u32)
        vap(0, 4, GL["FLOAT"], 0, 28, 0)
        vap(1, 2, GL["FLOAT"], 0, 28, 16)
        vap(2, 1, GL["FLOAT"], 0, 28, 24)
        for k in range(3):
            en(k)
        tex = RectTexture(self.W, self.H)
        common_state(self.W, self.H, depth_func)
        if not depth_test:
            g.fn("glDisable", None, u32)(GL["DEPTH_TEST"])
        self.fbo.attach([tex])
        self.prog.set(disc=bool(disc))
        self.prog.use()
        clear()
        first = (i32 * n)(*range(0, 4 * n, 4))
        count = (i32 * n)(*([4] * n))
        g.fn("glMultiDrawArrays", None, u32, vp, vp, i32)(GL["TRIANGLE_STRIP"], first, count, n)
        g.fn("glFinish", None)()
        g.check("pass-through quads")
        out = tex.read()
        return out[..., 0].astype(np.int64) - 1, out[..., 1]  # winner id per pixel (-1: none), window-space depth


def limits():
    g = Context.get()
    v = i32(0)
    g.fn("glGetIntegerv", None, u32, C.POINTER(i32))(GL["SUBPIXEL_BITS"], C.byref(v))
    return {"version": g.version, "renderer": g.renderer, "subpixel_bits": v.value}


class Sampler:
    """glow::GlSampler of Frame2Model (Frame2Model.cpp:99-109): CLAMP_TO_BORDER, LINEAR or NEAREST, one object on all units"""

    def __init__(self, linear):
        g = Context.get()
        self.g, self.id = g, gen("Samplers")
        sp = g.fn("glSamplerParameteri", None, u32, u32, i32)
        f = GL["LINEAR"] if linear else GL["NEAREST"]
        for pname, val in ((GL["TEXTURE_MIN_FILTER"], f), (GL["TEXTURE_MAG_FILTER"], f), (GL["TEXTURE_WRAP_S"], GL["CLAMP_TO_BORDER"]),
                           (GL["TEXTURE_WRAP_T"], GL["CLAMP_TO_BORDER"])):
            sp(self.id, pname, val)

    def bind(self, unit):
        self.g.fn("glBindSampler", None, u32, u32)(unit, self.id)


class Jacobians:
    """Frame2Model::jacobianProducts (Frame2Model.cpp:14-63 set-up, :136-261 the call) with the reference's
    Frame2Model_jacobians.{vert,geom,frag}: ceil(W / 64) * H points, each geometry-shader invocation walks 64 texels and
    emits 16 points into a 2 x 8 RGB32F target with GL_ONE / GL_ONE blending.  Returns the 48 floats the host reads
    back (:214-227 unpacks them)."""

    def __init__(self, params, entries_per_kernel=64, order=None):
        import math
        p = self.p = params
        self.epk = entries_per_kernel
        self.W, self.H = p.data_width, p.data_height
        self.prog = Program({"VERTEX_SHADER": "Frame2Model_jacobians.vert", "GEOMETRY_SHADER": "Frame2Model_jacobians.geom",
                             "FRAGMENT_SHADER": "Frame2Model_jacobians.frag"})
        self.prog.set(vertex_model=0, normal_model=1, vertex_data=2, normal_data=3, semantic_model=4, semantic_data=5,
                      entries_per_kernel=int(entries_per_kernel))
        fov_up, fov_down = abs(float(np.float32(p.data_fov_up))), abs(float(np.float32(p.data_fov_down)))
        self.prog.set(distance_outliers=bool(p.weight_function == 0), weight_function=int(p.weight_function),
                      factor=float(p.factor), distance_thresh=float(p.icp_max_distance),
                      angle_thresh=float(np.float32(math.cos(float(np.float32(p.icp_max_angle)) * math.pi / 180.0))),
                      fov_up=fov_up, fov_down=fov_down, fov=float(np.float32(fov_up) + np.float32(fov_down)),
                      min_depth=float(p.min_depth), max_depth=float(p.max_depth), cutoff_threshold=0.0)
        coords = np.array([(i + 0.5, j + 0.5) for i in range(0, self.W, entries_per_kernel) for j in range(self.H)], dtype=np.float32)
        if order is not None:  # control experiment only (tests/test_gl_controls.py): the same points, drawn in another order
            coords = np.ascontiguousarray(coords[np.asarray(order)])
        self.n = coords.shape[0]
        g = Context.get()
        self.vbo = Buffer(coords)
        self.vao = gen("VertexArrays")
        g.fn("glBindVertexArray", None, u32)(self.vao)
        g.fn("glBindBuffer", None, u32, u32)(GL["ARRAY_BUFFER"], self.vbo.id)
        g.fn("glVertexAttribPointer", None, u32, i32, u32, C.c_ubyte, i32, vp)(0, 2, GL["FLOAT"], 0, 8, 0)
        g.fn("glEnableVertexAttribArray", None, u32)(0)
        self.sampler = Sampler(bool(p.bilinear_sampling))
        self.target = RectTexture(2, 8)   # RGBA32F here; the reference's RGB_FLOAT target has no alpha to blend either
        self.fbo = Framebuffer(2, 8)

    def run(self, cur, model, pose, iteration):
        """cur / model: three (H, W, 4) maps each (vertex, normal, semantic)"""
        g = Context.get()
        tex = [RectTexture(m.shape[1], m.shape[0], m) for m in (model[0], model[1], cur[0], cur[1], model[2], cur[2])]
        g.fn("glPointSize", None, f32)(1.0)
        g.fn("glClearColor", None, f32, f32, f32, f32)(0, 0, 0, 0)
        g.fn("glDisable", None, u32)(GL["DEPTH_TEST"])
        for unit, t in enumerate(tex):
            t.bind(unit)
            self.sampler.bind(unit)
        self.fbo.attach([self.target])
        g.fn("glViewport", None, i32, i32, i32, i32)(0, 0, 2, 8)
        clear()
        g.fn("glEnable", None, u32)(GL["BLEND"])
        g.fn("glBlendFunc", None, u32, u32)(GL["ONE"], GL["ONE"])
        self.prog.set(pose=np.asarray(pose, dtype=np.float64).astype(np.float32), iteration=int(iteration))
        self.prog.use()
        draw_points(self.vao, self.n)
        g.fn("glDisable", None, u32)(GL["BLEND"])
        g.fn("glFinish", None)()
        g.check("Frame2Model_jacobians draw")
        for unit in range(6):
            g.fn("glBindSampler", None, u32, u32)(unit, 0)
        return self.target.read()[..., :3].reshape(-1)  # 8 rows x 2 texels x RGB = 48 floats, PixelFormat::RGB download


class VertexMap:
    """Preprocessing::process, first pass (Preprocessing.cpp:24-60 set-up, :120-189 the draw) with the reference's
    gen_vertexmap.{vert,frag}: GL_POINTS of size 1 into the vertex map + semantic map, DEPTH24_STENCIL8, GL_LESS.  The
    label / prob attribute pointers start 4 / 5 floats into their buffers, as the reference sets them
    (Preprocessing.cpp:142-145, quirk B-1); the buffers are zero-padded so that the fetches stay inside."""

    def __init__(self, params):
        p = self.p = params
        self.W, self.H = p.data_width, p.data_height
        self.prog = Program({"VERTEX_SHADER": "gen_vertexmap.vert", "FRAGMENT_SHADER": "gen_vertexmap.frag"})
        # Preprocessing::setParameters, Preprocessing.cpp:77-100
        self.prog.set(width=float(self.W), height=float(self.H), fov_up=abs(float(np.float32(p.data_fov_up))),
                      fov_down=abs(float(np.float32(p.data_fov_down))), min_depth=float(p.min_depth), max_depth=float(p.max_depth))
        self.fbo = Framebuffer(self.W, self.H)

    def run(self, points, labels, probs, timestamp, label_offset=4, prob_offset=5):
        g = Context.get()
        n = points.shape[0]
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(n, 4)
        lab = np.zeros(n + 8, dtype=np.float32)
        prb = np.zeros(n + 8, dtype=np.float32)
        lab[:n] = labels
        prb[:n] = probs
        bp, bl, bq = Buffer(pts), Buffer(lab), Buffer(prb)
        vao = gen("VertexArrays")
        g.fn("glBindVertexArray", None, u32)(vao)
        vap = g.fn("glVertexAttribPointer", None, u32, i32, u32, C.c_ubyte, i32, vp)
        bind = g.fn("glBindBuffer", None, u32, u32)
        en = g.fn("glEnableVertexAttribArray", None, u32)
        bind(GL["ARRAY_BUFFER"], bp.id)
        vap(0, 4, GL["FLOAT"], 0, 16, 0)
        bind(GL["ARRAY_BUFFER"], bl.id)
        vap(1, 1, GL["FLOAT"], 0, 4, 4 * label_offset)
        bind(GL["ARRAY_BUFFER"], bq.id)
        vap(2, 1, GL["FLOAT"], 0, 4, 4 * prob_offset)
        for k in range(3):
            en(k)
        vmap, smap = RectTexture(self.W, self.H), RectTexture(self.W, self.H)
        g.fn("glPointSize", None, f32)(1.0)
        common_state(self.W, self.H, "LESS")
        self.fbo.attach([vmap, smap])
        self.prog.set(isfirst=bool(timestamp < 10))
        self.prog.use()
        clear()
        draw_points(vao, n)
        g.fn("glFinish", None)()
        g.check("gen_vertexmap draw")
        return vmap.read(), smap.read()


class IndexMap:
    """SurfelMap::renderIndexmap (SurfelMap.cpp:586-604; framebuffer :100-110, program :140-150, state :507-509) with the
    reference's gen_indexmap.{vert,frag}: every surfel of the map as a GL_POINT of size 1 at the centre of the data texel
    it projects to, three colour attachments, DEPTH24_STENCIL8 + GL_LESS -- the nearest front-facing surfel per texel,
    ties to the surfel drawn first.  Returns the index attachment as uint32 (gl_VertexID + 1; 0 = no surfel)."""

    def __init__(self, params):
        p = self.p = params
        self.W, self.H = p.data_width, p.data_height
        self.prog = Program({"VERTEX_SHADER": "gen_indexmap.vert", "FRAGMENT_SHADER": "gen_indexmap.frag"})
        self.prog.set(fov_up=abs(float(np.float32(p.data_fov_up))), fov_down=abs(float(np.float32(p.data_fov_down))),
                      min_depth=float(p.min_depth), max_depth=float(p.max_depth), width=float(self.W), height=float(self.H),
                      poseBuffer=5)
        self.fbo = Framebuffer(self.W, self.H)

    def run(self, surfels, poses, pose, inv_pose):
        g = Context.get()
        surfels = np.ascontiguousarray(surfels)
        vbo = Buffer(surfels.view(np.uint8))
        vao = surfel_vao(vbo)
        ptex = BufferTexture(np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 4))
        targets = [RectTexture(self.W, self.H) for _ in range(3)]  # the reference's first attachment is R32F: same raster
        g.fn("glPointSize", None, f32)(1.0)
        common_state(self.W, self.H, "LESS")
        ptex.bind(5)
        self.fbo.attach(targets)
        self.prog.set(pose=np.asarray(pose, dtype=np.float32), inv_pose=np.asarray(inv_pose, dtype=np.float32))
        self.prog.use()
        clear()
        draw_points(vao, surfels.shape[0])
        g.fn("glFinish", None)()
        g.check("gen_indexmap draw")
        return targets[0].read()[..., 0].astype(np.uint32), targets[1].read()


class NormalsLabels:
    """Preprocessing::process, passes 2 and 3 (Preprocessing.cpp:238-327; programs :45-53, sampler :68-70) with the
    reference's empty.vert + quad.geom + gen_normalmap.frag / floodfill.frag: one GL_POINT expanded to a full-screen quad,
    interpolated texCoords, NEAREST + CLAMP_TO_BORDER sampler objects on both units, no depth test.  Input: the vertex
    map and the raw semantic map of pass 1.  Returns (normal map, eroded labels, refined labels)."""

    def __init__(self, params):
        self.W, self.H = params.data_width, params.data_height
        stages = lambda frag: {"VERTEX_SHADER": "empty.vert", "GEOMETRY_SHADER": "quad.geom", "FRAGMENT_SHADER": frag}
        self.normal = Program(stages("gen_normalmap.frag"))
        self.flood = Program(stages("floodfill.frag"))
        for prog in (self.normal, self.flood):
            prog.set(vertex_map=0, semantic_map=1)  # the only uniforms the reference sets (Preprocessing.cpp:113-116)
        self.fbo = Framebuffer(self.W, self.H)
        self.sampler = Sampler(linear=False)
        self.vao = gen("VertexArrays")  # vao_no_points_

    def _pass(self, prog, vmap_tex, sem_tex, n_out):
        g = Context.get()
        outs = [RectTexture(self.W, self.H) for _ in range(n_out)]
        g.fn("glDisable", None, u32)(GL["DEPTH_TEST"])
        g.fn("glClearColor", None, f32, f32, f32, f32)(0, 0, 0, 0)
        g.fn("glViewport", None, i32, i32, i32, i32)(0, 0, self.W, self.H)
        self.fbo.attach(outs)
        vmap_tex.bind(0)
        sem_tex.bind(1)
        self.sampler.bind(0)
        self.sampler.bind(1)
        prog.use()
        clear()
        draw_points(self.vao, 1)
        g.fn("glFinish", None)()
        g.fn("glBindSampler", None, u32, u32)(0, 0)
        g.fn("glBindSampler", None, u32, u32)(1, 0)
        g.check("full-screen pass")
        return outs

    def run(self, vmap, smap):
        v = RectTexture(self.W, self.H, vmap)
        s = RectTexture(self.W, self.H, smap)
        nmap, eroded = self._pass(self.normal, v, s, 2)
        (refined,) = self._pass(self.flood, v, eroded, 1)
        return nmap.read(), eroded.read(), refined.read()


class SurfelUpdate:
    """SurfelMap::updateSurfels, first draw (K9; SurfelMap.cpp:621-644, program :152-157, uniforms :399-438, state
    :507-545) with the reference's update_surfels.{vert,geom,frag}: every surfel of the map as a GL_POINT, TRANSFORM
    FEEDBACK of the five interleaved varyings (the 64-byte record) in primitive order -- the geometry shader drops the
    surfels the update removes -- while the surviving points are rasterised into the integration mask.  The map's sampler
    object (MIN NEAREST / MAG LINEAR / CLAMP_TO_BORDER, :168-170) sits on units 0-6."""

    VARYINGS = ["sfl_position_radius", "sfl_normal_confidence", "sfl_timestamp", "sfl_color_weight_count", "sfl_semantic_map"]

    def __init__(self, params, tag_sources=False):
        """tag_sources: the geometry shader (a pass-through of the vertex stage's record) gets ONE more output, the index
        of the input primitive, captured as a sixth feedback word group -- test instrumentation that lets two runs be
        aligned surfel by surfel when a handful of borderline surfels survive in one and not in the other; the vertex
        shader, where the update is decided and computed, stays the reference's text"""
        self.W, self.H = params.data_width, params.data_height
        self.tagged = bool(tag_sources)
        geom = shader_source("update_surfels.geom")
        if tag_sources:
            assert "out vec4 sfl_semantic_map;" in geom and "EmitVertex();" in geom
            geom = geom.replace("out vec4 sfl_semantic_map;", "out vec4 sfl_semantic_map;\nout float sfl_source;", 1)
            geom = geom.replace("EmitVertex();", "sfl_source = float(gl_PrimitiveIDIn);\n    EmitVertex();", 1)
        self.prog = Program({"VERTEX_SHADER": shader_source("update_surfels.vert"), "GEOMETRY_SHADER": geom,
                             "FRAGMENT_SHADER": shader_source("update_surfels.frag")},
                            tf_varyings=self.VARYINGS + (["sfl_source"] if tag_sources else []), from_reference=False)
        self.prog.set(vertex_map=0, normal_map=1, radiusConfidence_map=2, index_map=3, poseBuffer=5, semantic_map_in=6)
        self.fbo = Framebuffer(self.W, self.H)
        g = Context.get()
        self.sampler = gen("Samplers")
        sp = g.fn("glSamplerParameteri", None, u32, u32, i32)
        for pname, val in (("TEXTURE_MIN_FILTER", "NEAREST"), ("TEXTURE_MAG_FILTER", "LINEAR"),
                           ("TEXTURE_WRAP_S", "CLAMP_TO_BORDER"), ("TEXTURE_WRAP_T", "CLAMP_TO_BORDER")):
            sp(self.sampler, GL[pname], GL[val])

    def run(self, uniforms, surfels, poses, frame, radconf, index_map_float):
        """uniforms: [(name, value, kind)] as oracle/pyref.py update_uniforms() lists them.  Returns (the records
        transform feedback wrote, in order; the integration mask (H, W) of the red channel)."""
        g = Context.get()
        W, H = self.W, self.H
        kw = {}
        for name, value, kind in uniforms:
            kw[name] = int(value) if kind == "i" else (np.asarray(value, dtype=np.float32) if kind in "mv" else float(np.float32(value)))
        self.prog.set(**kw)
        surfels = np.ascontiguousarray(surfels)
        n = surfels.shape[0]
        vao = surfel_vao(Buffer(surfels.view(np.uint8)))
        idx4 = np.zeros((H, W, 4), dtype=np.float32)
        idx4[..., 0] = index_map_float  # the reference's texture is R32F: texture().x is all the shader reads
        tex = [RectTexture(W, H, frame[0]), RectTexture(W, H, frame[1]), RectTexture(W, H, radconf), RectTexture(W, H, idx4)]
        sem = RectTexture(W, H, frame[2])
        ptex = BufferTexture(np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 4))
        mask = RectTexture(W, H)  # created BEFORE the units are populated: creating a texture binds it to the active unit
        for unit, t in enumerate(tex):
            t.bind(unit)
        ptex.bind(5)
        sem.bind(6)
        for unit in range(7):
            g.fn("glBindSampler", None, u32, u32)(unit, self.sampler)
        g.fn("glActiveTexture", None, u32)(GL["TEXTURE0"] + 7)
        g.fn("glPointSize", None, f32)(1.0)
        common_state(W, H, "LESS")
        self.fbo.attach([mask])
        words = 17 if self.tagged else 16
        out = Buffer(nbytes=max(n, 1) * 4 * words)
        g.fn("glBindBufferBase", None, u32, u32, u32)(GL["TRANSFORM_FEEDBACK_BUFFER"], 0, out.id)
        q = gen("Queries")
        self.prog.use()
        clear()
        g.fn("glBeginQuery", None, u32, u32)(GL["TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN"], q)
        g.fn("glBeginTransformFeedback", None, u32)(GL["POINTS"])
        draw_points(vao, n)
        g.fn("glEndTransformFeedback", None)()
        g.fn("glEndQuery", None, u32)(GL["TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN"])
        g.fn("glFinish", None)()
        written = u32(0)
        g.fn("glGetQueryObjectuiv", None, u32, u32, C.POINTER(u32))(q, GL["QUERY_RESULT"], C.byref(written))
        for unit in range(7):
            g.fn("glBindSampler", None, u32, u32)(unit, 0)
        g.check("update_surfels draw")
        g.fn("glBindBuffer", None, u32, u32)(GL["TRANSFORM_FEEDBACK_BUFFER"], out.id)
        rec = np.empty(written.value * words, dtype=np.float32)
        if written.value:
            g.fn("glGetBufferSubData", None, u32, C.c_ssize_t, C.c_ssize_t, vp)(GL["TRANSFORM_FEEDBACK_BUFFER"], 0, rec.nbytes,
                                                                               rec.ctypes.data)
        rec = rec.reshape(-1, words)
        if self.tagged:
            return np.ascontiguousarray(rec[:, :16]), mask.read()[..., 0], rec[:, 16].astype(np.uint32)
        return rec, mask.read()[..., 0]


class SurfelGenerate:
    """SurfelMap::updateSurfels, second draw (K10; SurfelMap.cpp:646-664, program :57-66, uniforms :360-376) with the
    reference's gen_surfels.{vert,geom,frag}: one GL_POINT per data texel in the x-major order of vbo_img_coords_
    (:84-93), GL_RASTERIZER_DISCARD, transform feedback of the new surfels.  Sampler objects as in SurfelUpdate."""

    def __init__(self, params):
        self.W, self.H = params.data_width, params.data_height
        # no fragment stage: the draw runs under GL_RASTERIZER_DISCARD, and Mesa's compiler rejects gen_surfels.frag (an
        # integer fragment input without `flat`, which the reference's NVIDIA compiler lets pass)
        self.prog = Program({"VERTEX_SHADER": "gen_surfels.vert", "GEOMETRY_SHADER": "gen_surfels.geom"},
                            tf_varyings=SurfelUpdate.VARYINGS)
        self.prog.set(vertex_map=0, normal_map=1, radiusConfidence_map=2, measurementIntegrated_map=4, semantic_map=6,
                      model_semantic_map=9, prior_map=8)
        g = Context.get()
        self.sampler = gen("Samplers")
        sp = g.fn("glSamplerParameteri", None, u32, u32, i32)
        for pname, val in (("TEXTURE_MIN_FILTER", "NEAREST"), ("TEXTURE_MAG_FILTER", "LINEAR"),
                           ("TEXTURE_WRAP_S", "CLAMP_TO_BORDER"), ("TEXTURE_WRAP_T", "CLAMP_TO_BORDER")):
            sp(self.sampler, GL[pname], GL[val])

    def run(self, uniforms, frame, radconf, integrated4):
        g = Context.get()
        W, H = self.W, self.H
        kw = {}
        for name, value, kind in uniforms:
            kw[name] = int(value) if kind == "i" else (np.asarray(value, dtype=np.float32) if kind in "mv" else float(np.float32(value)))
        self.prog.set(**kw)
        xs, ys = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="ij")
        coords = np.stack([xs + np.float32(0.5), ys + np.float32(0.5)], axis=-1).reshape(-1, 2)  # x-major, SurfelMap.cpp:87-91
        vbo = Buffer(np.ascontiguousarray(coords))
        vao = gen("VertexArrays")
        g.fn("glBindVertexArray", None, u32)(vao)
        g.fn("glBindBuffer", None, u32, u32)(GL["ARRAY_BUFFER"], vbo.id)
        g.fn("glVertexAttribPointer", None, u32, i32, u32, C.c_ubyte, i32, vp)(0, 2, GL["FLOAT"], 0, 8, 0)
        g.fn("glEnableVertexAttribArray", None, u32)(0)
        tex = {0: RectTexture(W, H, frame[0]), 1: RectTexture(W, H, frame[1]), 2: RectTexture(W, H, radconf),
               4: RectTexture(W, H, integrated4), 6: RectTexture(W, H, frame[2])}
        for unit, t in tex.items():
            t.bind(unit)
        for unit in range(7):
            g.fn("glBindSampler", None, u32, u32)(unit, self.sampler)
        g.fn("glActiveTexture", None, u32)(GL["TEXTURE0"] + 7)
        n = W * H
        out = Buffer(nbytes=n * 64)
        g.fn("glBindBufferBase", None, u32, u32, u32)(GL["TRANSFORM_FEEDBACK_BUFFER"], 0, out.id)
        q = gen("Queries")
        self.prog.use()
        g.fn("glDisable", None, u32)(GL["DEPTH_TEST"])
        g.fn("glEnable", None, u32)(GL["RASTERIZER_DISCARD"])
        g.fn("glBeginQuery", None, u32, u32)(GL["TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN"], q)
        g.fn("glBeginTransformFeedback", None, u32)(GL["POINTS"])
        draw_points(vao, n)
        g.fn("glEndTransformFeedback", None)()
        g.fn("glEndQuery", None, u32)(GL["TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN"])
        g.fn("glDisable", None, u32)(GL["RASTERIZER_DISCARD"])
        g.fn("glFinish", None)()
        written = u32(0)
        g.fn("glGetQueryObjectuiv", None, u32, u32, C.POINTER(u32))(q, GL["QUERY_RESULT"], C.byref(written))
        for unit in range(7):
            g.fn("glBindSampler", None, u32, u32)(unit, 0)
        g.check("gen_surfels draw")
        g.fn("glBindBuffer", None, u32, u32)(GL["TRANSFORM_FEEDBACK_BUFFER"], out.id)
        rec = np.empty(written.value * 16, dtype=np.float32)
        if written.value:
            g.fn("glGetBufferSubData", None, u32, C.c_ssize_t, C.c_ssize_t, vp)(GL["TRANSFORM_FEEDBACK_BUFFER"], 0, rec.nbytes,
                                                                               rec.ctypes.data)
        return rec.reshape(-1, 16)


class SurfelFilter:
    """SurfelMap::copySurfels (K11, copy_surfels.vert + copy_surfels.geom, SurfelMap.cpp:160-166, 667-698) and
    SurfelMap::extractSurfels (K12, extract_surfels.vert + copy_surfels.geom, :280-286, 708-742): the surfel buffer as
    GL_POINTS, GL_RASTERIZER_DISCARD, transform feedback of the records the vertex stage marks valid.  One feedback
    object spans several draws (copySurfels appends the new surfels behind the updated ones)."""

    def __init__(self, vertex_shader):
        self.prog = Program({"VERTEX_SHADER": vertex_shader, "GEOMETRY_SHADER": "copy_surfels.geom",
                             "FRAGMENT_SHADER": "empty.frag"}, tf_varyings=SurfelUpdate.VARYINGS)
        self.prog.set(poseBuffer=5)

    def run(self, buffers, poses, center, extent):
        g = Context.get()
        self.prog.set(submap_center=np.asarray(center, dtype=np.float32), submap_extent=float(np.float32(extent)))
        ptex = BufferTexture(np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 4))
        total = sum(b.shape[0] for b in buffers)
        out = Buffer(nbytes=max(total, 1) * 64)
        vaos = [(surfel_vao(Buffer(np.ascontiguousarray(b).view(np.uint8))), b.shape[0]) for b in buffers]
        ptex.bind(5)
        g.fn("glBindBufferBase", None, u32, u32, u32)(GL["TRANSFORM_FEEDBACK_BUFFER"], 0, out.id)
        q = gen("Queries")
        self.prog.use()
        g.fn("glEnable", None, u32)(GL["RASTERIZER_DISCARD"])
        g.fn("glBeginQuery", None, u32, u32)(GL["TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN"], q)
        g.fn("glBeginTransformFeedback", None, u32)(GL["POINTS"])
        for vao, n in vaos:
            if n:
                draw_points(vao, n)
        g.fn("glEndTransformFeedback", None)()
        g.fn("glEndQuery", None, u32)(GL["TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN"])
        g.fn("glDisable", None, u32)(GL["RASTERIZER_DISCARD"])
        g.fn("glFinish", None)()
        written = u32(0)
        g.fn("glGetQueryObjectuiv", None, u32, u32, C.POINTER(u32))(q, GL["QUERY_RESULT"], C.byref(written))
        g.check("surfel filter draw")
        g.fn("glBindBuffer", None, u32, u32)(GL["TRANSFORM_FEEDBACK_BUFFER"], out.id)
        rec = np.empty(written.value * 16, dtype=np.float32)
        if written.value:
            g.fn("glGetBufferSubData", None, u32, C.c_ssize_t, C.c_ssize_t, vp)(GL["TRANSFORM_FEEDBACK_BUFFER"], 0, rec.nbytes,
                                                                               rec.ctypes.data)
        return rec.reshape(-1, 16)


def eval_vertex_shader(vs_text, attribs, out_components, out_name="r", prelude="driver"):
    """run an OWN one-stage shader over arrays of float attributes and capture ONE vecN output per vertex through
    transform feedback (rasteriser discarded): how tests look at the GL implementation's built-in functions value by
    value.  attribs: {attribute name: float32 array}; prelude: "driver" | "detmath" (transcendentals, above)."""
    import re
    g = Context.get()
    text = vs_text
    pre = {"driver": "", "detmath": DETMATH_PRELUDE}[prelude]
    if pre:
        text = re.sub(r"^(\s*#version[^\n]*\n)", lambda m: m.group(1) + "#extension GL_ARB_gpu_shader5 : require\n" + pre, text, count=1)
    prog = Program({"VERTEX_SHADER": text}, tf_varyings=[out_name], from_reference=False)
    n = len(next(iter(attribs.values())))
    vao = gen("VertexArrays")
    g.fn("glBindVertexArray", None, u32)(vao)
    keep = []
    for name, arr in attribs.items():
        loc = g.fn("glGetAttribLocation", i32, u32, C.c_char_p)(prog.id, name.encode())
        if loc < 0:
            continue
        b = Buffer(np.ascontiguousarray(arr, dtype=np.float32))
        keep.append(b)
        g.fn("glBindBuffer", None, u32, u32)(GL["ARRAY_BUFFER"], b.id)
        g.fn("glVertexAttribPointer", None, u32, i32, u32, C.c_ubyte, i32, vp)(loc, 1, GL["FLOAT"], 0, 4, 0)
        g.fn("glEnableVertexAttribArray", None, u32)(loc)
    out = Buffer(nbytes=max(n, 1) * 4 * out_components)
    g.fn("glBindBufferBase", None, u32, u32, u32)(GL["TRANSFORM_FEEDBACK_BUFFER"], 0, out.id)
    prog.use()
    g.fn("glEnable", None, u32)(GL["RASTERIZER_DISCARD"])
    g.fn("glBeginTransformFeedback", None, u32)(GL["POINTS"])
    draw_points(vao, n)
    g.fn("glEndTransformFeedback", None)()
    g.fn("glDisable", None, u32)(GL["RASTERIZER_DISCARD"])
    g.fn("glFinish", None)()
    g.check("eval_vertex_shader")
    res = np.empty(n * out_components, dtype=np.float32)
    g.fn("glBindBuffer", None, u32, u32)(GL["TRANSFORM_FEEDBACK_BUFFER"], out.id)
    g.fn("glGetBufferSubData", None, u32, C.c_ssize_t, C.c_ssize_t, vp)(GL["TRANSFORM_FEEDBACK_BUFFER"], 0, res.nbytes, res.ctypes.data)
    return res.reshape(n, out_components)


def _map_sampler():
    """the map's sampler object, SurfelMap.cpp:168-170: MIN NEAREST / MAG LINEAR / CLAMP_TO_BORDER"""
    g = Context.get()
    s = gen("Samplers")
    sp = g.fn("glSamplerParameteri", None, u32, u32, i32)
    for pname, val in (("TEXTURE_MIN_FILTER", "NEAREST"), ("TEXTURE_MAG_FILTER", "LINEAR"),
                       ("TEXTURE_WRAP_S", "CLAMP_TO_BORDER"), ("TEXTURE_WRAP_T", "CLAMP_TO_BORDER")):
        sp(s, GL[pname], GL[val])
    return s


class RadiusConfidence:
    """SurfelMap::generateDataSurfels (K8; SurfelMap.cpp:606-619, framebuffer :116-126, uniforms :380-397) with the
    reference's init_radiusConf.{vert,frag}: one GL_POINT per data texel at its centre, two colour attachments written
    (centred vertex, radius / confidence), depth test LESS against a cleared buffer, the map's sampler on units 0 / 1."""

    def __init__(self, params):
        self.W, self.H = params.data_width, params.data_height
        self.prog = Program({"VERTEX_SHADER": "init_radiusConf.vert", "FRAGMENT_SHADER": "init_radiusConf.frag"})
        self.prog.set(vertex_map=0, normal_map=1)
        self.fbo = Framebuffer(self.W, self.H)
        self.sampler = _map_sampler()

    def run(self, uniforms, vmap, nmap):
        g = Context.get()
        W, H = self.W, self.H
        self.prog.set(**uniforms)
        xs, ys = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="ij")
        coords = np.stack([xs + np.float32(0.5), ys + np.float32(0.5)], axis=-1).reshape(-1, 2)
        vbo = Buffer(np.ascontiguousarray(coords))
        vao = gen("VertexArrays")
        g.fn("glBindVertexArray", None, u32)(vao)
        g.fn("glBindBuffer", None, u32, u32)(GL["ARRAY_BUFFER"], vbo.id)
        g.fn("glVertexAttribPointer", None, u32, i32, u32, C.c_ubyte, i32, vp)(0, 2, GL["FLOAT"], 0, 8, 0)
        g.fn("glEnableVertexAttribArray", None, u32)(0)
        outs = [RectTexture(W, H), RectTexture(W, H)]
        tv, tn = RectTexture(W, H, vmap), RectTexture(W, H, nmap)
        tv.bind(0)
        tn.bind(1)
        for unit in (0, 1):
            g.fn("glBindSampler", None, u32, u32)(unit, self.sampler)
        g.fn("glActiveTexture", None, u32)(GL["TEXTURE0"] + 7)
        g.fn("glPointSize", None, f32)(1.0)
        common_state(W, H, "LESS")
        self.fbo.attach(outs)
        self.prog.use()
        clear()
        draw_points(vao, W * H)
        g.fn("glFinish", None)()
        for unit in (0, 1):
            g.fn("glBindSampler", None, u32, u32)(unit, 0)
        g.check("init_radiusConf draw")
        return outs[1].read(), outs[0].read()


class Compose:
    """the compose pass of SurfelMap::render (K5; SurfelMap.cpp:249-263 program, :911-961 draw) with the reference's
    empty.vert + quad.geom + render_compose.frag: old / new vertex, normal and semantic maps on units 0-4 and 6 under the
    map's sampler (MAG LINEAR at a 1:1 mapping), three colour attachments."""

    def __init__(self, params):
        self.W, self.H = params.model_width, params.model_height
        self.prog = Program({"VERTEX_SHADER": "empty.vert", "GEOMETRY_SHADER": "quad.geom", "FRAGMENT_SHADER": "render_compose.frag"})
        self.prog.set(old_vertexmap=0, old_normalmap=1, new_vertexmap=2, new_normalmap=3, poseBuffer=5, new_semanticmap=4,
                      old_semanticmap=6, max_distance=float(np.float32(params.max_loop_closure_distance)))
        self.fbo = Framebuffer(self.W, self.H)
        self.sampler = _map_sampler()
        self.vao = gen("VertexArrays")

    def run(self, old, new):
        g = Context.get()
        W, H = self.W, self.H
        outs = [RectTexture(W, H) for _ in range(3)]
        units = {0: old[0], 1: old[1], 2: new[0], 3: new[1], 4: new[2], 6: old[2]}
        tex = {u_: RectTexture(W, H, a) for u_, a in units.items()}
        for u_, t in tex.items():
            t.bind(u_)
            g.fn("glBindSampler", None, u32, u32)(u_, self.sampler)
        g.fn("glActiveTexture", None, u32)(GL["TEXTURE0"] + 7)
        common_state(W, H, "LESS")
        self.fbo.attach(outs)
        self.prog.use()
        clear()
        draw_points(self.vao, 1)
        g.fn("glFinish", None)()
        for u_ in tex:
            g.fn("glBindSampler", None, u32, u32)(u_, 0)
        g.check("render_compose draw")
        return [o.read() for o in outs]
