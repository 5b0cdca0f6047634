/*
 * tools/gl_interop_probe.cpp -- round-4 probe for SURVEY.md 8(f)-2 (HIP -> GL hand-over for the untouched viewer,
 * SurfelMap.cpp:1167-1230, ViewportWidget.cpp:404-434): what GL does the MI355X box offer, and does the HIP runtime
 * accept a buffer object of it?  Steps, each reported:
 *   1. /dev/dri render nodes, the GL / EGL / GBM libraries the loader knows;
 *   2. a GL context WITHOUT a window system through Mesa's DRI software-rasteriser interface (oracle/glref/gl_ctx.c,
 *      the context the GL-backed parity tests use): version, renderer;
 *   3. a real buffer object of that context (glGenBuffers / glBufferData, 1 MB) handed to hipGraphicsGLRegisterBuffer;
 *      if the registration succeeds, the surfels of a small map are copied into it as examples/gl_interop.cpp does and
 *      read back through glGetBufferSubData for comparison with suma_map_download.
 * HIP's GL interop (ROCclr) asks the CURRENT GLX or EGL context for its device through Mesa's interop entry points
 * (MesaGLInteropGLXQueryDeviceInfo / ...EGL...); a context made through the raw DRI interface is neither, and llvmpipe
 * buffers live in host memory anyway -- so step 3 is expected to be refused.  The log says what was found.
 * build: g++ -std=c++11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/gl_interop_probe.cpp -o probe \
 *            -Lsemantic_suma_amd -lsuma_hip -L/opt/rocm/lib -lamdhip64 -ldl
 */
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <hip/hip_gl_interop.h>

#include "suma_hip.h"

typedef int (*ctx_create_fn)(const char*);
typedef const char* (*ctx_error_fn)(void);
typedef void* (*ctx_proc_fn)(const char*);

int main(int argc, char** argv) {
  const char* shim = argc > 1 ? argv[1] : "oracle/_ref/libsuma_glctx.so";
  std::printf("== 1. devices and libraries\n");
  std::fflush(stdout);
  std::system("ls -l /dev/dri 2>&1 | head -8; ldconfig -p | grep -i -E 'libEGL|libgbm|libOSMesa|libGLX_|libGL\\.so' | head -8; "
              "ls /usr/lib/x86_64-linux-gnu/dri 2>/dev/null | tr '\\n' ' '; echo");
  std::printf("== 2. GL context through the DRI software-rasteriser interface (%s)\n", shim);
  void* h = dlopen(shim, RTLD_NOW);
  if (!h) {
    std::printf("shim not loadable: %s\n", dlerror());
    return 0;
  }
  ctx_create_fn create = (ctx_create_fn)dlsym(h, "gl_ctx_create");
  ctx_error_fn cerr = (ctx_error_fn)dlsym(h, "gl_ctx_error");
  ctx_proc_fn proc = (ctx_proc_fn)dlsym(h, "gl_ctx_proc");
  if (!create || create("") != 0) {
    std::printf("no context: %s\n", cerr ? cerr() : "?");
    return 0;
  }
  typedef const unsigned char* (*get_string_fn)(unsigned);
  get_string_fn get_string = (get_string_fn)proc("glGetString");
  std::printf("GL_VERSION %s | GL_RENDERER %s\n", get_string(0x1F02), get_string(0x1F01));
  std::printf("== 3. a buffer object of that context -> hipGraphicsGLRegisterBuffer\n");
  typedef void (*gen_fn)(int, unsigned*);
  typedef void (*bind_fn)(unsigned, unsigned);
  typedef void (*data_fn)(unsigned, long, const void*, unsigned);
  typedef void (*getsub_fn)(unsigned, long, long, void*);
  unsigned vbo = 0;
  ((gen_fn)proc("glGenBuffers"))(1, &vbo);
  ((bind_fn)proc("glBindBuffer"))(0x8892, vbo);
  ((data_fn)proc("glBufferData"))(0x8892, 1 << 20, nullptr, 0x88EA);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::printf("no HIP device here: the registration cannot be tried\n");
    return 0;
  }
  unsigned int gl_devs = 0;
  int devs[8];
  hipError_t e = hipGLGetDevices(&gl_devs, devs, 8, hipGLDeviceListAll);
  std::printf("hipGLGetDevices: %s (%u devices)\n", hipGetErrorName(e), gl_devs);
  (void)hipGetLastError();
  hipGraphicsResource_t res = nullptr;
  e = hipGraphicsGLRegisterBuffer(&res, vbo, hipGraphicsRegisterFlagsWriteDiscard);
  std::printf("hipGraphicsGLRegisterBuffer(vbo %u): %s\n", vbo, hipGetErrorName(e));
  (void)hipGetLastError();
  if (e != hipSuccess) {
    std::printf("RESULT: no HIP <-> GL interop on this machine (a raw-DRI llvmpipe context is not a GLX / EGL context of the "
                "GPU's driver); f2 stays code that compiles and links (examples/gl_interop.cpp), not a measured path\n");
    return 0;
  }
  /* registered after all: copy a map into the VBO and compare */
  suma_params p;
  suma_params_default(&p);
  p.data_width = p.model_width = 360;
  p.data_height = p.model_height = 32;
  suma_ctx* ctx = nullptr;
  if (suma_ctx_create(&p, 0, &ctx) != SUMA_OK) return 1;
  std::vector<suma_surfel> host(1000);
  std::memset(host.data(), 0, host.size() * sizeof(suma_surfel));
  for (size_t k = 0; k < host.size(); ++k) host[k].x = (float)k;
  suma_map_upload(ctx, host.data(), (uint32_t)host.size(), 1);
  hipStream_t stream = (hipStream_t)suma_ctx_stream(ctx);
  void* d_src = nullptr;
  uint32_t n = 0;
  suma_map_export_surfels(ctx, &d_src, &n);
  void* d_vbo = nullptr;
  size_t bytes = 0;
  bool ok = hipGraphicsMapResources(1, &res, stream) == hipSuccess &&
            hipGraphicsResourceGetMappedPointer(&d_vbo, &bytes, res) == hipSuccess &&
            hipMemcpyAsync(d_vbo, d_src, n * sizeof(suma_surfel), hipMemcpyDeviceToDevice, stream) == hipSuccess &&
            hipGraphicsUnmapResources(1, &res, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
  std::vector<suma_surfel> back(n);
  if (ok) ((getsub_fn)proc("glGetBufferSubData"))(0x8892, 0, (long)(n * sizeof(suma_surfel)), back.data());
  ok = ok && std::memcmp(back.data(), host.data(), n * sizeof(suma_surfel)) == 0;
  std::printf("RESULT: interop copy %s (%u surfels)\n", ok ? "matches suma_map_download" : "FAILED", n);
  suma_ctx_destroy(ctx);
  return ok ? 0 : 1;
}
