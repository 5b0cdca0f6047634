/*
 * examples/gl_interop.cpp -- the HIP -> GL hand-over of INTEGRATION.md 2a as code that compiles and links.
 *
 * The reference's visualizer draws the surfel VBO (SurfelMap::draw, src/core/SurfelMap.cpp:1167-1230) and binds
 * Frame::vertex_map / normal_map / semantic_map as rectangle textures (src/visualizer/ViewportWidget.cpp:404-434).
 * With the map and the frames living in HIP memory, the viewer's GL objects are registered once with the HIP runtime
 * and refreshed by device-to-device copies on the ctx stream -- no host copy.  This file is what a maintainer pastes
 * into the GL thread.  It needs a current GL context to DO anything: neither the build container nor the GPU box of
 * this project has one, so the test suite compiles it, links it against libamdhip64 + libsuma_hip, and runs main(),
 * which calls the registration with a buffer name that cannot exist and checks that the failure is reported, not
 * crashed on (tests/test_abi.py, tests/test_gpu_cpp.py).
 *
 * build: g++ -std=c++11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/gl_interop.cpp \
 *            -Lsemantic_suma_amd -lsuma_hip -L/opt/rocm/lib -lamdhip64
 */
#include <cstdio>

#include <hip/hip_runtime_api.h>
#include <hip/hip_gl_interop.h>

#include "suma_hip.h"

#ifndef GL_TEXTURE_RECTANGLE
#define GL_TEXTURE_RECTANGLE 0x84F5
#endif

namespace suma_gl {

/* the viewer's GL objects, registered once (in the thread that owns the GL context) */
struct ViewerLink {
  hipGraphicsResource_t surfel_vbo = nullptr;
  hipGraphicsResource_t map_tex[3] = {nullptr, nullptr, nullptr}; /* vertex / normal / semantic rectangle textures */
};

/* glow::GlBuffer<Surfel>::id() of SurfelMap::surfels_, GlTextureRectangle::id() of the three maps of a Frame */
inline hipError_t register_viewer(ViewerLink* link, GLuint surfel_vbo, const GLuint map_textures[3]) {
  hipError_t e = hipGraphicsGLRegisterBuffer(&link->surfel_vbo, surfel_vbo, hipGraphicsRegisterFlagsWriteDiscard);
  for (int m = 0; m < 3 && e == hipSuccess; ++m)
    e = hipGraphicsGLRegisterImage(&link->map_tex[m], map_textures[m], GL_TEXTURE_RECTANGLE,
                                   hipGraphicsRegisterFlagsWriteDiscard);
  return e;
}

inline void unregister_viewer(ViewerLink* link) {
  if (link->surfel_vbo) hipGraphicsUnregisterResource(link->surfel_vbo);
  for (int m = 0; m < 3; ++m)
    if (link->map_tex[m]) hipGraphicsUnregisterResource(link->map_tex[m]);
  *link = ViewerLink();
}

/* per displayed frame: the active map into the viewer's VBO (64-byte records, the VAO layout of SurfelMap.cpp:46-55);
 * returns the number of surfels to draw in *n.  Enqueued on the ctx stream, i.e. behind the update that produced them. */
inline hipError_t refresh_surfels(suma_ctx* ctx, ViewerLink* link, uint32_t* n) {
  hipStream_t stream = (hipStream_t)suma_ctx_stream(ctx);
  void* d_src = nullptr;
  if (suma_map_export_surfels(ctx, &d_src, n) != SUMA_OK) return hipErrorUnknown;
  hipError_t e = hipGraphicsMapResources(1, &link->surfel_vbo, stream);
  if (e != hipSuccess) return e;
  void* d_vbo = nullptr;
  size_t bytes = 0;
  e = hipGraphicsResourceGetMappedPointer(&d_vbo, &bytes, link->surfel_vbo);
  if (e == hipSuccess) {
    size_t want = (size_t)*n * sizeof(suma_surfel);
    if (want > bytes) want = bytes; /* the VBO was sized for maxNumSurfels_ (SurfelMap.cpp:42) */
    e = hipMemcpyAsync(d_vbo, d_src, want, hipMemcpyDeviceToDevice, stream);
  }
  hipError_t u = hipGraphicsUnmapResources(1, &link->surfel_vbo, stream);
  return e != hipSuccess ? e : u;
}

/* the three maps of a frame into the viewer's rectangle textures (RGBA32F, row 0 = lowest beam = GL's bottom row) */
inline hipError_t refresh_frame(suma_ctx* ctx, ViewerLink* link, const suma_frame* frame) {
  hipStream_t stream = (hipStream_t)suma_ctx_stream(ctx);
  for (int m = 0; m < 3; ++m) {
    void* d_map = nullptr;
    uint32_t w = 0, h = 0, row_bytes = 0;
    if (suma_frame_export(ctx, frame, m, &d_map, &w, &h, &row_bytes) != SUMA_OK) return hipErrorUnknown;
    hipError_t e = hipGraphicsMapResources(1, &link->map_tex[m], stream);
    if (e != hipSuccess) return e;
    hipArray_t arr = nullptr;
    e = hipGraphicsSubResourceGetMappedArray(&arr, link->map_tex[m], 0, 0);
    if (e == hipSuccess) e = hipMemcpy2DToArrayAsync(arr, 0, 0, d_map, row_bytes, row_bytes, h, hipMemcpyDeviceToDevice, stream);
    hipError_t u = hipGraphicsUnmapResources(1, &link->map_tex[m], stream);
    if (e != hipSuccess) return e;
    if (u != hipSuccess) return u;
  }
  return hipSuccess;
}

}  // namespace suma_gl

int main() {
  /* no GL context here: the registration must fail with an error code (and nothing may crash) */
  suma_gl::ViewerLink link;
  const GLuint textures[3] = {0xdead0001u, 0xdead0002u, 0xdead0003u};
  const hipError_t e = suma_gl::register_viewer(&link, 0xdead0000u, textures);
  (void)hipGetLastError();
  std::printf("hipGraphicsGLRegisterBuffer without a GL context: %s\n", e == hipSuccess ? "registered (?)" : hipGetErrorName(e));
  suma_gl::unregister_viewer(&link);
  /* keeps the refresh paths referenced so that they are compiled and linked */
  std::printf("refresh entry points at %p %p\n", (void*)&suma_gl::refresh_surfels, (void*)&suma_gl::refresh_frame);
  return e == hipSuccess ? 1 : 0;
}
