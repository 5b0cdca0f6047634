/*
 * gl_ctx.c -- an OpenGL 3.3+ core context WITHOUT a window system, for the test infrastructure only.
 *
 * Neither the build container nor the GPU box has an X server, EGL or OSMesa, which is why rounds 1-3 called the
 * fixed-function half of the reference's render passes "unpinned" (point / triangle rasterisation, depth test,
 * transform-feedback order, texture filtering: GL behaviour that lives in the driver, not in src/shader).  But the
 * image does ship Mesa's DRI drivers, and swrast_dri.so (llvmpipe, a conformant software GL 4.5) can be driven
 * directly through the DRI software-rasteriser interface (GL/internal/dri_interface.h: __DRI_CORE + __DRI_SWRAST with
 * a loader that never presents anything) -- the way GBM / OSMesa do internally.  With that, the reference's OWN GLSL
 * (/root/reference/src/shader/ *.vert / .geom / .frag, read where it lies) runs in a real GL implementation here, and
 * oracle/glref.py replays the draw calls of src/core/SurfelMap.cpp / Preprocessing.cpp around it.
 *
 * This file only creates the context and hands out GL entry points; it contains no algorithm of the hot path and is
 * never linked into the product.  Built by oracle/ref_build.py into oracle/_ref/libsuma_glctx.so (git-ignored).
 */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <GL/gl.h>
#include <GL/internal/dri_interface.h>

static void get_drawable_info(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* priv) {
  (void)d; (void)priv;
  *x = *y = 0;
  *w = *h = 16;
}
static void put_image(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* priv) {
  (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)priv;
}
static void get_image(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* priv) {
  (void)d; (void)x; (void)y; (void)priv;
  memset(data, 0, (size_t)w * (size_t)h * 4);
}
static const __DRIswrastLoaderExtension swrast_loader = {
    .base = {__DRI_SWRAST_LOADER, 1},
    .getDrawableInfo = get_drawable_info,
    .putImage = put_image,
    .getImage = get_image,
};
static const __DRIextension* loader_exts[] = {&swrast_loader.base, NULL};

static void* (*g_gpa)(const char*) = NULL;
static char g_error[256] = "";

const char* gl_ctx_error(void) { return g_error; }

/* 0 on success; the context stays current on the calling thread for the life of the process */
int gl_ctx_create(const char* dri_path) {
  if (g_gpa) return 0;
  const char* paths[] = {dri_path, "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", "/usr/lib64/dri/swrast_dri.so", NULL};
  void* h = NULL;
  for (int i = 0; !h && i < 3; ++i)
    if (paths[i] && paths[i][0]) h = dlopen(paths[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    snprintf(g_error, sizeof(g_error), "swrast_dri.so: %s", dlerror());
    return -1;
  }
  const __DRIextension** (*get)(void) = (const __DRIextension** (*)(void))dlsym(h, "__driDriverGetExtensions_swrast");
  if (!get) {
    snprintf(g_error, sizeof(g_error), "__driDriverGetExtensions_swrast not exported");
    return -2;
  }
  const __DRIextension** exts = get();
  const __DRIcoreExtension* core = NULL;
  const __DRIswrastExtension* sw = NULL;
  for (int i = 0; exts[i]; ++i) {
    if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension*)exts[i];
    if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = (const __DRIswrastExtension*)exts[i];
  }
  if (!core || !sw || sw->base.version < 4) {
    snprintf(g_error, sizeof(g_error), "DRI_Core / DRI_SWRast (v4) not offered by the driver");
    return -3;
  }
  const __DRIconfig** configs = NULL;
  __DRIscreen* scr = sw->createNewScreen2(0, loader_exts, exts, &configs, NULL);
  if (!scr || !configs || !configs[0]) {
    snprintf(g_error, sizeof(g_error), "createNewScreen2 failed");
    return -4;
  }
  unsigned attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 3};
  unsigned err = 0;
  __DRIcontext* ctx = sw->createContextAttribs(scr, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
  if (!ctx) {
    snprintf(g_error, sizeof(g_error), "createContextAttribs(3.3 core) failed: %u", err);
    return -5;
  }
  __DRIdrawable* dr = sw->createNewDrawable(scr, configs[0], NULL);
  if (!dr || !core->bindContext(ctx, dr, dr)) {
    snprintf(g_error, sizeof(g_error), "createNewDrawable / bindContext failed");
    return -6;
  }
  void* glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
  if (!glapi) {
    snprintf(g_error, sizeof(g_error), "libglapi.so.0: %s", dlerror());
    return -7;
  }
  g_gpa = (void* (*)(const char*))dlsym(glapi, "_glapi_get_proc_address");
  if (!g_gpa) {
    snprintf(g_error, sizeof(g_error), "_glapi_get_proc_address not exported");
    return -8;
  }
  return 0;
}

void* gl_ctx_proc(const char* name) { return g_gpa ? g_gpa(name) : NULL; }
