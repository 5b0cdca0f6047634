/*
 * oracle/ref_driver.cpp -- TEST INFRASTRUCTURE.  Never shipped, never linked by the product.
 *
 * Runs the reference's own GLSL shaders (compiled to C++ by oracle/ref_build.py from
 * /root/reference/src/shader; see oracle/glsl_compat.hpp) on the CPU.  This file restates only what the
 * OpenGL 3.3 fixed-function pipeline does AROUND the shaders for the draw calls of the hot path:
 *   vertex fetch from the bound VBOs (attribute layout: SurfelMap.cpp:46-55, Preprocessing.cpp:140-145,
 *   Frame2Model.cpp:25-36), VS -> GS hand-over of interface blocks, point clipping against the clip volume,
 *   viewport transform, point rasterisation (size 1 -> the pixel that contains the point), 24-bit depth test
 *   GL_LESS in draw order, fragment outputs to the colour attachments, GL_ONE/GL_ONE blending (K6),
 *   transform-feedback capture in draw order (K9 / K10 / K11 / K12).
 * Triangle rasterisation (K4) is NOT restated here: ref_render_quads returns what the geometry shader emits
 * (gate, the four clip-space corners, the flat attributes); coverage is GL-implementation-defined and
 * modelled in oracle/o_map.c.
 *
 * Uniform values, texture bindings and sampler state are set by the caller (tests/test_ref_shaders.py),
 * which cites the host line of the reference each of them comes from.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>

#include "glsl_compat.hpp"
#include "../include/suma_types.h"

enum { K_FLOAT = 1, K_INT, K_BOOL, K_VEC2, K_VEC3, K_VEC4, K_MAT4, K_SAMPLER_RECT, K_SAMPLER_BUFFER };
#include "shaders_gen.inc" /* oracle/_ref/shaders_gen.inc, generated */

#include "lie_algebra.h" /* the reference's header, resolved through -I <reference>/src/core */

using namespace glsl;

static bool stage_of(const char* stage, const char* prog) {
  size_t n = strlen(prog);
  return strncmp(stage, prog, n) == 0 && stage[n] == '.';
}

extern "C" {

/* glUniform*: sets `name` in every stage of program `prog` (= shader base name) that declares it.  data: doubles
 * (1 scalar, 2/3/4 components, or 16 column-major matrix elements), converted to the declared GLSL type -- every
 * value the caller passes is exactly representable.  Returns the number of stages hit. */
int ref_set_uniform(const char* prog, const char* name, const double* d) {
  int hits = 0;
  for (const uniform_entry* e = uniform_table; e->stage; ++e) {
    if (!stage_of(e->stage, prog) || strcmp(e->name, name) != 0) continue;
    switch (e->kind) {
      case K_FLOAT: *(float*)e->ptr = (float)d[0]; break;
      case K_INT: *(int*)e->ptr = (int)d[0]; break;
      case K_BOOL: *(bool*)e->ptr = d[0] != 0.0; break;
      case K_VEC2: *(vec2*)e->ptr = vec2((float)d[0], (float)d[1]); break;
      case K_VEC3: *(vec3*)e->ptr = vec3((float)d[0], (float)d[1], (float)d[2]); break;
      case K_VEC4: *(vec4*)e->ptr = vec4((float)d[0], (float)d[1], (float)d[2], (float)d[3]); break;
      case K_MAT4: {
        mat4& m = *(mat4*)e->ptr;
        for (int c = 0; c < 4; ++c) m[c] = vec4((float)d[4 * c], (float)d[4 * c + 1], (float)d[4 * c + 2], (float)d[4 * c + 3]);
        break;
      }
      default: continue;
    }
    ++hits;
  }
  return hits;
}

/* glBindTexture + sampler state for a rectangle texture (ch = 4: RGBA32F, 1: R32F); filter 0 NEAREST, 1 LINEAR;
 * wrap is CLAMP_TO_BORDER, border colour 0, everywhere on the hot path */
int ref_bind_texture(const char* prog, const char* name, const float* data, int w, int h, int ch, int filter) {
  int hits = 0;
  for (const uniform_entry* e = uniform_table; e->stage; ++e) {
    if (e->kind != K_SAMPLER_RECT || !stage_of(e->stage, prog) || strcmp(e->name, name) != 0) continue;
    sampler2DRect& s = *(sampler2DRect*)e->ptr;
    s.data = data;
    s.w = w;
    s.h = h;
    s.ch = ch;
    s.filter = filter & 1;
    s.edge = (filter & GLSL_CLAMP_TO_EDGE) != 0;
    ++hits;
  }
  return hits;
}
/* texture buffer (poseBuffer: 4 RGBA32F texels per pose, SurfelMap.h:205-208) */
int ref_bind_buffer(const char* prog, const char* name, const float* data, int n_texels) {
  int hits = 0;
  for (const uniform_entry* e = uniform_table; e->stage; ++e) {
    if (e->kind != K_SAMPLER_BUFFER || !stage_of(e->stage, prog) || strcmp(e->name, name) != 0) continue;
    samplerBuffer& s = *(samplerBuffer*)e->ptr;
    s.data = data;
    s.n = n_texels;
    ++hits;
  }
  return hits;
}

}  // extern "C"

/* ---- fixed function ---- */
struct raster_point {
  int px, py;
  uint32_t z24;
};
/* point primitive: clip against -w <= x,y,z <= w, perspective divide, viewport (0,0,W,H), depth range [0,1] */
static bool rasterize_point(const vec4& clip, int W, int H, raster_point* out) {
  const float w = clip.w;
  if (!(clip.x >= -w && clip.x <= w && clip.y >= -w && clip.y <= w && clip.z >= -w && clip.z <= w)) return false;
  const float xd = clip.x / w, yd = clip.y / w, zd = clip.z / w;
  const float xw = (0.5f * (float)W) * xd + 0.5f * (float)W;
  const float yw = (0.5f * (float)H) * yd + 0.5f * (float)H;
  const float zw = 0.5f * zd + 0.5f;
  int px = (int)std::floor(xw), py = (int)std::floor(yw);
  if (px < 0 || py < 0 || px >= W || py >= H) return false; /* on the far clip edge: no pixel centre covered */
  out->px = px;
  out->py = py;
  out->z24 = (uint32_t)__builtin_rintf(zw * 16777215.0f); /* GL_DEPTH24_STENCIL8: unorm24, nearest-even of the fp32 product (pinned against llvmpipe, oracle/glref.py) */
  return true;
}

static void store4(float* map, size_t pix, const vec4& v) {
  map[4 * pix] = v.x;
  map[4 * pix + 1] = v.y;
  map[4 * pix + 2] = v.z;
  map[4 * pix + 3] = v.w;
}

extern "C" {

/* ------------------------------------------------------------------------------------------------------
 * K1  Preprocessing::process, first pass (Preprocessing.cpp:120-189): gen_vertexmap.vert + .frag, GL_POINTS,
 * depth test LESS, clear colour 0.  label / prob are the attribute values AS FETCHED (the caller applies the
 * byte offsets of Preprocessing.cpp:142-145). */
void ref_draw_vertexmap(const float* pts4, const float* label, const float* prob, uint32_t n, int W, int H, float* vmap4,
                        float* smap4) {
  namespace VS = s_gen_vertexmap_vert;
  namespace FS = s_gen_vertexmap_frag;
  const size_t P = (size_t)W * H;
  memset(vmap4, 0, P * 16);
  memset(smap4, 0, P * 16);
  uint32_t* depth = (uint32_t*)malloc(P * 4);
  for (size_t i = 0; i < P; ++i) depth[i] = 0xffffffu; /* glClear: depth 1.0 */
  for (uint32_t i = 0; i < n; ++i) {
    VS::position = vec4(pts4[4 * i], pts4[4 * i + 1], pts4[4 * i + 2], pts4[4 * i + 3]);
    VS::label = label[i];
    VS::prob = prob[i];
    VS::gl_VertexID = (int)i;
    VS::shader_main();
    raster_point rp;
    if (!rasterize_point(VS::gl_Position, W, H, &rp)) continue;
    size_t pix = (size_t)rp.py * W + rp.px;
    if (!(rp.z24 < depth[pix])) continue;
    depth[pix] = rp.z24;
    FS::vertex_coord = VS::vertex_coord;
    FS::vert_label = VS::vert_label;
    FS::vert_label_prob = VS::vert_label_prob;
    FS::shader_main();
    store4(vmap4, pix, FS::color);
    store4(smap4, pix, FS::semantic_map);
  }
  free(depth);
}

/* K1 with avgVertexmap_ (Preprocessing.cpp:150,160-166): the same draw with the depth test disabled and
 * glBlendFunc(GL_ONE, GL_ONE) on both attachments; fragments blend in primitive order. */
void ref_draw_vertexmap_blend(const float* pts4, const float* label, const float* prob, uint32_t n, int W, int H,
                              float* vmap4, float* smap4) {
  namespace VS = s_gen_vertexmap_vert;
  namespace FS = s_gen_vertexmap_frag;
  const size_t P = (size_t)W * H;
  memset(vmap4, 0, P * 16);
  memset(smap4, 0, P * 16);
  for (uint32_t i = 0; i < n; ++i) {
    VS::position = vec4(pts4[4 * i], pts4[4 * i + 1], pts4[4 * i + 2], pts4[4 * i + 3]);
    VS::label = label[i];
    VS::prob = prob[i];
    VS::gl_VertexID = (int)i;
    VS::shader_main();
    raster_point rp;
    if (!rasterize_point(VS::gl_Position, W, H, &rp)) continue;
    size_t pix = (size_t)rp.py * W + rp.px;
    FS::vertex_coord = VS::vertex_coord;
    FS::vert_label = VS::vert_label;
    FS::vert_label_prob = VS::vert_label_prob;
    FS::shader_main();
    const vec4 src[2] = {FS::color, FS::semantic_map};
    float* dst[2] = {vmap4 + 4 * pix, smap4 + 4 * pix};
    for (int a = 0; a < 2; ++a) { /* C = Cs * 1 + Cd * 1 in the fp32 colour buffer */
      dst[a][0] = src[a].x * 1.0f + dst[a][0] * 1.0f;
      dst[a][1] = src[a].y * 1.0f + dst[a][1] * 1.0f;
      dst[a][2] = src[a].z * 1.0f + dst[a][2] * 1.0f;
      dst[a][3] = src[a].w * 1.0f + dst[a][3] * 1.0f;
    }
  }
}

/* full-screen passes: empty.vert + quad.geom (texCoords in [0,1]^2, interpolated: (x + 1/2) / W at a pixel
 * centre) + a fragment shader.  Preprocessing.cpp:238-327, SurfelMap.cpp:911-940 */
void ref_pass_normalmap(int W, int H, float* normal4, float* eroded4) {
  namespace FS = s_gen_normalmap_frag;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      FS::texCoords = vec2(((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H);
      FS::shader_main();
      store4(normal4, (size_t)y * W + x, FS::normal);
      store4(eroded4, (size_t)y * W + x, FS::eroded_semantic_map);
    }
}
void ref_pass_floodfill(int W, int H, float* refined4) {
  namespace FS = s_floodfill_frag;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      FS::texCoords = vec2(((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H);
      FS::shader_main();
      store4(refined4, (size_t)y * W + x, FS::refined_semantic_map);
    }
}
void ref_pass_compose(int W, int H, float* v4, float* n4, float* s4) {
  namespace FS = s_render_compose_frag;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      FS::texCoords = vec2(((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H);
      FS::shader_main();
      store4(v4, (size_t)y * W + x, FS::vertexmap);
      store4(n4, (size_t)y * W + x, FS::normalmap);
      store4(s4, (size_t)y * W + x, FS::semanticmap);
    }
}
/* K1b / K1c optional filters (Preprocessing.cpp:191-236) */
void ref_pass_avg_vertexmap(int W, int H, float* out4) {
  namespace FS = s_avg_vertexmap_frag;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      FS::texCoords = vec2(((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H);
      FS::shader_main();
      store4(out4, (size_t)y * W + x, FS::out_vertexmap);
    }
}
void ref_pass_bilateral(int W, int H, float* out4) {
  namespace FS = s_bilateral_filter_frag;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      FS::texCoords = vec2(((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H);
      FS::shader_main();
      store4(out4, (size_t)y * W + x, FS::out_vertexmap);
    }
}

}  // extern "C"

/* ------------------------------------------------------------------------------------------------------
 * K6  Frame2Model::jacobianProducts (Frame2Model.cpp:136-261): GL_POINTS over vbo_img_coords_
 * (Frame2Model.cpp:25-30: x in steps of entriesPerKernel, then y), VS -> GS, every EmitVertex is a point in a
 * 2 x 8 RGB32F target, blending GL_ONE/GL_ONE, no depth test. */
static float* g_blend;
static int64_t* g_fix;
static float* g_emit_log;
static size_t g_emit_count, g_emit_cap;
static void k6_emit() {
  namespace GS = s_Frame2Model_jacobians_geom;
  namespace FS = s_Frame2Model_jacobians_frag;
  raster_point rp;
  if (!rasterize_point(GS::gl_Position, 2, 8, &rp)) return;
  FS::values = GS::values;
  FS::shader_main();
  const int texel = rp.py * 2 + rp.px;
  for (int c = 0; c < 3; ++c) {
    float v = FS::result[c];
    g_blend[3 * texel + c] += v; /* ROP add, here in draw order (GL leaves the order open) */
    if (g_fix) g_fix[3 * texel + c] += (int64_t)llrint((double)v * SUMA_ACC_SCALE);
  }
}
extern "C" void ref_draw_jacobians(int W, int H, int entries_per_kernel, float* blend48, int64_t* fix48) {
  namespace VS = s_Frame2Model_jacobians_vert;
  namespace GS = s_Frame2Model_jacobians_geom;
  memset(blend48, 0, 48 * sizeof(float)); /* glClear */
  if (fix48) memset(fix48, 0, 48 * sizeof(int64_t));
  g_blend = blend48;
  g_fix = fix48;
  GS::emit_cb = k6_emit;
  for (int i = 0; i < W; i += entries_per_kernel)
    for (int j = 0; j < H; ++j) {
      VS::texCoords = vec2((float)i + 0.5f, (float)j + 0.5f);
      VS::shader_main();
      VS::export_vs_out(GS::gs_in[0]);
      GS::shader_main();
    }
  GS::emit_cb = nullptr;
}

/* ------------------------------------------------------------------------------------------------------
 * surfel vertex fetch: SurfelMap.cpp:46-55 (attribute 2 is an INT, attribute 3 three floats from `color`) */
#define LOAD_SURFEL(NS, A0, A1, A2, A3, A4)              \
  NS::A0 = vec4(s.x, s.y, s.z, s.radius);                \
  NS::A1 = vec4(s.nx, s.ny, s.nz, s.confidence);         \
  NS::A2 = (int)s.timestamp;                             \
  NS::A3 = vec3(s.color, s.weight, s.count);             \
  NS::A4 = vec4(s.r, s.g, s.b, s.w);                     \
  NS::gl_VertexID = id;

/* K4 geometry: render_surfels.vert + .geom per surfel (SurfelMap.cpp:847-1165).  emitted[i] = number of strip
 * vertices (0 or 4); pos = their gl_Position (4 x xyzw); tex = texCoords (4 x 2); attr = flat outputs of the last
 * vertex: vertex(4) normal(4) semantic(4) confidence(1). */
static float *g_q_pos, *g_q_tex, *g_q_attr;
static int g_q_count;
static void k4_emit() {
  namespace GS = s_render_surfels_geom;
  if (g_q_count < 4) {
    for (int c = 0; c < 4; ++c) g_q_pos[4 * g_q_count + c] = GS::gl_Position[c];
    g_q_tex[2 * g_q_count] = GS::texCoords.x;
    g_q_tex[2 * g_q_count + 1] = GS::texCoords.y;
    for (int c = 0; c < 4; ++c) {
      g_q_attr[c] = GS::vertex[c];
      g_q_attr[4 + c] = GS::normal[c];
      g_q_attr[8 + c] = GS::semantic[c];
    }
    g_q_attr[12] = GS::confidence;
  }
  ++g_q_count;
}
extern "C" void ref_render_quads(const suma_surfel* surfels, uint32_t n, uint8_t* emitted, float* pos, float* tex,
                                 float* attr) {
  namespace VS = s_render_surfels_vert;
  namespace GS = s_render_surfels_geom;
  GS::emit_cb = k4_emit;
  for (uint32_t i = 0; i < n; ++i) {
    const suma_surfel& s = surfels[i];
    const int id = (int)i;
    LOAD_SURFEL(VS, position_radius, normal_confidence, timestamp, surfel_color_weight_count, surfel_semantic_map)
    VS::shader_main();
    VS::export_vs_out(GS::gs_in[0]);
    g_q_pos = pos + 16 * (size_t)i;
    g_q_tex = tex + 8 * (size_t)i;
    g_q_attr = attr + 13 * (size_t)i;
    g_q_count = 0;
    GS::shader_main();
    emitted[i] = (uint8_t)g_q_count;
  }
  GS::emit_cb = nullptr;
}

extern "C" {

/* K7  SurfelMap::renderIndexmap (SurfelMap.cpp:586-604): gen_indexmap.vert + .frag, depth LESS, R32F target */
void ref_draw_indexmap(const suma_surfel* surfels, uint32_t n, int W, int H, float* index_map) {
  namespace VS = s_gen_indexmap_vert;
  namespace FS = s_gen_indexmap_frag;
  const size_t P = (size_t)W * H;
  memset(index_map, 0, P * 4);
  uint32_t* depth = (uint32_t*)malloc(P * 4);
  for (size_t i = 0; i < P; ++i) depth[i] = 0xffffffu;
  for (uint32_t i = 0; i < n; ++i) {
    const suma_surfel& s = surfels[i];
    const int id = (int)i;
    LOAD_SURFEL(VS, surfel_position_radius, surfel_normal_confidence, surfel_timestamp, surfel_color_weight_count,
                sfl_semantic_map)
    VS::shader_main();
    raster_point rp;
    if (!rasterize_point(VS::gl_Position, W, H, &rp)) continue;
    size_t pix = (size_t)rp.py * W + rp.px;
    if (!(rp.z24 < depth[pix])) continue;
    depth[pix] = rp.z24;
    FS::index = VS::index;
    FS::vertex = VS::vertex;
    FS::normal = VS::normal;
    FS::shader_main();
    index_map[pix] = FS::indexmap;
  }
  free(depth);
}

/* K8  SurfelMap::generateDataSurfels (SurfelMap.cpp:606-619): init_radiusConf.vert + .frag over vbo_img_coords_ */
void ref_draw_radius_conf(int W, int H, float* radconf4) {
  namespace VS = s_init_radiusConf_vert;
  namespace FS = s_init_radiusConf_frag;
  const size_t P = (size_t)W * H;
  memset(radconf4, 0, P * 16);
  for (int x = 0; x < W; ++x)
    for (int y = 0; y < H; ++y) {
      VS::img_coords = vec2((float)x + 0.5f, (float)y + 0.5f);
      VS::shader_main();
      raster_point rp;
      if (!rasterize_point(VS::gl_Position, W, H, &rp)) continue;
      FS::valid = VS::valid;
      FS::centerized_vertex = VS::centerized_vertex;
      FS::radius = VS::radius;
      FS::confidence = VS::confidence;
      FS::shader_main();
      store4(radconf4, (size_t)rp.py * W + rp.px, FS::radius_confidence_map);
    }
}

}  // extern "C"

/* K9  SurfelMap::updateSurfels part 1 (SurfelMap.cpp:621-644): update_surfels.vert + .geom + .frag, transform
 * feedback of the five sfl_* varyings (SurfelMap.cpp:38-40) + the integration mask (depth LESS, colour (1,0,0,0)) */
static suma_surfel* g_tf_out;
static uint32_t g_tf_n, g_tf_cap;
static float* g_mask4;
static uint32_t* g_mask_depth;
static int g_W, g_H;
#define TF_CAPTURE(GS)                                   \
  if (g_tf_n < g_tf_cap) {                               \
    suma_surfel& o = g_tf_out[g_tf_n];                   \
    o.x = GS::sfl_position_radius.x;                     \
    o.y = GS::sfl_position_radius.y;                     \
    o.z = GS::sfl_position_radius.z;                     \
    o.radius = GS::sfl_position_radius.w;                \
    o.nx = GS::sfl_normal_confidence.x;                  \
    o.ny = GS::sfl_normal_confidence.y;                  \
    o.nz = GS::sfl_normal_confidence.z;                  \
    o.confidence = GS::sfl_normal_confidence.w;          \
    o.timestamp = (uint32_t)GS::sfl_timestamp;           \
    o.color = GS::sfl_color_weight_count.x;              \
    o.weight = GS::sfl_color_weight_count.y;             \
    o.count = GS::sfl_color_weight_count.z;              \
    o.r = GS::sfl_semantic_map.x;                        \
    o.g = GS::sfl_semantic_map.y;                        \
    o.b = GS::sfl_semantic_map.z;                        \
    o.w = GS::sfl_semantic_map.w;                        \
  }                                                      \
  ++g_tf_n; /* counts primitives written; the buffer keeps the first cap of them */
static void k9_emit() {
  namespace GS = s_update_surfels_geom;
  namespace FS = s_update_surfels_frag;
  TF_CAPTURE(GS)
  raster_point rp;
  if (!rasterize_point(GS::gl_Position, g_W, g_H, &rp)) return;
  size_t pix = (size_t)rp.py * g_W + rp.px;
  if (!(rp.z24 < g_mask_depth[pix])) return;
  g_mask_depth[pix] = rp.z24;
  FS::shader_main();
  store4(g_mask4, pix, FS::color);
}
/* optional: src[k] = index of the input surfel that became output record k of the next ref_draw_update (transform
 * feedback closes up the dropped primitives; tests/test_gl_reference.py aligns two runs by it) */
static uint32_t* g_update_src = nullptr;
extern "C" void ref_set_update_sources(uint32_t* src) { g_update_src = src; }
extern "C" uint32_t ref_draw_update(const suma_surfel* surfels, uint32_t n, int W, int H, suma_surfel* out, uint32_t cap,
                                    float* integrated4) {
  namespace VS = s_update_surfels_vert;
  namespace GS = s_update_surfels_geom;
  const size_t P = (size_t)W * H;
  memset(integrated4, 0, P * 16);
  g_mask4 = integrated4;
  g_mask_depth = (uint32_t*)malloc(P * 4);
  for (size_t i = 0; i < P; ++i) g_mask_depth[i] = 0xffffffu;
  g_W = W;
  g_H = H;
  g_tf_out = out;
  g_tf_n = 0;
  g_tf_cap = cap;
  GS::emit_cb = k9_emit;
  for (uint32_t i = 0; i < n; ++i) {
    const suma_surfel& s = surfels[i];
    const int id = (int)i;
    LOAD_SURFEL(VS, surfel_position_radius, surfel_normal_confidence, surfel_timestamp, surfel_color_weight_count,
                surfel_semantic_map)
    VS::shader_main();
    VS::export_vs_out(GS::gs_in[0]);
    GS::gl_in[0].gl_Position = VS::gl_Position;
    const uint32_t n0 = g_tf_n;
    GS::shader_main();
    if (g_update_src && g_tf_n > n0 && n0 < cap) g_update_src[n0] = i;
  }
  GS::emit_cb = nullptr;
  g_update_src = nullptr;
  free(g_mask_depth);
  return g_tf_n < cap ? g_tf_n : cap;
}

/* K10  SurfelMap::updateSurfels part 2 (SurfelMap.cpp:646-664): gen_surfels.vert + .geom over vbo_img_coords_
 * (x-major, SurfelMap.cpp:88-92), rasteriser discard, transform feedback */
static void k10_emit() {
  namespace GS = s_gen_surfels_geom;
  TF_CAPTURE(GS)
}
extern "C" uint32_t ref_draw_generate(int W, int H, suma_surfel* out, uint32_t cap) {
  namespace VS = s_gen_surfels_vert;
  namespace GS = s_gen_surfels_geom;
  g_tf_out = out;
  g_tf_n = 0;
  g_tf_cap = cap;
  GS::emit_cb = k10_emit;
  for (int x = 0; x < W; ++x)
    for (int y = 0; y < H; ++y) {
      VS::img_coords = vec2((float)x + 0.5f, (float)y + 0.5f);
      VS::shader_main();
      VS::export_vs_out(GS::gs_in[0]);
      GS::shader_main();
    }
  GS::emit_cb = nullptr;
  return g_tf_n < cap ? g_tf_n : cap;
}

/* K11 / K12  copySurfels (SurfelMap.cpp:667-698) and extractSurfels (SurfelMap.cpp:708-742): vertex shader +
 * copy_surfels.geom, transform feedback appended behind out[0 .. n_before) */
static void copy_emit() {
  namespace GS = s_copy_surfels_geom;
  TF_CAPTURE(GS)
}
extern "C" uint32_t ref_draw_copy(const suma_surfel* surfels, uint32_t n, suma_surfel* out, uint32_t n_before,
                                  uint32_t cap) {
  namespace VS = s_copy_surfels_vert;
  namespace GS = s_copy_surfels_geom;
  g_tf_out = out;
  g_tf_n = n_before;
  g_tf_cap = cap;
  GS::emit_cb = copy_emit;
  for (uint32_t i = 0; i < n; ++i) {
    const suma_surfel& s = surfels[i];
    const int id = (int)i;
    LOAD_SURFEL(VS, position_radius, normal_confidence, in_timestamp, surfel_color_weight_count, sfl_semantic_map)
    VS::shader_main();
    VS::export_vs_out(GS::gs_in[0]);
    GS::shader_main();
  }
  GS::emit_cb = nullptr;
  return g_tf_n < cap ? g_tf_n : cap;
}
extern "C" uint32_t ref_draw_extract(const suma_surfel* surfels, uint32_t n, suma_surfel* out, uint32_t cap) {
  namespace VS = s_extract_surfels_vert;
  namespace GS = s_copy_surfels_geom;
  g_tf_out = out;
  g_tf_n = 0;
  g_tf_cap = cap;
  GS::emit_cb = copy_emit;
  for (uint32_t i = 0; i < n; ++i) {
    const suma_surfel& s = surfels[i];
    const int id = (int)i;
    LOAD_SURFEL(VS, position_radius, normal_confidence, in_timestamp, surfel_color_weight_count, sfl_semantic_map)
    VS::shader_main();
    VS::export_vs_out(GS::gs_in[0]);
    GS::shader_main();
  }
  GS::emit_cb = nullptr;
  return g_tf_n < cap ? g_tf_n : cap;
}

extern "C" {

/* src/core/lie_algebra.cpp, compiled where it lies against oracle/eigen_shim */
void ref_se3_exp(const double x[6], double T[16]) {
  Eigen::VectorXd v(6);
  for (int i = 0; i < 6; ++i) v[i] = x[i];
  Eigen::Matrix4d M = SE3::exp(v);
  memcpy(T, M.data(), 16 * sizeof(double)); /* column major */
}
void ref_se3_log(const double T[16], double x[6]) {
  Eigen::Matrix4d M;
  memcpy(M.data(), T, 16 * sizeof(double));
  Eigen::VectorXd v = SE3::log(M);
  for (int i = 0; i < 6; ++i) x[i] = v[i];
}

/* GLSL built-ins exposed for the deviation tests */
void ref_glsl_inverse(const float m[16], float out[16]) {
  mat4 M;
  for (int c = 0; c < 4; ++c) M[c] = vec4(m[4 * c], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]);
  mat4 R = inverse_cofactor(M);
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) out[4 * c + r] = R[c][r];
}
float ref_glsl_pack(float r, float g, float b) { return s_update_surfels_vert::pack(vec3(r, g, b)); }
void ref_glsl_slerp(const float v0[3], const float v1[3], float w, float out[3]) {
  vec3 r = s_update_surfels_vert::slerp(vec3(v0[0], v0[1], v0[2]), vec3(v1[0], v1[1], v1[2]), w);
  out[0] = r.x;
  out[1] = r.y;
  out[2] = r.z;
}
int ref_uses_libm(void) {
#ifdef REF_USE_LIBM
  return 1;
#else
  return 0;
#endif
}

}  // extern "C"
