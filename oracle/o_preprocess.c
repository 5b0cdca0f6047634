/*
 * oracle/o_preprocess.c -- TEST INFRASTRUCTURE (CPU oracle).
 * Restates Preprocessing::process (reference src/core/Preprocessing.cpp:120-339) and its shaders:
 *   K1 gen_vertexmap.vert:73-103 / gen_vertexmap.frag:16-23  (z-buffered spherical scatter)
 *   K2 gen_normalmap.frag:41-99                                (cross-stencil normals + label erosion)
 *   K3 floodfill.frag:34-84                                    (label flood fill)
 */
#include "o_ctx.h"

static inline int32_t o_wrap(int32_t x, int32_t w) {
  /* gen_normalmap.frag:24-32 wrap() */
  while (x >= w) x -= w;
  while (x < 0) x += w;
  return x;
}

/* K1.  GL semantics restated: GL_POINTS with the vertex snapped to its texel centre
 * (gen_vertexmap.vert:88-89), point clipping against the unit cube, depth test GL_LESS on a
 * 24-bit depth buffer (Preprocessing.cpp:56,158,168), primitives processed in order => the
 * smaller quantised depth wins, ties go to the lower point index. */
static void o_k1_vertexmap(const ora_ctx* c, const suma_float4* pts, const float* labels, const float* probs,
                           uint32_t n, uint32_t timestamp, ora_frame* f, uint64_t* zbuf) {
  const suma_params* p = &c->p;
  const int32_t W = (int32_t)p->data_width, H = (int32_t)p->data_height;
  const float width = (float)W, height = (float)H;
  const float fov_up = fabsf(p->data_fov_up), fov_down = fabsf(p->data_fov_down);
  const float fov = fabsf(fov_up) + fabsf(fov_down);
  const float min_depth = p->min_depth, max_depth = p->max_depth;
  const size_t P = (size_t)W * (size_t)H;
  for (size_t i = 0; i < P; ++i) zbuf[i] = ~(uint64_t)0;

#pragma omp parallel for num_threads(c->threads) schedule(static)
  for (uint32_t i = 0; i < n; ++i) {
    ov3 pos = ov3_make(pts[i].x, pts[i].y, pts[i].z);
    float depth = ov3_len(pos);
    float yaw = sdm_atan2(pos.y, pos.x);
    float pitch = -sdm_asin(pos.z / depth);
    float x = (-yaw * SUMA_INV_PI_F);
    float y = (1.0f - (2.0f * ((pitch * SUMA_RAD2DEG_F) + fov_up)) / fov);
    float z = 2.0f * ((depth - min_depth) / (max_depth - min_depth)) - 1.0f;
    float fx = sdm_floor((0.5f * (x + 1.0f)) * width);
    float fy = sdm_floor((0.5f * (y + 1.0f)) * height);
    if (!(fx >= 0.0f && fx < width && fy >= 0.0f && fy < height)) continue; /* clipped (or NaN) */
    if (!(z >= -1.0f && z <= 1.0f)) continue;
    float zw = 0.5f * z + 0.5f;
    uint64_t key = ((uint64_t)o_depth24(zw) << 32) | (uint64_t)i;
    size_t pix = (size_t)(int32_t)fy * (size_t)W + (size_t)(int32_t)fx;
    o_zmin(&zbuf[pix], key);
  }

  const int isfirst = (timestamp < 10); /* Preprocessing.cpp:176-179 */
#pragma omp parallel for num_threads(c->threads) schedule(static)
  for (size_t pix = 0; pix < P; ++pix) {
    uint64_t key = zbuf[pix];
    if (key == ~(uint64_t)0) {
      f->vertex[pix] = o_f4(0.f, 0.f, 0.f, 0.f); /* glClearColor(0,0,0,0) */
      f->semantic[pix] = o_f4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    uint32_t i = (uint32_t)(key & 0xffffffffu);
    /* quirk B-1: attribute offsets of Preprocessing.cpp:142-145 */
    uint64_t li = (uint64_t)i + p->label_offset, pi = (uint64_t)i + p->prob_offset;
    float label = (labels != NULL && li < n) ? labels[li] : 0.0f;
    float prob = (probs != NULL && pi < n) ? probs[pi] : 0.0f;
    suma_float4 v = o_f4(pts[i].x, pts[i].y, pts[i].z, 1.0f);
    if (isfirst && o_is_dynamic_label(label)) v = o_f4(0.f, 0.f, 0.f, 0.f); /* gen_vertexmap.vert:95-102 */
    float l = label / 255.0f;                                               /* gen_vertexmap.frag:20 */
    f->vertex[pix] = v;
    f->semantic[pix] = o_f4(l, l, l, prob);
  }
}

/* K2. gen_normalmap.frag:41-99.  NEAREST + CLAMP_TO_BORDER sampler (Preprocessing.cpp:68-70),
 * x wraps through wrap(), y falls off into the zero border. */
static void o_k2_normals(const ora_ctx* c, const ora_frame* f, suma_float4* normal, suma_float4* eroded) {
  const int32_t W = (int32_t)c->p.data_width, H = (int32_t)c->p.data_height;
  const suma_float4* V = f->vertex;
  const suma_float4* S = f->semantic;
#pragma omp parallel for num_threads(c->threads) schedule(static)
  for (int32_t y = 0; y < H; ++y) {
    for (int32_t x = 0; x < W; ++x) {
      size_t pix = (size_t)y * W + x;
      suma_float4 nrm = o_f4(0.f, 0.f, 0.f, 1.f);
      suma_float4 ero = o_f4(0.f, 0.f, 0.f, 1.f);
      suma_float4 p = V[pix];
      if (p.w > 0.0f) {
        nrm.w = 1.0f;
        suma_float4 u = o_texel(V, W, H, o_wrap(x + 1, W), y);
        suma_float4 v = o_texel(V, W, H, x, y + 1);
        suma_float4 s = o_texel(V, W, H, o_wrap(x - 1, W), y);
        suma_float4 t = o_texel(V, W, H, x, y - 1);
        ov3 pp = ov3_make(p.x, p.y, p.z);
        ov3 un = ov3_normalize(ov3_sub(ov3_make(u.x, u.y, u.z), pp));
        ov3 vn = ov3_normalize(ov3_sub(ov3_make(v.x, v.y, v.z), pp));
        /* s, t directions are computed by the shader but only their .w is used */
        if (u.w < 1.0f && v.w < 1.0f) nrm.w = 0.0f;
        if (s.w < 1.0f && t.w < 1.0f) nrm.w = 0.0f;
        if (!(u.w > 0.5f) || !(v.w > 0.5f)) nrm.w = 0.0f;

        /* erosion, kernel_size = 2 -> offset 1 only (gen_normalmap.frag:69-85) */
        ero = S[pix];
        float pl = S[pix].x;
        float ul = o_texel(S, W, H, o_wrap(x + 1, W), y).x;
        float vl = o_texel(S, W, H, x, y + 1).x;
        float sl = o_texel(S, W, H, o_wrap(x - 1, W), y).x;
        float tl = o_texel(S, W, H, x, y - 1).x;
        if ((pl != ul && ul != 0.0f) || (pl != vl && vl != 0.0f) || (pl != sl && sl != 0.0f) ||
            (pl != tl && tl != 0.0f))
          ero = o_f4(0.f, 0.f, 0.f, 1.f);

        if (nrm.w > 0.0f) {
          ov3 w = ov3_cross(un, vn);
          float len = ov3_len(w);
          nrm = o_f4(w.x / len, w.y / len, w.z / len, (len > 0.0000001f) ? 1.0f : 0.0f);
        }
      }
      normal[pix] = nrm;
      eroded[pix] = ero;
    }
  }
}

/* K3. floodfill.frag:34-84, kernel_size = 3 -> offsets 1, 2; neighbour order +x, +y, -x, -y. */
static void o_k3_floodfill(const ora_ctx* c, const suma_float4* V, const suma_float4* E, suma_float4* refined) {
  const int32_t W = (int32_t)c->p.data_width, H = (int32_t)c->p.data_height;
  const float threshold = 0.007f;
#pragma omp parallel for num_threads(c->threads) schedule(static)
  for (int32_t y = 0; y < H; ++y) {
    for (int32_t x = 0; x < W; ++x) {
      size_t pix = (size_t)y * W + x;
      suma_float4 out = E[pix];
      suma_float4 p = V[pix];
      float lp = ov3_len(ov3_make(p.x, p.y, p.z));
      float plabel = E[pix].x;
      for (int32_t offset = 1; offset < 3; ++offset) {
        const int32_t nx[4] = {o_wrap(x + offset, W), x, o_wrap(x - offset, W), x};
        const int32_t ny[4] = {y, y + offset, y, y - offset};
        int hit = 0;
        for (int k = 0; k < 4 && !hit; ++k) {
          suma_float4 q = o_texel(V, W, H, nx[k], ny[k]);
          suma_float4 ql = o_texel(E, W, H, nx[k], ny[k]);
          float lq = ov3_len(ov3_make(q.x, q.y, q.z));
          if (plabel == 0.0f && ql.x != 0.0f && fabsf(lp - lq) < threshold * lp) {
            out = o_f4(ql.x, ql.y, ql.z, ql.w / (float)(offset + 1));
            hit = 1;
          }
        }
        if (hit) break;
      }
      refined[pix] = out;
    }
  }
}

void ora_preprocess(ora_ctx* c, const suma_float4* points, const float* labels, const float* probs, uint32_t n,
                    uint32_t timestamp, ora_frame* out) {
  const size_t P = (size_t)c->p.data_width * c->p.data_height;
  o_k1_vertexmap(c, points, labels, probs, n, timestamp, out, c->zbuf_data);
  suma_float4* eroded = (suma_float4*)malloc(P * sizeof(suma_float4));
  o_k2_normals(c, out, out->normal, eroded);
  suma_float4* refined = (suma_float4*)malloc(P * sizeof(suma_float4));
  o_k3_floodfill(c, out->vertex, eroded, refined);
  memcpy(out->semantic, refined, P * sizeof(suma_float4)); /* Preprocessing.cpp:327 */
  free(eroded);
  free(refined);
}
