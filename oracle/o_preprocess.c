/*
 * oracle/o_preprocess.c -- TEST INFRASTRUCTURE (CPU oracle).
 * Restates Preprocessing::process (reference src/core/Preprocessing.cpp:120-339) and its shaders:
 *   K1 gen_vertexmap.vert:73-103 / gen_vertexmap.frag:16-23  (z-buffered spherical scatter)
 *   K2 gen_normalmap.frag:41-99                                (cross-stencil normals + label erosion)
 *   K3 floodfill.frag:34-84                                    (label flood fill)
 * and the optional filters between K1 and K2 (Preprocessing.cpp:150-236; off in config/default.xml):
 *   K1 in its `avg_vertexmap` mode (no depth test, additive blending), K1b avg_vertexmap.frag:14-21,
 *   K1c bilateral_filter.frag:28-83
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
          const ov3 wn = ov3_divs(w, len); /* vec3 / float */
          nrm = o_f4(wn.x, wn.y, wn.z, (len > 0.0000001f) ? 1.0f : 0.0f);
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

/* K1 with avgVertexmap_ (Preprocessing.cpp:150,160-166): depth test off, glBlendFunc(GL_ONE, GL_ONE) on both colour
 * attachments.  GL blends the fragments of one pixel in primitive order, so each texel ends up with the fp32 sum
 * ((f_0 + f_1) + f_2) ... of its points taken by ascending index; clipping is that of the z-buffered pass. */
static void o_k1_vertexmap_sum(const ora_ctx* c, const suma_float4* pts, const float* labels, const float* probs,
                               uint32_t n, uint32_t timestamp, suma_float4* vsum, suma_float4* ssum) {
  const suma_params* p = &c->p;
  const int32_t W = (int32_t)p->data_width, H = (int32_t)p->data_height;
  const float width = (float)W, height = (float)H;
  const float fov_up = fabsf(p->data_fov_up), fov_down = fabsf(p->data_fov_down);
  const float fov = fabsf(fov_up) + fabsf(fov_down);
  const float min_depth = p->min_depth, max_depth = p->max_depth;
  const size_t P = (size_t)W * (size_t)H;
  const int isfirst = (timestamp < 10);
  for (size_t i = 0; i < P; ++i) vsum[i] = ssum[i] = o_f4(0.f, 0.f, 0.f, 0.f);
  for (uint32_t i = 0; i < n; ++i) { /* in order: the sums are not associative */
    ov3 pos = ov3_make(pts[i].x, pts[i].y, pts[i].z);
    float depth = ov3_len(pos);
    float yaw = sdm_atan2(pos.y, pos.x);
    float pitch = -sdm_asin(pos.z / depth);
    float x = (-yaw * SUMA_INV_PI_F);
    float y = (1.0f - (2.0f * ((pitch * SUMA_RAD2DEG_F) + fov_up)) / fov);
    float z = 2.0f * ((depth - min_depth) / (max_depth - min_depth)) - 1.0f;
    float fx = sdm_floor((0.5f * (x + 1.0f)) * width);
    float fy = sdm_floor((0.5f * (y + 1.0f)) * height);
    if (!(fx >= 0.0f && fx < width && fy >= 0.0f && fy < height)) continue;
    if (!(z >= -1.0f && z <= 1.0f)) continue;
    size_t pix = (size_t)(int32_t)fy * (size_t)W + (size_t)(int32_t)fx;
    uint64_t li = (uint64_t)i + p->label_offset, pi = (uint64_t)i + p->prob_offset;
    float label = (labels != NULL && li < n) ? labels[li] : 0.0f;
    float prob = (probs != NULL && pi < n) ? probs[pi] : 0.0f;
    suma_float4 v = o_f4(pts[i].x, pts[i].y, pts[i].z, 1.0f);
    if (isfirst && o_is_dynamic_label(label)) v = o_f4(0.f, 0.f, 0.f, 0.f);
    float l = label / 255.0f;
    suma_float4* a = &vsum[pix];
    a->x = v.x + a->x, a->y = v.y + a->y, a->z = v.z + a->z, a->w = v.w + a->w;
    suma_float4* b = &ssum[pix];
    b->x = l + b->x, b->y = l + b->y, b->z = l + b->z, b->w = prob + b->w;
  }
}

/* What texture(sampler2DRect, (x, y)) returns at the INTEGER coordinate (x, y) -- all the two filter shaders ever
 * ask for (avg_vertexmap.frag:17-19, bilateral_filter.frag:30-33,62).  No sampler object is bound in these passes,
 * the texture's own state decides (suma_types.h, filter_sampling):
 *   NEAREST            texel (x, y);
 *   GL initial state   LINEAR + CLAMP_TO_EDGE, GL 3.3 core 3.8.11: u - 1/2 = x - 1/2, so i0 = x - 1, alpha = 1/2
 *                      (likewise beta): the four texels around the corner, each weighted (1/2)(1/2), summed in the
 *                      order of the spec's formula; indices clamped to the edge. */
static suma_float4 o_filter_fetch(const ora_ctx* c, const suma_float4* M, int32_t x, int32_t y) {
  const int32_t W = (int32_t)c->p.data_width;
  if (c->p.filter_sampling == SUMA_FILTER_SAMPLING_NEAREST) return M[(size_t)y * W + x];
  const int32_t i0 = x - 1 < 0 ? 0 : x - 1, j0 = y - 1 < 0 ? 0 : y - 1;
  const float a = 0.5f, b = 0.5f;
  const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  const suma_float4 t00 = M[(size_t)j0 * W + i0], t10 = M[(size_t)j0 * W + x], t01 = M[(size_t)y * W + i0],
                    t11 = M[(size_t)y * W + x];
  return o_f4(((t00.x * w00 + t10.x * w10) + t01.x * w01) + t11.x * w11,
              ((t00.y * w00 + t10.y * w10) + t01.y * w01) + t11.y * w11,
              ((t00.z * w00 + t10.z * w10) + t01.z * w01) + t11.z * w11,
              ((t00.w * w00 + t10.w * w10) + t01.w * w01) + t11.w * w11);
}

/* the texel coordinate the filter shaders derive from the interpolated texCoords (quad.geom: (x + 1/2) / W at the
 * pixel centre): int(texCoords.x * width) */
static inline int32_t o_filter_coord(int32_t x, int32_t w) { return (int32_t)((((float)x + 0.5f) / (float)w) * (float)w); }

/* K1b avg_vertexmap.frag:14-21 */
static void o_k1b_average(const ora_ctx* c, const suma_float4* vsum, suma_float4* out) {
  const int32_t W = (int32_t)c->p.data_width, H = (int32_t)c->p.data_height;
#pragma omp parallel for num_threads(c->threads) schedule(static)
  for (int32_t y = 0; y < H; ++y)
    for (int32_t x = 0; x < W; ++x) {
      suma_float4 v = o_filter_fetch(c, vsum, o_filter_coord(x, W), o_filter_coord(y, H));
      if (v.w > 0.5f) { /* vec4 / float: one reciprocal, four multiplies (o_math.h) */
        const float rw = 1.0f / v.w;
        v = o_f4(v.x * rw, v.y * rw, v.z * rw, v.w * rw);
      }
      out[(size_t)y * W + x] = v;
    }
}

/* K1c bilateral_filter.frag:28-83: 13 x 13 window (columns wrap, rows stop at the image), weights from the pixel
 * distance and the range difference; the neighbour's "range" is length(vec4), w included (:64) */
static void o_k1c_bilateral(const ora_ctx* c, const suma_float4* V, suma_float4* out) {
  const int32_t W = (int32_t)c->p.data_width, H = (int32_t)c->p.data_height;
  const float width = (float)W, height = (float)H;
  const float sigma_space = c->p.bilateral_sigma_space, sigma_range = c->p.bilateral_sigma_range;
#pragma omp parallel for num_threads(c->threads) schedule(dynamic, 1)
  for (int32_t py = 0; py < H; ++py)
    for (int32_t px = 0; px < W; ++px) {
      const int32_t x = (int32_t)((((float)px + 0.5f) / width) * width), y = (int32_t)((((float)py + 0.5f) / height) * height);
      suma_float4 vertex = o_filter_fetch(c, V, x, y);
      suma_float4 res = vertex;
      if (vertex.w > 0.5f) {
        float range = ov3_len(ov3_make(vertex.x, vertex.y, vertex.z));
        ov3 ray = ov3_divs(ov3_make(vertex.x, vertex.y, vertex.z), range);
        float sigma_space_factor = -0.5f / (sigma_space * sigma_space);
        float sigma_range_factor = -0.5f / (sigma_range * sigma_range);
        const int32_t R = 6, D = R * 2 + 1;
        int32_t tx = x - D / 2 + D;
        int32_t ty = y - D / 2 + D < (int32_t)height ? y - D / 2 + D : (int32_t)height;
        float sum1 = 0.0f, sum2 = 0.0f;
        for (int32_t cy = (y - D / 2 > 0 ? y - D / 2 : 0); cy < ty; ++cy)
          for (int32_t cx = x - D / 2; cx < tx; ++cx) {
            float xx = (float)cx; /* bilateral_filter.frag:18-26 wrap() on floats */
            while (xx >= width) xx = xx - width;
            while (xx < 0.0f) xx = xx + width;
            suma_float4 tmp = o_filter_fetch(c, V, (int32_t)xx, cy);
            if (tmp.w < 0.5f) continue;
            /* length(vec4) = sqrt(dot): the fused chain of o_math.h over four components */
            float tmp_range = sdm_sqrt(O_FMA(tmp.w, tmp.w, O_FMA(tmp.z, tmp.z, O_FMA(tmp.y, tmp.y, tmp.x * tmp.x))));
            float dx = (float)x - xx;
            float diff_space2 = dx * dx + (float)((y - cy) * (y - cy));
            float diff_range2 = (range - tmp_range) * (range - tmp_range);
            float weight = sdm_exp(diff_space2 * sigma_space_factor + diff_range2 * sigma_range_factor);
            sum1 += tmp_range * weight;
            sum2 += weight;
          }
        float filtered_range = sum1 / sum2;
        res = o_f4(filtered_range * ray.x, filtered_range * ray.y, filtered_range * ray.z, 1.0f);
      }
      out[(size_t)py * W + px] = res;
    }
}

void ora_preprocess(ora_ctx* c, const suma_float4* points, const float* labels, const float* probs, uint32_t n,
                    uint32_t timestamp, ora_frame* out) {
  const size_t P = (size_t)c->p.data_width * c->p.data_height;
  if (c->p.avg_vertexmap) { /* Preprocessing.cpp:150,160-166,191-213 */
    suma_float4* temp = (suma_float4*)malloc(P * sizeof(suma_float4));
    o_k1_vertexmap_sum(c, points, labels, probs, n, timestamp, temp, out->semantic);
    o_k1b_average(c, temp, out->vertex);
    free(temp);
  } else {
    o_k1_vertexmap(c, points, labels, probs, n, timestamp, out, c->zbuf_data);
  }
  if (c->p.filter_vertexmap) { /* :215-236: computed always, used only with use_filtered_vertexmap */
    suma_float4* temp = (suma_float4*)malloc(P * sizeof(suma_float4));
    o_k1c_bilateral(c, out->vertex, temp);
    if (c->p.use_filtered_vertexmap) memcpy(out->vertex, temp, P * sizeof(suma_float4));
    free(temp);
  }
  suma_float4* eroded = (suma_float4*)malloc(P * sizeof(suma_float4));
  o_k2_normals(c, out, out->normal, eroded);
  suma_float4* refined = (suma_float4*)malloc(P * sizeof(suma_float4));
  o_k3_floodfill(c, out->vertex, eroded, refined);
  memcpy(out->semantic, refined, P * sizeof(suma_float4)); /* Preprocessing.cpp:327 */
  free(eroded);
  free(refined);
}
