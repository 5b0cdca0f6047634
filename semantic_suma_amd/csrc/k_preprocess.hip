/*
 * k_preprocess.hip -- K1..K3: Preprocessing::process on gfx950.
 *
 * Replaces (reference, citations relative to /root/reference):
 *   K1 src/core/Preprocessing.cpp:120-189 + src/shader/gen_vertexmap.vert:73-103 / .frag:16-23
 *      GL: GL_POINTS scatter with depth test GL_LESS on a 24-bit depth buffer.
 *      Here: 64-bit atomicMin of (depth24 << 32 | point index) -- the same winner GL's in-order
 *      depth test picks (smaller quantised depth, then lower index) -- then a resolve pass that
 *      also re-arms the z-buffer (no clear launch).
 *   K2 Preprocessing.cpp:238-279 + gen_normalmap.frag:41-99 (cross-stencil normal, label erosion)
 *   K3 Preprocessing.cpp:281-327 + floodfill.frag:34-84     (label flood fill)
 *
 * Layout: every map is a row-major W x H array of float4 (row 0 = lowest beam); one thread per
 * texel, 16 B per lane, a wave covers 1 KiB of one row.
 */
#include "suma_internal.h"

__global__ void __launch_bounds__(256)
    k1_scatter(const float4* __restrict__ pts, uint32_t n, proj_t q, unsigned long long* __restrict__ zbuf) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 pt = pts[i];
  v3 pos = mk3(pt.x, pt.y, pt.z);
  float depth = len3(pos);
  float yaw = sdm_atan2(pos.y, pos.x);
  float pitch = -sdm_asin(pos.z / depth);
  /* gen_vertexmap.vert:83-85: NDC coordinates */
  float x = (-yaw * SUMA_INV_PI_F);
  float y = (1.0f - (2.0f * ((pitch * SUMA_RAD2DEG_F) + q.fov_up)) / q.fov);
  float z = 2.0f * ((depth - q.min_depth) / (q.max_depth - q.min_depth)) - 1.0f;
  /* :88-89 snap to the texel centre */
  float fx = sdm_floor((0.5f * (x + 1.0f)) * q.width);
  float fy = sdm_floor((0.5f * (y + 1.0f)) * q.height);
  if (!(fx >= 0.0f && fx < q.width && fy >= 0.0f && fy < q.height)) return; /* clipped (or NaN) */
  if (!(z >= -1.0f && z <= 1.0f)) return;
  float zw = 0.5f * z + 0.5f;
  unsigned long long key = ((unsigned long long)depth24(zw) << 32) | (unsigned long long)i;
  size_t pix = (size_t)(int32_t)fy * (size_t)q.W + (size_t)(int32_t)fx;
  atomicMin(&zbuf[pix], key);
}

__global__ void __launch_bounds__(256)
    k1_resolve(unsigned long long* __restrict__ zbuf, const float4* __restrict__ pts,
               const float* __restrict__ labels, const float* __restrict__ probs, uint32_t n, uint32_t label_offset,
               uint32_t prob_offset, int isfirst, float4* __restrict__ vertex, float4* __restrict__ semantic,
               uint32_t P) {
  uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= P) return;
  unsigned long long key = zbuf[pix];
  zbuf[pix] = SUMA_EMPTY_KEY; /* the data-sized z-buffer is shared with K7 and is always left cleared */
  if (key == SUMA_EMPTY_KEY) {
    vertex[pix] = f4(0.f, 0.f, 0.f, 0.f); /* glClearColor(0,0,0,0) */
    semantic[pix] = f4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  uint32_t i = (uint32_t)(key & 0xffffffffull);
  /* quirk B-1 (Preprocessing.cpp:142-145): the label / prob attributes are bound at offsets 4 / 5 */
  unsigned long long li = (unsigned long long)i + label_offset, pi = (unsigned long long)i + prob_offset;
  float label = (labels != nullptr && li < n) ? labels[li] : 0.0f;
  float prob = (probs != nullptr && pi < n) ? probs[pi] : 0.0f;
  float4 pt = pts[i];
  float4 v = f4(pt.x, pt.y, pt.z, 1.0f);
  if (isfirst && is_dynamic_label(label)) v = f4(0.f, 0.f, 0.f, 0.f); /* gen_vertexmap.vert:95-102 */
  float l = label / 255.0f;                                            /* gen_vertexmap.frag:20 */
  vertex[pix] = v;
  semantic[pix] = f4(l, l, l, prob);
}

__device__ __forceinline__ int32_t wrapx(int32_t x, int32_t w) {
  /* gen_normalmap.frag:24-32 wrap() for |offset| < w */
  if (x >= w) x -= w;
  if (x < 0) x += w;
  return x;
}

__global__ void __launch_bounds__(256)
    k2_normals(const float4* __restrict__ V, const float4* __restrict__ S, float4* __restrict__ normal,
               float4* __restrict__ eroded, int32_t W, int32_t H) {
  int32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  int32_t y = blockIdx.y;
  if (x >= W) return;
  size_t pix = (size_t)y * W + x;
  float4 nrm = f4(0.f, 0.f, 0.f, 1.f);
  float4 ero = f4(0.f, 0.f, 0.f, 1.f);
  float4 p = V[pix];
  if (p.w > 0.0f) {
    nrm.w = 1.0f;
    int32_t xp = wrapx(x + 1, W), xm = wrapx(x - 1, W);
    float4 u = texel(V, W, H, xp, y);
    float4 v = texel(V, W, H, x, y + 1);
    float4 s = texel(V, W, H, xm, y);
    float4 t = texel(V, W, H, x, y - 1);
    v3 pp = xyz(p);
    v3 un = normalize3(sub3(xyz(u), pp));
    v3 vn = normalize3(sub3(xyz(v), pp));
    if (u.w < 1.0f && v.w < 1.0f) nrm.w = 0.0f;
    if (s.w < 1.0f && t.w < 1.0f) nrm.w = 0.0f;
    if (!(u.w > 0.5f) || !(v.w > 0.5f)) nrm.w = 0.0f;

    /* erosion, kernel_size 2 -> offset 1 (gen_normalmap.frag:69-85) */
    float4 sp = S[pix];
    ero = sp;
    float pl = sp.x;
    float ul = texel(S, W, H, xp, y).x;
    float vl = texel(S, W, H, x, y + 1).x;
    float sl = texel(S, W, H, xm, y).x;
    float tl = texel(S, W, H, x, y - 1).x;
    if ((pl != ul && ul != 0.0f) || (pl != vl && vl != 0.0f) || (pl != sl && sl != 0.0f) || (pl != tl && tl != 0.0f))
      ero = f4(0.f, 0.f, 0.f, 1.f);

    if (nrm.w > 0.0f) {
      v3 w = cross3(un, vn);
      float len = len3(w);
      nrm = f4(w.x / len, w.y / len, w.z / len, (len > 0.0000001f) ? 1.0f : 0.0f);
    }
  }
  normal[pix] = nrm;
  eroded[pix] = ero;
}

__global__ void __launch_bounds__(256)
    k3_floodfill(const float4* __restrict__ V, const float4* __restrict__ E, float4* __restrict__ refined, int32_t W,
                 int32_t H) {
  int32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  int32_t y = blockIdx.y;
  if (x >= W) return;
  size_t pix = (size_t)y * W + x;
  const float threshold = 0.007f;
  float4 out = E[pix];
  float plabel = out.x;
  if (plabel == 0.0f) { /* only unlabeled texels can change (floodfill.frag:52) */
    float lp = len3(xyz(V[pix]));
    bool hit = false;
    for (int32_t offset = 1; offset < 3 && !hit; ++offset) {
      const int32_t nx[4] = {wrapx(x + offset, W), x, wrapx(x - offset, W), x};
      const int32_t ny[4] = {y, y + offset, y, y - offset};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (hit) break;
        float4 ql = texel(E, W, H, nx[k], ny[k]);
        if (ql.x != 0.0f) {
          float lq = len3(xyz(texel(V, W, H, nx[k], ny[k])));
          if (sdm_abs(lp - lq) < threshold * lp) {
            out = f4(ql.x, ql.y, ql.z, ql.w / (float)(offset + 1));
            hit = true;
          }
        }
      }
    }
  }
  refined[pix] = out;
}

hipError_t launch_preprocess(suma_ctx* c, const float4* d_pts, const float* d_labels, const float* d_probs, uint32_t n,
                             uint32_t timestamp, suma_frame* out) {
  const uint32_t P = (uint32_t)c->P;
  const int32_t W = c->pd.W, H = c->pd.H;
  hipStream_t st = c->stream;
  {
    ProfScope ps(c, "k1_vertexmap", 24.0 * n + 32.0 * P);
    if (n > 0) k1_scatter<<<(n + 255) / 256, 256, 0, st>>>(d_pts, n, c->pd, c->zbuf_data);
    k1_resolve<<<(P + 255) / 256, 256, 0, st>>>(c->zbuf_data, d_pts, d_labels, d_probs, n, c->p.label_offset,
                                                 c->p.prob_offset, timestamp < 10 ? 1 : 0, out->map[SUMA_MAP_VERTEX],
                                                 out->map[SUMA_MAP_SEMANTIC], P);
  }
  {
    ProfScope ps(c, "k2k3_normals_labels", 64.0 * P);
    dim3 grid((W + 255) / 256, H);
    k2_normals<<<grid, 256, 0, st>>>(out->map[SUMA_MAP_VERTEX], out->map[SUMA_MAP_SEMANTIC], out->map[SUMA_MAP_NORMAL],
                                     c->eroded, W, H);
    k3_floodfill<<<grid, 256, 0, st>>>(out->map[SUMA_MAP_VERTEX], c->eroded, out->map[SUMA_MAP_SEMANTIC], W, H);
  }
  return hipGetLastError();
}
