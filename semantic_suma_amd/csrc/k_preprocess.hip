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
 * texel, 16 B per lane.  K2 + K3 run as ONE kernel over LDS-staged tiles (k23_normals_labels).
 * The optional filters between K1 and K2 (avg_vertexmap, filter_vertexmap: off in config/default.xml) live in
 * k_filters.hip.
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

/* K2 + K3 fused over LDS tiles.  A block owns a PRE_TX x PRE_TY patch of the image and stages the
 * vertex and semantic maps of the patch plus a halo of 3 texels (x wraps around the 360-degree
 * image, rows outside the image are the zero border) in LDS: the 5-point normal / erosion stencil
 * needs +-1, the flood fill reads ERODED labels at +-1 / +-2, and erosion itself looks one further.
 * Stage B writes the normal map and keeps the eroded labels of patch + halo 2 in LDS, stage C does
 * the flood fill from LDS -- the intermediate "eroded" image never exists in HBM and the per-texel
 * neighbour fetches (9 vertex + 13 label reads) are LDS reads. */
#define PRE_TX 64
#define PRE_TY 8
#define PRE_H3 3
#define PRE_SW (PRE_TX + 2 * PRE_H3) /* 70 */
#define PRE_SH (PRE_TY + 2 * PRE_H3) /* 14 */
#define PRE_EW (PRE_TX + 4)          /* eroded: halo 2 */
#define PRE_EH (PRE_TY + 4)

__global__ void __launch_bounds__(PRE_TX* PRE_TY)
    k23_normals_labels(const float4* __restrict__ V, const float4* __restrict__ S, float4* __restrict__ normal,
                       float4* __restrict__ refined, int32_t W, int32_t H) {
  __shared__ float4 sV[PRE_SH][PRE_SW];
  __shared__ float4 sS[PRE_SH][PRE_SW];
  __shared__ float4 sE[PRE_EH][PRE_EW];
  const int32_t x0 = blockIdx.x * PRE_TX, y0 = blockIdx.y * PRE_TY;
  const int tid = threadIdx.x;
  /* ---- stage A: patch + halo 3 ---- */
  for (int t = tid; t < PRE_SW * PRE_SH; t += PRE_TX * PRE_TY) {
    const int ly = t / PRE_SW, lx = t - ly * PRE_SW;
    int32_t gx = x0 + lx - PRE_H3, gy = y0 + ly - PRE_H3;
    if (gx >= W) gx -= W; /* gen_normalmap.frag:24-32 wrap() */
    if (gx < 0) gx += W;
    /* a patch that overhangs the right image edge (W not a multiple of the tile) may wrap twice */
    if (gx >= W) gx -= W;
    float4 v = f4(0.f, 0.f, 0.f, 0.f), s = v;
    if (gy >= 0 && gy < H) {
      v = V[(size_t)gy * W + gx];
      s = S[(size_t)gy * W + gx];
    }
    sV[ly][lx] = v;
    sS[ly][lx] = s;
  }
  __syncthreads();
  /* ---- stage B: eroded labels for patch + halo 2 (gen_normalmap.frag:69-85) ---- */
  for (int t = tid; t < PRE_EW * PRE_EH; t += PRE_TX * PRE_TY) {
    const int ey = t / PRE_EW, ex = t - ey * PRE_EW;
    const int ly = ey + 1, lx = ex + 1; /* position in the halo-3 arrays */
    const int32_t gy = y0 + ey - 2;
    float4 ero = f4(0.f, 0.f, 0.f, 0.f); /* outside the image: the sampler's border colour */
    if (gy >= 0 && gy < H) {
      ero = f4(0.f, 0.f, 0.f, 1.f);
      if (sV[ly][lx].w > 0.0f) {
        const float4 sp = sS[ly][lx];
        ero = sp;
        const float pl = sp.x;
        const float ul = sS[ly][lx + 1].x, vl = sS[ly + 1][lx].x, sl = sS[ly][lx - 1].x, tl = sS[ly - 1][lx].x;
        if ((pl != ul && ul != 0.0f) || (pl != vl && vl != 0.0f) || (pl != sl && sl != 0.0f) || (pl != tl && tl != 0.0f))
          ero = f4(0.f, 0.f, 0.f, 1.f);
      }
    }
    sE[ey][ex] = ero;
  }
  /* normals of the patch itself (gen_normalmap.frag:41-66, 87-99) */
  const int px = tid % PRE_TX, py = tid / PRE_TX;
  const int32_t gx = x0 + px, gy = y0 + py;
  const bool inside = gx < W && gy < H;
  {
    const int ly = py + PRE_H3, lx = px + PRE_H3;
    float4 nrm = f4(0.f, 0.f, 0.f, 1.f);
    const float4 p = sV[ly][lx];
    if (p.w > 0.0f) {
      nrm.w = 1.0f;
      const float4 u = sV[ly][lx + 1], v = sV[ly + 1][lx], s = sV[ly][lx - 1], t = sV[ly - 1][lx];
      v3 pp = xyz(p);
      v3 un = normalize3(sub3(xyz(u), pp));
      v3 vn = normalize3(sub3(xyz(v), pp));
      if (u.w < 1.0f && v.w < 1.0f) nrm.w = 0.0f;
      if (s.w < 1.0f && t.w < 1.0f) nrm.w = 0.0f;
      if (!(u.w > 0.5f) || !(v.w > 0.5f)) nrm.w = 0.0f;
      if (nrm.w > 0.0f) {
        v3 w = cross3(un, vn);
        float len = len3(w);
        const v3 wn = divs3(w, len); /* vec3 / float */
        nrm = f4(wn.x, wn.y, wn.z, (len > 0.0000001f) ? 1.0f : 0.0f);
      }
    }
    if (inside) normal[(size_t)gy * W + gx] = nrm;
  }
  __syncthreads();
  /* ---- stage C: flood fill (floodfill.frag:34-84) ---- */
  if (inside) {
    const int ey = py + 2, ex = px + 2;
    const int ly = py + PRE_H3, lx = px + PRE_H3;
    const float threshold = 0.007f;
    float4 out = sE[ey][ex];
    if (out.x == 0.0f) { /* only unlabeled texels can change */
      const float lp = len3(xyz(sV[ly][lx]));
      bool hit = false;
      for (int offset = 1; offset < 3 && !hit; ++offset) {
        const int dx[4] = {offset, 0, -offset, 0}, dy[4] = {0, offset, 0, -offset};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (hit) break;
          const float4 ql = sE[ey + dy[k]][ex + dx[k]];
          if (ql.x != 0.0f) {
            const float lq = len3(xyz(sV[ly + dy[k]][lx + dx[k]]));
            if (sdm_abs(lp - lq) < threshold * lp) {
              out = f4(ql.x, ql.y, ql.z, ql.w / (float)(offset + 1));
              hit = true;
            }
          }
        }
      }
    }
    refined[(size_t)gy * W + gx] = out;
  }
}

hipError_t launch_preprocess(suma_ctx* c, const float4* d_pts, const float* d_labels, const float* d_probs, uint32_t n,
                             uint32_t timestamp, suma_frame* out) {
  const uint32_t P = (uint32_t)c->P;
  const int32_t W = c->pd.W, H = c->pd.H;
  hipStream_t st = c->ls;
  /* on the scan pipeline's side stream K1 runs while K7 / K10 of the previous scan may still use zbuf_data */
  unsigned long long* zbuf = (st != c->stream && c->zbuf_k1) ? c->zbuf_k1 : c->zbuf_data;
  if (c->p.avg_vertexmap) { /* Preprocessing.cpp:150,160-166,191-213 (k_filters.hip) */
    hipError_t e = launch_k1_average(c, d_pts, d_labels, d_probs, n, timestamp, out->map[SUMA_MAP_VERTEX], c->eroded);
    if (e != hipSuccess) return e;
  } else {
    ProfScope ps(c, "k1_vertexmap", 24.0 * n + 32.0 * P);
    if (n > 0) k1_scatter<<<(n + 255) / 256, 256, 0, st>>>(d_pts, n, c->pd, zbuf);
    k1_resolve<<<(P + 255) / 256, 256, 0, st>>>(zbuf, d_pts, d_labels, d_probs, n, c->p.label_offset,
                                                 c->p.prob_offset, timestamp < 10 ? 1 : 0, out->map[SUMA_MAP_VERTEX],
                                                 c->eroded /* raw labels: scratch */, P);
  }
  if (c->p.filter_vertexmap && c->p.use_filtered_vertexmap) { /* :215-236; without use_filtered the result is dropped */
    hipError_t e = launch_k1c_bilateral(c, out->map[SUMA_MAP_VERTEX]);
    if (e != hipSuccess) return e;
  }
  {
    ProfScope ps(c, "k2k3_normals_labels", 64.0 * P);
    /* raw labels go to the scratch map, the fused kernel writes the refined ones into the frame
     * (Preprocessing.cpp:327 copies the refined texture over frame.semantic_map) */
    dim3 grid((W + PRE_TX - 1) / PRE_TX, (H + PRE_TY - 1) / PRE_TY);
    k23_normals_labels<<<grid, PRE_TX * PRE_TY, 0, st>>>(out->map[SUMA_MAP_VERTEX], c->eroded, out->map[SUMA_MAP_NORMAL],
                                                       out->map[SUMA_MAP_SEMANTIC], W, H);
  }
  return hipGetLastError();
}
