/*
 * k_filters.hip -- the optional vertex-map filters of Preprocessing::process on gfx950 (off in config/default.xml).
 *
 * Replaces (reference, citations relative to /root/reference):
 *   K1 in its avg_vertexmap mode  src/core/Preprocessing.cpp:150,160-166 + gen_vertexmap.vert/.frag
 *      GL: the K1 point scatter with the depth test OFF and glBlendFunc(GL_ONE, GL_ONE) on both colour attachments.
 *      Fragments of one pixel blend in primitive order, so a texel ends up with the fp32 sum ((f0 + f1) + f2) ... of
 *      its points taken by ascending point index -- not associative, so neither atomics nor a tree will do.
 *      Here: one 64-bit key (pixel << 32 | point index) per point, sorted (rocPRIM radix sort: a plain library sort;
 *      the keys are unique, so the result does not depend on its stability), then one lane per pixel run adds its
 *      points in key order = ascending index.
 *   K1b Preprocessing.cpp:191-213 + avg_vertexmap.frag:14-21       (sum / count)
 *   K1c Preprocessing.cpp:215-236 + bilateral_filter.frag:28-83    (13 x 13 bilateral range filter)
 *
 * Sampling: both filter shaders read their input with texture(sampler2DRect, INTEGER coordinate) and no sampler
 * object bound; `filter_sampling` (suma_types.h) names the texture state -- NEAREST, or the GL initial state of a
 * rectangle texture (LINEAR + CLAMP_TO_EDGE), under which an integer coordinate is a texel corner and the fetch is
 * the mean of the four texels around it (filter_fetch below, operation order of GL 3.3 core 3.8.11).
 * The texel a fragment addresses, int(texCoords.x * width) with texCoords = (x + 1/2) / width, is x itself for every
 * width up to 8192 (checked exhaustively, tests/test_oracle_kat.py); larger images are refused with the filters on.
 */
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "suma_internal.h"

/* ---- K1, sum mode ---------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
    k1s_keys(const float4* __restrict__ pts, uint32_t n, proj_t q, uint32_t P, unsigned long long* __restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 pt = pts[i];
  const v3 pos = mk3(pt.x, pt.y, pt.z);
  const float depth = len3(pos);
  const float yaw = sdm_atan2(pos.y, pos.x);
  const float pitch = -sdm_asin(pos.z / depth);
  const float x = (-yaw * SUMA_INV_PI_F);
  const float y = (1.0f - (2.0f * ((pitch * SUMA_RAD2DEG_F) + q.fov_up)) / q.fov);
  const float z = 2.0f * ((depth - q.min_depth) / (q.max_depth - q.min_depth)) - 1.0f;
  const float fx = sdm_floor((0.5f * (x + 1.0f)) * q.width);
  const float fy = sdm_floor((0.5f * (y + 1.0f)) * q.height);
  uint32_t key = P; /* clipped (or NaN): sorted behind every pixel */
  if ((fx >= 0.0f && fx < q.width && fy >= 0.0f && fy < q.height) && (z >= -1.0f && z <= 1.0f))
    key = (uint32_t)(int32_t)fy * (uint32_t)q.W + (uint32_t)(int32_t)fx;
  keys[i] = ((unsigned long long)key << 32) | (unsigned long long)i;
}

/* one lane per sorted position; the lane at the head of a pixel's run walks the run */
__global__ void __launch_bounds__(256)
    k1s_sum(const unsigned long long* __restrict__ keys, uint32_t n, uint32_t P,
            const float4* __restrict__ pts, const float* __restrict__ labels, const float* __restrict__ probs,
            uint32_t label_offset, uint32_t prob_offset, int isfirst, float4* __restrict__ vsum,
            float4* __restrict__ ssum) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t pix = (uint32_t)(keys[j] >> 32);
  if (pix >= P || (j > 0 && (uint32_t)(keys[j - 1] >> 32) == pix)) return;
  float4 a = f4(0.f, 0.f, 0.f, 0.f), b = a; /* glClear */
  for (uint32_t k = j; k < n; ++k) {
    const unsigned long long key = keys[k];
    if ((uint32_t)(key >> 32) != pix) break;
    const uint32_t i = (uint32_t)(key & 0xffffffffull);
    const unsigned long long li = (unsigned long long)i + label_offset, pi = (unsigned long long)i + prob_offset;
    const float label = (labels != nullptr && li < n) ? labels[li] : 0.0f;
    const float prob = (probs != nullptr && pi < n) ? probs[pi] : 0.0f;
    const float4 pt = pts[i];
    float4 v = f4(pt.x, pt.y, pt.z, 1.0f);
    if (isfirst && is_dynamic_label(label)) v = f4(0.f, 0.f, 0.f, 0.f);
    const float l = label / 255.0f;
    a = f4(v.x + a.x, v.y + a.y, v.z + a.z, v.w + a.w);
    b = f4(l + b.x, l + b.y, l + b.z, prob + b.w);
  }
  vsum[pix] = a;
  ssum[pix] = b;
}

/* ---- texture(in_vertexmap, (x, y)) at an integer coordinate ------------------------------------------------ */
__device__ __forceinline__ float4 filter_fetch(const float4* __restrict__ M, int32_t W, int32_t x, int32_t y, int nearest) {
  if (nearest) return M[(size_t)y * W + x];
  const int32_t i0 = x - 1 < 0 ? 0 : x - 1, j0 = y - 1 < 0 ? 0 : y - 1;
  const float a = 0.5f, b = 0.5f;
  const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  const float4 t00 = M[(size_t)j0 * W + i0], t10 = M[(size_t)j0 * W + x], t01 = M[(size_t)y * W + i0],
               t11 = M[(size_t)y * W + x];
  return f4(((t00.x * w00 + t10.x * w10) + t01.x * w01) + t11.x * w11,
            ((t00.y * w00 + t10.y * w10) + t01.y * w01) + t11.y * w11,
            ((t00.z * w00 + t10.z * w10) + t01.z * w01) + t11.z * w11,
            ((t00.w * w00 + t10.w * w10) + t01.w * w01) + t11.w * w11);
}

/* ---- K1b avg_vertexmap.frag:14-21 ---------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
    k1b_average(const float4* __restrict__ vsum, float4* __restrict__ out, int32_t W, int32_t H, int nearest) {
  const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (uint32_t)W * (uint32_t)H) return;
  const int32_t y = (int32_t)(pix / (uint32_t)W), x = (int32_t)(pix - (uint32_t)y * (uint32_t)W);
  float4 v = filter_fetch(vsum, W, x, y, nearest);
  if (v.w > 0.5f) { /* vec4 / float: one reciprocal, four multiplies (dev_math.h divs3) */
    const float rw = 1.0f / v.w;
    v = f4(v.x * rw, v.y * rw, v.z * rw, v.w * rw);
  }
  out[pix] = v;
}

/* ---- K1c bilateral_filter.frag:28-83 ---------------------------------------------------------------------------
 * A block owns a BF_TX x BF_TY patch and stages the FETCHED texels of the patch + 6 columns / rows around it in
 * LDS (columns through the shader's wrap(), rows outside the image never read: the window is cut at the image in
 * y); each lane then runs the 13 x 13 window of its pixel from LDS in the shader's order (rows outer, columns
 * inner: the two sums are sequential fp32 additions). */
#define BF_R 6
#define BF_TX 64
#define BF_TY 4
#define BF_SW (BF_TX + 2 * BF_R)
#define BF_SH (BF_TY + 2 * BF_R)

__device__ __forceinline__ float bf_wrap(float x, float dim) { /* bilateral_filter.frag:18-26 */
  float value = x;
  while (value >= dim) value = (value - dim);
  while (value < 0.0f) value = (value + dim);
  return value;
}

__global__ void __launch_bounds__(BF_TX* BF_TY)
    k1c_bilateral(const float4* __restrict__ V, float4* __restrict__ out, int32_t W, int32_t H, float sigma_space,
                  float sigma_range, int nearest) {
  __shared__ float4 sT[BF_SH][BF_SW];
  const int32_t x0 = blockIdx.x * BF_TX, y0 = blockIdx.y * BF_TY;
  const float width = (float)W, height = (float)H;
  for (int t = threadIdx.x; t < BF_SW * BF_SH; t += BF_TX * BF_TY) {
    const int ly = t / BF_SW, lx = t - ly * BF_SW;
    const int32_t cy = y0 + ly - BF_R;
    float4 v = f4(0.f, 0.f, 0.f, 0.f);
    if (cy >= 0 && cy < H) v = filter_fetch(V, W, (int32_t)bf_wrap((float)(x0 + lx - BF_R), width), cy, nearest);
    sT[ly][lx] = v;
  }
  __syncthreads();
  const int px = threadIdx.x % BF_TX, py = threadIdx.x / BF_TX;
  const int32_t x = x0 + px, y = y0 + py;
  if (x >= W || y >= H) return;
  const float4 vertex = sT[py + BF_R][px + BF_R];
  float4 res = vertex;
  if (vertex.w > 0.5f) {
    const float range = len3(xyz(vertex));
    const v3 ray = divs3(xyz(vertex), range);
    const float sigma_space_factor = -0.5f / (sigma_space * sigma_space);
    const float sigma_range_factor = -0.5f / (sigma_range * sigma_range);
    const int32_t D = BF_R * 2 + 1;
    const int32_t tx = x - D / 2 + D;
    const int32_t ty = y - D / 2 + D < (int32_t)height ? y - D / 2 + D : (int32_t)height;
    float sum1 = 0.0f, sum2 = 0.0f;
    for (int32_t cy = (y - D / 2 > 0 ? y - D / 2 : 0); cy < ty; ++cy) {
      for (int32_t cx = x - D / 2; cx < tx; ++cx) {
        const float xx = bf_wrap((float)cx, width);
        const float4 tmp = sT[cy - y0 + BF_R][cx - x0 + BF_R];
        if (tmp.w < 0.5f) continue;
        /* length(vec4) = sqrt(dot): the fused chain of dev_math.h over four components */
        const float tmp_range = sdm_sqrt(SDEV_FMA(tmp.w, tmp.w, SDEV_FMA(tmp.z, tmp.z, SDEV_FMA(tmp.y, tmp.y, tmp.x * tmp.x))));
        const float dx = (float)x - xx;
        const float diff_space2 = dx * dx + (float)((y - cy) * (y - cy));
        const float diff_range2 = (range - tmp_range) * (range - tmp_range);
        const float weight = sdm_exp(diff_space2 * sigma_space_factor + diff_range2 * sigma_range_factor);
        sum1 += tmp_range * weight;
        sum2 += weight;
      }
    }
    const float filtered_range = sum1 / sum2;
    res = f4(filtered_range * ray.x, filtered_range * ray.y, filtered_range * ray.z, 1.0f);
  }
  out[(size_t)y * W + x] = res;
}

/* ---- host side ----------------------------------------------------------------------------------------------- */
static uint32_t key_bits(uint32_t P) { /* pixel part 0 .. P above the 32 index bits */
  uint32_t b = 1;
  while (b < 32 && (P >> b) != 0) ++b;
  return 32 + b;
}

static hipError_t filters_reserve(suma_ctx* c, uint32_t n) {
  hipError_t e;
  if (!c->filt_temp && (e = hipMalloc((void**)&c->filt_temp, c->P * sizeof(float4))) != hipSuccess) return e;
  if (n <= c->filt_cap) return hipSuccess;
  const uint32_t cap = n + n / 4 + 1024;
  /* the buffers may still be read by a sort in flight on either stream: drain before freeing */
  if (c->filt_sort || c->filt_sort_tmp) {
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return e;
    if (c->side_stream && (e = hipStreamSynchronize(c->side_stream)) != hipSuccess) return e;
  }
  if (c->filt_sort) hipFree(c->filt_sort);
  if (c->filt_sort_tmp) hipFree(c->filt_sort_tmp);
  c->filt_sort = nullptr;
  c->filt_sort_tmp = nullptr;
  c->filt_cap = 0;
  if ((e = hipMalloc((void**)&c->filt_sort, (size_t)2 * cap * sizeof(unsigned long long))) != hipSuccess) return e;
  size_t bytes = 0;
  unsigned long long* k = c->filt_sort;
  e = rocprim::radix_sort_keys(nullptr, bytes, k, k + cap, cap, 0, key_bits((uint32_t)c->P), c->stream);
  if (e != hipSuccess) return e;
  if ((e = hipMalloc(&c->filt_sort_tmp, bytes ? bytes : 16)) != hipSuccess) return e;
  c->filt_sort_tmp_bytes = bytes;
  c->filt_cap = cap;
  return hipSuccess;
}

/* K1 in sum mode + K1b: the averaged vertex map goes to `vertex`, the summed label texels to `raw_semantic` */
hipError_t launch_k1_average(suma_ctx* c, const float4* d_pts, const float* d_labels, const float* d_probs, uint32_t n,
                             uint32_t timestamp, float4* vertex, float4* raw_semantic) {
  const uint32_t P = (uint32_t)c->P;
  hipStream_t st = c->ls;
  hipError_t e = filters_reserve(c, n);
  if (e != hipSuccess) return e;
  ProfScope ps(c, "k1_sum_k1b_average", 40.0 * n + 64.0 * P);
  if ((e = hipMemsetAsync(c->filt_temp, 0, (size_t)P * sizeof(float4), st)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(raw_semantic, 0, (size_t)P * sizeof(float4), st)) != hipSuccess) return e;
  if (n > 0) {
    unsigned long long *keys = c->filt_sort, *skeys = keys + c->filt_cap;
    k1s_keys<<<(n + 255) / 256, 256, 0, st>>>(d_pts, n, c->pd, P, keys);
    size_t bytes = c->filt_sort_tmp_bytes;
    e = rocprim::radix_sort_keys(c->filt_sort_tmp, bytes, keys, skeys, n, 0, key_bits(P), st);
    if (e != hipSuccess) return e;
    k1s_sum<<<(n + 255) / 256, 256, 0, st>>>(skeys, n, P, d_pts, d_labels, d_probs, c->p.label_offset,
                                            c->p.prob_offset, timestamp < 10 ? 1 : 0, c->filt_temp, raw_semantic);
  }
  k1b_average<<<(P + 255) / 256, 256, 0, st>>>(c->filt_temp, vertex, c->pd.W, c->pd.H,
                                               c->p.filter_sampling == SUMA_FILTER_SAMPLING_NEAREST);
  return hipGetLastError();
}

/* K1c; `vertex` is replaced by its filtered version (the caller runs this only with use_filtered_vertexmap: the
 * reference computes the filter either way and drops the result otherwise, Preprocessing.cpp:234) */
hipError_t launch_k1c_bilateral(suma_ctx* c, float4* vertex) {
  const int32_t W = c->pd.W, H = c->pd.H;
  hipStream_t st = c->ls;
  hipError_t e = filters_reserve(c, 0);
  if (e != hipSuccess) return e;
  ProfScope ps(c, "k1c_bilateral", 48.0 * c->P);
  dim3 grid((W + BF_TX - 1) / BF_TX, (H + BF_TY - 1) / BF_TY);
  k1c_bilateral<<<grid, BF_TX * BF_TY, 0, st>>>(vertex, c->filt_temp, W, H, c->p.bilateral_sigma_space,
                                                c->p.bilateral_sigma_range,
                                                c->p.filter_sampling == SUMA_FILTER_SAMPLING_NEAREST);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return hipMemcpyAsync(vertex, c->filt_temp, c->P * sizeof(float4), hipMemcpyDeviceToDevice, st);
}
