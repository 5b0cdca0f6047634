// How far apart do the workgroups of one launch START?  (wall_clock64 = 100 MHz constant clock)
// usage: dispatch_stagger   -> prints, per (blocks x threads, LDS bytes), the start-time spread in us
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ void warm(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.0f; }
// a predecessor shaped like K9: 256 blocks x 1024 threads, 128 KB of LDS, a few microseconds of work, uneven finish
__global__ void __launch_bounds__(1024) heavy(float* p, int spin) {
  __shared__ float s[32768];
  s[threadIdx.x] = p[threadIdx.x];
  __syncthreads();
  float a = s[(threadIdx.x * 7) & 1023];
  const int n = spin * (1 + (blockIdx.x & 3));
  for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) p[0] = a;
}
template <int LDS>
__global__ void probe(unsigned long long* t) {
  __shared__ char s[LDS > 0 ? LDS : 1];
  unsigned long long t0 = wall_clock64();
  if (LDS > 0) s[threadIdx.x] = (char)t0;
  __syncthreads();
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = wall_clock64() + (LDS > 0 ? s[1] & 0 : 0); }
}
template <int LDS>
static void run(int blocks, int threads, float* w, unsigned long long* d, int pred = 0) {
  std::vector<unsigned long long> h(2 * blocks);
  double spread = 0, bar = 0;
  const int reps = 20;
  for (int r = 0; r < reps; ++r) {
    if (pred) heavy<<<256, 1024>>>(w, 2000); else warm<<<1024, 256>>>(w);  // a predecessor in the same stream, as in the pipeline
    probe<LDS><<<blocks, threads>>>(d);
    (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0, b = 0;
    for (int i = 0; i < blocks; ++i) { lo = std::min(lo, h[2 * i]); hi = std::max(hi, h[2 * i]); b = std::max(b, h[2 * i + 1] - h[2 * i]); }
    if (r >= 5) { spread += (hi - lo) / 100.0; bar += b / 100.0; }
  }
  printf("%s%5d blocks x %4d threads, %6d B LDS: first-to-last block start %6.2f us, slowest first barrier %5.2f us\n", pred ? "[after a K9-shaped kernel] " : "", blocks, threads,
         LDS, spread / (reps - 5), bar / (reps - 5));
}
int main() {
  float* w; unsigned long long* d;
  (void)hipMalloc(&w, 1024 * 256 * 4); (void)hipMemset(w, 0, 1024 * 256 * 4); (void)hipMalloc(&d, 16 * 8192);
  run<0>(128, 1024, w, d); run<0>(256, 1024, w, d); run<0>(256, 512, w, d); run<0>(512, 256, w, d); run<0>(2048, 256, w, d);
  run<65536>(256, 1024, w, d); run<65536>(256, 512, w, d); run<8192>(128, 1024, w, d);
  run<8192>(128, 1024, w, d, 1); run<8192>(512, 256, w, d, 1); run<0>(2048, 256, w, d, 1);
  return 0;
}
