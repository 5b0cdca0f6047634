// per-kernel floor of a dependent chain on one stream: N back-to-back launches of (a) an empty kernel, (b) a kernel
// that stores one word, (c) 256 blocks x 512 threads storing one word per block
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_store(int* p) { if (threadIdx.x == 0) p[blockIdx.x] = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d; hipMalloc(&d, 4096);
  const int N = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipStreamSynchronize(s); double t = now();
    for (int i = 0; i < N; ++i) k_empty<<<1, 64, 0, s>>>();
    hipStreamSynchronize(s); double a = (now() - t) / N * 1e6;
    t = now();
    for (int i = 0; i < N; ++i) k_store<<<1, 64, 0, s>>>(d);
    hipStreamSynchronize(s); double b = (now() - t) / N * 1e6;
    t = now();
    for (int i = 0; i < N; ++i) k_store<<<256, 512, 0, s>>>(d);
    hipStreamSynchronize(s); double c = (now() - t) / N * 1e6;
    printf("us per launch in a dependent chain: empty 1x64: %.2f | store 1x64: %.2f | store 256x512: %.2f\n", a, b, c);
  }
  return 0;
}
