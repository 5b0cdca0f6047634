// per-kernel floor of a dependent chain on one stream: N back-to-back launches of (a) an empty kernel, (b) a kernel
// that stores one word, (c) 256 blocks x 512 threads storing one word per block
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_store(int* p) { if (threadIdx.x == 0) p[blockIdx.x] = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int graph_main();
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
  graph_main();
  return 0;
}
// appended: the same dependent chain as a hipGraph (stream capture), launched 20 times
int graph_main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d; hipMalloc(&d, 4096);
  const int N = 200;
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < N; ++i) k_store<<<256, 512, 0, s>>>(d);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  double t = now();
  for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  printf("hipGraph of %d dependent store kernels: %.2f us per kernel node\n", N, (now() - t) / (20.0 * N) * 1e6);
  // a short graph inside a stream of normal launches, as a Gauss-Newton chain would use it:
  //   (a) 11 normal launches  vs  (b) 1 normal + graph of 9 + 1 normal, repeated
  hipGraph_t g9; hipGraphExec_t ge9;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 9; ++i) k_store<<<256, 512, 0, s>>>(d);
  hipStreamEndCapture(s, &g9);
  hipGraphInstantiate(&ge9, g9, nullptr, nullptr, 0);
  hipGraphLaunch(ge9, s); hipStreamSynchronize(s);
  const int R = 300;
  t = now();
  for (int r = 0; r < R; ++r) for (int i = 0; i < 11; ++i) k_store<<<256, 512, 0, s>>>(d);
  hipStreamSynchronize(s); double a = (now() - t) / R * 1e6;
  t = now();
  for (int r = 0; r < R; ++r) { k_store<<<256, 512, 0, s>>>(d); hipGraphLaunch(ge9, s); k_store<<<256, 512, 0, s>>>(d); }
  hipStreamSynchronize(s); double b = (now() - t) / R * 1e6;
  printf("11 dependent launches: %.1f us | 1 + graph(9) + 1: %.1f us\n", a, b);
  return 0;
}
