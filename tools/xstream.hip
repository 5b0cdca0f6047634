// micro-benchmark: cost of a cross-stream event dependency vs same-stream ordering, and overlap of two kernels
// hipcc --offload-arch=gfx950 -O2 tools/xstream.hip -o tools/xstream.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int* sink) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t e[64];
  for (auto& x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
  int* d; hipMalloc(&d, 4);
  const long long us10 = 1000;  // wall_clock64 ticks at 100 MHz: 1000 ticks = 10 us
  const int N = 200;
  for (int rep = 0; rep < 2; ++rep) {
    // (a) same stream: N x (A, B)
    hipDeviceSynchronize(); double t = now();
    for (int i = 0; i < N; ++i) { spin<<<256, 256, 0, s1>>>(us10, d); spin<<<256, 256, 0, s1>>>(us10, d); }
    hipStreamSynchronize(s1); double a = (now() - t) / N * 1e6;
    // (b) ping-pong: A on s1, B on s2 waits A, next A waits B
    hipDeviceSynchronize(); t = now();
    for (int i = 0; i < N; ++i) {
      spin<<<256, 256, 0, s1>>>(us10, d); hipEventRecord(e[0], s1); hipStreamWaitEvent(s2, e[0], 0);
      spin<<<256, 256, 0, s2>>>(us10, d); hipEventRecord(e[1], s2); hipStreamWaitEvent(s1, e[1], 0);
    }
    hipDeviceSynchronize(); double b = (now() - t) / N * 1e6;
    // (c) fork-join: A on s1; then B (s2, 10us, 64 blocks) concurrent with C (s1, 10 us, 192 blocks); join
    hipDeviceSynchronize(); t = now();
    for (int i = 0; i < N; ++i) {
      spin<<<256, 256, 0, s1>>>(us10, d); hipEventRecord(e[0], s1); hipStreamWaitEvent(s2, e[0], 0);
      spin<<<64, 256, 0, s2>>>(us10, d); hipEventRecord(e[1], s2);
      spin<<<192, 256, 0, s1>>>(us10, d); hipStreamWaitEvent(s1, e[1], 0);
    }
    hipDeviceSynchronize(); double c = (now() - t) / N * 1e6;
    // (d) side stream lags: A,C,C on s1 per iteration; B on s2 depends on A only; s1 waits for B of the PREVIOUS iteration
    hipDeviceSynchronize(); t = now();
    for (int i = 0; i < N; ++i) {
      if (i > 0) hipStreamWaitEvent(s1, e[1 + ((i - 1) & 1)], 0);
      spin<<<256, 256, 0, s1>>>(us10, d); hipEventRecord(e[0], s1); hipStreamWaitEvent(s2, e[0], 0);
      spin<<<64, 256, 0, s2>>>(2 * us10, d); hipEventRecord(e[1 + (i & 1)], s2);
      spin<<<192, 256, 0, s1>>>(us10, d); spin<<<192, 256, 0, s1>>>(us10, d);
    }
    hipDeviceSynchronize(); double dd = (now() - t) / N * 1e6;
    printf("per iteration: same-stream A,B (20 us of work) %.1f us | ping-pong across streams %.1f us | fork-join (A, then B||C: 20 us critical) %.1f us | lagging side stream (30 us main, 20 us side) %.1f us\n", a, b, c, dd);
  }
  return 0;
}
