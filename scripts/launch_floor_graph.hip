// Chain of dependent kernel launches: plain stream launches against one hipGraph of the same chain (captured once,
// launched repeatedly).  Does a graph lower the per-launch floor of the 2 x 10^4 dependent launches of the tridiagonalisation?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main() {
  float* p; hipMalloc(&p, 1 << 24); hipMemset(p, 0, 1 << 24);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int N = 10000, g1 = 157, g2 = 850;
  for (int rep = 0; rep < 2; ++rep) {
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < N; ++i) {
      hipLaunchKernelGGL(k_touch, dim3(g1), dim3(320), 0, st, p, g1 * 320);
      hipLaunchKernelGGL(k_touch, dim3(g2), dim3(512), 0, st, p, g2 * 512);
    }
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("stream launches: %.1f ms for %d pairs = %.2f us per pair\n", ms, N, ms * 1e3 / N);
  }
  hipGraph_t graph; hipGraphExec_t exec;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < N; ++i) {
    hipLaunchKernelGGL(k_touch, dim3(g1), dim3(320), 0, st, p, g1 * 320);
    hipLaunchKernelGGL(k_touch, dim3(g2), dim3(512), 0, st, p, g2 * 512);
  }
  hipStreamEndCapture(st, &graph);
  hipEventRecord(a, st);
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  hipEventRecord(b, st); hipEventSynchronize(b);
  float msi; hipEventElapsedTime(&msi, a, b);
  printf("instantiate: %s, %.1f ms\n", hipGetErrorString(e), msi);
  for (int rep = 0; rep < 3; ++rep) {
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    hipGraphLaunch(exec, st);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("graph launch: %.1f ms for %d pairs = %.2f us per pair\n", ms, N, ms * 1e3 / N);
  }
  return 0;
}
