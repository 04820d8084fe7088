// Floor of a chain of dependent kernel launches on one stream (grid shapes of the sytrd kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 1.f; }
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main() {
  float* p; hipMalloc(&p, 1 << 24); hipMemset(p, 0, 1 << 24);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 4; ++mode) {
    const int g1 = mode < 2 ? 40 : 157, g2 = mode < 2 ? 60 : 850;
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10000; ++i) {
      if (mode % 2 == 0) { hipLaunchKernelGGL(k_empty, dim3(g1), dim3(320), 0, 0, p); hipLaunchKernelGGL(k_empty, dim3(g2), dim3(512), 0, 0, p); }
      else { hipLaunchKernelGGL(k_touch, dim3(g1), dim3(320), 0, 0, p, g1 * 320); hipLaunchKernelGGL(k_touch, dim3(g2), dim3(512), 0, 0, p, g2 * 512); }
    }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("mode %d (%s, grids %d/%d): %.1f ms for 10000 pairs = %.2f us per pair\n", mode, mode % 2 ? "touch" : "empty", g1, g2, ms, ms / 10.0);
  }
  return 0;
}
