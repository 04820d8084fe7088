// Empirical operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
  const int l = threadIdx.x;
  // hypothesis: A[m = l%16][k = l/16], B[k = l/16][n = l%16]
  const double a = 100.0 * (l % 16) + (l / 16);      // A[m][k] = 100 m + k
  const double b = (l / 16 == 2) ? 1.0 * (l % 16 + 1) : 0.0;  // B[k][n] = (k == 2) * (n + 1)
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];   // expect D[m][n] = A[m][2] * (n+1) = (100 m + 2)(n + 1)
}
int main() {
  double* d; hipMalloc(&d, 256 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 15, 16, 17, 32, 48, 63})
    for (int i = 0; i < 4; ++i) {
      const double v = h[l * 4 + i];
      // decode m, n from v = (100 m + 2)(n + 1): try all
      int fm = -1, fn = -1;
      for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) if (v == (100.0 * m + 2) * (n + 1)) { fm = m; fn = n; }
      printf("lane %2d reg %d -> %8.0f  = D[%d][%d]\n", l, i, v, fm, fn);
    }
  return 0;
}
