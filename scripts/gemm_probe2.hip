// R = M X1 in row blocks: rate of rocblas_sgemm(N, N, rp, rows, m) as a function of the block height.
// build: hipcc -O2 --offload-arch=gfx950 scripts/gemm_probe2.hip -o scripts/gemm_probe2.bin -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("fail %s = %d\n", #x, (int)e_); exit(1);} } while (0)
static float* dalloc(size_t n) { float* p; CK(hipMalloc(&p, n * sizeof(float))); CK(hipMemset(p, 0, n * sizeof(float))); return p; }
template <class F> static float timeit(F f, int reps = 3) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
  const int Rc = 55694, m = 9999;
  rocblas_handle h; CK(rocblas_create_handle(&h));
  float* M = dalloc((size_t)Rc * 10000); float* X1 = dalloc((size_t)10000 * 10000); float* R = dalloc((size_t)Rc * 10000);
  const float one = 1.f, zero = 0.f;
  for (int rows : {6962, 7168, 9283, 11139, 13924, 14336, 18565, 27847, 55694}) {
    float t = timeit([&] { CK(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, m, rows, m, &one, X1, m, M, 10000, &zero, R, m)); });
    printf("rows %6d  %.2f ms  %.0f TF/s   (x%d blocks = %.1f ms)\n", rows, t, 2.0 * m * (double)rows * m * 1e-9 / t, (Rc + rows - 1) / rows,
           t * ((Rc + rows - 1) / rows));
  }
  return 0;
}
