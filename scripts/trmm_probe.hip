// rocblas_strmm (out of place) vs sgemm for the two triangular products of the final stage (Et lower triangular).
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("fail %s = %d\n", #x, (int)e_); exit(1);} } while (0)
static float* dalloc(size_t n) { float* p; CK(hipMalloc(&p, n * sizeof(float))); CK(hipMemset(p, 0, n * sizeof(float))); return p; }
template <class F> static float timeit(F f, int reps = 3) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
  const int m = 9999, T = 10000;
  rocblas_handle h; CK(rocblas_create_handle(&h));
  float* Et = dalloc((size_t)T * T); float* W = dalloc((size_t)T * T); float* O = dalloc((size_t)T * T);
  const float one = 1.f, zero = 0.f;
  // Vp (row-major m x T) = Et (m x m lower) W1 (m x T): column-major view Vp^T (T x m) = W1^T (T x m) * Et^T (upper, right side)
  float g1 = timeit([&] { CK(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, T, m, m, &one, W, T, Et, T, &zero, O, T)); });
  float t1 = timeit([&] { CK(rocblas_strmm(h, rocblas_side_right, rocblas_fill_upper, rocblas_operation_none, rocblas_diagonal_non_unit, T, m, &one, Et, T, W, T, O, T)); });
  // X1 (row-major m x rp) = Et^T W: column-major X1^T (rp x m) = W^T (rp x m) * Et (col-major view of row-major Et is Et^T...) -> right side, transposed
  float g2 = timeit([&] { CK(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_transpose, m, m, m, &one, W, T, Et, T, &zero, O, T)); });
  float t2 = timeit([&] { CK(rocblas_strmm(h, rocblas_side_right, rocblas_fill_upper, rocblas_operation_transpose, rocblas_diagonal_non_unit, m, m, &one, Et, T, W, T, O, T)); });
  printf("Vp: sgemm %.2f ms  strmm %.2f ms | X1: sgemm %.2f ms  strmm %.2f ms\n", g1, t1, g2, t2);
  return 0;
}
