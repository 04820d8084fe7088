// R = M X1 written straight into pinned host memory (zero-copy) vs device memory + copy.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("fail %s = %d\n", #x, (int)e_); exit(1);} } while (0)
static float* dalloc(size_t n) { float* p; CK(hipMalloc(&p, n * sizeof(float))); CK(hipMemset(p, 0, n * sizeof(float))); return p; }
template <class F> static float timeit(F f, int reps = 2) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
  const int Rc = 55694, m = 9999;
  rocblas_handle h; CK(rocblas_create_handle(&h));
  float* M = dalloc((size_t)Rc * 10000); float* X1 = dalloc((size_t)10000 * 10000); float* R = dalloc((size_t)Rc * 10000);
  float* Rh; CK(hipHostMalloc(&Rh, (size_t)Rc * m * sizeof(float), hipHostMallocDefault));
  float* Rh_dev; CK(hipHostGetDevicePointer((void**)&Rh_dev, Rh, 0));
  const float one = 1.f, zero = 0.f;
  float t_dev = timeit([&] { CK(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, m, Rc, m, &one, X1, m, M, 10000, &zero, R, m)); });
  float t_host = timeit([&] { CK(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, m, Rc, m, &one, X1, m, M, 10000, &zero, Rh_dev, m)); });
  float t_copy = timeit([&] { CK(hipMemcpyAsync(Rh, R, (size_t)Rc * m * sizeof(float), hipMemcpyDeviceToHost, 0)); });
  printf("gemm -> device %.1f ms | gemm -> pinned host %.1f ms | D2H copy alone %.1f ms\n", t_dev, t_host, t_copy);
  return 0;
}
