// Timing probe for the rocBLAS calls of the global stage (config 3 shapes).  Not part of the library.
// build: hipcc -O2 --offload-arch=gfx950 scripts/gemm_probe.hip -o scripts/gemm_probe.bin -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("fail %s = %d\n", #x, (int)e_); exit(1);} } while (0)

static float* dalloc(size_t n, float val) {
  float* p; CK(hipMalloc(&p, n * sizeof(float)));
  std::vector<float> h(1 << 20);
  for (auto& x : h) x = val * ((rand() % 2001) - 1000) / 1000.f;
  for (size_t o = 0; o < n; o += h.size()) CK(hipMemcpy(p + o, h.data(), sizeof(float) * std::min(h.size(), n - o), hipMemcpyHostToDevice));
  return p;
}

template <class F> static float timeit(F f, int reps = 2) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main(int argc, char** argv) {
  const int Rc = 55694, T = 10000;
  rocblas_handle h; CK(rocblas_create_handle(&h));
  float* M = dalloc((size_t)Rc * T, 1.f);
  float* GM = dalloc((size_t)Rc * T, 1.f);
  float* C = dalloc((size_t)T * T, 0.f);
  float* X1 = dalloc((size_t)T * T, 1.f);
  float* Rm = dalloc((size_t)Rc * T, 0.f);
  const float one = 1.f, zero = 0.f;
  for (int m : {9999, 10000, 9984}) {
    // (1) C = M^T GM  (row-major)  -> col-major: C = GM_cm * M_cm^T
    float t1 = timeit([&] { CK(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_transpose, m, m, Rc, &one, GM, T, M, T, &zero, C, T)); });
    // (2) W1 = M^T Z (m x T)
    float t2 = timeit([&] { CK(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_transpose, T, m, Rc, &one, GM, T, M, T, &zero, C, T)); });
    // (3) R = M X1 : col-major (rp x Rc) = X1_cm (rp x m) * M_cm (m x Rc)
    float t3 = timeit([&] { CK(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, m, Rc, m, &one, X1, m, M, T, &zero, Rm, m)); });
    float t3b = timeit([&] { CK(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, m, Rc, m, &one, X1, T, M, T, &zero, Rm, T)); });
    // syrkx for (1)
    float t4 = timeit([&] { CK(rocblas_ssyrkx(h, rocblas_fill_upper, rocblas_operation_none, m, Rc, &one, GM, T, M, T, &zero, C, T)); });
    // square m x m x m (Gram of Vp, Vt = Wt Vp ...)
    float t5 = timeit([&] { CK(rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, m, m, T, &one, X1, T, X1, T, &zero, C, T)); });
    float t6 = timeit([&] { CK(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, T, m, m, &one, X1, T, C, T, &zero, Rm, T)); });
    float t7 = timeit([&] { CK(rocblas_ssyrk(h, rocblas_fill_upper, rocblas_operation_transpose, m, T, &one, X1, T, &zero, C, T)); });
    const double f_big = 2.0 * m * (double)T * Rc * 1e-9, f_sq = 2.0 * m * (double)m * T * 1e-9;
    printf("m=%d  C=MtGM %.1f ms (%.0f TF)  W1=MtZ %.1f ms (%.0f TF)  R=M*X1 ld=m %.1f ms  ld=T %.1f ms (%.0f TF)  syrkx %.1f ms | sq TN %.1f (%.0f TF) sq NN %.1f  syrk %.1f\n",
           m, t1, f_big / t1, t2, f_big / t2, t3, t3b, f_big / t3b, t4, t5, f_sq / t5, t6, t7);
    fflush(stdout);
  }
  return 0;
}
