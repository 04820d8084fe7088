// Probe (not part of the library): the three large fp32 products of the global stage computed as sums of bf16
// products with fp32 accumulation (a = a1 + a2 (+ a3), bf16 pieces; "bf16x3" keeps a1b1 + a1b2 + a2b1, "bf16x6"
// adds a1b3 + a2b2 + a3b1), against rocBLAS sgemm: time and error relative to an fp64 reference on sampled entries.
// build: hipcc -O3 --offload-arch=gfx950 scripts/gemm_split_probe.hip -o scripts/gemm_split_probe.bin -lrocblas
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <rocblas/rocblas.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto e_ = (x); if ((int)e_ != 0) { printf("fail %s = %d (line %d)\n", #x, (int)e_, __LINE__); exit(1);} } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)(i * 2654435761u) ^ seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = scale * ((x & 0xffffff) / 8388608.0f - 1.0f);
  }
}
__global__ void split_kernel(const float* a, size_t n, rocblas_bfloat16* a1, rocblas_bfloat16* a2, rocblas_bfloat16* a3) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float x = a[i];
    __hip_bfloat16 h1 = __float2bfloat16(x); float r = x - __bfloat162float(h1);
    __hip_bfloat16 h2 = __float2bfloat16(r); r -= __bfloat162float(h2);
    __hip_bfloat16 h3 = __float2bfloat16(r);
    a1[i] = *reinterpret_cast<rocblas_bfloat16*>(&h1); a2[i] = *reinterpret_cast<rocblas_bfloat16*>(&h2); a3[i] = *reinterpret_cast<rocblas_bfloat16*>(&h3);
  }
}

template <class F> static float timeit(F f, int reps = 2) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

// column-major C(m x n) = opA(A) opB(B), fp32 via sgemm or bf16 pieces via gemm_ex
static void run_case(rocblas_handle h, const char* name, rocblas_operation ta, rocblas_operation tb, int m, int n, int k) {
  const size_t na = (size_t)m * k, nb = (size_t)k * n, nc = (size_t)m * n;
  const int lda = (ta == rocblas_operation_none) ? m : k, ldb = (tb == rocblas_operation_none) ? k : n;
  float *A, *B, *C, *C3, *C6;
  CK(hipMalloc(&A, na * 4)); CK(hipMalloc(&B, nb * 4)); CK(hipMalloc(&C, nc * 4)); CK(hipMalloc(&C3, nc * 4)); CK(hipMalloc(&C6, nc * 4));
  rocblas_bfloat16 *A1, *A2, *A3, *B1, *B2, *B3;
  CK(hipMalloc(&A1, na * 2)); CK(hipMalloc(&A2, na * 2)); CK(hipMalloc(&A3, na * 2));
  CK(hipMalloc(&B1, nb * 2)); CK(hipMalloc(&B2, nb * 2)); CK(hipMalloc(&B3, nb * 2));
  fill_kernel<<<4096, 256>>>(A, na, 1u, 1.f); fill_kernel<<<4096, 256>>>(B, nb, 2u, 1.f);
  const float one = 1.f, zero = 0.f;
  const float t_split = timeit([&] { split_kernel<<<8192, 256>>>(A, na, A1, A2, A3); split_kernel<<<8192, 256>>>(B, nb, B1, B2, B3); });
  const float t32 = timeit([&] { CK(rocblas_sgemm(h, ta, tb, m, n, k, &one, A, lda, B, ldb, &zero, C, m)); });
  auto bf = [&](const rocblas_bfloat16* a, const rocblas_bfloat16* b, float* c, const float* beta) {
    CK(rocblas_gemm_ex(h, ta, tb, m, n, k, &one, a, rocblas_datatype_bf16_r, lda, b, rocblas_datatype_bf16_r, ldb, beta, c,
                       rocblas_datatype_f32_r, m, c, rocblas_datatype_f32_r, m, rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0));
  };
  const float t1 = timeit([&] { bf(A1, B1, C3, &zero); });
  const float t3 = timeit([&] { bf(A2, B1, C3, &zero); bf(A1, B2, C3, &one); bf(A1, B1, C3, &one); });
  const float t6 = timeit([&] { bf(A3, B1, C6, &zero); bf(A2, B2, C6, &one); bf(A1, B3, C6, &one); bf(A2, B1, C6, &one); bf(A1, B2, C6, &one); bf(A1, B1, C6, &one); });
  // errors on sampled entries against fp64
  std::vector<float> hA(na), hB(nb);
  CK(hipMemcpy(hA.data(), A, na * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hB.data(), B, nb * 4, hipMemcpyDeviceToHost));
  double e32 = 0, e3 = 0, e6 = 0, nrm = 0;
  for (int s = 0; s < 200; ++s) {
    const int i = (int)((s * 7919ull) % m), j = (int)((s * 104729ull) % n);
    double ref = 0;
    for (int kk = 0; kk < k; ++kk) {
      const double a = (ta == rocblas_operation_none) ? hA[(size_t)kk * lda + i] : hA[(size_t)i * lda + kk];
      const double b = (tb == rocblas_operation_none) ? hB[(size_t)j * ldb + kk] : hB[(size_t)kk * ldb + j];
      ref += a * b;
    }
    float c32, c3, c6;
    CK(hipMemcpy(&c32, C + (size_t)j * m + i, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&c3, C3 + (size_t)j * m + i, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&c6, C6 + (size_t)j * m + i, 4, hipMemcpyDeviceToHost));
    e32 += (c32 - ref) * (c32 - ref); e3 += (c3 - ref) * (c3 - ref); e6 += (c6 - ref) * (c6 - ref); nrm += ref * ref;
  }
  const double fl = 2.0 * m * (double)n * k * 1e-9;
  printf("%-10s m=%d n=%d k=%d | sgemm %.1f ms (%.0f TF/s) err %.2e | bf16 x1 %.1f ms (%.0f TF/s) | x3 %.1f ms err %.2e | x6 %.1f ms err %.2e | split %.1f ms\n",
         name, m, n, k, t32, fl / t32, sqrt(e32 / nrm), t1, fl / t1, t3, sqrt(e3 / nrm), t6, sqrt(e6 / nrm), t_split);
  fflush(stdout);
  hipFree(A); hipFree(B); hipFree(C); hipFree(C3); hipFree(C6); hipFree(A1); hipFree(A2); hipFree(A3); hipFree(B1); hipFree(B2); hipFree(B3);
}

int main() {
  rocblas_handle h; CK(rocblas_create_handle(&h));
  const int Rc = 55804, T = 10000, m = 9999;
  // the library's row-major products as column-major calls:
  run_case(h, "W1=M^T Z", rocblas_operation_none, rocblas_operation_none, T, m, Rc);        // (T x Rc)(Rc x m): plain NN after the explicit transpose
  run_case(h, "R=M X1", rocblas_operation_none, rocblas_operation_none, m, Rc, m);           // (rp x m)(m x Rc)
  run_case(h, "C blocks", rocblas_operation_none, rocblas_operation_transpose, m, 2000, Rc); // one 2000-row block of M^T GM
  return 0;
}
