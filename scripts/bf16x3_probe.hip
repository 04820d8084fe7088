// Probe: fp32 products of the global stage through the bf16 matrix cores (three-way bf16 split of both operands, six
// bf16 x bf16 -> fp32 products), against rocBLAS sgemm.  Rates for the three big shapes of the headline step and the
// error of each form against a double-precision product.
//   hipcc -O2 --offload-arch=gfx950 scripts/bf16x3_probe.hip -o scripts/bf16x3_probe.bin -lrocblas -lhipblaslt
#include <hip/hip_runtime.h>
#include <hip/hip_bfloat16.h>
#include <hipblaslt/hipblaslt.h>
#include <rocblas/rocblas.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define BK(x) do { rocblas_status s_ = (x); if (s_ != rocblas_status_success) { printf("rocBLAS error %d at %d\n", (int)s_, __LINE__); exit(1); } } while (0)
#define LK(x) do { hipblasStatus_t s_ = (x); if (s_ != HIPBLAS_STATUS_SUCCESS) { printf("hipBLASLt error %d at %d\n", (int)s_, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* x, long n, unsigned seed, float scale) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  unsigned h2 = h * 747796405u + 2891336453u;
  h2 ^= h2 >> 16;
  float u1 = ((h >> 8) + 1) * (1.0f / 16777217.0f), u2 = (h2 >> 8) * (1.0f / 16777216.0f);
  x[i] = scale * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

static __device__ inline unsigned short bf16_rn(float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static __device__ inline float bf16_f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// x = x1 + x2 + x3 exactly (8 significant bits each), unless the tail underflows
__global__ void split3_kernel(const float* __restrict__ x, long n, unsigned short* __restrict__ x1, unsigned short* __restrict__ x2,
                              unsigned short* __restrict__ x3) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  unsigned short a = bf16_rn(v);
  float r = v - bf16_f(a);
  unsigned short b = bf16_rn(r);
  r -= bf16_f(b);
  x1[i] = a; x2[i] = b; x3[i] = bf16_rn(r);
}

// x * 2^-e = h1 + 2^-11 h2 with fp16 pieces (11 significant bits each): |x 2^-e - h1 - 2^-11 h2| <= 2^-24 |x 2^-e|
__global__ void absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned u = __float_as_uint(x[i]) & 0x7fffffffu;
    m = u > m ? u : m;
  }
  for (int o = 32; o; o >>= 1) { unsigned t = __shfl_xor(m, o); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
// scal[0] = 2^-e with max |x| 2^-e in [2^13, 2^14);  scal[1] = 2^e
__global__ void expo_kernel(const unsigned* __restrict__ amax, float* __restrict__ scal) {
  int be = (int)((*amax >> 23) & 0xff);   // biased exponent of the largest magnitude
  if (be == 0) be = 1;
  int e = be - 127 - 13;
  scal[0] = __uint_as_float((unsigned)(127 - e) << 23);
  scal[1] = __uint_as_float((unsigned)(127 + e) << 23);
}
__global__ void split2h_kernel(const float* __restrict__ x, long n, const float* __restrict__ scal, _Float16* __restrict__ h1,
                               _Float16* __restrict__ h2) {
  const float sc = scal[0];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = x[i] * sc;
    _Float16 a = (_Float16)v;
    float r = (v - (float)a) * 2048.f;
    h1[i] = a;
    h2[i] = (_Float16)r;
  }
}
// alphas[0] = 2^(ea + eb), alphas[1] = 2^(ea + eb - 11), alphas[2] = 1, alphas[3] = 0
__global__ void alphas_kernel(const float* sa, const float* sb, float* alphas) {
  alphas[0] = sa[1] * sb[1];
  alphas[1] = sa[1] * sb[1] * (1.f / 2048.f);
  alphas[2] = 1.f;
  alphas[3] = 0.f;
}
__global__ void abs_kernel(float* x, long n) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) x[i] = fabsf(x[i]);
}
__global__ void rowscale_kernel(float* x, long rows, long cols) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < rows * cols) { long c = i % cols; x[i] *= exp10f(-6.f * (float)((c * 7919) % cols) / (float)cols); }
}

__global__ void f2d_kernel(const float* x, double* y, long n) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i];
}

struct Lt {
  hipblasLtHandle_t h;
  void* ws; size_t ws_bytes;
};

// row-major C (m x n, fp32) (+)= op(A) op(B), A / B bf16 row-major; returns 0 when no algorithm was found
static int lt_gemm_x(Lt& lt, hipStream_t st, hipDataType dt, int tA, int tB, int m, int n, int k, const void* A, long lda, const void* B, long ldb,
                     const float* alpha_dev, const float* beta_dev, float* C, long ldc) {
  hipblasLtMatmulDesc_t desc;
  LK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  hipblasOperation_t opa = tB ? HIPBLAS_OP_T : HIPBLAS_OP_N, opb = tA ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  LK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa)));
  LK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb)));
  int32_t pm = HIPBLASLT_POINTER_MODE_DEVICE;
  LK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_POINTER_MODE, &pm, sizeof(pm)));
  hipblasLtMatrixLayout_t la, lb, lc;
  LK(hipblasLtMatrixLayoutCreate(&la, dt, tB ? k : n, tB ? n : k, ldb));
  LK(hipblasLtMatrixLayoutCreate(&lb, dt, tA ? m : k, tA ? k : m, lda));
  LK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, n, m, ldc));
  hipblasLtMatmulPreference_t pref;
  LK(hipblasLtMatmulPreferenceCreate(&pref));
  LK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &lt.ws_bytes, sizeof(lt.ws_bytes)));
  hipblasLtMatmulHeuristicResult_t res[1];
  int found = 0;
  hipblasStatus_t s = hipblasLtMatmulAlgoGetHeuristic(lt.h, desc, la, lb, lc, lc, pref, 1, res, &found);
  int ok = 0;
  if (s == HIPBLAS_STATUS_SUCCESS && found > 0) {
    hipblasStatus_t s2 = hipblasLtMatmul(lt.h, desc, alpha_dev, B, la, A, lb, beta_dev, C, lc, C, lc, &res[0].algo, lt.ws, lt.ws_bytes, st);
    ok = s2 == HIPBLAS_STATUS_SUCCESS;
    if (!ok) printf("  (hipblasLtMatmul device pointer mode: status %d)\n", (int)s2);
  }
  hipblasLtMatmulPreferenceDestroy(pref);
  hipblasLtMatrixLayoutDestroy(la); hipblasLtMatrixLayoutDestroy(lb); hipblasLtMatrixLayoutDestroy(lc);
  hipblasLtMatmulDescDestroy(desc);
  return ok;
}

static int lt_gemm(Lt& lt, hipStream_t st, int tA, int tB, int m, int n, int k, const void* A, long lda, const void* B, long ldb, float beta,
                   float* C, long ldc, hipDataType dt = HIP_R_16BF, float alpha = 1.f) {
  // column-major view: C^T (n x m) = op(B)^T op(A)^T
  hipblasLtMatmulDesc_t desc;
  LK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  hipblasOperation_t opa = tB ? HIPBLAS_OP_T : HIPBLAS_OP_N, opb = tA ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  LK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa)));
  LK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb)));
  hipblasLtMatrixLayout_t la, lb, lc;
  // first operand: B seen column-major: (tB ? k x n : n x k) with ld ldb
  LK(hipblasLtMatrixLayoutCreate(&la, dt, tB ? k : n, tB ? n : k, ldb));
  LK(hipblasLtMatrixLayoutCreate(&lb, dt, tA ? m : k, tA ? k : m, lda));
  LK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, n, m, ldc));
  hipblasLtMatmulPreference_t pref;
  LK(hipblasLtMatmulPreferenceCreate(&pref));
  LK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &lt.ws_bytes, sizeof(lt.ws_bytes)));
  hipblasLtMatmulHeuristicResult_t res[1];
  int found = 0;
  hipblasStatus_t s = hipblasLtMatmulAlgoGetHeuristic(lt.h, desc, la, lb, lc, lc, pref, 1, res, &found);
  int ok = 0;
  if (s == HIPBLAS_STATUS_SUCCESS && found > 0) {
    LK(hipblasLtMatmul(lt.h, desc, &alpha, B, la, A, lb, &beta, C, lc, C, lc, &res[0].algo, lt.ws, lt.ws_bytes, st));
    ok = 1;
  }
  hipblasLtMatmulPreferenceDestroy(pref);
  hipblasLtMatrixLayoutDestroy(la); hipblasLtMatrixLayoutDestroy(lb); hipblasLtMatrixLayoutDestroy(lc);
  hipblasLtMatmulDescDestroy(desc);
  return ok;
}

static void rb_gemm_bf16(rocblas_handle h, int tA, int tB, int m, int n, int k, const void* A, long lda, const void* B, long ldb, float beta,
                         float* C, long ldc) {
  float alpha = 1.f;
  BK(rocblas_gemm_ex(h, tB ? rocblas_operation_transpose : rocblas_operation_none, tA ? rocblas_operation_transpose : rocblas_operation_none,
                     n, m, k, &alpha, B, rocblas_datatype_bf16_r, (rocblas_int)ldb, A, rocblas_datatype_bf16_r, (rocblas_int)lda, &beta, C,
                     rocblas_datatype_f32_r, (rocblas_int)ldc, C, rocblas_datatype_f32_r, (rocblas_int)ldc, rocblas_datatype_f32_r,
                     rocblas_gemm_algo_standard, 0, 0));
}

static void rb_sgemm(rocblas_handle h, int tA, int tB, int m, int n, int k, const float* A, long lda, const float* B, long ldb, float beta, float* C,
                     long ldc) {
  float alpha = 1.f;
  BK(rocblas_sgemm(h, tB ? rocblas_operation_transpose : rocblas_operation_none, tA ? rocblas_operation_transpose : rocblas_operation_none, n, m, k,
                   &alpha, B, (rocblas_int)ldb, A, (rocblas_int)lda, &beta, C, (rocblas_int)ldc));
}

static void rb_dgemm(rocblas_handle h, int tA, int tB, int m, int n, int k, const double* A, long lda, const double* B, long ldb, double* C, long ldc) {
  double alpha = 1., beta = 0.;
  BK(rocblas_dgemm(h, tB ? rocblas_operation_transpose : rocblas_operation_none, tA ? rocblas_operation_transpose : rocblas_operation_none, n, m, k,
                   &alpha, B, (rocblas_int)ldb, A, (rocblas_int)lda, &beta, C, (rocblas_int)ldc));
}

struct SplitH { _Float16 *h1, *h2; float* scal; unsigned* amax; };
static SplitH splith(const float* x, long n, hipStream_t st) {
  SplitH s;
  CK(hipMalloc(&s.h1, n * 2)); CK(hipMalloc(&s.h2, n * 2)); CK(hipMalloc(&s.scal, 8)); CK(hipMalloc(&s.amax, 4));
  CK(hipMemsetAsync(s.amax, 0, 4, st));
  hipLaunchKernelGGL(absmax_kernel, dim3(4096), dim3(256), 0, st, x, n, s.amax);
  hipLaunchKernelGGL(expo_kernel, dim3(1), dim3(1), 0, st, s.amax, s.scal);
  hipLaunchKernelGGL(split2h_kernel, dim3(8192), dim3(256), 0, st, x, n, s.scal, s.h1, s.h2);
  return s;
}
// three fp16 products with host scalars (2^-11 on the two small ones), then C *= 2^(ea + eb) read on the device
__global__ void unscale_kernel(float* C, long n, const float* sa, const float* sb) {
  const float f = sa[1] * sb[1];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) C[i] *= f;
}
static int g_devmode = 0;
static int three(Lt& lt, hipStream_t st, float* alphas, int tA, int tB, int m, int n, int k, const SplitH& a, long aoff, long lda, const SplitH& b,
                 long boff, long ldb, int accumulate, float* C, long ldc) {
  int ok = 1;
  if (g_devmode) {
    hipLaunchKernelGGL(alphas_kernel, dim3(1), dim3(1), 0, st, a.scal, b.scal, alphas);
    ok &= lt_gemm_x(lt, st, HIP_R_16F, tA, tB, m, n, k, a.h1 + aoff, lda, b.h2 + boff, ldb, alphas + 1, alphas + (accumulate ? 2 : 3), C, ldc);
    ok &= lt_gemm_x(lt, st, HIP_R_16F, tA, tB, m, n, k, a.h2 + aoff, lda, b.h1 + boff, ldb, alphas + 1, alphas + 2, C, ldc);
    ok &= lt_gemm_x(lt, st, HIP_R_16F, tA, tB, m, n, k, a.h1 + aoff, lda, b.h1 + boff, ldb, alphas + 0, alphas + 2, C, ldc);
    return ok;
  }
  // (accumulate: C holds a scaled partial sum from the previous chunk; the caller unscales once at the end)
  ok &= lt_gemm(lt, st, tA, tB, m, n, k, a.h1 + aoff, lda, b.h2 + boff, ldb, accumulate ? 1.f : 0.f, C, ldc, HIP_R_16F, 1.f / 2048.f);
  ok &= lt_gemm(lt, st, tA, tB, m, n, k, a.h2 + aoff, lda, b.h1 + boff, ldb, 1.f, C, ldc, HIP_R_16F, 1.f / 2048.f);
  ok &= lt_gemm(lt, st, tA, tB, m, n, k, a.h1 + aoff, lda, b.h1 + boff, ldb, 1.f, C, ldc, HIP_R_16F, 1.f);
  return ok;
}
static void unscale(hipStream_t st, float* C, long n, const SplitH& a, const SplitH& b) {
  hipLaunchKernelGGL(unscale_kernel, dim3(4096), dim3(256), 0, st, C, n, a.scal, b.scal);
}

struct Split { unsigned short *p1, *p2, *p3; };
static Split split(const float* x, long n, hipStream_t st) {
  Split s;
  CK(hipMalloc(&s.p1, n * 2)); CK(hipMalloc(&s.p2, n * 2)); CK(hipMalloc(&s.p3, n * 2));
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, s.p1, s.p2, s.p3);
  return s;
}
static void drop(Split& s) { hipFree(s.p1); hipFree(s.p2); hipFree(s.p3); }

// six products, small terms first
template <class F>
static void six(F&& g, const Split& a, const Split& b, float beta0) {
  g(a.p1, b.p3, beta0); g(a.p3, b.p1, 1.f); g(a.p2, b.p2, 1.f); g(a.p1, b.p2, 1.f); g(a.p2, b.p1, 1.f); g(a.p1, b.p1, 1.f);
}

int main(int argc, char** argv) {
  const int big = argc > 1 ? atoi(argv[1]) : 1;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  rocblas_handle rb;
  BK(rocblas_create_handle(&rb));
  BK(rocblas_set_stream(rb, st));
  Lt lt;
  LK(hipblasLtCreate(&lt.h));
  lt.ws_bytes = 256u << 20;
  CK(hipMalloc(&lt.ws, lt.ws_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double flops, int reps, auto&& fn) {
    fn();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("  %-44s %9.2f ms  %8.1f TFLOP/s (fp32-equivalent)\n", name, ms, flops / ms * 1e-9);
    fflush(stdout);
  };

  // ---- accuracy: C = A^T B with a long inner dimension, against double
  float* alphas;
  CK(hipMalloc(&alphas, 64));
  for (int kind = 0; kind < 3; ++kind) {
    const int m = 768, n = 640, k = 55808;
    float *A, *B, *C;
    double *Ad, *Bd, *Cd;
    CK(hipMalloc(&A, (long)k * m * 4)); CK(hipMalloc(&B, (long)k * n * 4)); CK(hipMalloc(&C, (long)m * n * 4));
    CK(hipMalloc(&Ad, (long)k * m * 8)); CK(hipMalloc(&Bd, (long)k * n * 8)); CK(hipMalloc(&Cd, (long)m * n * 8));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((long)k * m + 255) / 256)), dim3(256), 0, st, A, (long)k * m, 1u, 1.f);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((long)k * n + 255) / 256)), dim3(256), 0, st, B, (long)k * n, 2u, 1.f);
    if (kind == 1) {   // all terms positive (Gram-like sums: rounding errors of one chain do not average out)
      hipLaunchKernelGGL(abs_kernel, dim3((unsigned)(((long)k * m + 255) / 256)), dim3(256), 0, st, A, (long)k * m);
      hipLaunchKernelGGL(abs_kernel, dim3((unsigned)(((long)k * n + 255) / 256)), dim3(256), 0, st, B, (long)k * n);
    }
    if (kind == 2) {   // columns of A and B spread over six decades
      hipLaunchKernelGGL(rowscale_kernel, dim3((unsigned)(((long)k * m + 255) / 256)), dim3(256), 0, st, A, (long)k, (long)m);
      hipLaunchKernelGGL(rowscale_kernel, dim3((unsigned)(((long)k * n + 255) / 256)), dim3(256), 0, st, B, (long)k, (long)n);
    }
    hipLaunchKernelGGL(f2d_kernel, dim3((unsigned)(((long)k * m + 255) / 256)), dim3(256), 0, st, A, Ad, (long)k * m);
    hipLaunchKernelGGL(f2d_kernel, dim3((unsigned)(((long)k * n + 255) / 256)), dim3(256), 0, st, B, Bd, (long)k * n);
    rb_dgemm(rb, 1, 0, m, n, k, Ad, m, Bd, n, Cd, n);
    std::vector<double> ref((long)m * n);
    std::vector<float> got((long)m * n);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(ref.data(), Cd, ref.size() * 8, hipMemcpyDeviceToHost));
    auto report = [&](const char* name) {
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(got.data(), C, got.size() * 4, hipMemcpyDeviceToHost));
      double num = 0, den = 0, mx = 0, mrel = 0;
      for (size_t i = 0; i < ref.size(); ++i) {
        double d = got[i] - ref[i];
        num += d * d; den += ref[i] * ref[i];
        if (fabs(d) > mx) mx = fabs(d);
        if (kind && fabs(d) / fabs(ref[i]) > mrel) mrel = fabs(d) / fabs(ref[i]);
      }
      printf("  %-44s rel. Frobenius error %.3e   max abs %.3e   max entrywise rel %.3e\n", name, sqrt(num / den), mx, mrel);
    };
    printf("accuracy, C = A^T B, %d x %d x %d, %s\n", m, n, k,
           kind == 0 ? "standard normal entries" : kind == 1 ? "|normal| entries (all terms positive)" : "normal entries, columns scaled over six decades");
    rb_sgemm(rb, 1, 0, m, n, k, A, m, B, n, 0.f, C, n);
    report("rocBLAS sgemm, one chain");
    for (int k0 = 0; k0 < k; k0 += 1024) {
      int kk = k - k0 < 1024 ? k - k0 : 1024;
      rb_sgemm(rb, 1, 0, m, n, kk, A + (long)k0 * m, m, B + (long)k0 * n, n, k0 ? 1.f : 0.f, C, n);
    }
    report("rocBLAS sgemm, chunks of 1024");
    Split a = split(A, (long)k * m, st), b = split(B, (long)k * n, st);
    six([&](const void* x, const void* y, float beta) { rb_gemm_bf16(rb, 1, 0, m, n, k, x, m, y, n, beta, C, n); }, a, b, 0.f);
    report("rocBLAS gemm_ex bf16x3 (6 products)");
    int ok = 1;
    six([&](const void* x, const void* y, float beta) { ok &= lt_gemm(lt, st, 1, 0, m, n, k, x, m, y, n, beta, C, n); }, a, b, 0.f);
    if (ok) report("hipBLASLt bf16x3 (6 products)");
    else printf("  hipBLASLt: no algorithm\n");
    for (int k0 = 0; k0 < k; k0 += 4096) {
      int kk = k - k0 < 4096 ? k - k0 : 4096;
      six([&](const void* x, const void* y, float beta) {
        lt_gemm(lt, st, 1, 0, m, n, kk, (const unsigned short*)x + (long)k0 * m, m, (const unsigned short*)y + (long)k0 * n, n, beta, C, n);
      }, a, b, k0 ? 1.f : 0.f);
    }
    report("hipBLASLt bf16x3, chunks of 4096");
    {
      SplitH ah = splith(A, (long)k * m, st), bh = splith(B, (long)k * n, st);
      g_devmode = 1;
      int ok3 = three(lt, st, alphas, 1, 0, m, n, k, ah, 0, m, bh, 0, n, 0, C, n);
      if (ok3) report("hipBLASLt fp16x2 (3 products, device scalars)");
      else printf("  hipBLASLt fp16x2, device scalars: failed\n");
      g_devmode = 0;
      ok3 = three(lt, st, alphas, 1, 0, m, n, k, ah, 0, m, bh, 0, n, 0, C, n);
      unscale(st, C, (long)m * n, ah, bh);
      if (ok3) report("hipBLASLt fp16x2 (3 products + unscale)");
      else printf("  hipBLASLt fp16x2: failed\n");
      for (int k0 = 0; k0 < k; k0 += 8192) {
        int kk = k - k0 < 8192 ? k - k0 : 8192;
        three(lt, st, alphas, 1, 0, m, n, kk, ah, (long)k0 * m, m, bh, (long)k0 * n, n, k0 ? 1 : 0, C, n);
      }
      unscale(st, C, (long)m * n, ah, bh);
      report("hipBLASLt fp16x2, chunks of 8192");
      hipFree(ah.h1); hipFree(ah.h2); hipFree(bh.h1); hipFree(bh.h2);
    }
    // three products only (bf16x2-like)
    rb_gemm_bf16(rb, 1, 0, m, n, k, a.p1, m, b.p2, n, 0.f, C, n);
    rb_gemm_bf16(rb, 1, 0, m, n, k, a.p2, m, b.p1, n, 1.f, C, n);
    rb_gemm_bf16(rb, 1, 0, m, n, k, a.p1, m, b.p1, n, 1.f, C, n);
    report("bf16x2 (3 products)");
    drop(a); drop(b);
    hipFree(A); hipFree(B); hipFree(C); hipFree(Ad); hipFree(Bd); hipFree(Cd);
  }
  if (!big) return 0;

  // ---- rates on the shapes of the headline step (R = 55 808 rows, m = T = 10 000)
  const int R = 55808, m = 10000, T = 10000;
  float *M, *Z, *C;
  CK(hipMalloc(&M, (long)R * m * 4)); CK(hipMalloc(&Z, (long)R * T * 4)); CK(hipMalloc(&C, (long)R * m * 4));
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((long)R * m + 255) / 256)), dim3(256), 0, st, M, (long)R * m, 3u, 1.f);
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((long)R * T + 255) / 256)), dim3(256), 0, st, Z, (long)R * T, 4u, 1.f);
  Split ms = split(M, (long)R * m, st), zs = split(Z, (long)R * T, st);
  CK(hipStreamSynchronize(st));
  {
    hipEvent_t s0, s1;
    CK(hipEventCreate(&s0)); CK(hipEventCreate(&s1));
    CK(hipEventRecord(s0, st));
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)(((long)R * m + 255) / 256)), dim3(256), 0, st, M, (long)R * m, ms.p1, ms.p2, ms.p3);
    CK(hipEventRecord(s1, st));
    CK(hipEventSynchronize(s1));
    float t;
    CK(hipEventElapsedTime(&t, s0, s1));
    printf("split of a %d x %d operand: %.2f ms\n", R, m, t);
  }
  SplitH mh = splith(M, (long)R * m, st), zh = splith(Z, (long)R * T, st);
  CK(hipStreamSynchronize(st));
  struct Shape { const char* name; int tA, tB, mm, nn, kk; const float *A, *B; long lda, ldb; const Split *as, *bs; };
  // (a) C = M^T GM (A^T B, inner R), (b) W1 = Mt Z with Mt = m x R row-major (plain, inner R), (c) R_out = M X1 (plain, inner m)
  Shape shapes[3] = {
      {"(a) m x m = (R x m)^T (R x m)", 1, 0, m, m, R, M, Z, m, T, &ms, &zs},
      {"(b) m x T = (m x R) (R x T)", 0, 0, m, T, R, M, Z, R, T, &ms, &zs},
      {"(c) R x m = (R x m) (m x m)", 0, 0, R, m, m, M, Z, m, m, &ms, &zs},
  };
  for (auto& s : shapes) {
    printf("%s\n", s.name);
    const double fl = 2.0 * s.mm * s.nn * s.kk;
    timeit("rocBLAS sgemm", fl, 2, [&] { rb_sgemm(rb, s.tA, s.tB, s.mm, s.nn, s.kk, s.A, s.lda, s.B, s.ldb, 0.f, C, s.nn); });
    timeit("rocBLAS gemm_ex bf16 -> f32, ONE product", fl, 2,
           [&] { rb_gemm_bf16(rb, s.tA, s.tB, s.mm, s.nn, s.kk, s.as->p1, s.lda, s.bs->p1, s.ldb, 0.f, C, s.nn); });
    timeit("hipBLASLt bf16 -> f32, ONE product", fl, 2,
           [&] { lt_gemm(lt, st, s.tA, s.tB, s.mm, s.nn, s.kk, s.as->p1, s.lda, s.bs->p1, s.ldb, 0.f, C, s.nn); });
    timeit("hipBLASLt fp16x2, three products", fl, 3, [&] {
      three(lt, st, alphas, s.tA, s.tB, s.mm, s.nn, s.kk, mh, 0, s.lda, zh, 0, s.ldb, 0, C, s.nn);
    });
    timeit("split of both operands + three products", fl, 3, [&] {
      const long na = (long)(s.tA ? s.kk : s.mm) * s.lda, nb = (long)(s.tB ? s.nn : s.kk) * s.ldb;
      CK(hipMemsetAsync(mh.amax, 0, 4, st)); CK(hipMemsetAsync(zh.amax, 0, 4, st));
      hipLaunchKernelGGL(absmax_kernel, dim3(4096), dim3(256), 0, st, s.A, na, mh.amax);
      hipLaunchKernelGGL(expo_kernel, dim3(1), dim3(1), 0, st, mh.amax, mh.scal);
      hipLaunchKernelGGL(split2h_kernel, dim3(8192), dim3(256), 0, st, s.A, na, mh.scal, mh.h1, mh.h2);
      hipLaunchKernelGGL(absmax_kernel, dim3(4096), dim3(256), 0, st, s.B, nb, zh.amax);
      hipLaunchKernelGGL(expo_kernel, dim3(1), dim3(1), 0, st, zh.amax, zh.scal);
      hipLaunchKernelGGL(split2h_kernel, dim3(8192), dim3(256), 0, st, s.B, nb, zh.scal, zh.h1, zh.h2);
      three(lt, st, alphas, s.tA, s.tB, s.mm, s.nn, s.kk, mh, 0, s.lda, zh, 0, s.ldb, 0, C, s.nn);
      unscale(st, C, (long)s.mm * s.nn, mh, zh);
    });
    timeit("rocBLAS gemm_ex bf16x3, six products", fl, 2, [&] {
      six([&](const void* x, const void* y, float beta) { rb_gemm_bf16(rb, s.tA, s.tB, s.mm, s.nn, s.kk, x, s.lda, y, s.ldb, beta, C, s.nn); }, *s.as,
          *s.bs, 0.f);
    });
    timeit("hipBLASLt bf16x3, six products", fl, 2, [&] {
      six([&](const void* x, const void* y, float beta) { lt_gemm(lt, st, s.tA, s.tB, s.mm, s.nn, s.kk, x, s.lda, y, s.ldb, beta, C, s.nn); }, *s.as,
          *s.bs, 0.f);
    });
  }
  return 0;
}
