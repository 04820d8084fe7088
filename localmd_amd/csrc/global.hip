// Global recombination: U^T U, orthogonalising mixing matrix P, V projection helpers and the
// projected SVD (decomposition.py:912-1137, pmd_loader.py:316-346, :392-414).
// Dense fp32 GEMMs go to rocBLAS and the dense symmetric eigenproblem to rocSOLVER (ssyevd);
// everything else is hand-written.  Matrices are row-major; the rocBLAS calls use the usual
// operand swap (C^T = B^T A^T).
#include "pmd_internal.h"
#include <rocsolver/rocsolver.h>
#include <algorithm>
#include <cmath>
#include <vector>

#define RUN(call)                 \
  do {                            \
    int rc__ = (call);            \
    if (rc__ != PMD_OK) return rc__; \
  } while (0)

#define PMD_BLAS(ctx, call)                                                     \
  do {                                                                          \
    rocblas_status s__ = (call);                                                \
    if (s__ != rocblas_status_success) return pmd_fail(ctx, PMD_ERR_BLAS, #call, rocblas_status_to_string(s__)); \
  } while (0)

// row-major C(m x n) = alpha * op(A) * op(B) + beta * C
// ---------------------------------------------------------------- long inner dimension ------
// A product with few output tiles and a very long inner dimension (M^T G M and M^T Z when R = 3e5 tile components meet
// 1e3 frames: 1000 x 1000 outputs, k = 3e5) leaves most CUs idle in rocBLAS' sgemm (measured 20 TFLOP/s on the many-tile
// workloads).  The inner dimension is cut into S slices that run as ONE strided-batched sgemm (S x the workgroups) into
// S partial outputs, which a second kernel sums in a fixed order (reproducible).
__global__ void sum_partials_kernel(const float* __restrict__ part, long mn, int n, int S, float beta, float* __restrict__ C, long ldc) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < mn; i += (long)gridDim.x * blockDim.x) {
    const long r = i / n;
    const int c = (int)(i - r * n);
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += part[(long)s * mn + i];
    float* o = C + r * ldc + c;
    *o = (beta == 0.f) ? acc : acc + beta * *o;
  }
}

// Length of one fp32 accumulation chain in the library's GEMM-shaped products (PMD_GEMM_KCHUNK overrides; 0 = whole k):
// 1024 up to k = 16384, 2048 beyond (the output is re-read once per chunk: 12 % / 6 % of the product's time).
int pmd_gemm_k_chunk(int k) {
  static int forced = -2;
  if (forced == -2) { const char* e = getenv("PMD_GEMM_KCHUNK"); forced = e ? atoi(e) : -1; }
  if (forced == 0) return k;
  if (forced > 0) return forced;
  if (k <= 3072) return k;
  return k <= 16384 ? 1024 : 2048;
}

bool pmd_is_host_pointer(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return attr.type == hipMemoryTypeHost;
}

static int gemm_rm_splitk(pmd_ctx* ctx, int transA, int transB, int m, int n, int k, int S, float alpha, const float* A, long lda,
                          const float* B, long ldb, float beta, float* C, long ldc) {
  pmd_prof_scope prof__(ctx, "rocblas_sgemm_splitk");
  const int kc = (k + S - 1) / S;
  const int full = k / kc, rem = k - full * kc;   // `full` slices of kc, one more of `rem`
  const int slices = full + (rem ? 1 : 0);
  const long mn = (long)m * n;
  const size_t need = (size_t)slices * mn * sizeof(float) + 4096;
  void* scratch = nullptr;
  RUN(pmd_split_scratch(ctx, need, &scratch));
  if (!scratch) return pmd_fail(ctx, PMD_ERR_HIP, "gemm_rm_splitk", "out of device memory for the partial sums");
  float* part = (float*)scratch;
  const float zero = 0.f;
  const rocblas_operation opA = transA ? rocblas_operation_transpose : rocblas_operation_none;
  const rocblas_operation opB = transB ? rocblas_operation_transpose : rocblas_operation_none;
  // slice s covers inner indices [s kc, (s + 1) kc): A advances by kc columns (or rows when transposed), B by kc rows (columns)
  const long strideA = transA ? (long)kc * lda : kc, strideB = transB ? kc : (long)kc * ldb;
  PMD_BLAS(ctx, rocblas_sgemm_strided_batched(ctx->blas, opB, opA, n, m, kc, &alpha, B, (rocblas_int)ldb, strideB, A, (rocblas_int)lda,
                                              strideA, &zero, part, n, mn, full));
  if (rem)
    PMD_BLAS(ctx, rocblas_sgemm(ctx->blas, opB, opA, n, m, rem, &alpha, B + (long)full * strideB, (rocblas_int)ldb,
                                A + (long)full * strideA, (rocblas_int)lda, &zero, part + (long)full * mn, n));
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)std::min<long>((mn + 255) / 256, 4096)), dim3(256), 0, ctx->stream, part, mn, n,
                     slices, beta, C, ldc);
  PMD_LAUNCH_CHECK(ctx, "sum_partials_kernel");
  return PMD_OK;
}

int pmd_gemm_rm(pmd_ctx* ctx, int transA, int transB, int m, int n, int k, float alpha, const float* A, long lda,
                const float* B, long ldb, float beta, float* C, long ldc) {
  if (m <= 0 || n <= 0) return PMD_OK;
  if (k > 0 && pmd_f16x2_wanted(ctx, m, n, k) && !pmd_is_host_pointer(C)) {
    // large products: three fp16-piece products on the fp16 matrix cores, error below the fp32 path's (gemm_f16x2.hip);
    // done = 0 (Inf / NaN / all-zero operands, no kernel for the shape): C is untouched and the fp32 path below runs
    int done = 0;
    RUN(pmd_gemm_f16x2(ctx, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, &done));
    if (done) return PMD_OK;
  }
  {
    // fewer than one 128 x 128 output tile per CU and an inner dimension that leaves >= 4096 per slice
    static int mode = -1;   // PMD_GEMM_SPLITK=0 switches the path off (A/B runs)
    if (mode < 0) { const char* e = getenv("PMD_GEMM_SPLITK"); mode = (e && !strcmp(e, "0")) ? 0 : 1; }
    const long tiles = (long)((m + 127) / 128) * ((n + 127) / 128);
    if (mode && tiles < 256 && k >= 16384) {
      int S = (int)std::min<long>(std::min<long>(64, k / 4096), (512 + tiles - 1) / tiles);
      if (S >= 2) return gemm_rm_splitk(ctx, transA, transB, m, n, k, S, alpha, A, lda, B, ldb, beta, C, ldc);
    }
  }
  pmd_prof_scope prof__(ctx, "rocblas_sgemm");
  const int kc = pmd_gemm_k_chunk(k);
  if (kc < k && !pmd_is_host_pointer(C)) {
    // Long inner dimension: rocBLAS carries ONE fp32 accumulation chain over all of k (measured on MI355X, scripts/
    // gemm_bias_probe.py: entry errors of 4.9e-6 of the diagonal at k = 10^4, against 3.8e-7 when the chain is cut every
    // 1024 terms - what a CPU BLAS does through its k blocking, and what the NumPy oracle therefore sees).  On the Gram
    // matrix of a 2000 x 10^4 trace matrix those entry errors add up to 4e-5 on the leading singular values and 6e-4 on
    // right singular vectors with 2 % gaps.  Chunks of kc terms accumulated into C (one rounding per chunk).
    const rocblas_operation opA = transA ? rocblas_operation_transpose : rocblas_operation_none;
    const rocblas_operation opB = transB ? rocblas_operation_transpose : rocblas_operation_none;
    const float one = 1.f;
    for (int k0 = 0; k0 < k; k0 += kc) {
      const int kk = std::min(kc, k - k0);
      const float* a = A + (transA ? (long)k0 * lda : (long)k0);
      const float* b = B + (transB ? (long)k0 : (long)k0 * ldb);
      PMD_BLAS(ctx, rocblas_sgemm(ctx->blas, opB, opA, n, m, kk, &alpha, b, (rocblas_int)ldb, a, (rocblas_int)lda, k0 == 0 ? &beta : &one, C,
                                  (rocblas_int)ldc));
    }
    return PMD_OK;
  }
  PMD_BLAS(ctx, rocblas_sgemm(ctx->blas, transB ? rocblas_operation_transpose : rocblas_operation_none,
                              transA ? rocblas_operation_transpose : rocblas_operation_none, n, m, k, &alpha, B,
                              (rocblas_int)ldb, A, (rocblas_int)lda, &beta, C, (rocblas_int)ldc));
  return PMD_OK;
}

// ---------------------------------------------------------------- weighted tile bases ------
// Uw[tile][c][q] = Ut[tile][c][q] * w[q] / cumw[pix[tile][q]] for c < ranks[tile], else 0
// (decomposition.py:812-816, :847-853).
__global__ void weight_tiles_kernel(const float* __restrict__ Ut, long tile_stride, int ld, const int* __restrict__ pix,
                                    int d, const float* __restrict__ w, const float* __restrict__ cumw,
                                    const int* __restrict__ ranks, float* __restrict__ Uw) {
  const int tile = blockIdx.y;
  const int rk = ranks[tile];
  // every one of the 64 x ld entries is written: the padding feeds MFMA operands (0 * x must be 0)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < PMD_RPAD * ld; i += gridDim.x * blockDim.x) {
    const int c = i / ld, q = i - c * ld;
    float v = 0.f;
    if (c < rk && q < d) v = Ut[(long)tile * tile_stride + (long)c * ld + q] * w[q] / cumw[pix[(long)tile * d + q]];
    Uw[(long)tile * tile_stride + (long)c * ld + q] = v;
  }
}

int pmd_launch_weight_tiles(pmd_ctx* ctx, const float* Ut, int dpad, const int* pix, int d, const float* w,
                            const float* cumw, const int* ranks, float* Uw, int n_tiles) {
  pmd_prof_scope prof__(ctx, "weight_tiles");
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    hipLaunchKernelGGL(weight_tiles_kernel, dim3(32, tn), dim3(256), 0, ctx->stream, Ut + (long)t0 * 64 * dpad,
                       64L * dpad, dpad, pix + (long)t0 * d, d, w, cumw, ranks + t0, Uw + (long)t0 * 64 * dpad);
    PMD_LAUNCH_CHECK(ctx, "weight_tiles_kernel");
  }
  return PMD_OK;
}

// Z[col_off[tile] + c][t] = Out[tile][c][t], c < ranks[tile]
__global__ void compact_rows_kernel(const float* __restrict__ Out, long tile_stride, long ldo,
                                    const int* __restrict__ col_off, const int* __restrict__ ranks, int T,
                                    float* __restrict__ Z, long ldz) {
  const int tile = blockIdx.y;
  const int rk = ranks[tile];
  const long off = col_off[tile];
  for (int c = 0; c < rk; ++c)
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x)
      Z[(off + c) * ldz + t] = Out[(long)tile * tile_stride + (long)c * ldo + t];
}

int pmd_launch_compact_rows(pmd_ctx* ctx, const float* Out, long tile_stride, long ldo, const int* col_off,
                            const int* ranks, int T, float* Z, long ldz, int n_tiles) {
  pmd_prof_scope prof__(ctx, "compact_rows");
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    hipLaunchKernelGGL(compact_rows_kernel, dim3(16, tn), dim3(256), 0, ctx->stream, Out + (long)t0 * tile_stride,
                       tile_stride, ldo, col_off + t0, ranks + t0, T, Z, ldz);
    PMD_LAUNCH_CHECK(ctx, "compact_rows_kernel");
  }
  return PMD_OK;
}

// ---------------------------------------------------------------- G = U^T U ----------------
// pairs[p] = (tile a, tile b, i0, i1, j0, j1): overlap rectangle in FOV coordinates.
// origins[tile] = (k, j).  Writes G[off_a + c][off_b + c'] and its transpose.
__global__ __launch_bounds__(256) void gram_pairs_kernel(const float* __restrict__ Uw, int dpad, int b1,
                                                         const int* __restrict__ pairs, const int* __restrict__ origins,
                                                         const int* __restrict__ col_off, const int* __restrict__ ranks,
                                                         float* __restrict__ G, long ldg) {
  __shared__ float ua[64][65];
  __shared__ float ub[64][65];
  const int* pr = pairs + (long)blockIdx.x * 6;
  const int ta = pr[0], tb = pr[1], i0 = pr[2], i1 = pr[3], j0 = pr[4], j1 = pr[5];
  const int ra = ranks[ta], rb = ranks[tb];
  if (ra == 0 || rb == 0) return;
  const int ka = origins[2 * ta], ja = origins[2 * ta + 1], kb = origins[2 * tb], jb = origins[2 * tb + 1];
  const int h = i1 - i0, npix = h * (j1 - j0);
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  const float* pa = Uw + (long)ta * 64 * dpad;
  const float* pb = Uw + (long)tb * 64 * dpad;
  for (int p0 = 0; p0 < npix; p0 += 64) {
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
      const int c = i >> 6, pl = i & 63;
      const int p = p0 + pl;
      float va = 0.f, vb = 0.f;
      if (p < npix) {
        const int jj = p / h, ii = p - jj * h;
        const int gi = i0 + ii, gj = j0 + jj;
        if (c < ra) va = pa[(long)c * dpad + (gi - ka) + b1 * (gj - ja)];
        if (c < rb) vb = pb[(long)c * dpad + (gi - kb) + b1 * (gj - jb)];
      }
      ua[c][pl] = va;
      ub[c][pl] = vb;
    }
    __syncthreads();
#pragma unroll 4
    for (int pl = 0; pl < 64; ++pl) {
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) { av[a] = (double)ua[4 * ti + a][pl]; bv[a] = (double)ub[4 * tj + a][pl]; }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
    }
    __syncthreads();
  }
  const long oa = col_off[ta], ob = col_off[tb];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int c = 4 * ti + a, cp = 4 * tj + b;
      if (c < ra && cp < rb) {
        const float v = (float)acc[a][b];
        G[(oa + c) * ldg + ob + cp] = v;
        G[(ob + cp) * ldg + oa + c] = v;
      }
    }
}

// tile x background block: G[off + c][Rt + k] = sum_q Uw[tile][c][q] * basis[pix[q]][k]
__global__ __launch_bounds__(256) void gram_bg_kernel(const float* __restrict__ Uw, int dpad, int d,
                                                      const int* __restrict__ pix, const float* __restrict__ basis,
                                                      int K, const int* __restrict__ col_off,
                                                      const int* __restrict__ ranks, int Rt, float* __restrict__ G,
                                                      long ldg) {
  const int tile = blockIdx.x;
  const int rk = ranks[tile];
  const long off = col_off[tile];
  for (int i = threadIdx.x; i < rk * K; i += 256) {
    const int c = i / K, k = i - c * K;
    double s = 0.0;
    for (int q = 0; q < d; ++q)
      s += (double)Uw[(long)tile * 64 * dpad + (long)c * dpad + q] * (double)basis[(long)pix[(long)tile * d + q] * K + k];
    G[(off + c) * ldg + Rt + k] = (float)s;
    G[(long)(Rt + k) * ldg + off + c] = (float)s;
  }
}

// background x background: one workgroup per entry
__global__ __launch_bounds__(256) void gram_bgbg_kernel(const float* __restrict__ basis, long D, int K, int Rt,
                                                        float* __restrict__ G, long ldg) {
  __shared__ double red[256];
  const int k1 = blockIdx.x, k2 = blockIdx.y;
  if (k2 < k1) return;
  double s = 0.0;
  for (long c = threadIdx.x; c < D; c += 256) s += (double)basis[c * K + k1] * (double)basis[c * K + k2];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    G[(long)(Rt + k1) * ldg + Rt + k2] = (float)red[0];
    G[(long)(Rt + k2) * ldg + Rt + k1] = (float)red[0];
  }
}

int pmd_gram_u_impl(pmd_ctx* ctx, const float* Uw, int dpad, int b1, int b2, const int* pix, const int* pairs,
                    int n_pairs, const int* origins, const int* col_off, const int* ranks, int n_tiles, int Rt,
                    const float* basis, long D, int K, float* G, long ldg) {
  pmd_prof_scope prof__(ctx, "gram_u");
  const long R = Rt + K;
  PMD_HIP(ctx, hipMemsetAsync(G, 0, (size_t)R * ldg * sizeof(float), ctx->stream));
  if (n_pairs > 0) {
    hipLaunchKernelGGL(gram_pairs_kernel, dim3(n_pairs), dim3(256), 0, ctx->stream, Uw, dpad, b1, pairs, origins,
                       col_off, ranks, G, ldg);
    PMD_LAUNCH_CHECK(ctx, "gram_pairs_kernel");
  }
  if (K > 0) {
    hipLaunchKernelGGL(gram_bg_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, Uw, dpad, b1 * b2, pix, basis, K,
                       col_off, ranks, Rt, G, ldg);
    PMD_LAUNCH_CHECK(ctx, "gram_bg_kernel");
    hipLaunchKernelGGL(gram_bgbg_kernel, dim3(K, K), dim3(256), 0, ctx->stream, basis, D, K, Rt, G, ldg);
    PMD_LAUNCH_CHECK(ctx, "gram_bgbg_kernel");
  }
  return PMD_OK;
}

// ---------------------------------------------------------------- helpers for eigen output -
// dst[c][x] = src[perm[c]][x] * scale[c]   (row gather of the eigenvector matrix)
__global__ void gather_rows_scale_kernel(const float* __restrict__ src, long lds_, const int* __restrict__ perm,
                                         const float* __restrict__ scale, int ncols, float* __restrict__ dst, long ldd) {
  const int c = blockIdx.y;
  const float s = scale[c];
  const long r = perm[c];
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < ncols; x += gridDim.x * blockDim.x)
    dst[(long)c * ldd + x] = src[r * lds_ + x] * s;
}

__global__ void transpose_kernel(const float* __restrict__ src, long lds_, int rows, int cols, float* __restrict__ dst,
                                 long ldd) {
  __shared__ float t[32][33];
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8)
    if (y0 + r < rows && x0 + tx < cols) t[r][tx] = src[(long)(y0 + r) * lds_ + x0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (x0 + r < cols && y0 + tx < rows) dst[(long)(x0 + r) * ldd + y0 + tx] = t[tx][r];
}

__global__ void scale_rows2_kernel(float* __restrict__ x, long ld, int ncols, const float* __restrict__ scale) {
  const float s = scale[blockIdx.y];
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += gridDim.x * blockDim.x) x[(long)blockIdx.y * ld + c] *= s;
}

__global__ void scale_cols_kernel(float* __restrict__ x, long ld, int ncols, const float* __restrict__ scale) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += gridDim.x * blockDim.x) x[(long)blockIdx.y * ld + c] *= scale[c];
}

static int launch_gather_rows(pmd_ctx* ctx, const float* src, long lds_, const int* perm, const float* scale, int nrows,
                              int ncols, float* dst, long ldd) {
  for (int r0 = 0; r0 < nrows; r0 += 32768) {
    const int rn = (nrows - r0 < 32768) ? nrows - r0 : 32768;
    hipLaunchKernelGGL(gather_rows_scale_kernel, dim3(8, rn), dim3(256), 0, ctx->stream, src, lds_, perm + r0,
                       scale + r0, ncols, dst + (long)r0 * ldd, ldd);
    PMD_LAUNCH_CHECK(ctx, "gather_rows_scale_kernel");
  }
  return PMD_OK;
}

static int launch_transpose(pmd_ctx* ctx, const float* src, long lds_, int rows, int cols, float* dst, long ldd) {
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, ctx->stream, src, lds_,
                     rows, cols, dst, ldd);
  PMD_LAUNCH_CHECK(ctx, "transpose_kernel");
  return PMD_OK;
}

// ---------------------------------------------------------------- A15 ----------------------
// compute_lowrank_factorized_svd(only_left=True), decomposition.py:974-999.
// G: R x R (overwritten).  M: R x m right matrix (row-major, ldm) or NULL for the identity
// (reference: right_mat = v if R > v.shape[1] else eye(R)).  P_out: R x R' (ldp >= m).
// Host sync: the eigenvalues come back to count R' (reference: good_components = eig_vals > 0).
size_t pmd_orthogonalize_workspace_bytes_impl(int R, int m, int has_m) {
  size_t b = 0;
  b += (size_t)m * sizeof(float) * 4 + (size_t)m * sizeof(int) + 4096;  // w, work, scale, perm, info
  if (has_m) b += (size_t)R * m * sizeof(float) + (size_t)m * m * sizeof(float);  // GM, C
  b += (size_t)m * m * sizeof(float);                                             // Et
  return b + 8192;
}

int pmd_orthogonalize_impl(pmd_ctx* ctx, float* G, int R, const float* M, int m, long ldm, float* P_out, long ldp,
                           int* rprime_out, void* ws, size_t ws_bytes) {
  pmd_arena ar(ws, ws_bytes);
  float* w = ar.take_n<float>(m);
  float* work = ar.take_n<float>(m);
  float* scale = ar.take_n<float>(m);
  int* perm = ar.take_n<int>(m);
  int* info = ar.take_n<int>(4);
  float* C = nullptr;
  float* GM = nullptr;
  if (M) {
    GM = ar.take_n<float>((size_t)R * m);
    C = ar.take_n<float>((size_t)m * m);
  } else {
    if (m != R) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_orthogonalize", "identity right matrix needs m == R");
    C = G;
  }
  float* Et = ar.take_n<float>((size_t)m * m);
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_orthogonalize", "workspace too small");
  long ldc = M ? m : R;
  if (M) {
    RUN(pmd_gemm_rm(ctx, 0, 0, R, m, R, 1.f, G, R, M, ldm, 0.f, GM, m));
    RUN(pmd_gemm_rm(ctx, 1, 0, m, m, R, 1.f, M, ldm, GM, m, 0.f, C, m));
  }
  RUN(pmd_syevd(ctx, m, C, ldc, w, work, info));
  std::vector<float> hw(m);
  int hinfo = 0;
  PMD_HIP(ctx, hipMemcpyAsync(hw.data(), w, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  PMD_HIP(ctx, hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (hinfo != 0) return pmd_fail(ctx, PMD_ERR_BLAS, "rocsolver_ssyevd", "did not converge");
  // jnp.linalg.svd(hermitian=True): sort by |lambda| descending
  std::vector<int> idx(m);
  for (int i = 0; i < m; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return std::fabs(hw[a]) > std::fabs(hw[b]); });
  std::vector<int> hperm;
  std::vector<float> hscale;
  // jnp.linalg.svd(hermitian=True) returns |lambda| and u = v sign(lambda): `eig_vals > 0` (decomposition.py:988)
  // keeps every direction whose eigenvalue is not exactly zero.  null_cutoff >= 0 (pmd_ctx_set_null_cutoff):
  // only lambda > cutoff * lambda_max.
  const float lmax = m > 0 ? std::fabs(hw[idx[0]]) : 0.f;
  for (int i = 0; i < m; ++i) {
    const float lam = hw[idx[i]];
    const bool keep = ctx->null_cutoff < 0.f ? (lam != 0.f && lam == lam) : (lam > ctx->null_cutoff * lmax);
    if (keep) {
      hperm.push_back(idx[i]);
      hscale.push_back((lam < 0.f ? -1.0f : 1.0f) / std::sqrt(std::fabs(lam)));
    }
  }
  const int rp = (int)hperm.size();
  *rprime_out = rp;
  if (rp == 0) return PMD_OK;
  PMD_HIP(ctx, hipMemcpyAsync(perm, hperm.data(), (size_t)rp * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  PMD_HIP(ctx, hipMemcpyAsync(scale, hscale.data(), (size_t)rp * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  RUN(launch_gather_rows(ctx, C, ldc, perm, scale, rp, m, Et, m));
  if (M) RUN(pmd_gemm_rm(ctx, 0, 1, R, rp, m, 1.f, M, ldm, Et, m, 0.f, P_out, ldp));
  else RUN(launch_transpose(ctx, Et, m, rp, m, P_out, ldp));
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));  // hperm/hscale are host temporaries
  return PMD_OK;
}

// ---------------------------------------------------------------- A17 ----------------------
// projected_svd (decomposition.py:1042-1060) with fewer_rows (:1089-1099) / fewer_columns
// (:1128-1137).  V: n1 x n2 (row-major, overwritten when n1 <= n2 is false? no: preserved).
// Outputs: R_out (rows_p x nk), s_out (nk), Vt_out (nk x n2), nk = min(n1, n2).
size_t pmd_projected_svd_workspace_bytes_impl(int rows_p, int n1, int n2) {
  const size_t nk = (size_t)std::min(n1, n2);
  size_t b = (nk + 4) * nk * sizeof(float) * 2 + nk * (sizeof(float) * 5 + sizeof(int)) + 8192;
  if (n1 > n2) b += (size_t)n1 * n2 * sizeof(float);
  return b + 8192;
}

int pmd_projected_svd_impl(pmd_ctx* ctx, const float* P, int rows_p, long ldp, const float* V, int n1, int n2, long ldv,
                           float* R_out, long ldr, float* s_out, float* Vt_out, long ldvt, void* ws, size_t ws_bytes) {
  const int nk = std::min(n1, n2);
  pmd_arena ar(ws, ws_bytes);
  const long ldc = pmd_round_up(nk, 4);  // the library's own tridiagonalisation wants 16-byte aligned rows
  float* C = ar.take_n<float>((size_t)ldc * nk);
  float* Wt = ar.take_n<float>((size_t)nk * nk);
  float* w = ar.take_n<float>(nk);
  float* work = ar.take_n<float>(nk);
  float* sgn = ar.take_n<float>(nk);
  float* inv = ar.take_n<float>(nk);
  int* perm = ar.take_n<int>(nk);
  int* info = ar.take_n<int>(4);
  float* left = (n1 > n2) ? ar.take_n<float>((size_t)n1 * n2) : nullptr;
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_projected_svd", "workspace too small");
  {
    // Gram matrix: only the row-major upper triangle (= column-major lower), the one pmd_syevd reads
    pmd_prof_scope prof__(ctx, "rocblas_ssyrk");
    const float one = 1.f, zero = 0.f;
    // (inner dimension in chunks: one fp32 accumulation chain per chunk, see pmd_gemm_rm)
    const int kfull = (n1 <= n2) ? n2 : n1;
    const int kc = pmd_gemm_k_chunk(kfull);
    for (int k0 = 0; k0 < kfull; k0 += kc) {
      const int kk = std::min(kc, kfull - k0);
      if (n1 <= n2)  // V V^T
        PMD_BLAS(ctx, rocblas_ssyrk(ctx->blas, rocblas_fill_lower, rocblas_operation_transpose, n1, kk, &one, V + k0, (rocblas_int)ldv,
                                    k0 == 0 ? &zero : &one, C, (rocblas_int)ldc));
      else           // V^T V
        PMD_BLAS(ctx, rocblas_ssyrk(ctx->blas, rocblas_fill_lower, rocblas_operation_none, n2, kk, &one, V + (long)k0 * ldv, (rocblas_int)ldv,
                                    k0 == 0 ? &zero : &one, C, (rocblas_int)ldc));
    }
  }
  RUN(pmd_syevd(ctx, nk, C, ldc, w, work, info));
  std::vector<float> hw(nk);
  int hinfo = 0;
  PMD_HIP(ctx, hipMemcpyAsync(hw.data(), w, (size_t)nk * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  PMD_HIP(ctx, hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (hinfo != 0) return pmd_fail(ctx, PMD_ERR_BLAS, "rocsolver_ssyevd", "did not converge");
  std::vector<int> idx(nk);
  for (int i = 0; i < nk; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return std::fabs(hw[a]) > std::fabs(hw[b]); });
  std::vector<float> hs(nk), hsgn(nk), hinv(nk);
  for (int i = 0; i < nk; ++i) {
    const float lam = hw[idx[i]];
    hs[i] = std::sqrt(std::fabs(lam));
    hsgn[i] = (lam < 0.f) ? -1.f : 1.f;
    hinv[i] = (hs[i] == 0.f) ? 1.f : 1.f / hs[i];
  }
  PMD_HIP(ctx, hipMemcpyAsync(perm, idx.data(), (size_t)nk * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  PMD_HIP(ctx, hipMemcpyAsync(sgn, hsgn.data(), (size_t)nk * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  PMD_HIP(ctx, hipMemcpyAsync(inv, hinv.data(), (size_t)nk * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  PMD_HIP(ctx, hipMemcpyAsync(s_out, hs.data(), (size_t)nk * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  // Wt[c][:] = sign_c * eigenvector perm[c]
  RUN(launch_gather_rows(ctx, C, ldc, perm, sgn, nk, nk, Wt, nk));
  if (n1 <= n2) {
    // Vt = (W^T V) / s ; R = P W
    RUN(pmd_gemm_rm(ctx, 0, 0, nk, n2, n1, 1.f, Wt, nk, V, ldv, 0.f, Vt_out, ldvt));
    hipLaunchKernelGGL(scale_rows2_kernel, dim3(8, nk), dim3(256), 0, ctx->stream, Vt_out, ldvt, n2, inv);
    PMD_LAUNCH_CHECK(ctx, "scale_rows2_kernel");
    if (P) RUN(pmd_gemm_rm(ctx, 0, 1, rows_p, nk, n1, 1.f, P, ldp, Wt, nk, 0.f, R_out, ldr));
    else RUN(launch_transpose(ctx, Wt, nk, nk, n1, R_out, ldr));  // no projection: R_out = W (n1 x nk)
  } else {
    // right_t = W; left = V (W / s); R = P left; Vt = W^T
    RUN(pmd_gemm_rm(ctx, 0, 1, n1, nk, n2, 1.f, V, ldv, Wt, nk, 0.f, left, nk));
    hipLaunchKernelGGL(scale_cols_kernel, dim3(8, n1), dim3(256), 0, ctx->stream, left, (long)nk, nk, inv);
    PMD_LAUNCH_CHECK(ctx, "scale_cols_kernel");
    RUN(pmd_gemm_rm(ctx, 0, 0, rows_p, nk, n1, 1.f, P, ldp, left, nk, 0.f, R_out, ldr));
    PMD_HIP(ctx, hipMemcpy2DAsync(Vt_out, ldvt * sizeof(float), Wt, (size_t)nk * sizeof(float), (size_t)nk * sizeof(float), nk, hipMemcpyDeviceToDevice, ctx->stream));
  }
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PMD_OK;
}

// ---------------------------------------------------------------- A17 split by frame columns ------------------
// Multi-GPU form of the factored projected SVD (the frames x frames products of the last stage sharded by frame columns):
//   pmd_psvd_vp_gram : Vp_r = Et W1[:, c0:c1) for this rank's frame columns, C_r = Vp_r Vp_r^T (row-major upper triangle)
//   [the caller sums C_r over the ranks]
//   pmd_psvd_finish  : eigendecomposition of the summed C (replicated), s, W, and this rank's columns of Vt = W^T Vp / s
// One rank with all columns reproduces pmd_projected_svd_factored (same kernels, same order of operations).
static int psvd_gram_rows(pmd_ctx* ctx, const float* V, int n1, int n2, long ldv, float* C, long ldc) {
  pmd_prof_scope prof__(ctx, "rocblas_ssyrk");
  const float one = 1.f, zero = 0.f;
  const int kc = pmd_gemm_k_chunk(n2);
  if (n2 <= 0) { PMD_HIP(ctx, hipMemsetAsync(C, 0, (size_t)n1 * ldc * sizeof(float), ctx->stream)); return PMD_OK; }
  for (int k0 = 0; k0 < n2; k0 += kc) {
    const int kk = std::min(kc, n2 - k0);
    PMD_BLAS(ctx, rocblas_ssyrk(ctx->blas, rocblas_fill_lower, rocblas_operation_transpose, n1, kk, &one, V + k0, (rocblas_int)ldv,
                                k0 == 0 ? &zero : &one, C, (rocblas_int)ldc));
  }
  return PMD_OK;
}

int pmd_psvd_vp_gram_impl(pmd_ctx* ctx, const float* Et, int rp, int m, long lde, const float* W1, int nc, long ldw, int et_lower,
                          float* Vp, long ldv, float* C, long ldc) {
  if (nc > 0) {
    // (a triangular product does half the flops in fp32; the full product from fp16 pieces is faster still where it applies)
    const bool tri = et_lower && rp == m && !pmd_f16x2_wanted(ctx, rp, nc, m);
    const float one = 1.f;
    if (tri) {
      pmd_prof_scope prof__(ctx, "rocblas_strmm");
      PMD_BLAS(ctx, rocblas_strmm(ctx->blas, rocblas_side_right, rocblas_fill_upper, rocblas_operation_none, rocblas_diagonal_non_unit, nc, m,
                                  &one, Et, (rocblas_int)lde, W1, (rocblas_int)ldw, Vp, (rocblas_int)ldv));
    } else {
      RUN(pmd_gemm_rm(ctx, 0, 0, rp, nc, m, 1.f, Et, lde, W1, ldw, 0.f, Vp, ldv));
    }
  }
  return psvd_gram_rows(ctx, Vp, rp, nc, ldv, C, ldc);
}

size_t pmd_psvd_finish_workspace_bytes_impl(int rp) {
  return (size_t)rp * rp * sizeof(float) + (size_t)rp * (sizeof(float) * 5 + sizeof(int)) + 16384;
}

// C: summed Gram matrix (rp x rp, ld ldc >= round_up(rp, 4); overwritten), Vp: this rank's columns (rp x nc).
// Outputs: W_out (rp x rp, row c' = component c' of the left vectors, i.e. W[c'][c]), s_out (rp), Vt_out (rp x nc).
int pmd_psvd_finish_impl(pmd_ctx* ctx, float* C, long ldc, int rp, const float* Vp, int nc, long ldv, float* W_out, long ldw, float* s_out,
                         float* Vt_out, long ldvt, void* ws, size_t ws_bytes) {
  pmd_arena ar(ws, ws_bytes);
  float* Wt = ar.take_n<float>((size_t)rp * rp);
  float* w = ar.take_n<float>(rp);
  float* work = ar.take_n<float>(rp);
  float* sgn = ar.take_n<float>(rp);
  float* inv = ar.take_n<float>(rp);
  int* perm = ar.take_n<int>(rp);
  int* info = ar.take_n<int>(4);
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_psvd_finish", "workspace too small");
  RUN(pmd_syevd(ctx, rp, C, ldc, w, work, info));
  std::vector<float> hw(rp);
  int hinfo = 0;
  PMD_HIP(ctx, hipMemcpyAsync(hw.data(), w, (size_t)rp * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  PMD_HIP(ctx, hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (hinfo != 0) return pmd_fail(ctx, PMD_ERR_BLAS, "rocsolver_ssyevd", "did not converge");
  std::vector<int> idx(rp);
  for (int i = 0; i < rp; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return std::fabs(hw[a]) > std::fabs(hw[b]); });
  std::vector<float> hs(rp), hsgn(rp), hinv(rp);
  for (int i = 0; i < rp; ++i) {
    const float lam = hw[idx[i]];
    hs[i] = std::sqrt(std::fabs(lam));
    hsgn[i] = (lam < 0.f) ? -1.f : 1.f;
    hinv[i] = (hs[i] == 0.f) ? 1.f : 1.f / hs[i];
  }
  PMD_HIP(ctx, hipMemcpyAsync(perm, idx.data(), (size_t)rp * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  PMD_HIP(ctx, hipMemcpyAsync(sgn, hsgn.data(), (size_t)rp * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  PMD_HIP(ctx, hipMemcpyAsync(inv, hinv.data(), (size_t)rp * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  PMD_HIP(ctx, hipMemcpyAsync(s_out, hs.data(), (size_t)rp * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  RUN(launch_gather_rows(ctx, C, ldc, perm, sgn, rp, rp, Wt, rp));      // Wt[c][:] = sign_c * eigenvector perm[c]
  if (nc > 0) {
    RUN(pmd_gemm_rm(ctx, 0, 0, rp, nc, rp, 1.f, Wt, rp, Vp, ldv, 0.f, Vt_out, ldvt));
    hipLaunchKernelGGL(scale_rows2_kernel, dim3(8, rp), dim3(256), 0, ctx->stream, Vt_out, ldvt, nc, inv);
    PMD_LAUNCH_CHECK(ctx, "scale_rows2_kernel");
  }
  RUN(launch_transpose(ctx, Wt, rp, rp, rp, W_out, ldw));
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the host vectors above are temporaries
  return PMD_OK;
}

// ---------------------------------------------------------------- background projection ----
// out[k][t] = sum_c basis[c][k] * xs[c][t]  (pmd_loader.py:386, and the K background rows of
// U^T X in :411).  xs must have round_up(D, 1024) rows allocated (rows >= D zero).
#define BGP_BLK 1024

__global__ void block_basis_kernel(const float* __restrict__ basis, long D, int K, float* __restrict__ At, int kstride) {
  // At[blk][k][q] = basis[blk*BGP_BLK + q][k]   (K <= 64 columns of a basis with kstride columns per pixel)
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long blk = c / BGP_BLK;
  const int q = (int)(c - blk * BGP_BLK);
  for (int k = 0; k < PMD_RPAD; ++k) At[blk * PMD_RPAD * BGP_BLK + (long)k * BGP_BLK + q] = (c < D && k < K) ? basis[c * kstride + k] : 0.f;
}

size_t pmd_bg_project_workspace_bytes_impl(long D, int T) {
  const size_t nblk = (size_t)((D + BGP_BLK - 1) / BGP_BLK);
  return nblk * 64 * BGP_BLK * sizeof(float) + nblk * 64 * (size_t)pmd_time_ld(T) * sizeof(float) + 8192;
}

int pmd_bg_project_impl(pmd_ctx* ctx, const float* xs, long D, int T, long ld, const float* basis, int K, float* out,
                        long ldo, void* ws, size_t ws_bytes) {
  if (K < 1) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_bg_project", "background rank must be >= 1");
  const int nblk = (int)((D + BGP_BLK - 1) / BGP_BLK);
  const long ldt = pmd_time_ld(T);
  pmd_arena ar(ws, ws_bytes);
  float* At = ar.take_n<float>((size_t)nblk * 64 * BGP_BLK);
  float* part = ar.take_n<float>((size_t)nblk * 64 * ldt);
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_bg_project", "workspace too small");
  // 64 basis columns per pass over the movie (ranks above 64: several passes)
  for (int k0 = 0; k0 < K; k0 += 64) {
    const int kc = (K - k0 < 64) ? K - k0 : 64;
    hipLaunchKernelGGL(block_basis_kernel, dim3(nblk * (BGP_BLK / 256)), dim3(256), 0, ctx->stream, basis + k0, D, kc, At, K);
    PMD_LAUNCH_CHECK(ctx, "block_basis_kernel");
    ctx->atx_rows = kc;   // (<= 16 columns - the default rank is 15 - run on one row tile: a quarter of the MFMA work)
    const int rc_atx = pmd_launch_tile_atx(ctx, xs, ld, nullptr, 0, BGP_BLK, BGP_BLK, At, 64L * BGP_BLK, BGP_BLK, part, 64L * ldt, ldt, nblk, T, 8);
    ctx->atx_rows = 0;
    RUN(rc_atx);
    // sum the block partials row by row into out[k][0:T]
    for (int k = 0; k < kc; ++k)
      RUN(pmd_launch_reduce_slices(ctx, part + (long)k * ldt, 0, 64L * ldt, nblk, T, out + (long)(k0 + k) * ldo, 0, 1));
  }
  return PMD_OK;
}

// =============================================================================================
// Block-sparse form of G = U^T U (used when the right matrix is not the identity, i.e. R > frames:
// decomposition.py:976-981).  G is never densified: tile pairs give 64x64 blocks, the background
// columns give one 64x64 block per tile plus a dense K x Rc strip.
// =============================================================================================

// Gblk[p][c][c'] = sum over the overlap of pair p of Uw[a][c][.] * Uw[b][c'][.]   (a <= b)
__global__ __launch_bounds__(256) void gram_pair_blocks_kernel(const float* __restrict__ Uw, int dpad, int b1,
                                                               const int* __restrict__ pairs,
                                                               const int* __restrict__ origins,
                                                               const int* __restrict__ ranks,
                                                               float* __restrict__ Gblk) {
  __shared__ float ua[64][65];
  __shared__ float ub[64][65];
  const int* pr = pairs + (long)blockIdx.x * 6;
  const int ta = pr[0], tb = pr[1], i0 = pr[2], i1 = pr[3], j0 = pr[4], j1 = pr[5];
  const int ra = ranks[ta], rb = ranks[tb];
  const int ka = origins[2 * ta], ja = origins[2 * ta + 1], kb = origins[2 * tb], jb = origins[2 * tb + 1];
  const int h = i1 - i0, npix = h * (j1 - j0);
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  const float* pa = Uw + (long)ta * 64 * dpad;
  const float* pb = Uw + (long)tb * 64 * dpad;
  if (ra > 0 && rb > 0) {
    for (int p0 = 0; p0 < npix; p0 += 64) {
      for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, pl = i & 63;
        const int p = p0 + pl;
        float va = 0.f, vb = 0.f;
        if (p < npix) {
          const int jj = p / h, ii = p - jj * h;
          const int gi = i0 + ii, gj = j0 + jj;
          if (c < ra) va = pa[(long)c * dpad + (gi - ka) + b1 * (gj - ja)];
          if (c < rb) vb = pb[(long)c * dpad + (gi - kb) + b1 * (gj - jb)];
        }
        ua[c][pl] = va;
        ub[c][pl] = vb;
      }
      __syncthreads();
#pragma unroll 4
      for (int pl = 0; pl < 64; ++pl) {
        double av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { av[a] = (double)ua[4 * ti + a][pl]; bv[a] = (double)ub[4 * tj + a][pl]; }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
      }
      __syncthreads();
    }
  }
  float* g = Gblk + (long)blockIdx.x * 4096;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) g[(4 * ti + a) * 64 + 4 * tj + b] = (float)acc[a][b];
}

// Gbg[tile][c][k] = sum_q Uw[tile][c][q] * basis[pix[q]][k]; also scattered into the dense strip
// Gstrip[k][off + c] (K x ldgs).
__global__ __launch_bounds__(256) void gram_bg_blocks_kernel(const float* __restrict__ Uw, int dpad, int d,
                                                             const int* __restrict__ pix,
                                                             const float* __restrict__ basis, int K,
                                                             const int* __restrict__ col_off,
                                                             const int* __restrict__ ranks, float* __restrict__ Gbg,
                                                             float* __restrict__ Gstrip, long ldgs, int kstride) {
  const int tile = blockIdx.x;
  const int rk = ranks[tile];
  const long off = col_off[tile];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, k = i & 63;
    float v = 0.f;
    if (c < rk && k < K) {
      double s = 0.0;
      for (int q = 0; q < d; ++q)
        s += (double)Uw[(long)tile * 64 * dpad + (long)c * dpad + q] * (double)basis[(long)pix[(long)tile * d + q] * kstride + k];
      v = (float)s;
      Gstrip[(long)k * ldgs + off + c] = v;
    }
    Gbg[(long)tile * 4096 + i] = v;
  }
}

__global__ __launch_bounds__(256) void gram_bgbg_strip_kernel(const float* __restrict__ basis, long D, int K, int Rt,
                                                              float* __restrict__ Gstrip, long ldgs) {
  __shared__ double red[256];
  const int k1 = blockIdx.x, k2 = blockIdx.y;
  if (k2 < k1) return;
  double s = 0.0;
  for (long c = threadIdx.x; c < D; c += 256) s += (double)basis[c * K + k1] * (double)basis[c * K + k2];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Gstrip[(long)k1 * ldgs + Rt + k2] = (float)red[0];
    Gstrip[(long)k2 * ldgs + Rt + k1] = (float)red[0];
  }
}

// GM[off_a + c][x] = sum over neighbour blocks (b, c') of G_ab[c][c'] * M[off_b + c'][x]
// nbr_ptr[a] .. nbr_ptr[a+1]: entries (row offset of the block's M rows, rows in the block,
// block index into Gblk/Gbg, flags: bit0 = use the transposed block, bit1 = background block).
template <int NOUT>
__global__ __launch_bounds__(256) void gram_apply_kernel(const float* __restrict__ Gblk, const float* __restrict__ Gbg,
                                                         const int* __restrict__ nbr_ptr, const int* __restrict__ nbr,
                                                         const int* __restrict__ col_off, const int* __restrict__ ranks,
                                                         const float* __restrict__ M, long ldm, int ncols,
                                                         float* __restrict__ GM, long ldgm) {
  __shared__ float g[64][NOUT + 1];  // g[c'][c]
  const int a = blockIdx.y;
  const int ra = ranks[a];
  if (ra == 0) return;
  const int x = blockIdx.x * 256 + threadIdx.x;
  // fp32 accumulation over the few hundred terms of a block row: the product feeds fp32 GEMMs
  // (the reference rounds its float64 U^T U v to float32 at decomposition.py:982)
  float acc[NOUT];
#pragma unroll
  for (int c = 0; c < NOUT; ++c) acc[c] = 0.f;
  for (int e = nbr_ptr[a]; e < nbr_ptr[a + 1]; ++e) {
    const int row0 = nbr[4 * e], rb = nbr[4 * e + 1], blk = nbr[4 * e + 2], flags = nbr[4 * e + 3];
    const float* src = (flags & 2) ? Gbg + (long)blk * 4096 : Gblk + (long)blk * 4096;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * NOUT; i += 256) {
      const int cp = i / NOUT, c = i - cp * NOUT;
      // stored block is [row-tile comp][col-tile comp]; transposed flag: this tile is the column tile
      g[cp][c] = src[(flags & 1) ? cp * 64 + c : c * 64 + cp];
    }
    __syncthreads();
    // eight rows of M in flight per thread (unconditional loads on clamped indices), FMAs only for the
    // 8-column groups below this tile's rank
    const long xc = (x < ncols) ? x : ncols - 1;
    for (int cp0 = 0; cp0 < rb; cp0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = M[(long)(row0 + ((cp0 + u < rb) ? cp0 + u : rb - 1)) * ldm + xc];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (cp0 + u < rb) {
#pragma unroll
          for (int c0 = 0; c0 < NOUT; c0 += 8) {
            if (c0 < ra) {
#pragma unroll
              for (int c = c0; c < c0 + 8; ++c) acc[c] = fmaf(g[cp0 + u][c], v[u], acc[c]);
            }
          }
        }
      }
    }
  }
  if (x < ncols) {
    const long off = col_off[a];
#pragma unroll
    for (int c = 0; c < NOUT; ++c)
      if (c < ra) GM[(off + c) * ldgm + x] = acc[c];
  }
}

// The same product on the fp32 MFMA path: a workgroup owns tile a x 256 columns (a wave 64 of them).  Per neighbour
// block the (rank-masked) G block goes through LDS as the A operand (m = component of tile a, k = component of the
// neighbour), a lane's float4 of M row cp0 + k is the B operand of four N tiles at once (N tile s = columns
// 4 n + s: the permutation is undone by the float4 store of the four accumulators).  16 x CT x 4 accumulators.
typedef float ga_f32x4 __attribute__((ext_vector_type(4)));
constexpr int GA_LD = 80;  // g[cp][c] row length: 80 = 16 mod 64, the four k rows of a fragment hit disjoint banks

template <int CT>
__global__ __launch_bounds__(256) void gram_apply_mfma_kernel(const float* __restrict__ Gblk, const float* __restrict__ Gbg,
                                                              const int* __restrict__ nbr_ptr, const int* __restrict__ nbr,
                                                              const int* __restrict__ col_off, const int* __restrict__ ranks,
                                                              const float* __restrict__ M, long ldm, int ncols,
                                                              float* __restrict__ GM, long ldgm) {
  __shared__ float g[64][GA_LD];
  const int a = blockIdx.y;
  const int ra = ranks[a];
  if (ra == 0) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform, and known to be)
  const int n16 = lane & 15, kk = lane >> 4;
  const int x0 = blockIdx.x * 256 + wave * 64 + 4 * n16;
  const long xc = (x0 + 3 < ldm) ? x0 : 0;  // idle lanes read valid columns; their results are not stored
  ga_f32x4 acc[CT][4];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) acc[ct][sx] = (ga_f32x4){0.f, 0.f, 0.f, 0.f};
  for (int e = nbr_ptr[a]; e < nbr_ptr[a + 1]; ++e) {
    const int row0 = nbr[4 * e], rb = nbr[4 * e + 1], blk = nbr[4 * e + 2], flags = nbr[4 * e + 3];
    const float* src = (flags & 2) ? Gbg + (long)blk * 4096 : Gblk + (long)blk * 4096;
    __syncthreads();
    // (only the part the MFMAs below read: cp < round_up(rb, 16), c < round_up(ra, 16); consecutive threads read
    // consecutive addresses of the stored block in either orientation)
    const int cpn = (rb + 15) & ~15, cn = (ra + 15) & ~15;
    if (flags & 1) {
      for (int i = threadIdx.x; i < cpn * cn; i += 256) {
        const int cp = i / cn, c = i - cp * cn;
        const float v = src[cp * 64 + c];
        g[cp][c] = (cp < rb && c < ra) ? v : 0.f;  // masked: rows / columns beyond the ranks hold other tiles' data
      }
    } else {
      for (int i = threadIdx.x; i < cpn * cn; i += 256) {
        const int c = i / cpn, cp = i - c * cpn;
        const float v = src[c * 64 + cp];
        g[cp][c] = (cp < rb && c < ra) ? v : 0.f;
      }
    }
    __syncthreads();
    // 16 rows of M (four k steps) per batch, all four loads in flight at once, unconditional (rows clamped into the
    // block; the masked G makes their contribution zero)
    const float* mp = M + (long)row0 * ldm + xc;
    for (int cp0 = 0; cp0 < rb; cp0 += 16) {
      ga_f32x4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const ga_f32x4*>(mp + (long)min(cp0 + 4 * u + kk, rb - 1) * ldm);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (cp0 + 4 * u < rb) {  // workgroup-uniform
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            if (16 * ct < ra) {  // workgroup-uniform: component tiles beyond this tile's rank are skipped
              const float av = g[cp0 + 4 * u + kk][16 * ct + n16];
#pragma unroll
              for (int sx = 0; sx < 4; ++sx) acc[ct][sx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u][sx], acc[ct][sx], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  if (x0 >= ncols) return;
  const long off = col_off[a];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = 16 * ct + 4 * kk + i;
      if (c >= ra) continue;
      float* o = GM + (off + c) * ldgm + x0;
      if (x0 + 3 < ncols) {
        *reinterpret_cast<ga_f32x4*>(o) = (ga_f32x4){acc[ct][0][i], acc[ct][1][i], acc[ct][2][i], acc[ct][3][i]};
      } else {
        for (int sx = 0; sx < 4 && x0 + sx < ncols; ++sx) o[sx] = acc[ct][sx][i];
      }
    }
}

// blocks: Gblk [n_pairs][64][64], Gbg [n_tiles][64][64], Gstrip [K][ldgs] (ldgs >= Rt + K)
int pmd_gram_blocks_impl(pmd_ctx* ctx, const float* Uw, int dpad, int b1, int b2, const int* pix, const int* pairs,
                         int n_pairs, const int* origins, const int* col_off, const int* ranks, int n_tiles, int Rt,
                         const float* basis, long D, int K, float* Gblk, float* Gbg, float* Gstrip, long ldgs) {
  pmd_prof_scope prof__(ctx, "gram_blocks");
  if (n_pairs > 0) {
    hipLaunchKernelGGL(gram_pair_blocks_kernel, dim3(n_pairs), dim3(256), 0, ctx->stream, Uw, dpad, b1, pairs, origins,
                       ranks, Gblk);
    PMD_LAUNCH_CHECK(ctx, "gram_pair_blocks_kernel");
  }
  if (K > 0) {
    PMD_HIP(ctx, hipMemsetAsync(Gstrip, 0, (size_t)K * ldgs * sizeof(float), ctx->stream));
    // Gbg: [K block of 64][tile][64][64] (one block of 64 background columns per launch)
    for (int k0 = 0; k0 < K; k0 += 64) {
      const int kc = (K - k0 < 64) ? K - k0 : 64;
      hipLaunchKernelGGL(gram_bg_blocks_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, Uw, dpad, b1 * b2, pix, basis + k0, kc,
                         col_off, ranks, Gbg + (long)(k0 / 64) * n_tiles * 4096, Gstrip + (long)k0 * ldgs, ldgs, K);
      PMD_LAUNCH_CHECK(ctx, "gram_bg_blocks_kernel");
    }
    hipLaunchKernelGGL(gram_bgbg_strip_kernel, dim3(K, K), dim3(256), 0, ctx->stream, basis, D, K, Rt, Gstrip, ldgs);
    PMD_LAUNCH_CHECK(ctx, "gram_bgbg_strip_kernel");
  }
  return PMD_OK;
}

// GM = G M for the block-sparse G (rows of the K background columns via one small GEMM).
int pmd_gram_apply_impl(pmd_ctx* ctx, const float* Gblk, const float* Gbg, const float* Gstrip, long ldgs,
                        const int* nbr_ptr, const int* nbr, const int* col_off, const int* ranks, int n_tiles, int Rt,
                        int K, int max_rank, const float* M, long ldm, int ncols, float* GM, long ldgm) {
  {
    pmd_prof_scope prof__(ctx, "gram_apply");
    dim3 grid((ncols + 255) / 256, n_tiles);
    const char* gam = getenv("PMD_GRAM_APPLY_MFMA");
    const bool mfma_ok = !(gam && atoi(gam) == 0) && ldm % 4 == 0 && ldgm % 4 == 0 && ((uintptr_t)M & 15) == 0 &&
                         ((uintptr_t)GM & 15) == 0 && max_rank <= 64;
    if (mfma_ok) {
      const int ct = (max_rank + 15) / 16;
#define GA_LAUNCH(CT_)                                                                                                   \
  hipLaunchKernelGGL(gram_apply_mfma_kernel<CT_>, grid, dim3(256), 0, ctx->stream, Gblk, Gbg, nbr_ptr, nbr, col_off, ranks, \
                     M, ldm, ncols, GM, ldgm)
      if (ct <= 1) GA_LAUNCH(1);
      else if (ct == 2) GA_LAUNCH(2);
      else if (ct == 3) GA_LAUNCH(3);
      else GA_LAUNCH(4);
#undef GA_LAUNCH
    } else if (max_rank <= 16)
      hipLaunchKernelGGL(gram_apply_kernel<16>, grid, dim3(256), 0, ctx->stream, Gblk, Gbg, nbr_ptr, nbr, col_off, ranks,
                         M, ldm, ncols, GM, ldgm);
    else if (max_rank <= 32)
      hipLaunchKernelGGL(gram_apply_kernel<32>, grid, dim3(256), 0, ctx->stream, Gblk, Gbg, nbr_ptr, nbr, col_off, ranks,
                         M, ldm, ncols, GM, ldgm);
    else
      hipLaunchKernelGGL(gram_apply_kernel<64>, grid, dim3(256), 0, ctx->stream, Gblk, Gbg, nbr_ptr, nbr, col_off, ranks,
                         M, ldm, ncols, GM, ldgm);
    PMD_LAUNCH_CHECK(ctx, "gram_apply_kernel");
  }
  if (K > 0) RUN(pmd_gemm_rm(ctx, 0, 0, K, ncols, Rt + K, 1.f, Gstrip, ldgs, M, ldm, 0.f, GM + (long)Rt * ldgm, ldgm));
  return PMD_OK;
}

// =============================================================================================
// CSR assembly of the sparse spatial matrix on the device (decomposition.py:812-853, :929-930):
// row p (output pixel order) holds, for every covering tile in tile order, rank_t entries
// (float64(U) * w) * (1/cumw), then the K background entries.  cover1[i][0..3] / cover2[j][0..3]
// list the indices of the tile-row / tile-column origins covering FOV row i / column j (-1 pads).
// =============================================================================================
__device__ __forceinline__ void csr_pixel(long p, int d1, int d2, int order_f, int* i, int* j) {
  if (order_f) { *j = (int)(p / d1); *i = (int)(p - (long)(*j) * d1); }
  else { *i = (int)(p / d2); *j = (int)(p - (long)(*i) * d2); }
}

__global__ void csr_count_kernel(int d1, int d2, int order_f, const int* __restrict__ cover1,
                                 const int* __restrict__ cover2, int n2, const int* __restrict__ ranks, int K,
                                 long* __restrict__ row_nnz) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long)d1 * d2) return;
  int i, j;
  csr_pixel(p, d1, d2, order_f, &i, &j);
  long n = K;
  for (int a = 0; a < 4; ++a) {
    const int ki = cover1[4 * i + a];
    if (ki < 0) continue;
    for (int b = 0; b < 4; ++b) {
      const int ji = cover2[4 * j + b];
      if (ji >= 0) n += ranks[ki * n2 + ji];
    }
  }
  row_nnz[p] = n;
}

__global__ void csr_fill_kernel(int d1, int d2, int order_f, int b1, const int* __restrict__ cover1,
                                const int* __restrict__ cover2, const int* __restrict__ orig1,
                                const int* __restrict__ orig2, int n2, const int* __restrict__ ranks,
                                const int* __restrict__ col_off, const float* __restrict__ Ut, int dpad,
                                const float* __restrict__ w, const double* __restrict__ inv_cumw,
                                const float* __restrict__ basis, int K, int Rt, const long* __restrict__ indptr,
                                double* __restrict__ data, int* __restrict__ indices, int* __restrict__ zero_count, int rpad) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long)d1 * d2) return;
  int i, j;
  csr_pixel(p, d1, d2, order_f, &i, &j);
  long pos = indptr[p];
  const double inv = inv_cumw[(long)i * d2 + j];
  int zeros = 0;
  for (int a = 0; a < 4; ++a) {
    const int ki = cover1[4 * i + a];
    if (ki < 0) continue;
    for (int b = 0; b < 4; ++b) {
      const int ji = cover2[4 * j + b];
      if (ji < 0) continue;
      const int tile = ki * n2 + ji;
      const int q = (i - orig1[ki]) + b1 * (j - orig2[ji]);
      const double wq = (double)w[q];
      const int rk = ranks[tile];
      const int off = col_off[tile];
      for (int c = 0; c < rk; ++c) {
        const double v = ((double)Ut[(long)tile * rpad * dpad + (long)c * dpad + q] * wq) * inv;
        zeros += (v == 0.0);
        data[pos] = v;
        indices[pos] = off + c;
        ++pos;
      }
    }
  }
  for (int k = 0; k < K; ++k) {
    const double v = (double)basis[((long)i * d2 + j) * K + k];
    zeros += (v == 0.0);
    data[pos] = v;
    indices[pos] = Rt + k;
    ++pos;
  }
  if (zeros) atomicAdd(zero_count, zeros);
}

int pmd_csr_count_impl(pmd_ctx* ctx, int d1, int d2, int order_f, const int* cover1, const int* cover2, int n2,
                       const int* ranks, int K, long* row_nnz) {
  pmd_prof_scope prof__(ctx, "csr_assembly");
  const long D = (long)d1 * d2;
  hipLaunchKernelGGL(csr_count_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, d1, d2, order_f,
                     cover1, cover2, n2, ranks, K, row_nnz);
  PMD_LAUNCH_CHECK(ctx, "csr_count_kernel");
  return PMD_OK;
}

int pmd_csr_fill_impl(pmd_ctx* ctx, int d1, int d2, int order_f, int b1, const int* cover1, const int* cover2,
                      const int* orig1, const int* orig2, int n2, const int* ranks, const int* col_off, const float* Ut,
                      int dpad, const float* w, const double* inv_cumw, const float* basis, int K, int Rt,
                      const long* indptr, double* data, int* indices, int* zero_count, int rpad) {
  pmd_prof_scope prof__(ctx, "csr_assembly");
  const long D = (long)d1 * d2;
  PMD_HIP(ctx, hipMemsetAsync(zero_count, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(csr_fill_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, d1, d2, order_f, b1,
                     cover1, cover2, orig1, orig2, n2, ranks, col_off, Ut, dpad, w, inv_cumw, basis, K, Rt, indptr, data,
                     indices, zero_count, rpad);
  PMD_LAUNCH_CHECK(ctx, "csr_fill_kernel");
  return PMD_OK;
}

// =============================================================================================
// Factored form of A15 + A16 + A17 for R > frames (P = M E^T is never formed):
//   pmd_orthogonalize_factored: C = M^T (G M) -> Et (R' x m), rows = eigenvectors / sqrt(lambda)
//   pmd_projected_svd_factored: V = Et (M^T Z); SVD of V; R_out = M (Et^T W)
// =============================================================================================
size_t pmd_orthogonalize_factored_workspace_bytes_impl(int m) {
  return (size_t)m * m * sizeof(float) + (size_t)m * (3 * sizeof(float) + sizeof(int)) + 8192;
}

int pmd_orthogonalize_factored_impl(pmd_ctx* ctx, const float* M, int Rc, int m, long ldm, const float* GM, long ldgm,
                                    float* Et_out, long lde, int* rprime_out, void* ws, size_t ws_bytes) {
  pmd_arena ar(ws, ws_bytes);
  float* C = ar.take_n<float>((size_t)m * m);
  float* w = ar.take_n<float>(m);
  float* work = ar.take_n<float>(m);
  float* scale = ar.take_n<float>(m);
  int* perm = ar.take_n<int>(m);
  int* info = ar.take_n<int>(4);
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_orthogonalize_factored", "workspace too small");
  RUN(pmd_gemm_rm(ctx, 1, 0, m, m, Rc, 1.f, M, ldm, GM, ldgm, 0.f, C, m));
  RUN(pmd_syevd(ctx, m, C, m, w, work, info));
  std::vector<float> hw(m);
  int hinfo = 0;
  PMD_HIP(ctx, hipMemcpyAsync(hw.data(), w, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  PMD_HIP(ctx, hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (hinfo != 0) return pmd_fail(ctx, PMD_ERR_BLAS, "rocsolver_ssyevd", "did not converge");
  std::vector<int> idx(m);
  for (int i = 0; i < m; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return std::fabs(hw[a]) > std::fabs(hw[b]); });
  std::vector<int> hperm;
  std::vector<float> hscale;
  // jnp.linalg.svd(hermitian=True) returns |lambda| and u = v sign(lambda): `eig_vals > 0` (decomposition.py:988)
  // keeps every direction whose eigenvalue is not exactly zero.  null_cutoff >= 0 (pmd_ctx_set_null_cutoff):
  // only lambda > cutoff * lambda_max.
  const float lmax = m > 0 ? std::fabs(hw[idx[0]]) : 0.f;
  for (int i = 0; i < m; ++i) {
    const float lam = hw[idx[i]];
    const bool keep = ctx->null_cutoff < 0.f ? (lam != 0.f && lam == lam) : (lam > ctx->null_cutoff * lmax);
    if (keep) {
      hperm.push_back(idx[i]);
      hscale.push_back((lam < 0.f ? -1.0f : 1.0f) / std::sqrt(std::fabs(lam)));
    }
  }
  const int rp = (int)hperm.size();
  *rprime_out = rp;
  if (rp == 0) return PMD_OK;
  PMD_HIP(ctx, hipMemcpyAsync(perm, hperm.data(), (size_t)rp * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  PMD_HIP(ctx, hipMemcpyAsync(scale, hscale.data(), (size_t)rp * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  RUN(launch_gather_rows(ctx, C, m, perm, scale, rp, m, Et_out, lde));
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PMD_OK;
}

size_t pmd_projected_svd_factored_workspace_bytes_impl(int Rc, int m, int rp, int T) {
  return (size_t)m * T * sizeof(float) + (size_t)rp * T * sizeof(float) + (size_t)m * rp * sizeof(float) +
         (size_t)m * pmd_round_up(Rc, 64) * sizeof(float) + pmd_projected_svd_workspace_bytes_impl(1, rp, T) + 16384;
}

// M: Rc x m; Et: rp x m; Z: Rc x T.  Outputs R_out (Rc x nk), s (nk), Vt (nk x T), nk = min(rp, T);
// Vp_out (rp x T, optional, may be NULL) receives V = P^T Z.
int pmd_projected_svd_factored_impl(pmd_ctx* ctx, const float* M, int Rc, int m, long ldm, const float* Et, int rp,
                                    long lde, const float* Z, int T, long ldz, float* R_out, long ldr, float* s_out,
                                    float* Vt_out, long ldvt, float* Vp_out, long ldvp, float* X1_out,
                                    const float* W1_in, int et_lower, void* ws, size_t ws_bytes) {
  if (rp > T) return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "pmd_projected_svd_factored", "needs R' <= T");
  pmd_arena ar(ws, ws_bytes);
  float* W1 = ar.take_n<float>((size_t)m * T);
  float* Vp = Vp_out ? Vp_out : ar.take_n<float>((size_t)rp * T);
  const long ldv = Vp_out ? ldvp : T;
  float* X1 = X1_out ? X1_out : ar.take_n<float>((size_t)m * rp);
  const long ldt = pmd_round_up(Rc, 64);   // (rows of the transposed copy on 256-byte boundaries, see pmd_gram_mtgm_ld_impl)
  float* Mt = ar.take_n<float>((size_t)m * ldt);
  const size_t sub_bytes = pmd_projected_svd_workspace_bytes_impl(1, rp, T);
  void* sub = ar.take(sub_bytes);
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_projected_svd_factored", "workspace too small");
  // M^T Z as a plain (non-transposed) product of an explicit copy of M^T: rocBLAS' transposed-A kernels
  // reach half the rate of the plain ones at K = Rc ~ 5e4 (72 vs 147 TFLOP/s, scripts/gemm_probe.hip),
  // and the copy is one 2 x 4 Rc m byte pass
  if (!W1_in) {
    RUN(launch_transpose(ctx, M, ldm, Rc, m, Mt, ldt));
    RUN(pmd_gemm_rm(ctx, 0, 0, m, T, Rc, 1.f, Mt, ldt, Z, ldz, 0.f, W1, T));      // M^T Z
  }
  // (W1_in: the caller has formed M^T Z already, e.g. as an all-reduced sum of per-rank row-range partials)
  // Cholesky route: Et is lower triangular (zeros above the diagonal), strmm does half the work in fp32; where the full
  // product runs from fp16 pieces (gemm_f16x2.hip) that is faster still (6 against 10 ms at order 10^4)
  const bool tri_any = et_lower && rp == m;
  const bool tri = tri_any && !pmd_f16x2_wanted(ctx, rp, T, m);
  const float one = 1.f;
  if (tri) {
    // row-major Vp = Et W1  <=>  column-major Vp^T = W1^T Et^T with Et^T upper on the right
    pmd_prof_scope prof__(ctx, "rocblas_strmm");
    PMD_BLAS(ctx, rocblas_strmm(ctx->blas, rocblas_side_right, rocblas_fill_upper, rocblas_operation_none,
                                rocblas_diagonal_non_unit, T, m, &one, Et, (rocblas_int)lde, W1_in ? W1_in : W1, T, Vp,
                                (rocblas_int)ldv));
  } else {
    RUN(pmd_gemm_rm(ctx, 0, 0, rp, T, m, 1.f, Et, lde, W1_in ? W1_in : W1, T, 0.f, Vp, ldv));     // V = Et (M^T Z)
  }
  // SVD of V with the identity as projection: the "R" it returns is W (rp x rp), reuse W1's memory
  float* Wmat = W1;  // rp x rp  (rp <= m, T)
  RUN(pmd_projected_svd_impl(ctx, nullptr, 0, 0, Vp, rp, T, ldv, Wmat, rp, s_out, Vt_out, ldvt, sub, sub_bytes));
  // R = M (Et^T W)
  if (tri_any && !pmd_f16x2_wanted(ctx, m, rp, rp)) {
    // row-major X1 = Et^T W  <=>  column-major X1^T = W^T (Et^T)^T
    pmd_prof_scope prof__(ctx, "rocblas_strmm");
    PMD_BLAS(ctx, rocblas_strmm(ctx->blas, rocblas_side_right, rocblas_fill_upper, rocblas_operation_transpose,
                                rocblas_diagonal_non_unit, rp, m, &one, Et, (rocblas_int)lde, Wmat, rp, X1, rp));
  } else {
    RUN(pmd_gemm_rm(ctx, 1, 0, m, rp, rp, 1.f, Et, lde, Wmat, rp, 0.f, X1, rp));
  }
  if (R_out) RUN(pmd_gemm_rm(ctx, 0, 0, Rc, rp, m, 1.f, M, ldm, X1, rp, 0.f, R_out, ldr));
  return PMD_OK;
}

// =============================================================================================
// Cholesky form of the orthogonalisation (R > frames): any P with the column space of `right` and
// P^T G P = I gives the same final R, s, Vt (P_chol = P_eigh Q with Q orthogonal, and the
// projected SVD absorbs Q).  C = M^T G M = U_c^T U_c (upper Cholesky), P = M U_c^{-1}, i.e.
// Et = U_c^{-T} (lower triangular).  ok_host = 0 when C is not numerically positive definite
// (the caller then uses the eigendecomposition, decomposition.py:984-996).
// =============================================================================================
// Cholesky factor of one diagonal block (nb <= 128): A = L L^T, row-major lower triangle in place.
// One workgroup of 16 x 16 threads; thread (tr, tc) keeps the entries r = tr (mod 16), c = tc (mod 16) in
// registers for the whole factorisation, only the pivot column travels through LDS (one barrier per step).
// *info = 1-based global index of the first non-positive pivot (left untouched otherwise).
#define CHOL_NB 128
// n_abs_last > 0: a negative pivot number n_abs_last (1-based, the LAST pivot of the whole matrix) is replaced by its
// absolute value - the Cholesky-route form of the reference keeping a numerically null direction through |lambda|.
// linv_out != NULL: the inverse of the factor, row-major lower triangle L^{-1} with leading dimension CHOL_NB (zeros above the
// diagonal), is formed in the same launch (dynamic LDS: two CHOL_NB x CHOL_LS float arrays): the factor goes to LDS, thread c
// solves L x = e_c by forward substitution (column c of the inverse, kept in LDS).  This replaces rocBLAS' strtri on the
// block (five launches, 150 us in its diagonal kernel) in the latency-bound chain of the blocked factorisation.
#define CHOL_LS (CHOL_NB + 1)
__global__ __launch_bounds__(256) void potf2_block_kernel(float* __restrict__ A, long ld, int nb, int k0, int* __restrict__ info,
                                                          int n_abs_last, float* __restrict__ linv_out) {
  extern __shared__ float chol_dyn[];
  __shared__ float s_col[2][CHOL_NB];
  const int tid = threadIdx.x;
  const int tr = tid >> 4, tc = tid & 15;
  float a[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = tr + 16 * i, c = tc + 16 * j;
      const bool in = r < nb && c <= r;
      const float v = A[(long)(in ? r : 0) * ld + (in ? c : 0)];  // unconditional load, masked value
      a[i][j] = in ? v : 0.f;
    }
  int bad = 0;
  for (int k = 0; k < nb; ++k) {
    float* col = s_col[k & 1];
    const int jk = k >> 4;
    if (tc == (k & 15)) {
      // this thread owns entries of column k: publish the part on and below the diagonal
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = tr + 16 * i;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v = (j == jk) ? a[i][j] : v;
        if (r >= k && r < nb) col[r] = v;
      }
    }
    __syncthreads();
    float piv = col[k];
    if (!(piv > 0.f)) {
      if (k0 + k + 1 == n_abs_last && piv < 0.f) {
        piv = -piv;
      } else {
        bad = k0 + k + 1;
        break;  // uniform: every thread reads the same pivot
      }
    }
    const float dk = sqrtf(piv);
    const float inv = 1.f / dk;
    float lr[8], lc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = tr + 16 * i, c = tc + 16 * i;
      const float vr = col[r], vc = col[c];  // unconditional LDS reads (stale entries above the pivot are masked)
      lr[i] = (r > k && r < nb) ? vr * inv : 0.f;
      lc[i] = (c > k && c < nb) ? vc * inv : 0.f;
    }
    // only register columns j >= k / 16 and rows i >= j can still change (uniform tests: whole slabs are skipped)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < jk) continue;
#pragma unroll
      for (int i = j; i < 8; ++i) {
        const int r = tr + 16 * i, c = tc + 16 * j;
        // trailing update (c > k, r >= c): lr, lc are zero outside it except for r < c, which is masked here
        if (r >= c) a[i][j] -= lr[i] * lc[j];
        // column k itself: the scaled entries and the pivot
        if (j == jk && c == k) a[i][j] = (r == k) ? dk : ((r > k) ? lr[i] : a[i][j]);
      }
    }
  }
  if (bad) {
    if (tid == 0 && *info == 0) *info = bad;
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = tr + 16 * i, c = tc + 16 * j;
      if (r < nb && c <= r) A[(long)r * ld + c] = a[i][j];
    }
  if (!linv_out) return;
  float* Ls = chol_dyn;
  float* Xs = chol_dyn + CHOL_NB * CHOL_LS;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = tr + 16 * i, c = tc + 16 * j;
      if (r < nb && c <= r) Ls[r * CHOL_LS + c] = a[i][j];
    }
  __syncthreads();
  {
    // two lanes per column (even / odd k), 32 columns per wave; the inner sums take four (k, k + 2, k + 4, k + 6) terms per
    // lane and trip with all eight LDS reads issued before the first product (a plain loop waits out one LDS latency per term:
    // 430 us instead of 40).  A wave only reads inverse entries it wrote itself: wave-level ordering is enough.
    const int c = tid >> 1, h = tid & 1;
    const int c0 = __builtin_amdgcn_readfirstlane(c & ~31);
    if (h == 0 && c < nb) Xs[c * CHOL_LS + c] = 1.f / Ls[c * CHOL_LS + c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = c0 + 1; i < nb; ++i) {
      const float* lrow = Ls + i * CHOL_LS;
      float acc0 = 0.f, acc1 = 0.f;
      for (int k = c0 + h; k < i; k += 8) {
        float lv[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kk = min(k + 2 * u, CHOL_NB - 1);
          lv[u] = lrow[kk];
          xv[u] = Xs[kk * CHOL_LS + min(c, CHOL_NB - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kk = k + 2 * u;
          const float t = (kk < i && kk >= c) ? lv[u] * xv[u] : 0.f;   // (entries never written are masked, not multiplied)
          if (u & 1) acc1 += t; else acc0 += t;
        }
      }
      float acc = acc0 + acc1;
      acc += __shfl_xor(acc, 1);
      if (h == 0 && i > c && c < nb) Xs[i * CHOL_LS + c] = -acc / lrow[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __syncthreads();
  for (int idx = tid; idx < nb * nb; idx += 256) {
    const int i = idx / nb, c = idx - i * nb;
    linv_out[(long)i * CHOL_NB + c] = (c <= i) ? Xs[i * CHOL_LS + c] : 0.f;
  }
}

// Panel of the blocked factorisation, in place: P (rest x nb, row-major, leading dimension ld) <- P L^{-T}, with
// linv = row-major L^{-1} (leading dimension CHOL_NB); a compact copy (leading dimension CHOL_NB) feeds the trailing update.
// 16 rows per workgroup; the inverse passes through LDS in four slabs of 32 inner indices, transposed on the way in.
// 25 KB of LDS: the chain runs next to large products on the main stream, and a workgroup that asks for more than the
// CU has left waits for one of theirs to retire (measured: a 83 KB version of this kernel and a 132 KB fused
// factor-and-invert kernel were faster alone and 11 ms slower in the step).
// (rocBLAS picks 16 x 16 / 128 x 64 macro tiles for this 128-deep product: 130-180 us per panel.)
#define CHOL_PR 16
__global__ __launch_bounds__(256) void chol_panel_kernel(float* __restrict__ P, long ld, int rest, int nb, const float* __restrict__ linv,
                                                         float* __restrict__ compact) {
  __shared__ float Ps[CHOL_PR][CHOL_LS];
  __shared__ float Lt[32][CHOL_LS];            // Lt[k - k0][j] = linv[j][k]
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * CHOL_PR;
  for (int idx = tid; idx < CHOL_PR * CHOL_NB; idx += 256) {
    const int r = idx / CHOL_NB, k = idx - r * CHOL_NB;
    Ps[r][k] = (r0 + r < rest && k < nb) ? P[(long)(r0 + r) * ld + k] : 0.f;
  }
  const int ty = tid >> 5, tx = tid & 31;   // rows 2 ty, 2 ty + 1; columns tx + 32 q
  float acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  for (int k0 = 0; k0 < nb; k0 += 32) {
    __syncthreads();
    for (int idx = tid; idx < CHOL_NB * 32; idx += 256) {
      const int j = idx >> 5, kk = idx & 31;
      Lt[kk][j] = (j < nb && k0 + kk <= j) ? linv[(long)j * CHOL_NB + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      const float p0 = Ps[2 * ty][k0 + kk], p1 = Ps[2 * ty + 1][k0 + kk];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float lv = Lt[kk][tx + 32 * b];
        acc[0][b] += p0 * lv;
        acc[1][b] += p1 * lv;
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int r = r0 + 2 * ty + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = tx + 32 * b;
      if (r < rest && j < nb) {
        P[(long)r * ld + j] = acc[a][b];
        compact[(long)r * CHOL_NB + j] = acc[a][b];
      }
    }
  }
}

// Blocked right-looking Cholesky of the row-major lower triangle (= column-major upper, A = U^T U):
// diagonal block in LDS, panel by rocBLAS strsm, trailing update by ssyrk.  rocSOLVER's spotrf spends half
// of its 35 ms at n = 10^4 in its unblocked diagonal-block kernel.
// tmp: n x CHOL_NB floats (the panel before it is copied back), linv: CHOL_NB x CHOL_NB floats
static int chol_lower_rm(pmd_ctx* ctx, int n, float* A, long ld, int* info, float* tmp, float* linv, int abs_last_pivot) {
  pmd_prof_scope prof__(ctx, "cholesky");
  PMD_HIP(ctx, hipMemsetAsync(info, 0, sizeof(int), ctx->stream));
  PMD_HIP(ctx, hipMemsetAsync(linv, 0, sizeof(float) * CHOL_NB * CHOL_NB, ctx->stream));
  const float one = 1.f, minus1 = -1.f;
  // PMD_CHOL_CHAIN=rocblas: the earlier chain (strtri on the block, panel by sgemm into a copy) for A/B runs
  static int own_chain = -1;
  if (own_chain < 0) { const char* e = getenv("PMD_CHOL_CHAIN"); own_chain = (e && !strcmp(e, "rocblas")) ? 0 : 1; }
  // PMD_CHOL_FUSED=1: factor and invert the diagonal block in one launch (132 KB of LDS: faster alone, slower next to the
  // large products of the main stream, see chol_panel_kernel)
  static int fused_inv = -1;
  if (fused_inv < 0) { const char* e = getenv("PMD_CHOL_FUSED"); fused_inv = (e && !strcmp(e, "1")) ? 1 : 0; }
  const size_t lds = 2 * (size_t)CHOL_NB * CHOL_LS * sizeof(float);
  if (own_chain && fused_inv)
    PMD_HIP(ctx, hipFuncSetAttribute((const void*)potf2_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int k0 = 0; k0 < n; k0 += CHOL_NB) {
    const int nb = std::min(CHOL_NB, n - k0);
    float* D = A + (long)k0 * ld + k0;
    const int rest = n - k0 - nb;
    const bool fused = own_chain && fused_inv && rest > 0;
    hipLaunchKernelGGL(potf2_block_kernel, dim3(1), dim3(256), fused ? lds : 0, ctx->stream, D, ld, nb, k0, info, abs_last_pivot ? n : -1,
                       fused ? linv : (float*)nullptr);
    PMD_LAUNCH_CHECK(ctx, "potf2_block_kernel");
    if (rest <= 0) break;
    float* P = A + (long)(k0 + nb) * ld + k0;        // row-major rest x nb  ==  column-major nb x rest
    float* T22 = A + (long)(k0 + nb) * ld + (k0 + nb);
    if (own_chain) {
      // L21 = A21 L_kk^{-T} in place (+ a compact copy for the trailing update)
      if (!fused)
        PMD_BLAS(ctx, rocblas_strtri(ctx->blas, rocblas_fill_upper, rocblas_diagonal_non_unit, nb, D, (rocblas_int)ld, linv, CHOL_NB));
      hipLaunchKernelGGL(chol_panel_kernel, dim3((rest + CHOL_PR - 1) / CHOL_PR), dim3(256), 0, ctx->stream, P, ld, rest, nb, linv, tmp);
      PMD_LAUNCH_CHECK(ctx, "chol_panel_kernel");
      PMD_BLAS(ctx, rocblas_ssyrk(ctx->blas, rocblas_fill_upper, rocblas_operation_transpose, rest, nb, &minus1, tmp, CHOL_NB, &one,
                                  T22, (rocblas_int)ld));
      continue;
    }
    // L21 = A21 L_kk^{-T}: invert the 128 x 128 block (column-major upper view of the row-major lower block)
    // and multiply - rocBLAS' strsm spends ~20 small launches per panel on the same thing
    PMD_BLAS(ctx, rocblas_strtri(ctx->blas, rocblas_fill_upper, rocblas_diagonal_non_unit, nb, D, (rocblas_int)ld, linv, CHOL_NB));
    RUN(pmd_gemm_rm(ctx, 0, 1, rest, nb, nb, 1.f, P, ld, linv, CHOL_NB, 0.f, tmp, CHOL_NB));
    PMD_HIP(ctx, hipMemcpy2DAsync(P, (size_t)ld * sizeof(float), tmp, (size_t)CHOL_NB * sizeof(float), (size_t)nb * sizeof(float),
                                  rest, hipMemcpyDeviceToDevice, ctx->stream));
    PMD_BLAS(ctx, rocblas_ssyrk(ctx->blas, rocblas_fill_upper, rocblas_operation_transpose, rest, nb, &minus1, tmp,
                                CHOL_NB, &one, T22, (rocblas_int)ld));
  }
  return PMD_OK;
}

__global__ void tril_mask_kernel(float* __restrict__ A, long ld, int n) {
  const int i = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    if (j > i) A[(long)i * ld + j] = 0.f;
}

size_t pmd_orthogonalize_chol_workspace_bytes_impl(int Rc, int m) {
  return std::max(pmd_gram_mtgm_workspace_bytes_impl(Rc, m), pmd_chol_inverse_workspace_bytes_impl(m)) + ((size_t)m + CHOL_NB) * CHOL_NB * sizeof(float) + 16384;
}

// C (row-major lower block triangle, the part the Cholesky step reads) = M^T GM over `rows` rows of both.
// With the tile rows sharded over ranks every rank passes its own row range and the partial C are all-reduced.
// The transposed copy M^T (m x rows) at the start of the workspace has leading dimension pmd_gram_mtgm_ld(rows): a multiple of
// 64 floats.  (With ld = rows = 113 305 at BASELINE config 4 every row of the copy started off a 16-byte boundary and rocBLAS
// fell back to element-wise loads: 51 TFLOP/s for this product instead of 130.)
long pmd_gram_mtgm_ld_impl(int rows) { return pmd_round_up(rows, 64); }
size_t pmd_gram_mtgm_workspace_bytes_impl(int rows, int m) { return (size_t)m * pmd_gram_mtgm_ld_impl(rows) * sizeof(float) + 4096; }

int pmd_gram_mtgm_impl(pmd_ctx* ctx, const float* M, int rows, int m, long ldm, const float* GM, long ldgm, float* C,
                       long ldc, void* ws, size_t ws_bytes) {
  if (rows <= 0) return PMD_OK;
  pmd_arena ar(ws, ws_bytes);
  const long ldt = pmd_gram_mtgm_ld_impl(rows);
  float* Mt = ar.take_n<float>((size_t)m * ldt);
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_gram_mtgm", "workspace too small");
  // row blocks C[i0:i0+bs, 0:i0+bs] = Mt[i0:i0+bs, :] GM[:, 0:i0+bs]  (3/5 of the flops at 5 blocks)
  RUN(launch_transpose(ctx, M, ldm, rows, m, Mt, ldt));
  const char* cbenv = getenv("PMD_C_BLOCKS");
  const int nblk = cbenv ? std::max(1, atoi(cbenv)) : 5;  // 2/3/4/5/6/8/12/16 blocks at m = 10^4: 70/64/57/52/59/58/60/67 ms
  const int bs = std::max(256, ((m + nblk - 1) / nblk + 255) / 256 * 256);
  // fp16-piece products (gemm_f16x2.hip): both operands are split ONCE, the row blocks are views of the pieces
  bool pieces = false;
  pmd_f16x2_op ma, gb;
  // C is singular by construction on the R > frames route (its last pivot is the null direction, 3e-12 of the mean diagonal on
  // the headline fixture) and the Cholesky step needs every other pivot positive: this is the one product of the stage whose
  // error structure decides whether the route works at all.  Measured on that fixture (scripts/debug_headline.py): two fp16
  // pieces per operand (22-23 bits) fail the factorisation or pass it with wrong factors (s off by 0.57); THREE pieces hold an
  // fp32 number exactly, six of the nine piece products (down to 2^-22; the rest is 2^-33) are exact products accumulated in
  // fp32 - sgemm's own arithmetic - and then the length of the accumulation chain decides: chunks of 2048 / 3072 inner
  // indices pass with the figures of the fp32 path, chunks of 4096 and more fail.
  // PMD_F16X2_MTGM: 6 (default) = those six products as ONE matrix product per chunk of pmd_gemm_k_chunk(k) inner indices
  // (the "concatenated" form of gemm_f16x2.hip: C is revisited once per chunk, as on the fp32 path: 34 ms against 52 at
  // config 3, s 1.18e-4 / Vt 8.5e-4 / U R 1.05e-1 against the arbiter where sgemm gives 1.11e-4 / 1.2-1.5e-3 / 5.5-6.8e-2);
  // 3 = six separate products per chunk (42 ms at config 3, but 870 against 690 ms at BASELINE config 4: C is re-read and
  // re-written six times per chunk); 2 = two pieces (breaks the fixture); 0 = sgemm.
  static int mtgm_pieces = -1;
  if (mtgm_pieces < 0) {
    const char* e = getenv("PMD_F16X2_MTGM");
    mtgm_pieces = e ? atoi(e) : 6;
    if (mtgm_pieces != 2 && mtgm_pieces != 3 && mtgm_pieces != 6) mtgm_pieces = 0;
  }
  int pm = mtgm_pieces;
  // (not where the output is small and the inner dimension huge - the many-tile workloads: 10^3 x 10^3 outputs, k = 3 10^5 -
  // there one strided-batched split-K sgemm, pmd_gemm_rm, beats 150 chunks of tiny products)
  const long out_tiles = (long)((std::min(bs, m) + 127) / 128) * ((m + 127) / 128);
  if (pm == 6 && out_tiles < 256) pm = 0;
  if (pm == 6 && pmd_f16x2_wanted(ctx, std::min(bs, m), m, rows)) {
    // PMD_F16X2_MTGM=6: the six piece products of three exact pieces per operand as ONE matrix product per accumulation chunk
    // (gemm_f16x2.hip, "concatenated" form): C is revisited once per chunk, as on the fp32 path, not six times
    const float* X[2] = {Mt, GM};
    const int xr[2] = {m, rows}, xc[2] = {rows, m};
    const long xl[2] = {ldt, ldgm};
    int ex[2] = {0, 0}, usable = 0;
    RUN(pmd_f16x2_exponents(ctx, 2, X, xr, xc, xl, ex, &usable));
    const int kc = pmd_gemm_k_chunk(rows);
    const long lda6 = 6L * pmd_round_up(kc, 8), ldb6 = pmd_round_up(m, 8);
    void* w = nullptr;
    if (usable) RUN(pmd_split_scratch(ctx, ((size_t)bs * lda6 + 6 * (size_t)kc * ldb6) * sizeof(_Float16) + 512, &w));
    const float alpha6 = ldexpf(1.f, ex[0] + ex[1]);
    if (usable && w && std::isfinite(alpha6) && alpha6 > 0.f) {
      _Float16* A6 = (_Float16*)w;
      _Float16* B6 = A6 + (size_t)bs * lda6;
      bool ok6 = true;
      for (int k0 = 0; k0 < rows && ok6; k0 += kc) {
        const int kk = std::min(kc, rows - k0);
        RUN(pmd_f16cat_b(ctx, GM + (long)k0 * ldgm, kk, m, ldgm, ex[1], B6, ldb6));
        for (int i0 = 0; i0 < m && ok6; i0 += bs) {
          const int nr = std::min(bs, m - i0);
          RUN(pmd_f16cat_a(ctx, Mt + (long)i0 * ldt + k0, nr, kk, ldt, ex[0], A6, lda6));
          int done = 0;
          RUN(pmd_f16_plain_matmul(ctx, nr, i0 + nr, 6 * kk, alpha6, A6, lda6, B6, ldb6, k0 ? 1.f : 0.f, C + (long)i0 * ldc, ldc, &done));
          if (!done) ok6 = false;
        }
        if (!ok6 && k0 > 0) return pmd_fail(ctx, PMD_ERR_BLAS, "pmd_gram_mtgm", "no hipBLASLt kernel for a later chunk");
      }
      if (ok6) return PMD_OK;
    }
  }
  if (pm == 6) pm = 0;
  if (pm && pmd_f16x2_wanted(ctx, std::min(bs, m), m, rows)) {
    const size_t na = pmd_f16x2_bytes(m, rows, pm), nb = pmd_f16x2_bytes(rows, m, pm);
    void* w = nullptr;
    RUN(pmd_split_scratch(ctx, na + nb, &w));
    const float* X[2] = {Mt, GM};
    if (w) {
    const int xr[2] = {m, rows}, xc[2] = {rows, m};
    const long xl[2] = {ldt, ldgm};
    void* buf[2] = {w, (char*)w + na};
    pmd_f16x2_op ops[2];
    int usable = 0;
    RUN(pmd_f16x2_split(ctx, 2, X, xr, xc, xl, buf, ops, &usable, pm));
    if (usable) { pieces = true; ma = ops[0]; gb = ops[1]; }
    }
  }
  for (int i0 = 0; i0 < m; i0 += bs) {
    const int nr = std::min(bs, m - i0);
    if (pieces) {
      pmd_f16x2_op a = ma;
      a.h1 += (long)i0 * ma.ld;
      a.h2 += (long)i0 * ma.ld;
      if (a.h3) a.h3 += (long)i0 * ma.ld;
      // inner dimension in chunks: one fp32 accumulation chain per chunk, of the length the fp32 path uses (the matrix-core
      // kernel otherwise carries one chain over all of k); PMD_F16X2_MTGM_KCHUNK overrides
      static int kch_env = -2;
      if (kch_env == -2) { const char* e = getenv("PMD_F16X2_MTGM_KCHUNK"); kch_env = e ? atoi(e) : -1; }
      const int kch = kch_env > 0 ? kch_env : (kch_env == 0 ? rows : pmd_gemm_k_chunk(rows));
      int done = 1;
      for (int k0 = 0; k0 < rows && done; k0 += kch) {
        const int kk = std::min(kch, rows - k0);
        pmd_f16x2_op ak = a, bk = gb;
        ak.h1 += k0; ak.h2 += k0;
        if (ak.h3) ak.h3 += k0;
        bk.h1 += (long)k0 * gb.ld; bk.h2 += (long)k0 * gb.ld;
        if (bk.h3) bk.h3 += (long)k0 * gb.ld;
        RUN(pmd_f16x2_matmul(ctx, 0, 0, nr, i0 + nr, kk, 1.f, ak, bk, k0 ? 1.f : 0.f, C + (long)i0 * ldc, ldc, &done));
      }
      if (done) continue;
      pieces = false;   // (the product below may reuse the scratch that held the pieces)
    }
    const int keep = ctx->gemm_split;
    if (!mtgm_pieces) ctx->gemm_split = 0;
    const int rc = pmd_gemm_rm(ctx, 0, 0, nr, i0 + nr, rows, 1.f, Mt + (long)i0 * ldt, ldt, GM, ldgm, 0.f, C + (long)i0 * ldc, ldc);
    ctx->gemm_split = keep;
    RUN(rc);
  }
  // the pieces of both operands are three quarters of M and GM together (27 GB at BASELINE config 4): a scratch of that
  // size goes back to the device before the eigensolver asks for its workspace
  if (pieces) RUN(pmd_split_scratch_trim(ctx, (size_t)8 << 30));
  return PMD_OK;
}

// Small orders in double precision (the policy of pmd_syevd, sytrd.hip: global-stage problems up to order 512 are factorised
// in fp64 and rounded).  The Cholesky route amplifies the factorisation error by the condition number of C, which the
// reference's rank_prune and R > frames routes drive to 1e5 ... 1e7: in fp32 the signal singular values of such draws came
// out 27 % off where NumPy's (double-precision LAPACK on fp32 data) are 2 % off (seeded fuzz, options 104 / 42).
// Orders up to which the Cholesky step runs in double precision (rocSOLVER dpotrf + dtrtri on a widened copy, results
// rounded).  PMD_CHOL_F64_MAX overrides.
static int chol_f64_max() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("PMD_CHOL_F64_MAX"); v = e ? atoi(e) : 512; }
  return v;
}
#define CHOL_F64_MAX chol_f64_max()

namespace {
__global__ void chol_widen_kernel(const float* __restrict__ src, long lds_, double* __restrict__ dst, long ldd, int n) {
  const int r = blockIdx.y;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) dst[(long)r * ldd + c] = (double)src[(long)r * lds_ + c];
}
// the factor lives in the row-major lower triangle; everything above the diagonal becomes zero
__global__ void chol_narrow_lower_kernel(const double* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, int n) {
  const int r = blockIdx.y;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x)
    dst[(long)r * ldd + c] = (c <= r) ? (float)src[(long)r * lds_ + c] : 0.f;
}
// last pivot through its absolute value (abs_last_pivot of pmd_chol_inverse): memory row m - 1 holds y = U_11^{-T} c in its first
// m - 1 entries and C_mm in the last; U_mm = sqrt(|C_mm - y^T y|)
__global__ void chol_last_pivot_kernel(double* __restrict__ U, long ld, int m, int* __restrict__ info) {
  __shared__ double red[256];
  double* row = U + (long)(m - 1) * ld;
  double s = 0.0;
  for (int i = threadIdx.x; i < m - 1; i += 256) s += row[i] * row[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double p = row[m - 1] - red[0];
    if (!(fabs(p) > 0.0)) { if (*info == 0) *info = m; }
    else row[m - 1] = sqrt(fabs(p));
  }
}
}  // namespace

// In place: C = U_c^T U_c (row-major lower triangle read) -> Et = U_c^{-T} (row-major lower, rest zeroed).
size_t pmd_chol_inverse_workspace_bytes_impl(int m) {
  return ((size_t)m + CHOL_NB) * CHOL_NB * sizeof(float) + (m <= CHOL_F64_MAX ? (size_t)m * m * sizeof(double) + 64 : 0) + 8192;
}

int pmd_chol_inverse_impl(pmd_ctx* ctx, float* C, int m, long ldc, int abs_last_pivot, int* ok_host, void* ws, size_t ws_bytes) {
  pmd_arena ar(ws, ws_bytes);
  int* info = ar.take_n<int>(4);
  float* chol_tmp = ar.take_n<float>((size_t)m * CHOL_NB);
  float* chol_linv = ar.take_n<float>((size_t)CHOL_NB * CHOL_NB);
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_chol_inverse", "workspace too small");
  *ok_host = 0;
  int hinfo = 0;
  if (m <= CHOL_F64_MAX && m >= 1 && !getenv("PMD_CHOLESKY")) {
    pmd_prof_scope prof__(ctx, "cholesky_f64");
    double* Cd = ar.take_n<double>((size_t)m * m);
    if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_chol_inverse", "workspace too small");
    PMD_HIP(ctx, hipMemsetAsync(info, 0, 2 * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(chol_widen_kernel, dim3((m + 255) / 256, m), dim3(256), 0, ctx->stream, C, ldc, Cd, (long)m, m);
    PMD_LAUNCH_CHECK(ctx, "chol_widen_kernel");
    // the row-major lower triangle is the column-major upper one: C = U^T U
    const int mf = abs_last_pivot ? m - 1 : m;
    if (mf > 0) PMD_BLAS(ctx, rocsolver_dpotrf(ctx->blas, rocblas_fill_upper, mf, Cd, m, info));
    if (abs_last_pivot) {
      if (m > 1)
        PMD_BLAS(ctx, rocblas_dtrsv(ctx->blas, rocblas_fill_upper, rocblas_operation_transpose, rocblas_diagonal_non_unit, m - 1, Cd, m,
                                    Cd + (long)(m - 1) * m, 1));
      hipLaunchKernelGGL(chol_last_pivot_kernel, dim3(1), dim3(256), 0, ctx->stream, Cd, (long)m, m, info + 1);
      PMD_LAUNCH_CHECK(ctx, "chol_last_pivot_kernel");
    }
    int h2[2] = {0, 0};
    PMD_HIP(ctx, hipMemcpyAsync(h2, info, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h2[0] != 0 || h2[1] != 0) return PMD_OK;   // not positive definite: ok_host stays 0, C is untouched
    PMD_BLAS(ctx, rocsolver_dtrtri(ctx->blas, rocblas_fill_upper, rocblas_diagonal_non_unit, m, Cd, m, info));
    PMD_HIP(ctx, hipMemcpyAsync(h2, info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    hipLaunchKernelGGL(chol_narrow_lower_kernel, dim3((m + 255) / 256, m), dim3(256), 0, ctx->stream, Cd, (long)m, C, ldc, m);
    PMD_LAUNCH_CHECK(ctx, "chol_narrow_lower_kernel");
    PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h2[0] != 0) return pmd_fail(ctx, PMD_ERR_BLAS, "rocsolver_dtrtri", "singular factor");
    *ok_host = 1;
    return PMD_OK;
  }
  {
    const char* cmode = getenv("PMD_CHOLESKY");
    if (cmode && !strcmp(cmode, "rocsolver") && !abs_last_pivot) {
      pmd_prof_scope prof__(ctx, "rocsolver_spotrf");
      PMD_BLAS(ctx, rocsolver_spotrf(ctx->blas, rocblas_fill_upper, m, C, (rocblas_int)ldc, info));
    } else {
      RUN(chol_lower_rm(ctx, m, C, ldc, info, chol_tmp, chol_linv, abs_last_pivot));
    }
  }
  PMD_HIP(ctx, hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (hinfo != 0) return PMD_OK;  // not positive definite: ok_host stays 0
  {
    pmd_prof_scope prof__(ctx, "rocsolver_strtri");
    PMD_BLAS(ctx, rocsolver_strtri(ctx->blas, rocblas_fill_upper, rocblas_diagonal_non_unit, m, C, (rocblas_int)ldc, info));
  }
  PMD_HIP(ctx, hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  // column-major upper U^{-1} == row-major lower U^{-T} = Et; clear the other triangle (old C entries)
  hipLaunchKernelGGL(tril_mask_kernel, dim3(8, m), dim3(256), 0, ctx->stream, C, ldc, m);
  PMD_LAUNCH_CHECK(ctx, "tril_mask_kernel");
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (hinfo != 0) return PMD_OK;
  *ok_host = 1;
  return PMD_OK;
}

int pmd_orthogonalize_chol_impl(pmd_ctx* ctx, const float* M, int Rc, int m, long ldm, const float* GM, long ldgm,
                                float* Et_out, long lde, int* ok_host, void* ws, size_t ws_bytes) {
  RUN(pmd_gram_mtgm_impl(ctx, M, Rc, m, ldm, GM, ldgm, Et_out, lde, ws, ws_bytes));
  return pmd_chol_inverse_impl(ctx, Et_out, m, lde, 0, ok_host, ws, ws_bytes);
}

int pmd_transpose_impl(pmd_ctx* ctx, const float* src, long lds_, int rows, int cols, float* dst, long ldd) {
  return launch_transpose(ctx, src, lds_, rows, cols, dst, ldd);
}
