// Generic-width forms of the per-tile small-matrix kernels, for sketches wider than the 64 component rows the main
// tile kernels are built on (max_components + 10 > 64, background_rank + 10 > 64; the reference's arguments are
// unbounded: decomposition.py:643-665, :59-67).  Per-tile arrays are [tile][rp][x] with rp = a multiple of 64
// (pmd_tile_rpad); the streaming contractions run as row blocks of 64 through tile_atx / tile_xbt (pmd_launch_tile_*_rp
// below), the small dense algebra through the kernels of this file:
//   wide_gram   : G[tile][slice][rp][rp] = In In^T over a slice of positions, fp64 accumulation of fp32 inputs
//   wide_eig    : symmetric eigendecomposition of the summed n x n Gram matrices (rocSOLVER dsyevd, strided batch),
//                 vectors ordered by descending eigenvalue; mode 1 = scaled by 1/sqrt(lambda) with the null rule of
//                 small_eig (lambda <= tol * lambda_max -> zero column)
//   wide_rowmix : Out[tile][c][x] = sum_c' N[tile][c'][c] In[tile][c'][x], fp64 accumulation, in-place safe
// Same numerical policy as the 64-row path: every Gram matrix behind an SVD in fp64 from the fp32
// data, eigenvectors in fp64.  This path is sized for correctness at any width, not tuned: the default arguments
// (max_components = 50, background_rank = 15) never reach it.
#include "pmd_internal.h"
#include <rocsolver/rocsolver.h>

#define RUN(call)                    \
  do {                               \
    int rc__ = (call);               \
    if (rc__ != PMD_OK) return rc__; \
  } while (0)

#define WIDE_BLAS(ctx, call)                                                                   \
  do {                                                                                         \
    rocblas_status s__ = (call);                                                               \
    if (s__ != rocblas_status_success) return pmd_fail(ctx, PMD_ERR_BLAS, #call, "rocBLAS / rocSOLVER call failed"); \
  } while (0)

int pmd_tile_rpad(int r) {
  const int l = r + 10;
  return l <= 64 ? 64 : (int)pmd_round_up(l, 64);
}

// ---------------------------------------------------------------- Gram -----------------------------------------
// one workgroup = one 64 x 64 block (bi, bj) of one tile's slice; 256 threads = 16 x 16, each a 4 x 4 sub-block
__global__ __launch_bounds__(256) void wide_gram_kernel(const float* __restrict__ In, const float* __restrict__ In2, long tile_stride,
                                                        long ld, int len, int chunk_per_slice, int rp, double* __restrict__ G,
                                                        long g_tile_stride) {
  __shared__ float sa[64][33];
  __shared__ float sb[64][33];
  const int tile = blockIdx.x, slice = blockIdx.y;
  const int nblk = rp >> 6;
  const int bi = blockIdx.z / nblk, bj = blockIdx.z - bi * nblk;
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  const float* a = In + (long)tile * tile_stride + (long)(64 * bi) * ld;
  const float* b = In2 + (long)tile * tile_stride + (long)(64 * bj) * ld;
  const int x_begin = slice * chunk_per_slice;
  const int x_end = min(len, x_begin + chunk_per_slice);
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int x0 = x_begin; x0 < x_end; x0 += 32) {
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
      const int r = i >> 5, cx = i & 31;
      const bool in = x0 + cx < x_end;
      sa[r][cx] = in ? a[(long)r * ld + x0 + cx] : 0.f;
      sb[r][cx] = in ? b[(long)r * ld + x0 + cx] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int cx = 0; cx < 32; ++cx) {
      double av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = (double)sa[4 * ti + i][cx]; bv[i] = (double)sb[4 * tj + i][cx]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fma(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  double* g = G + (long)tile * g_tile_stride + (long)slice * rp * rp;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) g[(long)(64 * bi + 4 * ti + i) * rp + 64 * bj + 4 * tj + j] = acc[i][j];
}

// G: [tile][slices][rp][rp] doubles; In2 (same strides) = the second factor of a cross Gram matrix G[c][c'] = sum_x In[c][x] In2[c'][x]
// (NULL: In itself)
int pmd_launch_wide_gram(pmd_ctx* ctx, const float* In, long tile_stride, long ld, int len, int n_tiles, int slices, int rp,
                         double* G, const float* In2) {
  pmd_prof_scope prof__(ctx, "wide_gram");
  if (n_tiles <= 0) return PMD_OK;
  if (slices < 1) slices = 1;
  int cps = (len + slices - 1) / slices;
  cps = (int)pmd_round_up(cps, 32);
  const int nblk = rp / 64;
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    hipLaunchKernelGGL(wide_gram_kernel, dim3(tn, slices, nblk * nblk), dim3(256), 0, ctx->stream, In + (long)t0 * tile_stride,
                       (In2 ? In2 : In) + (long)t0 * tile_stride, tile_stride, ld, len, cps, rp, G + (long)t0 * slices * rp * rp,
                       (long)slices * rp * rp);
    PMD_LAUNCH_CHECK(ctx, "wide_gram_kernel");
  }
  return PMD_OK;
}

// ---------------------------------------------------------------- eigendecomposition ---------------------------
// A[tile] (n x n, ld n) = symmetrised sum over the slices of the leading n x n block of G[tile]
__global__ void wide_sym_sum_kernel(const double* __restrict__ G, long g_tile_stride, int slices, int rp, int n,
                                    double* __restrict__ A) {
  const int tile = blockIdx.y;
  const double* g = G + (long)tile * g_tile_stride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += gridDim.x * blockDim.x) {
    const int r = i / n, c = i - r * n;
    double s = 0.0;
    for (int k = 0; k < slices; ++k) s += g[(long)k * rp * rp + (long)r * rp + c] + g[(long)k * rp * rp + (long)c * rp + r];
    A[(long)tile * n * n + i] = 0.5 * s;
  }
}

// rocSOLVER returns ascending eigenvalues w[tile][n] and column-major eigenvectors A[tile][col j][row i].
// Nout[tile][c'][c] (ld rp) = component c' of the eigenvector with the c-th LARGEST eigenvalue; entries outside n x n zero.
__global__ void wide_eig_finish_kernel(const double* __restrict__ A, const double* __restrict__ w, int n, int rp, int mode,
                                       double tol, double* __restrict__ Nout, double* __restrict__ lam_out) {
  const int tile = blockIdx.y;
  const double* a = A + (long)tile * n * n;
  const double* wt = w + (long)tile * n;
  const double lmax = wt[n - 1];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rp * rp; i += gridDim.x * blockDim.x) {
    const int r = i / rp, c = i - r * rp;
    double v = 0.0;
    if (r < n && c < n) {
      const int src = n - 1 - c;
      v = a[(long)src * n + r];
      if (mode == 1) {
        const double lam = wt[src];
        v = (lam > tol * lmax && lam > 0.0) ? v / sqrt(lam) : 0.0;
      }
    }
    Nout[(long)tile * rp * rp + i] = v;
  }
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < rp; c += blockDim.x) lam_out[(long)tile * rp + c] = (c < n) ? wt[n - 1 - c] : 0.0;
}

size_t pmd_wide_eig_workspace_bytes(int n, int n_tiles) {
  return (size_t)n_tiles * n * n * sizeof(double) + 2 * (size_t)n_tiles * (n + 2) * sizeof(double) + (size_t)n_tiles * sizeof(int) + 4096;
}

// G: [tile][slices][rp][rp]; Nout: [tile][rp][rp]; lam_out: [tile][rp]; ws: pmd_wide_eig_workspace_bytes(n, n_tiles)
int pmd_launch_wide_eig(pmd_ctx* ctx, const double* G, int slices, int rp, int n, int mode, double tol, double* Nout,
                        double* lam_out, int n_tiles, void* ws, size_t ws_bytes) {
  pmd_prof_scope prof__(ctx, "wide_eig");
  if (n_tiles <= 0) return PMD_OK;
  if (n < 1 || n > rp) return pmd_fail(ctx, PMD_ERR_ARG, "wide_eig", "order out of range");
  pmd_arena ar(ws, ws_bytes);
  double* A = ar.take_n<double>((size_t)n_tiles * n * n);
  double* w = ar.take_n<double>((size_t)n_tiles * n);
  double* e = ar.take_n<double>((size_t)n_tiles * (n + 2));
  int* info = ar.take_n<int>(n_tiles);
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "wide_eig", "workspace too small");
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    hipLaunchKernelGGL(wide_sym_sum_kernel, dim3(16, tn), dim3(256), 0, ctx->stream, G + (long)t0 * slices * rp * rp,
                       (long)slices * rp * rp, slices, rp, n, A + (long)t0 * n * n);
    PMD_LAUNCH_CHECK(ctx, "wide_sym_sum_kernel");
  }
  // rocSOLVER's batched Jacobi solver (dsyevj; PMD_WIDE_EIG=syevd selects the divide-and-conquer one, which runs the
  // problems of a batch one after the other: 640 us per 60 x 60 problem measured, 10 s for a 65 000-problem batch)
  static int use_syevd = -1;
  if (use_syevd < 0) { const char* m = getenv("PMD_WIDE_EIG"); use_syevd = (m && !strcmp(m, "syevd")) ? 1 : 0; }
  if (use_syevd) {
    WIDE_BLAS(ctx, rocsolver_dsyevd_strided_batched(ctx->blas, rocblas_evect_original, rocblas_fill_lower, n, A, n, (rocblas_stride)n * n, w,
                                                    (rocblas_stride)n, e, (rocblas_stride)n, info, n_tiles));
  } else {
    // residual / sweep counts go to the (unused) off-diagonal scratch `e` and its int view
    double* resid = e;
    rocblas_int* nsweeps = reinterpret_cast<rocblas_int*>(e + n_tiles);
    WIDE_BLAS(ctx, rocsolver_dsyevj_strided_batched(ctx->blas, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_lower, n, A, n,
                                                    (rocblas_stride)n * n, 0.0, resid, 100, nsweeps, w, (rocblas_stride)n, info, n_tiles));
  }
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    hipLaunchKernelGGL(wide_eig_finish_kernel, dim3(16, tn), dim3(256), 0, ctx->stream, A + (long)t0 * n * n, w + (long)t0 * n, n, rp,
                       mode, tol, Nout + (long)t0 * rp * rp, lam_out + (long)t0 * rp);
    PMD_LAUNCH_CHECK(ctx, "wide_eig_finish_kernel");
  }
  return PMD_OK;
}

// ---------------------------------------------------------------- row mixing -----------------------------------
// One workgroup = one tile x 32 positions.  The n_in input rows of those positions are staged in LDS before anything is
// stored (in-place safe); thread (x = tid & 31, g = tid >> 5) forms outputs c = g, g + 8, ... one after the other.
// Rows c in [n_out, rp) are zeroed.
__global__ __launch_bounds__(256) void wide_rowmix_kernel(const float* __restrict__ In, long in_tile_stride, long ld_in,
                                                          const double* __restrict__ N, long n_tile_stride, int rp,
                                                          int n_in, int n_out, float* __restrict__ Out,
                                                          long out_tile_stride, long ld_out, int len) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [n_in][33]
  const int tile = blockIdx.y;
  const int x = threadIdx.x & 31, g = threadIdx.x >> 5;
  const float* in = In + (long)tile * in_tile_stride;
  float* out = Out + (long)tile * out_tile_stride;
  const double* nm = N + (long)tile * n_tile_stride;
  for (int x0 = blockIdx.x * 32; x0 < len; x0 += gridDim.x * 32) {
    for (int i = threadIdx.x; i < n_in * 32; i += 256) {
      const int r = i >> 5, cx = i & 31;
      xs[r * 33 + cx] = (x0 + cx < len) ? in[(long)r * ld_in + x0 + cx] : 0.f;
    }
    __syncthreads();
    for (int c = g; c < rp; c += 8) {
      double acc = 0.0;
      if (c < n_out)
        for (int cp = 0; cp < n_in; ++cp) acc = fma(nm[(long)cp * rp + c], (double)xs[cp * 33 + x], acc);
      if (x0 + x < len) out[(long)c * ld_out + x0 + x] = (float)acc;
    }
    __syncthreads();
  }
}

// N: [tile][rp][rp] doubles (n_tile_stride = 0 shares one matrix)
int pmd_launch_wide_rowmix(pmd_ctx* ctx, const float* In, long in_tile_stride, long ld_in, const double* N, long n_tile_stride,
                           int rp, int n_in, int n_out, float* Out, long out_tile_stride, long ld_out, int len, int n_tiles) {
  pmd_prof_scope prof__(ctx, "wide_rowmix");
  if (n_tiles <= 0 || len <= 0) return PMD_OK;
  if (n_in > rp || n_out > rp) return pmd_fail(ctx, PMD_ERR_ARG, "wide_rowmix", "more rows than the padded height");
  int bx = (len + 31) / 32;
  if (bx > 64) bx = 64;
  const size_t lds = (size_t)(n_in > 0 ? n_in : 1) * 33 * sizeof(float);
  if (lds > 160 * 1024) return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "wide_rowmix", "more input rows than LDS holds");
  PMD_HIP(ctx, hipFuncSetAttribute((const void*)wide_rowmix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    hipLaunchKernelGGL(wide_rowmix_kernel, dim3(bx, tn), dim3(256), lds, ctx->stream, In + (long)t0 * in_tile_stride, in_tile_stride,
                       ld_in, N + (long)t0 * n_tile_stride, n_tile_stride, rp, n_in, n_out, Out + (long)t0 * out_tile_stride,
                       out_tile_stride, ld_out, len);
    PMD_LAUNCH_CHECK(ctx, "wide_rowmix_kernel");
  }
  return PMD_OK;
}

// ---------------------------------------------------------------- streaming contractions in row blocks of 64 ----
// A: [tile][rp][a_ld], Out: [tile][rp][ldo]; rows [0, nrows) are computed (whole blocks of 64), the tile strides are the
// caller's (rp * ld)
int pmd_launch_tile_atx_rp(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                           const float* A, long a_tile_stride, int a_ld, float* Out, long out_tile_stride, long ldo, int n_tiles,
                           int T, int slices, int nrows) {
  for (int h = 0; 64 * h < nrows; ++h)
    RUN(pmd_launch_tile_atx(ctx, X, ldx, pix, pix_stride, row0_stride, d, A + (long)64 * h * a_ld, a_tile_stride, a_ld,
                            Out + (long)64 * h * ldo, out_tile_stride, ldo, n_tiles, T, slices));
  return PMD_OK;
}

// B: [tile][rp][ldb], S: [tile][slice][rp][s_ld] (s_slice_stride = rp * s_ld)
int pmd_launch_tile_xbt_rp(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                           const float* B, long b_tile_stride, long ldb, float* S, long s_tile_stride, long s_slice_stride, int s_ld,
                           int n_tiles, int T, int slices, int nrows) {
  for (int h = 0; 64 * h < nrows; ++h)
    RUN(pmd_launch_tile_xbt(ctx, X, ldx, pix, pix_stride, row0_stride, d, B + (long)64 * h * ldb, b_tile_stride, ldb,
                            S + (long)64 * h * s_ld, s_tile_stride, s_slice_stride, s_ld, n_tiles, T, slices));
  return PMD_OK;
}
