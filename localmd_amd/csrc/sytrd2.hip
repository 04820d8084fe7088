// Two-stage tridiagonalisation of the dense symmetric eigenproblem of the global stage (decomposition.py:984, :1090:
// jnp.linalg.svd(..., hermitian=True)), the alternative to the one-stage reduction of sytrd.hip whose symv re-reads the
// trailing triangle for every column (670 GB at n = 10^4, 5.3 TB at n = 2 10^4).
//
//   stage 1  A = Q1 B Q1^T   dense -> band (half bandwidth SB = 64).  Per 64-column panel: CholeskyQR2 of the sub-band block
//                            (fp64 Gram / Cholesky, kernels of the tile pipeline) + Householder reconstruction
//                            (Q - [S; 0] = L U with the signs S chosen during the elimination: V = L is unit lower
//                            trapezoidal, T = -U S L1^{-T} upper triangular, H = I - V T V^T maps [S R; 0] to the panel),
//                            then the two-sided update of the trailing block as rocBLAS GEMMs:
//                            A22 -= V W^T + W V^T,  W = Y - V (T^T V^T Y) / 2,  Y = A22 V T.
//   stage 2  B = Q2 T Q2^T   band -> tridiagonal by bulge chasing (sweep s eliminates column s below the first subdiagonal,
//                            task k of the sweep acts on rows s + 1 + k SB .. + SB); task (s, k) needs (s, k - 1) and
//                            (s - 1, k + 1).  A workgroup runs SW consecutive sweeps in lockstep (one step per barrier, sweep
//                            i two tasks behind sweep i - 1); consecutive workgroups are kept apart by launches.
//   T = Z L Z^T              rocSOLVER sstedc
//   E = Q1 (Q2 Z)            Q2: the length-64 reflectors of stage 2 applied to 32-vector slabs (one workgroup per slab, no
//                            synchronisation between slabs); Q1: the compact-WY path of sytrd.hip with reflector offset 64.
//
// Conventions as in sytrd.hip: memory row c of the buffer holds column c of the symmetric matrix ("positions" r along the
// row); on entry only positions r >= c are read.
#include "pmd_internal.h"
#include <rocsolver/rocsolver.h>
#include <algorithm>
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

namespace {

constexpr int SB = 64;   // half bandwidth after stage 1 = panel width = reflector length of stage 2

// A[r][c] = A[c][r] for r > c: the GEMMs of stage 1 work on the full symmetric matrix
__global__ void symmetrize_kernel(float* __restrict__ A, long lda, int n) {
  __shared__ float t[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;   // tile (bi, bj) of the upper part, bj >= bi
  if (bj < bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    t[r][tx] = (i < n && j < n) ? A[(long)i * lda + j] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = bj * 32 + r, j = bi * 32 + tx;   // transposed tile
    if (i < n && j < n && i > j) A[(long)i * lda + j] = t[tx][r];
  }
}

// rows 0..w-1 of the panel (memory rows j0.., positions r0..r0+m-1) -> Pbuf[c][i]; rows w..63 and the padding are zeroed
__global__ void panel_copy_in_kernel(const float* __restrict__ A, long lda, int j0, int r0, int w, int m, float* __restrict__ Pbuf,
                                     long ldp) {
  const int c = blockIdx.y;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ldp; i += (long)gridDim.x * blockDim.x)
    Pbuf[(long)c * ldp + i] = (c < w && i < m) ? A[(long)(j0 + c) * lda + r0 + i] : 0.f;
}

__global__ void panel_copy_out_kernel(const float* __restrict__ Pbuf, long ldp, int j0, int r0, int w, int m, float* __restrict__ A,
                                      long lda) {
  const int c = blockIdx.y;
  if (c >= w) return;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long)gridDim.x * blockDim.x)
    A[(long)(j0 + c) * lda + r0 + i] = Pbuf[(long)c * ldp + i];
}

// Cholesky of the w x w Gram matrix (sum of `slices` partials, symmetrised) in fp64: G = R^T R.  Rout = R (upper, row-major
// [64][64]), Nout[c'][c] = (R^{-1})[c'][c] (the layout tile_rowmix expects).  A non-positive pivot sets *flag (the panel is
// rank deficient: the caller falls back to the one-stage reduction).
__global__ __launch_bounds__(256) void panel_chol_kernel(const double* __restrict__ G, int slices, int w, double* __restrict__ Rout,
                                                         double* __restrict__ Nout, int* __restrict__ flag) {
  __shared__ double R[64][65];
  __shared__ double Ri[64][65];
  __shared__ int bad_s;
  const int tid = threadIdx.x;
  const int t = tid & 63, q = tid >> 6;          // column t, quarter q of the rows / of an inner sum
  if (tid == 0) bad_s = 0;
  for (int i = q; i < 64; i += 4) {
    double s = 0.0;
    if (i < w && t < w)
      for (int k = 0; k < slices; ++k) s += G[(long)k * 4096 + i * 64 + t] + G[(long)k * 4096 + t * 64 + i];
    R[i][t] = 0.5 * s;
    Ri[i][t] = 0.0;
  }
  __syncthreads();
  double dmax = 0.0;
  for (int i = 0; i < w; ++i) dmax = fmax(dmax, R[i][i]);
  for (int k = 0; k < w; ++k) {
    const double piv = R[k][k];
    const bool bad = !(piv > 1e-13 * dmax);
    const double rkk = bad ? 1.0 : sqrt(piv);
    double rkt = 0.0;
    if (t >= k && t < w) rkt = bad ? ((t == k) ? 1.0 : 0.0) : R[k][t] / rkk;
    __syncthreads();
    if (q == 0 && t >= k && t < w) R[k][t] = rkt;
    if (tid == 0 && bad) bad_s = 1;
    __syncthreads();
    if (!bad && t > k && t < w)
      for (int i = k + 1 + q; i <= t; i += 4) R[i][t] -= R[k][i] * rkt;
    __syncthreads();
  }
  // R^{-1} (upper): column t by back substitution, the inner sum split over the four threads of the column
  // (tid = 64 q + t: the four threads of a column sit in four different waves, so the partial sums meet in LDS)
  __shared__ double part[4][64];
  for (int i = w - 1; i >= 0; --i) {
    double s = 0.0;
    if (t < w && i <= t)
      for (int j = i + 1 + q; j <= t; j += 4) s += R[i][j] * Ri[j][t];
    part[q][t] = s;
    __syncthreads();
    if (q == 0 && t < w && i <= t) Ri[i][t] = (((i == t) ? 1.0 : 0.0) - (part[0][t] + part[1][t] + part[2][t] + part[3][t])) / R[i][i];
    __syncthreads();
  }
  for (int i = q; i < 64; i += 4) {
    Rout[i * 64 + t] = (i < w && t < w && t >= i) ? R[i][t] : 0.0;
    Nout[i * 64 + t] = (i < w && t < w) ? Ri[i][t] : 0.0;
  }
  if (tid == 0 && bad_s) *flag = 1;
}

// Householder reconstruction (Ballard, Demmel, Grigori, Jacquelin, Nguyen, Solomonik 2014) of the panel's orthonormal factor.
// Pbuf[c][i], i < w: top block Q1[i][c] of Q (m x w, orthonormal columns).  Computes S (signs), Q1 - S = L1 U (no pivoting:
// S_ii = -sign of the current diagonal entry makes every pivot >= 1 in magnitude), then
//   Ninv[c'][c] = (U^{-1})[c'][c]            (tile_rowmix turns the rows below the top block into V2 = Q2 U^{-1})
//   T = -U S L1^{-T}                          (upper triangular; H = I - V T V^T, V = [L1; V2])
//   Rs = S R2 R1                              (the panel becomes [Rs; 0])
// and writes the top block of the panel as LAPACK stores it: Rs on and above the diagonal, L1 below; tau[c] = T[c][c].
__global__ __launch_bounds__(256) void panel_hr_kernel(float* __restrict__ Pbuf, long ldp, int w, const double* __restrict__ R1,
                                                       const double* __restrict__ R2, double* __restrict__ Ninv,
                                                       float* __restrict__ T32, float* __restrict__ tau_out) {
  __shared__ double M[64][65];    // Q1 - S -> L1 (strictly lower) and U (upper)
  __shared__ double X[64][65];    // U^{-1}, then L1^{-T}
  __shared__ double Y[64][65];    // R = R2 R1, then T
  __shared__ double S[64];
  const int tid = threadIdx.x;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int i = e >> 6, c = e & 63;
    M[i][c] = (i < w && c < w) ? (double)Pbuf[(long)c * ldp + i] : 0.0;
    X[i][c] = 0.0;
    double r = 0.0;
    if (i < w && c < w)
      for (int k = i; k <= c; ++k) r += R2[i * 64 + k] * R1[k * 64 + c];   // upper x upper
    Y[i][c] = r;
  }
  __syncthreads();
  // LU without pivoting, sign chosen per step
  for (int i = 0; i < w; ++i) {
    if (tid == 0) {
      const double sgn = (M[i][i] >= 0.0) ? -1.0 : 1.0;
      S[i] = sgn;
      M[i][i] -= sgn;
    }
    __syncthreads();
    const double piv = M[i][i];
    for (int r = i + 1 + tid; r < w; r += 256) M[r][i] /= piv;
    __syncthreads();
    for (int e = tid; e < (w - i - 1) * (w - i - 1); e += 256) {
      const int r = i + 1 + e / (w - i - 1), c = i + 1 + e % (w - i - 1);
      M[r][c] -= M[r][i] * M[i][c];
    }
    __syncthreads();
  }
  // X = U^{-1} (upper): thread c solves column c
  if (tid < w) {
    const int c = tid;
    for (int i = c; i >= 0; --i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int j = i + 1; j <= c; ++j) s -= M[i][j] * X[j][c];
      X[i][c] = s / M[i][i];
    }
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += 256) {
    const int i = e >> 6, c = e & 63;
    Ninv[e] = (i < w && c < w) ? X[i][c] : 0.0;
  }
  __syncthreads();
  // the top block of the panel: Rs = S R above / on the diagonal, L1 below
  for (int e = tid; e < w * w; e += 256) {
    const int i = e / w, c = e % w;
    Pbuf[(long)c * ldp + i] = (float)((i <= c) ? S[i] * Y[i][c] : M[i][c]);
  }
  __syncthreads();
  // X = L1^{-T}: L1^T is unit upper triangular with entries L1^T[i][j] = M[j][i], j > i; thread c solves column c
  if (tid < w) {
    const int c = tid;
    for (int i = 0; i < w; ++i) X[i][c] = 0.0;
    for (int i = c; i >= 0; --i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int j = i + 1; j <= c; ++j) s -= M[j][i] * X[j][c];
      X[i][c] = s;
    }
  }
  __syncthreads();
  // T = -U S L1^{-T}
  for (int e = tid; e < 64 * 64; e += 256) {
    const int i = e >> 6, c = e & 63;
    double tsum = 0.0;
    if (i < w && c < w && c >= i)
      for (int k = i; k <= c; ++k) tsum -= M[i][k] * S[k] * X[k][c];
    T32[e] = (float)tsum;
    if (i == c && i < w) tau_out[i] = (float)tsum;
  }
}

// Pbuf top block -> clean V: zeros above the diagonal, one on it (in place; the panel has been written back before)
__global__ void panel_clean_v_kernel(float* __restrict__ Pbuf, long ldp, int w) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < w * w; e += gridDim.x * blockDim.x) {
    const int i = e / w, c = e % w;
    if (i < c) Pbuf[(long)c * ldp + i] = 0.f;
    else if (i == c) Pbuf[(long)c * ldp + i] = 1.f;
  }
}

// Wt = Yt - 0.5 K^T Vt as St1 = [Vt; Wt], St2 = [Wt; Vt] needs Wt explicitly: Wt[c][i] = Yt[c][i] - 0.5 sum_c' K[c'][c] Vt[c'][i]
__global__ __launch_bounds__(256) void panel_w_kernel(const float* __restrict__ Vt, const float* __restrict__ Yt, long ld, int w, int m,
                                                      const float* __restrict__ K, float* __restrict__ St1, float* __restrict__ St2,
                                                      long lds_) {
  __shared__ float k_s[64][65];
  for (int e = threadIdx.x; e < 64 * 64; e += 256) k_s[e >> 6][e & 63] = ((e >> 6) < w && (e & 63) < w) ? K[e] : 0.f;
  __syncthreads();
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  float v[64];   // constant trip counts: the array stays in registers (rows >= w of Vt are zero)
#pragma unroll
  for (int c = 0; c < 64; ++c) v[c] = Vt[(long)c * ld + i];
#pragma unroll 4
  for (int c = 0; c < 64; ++c) {
    if (c >= w) break;
    float acc = 0.f;
#pragma unroll
    for (int cp = 0; cp < 64; ++cp) acc = fmaf(k_s[cp][c], v[cp], acc);
    const float wv = Yt[(long)c * ld + i] - 0.5f * acc;
    St1[(long)c * lds_ + i] = v[c];
    St1[(long)(w + c) * lds_ + i] = wv;
    St2[(long)c * lds_ + i] = wv;
    St2[(long)(w + c) * lds_ + i] = v[c];
  }
}

}  // namespace

size_t pmd_sy2sb_workspace_bytes_impl(int n) {
  const size_t n4 = (size_t)pmd_round_up(n, 64) + 64;
  return (64 * n4 * 3 + 128 * n4 * 2) * sizeof(float) + (64 + 4) * 4096 * sizeof(double) + 4 * 4096 * sizeof(float) + 65536;
}

// Stage 1.  A: n x n, lda % 4 == 0; on entry positions r >= c of memory row c are valid; on exit the band (positions
// c .. c + 64 of memory row c) holds B, the positions beyond it the reflectors of Q1 (unit entry of reflector c at position
// c + 64), tau1[c] their scalars.  *flag_host != 0: a panel was numerically rank deficient, the result is not usable.
int pmd_sy2sb_impl(pmd_ctx* ctx, int n, float* A, long lda, float* tau1, int* flag_host, void* ws, size_t ws_bytes) {
  pmd_prof_scope prof__(ctx, "sy2sb");
  *flag_host = 0;
  pmd_arena ar(ws, ws_bytes);
  const long ldp = pmd_round_up(n, 64) + 64;
  float* Pbuf = ar.take_n<float>(64 * ldp);
  float* Xt = ar.take_n<float>(64 * ldp);
  float* Yt = ar.take_n<float>(64 * ldp);
  float* St1 = ar.take_n<float>(128 * ldp);
  float* St2 = ar.take_n<float>(128 * ldp);
  const int GS = 64;   // slices of the panel Gram matrix
  double* gpart = ar.take_n<double>((size_t)GS * 4096);
  double* R1 = ar.take_n<double>(4096);
  double* R2 = ar.take_n<double>(4096);
  double* N1 = ar.take_n<double>(4096);
  double* Ninv = ar.take_n<double>(4096);
  float* T32 = ar.take_n<float>(4096);
  float* Kp = ar.take_n<float>(4096);
  float* Km = ar.take_n<float>(4096);
  int* flag = ar.take_n<int>(4);
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_sy2sb", "workspace too small");
  hipStream_t st = ctx->stream;
  PMD_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(int), st));
  PMD_HIP(ctx, hipMemsetAsync(tau1, 0, (size_t)n * sizeof(float), st));
  {
    const int nt = (n + 31) / 32;
    hipLaunchKernelGGL(symmetrize_kernel, dim3(nt, nt), dim3(256), 0, st, A, lda, n);
    PMD_LAUNCH_CHECK(ctx, "symmetrize_kernel");
  }
  for (int j0 = 0; j0 + SB < n; j0 += SB) {
    const int r0 = j0 + SB;
    const int m = n - r0;
    const int w = std::min(SB, m);
    if (m < 2 && w < 1) break;
    const unsigned bx = (unsigned)std::min<long>((ldp + 255) / 256, 1024);
    hipLaunchKernelGGL(panel_copy_in_kernel, dim3(bx, 64), dim3(256), 0, st, A, lda, j0, r0, w, m, Pbuf, ldp);
    PMD_LAUNCH_CHECK(ctx, "panel_copy_in_kernel");
    // CholeskyQR2: Q = P R1^{-1} R2^{-1}
    for (int pass = 0; pass < 2; ++pass) {
      const int slices = std::max(1, std::min(GS, m / 256));
      RUN(pmd_launch_tile_gram(ctx, Pbuf, 64 * ldp, ldp, m, 1, slices, gpart));
      hipLaunchKernelGGL(panel_chol_kernel, dim3(1), dim3(256), 0, st, gpart, slices, w, pass == 0 ? R1 : R2, N1, flag);
      PMD_LAUNCH_CHECK(ctx, "panel_chol_kernel");
      RUN(pmd_launch_tile_rowmix(ctx, Pbuf, 64 * ldp, ldp, N1, 4096, w, w, Pbuf, 64 * ldp, ldp, m, 1));
    }
    hipLaunchKernelGGL(panel_hr_kernel, dim3(1), dim3(256), 0, st, Pbuf, ldp, w, R1, R2, Ninv, T32, tau1 + j0);
    PMD_LAUNCH_CHECK(ctx, "panel_hr_kernel");
    if (m > w) RUN(pmd_launch_tile_rowmix(ctx, Pbuf + w, 64 * ldp, ldp, Ninv, 4096, w, w, Pbuf + w, 64 * ldp, ldp, m - w, 1));
    hipLaunchKernelGGL(panel_copy_out_kernel, dim3(bx, 64), dim3(256), 0, st, Pbuf, ldp, j0, r0, w, m, A, lda);
    PMD_LAUNCH_CHECK(ctx, "panel_copy_out_kernel");
    hipLaunchKernelGGL(panel_clean_v_kernel, dim3(16), dim3(256), 0, st, Pbuf, ldp, w);
    PMD_LAUNCH_CHECK(ctx, "panel_clean_v_kernel");
    if (w < SB) {
      // the last, narrow panel (w = m < 64 columns): H also acts on the in-band entries of the columns j0 + w .. r0 - 1 below
      // row r0 (a full panel has none): rows c of that block, as stored, become  a^T - (a^T V) T V^T
      const int nr = SB - w;
      float* Rt = A + (long)(j0 + w) * lda + r0;   // nr x m, row-major, ld lda
      RUN(pmd_gemm_rm(ctx, 0, 1, nr, w, m, 1.f, Rt, lda, Pbuf, ldp, 0.f, Xt, ldp));
      RUN(pmd_gemm_rm(ctx, 0, 0, nr, w, w, 1.f, Xt, ldp, T32, 64, 0.f, Yt, ldp));
      RUN(pmd_gemm_rm(ctx, 0, 0, nr, m, w, -1.f, Yt, ldp, Pbuf, ldp, 1.f, Rt, lda));
    }
    if (m <= 1) continue;   // a 1 x 1 trailing block is invariant under H = +-1
    float* A22 = A + (long)r0 * lda + r0;
    // Xt = Vt A22 (w x m);  Yt = T^T Xt;  Kp = Vt Yt^T;  K = T^T Kp;  W = Y - V K / 2;  A22 -= [V W][W V]^T
    RUN(pmd_gemm_rm(ctx, 0, 0, w, m, m, 1.f, Pbuf, ldp, A22, lda, 0.f, Xt, ldp));
    RUN(pmd_gemm_rm(ctx, 1, 0, w, m, w, 1.f, T32, 64, Xt, ldp, 0.f, Yt, ldp));
    RUN(pmd_gemm_rm(ctx, 0, 1, w, w, m, 1.f, Pbuf, ldp, Yt, ldp, 0.f, Kp, 64));
    RUN(pmd_gemm_rm(ctx, 1, 0, w, w, w, 1.f, T32, 64, Kp, 64, 0.f, Km, 64));
    hipLaunchKernelGGL(panel_w_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, Pbuf, Yt, ldp, w, m, Km, St1, St2, ldp);
    PMD_LAUNCH_CHECK(ctx, "panel_w_kernel");
    RUN(pmd_gemm_rm(ctx, 1, 0, m, m, 2 * w, -1.f, St1, ldp, St2, ldp, 1.f, A22, lda));
  }
  PMD_HIP(ctx, hipMemcpyAsync(flag_host, flag, sizeof(int), hipMemcpyDeviceToHost, st));
  PMD_HIP(ctx, hipStreamSynchronize(st));
  return PMD_OK;
}

// =============================================================================================
// Stage 2: band -> tridiagonal by bulge chasing (oracle of the task structure: tests/two_stage_ref.py).
// Band storage AB[j][d] = B(j + d, j), d < LDB = 2 SB (the bulges reach half bandwidth 2 SB - 1).
// Task (s, k): rows r_k = [r0, r0 + L), r0 = s + 1 + k SB, L = min(SB, n - r0).
//   k = 0: reflector from x = B(r_0, s);   k >= 1: the block Bm = B(r_k, r_{k-1}) first takes the previous reflector of the
//   sweep from the right, then its first column gives the new reflector, which is applied to the rest of Bm from the left;
//   in both cases the diagonal block D = B(r_k, r_k) is updated from both sides.  The reflector (length SB, zero padded) goes
//   to V2[k][s][.] / tau2[k][s] for the back-transformation.
// Scheduling: task (s, k) runs at step t = 2 s + k (needs (s, k - 1) and (s - 1, k + 1), both at t - 1).  A workgroup owns
// SW consecutive sweeps (one 256-thread team each) and walks the steps of its window with one barrier per phase; workgroup g
// runs window q = L - g in launch L, so whatever its first sweep needs from workgroup g - 1 was finished by an earlier launch.
// =============================================================================================
namespace {

constexpr int SW = 4;          // sweeps per workgroup
constexpr int LDB = 2 * SB;    // band storage row length

__global__ void band_extract_kernel(const float* __restrict__ A, long lda, int n, float* __restrict__ AB) {
  const int j = blockIdx.x;
  for (int d = threadIdx.x; d < LDB; d += blockDim.x) AB[(long)j * LDB + d] = (d <= SB && j + d < n) ? A[(long)j * lda + j + d] : 0.f;
}

__global__ void band_diag_kernel(const float* __restrict__ AB, int n, float* __restrict__ d, float* __restrict__ e) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  d[j] = AB[(long)j * LDB];
  if (j < n - 1) e[j] = AB[(long)j * LDB + 1];
}

// team-wide sum of one value per thread of the 256-thread team (4 waves): returns the total to every thread of the team.
// red: SW x 4 floats of shared memory; two barriers (the whole workgroup executes it in lockstep)
__device__ __forceinline__ float team_sum(float v, float* red, int team, int tt) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((tt & 63) == 0) red[team * 4 + (tt >> 6)] = v;
  __syncthreads();
  return red[team * 4] + red[team * 4 + 1] + red[team * 4 + 2] + red[team * 4 + 3];
}

__global__ __launch_bounds__(SW * 256) void sb2st_kernel(float* __restrict__ AB, int n, int launch, int h, int g_lo,
                                                         float* __restrict__ V2, float* __restrict__ tau2) {
  __shared__ float Bs[SW][SB][SB + 1];
  __shared__ float Ds[SW][SB][SB + 1];
  __shared__ float vcur[SW][SB], vprev[SW][SB], wv[SW][SB], zv[SW][SB];
  __shared__ float sc[SW][4];      // tau_prev, tau, beta, alpha
  __shared__ float red[SW * 4];
  const int g = g_lo + blockIdx.x;
  const int q = launch - g;
  const int team = threadIdx.x >> 8, tt = threadIdx.x & 255;
  const int i = tt >> 2, qd = tt & 3;          // row i, column quarter qd (columns qd * 16 .. + 15)
  const int s = g * SW + team;
  if (q < 0) return;
  // the previous reflector of the sweep (task k - 1 ran in the previous step, possibly in the previous launch)
  {
    const int k_first = q * h - 2 * s;
    float tp = 0.f;
    if (s < n - 2 && k_first >= 1 && s + 1 + (long)(k_first - 1) * SB < n) {   // (a window may start behind the sweep's last task)
      const long idx = (long)(k_first - 1) * n + s;
      if (tt < SB) vprev[team][tt] = V2[idx * SB + tt];
      tp = tau2[idx];
    } else if (tt < SB) vprev[team][tt] = 0.f;
    if (tt == 0) sc[team][0] = tp;
  }
  __syncthreads();
  for (int t = q * h; t < (q + 1) * h; ++t) {
    const int k = t - 2 * s;
    const int r0 = s + 1 + k * SB;
    const bool exists = s < n - 2 && k >= 0 && r0 < n && !(k == 0 && n - r0 < 2);
    const int L = exists ? min(SB, n - r0) : 0;
    const int c0 = r0 - SB;
    // ---- load Bm (k >= 1) and D
    if (exists) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int j = qd * 16 + u;
        float bv = 0.f, dv = 0.f;
        if (i < L) {
          if (k >= 1) bv = AB[(long)(c0 + j) * LDB + SB + i - j];
          if (j < L) dv = (i >= j) ? AB[(long)(r0 + j) * LDB + i - j] : AB[(long)(r0 + i) * LDB + j - i];
        }
        Bs[team][i][j] = bv;
        Ds[team][i][j] = dv;
      }
      if (k == 0 && tt < SB) Bs[team][tt][0] = (tt < L) ? AB[(long)s * LDB + 1 + tt] : 0.f;   // x = B(r_0, s) as column 0
    }
    __syncthreads();
    // ---- (a) Bm <- Bm (I - tau_prev v_prev v_prev^T)
    float part = 0.f;
    if (exists && k >= 1) {
#pragma unroll
      for (int u = 0; u < 16; ++u) part = fmaf(Bs[team][i][qd * 16 + u], vprev[team][qd * 16 + u], part);
    }
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    if (exists && k >= 1) {
      const float f = sc[team][0] * part;
#pragma unroll
      for (int u = 0; u < 16; ++u) Bs[team][i][qd * 16 + u] -= f * vprev[team][qd * 16 + u];
    }
    __syncthreads();
    // ---- (b) reflector from column 0 (rows 0 .. L - 1)
    const float x = (exists && tt < L && tt >= 1) ? Bs[team][tt][0] : 0.f;
    const float xn2 = team_sum(x * x, red, team, tt);
    if (tt == 0) {
      float tau = 0.f, beta = 0.f, scale = 0.f;
      if (exists && L >= 2) {
        const float alpha = Bs[team][0][0];
        beta = alpha;
        if (xn2 > 0.f) {
          beta = -copysignf(sqrtf(alpha * alpha + xn2), alpha);
          tau = (beta - alpha) / beta;
          scale = 1.f / (alpha - beta);
        }
      } else if (exists) {
        beta = Bs[team][0][0];
      }
      sc[team][1] = tau; sc[team][2] = beta; sc[team][3] = scale;
    }
    __syncthreads();
    const float tau = sc[team][1];
    if (tt < SB) vcur[team][tt] = (exists && tt < L) ? ((tt == 0) ? 1.f : Bs[team][tt][0] * sc[team][3]) : 0.f;
    __syncthreads();
    // ---- (c) rest of Bm from the left: z[j] = sum_i v[i] Bm[i][j]; thread (j = i, quarter of the rows = qd)
    if (exists && k >= 1 && tau != 0.f) {
      float zp = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) zp = fmaf(vcur[team][qd * 16 + u], Bs[team][qd * 16 + u][i], zp);
      zp += __shfl_xor(zp, 1);
      zp += __shfl_xor(zp, 2);
      if (qd == 0) zv[team][i] = zp;
    }
    __syncthreads();
    if (exists && k >= 1 && tau != 0.f) {
      const float f = tau * vcur[team][i];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int j = qd * 16 + u;
        if (j >= 1) Bs[team][i][j] -= f * zv[team][j];
      }
    }
    // ---- (d) D <- H D H:  p = tau D v, alpha = -tau p^T v / 2, qv = p + alpha v, D -= v qv^T + qv v^T
    float pp = 0.f;
    if (exists && tau != 0.f) {
#pragma unroll
      for (int u = 0; u < 16; ++u) pp = fmaf(Ds[team][i][qd * 16 + u], vcur[team][qd * 16 + u], pp);
    }
    pp += __shfl_xor(pp, 1);
    pp += __shfl_xor(pp, 2);
    pp *= tau;
    if (qd == 0) wv[team][i] = pp;
    const float pv = team_sum((qd == 0) ? pp * vcur[team][i] : 0.f, red, team, tt);
    const float alpha = -0.5f * tau * pv;
    if (exists && tau != 0.f) {
      if (qd == 0) wv[team][i] = pp + alpha * vcur[team][i];
    }
    __syncthreads();
    // ---- store
    if (exists) {
      if (tau != 0.f) {
        const float vi = vcur[team][i], qi = wv[team][i];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int j = qd * 16 + u;
          if (i < L && j <= i) AB[(long)(r0 + j) * LDB + i - j] = Ds[team][i][j] - vi * wv[team][j] - qi * vcur[team][j];
        }
      }
      if (k >= 1) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int j = qd * 16 + u;
          if (i < L) {
            float bv = Bs[team][i][j];
            if (j == 0 && L >= 2) bv = (i == 0) ? sc[team][2] : 0.f;
            AB[(long)(c0 + j) * LDB + SB + i - j] = bv;
          }
        }
      } else if (tt < L) {
        AB[(long)s * LDB + 1 + tt] = (tt == 0) ? sc[team][2] : 0.f;
      }
      const long idx = (long)k * n + s;
      if (tt < SB) V2[idx * SB + tt] = vcur[team][tt];
      if (tt == 0) tau2[idx] = tau;
    }
    __syncthreads();
    if (tt < SB) vprev[team][tt] = exists ? vcur[team][tt] : vprev[team][tt];
    if (tt == 0 && exists) sc[team][0] = tau;
    __syncthreads();
  }
}


// ---- second form: one WAVE per task (lane i = row i of the 64 x 64 blocks).  The off-diagonal block lives in the lane's
// registers (row access) and in LDS (column access for z = v^T B), the diagonal block in registers only; vectors are
// broadcast with readlane, reductions run on DPP, nothing inside a step needs a workgroup barrier: one barrier per step keeps
// the WSW sweeps of a workgroup in lockstep.  Same scheduling as above (task (s, k) at step 2 s + k, workgroup g runs window
// L - g in launch L).
constexpr int WSW = 8;         // sweeps (= waves) per workgroup

__device__ __forceinline__ float wave_sum64(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

__device__ __forceinline__ float lane_bcast(float v, int l) {   // l: compile-time constant after unrolling
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// The band is addressed through a raw buffer resource: an access of a lane that has no element (rows beyond the block, the
// upper triangle of the diagonal block) gets an offset beyond the buffer - the hardware returns 0 for such a load and drops
// such a store - instead of an exec-mask region per access (the first form spent a third of its instructions on those).
constexpr unsigned BAND_OOB = 0x10000000u;     // element offset beyond any band (1 GiB in bytes; a band has n * 128 elements)
__device__ __forceinline__ float band_ld(__amdgpu_buffer_rsrc_t r, unsigned elem) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(elem * 4u), 0, 0));
}
__device__ __forceinline__ void band_st(__amdgpu_buffer_rsrc_t r, unsigned elem, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)(elem * 4u), 0, 0);
}

__global__ __launch_bounds__(WSW * 64) void sb2st_wave_kernel(float* __restrict__ AB, int n, int launch, int h, int g_lo,
                                                              float* __restrict__ V2, float* __restrict__ tau2) {
  __shared__ float Bs[WSW][SB][SB + 1];
  // vectors every lane needs in full are published in LDS and read back four at a time (one broadcast read serves the wave)
  __shared__ __attribute__((aligned(16))) float vec[WSW][4][SB];     // 0: v_prev, 1: v, 2: z, 3: q
  const int g = g_lo + blockIdx.x;
  const int q = launch - g;
  // (readfirstlane: the wave index is uniform, but only this tells the compiler - without it every per-task quantity lives in
  // vector registers and every "uniform" branch below becomes an exec-mask region)
  const int team = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), i = threadIdx.x & 63;   // lane i = row i of the task's blocks
  const int s = g * WSW + team;
  if (q < 0) return;
  const __amdgpu_buffer_rsrc_t band = __builtin_amdgcn_make_buffer_rsrc(AB, 0, n * LDB * 4, 0x00020000);
  float* vp_s = vec[team][0];
  float* v_s = vec[team][1];
  float* z_s = vec[team][2];
  float* q_s = vec[team][3];
  // the previous reflector of the sweep (task k - 1 ran in the previous step, possibly in the previous launch)
  float tauprev = 0.f;
  {
    const int k_first = q * h - 2 * s;
    float vp = 0.f;
    if (s < n - 2 && k_first >= 1 && s + 1 + (long)(k_first - 1) * SB < n) {   // (a window may start behind the sweep's last task)
      const long idx = (long)(k_first - 1) * n + s;
      vp = V2[idx * SB + i];
      tauprev = tau2[idx];
    }
    tauprev = lane_bcast(tauprev, 0);
    vp_s[i] = vp;
  }
  for (int t = q * h; t < (q + 1) * h; ++t) {
    const int k = t - 2 * s;
    const int r0 = s + 1 + k * SB;
    const bool exists = s < n - 2 && k >= 0 && r0 < n && !(k == 0 && n - r0 < 2);
    if (exists) {
      const int L = min(SB, n - r0);
      const bool row = i < L;
      // ---- load: row i of Bm = B(r_k, r_{k-1}) (k >= 1) and of D = B(r_k, r_k); every load is issued before the first use
      float b[SB], d[SB];
      // Bm(i, j) = AB[(c0 + j) * LDB + SB + i - j] = AB[ob + j * (LDB - 1)]
      const unsigned ob = row ? (unsigned)((r0 - SB) * LDB + SB + i) : BAND_OOB;
      // D(i, j), j <= i: AB[(r0 + j) * LDB + i - j] = AB[ol + j * (LDB - 1)];  j > i: AB[(r0 + i) * LDB + j - i] = AB[ou + j]
      const unsigned ol = row ? (unsigned)(r0 * LDB + i) : BAND_OOB, ou = row ? (unsigned)((r0 + i) * LDB - i) : BAND_OOB;
      if (k >= 1) {
#pragma unroll
        for (int j = 0; j < SB; ++j) b[j] = band_ld(band, ob + (unsigned)(j * (LDB - 1)));
      } else {
#pragma unroll
        for (int j = 0; j < SB; ++j) b[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < SB; ++j)
        d[j] = band_ld(band, (j < L) ? ((j <= i) ? ol + (unsigned)(j * (LDB - 1)) : ou + (unsigned)j) : BAND_OOB);
      float x;   // the column the new reflector annihilates: B(r_0, s) for k = 0, column 0 of Bm (after (a)) otherwise
      if (k >= 1) {
        // ---- (a) Bm <- Bm (I - tau_prev v_prev v_prev^T)
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < SB; j += 4) {
          const float4 w = *reinterpret_cast<const float4*>(vp_s + j);
          part = fmaf(b[j], w.x, part); part = fmaf(b[j + 1], w.y, part); part = fmaf(b[j + 2], w.z, part); part = fmaf(b[j + 3], w.w, part);
        }
        const float f = tauprev * part;
#pragma unroll
        for (int j = 0; j < SB; j += 4) {
          const float4 w = *reinterpret_cast<const float4*>(vp_s + j);
          b[j] = fmaf(-f, w.x, b[j]); b[j + 1] = fmaf(-f, w.y, b[j + 1]); b[j + 2] = fmaf(-f, w.z, b[j + 2]); b[j + 3] = fmaf(-f, w.w, b[j + 3]);
        }
#pragma unroll
        for (int j = 0; j < SB; ++j) Bs[team][i][j] = b[j];
        x = b[0];
      } else {
        x = band_ld(band, row ? (unsigned)(s * LDB + 1 + i) : BAND_OOB);
      }
      // ---- (b) reflector from x (rows 0 .. L - 1)
      const float xn2 = wave_sum64((i >= 1 && row) ? x * x : 0.f);
      const float alpha = lane_bcast(x, 0);
      float tau_ = 0.f, beta_ = alpha, scale_ = 0.f;
      if (L >= 2 && xn2 > 0.f) {
        beta_ = -copysignf(sqrtf(alpha * alpha + xn2), alpha);
        tau_ = (beta_ - alpha) / beta_;
        scale_ = 1.f / (alpha - beta_);
      }
      const float tau = lane_bcast(tau_, 0), beta = lane_bcast(beta_, 0), scale = lane_bcast(scale_, 0);   // uniform, and known to be
      const float v = row ? ((i == 0) ? 1.f : x * scale) : 0.f;
      v_s[i] = v;
      // ---- (c) rest of Bm from the left: z[j] = sum_i v[i] Bm[i][j] (lane j walks column j in LDS), Bm -= tau v z^T
      if (k >= 1 && tau != 0.f) {
        float z = 0.f;
#pragma unroll
        for (int r = 0; r < SB; r += 4) {
          const float4 w = *reinterpret_cast<const float4*>(v_s + r);
          z = fmaf(w.x, Bs[team][r][i], z); z = fmaf(w.y, Bs[team][r + 1][i], z); z = fmaf(w.z, Bs[team][r + 2][i], z); z = fmaf(w.w, Bs[team][r + 3][i], z);
        }
        z_s[i] = z;
        const float f = tau * v;
#pragma unroll
        for (int j = 0; j < SB; j += 4) {
          const float4 w = *reinterpret_cast<const float4*>(z_s + j);
          if (j > 0) b[j] = fmaf(-f, w.x, b[j]);
          b[j + 1] = fmaf(-f, w.y, b[j + 1]); b[j + 2] = fmaf(-f, w.z, b[j + 2]); b[j + 3] = fmaf(-f, w.w, b[j + 3]);
        }
      }
      // ---- (d) D <- H D H:  p = tau D v, alpha2 = -tau p^T v / 2, qv = p + alpha2 v, D -= v qv^T + qv v^T
      if (tau != 0.f) {
        float p = 0.f;
#pragma unroll
        for (int j = 0; j < SB; j += 4) {
          const float4 w = *reinterpret_cast<const float4*>(v_s + j);
          p = fmaf(d[j], w.x, p); p = fmaf(d[j + 1], w.y, p); p = fmaf(d[j + 2], w.z, p); p = fmaf(d[j + 3], w.w, p);
        }
        p *= tau;
        const float pv = wave_sum64(p * v);
        const float qv = fmaf(-0.5f * tau * pv, v, p);
        q_s[i] = qv;
#pragma unroll
        for (int j = 0; j < SB; j += 4) {
          const float4 wq = *reinterpret_cast<const float4*>(q_s + j);
          const float4 wv = *reinterpret_cast<const float4*>(v_s + j);
          // lower triangle only (j <= i): the sign of i - j, spread over the word, pushes the other lanes out of range
          band_st(band, ol + (unsigned)(j * (LDB - 1)) + ((unsigned)((i - j) >> 31) & BAND_OOB), d[j] - v * wq.x - qv * wv.x);
          band_st(band, ol + (unsigned)((j + 1) * (LDB - 1)) + ((unsigned)((i - j - 1) >> 31) & BAND_OOB), d[j + 1] - v * wq.y - qv * wv.y);
          band_st(band, ol + (unsigned)((j + 2) * (LDB - 1)) + ((unsigned)((i - j - 2) >> 31) & BAND_OOB), d[j + 2] - v * wq.z - qv * wv.z);
          band_st(band, ol + (unsigned)((j + 3) * (LDB - 1)) + ((unsigned)((i - j - 3) >> 31) & BAND_OOB), d[j + 3] - v * wq.w - qv * wv.w);
        }
      }
      // ---- store Bm / the eliminated column, the reflector
      if (k >= 1) {
        if (L >= 2) b[0] = (i == 0) ? beta : 0.f;
#pragma unroll
        for (int j = 0; j < SB; ++j) band_st(band, ob + (unsigned)(j * (LDB - 1)), b[j]);
      } else {
        band_st(band, row ? (unsigned)(s * LDB + 1 + i) : BAND_OOB, (i == 0) ? beta : 0.f);
      }
      const long idx = (long)k * n + s;
      V2[idx * SB + i] = v;
      if (i == 0) tau2[idx] = tau;
      vp_s[i] = v;
      tauprev = tau;
    }
    __syncthreads();    // the sweeps of the workgroup advance in lockstep: what this step wrote, the next one reads
  }
}

}  // namespace

size_t pmd_sb2st_workspace_bytes_impl(int n) {
  const size_t K = (size_t)(n / SB + 2);
  return (size_t)n * LDB * sizeof(float) + K * n * SB * sizeof(float) + K * n * sizeof(float) + 65536;
}

// Stage 2 on the band left by pmd_sy2sb_impl in A.  Outputs d[n], e[n - 1] and, in the workspace, the reflectors V2 / tau2
// (pointers returned) for pmd_sb2st_apply_q2_impl.
int pmd_sb2st_impl(pmd_ctx* ctx, int n, const float* A, long lda, float* d, float* e, float** V2_out, float** tau2_out,
                   void* ws, size_t ws_bytes) {
  pmd_prof_scope prof__(ctx, "sb2st");
  pmd_arena ar(ws, ws_bytes);
  const size_t K = (size_t)(n / SB + 2);
  float* AB = ar.take_n<float>((size_t)n * LDB);
  float* V2 = ar.take_n<float>(K * n * SB);
  float* tau2 = ar.take_n<float>(K * n);
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_sb2st", "workspace too small");
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(band_extract_kernel, dim3(n), dim3(128), 0, st, A, lda, n, AB);
  PMD_LAUNCH_CHECK(ctx, "band_extract_kernel");
  PMD_HIP(ctx, hipMemsetAsync(tau2, 0, K * n * sizeof(float), st));
  PMD_HIP(ctx, hipMemsetAsync(V2, 0, K * n * SB * sizeof(float), st));
  if (n > 2) {
    const int n_sweeps = n - 2;
    static int team_form = -1;   // PMD_SB2ST=team: the first form (256-thread teams, four sweeps per workgroup)
    if (team_form < 0) { const char* e_ = getenv("PMD_SB2ST"); team_form = (e_ && !strcmp(e_, "team")) ? 1 : 0; }
    // steps per launch: measured at n = 10^4 - wave form 373 / 393 / 450 ms for h = 2 / 4 / 8, team form 667 ms at h = 8
    static int h_env = 0;
    if (!h_env) { const char* e_ = getenv("PMD_SB2ST_H"); h_env = e_ ? std::max(1, atoi(e_)) : (team_form ? 8 : 2); }
    const int h = h_env;
    const int SWX = team_form ? SW : WSW;
    const int n_wg = (n_sweeps + SWX - 1) / SWX;
    auto tasks_of = [&](int s) { return (n - s - 1 + SB - 1) / SB; };   // K_s
    // last step of the whole reduction, windows per workgroup
    long t_max = 0;
    for (int s = 0; s < n_sweeps; s += std::max(1, n_sweeps / 64)) t_max = std::max<long>(t_max, 2L * s + tasks_of(s) - 1);
    t_max = std::max<long>(t_max, 2L * (n_sweeps - 1) + tasks_of(n_sweeps - 1) - 1);
    const long Q = t_max / h + 1;
    for (long L = 0; L < Q + n_wg - 1; ++L) {
      // workgroups with work in launch L: window q = L - g, steps [q h, q h + h) against the workgroup's steps
      // [2 g SW, 2 (g SW + SW - 1) + K_{g SW} - 1]
      long g_lo = std::max<long>(0, L - Q + 1), g_hi = std::min<long>(n_wg - 1, L);
      while (g_lo <= g_hi) {   // drop workgroups whose window lies before / behind their steps
        const long q = L - g_lo, t_first = 2L * g_lo * SWX, t_last = 2L * (g_lo * SWX + SWX - 1) + tasks_of((int)(g_lo * SWX)) - 1;
        if ((q + 1) * h - 1 < t_first || q * h > t_last) ++g_lo; else break;
      }
      while (g_hi >= g_lo) {
        const long q = L - g_hi, t_first = 2L * g_hi * SWX, t_last = 2L * (g_hi * SWX + SWX - 1) + tasks_of((int)(g_hi * SWX)) - 1;
        if ((q + 1) * h - 1 < t_first || q * h > t_last) --g_hi; else break;
      }
      if (g_lo > g_hi) continue;
      if (team_form)
        hipLaunchKernelGGL(sb2st_kernel, dim3((unsigned)(g_hi - g_lo + 1)), dim3(SW * 256), 0, st, AB, n, (int)L, h, (int)g_lo, V2, tau2);
      else
        hipLaunchKernelGGL(sb2st_wave_kernel, dim3((unsigned)(g_hi - g_lo + 1)), dim3(WSW * 64), 0, st, AB, n, (int)L, h, (int)g_lo, V2, tau2);
    }
    PMD_LAUNCH_CHECK(ctx, "sb2st_kernel");
  }
  hipLaunchKernelGGL(band_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, AB, n, d, e);
  PMD_LAUNCH_CHECK(ctx, "band_diag_kernel");
  *V2_out = V2;
  *tau2_out = tau2;
  return PMD_OK;
}

// =============================================================================================
// Back-transformation of stage 2: Z <- Q2 Z, memory row m of Z = one vector over the positions.  Q2 = product of the
// reflectors in creation order (sweep-major); reflectors with disjoint positions commute, and the order "sweep groups of 64
// descending; inside a group task index k ascending; inside (group, k) sweeps descending" is a valid rearrangement
// (tests/two_stage_ref.py checks it against the plain order).  The 64 reflectors of a (group, k) block touch 127
// consecutive positions.  One workgroup owns 32 vectors and applies every block to its slab: no synchronisation between
// workgroups.  A vector is handled by 8 adjacent lanes (8 positions each), so consecutive reflectors of a block - whose
// windows are shifted by one position - only need the in-order LDS accesses of one wave, no workgroup barrier.
// =============================================================================================
namespace {

constexpr int Q2G = 64;    // sweeps per block
constexpr int Q2V = 32;    // vectors per workgroup
constexpr int Q2W = Q2G + SB - 1;   // positions per block (127)

// sum over the 8 lanes of a vector's lane group, result in all of them (quad butterflies + the mirror of a half row)
__device__ __forceinline__ float q2_sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return v;
}

// One reflector of a block: the lane group of a vector holds the reflector's 64-position window in registers (8 positions a
// lane; logical position u of step count C lives in r[(u - C) & 7], so that sliding the window by one position - the next
// reflector sits one position lower - moves no register: the slot of the position that leaves takes the one that enters).
#define Q2_STEP(C)                                                                                                          \
  {                                                                                                                         \
    /* operands of this step were requested one step ahead; request the next step's now (row a - 1; row 0 again at the end) */ \
    const float4 va = nva, vb = nvb;                                                                                        \
    const float tau = ntau, zin = nzin;                                                                                     \
    const int an = a > 0 ? a - 1 : 0;                                                                                       \
    nva = *reinterpret_cast<const float4*>(&vsc[an][8 * part]);                                                             \
    nvb = *reinterpret_cast<const float4*>(&vsc[an][8 * part + 4]);                                                         \
    ntau = (an < ns) ? tauc[an] : 0.f;                                                                                      \
    nzin = zt[v][an > 0 ? an - 1 : 0];       /* the position that enters after step an (used by part 0 only) */             \
    float d0 = r[(0 - C) & 7] * va.x, d1 = r[(1 - C) & 7] * va.y;                                                            \
    d0 = fmaf(r[(2 - C) & 7], va.z, d0); d1 = fmaf(r[(3 - C) & 7], va.w, d1);                                                \
    d0 = fmaf(r[(4 - C) & 7], vb.x, d0); d1 = fmaf(r[(5 - C) & 7], vb.y, d1);                                                \
    d0 = fmaf(r[(6 - C) & 7], vb.z, d0); d1 = fmaf(r[(7 - C) & 7], vb.w, d1);                                                \
    const float f = tau * q2_sum8(d0 + d1);                                                                                 \
    r[(0 - C) & 7] = fmaf(-f, va.x, r[(0 - C) & 7]); r[(1 - C) & 7] = fmaf(-f, va.y, r[(1 - C) & 7]);                        \
    r[(2 - C) & 7] = fmaf(-f, va.z, r[(2 - C) & 7]); r[(3 - C) & 7] = fmaf(-f, va.w, r[(3 - C) & 7]);                        \
    r[(4 - C) & 7] = fmaf(-f, vb.x, r[(4 - C) & 7]); r[(5 - C) & 7] = fmaf(-f, vb.y, r[(5 - C) & 7]);                        \
    r[(6 - C) & 7] = fmaf(-f, vb.z, r[(6 - C) & 7]); r[(7 - C) & 7] = fmaf(-f, vb.w, r[(7 - C) & 7]);                        \
    if (a > 0) {                                                                                                            \
      const float out = r[(7 - C) & 7];                                                                                     \
      float in = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, out), 0x111, 0xF, 0xF, true)); /* row_shr:1 */ \
      if (part == 7) zt[v][a + SB - 1] = out;                                                                               \
      r[(7 - C) & 7] = (part == 0) ? zin : in;                                                                              \
    }                                                                                                                       \
    --a;                                                                                                                    \
  }

typedef __attribute__((address_space(3))) void* q2_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* q2_gbl_ptr_t;

__global__ __launch_bounds__(256) void apply_q2_kernel(float* __restrict__ Z, long ldz, int n, int nvec, const float* __restrict__ V2,
                                                       const float* __restrict__ tau2) {
  __shared__ float zt[Q2V][Q2W + 2];
  __shared__ __attribute__((aligned(16))) float zn[Q2V][SB];            // the 64 positions the next block adds (LDS-DMA target)
  __shared__ __attribute__((aligned(16))) float vs[2][Q2G][SB];         // reflectors of the current / the next block
  __shared__ __attribute__((aligned(16))) float taus[2][Q2G];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int v = tid >> 3, part = tid & 7;      // vector v of the slab; positions 8 part .. 8 part + 7 of the reflector window
  const long v0 = (long)blockIdx.x * Q2V;
  const int n_sweeps = n - 2;
  const int n_groups = (n_sweeps + Q2G - 1) / Q2G;
  for (int S = n_groups - 1; S >= 0; --S) {
    const int s_lo = S * Q2G;
    const int ns = min(Q2G, n_sweeps - s_lo);           // sweeps of the group; reflectors beyond them count with tau = 0
    const int kmax = (n - s_lo - 1 + SB - 1) / SB;      // blocks with p0 = s_lo + 1 + k SB < n
    // Block k + 1 needs 64 more positions of every vector and its own 64 reflectors: requested as LDS-DMA transfers
    // (global_load_lds: no staging registers, nothing the compiler can move behind the compute loop) while block k computes.
    // Addresses are clamped into the arrays; what lies beyond the matrix is masked when it is used.
    auto prefetch = [&](int k_next, int q0, int buf) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int vv = 8 * wid + j;
        const long row = (v0 + vv < nvec) ? v0 + vv : v0;
        __builtin_amdgcn_global_load_lds((q2_gbl_ptr_t)(Z + row * ldz + min(q0 + lane, n - 1)), (q2_lds_ptr_t)&zn[vv][0], 4, 0, 0);
      }
      const float* blk = V2 + ((long)k_next * n + s_lo) * SB;      // 64 x 64 floats, contiguous, 256-byte aligned
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = 4 * wid + j;                                 // 1 KiB piece m: rows 4 m .. 4 m + 3
        __builtin_amdgcn_global_load_lds((q2_gbl_ptr_t)(blk + 256 * m + 4 * lane), (q2_lds_ptr_t)(&vs[buf][0][0] + 256 * m), 16, 0, 0);
      }
      if (wid == 0)
        __builtin_amdgcn_global_load_lds((q2_gbl_ptr_t)(tau2 + (long)k_next * n + s_lo + lane), (q2_lds_ptr_t)&taus[buf][0], 4, 0, 0);
    };
    {
      // first block of the group: the leading 63 positions through registers, the rest and the reflectors as for every block
      const int p0 = s_lo + 1;
      float ta[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i, vv = e >> 6, q = e & 63;
        const bool ok = p0 + q < n && v0 + vv < nvec;
        ta[i] = Z[(v0 + (ok ? vv : 0)) * ldz + (ok ? p0 + q : 0)];
      }
      __syncthreads();
      prefetch(0, p0 + 63, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i, vv = e >> 6, q = e & 63;
        if (q < 63) zt[vv][q] = (p0 + q < n && v0 + vv < nvec) ? ta[i] : 0.f;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the barrier alone does not wait for vector-memory transfers
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i, vv = e >> 6, q = e & 63;
        zt[vv][63 + q] = (p0 + 63 + q < n && v0 + vv < nvec) ? zn[vv][q] : 0.f;
      }
      __syncthreads();
    }
    for (int k = 0; k < kmax; ++k) {
      const int p0 = s_lo + 1 + k * SB;
      const bool more = k + 1 < kmax;
      const int cur = k & 1;
      const float (*vsc)[SB] = vs[cur];
      const float* tauc = taus[cur];
      if (more) prefetch(k + 1, p0 + 127, cur ^ 1);
      float r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = zt[v][Q2G - 1 + 8 * part + u];     // window of the first reflector (a = 63)
      int a = Q2G - 1;
      float4 nva = *reinterpret_cast<const float4*>(&vsc[a][8 * part]), nvb = *reinterpret_cast<const float4*>(&vsc[a][8 * part + 4]);
      float ntau = (a < ns) ? tauc[a] : 0.f, nzin = zt[v][a - 1];
      for (int it = 0; it < Q2G / 8; ++it) {
        Q2_STEP(0) Q2_STEP(1) Q2_STEP(2) Q2_STEP(3) Q2_STEP(4) Q2_STEP(5) Q2_STEP(6) Q2_STEP(7)
      }
      // a = 0 was the last reflector (no slide after it): logical position u of step count 63 sits in r[(u - 63) & 7] = r[(u + 1) & 7]
#pragma unroll
      for (int u = 0; u < 8; ++u) zt[v][8 * part + u] = r[(u + 1) & 7];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's transfers for the next block have landed ...
      __syncthreads();                                   // ... and everybody's; the window is back in the tile
      // positions 0 .. 63 of the tile are final for this group (all of it after the last block); the rest moves down by 64
      float keep[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i, vv = e >> 6, q = e & 63;
        if (p0 + q < n && v0 + vv < nvec) Z[(v0 + vv) * ldz + p0 + q] = zt[vv][q];
        keep[i] = zt[vv][64 + q];
        if (!more && q < 63 && p0 + 64 + q < n && v0 + vv < nvec) Z[(v0 + vv) * ldz + p0 + 64 + q] = keep[i];
      }
      if (more) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int e = tid + 256 * i, vv = e >> 6, q = e & 63;
          if (q < 63) zt[vv][q] = keep[i];
          zt[vv][63 + q] = (p0 + 127 + q < n && v0 + vv < nvec) ? zn[vv][q] : 0.f;
        }
        __syncthreads();
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no transfer may outlive the workgroup's LDS allocation
}
#undef Q2_STEP

}  // namespace

int pmd_sb2st_apply_q2_impl(pmd_ctx* ctx, int n, const float* V2, const float* tau2, float* Z, long ldz, int nvec) {
  pmd_prof_scope prof__(ctx, "apply_q2");
  if (n <= 2 || nvec <= 0) return PMD_OK;
  hipLaunchKernelGGL(apply_q2_kernel, dim3((unsigned)((nvec + Q2V - 1) / Q2V)), dim3(256), 0, ctx->stream, Z, ldz, n, nvec, V2, tau2);
  PMD_LAUNCH_CHECK(ctx, "apply_q2_kernel");
  return PMD_OK;
}
