// Local correlation images: the per-pixel Python loops of /root/reference/localmd/diagnostic_plots.py
//   make_residual_correlation_image (:100-163), make_pmd_correlation_image (:166-223),
//   make_correlation_image (:225-271), make_autocorrelation_image (:274-304)
// as two HBM-bound passes.  All four are functions of first and second moments of pixel traces:
//   neighbour_moments : for every pixel p and its 8 neighbours q, sum_t x_p, sum_t x_p^2, sum_t x_p x_q, with
//                       x = A - B (B optional) and every trace shifted by its value in a reference frame (covariances
//                       are shift invariant; the products then stay of the order of the variance, so fp32 products and
//                       fp64 sums lose nothing on movies whose mean is 100 x their standard deviation);
//   lag_moments       : the five sums behind corr(x[lag:], x[:-lag]);
//   *_image           : the reference's normalisations and its max / mean over the neighbours that exist.
// The movie is frames-first, X[t][p], p = i * d2 + j: a wave reads 64 consecutive pixels of a frame (one 256-B line
// per load), the eight neighbour loads of the same frame hit L1 / L2, so each frame is fetched from HBM once.
#include "pmd_internal.h"

namespace {

constexpr int DIAG_SLICE = 64;   // frames whose products are summed in fp32 before they enter the fp64 sums

// partial[slice][10][D]; slice = blockIdx.y owns frames [t0, t1)
__global__ __launch_bounds__(256) void neighbour_moments_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                const float* __restrict__ ref, long T, int d1, int d2,
                                                                int frames_per_block, double* __restrict__ partial) {
  const long D = (long)d1 * d2;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const bool ok = p < D;
  const long pc = ok ? p : D - 1;
  const int i = (int)(pc / d2), j = (int)(pc - (long)i * d2);
  long q[8];
  float rq[8];
  int k = 0;
#pragma unroll
  for (int di = -1; di <= 1; ++di)
#pragma unroll
    for (int dj = -1; dj <= 1; ++dj) {
      if (di == 0 && dj == 0) continue;
      const int ii = min(max(i + di, 0), d1 - 1), jj = min(max(j + dj, 0), d2 - 1);  // clamped: the image step ignores them
      q[k] = (long)ii * d2 + jj;
      rq[k] = ref[q[k]];
      ++k;
    }
  const float rp = ref[pc];
  double acc[10];
#pragma unroll
  for (int m = 0; m < 10; ++m) acc[m] = 0.0;
  const long t0 = (long)blockIdx.y * frames_per_block;
  const long t1 = min(T, t0 + frames_per_block);
  for (long ts = t0; ts < t1; ts += DIAG_SLICE) {
    float f[10];
#pragma unroll
    for (int m = 0; m < 10; ++m) f[m] = 0.f;
    const long te = min(t1, ts + DIAG_SLICE);
    for (long t = ts; t < te; ++t) {
      const float* a = A + t * D;
      float xp = a[pc];
      float xq[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) xq[m] = a[q[m]];
      if (B) {
        const float* b = B + t * D;
        xp -= b[pc];
#pragma unroll
        for (int m = 0; m < 8; ++m) xq[m] -= b[q[m]];
      }
      xp -= rp;
      f[0] += xp;
      f[1] += xp * xp;
#pragma unroll
      for (int m = 0; m < 8; ++m) f[2 + m] += xp * (xq[m] - rq[m]);
    }
#pragma unroll
    for (int m = 0; m < 10; ++m) acc[m] += (double)f[m];
  }
  if (!ok) return;
  double* o = partial + (long)blockIdx.y * 10 * D + p;
#pragma unroll
  for (int m = 0; m < 10; ++m) o[(long)m * D] = acc[m];
}

// partial[slice][5][D]: sum x_t (t >= lag), sum x_t^2 (t >= lag), sum x_t (t < T - lag), sum x_t^2 (t < T - lag), sum x_t x_{t-lag}
__global__ __launch_bounds__(256) void lag_moments_kernel(const float* __restrict__ A, const float* __restrict__ ref, long T, long D,
                                                          int lag, int frames_per_block, double* __restrict__ partial) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const bool ok = p < D;
  const long pc = ok ? p : D - 1;
  const float rp = ref[pc];
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const long t0 = (long)blockIdx.y * frames_per_block + lag;   // t runs over [lag, T)
  const long t1 = min(T, t0 + frames_per_block);
  for (long ts = t0; ts < t1; ts += DIAG_SLICE) {
    float f[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const long te = min(t1, ts + DIAG_SLICE);
    for (long t = ts; t < te; ++t) {
      const float x = A[t * D + pc] - rp, y = A[(t - lag) * D + pc] - rp;
      f[0] += x; f[1] += x * x; f[2] += y; f[3] += y * y; f[4] += x * y;
    }
#pragma unroll
    for (int m = 0; m < 5; ++m) acc[m] += (double)f[m];
  }
  if (!ok) return;
  double* o = partial + (long)blockIdx.y * 5 * D + p;
#pragma unroll
  for (int m = 0; m < 5; ++m) o[(long)m * D] = acc[m];
}

// out[e] (+)= sum over slices of partial[slice][e], e < n_elems (fixed order: reproducible)
__global__ void sum_slices_kernel(const double* __restrict__ partial, long n_elems, int slices, int accumulate, double* __restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_elems) return;
  double s = accumulate ? out[e] : 0.0;
  for (int k = 0; k < slices; ++k) s += partial[(long)k * n_elems + e];
  out[e] = s;
}

// kind 0: Pearson correlation of the numerator movie (diagnostic_plots.py:235-241)
// kind 1: cov_{ddof=1}(x_p, x_q) / sqrt(var_{ddof=0}(y_p) var_{ddof=0}(y_q)), y = denominator movie (:117-123, :187-193)
// mode 0: max over the existing neighbours, starting from 0 (:150-151); mode 1: their mean (:148-149, :157-158)
__global__ __launch_bounds__(256) void neighbour_image_kernel(const double* __restrict__ num, const double* __restrict__ den, long T,
                                                              int d1, int d2, int kind, int mode, double* __restrict__ out) {
  const long D = (long)d1 * d2;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= D) return;
  const int i = (int)(p / d2), j = (int)(p - (long)i * d2);
  const double n = (double)T;
  const double sp = num[p];
  const double* dsum = kind == 0 ? num : den;
  const double* dsq = kind == 0 ? num + D : den + D;
  const double ssp = dsq[p] - dsum[p] * dsum[p] / n;   // centred sum of squares of the normalising movie
  double best = 0.0, total = 0.0;
  int count = 0, k = 0;
  for (int di = -1; di <= 1; ++di)
    for (int dj = -1; dj <= 1; ++dj) {
      if (di == 0 && dj == 0) continue;
      const int ii = i + di, jj = j + dj;
      if (ii >= 0 && ii < d1 && jj >= 0 && jj < d2) {
        const long q = (long)ii * d2 + jj;
        const double cross = num[(long)(2 + k) * D + p] - sp * num[q] / n;   // centred cross product
        const double ssq = dsq[q] - dsum[q] * dsum[q] / n;
        double val;
        if (kind == 0) val = cross / sqrt(ssp * ssq);
        else val = (cross / (n - 1.0)) / sqrt((ssp / n) * (ssq / n));
        total += val;
        // Python's max(cov, net_corr) (diagnostic_plots.py:150-151) keeps net_corr only if `net_corr > cov` is true: a NaN value
        // (zero-variance neighbour) REPLACES the running maximum, and the next neighbour then replaces the NaN, without the 0 floor
        best = (best > val) ? best : val;
        ++count;
      }
      ++k;
    }
  out[p] = mode == 0 ? best : total / (double)count;
}

__global__ void lag_image_kernel(const double* __restrict__ mom, long D, double n, double* __restrict__ out) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= D) return;
  const double sx = mom[p], sxx = mom[D + p], sy = mom[2 * D + p], syy = mom[3 * D + p], sxy = mom[4 * D + p];
  out[p] = (sxy - sx * sy / n) / (sqrt(sxx - sx * sx / n) * sqrt(syy - sy * sy / n));
}

}  // namespace

size_t pmd_diag_workspace_bytes_impl(long T, long D) {
  const int fpb = 1024;
  const long slices = (T + fpb - 1) / fpb;
  return (size_t)slices * 10 * (size_t)D * sizeof(double) + 4096;
}

// moments[10][D] (+)= neighbour moments of the frames given (A - B, B may be NULL); ref[D] = reference frame
int pmd_neighbour_moments_impl(pmd_ctx* ctx, const float* A, const float* B, const float* ref, long T, int d1, int d2,
                               int accumulate, double* moments, void* ws, size_t ws_bytes) {
  pmd_prof_scope prof__(ctx, "diag_moments");
  if (T <= 0) return PMD_OK;
  const long D = (long)d1 * d2;
  const int fpb = 1024;
  const int slices = (int)((T + fpb - 1) / fpb);
  if ((size_t)slices * 10 * (size_t)D * sizeof(double) > ws_bytes) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_neighbour_moments", "workspace too small");
  double* partial = static_cast<double*>(ws);
  hipLaunchKernelGGL(neighbour_moments_kernel, dim3((unsigned)((D + 255) / 256), slices), dim3(256), 0, ctx->stream, A, B, ref, T, d1,
                     d2, fpb, partial);
  PMD_LAUNCH_CHECK(ctx, "neighbour_moments_kernel");
  hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((10 * D + 255) / 256)), dim3(256), 0, ctx->stream, partial, 10 * D, slices,
                     accumulate, moments);
  PMD_LAUNCH_CHECK(ctx, "sum_slices_kernel");
  return PMD_OK;
}

// moments[5][D] (+)= lag moments of the T resident frames: pairs (t, t - lag) for t in [lag, T), see lag_moments_kernel
int pmd_lag_moments_impl(pmd_ctx* ctx, const float* A, const float* ref, long T, long D, int lag, int accumulate, double* moments,
                         void* ws, size_t ws_bytes) {
  pmd_prof_scope prof__(ctx, "diag_moments");
  if (lag < 1 || lag >= T) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_lag_moments", "need 1 <= lag < frames");
  const int fpb = 1024;
  const long n = T - lag;
  const int slices = (int)((n + fpb - 1) / fpb);
  if ((size_t)slices * 5 * (size_t)D * sizeof(double) > ws_bytes) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_lag_moments", "workspace too small");
  double* partial = static_cast<double*>(ws);
  hipLaunchKernelGGL(lag_moments_kernel, dim3((unsigned)((D + 255) / 256), slices), dim3(256), 0, ctx->stream, A, ref, T, D, lag, fpb,
                     partial);
  PMD_LAUNCH_CHECK(ctx, "lag_moments_kernel");
  hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((5 * D + 255) / 256)), dim3(256), 0, ctx->stream, partial, 5 * D, slices,
                     accumulate, moments);
  PMD_LAUNCH_CHECK(ctx, "sum_slices_kernel");
  return PMD_OK;
}

int pmd_neighbour_image_impl(pmd_ctx* ctx, const double* num, const double* den, long T, int d1, int d2, int kind, int mode,
                             double* out) {
  pmd_prof_scope prof__(ctx, "diag_image");
  if (kind != 0 && !den) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_neighbour_image", "kind 1 needs the moments of the normalising movie");
  const long D = (long)d1 * d2;
  hipLaunchKernelGGL(neighbour_image_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, num, den, T, d1, d2, kind,
                     mode, out);
  PMD_LAUNCH_CHECK(ctx, "neighbour_image_kernel");
  return PMD_OK;
}

int pmd_lag_image_impl(pmd_ctx* ctx, const double* moments, long D, long n, double* out) {
  pmd_prof_scope prof__(ctx, "diag_image");
  hipLaunchKernelGGL(lag_image_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, moments, D, (double)n, out);
  PMD_LAUNCH_CHECK(ctx, "lag_image_kernel");
  return PMD_OK;
}
