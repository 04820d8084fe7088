// Full-movie passes that feed the tile decomposition:
//   * per-pixel mean and Welch noise estimate   (pmd_loader.py:203-291, preprocessing_utils.py:10-40)
//   * (Y-mu)/sigma + transpose to pixel-major    (pmd_loader.py:293-298, :374-378, :396-397)
//   * background filter X - B (B^T X)            (pmd_loader.py:386-387)
//   * temporal bin average / spatial pooling     (decomposition.py:279-290)
// Movie layout on entry: Y[t][c], c = i*d2 + j (frames-first, as the dataset hands it over).
// Working layout: X[c][t], leading dimension ld (multiple of 64, zero padded): each pixel's
// trace is contiguous, so a tile is a gather of d rows and every row segment is a whole
// 128-B line whatever the tile origin is.
#include "pmd_internal.h"
#include "fft_tables.h"

// ---- tables: [0,256) Hann window (periodic); [256,320) cos, [320,384) sin of 2*pi*k/128;
//              [384,513) cos, [513,642) sin of 2*pi*k/256 (k = 0..128)
#define TAB_WIN 0
#define TAB_C128 256
#define TAB_S128 320
#define TAB_C256 384
#define TAB_S256 513
#define TAB_SIZE 642

int pmd_init_tables(pmd_ctx* ctx) {
  float h[TAB_SIZE];
  const double two_pi = 6.283185307179586476925286766559;
  for (int i = 0; i < 256; ++i) h[TAB_WIN + i] = (float)(0.5 - 0.5 * cos(two_pi * i / 256.0));
  for (int k = 0; k < 64; ++k) {
    h[TAB_C128 + k] = (float)cos(two_pi * k / 128.0);
    h[TAB_S128 + k] = (float)sin(two_pi * k / 128.0);
  }
  for (int k = 0; k <= 128; ++k) {
    h[TAB_C256 + k] = (float)cos(two_pi * k / 256.0);
    h[TAB_S256 + k] = (float)sin(two_pi * k / 256.0);
  }
  PMD_HIP(ctx, hipMalloc((void**)&ctx->tables, sizeof(h)));
  PMD_HIP(ctx, hipMemcpy(ctx->tables, h, sizeof(h), hipMemcpyHostToDevice));
  return PMD_OK;
}

__device__ __forceinline__ int bitrev7(int x) { return (int)(__brev((unsigned)x) >> 25); }

// One lane = one pixel x one 1024-frame chunk (frames-first movie: a wave reads 64 consecutive pixels of a
// frame, 256 B).  Every lane runs its own Welch estimate (pmd_loader.py:203-291, preprocessing_utils.py:10-40):
// 256-frame Hann windows at 50 % overlap, constant detrend, one-sided density averaged over the upper half band.
// The 256 frames of a window are loaded into registers in one batch (256 independent loads in flight per lane),
// and the real-input FFT (z[n] = y[2n] + i y[2n+1], 128 complex points) runs IN REGISTERS, fully unrolled with
// compile-time twiddles (fft_tables.h): no LDS, so occupancy is set by the 256-VGPR window, not by a 64 KB
// LDS column block.  Decimation in frequency: natural-order input, bit-reversed output positions (compile time).
__device__ __forceinline__ constexpr int bitrev7c(int x) {
  int r = 0;
  for (int b = 0; b < 7; ++b) r |= ((x >> b) & 1) << (6 - b);
  return r;
}

__global__ __launch_bounds__(64) void stats_chunk_kernel(const float* __restrict__ Y, int T, long D, int frame_const,
                                                         int do_noise, const float* __restrict__ tab,
                                                         double* __restrict__ chunk_sum, float* __restrict__ chunk_noise) {
  (void)tab;
  const int lane = threadIdx.x;
  const long c = (long)blockIdx.x * 64 + lane;
  const bool valid = c < D;
  const long cc = valid ? c : D - 1;
  const int chunk = blockIdx.y;
  const int t0 = chunk * frame_const;
  const int t1 = min(T, t0 + frame_const);
  const int n = t1 - t0;
  const bool noise = do_noise && n >= 256;
  const int nseg = noise ? (n - 128) / 128 : 0;
  const int nhalf = noise ? nseg + 1 : 0;  // full 128-frame halves consumed by the windows

  double s = 0.0;
  float acc = 0.f;
  const float* ybase = Y + (long)t0 * D + cc;
  for (int seg = 0; seg < nseg; ++seg) {
    const float* yp = ybase + (long)seg * 128 * D;
    float re[128], im[128];
#pragma unroll
    for (int u = 0; u < 128; ++u) {
      re[u] = yp[(long)(2 * u) * D];
      im[u] = yp[(long)(2 * u + 1) * D];
    }
    // sums of the two halves (frames 0..127 = points 0..63): 32-term float partials, then double
    double h0 = 0.0, h1 = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float p0 = 0.f, p1 = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        p0 += re[g * 16 + u];
        p1 += im[g * 16 + u];
      }
      if (g < 4) h0 += (double)p0 + (double)p1; else h1 += (double)p0 + (double)p1;
    }
    s += (seg == 0) ? (h0 + h1) : h1;  // every frame enters the chunk sum once
    const float m = (float)((h0 + h1) * (1.0 / 256.0));
#pragma unroll
    for (int u = 0; u < 128; ++u) {
      re[u] = (re[u] - m) * kWin256[2 * u];
      im[u] = (im[u] - m) * kWin256[2 * u + 1];
    }
    // radix-2 decimation in frequency, in place: X[k] ends at position bitrev7(k)
#pragma unroll
    for (int len = 128; len >= 2; len >>= 1) {
      const int half = len >> 1, step = 128 / len;
#pragma unroll
      for (int blk = 0; blk < 128; blk += len) {
#pragma unroll
        for (int k = 0; k < half; ++k) {
          const int i0 = blk + k, i1 = i0 + half;
          const float wr = kC128[k * step], wi = -kS128[k * step];
          const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
          re[i0] = ar + br;
          im[i0] = ai + bi;
          const float dr = ar - br, di = ai - bi;
          re[i1] = dr * wr - di * wi;
          im[i1] = dr * wi + di * wr;
        }
      }
    }
    // real-input unpack for bins 65..128: sum of one-sided power (x2 except Nyquist)
    float p = 0.f;
#pragma unroll
    for (int k = 65; k < 128; ++k) {
      const float zkr = re[bitrev7c(k)], zki = im[bitrev7c(k)];
      const float cr = re[bitrev7c(128 - k)], ci = -im[bitrev7c(128 - k)];
      const float er = 0.5f * (zkr + cr), ei = 0.5f * (zki + ci);
      // O = (Z - conj(Z'))/(2i) = (-i/2) * (dr + i di) = (di/2, -dr/2)
      const float dr = zkr - cr, di = zki - ci;
      const float orr = 0.5f * di, oi = -0.5f * dr;
      const float wr = kC256[k], wi = -kS256[k];
      const float xr = er + (orr * wr - oi * wi);
      const float xi = ei + (orr * wi + oi * wr);
      p += 2.0f * (xr * xr + xi * xi);
    }
    const float xn = re[0] - im[0];
    p += xn * xn;
    acc += p;
  }
  // frames not covered by a full half (chunk tail, or the whole chunk when it is too short for a window)
  for (int t = t0 + nhalf * 128; t < t1; t += 32) {
    float v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = Y[(long)min(t + u, t1 - 1) * D + cc];
    double part = 0.0;
#pragma unroll
    for (int u = 0; u < 32; ++u) part += (t + u < t1) ? (double)v[u] : 0.0;
    s += part;
  }
  if (valid) chunk_sum[(long)chunk * D + c] = s;
  if (!do_noise) return;
  // density scaling 1/(fs*sum w^2) = 1/96; mean over segments; 0.5 * Pxx averaged over 64 bins
  const float val = noise ? sqrtf(acc * (1.0f / 96.0f) / (float)nseg * 0.5f / 64.0f) : 0.f;
  if (valid) chunk_noise[(long)chunk * D + c] = val;
}

__global__ void stats_finalize_kernel(const double* __restrict__ chunk_sum, const float* __restrict__ chunk_noise,
                                      int nchunks, int ncounted, long D, int T, int do_noise,
                                      float* __restrict__ mean_out, float* __restrict__ std_out) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  double s = 0.0;
  for (int k = 0; k < nchunks; ++k) s += chunk_sum[(long)k * D + c];
  mean_out[c] = (float)(s / (double)T);
  if (do_noise && ncounted > 0) {
    float a = 0.f;
    for (int k = 0; k < nchunks; ++k) a += chunk_noise[(long)k * D + c] / (float)nchunks;
    a *= (float)nchunks / (float)ncounted;
    std_out[c] = (a == 0.f) ? 1.0f : a;
  } else {
    std_out[c] = 1.0f;
  }
}

size_t pmd_stats_workspace_bytes(int T, long D, int frame_const) {
  const int nchunks = (T + frame_const - 1) / frame_const;
  return (size_t)nchunks * D * (sizeof(double) + sizeof(float)) + 1024;
}

int pmd_launch_stats(pmd_ctx* ctx, const float* movie, int T, long D, int frame_const, int do_noise, float* mean_out,
                     float* std_out, void* ws, size_t ws_bytes) {
  pmd_prof_scope prof__(ctx, "stats_welch");
  const int nchunks = (T + frame_const - 1) / frame_const;
  pmd_arena ar(ws, ws_bytes);
  double* csum = ar.take_n<double>((size_t)nchunks * D);
  float* cnoise = ar.take_n<float>((size_t)nchunks * D);
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_stats", "workspace too small");
  if (T < 256) do_noise = 0;  // pmd_loader.py:213-214
  int ncounted = 0;
  for (int k = 0; k < nchunks; ++k) {
    const int n = (k + 1 == nchunks) ? T - k * frame_const : frame_const;
    if (n >= 256) ncounted++;
  }
  hipLaunchKernelGGL(stats_chunk_kernel, dim3((unsigned)((D + 63) / 64), nchunks), dim3(64), 0, ctx->stream, movie, T,
                     D, frame_const, do_noise, ctx->tables, csum, cnoise);
  PMD_LAUNCH_CHECK(ctx, "stats_chunk_kernel");
  hipLaunchKernelGGL(stats_finalize_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, csum, cnoise,
                     nchunks, ncounted, D, T, do_noise, mean_out, std_out);
  PMD_LAUNCH_CHECK(ctx, "stats_finalize_kernel");
  return PMD_OK;
}

// ---- standardise + transpose: out[c][f] = (Y[frames[f]][c] - mean[c]) / std[c] ------------
// 64 pixels x 64 frames per workgroup through a padded LDS tile; columns f in [nf, ld) are
// written as zeros so that consumers may read whole padded rows.
__global__ __launch_bounds__(256) void standardize_transpose_kernel(const float* __restrict__ Y, long D,
                                                                    const int* __restrict__ frames, int nf,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ stdv,
                                                                    float* __restrict__ out, long ld) {
  __shared__ float tile[64][65];
  const long c0 = (long)blockIdx.x * 64;
  const int f0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform, and known to be)
  const long c = c0 + tx;
  const float mu = (c < D) ? mean[c] : 0.f;
  const float sg = (c < D) ? stdv[c] : 1.f;
  // the 16 loads of a thread are unconditional (clamped frame / pixel, masked value) and issued back to back: a
  // guarded load compiles to a branch with a full wait behind it
  const long cc = (c < D) ? c : D - 1;
  float y[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int f = min(f0 + ty + 4 * u, nf - 1);
    const long t = frames ? (long)frames[f] : (long)f;
    y[u] = Y[t * D + cc];
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int r = ty + 4 * u;
    tile[r][tx] = (f0 + r < nf && c < D) ? (y[u] - mu) / sg : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const long cw = c0 + r;
    const long f = f0 + tx;
    if (cw < D && f < ld) out[cw * ld + f] = tile[tx][r];
  }
}

int pmd_launch_standardize_transpose(pmd_ctx* ctx, const float* movie, long D, const int* frames, int nf,
                                     const float* mean, const float* stdv, float* out, long ld) {
  pmd_prof_scope prof__(ctx, "standardize_transpose");
  dim3 grid((unsigned)((D + 63) / 64), (unsigned)((ld + 63) / 64));
  hipLaunchKernelGGL(standardize_transpose_kernel, grid, dim3(256), 0, ctx->stream, movie, D, frames, nf, mean, stdv,
                     out, ld);
  PMD_LAUNCH_CHECK(ctx, "standardize_transpose_kernel");
  return PMD_OK;
}

// ---- background filter: out[c][f] = in[c][f] - sum_k basis[c][k] * pj[k][f] ----------------
// basis is [c][k] row-major (ldb = K); pj is [k][f] with leading dimension ldp.  A thread owns one
// frame column: its K projections stay in registers while the workgroup walks 64 pixel rows, so the
// pass is one read + one write of the movie (the basis row is a wave-uniform broadcast load).
template <int KMAX>
// `in` and `out` may be the SAME array (the single-copy memory plan filters in place): neither is __restrict__, every live
// thread reads and writes only its own column, and the clamped loads of the padding threads are discarded.
__global__ __launch_bounds__(256) void filter_kernel(const float* in, float* out, long D,
                                                     int nf, long ld, const float* __restrict__ basis, int K,
                                                     const float* __restrict__ pj, long ldp, int kstride) {
  const long f = (long)blockIdx.x * 256 + threadIdx.x;
  const long c0 = (long)blockIdx.y * 64;
  const bool live = f < nf;
  const bool pad = !live && f < ld;   // padding columns [nf, ld) are written as zeros (consumers read whole rows)
  const long fc = live ? f : nf - 1;
  // loads are unconditional (clamped index, masked value): a guarded load compiles to a branch with a full wait
  float p[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    p[k] = pj[(long)((k < K) ? k : 0) * ldp + fc];
    if (k >= K) p[k] = 0.f;
  }
  const long c1 = (c0 + 64 < D) ? c0 + 64 : D;
  for (long c = c0; c < c1; c += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = in[((c + u < c1) ? c + u : c1 - 1) * ld + fc];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long cu = (c + u < c1) ? c + u : c1 - 1;
      const float* b = basis + cu * kstride;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) acc = fmaf(b[(k < K) ? k : 0], p[k], acc);
      if ((live || pad) && c + u < c1) out[cu * ld + f] = live ? v[u] - acc : 0.f;
    }
  }
}

int pmd_launch_filter(pmd_ctx* ctx, const float* in, float* out, long D, int nf, long ld, const float* basis, int K,
                      const float* pj, long ldp) {
  pmd_prof_scope prof__(ctx, "bg_filter");
  const int bx = (int)((ld + 255) / 256);
  const long rows_per_launch = 65535L * 64;
  // background ranks above 64: blocks of 64 basis columns, each one more in-place pass X <- X - B_blk (B_blk^T X) (the
  // projections pj were all formed from the unfiltered X, and the passes add up)
  for (int k0 = 0; k0 < K; k0 += 64) {
    const int kc = (K - k0 < 64) ? K - k0 : 64;
    const float* src = (k0 == 0) ? in : out;
    for (long c0 = 0; c0 < D; c0 += rows_per_launch) {
      const long cn = (D - c0 < rows_per_launch) ? D - c0 : rows_per_launch;
      dim3 grid(bx, (unsigned)((cn + 63) / 64));
      if (kc <= 16)
        hipLaunchKernelGGL(filter_kernel<16>, grid, dim3(256), 0, ctx->stream, src + c0 * ld, out + c0 * ld, cn, nf, ld,
                           basis + c0 * K + k0, kc, pj + (long)k0 * ldp, ldp, K);
      else
        hipLaunchKernelGGL(filter_kernel<64>, grid, dim3(256), 0, ctx->stream, src + c0 * ld, out + c0 * ld, cn, nf, ld,
                           basis + c0 * K + k0, kc, pj + (long)k0 * ldp, ldp, K);
      PMD_LAUNCH_CHECK(ctx, "filter_kernel");
    }
  }
  return PMD_OK;
}

// ---- per-pixel scaling of rows (pixel_weighting, decomposition.py:717-718) ------------------
__global__ void scale_rows_kernel(float* __restrict__ x, long D, int nf, long ld, const float* __restrict__ w) {
  const long c = blockIdx.y;
  const float s = w[c];
  for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (long)gridDim.x * blockDim.x) x[c * ld + f] *= s;
}

int pmd_launch_scale_rows(pmd_ctx* ctx, float* x, long D, int nf, long ld, const float* w) {
  for (long c0 = 0; c0 < D; c0 += 32768) {
    const long cn = (D - c0 < 32768) ? D - c0 : 32768;
    hipLaunchKernelGGL(scale_rows_kernel, dim3(8, (unsigned)cn), dim3(256), 0, ctx->stream, x + c0 * ld, D, nf, ld,
                       w + c0);
    PMD_LAUNCH_CHECK(ctx, "scale_rows_kernel");
  }
  return PMD_OK;
}

// ---- pooled + temporally binned sketch matrix of every tile (decomposition.py:279-290) -----
// Two steps so that the full-resolution movie is read once, coalesced:
//   xbar[c][cb]          = mean_{tau < a} X[c][a*cb + tau]                       (bin_average)
//   abar[tile][p][cb]    = (1/|window p|) * sum_{q in window p} xbar[pix[tile][q]][cb]   (tile_pool)
// pool_q: [P][pool_max] local pixel ids of window p (-1 = padding).
__global__ __launch_bounds__(256) void bin_average_kernel(const float* __restrict__ X, long ldx, int a, int nbins,
                                                          float* __restrict__ xbar, long ldb) {
  const long c = blockIdx.y;
  const float inv = 1.0f / (float)a;
  for (int cb = blockIdx.x * blockDim.x + threadIdx.x; cb < nbins; cb += gridDim.x * blockDim.x) {
    const float* row = X + c * ldx + (long)cb * a;
    float s = 0.f;
    for (int tau = 0; tau < a; ++tau) s += row[tau];
    xbar[c * ldb + cb] = s * inv;
  }
}

__global__ __launch_bounds__(256) void tile_pool_kernel(const float* __restrict__ xbar, long ldb,
                                                        const int* __restrict__ pix, int d,
                                                        const int* __restrict__ pool_q, int pool_max, int P, int nbins,
                                                        float* __restrict__ abar, long ld_ab, long tile_stride) {
  const int tile = blockIdx.y;
  const int p = blockIdx.z;
  const int* pq = pool_q + (long)p * pool_max;
  const int* px = pix + (long)tile * d;
  int cnt = 0;
  for (int w = 0; w < pool_max; ++w) cnt += (pq[w] >= 0);
  const float inv = 1.0f / (float)cnt;
  for (int cb = blockIdx.x * blockDim.x + threadIdx.x; cb < nbins; cb += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int w = 0; w < pool_max; ++w) {
      const int q = pq[w];
      if (q >= 0) s += xbar[(long)px[q] * ldb + cb];
    }
    abar[(long)tile * tile_stride + (long)p * ld_ab + cb] = s * inv;
  }
}

int pmd_launch_bin_average(pmd_ctx* ctx, const float* X, long ldx, long n_rows, int a, int nbins, float* xbar, long ldb) {
  int bx = (nbins + 255) / 256;
  if (bx > 8) bx = 8;
  for (long c0 = 0; c0 < n_rows; c0 += 32768) {
    const long cn = (n_rows - c0 < 32768) ? n_rows - c0 : 32768;
    hipLaunchKernelGGL(bin_average_kernel, dim3(bx, (unsigned)cn), dim3(256), 0, ctx->stream, X + c0 * ldx, ldx, a, nbins,
                       xbar + c0 * ldb, ldb);
    PMD_LAUNCH_CHECK(ctx, "bin_average_kernel");
  }
  return PMD_OK;
}

// xbar: n_rows x ld_ab scratch (n_rows = pixel rows of X)
int pmd_launch_tile_pool_bin(pmd_ctx* ctx, const float* X, long ldx, long n_rows, const int* pix, int n_tiles, int d,
                             const int* pool_q, int pool_max, int P, int a, int nbins, float* xbar, float* abar,
                             long ld_ab, long tile_stride) {
  pmd_prof_scope prof__(ctx, "tile_pool_bin");
  int bx = (nbins + 255) / 256;
  if (bx > 8) bx = 8;
  for (long c0 = 0; c0 < n_rows; c0 += 32768) {
    const long cn = (n_rows - c0 < 32768) ? n_rows - c0 : 32768;
    hipLaunchKernelGGL(bin_average_kernel, dim3(bx, (unsigned)cn), dim3(256), 0, ctx->stream, X + c0 * ldx, ldx, a, nbins,
                       xbar + c0 * ld_ab, ld_ab);
    PMD_LAUNCH_CHECK(ctx, "bin_average_kernel");
  }
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    hipLaunchKernelGGL(tile_pool_kernel, dim3(bx, tn, P), dim3(256), 0, ctx->stream, xbar, ld_ab, pix + (long)t0 * d, d,
                       pool_q, pool_max, P, nbins, abar + (long)t0 * tile_stride, ld_ab, tile_stride);
    PMD_LAUNCH_CHECK(ctx, "tile_pool_kernel");
  }
  return PMD_OK;
}
