// Batched small dense factorizations, one workgroup per tile, fp64 in LDS.
//   small_qr   : reduced Householder QR (LAPACK dgeqr2 + dorg2r conventions), replaces
//                jnp.linalg.qr at decomposition.py:64 / pmd_loader.py:58
//   small_eig  : cyclic parallel Jacobi eigensolver of a symmetric n x n Gram matrix (n <= 64);
//                with the streaming Gram/rowmix kernels it replaces the jnp.linalg.svd calls at
//                decomposition.py:66, :301, :315-317, :319 (SVD of M as eigh(M M^T))
//   roughness statistics and the keep/discard scan (evaluation.py:84-126, :133-222)
#include "pmd_internal.h"

// ---------------------------------------------------------------- Householder QR ----------
// Yt: [tile][comp][q] (fp32), P rows (q < P), l columns (comp < l).  Qt out: [tile][comp][q],
// rows >= min(P,l) and columns >= P are left untouched (callers pre-zero the buffer).
template <typename ST>
__global__ __launch_bounds__(256) void small_qr_kernel(const float* __restrict__ Yt, long y_tile_stride, int y_ld,
                                                       int P, int l, float* __restrict__ Qt, long q_tile_stride,
                                                       int q_ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nref = min(P, l);
  const int ldm = P | 1;  // odd leading dimension: columns start on different banks
  ST* M = reinterpret_cast<ST*>(smem);                                   // M[c*ldm + i] = Y[i][c]
  double* tau = reinterpret_cast<double*>(smem + (((size_t)l * ldm * sizeof(ST) + 15) & ~size_t(15)));  // [64]
  double* scratch = tau + 64;                                            // [256 + 4]
  const int tid = threadIdx.x;
  const float* y = Yt + (long)blockIdx.x * y_tile_stride;
  for (int i = tid; i < l * P; i += 256) {
    const int c = i / P, q = i - c * P;
    M[c * ldm + q] = (ST)y[(long)c * y_ld + q];
  }
  __syncthreads();

  const int col = tid >> 2, sub = tid & 3;  // four threads per column
  for (int k = 0; k < nref; ++k) {
    // sigma = sum_{i>k} Y[i][k]^2
    double part = 0.0;
    for (int i = k + 1 + tid; i < P; i += 256) { const double v = (double)M[k * ldm + i]; part += v * v; }
    scratch[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) scratch[tid] += scratch[tid + s];
      __syncthreads();
    }
    if (tid == 0) {
      const double sigma = scratch[0];
      const double alpha = (double)M[k * ldm + k];
      double t = 0.0, scale = 0.0, beta = alpha;
      if (sigma > 0.0) {
        const double nrm = sqrt(alpha * alpha + sigma);
        beta = (alpha >= 0.0) ? -nrm : nrm;
        t = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      }
      tau[k] = t;
      scratch[256] = scale;
      M[k * ldm + k] = (ST)beta;
    }
    __syncthreads();
    const double scale = scratch[256];
    const double t = tau[k];
    for (int i = k + 1 + tid; i < P; i += 256) M[k * ldm + i] = (ST)((double)M[k * ldm + i] * scale);
    __syncthreads();
    if (t != 0.0) {
      for (int j = k + 1 + col; j < l; j += 64) {
        double dot = 0.0;
        for (int i = k + 1 + sub; i < P; i += 4) dot += (double)M[k * ldm + i] * (double)M[j * ldm + i];
        dot += __shfl_xor(dot, 1);
        dot += __shfl_xor(dot, 2);
        const double w = t * ((double)M[j * ldm + k] + dot);
        for (int i = k + 1 + sub; i < P; i += 4) M[j * ldm + i] = (ST)((double)M[j * ldm + i] - w * (double)M[k * ldm + i]);
        if (sub == 0) M[j * ldm + k] = (ST)((double)M[j * ldm + k] - w);
      }
    }
    __syncthreads();
  }
  // columns beyond nref hold only R entries: they are not part of Q
  // form Q in place (dorg2r), k = nref-1 .. 0
  for (int k = nref - 1; k >= 0; --k) {
    const double t = tau[k];
    if (t != 0.0) {
      for (int j = k + 1 + col; j < nref; j += 64) {
        double dot = 0.0;
        for (int i = k + 1 + sub; i < P; i += 4) dot += (double)M[k * ldm + i] * (double)M[j * ldm + i];
        dot += __shfl_xor(dot, 1);
        dot += __shfl_xor(dot, 2);
        const double w = t * ((double)M[j * ldm + k] + dot);
        for (int i = k + 1 + sub; i < P; i += 4) M[j * ldm + i] = (ST)((double)M[j * ldm + i] - w * (double)M[k * ldm + i]);
        if (sub == 0) M[j * ldm + k] = (ST)((double)M[j * ldm + k] - w);
      }
    }
    __syncthreads();
    for (int i = tid; i < P; i += 256) {
      double v;
      if (i < k) v = 0.0;
      else if (i == k) v = 1.0 - t;
      else v = -t * (double)M[k * ldm + i];
      M[k * ldm + i] = (ST)v;
    }
    __syncthreads();
  }
  float* qo = Qt + (long)blockIdx.x * q_tile_stride;
  for (int i = tid; i < nref * P; i += 256) {
    const int c = i / P, q = i - c * P;
    qo[(long)c * q_ld + q] = (float)M[c * ldm + q];
  }
}

// does the P x l matrix (fp32 storage at least) fit the 160 KB of LDS a workgroup can have?
bool pmd_small_qr_fits(int P, int l) {
  const size_t tail = (64 + 260) * sizeof(double) + 16;
  return l <= 64 && (size_t)l * (P | 1) * sizeof(float) + tail <= 160 * 1024;
}

int pmd_launch_small_qr(pmd_ctx* ctx, const float* Yt, long y_tile_stride, int y_ld, int P, int l, float* Qt,
                        long q_tile_stride, int q_ld, int n_tiles) {
  pmd_prof_scope prof__(ctx, "small_qr");
  if (n_tiles <= 0) return PMD_OK;
  if (l > 64) return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "small_qr", "more than 64 columns");
  const int ldm = P | 1;
  const size_t tail = (64 + 260) * sizeof(double) + 16;
  size_t bytes = (size_t)l * ldm * sizeof(double) + tail;
  const bool use_double = bytes <= 150 * 1024;
  if (!use_double) bytes = (size_t)l * ldm * sizeof(float) + tail;
  if (bytes > 160 * 1024) return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "small_qr", "matrix does not fit LDS");
  if (use_double) {
    PMD_HIP(ctx, hipFuncSetAttribute((const void*)small_qr_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL(small_qr_kernel<double>, dim3(n_tiles), dim3(256), bytes, ctx->stream, Yt, y_tile_stride, y_ld, P,
                       l, Qt, q_tile_stride, q_ld);
  } else {
    PMD_HIP(ctx, hipFuncSetAttribute((const void*)small_qr_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL(small_qr_kernel<float>, dim3(n_tiles), dim3(256), bytes, ctx->stream, Yt, y_tile_stride, y_ld, P,
                       l, Qt, q_tile_stride, q_ld);
  }
  PMD_LAUNCH_CHECK(ctx, "small_qr_kernel");
  return PMD_OK;
}

// ---------------------------------------------------------------- Jacobi eigensolver ------
// G: [tile][slices][64][64] doubles (summed on load; only the leading n x n block is used).
// Nout[tile][c'][c] = eigenvector c (descending eigenvalue), component c'.
//   mode 0: plain eigenvectors
//   mode 1: column c scaled by 1/sqrt(lambda_c); columns with lambda_c <= tol*lambda_max zeroed
// lam_out[tile][c] = eigenvalue c (descending); entries >= n are zero.
#define EIG_LD 65
// Threads per problem (PMD_EIG_THREADS=256|512|1024, default 1024).  A problem holds 66.5 KB of LDS (A and V in fp64), so
// two share a CU whatever the workgroup size; one rotation step is ~18 000 fp64 FMAs (A <- J^T A J and V <- V J at
// n = 60) between two barriers.  Measured (round 2, 16 129 tiles x 4 launches): 83 / 96 / 130 ms per step with 1024 /
// 512 / 256 threads - the step is bound by its chain of dependent LDS round trips, which more threads shorten, not by
// the fp64 rate.
template <int EIG_THREADS>
__global__ __launch_bounds__(EIG_THREADS) void small_eig_kernel(const double* __restrict__ G, long g_tile_stride, int slices,
                                                        int n, int mode, double tol, double* __restrict__ Nout,
                                                        double* __restrict__ lam_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* A = reinterpret_cast<double*>(smem);  // [64][EIG_LD]
  double* V = A + 64 * EIG_LD;                  // [64][EIG_LD]
  double* cs = V + 64 * EIG_LD;                 // [32][2]
  int* pq = reinterpret_cast<int*>(cs + 64);    // [32][2]
  int* flag = pq + 64;                          // [2]
  int* order = flag + 2;                        // [64]
  const int tid = threadIdx.x;
  const double* g = G + (long)blockIdx.x * g_tile_stride;
  for (int i = tid; i < 64 * 64; i += EIG_THREADS) {
    const int r = i >> 6, c = i & 63;
    double s = 0.0;
    if (r < n && c < n)
      for (int k = 0; k < slices; ++k) s += g[(long)k * 4096 + r * 64 + c] + g[(long)k * 4096 + c * 64 + r];
    A[r * EIG_LD + c] = 0.5 * s;
    V[r * EIG_LD + c] = (r == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int npad = (n + 1) & ~1;
  const int half = npad / 2;
  for (int sweep = 0; sweep < 40; ++sweep) {
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    for (int m = 0; m < npad - 1; ++m) {
      if (tid < half) {
        // round-robin pairing: fix player npad-1, rotate the rest
        int p, q;
        if (tid == 0) { p = npad - 1; q = m; }
        else { p = (m + tid) % (npad - 1); q = (m - tid + (npad - 1)) % (npad - 1); }
        if (p > q) { const int t = p; p = q; q = t; }
        double c = 1.0, s = 0.0;
        if (q < n) {
          const double app = A[p * EIG_LD + p], aqq = A[q * EIG_LD + q], apq = A[p * EIG_LD + q];
          const double prod = fabs(app * aqq), apq2 = apq * apq;
          if (apq2 > 1e-34 * prod && fabs(apq) > 1e-300) {
            // This chain sits between two barriers of every step, so it is kept short: t = tan(theta) from fp32
            // arithmetic on exponent-normalised operands (an error of 1e-7 in t leaves 1e-7 apq behind instead of
            // zero - the next sweep takes it; convergence stays quadratic), then c = 1/sqrt(1 + t^2), s = t c in
            // fp64, which is what keeps the accumulated rotations orthogonal to fp64 accuracy.
            const double num = aqq - app, den = 2.0 * apq;
            const int ex = max(__builtin_amdgcn_frexp_exp(num), __builtin_amdgcn_frexp_exp(den));
            const float th = (float)ldexp(num, -ex) / (float)ldexp(den, -ex);
            const float ath = fabsf(th);
            const float t32 = (ath > 1e9f) ? 0.5f / th : copysignf(1.0f, th) / (ath + sqrtf(1.0f + th * th));
            const double t = (double)t32;
            c = rsqrt(1.0 + t * t);
            s = t * c;
            // another sweep is needed only if this one still met an off-diagonal above 1e-6 sqrt(app aqq): the rotations of
            // this sweep take such entries to ~1e-12 (quadratic convergence), fp64 grade for vectors that leave as fp32.
            // (1e-24 here ran one more full sweep of 59 two-barrier steps to verify what the bound already says: 12 % of the
            // kernel on the many-tile workloads.)
            if (apq2 > 1e-12 * prod) flag[0] = 1;
          }
        }
        cs[2 * tid] = c; cs[2 * tid + 1] = s;
        pq[2 * tid] = p; pq[2 * tid + 1] = q;
      }
      __syncthreads();
      // A <- J^T A J in one pass: the 2x2 block (pair k rows) x (pair k' columns) gets both rotations;
      // blocks are disjoint, so one barrier per step suffices.  V <- V J rides in the same phase.
      for (int i = tid; i < half * half; i += EIG_THREADS) {
        const int k = i / half, kp = i - k * half;
        const double c = cs[2 * k], s = cs[2 * k + 1], c2 = cs[2 * kp], s2 = cs[2 * kp + 1];
        if (s != 0.0 || s2 != 0.0) {
          const int p = pq[2 * k], q = pq[2 * k + 1], p2 = pq[2 * kp], q2 = pq[2 * kp + 1];
          const double a11 = A[p * EIG_LD + p2], a12 = A[p * EIG_LD + q2];
          const double a21 = A[q * EIG_LD + p2], a22 = A[q * EIG_LD + q2];
          const double r11 = c * a11 - s * a21, r12 = c * a12 - s * a22;   // rows rotated
          const double r21 = s * a11 + c * a21, r22 = s * a12 + c * a22;
          A[p * EIG_LD + p2] = c2 * r11 - s2 * r12;
          A[p * EIG_LD + q2] = s2 * r11 + c2 * r12;
          A[q * EIG_LD + p2] = c2 * r21 - s2 * r22;
          A[q * EIG_LD + q2] = s2 * r21 + c2 * r22;
        }
      }
      for (int i = tid; i < half * n; i += EIG_THREADS) {
        const int k = i / n, j = i - k * n;
        const double c = cs[2 * k], s = cs[2 * k + 1];
        if (s != 0.0) {
          const int p = pq[2 * k], q = pq[2 * k + 1];
          const double vp = V[j * EIG_LD + p], vq = V[j * EIG_LD + q];
          V[j * EIG_LD + p] = c * vp - s * vq;
          V[j * EIG_LD + q] = s * vp + c * vq;
        }
      }
      __syncthreads();
    }
    if (flag[0] == 0) break;
    __syncthreads();
  }
  // rank eigenvalues (descending, ties by index)
  if (tid < 64) {
    int rank = 0;
    if (tid < n) {
      const double li = A[tid * EIG_LD + tid];
      for (int j = 0; j < n; ++j) {
        const double lj = A[j * EIG_LD + j];
        rank += (lj > li) || (lj == li && j < tid);
      }
      order[rank] = tid;
    }
  }
  __syncthreads();
  double lmax = (n > 0) ? A[order[0] * EIG_LD + order[0]] : 0.0;
  double* no = Nout + (long)blockIdx.x * 4096;
  for (int i = tid; i < 4096; i += EIG_THREADS) {
    const int r = i >> 6, c = i & 63;
    double v = 0.0;
    if (r < n && c < n) {
      const int src = order[c];
      v = V[r * EIG_LD + src];
      if (mode == 1) {
        const double lam = A[src * EIG_LD + src];
        v = (lam > tol * lmax && lam > 0.0) ? v / sqrt(lam) : 0.0;
      }
    }
    no[i] = v;
  }
  if (tid < 64) lam_out[(long)blockIdx.x * 64 + tid] = (tid < n) ? A[order[tid] * EIG_LD + order[tid]] : 0.0;
}

// ---------------------------------------------------------------- tridiagonal QL eigensolver, one wave per problem ----
// The same contract as small_eig_kernel (G summed over its slices and symmetrised on load; Nout / lam_out in descending
// order; mode 1 scaling and null rule), by Householder tridiagonalisation (LAPACK dsytd2 'L' conventions), in-place
// formation of Q (dorg2r on the shifted array, as dorgtr does) and the implicit QL iteration with the rotations
// accumulated into Q (EISPACK tql2) - all in ONE wave on one n x (n + 1) fp64 array in LDS (29 KB at n = 60: five problems
// share a CU; the Jacobi kernel above holds A and V, 66 KB, two per CU, and spends two 1024-thread barriers on every one
// of its ~470 rotation steps).  The tridiagonal matrix lives in REGISTERS: lane i holds d[i], e[i], tau[i]; a scalar is
// fetched with v_readlane and stored by a predicated move, so the dependent chain of a rotation is ~250 cycles of fp64
// arithmetic with no memory access in it.  The accumulated rotations touch a lane's own row of Q only (no cross-lane
// hazard in the whole QL phase), and the column two rotations share stays in a register.  Cross-lane traffic through LDS
// exists only in the two Householder phases (reflector broadcast), fenced by single-wave barriers.
__device__ __forceinline__ double pmd_readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void small_eig_ql_kernel(const double* __restrict__ G, long g_tile_stride, int slices, int n,
                                                          int mode, double tol, double* __restrict__ Nout,
                                                          double* __restrict__ lam_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int LD = n + 1;
  double* A = reinterpret_cast<double*>(smem);                 // [n][LD]; ends as the eigenvector matrix
  double* vv = A + (size_t)n * LD;                             // [64] reflector
  double* ww = vv + 64;                                        // [64]
  int* order = reinterpret_cast<int*>(ww + 64);                // [64]
  const int lane = threadIdx.x;
  const double* g = G + (long)blockIdx.x * g_tile_stride;
  for (int i = lane; i < n * n; i += 64) {
    const int r = i / n, c = i - r * n;
    double sm = 0.0;
    for (int k = 0; k < slices; ++k) sm += g[(long)k * 4096 + r * 64 + c] + g[(long)k * 4096 + c * 64 + r];
    A[r * LD + c] = 0.5 * sm;
  }
  auto wsum = [](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  double dreg = 0.0, ereg = 0.0, treg = 0.0;                   // d[lane], e[lane] (couples lane, lane + 1), tau[lane]
  double* row = A + (size_t)(lane < n ? lane : 0) * LD;        // this lane's row (lanes >= n never store)
  const bool live = lane < n;
  // ---- Householder tridiagonalisation: A = Q T Q^T, reflector k in A[k+2.., k], tau[k]
  for (int k = 0; k + 1 < n; ++k) {
    __syncthreads();                                           // rows updated by the previous step are visible
    const double alpha = A[(k + 1) * LD + k];
    const double xi = (lane > k + 1 && live) ? row[k] : 0.0;
    const double sigma = wsum(xi * xi);
    if (lane == k) dreg = row[k];
    if (sigma == 0.0) {
      if (lane == k) { ereg = alpha; treg = 0.0; }
      continue;
    }
    const double nrm = sqrt(alpha * alpha + sigma);
    const double beta = (alpha >= 0.0) ? -nrm : nrm;
    const double t = (beta - alpha) / beta;
    const double scale = 1.0 / (alpha - beta);
    const double vi = (lane == k + 1) ? 1.0 : xi * scale;     // zero for lane <= k and lane >= n
    if (lane == k) { ereg = beta; treg = t; }
    vv[lane] = vi;
    if (lane > k + 1 && live) row[k] = vi;
    __syncthreads();
    // p = tau A22 v (row `lane` of the full symmetric trailing block)
    double pi = 0.0;
    if (lane > k && live) {
      double p0 = 0.0, p1 = 0.0;
      int j = k + 1;
      for (; j + 1 < n; j += 2) { p0 = fma(row[j], vv[j], p0); p1 = fma(row[j + 1], vv[j + 1], p1); }
      if (j < n) p0 = fma(row[j], vv[j], p0);
      pi = (p0 + p1) * t;
    }
    const double a2 = -0.5 * t * wsum(pi * vi);
    const double wi = pi + a2 * vi;
    ww[lane] = wi;
    __syncthreads();
    if (lane > k && live)
      for (int j = k + 1; j < n; ++j) row[j] -= vi * ww[j] + wi * vv[j];
  }
  if (lane == n - 1) { dreg = row[n - 1]; ereg = 0.0; }
  __syncthreads();
  // ---- Q in place.  B[i'][j'] = A[i' + 1][j'] (i', j' < m = n - 1) holds reflector k below its diagonal in column k:
  //      dorg2r on B, then the columns move one to the right and row / column 0 become those of the unit matrix.
  const int m = n - 1;
  if (m >= 1) {
    const int kr = m - 1;                                     // reflectors 0 .. kr - 1
    double* B = A + LD;                                        // B(i, j) = B[i * LD + j]
    if (lane < m)
      for (int j = kr; j < m; ++j) B[lane * LD + j] = (lane == j) ? 1.0 : 0.0;   // columns kr .. m - 1 of the unit matrix
    for (int i = kr - 1; i >= 0; --i) {
      __syncthreads();
      const double t = pmd_readlane_f64(treg, i);
      // apply H(i) to B(i:m, i+1:m) from the left; lane = column j
      if (lane > i && lane < m) {
        double w0 = B[i * LD + lane], w1 = 0.0;                // v[i] = 1
        int r = i + 1;
        for (; r + 1 < m; r += 2) { w0 = fma(B[r * LD + i], B[r * LD + lane], w0); w1 = fma(B[(r + 1) * LD + i], B[(r + 1) * LD + lane], w1); }
        if (r < m) w0 = fma(B[r * LD + i], B[r * LD + lane], w0);
        const double wj = (w0 + w1) * t;
        B[i * LD + lane] -= wj;
        for (r = i + 1; r < m; ++r) B[r * LD + lane] -= wj * B[r * LD + i];
      }
      __syncthreads();
      // column i becomes H(i) e_i (lane = row)
      if (lane < m) {
        const double vr = B[lane * LD + i];
        B[lane * LD + i] = (lane < i) ? 0.0 : (lane == i) ? 1.0 - t : -t * vr;
      }
    }
    __syncthreads();
    // shift the columns one to the right (lane = row of A), unit first row and column
    if (lane >= 1 && live) {
      for (int j = m; j >= 1; --j) row[j] = row[j - 1];
      row[0] = 0.0;
    }
    __syncthreads();
    if (live) A[lane] = (lane == 0) ? 1.0 : 0.0;
  } else if (lane == 0) {
    A[0] = 1.0;
  }
  __syncthreads();
  // ---- implicit QL with accumulated rotations (EISPACK tql2); e[i] couples i and i + 1, e[n-1] = 0.
  //      From here on a lane touches its own row of A only.
  {
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; ++l) {
      const double dl0 = pmd_readlane_f64(dreg, l), el0 = pmd_readlane_f64(ereg, l);
      tst1 = fmax(tst1, fabs(dl0) + fabs(el0));
      // first m >= l with a negligible e[m] (e[n-1] = 0 guarantees one)
      const unsigned long long small = __ballot(live && lane >= l && !(fabs(ereg) > eps * tst1));
      const int mm = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(small | (1ull << (n - 1))));
      if (mm > l) {
        int iter = 0;
        double el = el0;
        do {
          ++iter;
          double gg = pmd_readlane_f64(dreg, l);
          const double dnext = pmd_readlane_f64(dreg, l + 1);
          double p = (dnext - gg) / (2.0 * el);
          double r = sqrt(p * p + 1.0);
          if (p < 0.0) r = -r;
          const double dl = el / (p + r);
          const double dl1 = el * (p + r);
          double h = gg - dl;
          if (lane == l) dreg = dl;
          else if (lane == l + 1) dreg = dl1;
          else if (lane >= l + 2 && live) dreg -= h;
          f += h;
          p = pmd_readlane_f64(dreg, mm);
          double c = 1.0, c2 = 1.0, c3 = 1.0, s_ = 0.0, s2 = 0.0;
          const double el1 = pmd_readlane_f64(ereg, l + 1);
          double zhi = live ? row[mm] : 0.0;                   // column i + 1 of this lane's row, carried between rotations
          for (int i = mm - 1; i >= l; --i) {
            const double zlo = live ? row[i] : 0.0;
            c3 = c2;
            c2 = c;
            s2 = s_;
            const double ei = pmd_readlane_f64(ereg, i), di = pmd_readlane_f64(dreg, i);
            gg = c * ei;
            h = c * p;
            const double q2 = p * p + ei * ei;
            const double rinv = (q2 > 0.0) ? rsqrt(q2) : 0.0;
            const double enew = s_ * (q2 * rinv);
            if (q2 > 0.0) { s_ = ei * rinv; c = p * rinv; }
            else { s_ = 0.0; c = 1.0; }
            p = c * di - s_ * gg;
            const double dnew = h + s_ * (c * gg + s_ * di);
            if (lane == i + 1) { ereg = enew; dreg = dnew; }
            if (live) row[i + 1] = s_ * zlo + c * zhi;
            zhi = c * zlo - s_ * zhi;
          }
          if (live) row[l] = zhi;
          p = -s_ * s2 * c3 * el1 * el / dl1;
          el = s_ * p;
          if (lane == l) { ereg = el; dreg = c * p; }
        } while (fabs(el) > eps * tst1 && iter < 80);
      }
      if (lane == l) { dreg += f; ereg = 0.0; }
    }
  }
  // ---- rank the eigenvalues (descending, ties by index), write out
  {
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const double lj = pmd_readlane_f64(dreg, j);
      rank += (lj > dreg) || (lj == dreg && j < lane);
    }
    if (live) order[rank] = lane;
  }
  __syncthreads();
  {
    // lane = output column c: eigenvector order[c], its eigenvalue and (mode 1) its scale; rows are walked in a loop so that
    // the global stores are contiguous
    const int src = live ? order[lane] : 0;
    const double lam = __shfl(dreg, src);
    const double lmax = pmd_readlane_f64(lam, 0);
    double scl = 1.0;
    if (mode == 1) scl = (lam > tol * lmax && lam > 0.0) ? 1.0 / sqrt(lam) : 0.0;
    double* no = Nout + (long)blockIdx.x * 4096;
    for (int r = 0; r < 64; ++r) no[r * 64 + lane] = (r < n && live) ? A[r * LD + src] * scl : 0.0;
    lam_out[(long)blockIdx.x * 64 + lane] = live ? lam : 0.0;
  }
}

int pmd_launch_small_eig(pmd_ctx* ctx, const double* G, int slices, int n, int mode, double tol, double* Nout,
                         double* lam_out, int n_tiles) {
  pmd_prof_scope prof__(ctx, "small_eig");
  if (n_tiles <= 0) return PMD_OK;
  if (n > 64 || n < 1) return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "small_eig", "n must be in [1, 64]");
  {
    // A/B: PMD_SMALL_EIG=rocsolver routes the batch through rocSOLVER's strided-batched dsyevd (wide.hip, rp = 64)
    static int lib_mode = -1;
    if (lib_mode < 0) { const char* e = getenv("PMD_SMALL_EIG"); lib_mode = (e && !strcmp(e, "rocsolver")) ? 1 : 0; }
    if (lib_mode) {
      const size_t need = pmd_wide_eig_workspace_bytes(64, n_tiles);
      if (ctx->scratch2_bytes < need) {
        PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->scratch2) (void)hipFree(ctx->scratch2);
        ctx->scratch2 = nullptr;
        ctx->scratch2_bytes = 0;
        PMD_HIP(ctx, hipMalloc(&ctx->scratch2, need));
        ctx->scratch2_bytes = need;
      }
      return pmd_launch_wide_eig(ctx, G, slices, 64, n, mode, tol, Nout, lam_out, n_tiles, ctx->scratch2, ctx->scratch2_bytes);
    }
  }
  {
    // default: the one-wave QL kernel; PMD_SMALL_EIG=jacobi restores the parallel Jacobi kernel (A/B runs)
    static int ql_mode = -1;
    if (ql_mode < 0) { const char* e = getenv("PMD_SMALL_EIG"); ql_mode = (e && !strcmp(e, "jacobi")) ? 0 : 1; }
    if (ql_mode) {
      const size_t lds = ((size_t)n * (n + 1) + 2 * 64) * sizeof(double) + 64 * sizeof(int) + 64;
      PMD_HIP(ctx, hipFuncSetAttribute((const void*)small_eig_ql_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(small_eig_ql_kernel, dim3(n_tiles), dim3(64), lds, ctx->stream, G, (long)slices * 4096, slices, n, mode, tol,
                         Nout, lam_out);
      PMD_LAUNCH_CHECK(ctx, "small_eig_ql_kernel");
      return PMD_OK;
    }
  }
  const size_t bytes = (size_t)2 * 64 * EIG_LD * sizeof(double) + 64 * sizeof(double) + (64 + 2 + 64) * sizeof(int) + 64;
  static int threads = 0;
  if (!threads) {
    const char* e = getenv("PMD_EIG_THREADS");
    threads = e ? atoi(e) : 1024;
    if (threads != 256 && threads != 512 && threads != 1024) threads = 1024;
  }
#define PMD_EIG_LAUNCH(TH)                                                                                                      \
  do {                                                                                                                          \
    PMD_HIP(ctx, hipFuncSetAttribute((const void*)small_eig_kernel<TH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)); \
    hipLaunchKernelGGL(small_eig_kernel<TH>, dim3(n_tiles), dim3(TH), bytes, ctx->stream, G, (long)slices * 4096, slices, n,    \
                       mode, tol, Nout, lam_out);                                                                               \
  } while (0)
  if (threads == 1024) PMD_EIG_LAUNCH(1024);
  else if (threads == 512) PMD_EIG_LAUNCH(512);
  else PMD_EIG_LAUNCH(256);  // 83 / 96 / 130 ms per step with 1024 / 512 / 256 threads on the 16 129-tile workload: latency, not fp64 rate, bounds a step
#undef PMD_EIG_LAUNCH
  PMD_LAUNCH_CHECK(ctx, "small_eig_kernel");
  return PMD_OK;
}

// ---------------------------------------------------------------- batched Cholesky whitening
// Same interface as small_eig mode 1 where only an ORTHONORMAL BASIS is wanted, not the eigenvectors: G = R^T R
// (upper R), Nout[tile][c'][c] = (R^{-1})[c'][c], so that tile_rowmix turns rows with Gram matrix G into orthonormal
// rows (new row c = sum_c' N[c'][c] old row c').  A pivot <= tol * max diagonal marks a direction that depends on the
// earlier ones: its column is zero (the counterpart of small_eig's lambda <= tol * lambda_max).  One wave per problem,
// ~50 us against ~0.7 ms for the Jacobi solver: two of the four factorisations of single_block_md
// (decomposition.py:301, :315-317) only feed spans - span(S) does not depend on the basis of the row space of V_ds, and
// U = U0 Wl does not depend on the basis U0 of span(S) - and take this path when no denoiser hook reads the vectors.
__global__ __launch_bounds__(64) void small_chol_kernel(const double* __restrict__ G, long g_tile_stride, int slices, int n,
                                                        double tol, double* __restrict__ Nout) {
  __shared__ double R[64][65];
  __shared__ double Ri[64][65];
  __shared__ int dead[64];
  const int t = threadIdx.x;
  const double* g = G + (long)blockIdx.x * g_tile_stride;
  for (int i = 0; i < 64; ++i) {
    double s = 0.0;
    if (i < n && t < n)
      for (int k = 0; k < slices; ++k) s += g[(long)k * 4096 + i * 64 + t] + g[(long)k * 4096 + t * 64 + i];
    R[i][t] = 0.5 * s;
    Ri[i][t] = 0.0;
  }
  __syncthreads();
  double dmax = 0.0;
  for (int i = 0; i < n; ++i) dmax = fmax(dmax, R[i][i]);
  // right-looking Cholesky, upper factor stored in R (row k = R[k][k:])
  for (int k = 0; k < n; ++k) {
    const double piv = R[k][k];
    const bool bad = !(piv > tol * dmax);
    if (t == 0) dead[k] = bad;
    const double rkk = bad ? 1.0 : sqrt(piv);
    double rkt = 0.0;
    if (t >= k && t < n) rkt = bad ? ((t == k) ? 1.0 : 0.0) : R[k][t] / rkk;
    __syncthreads();
    if (t >= k && t < n) R[k][t] = rkt;
    __syncthreads();
    // trailing update: R[i][j] -= R[k][i] * R[k][j], i, j > k; thread t owns column j = t
    if (!bad && t > k && t < n)
      for (int i = k + 1; i <= t; ++i) R[i][t] -= R[k][i] * rkt;
    __syncthreads();
  }
  // invert the upper-triangular factor: thread t solves column t of R X = I
  if (t < n) {
    for (int i = t; i >= 0; --i) {
      double s = (i == t) ? 1.0 : 0.0;
      for (int j = i + 1; j <= t; ++j) s -= R[i][j] * Ri[j][t];
      Ri[i][t] = s / R[i][i];
    }
  }
  __syncthreads();
  double* no = Nout + (long)blockIdx.x * 4096;
  for (int i = 0; i < 64; ++i) {
    double v = (i < n && t < n) ? Ri[i][t] : 0.0;
    if (t < n && dead[t]) v = 0.0;
    no[i * 64 + t] = v;
  }
}

int pmd_launch_small_chol(pmd_ctx* ctx, const double* G, int slices, int n, double tol, double* Nout, int n_tiles) {
  pmd_prof_scope prof__(ctx, "small_chol");
  if (n_tiles <= 0) return PMD_OK;
  if (n > 64 || n < 1) return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "small_chol", "n must be in [1, 64]");
  hipLaunchKernelGGL(small_chol_kernel, dim3(n_tiles), dim3(64), 0, ctx->stream, G, (long)slices * 4096, slices, n, tol, Nout);
  PMD_LAUNCH_CHECK(ctx, "small_chol_kernel");
  return PMD_OK;
}

// ---------------------------------------------------------------- pooled basis expansion ---
// Out[tile][c][q] = In[tile][c][pool_idx[q]] * pool_w[q]   (U_ds^T composed with the pooling map)
__global__ void expand_pooled_kernel(const float* __restrict__ In, long in_tile_stride, int in_ld,
                                     const int* __restrict__ pool_idx, const float* __restrict__ pool_w, int d, int r,
                                     float* __restrict__ Out, long out_tile_stride, int out_ld) {
  const int tile = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < r * d; i += gridDim.x * blockDim.x) {
    const int c = i / d, q = i - c * d;
    Out[(long)tile * out_tile_stride + (long)c * out_ld + q] =
        In[(long)tile * in_tile_stride + (long)c * in_ld + pool_idx[q]] * pool_w[q];
  }
}

int pmd_launch_expand_pooled(pmd_ctx* ctx, const float* In, long in_tile_stride, int in_ld, const int* pool_idx,
                             const float* pool_w, int d, int r, float* Out, long out_tile_stride, int out_ld,
                             int n_tiles) {
  pmd_prof_scope prof__(ctx, "expand_pooled");
  int bx = (r * d + 255) / 256;
  if (bx > 32) bx = 32;
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    hipLaunchKernelGGL(expand_pooled_kernel, dim3(bx, tn), dim3(256), 0, ctx->stream, In + (long)t0 * in_tile_stride,
                       in_tile_stride, in_ld, pool_idx, pool_w, d, r, Out + (long)t0 * out_tile_stride,
                       out_tile_stride, out_ld);
    PMD_LAUNCH_CHECK(ctx, "expand_pooled_kernel");
  }
  return PMD_OK;
}

// ---------------------------------------------------------------- roughness statistics ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// evaluation.py:84-111.  Ut: [tile][comp][q], q = i + b1*j.  One wave per (tile, comp).
__global__ __launch_bounds__(64) void spatial_stat_kernel(const float* __restrict__ Ut, long tile_stride, int ld, int b1,
                                                          int b2, float* __restrict__ stats, int rp) {
  const int tile = blockIdx.y, comp = blockIdx.x;
  const float* u = Ut + (long)tile * tile_stride + (long)comp * ld;
  const int d = b1 * b2;
  float sv = 0.f, sh = 0.f, sa = 0.f;
  for (int q = threadIdx.x; q < d; q += 64) {
    const float x = u[q];
    const int i = q % b1;
    sa += fabsf(x);
    if (i + 1 < b1) sv += fabsf(u[q + 1] - x);
    if (q + b1 < d) sh += fabsf(x - u[q + b1]);
  }
  sv = wave_sum(sv); sh = wave_sum(sh); sa = wave_sum(sa);
  if (threadIdx.x == 0) {
    const float avg_diff = (sv + sh) / (float)((b1 - 1) * b2 + b1 * (b2 - 1));
    const float avg_elem = sa / (float)d;
    stats[((long)tile * rp + comp) * 2 + 0] = avg_diff / avg_elem;
  }
}

// evaluation.py:114-126.  V: [tile][comp][t].  One workgroup per (tile, comp).
__global__ __launch_bounds__(256) void temporal_stat_kernel(const float* __restrict__ V, long tile_stride, long ld, int T,
                                                            float* __restrict__ stats, int rp) {
  __shared__ float red[2][4];
  const int tile = blockIdx.y, comp = blockIdx.x;
  const float* v = V + (long)tile * tile_stride + (long)comp * ld;
  float num = 0.f, den = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    const float x = v[t];
    den += fabsf(x);
    if (t >= 1 && t + 1 < T) num += fabsf(v[t - 1] + v[t + 1] - 2.0f * x);
  }
  num = wave_sum(num); den = wave_sum(den);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = num; red[1][threadIdx.x >> 6] = den; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float n4 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float d4 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    stats[((long)tile * rp + comp) * 2 + 1] = (n4 / (float)(T - 2)) / (d4 / (float)T);
  }
}

int pmd_launch_stats_roughness(pmd_ctx* ctx, const float* Ut, long u_tile_stride, int u_ld, int b1, int b2,
                               const float* V, long v_tile_stride, long v_ld, int T, int r, float* stats,
                               int n_tiles, int rp) {
  pmd_prof_scope prof__(ctx, "roughness");
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    if (Ut) {
      hipLaunchKernelGGL(spatial_stat_kernel, dim3(r, tn), dim3(64), 0, ctx->stream, Ut + (long)t0 * u_tile_stride,
                         u_tile_stride, u_ld, b1, b2, stats + (long)t0 * rp * 2, rp);
      PMD_LAUNCH_CHECK(ctx, "spatial_stat_kernel");
    }
    if (V) {
      hipLaunchKernelGGL(temporal_stat_kernel, dim3(r, tn), dim3(256), 0, ctx->stream, V + (long)t0 * v_tile_stride,
                         v_tile_stride, v_ld, T, stats + (long)t0 * rp * 2, rp);
      PMD_LAUNCH_CHECK(ctx, "temporal_stat_kernel");
    }
  }
  return PMD_OK;
}

// evaluation.py:133-164 + :195-222.  keep[tile][c] in {0,1}; ranks[tile] = number kept (capped).
__global__ void decide_kernel(const float* __restrict__ stats, int r, float thr_s, float thr_t, int max_fail, int cap,
                              int n_tiles, int* __restrict__ good, int* __restrict__ keep, int* __restrict__ ranks, int rp) {
  const int tile = blockIdx.x * blockDim.x + threadIdx.x;
  if (tile >= n_tiles) return;
  int fails = 0, count = 0;
  bool all_fails = false;
  for (int c = 0; c < rp; ++c) {
    int g = 0, k = 0;
    if (c < r) {
      const float sp = stats[((long)tile * rp + c) * 2 + 0];
      const float tp = stats[((long)tile * rp + c) * 2 + 1];
      g = (sp < thr_s) && (tp < thr_t);
      if (all_fails) k = 0;
      else if (!g) {
        fails++;
        k = 1;
        if (fails == max_fail) all_fails = true;
      } else {
        fails = 0;
        k = 1;
      }
      if (k && count >= cap) k = 0;
      count += k;
    }
    good[(long)tile * rp + c] = g;
    keep[(long)tile * rp + c] = k;
  }
  ranks[tile] = count;
}

int pmd_launch_decide(pmd_ctx* ctx, const float* stats, int r, float thr_s, float thr_t, int max_fail, int cap,
                      int n_tiles, int* good, int* keep, int* ranks, int rp) {
  pmd_prof_scope prof__(ctx, "decide");
  hipLaunchKernelGGL(decide_kernel, dim3((n_tiles + 63) / 64), dim3(64), 0, ctx->stream, stats, r, thr_s, thr_t,
                     max_fail, cap, n_tiles, good, keep, ranks, rp);
  PMD_LAUNCH_CHECK(ctx, "decide_kernel");
  return PMD_OK;
}

// ---------------------------------------------------------------- residual-window helpers --
// (single_residual_block_md, decomposition.py:364-387)

// cross Gram of two [comp][x] arrays: Gx[tile][c][c'] = sum_x A[tile][c][x] * B[tile][c'][x]  (fp64)
__global__ __launch_bounds__(256) void tile_cross_gram_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              long tile_stride, int ld, int len,
                                                              double* __restrict__ G) {
  __shared__ float sa[64][33];
  __shared__ float sb[64][33];
  const int tile = blockIdx.x;
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  const float* a = A + (long)tile * tile_stride;
  const float* b = B + (long)tile * tile_stride;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int x0 = 0; x0 < len; x0 += 32) {
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
      const int r = i >> 5, cx = i & 31;
      const bool in = x0 + cx < len;
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
  double* g = G + (long)tile * 4096;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) g[(4 * ti + i) * 64 + 4 * tj + j] = acc[i][j];
}

int pmd_launch_tile_cross_gram(pmd_ctx* ctx, const float* A, const float* B, long tile_stride, int ld, int len,
                               double* G, int n_tiles) {
  pmd_prof_scope prof__(ctx, "tile_cross_gram");
  if (n_tiles <= 0) return PMD_OK;
  hipLaunchKernelGGL(tile_cross_gram_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, A, B, tile_stride, ld, len, G);
  PMD_LAUNCH_CHECK(ctx, "tile_cross_gram_kernel");
  return PMD_OK;
}

// out[tile][q][x] = X[pix[tile][q]][x] - sum_{c<r} E[tile][c][q] * W[tile][c][x]      (x < len)
__global__ __launch_bounds__(256) void tile_residual_rows_kernel(const float* __restrict__ X, long ldx,
                                                                 const int* __restrict__ pix, int d,
                                                                 const float* __restrict__ E, int e_ld,
                                                                 const float* __restrict__ W, long w_ld, int r, int len,
                                                                 float* __restrict__ out, long out_ld, int rp) {
  const int tile = blockIdx.z, q = blockIdx.y;
  const float* e = E + (long)tile * rp * e_ld + q;
  const float* w = W + (long)tile * rp * w_ld;
  const float* xr = X + (long)pix[(long)tile * d + q] * ldx;
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < len; x += gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int c = 0; c < r; ++c) acc = fmaf(e[(long)c * e_ld], w[(long)c * w_ld + x], acc);
    out[((long)tile * d + q) * out_ld + x] = xr[x] - acc;
  }
}

int pmd_launch_tile_residual_rows(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int d, const float* E,
                                  int e_ld, const float* W, long w_ld, int r, int len, float* out, long out_ld,
                                  int n_tiles, int rp) {
  pmd_prof_scope prof__(ctx, "tile_residual_rows");
  if (n_tiles <= 0) return PMD_OK;
  int bx = (len + 255) / 256;
  if (bx > 4) bx = 4;
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    hipLaunchKernelGGL(tile_residual_rows_kernel, dim3(bx, d, tn), dim3(256), 0, ctx->stream, X, ldx, pix + (long)t0 * d, d,
                       E + (long)t0 * rp * e_ld, e_ld, W + (long)t0 * rp * w_ld, w_ld, r, len,
                       out + (long)t0 * d * out_ld, out_ld, rp);
    PMD_LAUNCH_CHECK(ctx, "tile_residual_rows_kernel");
  }
  return PMD_OK;
}

// a[tile][c][x] -= b[tile][c][x]   (c < 64, x < len)
__global__ void tile_sub_kernel(float* __restrict__ a, const float* __restrict__ b, long tile_stride, int ld, int len, int rp) {
  const int tile = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rp * len; i += gridDim.x * blockDim.x) {
    const int c = i / len, x = i - c * len;
    a[(long)tile * tile_stride + (long)c * ld + x] -= b[(long)tile * tile_stride + (long)c * ld + x];
  }
}

int pmd_launch_tile_sub(pmd_ctx* ctx, float* a, const float* b, long tile_stride, int ld, int len, int n_tiles, int rp) {
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    hipLaunchKernelGGL(tile_sub_kernel, dim3(16, tn), dim3(256), 0, ctx->stream, a + (long)t0 * tile_stride,
                       b + (long)t0 * tile_stride, tile_stride, ld, len, rp);
    PMD_LAUNCH_CHECK(ctx, "tile_sub_kernel");
  }
  return PMD_OK;
}

// decisions of a residual window + append (decomposition.py:501-515): the kept prefix of Unew (capped by
// the remaining capacity) is copied behind the counts[tile] components already in Ucur.
__global__ __launch_bounds__(256) void tile_append_kernel(const float* __restrict__ stats, int r, float thr_s,
                                                          float thr_t, int max_fail, int cap,
                                                          const float* __restrict__ Unew, float* __restrict__ Ucur,
                                                          int ld, int* __restrict__ counts, int* __restrict__ good,
                                                          int* __restrict__ keep, int rp) {
  __shared__ int s_take, s_base;
  const int tile = blockIdx.x;
  if (threadIdx.x == 0) {
    int fails = 0, kept = 0;
    bool all_fails = false;
    for (int c = 0; c < rp; ++c) {
      int g = 0, k = 0;
      if (c < r) {
        const float sp = stats[((long)tile * rp + c) * 2 + 0];
        const float tp = stats[((long)tile * rp + c) * 2 + 1];
        g = (sp < thr_s) && (tp < thr_t);
        if (all_fails) k = 0;
        else if (!g) { fails++; k = 1; if (fails == max_fail) all_fails = true; }
        else { fails = 0; k = 1; }
        kept += k;
      }
      good[(long)tile * rp + c] = g;
      keep[(long)tile * rp + c] = k;
    }
    const int base = counts[tile];
    const int remaining = cap - base;
    s_take = kept < remaining ? kept : remaining;
    s_base = base;
    counts[tile] = base + s_take;
  }
  __syncthreads();
  const int take = s_take, base = s_base;
  for (int i = threadIdx.x; i < take * ld; i += 256) {
    const int c = i / ld, x = i - c * ld;
    Ucur[(long)tile * rp * ld + (long)(base + c) * ld + x] = Unew[(long)tile * rp * ld + (long)c * ld + x];
  }
}

int pmd_launch_tile_append(pmd_ctx* ctx, const float* stats, int r, float thr_s, float thr_t, int max_fail, int cap,
                           const float* Unew, float* Ucur, int ld, int* counts, int* good, int* keep, int n_tiles, int rp) {
  if (n_tiles <= 0) return PMD_OK;
  hipLaunchKernelGGL(tile_append_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, stats, r, thr_s, thr_t, max_fail, cap,
                     Unew, Ucur, ld, counts, good, keep, rp);
  PMD_LAUNCH_CHECK(ctx, "tile_append_kernel");
  return PMD_OK;
}

// rows c >= counts[tile] of U[tile][c][:] are cleared (components that were not kept)
__global__ void tile_truncate_kernel(float* __restrict__ U, int ld, const int* __restrict__ counts, int rp) {
  const int tile = blockIdx.y;
  const int k = counts[tile];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (rp - k) * ld; i += gridDim.x * blockDim.x)
    U[(long)tile * rp * ld + (long)k * ld + i] = 0.f;
}

int pmd_launch_tile_truncate(pmd_ctx* ctx, float* U, int ld, const int* counts, int n_tiles, int rp) {
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    hipLaunchKernelGGL(tile_truncate_kernel, dim3(16, tn), dim3(256), 0, ctx->stream, U + (long)t0 * rp * ld, ld, counts + t0, rp);
    PMD_LAUNCH_CHECK(ctx, "tile_truncate_kernel");
  }
  return PMD_OK;
}
