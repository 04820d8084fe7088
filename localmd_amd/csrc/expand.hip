// Expansion of a decomposition back into pixels (pmdarray.py:132-171, PMDArray.__getitem__):
//   out[frame][pixel] = std[pixel] * sum_k U[pixel][k] * C[k][frame] + mean[pixel]
// with U the sparse spatial matrix (CSR) and C = (R diag(s)) Vt[:, frames].  Two kernels:
//   csr_rows_spmm   : acc[p][f] = sum_i data[i] * B[indices[i]][f] over the nonzeros of row rows[p]
//                     (B = C, or B = R diag(s) when the dense product with Vt comes second)
//   transpose_affine: out[f][p] = acc[p][f] * scale[p] + shift[p]
// Both are bandwidth-bound: the nonzeros of a row are workgroup-uniform (scalar loads), the rows
// of B are read coalesced along the frame axis and neighbouring pixels share their tiles'
// columns, so B stays in L2.
#include "pmd_internal.h"

namespace {

template <int VEC>
__global__ __launch_bounds__(256) void csr_rows_spmm_kernel(const long* __restrict__ indptr, const int* __restrict__ indices,
                                                            const float* __restrict__ data, const int* __restrict__ rows,
                                                            const float* __restrict__ B, long ldb, int ncols,
                                                            float* __restrict__ out, long ldo) {
  const long p = blockIdx.x;
  const int row = rows ? rows[p] : (int)p;
  const long lo = indptr[row], hi = indptr[row + 1];
  const int f = (blockIdx.y * 256 + threadIdx.x) * VEC;
  const bool ok = f < ncols;
  const int fc = ok ? f : 0;  // idle lanes read a valid column: no branch around the loads
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  constexpr int UN = 8;
  long i = lo;
  for (; i + UN <= hi; i += UN) {
    float b[UN][VEC], a[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      a[u] = data[i + u];
      const float* bp = B + (long)indices[i + u] * ldb + fc;
      if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(bp);
        b[u][0] = t.x; b[u][1] = t.y; b[u][2] = t.z; b[u][3] = t.w;
      } else {
        b[u][0] = bp[0];
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] += a[u] * b[u][v];
  }
  for (; i < hi; ++i) {
    const float a = data[i];
    const float* bp = B + (long)indices[i] * ldb + fc;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] += a * bp[v];
  }
  if (!ok) return;
  float* op = out + p * ldo + f;
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(op) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
    op[0] = acc[0];
  }
}

__global__ __launch_bounds__(256) void transpose_affine_kernel(const float* __restrict__ src, long lds_, long rows, int cols,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               float* __restrict__ dst, long ldd) {
  __shared__ float t[32][33];
  const long y0 = (long)blockIdx.x * 32;  // rows of src (pixels): the long axis goes on grid.x
  const int x0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const long y = y0 + r;
    if (y < rows && x0 + tx < cols) {
      const float sc = scale ? scale[y] : 1.f, sh = shift ? shift[y] : 0.f;
      t[r][tx] = src[y * lds_ + x0 + tx] * sc + sh;
    }
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (x0 + r < cols && y0 + tx < rows) dst[(long)(x0 + r) * ldd + y0 + tx] = t[tx][r];
}

}  // namespace

int pmd_csr_rows_spmm_impl(pmd_ctx* ctx, const long* indptr, const int* indices, const float* data, const int* rows,
                           long n_sel, const float* B, long ldb, int ncols, float* out, long ldo) {
  if (n_sel <= 0 || ncols <= 0) return PMD_OK;
  if (n_sel > 2147483647L) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_csr_rows_spmm", "too many rows in one call");
  const bool vec = ncols >= 256 && ldb % 4 == 0 && ldo % 4 == 0 && ncols % 4 == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)out & 15) == 0;
  if (vec) {
    const unsigned gy = (unsigned)((ncols / 4 + 255) / 256);
    if (gy > 65535) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_csr_rows_spmm", "too many columns in one call");
    hipLaunchKernelGGL(csr_rows_spmm_kernel<4>, dim3((unsigned)n_sel, gy), dim3(256), 0, ctx->stream, indptr, indices, data,
                       rows, B, ldb, ncols, out, ldo);
  } else {
    const unsigned gy = (unsigned)((ncols + 255) / 256);
    if (gy > 65535) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_csr_rows_spmm", "too many columns in one call");
    hipLaunchKernelGGL(csr_rows_spmm_kernel<1>, dim3((unsigned)n_sel, gy), dim3(256), 0, ctx->stream, indptr, indices, data,
                       rows, B, ldb, ncols, out, ldo);
  }
  PMD_LAUNCH_CHECK(ctx, "csr_rows_spmm_kernel");
  return PMD_OK;
}

int pmd_transpose_affine_impl(pmd_ctx* ctx, const float* src, long lds_, long rows, int cols, const float* scale,
                              const float* shift, float* dst, long ldd) {
  if (rows <= 0 || cols <= 0) return PMD_OK;
  const long gx = (rows + 31) / 32;
  const unsigned gy = (unsigned)((cols + 31) / 32);
  if (gx > 2147483647L || gy > 65535) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_transpose_affine", "matrix too large for one call");
  hipLaunchKernelGGL(transpose_affine_kernel, dim3((unsigned)gx, gy), dim3(256), 0, ctx->stream, src, lds_, rows, cols, scale,
                     shift, dst, ldd);
  PMD_LAUNCH_CHECK(ctx, "transpose_affine_kernel");
  return PMD_OK;
}
