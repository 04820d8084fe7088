// Large fp32 products on the fp16 matrix cores (v_mfma_f32_32x32x16_f16 through hipBLASLt: 2.5 PFLOP/s dense on MI355X
// against 157 TFLOP/s for fp32 MFMA), with fp32-level accuracy.
//
// Every operand X (fp32) is written once as two fp16 arrays with a power-of-two scale 2^-e chosen from max |X|:
//     X 2^-e = H1 + 2^-11 H2,   H1 = fp16(X 2^-e),   H2 = fp16((X 2^-e - H1) 2^11)
// H1 carries the leading 11 significant bits, H2 the next 11 (the residual is exact in fp32 and is scaled back into the
// normal fp16 range before rounding, so entries down to 2^-38 of the largest keep all their bits):
//     |X 2^-e - H1 - 2^-11 H2| <= 2^-24 |X 2^-e|,   the rounding error of an fp32 number.
// The product is then three fp16 x fp16 -> fp32 matrix products accumulated in fp32, smallest first:
//     A B = 2^(ea + eb) [ 2^-11 (A1 B2 + A2 B1) + A1 B1 ]  +  O(2^-22 |A| |B|) entrywise before accumulation
// (the dropped term 2^-22 A2 B2 is below the representation error).  Products of fp16 numbers are exact in fp32; all
// scales are powers of two.  Measured against a double-precision product (scripts/bf16x3_probe.hip, 768 x 640 outputs,
// inner dimension 55 808, relative Frobenius error): this path 4.1e-7 (random signs) / 1.2e-7 (all terms positive) /
// 4.3e-7 (columns spread over six decades); rocBLAS sgemm 2.4e-6 / 1.2e-6 / 2.4e-6 with one accumulation chain and
// 5.9e-7 / 1.3e-7 / 5.9e-7 with the chain cut every 1024 terms (pmd_gemm_rm's fp32 path).  Rates on the headline shapes,
// both splits included: 32 ms against 76-80 ms (157 ms for the transposed-A form).
//
// The two exponents are read back on the host (one 8-byte copy and one stream synchronisation per product, ~30 us): the
// scalars of hipBLASLt's device-pointer mode are not honoured by the kernels picked for these shapes on this stack.
// Operands holding Inf / NaN, all-zero operands and fp32-subnormal maxima go to the fp32 path.
#include "pmd_internal.h"
#include <hipblaslt/hipblaslt.h>
#include <cmath>
#include <map>
#include <tuple>

#define PMD_LT(ctx, call)                                                                               \
  do {                                                                                                  \
    hipblasStatus_t s__ = (call);                                                                       \
    if (s__ != HIPBLAS_STATUS_SUCCESS) {                                                                \
      char buf__[64];                                                                                   \
      snprintf(buf__, sizeof(buf__), "hipBLASLt status %d", (int)s__);                                  \
      return pmd_fail(ctx, PMD_ERR_BLAS, #call, buf__);                                                 \
    }                                                                                                   \
  } while (0)

#define RUN_OK(call)                 \
  do {                               \
    int rc__ = (call);               \
    if (rc__ != PMD_OK) return rc__; \
  } while (0)

namespace {

struct lt_plan {
  hipblasLtMatmulDesc_t desc;
  hipblasLtMatrixLayout_t la, lb, lc;
  hipblasLtMatmulAlgo_t algo;
  bool ok;
};

using lt_key = std::tuple<int, int, int, int, int, long, long, long>;

struct f16x2_state {
  hipblasLtHandle_t lt = nullptr;
  void* ws = nullptr;
  size_t ws_bytes = 0;
  unsigned* amax_dev = nullptr;    // [2] maxima, then [2][ABSMAX_BLOCKS] per-workgroup partial maxima
  unsigned* amax_host = nullptr;   // pinned [2]
  std::map<lt_key, lt_plan> plans;
};

constexpr size_t LT_WORKSPACE = 64u << 20;
constexpr int ABSMAX_BLOCKS = 2048;

}  // namespace

// ---------------------------------------------------------------------------------------------
// largest magnitude (as raw bits: for non-negative floats the integer order is the float order, NaN > Inf > finite)
__global__ __launch_bounds__(256) void f16x2_absmax_kernel(const float* __restrict__ x, int rows, int cols, long ld, int vec,
                                                           unsigned* __restrict__ out) {
  // vec: rows start on 16-byte boundaries (base aligned, ld % 4 == 0): float4 loads over the first cols / 4 * 4 entries of a
  // row, the 0-3 entries behind them one by one.  Four independent running maxima: a single one serialises the loads
  // behind its dependency chain (measured: 100 GB/s on a 9999-column operand against 4 TB/s).
  unsigned m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  const int c4 = vec ? cols / 4 : 0;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float* row = x + (long)r * ld;
    const float4* row4 = reinterpret_cast<const float4*>(row);
    for (int c = threadIdx.x; c < c4; c += 256) {
      const float4 v = row4[c];
      m0 = max(m0, __float_as_uint(v.x) & 0x7fffffffu);
      m1 = max(m1, __float_as_uint(v.y) & 0x7fffffffu);
      m2 = max(m2, __float_as_uint(v.z) & 0x7fffffffu);
      m3 = max(m3, __float_as_uint(v.w) & 0x7fffffffu);
    }
    int c = 4 * c4 + threadIdx.x;
    for (; c + 768 < cols; c += 1024) {
      m0 = max(m0, __float_as_uint(row[c]) & 0x7fffffffu);
      m1 = max(m1, __float_as_uint(row[c + 256]) & 0x7fffffffu);
      m2 = max(m2, __float_as_uint(row[c + 512]) & 0x7fffffffu);
      m3 = max(m3, __float_as_uint(row[c + 768]) & 0x7fffffffu);
    }
    for (; c < cols; c += 256) m0 = max(m0, __float_as_uint(row[c]) & 0x7fffffffu);
  }
  unsigned m = max(max(m0, m1), max(m2, m3));
  // one partial per workgroup, no atomics: device-scope atomics on one address serialise at ~0.5 us each across the XCDs
  // (16 384 of them made this kernel 8.7 ms on a 2.2 GB operand; measured)
  __shared__ unsigned wmax[4];
  for (int o = 32; o; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
}

__global__ __launch_bounds__(256) void f16x2_absmax_final_kernel(const unsigned* __restrict__ part, int n, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (int i = threadIdx.x; i < n; i += 256) m = max(m, part[i]);
  __shared__ unsigned wmax[4];
  for (int o = 32; o; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) *out = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
}

struct half4_t { _Float16 x, y, z, w; };

__device__ inline void f16x2_pieces(float v, _Float16& a, _Float16& b, _Float16& c) {
  a = (_Float16)v;
  const float r = (v - (float)a) * 2048.f;       // exact: the residual has at most 13 significant bits
  b = (_Float16)r;
  c = (_Float16)((r - (float)b) * 2048.f);       // exact: at most 2 bits are left - three pieces hold an fp32 number exactly
}

// h3 != NULL: third piece (X 2^-e = h1 + 2^-11 h2 + 2^-22 h3, exact)
__global__ __launch_bounds__(256) void f16x2_split_kernel(const float* __restrict__ x, int rows, int cols, long ld, int vec, float scale,
                                                          _Float16* __restrict__ h1, _Float16* __restrict__ h2, _Float16* __restrict__ h3,
                                                          long ldh) {
  const int c4 = vec ? cols / 4 : 0;   // (ldh is a multiple of 8: the pieces' rows are 16-byte aligned too)
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float* row = x + (long)r * ld;
    _Float16* o1 = h1 + (long)r * ldh;
    _Float16* o2 = h2 + (long)r * ldh;
    _Float16* o3 = h3 ? h3 + (long)r * ldh : nullptr;
    const float4* row4 = reinterpret_cast<const float4*>(row);
    half4_t* p1 = reinterpret_cast<half4_t*>(o1);
    half4_t* p2 = reinterpret_cast<half4_t*>(o2);
    half4_t* p3 = reinterpret_cast<half4_t*>(o3);
    for (int c = threadIdx.x; c < c4; c += 256) {
      const float4 v = row4[c];
      half4_t a, b, d;
      f16x2_pieces(v.x * scale, a.x, b.x, d.x);
      f16x2_pieces(v.y * scale, a.y, b.y, d.y);
      f16x2_pieces(v.z * scale, a.z, b.z, d.z);
      f16x2_pieces(v.w * scale, a.w, b.w, d.w);
      p1[c] = a;
      p2[c] = b;
      if (o3) p3[c] = d;
    }
    for (int c = 4 * c4 + threadIdx.x; c < cols; c += 256) {
      _Float16 a, b, d;
      f16x2_pieces(row[c] * scale, a, b, d);
      o1[c] = a;
      o2[c] = b;
      if (o3) o3[c] = d;
    }
  }
}

// ---------------------------------------------------------------------------------------------
static f16x2_state* state_of(pmd_ctx* ctx) { return (f16x2_state*)ctx->f16x2; }

static int ensure_state(pmd_ctx* ctx) {
  if (ctx->f16x2) return PMD_OK;
  f16x2_state* st = new f16x2_state();
  if (hipblasLtCreate(&st->lt) != HIPBLAS_STATUS_SUCCESS) { delete st; return pmd_fail(ctx, PMD_ERR_BLAS, "hipblasLtCreate", "failed"); }
  if (hipMalloc(&st->ws, LT_WORKSPACE) != hipSuccess || hipMalloc((void**)&st->amax_dev, (2 + 2 * ABSMAX_BLOCKS) * sizeof(unsigned)) != hipSuccess ||
      hipHostMalloc((void**)&st->amax_host, 2 * sizeof(unsigned)) != hipSuccess) {
    (void)hipGetLastError();
    if (st->ws) (void)hipFree(st->ws);
    if (st->amax_dev) (void)hipFree(st->amax_dev);
    hipblasLtDestroy(st->lt);
    delete st;
    return pmd_fail(ctx, PMD_ERR_HIP, "gemm_f16x2", "allocation failed");
  }
  st->ws_bytes = LT_WORKSPACE;
  ctx->f16x2 = st;
  return PMD_OK;
}

void pmd_f16x2_destroy(pmd_ctx* ctx) {
  f16x2_state* st = state_of(ctx);
  if (!st) return;
  for (auto& kv : st->plans) {
    if (!kv.second.ok) continue;
    hipblasLtMatrixLayoutDestroy(kv.second.la);
    hipblasLtMatrixLayoutDestroy(kv.second.lb);
    hipblasLtMatrixLayoutDestroy(kv.second.lc);
    hipblasLtMatmulDescDestroy(kv.second.desc);
  }
  if (st->ws) (void)hipFree(st->ws);
  if (st->amax_dev) (void)hipFree(st->amax_dev);
  if (st->amax_host) (void)hipHostFree(st->amax_host);
  if (st->lt) hipblasLtDestroy(st->lt);
  delete st;
  ctx->f16x2 = nullptr;
}

// which products take this path: PMD_GEMM_SPLIT=0 none; PMD_GEMM_SPLIT_MIN_GFLOP (default 100) sets the size gate
bool pmd_f16x2_wanted(const pmd_ctx* ctx, int m, int n, int k) {
  if (!ctx->gemm_split) return false;
  if (m < ctx->gemm_split_min_dim || n < ctx->gemm_split_min_dim || k < ctx->gemm_split_min_dim) return false;
  return 2.0 * m * (double)n * k >= ctx->gemm_split_min_flop;
}

static inline int vec_ok(const void* p, int cols, long ld) { (void)cols; return (ld % 4 == 0) && (((uintptr_t)p & 15) == 0); }

long pmd_f16x2_ld(int cols) { return pmd_round_up(cols, 8); }

size_t pmd_f16x2_bytes(int rows, int cols, int pieces) { return pieces * sizeof(_Float16) * (size_t)rows * pmd_f16x2_ld(cols) + 256; }

// Splits up to two operands with ONE read-back of their maxima.  h1 of operand i = (char*)buf_i, h2 follows it.
// *usable = 0 when an operand cannot take the path (Inf / NaN, all zero, subnormal maximum): nothing was written then.
// Exponents e_i with max |X_i| 2^-e_i in [2^13, 2^14) for up to two operands, with ONE read-back.  *usable = 0: an operand
// cannot take the path (Inf / NaN, all zero, subnormal maximum).
int pmd_f16x2_exponents(pmd_ctx* ctx, int count, const float* const* X, const int* rows, const int* cols, const long* ld, int* e_out,
                        int* usable) {
  RUN_OK(ensure_state(ctx));
  f16x2_state* st = state_of(ctx);
  *usable = 0;
  if (count < 1 || count > 2) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_f16x2_exponents", "one or two operands");
  for (int i = 0; i < count; ++i) {
    const int blocks = rows[i] < ABSMAX_BLOCKS ? rows[i] : ABSMAX_BLOCKS;
    unsigned* part = st->amax_dev + 2 + (size_t)i * ABSMAX_BLOCKS;
    hipLaunchKernelGGL(f16x2_absmax_kernel, dim3(blocks), dim3(256), 0, ctx->stream, X[i], rows[i], cols[i], ld[i], vec_ok(X[i], cols[i], ld[i]),
                       part);
    hipLaunchKernelGGL(f16x2_absmax_final_kernel, dim3(1), dim3(256), 0, ctx->stream, part, blocks, st->amax_dev + i);
  }
  PMD_LAUNCH_CHECK(ctx, "f16x2_absmax_kernel");
  PMD_HIP(ctx, hipMemcpyAsync(st->amax_host, st->amax_dev, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < count; ++i) {
    const int be = (int)(st->amax_host[i] >> 23);
    if (be == 0 || be >= 255) return PMD_OK;
    e_out[i] = be - 127 - 13;
  }
  *usable = 1;
  return PMD_OK;
}

// ---------------------------------------------------------------------------------------------
// "Concatenated" form for products whose accumulator is expensive to revisit (M^T G M: the six piece products of three exact
// pieces per operand in ONE matrix product).  For a chunk of kk inner indices the operands become
//     A' (rows x 6 kk) = [ a1 2^-11 | a3 2^-11 | a2 2^-11 | a1 2^-6 | a2 2^-5 | a1 ]
//     B' (6 kk x cols) = [ b3 2^-11 ; b1 2^-11 ; b2 2^-11 ; b2 2^-5 ; b1 2^-6 ; b1 ]
// so that A' B' = 2^-22 (a1 b3 + a3 b1 + a2 b2) + 2^-11 (a1 b2 + a2 b1) + a1 b1, the small terms first in the accumulation
// order.  The scaled copies lose bits only where an entry is below 2^-17 of the largest (2^-22 terms: three bits needed) or
// 2^-22 of it (2^-11 terms).
__global__ __launch_bounds__(256) void f16cat_a_kernel(const float* __restrict__ x, int rows, int kk, long ld, float scale, _Float16* __restrict__ out,
                                                       long ldo) {
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float* row = x + (long)r * ld;
    _Float16* o = out + (long)r * ldo;
    for (int c = threadIdx.x; c < kk; c += 256) {
      _Float16 a, b, d;
      f16x2_pieces(row[c] * scale, a, b, d);
      const float fa = (float)a, fb = (float)b, fd = (float)d;
      o[c] = (_Float16)(fa * (1.f / 2048.f));
      o[kk + c] = (_Float16)(fd * (1.f / 2048.f));
      o[2 * kk + c] = (_Float16)(fb * (1.f / 2048.f));
      o[3 * kk + c] = (_Float16)(fa * (1.f / 64.f));
      o[4 * kk + c] = (_Float16)(fb * (1.f / 32.f));
      o[5 * kk + c] = a;
    }
  }
}

__global__ __launch_bounds__(256) void f16cat_b_kernel(const float* __restrict__ x, int kk, int cols, long ld, float scale, _Float16* __restrict__ out,
                                                       long ldo) {
  for (int r = blockIdx.x; r < kk; r += gridDim.x) {
    const float* row = x + (long)r * ld;
    for (int c = threadIdx.x; c < cols; c += 256) {
      _Float16 a, b, d;
      f16x2_pieces(row[c] * scale, a, b, d);
      const float fa = (float)a, fb = (float)b, fd = (float)d;
      out[(long)r * ldo + c] = (_Float16)(fd * (1.f / 2048.f));
      out[(long)(kk + r) * ldo + c] = (_Float16)(fa * (1.f / 2048.f));
      out[(long)(2 * kk + r) * ldo + c] = (_Float16)(fb * (1.f / 2048.f));
      out[(long)(3 * kk + r) * ldo + c] = (_Float16)(fb * (1.f / 32.f));
      out[(long)(4 * kk + r) * ldo + c] = (_Float16)(fa * (1.f / 64.f));
      out[(long)(5 * kk + r) * ldo + c] = a;
    }
  }
}

int pmd_f16cat_a(pmd_ctx* ctx, const float* X, int rows, int kk, long ld, int e, _Float16* out, long ldo) {
  hipLaunchKernelGGL(f16cat_a_kernel, dim3(rows < 4096 ? rows : 4096), dim3(256), 0, ctx->stream, X, rows, kk, ld, ldexpf(1.f, -e), out, ldo);
  PMD_LAUNCH_CHECK(ctx, "f16cat_a_kernel");
  return PMD_OK;
}

int pmd_f16cat_b(pmd_ctx* ctx, const float* X, int kk, int cols, long ld, int e, _Float16* out, long ldo) {
  hipLaunchKernelGGL(f16cat_b_kernel, dim3(kk < 4096 ? kk : 4096), dim3(256), 0, ctx->stream, X, kk, cols, ld, ldexpf(1.f, -e), out, ldo);
  PMD_LAUNCH_CHECK(ctx, "f16cat_b_kernel");
  return PMD_OK;
}

int pmd_f16x2_split(pmd_ctx* ctx, int count, const float* const* X, const int* rows, const int* cols, const long* ld, void* const* buf,
                    pmd_f16x2_op* ops, int* usable, int pieces) {
  RUN_OK(ensure_state(ctx));
  f16x2_state* st = state_of(ctx);
  *usable = 0;
  if (count < 1 || count > 2) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_f16x2_split", "one or two operands");
  pmd_prof_scope prof__(ctx, "f16x2_split");
  for (int i = 0; i < count; ++i) {
    const int blocks = rows[i] < ABSMAX_BLOCKS ? rows[i] : ABSMAX_BLOCKS;
    unsigned* part = st->amax_dev + 2 + (size_t)i * ABSMAX_BLOCKS;
    hipLaunchKernelGGL(f16x2_absmax_kernel, dim3(blocks), dim3(256), 0, ctx->stream, X[i], rows[i], cols[i], ld[i], vec_ok(X[i], cols[i], ld[i]),
                       part);
    hipLaunchKernelGGL(f16x2_absmax_final_kernel, dim3(1), dim3(256), 0, ctx->stream, part, blocks, st->amax_dev + i);
  }
  PMD_LAUNCH_CHECK(ctx, "f16x2_absmax_kernel");
  PMD_HIP(ctx, hipMemcpyAsync(st->amax_host, st->amax_dev, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < count; ++i) {
    const unsigned bits = st->amax_host[i];
    const int be = (int)(bits >> 23);
    if (be == 0 || be >= 255) return PMD_OK;   // zero / subnormal maximum, Inf or NaN present
    ops[i].e = be - 127 - 13;                  // max |X| 2^-e in [2^13, 2^14): sums of two pieces stay below fp16's 65504
  }
  for (int i = 0; i < count; ++i) {
    const long ldh = pmd_f16x2_ld(cols[i]);
    _Float16* h1 = (_Float16*)buf[i];
    _Float16* h2 = h1 + (size_t)rows[i] * ldh;
    _Float16* h3 = pieces == 3 ? h2 + (size_t)rows[i] * ldh : nullptr;
    ops[i].h1 = h1;
    ops[i].h2 = h2;
    ops[i].h3 = h3;
    ops[i].ld = ldh;
    const int blocks = rows[i] < 8192 ? rows[i] : 8192;
    hipLaunchKernelGGL(f16x2_split_kernel, dim3(blocks), dim3(256), 0, ctx->stream, X[i], rows[i], cols[i], ld[i], vec_ok(X[i], cols[i], ld[i]),
                       ldexpf(1.f, -ops[i].e), h1, h2, h3, ldh);
  }
  PMD_LAUNCH_CHECK(ctx, "f16x2_split_kernel");
  *usable = 1;
  return PMD_OK;
}

static int plan_for(pmd_ctx* ctx, f16x2_state* st, int tA, int tB, int m, int n, int k, long lda, long ldb, long ldc, lt_plan** out) {
  const lt_key key(tA, tB, m, n, k, lda, ldb, ldc);
  auto it = st->plans.find(key);
  if (it != st->plans.end()) { *out = &it->second; return PMD_OK; }
  lt_plan p;
  p.ok = false;
  // column-major view of the row-major product: C^T (n x m) = op(B)^T op(A)^T, so B is hipBLASLt's first operand
  PMD_LT(ctx, hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  const hipblasOperation_t op1 = tB ? HIPBLAS_OP_T : HIPBLAS_OP_N, op2 = tA ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  PMD_LT(ctx, hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op1, sizeof(op1)));
  PMD_LT(ctx, hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op2, sizeof(op2)));
  PMD_LT(ctx, hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16F, tB ? k : n, tB ? n : k, ldb));
  PMD_LT(ctx, hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_16F, tA ? m : k, tA ? k : m, lda));
  PMD_LT(ctx, hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_32F, n, m, ldc));
  hipblasLtMatmulPreference_t pref;
  PMD_LT(ctx, hipblasLtMatmulPreferenceCreate(&pref));
  PMD_LT(ctx, hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &st->ws_bytes, sizeof(st->ws_bytes)));
  hipblasLtMatmulHeuristicResult_t res[1];
  int found = 0;
  const hipblasStatus_t hs = hipblasLtMatmulAlgoGetHeuristic(st->lt, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, res, &found);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (hs == HIPBLAS_STATUS_SUCCESS && found > 0) {
    p.algo = res[0].algo;
    p.ok = true;
  } else {
    hipblasLtMatrixLayoutDestroy(p.la);
    hipblasLtMatrixLayoutDestroy(p.lb);
    hipblasLtMatrixLayoutDestroy(p.lc);
    hipblasLtMatmulDescDestroy(p.desc);
  }
  *out = &st->plans.emplace(key, p).first->second;
  return PMD_OK;
}

// row-major C (m x n) = alpha op(A) op(B) + beta C from split operands (A: m x k or k x m, B: k x n or n x k; the pieces may
// be sub-blocks of a split array: same leading dimension, pointers advanced).  *done = 0: no hipBLASLt kernel for the
// shape, nothing was written.
int pmd_f16x2_matmul(pmd_ctx* ctx, int tA, int tB, int m, int n, int k, float alpha, const pmd_f16x2_op& a, const pmd_f16x2_op& b, float beta,
                     float* C, long ldc, int* done) {
  RUN_OK(ensure_state(ctx));
  f16x2_state* st = state_of(ctx);
  *done = 0;
  lt_plan* p = nullptr;
  RUN_OK(plan_for(ctx, st, tA, tB, m, n, k, a.ld, b.ld, ldc, &p));
  if (!p->ok) return PMD_OK;
  pmd_prof_scope prof__(ctx, "gemm_f16x2");
  const float a_main = alpha * ldexpf(1.f, a.e + b.e), a_small = alpha * ldexpf(1.f, a.e + b.e - 11), one = 1.f;
  const float a_tiny = alpha * ldexpf(1.f, a.e + b.e - 22);
  const bool six = a.h3 && b.h3;
  if (!std::isfinite(a_main) || a_small == 0.f || (six && a_tiny == 0.f)) return PMD_OK;   // scales outside fp32: the fp32 path decides
  if (six) {
    // three exact pieces per operand: A1 B3 + A3 B1 + A2 B2 (2^-22), then the three products below; what is dropped
    // (A2 B3, A3 B2, A3 B3) is 2^-33 of the product - every kept product is exact, only the fp32 accumulation rounds
    PMD_LT(ctx, hipblasLtMatmul(st->lt, p->desc, &a_tiny, b.h3, p->la, a.h1, p->lb, &beta, C, p->lc, C, p->lc, &p->algo, st->ws, st->ws_bytes,
                                ctx->stream));
    PMD_LT(ctx, hipblasLtMatmul(st->lt, p->desc, &a_tiny, b.h1, p->la, a.h3, p->lb, &one, C, p->lc, C, p->lc, &p->algo, st->ws, st->ws_bytes,
                                ctx->stream));
    PMD_LT(ctx, hipblasLtMatmul(st->lt, p->desc, &a_tiny, b.h2, p->la, a.h2, p->lb, &one, C, p->lc, C, p->lc, &p->algo, st->ws, st->ws_bytes,
                                ctx->stream));
  }
  const float* beta1 = six ? &one : &beta;
  PMD_LT(ctx, hipblasLtMatmul(st->lt, p->desc, &a_small, b.h2, p->la, a.h1, p->lb, beta1, C, p->lc, C, p->lc, &p->algo, st->ws, st->ws_bytes,
                              ctx->stream));
  PMD_LT(ctx, hipblasLtMatmul(st->lt, p->desc, &a_small, b.h1, p->la, a.h2, p->lb, &one, C, p->lc, C, p->lc, &p->algo, st->ws, st->ws_bytes,
                              ctx->stream));
  PMD_LT(ctx, hipblasLtMatmul(st->lt, p->desc, &a_main, b.h1, p->la, a.h1, p->lb, &one, C, p->lc, C, p->lc, &p->algo, st->ws, st->ws_bytes,
                              ctx->stream));
  *done = 1;
  return PMD_OK;
}

// row-major C (m x n) = alpha A B + beta C with fp16 operands as they are (A: m x k, B: k x n)
int pmd_f16_plain_matmul(pmd_ctx* ctx, int m, int n, int k, float alpha, const _Float16* A, long lda, const _Float16* B, long ldb, float beta,
                         float* C, long ldc, int* done) {
  RUN_OK(ensure_state(ctx));
  f16x2_state* st = state_of(ctx);
  *done = 0;
  lt_plan* p = nullptr;
  RUN_OK(plan_for(ctx, st, 0, 0, m, n, k, lda, ldb, ldc, &p));
  if (!p->ok) return PMD_OK;
  pmd_prof_scope prof__(ctx, "gemm_f16x2");
  PMD_LT(ctx, hipblasLtMatmul(st->lt, p->desc, &alpha, B, p->la, A, p->lb, &beta, C, p->lc, C, p->lc, &p->algo, st->ws, st->ws_bytes, ctx->stream));
  *done = 1;
  return PMD_OK;
}

// library-owned scratch for the pieces (grown on demand; shared with the split-K partial sums of pmd_gemm_rm).
// *out = NULL when the device has no room for it (the caller then takes its fp32 path); no error is left behind.
int pmd_split_scratch(pmd_ctx* ctx, size_t need, void** out) {
  *out = nullptr;
  if (ctx->split_ws_bytes < need) {
    PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->split_ws) (void)hipFree(ctx->split_ws);
    ctx->split_ws = nullptr;
    ctx->split_ws_bytes = 0;
    if (hipMalloc(&ctx->split_ws, need) != hipSuccess) {
      (void)hipGetLastError();
      ctx->split_ws = nullptr;
      return PMD_OK;
    }
    ctx->split_ws_bytes = need;
  }
  *out = ctx->split_ws;
  return PMD_OK;
}

// frees the scratch when it is larger than `keep_bytes` (end of a decomposition: the next one may need the room for its movie)
int pmd_split_scratch_trim(pmd_ctx* ctx, size_t keep_bytes) {
  if (ctx->split_ws && ctx->split_ws_bytes > keep_bytes) {
    PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->split_ws);
    ctx->split_ws = nullptr;
    ctx->split_ws_bytes = 0;
  }
  return PMD_OK;
}

// The whole product for pmd_gemm_rm.  *done = 0: the caller's fp32 path must run (C untouched).
int pmd_gemm_f16x2(pmd_ctx* ctx, int transA, int transB, int m, int n, int k, float alpha, const float* A, long lda, const float* B, long ldb,
                   float beta, float* C, long ldc, int* done) {
  *done = 0;
  const int a_rows = transA ? k : m, a_cols = transA ? m : k, b_rows = transB ? n : k, b_cols = transB ? k : n;
  const size_t na = pmd_f16x2_bytes(a_rows, a_cols, 2), nb = pmd_f16x2_bytes(b_rows, b_cols, 2);
  void* w = nullptr;
  RUN_OK(pmd_split_scratch(ctx, na + nb, &w));
  if (!w) return PMD_OK;
  const float* X[2] = {A, B};
  const int rows[2] = {a_rows, b_rows}, cols[2] = {a_cols, b_cols};
  const long ld[2] = {lda, ldb};
  void* buf[2] = {w, (char*)w + na};
  pmd_f16x2_op ops[2];
  int usable = 0;
  RUN_OK(pmd_f16x2_split(ctx, 2, X, rows, cols, ld, buf, ops, &usable, 2));
  if (!usable) return PMD_OK;
  return pmd_f16x2_matmul(ctx, transA, transB, m, n, k, alpha, ops[0], ops[1], beta, C, ldc, done);
}
