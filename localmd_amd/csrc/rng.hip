// Counter-based Gaussian generator: Philox4x32-10 + Box-Muller on 24-bit uniforms.
// Replaces jax.random.normal at /root/reference/localmd/decomposition.py:62, :127, :870 and
// pmd_loader.py:56.  Element e of logical array (stream, index) comes from Philox block
// q = e/4, lane e%4, counter (q_lo, q_hi, index, stream), key (seed_lo, seed_hi); the result
// is independent of launch geometry and GPU count.  oracle/philox.py restates it in NumPy.
#include "pmd_common.h"

__device__ __forceinline__ void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__device__ __forceinline__ void pmd_normal4(uint64_t seed, uint32_t stream, uint32_t index, uint64_t q, float z[4]) {
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), index, stream};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float two_m24 = 5.9604644775390625e-08f;
  const float u0 = ((float)(c[0] >> 8) + 0.5f) * two_m24;
  const float u1 = ((float)(c[1] >> 8) + 0.5f) * two_m24;
  const float u2 = ((float)(c[2] >> 8) + 0.5f) * two_m24;
  const float u3 = ((float)(c[3] >> 8) + 0.5f) * two_m24;
  const float r0 = sqrtf(-2.0f * logf(u0));
  const float r1 = sqrtf(-2.0f * logf(u2));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u1, &s0, &c0);
  sincosf(6.283185307179586f * u3, &s1, &c1);
  z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
}

// Batched fill.  Logical array (stream, index0 + b*index_step) has rows x cols elements,
// e = row*cols + col.  Written to out[b*batch_stride + row*ld + col] (transpose == 0) or
// out[b*batch_stride + col*ld + row] (transpose == 1).
__global__ void rng_normal_kernel(uint64_t seed, uint32_t stream, uint32_t index0, uint32_t index_step, long rows,
                                  int cols, int transpose, float* __restrict__ out, long ld, long batch_stride) {
  const long n = rows * cols;
  const long nq = (n + 3) / 4;
  float* o = out + (long)blockIdx.y * batch_stride;
  const uint32_t index = index0 + blockIdx.y * index_step;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
    float z[4];
    pmd_normal4(seed, stream, index, (uint64_t)q, z);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long e = 4 * q + i;
      if (e < n) {
        const long row = e / cols;
        const int col = (int)(e - row * cols);
        if (transpose) o[(long)col * ld + row] = z[i];
        else o[row * ld + col] = z[i];
      }
    }
  }
}

int pmd_launch_rng(pmd_ctx* ctx, uint64_t seed, uint32_t stream, uint32_t index0, uint32_t index_step, int batch,
                   long rows, int cols, int transpose, float* out, long ld, long batch_stride) {
  pmd_prof_scope prof__(ctx, "rng_normal");
  if (batch <= 0 || rows <= 0 || cols <= 0) return PMD_OK;
  const long nq = (rows * cols + 3) / 4;
  int bx = (int)((nq + 255) / 256);
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(rng_normal_kernel, dim3(bx, batch), dim3(256), 0, ctx->stream, seed, stream, index0,
                     index_step, rows, cols, transpose, out, ld, batch_stride);
  PMD_LAUNCH_CHECK(ctx, "rng_normal_kernel");
  return PMD_OK;
}
