// Streaming contractions over gathered tiles of the pixel-major movie X[c][t].
//
//   tile_atx :  Out[tile][comp][t] = sum_q A[tile][comp][q] * X[pix[tile][q]][t]      (r x d)(d x T)
//               V_ds = U_ds^T X_ds (decomposition.py:295-298), U0^T X (:318), U_b^T X (:390-407,
//               pmd_loader.py:411), Q^T A (decomposition.py:65)
//   tile_xbt :  S[tile][slice][comp][q] = sum_{t in slice} X[pix[tile][q]][t] * B[tile][comp][t]   (d x T)(T x r)
//               X V_b^T (decomposition.py:304-306), A Omega (:63)
//   tile_gram:  G[tile][slice][i][j] = sum_{x in slice} In[tile][i][x] * In[tile][j][x]  (fp64 accumulate)
//               Gram matrices behind the four small SVDs of single_block_md (:66, :301, :315-319)
//   tile_rowmix: Out[tile][c][x] = sum_c' N[tile][c'][c] * In[tile][c'][x]               (fp64 accumulate)
//
// Every per-tile [comp][x] array has PMD_RPAD = 64 rows (rows >= rank are zero) so that four
// waves each own one 16-row MFMA tile.  All fp32 products run on v_mfma_f32_16x16x4_f32
// (exact fp32 fmaf chains, 64 FLOP/clk/SIMD).
#include "pmd_common.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// tile_atx.  One workgroup = one tile x one run of 32-frame chunks.  Wave (mt, ks) keeps its
// 16 x (16*KJW) slice of A in registers for the whole run (KJW float4 = 4*KJW VGPRs) and reads
// the X chunk, shared by all waves, from LDS.  K is walked in a permuted order: k-step (J, s),
// lane slot kk <-> row q = 16*J + 4*kk + s, so one float4 of A feeds four MFMAs and the LDS
// rows of one 32-lane group sit 4 rows = 16 banks apart (row stride TC+4 floats): conflict-free.
// ------------------------------------------------------------------------------------------
// MT = row tiles of A that carry data (4: all 64 rows, one row tile per wave and K slice; 1: rows < 16 only - the background
// projection with its default 15 columns - where the waves are K slices of the ONE row tile: a quarter of the MFMA work,
// rows >= 16 of Out are not written).
template <int KJW, int KS, int NS, int MT = 4>
__global__ __launch_bounds__(64 * MT * KS * NS) void tile_atx_kernel(const float* __restrict__ X, long ldx,
                                                            const int* __restrict__ pix, int pix_stride,
                                                            long row0_stride, int d,
                                                            const float* __restrict__ A, long a_tile_stride, int a_ld,
                                                            float* __restrict__ Out, long out_tile_stride, long ldo,
                                                            int n_chunks_total, int chunks_per_slice) {
  constexpr int NTW = 2;
  constexpr int TC = 16 * NTW;
  constexpr int LSTR = TC + 4;
  constexpr int DPAD = 16 * KJW * KS;
  // NS = 2: a second group of four waves takes the second 16-frame N tile, so every SIMD hosts two
  // waves whose LDS refills, barriers and stores overlap with the other one's MFMAs.
  static_assert(NS == 1 || KS == 1, "the N split is only built for KS = 1");
  static_assert(MT == 4 || (MT == 1 && NS == 1), "one row tile: K slices only");
  constexpr int NTHREADS = 64 * MT * KS * NS;
  constexpr int RED = (KS - 1) * MT * 512;   // floats of K-split scratch
  constexpr int QUADS = TC / 4;
  constexpr int NPRE = (DPAD * QUADS + NTHREADS - 1) / NTHREADS;
  // two LDS buffers (one barrier per chunk) whenever they fit next to the K-split scratch
  constexpr bool DB = (2 * DPAD * LSTR + RED + 4) * 4 <= 160 * 1024;
  constexpr int NBUF = DB ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* red = lds + NBUF * DPAD * LSTR;
  constexpr int TRASH = NBUF * DPAD * LSTR + RED;  // 16-byte slot nobody reads

  const int tile = pmd_xcd_tile();
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and known to be)
  const int mt = (MT == 4) ? (wid & 3) : 0, ks = (NS == 1) ? ((MT == 4) ? (wid >> 2) : wid) : 0, nh = (NS == 1) ? 0 : (wid >> 2);
  const int n16 = lane & 15, kk = lane >> 4;
  const int kz = blockIdx.z;           // grid-level K split (d > DPAD)
  const int qbase = kz * DPAD;
  const int dloc = min(d - qbase, DPAD);

  const int c_begin = blockIdx.y * chunks_per_slice;
  const int c_end = min(n_chunks_total, c_begin + chunks_per_slice);
  if (c_begin >= c_end) return;

  // A fragments
  f32x4 areg[KJW];
  {
    const float* ap = A + (long)tile * a_tile_stride + (long)(16 * mt + n16) * a_ld + qbase + ks * (16 * KJW) + 4 * kk;
#pragma unroll
    for (int J = 0; J < KJW; ++J) areg[J] = *reinterpret_cast<const f32x4*>(ap + 16 * J);
  }

  // loader items of this thread
  long goff[NPRE];
  int loff[NPRE];
#pragma unroll
  for (int k = 0; k < NPRE; ++k) {
    const int i = tid + k * NTHREADS;
    const int q = i / QUADS, j = i - q * QUADS;
    // branch-free: an item beyond the tile reads the tile's first row and lands in a trash slot
    const int qc = (q < dloc) ? q : 0;
    const long row = pix ? (long)pix[(long)tile * pix_stride + qbase + qc] : (long)tile * row0_stride + qbase + qc;
    goff[k] = row * ldx + 4 * j;
    loff[k] = (q < dloc) ? q * LSTR + 4 * j : TRASH;
  }
  for (int i = tid; i < NBUF * DPAD * LSTR; i += NTHREADS) lds[i] = 0.f;
  __syncthreads();

  f32x4 pre[NPRE];
#pragma unroll
  for (int k = 0; k < NPRE; ++k) pre[k] = *reinterpret_cast<const f32x4*>(X + goff[k] + (long)c_begin * TC);
#pragma unroll
  for (int k = 0; k < NPRE; ++k) *reinterpret_cast<f32x4*>(lds + (loff[k] == TRASH ? TRASH : loff[k])) = pre[k];
  __syncthreads();

  float* outp = Out + (long)tile * out_tile_stride + (long)(16 * mt + 4 * kk) * ldo + n16 + 16 * nh;
  const int rd_off = (ks * 16 * KJW + 4 * kk) * LSTR + n16 + 16 * nh;

  for (int c = c_begin; c < c_end; ++c) {
    const int cur = DB ? ((c - c_begin) & 1) : 0;
    const float* xrd = lds + cur * (DPAD * LSTR) + rd_off;
    const bool more = c + 1 < c_end;
    if (more) {
#pragma unroll
      for (int k = 0; k < NPRE; ++k) pre[k] = *reinterpret_cast<const f32x4*>(X + goff[k] + (long)(c + 1) * TC);
    }
    // LDS operands one J ahead of the MFMAs that consume them
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NS == 1) {
      float b0[4], b1[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { b0[s] = xrd[s * LSTR]; b1[s] = xrd[s * LSTR + 16]; }
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // the prologue reads form their own group
#pragma unroll
      for (int J = 0; J < KJW; ++J) {
        float n0[4], n1[4];
        if (J + 1 < KJW) {
#pragma unroll
          for (int s = 0; s < 4; ++s) { n0[s] = xrd[(16 * (J + 1) + s) * LSTR]; n1[s] = xrd[(16 * (J + 1) + s) * LSTR + 16]; }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][s], b0[s], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][s], b1[s], acc1, 0, 0, 0);
        }
        // pin the order: the four LDS reads of step J+1, then the eight MFMAs of step J
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        if (J + 1 < KJW) {
#pragma unroll
          for (int s = 0; s < 4; ++s) { b0[s] = n0[s]; b1[s] = n1[s]; }
        }
      }
    } else {
      // one N tile per wave; two accumulator chains (s even / odd) cover the 40-cycle MFMA latency
      float b0[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) b0[s] = xrd[s * LSTR];
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int J = 0; J < KJW; ++J) {
        float n0[4];
        if (J + 1 < KJW) {
#pragma unroll
          for (int s = 0; s < 4; ++s) n0[s] = xrd[(16 * (J + 1) + s) * LSTR];
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][0], b0[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][1], b0[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][2], b0[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][3], b0[3], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if (J + 1 < KJW) {
#pragma unroll
          for (int s = 0; s < 4; ++s) b0[s] = n0[s];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc0[i] += acc1[i];
    }
    if (KS > 1) {
      // sum the K slices of the waves that share an M tile
      if (ks > 0) {
        float* r = red + (((ks - 1) * MT + mt) * 2) * 256 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) { r[i * 64] = acc0[i]; r[256 + i * 64] = acc1[i]; }
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int o = 1; o < KS; ++o) {
          const float* r = red + (((o - 1) * MT + mt) * 2) * 256 + lane;
#pragma unroll
          for (int i = 0; i < 4; ++i) { acc0[i] += r[i * 64]; acc1[i] += r[256 + i * 64]; }
        }
      }
    }
    if (ks == 0) {
      float* o = outp + (long)c * TC;
      if (gridDim.z == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[(long)i * ldo] = acc0[i];
          if (NS == 1) o[(long)i * ldo + 16] = acc1[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          atomicAdd(&o[(long)i * ldo], acc0[i]);
          if (NS == 1) atomicAdd(&o[(long)i * ldo + 16], acc1[i]);
        }
      }
    }
    if (!DB) __syncthreads();  // single buffer: everyone must be done reading before the refill
    if (more) {
      float* xw = lds + (DB ? (cur ^ 1) * (DPAD * LSTR) : 0);
#pragma unroll
      for (int k = 0; k < NPRE; ++k) *reinterpret_cast<f32x4*>(loff[k] == TRASH ? lds + TRASH : xw + loff[k]) = pre[k];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// tile_atx with LDS-DMA staging (KS = NS = 1, DPAD = 16 KJW <= 400): the X chunk (DPAD x 32 floats) goes from
// global memory straight into LDS (global_load_lds_dwordx4: one wave instruction fills a contiguous 1 KiB =
// 8 rows x 8 quads), no staging registers and no ds_write phase.  Three dense buffers form a ring, loads run two
// chunks ahead, counted vmcnt + one raw barrier per chunk.  Rows are 32 floats with no padding; the quad index
// is XORed with 4 * ((row >> 2) & 1) - applied on the global-address side by the loader and on the read side
// by the MFMA operand fetch - so the four K slots of a lane group never share a bank.
// Rows q >= dloc load the tile's first row (finite values meeting zero columns of A).
// (Round 3: a three-row-tile form for the default 50 components - rows 0-47 as 3 row tiles x 2 frame tiles x 2 pixel halves on
// the matrix cores, rows 48-49 on the vector units, partial sums exchanged through LDS: 22 % fewer MFMAs, a quarter of the LDS
// operand reads - was built, verified against fp64 and measured at 13.5 ms against 13.4 ms for this kernel (14.6 ms without its
// vector-unit part, 13.9 ms without its stores): neither the MFMA count nor the stores bound this kernel; not kept.)
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* pmd_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* pmd_gbl_ptr_t;

template <int KJW>
__global__ __launch_bounds__(256) void tile_atx_dma_kernel(const float* __restrict__ X, long ldx,
                                                           const int* __restrict__ pix, int pix_stride,
                                                           long row0_stride, int d, const float* __restrict__ A,
                                                           long a_tile_stride, int a_ld, float* __restrict__ Out,
                                                           long out_tile_stride, long ldo, int n_chunks_total,
                                                           int chunks_per_slice, const int* __restrict__ tile_ranks) {
  constexpr int DPAD = 16 * KJW;
  constexpr int BUF = DPAD * 32;           // floats per ring buffer
  constexpr int NINS = DPAD / 8;           // 1-KiB pieces per chunk
  constexpr int NSLOT = (NINS + 3) / 4;    // pieces per wave
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tile = pmd_xcd_tile();
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and known to be)
  const int n16 = lane & 15, kk = lane >> 4;
  const int dloc = min(d, DPAD);
  const int c_begin = blockIdx.y * chunks_per_slice;
  const int c_end = min(n_chunks_total, c_begin + chunks_per_slice);
  if (c_begin >= c_end) return;

  // tile_ranks (projection launches): a tile that kept <= 32 components has only two non-zero 16-row tiles of A.  Its four
  // waves then take (row tile, frame tile) = (wid & 1, wid >> 1) - one frame tile each instead of two, half the MFMAs and
  // LDS operand reads per wave - and rows >= 32 of the output are not written (the callers compact rows < rank only).
  const bool half = tile_ranks != nullptr && __builtin_amdgcn_readfirstlane(tile_ranks[tile]) <= 32;
  const int mrow = half ? (wid & 1) : wid;
  const int nsel = wid >> 1;
  f32x4 areg[KJW];
  {
    const float* ap = A + (long)tile * a_tile_stride + (long)(16 * mrow + n16) * a_ld + 4 * kk;
#pragma unroll
    for (int J = 0; J < KJW; ++J) areg[J] = *reinterpret_cast<const f32x4*>(ap + 16 * J);
  }
  // loader pieces: piece m covers rows 8m .. 8m+7; wave w issues m = w, w+4, ...
  const float* gsrc[NSLOT];
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const int m = min(wid + 4 * k, NINS - 1);
    const int q = 8 * m + (lane >> 3), p = lane & 7;
    const int f = 4 * ((q >> 2) & 1);
    const int qc = (q < dloc) ? q : 0;
    const long row = pix ? (long)pix[(long)tile * pix_stride + qc] : (long)tile * row0_stride + qc;
    gsrc[k] = X + row * ldx + 4 * (p ^ f);
  }
  auto issue = [&](int c, int buf) {
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      if (wid + 4 * k < NINS)
        __builtin_amdgcn_global_load_lds((pmd_gbl_ptr_t)(gsrc[k] + (long)c * 32),
                                         (pmd_lds_ptr_t)(lds + buf * BUF + 8 * (wid + 4 * k) * 32), 16, 0, 0);
    }
  };
  // loads past the slice end read the next chunks of the same rows (rows carry PMD_LD_SLACK = 64 spare floats)
  issue(c_begin, 0);
  issue(c_begin + 1, 1);
  constexpr int MYINS = NSLOT;  // upper bound of this wave's pieces per chunk (the last slot may be void)
  // wait for chunk c_begin: at most the pieces of chunk c_begin + 1 may still be in flight
  if (wid + 4 * (NSLOT - 1) < NINS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MYINS) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MYINS - 1) : "memory");
  __builtin_amdgcn_s_barrier();
  const int f = 4 * (kk & 1);
  const int off0 = (4 * kk) * 32 + (((n16 >> 2) ^ f) << 2) + (n16 & 3);
  const int off1 = (4 * kk) * 32 + (((4 + (n16 >> 2)) ^ f) << 2) + (n16 & 3);
  float* outp = Out + (long)tile * out_tile_stride + (long)(16 * mrow + 4 * kk) * ldo + n16;
  const bool full = wid + 4 * (NSLOT - 1) < NINS;  // this wave issues NSLOT (else NSLOT - 1) pieces per chunk
  if (half) {
    const int offn = nsel ? off1 : off0;
    outp += 16 * nsel;
    for (int c = c_begin; c < c_end; ++c) {
      const int cur = (c - c_begin) % 3;
      issue(c + 2, (c - c_begin + 2) % 3);
      const float* xn = lds + cur * BUF + offn;
      // one frame tile per wave; two accumulator chains (s even / odd) cover the MFMA latency
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      float b0[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) b0[s] = xn[s * 32];
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int J = 0; J < KJW; ++J) {
        float n0[4];
        if (J + 1 < KJW) {
#pragma unroll
          for (int s = 0; s < 4; ++s) n0[s] = xn[(16 * (J + 1) + s) * 32];
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][0], b0[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][1], b0[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][2], b0[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][3], b0[3], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if (J + 1 < KJW) {
#pragma unroll
          for (int s = 0; s < 4; ++s) b0[s] = n0[s];
        }
      }
      {
        float* o = outp + (long)c * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[(long)i * ldo] = acc0[i] + acc1[i];
      }
      // chunk c + 1 must have landed: still allowed in flight are the pieces of chunk c + 2 and the 4 stores above
      if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MYINS + 4) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MYINS - 1 + 4) : "memory");
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  for (int c = c_begin; c < c_end; ++c) {
    const int cur = (c - c_begin) % 3;
    issue(c + 2, (c - c_begin + 2) % 3);
    const float* x0 = lds + cur * BUF + off0;
    const float* x1 = lds + cur * BUF + off1;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float b0[4], b1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { b0[s] = x0[s * 32]; b1[s] = x1[s * 32]; }
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int J = 0; J < KJW; ++J) {
      float n0[4], n1[4];
      if (J + 1 < KJW) {
#pragma unroll
        for (int s = 0; s < 4; ++s) { n0[s] = x0[(16 * (J + 1) + s) * 32]; n1[s] = x1[(16 * (J + 1) + s) * 32]; }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][s], b0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][s], b1[s], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      if (J + 1 < KJW) {
#pragma unroll
        for (int s = 0; s < 4; ++s) { b0[s] = n0[s]; b1[s] = n1[s]; }
      }
    }
    {
      float* o = outp + (long)c * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) { o[(long)i * ldo] = acc0[i]; o[(long)i * ldo + 16] = acc1[i]; }
    }
    // chunk c + 1 must have landed: still allowed in flight are the pieces of chunk c + 2 and the 8 stores above
    if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MYINS + 8) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MYINS - 1 + 8) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA write may outlive the workgroup's LDS allocation
}

template <int KJW>
static int launch_atx_dma(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                          const float* A, long a_tile_stride, int a_ld, float* Out, long out_tile_stride, long ldo,
                          int n_tiles, int T, int slices, const int* tile_ranks) {
  const size_t lds = (size_t)3 * 16 * KJW * 32 * sizeof(float);
  auto kern = tile_atx_dma_kernel<KJW>;
  PMD_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int n_chunks = (T + 31) / 32;
  if (slices < 1) slices = 1;
  if (slices > n_chunks) slices = n_chunks;
  const int cps = (n_chunks + slices - 1) / slices;
  const int ny = (n_chunks + cps - 1) / cps;
  hipLaunchKernelGGL(kern, dim3(n_tiles, ny, 1), dim3(256), lds, ctx->stream, X, ldx, pix, pix_stride, row0_stride, d, A,
                     a_tile_stride, a_ld, Out, out_tile_stride, ldo, n_chunks, cps, tile_ranks);
  PMD_LAUNCH_CHECK(ctx, "tile_atx_dma_kernel");
  return PMD_OK;
}

template <int KJW, int KS, int NS, int MT = 4>
static int launch_atx_variant(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride,
                              int d, const float* A, long a_tile_stride, int a_ld, float* Out, long out_tile_stride,
                              long ldo, int n_tiles, int T, int slices, int kz) {
  constexpr int DPAD = 16 * KJW * KS;
  constexpr int RED = (KS - 1) * MT * 512;
  constexpr bool DB = (2 * DPAD * 36 + RED + 4) * 4 <= 160 * 1024;
  const size_t lds = (size_t)((DB ? 2 : 1) * DPAD * 36 + RED + 4) * sizeof(float);
  static_assert(((DB ? 2 : 1) * DPAD * 36 + RED + 4) * 4 <= 160 * 1024, "LDS budget");
  auto kern = tile_atx_kernel<KJW, KS, NS, MT>;
  PMD_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int n_chunks = (T + 31) / 32;
  if (slices < 1) slices = 1;
  if (slices > n_chunks) slices = n_chunks;
  const int cps = (n_chunks + slices - 1) / slices;
  const int ny = (n_chunks + cps - 1) / cps;
  // tiles ride on gridDim.x (limit 2^31-1)
  hipLaunchKernelGGL(kern, dim3(n_tiles, ny, kz), dim3(64 * MT * KS * NS), lds, ctx->stream, X, ldx, pix, pix_stride,
                     row0_stride, d, A, a_tile_stride, a_ld, Out, out_tile_stride, ldo, n_chunks, cps);
  PMD_LAUNCH_CHECK(ctx, "tile_atx_kernel");
  return PMD_OK;
}

// a_ld must be the padded pixel count returned by pmd_tile_dpad(d); Out rows are PMD_RPAD.
int pmd_launch_tile_atx(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                        const float* A, long a_tile_stride, int a_ld, float* Out, long out_tile_stride, long ldo,
                        int n_tiles, int T, int slices) {
  pmd_prof_scope prof__(ctx, ctx->atx_label ? ctx->atx_label : "tile_atx");
  if (n_tiles <= 0 || T <= 0) return PMD_OK;
  int kz = 1;
  int dv = d;
  if (d > 1024) {
    kz = (d + 1023) / 1024;    // grid-level K split: slice z covers pixels [1024 z, 1024 (z + 1)), partial sums by atomicAdd
    dv = 1024;
    // the two K halves accumulate with atomics: clear the 64 output rows of every tile (only those: the tile stride may
    // cover more row blocks than this call writes, pmd_launch_tile_atx_rp)
    PMD_HIP(ctx, hipMemset2DAsync(Out, (size_t)out_tile_stride * sizeof(float), 0, (size_t)PMD_RPAD * ldo * sizeof(float), (size_t)n_tiles,
                                  ctx->stream));
  }
  pmd_dvariant v;
  if (!pmd_pick_dvariant(dv, &v)) return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "tile_atx", "no kernel variant");
  if (a_ld < kz * v.dpad) return pmd_fail(ctx, PMD_ERR_ARG, "tile_atx", "a_ld smaller than padded tile size");
#define ATX_CASE(KJW_, KS_, NS_)                                                                                         \
  if (v.kjw == KJW_ && v.ks == KS_)                                                                                      \
    return launch_atx_variant<KJW_, KS_, NS_>(ctx, X, ldx, pix, pix_stride, row0_stride, d, A, a_tile_stride, a_ld, Out, \
                                              out_tile_stride, ldo, n_tiles, T, slices, kz);
  // d in (256, 400]: LDS-DMA staged variant (needs the two spare chunks behind every row: ldx >= 32 (chunks + 2))
  {
    const char* dm = getenv("PMD_ATX_DMA");
    const bool dma_ok = !(dm && !strcmp(dm, "0")) && ldx >= 32L * ((T + 31) / 32 + 2) && (ldx % 4) == 0;
    if (v.kjw == 25 && v.ks == 1 && dma_ok)
      return launch_atx_dma<25>(ctx, X, ldx, pix, pix_stride, row0_stride, d, A, a_tile_stride, a_ld, Out, out_tile_stride,
                                ldo, n_tiles, T, slices, ctx->atx_ranks);
  }
  // rows < 16 only (ctx->atx_rows, set by the background projection) on a 1024-pixel tile: eight K slices of one row tile
  if (ctx->atx_rows > 0 && ctx->atx_rows <= 16 && v.kjw == 32 && v.ks == 2 && kz == 1)
    return launch_atx_variant<8, 8, 1, 1>(ctx, X, ldx, pix, pix_stride, row0_stride, d, A, a_tile_stride, a_ld, Out, out_tile_stride, ldo,
                                          n_tiles, T, slices, kz);
  ATX_CASE(16, 1, 2)
  ATX_CASE(25, 1, 1)
  ATX_CASE(32, 1, 1)
  ATX_CASE(25, 2, 1)
  ATX_CASE(32, 2, 1)
#undef ATX_CASE
  return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "tile_atx", "variant not built");
}

// ------------------------------------------------------------------------------------------
// tile_xbt.  No LDS: wave w of workgroup (tile, slice, mblock) owns MPW 16-pixel M tiles and all
// four 16-component N tiles; per 16-frame group every lane loads one float4 of X per M tile
// (16 rows x 64 B per wave instruction) and one float4 of B per N tile, then issues 4*MPW*4
// MFMAs.  K order inside a group is permuted like tile_atx (slot kk <-> frame 4*kk + s).
// Loads run one group ahead of the MFMAs (two register sets).
// ------------------------------------------------------------------------------------------
template <int MPW>
__global__ __launch_bounds__(256) void tile_xbt_kernel(const float* __restrict__ X, long ldx,
                                                       const int* __restrict__ pix, int pix_stride, long row0_stride,
                                                       int d, const float* __restrict__ B, long b_tile_stride, long ldb,
                                                       float* __restrict__ S, long s_tile_stride, long s_slice_stride,
                                                       int s_ld, int n_groups_total, int groups_per_slice) {
  const int tile = pmd_xcd_tile();
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform, and known to be)
  const int n16 = lane & 15, kk = lane >> 4;
  const int m0 = (blockIdx.z * 4 + wid) * MPW;  // first M tile of this wave
  const int g_begin = blockIdx.y * groups_per_slice;
  const int g_end = min(n_groups_total, g_begin + groups_per_slice);

  const float* xrow[MPW];
  bool mvalid[MPW];
  float rmask[MPW];
#pragma unroll
  for (int i = 0; i < MPW; ++i) {
    const int q = 16 * (m0 + i) + n16;
    mvalid[i] = 16 * (m0 + i) < d;  // wave-uniform
    const int qc = min(q, d - 1);
    const long row = pix ? (long)pix[(long)tile * pix_stride + qc] : (long)tile * row0_stride + qc;
    xrow[i] = X + row * ldx + 4 * kk;
    rmask[i] = (q < d) ? 1.f : 0.f;
  }
  const float* brow = B + (long)tile * b_tile_stride + (long)n16 * ldb + 4 * kk;

  f32x4 acc[MPW][4];
#pragma unroll
  for (int i = 0; i < MPW; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (g_begin < g_end && mvalid[0]) {
    f32x4 a0[MPW], b0[4], a1[MPW], b1[4];
    auto load = [&](f32x4* a, f32x4* b, int g) {
      const long t = (long)g * 16;
#pragma unroll
      for (int i = 0; i < MPW; ++i)
        if (mvalid[i]) a[i] = *reinterpret_cast<const f32x4*>(xrow[i] + t);
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const f32x4*>(brow + (long)(16 * n) * ldb + t);
    };
    auto compute = [&](const f32x4* a, const f32x4* b) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MPW; ++i)
          if (mvalid[i]) {
            const float av = a[i][s] * rmask[i];
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[n][s], acc[i][n], 0, 0, 0);
          }
    };
    load(a0, b0, g_begin);
    int g = g_begin;
    for (; g + 1 < g_end; g += 2) {
      load(a1, b1, g + 1);
      compute(a0, b0);
      if (g + 2 < g_end) load(a0, b0, g + 2);
      compute(a1, b1);
    }
    if (g < g_end) compute(a0, b0);
  }

  float* sp = S + (long)tile * s_tile_stride + (long)blockIdx.y * s_slice_stride;
#pragma unroll
  for (int i = 0; i < MPW; ++i) {
    if (!mvalid[i]) continue;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      // C layout: col = lane&15 -> component, row = 4*(lane>>4) + reg -> pixel
      float* o = sp + (long)(16 * n + n16) * s_ld + 16 * (m0 + i) + 4 * kk;
      *reinterpret_cast<f32x4*>(o) = acc[i][n];
    }
  }
}

// The same with B staged through LDS (round 3).  In the form above every wave loads all four 16-component tiles of B for
// every 16-frame group - the four waves of a workgroup fetch the same 4 KB four times, 27 GB of the 68 GB the kernel pulls
// through L2 per launch at config 3.  Here wave w fetches component tile w only and publishes it in LDS (two 4 KB buffers,
// one barrier per group: 112 MFMAs = 3 600 cycles apart); every wave reads its four operand fragments from there.
template <int MPW>
__global__ __launch_bounds__(256) void tile_xbt_lds_kernel(const float* __restrict__ X, long ldx,
                                                           const int* __restrict__ pix, int pix_stride, long row0_stride,
                                                           int d, const float* __restrict__ B, long b_tile_stride, long ldb,
                                                           float* __restrict__ S, long s_tile_stride, long s_slice_stride,
                                                           int s_ld, int n_groups_total, int groups_per_slice) {
  __shared__ f32x4 bs[2][4][64];
  const int tile = pmd_xcd_tile();
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n16 = lane & 15, kk = lane >> 4;
  const int m0 = (blockIdx.z * 4 + wid) * MPW;  // first M tile of this wave
  const int g_begin = blockIdx.y * groups_per_slice;
  const int g_end = min(n_groups_total, g_begin + groups_per_slice);

  const float* xrow[MPW];
  bool mvalid[MPW];
  float rmask[MPW];
#pragma unroll
  for (int i = 0; i < MPW; ++i) {
    const int q = 16 * (m0 + i) + n16;
    mvalid[i] = 16 * (m0 + i) < d;  // wave-uniform
    const int qc = min(q, d - 1);
    const long row = pix ? (long)pix[(long)tile * pix_stride + qc] : (long)tile * row0_stride + qc;
    xrow[i] = X + row * ldx + 4 * kk;
    rmask[i] = (q < d) ? 1.f : 0.f;
  }
  // this wave's share of B: component tile `wid`
  const float* bmine = B + (long)tile * b_tile_stride + (long)(16 * wid + n16) * ldb + 4 * kk;

  f32x4 acc[MPW][4];
#pragma unroll
  for (int i = 0; i < MPW; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // (every wave runs the loop - also one whose pixel tiles all lie beyond d: it carries a quarter of B and the barriers)
  if (g_begin < g_end) {
    f32x4 a0[MPW], a1[MPW], bnext;
    auto load_x = [&](f32x4* a, int g) {
      const long t = (long)g * 16;
#pragma unroll
      for (int i = 0; i < MPW; ++i)
        if (mvalid[i]) a[i] = *reinterpret_cast<const f32x4*>(xrow[i] + t);
    };
    auto compute = [&](const f32x4* a, int buf) {
      f32x4 b[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = bs[buf][n][lane];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MPW; ++i)
          if (mvalid[i]) {
            const float av = a[i][s] * rmask[i];
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[n][s], acc[i][n], 0, 0, 0);
          }
    };
    bs[0][wid][lane] = *reinterpret_cast<const f32x4*>(bmine + (long)g_begin * 16);
    load_x(a0, g_begin);
    __syncthreads();
    int g = g_begin;
    for (; g + 1 < g_end; g += 2) {
      // group g from (a0, buffer 0); group g + 1 from (a1, buffer 1)
      bnext = *reinterpret_cast<const f32x4*>(bmine + (long)(g + 1) * 16);
      load_x(a1, g + 1);
      compute(a0, 0);
      bs[1][wid][lane] = bnext;
      __syncthreads();
      const bool more = g + 2 < g_end;   // uniform over the workgroup
      if (more) {
        bnext = *reinterpret_cast<const f32x4*>(bmine + (long)(g + 2) * 16);
        load_x(a0, g + 2);
      }
      compute(a1, 1);
      if (more) bs[0][wid][lane] = bnext;
      __syncthreads();
    }
    if (g < g_end) compute(a0, 0);
  }

  float* sp = S + (long)tile * s_tile_stride + (long)blockIdx.y * s_slice_stride;
#pragma unroll
  for (int i = 0; i < MPW; ++i) {
    if (!mvalid[i]) continue;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float* o = sp + (long)(16 * n + n16) * s_ld + 16 * (m0 + i) + 4 * kk;
      *reinterpret_cast<f32x4*>(o) = acc[i][n];
    }
  }
}

// S rows have s_ld floats (>= 16*ceil(d/16)); slices partition the frame range.
int pmd_launch_tile_xbt(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                        const float* B, long b_tile_stride, long ldb, float* S, long s_tile_stride,
                        long s_slice_stride, int s_ld, int n_tiles, int T, int slices) {
  pmd_prof_scope prof__(ctx, "tile_xbt");
  if (n_tiles <= 0 || T <= 0) return PMD_OK;
  const int n_groups = (T + 15) / 16;
  if (slices < 1) slices = 1;
  const int slices_asked = slices;
  if (slices > n_groups) slices = n_groups;
  const int gps = (n_groups + slices - 1) / slices;
  slices = (n_groups + gps - 1) / gps;
  const int mtiles = (d + 15) / 16;
  if (s_ld < 16 * mtiles) return pmd_fail(ctx, PMD_ERR_ARG, "tile_xbt", "s_ld too small");
  if (slices < slices_asked && s_slice_stride > 0) {
    // Short rows (T <= 48, 65-96 or 129-144 with four slices): fewer slices have work than the caller's buffer holds, and
    // the caller sums all of them.  The slices without work are cleared here (found by the seeded fuzz with poisoned
    // allocations: a 133-frame movie picked up stale workspace contents through the Gram matrix of V_ds).
    PMD_HIP(ctx, hipMemset2DAsync(S + (long)slices * s_slice_stride, (size_t)s_tile_stride * sizeof(float), 0,
                                  (size_t)(slices_asked - slices) * s_slice_stride * sizeof(float), (size_t)n_tiles, ctx->stream));
  }
  // M tiles per wave: spread small tiles over the four waves, 7 per wave (112 accumulator VGPRs) at most
  // PMD_XBT_LDS=0: every wave loads all of B itself (the form of rounds 1-2, A/B runs)
  static int blds = -1;
  if (blds < 0) { const char* e = getenv("PMD_XBT_LDS"); blds = (e && !strcmp(e, "0")) ? 0 : 1; }
#define XBT_LAUNCH(MPW_)                                                                                              \
  {                                                                                                                   \
    const int mblocks = (mtiles + 4 * MPW_ - 1) / (4 * MPW_);                                                         \
    if (blds)                                                                                                         \
      hipLaunchKernelGGL(tile_xbt_lds_kernel<MPW_>, dim3(n_tiles, slices, mblocks), dim3(256), 0, ctx->stream, X, ldx, pix, \
                         pix_stride, row0_stride, d, B, b_tile_stride, ldb, S, s_tile_stride, s_slice_stride, s_ld,   \
                         n_groups, gps);                                                                              \
    else                                                                                                              \
      hipLaunchKernelGGL(tile_xbt_kernel<MPW_>, dim3(n_tiles, slices, mblocks), dim3(256), 0, ctx->stream, X, ldx, pix, \
                         pix_stride, row0_stride, d, B, b_tile_stride, ldb, S, s_tile_stride, s_slice_stride, s_ld,   \
                         n_groups, gps);                                                                              \
  }
  if (mtiles <= 4) XBT_LAUNCH(1)
  else if (mtiles <= 8) XBT_LAUNCH(2)
  else if (mtiles <= 16) XBT_LAUNCH(4)
  else XBT_LAUNCH(7)
#undef XBT_LAUNCH
  PMD_LAUNCH_CHECK(ctx, "tile_xbt_kernel");
  return PMD_OK;
}

// ------------------------------------------------------------------------------------------
// tile_gram (fp64 accumulation of fp32 inputs).  256 threads = 16 x 16, each a 4 x 4 block.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_gram_kernel(const float* __restrict__ In, long tile_stride, long ld,
                                                        int len, int chunk_per_slice, double* __restrict__ G,
                                                        long g_tile_stride) {
  __shared__ float buf[64][33];
  const int tile = blockIdx.x, slice = blockIdx.y;
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  const float* in = In + (long)tile * tile_stride;
  const int x_begin = slice * chunk_per_slice;
  const int x_end = min(len, x_begin + chunk_per_slice);
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int x0 = x_begin; x0 < x_end; x0 += 32) {
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
      const int r = i >> 5, cx = i & 31;
      buf[r][cx] = (x0 + cx < x_end) ? in[(long)r * ld + x0 + cx] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int cx = 0; cx < 32; ++cx) {
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) { av[a] = (double)buf[4 * ti + a][cx]; bv[a] = (double)buf[4 * tj + a][cx]; }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
    }
    __syncthreads();
  }
  double* g = G + (long)tile * g_tile_stride + (long)slice * 4096;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) g[(4 * ti + a) * 64 + 4 * tj + b] = acc[a][b];
}

// fp64-MFMA form (v_mfma_f64_16x16x4_f64: A[m = l%16][k = l/16], B[k = l/16][n = l%16], D[4i + l/16][l%16]).
// Wave w owns the 16-row tile w of the 64 x 64 result (four 16 x 16 accumulators).  Per macro step of 16
// positions a lane loads one float4 per row tile (row 16t + l%16, positions 4(l/16) .. +3); element s of those
// float4 feeds MFMA s, i.e. the K slots walk the positions in a permuted order - harmless for a sum over all of
// them as long as both operands use the same permutation, which they do (they are the same registers).
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void tile_gram_mfma_kernel(const float* __restrict__ In, long tile_stride, long ld,
                                                             int len, int chunk_per_slice, double* __restrict__ G,
                                                             long g_tile_stride) {
  const int tile = blockIdx.x, slice = blockIdx.y;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform, and known to be)
  const int m16 = lane & 15, kq = lane >> 4;
  const float* in = In + (long)tile * tile_stride + (long)m16 * ld + 4 * kq;
  const int x_begin = slice * chunk_per_slice;
  const int x_end = min(len, x_begin + chunk_per_slice);
  f64x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f64x4){0.0, 0.0, 0.0, 0.0};
  // one step = 32 positions = two quads per lane and row tile; the next step's eight loads are in flight
  // while the 32 MFMAs of the current one run
  auto load = [&](int x0, f32x4 (&v)[8]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // unconditional loads (rows always exist; positions are clamped to the row's last aligned quad), masked below
      const int xh = x0 + 16 * h;
      const int xq = min(xh, ((int)ld - 4) - 4 * kq);
#pragma unroll
      for (int t = 0; t < 4; ++t) v[4 * h + t] = *reinterpret_cast<const f32x4*>(in + (long)(16 * t) * ld + xq);
      const int xs = xh + 4 * kq;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (xs + e >= x_end || xq != xh) v[4 * h + t][e] = 0.f;
    }
  };
  f32x4 cur[8], nxt[8];
  if (x_begin < x_end) load(x_begin, cur);
  for (int x0 = x_begin; x0 < x_end; x0 += 32) {
    const bool more = x0 + 32 < x_end;
    if (more) load(x0 + 32, nxt);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double a = (double)(w == 0 ? cur[4 * h][e] : w == 1 ? cur[4 * h + 1][e] : w == 2 ? cur[4 * h + 2][e] : cur[4 * h + 3][e]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (double)cur[4 * h + t][e], acc[t], 0, 0, 0);
      }
    if (more) {
#pragma unroll
      for (int t = 0; t < 8; ++t) cur[t] = nxt[t];
    }
  }
  double* g = G + (long)tile * g_tile_stride + (long)slice * 4096;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) g[(16 * w + 4 * i + kq) * 64 + 16 * t + m16] = acc[t][i];
}

// G: [tile][slices][64][64] doubles, g_tile_stride = slices*4096.
int pmd_launch_tile_gram(pmd_ctx* ctx, const float* In, long tile_stride, long ld, int len, int n_tiles, int slices,
                         double* G) {
  pmd_prof_scope prof__(ctx, "tile_gram");
  if (n_tiles <= 0) return PMD_OK;
  if (slices < 1) slices = 1;
  int cps = (len + slices - 1) / slices;
  cps = (int)pmd_round_up(cps, 32);
  const char* gm = getenv("PMD_GRAM_MFMA");
  if (!(gm && !strcmp(gm, "0")) && ld % 4 == 0 && ld >= 16 && tile_stride % 4 == 0 && !((uintptr_t)In & 15))
    hipLaunchKernelGGL(tile_gram_mfma_kernel, dim3(n_tiles, slices), dim3(256), 0, ctx->stream, In, tile_stride, ld, len,
                       cps, G, (long)slices * 4096);
  else
    hipLaunchKernelGGL(tile_gram_kernel, dim3(n_tiles, slices), dim3(256), 0, ctx->stream, In, tile_stride, ld, len, cps,
                       G, (long)slices * 4096);
  PMD_LAUNCH_CHECK(ctx, "tile_gram_kernel");
  return PMD_OK;
}

// ------------------------------------------------------------------------------------------
// sum over slices:  out[tile][i] = sum_s in[tile][s][i]   (fp32 in, fp64 accumulate, fp32 out)
// ------------------------------------------------------------------------------------------
__global__ void reduce_slices_kernel(const float* __restrict__ in, long tile_stride, long slice_stride, int slices,
                                     long n, float* __restrict__ out, long out_tile_stride) {
  const int tile = blockIdx.y;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int k = 0; k < slices; ++k) s += (double)in[(long)tile * tile_stride + (long)k * slice_stride + i];
    out[(long)tile * out_tile_stride + i] = (float)s;
  }
}

// the same with 16-byte accesses, one quad per thread, all slice loads of a thread issued before the first add
// (four quads per thread in a loop was 2.2x SLOWER than the scalar kernel: each iteration waited for its own loads)
__global__ __launch_bounds__(256) void reduce_slices4_kernel(const float4* __restrict__ in, long tile_stride4, long slice_stride4,
                                                             int slices, long n4, float4* __restrict__ out, long out_tile_stride4) {
  const int tile = blockIdx.y;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4* src = in + (long)tile * tile_stride4 + i;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (slices == 4) {
    const float4 a = src[0], b = src[slice_stride4], c = src[2 * slice_stride4], d = src[3 * slice_stride4];
    s0 = (double)a.x + (double)b.x + (double)c.x + (double)d.x;
    s1 = (double)a.y + (double)b.y + (double)c.y + (double)d.y;
    s2 = (double)a.z + (double)b.z + (double)c.z + (double)d.z;
    s3 = (double)a.w + (double)b.w + (double)c.w + (double)d.w;
  } else {
    // many slices (the background projection sums one partial per block of 1024 pixels): eight loads in flight per thread
    int k = 0;
    for (; k + 8 <= slices; k += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(long)(k + u) * slice_stride4];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
    }
    for (; k < slices; ++k) {
      const float4 v = src[(long)k * slice_stride4];
      s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
    }
  }
  out[(long)tile * out_tile_stride4 + i] = make_float4((float)s0, (float)s1, (float)s2, (float)s3);
}

int pmd_launch_reduce_slices(pmd_ctx* ctx, const float* in, long tile_stride, long slice_stride, int slices, long n,
                             float* out, long out_tile_stride, int n_tiles) {
  pmd_prof_scope prof__(ctx, "reduce_slices");
  if (n % 4 == 0 && tile_stride % 4 == 0 && slice_stride % 4 == 0 && out_tile_stride % 4 == 0 && !((uintptr_t)in & 15) &&
      !((uintptr_t)out & 15)) {
    const long n4 = n / 4;
    const unsigned bx4 = (unsigned)((n4 + 255) / 256);
    for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
      const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
      hipLaunchKernelGGL(reduce_slices4_kernel, dim3(bx4, tn), dim3(256), 0, ctx->stream,
                         reinterpret_cast<const float4*>(in + (long)t0 * tile_stride), tile_stride / 4, slice_stride / 4, slices, n4,
                         reinterpret_cast<float4*>(out + (long)t0 * out_tile_stride), out_tile_stride / 4);
      PMD_LAUNCH_CHECK(ctx, "reduce_slices4_kernel");
    }
    return PMD_OK;
  }
  int bx = (int)((n + 255) / 256);
  if (bx > 64) bx = 64;
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3(bx, tn), dim3(256), 0, ctx->stream, in + (long)t0 * tile_stride,
                       tile_stride, slice_stride, slices, n, out + (long)t0 * out_tile_stride, out_tile_stride);
    PMD_LAUNCH_CHECK(ctx, "reduce_slices_kernel");
  }
  return PMD_OK;
}

// ------------------------------------------------------------------------------------------
// tile_rowmix: Out[tile][c][x] = sum_{c' < n_in} N[tile][c'][c] * In[tile][c'][x], c < n_out;
// rows c in [n_out, 64) are zeroed.  In-place safe (a thread owns column x).  N is
// [tile][64][64] doubles (n_tile_stride = 0 shares one matrix between tiles).
// ------------------------------------------------------------------------------------------
template <int NOUT>
__global__ __launch_bounds__(256) void tile_rowmix_kernel(const float* __restrict__ In, long in_tile_stride, long ld_in,
                                                          const double* __restrict__ N, long n_tile_stride, int n_in,
                                                          int n_out, float* __restrict__ Out, long out_tile_stride,
                                                          long ld_out, int len) {
  __shared__ double nm[64 * NOUT];
  const int tile = blockIdx.y;
  const double* nsrc = N + (long)tile * n_tile_stride;
  for (int i = threadIdx.x; i < 64 * NOUT; i += 256) {
    const int cp = i / NOUT, c = i - cp * NOUT;
    nm[i] = (cp < n_in && c < n_out) ? nsrc[cp * 64 + c] : 0.0;
  }
  __syncthreads();
  const float* in = In + (long)tile * in_tile_stride;
  float* out = Out + (long)tile * out_tile_stride;
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < len; x += gridDim.x * blockDim.x) {
    double acc[NOUT];
#pragma unroll
    for (int c = 0; c < NOUT; ++c) acc[c] = 0.0;
    // eight input rows in flight per thread (clamped index: no branch between the loads)
    for (int cp0 = 0; cp0 < n_in; cp0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = in[(long)((cp0 + u < n_in) ? cp0 + u : n_in - 1) * ld_in + x];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (cp0 + u < n_in) {
          const double vd = (double)v[u];
#pragma unroll
          for (int c = 0; c < NOUT; ++c) acc[c] = fma(nm[(cp0 + u) * NOUT + c], vd, acc[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NOUT; ++c) out[(long)c * ld_out + x] = (c < n_out) ? (float)acc[c] : 0.f;
    for (int c = NOUT; c < 64; ++c) out[(long)c * ld_out + x] = 0.f;
  }
}

// fp64-MFMA form for n_out > 4.  Wave w owns output rows 16 w .. 16 w + 15: its slice of N^T (A operand,
// A[m][k] = N[k][16 w + m]) stays in registers (16 doubles per lane) for the whole sweep; per chunk of 16
// positions a lane loads In[4 ks + l/16][x0 + l%16] for the 16 K steps (B operand) and issues 16 MFMAs.
// In-place safe: all four waves have consumed the chunk's inputs (barrier) before anyone stores its outputs;
// the next chunk (other positions) is prefetched before that barrier.
__global__ __launch_bounds__(256) void tile_rowmix_mfma_kernel(const float* __restrict__ In, long in_tile_stride,
                                                               long ld_in, const double* __restrict__ N,
                                                               long n_tile_stride, int n_in, int n_out,
                                                               float* __restrict__ Out, long out_tile_stride,
                                                               long ld_out, int len) {
  const int tile = blockIdx.y;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform, and known to be)
  const int n16 = lane & 15, kq = lane >> 4;
  const double* nsrc = N + (long)tile * n_tile_stride;
  double a[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int cp = 4 * ks + kq, c = 16 * w + n16;
    const double v = nsrc[cp * 64 + c];   // the 64 x 64 block always exists; entries outside n_in x n_out are masked
    a[ks] = (cp < n_in && c < n_out) ? v : 0.0;
  }
  const float* in = In + (long)tile * in_tile_stride;
  float* out = Out + (long)tile * out_tile_stride;
  const int ksteps = (n_in + 3) / 4;
  auto load = [&](int x0, float (&v)[16]) {
    const int x = min(x0 + n16, len - 1);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float t = in[(long)min(4 * ks + kq, 63) * ld_in + x];   // unconditional, rows < 64 exist
      v[ks] = (4 * ks + kq < n_in) ? t : 0.f;                       // rows >= n_in may hold anything: 0 * NaN
    }
  };
  float cur[16], nxt[16];
  int x0 = blockIdx.x * 16;
  if (x0 < len) load(x0, cur);
  for (; x0 < len; x0 += gridDim.x * 16) {
    const int xn = x0 + gridDim.x * 16;
    if (xn < len) load(xn, nxt);
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (ks < ksteps) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], (double)cur[ks], acc, 0, 0, 0);
    __syncthreads();   // every wave holds this chunk's inputs in registers: outputs may overwrite them now
    if (x0 + n16 < len) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 16 * w + 4 * i + kq;
        out[(long)c * ld_out + x0 + n16] = (c < n_out) ? (float)acc[i] : 0.f;
      }
    }
    if (xn < len) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) cur[ks] = nxt[ks];
    }
  }
}

// The same with 64 positions per step: a lane loads one float4 per K step (16 lanes x 16 B = 256 contiguous bytes per row,
// four rows per instruction) and element e of it is the B operand of MFMA e, whose N index n then stands for position
// x0 + 4 n + e; the four accumulators of a lane hold four consecutive positions of its output rows and leave as float4.
// A quarter of the barriers and of the load / store instructions of the 16-position form: 18 -> 8 ms per step on the
// 16 129-tile workload.  Needs 16-byte aligned rows (ld % 4 == 0) that can be read up to the next multiple of 4 behind len.
__global__ __launch_bounds__(256) void tile_rowmix_mfma4_kernel(const float* __restrict__ In, long in_tile_stride,
                                                                long ld_in, const double* __restrict__ N,
                                                                long n_tile_stride, int n_in, int n_out,
                                                                float* __restrict__ Out, long out_tile_stride,
                                                                long ld_out, int len) {
  const int tile = blockIdx.y;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform, and known to be)
  const int n16 = lane & 15, kq = lane >> 4;
  const double* nsrc = N + (long)tile * n_tile_stride;
  double a[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int cp = 4 * ks + kq, c = 16 * w + n16;
    const double v = nsrc[cp * 64 + c];   // the 64 x 64 block always exists; entries outside n_in x n_out are masked
    a[ks] = (cp < n_in && c < n_out) ? v : 0.0;
  }
  const float* in = In + (long)tile * in_tile_stride;
  float* out = Out + (long)tile * out_tile_stride;
  const int ksteps = (n_in + 3) / 4;
  const int len4 = (len + 3) & ~3;
  auto load = [&](int x0, f32x4 (&v)[16]) {
    const int x = min(x0 + 4 * n16, len4 - 4);                       // clamped: a valid quad of the row
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(in + (long)min(4 * ks + kq, 63) * ld_in + x);   // unconditional, rows < 64 exist
      const bool ok = 4 * ks + kq < n_in;                             // rows >= n_in may hold anything: 0 * NaN
      v[ks] = ok ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x4 cur[16], nxt[16];
  int x0 = blockIdx.x * 64;
  if (x0 < len) load(x0, cur);
  for (; x0 < len; x0 += gridDim.x * 64) {
    const int xn = x0 + gridDim.x * 64;
    if (xn < len) load(xn, nxt);
    f64x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (ks < ksteps) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], (double)cur[ks][e], acc[e], 0, 0, 0);
      }
    __syncthreads();   // every wave holds this chunk's inputs in registers: outputs may overwrite them now
    const int xs = x0 + 4 * n16;
    if (xs < len) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 16 * w + 4 * i + kq;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (c < n_out) ? (float)acc[e][i] : 0.f;
        float* dst = out + (long)c * ld_out + xs;
        if (xs + 3 < len) *reinterpret_cast<f32x4*>(dst) = o;
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (xs + e < len) dst[e] = o[e];
        }
      }
    }
    if (xn < len) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) cur[ks] = nxt[ks];
    }
  }
}

int pmd_launch_tile_rowmix(pmd_ctx* ctx, const float* In, long in_tile_stride, long ld_in, const double* N,
                           long n_tile_stride, int n_in, int n_out, float* Out, long out_tile_stride, long ld_out,
                           int len, int n_tiles) {
  pmd_prof_scope prof__(ctx, "tile_rowmix");
  if (n_tiles <= 0 || len <= 0) return PMD_OK;
  int bx = (len + 255) / 256;
  if (bx > 64) bx = 64;
  for (int t0 = 0; t0 < n_tiles; t0 += 32768) {
    const int tn = (n_tiles - t0 < 32768) ? n_tiles - t0 : 32768;
    const float* in = In + (long)t0 * in_tile_stride;
    const double* nn = N + (long)t0 * n_tile_stride;
    float* out = Out + (long)t0 * out_tile_stride;
    const char* rm = getenv("PMD_ROWMIX_MFMA");
    const bool quads = ld_in % 4 == 0 && ld_out % 4 == 0 && in_tile_stride % 4 == 0 && out_tile_stride % 4 == 0 && !((uintptr_t)in & 15) &&
                       !((uintptr_t)out & 15) && ld_in >= ((len + 3) & ~3) && len >= 4 && !(rm && !strcmp(rm, "16"));
    // (the 64-position form pays with 200 VGPRs - two workgroups per CU: it wins where tiles are plentiful and each one is
    // small - 13.3 against 18.0 ms at 16 129 tiles x 1000 frames - and loses at 2601 tiles x 10^4 frames, 11.6 against 7.7)
    if (n_out > 4 && !(rm && !strcmp(rm, "0")) && quads && (n_tiles >= 8192 || (rm && !strcmp(rm, "64")))) {
      // workgroups per tile: each one first loads its slice of the 64 x 64 mixing matrix (32 KB per tile and workgroup),
      // so no more of them than it takes to fill the chip (~8192 workgroups in all)
      int bxm = (len + 63) / 64;
      const int want = (8192 + tn - 1) / tn;
      if (bxm > want) bxm = want;
      if (bxm > 16) bxm = 16;
      hipLaunchKernelGGL(tile_rowmix_mfma4_kernel, dim3(bxm, tn), dim3(256), 0, ctx->stream, in, in_tile_stride, ld_in, nn,
                         n_tile_stride, n_in, n_out, out, out_tile_stride, ld_out, len);
    } else if (n_out > 4 && !(rm && !strcmp(rm, "0"))) {
      int bxm = (len + 15) / 16;
      if (bxm > 16) bxm = 16;
      hipLaunchKernelGGL(tile_rowmix_mfma_kernel, dim3(bxm, tn), dim3(256), 0, ctx->stream, in, in_tile_stride, ld_in, nn,
                         n_tile_stride, n_in, n_out, out, out_tile_stride, ld_out, len);
    } else if (n_out <= 4)
      hipLaunchKernelGGL(tile_rowmix_kernel<4>, dim3(bx, tn), dim3(256), 0, ctx->stream, in, in_tile_stride, ld_in, nn,
                         n_tile_stride, n_in, n_out, out, out_tile_stride, ld_out, len);
    else if (n_out <= 32)
      hipLaunchKernelGGL(tile_rowmix_kernel<32>, dim3(bx, tn), dim3(256), 0, ctx->stream, in, in_tile_stride, ld_in, nn,
                         n_tile_stride, n_in, n_out, out, out_tile_stride, ld_out, len);
    else
      hipLaunchKernelGGL(tile_rowmix_kernel<64>, dim3(bx, tn), dim3(256), 0, ctx->stream, in, in_tile_stride, ld_in, nn,
                         n_tile_stride, n_in, n_out, out, out_tile_stride, ld_out, len);
    PMD_LAUNCH_CHECK(ctx, "tile_rowmix_kernel");
  }
  return PMD_OK;
}

// float partial Gram blocks (tile_xbt output, [tile][slice][64][ld]) -> double [tile][slice][64][64]
__global__ void gram_f2d_kernel(const float* __restrict__ in, int ld, long n_blocks, double* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_blocks * 4096) return;
  const long b = i >> 12;
  const int r = (int)((i >> 6) & 63), c = (int)(i & 63);
  out[i] = (double)in[b * 64 * ld + (long)r * ld + c];
}

int pmd_launch_gram_f2d(pmd_ctx* ctx, const float* in, int ld, long n_blocks, double* out) {
  hipLaunchKernelGGL(gram_f2d_kernel, dim3((unsigned)((n_blocks * 4096 + 255) / 256)), dim3(256), 0, ctx->stream, in, ld,
                     n_blocks, out);
  PMD_LAUNCH_CHECK(ctx, "gram_f2d_kernel");
  return PMD_OK;
}
