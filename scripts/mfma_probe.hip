// Micro-probe: what limits v_mfma_f32_16x16x4_f32 fed from LDS?  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define KJW 25
#define LSTR 36

template <int MODE, int NWAVES>
__global__ __launch_bounds__(64 * NWAVES) void probe(const float* __restrict__ A, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n16 = lane & 15, kk = lane >> 4;
  for (int i = tid; i < 400 * LSTR; i += 64 * NWAVES) lds[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  f32x4 areg[KJW];
  for (int J = 0; J < KJW; ++J) areg[J] = *reinterpret_cast<const f32x4*>(A + (wid & 3) * 1600 + n16 * 100 + 4 * J);
  const float* xrd = lds + (4 * kk) * LSTR + n16;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // registers only
#pragma unroll
      for (int J = 0; J < KJW; ++J)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][s], areg[(J + 1) % KJW][s], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[J][s], areg[(J + 2) % KJW][s], acc1, 0, 0, 0);
        }
    } else {  // B from LDS, prefetched one step ahead
      float b0[4], b1[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { b0[s] = xrd[s * LSTR]; b1[s] = xrd[s * LSTR + 16]; }
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
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
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        if (J + 1 < KJW) {
#pragma unroll
          for (int s = 0; s < 4; ++s) { b0[s] = n0[s]; b1[s] = n1[s]; }
        }
      }
      if (MODE == 2) __syncthreads();
    }
  }
  out[blockIdx.x * 64 * NWAVES + tid] = acc0[0] + acc1[1] + acc0[2] + acc1[3];
}

// chunk loop with the refill path of tile_atx: MODE bit0 = global loads, bit1 = ds_write refill (double buffer),
// bit2 = output stores, bit3 = barrier per chunk
template <int MODE>
__global__ __launch_bounds__(256) void probe2(const float* __restrict__ A, const float* __restrict__ X, long ldx,
                                              float* __restrict__ out, long ldo, int chunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n16 = lane & 15, kk = lane >> 4;
  for (int i = tid; i < 2 * 400 * LSTR; i += 256) lds[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  f32x4 areg[KJW];
  for (int J = 0; J < KJW; ++J) areg[J] = *reinterpret_cast<const f32x4*>(A + (wid & 3) * 1600 + n16 * 100 + 4 * J);
  long goff[13]; int loff[13];
  for (int k = 0; k < 13; ++k) {
    const int i = tid + k * 256; const int q = (i / 8) % 400, j = i % 8;
    goff[k] = ((long)blockIdx.x * 400 + q) * ldx + 4 * j; loff[k] = q * LSTR + 4 * j;
  }
  f32x4 pre[13];
  for (int k = 0; k < 13; ++k) pre[k] = (f32x4){0, 0, 0, 0};
  float* outp = out + (long)blockIdx.x * 64 * ldo + (long)(16 * wid + 4 * kk) * ldo + n16;
  for (int c = 0; c < chunks; ++c) {
    const int cur = c & 1;
    const float* xrd = lds + cur * 400 * LSTR + (4 * kk) * LSTR + n16;
    if (MODE & 1) {
#pragma unroll
      for (int k = 0; k < 13; ++k) pre[k] = *reinterpret_cast<const f32x4*>(X + goff[k] + (long)(c + 1) * 32);
    }
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float b0[4], b1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { b0[s] = xrd[s * LSTR]; b1[s] = xrd[s * LSTR + 16]; }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
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
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      if (J + 1 < KJW) {
#pragma unroll
        for (int s = 0; s < 4; ++s) { b0[s] = n0[s]; b1[s] = n1[s]; }
      }
    }
    if (MODE & 4) {
      float* o = outp + (long)c * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) { o[(long)i * ldo] = acc0[i]; o[(long)i * ldo + 16] = acc1[i]; }
    } else {
      asm volatile("" ::"v"(acc0), "v"(acc1));
    }
    if (MODE & 2) {
      float* xw = lds + (cur ^ 1) * 400 * LSTR;
#pragma unroll
      for (int k = 0; k < 13; ++k) *reinterpret_cast<f32x4*>(xw + loff[k]) = pre[k];
    } else {
      asm volatile("" ::"v"(pre[0]), "v"(pre[5]), "v"(pre[12]));
    }
    if (MODE & 8) __syncthreads();
  }
  if (!(MODE & 4)) out[blockIdx.x * 256 + tid] = lds[tid];
}

// LDS-DMA variant: 3 dense, XOR-swizzled buffers; loads two chunks ahead; counted vmcnt + raw barrier.
// stores: STORES = 1 writes the output tile each chunk.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
template <int STORES>
__global__ __launch_bounds__(256) void probe3(const float* __restrict__ A, const float* __restrict__ X, long ldx,
                                              float* __restrict__ out, long ldo, int chunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BUF = 400 * 32;  // floats per buffer
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n16 = lane & 15, kk = lane >> 4;
  for (int i = tid; i < 3 * BUF; i += 256) lds[i] = 0.f;
  __syncthreads();
  f32x4 areg[KJW];
  for (int J = 0; J < KJW; ++J) areg[J] = *reinterpret_cast<const f32x4*>(A + (wid & 3) * 1600 + n16 * 100 + 4 * J);
  // DMA items: instruction m (0..49) covers rows 8m..8m+7; wave w issues m = w, w+4, ... (13 slots, last may be void)
  const float* gsrc[13];
  int ldst[13];
#pragma unroll
  for (int k = 0; k < 13; ++k) {
    const int m = wid + 4 * k;
    const int mm = m < 50 ? m : 49;
    const int q = 8 * mm + (lane >> 3), p = lane & 7;
    const int f = 4 * ((q >> 2) & 1);
    gsrc[k] = X + ((long)blockIdx.x * 400 + q) * ldx + 4 * (p ^ f);
    ldst[k] = (8 * mm) * 32;  // wave-uniform float offset of the 1-KiB piece
  }
  auto issue = [&](int c, int buf) {
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      if (wid + 4 * k < 50)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gsrc[k] + (long)c * 32), (lds_ptr_t)(lds + buf * BUF + ldst[k]), 16, 0, 0);
    }
  };
  issue(0, 0);
  issue(1, 1);
  asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int f = 4 * (kk & 1);
  const int off0 = (4 * kk) * 32 + (((n16 >> 2) ^ f) << 2) + (n16 & 3);
  const int off1 = (4 * kk) * 32 + (((4 + (n16 >> 2)) ^ f) << 2) + (n16 & 3);
  float* outp = out + (long)blockIdx.x * 64 * ldo + (long)(16 * wid + 4 * kk) * ldo + n16;
  for (int c = 0; c < chunks; ++c) {
    const int cur = c % 3;
    issue(c + 2, (c + 2) % 3);
    const float* x0 = lds + cur * BUF + off0;
    const float* x1 = lds + cur * BUF + off1;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
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
    if (STORES) {
      float* o = outp + (long)c * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) { o[(long)i * ldo] = acc0[i]; o[(long)i * ldo + 16] = acc1[i]; }
      asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
    } else {
      asm volatile("" ::"v"(acc0), "v"(acc1));
      asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!STORES) out[blockIdx.x * 256 + tid] = lds[tid];
}

template <int STORES>
void run3(const char* name, int nblk, int chunks) {
  float *A, *X, *out;
  const long ldx = 10112, ldo = 10112;
  hipMalloc(&A, 1 << 20); hipMemset(A, 0, 1 << 20);
  hipMalloc(&X, (size_t)nblk * 400 * ldx * 4); hipMemset(X, 0, (size_t)nblk * 400 * ldx * 4);
  hipMalloc(&out, (size_t)nblk * 64 * ldo * 4);
  const size_t lds = 3 * 400 * 32 * 4;
  hipFuncSetAttribute((const void*)probe3<STORES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe3<STORES>), dim3(nblk), dim3(256), lds, 0, A, X, ldx, out, ldo, 4);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe3<STORES>), dim3(nblk), dim3(256), lds, 0, A, X, ldx, out, ldo, chunks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)nblk * 4 * chunks * 200.0 * 2048.0;
  printf("%-52s %8.3f ms  %7.1f TF/s issued\n", name, ms, flops / ms / 1e9);
  hipFree(A); hipFree(X); hipFree(out);
}

template <int MODE>
void run2(const char* name, int nblk, int chunks) {
  float *A, *X, *out;
  const long ldx = 10112, ldo = 10112;
  hipMalloc(&A, 1 << 20); hipMemset(A, 0, 1 << 20);
  hipMalloc(&X, (size_t)nblk * 400 * ldx * 4); hipMemset(X, 0, (size_t)nblk * 400 * ldx * 4);
  hipMalloc(&out, (size_t)nblk * 64 * ldo * 4);
  const size_t lds = 2 * 400 * LSTR * 4 + 64;
  hipFuncSetAttribute((const void*)probe2<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe2<MODE>), dim3(nblk), dim3(256), lds, 0, A, X, ldx, out, ldo, 4);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe2<MODE>), dim3(nblk), dim3(256), lds, 0, A, X, ldx, out, ldo, chunks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)nblk * 4 * chunks * 200.0 * 2048.0;
  printf("%-52s %8.3f ms  %7.1f TF/s issued\n", name, ms, flops / ms / 1e9);
  hipFree(A); hipFree(X); hipFree(out);
}

template <int MODE, int NWAVES>
void run(const char* name, int blocks_per_cu) {
  float *A, *out;
  hipMalloc(&A, 1 << 20);
  hipMemset(A, 0, 1 << 20);
  const int nblk = 256 * blocks_per_cu;
  hipMalloc(&out, nblk * 64 * NWAVES * 4);
  const int iters = 2000;
  const size_t lds = 400 * LSTR * 4;
  hipFuncSetAttribute((const void*)probe<MODE, NWAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, NWAVES>), dim3(nblk), dim3(64 * NWAVES), lds, 0, A, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, NWAVES>), dim3(nblk), dim3(64 * NWAVES), lds, 0, A, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)nblk * NWAVES * iters * 200.0 * 2048.0;
  printf("%-44s %8.3f ms  %7.1f TF/s\n", name, ms, flops / ms / 1e9);
}

int main() {
  run<0, 4>("regs only, 4 waves/CU", 1);
  run<0, 8>("regs only, 8 waves/CU", 1);
  run<1, 4>("B from LDS (prefetch), 4 waves/CU", 1);
  run<1, 8>("B from LDS (prefetch), 8 waves/CU", 1);
  run<1, 4>("B from LDS, 2 blocks/CU x 4 waves", 2);
  run<2, 4>("B from LDS + barrier per 200 MFMA, 4 waves", 1);
  run<2, 4>("B from LDS + barrier, 2 blocks/CU", 2);
  const int nb = 2560, ch = 312;
  run2<8>("chunk loop: barrier only", nb, ch);
  run2<9>("chunk loop: + global loads (HBM stream)", nb, ch);
  run2<11>("chunk loop: + loads + LDS refill", nb, ch);
  run2<15>("chunk loop: + loads + refill + stores (= tile_atx)", nb, ch);
  run2<12>("chunk loop: stores only + barrier", nb, ch);
  run2<10>("chunk loop: LDS refill only (no loads) + barrier", nb, ch);
  run3<0>("LDS-DMA ring (3 buffers, 2 ahead), no stores", nb, ch - 2);
  run3<1>("LDS-DMA ring + stores", nb, ch - 2);
  return 0;
}
