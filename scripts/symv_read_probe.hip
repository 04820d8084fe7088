// Raw read rate of the triangle-tile access pattern of sytrd_symv (no reduction beyond a per-thread sum):
// which tile shape / loads in flight the memory system likes.  Not part of the library.
// build: hipcc -O3 --offload-arch=gfx950 scripts/symv_read_probe.hip -o scripts/symv_read_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != hipSuccess) { printf("fail %s = %d\n", #x, (int)e_); exit(1);} } while (0)

// tile: TR rows x TC positions, 512 threads: TC/4 float4 per row; each thread loads RPT = TR*TC/4/512 float4.
// Thread layout: lanes run along positions (coalesced), waves split (position groups) x (row groups).
template <int TR, int TC>
__global__ __launch_bounds__(512) void read_tiles(const float* __restrict__ A, long ld, int n, int cs, const int2* __restrict__ tiles,
                                                  float* __restrict__ out) {
  constexpr int F4_PER_ROW = TC / 4;                                     // float4 per tile row
  constexpr int LANES_PER_ROW = F4_PER_ROW < 512 ? F4_PER_ROW : 512;      // threads along a row
  constexpr int ROW_GROUPS = 512 / LANES_PER_ROW;                         // concurrent rows
  constexpr int PASSES = F4_PER_ROW / LANES_PER_ROW;                      // float4 per thread per row
  constexpr int RPT = TR / ROW_GROUPS;                                    // rows per thread
  const int2 t = tiles[blockIdx.x];
  const int cb = t.x, p0 = t.y;
  const int tid = threadIdx.x;
  const int lr = tid % LANES_PER_ROW, rg = tid / LANES_PER_ROW;
  float4 a[RPT * PASSES];
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    const int row = min(cb + rg * RPT + u, n - 1);
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      int pos = p0 + (p * LANES_PER_ROW + lr) * 4;
      pos = (pos + 3 < ld) ? pos : p0;
      a[u * PASSES + p] = *reinterpret_cast<const float4*>(A + (long)row * ld + pos);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < RPT * PASSES; ++u) s += (a[u].x + a[u].y) + (a[u].z + a[u].w);
  if (s == 12345.678f) out[blockIdx.x] = s;  // keep the loads
}

template <int TR, int TC>
static void run(const float* A, long ld, int n, int cs, float* out, int lds_bytes = 0) {
  std::vector<int2> tiles;
  for (int cb = cs; cb < n; cb += TR) {
    const int r0 = cb & ~3;
    for (int p0 = r0; p0 < n; p0 += TC) tiles.push_back(make_int2(cb, p0));
  }
  int2* dt; CK(hipMalloc(&dt, tiles.size() * sizeof(int2)));
  CK(hipMemcpy(dt, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  if (lds_bytes > 65536) CK(hipFuncSetAttribute((const void*)read_tiles<TR, TC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((read_tiles<TR, TC>), dim3(tiles.size()), dim3(512), lds_bytes, 0, A, ld, n, cs, dt, out);
  const int reps = 20;
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((read_tiles<TR, TC>), dim3(tiles.size()), dim3(512), lds_bytes, 0, A, ld, n, cs, dt, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double np = n - cs;
  const double alg = 4.0 * np * (np + 1) / 2, touched = (double)tiles.size() * TR * TC * 4;
  printf("lds %6d  n'=%5d tile %3d x %4d: %6zu tiles %7.1f us  algorithmic %.2f TB/s  touched %.2f TB/s\n", lds_bytes, n - cs, TR, TC, tiles.size(),
         ms * 1e3, alg / (ms * 1e-3) / 1e12, touched / (ms * 1e-3) / 1e12);
  CK(hipFree(dt));
}

int main() {
  const int n = 10000; const long ld = 10000;
  float* A; CK(hipMalloc(&A, (size_t)n * ld * 4 + 65536)); CK(hipMemset(A, 0, (size_t)n * ld * 4 + 65536));
  float* out; CK(hipMalloc(&out, 1 << 22));
  // workgroups per CU limited through the dynamic LDS allocation: 0 -> registers decide (3), 60 KB -> 2, 100 KB -> 1
  for (int cs : {1, 4001}) {
    for (int lds : {0, 40 * 1024, 60 * 1024, 100 * 1024}) {
      run<64, 512>(A, ld, n, cs, out, lds);
      run<32, 512>(A, ld, n, cs, out, lds);
    }
  }
  return 0;
}
