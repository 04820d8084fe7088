// Internal header of libpmd_hip.so (gfx950 only).  Public C ABI: include/pmd_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define PMD_OK 0
#define PMD_ERR_HIP -1
#define PMD_ERR_ARG -2
#define PMD_ERR_WORKSPACE -3
#define PMD_ERR_BLAS -4
#define PMD_ERR_UNSUPPORTED -5

#define PMD_RPAD 64          // component rows of every per-tile [comp][x] array
#define PMD_LD_SLACK 64      // extra floats at the end of every time-contiguous row

// random streams (oracle/philox.py holds the same numbers)
#define PMD_STREAM_BG_OMEGA 1
#define PMD_STREAM_SIM_NOISE 2
#define PMD_STREAM_SIM_OMEGA 3
#define PMD_STREAM_TILE_OMEGA 4
#define PMD_STREAM_PRUNE 5

struct pmd_prof_rec {
  const char* name;
  hipEvent_t start, stop;
};

struct pmd_ctx {
  int device;
  hipStream_t stream;
  rocblas_handle blas;
  float* tables;  // device: Hann window + FFT twiddles, see prep.hip
  char err[512];
  const char* atx_label;             // profiling name of the next tile_atx launches (NULL: "tile_atx")
  int atx_rows;                      // rows of A that carry data in the next tile_atx launches (0: all 64)
  const int* atx_ranks;              // per-tile ranks of the next tile_atx launches (projection: rows >= rank of A are zero), or NULL
  void* scratch;                     // library-owned device scratch of the eigensolver (sytrd.hip)
  size_t scratch_bytes;
  void* scratch2;                    // library-owned device scratch of the fp64 eigenvector refinement (sytrd.hip)
  size_t scratch2_bytes;
  int gemm_split;                    // 1 (default): large fp32 products as three fp16-piece products (gemm_f16x2.hip); 0: rocBLAS sgemm only
  double gemm_split_min_flop;        // ... for products with at least this many flops
  int gemm_split_min_dim;            // ... and no dimension below this
  void* f16x2;                       // state of gemm_f16x2.hip (hipBLASLt handle, plans), created on first use
  void* split_ws;                    // library-owned device scratch of the split products (fp16 pieces, split-K partial sums)
  size_t split_ws_bytes;
  void* comm;                        // RCCL communicator (pmd_comm_init), NULL without one
  int comm_rank, comm_world;
  float null_cutoff;                 // < 0: keep every direction with lambda != 0, scaled by 1/sqrt(|lambda|) (decomposition.py:984-996);
                                     // >= 0: keep lambda > null_cutoff * lambda_max only (pmd_ctx_set_null_cutoff)
  bool profile;                      // pmd_profile_enable: HIP events around every kernel group
  std::vector<pmd_prof_rec> recs;
};

// RAII: records a start/stop event pair on the context's stream around one launcher call
struct pmd_prof_scope {
  pmd_ctx* ctx;
  pmd_prof_rec rec;
  bool on;
  pmd_prof_scope(pmd_ctx* c, const char* name) : ctx(c), on(c && c->profile) {
    if (!on) return;
    rec.name = name;
    if (hipEventCreate(&rec.start) != hipSuccess || hipEventCreate(&rec.stop) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(rec.start, ctx->stream);
  }
  ~pmd_prof_scope() {
    if (!on) return;
    (void)hipEventRecord(rec.stop, ctx->stream);
    ctx->recs.push_back(rec);
  }
};

static inline int pmd_fail(pmd_ctx* ctx, int code, const char* what, const char* detail) {
  if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, detail ? detail : "");
  return code;
}

#define PMD_HIP(ctx, call)                                                        \
  do {                                                                            \
    hipError_t e__ = (call);                                                      \
    if (e__ != hipSuccess) return pmd_fail(ctx, PMD_ERR_HIP, #call, hipGetErrorString(e__)); \
  } while (0)

#define PMD_LAUNCH_CHECK(ctx, name)                                               \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) return pmd_fail(ctx, PMD_ERR_HIP, name, hipGetErrorString(e__)); \
  } while (0)

static inline long pmd_round_up(long x, long m) { return (x + m - 1) / m * m; }

// XCD-aware tile order.  Workgroups go to the 8 XCDs round-robin by linear workgroup id, and each XCD has its
// own L2.  Tiles overlap their neighbours by 50 %, so instead of dealing tiles 0,1,2,... across the XCDs, every
// XCD gets a contiguous run of the tile list and walks it in order: neighbouring tiles then run on the same XCD
// at about the same time and the second reader of a shared pixel row hits that XCD's L2.
// For a grid whose x dimension counts tiles: returns the tile of this workgroup (a bijection on [0, gridDim.x)).
#ifdef __HIPCC__
__device__ __forceinline__ int pmd_xcd_tile() {
  const int n = (int)gridDim.x, x = (int)blockIdx.x;
  if (n < 64) return x;
  const int row = (int)(blockIdx.y + gridDim.y * blockIdx.z);
  const int s = (int)(((long)n * row) & 7);      // XCD of the first workgroup of this grid row
  const int c = (x + s) & 7;                     // XCD of this workgroup
  const int x0 = (c - s) & 7;                    // first x of the row that lands on XCD c
  const int m = (x - x0) >> 3;                   // how many earlier workgroups of the row that XCD got
  // tiles before XCD c's run: XCD c' owns the x values x0' = (c' - s) & 7, x0' + 8, ... below n
  int off = 0;
  for (int cp = 0; cp < c; ++cp) {
    const int x0p = (cp - s) & 7;
    off += (n - x0p + 7) >> 3;
  }
  return off + m;
}
#endif

// carve a caller-provided workspace
struct pmd_arena {
  char* base;
  size_t size;
  size_t used;
  bool overflow;
  pmd_arena(void* p, size_t n) : base((char*)p), size(n), used(0), overflow(false) {}
  void* take(size_t bytes) {
    size_t off = (used + 255) & ~size_t(255);
    used = off + bytes;
    if (base == nullptr || used > size) {
      overflow = true;
      return nullptr;
    }
    return base + off;
  }
  template <typename T>
  T* take_n(size_t n) { return (T*)take(n * sizeof(T)); }
};

// padded pixel count of a tile and the tile_atx kernel variant that serves it
struct pmd_dvariant {
  int kjw;   // 16-row groups per wave
  int ks;    // K split (waves = 4*ks)
  int tc;    // frames per chunk
  int dpad;  // = 16*kjw*ks
};
static inline bool pmd_pick_dvariant(int d, pmd_dvariant* v) {
  static const pmd_dvariant table[] = {
      {16, 1, 32, 256}, {25, 1, 32, 400}, {32, 1, 32, 512},
      {25, 2, 32, 800}, {32, 2, 32, 1024}, {25, 4, 16, 1600}};
  for (const auto& t : table)
    if (d <= t.dpad) { *v = t; return true; }
  return false;
}
