// Host-side sequencing of the per-tile decomposition, the threshold simulation and the
// background rSVD.  Everything is enqueued on ctx->stream inside a caller-provided workspace;
// nothing here allocates or synchronises.
#include "pmd_internal.h"
#include <algorithm>

int pmd_tile_dpad(int d) {
  if (d > 65536) return -1;
  if (d > 1024) return (int)pmd_round_up(d, 1024);   // tile_atx splits the pixel axis over the grid in slices of 1024
  pmd_dvariant v;
  if (!pmd_pick_dvariant(d, &v)) return -1;
  return v.dpad;
}

long pmd_time_ld(long t) { return pmd_round_up(t, 64) + PMD_LD_SLACK; }

#define RUN(call)                 \
  do {                            \
    int rc__ = (call);            \
    if (rc__ != PMD_OK) return rc__; \
  } while (0)

// Orthonormal basis of the sketch's column space (jnp.linalg.qr at decomposition.py:64 / pmd_loader.py:58).  Householder QR
// in LDS while the P x l matrix fits a workgroup's 160 KB (every default configuration); beyond that - large tiles with
// spatial_avg_factor = 1 and a wide sketch, e.g. 30 x 40 pixels x 58 columns - CholeskyQR2 with the fp64 Gram / Cholesky
// kernels of the whitening steps: the same Q up to the signs of its columns (QR is unique up to them), and everything
// downstream (B = Q^T A, the SVD of B, U = Q W) is invariant under those signs.
static int tile_sketch_basis(pmd_ctx* ctx, const float* Yt, long stride, int ld, int P, int l, float* Qt, double* gpart, double* nmat,
                             int n) {
  if (pmd_small_qr_fits(P, l)) return pmd_launch_small_qr(ctx, Yt, stride, ld, P, l, Qt, stride, ld, n);
  const int nref = P < l ? P : l;
  const float* src = Yt;
  for (int pass = 0; pass < 2; ++pass) {
    RUN(pmd_launch_tile_gram(ctx, src, stride, ld, P, n, 1, gpart));
    RUN(pmd_launch_small_chol(ctx, gpart, 1, nref, 1e-12, nmat, n));
    RUN(pmd_launch_tile_rowmix(ctx, src, stride, ld, nmat, 4096, nref, nref, Qt, stride, ld, P, n));
    src = Qt;
  }
  return PMD_OK;
}

// ------------------------------------------------------------------------------------------
// per-tile decomposition (decomposition.py:235-330 single_block_md, one window, + :501-523)
// ------------------------------------------------------------------------------------------
struct tiles_plan {
  int nb, l, dpad, Ppad, nref, rp;
  long ld_b;
  float *abar, *omT, *yt, *qt, *bm, *udst, *ut0, *outA, *spart, *sst, *xbar, *g1f;
  double *gpart, *nmat, *lam;
  void* eig_ws;        // wide path only: workspace of pmd_launch_wide_eig
  size_t eig_ws_bytes;
  size_t zero_bytes;  // leading part of the workspace that must be zeroed
};

static const int GRAM_SLICES = 4;
static const int XBT_SLICES = 4;

// rp = component rows of every per-tile array: 64 on the main path (max_components + 10 <= 64), pmd_tile_rpad(r) beyond
static int plan_tiles(pmd_arena& ar, tiles_plan& p, int n, int d, int P, int r, int a, int t_crop, long ldv, long n_rows) {
  p.nb = t_crop / a;
  p.l = r + 10;
  p.rp = pmd_tile_rpad(r);
  const size_t rp = p.rp, rp2 = rp * rp;
  const bool wide = p.rp > 64;
  p.dpad = pmd_tile_dpad(d);
  p.Ppad = pmd_tile_dpad(P);
  p.nref = (P < p.l) ? P : p.l;
  p.ld_b = pmd_time_ld(p.nb);
  if (p.dpad < 0 || p.Ppad < 0) return PMD_ERR_UNSUPPORTED;
  // arrays with padding that is read before it is written come first (they get zeroed)
  p.abar = ar.take_n<float>((size_t)n * P * p.ld_b);
  p.omT = ar.take_n<float>((size_t)n * rp * p.ld_b);
  p.yt = ar.take_n<float>((size_t)n * rp * p.Ppad);
  p.qt = ar.take_n<float>((size_t)n * rp * p.Ppad);
  p.udst = ar.take_n<float>((size_t)n * rp * p.Ppad);
  p.ut0 = ar.take_n<float>((size_t)n * rp * p.dpad);
  p.spart = wide ? nullptr : ar.take_n<float>((size_t)n * XBT_SLICES * 64 * p.dpad);   // (the wide path runs S = X V^T in one slice)
  p.sst = ar.take_n<float>((size_t)n * rp * p.dpad);
  p.zero_bytes = ar.used;
  p.bm = ar.take_n<float>((size_t)n * rp * p.ld_b);
  p.xbar = ar.take_n<float>((size_t)n_rows * p.ld_b);
  p.g1f = wide ? nullptr : ar.take_n<float>((size_t)n * GRAM_SLICES * 4096);
  p.outA = ar.take_n<float>((size_t)n * rp * ldv);
  p.gpart = ar.take_n<double>((size_t)n * GRAM_SLICES * rp2);
  p.nmat = ar.take_n<double>((size_t)n * rp2);
  p.lam = ar.take_n<double>((size_t)n * rp);
  p.eig_ws = nullptr;
  p.eig_ws_bytes = 0;
  if (wide) {
    p.eig_ws_bytes = pmd_wide_eig_workspace_bytes(p.rp, n);
    p.eig_ws = ar.take(p.eig_ws_bytes);
  }
  return PMD_OK;
}

size_t pmd_tiles_workspace_bytes_impl(int n, int d, int P, int r, int a, int t_crop, long ldv, long n_rows) {
  pmd_arena ar((void*)0x1000, ~size_t(0) >> 1);
  tiles_plan p;
  if (plan_tiles(ar, p, n, d, P, r, a, t_crop, ldv, n_rows) != PMD_OK) return 0;
  return ar.used + 4096;
}

// scratch of the generic-width small-matrix steps
struct wide_ws {
  int rp;
  double *gpart, *nmat, *lam;
  void* eig_ws;
  size_t eig_ws_bytes;
};

// Orthonormal basis of the row space of a [comp][x] array by two rounds of Gram matrix -> eigen-whitening (E Lambda^-1/2,
// null directions zeroed) -> row mixing: the generic-width counterpart of CholeskyQR2 / small_qr.  Any orthonormal basis
// of the sketch's span serves the rSVD (B = Q^T A, U = Q u are invariant under Q -> Q O).
static int wide_orthonormalise(pmd_ctx* ctx, const wide_ws& p, const float* src, long stride, long ld, int len, int n_in, int n_out,
                               float* dst, int n) {
  for (int pass = 0; pass < 2; ++pass) {
    RUN(pmd_launch_wide_gram(ctx, src, stride, ld, len, n, 1, p.rp, p.gpart));
    RUN(pmd_launch_wide_eig(ctx, p.gpart, 1, p.rp, pass == 0 ? n_in : n_out, 1, 1e-12, p.nmat, p.lam, n, p.eig_ws, p.eig_ws_bytes));
    RUN(pmd_launch_wide_rowmix(ctx, src, stride, ld, p.nmat, (long)p.rp * p.rp, p.rp, pass == 0 ? n_in : n_out, n_out, dst, stride, ld, len, n));
    src = dst;
  }
  return PMD_OK;
}

// single_block_md (decomposition.py:235-330) with more than 64 component rows: the same sequence as the main path below,
// with the contractions in row blocks of 64 and the small dense algebra through wide.hip (eigen-whitening wherever the
// main path uses Householder QR or Cholesky whitening: only spans matter there, see the comments of the main path).
static int tiles_decompose_wide(pmd_ctx* ctx, const tiles_plan& p, const float* Xf, long ldx, long n_rows, int t_crop, const int* tile_pix,
                                int n, int b1, int b2, const int* pool_q, int pool_max, int P, const int* pool_idx, const float* pool_w,
                                int r, int a, float thr_s, float thr_t, int max_fail, uint64_t seed, uint32_t omega_index0,
                                uint32_t omega_index_step, float* Ut_out, float* V_out, long ldv, float* stats_out, int* good_out,
                                int* keep_out, int* ranks_out, double* sing_out, void* ws, int stages) {
  const int d = b1 * b2, rp = p.rp;
  const long srd = (long)rp * p.dpad, srP = (long)rp * p.Ppad, srb = (long)rp * p.ld_b, srv = (long)rp * ldv, rp2 = (long)rp * rp;
  if (stages & 1) {
    PMD_HIP(ctx, hipMemsetAsync(ws, 0, p.zero_bytes, ctx->stream));
    PMD_HIP(ctx, hipMemsetAsync(Ut_out, 0, (size_t)n * srd * sizeof(float), ctx->stream));
    RUN(pmd_launch_tile_pool_bin(ctx, Xf, ldx, n_rows, tile_pix, n, d, pool_q, pool_max, P, a, p.nb, p.xbar, p.abar, p.ld_b, (long)P * p.ld_b));
    for (int t0 = 0; t0 < n; t0 += 32768) {
      const int tn = (n - t0 < 32768) ? n - t0 : 32768;
      RUN(pmd_launch_rng(ctx, seed, PMD_STREAM_TILE_OMEGA, omega_index0 + (uint32_t)t0 * omega_index_step, omega_index_step, tn, p.nb,
                         p.l, 1, p.omT + (long)t0 * srb, p.ld_b, srb));
    }
    RUN(pmd_launch_tile_xbt_rp(ctx, p.abar, p.ld_b, nullptr, 0, P, P, p.omT, srb, p.ld_b, p.yt, srP, 0, p.Ppad, n, p.nb, 1, p.l));
    const wide_ws wws = {p.rp, p.gpart, p.nmat, p.lam, p.eig_ws, p.eig_ws_bytes};
    RUN(wide_orthonormalise(ctx, wws, p.yt, srP, p.Ppad, P, p.l, p.nref, p.qt, n));
    RUN(pmd_launch_tile_atx_rp(ctx, p.abar, p.ld_b, nullptr, 0, P, P, p.qt, srP, p.Ppad, p.bm, srb, p.ld_b, n, p.nb, 1, p.nref));
    RUN(pmd_launch_wide_gram(ctx, p.bm, srb, p.ld_b, p.nb, n, 1, rp, p.gpart));
    RUN(pmd_launch_wide_eig(ctx, p.gpart, 1, rp, p.nref, 0, 0.0, p.nmat, p.lam, n, p.eig_ws, p.eig_ws_bytes));
    RUN(pmd_launch_wide_rowmix(ctx, p.qt, srP, p.Ppad, p.nmat, rp2, rp, p.nref, r, p.udst, srP, p.Ppad, P, n));
    RUN(pmd_launch_expand_pooled(ctx, p.udst, srP, p.Ppad, pool_idx, pool_w, d, r, p.ut0, srd, p.dpad, n));
    ctx->atx_label = "tile_atx_main";
    RUN(pmd_launch_tile_atx_rp(ctx, Xf, ldx, tile_pix, d, 0, d, p.ut0, srd, p.dpad, p.outA, srv, ldv, n, t_crop, 2, r));
    ctx->atx_label = nullptr;
  }
  if (stages & 2) {
    RUN(pmd_launch_wide_gram(ctx, p.outA, srv, ldv, t_crop, n, GRAM_SLICES, rp, p.gpart));
    RUN(pmd_launch_wide_eig(ctx, p.gpart, GRAM_SLICES, rp, r, 1, 1e-10, p.nmat, p.lam, n, p.eig_ws, p.eig_ws_bytes));
    RUN(pmd_launch_tile_xbt_rp(ctx, Xf, ldx, tile_pix, d, 0, d, p.outA, srv, ldv, p.sst, srd, 0, p.dpad, n, t_crop, 1, r));
    RUN(pmd_launch_wide_rowmix(ctx, p.sst, srd, p.dpad, p.nmat, rp2, rp, r, r, p.sst, srd, p.dpad, d, n));
  }
  if (stages & 4) {
    RUN(pmd_launch_wide_gram(ctx, p.sst, srd, p.dpad, d, n, 1, rp, p.gpart));
    RUN(pmd_launch_wide_eig(ctx, p.gpart, 1, rp, r, 1, 1e-10, p.nmat, p.lam, n, p.eig_ws, p.eig_ws_bytes));
    RUN(pmd_launch_wide_rowmix(ctx, p.sst, srd, p.dpad, p.nmat, rp2, rp, r, r, p.sst, srd, p.dpad, d, n));
    ctx->atx_label = "tile_atx_main";
    RUN(pmd_launch_tile_atx_rp(ctx, Xf, ldx, tile_pix, d, 0, d, p.sst, srd, p.dpad, V_out, srv, ldv, n, t_crop, 2, r));
    ctx->atx_label = nullptr;
    RUN(pmd_launch_wide_gram(ctx, V_out, srv, ldv, t_crop, n, GRAM_SLICES, rp, p.gpart));
    RUN(pmd_launch_wide_eig(ctx, p.gpart, GRAM_SLICES, rp, r, 0, 0.0, p.nmat, sing_out ? sing_out : p.lam, n, p.eig_ws, p.eig_ws_bytes));
    RUN(pmd_launch_wide_rowmix(ctx, p.sst, srd, p.dpad, p.nmat, rp2, rp, r, r, Ut_out, srd, p.dpad, d, n));
    RUN(pmd_launch_wide_rowmix(ctx, V_out, srv, ldv, p.nmat, rp2, rp, r, r, V_out, srv, ldv, t_crop, n));
    RUN(pmd_launch_stats_roughness(ctx, Ut_out, srd, p.dpad, b1, b2, V_out, srv, ldv, t_crop, r, stats_out, n, rp));
    RUN(pmd_launch_decide(ctx, stats_out, r, thr_s, thr_t, max_fail, r, n, good_out, keep_out, ranks_out, rp));
  }
  return PMD_OK;
}

int pmd_tiles_decompose_impl(pmd_ctx* ctx, const float* Xf, long ldx, long n_rows, int t_crop, const int* tile_pix, int n, int b1,
                             int b2, const int* pool_q, int pool_max, int P, const int* pool_idx, const float* pool_w,
                             int r, int a, float thr_s, float thr_t, int max_fail, uint64_t seed, uint32_t omega_index0,
                             uint32_t omega_index_step, float* Ut_out, float* V_out, long ldv, float* stats_out,
                             int* good_out, int* keep_out, int* ranks_out, double* sing_out, void* ws, size_t ws_bytes,
                             int stages) {
  const int d = b1 * b2;
  if (r < 1) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_tiles_decompose", "max_components must be >= 1");
  if (a < 1 || t_crop % a != 0 || t_crop / a < 1) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_tiles_decompose", "t_crop must be a positive multiple of temporal_avg_factor");
  if (ldv < pmd_time_ld(t_crop) || ldx < pmd_time_ld(t_crop)) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_tiles_decompose", "leading dimension too small");
  pmd_arena ar(ws, ws_bytes);
  tiles_plan p;
  if (plan_tiles(ar, p, n, d, P, r, a, t_crop, ldv, n_rows) != PMD_OK)
    return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "pmd_tiles_decompose", "tile too large (max 65536 pixels)");
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_tiles_decompose", "workspace too small");
  // max_components beyond the number of time bins or of pooled pixels: the reference's rSVD (decomposition.py:59-73) keeps
  // its sketch of max_components + 10 columns and `u_final[:, :rank]` simply returns the min(rank, bins, pixels) columns
  // that exist; everything downstream then works on that many components
  if (r > p.nb) r = p.nb;
  if (r > p.nref) r = p.nref;
  if (p.rp > 64)
    return tiles_decompose_wide(ctx, p, Xf, ldx, n_rows, t_crop, tile_pix, n, b1, b2, pool_q, pool_max, P, pool_idx, pool_w, r, a, thr_s,
                                thr_t, max_fail, seed, omega_index0, omega_index_step, Ut_out, V_out, ldv, stats_out, good_out, keep_out,
                                ranks_out, sing_out, ws, stages);
  const long s64d = 64L * p.dpad, s64P = 64L * p.Ppad, s64b = 64L * p.ld_b, s64v = 64L * ldv;
  // PMD_TILE_WHITEN=eig restores the eigenvector form of the two pure orthonormalisation steps (A/B runs)
  static int whiten_mode = -1;
  if (whiten_mode < 0) { const char* e = getenv("PMD_TILE_WHITEN"); whiten_mode = (e && !strcmp(e, "eig")) ? 0 : 1; }
  const bool whiten_chol = whiten_mode && stages == 7;   // the spatial_denoiser hook sees S = X V_b^T column by column
  const bool whiten_chol_u0 = whiten_mode != 0;
  // Time slices of the two contractions over all frames: four per tile give a few thousand tiles enough workgroups to fill
  // the chip; with many tiles (the 16 x 16-pixel regime: 16 129 / 65 025 tiles) one slice does, and the partial results
  // and their reduction pass (10 of 300 ms at 1024 x 1024 x 1000, b = 16) disappear.
  const int xs = n >= 4096 ? 1 : XBT_SLICES;
  const int gs = n >= 4096 ? 1 : GRAM_SLICES;

  // stages: bit 0 = up to V_ds (p.outA; the temporal_denoiser hook of decomposition.py:300 acts on it),
  //         bit 1 = basis of its row space and S = X V_b^T (p.sst; spatial_denoiser hook, :310), bit 2 = the rest
  if (stages & 1) {
  PMD_HIP(ctx, hipMemsetAsync(ws, 0, p.zero_bytes, ctx->stream));
  PMD_HIP(ctx, hipMemsetAsync(Ut_out, 0, (size_t)n * s64d * sizeof(float), ctx->stream));

  // --- rSVD of the pooled, temporally binned tile (decomposition.py:279-294, :59-73)
  RUN(pmd_launch_tile_pool_bin(ctx, Xf, ldx, n_rows, tile_pix, n, d, pool_q, pool_max, P, a, p.nb, p.xbar, p.abar, p.ld_b, (long)P * p.ld_b));
  for (int t0 = 0; t0 < n; t0 += 32768) {
    const int tn = (n - t0 < 32768) ? n - t0 : 32768;
    RUN(pmd_launch_rng(ctx, seed, PMD_STREAM_TILE_OMEGA, omega_index0 + (uint32_t)t0 * omega_index_step,
                       omega_index_step, tn, p.nb, p.l, 1, p.omT + (long)t0 * s64b, p.ld_b, s64b));
  }
  RUN(pmd_launch_tile_xbt(ctx, p.abar, p.ld_b, nullptr, 0, P, P, p.omT, s64b, p.ld_b, p.yt, s64P, 0, p.Ppad, n, p.nb, 1));
  RUN(tile_sketch_basis(ctx, p.yt, s64P, p.Ppad, P, p.l, p.qt, p.gpart, p.nmat, n));
  RUN(pmd_launch_tile_atx(ctx, p.abar, p.ld_b, nullptr, 0, P, P, p.qt, s64P, p.Ppad, p.bm, s64b, p.ld_b, n, p.nb, 1));
  RUN(pmd_launch_tile_gram(ctx, p.bm, s64b, p.ld_b, p.nb, n, 1, p.gpart));
  RUN(pmd_launch_small_eig(ctx, p.gpart, 1, p.nref, 0, 0.0, p.nmat, p.lam, n));
  RUN(pmd_launch_tile_rowmix(ctx, p.qt, s64P, p.Ppad, p.nmat, 4096, p.nref, r, p.udst, s64P, p.Ppad, P, n));
  RUN(pmd_launch_expand_pooled(ctx, p.udst, s64P, p.Ppad, pool_idx, pool_w, d, r, p.ut0, s64d, p.dpad, n));

  // --- V_ds = U_ds^T X_ds; basis of its row space (decomposition.py:295-301)
  ctx->atx_label = "tile_atx_main";
  RUN(pmd_launch_tile_atx(ctx, Xf, ldx, tile_pix, d, 0, d, p.ut0, s64d, p.dpad, p.outA, s64v, ldv, n, t_crop, 2));
  ctx->atx_label = nullptr;
  }
  if (stages & 2) {
  // (this Gram only conditions the basis change -- span(S) does not depend on it -- so fp32 MFMA is enough)
  RUN(pmd_launch_tile_xbt(ctx, p.outA, ldv, nullptr, 0, 64, 64, p.outA, s64v, ldv, p.g1f, gs * 4096L, 4096, 64, n, t_crop, gs));
  RUN(pmd_launch_gram_f2d(ctx, p.g1f, 64, (long)n * gs, p.gpart));
  // only span(S) matters downstream: with no denoiser hook reading S component by component (stages == 7) any orthonormal
  // basis of the row space of V_ds serves, and the Cholesky whitening replaces the Jacobi eigensolver (small_la.hip)
  if (whiten_chol) RUN(pmd_launch_small_chol(ctx, p.gpart, gs, r, 1e-10, p.nmat, n));
  else RUN(pmd_launch_small_eig(ctx, p.gpart, gs, r, 1, 1e-10, p.nmat, p.lam, n));

  // --- S = X V_b^T and its left singular vectors U0 (decomposition.py:304-317)
  if (xs == 1) {
    RUN(pmd_launch_tile_xbt(ctx, Xf, ldx, tile_pix, d, 0, d, p.outA, s64v, ldv, p.sst, s64d, 0, p.dpad, n, t_crop, 1));
  } else {
    RUN(pmd_launch_tile_xbt(ctx, Xf, ldx, tile_pix, d, 0, d, p.outA, s64v, ldv, p.spart, XBT_SLICES * s64d, s64d, p.dpad, n, t_crop, XBT_SLICES));
    RUN(pmd_launch_reduce_slices(ctx, p.spart, XBT_SLICES * s64d, s64d, XBT_SLICES, s64d, p.sst, s64d, n));
  }
  RUN(pmd_launch_tile_rowmix(ctx, p.sst, s64d, p.dpad, p.nmat, 4096, r, r, p.sst, s64d, p.dpad, d, n));
  }
  if (stages & 4) {
  RUN(pmd_launch_tile_gram(ctx, p.sst, s64d, p.dpad, d, n, 1, p.gpart));
  // U = U0 Wl depends on span(U0) = span(S) only: an orthonormal basis of it is enough (always; no hook reads U0)
  if (whiten_chol_u0) RUN(pmd_launch_small_chol(ctx, p.gpart, 1, r, 1e-10, p.nmat, n));
  else RUN(pmd_launch_small_eig(ctx, p.gpart, 1, r, 1, 1e-10, p.nmat, p.lam, n));
  RUN(pmd_launch_tile_rowmix(ctx, p.sst, s64d, p.dpad, p.nmat, 4096, r, r, p.sst, s64d, p.dpad, d, n));

  // --- W = U0^T X, its SVD rotates U0 and gives sigma*V (decomposition.py:318-323)
  ctx->atx_label = "tile_atx_main";
  RUN(pmd_launch_tile_atx(ctx, Xf, ldx, tile_pix, d, 0, d, p.sst, s64d, p.dpad, V_out, s64v, ldv, n, t_crop, 2));
  ctx->atx_label = nullptr;
  RUN(pmd_launch_tile_gram(ctx, V_out, s64v, ldv, t_crop, n, gs, p.gpart));
  RUN(pmd_launch_small_eig(ctx, p.gpart, gs, r, 0, 0.0, p.nmat, sing_out ? sing_out : p.lam, n));
  RUN(pmd_launch_tile_rowmix(ctx, p.sst, s64d, p.dpad, p.nmat, 4096, r, r, Ut_out, s64d, p.dpad, d, n));
  RUN(pmd_launch_tile_rowmix(ctx, V_out, s64v, ldv, p.nmat, 4096, r, r, V_out, s64v, ldv, t_crop, n));

  // --- roughness statistics and keep/discard scan (evaluation.py:84-222)
  RUN(pmd_launch_stats_roughness(ctx, Ut_out, s64d, p.dpad, b1, b2, V_out, s64v, ldv, t_crop, r, stats_out, n));
  RUN(pmd_launch_decide(ctx, stats_out, r, thr_s, thr_t, max_fail, r, n, good_out, keep_out, ranks_out));
  }
  return PMD_OK;
}

int pmd_tiles_hook_offsets_impl(int n, int d, int P, int r, int a, int t_crop, long ldv, long n_rows, size_t* vds_off,
                                size_t* s_off) {
  pmd_arena ar((void*)0x1000, ~size_t(0) >> 1);
  tiles_plan p;
  if (plan_tiles(ar, p, n, d, P, r, a, t_crop, ldv, n_rows) != PMD_OK) return PMD_ERR_UNSUPPORTED;
  *vds_off = (size_t)((char*)p.outA - (char*)0x1000);
  *s_off = (size_t)((char*)p.sst - (char*)0x1000);
  return PMD_OK;
}

// ------------------------------------------------------------------------------------------
// residual window (decomposition.py:333-387 single_residual_block_md + :501-515): components of the
// part of the window that the current basis E does not explain.  (I - E E^T) commutes with the
// temporal bin average, so the residual is never materialised at full resolution.
// ------------------------------------------------------------------------------------------
struct resid_plan {
  int nb, l, dpad, nref, rp;
  long ld_b, ld_L;
  float *xbar, *wbar, *ar, *omT, *yt, *qt, *bm, *unew, *tmp, *util, *vmat;
  double *gpart, *nmat, *lam;
  void* eig_ws;
  size_t eig_ws_bytes;
  size_t zero_bytes;
};

static int plan_resid(pmd_arena& ar, resid_plan& p, int n, int d, int r, int a, int L, long n_rows) {
  p.nb = L / a;
  p.l = r + 10;
  p.rp = pmd_tile_rpad(r);
  const size_t rp = p.rp, rp2 = rp * rp;
  p.dpad = pmd_tile_dpad(d);
  if (p.dpad < 0) return PMD_ERR_UNSUPPORTED;
  p.nref = d < p.l ? d : p.l;
  p.ld_b = pmd_time_ld(p.nb);
  p.ld_L = pmd_time_ld(L);
  p.omT = ar.take_n<float>((size_t)n * rp * p.ld_b);
  p.ar = ar.take_n<float>((size_t)n * d * p.ld_b);
  p.yt = ar.take_n<float>((size_t)n * rp * p.dpad);
  p.qt = ar.take_n<float>((size_t)n * rp * p.dpad);
  p.unew = ar.take_n<float>((size_t)n * rp * p.dpad);
  p.tmp = ar.take_n<float>((size_t)n * rp * p.dpad);
  p.util = ar.take_n<float>((size_t)n * rp * p.dpad);
  p.zero_bytes = ar.used;
  p.xbar = ar.take_n<float>((size_t)n_rows * p.ld_b);
  p.wbar = ar.take_n<float>((size_t)n * rp * p.ld_b);
  p.bm = ar.take_n<float>((size_t)n * rp * p.ld_b);
  p.vmat = ar.take_n<float>((size_t)n * rp * p.ld_L);
  p.gpart = ar.take_n<double>((size_t)n * GRAM_SLICES * rp2);
  p.nmat = ar.take_n<double>((size_t)n * rp2);
  p.lam = ar.take_n<double>((size_t)n * rp);
  p.eig_ws = nullptr;
  p.eig_ws_bytes = 0;
  if (p.rp > 64) {
    p.eig_ws_bytes = pmd_wide_eig_workspace_bytes(p.rp, n);
    p.eig_ws = ar.take(p.eig_ws_bytes);
  }
  return PMD_OK;
}

size_t pmd_tiles_residual_workspace_bytes_impl(int n, int d, int r, int a, int L, long n_rows) {
  pmd_arena ar((void*)0x1000, ~size_t(0) >> 1);
  resid_plan p;
  if (plan_resid(ar, p, n, d, r, a, L, n_rows) != PMD_OK) return 0;
  return ar.used + 4096;
}

int pmd_tiles_residual_impl(pmd_ctx* ctx, const float* Xw, long ldx, long n_rows, int L, const int* tile_pix, int n,
                            int b1, int b2, int r, int a, float thr_s, float thr_t, int max_fail, uint64_t seed,
                            uint32_t omega_index0, uint32_t omega_index_step, float* Ucur, int* counts, float* stats_out,
                            int* good_out, int* keep_out, void* ws, size_t ws_bytes) {
  const int d = b1 * b2;
  if (r < 1) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_tiles_residual", "max_components must be >= 1");
  if (a < 1 || L % a != 0 || L / a < 1) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_tiles_residual", "window length must be a positive multiple of temporal_avg_factor");
  if (ldx < pmd_time_ld(L)) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_tiles_residual", "leading dimension too small");
  pmd_arena ar(ws, ws_bytes);
  resid_plan p;
  if (plan_resid(ar, p, n, d, r, a, L, n_rows) != PMD_OK) return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "pmd_tiles_residual", "tile too large");
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_tiles_residual", "workspace too small");
  if (p.rp > 64) {
    // generic-width form (max_components + 10 > 64): the same sequence through wide.hip and row blocks of 64
    const int rp = p.rp;
    const long srd = (long)rp * p.dpad, srb = (long)rp * p.ld_b, srL = (long)rp * p.ld_L, rp2 = (long)rp * rp;
    const wide_ws wws = {rp, p.gpart, p.nmat, p.lam, p.eig_ws, p.eig_ws_bytes};
    PMD_HIP(ctx, hipMemsetAsync(ws, 0, p.zero_bytes, ctx->stream));
    RUN(pmd_launch_bin_average(ctx, Xw, ldx, n_rows, a, p.nb, p.xbar, p.ld_b));
    RUN(pmd_launch_tile_atx_rp(ctx, p.xbar, p.ld_b, tile_pix, d, 0, d, Ucur, srd, p.dpad, p.wbar, srb, p.ld_b, n, p.nb, 1, r));
    RUN(pmd_launch_tile_residual_rows(ctx, p.xbar, p.ld_b, tile_pix, d, Ucur, p.dpad, p.wbar, p.ld_b, r, p.nb, p.ar, p.ld_b, n, rp));
    for (int t0 = 0; t0 < n; t0 += 32768) {
      const int tn = (n - t0 < 32768) ? n - t0 : 32768;
      RUN(pmd_launch_rng(ctx, seed, PMD_STREAM_TILE_OMEGA, omega_index0 + (uint32_t)t0 * omega_index_step, omega_index_step, tn, p.nb, p.l, 1,
                         p.omT + (long)t0 * srb, p.ld_b, srb));
    }
    RUN(pmd_launch_tile_xbt_rp(ctx, p.ar, p.ld_b, nullptr, 0, d, d, p.omT, srb, p.ld_b, p.yt, srd, 0, p.dpad, n, p.nb, 1, p.l));
    RUN(wide_orthonormalise(ctx, wws, p.yt, srd, p.dpad, d, p.l, p.nref, p.qt, n));
    RUN(pmd_launch_tile_atx_rp(ctx, p.ar, p.ld_b, nullptr, 0, d, d, p.qt, srd, p.dpad, p.bm, srb, p.ld_b, n, p.nb, 1, p.nref));
    RUN(pmd_launch_wide_gram(ctx, p.bm, srb, p.ld_b, p.nb, n, 1, rp, p.gpart));
    RUN(pmd_launch_wide_eig(ctx, p.gpart, 1, rp, p.nref, 0, 0.0, p.nmat, p.lam, n, p.eig_ws, p.eig_ws_bytes));
    const int rnw = std::min(r, std::min(p.nb, p.nref));
    RUN(pmd_launch_wide_rowmix(ctx, p.qt, srd, p.dpad, p.nmat, rp2, rp, p.nref, rnw, p.unew, srd, p.dpad, d, n));
    // utilde = u - E (E^T u): the cross Gram matrix E^T u is the mixing matrix of the row mix of E
    RUN(pmd_launch_wide_gram(ctx, Ucur, srd, p.dpad, d, n, 1, rp, p.nmat, p.unew));
    RUN(pmd_launch_wide_rowmix(ctx, Ucur, srd, p.dpad, p.nmat, rp2, rp, r, rnw, p.tmp, srd, p.dpad, d, n));
    PMD_HIP(ctx, hipMemcpyAsync(p.util, p.unew, (size_t)n * srd * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    RUN(pmd_launch_tile_sub(ctx, p.util, p.tmp, srd, p.dpad, d, n, rp));
    RUN(pmd_launch_tile_atx_rp(ctx, Xw, ldx, tile_pix, d, 0, d, p.util, srd, p.dpad, p.vmat, srL, p.ld_L, n, L, 2, rnw));
    RUN(pmd_launch_stats_roughness(ctx, p.unew, srd, p.dpad, b1, b2, p.vmat, srL, p.ld_L, L, rnw, stats_out, n, rp));
    RUN(pmd_launch_tile_append(ctx, stats_out, rnw, thr_s, thr_t, max_fail, r, p.unew, Ucur, p.dpad, counts, good_out, keep_out, n, rp));
    return PMD_OK;
  }
  const long s64d = 64L * p.dpad, s64b = 64L * p.ld_b, s64L = 64L * p.ld_L;
  PMD_HIP(ctx, hipMemsetAsync(ws, 0, p.zero_bytes, ctx->stream));

  // A_r = (I - E E^T) binavg(X_window)
  RUN(pmd_launch_bin_average(ctx, Xw, ldx, n_rows, a, p.nb, p.xbar, p.ld_b));
  RUN(pmd_launch_tile_atx(ctx, p.xbar, p.ld_b, tile_pix, d, 0, d, Ucur, s64d, p.dpad, p.wbar, s64b, p.ld_b, n, p.nb, 1));
  RUN(pmd_launch_tile_residual_rows(ctx, p.xbar, p.ld_b, tile_pix, d, Ucur, p.dpad, p.wbar, p.ld_b, r, p.nb, p.ar, p.ld_b, n));
  // rSVD of A_r (decomposition.py:378, :59-73)
  for (int t0 = 0; t0 < n; t0 += 32768) {
    const int tn = (n - t0 < 32768) ? n - t0 : 32768;
    RUN(pmd_launch_rng(ctx, seed, PMD_STREAM_TILE_OMEGA, omega_index0 + (uint32_t)t0 * omega_index_step, omega_index_step, tn,
                       p.nb, p.l, 1, p.omT + (long)t0 * s64b, p.ld_b, s64b));
  }
  RUN(pmd_launch_tile_xbt(ctx, p.ar, p.ld_b, nullptr, 0, d, d, p.omT, s64b, p.ld_b, p.yt, s64d, 0, p.dpad, n, p.nb, 1));
  RUN(tile_sketch_basis(ctx, p.yt, s64d, p.dpad, d, p.l, p.qt, p.gpart, p.nmat, n));
  RUN(pmd_launch_tile_atx(ctx, p.ar, p.ld_b, nullptr, 0, d, d, p.qt, s64d, p.dpad, p.bm, s64b, p.ld_b, n, p.nb, 1));
  RUN(pmd_launch_tile_gram(ctx, p.bm, s64b, p.ld_b, p.nb, n, 1, p.gpart));
  RUN(pmd_launch_small_eig(ctx, p.gpart, 1, p.nref, 0, 0.0, p.nmat, p.lam, n));
  // new components of this window: min(max_components, bins, pixels) of them exist (see pmd_tiles_decompose_impl)
  const int rn = std::min(r, std::min(p.nb, p.nref));
  RUN(pmd_launch_tile_rowmix(ctx, p.qt, s64d, p.dpad, p.nmat, 4096, p.nref, rn, p.unew, s64d, p.dpad, d, n));
  // v = u^T (I - E E^T) X = utilde^T X with utilde = u - E (E^T u)   (decomposition.py:370-371, :379)
  RUN(pmd_launch_tile_cross_gram(ctx, Ucur, p.unew, s64d, p.dpad, d, p.nmat, n));
  RUN(pmd_launch_tile_rowmix(ctx, Ucur, s64d, p.dpad, p.nmat, 4096, r, rn, p.tmp, s64d, p.dpad, d, n));
  PMD_HIP(ctx, hipMemcpyAsync(p.util, p.unew, (size_t)n * s64d * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  RUN(pmd_launch_tile_sub(ctx, p.util, p.tmp, s64d, p.dpad, d, n));
  RUN(pmd_launch_tile_atx(ctx, Xw, ldx, tile_pix, d, 0, d, p.util, s64d, p.dpad, p.vmat, s64L, p.ld_L, n, L, 2));
  // fitness, keep/discard scan, append behind the existing components
  RUN(pmd_launch_stats_roughness(ctx, p.unew, s64d, p.dpad, b1, b2, p.vmat, s64L, p.ld_L, L, rn, stats_out, n));
  RUN(pmd_launch_tile_append(ctx, stats_out, rn, thr_s, thr_t, max_fail, r, p.unew, Ucur, p.dpad, counts, good_out, keep_out, n));
  return PMD_OK;
}

// ------------------------------------------------------------------------------------------
// threshold simulation (decomposition.py:76-131, :147-181): rank-1 rSVD of N(0,1) tiles
// ------------------------------------------------------------------------------------------
static const int SIM_BATCH = 256;  // one batch for the reference's 250 iterations (~22 MB of workspace each)

struct sim_plan {
  int dpad, nbatch;
  long ld_t;
  float *yt, *ypart, *qt, *ut, *noise, *omT, *bm, *stats;
  double *gpart, *nmat, *lam;
  size_t zero_bytes;
};

static int plan_sim(pmd_arena& ar, sim_plan& p, int d, int t, int iters) {
  p.dpad = pmd_tile_dpad(d);
  if (p.dpad < 0) return PMD_ERR_UNSUPPORTED;
  p.nbatch = iters < SIM_BATCH ? iters : SIM_BATCH;
  p.ld_t = pmd_time_ld(t);
  const size_t nb = p.nbatch;
  p.ypart = ar.take_n<float>(nb * XBT_SLICES * 64 * p.dpad);
  p.yt = ar.take_n<float>(nb * 64 * p.dpad);
  p.qt = ar.take_n<float>(nb * 64 * p.dpad);
  p.ut = ar.take_n<float>(nb * 64 * p.dpad);
  p.omT = ar.take_n<float>(nb * 64 * p.ld_t);
  p.noise = ar.take_n<float>(nb * d * p.ld_t);
  p.zero_bytes = ar.used;
  p.bm = ar.take_n<float>(nb * 64 * p.ld_t);
  p.stats = ar.take_n<float>(nb * 64 * 2);
  p.gpart = ar.take_n<double>(nb * GRAM_SLICES * 4096);
  p.nmat = ar.take_n<double>(nb * 4096);
  p.lam = ar.take_n<double>(nb * 64);
  return PMD_OK;
}

size_t pmd_sim_workspace_bytes_impl(int d, int t, int iters) {
  pmd_arena ar((void*)0x1000, ~size_t(0) >> 1);
  sim_plan p;
  if (plan_sim(ar, p, d, t, iters) != PMD_OK) return 0;
  return ar.used + 4096;
}

__global__ void copy_sim_stats_kernel(const float* __restrict__ stats, int nb, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb) {
    out[2 * i + 0] = stats[(long)i * PMD_RPAD * 2 + 0];
    out[2 * i + 1] = stats[(long)i * PMD_RPAD * 2 + 1];
  }
}

int pmd_threshold_sim_impl(pmd_ctx* ctx, int b1, int b2, int t, int iters, uint64_t seed, float* stats_out, void* ws,
                           size_t ws_bytes) {
  const int d = b1 * b2;
  const int l = 11;  // num_comps = 1, ten oversamples (decomposition.py:59, :708)
  if (t < 3) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_threshold_sim", "need at least 3 frames");
  pmd_arena ar(ws, ws_bytes);
  sim_plan p;
  if (plan_sim(ar, p, d, t, iters) != PMD_OK) return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "pmd_threshold_sim", "tile too large");
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_threshold_sim", "workspace too small");
  const long s64d = 64L * p.dpad, s64t = 64L * p.ld_t;
  const int nref = d < l ? d : l;
  for (int it0 = 0; it0 < iters; it0 += p.nbatch) {
    const int nb = (iters - it0 < p.nbatch) ? iters - it0 : p.nbatch;
    PMD_HIP(ctx, hipMemsetAsync(ws, 0, p.zero_bytes, ctx->stream));
    RUN(pmd_launch_rng(ctx, seed, PMD_STREAM_SIM_NOISE, (uint32_t)it0, 1, nb, d, t, 0, p.noise, p.ld_t, (long)d * p.ld_t));
    RUN(pmd_launch_rng(ctx, seed, PMD_STREAM_SIM_OMEGA, (uint32_t)it0, 1, nb, t, l, 1, p.omT, p.ld_t, s64t));
    RUN(pmd_launch_tile_xbt(ctx, p.noise, p.ld_t, nullptr, 0, d, d, p.omT, s64t, p.ld_t, p.ypart, XBT_SLICES * s64d, s64d, p.dpad, nb, t, XBT_SLICES));
    RUN(pmd_launch_reduce_slices(ctx, p.ypart, XBT_SLICES * s64d, s64d, XBT_SLICES, s64d, p.yt, s64d, nb));
    RUN(tile_sketch_basis(ctx, p.yt, s64d, p.dpad, d, l, p.qt, p.gpart, p.nmat, nb));
    RUN(pmd_launch_tile_atx(ctx, p.noise, p.ld_t, nullptr, 0, d, d, p.qt, s64d, p.dpad, p.bm, s64t, p.ld_t, nb, t, 4));
    RUN(pmd_launch_tile_gram(ctx, p.bm, s64t, p.ld_t, t, nb, GRAM_SLICES, p.gpart));
    RUN(pmd_launch_small_eig(ctx, p.gpart, GRAM_SLICES, nref, 0, 0.0, p.nmat, p.lam, nb));
    RUN(pmd_launch_tile_rowmix(ctx, p.qt, s64d, p.dpad, p.nmat, 4096, nref, 1, p.ut, s64d, p.dpad, d, nb));
    RUN(pmd_launch_tile_rowmix(ctx, p.bm, s64t, p.ld_t, p.nmat, 4096, nref, 1, p.bm, s64t, p.ld_t, t, nb));
    RUN(pmd_launch_stats_roughness(ctx, p.ut, s64d, p.dpad, b1, b2, p.bm, s64t, p.ld_t, t, 1, p.stats, nb));
    hipLaunchKernelGGL(copy_sim_stats_kernel, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, p.stats, nb, stats_out + 2L * it0);
    PMD_LAUNCH_CHECK(ctx, "copy_sim_stats_kernel");
  }
  return PMD_OK;
}

// ------------------------------------------------------------------------------------------
// background basis (pmd_loader.py:46-68, :300-314): rSVD of the standardised sample, with the
// tall-skinny QR done as CholeskyQR2 in fp64 (Q differs from Householder's by column signs
// only, which cancel in U = Q u).
// ------------------------------------------------------------------------------------------
#define BG_BLK 256

__global__ void sum_gram_blocks_kernel(const double* __restrict__ g, int nblk, double* __restrict__ out, int count = 4096) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += g[(long)b * count + i];
  out[i] = s;
}

// G = R^T R (upper R); N[c'][c] = (R^{-1})[c'][c].  Pivots <= tol * max diagonal give a zero column.
__global__ __launch_bounds__(64) void chol_inverse_kernel(const double* __restrict__ G, int n, double tol,
                                                          double* __restrict__ N) {
  __shared__ double R[64][65];
  __shared__ double Ri[64][65];
  __shared__ int dead[64];
  const int t = threadIdx.x;
  for (int i = 0; i < 64; ++i) { R[i][t] = (i < n && t < n) ? 0.5 * (G[i * 64 + t] + G[t * 64 + i]) : 0.0; Ri[i][t] = 0.0; }
  __syncthreads();
  double dmax = 0.0;
  for (int i = 0; i < n; ++i) dmax = fmax(dmax, R[i][i]);
  // right-looking Cholesky, upper factor stored in R (row k = R[k][k:])
  for (int k = 0; k < n; ++k) {
    const double piv = R[k][k];
    const bool bad = !(piv > tol * dmax);
    if (t == 0) dead[k] = bad;
    __syncthreads();
    const double rkk = bad ? 1.0 : sqrt(piv);
    double rkt = 0.0;
    if (t >= k && t < n) rkt = bad ? ((t == k) ? 1.0 : 0.0) : R[k][t] / rkk;
    __syncthreads();
    if (t >= k && t < n) R[k][t] = rkt;
    __syncthreads();
    if (!bad) {
      // trailing update: R[i][j] -= R[k][i] * R[k][j], i,j > k ; thread t owns column j = t
      if (t > k && t < n)
        for (int i = k + 1; i <= t; ++i) R[i][t] -= R[k][i] * R[k][t];
    }
    __syncthreads();
  }
  // invert the upper-triangular factor: thread t solves column t of R * X = I
  if (t < n) {
    for (int i = t; i >= 0; --i) {
      double s = (i == t) ? 1.0 : 0.0;
      for (int j = i + 1; j <= t; ++j) s -= R[i][j] * Ri[j][t];
      Ri[i][t] = s / R[i][i];
    }
  }
  __syncthreads();
  for (int i = 0; i < 64; ++i) {
    double v = (i < n && t < n) ? Ri[i][t] : 0.0;
    if (t < n && dead[t]) v = 0.0;
    N[i * 64 + t] = v;
  }
}

__global__ void unblock_basis_kernel(const float* __restrict__ ubt, long D, int K, float* __restrict__ basis, int rp = 64) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  const long blk = c / BG_BLK;
  const int q = (int)(c - blk * BG_BLK);
  for (int k = 0; k < K; ++k) basis[c * K + k] = ubt[blk * rp * BG_BLK + (long)k * BG_BLK + q];
}

struct bg_plan {
  int nblk, rp;
  long ld;
  float *omT, *ypart, *yt, *qt, *bpart, *bm;
  double *gblk, *gsum, *nmat, *lam;
  void* eig_ws;
  size_t eig_ws_bytes;
  size_t zero_bytes;
};

// rp = 64 rows of every [comp][x] array while K + 10 <= 64, pmd_tile_rpad(K) beyond (wide.hip kernels)
static void plan_bg(pmd_arena& ar, bg_plan& p, long D, int n, int K) {
  p.nblk = (int)((D + BG_BLK - 1) / BG_BLK);
  p.ld = pmd_time_ld(n);
  p.rp = pmd_tile_rpad(K);
  const size_t nb = p.nblk, rp = p.rp;
  const bool wide = p.rp > 64;
  p.omT = ar.take_n<float>(rp * p.ld);
  p.ypart = wide ? nullptr : ar.take_n<float>(nb * XBT_SLICES * 64 * BG_BLK);
  p.yt = ar.take_n<float>(nb * rp * BG_BLK);
  p.qt = ar.take_n<float>(nb * rp * BG_BLK);
  p.zero_bytes = ar.used;
  p.bpart = ar.take_n<float>(nb * rp * p.ld);
  p.bm = ar.take_n<float>(rp * p.ld);
  p.gblk = ar.take_n<double>(nb * rp * rp);
  p.gsum = ar.take_n<double>(rp * rp);
  p.nmat = ar.take_n<double>(rp * rp);
  p.lam = ar.take_n<double>(rp);
  p.eig_ws = nullptr;
  p.eig_ws_bytes = 0;
  if (wide) {
    p.eig_ws_bytes = pmd_wide_eig_workspace_bytes(p.rp, 1);
    p.eig_ws = ar.take(p.eig_ws_bytes);
  }
}

size_t pmd_bg_workspace_bytes_impl(long D, int n, int K) {
  pmd_arena ar((void*)0x1000, ~size_t(0) >> 1);
  bg_plan p;
  plan_bg(ar, p, D, n, K);
  return ar.used + 4096;
}

// background rSVD with a sketch wider than 64 columns: same steps, generic-width kernels (wide.hip); the tall-skinny
// orthonormalisation is two rounds of (block Gram matrices, summed) -> eigen-whitening -> row mixing
static int background_rsvd_wide(pmd_ctx* ctx, const bg_plan& p, const float* xs, long D, int n, long ld, int K, int l, uint64_t seed,
                                float* basis_out, void* ws) {
  const int rp = p.rp, nblk = p.nblk;
  const long srb = (long)rp * BG_BLK, rp2 = (long)rp * rp;
  PMD_HIP(ctx, hipMemsetAsync(ws, 0, p.zero_bytes, ctx->stream));
  RUN(pmd_launch_rng(ctx, seed, PMD_STREAM_BG_OMEGA, 0, 0, 1, n, l, 1, p.omT, p.ld, 0));
  RUN(pmd_launch_tile_xbt_rp(ctx, xs, ld, nullptr, 0, BG_BLK, BG_BLK, p.omT, 0, p.ld, p.yt, srb, 0, BG_BLK, nblk, n, 1, l));
  const float* src = p.yt;
  for (int pass = 0; pass < 2; ++pass) {
    RUN(pmd_launch_wide_gram(ctx, src, srb, BG_BLK, BG_BLK, nblk, 1, rp, p.gblk));
    hipLaunchKernelGGL(sum_gram_blocks_kernel, dim3((unsigned)((rp2 + 255) / 256)), dim3(256), 0, ctx->stream, p.gblk, nblk, p.gsum, (int)rp2);
    PMD_LAUNCH_CHECK(ctx, "sum_gram_blocks_kernel");
    RUN(pmd_launch_wide_eig(ctx, p.gsum, 1, rp, l, 1, 1e-13, p.nmat, p.lam, 1, p.eig_ws, p.eig_ws_bytes));
    RUN(pmd_launch_wide_rowmix(ctx, src, srb, BG_BLK, p.nmat, 0, rp, l, l, p.qt, srb, BG_BLK, BG_BLK, nblk));
    src = p.qt;
  }
  RUN(pmd_launch_tile_atx_rp(ctx, xs, ld, nullptr, 0, BG_BLK, BG_BLK, p.qt, srb, BG_BLK, p.bpart, (long)rp * p.ld, p.ld, nblk, n, 1, l));
  RUN(pmd_launch_reduce_slices(ctx, p.bpart, 0, (long)rp * p.ld, nblk, (long)rp * p.ld, p.bm, 0, 1));
  RUN(pmd_launch_wide_gram(ctx, p.bm, 0, p.ld, n, 1, 1, rp, p.gblk));
  RUN(pmd_launch_wide_eig(ctx, p.gblk, 1, rp, l, 0, 0.0, p.nmat, p.lam, 1, p.eig_ws, p.eig_ws_bytes));
  RUN(pmd_launch_wide_rowmix(ctx, p.qt, srb, BG_BLK, p.nmat, 0, rp, l, K, p.qt, srb, BG_BLK, BG_BLK, nblk));
  hipLaunchKernelGGL(unblock_basis_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, p.qt, D, K, basis_out, rp);
  PMD_LAUNCH_CHECK(ctx, "unblock_basis_kernel");
  return PMD_OK;
}

// xs: standardised sample, pixel-major [c][f], leading dimension ld >= pmd_time_ld(n), with
// round_up(D, 256) rows allocated (rows >= D zero).  basis_out: [c][k], K columns.
int pmd_background_rsvd_impl(pmd_ctx* ctx, const float* xs, long D, int n, long ld, int K, uint64_t seed,
                             float* basis_out, void* ws, size_t ws_bytes) {
  const int l = K + 10;
  if (K < 1 || l > 1024) return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "pmd_background_rsvd", "background_rank must be in [1, 1014]");
  if (ld < pmd_time_ld(n)) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_background_rsvd", "leading dimension too small");
  pmd_arena ar(ws, ws_bytes);
  bg_plan p;
  plan_bg(ar, p, D, n, K);
  if (ar.overflow) return pmd_fail(ctx, PMD_ERR_WORKSPACE, "pmd_background_rsvd", "workspace too small");
  if (p.rp > 64) return background_rsvd_wide(ctx, p, xs, D, n, ld, K, l, seed, basis_out, ws);
  const long s64b = 64L * BG_BLK;
  const int nblk = p.nblk;
  PMD_HIP(ctx, hipMemsetAsync(ws, 0, p.zero_bytes, ctx->stream));
  RUN(pmd_launch_rng(ctx, seed, PMD_STREAM_BG_OMEGA, 0, 0, 1, n, l, 1, p.omT, p.ld, 0));
  // Y = X Omega, block by block (Y^T blocks)
  RUN(pmd_launch_tile_xbt(ctx, xs, ld, nullptr, 0, BG_BLK, BG_BLK, p.omT, 0, p.ld, p.ypart, XBT_SLICES * s64b, s64b, BG_BLK, nblk, n, XBT_SLICES));
  RUN(pmd_launch_reduce_slices(ctx, p.ypart, XBT_SLICES * s64b, s64b, XBT_SLICES, s64b, p.yt, s64b, nblk));
  // CholeskyQR2
  const float* src = p.yt;
  for (int pass = 0; pass < 2; ++pass) {
    RUN(pmd_launch_tile_gram(ctx, src, s64b, BG_BLK, BG_BLK, nblk, 1, p.gblk));
    hipLaunchKernelGGL(sum_gram_blocks_kernel, dim3(16), dim3(256), 0, ctx->stream, p.gblk, nblk, p.gsum);
    PMD_LAUNCH_CHECK(ctx, "sum_gram_blocks_kernel");
    hipLaunchKernelGGL(chol_inverse_kernel, dim3(1), dim3(64), 0, ctx->stream, p.gsum, l, 1e-13, p.nmat);
    PMD_LAUNCH_CHECK(ctx, "chol_inverse_kernel");
    RUN(pmd_launch_tile_rowmix(ctx, src, s64b, BG_BLK, p.nmat, 0, l, l, p.qt, s64b, BG_BLK, BG_BLK, nblk));
    src = p.qt;
  }
  // B = Q^T X (sum of block contributions), SVD via Gram
  RUN(pmd_launch_tile_atx(ctx, xs, ld, nullptr, 0, BG_BLK, BG_BLK, p.qt, s64b, BG_BLK, p.bpart, 64L * p.ld, p.ld, nblk, n, 1));
  RUN(pmd_launch_reduce_slices(ctx, p.bpart, 0, 64L * p.ld, nblk, 64L * p.ld, p.bm, 0, 1));
  RUN(pmd_launch_tile_gram(ctx, p.bm, 0, p.ld, n, 1, 1, p.gblk));
  RUN(pmd_launch_small_eig(ctx, p.gblk, 1, l, 0, 0.0, p.nmat, p.lam, 1));
  // U = Q u[:, :K]
  RUN(pmd_launch_tile_rowmix(ctx, p.qt, s64b, BG_BLK, p.nmat, 0, l, K, p.qt, s64b, BG_BLK, BG_BLK, nblk));
  hipLaunchKernelGGL(unblock_basis_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, p.qt, D, K, basis_out);
  PMD_LAUNCH_CHECK(ctx, "unblock_basis_kernel");
  return PMD_OK;
}
