// extern "C" surface of libpmd_hip.so (declared in include/pmd_hip.h).
#include "pmd_internal.h"
#include "../../include/pmd_hip.h"

#define CTX_CHECK(ctx) \
  if (!(ctx)) return PMD_ERR_ARG;

extern "C" {

int pmd_version(void) { return 1; }

int pmd_ctx_create(int device, void* hip_stream, pmd_ctx** out) {
  if (!out) return PMD_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return PMD_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return PMD_ERR_HIP;
  pmd_ctx* ctx = new pmd_ctx();
  ctx->device = device;
  ctx->stream = (hipStream_t)hip_stream;
  ctx->tables = nullptr;
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  ctx->scratch2 = nullptr;
  ctx->scratch2_bytes = 0;
  ctx->atx_ranks = nullptr;
  ctx->atx_rows = 0;
  ctx->split_ws = nullptr;
  ctx->split_ws_bytes = 0;
  {
    // PMD_GEMM_SPLIT=0: every product through rocBLAS sgemm; default: products of >= PMD_GEMM_SPLIT_MIN_GFLOP (100) GFLOP with no
    // dimension below PMD_GEMM_SPLIT_MIN_DIM (256) run as three fp16-piece products on the fp16 matrix cores (gemm_f16x2.hip)
    const char* gs = getenv("PMD_GEMM_SPLIT");
    ctx->gemm_split = (gs && !strcmp(gs, "0")) ? 0 : 1;
    const char* gm = getenv("PMD_GEMM_SPLIT_MIN_GFLOP");
    ctx->gemm_split_min_flop = (gm ? atof(gm) : 100.0) * 1e9;
    const char* gd = getenv("PMD_GEMM_SPLIT_MIN_DIM");
    ctx->gemm_split_min_dim = gd ? atoi(gd) : 256;
    ctx->f16x2 = nullptr;
  }
  ctx->blas = nullptr;
  ctx->err[0] = 0;
  ctx->profile = false;
  ctx->null_cutoff = -1.f;
  ctx->comm = nullptr;
  ctx->comm_rank = 0;
  ctx->comm_world = 0;
  ctx->atx_label = nullptr;
  if (rocblas_create_handle(&ctx->blas) != rocblas_status_success) { delete ctx; return PMD_ERR_BLAS; }
  rocblas_set_stream(ctx->blas, ctx->stream);
  rocblas_set_pointer_mode(ctx->blas, rocblas_pointer_mode_host);
  if (pmd_init_tables(ctx) != PMD_OK) { rocblas_destroy_handle(ctx->blas); delete ctx; return PMD_ERR_HIP; }
  *out = ctx;
  return PMD_OK;
}

int pmd_ctx_destroy(pmd_ctx* ctx) {
  CTX_CHECK(ctx);
  hipSetDevice(ctx->device);
  if (ctx->comm) pmd_comm_destroy_impl(ctx);
  if (ctx->tables) hipFree(ctx->tables);
  if (ctx->scratch) hipFree(ctx->scratch);
  if (ctx->scratch2) hipFree(ctx->scratch2);
  if (ctx->split_ws) hipFree(ctx->split_ws);
  pmd_f16x2_destroy(ctx);
  if (ctx->blas) rocblas_destroy_handle(ctx->blas);
  delete ctx;
  return PMD_OK;
}

int pmd_comm_unique_id(void* out128) { return pmd_comm_unique_id_impl(out128); }
int pmd_comm_init(pmd_ctx* ctx, const void* unique_id128, int rank, int world) {
  CTX_CHECK(ctx);
  return pmd_comm_init_impl(ctx, unique_id128, rank, world);
}
int pmd_comm_destroy(pmd_ctx* ctx) {
  CTX_CHECK(ctx);
  return pmd_comm_destroy_impl(ctx);
}
int pmd_comm_all_reduce_f32(pmd_ctx* ctx, float* buf, size_t count) {
  CTX_CHECK(ctx);
  if (!buf && count) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_comm_all_reduce_f32", "null pointer");
  return pmd_comm_all_reduce_f32_impl(ctx, buf, count);
}
int pmd_comm_all_gather(pmd_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank) {
  CTX_CHECK(ctx);
  if ((!send || !recv) && bytes_per_rank) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_comm_all_gather", "null pointer");
  return pmd_comm_all_gather_impl(ctx, send, recv, bytes_per_rank);
}

int pmd_ctx_set_null_cutoff(pmd_ctx* ctx, float rel_cutoff) {
  CTX_CHECK(ctx);
  ctx->null_cutoff = rel_cutoff;
  return PMD_OK;
}

int pmd_ctx_set_stream(pmd_ctx* ctx, void* hip_stream) {
  CTX_CHECK(ctx);
  ctx->stream = (hipStream_t)hip_stream;
  rocblas_set_stream(ctx->blas, ctx->stream);
  return PMD_OK;
}

int pmd_ctx_sync(pmd_ctx* ctx) {
  CTX_CHECK(ctx);
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PMD_OK;
}

const char* pmd_last_error(pmd_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int pmd_profile_enable(pmd_ctx* ctx, int on) {
  CTX_CHECK(ctx);
  for (auto& r : ctx->recs) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
  ctx->recs.clear();
  ctx->profile = on != 0;
  return PMD_OK;
}

// Sum of the event-timed durations of every launch group called `name` since the last
// pmd_profile_enable (synchronises the stream).  total_ms/count may be NULL.
int pmd_profile_query(pmd_ctx* ctx, const char* name, double* total_ms, int* count) {
  CTX_CHECK(ctx);
  PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double tot = 0.0;
  int n = 0;
  for (auto& r : ctx->recs) {
    if (strcmp(r.name, name) != 0) continue;
    float ms = 0.f;
    PMD_HIP(ctx, hipEventElapsedTime(&ms, r.start, r.stop));
    tot += ms;
    n++;
  }
  if (total_ms) *total_ms = tot;
  if (count) *count = n;
  return PMD_OK;
}

// Names seen so far, '\n'-separated, written into buf (truncated to cap-1 characters).
int pmd_profile_names(pmd_ctx* ctx, char* buf, int cap) {
  CTX_CHECK(ctx);
  if (!buf || cap < 1) return PMD_ERR_ARG;
  buf[0] = 0;
  std::vector<const char*> seen;
  for (auto& r : ctx->recs) {
    bool dup = false;
    for (auto s : seen) dup = dup || strcmp(s, r.name) == 0;
    if (dup) continue;
    seen.push_back(r.name);
    if ((int)(strlen(buf) + strlen(r.name) + 2) >= cap) break;
    strcat(buf, r.name);
    strcat(buf, "\n");
  }
  return PMD_OK;
}

int pmd_rng_normal(pmd_ctx* ctx, uint64_t seed, uint32_t stream, uint32_t index0, uint32_t index_step, int batch,
                   long rows, int cols, int transpose, float* out, long ld, long batch_stride) {
  CTX_CHECK(ctx);
  for (int b0 = 0; b0 < batch; b0 += 32768) {
    const int bn = (batch - b0 < 32768) ? batch - b0 : 32768;
    int rc = pmd_launch_rng(ctx, seed, stream, index0 + (uint32_t)b0 * index_step, index_step, bn, rows, cols, transpose,
                            out + (long)b0 * batch_stride, ld, batch_stride);
    if (rc != PMD_OK) return rc;
  }
  return PMD_OK;
}

int pmd_stats(pmd_ctx* ctx, const float* movie, int T, long D, int frame_const, int compute_normalizer, float* mean_out,
              float* std_out, void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_launch_stats(ctx, movie, T, D, frame_const, compute_normalizer, mean_out, std_out, ws, ws_bytes);
}

int pmd_standardize_transpose(pmd_ctx* ctx, const float* movie, long D, const int* frames, int nf, const float* mean,
                              const float* std, float* out, long ld) {
  CTX_CHECK(ctx);
  return pmd_launch_standardize_transpose(ctx, movie, D, frames, nf, mean, std, out, ld);
}

size_t pmd_background_rsvd_workspace_bytes(long D, int n, int K) { return pmd_bg_workspace_bytes_impl(D, n, K); }
int pmd_background_rsvd(pmd_ctx* ctx, const float* xs, long D, int n, long ld, int K, uint64_t seed, float* basis_out,
                        void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_background_rsvd_impl(ctx, xs, D, n, ld, K, seed, basis_out, ws, ws_bytes);
}

size_t pmd_bg_project_workspace_bytes(long D, int T) { return pmd_bg_project_workspace_bytes_impl(D, T); }
int pmd_bg_project(pmd_ctx* ctx, const float* xs, long D, int T, long ld, const float* basis, int K, float* pj_out,
                   long ldp, void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_bg_project_impl(ctx, xs, D, T, ld, basis, K, pj_out, ldp, ws, ws_bytes);
}
int pmd_bg_filter(pmd_ctx* ctx, const float* xs, float* xf_out, long D, int nf, long ld, const float* basis, int K,
                  const float* pj, long ldp) {
  CTX_CHECK(ctx);
  return pmd_launch_filter(ctx, xs, xf_out, D, nf, ld, basis, K, pj, ldp);
}
int pmd_scale_rows(pmd_ctx* ctx, float* x, long D, int nf, long ld, const float* w) {
  CTX_CHECK(ctx);
  return pmd_launch_scale_rows(ctx, x, D, nf, ld, w);
}

size_t pmd_threshold_sim_workspace_bytes(int b1, int b2, int t, int iters) {
  return pmd_sim_workspace_bytes_impl(b1 * b2, t, iters);
}
int pmd_threshold_sim(pmd_ctx* ctx, int b1, int b2, int t, int iters, uint64_t seed, float* stats_out, void* ws,
                      size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_threshold_sim_impl(ctx, b1, b2, t, iters, seed, stats_out, ws, ws_bytes);
}

size_t pmd_tiles_workspace_bytes(int n_tiles, int b1, int b2, int P, int r, int a, int t_crop, long ldv, long n_rows) {
  return pmd_tiles_workspace_bytes_impl(n_tiles, b1 * b2, P, r, a, t_crop, ldv, n_rows);
}
int pmd_tiles_decompose(pmd_ctx* ctx, const float* xf, long ldx, long n_rows, int t_crop, const int* tile_pix, int n_tiles, int b1,
                        int b2, const int* pool_q, int pool_max, int P, const int* pool_idx, const float* pool_w, int r,
                        int a, float thr_s, float thr_t, int max_fail, uint64_t seed, uint32_t omega_index0,
                        uint32_t omega_index_step, float* Ut_out, float* V_out, long ldv, float* stats_out,
                        int* good_out, int* keep_out, int* ranks_out, double* lam_out, void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_tiles_decompose_impl(ctx, xf, ldx, n_rows, t_crop, tile_pix, n_tiles, b1, b2, pool_q, pool_max, P, pool_idx,
                                  pool_w, r, a, thr_s, thr_t, max_fail, seed, omega_index0, omega_index_step, Ut_out,
                                  V_out, ldv, stats_out, good_out, keep_out, ranks_out, lam_out, ws, ws_bytes, 7);
}
int pmd_tiles_decompose_staged(pmd_ctx* ctx, const float* xf, long ldx, long n_rows, int t_crop, const int* tile_pix, int n_tiles,
                               int b1, int b2, const int* pool_q, int pool_max, int P, const int* pool_idx,
                               const float* pool_w, int r, int a, float thr_s, float thr_t, int max_fail, uint64_t seed,
                               uint32_t omega_index0, uint32_t omega_index_step, float* Ut_out, float* V_out, long ldv,
                               float* stats_out, int* good_out, int* keep_out, int* ranks_out, double* lam_out, void* ws,
                               size_t ws_bytes, int stages) {
  CTX_CHECK(ctx);
  if (stages < 1 || stages > 7) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_tiles_decompose_staged", "stages must be a mask of bits 0..2");
  return pmd_tiles_decompose_impl(ctx, xf, ldx, n_rows, t_crop, tile_pix, n_tiles, b1, b2, pool_q, pool_max, P, pool_idx,
                                  pool_w, r, a, thr_s, thr_t, max_fail, seed, omega_index0, omega_index_step, Ut_out,
                                  V_out, ldv, stats_out, good_out, keep_out, ranks_out, lam_out, ws, ws_bytes, stages);
}
int pmd_tiles_hook_offsets(int n_tiles, int b1, int b2, int P, int r, int a, int t_crop, long ldv, long n_rows,
                           size_t* vds_offset, size_t* s_offset) {
  if (!vds_offset || !s_offset) return PMD_ERR_ARG;
  return pmd_tiles_hook_offsets_impl(n_tiles, b1 * b2, P, r, a, t_crop, ldv, n_rows, vds_offset, s_offset);
}

size_t pmd_tiles_residual_workspace_bytes(int n_tiles, int b1, int b2, int r, int a, int L, long n_rows) {
  return pmd_tiles_residual_workspace_bytes_impl(n_tiles, b1 * b2, r, a, L, n_rows);
}
int pmd_tiles_residual(pmd_ctx* ctx, const float* xw, long ldx, long n_rows, int L, const int* tile_pix, int n_tiles,
                       int b1, int b2, int r, int a, float thr_s, float thr_t, int max_fail, uint64_t seed,
                       uint32_t omega_index0, uint32_t omega_index_step, float* Ucur, int* counts, float* stats_out,
                       int* good_out, int* keep_out, void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_tiles_residual_impl(ctx, xw, ldx, n_rows, L, tile_pix, n_tiles, b1, b2, r, a, thr_s, thr_t, max_fail, seed,
                                 omega_index0, omega_index_step, Ucur, counts, stats_out, good_out, keep_out, ws, ws_bytes);
}
int pmd_tiles_truncate(pmd_ctx* ctx, float* U, int dpad, const int* counts, int n_tiles, int rpad) {
  CTX_CHECK(ctx);
  if (rpad < 64 || rpad % 64) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_tiles_truncate", "rpad must be a positive multiple of 64");
  return pmd_launch_tile_truncate(ctx, U, dpad, counts, n_tiles, rpad);
}

int pmd_weight_tiles(pmd_ctx* ctx, const float* Ut, int dpad, const int* tile_pix, int d, const float* w,
                     const float* cumw, const int* ranks, float* Uw_out, int n_tiles) {
  CTX_CHECK(ctx);
  return pmd_launch_weight_tiles(ctx, Ut, dpad, tile_pix, d, w, cumw, ranks, Uw_out, n_tiles);
}

int pmd_tiles_project(pmd_ctx* ctx, const float* x, long ldx, int T, const int* tile_pix, int n_tiles, int d,
                      const float* A, int dpad, float* Out, long ldo, int slices) {
  CTX_CHECK(ctx);
  ctx->atx_label = "tile_atx_main";
  const int rc = pmd_launch_tile_atx(ctx, x, ldx, tile_pix, d, 0, d, A, 64L * dpad, dpad, Out, 64L * ldo, ldo, n_tiles, T, slices);
  ctx->atx_label = nullptr;
  return rc;
}

int pmd_tiles_project_ranked(pmd_ctx* ctx, const float* x, long ldx, int T, const int* tile_pix, int n_tiles, int d,
                             const float* A, int dpad, float* Out, long ldo, int slices, const int* ranks) {
  CTX_CHECK(ctx);
  ctx->atx_label = "tile_atx_proj";
  ctx->atx_ranks = ranks;
  const int rc = pmd_launch_tile_atx(ctx, x, ldx, tile_pix, d, 0, d, A, 64L * dpad, dpad, Out, 64L * ldo, ldo, n_tiles, T, slices);
  ctx->atx_ranks = nullptr;
  ctx->atx_label = nullptr;
  return rc;
}

int pmd_compact_rows(pmd_ctx* ctx, const float* Out, long ldo, const int* col_off, const int* ranks, int T, float* Z,
                     long ldz, int n_tiles) {
  CTX_CHECK(ctx);
  return pmd_launch_compact_rows(ctx, Out, 64L * ldo, ldo, col_off, ranks, T, Z, ldz, n_tiles);
}

int pmd_gram_u(pmd_ctx* ctx, const float* Uw, int dpad, int b1, int b2, const int* tile_pix, const int* pairs,
               int n_pairs, const int* origins, const int* col_off, const int* ranks, int n_tiles, int Rt,
               const float* basis, long D, int K, float* G, long ldg) {
  CTX_CHECK(ctx);
  return pmd_gram_u_impl(ctx, Uw, dpad, b1, b2, tile_pix, pairs, n_pairs, origins, col_off, ranks, n_tiles, Rt, basis, D,
                         K, G, ldg);
}

size_t pmd_orthogonalize_workspace_bytes(int R, int m, int has_m) { return pmd_orthogonalize_workspace_bytes_impl(R, m, has_m); }
int pmd_orthogonalize(pmd_ctx* ctx, float* G, int R, const float* M, int m, long ldm, float* P_out, long ldp,
                      int* rprime_host, void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_orthogonalize_impl(ctx, G, R, M, m, ldm, P_out, ldp, rprime_host, ws, ws_bytes);
}

size_t pmd_projected_svd_workspace_bytes(int rows_p, int n1, int n2) { return pmd_projected_svd_workspace_bytes_impl(rows_p, n1, n2); }
int pmd_projected_svd(pmd_ctx* ctx, const float* P, int rows_p, long ldp, const float* V, int n1, int n2, long ldv,
                      float* R_out, long ldr, float* s_out, float* Vt_out, long ldvt, void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_projected_svd_impl(ctx, P, rows_p, ldp, V, n1, n2, ldv, R_out, ldr, s_out, Vt_out, ldvt, ws, ws_bytes);
}

int pmd_scratch_trim(pmd_ctx* ctx, size_t keep_bytes) {
  CTX_CHECK(ctx);
  return pmd_split_scratch_trim(ctx, keep_bytes);
}

int pmd_gemm_split_active(pmd_ctx* ctx, int m, int n, int k) {
  if (!ctx) return 0;
  return pmd_f16x2_wanted(ctx, m, n, k) ? 1 : 0;
}

int pmd_gemm(pmd_ctx* ctx, int transA, int transB, int m, int n, int k, float alpha, const float* A, long lda,
             const float* B, long ldb, float beta, float* C, long ldc) {
  CTX_CHECK(ctx);
  return pmd_gemm_rm(ctx, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}

int pmd_csr_rows_spmm(pmd_ctx* ctx, const int64_t* indptr, const int* indices, const float* data, const int* rows, long n_sel,
                      const float* B, long ldb, int ncols, float* out, long ldo) {
  CTX_CHECK(ctx);
  if (!indptr || !indices || !data || !B || !out) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_csr_rows_spmm", "null pointer");
  return pmd_csr_rows_spmm_impl(ctx, (const long*)indptr, indices, data, rows, n_sel, B, ldb, ncols, out, ldo);
}
size_t pmd_diag_workspace_bytes(long T, long D) { return pmd_diag_workspace_bytes_impl(T, D); }
int pmd_neighbour_moments(pmd_ctx* ctx, const float* A, const float* B, const float* ref, long T, int d1, int d2, int accumulate,
                          double* moments, void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  if (!A || !ref || !moments || !ws || d1 < 1 || d2 < 1) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_neighbour_moments", "bad argument");
  return pmd_neighbour_moments_impl(ctx, A, B, ref, T, d1, d2, accumulate, moments, ws, ws_bytes);
}
int pmd_lag_moments(pmd_ctx* ctx, const float* A, const float* ref, long T, long D, int lag, int accumulate, double* moments,
                    void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  if (!A || !ref || !moments || !ws || D < 1) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_lag_moments", "bad argument");
  return pmd_lag_moments_impl(ctx, A, ref, T, D, lag, accumulate, moments, ws, ws_bytes);
}
int pmd_neighbour_image(pmd_ctx* ctx, const double* num, const double* den, long T, int d1, int d2, int kind, int mode, double* out) {
  CTX_CHECK(ctx);
  if (!num || !out || T < 2 || (kind != 0 && kind != 1) || (mode != 0 && mode != 1)) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_neighbour_image", "bad argument");
  return pmd_neighbour_image_impl(ctx, num, den, T, d1, d2, kind, mode, out);
}
int pmd_lag_image(pmd_ctx* ctx, const double* moments, long D, long n, double* out) {
  CTX_CHECK(ctx);
  if (!moments || !out || n < 1) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_lag_image", "bad argument");
  return pmd_lag_image_impl(ctx, moments, D, n, out);
}
int pmd_transpose_affine(pmd_ctx* ctx, const float* src, long lds_, long rows, int cols, const float* scale,
                         const float* shift, float* dst, long ldd) {
  CTX_CHECK(ctx);
  if (!src || !dst) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_transpose_affine", "null pointer");
  return pmd_transpose_affine_impl(ctx, src, lds_, rows, cols, scale, shift, dst, ldd);
}

int pmd_gram_blocks(pmd_ctx* ctx, const float* Uw, int dpad, int b1, int b2, const int* tile_pix, const int* pairs,
                    int n_pairs, const int* origins, const int* col_off, const int* ranks, int n_tiles, int Rt,
                    const float* basis, long D, int K, float* Gblk, float* Gbg, float* Gstrip, long ldgs) {
  CTX_CHECK(ctx);
  return pmd_gram_blocks_impl(ctx, Uw, dpad, b1, b2, tile_pix, pairs, n_pairs, origins, col_off, ranks, n_tiles, Rt, basis,
                              D, K, Gblk, Gbg, Gstrip, ldgs);
}
int pmd_gram_apply(pmd_ctx* ctx, const float* Gblk, const float* Gbg, const float* Gstrip, long ldgs, const int* nbr_ptr,
                   const int* nbr, const int* col_off, const int* ranks, int n_tiles, int Rt, int K, int max_rank,
                   const float* M, long ldm, int ncols, float* GM, long ldgm) {
  CTX_CHECK(ctx);
  return pmd_gram_apply_impl(ctx, Gblk, Gbg, Gstrip, ldgs, nbr_ptr, nbr, col_off, ranks, n_tiles, Rt, K, max_rank, M, ldm,
                             ncols, GM, ldgm);
}
int pmd_csr_count(pmd_ctx* ctx, int d1, int d2, int order_f, const int* cover1, const int* cover2, int n2,
                  const int* ranks, int K, long* row_nnz) {
  CTX_CHECK(ctx);
  return pmd_csr_count_impl(ctx, d1, d2, order_f, cover1, cover2, n2, ranks, K, row_nnz);
}
int pmd_csr_fill(pmd_ctx* ctx, int d1, int d2, int order_f, int b1, const int* cover1, const int* cover2,
                 const int* orig1, const int* orig2, int n2, const int* ranks, const int* col_off, const float* Ut,
                 int dpad, const float* w, const double* inv_cumw, const float* basis, int K, int Rt, const long* indptr,
                 double* data, int* indices, int* zero_count, int rpad) {
  CTX_CHECK(ctx);
  if (rpad < 64 || rpad % 64) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_csr_fill", "rpad must be a positive multiple of 64");
  return pmd_csr_fill_impl(ctx, d1, d2, order_f, b1, cover1, cover2, orig1, orig2, n2, ranks, col_off, Ut, dpad, w,
                           inv_cumw, basis, K, Rt, indptr, data, indices, zero_count, rpad);
}
size_t pmd_orthogonalize_factored_workspace_bytes(int m) { return pmd_orthogonalize_factored_workspace_bytes_impl(m); }
int pmd_orthogonalize_factored(pmd_ctx* ctx, const float* M, int Rc, int m, long ldm, const float* GM, long ldgm,
                               float* Et_out, long lde, int* rprime_host, void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_orthogonalize_factored_impl(ctx, M, Rc, m, ldm, GM, ldgm, Et_out, lde, rprime_host, ws, ws_bytes);
}
int pmd_orthogonalize_chol(pmd_ctx* ctx, const float* M, int Rc, int m, long ldm, const float* GM, long ldgm,
                           float* Et_out, long lde, int* ok_host, void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_orthogonalize_chol_impl(ctx, M, Rc, m, ldm, GM, ldgm, Et_out, lde, ok_host, ws, ws_bytes);
}
size_t pmd_orthogonalize_chol_workspace_bytes(int Rc, int m) { return pmd_orthogonalize_chol_workspace_bytes_impl(Rc, m); }
size_t pmd_gram_mtgm_workspace_bytes(int rows, int m) { return pmd_gram_mtgm_workspace_bytes_impl(rows, m); }
long pmd_gram_mtgm_ld(int rows) { return pmd_gram_mtgm_ld_impl(rows); }
int pmd_gram_mtgm(pmd_ctx* ctx, const float* M, int rows, int m, long ldm, const float* GM, long ldgm, float* C, long ldc,
                  void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_gram_mtgm_impl(ctx, M, rows, m, ldm, GM, ldgm, C, ldc, ws, ws_bytes);
}
size_t pmd_chol_inverse_workspace_bytes(int m) { return pmd_chol_inverse_workspace_bytes_impl(m); }
int pmd_chol_inverse(pmd_ctx* ctx, float* C, int m, long ldc, int abs_last_pivot, int* ok_host, void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_chol_inverse_impl(ctx, C, m, ldc, abs_last_pivot, ok_host, ws, ws_bytes);
}
int pmd_transpose(pmd_ctx* ctx, const float* src, long ld_src, int rows, int cols, float* dst, long ld_dst) {
  CTX_CHECK(ctx);
  return pmd_transpose_impl(ctx, src, ld_src, rows, cols, dst, ld_dst);
}
int pmd_psvd_vp_gram(pmd_ctx* ctx, const float* Et, int rp, int m, long lde, const float* W1, int nc, long ldw, int et_lower, float* Vp,
                     long ldv, float* C, long ldc) {
  CTX_CHECK(ctx);
  if (rp < 1 || m < rp || nc < 0 || ldc < rp) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_psvd_vp_gram", "bad shape");
  return pmd_psvd_vp_gram_impl(ctx, Et, rp, m, lde, W1, nc, ldw, et_lower, Vp, ldv, C, ldc);
}
size_t pmd_psvd_finish_workspace_bytes(int rp) { return pmd_psvd_finish_workspace_bytes_impl(rp); }
int pmd_psvd_finish(pmd_ctx* ctx, float* C, long ldc, int rp, const float* Vp, int nc, long ldv, float* W_out, long ldw, float* s_out,
                    float* Vt_out, long ldvt, void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  if (rp < 1 || nc < 0 || ldc < rp) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_psvd_finish", "bad shape");
  return pmd_psvd_finish_impl(ctx, C, ldc, rp, Vp, nc, ldv, W_out, ldw, s_out, Vt_out, ldvt, ws, ws_bytes);
}
size_t pmd_projected_svd_factored_workspace_bytes(int Rc, int m, int rp, int T) {
  return pmd_projected_svd_factored_workspace_bytes_impl(Rc, m, rp, T);
}
int pmd_projected_svd_factored(pmd_ctx* ctx, const float* M, int Rc, int m, long ldm, const float* Et, int rp, long lde,
                               const float* Z, int T, long ldz, float* R_out, long ldr, float* s_out, float* Vt_out,
                               long ldvt, float* Vp_out, long ldvp, float* X1_out, const float* W1_in, int et_lower,
                               void* ws, size_t ws_bytes) {
  CTX_CHECK(ctx);
  return pmd_projected_svd_factored_impl(ctx, M, Rc, m, ldm, Et, rp, lde, Z, T, ldz, R_out, ldr, s_out, Vt_out, ldvt,
                                         Vp_out, ldvp, X1_out, W1_in, et_lower, ws, ws_bytes);
}

// ---- kernel-level entry points ---------------------------------------------------------------
int pmdk_tile_atx(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                  const float* A, long a_tile_stride, int a_ld, float* Out, long out_tile_stride, long ldo, int n_tiles,
                  int T, int slices) {
  CTX_CHECK(ctx);
  return pmd_launch_tile_atx(ctx, X, ldx, pix, pix_stride, row0_stride, d, A, a_tile_stride, a_ld, Out, out_tile_stride, ldo, n_tiles, T, slices);
}
int pmdk_tile_atx_rows(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                       const float* A, long a_tile_stride, int a_ld, float* Out, long out_tile_stride, long ldo, int n_tiles,
                       int T, int slices, int rows) {
  CTX_CHECK(ctx);
  ctx->atx_rows = rows;
  const int rc = pmd_launch_tile_atx(ctx, X, ldx, pix, pix_stride, row0_stride, d, A, a_tile_stride, a_ld, Out, out_tile_stride, ldo, n_tiles, T, slices);
  ctx->atx_rows = 0;
  return rc;
}
int pmdk_tile_xbt(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                  const float* B, long b_tile_stride, long ldb, float* S, long s_tile_stride, long s_slice_stride,
                  int s_ld, int n_tiles, int T, int slices) {
  CTX_CHECK(ctx);
  return pmd_launch_tile_xbt(ctx, X, ldx, pix, pix_stride, row0_stride, d, B, b_tile_stride, ldb, S, s_tile_stride, s_slice_stride, s_ld, n_tiles, T, slices);
}
int pmdk_tile_gram(pmd_ctx* ctx, const float* In, long tile_stride, long ld, int len, int n_tiles, int slices, double* G) {
  CTX_CHECK(ctx);
  return pmd_launch_tile_gram(ctx, In, tile_stride, ld, len, n_tiles, slices, G);
}
int pmdk_tile_rowmix(pmd_ctx* ctx, const float* In, long in_tile_stride, long ld_in, const double* N, long n_tile_stride,
                     int n_in, int n_out, float* Out, long out_tile_stride, long ld_out, int len, int n_tiles) {
  CTX_CHECK(ctx);
  return pmd_launch_tile_rowmix(ctx, In, in_tile_stride, ld_in, N, n_tile_stride, n_in, n_out, Out, out_tile_stride, ld_out, len, n_tiles);
}
int pmdk_small_qr(pmd_ctx* ctx, const float* Yt, long y_tile_stride, int y_ld, int P, int l, float* Qt, long q_tile_stride,
                  int q_ld, int n_tiles) {
  CTX_CHECK(ctx);
  return pmd_launch_small_qr(ctx, Yt, y_tile_stride, y_ld, P, l, Qt, q_tile_stride, q_ld, n_tiles);
}
int pmdk_small_eig(pmd_ctx* ctx, const double* G, int slices, int n, int mode, double tol, double* Nout, double* lam_out,
                   int n_tiles) {
  CTX_CHECK(ctx);
  return pmd_launch_small_eig(ctx, G, slices, n, mode, tol, Nout, lam_out, n_tiles);
}
int pmdk_tile_pool_bin(pmd_ctx* ctx, const float* X, long ldx, long n_rows, const int* pix, int n_tiles, int d,
                       const int* pool_q, int pool_max, int P, int a, int nbins, float* xbar, float* abar, long ld_ab,
                       long tile_stride) {
  CTX_CHECK(ctx);
  return pmd_launch_tile_pool_bin(ctx, X, ldx, n_rows, pix, n_tiles, d, pool_q, pool_max, P, a, nbins, xbar, abar, ld_ab, tile_stride);
}
int pmdk_roughness(pmd_ctx* ctx, const float* Ut, long u_tile_stride, int u_ld, int b1, int b2, const float* V,
                   long v_tile_stride, long v_ld, int T, int r, float* stats, int n_tiles) {
  CTX_CHECK(ctx);
  return pmd_launch_stats_roughness(ctx, Ut, u_tile_stride, u_ld, b1, b2, V, v_tile_stride, v_ld, T, r, stats, n_tiles);
}
int pmdk_syevd(pmd_ctx* ctx, int n, float* A, long lda, float* w, float* work, int* info) {
  CTX_CHECK(ctx);
  return pmd_syevd(ctx, n, A, lda, w, work, info);
}

// stage 1 of the two-stage reduction alone (tests): dense -> band; the workspace is allocated and freed inside
int pmdk_sy2sb(pmd_ctx* ctx, int n, float* A, long lda, float* tau1, int* flag_host) {
  CTX_CHECK(ctx);
  const size_t bytes = pmd_sy2sb_workspace_bytes_impl(n);
  void* ws = nullptr;
  if (hipMalloc(&ws, bytes) != hipSuccess) return pmd_fail(ctx, PMD_ERR_HIP, "pmdk_sy2sb", "hipMalloc");
  const int rc = pmd_sy2sb_impl(ctx, n, A, lda, tau1, flag_host, ws, bytes);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(ws);
  return rc;
}

// stages 1 + 2 (tests): dense -> band -> tridiagonal; d[n], e[n - 1] on the device
int pmdk_sytrd2(pmd_ctx* ctx, int n, float* A, long lda, float* tau1, float* d, float* e, int* flag_host) {
  CTX_CHECK(ctx);
  const size_t b1 = pmd_sy2sb_workspace_bytes_impl(n), b2 = pmd_sb2st_workspace_bytes_impl(n);
  void *w1 = nullptr, *w2 = nullptr;
  if (hipMalloc(&w1, b1) != hipSuccess) return pmd_fail(ctx, PMD_ERR_HIP, "pmdk_sytrd2", "hipMalloc");
  if (hipMalloc(&w2, b2) != hipSuccess) {
    (void)hipFree(w1);
    return pmd_fail(ctx, PMD_ERR_HIP, "pmdk_sytrd2", "hipMalloc");
  }
  int rc = pmd_sy2sb_impl(ctx, n, A, lda, tau1, flag_host, w1, b1);
  float *V2 = nullptr, *tau2 = nullptr;
  if (rc == PMD_OK) rc = pmd_sb2st_impl(ctx, n, A, lda, d, e, &V2, &tau2, w2, b2);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(w1);
  (void)hipFree(w2);
  return rc;
}

int pmdk_sytrd(pmd_ctx* ctx, int n, float* A, long lda, float* d, float* e, float* tau, int impl) {
  CTX_CHECK(ctx);
  return pmd_sytrd_auto(ctx, n, A, lda, d, e, tau, impl);
}

}  // extern "C"
