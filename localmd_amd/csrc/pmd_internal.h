// Launchers shared between the translation units of libpmd_hip.so.
#pragma once
#include "pmd_common.h"

int pmd_init_tables(pmd_ctx* ctx);
extern "C" {
int pmd_tile_dpad(int d);
long pmd_time_ld(long t);
size_t pmd_stats_workspace_bytes(int T, long D, int frame_const);
}

// rng.hip
int pmd_launch_rng(pmd_ctx* ctx, uint64_t seed, uint32_t stream, uint32_t index0, uint32_t index_step, int batch,
                   long rows, int cols, int transpose, float* out, long ld, long batch_stride);

// prep.hip
int pmd_launch_stats(pmd_ctx* ctx, const float* movie, int T, long D, int frame_const, int do_noise, float* mean_out,
                     float* std_out, void* ws, size_t ws_bytes);
int pmd_launch_standardize_transpose(pmd_ctx* ctx, const float* movie, long D, const int* frames, int nf,
                                     const float* mean, const float* stdv, float* out, long ld);
int pmd_launch_filter(pmd_ctx* ctx, const float* in, float* out, long D, int nf, long ld, const float* basis, int K,
                      const float* pj, long ldp);
int pmd_launch_scale_rows(pmd_ctx* ctx, float* x, long D, int nf, long ld, const float* w);
int pmd_launch_tile_pool_bin(pmd_ctx* ctx, const float* X, long ldx, long n_rows, const int* pix, int n_tiles, int d,
                             const int* pool_q, int pool_max, int P, int a, int nbins, float* xbar, float* abar,
                             long ld_ab, long tile_stride);

// tile_gemm.hip
int pmd_launch_tile_atx(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                        const float* A, long a_tile_stride, int a_ld, float* Out, long out_tile_stride, long ldo,
                        int n_tiles, int T, int slices);
int pmd_launch_tile_xbt(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                        const float* B, long b_tile_stride, long ldb, float* S, long s_tile_stride,
                        long s_slice_stride, int s_ld, int n_tiles, int T, int slices);
int pmd_launch_tile_gram(pmd_ctx* ctx, const float* In, long tile_stride, long ld, int len, int n_tiles, int slices,
                         double* G);
int pmd_launch_gram_f2d(pmd_ctx* ctx, const float* in, int ld, long n_blocks, double* out);
int pmd_launch_reduce_slices(pmd_ctx* ctx, const float* in, long tile_stride, long slice_stride, int slices, long n,
                             float* out, long out_tile_stride, int n_tiles);
int pmd_launch_tile_rowmix(pmd_ctx* ctx, const float* In, long in_tile_stride, long ld_in, const double* N,
                           long n_tile_stride, int n_in, int n_out, float* Out, long out_tile_stride, long ld_out,
                           int len, int n_tiles);

// small_la.hip
bool pmd_small_qr_fits(int P, int l);
int pmd_launch_small_qr(pmd_ctx* ctx, const float* Yt, long y_tile_stride, int y_ld, int P, int l, float* Qt,
                        long q_tile_stride, int q_ld, int n_tiles);
int pmd_launch_small_eig(pmd_ctx* ctx, const double* G, int slices, int n, int mode, double tol, double* Nout,
                         double* lam_out, int n_tiles);
int pmd_launch_small_chol(pmd_ctx* ctx, const double* G, int slices, int n, double tol, double* Nout, int n_tiles);
int pmd_launch_expand_pooled(pmd_ctx* ctx, const float* In, long in_tile_stride, int in_ld, const int* pool_idx,
                             const float* pool_w, int d, int r, float* Out, long out_tile_stride, int out_ld,
                             int n_tiles);
int pmd_launch_stats_roughness(pmd_ctx* ctx, const float* Ut, long u_tile_stride, int u_ld, int b1, int b2,
                               const float* V, long v_tile_stride, long v_ld, int T, int r, float* stats, int n_tiles,
                               int rp = PMD_RPAD);
int pmd_launch_decide(pmd_ctx* ctx, const float* stats, int r, float thr_s, float thr_t, int max_fail, int cap,
                      int n_tiles, int* good, int* keep, int* ranks, int rp = PMD_RPAD);

// wide.hip: generic-width forms (per-tile arrays [tile][rp][x], rp = pmd_tile_rpad(r) > 64)
extern "C" int pmd_tile_rpad(int r);
int pmd_launch_wide_gram(pmd_ctx* ctx, const float* In, long tile_stride, long ld, int len, int n_tiles, int slices, int rp,
                         double* G, const float* In2 = nullptr);
size_t pmd_wide_eig_workspace_bytes(int n, int n_tiles);
int pmd_launch_wide_eig(pmd_ctx* ctx, const double* G, int slices, int rp, int n, int mode, double tol, double* Nout,
                        double* lam_out, int n_tiles, void* ws, size_t ws_bytes);
int pmd_launch_wide_rowmix(pmd_ctx* ctx, const float* In, long in_tile_stride, long ld_in, const double* N, long n_tile_stride,
                           int rp, int n_in, int n_out, float* Out, long out_tile_stride, long ld_out, int len, int n_tiles);
int pmd_launch_tile_atx_rp(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                           const float* A, long a_tile_stride, int a_ld, float* Out, long out_tile_stride, long ldo, int n_tiles,
                           int T, int slices, int nrows);
int pmd_launch_tile_xbt_rp(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                           const float* B, long b_tile_stride, long ldb, float* S, long s_tile_stride, long s_slice_stride, int s_ld,
                           int n_tiles, int T, int slices, int nrows);

// expand.hip
int pmd_csr_rows_spmm_impl(pmd_ctx* ctx, const long* indptr, const int* indices, const float* data, const int* rows,
                           long n_sel, const float* B, long ldb, int ncols, float* out, long ldo);
int pmd_comm_unique_id_impl(void* out128);
int pmd_comm_init_impl(pmd_ctx* ctx, const void* unique_id128, int rank, int world);
int pmd_comm_destroy_impl(pmd_ctx* ctx);
int pmd_comm_all_reduce_f32_impl(pmd_ctx* ctx, float* buf, size_t count);
int pmd_comm_all_gather_impl(pmd_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank);
size_t pmd_diag_workspace_bytes_impl(long T, long D);
int pmd_neighbour_moments_impl(pmd_ctx* ctx, const float* A, const float* B, const float* ref, long T, int d1, int d2,
                               int accumulate, double* moments, void* ws, size_t ws_bytes);
int pmd_lag_moments_impl(pmd_ctx* ctx, const float* A, const float* ref, long T, long D, int lag, int accumulate, double* moments,
                         void* ws, size_t ws_bytes);
int pmd_neighbour_image_impl(pmd_ctx* ctx, const double* num, const double* den, long T, int d1, int d2, int kind, int mode,
                             double* out);
int pmd_lag_image_impl(pmd_ctx* ctx, const double* moments, long D, long n, double* out);
int pmd_transpose_affine_impl(pmd_ctx* ctx, const float* src, long lds_, long rows, int cols, const float* scale,
                              const float* shift, float* dst, long ldd);

// pipeline.hip
size_t pmd_tiles_workspace_bytes_impl(int n, int d, int P, int r, int a, int t_crop, long ldv, long n_rows);
int pmd_tiles_decompose_impl(pmd_ctx* ctx, const float* Xf, long ldx, long n_rows, int t_crop, const int* tile_pix, int n, int b1,
                             int b2, const int* pool_q, int pool_max, int P, const int* pool_idx, const float* pool_w,
                             int r, int a, float thr_s, float thr_t, int max_fail, uint64_t seed, uint32_t omega_index0,
                             uint32_t omega_index_step, float* Ut_out, float* V_out, long ldv, float* stats_out,
                             int* good_out, int* keep_out, int* ranks_out, double* sing_out, void* ws, size_t ws_bytes,
                             int stages);
int pmd_tiles_hook_offsets_impl(int n, int d, int P, int r, int a, int t_crop, long ldv, long n_rows, size_t* vds_off,
                                size_t* s_off);
size_t pmd_sim_workspace_bytes_impl(int d, int t, int iters);
int pmd_threshold_sim_impl(pmd_ctx* ctx, int b1, int b2, int t, int iters, uint64_t seed, float* stats_out, void* ws,
                           size_t ws_bytes);
size_t pmd_bg_workspace_bytes_impl(long D, int n, int K);
int pmd_background_rsvd_impl(pmd_ctx* ctx, const float* xs, long D, int n, long ld, int K, uint64_t seed,
                             float* basis_out, void* ws, size_t ws_bytes);

// global.hip
int pmd_gemm_rm(pmd_ctx* ctx, int transA, int transB, int m, int n, int k, float alpha, const float* A, long lda,
                const float* B, long ldb, float beta, float* C, long ldc);
int pmd_gemm_k_chunk(int k);
// gemm_f16x2.hip: fp32 products from two fp16 pieces per operand (X 2^-e = h1 + 2^-11 h2)
struct pmd_f16x2_op {
  const _Float16* h1;
  const _Float16* h2;
  const _Float16* h3;   // third piece (2^-22), or NULL
  long ld;
  int e;
};
bool pmd_f16x2_wanted(const pmd_ctx* ctx, int m, int n, int k);
long pmd_f16x2_ld(int cols);
size_t pmd_f16x2_bytes(int rows, int cols, int pieces);
int pmd_f16x2_split(pmd_ctx* ctx, int count, const float* const* X, const int* rows, const int* cols, const long* ld, void* const* buf,
                    pmd_f16x2_op* ops, int* usable, int pieces);
int pmd_f16x2_matmul(pmd_ctx* ctx, int tA, int tB, int m, int n, int k, float alpha, const pmd_f16x2_op& a, const pmd_f16x2_op& b, float beta,
                     float* C, long ldc, int* done);
int pmd_gemm_f16x2(pmd_ctx* ctx, int transA, int transB, int m, int n, int k, float alpha, const float* A, long lda, const float* B, long ldb,
                   float beta, float* C, long ldc, int* done);
int pmd_f16x2_exponents(pmd_ctx* ctx, int count, const float* const* X, const int* rows, const int* cols, const long* ld, int* e_out,
                        int* usable);
int pmd_f16cat_a(pmd_ctx* ctx, const float* X, int rows, int kk, long ld, int e, _Float16* out, long ldo);
int pmd_f16cat_b(pmd_ctx* ctx, const float* X, int kk, int cols, long ld, int e, _Float16* out, long ldo);
int pmd_f16_plain_matmul(pmd_ctx* ctx, int m, int n, int k, float alpha, const _Float16* A, long lda, const _Float16* B, long ldb, float beta,
                         float* C, long ldc, int* done);
int pmd_split_scratch(pmd_ctx* ctx, size_t need, void** out);
int pmd_split_scratch_trim(pmd_ctx* ctx, size_t keep_bytes);
void pmd_f16x2_destroy(pmd_ctx* ctx);
bool pmd_is_host_pointer(const void* p);
int pmd_syevd(pmd_ctx* ctx, int n, float* A, long lda, float* w, float* work, int* info);
size_t pmd_sy2sb_workspace_bytes_impl(int n);
int pmd_sy2sb_impl(pmd_ctx* ctx, int n, float* A, long lda, float* tau1, int* flag_host, void* ws, size_t ws_bytes);
int pmd_apply_q_off_impl(pmd_ctx* ctx, int n, const float* A, long lda, const float* tau, float* Z, long ldz, void* ws,
                         size_t ws_bytes, int off);
int pmd_syevd_two_stage(pmd_ctx* ctx, int n, float* A, long lda, float* w, int* info, int* done);
int pmd_sb2st_apply_q2_impl(pmd_ctx* ctx, int n, const float* V2, const float* tau2, float* Z, long ldz, int nvec);
size_t pmd_sb2st_workspace_bytes_impl(int n);
int pmd_sb2st_impl(pmd_ctx* ctx, int n, const float* A, long lda, float* d, float* e, float** V2_out, float** tau2_out,
                   void* ws, size_t ws_bytes);
int pmd_sytrd_auto(pmd_ctx* ctx, int n, float* A, long lda, float* d, float* e, float* tau, int impl);
int pmd_launch_weight_tiles(pmd_ctx* ctx, const float* Ut, int dpad, const int* pix, int d, const float* w,
                            const float* cumw, const int* ranks, float* Uw, int n_tiles);
int pmd_launch_compact_rows(pmd_ctx* ctx, const float* Out, long tile_stride, long ldo, const int* col_off,
                            const int* ranks, int T, float* Z, long ldz, int n_tiles);
int pmd_gram_u_impl(pmd_ctx* ctx, const float* Uw, int dpad, int b1, int b2, const int* pix, const int* pairs,
                    int n_pairs, const int* origins, const int* col_off, const int* ranks, int n_tiles, int Rt,
                    const float* basis, long D, int K, float* G, long ldg);
size_t pmd_orthogonalize_workspace_bytes_impl(int R, int m, int has_m);
int pmd_orthogonalize_impl(pmd_ctx* ctx, float* G, int R, const float* M, int m, long ldm, float* P_out, long ldp,
                           int* rprime_out, void* ws, size_t ws_bytes);
size_t pmd_projected_svd_workspace_bytes_impl(int rows_p, int n1, int n2);
int pmd_projected_svd_impl(pmd_ctx* ctx, const float* P, int rows_p, long ldp, const float* V, int n1, int n2, long ldv,
                           float* R_out, long ldr, float* s_out, float* Vt_out, long ldvt, void* ws, size_t ws_bytes);
size_t pmd_bg_project_workspace_bytes_impl(long D, int T);
int pmd_bg_project_impl(pmd_ctx* ctx, const float* xs, long D, int T, long ld, const float* basis, int K, float* out,
                        long ldo, void* ws, size_t ws_bytes);

// global.hip: block-sparse Gram, device CSR assembly, factored orthogonalisation / SVD
int pmd_gram_blocks_impl(pmd_ctx* ctx, const float* Uw, int dpad, int b1, int b2, const int* pix, const int* pairs,
                         int n_pairs, const int* origins, const int* col_off, const int* ranks, int n_tiles, int Rt,
                         const float* basis, long D, int K, float* Gblk, float* Gbg, float* Gstrip, long ldgs);
int pmd_gram_apply_impl(pmd_ctx* ctx, const float* Gblk, const float* Gbg, const float* Gstrip, long ldgs,
                        const int* nbr_ptr, const int* nbr, const int* col_off, const int* ranks, int n_tiles, int Rt,
                        int K, int max_rank, const float* M, long ldm, int ncols, float* GM, long ldgm);
int pmd_csr_count_impl(pmd_ctx* ctx, int d1, int d2, int order_f, const int* cover1, const int* cover2, int n2,
                       const int* ranks, int K, long* row_nnz);
int pmd_csr_fill_impl(pmd_ctx* ctx, int d1, int d2, int order_f, int b1, const int* cover1, const int* cover2,
                      const int* orig1, const int* orig2, int n2, const int* ranks, const int* col_off, const float* Ut,
                      int dpad, const float* w, const double* inv_cumw, const float* basis, int K, int Rt,
                      const long* indptr, double* data, int* indices, int* zero_count, int rpad);
size_t pmd_orthogonalize_factored_workspace_bytes_impl(int m);
int pmd_orthogonalize_factored_impl(pmd_ctx* ctx, const float* M, int Rc, int m, long ldm, const float* GM, long ldgm,
                                    float* Et_out, long lde, int* rprime_out, void* ws, size_t ws_bytes);
size_t pmd_projected_svd_factored_workspace_bytes_impl(int Rc, int m, int rp, int T);
size_t pmd_orthogonalize_chol_workspace_bytes_impl(int Rc, int m);
int pmd_projected_svd_factored_impl(pmd_ctx* ctx, const float* M, int Rc, int m, long ldm, const float* Et, int rp,
                                    long lde, const float* Z, int T, long ldz, float* R_out, long ldr, float* s_out,
                                    float* Vt_out, long ldvt, float* Vp_out, long ldvp, float* X1_out,
                                    const float* W1_in, int et_lower, void* ws, size_t ws_bytes);
int pmd_psvd_vp_gram_impl(pmd_ctx* ctx, const float* Et, int rp, int m, long lde, const float* W1, int nc, long ldw, int et_lower,
                          float* Vp, long ldv, float* C, long ldc);
size_t pmd_psvd_finish_workspace_bytes_impl(int rp);
int pmd_psvd_finish_impl(pmd_ctx* ctx, float* C, long ldc, int rp, const float* Vp, int nc, long ldv, float* W_out, long ldw, float* s_out,
                         float* Vt_out, long ldvt, void* ws, size_t ws_bytes);
long pmd_gram_mtgm_ld_impl(int rows);
size_t pmd_gram_mtgm_workspace_bytes_impl(int rows, int m);
int pmd_gram_mtgm_impl(pmd_ctx* ctx, const float* M, int rows, int m, long ldm, const float* GM, long ldgm, float* C,
                       long ldc, void* ws, size_t ws_bytes);
size_t pmd_chol_inverse_workspace_bytes_impl(int m);
int pmd_chol_inverse_impl(pmd_ctx* ctx, float* C, int m, long ldc, int abs_last_pivot, int* ok_host, void* ws, size_t ws_bytes);
int pmd_transpose_impl(pmd_ctx* ctx, const float* src, long lds_, int rows, int cols, float* dst, long ldd);
int pmd_orthogonalize_chol_impl(pmd_ctx* ctx, const float* M, int Rc, int m, long ldm, const float* GM, long ldgm,
                                float* Et_out, long lde, int* ok_host, void* ws, size_t ws_bytes);

// residual windows (single_residual_block_md)
int pmd_launch_bin_average(pmd_ctx* ctx, const float* X, long ldx, long n_rows, int a, int nbins, float* xbar, long ldb);
int pmd_launch_tile_cross_gram(pmd_ctx* ctx, const float* A, const float* B, long tile_stride, int ld, int len,
                               double* G, int n_tiles);
int pmd_launch_tile_residual_rows(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int d, const float* E,
                                  int e_ld, const float* W, long w_ld, int r, int len, float* out, long out_ld,
                                  int n_tiles, int rp = PMD_RPAD);
int pmd_launch_tile_sub(pmd_ctx* ctx, float* a, const float* b, long tile_stride, int ld, int len, int n_tiles, int rp = PMD_RPAD);
int pmd_launch_tile_append(pmd_ctx* ctx, const float* stats, int r, float thr_s, float thr_t, int max_fail, int cap,
                           const float* Unew, float* Ucur, int ld, int* counts, int* good, int* keep, int n_tiles,
                           int rp = PMD_RPAD);
int pmd_launch_tile_truncate(pmd_ctx* ctx, float* U, int ld, const int* counts, int n_tiles, int rp = PMD_RPAD);
size_t pmd_tiles_residual_workspace_bytes_impl(int n, int d, int r, int a, int L, long n_rows);
int pmd_tiles_residual_impl(pmd_ctx* ctx, const float* Xw, long ldx, long n_rows, int L, const int* tile_pix, int n,
                            int b1, int b2, int r, int a, float thr_s, float thr_t, int max_fail, uint64_t seed,
                            uint32_t omega_index0, uint32_t omega_index_step, float* Ucur, int* counts, float* stats_out,
                            int* good_out, int* keep_out, void* ws, size_t ws_bytes);
