/*
 * libpmd_hip.so -- C ABI of the MI355X (gfx950) blockwise-PMD hot path.
 *
 * The reference (apasarkar/localmd @ 2025-01-31) has no FFI: its hot path is @jit-ed JAX
 * behind plain Python callables.  Each entry point below replaces the jit region(s) cited
 * next to it (paths under /root/reference/localmd/); INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add to bind them.
 *
 * Conventions
 *   - every function returns int32: 0 = OK, negative = error; pmd_last_error(ctx) has the text
 *   - all pointers are DEVICE pointers owned by the caller unless the name ends in _host
 *   - all work is enqueued on the context's HIP stream; functions that must read a result on
 *     the host (pmd_orthogonalize, pmd_projected_svd) synchronise that stream themselves
 *   - scratch memory is a caller-provided workspace; *_workspace_bytes() sizes it
 *   - one context per (host thread, device); distinct contexts are independent
 *
 * Layouts
 *   movie        Y[t][c]            frames-first, c = i*d2 + j (what the dataset hands over)
 *   pixel-major  X[c][t]            leading dimension ld = pmd_time_ld(T); zero padded
 *   tile pixels  pix[tile][q]       q = il + b1*jl (column-major inside the tile, as the
 *                                   reference's order="F" reshapes), value = FOV pixel c
 *   tile basis   Ut[tile][comp][q]  rp = pmd_tile_rpad(max_components) component rows (64 up to max_components = 54;
 *                                   rows >= rank are zero), row length pmd_tile_dpad(b1*b2)
 *   tile traces  V[tile][comp][t]   rp component rows, leading dimension ldv >= pmd_time_ld(T)
 *   The global-stage entry points (pmd_weight_tiles, pmd_gram_*, pmd_compact_rows, pmd_tiles_project) take blocks of 64
 *   component rows: an array with rp = 64 v rows is handed to them as v "virtual tiles" per tile, [n*v][64][x], same memory.
 */
#ifndef PMD_HIP_H
#define PMD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pmd_ctx pmd_ctx;

int pmd_version(void);
/* hip_stream: a hipStream_t (0 = the null stream).  The context never owns the stream. */
int pmd_ctx_create(int device, void* hip_stream, pmd_ctx** out);
int pmd_ctx_destroy(pmd_ctx* ctx);
int pmd_ctx_set_stream(pmd_ctx* ctx, void* hip_stream);
int pmd_ctx_sync(pmd_ctx* ctx);
const char* pmd_last_error(pmd_ctx* ctx);

/* Measurement: HIP events on the context's stream around every kernel group (bench.py's
 * roofline figures).  pmd_profile_enable(ctx, 1) resets and starts; pmd_profile_query sums the
 * durations of the groups called `name`; pmd_profile_names lists the names seen. */
int pmd_profile_enable(pmd_ctx* ctx, int on);
int pmd_profile_query(pmd_ctx* ctx, const char* name, double* total_ms, int* count);
int pmd_profile_names(pmd_ctx* ctx, char* buf, int cap);

/* padded sizes every caller needs to allocate buffers */
int pmd_tile_dpad(int d);       /* padded pixel count of a d-pixel tile (-1: unsupported)   */
int pmd_tile_rpad(int r);       /* component rows of the per-tile arrays for max_components = r: 64 while r + 10 <= 64
                                   (the MFMA-tiled main path), round_up(r + 10, 64) beyond (generic-width kernels) */
long pmd_time_ld(long t);       /* leading dimension of a time-contiguous row of t frames   */

/* Gaussian matrices (replaces jax.random.normal: decomposition.py:62,:127,:870; pmd_loader.py:56).
 * Logical array (stream, index0 + b*index_step), rows x cols, element e = row*cols + col, is
 * written to out[b*batch_stride + row*ld + col] (transpose=0) or [.. + col*ld + row] (1). */
int pmd_rng_normal(pmd_ctx* ctx, uint64_t seed, uint32_t stream, uint32_t index0, uint32_t index_step, int batch,
                   long rows, int cols, int transpose, float* out, long ld, long batch_stride);

/* A1: per-pixel mean and Welch noise sigma (pmd_loader.py:203-291; preprocessing_utils.py:10-40). */
size_t pmd_stats_workspace_bytes(int T, long D, int frame_const);
int pmd_stats(pmd_ctx* ctx, const float* movie, int T, long D, int frame_const, int compute_normalizer,
              float* mean_out, float* std_out, void* ws, size_t ws_bytes);

/* (Y - mean)/std of selected frames, transposed to pixel-major (pmd_loader.py:293-298, :376-377,
 * :396-397).  frames: device int32[nf] or NULL for frames 0..nf-1.  out: D rows x ld. */
int pmd_standardize_transpose(pmd_ctx* ctx, const float* movie, long D, const int* frames, int nf, const float* mean,
                              const float* std, float* out, long ld);

/* A2: background basis = rank-K rSVD of the standardised sample (pmd_loader.py:46-68, :300-314).
 * xs: pixel-major sample with round_up(D,1024) rows allocated (rows >= D zero). basis_out[c][k]. */
size_t pmd_background_rsvd_workspace_bytes(long D, int n, int K);
int pmd_background_rsvd(pmd_ctx* ctx, const float* xs, long D, int n, long ld, int K, uint64_t seed, float* basis_out,
                        void* ws, size_t ws_bytes);

/* A5: temporal projection B^T X (pmd_loader.py:386) and X - B (B^T X) (:387). */
size_t pmd_bg_project_workspace_bytes(long D, int T);
int pmd_bg_project(pmd_ctx* ctx, const float* xs, long D, int T, long ld, const float* basis, int K, float* pj_out,
                   long ldp, void* ws, size_t ws_bytes);
/* pmd_bg_filter: xf_out == xs (in place) is allowed - the pass is element-wise per (pixel, frame). */
int pmd_bg_filter(pmd_ctx* ctx, const float* xs, float* xf_out, long D, int nf, long ld, const float* basis, int K,
                  const float* pj, long ldp);
/* pixel_weighting (decomposition.py:717-718) */
int pmd_scale_rows(pmd_ctx* ctx, float* x, long D, int nf, long ld, const float* w);

/* A4: threshold simulation (decomposition.py:76-131, :147-181).  stats_out[iter][0..1] =
 * (spatial, temporal) roughness of the rank-1 rSVD of an N(0,1) tile; the percentile is host work. */
size_t pmd_threshold_sim_workspace_bytes(int b1, int b2, int t, int iters);
int pmd_threshold_sim(pmd_ctx* ctx, int b1, int b2, int t, int iters, uint64_t seed, float* stats_out, void* ws,
                      size_t ws_bytes);

/* A6-A12: per-tile decomposition, one window (decomposition.py:235-330 single_block_md,
 * evaluation.py:84-222, decomposition.py:501-523).  Omega of tile b is logical array
 * (PMD_STREAM_TILE_OMEGA = 4, omega_index0 + b*omega_index_step).
 * pool_q[P][pool_max]: local pixels of pooling window p (-1 pads); pool_idx[q]: window of pixel q;
 * pool_w[q] = 1/|window|.  n_rows = pixel rows of xf.  Outputs, with rp = pmd_tile_rpad(r): Ut_out[n][rp][dpad],
 * V_out[n][rp][ldv] (= sigma*V rows), stats_out[n][rp][2], good_out/keep_out[n][rp], ranks_out[n], lam_out[n][rp]
 * (sigma^2, may be NULL).  r is unbounded as in the reference (decomposition.py:643-665); r + 10 > 64 runs the
 * generic-width kernels of wide.hip. */
size_t pmd_tiles_workspace_bytes(int n_tiles, int b1, int b2, int P, int r, int a, int t_crop, long ldv, long n_rows);
int pmd_tiles_decompose(pmd_ctx* ctx, const float* xf, long ldx, long n_rows, int t_crop, const int* tile_pix, int n_tiles, int b1,
                        int b2, const int* pool_q, int pool_max, int P, const int* pool_idx, const float* pool_w, int r,
                        int a, float thr_s, float thr_t, int max_fail, uint64_t seed, uint32_t omega_index0,
                        uint32_t omega_index_step, float* Ut_out, float* V_out, long ldv, float* stats_out,
                        int* good_out, int* keep_out, int* ranks_out, double* lam_out, void* ws, size_t ws_bytes);
/* The same pipeline in three resumable parts, for the reference's denoiser hooks (arbitrary callables inside
 * single_block_md).  stages is a mask: bit 0 = up to V_ds = U_ds^T X_ds (decomposition.py:279-298); the
 * temporal_denoiser (:300) then acts on V_ds[n][rp][ldv] (rows < r, t_crop frames) at byte offset *vds_offset of
 * the workspace; bit 1 = basis of its row space and S = X V_b^T (:301-306); the spatial_denoiser (:310) acts on
 * S[n][rp][dpad] (row c = component c, tile pixel il + b1*jl) at *s_offset; bit 2 = the rest (:315-328 and the
 * keep/discard scan).  The workspace must be left untouched between the calls except for the hook arrays. */
int pmd_tiles_decompose_staged(pmd_ctx* ctx, const float* xf, long ldx, long n_rows, int t_crop, const int* tile_pix, int n_tiles,
                               int b1, int b2, const int* pool_q, int pool_max, int P, const int* pool_idx,
                               const float* pool_w, int r, int a, float thr_s, float thr_t, int max_fail, uint64_t seed,
                               uint32_t omega_index0, uint32_t omega_index_step, float* Ut_out, float* V_out, long ldv,
                               float* stats_out, int* good_out, int* keep_out, int* ranks_out, double* lam_out, void* ws,
                               size_t ws_bytes, int stages);
int pmd_tiles_hook_offsets(int n_tiles, int b1, int b2, int P, int r, int a, int t_crop, long ldv, long n_rows,
                           size_t* vds_offset, size_t* s_offset);

/* A9: one further temporal window (decomposition.py:333-387 single_residual_block_md, :489-515): xw = the
 * window's frames (pixel-major, L frames); Ucur[n][rp][dpad] (rp = pmd_tile_rpad(r)) holds counts[tile] components (other rows zero)
 * and receives the kept components of the residual fit; counts is updated.  pmd_tiles_truncate clears the
 * rows >= counts[tile] of a basis array (after the first window). */
size_t pmd_tiles_residual_workspace_bytes(int n_tiles, int b1, int b2, int r, int a, int L, long n_rows);
int pmd_tiles_residual(pmd_ctx* ctx, const float* xw, long ldx, long n_rows, int L, const int* tile_pix, int n_tiles,
                       int b1, int b2, int r, int a, float thr_s, float thr_t, int max_fail, uint64_t seed,
                       uint32_t omega_index0, uint32_t omega_index_step, float* Ucur, int* counts, float* stats_out,
                       int* good_out, int* keep_out, void* ws, size_t ws_bytes);
int pmd_tiles_truncate(pmd_ctx* ctx, float* U, int dpad, const int* counts, int n_tiles, int rpad);

/* A13: weight and normalise tile bases: Uw = Ut * w[q] / cumw[pix]  (decomposition.py:812-853). */
int pmd_weight_tiles(pmd_ctx* ctx, const float* Ut, int dpad, const int* tile_pix, int d, const float* w,
                     const float* cumw, const int* ranks, float* Uw_out, int n_tiles);

/* A12/A16: Out[tile] = A[tile] X[tile pixels]  (get_temporal_projector decomposition.py:390-407;
 * the sparse product of pmd_loader.py:411).  A[tile][64][dpad]; Out[tile][64][ldo]. */
int pmd_tiles_project(pmd_ctx* ctx, const float* x, long ldx, int T, const int* tile_pix, int n_tiles, int d,
                      const float* A, int dpad, float* Out, long ldo, int slices);
/* The same when rows >= ranks[tile] of A are zero (the weighted, rank-masked bases of the movie projection): a tile that kept
 * <= 32 components runs half the MFMA work, and rows >= 32 of its Out block are then NOT written (pmd_compact_rows reads
 * rows < rank only). */
int pmd_tiles_project_ranked(pmd_ctx* ctx, const float* x, long ldx, int T, const int* tile_pix, int n_tiles, int d,
                             const float* A, int dpad, float* Out, long ldo, int slices, const int* ranks);
/* Z[col_off[tile] + c][t] = Out[tile][c][t], c < ranks[tile] */
int pmd_compact_rows(pmd_ctx* ctx, const float* Out, long ldo, const int* col_off, const int* ranks, int T, float* Z,
                     long ldz, int n_tiles);

/* A15: G = U^T U for the block-sparse U (decomposition.py:974); pairs[p] = (tile a, tile b, i0, i1,
 * j0, j1) overlap rectangles, origins[tile] = (k, j).  G is (Rt+K) x (Rt+K), row-major, ldg. */
int pmd_gram_u(pmd_ctx* ctx, const float* Uw, int dpad, int b1, int b2, const int* tile_pix, const int* pairs,
               int n_pairs, const int* origins, const int* col_off, const int* ranks, int n_tiles, int Rt,
               const float* basis, long D, int K, float* G, long ldg);
/* RCCL at this boundary, for a caller that is not built on torch.distributed: one communicator per context,
 * collectives enqueued on the context's stream.  pmd_comm_unique_id: 128 bytes from rank 0, to be broadcast out of band
 * (ncclGetUniqueId); pmd_comm_init: ncclCommInitRank on the context's device; the sharded path needs an in-place fp32
 * sum (partial M^T G M / M^T Z / background projections) and an all-gather of equal-sized blocks
 * (per-tile results).  RCCL is resolved with dlopen at the first call: no link-time dependency, and a process that
 * already holds an RCCL (PyTorch's) shares it.  localmd_amd/parallel.py uses torch.distributed (backend "nccl" = RCCL). */
int pmd_comm_unique_id(void* out128);
int pmd_comm_init(pmd_ctx* ctx, const void* unique_id128, int rank, int world);
int pmd_comm_destroy(pmd_ctx* ctx);
int pmd_comm_all_reduce_f32(pmd_ctx* ctx, float* buf, size_t count);
int pmd_comm_all_gather(pmd_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank);
/* Which eigen-directions pmd_orthogonalize / pmd_orthogonalize_factored keep.  rel_cutoff < 0 (default): the
 * reference's rule - jnp.linalg.svd(hermitian=True) returns |lambda| and u = v sign(lambda), so `eig_vals > 0`
 * (decomposition.py:988) keeps every direction whose eigenvalue is not exactly zero, numerically null ones included.
 * rel_cutoff >= 0: keep lambda > rel_cutoff * lambda_max only ("expose a relative cutoff option"). */
int pmd_ctx_set_null_cutoff(pmd_ctx* ctx, float rel_cutoff);
/* A15: P with (U P)^T (U P) = I (decomposition.py:976-999, only_left=True).  M = right matrix
 * (R x m, NULL = identity, then m = R).  G is overwritten.  *rprime_host = columns kept. */
size_t pmd_orthogonalize_workspace_bytes(int R, int m, int has_m);
int pmd_orthogonalize(pmd_ctx* ctx, float* G, int R, const float* M, int m, long ldm, float* P_out, long ldp,
                      int* rprime_host, void* ws, size_t ws_bytes);
/* A17: projected_svd (decomposition.py:1013-1137).  V: n1 x n2.  nk = min(n1, n2). */
size_t pmd_projected_svd_workspace_bytes(int rows_p, int n1, int n2);
int pmd_projected_svd(pmd_ctx* ctx, const float* P, int rows_p, long ldp, const float* V, int n1, int n2, long ldv,
                      float* R_out, long ldr, float* s_out, float* Vt_out, long ldvt, void* ws, size_t ws_bytes);
/* Block-sparse form of the same Gram matrix, used when R > frames (right_mat = v, decomposition.py:976-981):
 * Gblk[pair][64][64], Gbg[kb][tile][64][64] (tile x block kb of 64 background columns; ceil(K / 64) blocks), Gstrip[K][ldgs]
 * (background rows of G).
 * pmd_gram_apply: GM = G M without densifying G.  nbr_ptr[n_tiles+1], nbr[e] = (first M row of the
 * block, rows in it, block index, flags: bit0 transposed, bit1 background block). */
int pmd_gram_blocks(pmd_ctx* ctx, const float* Uw, int dpad, int b1, int b2, const int* tile_pix, const int* pairs,
                    int n_pairs, const int* origins, const int* col_off, const int* ranks, int n_tiles, int Rt,
                    const float* basis, long D, int K, float* Gblk, float* Gbg, float* Gstrip, long ldgs);
int pmd_gram_apply(pmd_ctx* ctx, const float* Gblk, const float* Gbg, const float* Gstrip, long ldgs, const int* nbr_ptr,
                   const int* nbr, const int* col_off, const int* ranks, int n_tiles, int Rt, int K, int max_rank,
                   const float* M, long ldm, int ncols, float* GM, long ldgm);
/* A15/A16/A17 with P = M E^T kept factored (R > frames): Et rows = eigenvectors of M^T G M / sqrt(lambda)
 * (|lambda| descending; which directions are kept: pmd_ctx_set_null_cutoff); then V = Et (M^T Z), its SVD, and
 * R_out = M (Et^T W). */
size_t pmd_orthogonalize_factored_workspace_bytes(int m);
int pmd_orthogonalize_factored(pmd_ctx* ctx, const float* M, int Rc, int m, long ldm, const float* GM, long ldgm,
                               float* Et_out, long lde, int* rprime_host, void* ws, size_t ws_bytes);
/* Cholesky variant of the same step: Et = U_c^{-T} with M^T G M = U_c^T U_c.  P = M Et^T differs from the
 * eigenvector form by an orthogonal factor that the projected SVD absorbs, so R, s, Vt are unchanged.
 * *ok_host = 0 if the matrix is not numerically positive definite (use pmd_orthogonalize_factored then). */
int pmd_orthogonalize_chol(pmd_ctx* ctx, const float* M, int Rc, int m, long ldm, const float* GM, long ldgm,
                           float* Et_out, long lde, int* ok_host, void* ws, size_t ws_bytes);
size_t pmd_orthogonalize_chol_workspace_bytes(int Rc, int m);
/* The frames x frames part of pmd_projected_svd_factored split by frame columns (multi-GPU: every rank owns T / N columns of
 * W1 = M^T Z; decomposition.py:1089-1099 is the reference step).  pmd_psvd_vp_gram: Vp (rp x nc) = Et (rp x m) W1[:, c0:c0+nc)
 * (W1 points at the first owned column, leading dimension ldw; et_lower: Et is lower triangular) and C = Vp Vp^T (rp x rp,
 * ldc >= round_up(rp, 4), row-major upper triangle) - the caller sums C over the ranks (all-reduce).  pmd_psvd_finish:
 * eigendecomposition of the summed C (every rank the same), s_out (rp), W_out (rp x rp), Vt_out (rp x nc) = this rank's columns
 * of W^T Vp / s.  One rank with nc = T reproduces pmd_projected_svd_factored. */
int pmd_psvd_vp_gram(pmd_ctx* ctx, const float* Et, int rp, int m, long lde, const float* W1, int nc, long ldw, int et_lower, float* Vp,
                     long ldv, float* C, long ldc);
size_t pmd_psvd_finish_workspace_bytes(int rp);
int pmd_psvd_finish(pmd_ctx* ctx, float* C, long ldc, int rp, const float* Vp, int nc, long ldv, float* W_out, long ldw, float* s_out,
                    float* Vt_out, long ldvt, void* ws, size_t ws_bytes);
size_t pmd_projected_svd_factored_workspace_bytes(int Rc, int m, int rp, int T);
int pmd_projected_svd_factored(pmd_ctx* ctx, const float* M, int Rc, int m, long ldm, const float* Et, int rp, long lde,
                               const float* Z, int T, long ldz, float* R_out, long ldr, float* s_out, float* Vt_out,
                               long ldvt, float* Vp_out, long ldvp, float* X1_out, const float* W1_in, int et_lower,
                               void* ws, size_t ws_bytes);
/* (X1_out: optional m x rp, ld rp, receives Et^T W.  With R_out == NULL the last product R = M X1 is left to the
 * caller, who can then form R in row blocks with pmd_gemm and overlap their download with the next block.
 * W1_in: optional m x T, ld T: M^T Z formed by the caller, e.g. the all-reduced sum of per-rank row-range partials.
 * et_lower != 0: Et is square lower triangular (pmd_orthogonalize_chol / pmd_chol_inverse), strmm replaces two GEMMs.) */
/* The two halves of pmd_orthogonalize_chol, for callers that shard the rows of M over ranks: the partial
 * C = M[rows]^T GM[rows] (row-major lower block triangle; all-reduce it), then C -> Et in place. */
size_t pmd_gram_mtgm_workspace_bytes(int rows, int m);
/* leading dimension of the transposed copy M^T (m x rows) that pmd_gram_mtgm leaves at the start of its workspace (a multiple
 * of 64 floats; the host driver reuses the copy for the M^T Z product) */
long pmd_gram_mtgm_ld(int rows);
int pmd_gram_mtgm(pmd_ctx* ctx, const float* M, int rows, int m, long ldm, const float* GM, long ldgm, float* C, long ldc,
                  void* ws, size_t ws_bytes);
size_t pmd_chol_inverse_workspace_bytes(int m);
/* abs_last_pivot != 0: the caller has rotated a known null direction of C into the last row / column; a negative last
 * pivot is then replaced by its absolute value instead of failing - the Cholesky-route form of the reference keeping
 * numerically null directions through |lambda| (svd(hermitian=True) at decomposition.py:984, `eig_vals > 0` at :988). */
int pmd_chol_inverse(pmd_ctx* ctx, float* C, int m, long ldc, int abs_last_pivot, int* ok_host, void* ws, size_t ws_bytes);
/* dst (cols x rows, ld_dst) = src (rows x cols, ld_src)^T, row-major */
int pmd_transpose(pmd_ctx* ctx, const float* src, long ld_src, int rows, int cols, float* dst, long ld_dst);
/* A13/A14: CSR arrays of the sparse spatial matrix built on the device (decomposition.py:812-853, :929-930);
 * cover1[d1][4] / cover2[d2][4]: indices of the tile-row / tile-column origins covering each FOV row / column. */
int pmd_csr_count(pmd_ctx* ctx, int d1, int d2, int order_f, const int* cover1, const int* cover2, int n2,
                  const int* ranks, int K, long* row_nnz);
int pmd_csr_fill(pmd_ctx* ctx, int d1, int d2, int order_f, int b1, const int* cover1, const int* cover2,
                 const int* orig1, const int* orig2, int n2, const int* ranks, const int* col_off, const float* Ut,
                 int dpad, const float* w, const double* inv_cumw, const float* basis, int K, int Rt, const long* indptr,
                 double* data, int* indices, int* zero_count, int rpad);   /* rpad: component rows of Ut (pmd_tile_rpad) */
/* row-major C = alpha op(A) op(B) + beta C (the jnp.matmul calls of decomposition.py:873, :982, :993,
 * :1006; pmd_loader.py:412) */
int pmd_gemm(pmd_ctx* ctx, int transA, int transB, int m, int n, int k, float alpha, const float* A, long lda,
             const float* B, long ldb, float beta, float* C, long ldc);
/* 1 when pmd_gemm runs a product of this size as three fp16-piece products on the fp16 matrix cores (csrc/gemm_f16x2.hip:
 * fp32 operands and result, error below the sgemm path's; PMD_GEMM_SPLIT=0 switches it off).  Those products accumulate
 * into C, so C should then live in HBM (a host-mapped C takes the sgemm path). */
int pmd_gemm_split_active(pmd_ctx* ctx, int m, int n, int k);
/* Frees the library-owned device scratch of the large products (fp16 pieces, split-K partial sums) when it is larger than
 * keep_bytes; it is re-created on demand.  The host driver calls it at the end of every decomposition. */
int pmd_scratch_trim(pmd_ctx* ctx, size_t keep_bytes);

/* A18: expansion back into pixels (pmdarray.py:132-171, PMDArray.__getitem__: spatial.dot(temporal), times the
 * noise image, plus the mean image, frames first).  pmd_csr_rows_spmm: out[p][f] = sum over the nonzeros i of CSR
 * row rows[p] (rows == NULL: row p) of data[i] * B[indices[i]][f], f < ncols; B is the dense (R diag(s)) Vt[:, frames]
 * or R diag(s).  pmd_transpose_affine: dst[c][r] = src[r][c] * scale[r] + shift[r] (scale / shift may be NULL). */
int pmd_csr_rows_spmm(pmd_ctx* ctx, const int64_t* indptr, const int* indices, const float* data, const int* rows, long n_sel,
                      const float* B, long ldb, int ncols, float* out, long ldo);
/* SURVEY 8(f)4 - the local correlation images of /root/reference/localmd/diagnostic_plots.py (their O(D) Python loops of
 * jitted per-pair calls, :131-156, :195-221, :247-268, :293-302) from first and second moments of the pixel traces.
 * Movies are frames first (T, d1, d2) float32 on the device; ref[D] = one frame of the movie (the traces are shifted by it,
 * covariances do not change).  pmd_neighbour_moments: moments[10][D] (+)= {sum x, sum x^2, sum x_p x_q for the 8 neighbours
 * q in row-major (di, dj) order} of x = A - B (B may be NULL); call it once per resident chunk of frames with accumulate = 1
 * after the first.  pmd_neighbour_image: kind 0 = Pearson correlation (make_correlation_image), kind 1 = cov(ddof 1) of the
 * numerator movie over sqrt(var var) (ddof 0) of the normalising movie `den` = its moments[0..1] (make_pmd_correlation_image,
 * make_residual_correlation_image); mode 0 = max over the existing neighbours starting from 0, mode 1 = their mean.
 * pmd_lag_moments / pmd_lag_image: make_autocorrelation_image; moments[5][D] (+)= sums over the pairs (t, t - lag), t in
 * [lag, T) of the resident frames (a chunk therefore starts `lag` frames before its first new frame); n = T_total - lag. */
size_t pmd_diag_workspace_bytes(long T, long D);
int pmd_neighbour_moments(pmd_ctx* ctx, const float* A, const float* B, const float* ref, long T, int d1, int d2, int accumulate,
                          double* moments, void* ws, size_t ws_bytes);
int pmd_lag_moments(pmd_ctx* ctx, const float* A, const float* ref, long T, long D, int lag, int accumulate, double* moments,
                    void* ws, size_t ws_bytes);
int pmd_neighbour_image(pmd_ctx* ctx, const double* num, const double* den, long T, int d1, int d2, int kind, int mode, double* out);
int pmd_lag_image(pmd_ctx* ctx, const double* moments, long D, long n, double* out);
int pmd_transpose_affine(pmd_ctx* ctx, const float* src, long lds_, long rows, int cols, const float* scale,
                         const float* shift, float* dst, long ldd);

/* ---- kernel-level entry points (used by the parity tests; same kernels as above) ---------- */
int pmdk_tile_atx(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                  const float* A, long a_tile_stride, int a_ld, float* Out, long out_tile_stride, long ldo, int n_tiles,
                  int T, int slices);
/* the same with the number of rows of A that carry data (rows >= `rows` are zero): <= 16 rows on 1024-pixel tiles and 33-50 rows
 * on 257-400-pixel tiles run specialised kernels that leave the output rows beyond the last 16-row tile they touch unwritten */
int pmdk_tile_atx_rows(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                       const float* A, long a_tile_stride, int a_ld, float* Out, long out_tile_stride, long ldo, int n_tiles,
                       int T, int slices, int rows);
int pmdk_tile_xbt(pmd_ctx* ctx, const float* X, long ldx, const int* pix, int pix_stride, long row0_stride, int d,
                  const float* B, long b_tile_stride, long ldb, float* S, long s_tile_stride, long s_slice_stride,
                  int s_ld, int n_tiles, int T, int slices);
int pmdk_tile_gram(pmd_ctx* ctx, const float* In, long tile_stride, long ld, int len, int n_tiles, int slices,
                   double* G);
int pmdk_tile_rowmix(pmd_ctx* ctx, const float* In, long in_tile_stride, long ld_in, const double* N,
                     long n_tile_stride, int n_in, int n_out, float* Out, long out_tile_stride, long ld_out, int len,
                     int n_tiles);
int pmdk_small_qr(pmd_ctx* ctx, const float* Yt, long y_tile_stride, int y_ld, int P, int l, float* Qt,
                  long q_tile_stride, int q_ld, int n_tiles);
int pmdk_small_eig(pmd_ctx* ctx, const double* G, int slices, int n, int mode, double tol, double* Nout,
                   double* lam_out, int n_tiles);
int pmdk_tile_pool_bin(pmd_ctx* ctx, const float* X, long ldx, long n_rows, const int* pix, int n_tiles, int d,
                       const int* pool_q, int pool_max, int P, int a, int nbins, float* xbar, float* abar, long ld_ab,
                       long tile_stride);
int pmdk_roughness(pmd_ctx* ctx, const float* Ut, long u_tile_stride, int u_ld, int b1, int b2, const float* V,
                   long v_tile_stride, long v_ld, int T, int r, float* stats, int n_tiles);
int pmdk_syevd(pmd_ctx* ctx, int n, float* A, long lda, float* w, float* work, int* info);
/* Householder tridiagonalisation A = Q T Q^T with LAPACK ssytrd('L') conventions on the column-major view
 * of the buffer (memory row c holds A(r, c), r >= c, at offset r): d[n], e[n-1], tau[n-1], reflectors in A.
 * impl 0 = rocSOLVER, 1 = the library's own kernels (lda % 4 == 0, lda >= round_up(n, 4)). */
int pmdk_sytrd(pmd_ctx* ctx, int n, float* A, long lda, float* d, float* e, float* tau, int impl);
/* stage 1 of the two-stage reduction (sytrd2.hip): dense -> band of half bandwidth 64; reflectors of Q1 below the band
 * (unit entry of reflector c at position c + 64), tau1[n]; *flag_host != 0: a panel was rank deficient */
int pmdk_sy2sb(pmd_ctx* ctx, int n, float* A, long lda, float* tau1, int* flag_host);
/* stages 1 + 2: dense -> band -> tridiagonal (bulge chasing); d[n], e[n - 1] on the device */
int pmdk_sytrd2(pmd_ctx* ctx, int n, float* A, long lda, float* tau1, float* d, float* e, int* flag_host);

#ifdef __cplusplus
}
#endif
#endif
