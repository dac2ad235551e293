// ck_internal.h -- launch wrappers shared between the translation units of
// libcokrige_hip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ck_host.h"
#include "ck_math.h"

#define CK_NB 512   // outer block column width (panel width)
#define CK_IB 64    // inner block (diagonal factor / row solves)
#define CK_BM 256   // GEMM tile rows
#define CK_AUX_ALIGN 256
// every packed panel carries, behind its rows, the inverses of its eight 64 x 64 diagonal blocks
// (k_potrf64 -> k_trsm64m); they travel with the panel in the multi-GPU broadcast
#define CK_PANEL_TAIL ((CK_NB / CK_IB) * CK_IB * CK_IB)

// ---- covariance assembly (ck_cov.hip) ---------------------------------------
// per-site transform: degrees -> (lat_rad, lon_rad, cos lat) | (x, y, 0)
void ck_launch_prep_sites(hipStream_t s, const double* coords, int64_t n, int metric, double* c0, double* c1,
                          double* c2, double* u /* 3 x n chord vectors, may be null */);
// tabulated fast path (ck_math.h "Tabulated correlation")
void ck_launch_table_nodes(hipStream_t s, const CkMatern* m, int metric, const double* q, int64_t n, double* out);
void ck_launch_table_check(hipStream_t s, const CkMatern* m, int metric, CkTable tab, const double* coef,
                           unsigned long long* max_err_bits);
// Internal site order: process 0 in [0, n0), process 1 in [n0p, nend), n0p = roundup(n0, 64);
// every other index below npad is padding (identity in Sigma, zero in the right-hand sides).
struct CkLayout {
    long n0, n0p, nend, npad;
};
// entries the table kernels defer to the exact evaluator: (row, col) pairs + device counter
// Two counters, used alternately (round 4): the pass that evaluates one assembly's list (k_assemble_fix) zeroes the counter
// of the NEXT assembly, so that no assembly starts with a memset launch of its own.
struct CkWorklist {
    int2* items;
    unsigned* count;   // count[0]: deferred entries; count[1]: work queue of the assembly kernel (option "assemble_queue")
    unsigned cap;
    unsigned* reset;   // the other pair of counters (may be null)
};
// which panels one assembly launch covers: Sigma -- the owned panels (tile0[j] = index of the first
// 64-row tile of the j-th owned panel, panel_of[j] = its block column, sigptr[K] = its storage);
// right-hand sides -- n_panels panels of aux_tiles tiles each, contiguous from `aux`
struct CkPanelMap {
    const int* tile0;
    const int* panel_of;
    double* const* sigptr;
    int n_panels;
    double* aux;
    long aux_tiles;
    const int* order;   // Sigma, work-queue form: the strips sorted by Matern block (may be null: back to front)
};
// Sigma: every owned block column (rows K*NB.., ld = CK_NB) in ONE launch.
// c: 3 x npad exact-formula coordinates, u: 3 x npad chord vectors.  fast: table path.
void ck_launch_assemble_sigma(hipStream_t s, bool fast, const CkMatern* blk, const CkTable* tabs,
                              const double* const* coefs, int metric, const double* c, const double* u, CkLayout L,
                              CkPanelMap pm, int total_tiles, CkWorklist wl, int queue_slots = 0 /* > 0 (table path): that many
                              resident workgroups take their strips from a work queue (the word behind wl.count, zero at launch) */);
// right-hand-side rows, every block column in one launch: rows = prediction sites p in [0, m)
// (row m = data values z, rows > m zero), cols = data sites.
// raw (table path only; may be null): the prediction sites' coordinates as the caller gave them (m x 2, padded with zeros
// to mpad).  The launch then transforms them itself -- every workgroup its own 64 rows, the workgroups of block column 0
// also write pc / pu for the later users (exact pass, _verify_model) -- instead of waiting for a k_prep_sites launch.
void ck_launch_assemble_aux(hipStream_t s, bool fast, const CkMatern* blk, const CkTable* tabs,
                            const double* const* coefs, int metric, int i_pred, double* pc, double* pu,
                            int64_t m, int64_t mpad, const double* c, const double* u, const double* z, CkLayout L,
                            int n_panels, double* aux, CkWorklist wl, const double* raw = nullptr, int queue_slots = 0);
// evaluate the deferred entries of the preceding table-path launches (no-op when the list is empty)
void ck_launch_assemble_fix(hipStream_t s, bool aux_rows, const CkMatern* blk, int metric, int i_pred,
                            const double* pc, int64_t mpad, const double* c, CkLayout L, CkWorklist wl,
                            double* const* sigptr, double* aux);
// dense a x b block for one (i, j) Matern block; mode 0 = covariance, 1 = distance only
void ck_launch_cov_dense(hipStream_t s, const CkMatern* blk_ij, int metric, int add_nugget, int mode,
                         const double* a0, const double* a1, const double* a2, int64_t a, const double* b0,
                         const double* b1, const double* b2, int64_t b, double* out);
void ck_launch_model_variogram(hipStream_t s, const CkMatern* blk, double sill, int kind, const int* pi,
                               const int* pj, const double* lags, int64_t n, double* out);
void ck_launch_cov_lags(hipStream_t s, const CkMatern* blk_ij, int add_nugget, const double* lags, int64_t n,
                        double* out);

// ---- dense linear algebra (ck_la.hip) ----------------------------------------
// C (M x N, ldc) -= A (M x K, lda) * B (N x K, ldb)^T on FP64 MFMA.
// M % 256 == 0, N % 64 == 0, K % 16 == 0.  lower: skip tiles whose rows are all above the
// diagonal  row + diag_off == col.
// batch > 1 repeats the product over blockIdx.y with element strides sC / sA / sB.
void ck_launch_gemm_nt(hipStream_t s, double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                       int64_t ldb, int64_t M, int64_t N, int64_t K, int lower, int64_t diag_off, int batch,
                       int64_t sC, int64_t sA, int64_t sB);
// Cholesky trailing update of every owned block column J = J0 + y * Jstep (y < nJ) by panel K
// (device pointer table sigptr_dev[J], panel P = rows K*NB.. of L).
void ck_launch_cu_probe(hipStream_t s, unsigned* out, int n_wg, int spin);
void ck_launch_syrk_group(hipStream_t s, double* const* sigptr_dev, double* const* srcptr_dev, int K0, int np, int J0,
                          int Jstep, int nJ, int64_t Npad, int64_t nvalid /* rows / columns from here on are identity padding */,
                          unsigned long long* stamps = nullptr /* diagnostic: ck_debug_gemm_clock */);
// the whole panel step (diagonal blocks, inverses, row solves, panel-internal updates) in ONE launch: nrows / 64 workgroups
// that hand each other the pivot blocks through flags[0..7] == seq (ck_la.hip: k_panel_coop); *err != 0: a wait timed out
// X / xrows (round 4): right-hand-side rows of this block column, further workgroups of the same launch (null / 0: none)
void ck_launch_panel_coop(hipStream_t s, double* P, int64_t nrows, double* tail, int64_t g0, long long* info, unsigned* flags,
                          unsigned seq, unsigned* err, double* X = nullptr, int64_t xrows = 0, unsigned spins = 2000000u,
                          int drop = -1 /* tests: this chunk of the diagonal block never publishes */);
// one update of the TALL matrix [Sigma; c0^T; z^T]: block columns J0 .. J0 + nJ - 1 of Sigma and of the mpad right-hand-side
// rows by the panels K0 .. K0 + np - 1, one launch (ck_la.hip: k_tall_group_d)
void ck_launch_tall_group(hipStream_t s, double* const* sigptr_dev, double* aux, int64_t mpad, int K0, int np, int J0, int nJ,
                          int64_t nvalid, int64_t mrows = 0);
// diagnostic: the cooperative panel step with shader-clock stamps of its links 1 .. 7 (prof: 64 words)
void ck_launch_panel_coop_prof(hipStream_t s, double* P, int64_t nrows, double* tail, int64_t g0, long long* info,
                               unsigned* flags, unsigned seq, unsigned* err, long long* prof);
void ck_launch_aux_group(hipStream_t s, double* aux, int64_t mpad, double* const* sigptr_dev, int K0, int np, int J0,
                         int nJ, int64_t mrows, int64_t nvalid, int64_t live_rows = 0);
// S_J -= sum_p aux_p[rows of J..] aux_p[rows of block J]^T for the nJ block columns of the prediction sites' Schur
// complement (ck_verify_model); aux: np block columns of mpad x CK_NB solved right-hand-side rows
void ck_launch_schur_syrk(hipStream_t s, double* const* schur_dev, const double* aux, int64_t mpad, int np, int nJ,
                          int64_t Mpad);
// In-place Cholesky of the 64 x 64 diagonal block at A (ld); info_dev gets global_index0 + j + 1 of
// the first non-positive pivot (only if still 0).
void ck_launch_potrf64(hipStream_t s, double* A, int64_t ld, int64_t global_index0, long long* info_dev,
                       double* Linv);
// diagnostic: the same with shader-clock stamps at its phase boundaries (prof: 16 words)
void ck_launch_potrf64_prof(hipStream_t s, double* A, int64_t ld, long long* info, double* Linv, long long* prof);
// X L^T = A in place for `nrows` rows of A (ld), 64 columns; L (64 x 64 lower, ldl).  nrows % 64 == 0.
// fused panel step (ck_la.hip, option "panel_fused")
void ck_launch_panel_diag(hipStream_t s, double* P, int j, int64_t g0, long long* info, double* Linv);
void ck_launch_panel_rows_all(hipStream_t s, double* X, int64_t nrows, const double* P, const double* tail, double* X2 = nullptr,
                              int64_t nrows2 = 0);
void ck_launch_panel_rows(hipStream_t s, double* X, int64_t row_first, int64_t nrows, const double* P, int j,
                          const double* Linv);
void ck_launch_trsm64(hipStream_t s, double* A, int64_t ld, int64_t nrows, const double* Linv);
// pred[p] = sum_c X[p][c] y[c];  err[p] = nan_to_num(sqrt(c0 - sum_c X[p][c]^2)); X rows live in
// n_panels panels of width CK_NB at aux + K * mpad * CK_NB; y is row `zrow`.
// c0 < 0: raw mode, pred[p] = X_p . y and err[p] = |X_p|^2 (leave-one-out).
void ck_launch_reduce_pred(hipStream_t s, const double* aux, int64_t mpad, int n_panels, int64_t m, int64_t zrow,
                           double c0, double* pred, double* err);
void ck_launch_tri_matvec(hipStream_t s, double* const* sigptr_dev, int64_t npad, const double* v, double* out);
void ck_launch_loo_rows(hipStream_t s, double* aux, int64_t mpad, int64_t m, int64_t g0, const double* z,
                        int64_t npad);
void ck_launch_mfma_probe(hipStream_t s, int32_t* out);
int ck_launch_mfma_peak(hipStream_t s, int blocks, int waves_per_simd, int iters, double* sink);

// ---- empirical variogram (ck_vario.hip) ----------------------------------------------------
#ifndef CK_VG_JSUB
#define CK_VG_JSUB 128   // "j" points of a sub-chunk: the unit of the level-window decision (and of the third set of bounding
                         // balls).  Bin pass at 1 M soundings: 64 -> 85.5 ms, 128 -> 80.9, 256 -> 87.1, 512 -> 122.5 (smaller blocks,
                         // narrower windows, more set-up); at 4 M soundings 128 and 256 are within 2 %
#endif
#ifndef CK_VG_JCHUNK
#define CK_VG_JCHUNK 1024   // "j" points of a wave's pair tile: the unit of the first culling test and of the tile -> wave deal
#endif
#define CK_VG_MAXBINS 60   // levels sit one per lane of a wave (ck_vario.hip); a few lanes of slack for the windows
struct CkVarioExt {
    double rmin, rmax;
    long long imin, jmin, imax, jmax;
};
// CkVarioPair (a pair the kernels leave to the host): ck_host.h
void ck_launch_vario_prep(hipStream_t s, const double* coords, int64_t n, int metric, double* u0, double* u1,
                          double* u2);
int ck_vario_bin_grid(int64_t ni, int64_t nj);   // workgroups of the three pair passes (wave tiles of 64 x 1024 points)
// iu / ju: 3 x n SoA (unit vectors | x, y, 0); part: CkVarioExt[grid]; q = squared chord | squared distance;
// ib64 / jb1024 / jbsub: bounding balls of the 64-point "i" blocks, 1024-point "j" chunks and 128-point sub-chunks
void ck_launch_vario_extent(hipStream_t s, int grid, int same, const double* iu, int64_t ni, const double* ju,
                            int64_t nj, double qcap, void* part, int rank, int world, const double* ib64,
                            const double* jb1024, const double* jbsub, double cmax, unsigned long long* best /* 2 words */,
                            double qwin_lo /* pairs with qwin_lo <= q <= qcap go to the list */, CkVarioPair* list, unsigned* count,
                            unsigned cap);
void ck_launch_vario_collect(hipStream_t s, int grid, int same, const double* iu, int64_t ni, const double* ju,
                             int64_t nj, double qtop_lo, double qcap, double qbot_hi, CkVarioPair* list, unsigned* count,
                             unsigned cap, int rank, int world, const double* ib64, const double* jb1024,
                             const double* jbsub);
// tile culling (ck_vario.hip): bounding balls of blocks of `blk` consecutive points, 4 x nblk doubles
int64_t ck_vario_nblocks(int64_t n, int blk);
void ck_launch_vario_bounds(hipStream_t s, const double* u, int64_t n, int blk, double* out);
// levels 1 .. nlev (xa / xb / dthr indexed by level; x = q (Euclid) or q / 2 - 1 (haversine), see ck_vario.hip);
// counts: CK_VG_MAXBINS + 1 words, the last = pairs visited; args_dev: CK_VG_ARGS_BYTES of device memory
#define CK_VG_ARGS_BYTES 256
void ck_launch_vario_bin(hipStream_t s, int metric, int same, int covariogram, const double* iu, const double* iv,
                         int64_t ni, const double* ju, const double* jv, int64_t nj, int nlev, const double* xa,
                         const double* xb, const double* dthr, double cmax, const double* ib64, const double* jb1024,
                         const double* jbsub, int grid, double* part_sum, unsigned long long* part_cnt, CkVarioPair* list,
                         unsigned* count, unsigned cap, int rank, int world, int nb, double* sums, long long* counts,
                         void* args_dev);

// ---- local-neighbourhood cokriging (ck_local.hip) -------------------------------------------
// pc: 3 x mpad prediction-site coordinates, sc: 3 x npad site coordinates (exact-formula form)
// cb: chunk bounds of the sites (ck_launch_local_chunk_bounds: 4 x ceil(nend / 256) doubles), cmax: largest chord
// (distance in the space of the chord vectors su / pu) a neighbour can have, with its safety margin
void ck_launch_local_chunk_bounds(hipStream_t s, const double* su, CkLayout L, double* cb);
void ck_launch_local_count(hipStream_t s, int metric, int i_pred, int cv, double max_dist, const double* pc,
                           int64_t m, int64_t mpad, const double* sc, CkLayout L, int* counts, const double* cb,
                           double cmax, const double* pu);
// slab_off[p]: offset (doubles) of point p's scratch slab ((k + 2) k doubles + k ints) when its
// neighbourhood exceeds the LDS limit
void ck_launch_local_solve(hipStream_t s, const CkMatern* blk, int metric, int i_pred, int cv, double max_dist,
                           const double* pc, int64_t p_base, int64_t m, int64_t mpad, const double* sc, const double* z,
                           CkLayout L, const int* counts, const long long* slab_off, double* slab, double c0var,
                           double* pred, double* err, const CkTable* tabs, const double* const* coefs, int use_tab,
                           const double* su, const double* pu, int k_hi, const double* cb, double cmax);
int ck_local_lds_limit();

// Large neighbourhoods (k > k_hi above): the "tiled" path.  The systems of a batch are factored TOGETHER,
// 64 columns per step, by launches over all systems that still have columns left (diagonal block + its
// inverse, row solves, trailing update on 128 x 128 MFMA tiles) -- the tile kernels of the joint path with a
// system index in the grid.  A system's scratch (slab + off):
//   S     CK_LT_ROWS(kq) rows x ld doubles, ld = kq + 128, kq = k + 2 rounded up to 64: ONE padded symmetric
//         matrix.  Rows/cols [0, k): local covariance (lower triangle); [k, kq - 2): identity padding; rows
//         kq - 2 and kq - 1: the c and z rows, which ride along as in the other local kernels -- here as two
//         more matrix rows with a huge diagonal (CK_LT_BIG), so that the Cholesky recurrences themselves do
//         their forward substitution (L[r][j] = (S[r][j] - sum) / L[j][j]) and never see a bad pivot there.
//         The 128 rows / columns beyond kq exist only so that whole tiles can be read and written without
//         bounds checks; nothing valid depends on them.
//   Linv  CK_LT_NINV x 64 x 64 doubles (inverses of the diagonal blocks of the current column group)
//   idx   k ints (neighbour list)
struct CkLocalSys {
    long long off;      // doubles into the slab
    int k, kq, ld, p;   // neighbours, padded size, leading dimension, prediction point index
};
#define CK_LT_BIG 1e200
#define CK_LT_ROWS(kq) ((kq) + 128)
#define CK_LT_NINV 8   // inverses of the diagonal blocks of one column group kept side by side (option local_group <= 8)
static inline long long ck_local_tiled_kq(long long k) { return (k + 2 + 63) / 64 * 64; }
static inline long long ck_local_tiled_doubles(long long k) {
    const long long kq = ck_local_tiled_kq(k);
    return (CK_LT_ROWS(kq) * (kq + 128) + CK_LT_NINV * 64 * 64 + (k + 1) / 2 + 1) & ~1LL;
}
void ck_launch_local_assemble_t(hipStream_t s, const CkMatern* blk, int metric, int i_pred, int cv, double max_dist,
                                const double* pc, int64_t mpad, const double* sc, const double* z, CkLayout L,
                                const CkLocalSys* sys, int n_sys, double* slab, const CkTable* tabs,
                                const double* const* coefs, int use_tab, const double* su, const double* pu,
                                const double* cb, double cmax, int* k0buf /* n_sys ints of scratch */);
// Columns are processed in groups of g 64-column blocks [g0, g0 + 64 g): block i of a group first receives the
// updates of the group's earlier blocks (one pass, K = 64 i), then its diagonal block is factored and inverted and
// the rows below are solved; the trailing matrix behind the group is updated once with K = 64 g (a g-th of the
// read-modify-write traffic of updating after every block).  Systems sorted by k descending: the first n_active
// are the ones that still have the block / trailing columns in question.
void ck_launch_local_tiled_block(hipStream_t s, const CkLocalSys* sys, double* slab, int n_active, int g0, int i,
                                 const int* kq_host, long long* info, int group_blocks);
// the rows BELOW the group's diagonal region, through all of the group's blocks in one launch (group_blocks of them)
// kq_host: the padded sizes of the batch's systems on the host (largest first): the launches have exactly one workgroup
// per chunk / tile that exists (ck_tilemap.h)
void ck_launch_local_tiled_rows_all(hipStream_t s, const CkLocalSys* sys, double* slab, int n_active, int g0,
                                    int group_blocks, const int* kq_host);
void ck_launch_local_tiled_trailing(hipStream_t s, const CkLocalSys* sys, double* slab, int n_active, int g0, int K,
                                    const int* kq_host);
// left-looking form (round 4, option "local_left" = 1, the default): the columns [g0, g0 + W) of every system, rows g0 .., receive
// all updates from the columns to their left in ONE pass (K = g0) before the group is factored; W <= 256
void ck_launch_local_tiled_left(hipStream_t s, const CkLocalSys* sys, double* slab, int n_active, int g0, int W,
                                const int* kq_host);
void ck_launch_local_reduce_t(hipStream_t s, const CkLocalSys* sys, int n_sys, const double* slab,
                              const long long* info, double c0var, double* pred, double* err);
