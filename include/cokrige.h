/* cokrige.h -- C ABI of libcokrige_hip.so, the MI355X (gfx950) numeric core of
 * the bivariate Matern cokriging predictor.
 *
 * The reference (91Mrwu/sif-xco2-cokriging) is pure Python and has no FFI of its
 * own; its drop-in boundary is the Python class API
 *     joint_prediction.Predictor   (src/joint_prediction.py:13-92)
 *     point_prediction.Predictor   (src/point_prediction.py:21-96)
 *     MultivariateMatern.covariance/cross_covariance (src/model.py:193-207)
 *     fields.distance_matrix       (src/fields.py:318-342)
 *     MultiField.empirical_variograms (src/fields.py:192-252)
 * Each entry point below names the reference lines whose arithmetic it
 * replaces.  The Python host layer in sif-xco2-cokriging_amd/ binds these
 * symbols with ctypes (see INTEGRATION.md for the stub a maintainer of the
 * reference would add).
 *
 * Conventions
 *   - every function returns int: 0 = ok, < 0 = error (text from
 *     ck_last_error(), thread local).  Numerical failure of the Cholesky
 *     factorisation is NOT an error code: it is reported LAPACK-style through
 *     `info` (1-based index of the first non-positive pivot, 0 = success).
 *   - "host" pointers are caller-owned, C-contiguous float64 (numpy) buffers,
 *     read or written during the call and never retained.
 *   - "dev" pointers are device addresses (e.g. torch.Tensor.data_ptr()).
 *   - coordinates are rows [lat, lon] in degrees for CK_METRIC_HAVERSINE
 *     (fast_dist=True, km) or [x, y] for CK_METRIC_EUCLID (src/fields.py:326-342).
 *   - a handle owns one GPU and one HIP stream; it is not thread-safe, distinct
 *     handles may be used from distinct threads.
 */
#ifndef COKRIGE_H
#define COKRIGE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ck_handle ck_handle;

#define CK_METRIC_HAVERSINE 0 /* fast_dist=True : 6371 km * haversine (src/fields.py:332-336) */
#define CK_METRIC_EUCLID 1    /* fast_dist=False, units=None : Euclidean (src/fields.py:340-342) */

#define CK_APPLY_SIGMA 1 /* ck_panel_apply: update the local trailing block columns of Sigma */
#define CK_APPLY_AUX 2   /* ck_panel_apply: forward-substitute / update the right-hand-side rows */
/* Every panel buffer (ck_panel_buffer: the owner's storage or a receive slot) is followed by this many bytes the
 * library never reads or writes: room for the host to pad a panel to a multiple of world x 4 KB, so that the exchange
 * can be an in-place all-gather of equal pieces (distributed.py, exchange = "sag"). */
#define CK_PANEL_SLACK_BYTES (64 * 512 * 8)

/* ---- library ------------------------------------------------------------- */
const char* ck_last_error(void);
int ck_version(void);
int ck_device_count(int* n);

/* ---- handle ---------------------------------------------------------------- */
int ck_create(int device_id, ck_handle** out);
/* The multi-GPU form of ck_create (SURVEY.md section 8b lists ck_create(device_ids, n_dev, ...)): the run is one process
 * -- one handle -- per GPU; every rank passes the SAME device list and its own rank and gets a handle on device_ids[rank]
 * that is already partitioned (ck_set_partition(rank, n_dev)).  The panels then travel between the ranks' ck_panel_buffer
 * addresses by whatever the host framework provides (RCCL under torch.distributed in distributed.py / workers.py). */
int ck_create_partitioned(const int* device_ids, int n_dev, int rank, ck_handle** out);
int ck_destroy(ck_handle* h);
/* external != 0: launch on the caller's HIP stream (hipStream_t, e.g.
 * torch.cuda.current_stream().cuda_stream; NULL is the legacy default stream, which is what
 * torch uses unless told otherwise).  external == 0: back to the handle's own stream. */
int ck_set_stream(ck_handle* h, void* hip_stream, int external);
/* Let the caller provide all device storage (e.g. one torch uint8 tensor), so a
 * host framework owns the memory and can run collectives on slices of it.
 * Without an arena the library allocates with hipMalloc.  Must precede ck_set_data. */
int ck_set_arena(ck_handle* h, void* dev_base, int64_t nbytes);
int ck_synchronize(ck_handle* h);
/* Device bytes the handle will allocate for the data set so far, the current partition and
 * `m` prediction points (size an arena with it; call after ck_set_data/ck_set_partition). */
int ck_estimate_bytes(ck_handle* h, int64_t m, int64_t* out);

/* ---- model and data ------------------------------------------------------- */
/* Matern parameters in the reference's order (src/model.py:122-130,145-152):
 * sigma[n_procs], nu[3] = (11, 12, 22), len_scale[3] = (11, 12, 22),
 * nugget[n_procs], rho12.  n_procs = 1: nu[0], len_scale[0] only. */
int ck_set_model(ck_handle* h, int n_procs, const double* sigma, const double* nu, const double* len_scale,
                 const double* nugget, double rho12);
int ck_set_metric(ck_handle* h, int metric);
/* 1-D block-column-cyclic ownership of Sigma over `world` processes (one GPU
 * each); prediction points are sharded by the caller.  Default (0, 1). */
int ck_set_partition(ck_handle* h, int rank, int world);
/* Observation sites of process k: coords (n_k x 2), values (n_k): Field.coords_main /
 * Field.values_main (src/fields.py:78-81, consumed at src/joint_prediction.py:53-55,106-111,130-132). */
int ck_set_data(ck_handle* h, int k, const double* coords_host, const double* values_host, int64_t n_k);

/* ---- element-wise parity surface (tests, and the model/fields mirrors) ----- */
/* fields.distance_matrix(A, B) (src/fields.py:318-342) -> out (a x b). */
int ck_distance_dense(ck_handle* h, const double* A_host, int64_t a, const double* B_host, int64_t b,
                      double* out_host);
/* covariance(i, D(A,B), use_nugget) if i == j else cross_covariance(i, j, D(A,B))
 * (src/model.py:193-207 on src/fields.py:318-342) -> out (a x b). */
int ck_cov_dense(ck_handle* h, int i, int j, const double* A_host, int64_t a, const double* B_host, int64_t b,
                 int use_nugget, double* out_host);
/* the same at given lags h[n] (MultivariateMatern.covariance / cross_covariance on an array). */
int ck_cov_lags(ck_handle* h, int i, int j, const double* lags_host, int64_t n, int use_nugget, double* out_host);
/* Model (cross-)variograms row by row -- the "fit" column of MultivariateMatern._map_fit and the
 * curves of variograms() / FittedVariogram (src/model.py:209-260, 330-331): row r has process pair
 * (i[r], j[r]) and lag h[r]; kind 0: semivariance(i, h) if i == j else cross_semivariance(i, j, h);
 * kind 1 ("covariogram"): covariance(i, h) incl. nugget at h == 0 / cross_covariance(i, j, h). */
int ck_model_variogram(ck_handle* h, const int32_t* i_host, const int32_t* j_host, const double* lags_host, int64_t n,
                       int kind, double* out_host);

/* ---- joint (global) cokriging: src/joint_prediction.py:35-153 --------------- */
/* K1: assemble the lower block triangle of Sigma = [[C11, C12], [C12^T, C22]] for the
 * locally owned block columns (Predictor._joint_cov, src/joint_prediction.py:124-153). */
int ck_assemble_joint(ck_handle* h);
/* K3: in-place blocked Cholesky Sigma = L L^T (cho_factor(lower=True), src/joint_prediction.py:69).
 * info = 0, or the 1-based order of the leading minor that is not positive definite
 * (scipy raises LinAlgError with that number).  Single-process form; for world > 1 drive
 * ck_panel_factor / ck_panel_apply from the host with a broadcast in between. */
int ck_factor(ck_handle* h, int64_t* info);
/* K2 + K4: prediction and standard error of process i at pcoords (m x 2):
 * c0 (Predictor._pred_cross_cov, :104-122), forward substitution V = L^-1 [c0 | z],
 * pred = V^T y, pred_err = nan_to_num(sqrt(sigma_i^2 + nugget_i - |V_k|^2)) (:68-78).
 * Needs ck_factor; may be called repeatedly. */
int ck_predict(ck_handle* h, int i, const double* pcoords_host, int64_t m, double* pred_host, double* pred_err_host);
/* ck_factor + ck_predict in one call, for a Sigma that is assembled and not yet factored (Predictor.__call__,
 * src/joint_prediction.py:60-78, does exactly this sequence): the factorisation and the forward substitution of the
 * right-hand sides run as two overlapped sweeps, the substitution one panel group behind the factorisation, so that each
 * fills the other's idle stretches (panel chain, under-filled in-group launches, launch drains).  Same results to rounding,
 * same *info and error behaviour as ck_factor; *info != 0 leaves pred / pred_err untouched.  Beyond 128 panels (N > 65 536),
 * where the overlap no longer pays, the call runs the two sweeps one after the other (option "fused_sweeps": -1 this rule,
 * 0 never overlapped, 1 always).  The factor stays resident:
 * further ck_predict calls work as after ck_factor. */
int ck_factor_predict(ck_handle* h, int i, const double* pcoords_host, int64_t m, double* pred_host, double* pred_err_host,
                      int64_t* info);

/* _verify_model (src/joint_prediction.py:60-66, 260-274): is the stacked matrix [[C_pp, c0^T], [c0, Sigma]] of the
 * prediction sites of the LAST ck_predict positive definite?  Sigma is (ck_factor succeeded), so this is the
 * Cholesky of the m x m Schur complement C_pp - V^T V on the solved right-hand sides that ck_predict left on the
 * device (the reference factorises the (m + N) x (m + N) matrix).  info = 0: positive definite; > 0: LAPACK-style
 * index of the failing leading minor among the prediction sites -- the reference then warns
 * "Prediction joint covariance matrix is not positive definte".  The index counts the sites in the order the library
 * laid them out (for m >= 256 with option site_order = 1 that is a Hilbert-curve order, not the caller's): use it as
 * a verdict (zero / non-zero), as the reference does.  Only valid directly after the ck_predict whose sites are meant:
 * ck_set_model, ck_set_metric, ck_assemble_joint and ck_factor invalidate the solved right-hand sides and this call then
 * fails.  Consumes the data row of the right-hand sides (a second call gives the same verdict; ck_aux_finish must
 * not be repeated after it).  Needs m (m + 512) / 2 more doubles of device memory. */
int ck_verify_model(ck_handle* h, int64_t* info);

/* Leave-one-out cross-validation of process i at all its data sites from ONE factorisation
 * (Predictor.cross_validation, src/joint_prediction.py:207-257, which re-solves per datum):
 * pred_q = z_q - (Sigma^-1 z)_q / (Sigma^-1)_qq, pred_err_q = sqrt(1 / (Sigma^-1)_qq); n_i values
 * each.  Needs ck_factor. */
int ck_loocv(ck_handle* h, int i, double* pred_host, double* pred_err_host);

/* Simulation draw z = L eps in the caller's stacked order (process 0 sites, then process 1):
 * sim.BivariateRandomField._simulate (src/sim.py:52-54: cholesky(cmat, lower=True) @ noise).
 * n = number of observations; needs ck_factor. */
int ck_sample(ck_handle* h, const double* noise_host, double* out_host, int64_t n);

/* ---- step-wise form (multi-GPU, fused solve) ------------------------------ */
int ck_num_panels(ck_handle* h, int* n_panels, int* panel_width, int64_t* n_padded);
int ck_panel_owner(ck_handle* h, int K, int* owner_rank);
/* Right-hand-side rows for prediction of process i at pcoords (this rank's shard):
 * assembles c0^T rows and the data row z^T (K2). */
int ck_aux_begin(ck_handle* h, int i, const double* pcoords_host, int64_t m);
/* Factor block column K in place (owner only): diagonal blocks, panel solve. */
int ck_panel_factor(ck_handle* h, int K);
/* Device address / size of the packed panel K: the owner's storage, or this rank's
 * receive buffer (the host broadcasts owner -> all between factor and apply). */
int ck_panel_buffer(ck_handle* h, int K, void** dev_ptr, int64_t* nbytes);
/* Apply panel K to the local trailing block columns (CK_APPLY_SIGMA) and/or to the
 * right-hand-side rows (CK_APPLY_AUX). */
int ck_panel_apply(ck_handle* h, int K, int what);
/* The CK_APPLY_SIGMA part restricted to the locally owned block columns J in [J_lo, J_hi] (clipped to
 * K + 1 .. n_panels - 1).  With it the host can run the classical look-ahead: update column K + 1 first,
 * factor it, start its broadcast, and update the remaining columns under the broadcast. */
int ck_panel_apply_sigma(ck_handle* h, int K, int J_lo, int J_hi);
/* Grouped form (world >= 1): the panels K0 .. K0 + np - 1 -- all readable on this rank, in their owner's storage or in a
 * receive slot (remote panel K lands in slot K % recv_slots, option "recv_slots", default 2; at most recv_slots remote
 * panels are alive at a time) -- applied in ONE pass with the contraction dimension 512 np, i.e. a np-th of the
 * read-modify-write traffic of np calls of ck_panel_apply (single process: what ck_factor / ck_predict do for groups of 3).
 *   what & CK_APPLY_SIGMA: the locally owned block columns J of [max(J_lo, K0 + np), J_hi], every n_phase-th of them
 *                          starting with the phase-th (so that the host can split one group update into n_phase pieces
 *                          and start the next panels' steps and exchanges in between);
 *   what & CK_APPLY_AUX:   the update part only (no substitution) of the right-hand-side block columns of the same
 *                          range, piece `phase` of n_phase contiguous pieces.
 * ck_panel_aux_solve(K): the substitution of right-hand-side block column K with the diagonal block of panel K
 * (ck_panel_apply(K, CK_APPLY_AUX) = ck_panel_aux_solve(K) + the update of the block columns beyond K). */
int ck_panel_apply_group(ck_handle* h, int K0, int np, int what, int J_lo, int J_hi, int phase, int n_phase);
int ck_panel_aux_solve(ck_handle* h, int K);
/* pred / pred_err of the local shard after all panels were applied to the aux rows. */
int ck_aux_finish(ck_handle* h, double* pred_host, double* pred_err_host);
/* info flag of the factorisation so far (synchronises). */
int ck_factor_info(ck_handle* h, int64_t* info);

/* ---- local-neighbourhood cokriging: src/point_prediction.py:45-249 ------------------------- */
/* Prediction and standard error of process i at pcoords (m x 2) from the observations within
 * max_dist of each point (cv != 0: observations of process i at distance 0 are withheld,
 * src/point_prediction.py:141-143).  One workgroup per point: radius search, local covariance,
 * Cholesky with c and z as extra rows.  Points with no observation in range, or whose local
 * covariance is not positive definite, get NaN in both outputs (:218-233); their numbers and the
 * largest neighbourhood size are returned for the caller's warnings.  Needs ck_set_model /
 * ck_set_data only. */
int ck_predict_local(ck_handle* h, int i, const double* pcoords_host, int64_t m, double max_dist, int cv,
                     double* pred_host, double* pred_err_host, int64_t* n_empty, int64_t* n_not_pd,
                     int64_t* k_max);
/* The reference builds the local predictor's state once, in its constructor (src/point_prediction.py:24-43: the Sigma blocks
 * the neighbourhoods are gathered from).  Here that state is the scratch slab of the large-neighbourhood paths: nbytes > 0
 * reserves at least that much, 0 the automatic budget of ck_predict_local (a quarter of the free device memory, at most
 * 32 GiB; option "local_slab_mb" if set).  Afterwards no ck_predict_local whose batches fit pays a hipMalloc (tens of GiB
 * right after smaller buffers were freed: up to seconds -- ck_timings [14] shows what a call spent growing the slab). */
int ck_local_reserve(ck_handle* h, int64_t nbytes);

/* ---- empirical (cross-)semivariogram / covariogram: src/fields.py:192-232, 378-403 ----- */
/* Fields i and j: coords (n x 2), residuals = values minus their mean (src/fields.py:380).
 * same != 0: marginal variogram, strict upper triangle of the i-i pairs (:195-199; j args ignored);
 * else all n_i * n_j pairs (:200-203).  Uses the handle's metric.  At most 2^28 - 1 points per field. */
int ck_vario_begin(ck_handle* h, const double* coords_i_host, const double* resid_i_host, int64_t n_i,
                   const double* coords_j_host, const double* resid_j_host, int64_t n_j, int same);
/* Pass 1: lo = smallest positive and hi = largest pair distance among pairs with d <= max_dist
 * (src/fields.py:212, 394-395); n_positive = 0 if there is no such pair (lo, hi = NaN). */
int ck_vario_extent(ck_handle* h, double max_dist, double* lo, double* hi, int64_t* n_positive);
/* With ck_set_partition(rank, world), world > 1, both passes visit only this rank's share of the pair
 * tiles (tile t belongs to rank t mod world): the host combines the ranks' results -- MIN of lo / MAX of hi
 * over the ranks that found a pair, SUM of the per-bin sums and counts (distributed.DistributedVariogram). */
/* Pass 2: per-bin sum of cloud values and pair count for bins (e_b, e_b+1], first bin [0, e_1]
 * (pd.cut(include_lowest=True), :214-222); edges[0] must be 0; at most 60 bins.
 * cloud = 0.5 (a - b)^2, or a * b when covariogram != 0 (:378-386). */
int ck_vario_bin(ck_handle* h, double max_dist, const double* edges_host, int n_edges, int covariogram,
                 double* sums_host, int64_t* counts_host);
int ck_vario_end(ck_handle* h);
/* Counters of the last ck_vario_extent / ck_vario_bin: [0] pairs of the extent pass decided on the host,
 * [1] pairs of the binning pass decided on the host, [2] pairs the binning pass visited (tiles that cannot hold a
 * retained pair are skipped), [3] extra rounds of the extent pass. */
int ck_vario_stats(ck_handle* h, int64_t* out4, int n);
/* How the variogram passes decide ties.  The reference decides on the rounded distance d (`d <= max_dist`, pd.cut on
 * the edges: src/fields.py:212-216); the kernels compare a monotone function of d and hand every pair within the
 * rounding band of a threshold to the reference's own formula: Euclidean on the device (bit-identical to scipy's
 * cdist), haversine on the host through libm -- bit for bit sklearn's haversine_distances(np.radians(X)) * 6371
 * (src/fields.py:332-336), which device trigonometry is not.  That host function, for n coordinate pairs
 * (A, B: n x 2), no GPU needed: */
int ck_ref_distance(int metric, const double* A_host, const double* B_host, int64_t n, double* out_host);

/* The order in which the library lays n sites out on the device (option site_order = 1, the default, and always
 * for the variogram's points): along a Hilbert curve of order 16 through the sites' bounding box, sites of one cell
 * in the caller's order.  perm_host[k] = index of the site that comes k-th.  Host-only (no handle, no device): the
 * results of every entry point come back in the caller's order, so this is for inspection and tests. */
int ck_hilbert_order(const double* coords_host, int64_t n, int64_t* perm_host);

/* ---- diagnostics ------------------------------------------------------------- */
/* Copy the locally owned part of Sigma / L back as a dense (N x N) lower triangle
 * (upper triangle zero-filled); small N only (tests).  Rows / columns are in the handle's INTERNAL
 * site order: process 0 then process 1, each in the order ck_debug_site_order reports. */
int ck_debug_get_lower(ck_handle* h, double* out_host, int64_t n);
/* Entries (rows[e], cols[e]) of Sigma -- after ck_assemble_joint -- or of its factor L L^T = Sigma -- after ck_factor; then
 * the lower triangle, (r, c) and (c, r) both give L[max][min] -- with the indices in the CALLER's stacked order
 * (process 0 sites, then process 1), whatever the internal site order; any n (parity tests at sizes where the dense
 * matrix does not fit a host array: sampled entries of the table-path assembly, sampled rows of the factor). */
int ck_debug_get_entries(ck_handle* h, const int64_t* rows_host, const int64_t* cols_host, int64_t n, double* out_host);
/* perm_out[j] = caller's index (within process k) of the site at internal position j.  Identity
 * with option site_order = 0; a Hilbert-curve order with site_order = 1 (the default). */
int ck_debug_site_order(ck_handle* h, int k, int64_t* perm_out, int64_t n_k);
/* Phase profile of the 64 x 64 diagonal-block kernel (Cholesky + inverse, the latency-bound link of the panel chain):
 * out8[0..5] = microseconds of load | factorisation | scaling + store | inverse of the diagonal 16 x 16 blocks |
 * off-diagonal blocks of the inverse | store of the inverse; [6] / [7] = a whole launch (instrumented / product kernel). */
int ck_debug_potrf_profile(ck_handle* h, int iters, double* out8);
/* Where a link of the cooperative panel step's chain (k_panel_coop) spends its time, on a rows x 512 test panel: out64[8 b + k],
 * b = 1 .. 7 = microseconds (shader clock at 2.4 GHz) from chunk b seeing the pivot chunk's "rows final" flag to k = 1
 * accumulation done | 2 inverse flag seen | 3 rows solved and stored | 4 drained + rows flag set | 5 diagonal block updated |
 * 6 factored + inverse stored | 7 drained + flag set; out64[8 b] (b >= 2) = the period between links; out64[0] = the whole
 * launch (HIP events). */
int ck_debug_coop_profile(ck_handle* h, int64_t rows, double* out64);
/* Raw lane/register -> (row, col) map of v_mfma_f64_16x16x4_f64: out[64*4*3] ints (row, col, k-map check). */
int ck_debug_mfma_probe(ck_handle* h, int32_t* out_host);
/* FP64 MFMA issue-rate microbenchmark (v_mfma_f64_16x16x4_f64, operands in registers,
 * `waves_per_simd` waves on every SIMD of the chip): the measured ceiling the GEMM kernels are
 * compared with, next to the datasheet 78.6 TFLOP/s. */
/* Where the workgroups of a stream created with hipExtStreamCreateWithCUMask(cu_mask8[8]) run (NULL: an
 * unmasked stream): out[wg] = xcc_id << 16 | se_id << 8 | sh_id << 4 | cu_id from HW_REG_XCC_ID / HW_REG_HW_ID. */
int ck_debug_cu_probe(ck_handle* h, const uint32_t* cu_mask8, int n_wg, uint32_t* out_host);
int ck_debug_mfma_peak(ck_handle* h, int waves_per_simd, int iters, double* out3 /* TFLOP/s, shader MHz, cycles per MFMA per wave */);
/* The clock the chip holds under the Cholesky trailing-update kernel on the data of the last ck_factor (the chip lowers
 * its clock under load, by an amount that depends on the operands).  Needs option "gemm_stamps" = 1 (set after the first
 * ck_assemble_joint; the factorisation then runs a stamped instantiation of k_syrk_group_d): out6 = median, 5 % and 95 %
 * quantile of the in-kernel shader clock in MHz over the stamped workgroups, their number, a workgroup's median
 * lifetime in shader cycles and in microseconds. */
int ck_debug_gemm_clock(ck_handle* h, double* out6);
/* Diagnostic: n_side cooperative panel steps (scratch panel of `rows` rows) on the high-priority side stream against the
 * first trailing update of the factorisation on the main stream.  mode 0: the update alone, 1: the side kernels alone,
 * 2: both, released by one event, the first side kernel submitted in front of the update.  out[0] = update ms, out[1 + i] =
 * end of side kernel i after the common start in ms.  Needs an assembled, unfactored handle of >= 8 panels and destroys
 * Sigma (assemble again). */
int ck_debug_stream_overlap(ck_handle* h, int mode, int64_t rows, int n_side, double* out);
/* Host only (no device needed): the 128 x 128 tiles of one Cholesky trailing update -- block columns J0 + u Jstep, u < nJ,
 * of a matrix whose rows / columns from nvalid on are identity padding -- in the order the launch's workgroups take them
 * (csrc/ck_tilemap.h): out3[3 t .. 3 t + 2] = block column, tile row and tile column inside it, for t < min(total, cap).
 * Returns the number of tiles = the launch's grid size, or -1. */
int64_t ck_debug_tile_map(int64_t nvalid, int J0, int Jstep, int nJ, int32_t* out3, int64_t cap);
/* Host only: the same for a launch over the TALL matrix [Sigma; c0^T; z^T] (ck_factor_predict, round 4: one launch updates a
 * block column's triangle tiles and the aux_tile_rows x 4 tiles of the right-hand-side block below it):
 * out4[4 t .. 4 t + 3] = block column, tile row, tile column, 1 if the tile lies in the right-hand-side block (tile row and
 * column then count inside that block).  Returns the grid size, or -1. */
int64_t ck_debug_tall_map(int64_t nvalid, int J0, int nJ, int aux_tile_rows, int32_t* out4, int64_t cap);
/* Host only: the workgroups of a batched launch of the local predictor's tiled path over n_sys systems, largest first, with
 * counts[y] work units each (non-increasing): out2[2 b], out2[2 b + 1] = system (-1: a padding workgroup at the end of a
 * run of equal counts) and unit of workgroup b, for b < min(grid, cap).  Returns the grid size, or -1. */
int64_t ck_debug_run_map(const int32_t* counts, int n_sys, int32_t* out2, int64_t cap);
/* Raw stamps of the last stamped launch ("gemm_stamps" = 1: every launch, = 2 + K0: only the trailing update behind the
 * panel group that starts at K0): out[4 b .. 4 b + 3] = shader cycles and 100 MHz ticks of workgroup b's lifetime (0, 0 if
 * it returned at once), its start in 100 MHz ticks, XCC_ID << 32 | HW_ID; grid4 = grid x, grid y, first block column,
 * number of panels of that launch. */
int ck_debug_gemm_stamps(ck_handle* h, uint64_t* out_host, int64_t n_words, int64_t* grid4);
/* Stage timings of the last calls in milliseconds (HIP events):
 * [0] assemble Sigma, [1] factor, [2] assemble aux, [3] solve sweep, [4] reduce,
 * and, with option "time_gemm": [5]/[6] total ms / number of the Cholesky trailing-update
 * launches (k_syrk_panels) of the last ck_factor, [7]/[8] the same for the right-hand-side
 * trailing updates of the last ck_predict; [9] variogram binning pass (ck_vario_bin); [10] local prediction kernels (ck_predict_local);
 * [11] ck_verify_model (host wall clock, synchronised); [12] 1 if the last ck_factor had to repeat the factorisation because
 * a workgroup of the cooperative panel step (option "panel_fused" bit 4) timed out waiting for a pivot block (never
 * observed; the bit is then off for the handle); [13] after ck_factor_predict: the span of the two overlapped sweeps -- [1] is then the
 * factorisation's span inside it (it shares the chip with the substitution), [3] what the substitution adds behind the
 * factorisation's end, and [5] .. [8] are 0 (a launch's duration would include the other sweep's share of the chip) -- except
 * in the tall sweep (option "tall_sweep", the default): [5]/[6] = sum of the durations / number of its update launches
 * (k_tall_group_d; launches of the two streams overlap each other, so the sum exceeds the span); [10] counts device work only
 * (counting pass + solve), [14] = host milliseconds the last ck_predict_local / ck_local_reserve spent growing the scratch slab
 * (hipMalloc; 0 when it did not grow); [15] after a tall sweep with "time_gemm": the union of the intervals of its update launches
 * -- the time during which k_tall_group_d is running at all (its launches on the two streams overlap each other). */
int ck_timings(ck_handle* h, double* out, int n);
/* The assembly kernels evaluate the covariance through a per-block table of C = amp * rho over
 * the squared chord (built on the device from the exact K_nu evaluator and verified against it
 * when the data are laid out; error measure |table - C| / (|amp| max(rho, 1e-6)), gate 2e-13).
 * Per block (0 = 11, 1 = 12, 2 = 22): whether the table passed its check (else the assembly uses
 * the exact evaluator), its size / range, its measured error. */
int ck_table_info(ck_handle* h, int block, int* enabled, int* n_intervals, double* q_lo, double* q_hi,
                  double* max_rel_err);
/* Entries the table path deferred to the exact evaluator (pairs closer than the table's lower end
 * or beyond its upper end) in this handle's assemblies since the last reset. */
int ck_table_fallbacks(ck_handle* h, int reset, int64_t* count);
/* Options: "time_gemm" (0/1/2) brackets every trailing-update launch with HIP events (2: the Sigma updates only, for
 * the step-wise form, where Sigma and right-hand-side updates alternate; read back through ck_timings);
 * "exact_cov" (0/1) makes the assembly kernels evaluate K_nu per entry instead of the tables;
 * "recv_slots" (2..64, default 2; before the first assemble / ck_estimate_bytes): receive buffers for remote panels of a
 * multi-process run -- 2 for the per-panel look-ahead schedule, 2 G for the grouped one (ck_panel_apply_group);
 * "lookahead" (-1/0/1, default -1 = automatic) runs the panel step of column K+1 on a second stream under the trailing
 * update of the columns beyond it (per-panel updates); automatic: ck_factor uses it from 12 to 63 panels (where the update is
 * short against the panel step and the one-launch cooperative panel step, submitted in front of the update, really runs
 * beside it: N = 10 000: 12.8 -> 11.4 ms), grouped updates without it from 64 panels on; 1 also switches the solve sweep's
 * variant on;
 * "panel_group" (1..16; default 0 = automatic: 4 for 40 or more panels (3 in rounds 1-3), else 1) = panels per trailing update of
 * ck_factor / ck_predict;
 * "panel_fused" (0..31, default 18 = 2 | 16; bit 0: factorisation, bit 1: right-hand-side rows): inside a 512-column panel the
 * 64-column sub-blocks are processed left-looking with the update and the row solve fused into one launch; bit 2: the
 * 512 x 512 diagonal block first, then one launch for all rows below it; bit 4 (on by default): the WHOLE panel step of the
 * factorisation -- eight 64 x 64 Cholesky factorisations with their inverses, the row solves, the panel-internal updates --
 * in ONE launch of cooperating workgroups: workgroup b owns the 64-row chunk b of the panel and walks it through the
 * sub-blocks as their pivot chunks are published through flags in device memory (agent-scope release / coherent loads,
 * bounded waits), so that a chunk waits for the one chunk above it in the dependency chain instead of for 24 launch
 * boundaries;
 * "coop_spins" (default 2 000 000, ~2 s): polls a workgroup of that launch spends on one flag before it sets the error word and
 * leaves (ck_factor / ck_factor_predict then repeat the factorisation with one launch per dependency and switch bit 4 off:
 * ck_timings [12]); "coop_inject_panel" (default -1; tests): the cooperative step of that panel drops one flag store, once, so
 * that the bounded wait trips and the recovery runs;
 * "tall_sweep" (0/1, default 1): ck_factor_predict as ONE sweep over the tall matrix [Sigma; c0^T; z^T] -- the right-hand-side
 * rows are further workgroups of the cooperative panel step and further tiles of every update launch (k_tall_group_d), with
 * the look-ahead of "fused_la" -- instead of a factorisation and a substitution sweep that share the chip on two streams
 * (same bits either way);
 * "tall_split" (0 never / 1 every panel / 2 automatic, the default) and "tall_split_rows" (default 12 288): in the tall sweep the panel
 * steps of panels behind the first group with at least that many rows run as the cooperative launch on the 512 x 512 head only plus
 * ONE launch of k_panel_rows_all for every other row of the panel and of the right-hand sides -- no workgroup holds a slot while it
 * waits for the head (same bits; N = 40 000: 0.5 % faster);
 * "solve_la" (-1 automatic = from 40 panels, 0, 1): ck_predict's sweep on a resident factor with the chain of the next panel group (the
 * one-column in-group updates, which fill 55 % of the chip, and the rows' walk through each panel) on the high-priority stream UNDER the
 * bulk update of the current group instead of in front of it (N = 40 000: 210 -> 200 ms; same bits);
 * "tall_b2_stream" (0/1, default 1): the tall sweep's update "group g -> everything beyond the next two groups" on the handle's own stream
 * beside the update of the group after next, instead of behind it on one stream (both only wait for group g's panels; 0.2 %);
 * "tall_thin" (0/1, default 1): a last right-hand-side tile row with at most 16 rows in front of the padding (m + 1 = 8 834: two rows)
 * computes those rows' 16-row block only (same bits, 0.35 % at N = 40 000);
 * "assemble_queue" (-1 automatic, 0 off, else the number of workgroups): the table-path assembly kernels as a resident set of
 * workgroups that take their 64 x 512 strips (or halves / quarters of them) from a work queue -- the 48 KB table is loaded once per
 * workgroup instead of once per strip; automatic: the right-hand-side assembly (K2) with 768 workgroups from 8 strips per workgroup
 * on (N = 40 000: 0.62 -> 0.56 ms), Sigma (K1) never (faster on one box, slower on another);
 * "local_slab_mb" = scratch budget of ck_predict_local in MiB (0, default: a quarter of the free memory, at most
 * 32 GiB; the points are processed in batches that fit; the scratch is kept until ck_destroy and reused);
 * "local_tile_min" (default 64 = the LDS kernel's limit): neighbourhoods with more sites than this are factored by
 * the tiled path of ck_predict_local (batched 64-column steps on the matrix cores) instead of one workgroup per
 * point (in LDS up to 64 sites, on a global slab above);
 * "local_group" (1..8, default 4) = 64-column blocks per group of that path; "local_left" (0/1, default 1): a group's columns receive
 * everything from their left in one pass (K = the group's first column) before the group is factored, instead of a K = 64 x
 * local_group update of everything behind every group (same bits: the accumulation order per element is the same);
 * "group_first" (default -1 = automatic: half a group from 40 panels on; 0 = a whole group) / "group_tail" / "group_tail_panels"
 * (default 0 = off): the single-process sweeps' group boundaries -- a first group of so many panels, groups of group_tail panels for
 * the last group_tail_panels panels (the grouping does not change a bit of the results: an element's updates are accumulated k
 * ascending whatever the split; measured, DESIGN.md section 5: groups of four with a first group of two are 0.6 % ahead of groups of three, the tail
 * variants change nothing);
 * "site_order" (0/1, default 1; changing it after the first assemble lays the sites out again): 1 lays the sites of each process -- and
 * sets of >= 256 prediction points -- out along a Hilbert curve inside the library, so that the rows and
 * columns of an assembly tile are neighbours in space (fewer LDS bank conflicts in the table lookups,
 * less divergence in the exact evaluator).  Predictions, LOOCV results and variograms come back in the
 * caller's order either way and agree to rounding (Sigma is permuted symmetrically).  What depends on
 * the order: L itself (ck_debug_get_lower; ck_sample therefore insists on site_order = 0), and the
 * index of the failing leading minor of a Sigma that is not positive definite -- ck_factor repeats such
 * a factorisation in the caller's order to report numpy's index; ck_panel_* callers get the index in
 * factorisation order and can do the same by setting site_order = 0 and sweeping again
 * (distributed.DistributedJoint does). */
int ck_set_option(ck_handle* h, const char* name, int64_t value);
/* Plain C -= A B^T on device buffers through the MFMA kernel (tests / microbenchmarks).
 * A: M x K (lda), B: N x K (ldb), C: M x N (ldc), all row-major device doubles;
 * M % 256 == 0, N % 64 == 0, K % 16 == 0.  lower != 0 skips tiles strictly above the diagonal. */
int ck_dev_gemm_nt(ck_handle* h, double* C_dev, int64_t ldc, const double* A_dev, int64_t lda, const double* B_dev,
                   int64_t ldb, int64_t M, int64_t N, int64_t K, int lower);

#ifdef __cplusplus
}
#endif
#endif /* COKRIGE_H */
