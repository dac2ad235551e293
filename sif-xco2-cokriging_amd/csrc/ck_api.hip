// ck_api.hip -- the C ABI of libcokrige_hip.so (include/cokrige.h): handle, device storage,
// and the host-side drivers of the blocked algorithms.
//
// Storage (all FP64, row-major):
//   Sigma / L   lower block triangle in packed block columns ("panels") of width NB = 512:
//               panel K holds rows [K*NB, Npad) x cols [K*NB, (K+1)*NB), ld = NB.  With
//               world > 1 a process owns the panels K % world == rank.  Npad = N rounded up
//               to NB; padded rows/cols carry an identity so no kernel needs edge code.
//   aux         the right-hand-side rows c0^T (one row per prediction site) and z^T, in the
//               same panel format: aux panel K is mpad x NB at aux + K*mpad*NB.
// Algorithm: right-looking blocked Cholesky on the tall matrix [Sigma; c0^T; z^T]
// (the forward substitution of cho_solve IS the panel step applied to the extra rows),
// two-level blocking NB = 512 / IB = 64, trailing updates on FP64 MFMA.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <string>
#include <utility>
#include <thread>
#include <vector>

#include "../../include/cokrige.h"
#include "ck_host.h"
#include "ck_internal.h"
#include "ck_tilemap.h"
#include "ck_model.h"

static int64_t roundup(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

static int fail(const std::string& msg) { return ck_fail(msg); }   // thread-local text: ck_host.cpp (ck_last_error)
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" +   \
                        std::to_string(__LINE__) + ")");                                          \
    } while (0)
#define CHKH(h)                       \
    if (!(h)) return fail("null handle"); \
    HIPCHK(hipSetDevice((h)->device))

// device temporaries of one call: released on every return path
struct DevTemps {
    std::vector<void*> p;
    ~DevTemps() {
        for (void* x : p)
            if (x) (void)hipFree(x);
    }
    template <class T>
    hipError_t get(T** out, size_t bytes) {
        *out = nullptr;
        const hipError_t e = hipMalloc((void**)out, bytes ? bytes : 8);
        if (e == hipSuccess) p.push_back((void*)*out);
        return e;
    }
};

struct EvPair {
    hipEvent_t a, b;
};

struct ck_handle {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // arena
    char* arena = nullptr;
    int64_t arena_size = 0, arena_used = 0;
    std::vector<void*> owned;   // hipMalloc'ed blocks (no arena)
    // model
    int n_procs = 0;
    int metric = CK_METRIC_HAVERSINE;
    CkMatern blk[3];
    CkMatern* d_blk = nullptr;
    bool model_set = false;
    // partition
    int rank = 0, world = 1;
    // data
    std::vector<double> h_coords[2], h_values[2];
    int64_t n[2] = {0, 0};
    bool data_set[2] = {false, false};
    int64_t N = 0, Npad = 0;      // N: observations; Npad: padded order of the internal matrix
    int64_t n0p = 0, nend = 0;    // internal order: process 0 in [0, n0), process 1 in [n0p, nend)
    int nK = 0;
    bool layout_ready = false;
    // option "site_order": 1 = the sites of each process (and large sets of prediction points) are
    // laid out along a Hilbert curve, so that the 64 rows / columns of an assembly tile are
    // neighbours in space and the lanes of a wave look up neighbouring table intervals; 0 = the
    // caller's order.  perm[k][j] = caller's index of the site at internal position j of process k.
    int site_order = 1;
    std::vector<int64_t> perm[2], pperm;
    bool p_sorted = false;
    double *s0 = nullptr, *s1 = nullptr, *s2 = nullptr, *z = nullptr;   // stacked sites / values (Npad)
    double* su = nullptr;               // chord vectors of the stacked sites (3 x Npad)
    double* pu = nullptr;               // chord vectors of the prediction sites (3 x mpad)
    // tabulated correlation (fast assembly)
    CkTable tab[3];
    double* d_coef[3] = {nullptr, nullptr, nullptr};
    CkTable* d_tabs = nullptr;
    double** d_coefptr = nullptr;
    bool tables_built = false;
    bool exact_cov = false;             // option "exact_cov": bypass the tables
    CkWorklist wl = {nullptr, nullptr, 0, nullptr};   // entries deferred by the table kernels
    unsigned* wl_counts = nullptr;           // its two counters (used alternately: next_worklist)
    int64_t fallback_total = 0;
    std::vector<double*> sig;    // per panel; nullptr if not owned
    double** d_sigptr = nullptr;
    double** d_panelptr = nullptr;   // where panel K can be read on this rank: own storage or receive buffer K & 1
    int *d_tile0 = nullptr, *d_panel_of = nullptr;   // assembly launch map of the owned panels
    int* d_strip_order = nullptr;                    // the same strips sorted by Matern block (work-queue form of the assembly)
    int n_owned = 0, total_tiles = 0;
    // receive slots for remote panels (world > 1): panel K lands in slot K % recv_slots.  Two slots carry the per-panel
    // look-ahead schedule; 2 G slots the grouped one (the G panels of the group being applied + the G being received)
    int recv_slots = 2;
    std::vector<double*> recv;
    long long* d_info = nullptr;
    bool assembled = false, factored = false;
    // aux
    int i_pred = 0;
    int64_t m = 0, mpad = 0, aux_cap = 0;   // aux_cap in doubles
    double* aux = nullptr;
    double *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
    int64_t p_cap = 0;
    double *d_pred = nullptr, *d_err = nullptr;
    double* d_pcoords = nullptr;
    char* mv_buf = nullptr;   // ck_model_variogram's device rows (i, j, lag, out), kept between the cost evaluations of a fit
    int64_t mv_cap = 0;
    int aux_state = 0;   // 0: nothing usable | 1: right-hand sides assembled | 2: solved by ck_predict (rows = V^T, row m = y)
    // Schur complement of the prediction sites (ck_verify_model), kept between calls with the same padded order
    int64_t sch_M = 0;
    std::vector<double*> sch_sig;
    double** d_sch_ptr = nullptr;
    int *d_sch_tile0 = nullptr, *d_sch_panel_of = nullptr;
    int sch_tiles = 0;
    double *sch_pc = nullptr, *sch_c = nullptr, *sch_u = nullptr;
    CkMatern* d_sch_blk = nullptr;
    CkTable* d_sch_tabs = nullptr;
    double** d_sch_coefptr = nullptr;
    long long* d_sch_info = nullptr;
    // empirical variogram state (ck_vario_*)
    std::vector<double> vg_ci, vg_cj, vg_vi, vg_vj;   // host copies of coordinates / residuals in the device's point order
                                                      // (the pairs the kernels leave to the host are decided on these)
    double *vg_iu = nullptr, *vg_iv = nullptr, *vg_ju = nullptr, *vg_jv = nullptr;
    unsigned long long* vg_best = nullptr;       // extreme-pair hints of the extent pass (ck_vario.hip)
    double *vg_jb = nullptr, *vg_ib64 = nullptr, *vg_jbsub = nullptr;   // bounding balls: 1024-point "j" chunks, 64-point
                                                                        // "i" blocks (wave tiles), 128-point sub-chunks
    int64_t vg_ni = 0, vg_nj = 0;
    int vg_same = 0, vg_bgrid = 0;
    void* vg_part = nullptr;
    double* vg_psum = nullptr;
    unsigned long long* vg_pcnt = nullptr;
    double* vg_out = nullptr;          // xa[38] | xb[38] | dthr[38] | sums[36] | counts[37] (8-byte words) | kernel arguments
    CkVarioPair* vg_list = nullptr;    // pairs left to the host
    unsigned* vg_count = nullptr;
    unsigned vg_list_cap = 0;
    int64_t vg_stats[4] = {0, 0, 0, 0};   // host-decided pairs of the extent pass | of the binning pass | pairs visited by
                                          // the binning pass | extra extent rounds
    // timings
    double t_ms[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int time_gemm = 0;   // 1: bracket every trailing-update launch with events | 2: the Sigma updates only (step-wise form)
    std::vector<EvPair> gemm_ev;
    size_t gemm_ev_used = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    // look-ahead: the panel step of column K+1 runs on a second (high-priority) stream under the
    // trailing update of panel K
    hipStream_t side = nullptr;
    hipStream_t side_lo = nullptr;    // a second stream of ordinary priority (option "fused_prio" = 2)
    int fused_sweeps_opt = -1;        // ck_factor_predict: -1 automatic (overlapped up to 128 panels) | 0 sequential | 1 overlapped
    int fused_prio = 0;               // ck_factor_predict: which sweep runs on the high-priority stream (fused_sweeps)
    int fused_group = 0;              // ck_factor_predict: panels per group, 0 = as ck_factor
    int group_first = -1, group_tail = 0, group_tail_panels = 0;   // group_plan(): first group (-1 automatic) / last panels with other group sizes
    int assemble_queue = -1;          // option "assemble_queue": resident workgroups of the table-path assembly kernels on a work queue
                                      // (-1 automatic: 768 from 8 strips per workgroup on; 0 = one strip per workgroup)
    int tall_sweep = 1;               // ck_factor_predict: ONE sweep over the tall matrix [Sigma; c0^T; z^T] (tall_sweeps, round 4)
    int tall_split = 2;               // tall sweep, panel steps hidden under a bulk update: cooperative launch on the 512 x 512 head only,
                                      // every other row (Sigma's and the right-hand sides') through k_panel_rows_all (0: never, 1: every
                                      // panel, 2: panels behind the first group with at least tall_split_rows rows)
    int tall_split_rows = 24 * CK_NB; // tall_split 2: shorter panels keep the one cooperative launch
    int solve_la = -1;                // ck_predict's sweep with the chain of the next group under the bulk of the current one (-1: from 40 panels)
    int tall_b2_stream = 1;           // tall sweep: B2(g) on the handle's own stream beside B1(g)
    std::vector<hipEvent_t> ev_b2;
    int tall_thin = 1;                // tall sweep: a last right-hand-side tile row with <= 16 rows in front of the padding computes those only
    int fused_la = -1;                // ck_factor_predict: look-ahead inside the factorisation (fused_sweeps_la); -1: from 40 panels
                                      // (N = 40 000: 522.3 -> 518.1 ms, three interleaved repetitions; N = 10 000: no difference)
    std::vector<hipEvent_t> ev_col, ev_pan;   // [K]: column K fully updated | panel K done
    // option "panel_group": G panels are factored (left-looking inside the group) before the trailing
    // matrix is updated ONCE with K = 512 G (ck_la.hip: gemm_tile_m); 1 = update after every panel
    int panel_group = 0;   // 0 = automatic: 3 for 40 or more panels, else 1 (measured: the one-column in-group
                           // launches cost more than the saved C traffic on small matrices)
    int64_t local_slab_mb = 0;   // option "local_slab_mb": scratch budget of ck_predict_local (0 = automatic)
    int local_tile_min = 64;     // option "local_tile_min": neighbourhoods larger than this take the tiled path
    int local_group = 4;         // option "local_group": 64-column blocks per trailing update of the tiled path
    int local_left = 1;          // option "local_left": the tiled path's groups are updated left-looking (one pass with K = g0 in front
                                 // of each group) instead of right-looking (a K = 64 G update of everything behind each group)
    int panel_fused = 2 | 16;         // option "panel_fused", bit 0: factorisation, bit 1: right-hand-side rows --
                                      // left-looking 64-column sub-blocks inside a panel, fused launches (measured at
                                      // N = 40 000: solve sweep 228.2 -> 219.6 ms, factorisation 362.9 -> 365.0 ms);
                                      // bit 4: the whole panel step of the factorisation in one launch of cooperating
                                      // workgroups (k_panel_coop: 360.4 -> 350.0 ms; N = 10 000: 14.4 -> 12.8 ms)
    int64_t loo_g0 = -1;              // >= 0 during ck_loocv: right-hand-side row 1 + p is the unit vector of site loo_g0 + p
    double* d_chunkb = nullptr;       // chunk bounds of the sites for the radius search (ck_local.hip: LpSearch)
    double* local_slab = nullptr;     // scratch of ck_predict_local, kept between calls (allocating tens of GiB
    long long local_slab_doubles = 0; // costs up to seconds, erratically); grows when a call needs more
    // option "lookahead": the panel step of column K + 1 on a second stream under the trailing update of the columns beyond
    // it (per-panel updates, no grouping).  With 24 launches per panel step it gained nothing (the side queue's launches
    // starve behind the resident update kernel); with the ONE-launch cooperative panel step, submitted in front of
    // the update, it does where the update is short against the panel step: factor_ms at N = 10 000: 12.8 -> 11.4,
    // N = 18 000: 44.9 -> 41.6, N = 28 000: 130.7 -> 127.9, N = 40 000: 351.5 (groups of 3) against 356.8.
    // -1 = automatic: on for the factorisation from 12 to 63 panels when the panel step is cooperative and no panel_group is
    // forced; 0 / 1 = off / on (1 also switches the solve sweep's variant on)
    int lookahead = -1;
    // the cooperative panel step (ck_la.hip: k_panel_coop, option "panel_fused" bit 4): [0..15] its flags, [16] its error word
    unsigned* d_coop = nullptr;
    unsigned long long* d_stamps = nullptr;   // option "gemm_stamps": lifetime stamps of the trailing-update workgroups
    size_t n_stamps = 0;                      // (ck_debug_gemm_clock), 4 words per workgroup of the largest launch
    int64_t stamp_grid[4] = {0, 0, 0, 0};     // grid x, y, J0, panels of the last stamped launch
    int stamp_sel = 0;                        // 1: every launch | >= 2: only the trailing launch behind panel group K0 = stamp_sel - 2
    unsigned coop_seq = 0;
    unsigned coop_spins = 2000000;    // option "coop_spins": polls a workgroup of k_panel_coop spends on one flag before it gives up (~2 s)
    int coop_inject_panel = -1;       // option "coop_inject_panel" (tests): the cooperative step of this panel loses one flag store
};

extern "C" int ck_version(void) { return 100; }
extern "C" int ck_device_count(int* n) {
    HIPCHK(hipGetDeviceCount(n));
    return 0;
}

// ---------------------------------------------------------------------------------------
// memory
// ---------------------------------------------------------------------------------------
static int dev_alloc(ck_handle* h, void** out, int64_t bytes) {
    bytes = (bytes + 255) & ~(int64_t)255;
    if (h->arena) {
        if (h->arena_used + bytes > h->arena_size)
            return fail("arena too small: need " + std::to_string(h->arena_used + bytes) + " bytes, have " +
                        std::to_string(h->arena_size));
        *out = h->arena + h->arena_used;
        h->arena_used += bytes;
        return 0;
    }
    HIPCHK(hipMalloc(out, (size_t)bytes));
    h->owned.push_back(*out);
    return 0;
}
static void dev_free_one(ck_handle* h, void* p) {
    if (!p || h->arena) return;
    for (size_t i = 0; i < h->owned.size(); ++i)
        if (h->owned[i] == p) {
            (void)hipFree(p);
            h->owned.erase(h->owned.begin() + i);
            return;
        }
}

extern "C" int ck_create(int device_id, ck_handle** out) {
    if (!out) return fail("null out");
    int nd = 0;
    HIPCHK(hipGetDeviceCount(&nd));
    if (nd <= 0) return fail("no HIP device visible: libcokrige_hip needs an MI355X (gfx950)");
    if (device_id < 0 || device_id >= nd) return fail("bad device id");
    HIPCHK(hipSetDevice(device_id));
    ck_handle* h = new ck_handle();
    h->device = device_id;
    HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    HIPCHK(hipEventCreate(&h->ev0));
    HIPCHK(hipEventCreate(&h->ev1));
    HIPCHK(hipEventCreate(&h->ev2));
    HIPCHK(hipEventCreate(&h->ev3));
    {
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi));
    }
    HIPCHK(hipMalloc((void**)&h->d_coop, 32 * sizeof(unsigned)));
    HIPCHK(hipMemset(h->d_coop, 0, 32 * sizeof(unsigned)));
    HIPCHK(hipMalloc((void**)&h->d_blk, 3 * sizeof(CkMatern)));
    h->d_info = (long long*)(h->d_coop + 18);   // behind the cooperative step's error word: one 16-byte read-back for both
    *out = h;
    return 0;
}

extern "C" int ck_create_partitioned(const int* device_ids, int n_dev, int rank, ck_handle** out) {
    if (!device_ids || n_dev < 1 || rank < 0 || rank >= n_dev) return fail("bad device list / rank");
    if (ck_create(device_ids[rank], out)) return -1;
    (*out)->rank = rank;
    (*out)->world = n_dev;
    return 0;
}

static void vario_free(ck_handle* h);
static void schur_free(ck_handle* h);

extern "C" int ck_destroy(ck_handle* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    vario_free(h);
    schur_free(h);
    for (void* p : h->owned) (void)hipFree(p);
    (void)hipFree(h->d_blk);
    if (h->d_tile0) (void)hipFree(h->d_tile0);
    if (h->local_slab) (void)hipFree(h->local_slab);
    if (h->d_panel_of) (void)hipFree(h->d_panel_of);
    if (h->wl.items) (void)hipFree(h->wl.items);
    if (h->wl_counts) (void)hipFree(h->wl_counts);
    if (h->d_tabs) (void)hipFree(h->d_tabs);
    if (h->d_coefptr) (void)hipFree(h->d_coefptr);
    for (int b = 0; b < 3; ++b)
        if (h->d_coef[b]) (void)hipFree(h->d_coef[b]);
    for (auto& e : h->gemm_ev) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    (void)hipEventDestroy(h->ev0);
    (void)hipEventDestroy(h->ev1);
    (void)hipEventDestroy(h->ev2);
    (void)hipEventDestroy(h->ev3);
    if (h->mv_buf) (void)hipFree(h->mv_buf);
    for (auto e : h->ev_b2) (void)hipEventDestroy(e);
    for (auto e : h->ev_col) (void)hipEventDestroy(e);
    for (auto e : h->ev_pan) (void)hipEventDestroy(e);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->side_lo) (void)hipStreamDestroy(h->side_lo);
    if (h->d_coop) (void)hipFree(h->d_coop);
    if (h->d_stamps) (void)hipFree(h->d_stamps);
    (void)hipStreamDestroy(h->own_stream);
    delete h;
    return 0;
}

extern "C" int ck_set_stream(ck_handle* h, void* hip_stream, int external) {
    CHKH(h);
    h->stream = external ? (hipStream_t)hip_stream : h->own_stream;
    return 0;
}

extern "C" int ck_set_arena(ck_handle* h, void* dev_base, int64_t nbytes) {
    CHKH(h);
    if (h->layout_ready || !h->owned.empty()) return fail("ck_set_arena must precede any allocation");
    if (((uintptr_t)dev_base & 255) != 0) return fail("arena base must be 256-byte aligned");
    h->arena = (char*)dev_base;
    h->arena_size = nbytes;
    h->arena_used = 0;
    return 0;
}

extern "C" int ck_synchronize(ck_handle* h) {
    CHKH(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------
// model / data
// ---------------------------------------------------------------------------------------
extern "C" int ck_set_model(ck_handle* h, int n_procs, const double* sigma, const double* nu, const double* len_scale,
                            const double* nugget, double rho12) {
    CHKH(h);
    if (n_procs != 1 && n_procs != 2) return fail("n_procs must be 1 or 2");
    if (!sigma || !nu || !len_scale || !nugget) return fail("null parameter array");
    const int nb = n_procs == 1 ? 1 : 3;
    for (int k = 0; k < nb; ++k)
        if (!(nu[k] > 0.0) || !(len_scale[k] > 0.0)) return fail("nu and len_scale must be positive");
    ck_model_prepare(n_procs, sigma, nu, len_scale, nugget, rho12, h->blk);
    h->n_procs = n_procs;
    HIPCHK(hipMemcpyAsync(h->d_blk, h->blk, 3 * sizeof(CkMatern), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->model_set = true;
    h->assembled = h->factored = false;
    h->aux_state = 0;   // solved right-hand sides of the old model must not feed ck_verify_model
    return 0;
}

extern "C" int ck_set_metric(ck_handle* h, int metric) {
    CHKH(h);
    if (metric != CK_METRIC_HAVERSINE && metric != CK_METRIC_EUCLID) return fail("unknown metric");
    if (metric != h->metric) {
        h->metric = metric;
        h->layout_ready = false;   // site transforms depend on the metric
        h->assembled = h->factored = false;
        h->aux_state = 0;
    }
    return 0;
}

extern "C" int ck_set_partition(ck_handle* h, int rank, int world) {
    CHKH(h);
    if (world < 1 || rank < 0 || rank >= world) return fail("bad partition");
    if (h->layout_ready) return fail("ck_set_partition must precede ck_assemble_joint");
    h->rank = rank;
    h->world = world;
    return 0;
}

extern "C" int ck_set_data(ck_handle* h, int k, const double* coords, const double* values, int64_t n_k) {
    CHKH(h);
    if (k < 0 || k > 1) return fail("process index must be 0 or 1");
    if (n_k < 0 || (n_k > 0 && (!coords || !values))) return fail("bad data arrays");
    // also after ck_set_metric has invalidated the layout: the site arrays and panels keep their first sizes
    if (h->layout_ready || h->s0) return fail("data already laid out on the device; create a new handle to change it");
    h->h_coords[k].assign(coords, coords + 2 * n_k);
    h->h_values[k].assign(values, values + n_k);
    h->n[k] = n_k;
    h->data_set[k] = true;
    return 0;
}


// ---------------------------------------------------------------------------------------
// tabulated covariance: build C(q) = amp * rho tables for every Matern block of the model
// (ck_math.h "Tabulated covariance").  Node values come from the exact device evaluator;
// the Chebyshev fit per interval is done here in long double; the result is checked on the
// device against the exact evaluator (k_table_check) and disabled if it misses 2e-13.
// ---------------------------------------------------------------------------------------
static int build_tables(ck_handle* h, double qbox_euclid) {
    const int nblk = h->n_procs == 1 ? 1 : 3;
    if (!h->wl.items) {
        h->wl.cap = 1u << 22;   // 4 M deferred entries (32 MB); beyond that the assembly re-runs exactly
        HIPCHK(hipMalloc((void**)&h->wl.items, (size_t)h->wl.cap * sizeof(int2)));
        HIPCHK(hipMalloc((void**)&h->wl_counts, 4 * sizeof(unsigned)));   // two pairs (count, work queue), used alternately: see CkWorklist
        HIPCHK(hipMemset(h->wl_counts, 0, 4 * sizeof(unsigned)));
        h->wl.count = h->wl_counts;
        h->wl.reset = h->wl_counts + 2;
    }
    const int ND = CK_TAB_DEG + 1;
    if (!h->d_tabs) {
        HIPCHK(hipMalloc((void**)&h->d_tabs, 3 * sizeof(CkTable)));
        HIPCHK(hipMalloc((void**)&h->d_coefptr, 3 * sizeof(double*)));
        for (int b = 0; b < 3; ++b) HIPCHK(hipMalloc((void**)&h->d_coef[b], (size_t)ND * CK_TAB_STRIDE * 8));
    }
    DevTemps tmp;
    unsigned long long* d_err = nullptr;
    double *d_q = nullptr, *d_f = nullptr;
    HIPCHK(tmp.get(&d_err, (size_t)(8)));
    HIPCHK(tmp.get(&d_q, (size_t)((size_t)ND * CK_TAB_STRIDE * 8)));
    HIPCHK(tmp.get(&d_f, (size_t)((size_t)ND * CK_TAB_STRIDE * 8)));
    std::vector<double> hq(ND * CK_TAB_STRIDE), hf(ND * CK_TAB_STRIDE), hcoef(ND * CK_TAB_STRIDE);
    for (int b = 0; b < 3; ++b) {
        CkTable& T = h->tab[b];
        memset(&T, 0, sizeof(T));
        if (b >= nblk) {
            T = h->tab[0];
            continue;
        }
        const CkMatern& m = h->blk[b];
        int64_t base = 0;
        int n_int = ck_table_plan(&m, h->metric, qbox_euclid, &base, hq.data());
        if (n_int <= 0) continue;
        HIPCHK(hipMemcpyAsync(d_q, hq.data(), (size_t)n_int * ND * 8, hipMemcpyHostToDevice, h->stream));
        ck_launch_table_nodes(h->stream, h->d_blk + b, h->metric, d_q, (int64_t)n_int * ND, d_f);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(hf.data(), d_f, (size_t)n_int * ND * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        // node values must be finite (they are: rho underflows to 0 far in the tail)
        bool finite = true;
        for (int64_t i = 0; i < (int64_t)n_int * ND; ++i) finite = finite && (fabs(hf[i]) < 1e300);
        if (!finite) continue;
        ck_table_fit(hf.data(), n_int, base, hcoef.data());
        HIPCHK(hipMemcpyAsync(h->d_coef[b], hcoef.data(), (size_t)ND * CK_TAB_STRIDE * 8, hipMemcpyHostToDevice, h->stream));
        T.base = (int32_t)base;
        T.n_int = n_int;
        T.q_lo = ck_table_edge(base);
        T.q_hi = ck_table_edge(base + n_int);
        HIPCHK(hipMemsetAsync(d_err, 0, 8, h->stream));
        ck_launch_table_check(h->stream, h->d_blk + b, h->metric, T, h->d_coef[b], d_err);
        HIPCHK(hipGetLastError());
        unsigned long long eb = 0;
        HIPCHK(hipMemcpyAsync(&eb, d_err, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        memcpy(&T.max_rel_err, &eb, 8);
        T.enabled = (T.max_rel_err < 2e-13) ? 1 : 0;
    }
    HIPCHK(hipMemcpyAsync(h->d_tabs, h->tab, 3 * sizeof(CkTable), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_coefptr, h->d_coef, 3 * sizeof(double*), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->tables_built = true;
    return 0;
}

// (thread team, Hilbert order, bounding box, the reference's distance arithmetic, the variogram's level planning and tie
// decisions: ck_host.cpp -- host-only code, also built with the CPU sanitizers by tests/test_host_sanitize.py)
// Decide the padded layout, upload sites / values, allocate the owned panels.
// need_panels = false (ck_predict_local): sites, tables and chunk bounds only -- the local predictor never touches the Sigma
// panels or the receive buffers (a rank pool's local handle used to hold its share of a 40 GB Sigma beside the
// joint runner's arena); they are allocated by the first entry that needs them.
static int ensure_panels(ck_handle* h);
static int ensure_layout(ck_handle* h, bool need_panels = true) {
    if (h->layout_ready) return need_panels ? ensure_panels(h) : 0;
    if (!h->model_set) return fail("ck_set_model has not been called");
    for (int k = 0; k < h->n_procs; ++k)
        if (!h->data_set[k]) return fail("ck_set_data missing for process " + std::to_string(k));
    const int64_t n0 = h->n[0], n1 = h->n_procs == 2 ? h->n[1] : 0;
    h->N = n0 + n1;
    if (h->N <= 0) return fail("no observations");
    // process 1 starts on a 64-row tile boundary, so every assembly tile has ONE Matern block
    h->n0p = n1 > 0 ? roundup(n0, 64) : n0;
    h->nend = h->n0p + n1;
    h->Npad = roundup(h->nend, CK_NB);
    h->nK = (int)(h->Npad / CK_NB);
    const int64_t Np = h->Npad;
    if (!h->s0) {
        if (dev_alloc(h, (void**)&h->s0, 3 * Np * 8) || dev_alloc(h, (void**)&h->su, 3 * Np * 8) ||
            dev_alloc(h, (void**)&h->z, Np * 8))
            return -1;
        h->s1 = h->s0 + Np;
        h->s2 = h->s0 + 2 * Np;
    }
    // stage coords -> device, transform
    double blo[2] = {1e300, 1e300}, bhi[2] = {-1e300, -1e300};
    for (int k = 0; k < h->n_procs; ++k) ck_host_bounding_box(h->h_coords[k].data(), h->n[k], blo, bhi);
    std::vector<double> hc(2 * Np, 0.0), hz(Np, 0.0);
    for (int k = 0; k < h->n_procs; ++k) {
        const int64_t nk = h->n[k], off = k == 0 ? 0 : h->n0p;
        if (h->site_order) {
            ck_host_hilbert_order(h->h_coords[k].data(), nk, blo, bhi, h->perm[k]);
        } else {
            h->perm[k].resize((size_t)nk);
            for (int64_t j = 0; j < nk; ++j) h->perm[k][(size_t)j] = j;
        }
        for (int64_t j = 0; j < nk; ++j) {
            const int64_t e = h->perm[k][(size_t)j];
            hc[2 * (off + j)] = h->h_coords[k][2 * e];
            hc[2 * (off + j) + 1] = h->h_coords[k][2 * e + 1];
            hz[off + j] = h->h_values[k][e];
        }
    }
    DevTemps tmp;
    double* d_tmp = nullptr;
    HIPCHK(tmp.get(&d_tmp, (size_t)(2 * Np * 8)));
    HIPCHK(hipMemcpyAsync(d_tmp, hc.data(), 2 * Np * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->z, hz.data(), Np * 8, hipMemcpyHostToDevice, h->stream));
    ck_launch_prep_sites(h->stream, d_tmp, Np, h->metric, h->s0, h->s1, h->s2, h->su);
    if (!h->d_chunkb) {   // outside the arena: not part of ck_estimate_bytes
        HIPCHK(hipMalloc((void**)&h->d_chunkb, (size_t)(4 * ((h->nend + 255) / 256 + 1) * 8)));
        h->owned.push_back(h->d_chunkb);
    }
    ck_launch_local_chunk_bounds(h->stream, h->su, CkLayout{h->n[0], h->n0p, h->nend, h->Npad}, h->d_chunkb);
    HIPCHK(hipStreamSynchronize(h->stream));
    // squared bounding-box diagonal of the data sites (Euclidean table range)
    const double qbox = (bhi[0] - blo[0]) * (bhi[0] - blo[0]) + (bhi[1] - blo[1]) * (bhi[1] - blo[1]);
    if (build_tables(h, qbox)) return -1;
    h->layout_ready = true;
    return need_panels ? ensure_panels(h) : 0;
}

static int ensure_panels(ck_handle* h) {
    const int64_t Np = h->Npad;
    if (h->sig.empty()) {
        h->sig.assign(h->nK, nullptr);
        // the owned panels from ONE allocation (outside a caller's arena, where dev_alloc carves anyway): 79 hipMalloc /
        // hipFree calls of tens of MB each were most of what a cold pass spent in its first assemble and in ck_destroy
        // beyond the kernels (scripts/diag_cold_phases.py)
        auto panel_bytes = [&](int K) {
            return (((Np - (int64_t)K * CK_NB) * CK_NB + CK_PANEL_TAIL) * 8 + CK_PANEL_SLACK_BYTES + 255) & ~(int64_t)255;
        };
        int64_t slab_bytes = 0;
        for (int K = h->rank; K < h->nK; K += h->world) slab_bytes += panel_bytes(K);
        char* slab = nullptr;
        if (dev_alloc(h, (void**)&slab, slab_bytes)) return -1;
        for (int K = h->rank; K < h->nK; K += h->world) {
            h->sig[K] = (double*)slab;
            slab += panel_bytes(K);
        }
        if (dev_alloc(h, (void**)&h->d_sigptr, (int64_t)h->nK * sizeof(double*))) return -1;
        HIPCHK(hipMemcpy(h->d_sigptr, h->sig.data(), h->nK * sizeof(double*), hipMemcpyHostToDevice));
        std::vector<int> tile0, panel_of;
        int acc = 0;
        for (int K = h->rank; K < h->nK; K += h->world) {
            tile0.push_back(acc);
            panel_of.push_back(K);
            acc += (int)((Np - (int64_t)K * CK_NB) / 64);
        }
        tile0.push_back(acc);
        h->n_owned = (int)panel_of.size();
        h->total_tiles = acc;
        HIPCHK(hipMalloc((void**)&h->d_tile0, tile0.size() * sizeof(int)));
        HIPCHK(hipMalloc((void**)&h->d_panel_of, (panel_of.size() + 1) * sizeof(int)));
        HIPCHK(hipMemcpy(h->d_tile0, tile0.data(), tile0.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_panel_of, panel_of.data(), panel_of.size() * sizeof(int), hipMemcpyHostToDevice));
        {
            // work-queue form of the assembly: the 64-row strips of all owned panels sorted by Matern block (block 22 first, then
            // 12, then 11; inside a block back to front as before), so that a resident workgroup reloads its table twice per launch
            std::vector<int> order;
            order.reserve((size_t)acc);
            for (int cls = 2; cls >= 0; --cls)
                for (int j = (int)panel_of.size() - 1; j >= 0; --j) {
                    const int64_t K = panel_of[(size_t)j];
                    const int pc = K * CK_NB >= h->n0p ? 1 : 0;
                    for (int tile = tile0[(size_t)j + 1] - tile0[(size_t)j] - 1; tile >= 0; --tile) {
                        const int64_t rt = K * CK_NB + (int64_t)tile * 64;
                        if ((rt >= h->n0p ? 1 : 0) + pc == cls) order.push_back(tile0[(size_t)j] + tile);
                    }
                }
            HIPCHK(hipMalloc((void**)&h->d_strip_order, std::max<size_t>(order.size(), 1) * sizeof(int)));
            h->owned.push_back(h->d_strip_order);
            HIPCHK(hipMemcpy(h->d_strip_order, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        if (h->world > 1) {
            h->recv.assign((size_t)h->recv_slots, nullptr);
            for (int b = 0; b < h->recv_slots; ++b)
                if (dev_alloc(h, (void**)&h->recv[(size_t)b], (Np * CK_NB + CK_PANEL_TAIL) * 8 + CK_PANEL_SLACK_BYTES)) return -1;
        }
        {
            std::vector<double*> pp(h->nK);
            for (int K = 0; K < h->nK; ++K) pp[K] = h->sig[K] ? h->sig[K] : (h->world > 1 ? h->recv[(size_t)(K % h->recv_slots)] : nullptr);
            if (dev_alloc(h, (void**)&h->d_panelptr, (int64_t)h->nK * sizeof(double*))) return -1;
            HIPCHK(hipMemcpy(h->d_panelptr, pp.data(), h->nK * sizeof(double*), hipMemcpyHostToDevice));
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// element-wise parity surface
// ---------------------------------------------------------------------------------------
static int blk_index(ck_handle* h, int i, int j, int* idx) {
    if (!h->model_set) return fail("ck_set_model has not been called");
    if (i < 0 || j < 0 || i >= h->n_procs || j >= h->n_procs) return fail("process index out of range");
    *idx = i + j;
    return 0;
}

static int dense_common(ck_handle* h, int bidx, int add_nugget, int mode, const double* A, int64_t a, const double* B,
                        int64_t b, double* out) {
    if (a <= 0 || b <= 0) return 0;
    DevTemps tmp;
    double *dA = nullptr, *dB = nullptr, *dO = nullptr, *ta = nullptr, *tb = nullptr;
    HIPCHK(tmp.get(&dA, (size_t)(2 * a * 8)));
    HIPCHK(tmp.get(&dB, (size_t)(2 * b * 8)));
    HIPCHK(tmp.get(&ta, (size_t)(3 * a * 8)));
    HIPCHK(tmp.get(&tb, (size_t)(3 * b * 8)));
    HIPCHK(tmp.get(&dO, (size_t)(a * b * 8)));
    HIPCHK(hipMemcpyAsync(dA, A, 2 * a * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dB, B, 2 * b * 8, hipMemcpyHostToDevice, h->stream));
    ck_launch_prep_sites(h->stream, dA, a, h->metric, ta, ta + a, ta + 2 * a, nullptr);
    ck_launch_prep_sites(h->stream, dB, b, h->metric, tb, tb + b, tb + 2 * b, nullptr);
    ck_launch_cov_dense(h->stream, h->d_blk + bidx, h->metric, add_nugget, mode, ta, ta + a, ta + 2 * a, a, tb, tb + b,
                        tb + 2 * b, b, dO);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dO, a * b * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int ck_distance_dense(ck_handle* h, const double* A, int64_t a, const double* B, int64_t b, double* out) {
    CHKH(h);
    return dense_common(h, 0, 0, 1, A, a, B, b, out);
}

extern "C" int ck_cov_dense(ck_handle* h, int i, int j, const double* A, int64_t a, const double* B, int64_t b,
                            int use_nugget, double* out) {
    CHKH(h);
    int bidx;
    if (blk_index(h, i, j, &bidx)) return -1;
    return dense_common(h, bidx, (i == j) && use_nugget, 0, A, a, B, b, out);
}

extern "C" int ck_cov_lags(ck_handle* h, int i, int j, const double* lags, int64_t n, int use_nugget, double* out) {
    CHKH(h);
    int bidx;
    if (blk_index(h, i, j, &bidx)) return -1;
    if (n <= 0) return 0;
    DevTemps tmp;
    double *dl = nullptr, *dO = nullptr;
    HIPCHK(tmp.get(&dl, (size_t)(n * 8)));
    HIPCHK(tmp.get(&dO, (size_t)(n * 8)));
    HIPCHK(hipMemcpyAsync(dl, lags, n * 8, hipMemcpyHostToDevice, h->stream));
    ck_launch_cov_lags(h->stream, h->d_blk + bidx, (i == j) && use_nugget, dl, n, dO);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dO, n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int ck_model_variogram(ck_handle* h, const int32_t* pi, const int32_t* pj, const double* lags, int64_t n,
                                  int kind, double* out) {
    CHKH(h);
    if (!h->model_set) return fail("ck_set_model has not been called");
    if (kind != 0 && kind != 1) return fail("kind must be 0 (semivariogram) or 1 (covariogram)");
    if (n <= 0) return 0;
    if (!pi || !pj || !lags || !out) return fail("null array");
    for (int64_t r = 0; r < n; ++r)
        if (pi[r] < 0 || pj[r] < 0 || pi[r] >= h->n_procs || pj[r] >= h->n_procs)
            return fail("process index out of range in row " + std::to_string(r));
    // sill of the cross-semivariogram: nansum(sigma^2 + nugget) / 2 over the parameter arrays (model.py:219-221)
    double sill = h->blk[0].amp + h->blk[0].nugget;
    if (h->n_procs == 2) sill = 0.5 * (sill + (h->blk[2].amp + h->blk[2].nugget));
    else sill = 0.5 * sill;
    if (n > h->mv_cap) {   // one allocation for all four arrays; a fit calls this hundreds of times with the same n
        if (h->mv_buf) (void)hipFree(h->mv_buf);
        h->mv_buf = nullptr;
        h->mv_cap = 0;
        const int64_t cap = roundup(n, 256);
        HIPCHK(hipMalloc((void**)&h->mv_buf, (size_t)cap * 24));
        h->mv_cap = cap;
    }
    double* dl = reinterpret_cast<double*>(h->mv_buf);
    double* dO = dl + h->mv_cap;
    int* di = reinterpret_cast<int*>(dO + h->mv_cap);
    int* dj = di + h->mv_cap;
    HIPCHK(hipMemcpyAsync(di, pi, n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dj, pj, n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dl, lags, n * 8, hipMemcpyHostToDevice, h->stream));
    ck_launch_model_variogram(h->stream, h->d_blk, sill, kind, di, dj, dl, n, dO);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dO, n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------
// joint path
// ---------------------------------------------------------------------------------------
static CkLayout layout_of(const ck_handle* h) { return CkLayout{h->n[0], h->n0p, h->nend, h->Npad}; }
static bool tables_usable(const ck_handle* h) {
    if (h->exact_cov) return false;
    const int nblk = h->n_procs == 1 ? 1 : 3;
    for (int b = 0; b < nblk; ++b)
        if (!h->tab[b].enabled) return false;
    return true;
}
// The worklist of the next table-path assembly: its counter is zero already -- the exact pass of the previous assembly
// (k_assemble_fix) zeroed it, or the allocation did -- and ITS exact pass will zero the other one.  No memset launch.
static void next_worklist(ck_handle* h) {
    unsigned* cur = h->wl.reset;
    h->wl.reset = h->wl.count;
    h->wl.count = cur;
}

// internal (padded-order) 1-based index -> index in the caller's stacked order
static int64_t external_index(const ck_handle* h, int64_t g) { return g > h->n0p ? g - (h->n0p - h->n[0]) : g; }

extern "C" int ck_assemble_joint(ck_handle* h) {
    CHKH(h);
    if (ensure_layout(h)) return -1;

    HIPCHK(hipEventRecord(h->ev0, h->stream));
    bool fast_done = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool fast = tables_usable(h) && attempt == 0;
        if (fast) next_worklist(h);   // a zeroed counter, without a memset launch in front of the assembly
        {
            CkPanelMap pm{h->d_tile0, h->d_panel_of, h->d_sigptr, h->n_owned, nullptr, 0, h->d_strip_order};
            ck_launch_assemble_sigma(h->stream, fast, h->d_blk, h->d_tabs, h->d_coefptr, h->metric, h->s0, h->su,
                                     layout_of(h), pm, h->total_tiles, h->wl, fast ? h->assemble_queue : 0);
        }
        if (!fast) break;
        ck_launch_assemble_fix(h->stream, false, h->d_blk, h->metric, 0, nullptr, 0, h->s0, layout_of(h), h->wl,
                               h->d_sigptr, nullptr);
        HIPCHK(hipEventRecord(h->ev1, h->stream));   // table kernel + exact pass; the count check below is host latency
        unsigned cnt = 0;
        HIPCHK(hipMemcpyAsync(&cnt, h->wl.count, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->fallback_total += cnt;
        if (cnt <= h->wl.cap) {
            fast_done = true;
            break;
        }   // else: too many out-of-table pairs for the list -> exact kernels
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(h->d_info, 0, sizeof(long long), h->stream));
    if (!fast_done) HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->t_ms[0] = ms;
    h->assembled = true;
    h->factored = false;
    h->aux_state = 0;   // right-hand sides solved with the previous factor are stale from here on
    return 0;
}

extern "C" int ck_num_panels(ck_handle* h, int* n_panels, int* panel_width, int64_t* n_padded) {
    CHKH(h);
    if (ensure_layout(h)) return -1;
    if (n_panels) *n_panels = h->nK;
    if (panel_width) *panel_width = CK_NB;
    if (n_padded) *n_padded = h->Npad;
    return 0;
}

extern "C" int ck_panel_owner(ck_handle* h, int K, int* owner_rank) {
    CHKH(h);
    if (K < 0) return fail("bad panel index");
    *owner_rank = K % h->world;
    return 0;
}

static const double* panel_src(ck_handle* h, int K) {
    return h->sig[K] ? h->sig[K] : (h->recv.empty() ? nullptr : h->recv[(size_t)(K % h->recv_slots)]);
}

extern "C" int ck_panel_buffer(ck_handle* h, int K, void** dev_ptr, int64_t* nbytes) {
    CHKH(h);
    if (ensure_layout(h)) return -1;
    if (K < 0 || K >= h->nK) return fail("bad panel index");
    *dev_ptr = (void*)panel_src(h, K);
    *nbytes = ((h->Npad - (int64_t)K * CK_NB) * CK_NB + CK_PANEL_TAIL) * 8;   // rows + the diagonal blocks' inverses
    return 0;
}

static void gemm_timed_begin(ck_handle* h, hipStream_t st = nullptr) {
    if (!st) st = h->stream;
    if (!h->time_gemm) return;
    if (h->gemm_ev_used == h->gemm_ev.size()) {
        EvPair e;
        (void)hipEventCreate(&e.a);
        (void)hipEventCreate(&e.b);
        h->gemm_ev.push_back(e);
    }
    (void)hipEventRecord(h->gemm_ev[h->gemm_ev_used].a, st);
}
static void gemm_timed_end(ck_handle* h, hipStream_t st = nullptr) {
    if (!st) st = h->stream;
    if (!h->time_gemm) return;
    (void)hipEventRecord(h->gemm_ev[h->gemm_ev_used].b, st);
    h->gemm_ev_used++;
}

// sum the event pairs recorded since the last reset into t_ms[slot], t_ms[slot + 1]
// ref != nullptr (the tall sweep: launches of two streams overlap each other): also the UNION of the launches' intervals --
// the time during which the kernel is running at all -- into t_ms[15], from every launch's start / end relative to `ref`
static void gemm_timed_collect(ck_handle* h, int slot, hipEvent_t ref = nullptr) {
    if (!h->time_gemm) return;
    double tot = 0;
    std::vector<std::pair<float, float>> iv;
    for (size_t e = 0; e < h->gemm_ev_used; ++e) {
        float t = 0;
        (void)hipEventElapsedTime(&t, h->gemm_ev[e].a, h->gemm_ev[e].b);
        tot += t;
        if (ref) {
            float a = 0, b = 0;
            (void)hipEventElapsedTime(&a, ref, h->gemm_ev[e].a);
            (void)hipEventElapsedTime(&b, ref, h->gemm_ev[e].b);
            iv.push_back({a, b});
        }
    }
    h->t_ms[slot] = tot;
    h->t_ms[slot + 1] = (double)h->gemm_ev_used;
    if (ref) {
        std::sort(iv.begin(), iv.end());
        double un = 0, cs = 0, ce = -1;
        for (const auto& x : iv) {
            if (x.first > ce) {
                if (ce >= 0) un += ce - cs;
                cs = x.first;
                ce = x.second;
            } else if (x.second > ce) {
                ce = x.second;
            }
        }
        if (ce >= 0) un += ce - cs;
        h->t_ms[15] = un;
    }
    h->gemm_ev_used = 0;
}

// ---- the building blocks, on an explicit stream --------------------------------------------------
// two-level panel step on block column K: 8 x (64 x 64 Cholesky, row solves, K = 64 update)
static void syrk_update(ck_handle* h, hipStream_t st, int K0, int np, int J0, int Jstep, int nJ) {
    unsigned long long* stamps = nullptr;
    if (h->d_stamps) {   // diagnostic: only launches that fit the stamp buffer are stamped
        const size_t wgs = (size_t)((h->Npad - (int64_t)J0 * CK_NB) / 128) * (CK_NB / 128) * (size_t)nJ;
        if (wgs <= h->n_stamps && (h->stamp_sel == 1 || (nJ > 1 && K0 == h->stamp_sel - 2))) {
            stamps = h->d_stamps;
            h->stamp_grid[0] = (int64_t)ck_tilemap_make(h->nend, J0, Jstep, nJ).total;   // the launch's grid: one row
            h->stamp_grid[1] = 1;
            h->stamp_grid[2] = J0;
            h->stamp_grid[3] = np;
        }
    }
    ck_launch_syrk_group(st, h->d_sigptr, h->d_panelptr, K0, np, J0, Jstep, nJ, h->Npad, h->nend, stamps);
}

// with_aux (the tall sweep, cooperative panel step only): the right-hand-side rows of block column K walk through the panel as
// further workgroups of the same launch
// split (with_aux only): the cooperative launch on the 512 x 512 head alone, then every row below it and every right-hand-side row
// through k_panel_rows_all -- a chunk below the head spins until the head's chunks have published, and while it spins it holds a
// slot the bulk update running beside the chain would use; k_panel_rows_all's workgroups wait for nobody (same arithmetic per row).
static void panel_factor_on(ck_handle* h, int K, hipStream_t st, bool with_aux = false, bool split = false) {
    double* P = h->sig[K];
    const int64_t R = h->Npad - (int64_t)K * CK_NB;
    double* tail = P + R * CK_NB;   // inverses of the eight diagonal blocks (CK_PANEL_TAIL)
    if (h->panel_fused & 16) {
        // the whole panel step in ONE launch of cooperating workgroups (ck_la.hip: k_panel_coop): a chunk of rows waits for
        // the one chunk above it in the chain instead of for 24 grid-wide launch boundaries
        h->coop_seq += 1;
        int drop = -1;
        if (h->coop_inject_panel == K) {   // test hook (option "coop_inject_panel"): chunk 3 of this panel never publishes, once
            drop = 3;
            h->coop_inject_panel = -1;
        }
        double* X = with_aux ? h->aux + (int64_t)K * h->mpad * CK_NB : nullptr;
        if (split && with_aux && R > CK_NB) {
            ck_launch_panel_coop(st, P, CK_NB, tail, (int64_t)K * CK_NB, h->d_info, h->d_coop, h->coop_seq, h->d_coop + 16, nullptr, 0,
                                 h->coop_spins, drop);
            ck_launch_panel_rows_all(st, P + (int64_t)CK_NB * CK_NB, R - CK_NB, P, tail, X, h->mpad);
            return;
        }
        ck_launch_panel_coop(st, P, R, tail, (int64_t)K * CK_NB, h->d_info, h->d_coop, h->coop_seq, h->d_coop + 16, X, with_aux ? h->mpad : 0,
                             h->coop_spins, drop);
        return;
    }
    if (h->panel_fused & 4) {
        // The 512 x 512 diagonal block first -- right-looking over its eight 64-column sub-blocks, on its own 512
        // rows only (small launches: 64 x 64 Cholesky + inverse, row solves, K = 64 update) -- then every row below
        // it walks through the factored block on its own, in ONE launch (the kernel of the right-hand-side rows):
        // 25 launches per panel, 24 of them on 512 rows, instead of 24 launches over all rows of the panel.
        const int64_t RD = std::min<int64_t>(R, CK_NB);
        for (int q = 0; q < CK_NB / CK_IB; ++q) {
            double* diag = P + (int64_t)q * CK_IB * CK_NB + q * CK_IB;
            double* linv = tail + (int64_t)q * CK_IB * CK_IB;
            ck_launch_potrf64(st, diag, CK_NB, (int64_t)K * CK_NB + q * CK_IB, h->d_info, linv);
            const int64_t r1 = (int64_t)(q + 1) * CK_IB;
            if (RD > r1) ck_launch_trsm64(st, P + r1 * CK_NB + q * CK_IB, CK_NB, RD - r1, linv);
            const int64_t ncols = CK_NB - r1;
            if (ncols > 0 && RD > r1) {
                const int64_t ra = r1 / CK_BM * CK_BM;
                ck_launch_gemm_nt(st, P + ra * CK_NB + r1, CK_NB, P + ra * CK_NB + q * CK_IB, CK_NB,
                                  P + r1 * CK_NB + q * CK_IB, CK_NB, RD - ra, ncols, CK_IB, 1, ra - r1, 1, 0, 0, 0);
            }
        }
        if (R > CK_NB) ck_launch_panel_rows_all(st, P + (int64_t)CK_NB * CK_NB, R - CK_NB, P, tail);
        return;
    }
    if (h->panel_fused & 1) {
        for (int q = 0; q < CK_NB / CK_IB; ++q) {
            double* linv = tail + (int64_t)q * CK_IB * CK_IB;
            ck_launch_panel_diag(st, P, q, (int64_t)K * CK_NB, h->d_info, linv);
            const int64_t r1 = (int64_t)(q + 1) * CK_IB;
            ck_launch_panel_rows(st, P, r1, R - r1, P, q, linv);
        }
        return;
    }
    for (int q = 0; q < CK_NB / CK_IB; ++q) {
        double* diag = P + (int64_t)q * CK_IB * CK_NB + q * CK_IB;
        double* linv = tail + (int64_t)q * CK_IB * CK_IB;
        ck_launch_potrf64(st, diag, CK_NB, (int64_t)K * CK_NB + q * CK_IB, h->d_info, linv);
        const int64_t r1 = (int64_t)(q + 1) * CK_IB;
        ck_launch_trsm64(st, P + r1 * CK_NB + q * CK_IB, CK_NB, R - r1, linv);
        const int64_t ncols = CK_NB - r1;
        if (ncols > 0) {
            const int64_t ra = r1 / CK_BM * CK_BM;   // tile-aligned start row (rows above r1 only touch the unused upper triangle)
            ck_launch_gemm_nt(st, P + ra * CK_NB + r1, CK_NB, P + ra * CK_NB + q * CK_IB, CK_NB,
                              P + r1 * CK_NB + q * CK_IB, CK_NB, R - ra, ncols, CK_IB, 1, ra - r1, 1, 0, 0, 0);
        }
    }
}

// trailing update of the locally owned block columns J in [Jlo, Jhi] by panel K
static void apply_sigma_on(ck_handle* h, int K, const double* P, int Jlo, int Jhi, hipStream_t st, bool timed) {
    int J0 = Jlo;
    while (J0 <= Jhi && (J0 % h->world) != h->rank) ++J0;
    if (J0 > Jhi) return;
    const int nJ = (Jhi - J0) / h->world + 1;
    (void)P;   // the group kernel reads panel K through d_panelptr[K] (own storage or receive buffer)
    if (timed) gemm_timed_begin(h, st);
    syrk_update(h, st, K, 1, J0, h->world, nJ);
    if (timed) gemm_timed_end(h, st);
}

// forward substitution of the right-hand-side rows with the diagonal block of panel K
// Right-hand-side rows that can be nonzero in the columns of panels <= K (a multiple of CK_AUX_ALIGN).  Prediction: all.
// Leave-one-out (ck_loocv): row 0 is dense, row 1 + p starts at column loo_g0 + p -- the rows are sorted by their
// first nonzero column, so the sweep works on a growing prefix (process 0: 58 % of the full sweep's flops,
// process 1: 8 %).
static int64_t aux_rows(const ck_handle* h, int K) {
    if (h->loo_g0 < 0) return h->mpad;
    const int64_t live = (int64_t)(K + 1) * CK_NB - h->loo_g0 + 1;   // rows 0 .. live - 1
    return std::min(h->mpad, roundup(std::max<int64_t>(live, 1), CK_AUX_ALIGN));
}

// rows of the right-hand-side block in front of its padding, for the thin last tile row (0: treat every row as live)
static int64_t aux_live(const ck_handle* h) { return h->loo_g0 < 0 && h->tall_thin ? h->m + 1 : 0; }

static void aux_inner_on(ck_handle* h, int K, const double* P, hipStream_t st) {
    double* X = h->aux + (int64_t)K * h->mpad * CK_NB;
    const double* tail = P + (h->Npad - (int64_t)K * CK_NB) * CK_NB;
    const int64_t rows = aux_rows(h, K);
    if (h->panel_fused & 2) {
        ck_launch_panel_rows_all(st, X, rows, P, tail);
        return;
    }
    for (int q = 0; q < CK_NB / CK_IB; ++q) {
        ck_launch_trsm64(st, X + q * CK_IB, CK_NB, rows, tail + (int64_t)q * CK_IB * CK_IB);
        const int64_t r1 = (int64_t)(q + 1) * CK_IB;
        const int64_t ncols = CK_NB - r1;
        if (ncols > 0)
            ck_launch_gemm_nt(st, X + r1, CK_NB, X + q * CK_IB, CK_NB, P + r1 * CK_NB + q * CK_IB, CK_NB, rows,
                              ncols, CK_IB, 0, 0, 1, 0, 0, 0);
    }
}

// aux[J] -= aux[K] * P[(J-K)*NB .., :]^T for J in [Jlo, Jhi], batched over J
static void aux_update_on(ck_handle* h, int K, const double* P, int Jlo, int Jhi, hipStream_t st, bool timed) {
    const int nJ = Jhi - Jlo + 1;
    if (nJ <= 0) return;
    const int64_t rows = aux_rows(h, K);
    timed = timed && h->time_gemm == 1;   // 2: only the Sigma updates are timed (one event list per sweep)
    (void)P;
    if (timed) gemm_timed_begin(h, st);
    ck_launch_aux_group(st, h->aux, h->mpad, h->d_panelptr, K, 1, Jlo, nJ, rows, h->nend, aux_live(h));
    if (timed) gemm_timed_end(h, st);
}

static int ensure_events(ck_handle* h) {
    while ((int)h->ev_col.size() < h->nK + 1) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        h->ev_col.push_back(a);
        h->ev_pan.push_back(b);
    }
    while ((int)h->ev_b2.size() < h->nK + 1) {
        hipEvent_t a;
        HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        h->ev_b2.push_back(a);
    }
    return 0;
}

extern "C" int ck_panel_factor(ck_handle* h, int K) {
    CHKH(h);
    if (!h->assembled) return fail("ck_assemble_joint has not been called");
    if (K < 0 || K >= h->nK) return fail("bad panel index");
    if (!h->sig[K]) return fail("panel " + std::to_string(K) + " is not owned by this rank");
    panel_factor_on(h, K, h->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int ck_panel_apply(ck_handle* h, int K, int what) {
    CHKH(h);
    if (K < 0 || K >= h->nK) return fail("bad panel index");
    const double* P = panel_src(h, K);
    if (what & CK_APPLY_SIGMA) apply_sigma_on(h, K, P, K + 1, h->nK - 1, h->stream, true);
    if ((what & CK_APPLY_AUX) && h->mpad > 0) {
        aux_inner_on(h, K, P, h->stream);
        aux_update_on(h, K, P, K + 1, h->nK - 1, h->stream, true);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int ck_panel_apply_sigma(ck_handle* h, int K, int J_lo, int J_hi) {
    CHKH(h);
    if (K < 0 || K >= h->nK) return fail("bad panel index");
    J_lo = std::max(J_lo, K + 1);
    J_hi = std::min(J_hi, h->nK - 1);
    if (J_lo > J_hi) return 0;
    apply_sigma_on(h, K, panel_src(h, K), J_lo, J_hi, h->stream, true);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int ck_panel_apply_group(ck_handle* h, int K0, int np, int what, int J_lo, int J_hi, int phase, int n_phase) {
    CHKH(h);
    if (K0 < 0 || np < 1 || K0 + np > h->nK) return fail("bad panel group");
    if (n_phase < 1 || phase < 0 || phase >= n_phase) return fail("bad phase");
    for (int p = 0; p < np; ++p)
        if (!panel_src(h, K0 + p)) return fail("panel " + std::to_string(K0 + p) + " is not readable on this rank");
    if (h->world > 1 && np > h->recv_slots) return fail("panel group larger than recv_slots");
    J_lo = std::max(J_lo, K0 + np);
    J_hi = std::min(J_hi, h->nK - 1);
    if (J_lo > J_hi) return 0;
    if (what & CK_APPLY_SIGMA) {
        // the owned block columns of [J_lo, J_hi], every n_phase-th of them starting with the phase-th
        int J0 = J_lo;
        while (J0 <= J_hi && (J0 % h->world) != h->rank) ++J0;
        J0 += phase * h->world;
        if (J0 <= J_hi) {
            const int step = h->world * n_phase;
            const int nJ = (J_hi - J0) / step + 1;
            gemm_timed_begin(h);
            syrk_update(h, h->stream, K0, np, J0, step, nJ);
            gemm_timed_end(h);
        }
    }
    if ((what & CK_APPLY_AUX) && h->mpad > 0) {
        // right-hand-side block columns: piece `phase` of n_phase contiguous pieces of [J_lo, J_hi]
        const int tot = J_hi - J_lo + 1, per = (tot + n_phase - 1) / n_phase;
        const int a = J_lo + phase * per, b = std::min(J_hi, a + per - 1);
        if (a <= b) {
            const bool timed = h->time_gemm == 1;
            if (timed) gemm_timed_begin(h);
            ck_launch_aux_group(h->stream, h->aux, h->mpad, h->d_panelptr, K0, np, a, b - a + 1, aux_rows(h, K0 + np - 1), h->nend, aux_live(h));
            if (timed) gemm_timed_end(h);
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int ck_panel_aux_solve(ck_handle* h, int K) {
    CHKH(h);
    if (K < 0 || K >= h->nK) return fail("bad panel index");
    const double* P = panel_src(h, K);
    if (!P) return fail("panel " + std::to_string(K) + " is not readable on this rank");
    if (h->mpad > 0) aux_inner_on(h, K, P, h->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

// the info word alone (ck_factor / ck_factor_predict look at the cooperative step's error word themselves)
static int factor_info_raw(ck_handle* h, int64_t* info) {
    long long v = 0;
    HIPCHK(hipMemcpyAsync(&v, h->d_info, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *info = external_index(h, (int64_t)v);   // pivots inside the identity padding cannot fail
    return 0;
}

extern "C" int ck_factor_info(ck_handle* h, int64_t* info) {
    CHKH(h);
    long long v = 0;
    HIPCHK(hipMemcpyAsync(&v, h->d_info, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->panel_fused & 16) {                     // step-wise form: a timed-out cooperative panel step is an error here
        unsigned werr = 0;                         // (ck_factor repeats the factorisation instead)
        HIPCHK(hipMemcpy(&werr, h->d_coop + 16, sizeof(werr), hipMemcpyDeviceToHost));
        if (werr != 0) {
            HIPCHK(hipMemset(h->d_coop + 16, 0, sizeof(unsigned)));
            h->panel_fused &= ~16;
            return fail("cooperative panel step timed out waiting for a pivot block (option panel_fused bit 4 now off): sweep again");
        }
    }
    *info = external_index(h, (int64_t)v);   // pivots inside the identity padding cannot fail
    return 0;
}

// automatic: 4 from 40 panels on (round 4; 3 in rounds 1-3: interleaved A/B at N = 40 000, G = 3 / 4 / 5 / 6 with their best first
// groups: 491.0 / 488.2 / 489.5 / 489.2 ms), 1 below
// automatic group size (measured with scripts/ab_tall.py, round 4: groups of two pay from 14 panels -- N = 10 000: 23.2 -> 22.4 ms --,
// groups of four from 40)
static int eff_group(const ck_handle* h) { return h->panel_group > 0 ? h->panel_group : (h->nK >= 40 ? 4 : h->nK >= 14 ? 2 : 1); }

// Group boundaries of the single-process sweeps (every form -- ck_factor / ck_predict, the two overlapped sweeps, the tall
// sweep -- takes them from here).  The grouping does not touch the results: an element's updates are accumulated k ascending inside
// a launch and the tile is stored and re-read exactly between launches, so any split performs the same MFMAs per element.
// Groups of G panels; options "group_first" (panels of the FIRST group, 0 = G: a short first group shortens the one chain
// nothing can hide, at the price of one pass over the matrix with a short K) and "group_tail" / "group_tail_panels" (group
// size for the last so-many panels, where a group's updates are shorter than its chain).  starts[g] .. starts[g + 1] - 1.
static std::vector<int> group_plan(const ck_handle* h, int G) {
    std::vector<int> st;
    G = std::max(1, G);
    const int nK = h->nK;
    int K = 0;
    // automatic (-1): from 40 panels on the first group has G / 2 panels -- the one chain nothing can hide under is half as long,
    // for one pass over the matrix with K = 512 G / 2 (G = 4: 489.3 -> 488.2 ms)
    const int gf = h->group_first >= 0 ? h->group_first : (nK >= 40 ? G / 2 : 0);
    if (gf > 0 && gf < G && nK > gf) {
        st.push_back(0);
        K = gf;
    }
    const int tailG = h->group_tail > 0 ? std::min(h->group_tail, G) : G;
    while (K < nK) {
        st.push_back(K);
        K += (nK - K <= h->group_tail_panels) ? tailG : G;
    }
    st.push_back(nK);
    return st;
}

static bool factor_lookahead(const ck_handle* h) {
    if (h->lookahead >= 0) return h->lookahead != 0;
    return (h->panel_fused & 16) && h->panel_group == 0 && h->world == 1 && h->nK >= 12 && h->nK < 64;
}

static int factor_sweep(ck_handle* h) {
    h->gemm_ev_used = 0;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    if (factor_lookahead(h)) {
        // Look-ahead: as soon as panel K has updated block column K + 1, the panel step of K + 1
        // starts on the side stream and runs under the update of the columns K + 2.. by panel K.
        if (ensure_events(h)) return -1;
        hipStream_t M = h->stream, S = h->side;
        panel_factor_on(h, 0, M);
        for (int K = 0; K < h->nK; ++K) {
            if (K > 0) HIPCHK(hipStreamWaitEvent(M, h->ev_pan[K], 0));
            if (K + 1 < h->nK) {
                apply_sigma_on(h, K, h->sig[K], K + 1, K + 1, M, false);
                HIPCHK(hipEventRecord(h->ev_col[K + 1], M));
                HIPCHK(hipStreamWaitEvent(S, h->ev_col[K + 1], 0));
                panel_factor_on(h, K + 1, S);
                HIPCHK(hipEventRecord(h->ev_pan[K + 1], S));
                apply_sigma_on(h, K, h->sig[K], K + 2, h->nK - 1, M, true);
            }
        }
        HIPCHK(hipGetLastError());
    } else if (eff_group(h) <= 1) {
        for (int K = 0; K < h->nK; ++K) {
            if (ck_panel_factor(h, K)) return -1;
            if (ck_panel_apply(h, K, CK_APPLY_SIGMA)) return -1;
        }
    } else {
        // Groups of G panels: inside a group block column K first receives the updates of the group's
        // earlier panels in one pass (K dimension 512 g), then its panel step; the trailing matrix
        // beyond the group is updated once with K = 512 G -- a quarter of the C traffic of G = 1.
        const std::vector<int> gs = group_plan(h, eff_group(h));
        for (size_t gi = 0; gi + 1 < gs.size(); ++gi) {
            const int K0 = gs[gi], Gc = gs[gi + 1] - gs[gi];
            for (int g = 0; g < Gc; ++g) {
                if (g > 0) {
                    gemm_timed_begin(h);
                    syrk_update(h, h->stream, K0, g, K0 + g, 1, 1);
                    gemm_timed_end(h);
                }
                panel_factor_on(h, K0 + g, h->stream);
            }
            if (K0 + Gc < h->nK) {
                gemm_timed_begin(h);
                syrk_update(h, h->stream, K0, Gc, K0 + Gc, 1, h->nK - K0 - Gc);
                gemm_timed_end(h);
            }
        }
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    return 0;
}

// The same sweep with the look-ahead of tall_sweeps (round 4; ck_predict on a resident factor -- the second field of a
// Predictor, the sites of a new grid): stream C (high priority) carries the chain of group g -- the one-column in-group updates
// (280 tiles: they fill 55 % of the chip) and the rows' walk through each panel -- and A(g), the update of the NEXT group's block
// columns; stream T carries B1(g) / B2(g), the update of everything beyond.  The chain of group g + 1 runs under the bulk of group g
// instead of in front of it.  Same launches on the same operands in the same order per block column: same bits.
static int solve_sweep_la(ck_handle* h) {
    if (ensure_events(h)) return -1;
    if (!h->side_lo) HIPCHK(hipStreamCreateWithFlags(&h->side_lo, hipStreamNonBlocking));
    hipStream_t C = h->side, T = h->side_lo, M = h->stream;
    const std::vector<int> gs = group_plan(h, eff_group(h));
    const int ng = (int)gs.size() - 1;
    auto first = [&](int g) { return gs[(size_t)g]; };
    auto count = [&](int g) { return gs[(size_t)g + 1] - gs[(size_t)g]; };
    auto update = [&](hipStream_t st, int K0, int np, int J0, int nJ) {
        if (nJ <= 0) return;
        gemm_timed_begin(h, st);
        ck_launch_aux_group(st, h->aux, h->mpad, h->d_panelptr, K0, np, J0, nJ, aux_rows(h, K0 + np - 1), h->nend, aux_live(h));
        gemm_timed_end(h, st);
    };
    const bool b2m = h->tall_b2_stream != 0;
    HIPCHK(hipEventRecord(h->ev0, M));   // (ev0 / ev1 are free between ck_aux_begin and ck_aux_finish)
    HIPCHK(hipStreamWaitEvent(C, h->ev0, 0));
    HIPCHK(hipStreamWaitEvent(T, h->ev0, 0));
    for (int g = 0; g < ng; ++g) {
        const int K0 = first(g), Gc = count(g);
        for (int q = 0; q < Gc; ++q) {
            if (q > 0) update(C, K0, q, K0 + q, 1);
            aux_inner_on(h, K0 + q, h->sig[K0 + q], C);
        }
        HIPCHK(hipEventRecord(h->ev_pan[g], C));   // group g's right-hand-side block columns are final
        if (g + 1 < ng) {
            if (g >= 1) HIPCHK(hipStreamWaitEvent(C, h->ev_col[g - 1], 0));   // B1(g - 1) wrote the same block columns
            update(C, K0, Gc, first(g + 1), count(g + 1));                    // A(g)
        }
        HIPCHK(hipStreamWaitEvent(T, h->ev_pan[g], 0));
        if (g + 2 < ng) {
            if (b2m && g >= 1) HIPCHK(hipStreamWaitEvent(T, h->ev_b2[g - 1], 0));   // B2(g - 1) wrote these block columns last
            update(T, K0, Gc, first(g + 2), count(g + 2));                    // B1(g)
            HIPCHK(hipEventRecord(h->ev_col[g], T));
        }
        if (g + 3 < ng) {
            if (b2m) HIPCHK(hipStreamWaitEvent(M, h->ev_pan[g], 0));
            update(b2m ? M : T, K0, Gc, first(g + 3), h->nK - first(g + 3));   // B2(g): beside B1(g) on the handle's own stream
        }
        if (b2m) HIPCHK(hipEventRecord(h->ev_b2[g], M));
    }
    HIPCHK(hipEventRecord(h->ev1, C));
    HIPCHK(hipEventRecord(h->ev_col[(size_t)h->nK], T));   // (ensure_events: nK + 1 entries, the groups use at most nK)
    HIPCHK(hipStreamWaitEvent(M, h->ev1, 0));
    HIPCHK(hipStreamWaitEvent(M, h->ev_col[(size_t)h->nK], 0));
    HIPCHK(hipGetLastError());
    return 0;
}

// forward sweep of the right-hand-side rows through the factor (the L panels are final)
static int solve_sweep(ck_handle* h) {
    const bool la = h->solve_la >= 0 ? h->solve_la != 0 : h->nK >= 40;
    if (la && h->world == 1 && h->loo_g0 < 0 && (h->panel_fused & 2) && eff_group(h) > 1 && h->side) return solve_sweep_la(h);
    if (h->lookahead > 0) {
        if (ensure_events(h)) return -1;
        hipStream_t M = h->stream, S = h->side;
        aux_inner_on(h, 0, h->sig[0], M);
        for (int K = 0; K < h->nK; ++K) {
            if (K > 0) HIPCHK(hipStreamWaitEvent(M, h->ev_pan[K], 0));
            if (K + 1 < h->nK) {
                aux_update_on(h, K, h->sig[K], K + 1, K + 1, M, false);
                HIPCHK(hipEventRecord(h->ev_col[K + 1], M));
                HIPCHK(hipStreamWaitEvent(S, h->ev_col[K + 1], 0));
                aux_inner_on(h, K + 1, h->sig[K + 1], S);
                HIPCHK(hipEventRecord(h->ev_pan[K + 1], S));
                aux_update_on(h, K, h->sig[K], K + 2, h->nK - 1, M, true);
            }
        }
    } else if (eff_group(h) <= 1) {
        for (int K = 0; K < h->nK; ++K)
            if (ck_panel_apply(h, K, CK_APPLY_AUX)) return -1;
    } else {
        const std::vector<int> gs = group_plan(h, eff_group(h));
        for (size_t gi = 0; gi + 1 < gs.size(); ++gi) {
            const int K0 = gs[gi], Gc = gs[gi + 1] - gs[gi];
            for (int g = 0; g < Gc; ++g) {
                if (g > 0) {
                    gemm_timed_begin(h);
                    ck_launch_aux_group(h->stream, h->aux, h->mpad, h->d_panelptr, K0, g, K0 + g, 1, aux_rows(h, K0 + g - 1), h->nend, aux_live(h));
                    gemm_timed_end(h);
                }
                aux_inner_on(h, K0 + g, h->sig[K0 + g], h->stream);
            }
            if (K0 + Gc < h->nK) {
                gemm_timed_begin(h);
                ck_launch_aux_group(h->stream, h->aux, h->mpad, h->d_panelptr, K0, Gc, K0 + Gc, h->nK - K0 - Gc,
                                    aux_rows(h, K0 + Gc - 1), h->nend, aux_live(h));
                gemm_timed_end(h);
            }
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// Both sweeps AT ONCE (ck_factor_predict): the factorisation on the high-priority side stream, the forward substitution of
// the right-hand-side rows on the main stream one panel group behind it.  Each sweep alone leaves the chip idle in places
// the other can fill: the panel chain of the factorisation (one launch, ~0.3 ms per panel, a chain of eight 28 us links
// that occupies a handful of CUs), the under-filled in-group launches of the substitution (280 tiles for 512 slots), and
// the drain of every launch (half a tile lifetime on average: ~0.2 ms x 78 launches per sweep).  The sweeps share nothing
// but the L panels, which the substitution reads a group behind their completion (event per group).  Kernels of two
// streams do interleave on this chip while a large grid is being dispatched (ck_debug_stream_overlap: dependent
// side-stream kernels ran, at ~3x their solo latency, under a 32 ms update that they slowed by 1.5 %).
static int fused_sweeps(ck_handle* h) {
    if (ensure_events(h)) return -1;
    if (h->fused_prio == 2 && !h->side_lo) HIPCHK(hipStreamCreateWithFlags(&h->side_lo, hipStreamNonBlocking));
    // option "fused_prio": 0 the factorisation on the high-priority stream | 1 the substitution | 2 neither
    hipStream_t F = h->fused_prio == 0 ? h->side : h->fused_prio == 1 ? h->stream : h->side_lo;
    hipStream_t M = h->fused_prio == 1 ? h->side : h->stream;
    const std::vector<int> gs = group_plan(h, h->fused_group > 0 ? h->fused_group : eff_group(h));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    if (F != h->stream) HIPCHK(hipStreamWaitEvent(F, h->ev0, 0));
    if (M != h->stream) HIPCHK(hipStreamWaitEvent(M, h->ev0, 0));
    for (int ge = 0; ge + 1 < (int)gs.size(); ++ge) {
        const int K0 = gs[ge], Gc = gs[ge + 1] - gs[ge];
        for (int g = 0; g < Gc; ++g) {
            if (g > 0) syrk_update(h, F, K0, g, K0 + g, 1, 1);
            panel_factor_on(h, K0 + g, F);
        }
        HIPCHK(hipEventRecord(h->ev_pan[ge], F));   // the group's panels are final
        if (K0 + Gc < h->nK) syrk_update(h, F, K0, Gc, K0 + Gc, 1, h->nK - K0 - Gc);
        HIPCHK(hipStreamWaitEvent(M, h->ev_pan[ge], 0));
        for (int g = 0; g < Gc; ++g) {
            if (g > 0) ck_launch_aux_group(M, h->aux, h->mpad, h->d_panelptr, K0, g, K0 + g, 1, aux_rows(h, K0 + g - 1), h->nend, aux_live(h));
            aux_inner_on(h, K0 + g, h->sig[K0 + g], M);
        }
        if (K0 + Gc < h->nK)
            ck_launch_aux_group(M, h->aux, h->mpad, h->d_panelptr, K0, Gc, K0 + Gc, h->nK - K0 - Gc, aux_rows(h, K0 + Gc - 1), h->nend, aux_live(h));
    }
    HIPCHK(hipEventRecord(h->ev1, F));               // end of the factorisation
    HIPCHK(hipEventRecord(h->ev2, M));               // end of the substitution
    if (F != h->stream) HIPCHK(hipStreamWaitEvent(h->stream, h->ev1, 0));
    if (M != h->stream) HIPCHK(hipStreamWaitEvent(h->stream, h->ev2, 0));
    HIPCHK(hipEventRecord(h->ev3, h->stream));       // end of both
    HIPCHK(hipGetLastError());
    return 0;
}

// The same with a LOOK-AHEAD inside the factorisation (option "fused_la"): three streams.  C (high priority) carries the
// critical path -- per panel group g its chain (panel steps and in-group updates) and then A(g), the update of the NEXT
// group's block columns by group g, which is all the next chain waits for; T carries the bulk of the trailing updates,
// B1(g) = group g -> block columns of group g + 2 and B2(g) = group g -> everything beyond; the main stream carries the
// substitution as before.  Every block column still receives its updates in the order of the sequential sweep (group 0,
// 1, 2, ... each in one launch with K = 512 G), so the factor has the same bits; what changes is that the latency-bound
// chain of group g + 1 runs UNDER the bulk of group g instead of in front of it.
//   C:  chain(g) -> [ev_chain g] -> wait B1(g - 1) -> A(g) -> chain(g + 1) ...
//   T:  wait ev_chain g -> B1(g) -> [ev_B1 g] -> B2(g) -> wait ev_chain g + 1 ...
//   M:  wait ev_chain g -> substitution of group g
// (an event is always recorded, in host order, before the wait on it is enqueued: a wait on a never-recorded event is a no-op)
static int fused_sweeps_la(ck_handle* h) {
    if (ensure_events(h)) return -1;
    if (!h->side_lo) HIPCHK(hipStreamCreateWithFlags(&h->side_lo, hipStreamNonBlocking));
    hipStream_t C = h->side, T = h->side_lo, M = h->stream;
    const std::vector<int> gs = group_plan(h, h->fused_group > 0 ? h->fused_group : eff_group(h));
    const int ng = (int)gs.size() - 1;
    auto first = [&](int g) { return gs[(size_t)g]; };
    auto count = [&](int g) { return gs[(size_t)g + 1] - gs[(size_t)g]; };
    HIPCHK(hipEventRecord(h->ev0, M));
    HIPCHK(hipStreamWaitEvent(C, h->ev0, 0));
    HIPCHK(hipStreamWaitEvent(T, h->ev0, 0));
    for (int g = 0; g < ng; ++g) {
        const int K0 = first(g), Gc = count(g);
        for (int q = 0; q < Gc; ++q) {
            if (q > 0) syrk_update(h, C, K0, q, K0 + q, 1, 1);
            panel_factor_on(h, K0 + q, C);
        }
        HIPCHK(hipEventRecord(h->ev_pan[g], C));   // group g's panels are final
        if (g + 1 < ng) {
            if (g >= 1) HIPCHK(hipStreamWaitEvent(C, h->ev_col[g - 1], 0));   // B1(g - 1) wrote the same block columns
            syrk_update(h, C, K0, Gc, first(g + 1), 1, count(g + 1));         // A(g)
        }
        HIPCHK(hipStreamWaitEvent(T, h->ev_pan[g], 0));
        if (g + 2 < ng) {
            syrk_update(h, T, K0, Gc, first(g + 2), 1, count(g + 2));         // B1(g)
            HIPCHK(hipEventRecord(h->ev_col[g], T));
        }
        if (g + 3 < ng) syrk_update(h, T, K0, Gc, first(g + 3), 1, h->nK - first(g + 3));   // B2(g)
        HIPCHK(hipStreamWaitEvent(M, h->ev_pan[g], 0));
        for (int q = 0; q < Gc; ++q) {
            if (q > 0) ck_launch_aux_group(M, h->aux, h->mpad, h->d_panelptr, K0, q, K0 + q, 1, aux_rows(h, K0 + q - 1), h->nend, aux_live(h));
            aux_inner_on(h, K0 + q, h->sig[K0 + q], M);
        }
        if (K0 + Gc < h->nK)
            ck_launch_aux_group(M, h->aux, h->mpad, h->d_panelptr, K0, Gc, K0 + Gc, h->nK - K0 - Gc, aux_rows(h, K0 + Gc - 1), h->nend, aux_live(h));
    }
    HIPCHK(hipEventRecord(h->ev1, C));               // end of the factorisation's chain: the last panel is final
    HIPCHK(hipEventRecord(h->ev2, T));
    HIPCHK(hipStreamWaitEvent(M, h->ev1, 0));
    HIPCHK(hipStreamWaitEvent(M, h->ev2, 0));
    HIPCHK(hipEventRecord(h->ev3, M));               // end of everything
    HIPCHK(hipGetLastError());
    return 0;
}

// ONE sweep over the tall matrix [Sigma; c0^T; z^T] (round 4, option "tall_sweep"): the forward substitution is the panel
// step applied to more rows, so every launch of the factorisation takes the right-hand-side rows along --
// the cooperative panel step as further workgroups (k_panel_coop), the updates as further tiles of the same grid
// (k_tall_group_d) -- instead of a second sweep that shares the chip with the first (fused_sweeps_la: 156 big launches per
// pass on two streams, the substitution's one-column launches filling 280 of 512 slots).  The look-ahead is kept: stream C
// (high priority) carries the chain of group g and A(g), the update of the NEXT group's block columns by group g; stream T
// carries B1(g) = group g -> block columns of group g + 2 and B2(g) = group g -> everything beyond.  Every block column --
// of Sigma and of the right-hand-side rows -- receives its updates in the order of the sequential sweeps, each in one launch
// with K = 512 G, and every tile computes what it computed there: same bits as ck_factor + ck_predict.
//   C:  chain(g) -> [ev_pan g] -> wait B1(g - 1) -> A(g) -> chain(g + 1) ...
//   T:  wait ev_pan g -> B1(g) -> [ev_col g] -> B2(g) -> wait ev_pan g + 1 ...
static int tall_sweeps(ck_handle* h) {
    if (ensure_events(h)) return -1;
    if (!h->side_lo) HIPCHK(hipStreamCreateWithFlags(&h->side_lo, hipStreamNonBlocking));
    hipStream_t C = h->side, T = h->side_lo, M = h->stream;
    const std::vector<int> gs = group_plan(h, h->fused_group > 0 ? h->fused_group : eff_group(h));
    const int ng = (int)gs.size() - 1;
    auto first = [&](int g) { return gs[(size_t)g]; };
    auto count = [&](int g) { return gs[(size_t)g + 1] - gs[(size_t)g]; };
    auto update = [&](hipStream_t st, int K0, int np, int J0, int nJ) {
        if (nJ <= 0) return;
        gemm_timed_begin(h, st);
        ck_launch_tall_group(st, h->d_sigptr, h->aux, h->mpad, K0, np, J0, nJ, h->nend, h->tall_thin ? h->m + 1 : 0);
        gemm_timed_end(h, st);
    };
    // option "tall_b2_stream": B2(g) on the handle's own stream instead of behind B1(g) on T -- the two only need group g's panels,
    // so the partly filled last round of B1(g) runs beside B2(g)'s tiles instead of in front of them
    const bool b2m = h->tall_b2_stream != 0;
    HIPCHK(hipEventRecord(h->ev0, M));
    HIPCHK(hipStreamWaitEvent(C, h->ev0, 0));
    HIPCHK(hipStreamWaitEvent(T, h->ev0, 0));
    for (int g = 0; g < ng; ++g) {
        const int K0 = first(g), Gc = count(g);
        for (int q = 0; q < Gc; ++q) {
            if (q > 0) update(C, K0, q, K0 + q, 1);
            const int K = K0 + q;
            const int64_t R = h->Npad - (int64_t)K * CK_NB;
            // panel steps that run under a bulk update and have many chunks: the head cooperatively, the rows on their own
            const bool split = h->tall_split == 1 || (h->tall_split == 2 && g >= 1 && R >= h->tall_split_rows);
            panel_factor_on(h, K, C, true, split);
        }
        HIPCHK(hipEventRecord(h->ev_pan[g], C));   // group g's panels and right-hand-side block columns are final
        if (g + 1 < ng) {
            if (g >= 1) HIPCHK(hipStreamWaitEvent(C, h->ev_col[g - 1], 0));   // B1(g - 1) wrote the same block columns
            update(C, K0, Gc, first(g + 1), count(g + 1));                    // A(g)
        }
        HIPCHK(hipStreamWaitEvent(T, h->ev_pan[g], 0));
        if (g + 2 < ng) {
            // (B2 on its own stream: B2(g - 1) was the last writer of the block columns B1(g) updates)
            if (b2m && g >= 1 && g + 2 < ng) HIPCHK(hipStreamWaitEvent(T, h->ev_b2[g - 1], 0));
            update(T, K0, Gc, first(g + 2), count(g + 2));                    // B1(g)
            HIPCHK(hipEventRecord(h->ev_col[g], T));
        }
        if (g + 3 < ng) {
            hipStream_t B = b2m ? M : T;
            if (b2m) HIPCHK(hipStreamWaitEvent(M, h->ev_pan[g], 0));
            update(B, K0, Gc, first(g + 3), h->nK - first(g + 3));            // B2(g)
        }
        if (b2m) HIPCHK(hipEventRecord(h->ev_b2[g], M));
    }
    HIPCHK(hipEventRecord(h->ev1, C));               // end of the chain: the last panel is final
    HIPCHK(hipEventRecord(h->ev2, T));
    HIPCHK(hipStreamWaitEvent(M, h->ev1, 0));
    HIPCHK(hipStreamWaitEvent(M, h->ev2, 0));
    HIPCHK(hipEventRecord(h->ev3, M));               // end of everything
    HIPCHK(hipGetLastError());
    return 0;
}

// ck_assemble_joint must have been called; = ck_factor + ck_predict with the two sweeps overlapped.  *info != 0: Sigma is not
// positive definite (pred / pred_err untouched), reported exactly as ck_factor reports it.
extern "C" int ck_factor_predict(ck_handle* h, int i, const double* pcoords, int64_t m, double* pred, double* pred_err,
                                 int64_t* info) {
    CHKH(h);
    if (h->world != 1) return fail("ck_factor_predict is the single-process form");
    if (!h->assembled) return fail("ck_assemble_joint has not been called");
    if (h->factored) return fail("Sigma is already factored; call ck_predict, or ck_assemble_joint again");
    if (!info) return fail("null info");
    // Overlapping pays while a sweep leaves the chip EMPTY for a noticeable share of its time (panel chain, launch drains):
    // 15-20 % at 20 panels, 2-4 % at 79, and at 196 panels (N = 100 000) between +1.5 % and -2 % depending on the device --
    // the launches are long against their drains there, and two kernels sharing the chip cost each other efficiency.
    // Automatic rule (option "fused_sweeps" = -1): overlapped up to 128 panels, one sweep after the other beyond.
    const bool overlap = h->fused_sweeps_opt >= 0 ? h->fused_sweeps_opt != 0 : h->nK <= 128;
    if (!(h->panel_fused & 2) || h->loo_g0 >= 0 || !overlap) {   // (also: options of the A/B scripts) the plain sequence
        h->t_ms[13] = 0.0;   // no overlapped span: ck_timings [1] and [3] are the two sweeps' own times
        if (ck_factor(h, info)) return -1;
        return *info == 0 ? ck_predict(h, i, pcoords, m, pred, pred_err) : 0;
    }
    if (ck_aux_begin(h, i, pcoords, m)) return -1;
    h->gemm_ev_used = 0;
    const bool la = h->fused_la >= 0 ? h->fused_la != 0 : h->nK >= 40;
    const bool tall = (h->panel_fused & 16) && h->tall_sweep != 0;
    if (tall ? tall_sweeps(h) : la ? fused_sweeps_la(h) : fused_sweeps(h)) return -1;
    struct { unsigned werr, pad; long long info; } status = {0u, 0u, 0};   // d_coop[16 .. 19]: the error word and the info word
    HIPCHK(hipMemcpyAsync(&status, h->d_coop + 16, sizeof(status), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *info = external_index(h, (int64_t)status.info);
    const unsigned werr = status.werr;
    if (*info != 0 || werr != 0) {
        // not positive definite, or a cooperative panel step timed out: ck_factor's own handling (redo in the caller's
        // order for numpy's minor index / without the cooperative step), then the substitution on the finished factor.
        // A timed-out step: clear its error word and switch the cooperative step off HERE, so that ck_factor runs the
        // factorisation once, the plain way, instead of repeating a cooperative one and discarding it for the stale flag.
        if (werr != 0) {
            HIPCHK(hipMemset(h->d_coop + 16, 0, sizeof(unsigned)));
            h->panel_fused &= ~16;
        }
        h->gemm_ev_used = 0;
        h->assembled = false;
        h->aux_state = 0;
        if (ck_assemble_joint(h)) return -1;
        if (ck_factor(h, info)) return -1;
        if (werr != 0) h->t_ms[12] = 1.0;   // visible in ck_timings: the factorisation was redone
        if (*info != 0) return 0;
        const double redone = h->t_ms[12];
        if (ck_predict(h, i, pcoords, m, pred, pred_err)) return -1;
        h->t_ms[12] = redone;
        return 0;
    }
    h->factored = true;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->t_ms[1] = ms;    // the factorisation's span (it shares the chip with the substitution)
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev3));
    h->t_ms[13] = ms;   // both sweeps
    h->t_ms[3] = h->t_ms[13] - h->t_ms[1];   // what the substitution adds behind the factorisation
    h->t_ms[12] = 0.0;
    h->t_ms[5] = h->t_ms[6] = h->t_ms[7] = h->t_ms[8] = 0.0;
    h->t_ms[15] = 0.0;
    if (tall) gemm_timed_collect(h, 5, h->ev0);   // the tall sweep's update launches: sum of their durations and union of their intervals
    else h->gemm_ev_used = 0;
    if (ck_aux_finish(h, pred, pred_err)) return -1;
    h->aux_state = 2;
    return 0;
}

extern "C" int ck_factor(ck_handle* h, int64_t* info) {
    CHKH(h);
    if (h->world != 1) return fail("ck_factor is the single-process form; drive ck_panel_* for world > 1");
    if (!h->assembled) return fail("ck_assemble_joint has not been called");
    if (h->factored) return fail("Sigma is already factored; call ck_assemble_joint again");
    h->aux_state = 0;
    h->t_ms[12] = 0.0;
    if (factor_sweep(h)) return -1;
    {
        // FIRST the cooperative panel step's error word: a workgroup of k_panel_coop that gave up waiting for a pivot block
        // (bounded spin, option "coop_spins": ~2 s; never observed outside the test hook "coop_inject_panel") leaves a factor --
        // and an info word -- that are not to be trusted.  The cooperative step is switched off for this handle and the
        // factorisation repeated with one launch per dependency.
        HIPCHK(hipStreamSynchronize(h->stream));
        unsigned werr = 0;
        HIPCHK(hipMemcpy(&werr, h->d_coop + 16, sizeof(werr), hipMemcpyDeviceToHost));
        if (werr != 0) {
            HIPCHK(hipMemset(h->d_coop + 16, 0, sizeof(unsigned)));
            h->panel_fused &= ~16;
            h->assembled = false;
            if (ck_assemble_joint(h)) return -1;
            if (factor_sweep(h)) return -1;
            h->t_ms[12] = 1.0;                  // visible in ck_timings
        }
    }
    if (factor_info_raw(h, info)) return -1;
    if (*info != 0 && h->site_order) {
        // Not positive definite.  numpy names the failing leading minor of Sigma in the CALLER's
        // order (cho_factor, joint_prediction.py:68-69, raises LinAlgError): redo the factorisation in that
        // order so that the reported index is the reference's.  The handle stays in site_order 0.
        h->site_order = 0;
        h->layout_ready = false;
        h->assembled = false;
        if (ck_assemble_joint(h)) return -1;
        if (factor_sweep(h)) return -1;
        if (factor_info_raw(h, info)) return -1;
    }
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->t_ms[1] = ms;
    gemm_timed_collect(h, 5);
    h->factored = true;
    return 0;
}

// pcoords == nullptr: storage only (ck_loocv fills the right-hand-side rows itself)
static int aux_begin_impl(ck_handle* h, int i, const double* pcoords, int64_t m, bool may_sort) {
    if (ensure_layout(h)) return -1;
    if (i < 0 || i >= h->n_procs) return fail("process index out of range");
    if (m < 0) return fail("bad pcoords");
    const int64_t mpad = roundup(m + 1, CK_AUX_ALIGN);
    const int64_t need = mpad * h->Npad;
    if (need > h->aux_cap) {
        dev_free_one(h, h->aux);
        h->aux = nullptr;
        if (dev_alloc(h, (void**)&h->aux, need * 8)) return -1;
        h->aux_cap = need;
    }
    if (mpad > h->p_cap) {
        dev_free_one(h, h->p0);
        dev_free_one(h, h->d_pcoords);
        dev_free_one(h, h->d_pred);
        dev_free_one(h, h->pu);
        if (dev_alloc(h, (void**)&h->p0, 3 * mpad * 8)) return -1;
        if (dev_alloc(h, (void**)&h->pu, 3 * mpad * 8)) return -1;
        if (dev_alloc(h, (void**)&h->d_pcoords, 2 * mpad * 8)) return -1;
        if (dev_alloc(h, (void**)&h->d_pred, 2 * mpad * 8)) return -1;
        h->p_cap = mpad;
    }
    h->p1 = h->p0 + mpad;
    h->p2 = h->p0 + 2 * mpad;
    h->d_err = h->d_pred + mpad;
    h->i_pred = i;
    h->m = m;
    h->mpad = mpad;
    h->aux_state = pcoords ? 1 : 0;
    // large sets of prediction points are laid out along the Hilbert curve like the data sites
    // (site_order above); ck_aux_finish hands the results back in the caller's order
    std::vector<double> sorted;
    if (!pcoords) {
        h->p_sorted = false;
        return 0;
    }
    bool fast_done = false;
    h->p_sorted = may_sort && h->site_order && m >= 256;
    if (h->p_sorted) {
        double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
        ck_host_bounding_box(pcoords, m, lo, hi);
        ck_host_hilbert_order(pcoords, m, lo, hi, h->pperm);
        sorted.resize((size_t)(2 * m));
        for (int64_t j = 0; j < m; ++j) {
            sorted[2 * j] = pcoords[2 * h->pperm[(size_t)j]];
            sorted[2 * j + 1] = pcoords[2 * h->pperm[(size_t)j] + 1];
        }
        pcoords = sorted.data();   // alive until the event synchronisation at the end of this function
    }
    HIPCHK(hipMemsetAsync(h->d_pcoords, 0, 2 * mpad * 8, h->stream));
    if (m > 0) HIPCHK(hipMemcpyAsync(h->d_pcoords, pcoords, 2 * m * 8, hipMemcpyHostToDevice, h->stream));
    // timed like ck_assemble_joint: the device work of K2 (site transform, table kernel, exact pass) -- not the upload of the
    // prediction coordinates in front of it nor the host round trip for the worklist count behind it
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    // table path: ONE launch -- the assembly transforms the prediction sites itself (every strip its 64 rows; the strips of
    // block column 0 store p0 / pu for the exact pass and the later users) and starts on a counter the previous exact pass
    // zeroed.  Round 3 had k_prep_sites, a memset, the assembly and the exact pass: the gaps between the four were a tenth of
    // the stage.  The exact path (no tables) keeps the separate transform.
    const bool fold = tables_usable(h);
    if (!fold) ck_launch_prep_sites(h->stream, h->d_pcoords, mpad, h->metric, h->p0, h->p1, h->p2, h->pu);
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool fast = tables_usable(h) && attempt == 0;
        if (fast) next_worklist(h);
        ck_launch_assemble_aux(h->stream, fast, h->d_blk, h->d_tabs, h->d_coefptr, h->metric, i, h->p0, h->pu, m,
                               mpad, h->s0, h->su, h->z, layout_of(h), h->nK, h->aux, h->wl, fast && fold ? h->d_pcoords : nullptr,
                               fast ? h->assemble_queue : 0);
        if (!fast) break;
        ck_launch_assemble_fix(h->stream, true, h->d_blk, h->metric, i, h->p0, mpad, h->s0, layout_of(h), h->wl,
                               h->d_sigptr, h->aux);
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        unsigned cnt = 0;
        HIPCHK(hipMemcpyAsync(&cnt, h->wl.count, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->fallback_total += cnt;
        if (cnt <= h->wl.cap) {
            fast_done = true;
            break;
        }
    }
    HIPCHK(hipGetLastError());
    if (!fast_done) HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));   // pcoords is caller memory: do not return before the copy is done
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->t_ms[2] = ms;
    return 0;
}

extern "C" int ck_aux_begin(ck_handle* h, int i, const double* pcoords, int64_t m) {
    CHKH(h);
    if (m > 0 && !pcoords) return fail("bad pcoords");
    return aux_begin_impl(h, i, pcoords, m, true);
}

extern "C" int ck_aux_finish(ck_handle* h, double* pred, double* pred_err) {
    CHKH(h);
    if (h->mpad <= 0) return fail("ck_aux_begin has not been called");
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    const double c0 = h->blk[2 * h->i_pred].amp + h->blk[2 * h->i_pred].nugget;   // sigma_i^2 + nugget_i (model.py:194-196 at h = 0)
    ck_launch_reduce_pred(h->stream, h->aux, h->mpad, h->nK, h->m, h->m, c0, h->d_pred, h->d_err);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    if (h->m > 0 && !h->p_sorted) {
        HIPCHK(hipMemcpyAsync(pred, h->d_pred, h->m * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(pred_err, h->d_err, h->m * 8, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->time_gemm == 2) gemm_timed_collect(h, 5);   // step-wise form: the Sigma updates of this sweep
    if (h->m > 0 && h->p_sorted) {   // back to the caller's order
        std::vector<double> tp((size_t)h->m), te((size_t)h->m);
        HIPCHK(hipMemcpyAsync(tp.data(), h->d_pred, h->m * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(te.data(), h->d_err, h->m * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int64_t j = 0; j < h->m; ++j) {
            pred[h->pperm[(size_t)j]] = tp[(size_t)j];
            pred_err[h->pperm[(size_t)j]] = te[(size_t)j];
        }
    }
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->t_ms[4] = ms;
    return 0;
}

extern "C" int ck_predict(ck_handle* h, int i, const double* pcoords, int64_t m, double* pred, double* pred_err) {
    CHKH(h);
    if (h->world != 1) return fail("ck_predict is the single-process form");
    if (!h->factored) return fail("ck_factor has not been called");
    if (ck_aux_begin(h, i, pcoords, m)) return -1;
    h->gemm_ev_used = 0;
    HIPCHK(hipEventRecord(h->ev2, h->stream));   // the handle's second event pair: ev0 / ev1 are used inside
    if (solve_sweep(h)) return -1;
    HIPCHK(hipEventRecord(h->ev3, h->stream));
    if (ck_aux_finish(h, pred, pred_err)) return -1;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev2, h->ev3));
    h->t_ms[3] = ms;
    gemm_timed_collect(h, 7);
    h->aux_state = 2;
    return 0;
}

// ---------------------------------------------------------------------------------------
// _verify_model (src/joint_prediction.py:60-66, 260-274)
// ---------------------------------------------------------------------------------------
// The reference factorises the stacked (m + N) x (m + N) matrix [[C_pp, c0^T], [c0, Sigma]] and warns when
// that fails.  Sigma is positive definite here (ck_factor succeeded), so the stacked matrix is positive definite
// iff the Schur complement  S = C_pp - c0^T Sigma^-1 c0 = C_pp - V^T V  is, and V^T = (L^-1 c0)^T are exactly the
// solved right-hand-side rows the last ck_predict left on the device: assemble C_pp (the auto-covariance of
// process i at the prediction sites, nugget where h == 0, :94-102) in the packed panel format of Sigma, subtract
// V^T V in one pass over the rows (ck_la.hip: k_schur_syrk_d), and run the same blocked Cholesky on it --
// 2 N m^2 / 2 + m^3 / 3 flop instead of (m + N)^3 / 3.
static void schur_free(ck_handle* h) {
    for (double* p : h->sch_sig)
        if (p) (void)hipFree(p);
    h->sch_sig.clear();
    void* ps[] = {h->d_sch_ptr, h->d_sch_tile0, h->d_sch_panel_of, h->sch_pc, h->sch_c, h->sch_u,
                  h->d_sch_blk, h->d_sch_tabs, h->d_sch_coefptr, h->d_sch_info};
    for (void* p : ps)
        if (p) (void)hipFree(p);
    h->d_sch_ptr = nullptr;
    h->d_sch_tile0 = h->d_sch_panel_of = nullptr;
    h->sch_pc = h->sch_c = h->sch_u = nullptr;
    h->d_sch_blk = nullptr;
    h->d_sch_tabs = nullptr;
    h->d_sch_coefptr = nullptr;
    h->d_sch_info = nullptr;
    h->sch_M = 0;
}

// the factorisation drivers work on the handle's matrix: point them at the Schur complement for one sweep
struct SchurSwap {
    ck_handle* h;
    std::vector<double*> sig;
    double **d_sigptr, **d_panelptr;
    int nK, time_gemm, world, rank;
    int64_t Npad, nend;
    long long* d_info;
    int lookahead;
    bool assembled;
    explicit SchurSwap(ck_handle* hh, int nJ, int64_t Mp, int64_t m_valid) : h(hh) {
        sig = h->sig;
        d_sigptr = h->d_sigptr;
        d_panelptr = h->d_panelptr;
        nK = h->nK;
        Npad = h->Npad;
        nend = h->nend;
        d_info = h->d_info;
        time_gemm = h->time_gemm;
        lookahead = h->lookahead;
        assembled = h->assembled;
        world = h->world;
        rank = h->rank;
        h->sig = h->sch_sig;
        h->d_sigptr = h->d_panelptr = h->d_sch_ptr;
        h->nK = nJ;
        h->Npad = Mp;
        h->nend = m_valid;
        h->d_info = h->d_sch_info;
        h->time_gemm = 0;
        h->lookahead = 0;
        h->assembled = true;
        h->world = 1;
        h->rank = 0;
    }
    ~SchurSwap() {
        h->sig = sig;
        h->d_sigptr = d_sigptr;
        h->d_panelptr = d_panelptr;
        h->nK = nK;
        h->Npad = Npad;
        h->nend = nend;
        h->d_info = d_info;
        h->time_gemm = time_gemm;
        h->lookahead = lookahead;
        h->assembled = assembled;
        h->world = world;
        h->rank = rank;
    }
};

static int factor_sweep(ck_handle* h);

extern "C" int ck_verify_model(ck_handle* h, int64_t* info) {
    CHKH(h);
    if (!info) return fail("null info");
    if (h->world != 1) return fail("ck_verify_model is the single-process form");
    if (h->aux_state != 2) return fail("ck_verify_model needs the solved right-hand sides of a preceding ck_predict");
    const int64_t m = h->m, mpad = h->mpad;
    *info = 0;
    if (m <= 0) return 0;
    const int64_t Mp = roundup(m, CK_NB);
    const int nJ = (int)(Mp / CK_NB);
    if (h->sch_M != Mp) {
        schur_free(h);
        h->sch_sig.assign((size_t)nJ, nullptr);
        std::vector<int> tile0, panel_of;
        int acc = 0;
        for (int J = 0; J < nJ; ++J) {
            const int64_t rows = Mp - (int64_t)J * CK_NB;
            HIPCHK(hipMalloc((void**)&h->sch_sig[(size_t)J], (size_t)(rows * CK_NB + CK_PANEL_TAIL) * 8));
            tile0.push_back(acc);
            panel_of.push_back(J);
            acc += (int)(rows / 64);
        }
        tile0.push_back(acc);
        h->sch_tiles = acc;
        HIPCHK(hipMalloc((void**)&h->d_sch_ptr, (size_t)nJ * sizeof(double*)));
        HIPCHK(hipMemcpy(h->d_sch_ptr, h->sch_sig.data(), (size_t)nJ * sizeof(double*), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void**)&h->d_sch_tile0, tile0.size() * sizeof(int)));
        HIPCHK(hipMalloc((void**)&h->d_sch_panel_of, (panel_of.size() + 1) * sizeof(int)));
        HIPCHK(hipMemcpy(h->d_sch_tile0, tile0.data(), tile0.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_sch_panel_of, panel_of.data(), panel_of.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void**)&h->sch_pc, (size_t)(2 * Mp) * 8));
        HIPCHK(hipMalloc((void**)&h->sch_c, (size_t)(3 * Mp) * 8));
        HIPCHK(hipMalloc((void**)&h->sch_u, (size_t)(3 * Mp) * 8));
        HIPCHK(hipMalloc((void**)&h->d_sch_blk, 3 * sizeof(CkMatern)));
        HIPCHK(hipMalloc((void**)&h->d_sch_tabs, 3 * sizeof(CkTable)));
        HIPCHK(hipMalloc((void**)&h->d_sch_coefptr, 3 * sizeof(double*)));
        HIPCHK(hipMalloc((void**)&h->d_sch_info, sizeof(long long)));
        h->sch_M = Mp;
    }
    const auto t_begin = std::chrono::steady_clock::now();
    // the prediction sites as a one-process site set: every tile uses the auto-block (i, i), replicated into all
    // three slots of the block / table arrays
    const int bi = 2 * h->i_pred;
    CkMatern hb[3] = {h->blk[bi], h->blk[bi], h->blk[bi]};
    CkTable ht[3] = {h->tab[bi], h->tab[bi], h->tab[bi]};
    double* hc[3] = {h->d_coef[bi], h->d_coef[bi], h->d_coef[bi]};
    HIPCHK(hipMemcpyAsync(h->d_sch_blk, hb, sizeof(hb), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_sch_tabs, ht, sizeof(ht), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_sch_coefptr, hc, sizeof(hc), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->sch_pc, 0, (size_t)(2 * Mp) * 8, h->stream));
    HIPCHK(hipMemcpyAsync(h->sch_pc, h->d_pcoords, (size_t)(2 * m) * 8, hipMemcpyDeviceToDevice, h->stream));
    ck_launch_prep_sites(h->stream, h->sch_pc, Mp, h->metric, h->sch_c, h->sch_c + Mp, h->sch_c + 2 * Mp, h->sch_u);
    HIPCHK(hipStreamSynchronize(h->stream));   // hb / ht / hc are stack memory
    const CkLayout L{m, Mp, Mp, Mp};
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool fast = tables_usable(h) && attempt == 0;
        if (fast) next_worklist(h);
        CkPanelMap pm{h->d_sch_tile0, h->d_sch_panel_of, h->d_sch_ptr, nJ, nullptr, 0, nullptr};
        ck_launch_assemble_sigma(h->stream, fast, h->d_sch_blk, h->d_sch_tabs, h->d_sch_coefptr, h->metric, h->sch_c,
                                 h->sch_u, L, pm, h->sch_tiles, h->wl, fast ? h->assemble_queue : 0);
        if (!fast) break;
        ck_launch_assemble_fix(h->stream, false, h->d_sch_blk, h->metric, 0, nullptr, 0, h->sch_c, L, h->wl, h->d_sch_ptr,
                               nullptr);
        unsigned cnt = 0;
        HIPCHK(hipMemcpyAsync(&cnt, h->wl.count, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->fallback_total += cnt;
        if (cnt <= h->wl.cap) break;
    }
    HIPCHK(hipGetLastError());
    // row m of the right-hand sides is y = L^-1 z, not a prediction site (ck_aux_finish has consumed it)
    HIPCHK(hipMemset2DAsync(h->aux + m * CK_NB, (size_t)mpad * CK_NB * 8, 0, (size_t)CK_NB * 8, (size_t)h->nK, h->stream));
    ck_launch_schur_syrk(h->stream, h->d_sch_ptr, h->aux, mpad, h->nK, nJ, Mp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(h->d_sch_info, 0, sizeof(long long), h->stream));
    long long v = 0;
    {
        SchurSwap swap(h, nJ, Mp, m);
        if (factor_sweep(h)) return -1;   // records ev1 at its end
        HIPCHK(hipMemcpyAsync(&v, h->d_info, sizeof(v), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        unsigned werr = 0;
        HIPCHK(hipMemcpy(&werr, h->d_coop + 16, sizeof(werr), hipMemcpyDeviceToHost));
        if (werr != 0) {
            HIPCHK(hipMemset(h->d_coop + 16, 0, sizeof(unsigned)));
            h->panel_fused &= ~16;
            return fail("cooperative panel step timed out (option panel_fused bit 4 now off): call ck_predict and ck_verify_model again");
        }
    }
    *info = (int64_t)v;   // 1-based index among the prediction sites in the library's internal order, 0 = positive definite
    h->t_ms[11] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return 0;
}

// ---------------------------------------------------------------------------------------
// simulation draw z = L eps  (sim.BivariateRandomField._simulate, src/sim.py:52-54)
// ---------------------------------------------------------------------------------------
extern "C" int ck_sample(ck_handle* h, const double* noise, double* out, int64_t n) {
    CHKH(h);
    if (h->world != 1) return fail("ck_sample is the single-process form");
    if (!h->factored) return fail("ck_factor has not been called");
    if (n != h->N) return fail("n must equal the number of observations");
    if (h->site_order)
        return fail("ck_sample: L eps depends on the order of the sites; set option site_order = 0 before assembling");
    const int64_t Np = h->Npad, n0 = h->n[0], gap = h->n0p - n0;
    std::vector<double> hv(Np, 0.0), ho(Np);
    memcpy(hv.data(), noise, n0 * 8);
    if (n > n0) memcpy(hv.data() + h->n0p, noise + n0, (n - n0) * 8);
    DevTemps tmp;
    double *dv = nullptr, *dout = nullptr;
    HIPCHK(tmp.get(&dv, (size_t)(Np * 8)));
    HIPCHK(tmp.get(&dout, (size_t)(Np * 8)));
    HIPCHK(hipMemcpyAsync(dv, hv.data(), Np * 8, hipMemcpyHostToDevice, h->stream));
    ck_launch_tri_matvec(h->stream, h->d_sigptr, Np, dv, dout);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ho.data(), dout, Np * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(out, ho.data(), n0 * 8);
    if (n > n0) memcpy(out + n0, ho.data() + n0 + gap, (n - n0) * 8);
    return 0;
}

// ---------------------------------------------------------------------------------------
// leave-one-out cross-validation from ONE factorisation
// ---------------------------------------------------------------------------------------
// The reference re-assembles and re-factorises everything once per withheld datum
// (src/joint_prediction.py:207-257).  Withholding datum q of process i and predicting it from all
// other data is the Gaussian conditional of z_q given z_-q under N(0, Sigma) -- c0 of that
// prediction IS the Sigma column of q (nugget only at h == 0, :112-121) and its prior variance IS
// Sigma_qq -- so
//     pred_q = z_q - (Sigma^-1 z)_q / (Sigma^-1)_qq,      pred_err_q = sqrt(1 / (Sigma^-1)_qq).
// With V = L^-1 E_i (unit vectors of the data of process i as right-hand sides) and y = L^-1 z:
// (Sigma^-1 z)_q = V_q . y and (Sigma^-1)_qq = |V_q|^2: the ordinary forward sweep + reduction.
extern "C" int ck_loocv(ck_handle* h, int i, double* pred, double* pred_err) {
    CHKH(h);
    if (h->world != 1) return fail("ck_loocv is the single-process form");
    if (!h->factored) return fail("ck_factor has not been called");
    if (i < 0 || i >= h->n_procs) return fail("process index out of range");
    const int64_t m = h->n[i];
    if (m <= 0) return 0;
    // reuse the aux machinery: storage as for m prediction points; m + 1 rows:
    // row 0 = z, row 1 + q = unit vector of datum q
    if (aux_begin_impl(h, i, nullptr, m, false)) return -1;
    HIPCHK(hipMemsetAsync(h->aux, 0, (size_t)h->mpad * h->Npad * 8, h->stream));
    h->loo_g0 = i == 0 ? 0 : h->n0p;
    ck_launch_loo_rows(h->stream, h->aux, h->mpad, m, h->loo_g0, h->z, h->Npad);
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    const int rc = solve_sweep(h);
    h->loo_g0 = -1;
    if (rc) return -1;
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    ck_launch_reduce_pred(h->stream, h->aux, h->mpad, h->nK, m + 1, 0, -1.0, h->d_pred, h->d_err);
    HIPCHK(hipGetLastError());
    std::vector<double> s1(m), s2(m);
    HIPCHK(hipMemcpyAsync(s1.data(), h->d_pred + 1, m * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(s2.data(), h->d_err + 1, m * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        h->t_ms[3] = ms;
    }
    const double* zi = h->h_values[i].data();
    for (int64_t q = 0; q < m; ++q) {   // q: internal position; results go to the caller's index
        const int64_t x = h->perm[i][(size_t)q];
        pred[x] = zi[x] - s1[q] / s2[q];
        const double e = sqrt(1.0 / s2[q]);
        pred_err[x] = (e == e) ? e : 0.0;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// local-neighbourhood cokriging: src/point_prediction.py:45-249
// ---------------------------------------------------------------------------------------
extern "C" int ck_predict_local(ck_handle* h, int i, const double* pcoords, int64_t m, double max_dist, int cv,
                                double* pred, double* pred_err, int64_t* n_empty, int64_t* n_not_pd,
                                int64_t* k_max) {
    CHKH(h);
    if (ensure_layout(h, false)) return -1;   // sites, tables, chunk bounds -- no Sigma panels
    if (i < 0 || i >= h->n_procs) return fail("process index out of range");
    if (m < 0 || (m > 0 && !pcoords)) return fail("bad pcoords");
    if (n_empty) *n_empty = 0;
    if (n_not_pd) *n_not_pd = 0;
    if (k_max) *k_max = 0;
    if (m == 0) return 0;
    const int64_t mp = roundup(m, 64);
    DevTemps tmp;
    double *d_pc = nullptr, *d_p3 = nullptr, *d_pu = nullptr, *d_out = nullptr, *d_slab = nullptr;
    int* d_cnt = nullptr;
    long long *d_off = nullptr, *d_linfo = nullptr;
    int* d_k0 = nullptr;
    CkLocalSys* d_sys = nullptr;
    HIPCHK(tmp.get(&d_pc, (size_t)(2 * mp * 8)));
    HIPCHK(tmp.get(&d_p3, (size_t)(3 * mp * 8)));
    HIPCHK(tmp.get(&d_pu, (size_t)(3 * mp * 8)));
    HIPCHK(tmp.get(&d_out, (size_t)(2 * mp * 8)));
    HIPCHK(tmp.get(&d_cnt, (size_t)(mp * sizeof(int))));
    HIPCHK(tmp.get(&d_off, (size_t)(mp * sizeof(long long))));
    HIPCHK(hipMemsetAsync(d_pc, 0, 2 * mp * 8, h->stream));
    HIPCHK(hipMemcpyAsync(d_pc, pcoords, 2 * m * 8, hipMemcpyHostToDevice, h->stream));
    ck_launch_prep_sites(h->stream, d_pc, mp, h->metric, d_p3, d_p3 + mp, d_p3 + 2 * mp, d_pu);
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    // largest chord (in the space of su / pu) a neighbour can have, with a margin (ck_local.hip: LpSearch)
    double cmax = max_dist;
    if (h->metric == CK_METRIC_HAVERSINE) {
        const double half = max_dist / (2.0 * CK_EARTH_RADIUS_KM);
        cmax = half >= 1.5 ? 4.0 : 2.0 * sin(half);   // beyond ~ a quarter of the globe: no culling
    }
    cmax = cmax * (1.0 + 1e-9) + 1e-12;
    if (!(cmax == cmax)) cmax = INFINITY;
    ck_launch_local_count(h->stream, h->metric, i, cv ? 1 : 0, max_dist, d_p3, m, mp, h->s0, layout_of(h), d_cnt,
                          h->d_chunkb, cmax, d_pu);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));   // [ev0, ev1]: the counting pass; the host-side planning and a growth of the
                                                 // scratch slab (hipMalloc: up to seconds) lie between the two device windows
    std::vector<int> cnt(m);
    HIPCHK(hipMemcpyAsync(cnt.data(), d_cnt, m * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    // Scratch slabs for neighbourhoods beyond the LDS limit.  Three size classes:
    //   k <= LDS limit               k_local_solve, system in LDS
    //   k <= local_tile_min          k_local_solve_big, one workgroup per point on a slab in global memory
    //   larger                       the tiled path (ck_internal.h: CkLocalSys): a batch of systems factored
    //                                together, three launches per 64 columns, updates on the MFMA tiles
    // Both slab users work in batches that fit a budget (a quarter of the free device memory, at most 32 GiB),
    // so that large radii over many points do not need sum_p k_p^2 doubles at once.
    const int kl = ck_local_lds_limit();
    const int k_hi = h->local_tile_min;   // may lie below the LDS limit: then the LDS kernel only sees k <= k_hi
    std::vector<long long> off(m, 0), need(m, 0);
    std::vector<int64_t> tiled;   // points of the third class
    int64_t kmx = 0, nempty = 0;
    long long need_max = 0;
    for (int64_t p = 0; p < m; ++p) {
        const long long k = cnt[p];
        kmx = k > kmx ? k : kmx;
        if (k == 0) ++nempty;
        long long nd = 0;
        if (k > k_hi) {
            tiled.push_back(p);
            nd = ck_local_tiled_doubles(k);
        } else if (k > kl) {
            nd = need[p] = ((k + 2) * k + (k + 1) / 2 + 2 + 1) & ~1LL;   // matrix + index list (ints), kept 16-byte aligned
        }
        need_max = nd > need_max ? nd : need_max;
    }
    size_t mem_free = 0, mem_total = 0;
    HIPCHK(hipMemGetInfo(&mem_free, &mem_total));
    mem_free += (size_t)h->local_slab_doubles * 8;   // the slab kept from an earlier call is ours to reuse
    long long budget = (long long)std::min<size_t>(mem_free / 4, (size_t)32 << 30) / 8;   // doubles
    if (h->local_slab_mb > 0) budget = (long long)h->local_slab_mb * (1 << 20) / 8;      // option "local_slab_mb" (tests)
    if (budget < need_max) budget = need_max;
    if ((size_t)need_max * 8 > mem_free) return fail("ck_predict_local: a neighbourhood of " + std::to_string(kmx) + " sites does not fit the device memory");
    std::vector<std::pair<int64_t, int64_t>> batches;   // [begin, end)
    long long slab_doubles = 0;
    {
        int64_t b0 = 0;
        long long acc = 0;
        for (int64_t p = 0; p < m; ++p) {
            if (acc + need[p] > budget && p > b0) {
                batches.push_back({b0, p});
                slab_doubles = acc > slab_doubles ? acc : slab_doubles;
                b0 = p;
                acc = 0;
            }
            off[p] = acc;
            acc += need[p];
        }
        batches.push_back({b0, m});
        slab_doubles = acc > slab_doubles ? acc : slab_doubles;
    }
    // tiled class: largest neighbourhoods first, so that the systems still active at a column are a prefix
    std::sort(tiled.begin(), tiled.end(), [&](int64_t a, int64_t b) { return cnt[a] != cnt[b] ? cnt[a] > cnt[b] : a < b; });
    std::vector<CkLocalSys> sysv(tiled.size());
    std::vector<std::pair<size_t, size_t>> tbatches;
    {
        size_t b0 = 0;
        long long acc = 0;
        for (size_t t = 0; t < tiled.size(); ++t) {
            const long long k = cnt[tiled[t]], nd = ck_local_tiled_doubles(k);
            if (acc + nd > budget && t > b0) {
                tbatches.push_back({b0, t});
                slab_doubles = acc > slab_doubles ? acc : slab_doubles;
                b0 = t;
                acc = 0;
            }
            const int kq = (int)ck_local_tiled_kq(k);
            sysv[t] = CkLocalSys{acc, (int)k, kq, kq + 128, (int)tiled[t]};
            acc += nd;
        }
        if (!tiled.empty()) {
            tbatches.push_back({b0, tiled.size()});
            slab_doubles = acc > slab_doubles ? acc : slab_doubles;
        }
    }
    h->t_ms[14] = 0.0;
    const auto t_grow = std::chrono::steady_clock::now();
    if (slab_doubles > h->local_slab_doubles) {
        // Growing is what stalls: hipMalloc of tens of GiB right after a hipFree of a few GiB that were written
        // took 1-4 s every time (scripts/diag_malloc.py).  So: beyond 1 GiB take the whole budget at once (the
        // slab then never grows again), and allocate the new slab before releasing the old one.
        const long long want = slab_doubles * 8 > (1LL << 30) ? std::max(slab_doubles, budget) : slab_doubles;
        double* fresh = nullptr;
        if (hipMalloc((void**)&fresh, (size_t)want * 8) != hipSuccess) {
            (void)hipGetLastError();
            if (h->local_slab) (void)hipFree(h->local_slab);   // not enough room for both
            h->local_slab = nullptr;
            h->local_slab_doubles = 0;
            HIPCHK(hipMalloc((void**)&fresh, (size_t)slab_doubles * 8));
            h->local_slab_doubles = slab_doubles;
        } else {
            if (h->local_slab) (void)hipFree(h->local_slab);
            h->local_slab_doubles = want;
        }
        h->local_slab = fresh;
        // visible in ck_timings [14]; profiles/r03c_local_predictor.json's 100 km row (1 262 ms for a 3 ms call) was this
        // allocation inside the one event window of round 3
        h->t_ms[14] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_grow).count();
    }
    if (slab_doubles > 0) d_slab = h->local_slab;
    HIPCHK(hipEventRecord(h->ev2, h->stream));
    HIPCHK(hipMemcpyAsync(d_off, off.data(), m * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    const double c0var = h->blk[2 * i].amp + h->blk[2 * i].nugget;   // covariance(i, 0)[0], point_prediction.py:66
    const int use_tab = tables_usable(h) ? 1 : 0;
    for (const auto& bt : batches)
        ck_launch_local_solve(h->stream, h->d_blk, h->metric, i, cv ? 1 : 0, max_dist, d_p3, bt.first, bt.second - bt.first,
                              mp, h->s0, h->z, layout_of(h), d_cnt, d_off, d_slab, c0var, d_out, d_out + mp, h->d_tabs,
                              h->d_coefptr, use_tab, h->su, d_pu, k_hi, h->d_chunkb, cmax);
    HIPCHK(hipGetLastError());
    if (!tiled.empty()) {
        HIPCHK(tmp.get(&d_sys, (size_t)(sysv.size() * sizeof(CkLocalSys))));
        HIPCHK(tmp.get(&d_linfo, (size_t)(sysv.size() * sizeof(long long))));
        HIPCHK(tmp.get(&d_k0, (size_t)(sysv.size() * sizeof(int))));
        HIPCHK(hipMemcpyAsync(d_sys, sysv.data(), sysv.size() * sizeof(CkLocalSys), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemsetAsync(d_linfo, 0, sysv.size() * sizeof(long long), h->stream));
        for (const auto& tb : tbatches) {
            const CkLocalSys* bsys = d_sys + tb.first;
            const int nb = (int)(tb.second - tb.first);
            ck_launch_local_assemble_t(h->stream, h->d_blk, h->metric, i, cv ? 1 : 0, max_dist, d_p3, mp, h->s0, h->z,
                                       layout_of(h), bsys, nb, d_slab, h->d_tabs, h->d_coefptr, use_tab, h->su, d_pu,
                                       h->d_chunkb, cmax, d_k0 + tb.first);
            const int kq_max = sysv[tb.first].kq;
            std::vector<int> kqv(nb);
            for (int y = 0; y < nb; ++y) kqv[y] = sysv[tb.first + y].kq;
            int na = nb;
            const int G = h->local_group;
            const bool left = h->local_left != 0 && 64 * G <= 256;
            for (int g0 = 0; g0 < kq_max; g0 += 64 * G) {
                // left-looking (round 4): the group's columns first receive everything from their left in one pass (K = g0) --
                // every tile of a system is read and written once instead of once per earlier group with K = 64 G
                if (left && g0 > 0) {
                    while (na > 0 && sysv[tb.first + na - 1].kq <= g0) --na;
                    ck_launch_local_tiled_left(h->stream, bsys, d_slab, na, g0, 64 * G, kqv.data());
                }
                // the group's diagonal region block by block (diagonal block, then the few chunks of rows inside the region) ...
                for (int b = 0; b < G && g0 + 64 * b < kq_max; ++b) {
                    while (na > 0 && sysv[tb.first + na - 1].kq <= g0 + 64 * b) --na;   // finished systems drop off the end
                    ck_launch_local_tiled_block(h->stream, bsys, d_slab, na, g0, b, kqv.data(), d_linfo + tb.first, G);
                }
                // ... then every row below it through all of the group's blocks in one launch, then the trailing update
                while (na > 0 && sysv[tb.first + na - 1].kq <= g0 + 64 * G) --na;
                ck_launch_local_tiled_rows_all(h->stream, bsys, d_slab, na, g0, G, kqv.data());
                if (!left) ck_launch_local_tiled_trailing(h->stream, bsys, d_slab, na, g0, 64 * G, kqv.data());
            }
            ck_launch_local_reduce_t(h->stream, bsys, nb, d_slab, d_linfo + tb.first, c0var, d_out, d_out + mp);
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev3, h->stream));
    HIPCHK(hipMemcpyAsync(pred, d_out, m * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(pred_err, d_out + mp, m * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0, ms2 = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    HIPCHK(hipEventElapsedTime(&ms2, h->ev2, h->ev3));
    h->t_ms[10] = (double)ms + (double)ms2;   // device work: counting pass + assembly / factorisations / reductions
    int64_t npd = 0;
    for (int64_t p = 0; p < m; ++p)
        if (cnt[p] > 0 && pred[p] != pred[p]) ++npd;
    if (n_empty) *n_empty = nempty;
    if (n_not_pd) *n_not_pd = npd;
    if (k_max) *k_max = kmx;
    return 0;
}

// The reference builds the local predictor's state once, in its constructor (src/point_prediction.py:24-43: the full Sigma
// blocks); here that state is the scratch slab of the large-neighbourhood paths.  nbytes > 0: at least that much; 0: the
// automatic budget of ck_predict_local (a quarter of the free memory, at most 32 GiB; option "local_slab_mb" if set).  After
// this call no ck_predict_local whose batches fit pays a hipMalloc.
extern "C" int ck_local_reserve(ck_handle* h, int64_t nbytes) {
    CHKH(h);
    if (nbytes < 0) return fail("ck_local_reserve: negative size");
    long long want = nbytes / 8;
    if (nbytes == 0) {
        size_t mem_free = 0, mem_total = 0;
        HIPCHK(hipMemGetInfo(&mem_free, &mem_total));
        mem_free += (size_t)h->local_slab_doubles * 8;
        want = (long long)std::min<size_t>(mem_free / 4, (size_t)32 << 30) / 8;
        if (h->local_slab_mb > 0) want = (long long)h->local_slab_mb * (1 << 20) / 8;
    }
    if (want <= h->local_slab_doubles) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->local_slab) (void)hipFree(h->local_slab);
    h->local_slab = nullptr;
    h->local_slab_doubles = 0;
    HIPCHK(hipMalloc((void**)&h->local_slab, (size_t)want * 8));
    h->local_slab_doubles = want;
    h->t_ms[14] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

// ---------------------------------------------------------------------------------------
// empirical (cross-)variogram: src/fields.py:192-232
// ---------------------------------------------------------------------------------------
static void vario_free(ck_handle* h) {
    void* ps[] = {h->vg_iu, h->vg_iv, h->vg_same ? nullptr : h->vg_ju, h->vg_same ? nullptr : h->vg_jv,
                  h->vg_part, h->vg_psum, h->vg_pcnt, h->vg_out, h->vg_jb, h->vg_ib64, h->vg_jbsub,
                  h->vg_best, h->vg_list, h->vg_count};
    for (void* p : ps)
        if (p) (void)hipFree(p);
    h->vg_iu = h->vg_iv = h->vg_ju = h->vg_jv = nullptr;
    h->vg_jb = h->vg_ib64 = h->vg_jbsub = nullptr;
    h->vg_best = nullptr;
    h->vg_part = nullptr;
    h->vg_psum = nullptr;
    h->vg_pcnt = nullptr;
    h->vg_out = nullptr;
    h->vg_list = nullptr;
    h->vg_count = nullptr;
    h->vg_list_cap = 0;
}

// Upload one field's points.  From 2 048 points on (and unless site_order = 0) they are first laid out along a
// Hilbert curve: bins sums and counts do not depend on the order of the points (up to rounding of the sums), and
// compact blocks of points are what lets the kernels skip whole pair tiles (ck_vario.hip, "tile culling").
// host_coords / host_vals receive the points in the order the device sees them (pairs come back as indices).
static int vario_upload(ck_handle* h, const double* coords, const double* vals, int64_t n, double** u, double** v,
                        std::vector<double>& host_coords, std::vector<double>& host_vals) {
    host_coords.assign(coords, coords + 2 * n);
    host_vals.assign(vals, vals + n);
    if (h->site_order && n >= 2048) {
        double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
        ck_host_bounding_box(coords, n, lo, hi);
        std::vector<int64_t> perm;
        ck_host_hilbert_order(coords, n, lo, hi, perm);
        ck_host_parallel(n, [&](int, int64_t b, int64_t e2) {
            for (int64_t k = b; k < e2; ++k) {
                const int64_t e = perm[(size_t)k];
                host_coords[2 * k] = coords[2 * e];
                host_coords[2 * k + 1] = coords[2 * e + 1];
                host_vals[(size_t)k] = vals[e];
            }
        });
    }
    DevTemps tmp;
    double* stage = nullptr;
    HIPCHK(hipMalloc((void**)u, 3 * n * 8));
    HIPCHK(hipMalloc((void**)v, n * 8));
    HIPCHK(tmp.get(&stage, (size_t)(2 * n * 8)));
    HIPCHK(hipMemcpyAsync(stage, host_coords.data(), 2 * n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(*v, host_vals.data(), n * 8, hipMemcpyHostToDevice, h->stream));
    ck_launch_vario_prep(h->stream, stage, n, h->metric, *u, *u + n, *u + 2 * n);
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int ck_vario_begin(ck_handle* h, const double* coords_i, const double* resid_i, int64_t n_i,
                              const double* coords_j, const double* resid_j, int64_t n_j, int same) {
    CHKH(h);
    if (n_i <= 0 || !coords_i || !resid_i) return fail("bad field i");
    if (!same && (n_j <= 0 || !coords_j || !resid_j)) return fail("bad field j");
    // the binning kernel addresses the "j" arrays with 32-bit byte offsets (8 n_j < 2^32) and lists pairs as int indices
    if (n_i >= (1LL << 28) || n_j >= (1LL << 28)) return fail("at most 2^28 - 1 points per field");
    vario_free(h);
    h->vg_same = same ? 1 : 0;
    h->vg_ni = n_i;
    if (vario_upload(h, coords_i, resid_i, n_i, &h->vg_iu, &h->vg_iv, h->vg_ci, h->vg_vi)) return -1;
    if (same) {
        h->vg_nj = n_i;
        h->vg_cj = h->vg_ci;
        h->vg_vj = h->vg_vi;
        h->vg_ju = h->vg_iu;
        h->vg_jv = h->vg_iv;
    } else {
        h->vg_nj = n_j;
        if (vario_upload(h, coords_j, resid_j, n_j, &h->vg_ju, &h->vg_jv, h->vg_cj, h->vg_vj)) return -1;
    }
    HIPCHK(hipMalloc((void**)&h->vg_ib64, (size_t)(4 * ck_vario_nblocks(h->vg_ni, 64) * 8)));
    HIPCHK(hipMalloc((void**)&h->vg_jb, (size_t)(4 * ck_vario_nblocks(h->vg_nj, CK_VG_JCHUNK) * 8)));
    HIPCHK(hipMalloc((void**)&h->vg_jbsub, (size_t)(4 * ck_vario_nblocks(h->vg_nj, CK_VG_JSUB) * 8)));
    HIPCHK(hipMalloc((void**)&h->vg_best, 16));
    ck_launch_vario_bounds(h->stream, h->vg_iu, h->vg_ni, 64, h->vg_ib64);
    ck_launch_vario_bounds(h->stream, h->vg_ju, h->vg_nj, CK_VG_JCHUNK, h->vg_jb);
    ck_launch_vario_bounds(h->stream, h->vg_ju, h->vg_nj, CK_VG_JSUB, h->vg_jbsub);
    HIPCHK(hipGetLastError());
    h->vg_bgrid = ck_vario_bin_grid(h->vg_ni, h->vg_nj);
    HIPCHK(hipMalloc(&h->vg_part, h->vg_bgrid * sizeof(CkVarioExt)));
    HIPCHK(hipMalloc((void**)&h->vg_psum, (size_t)h->vg_bgrid * CK_VG_MAXBINS * 8));
    HIPCHK(hipMalloc((void**)&h->vg_pcnt, (size_t)h->vg_bgrid * (CK_VG_MAXBINS + 1) * 8));
    HIPCHK(hipMalloc((void**)&h->vg_out, (3 * (CK_VG_MAXBINS + 2) + CK_VG_MAXBINS + CK_VG_MAXBINS + 1) * 8 + CK_VG_ARGS_BYTES));
    HIPCHK(hipMalloc((void**)&h->vg_count, sizeof(unsigned)));
    h->vg_list_cap = 1u << 20;
    HIPCHK(hipMalloc((void**)&h->vg_list, (size_t)h->vg_list_cap * sizeof(CkVarioPair)));
    for (int k = 0; k < 4; ++k) h->vg_stats[k] = 0;
    return 0;
}

// the list a kernel has just filled, on the host; *overflow = the kernel wanted more room than the list has --
// the list is then re-allocated and the caller runs the kernel again
static int vario_fetch_list(ck_handle* h, std::vector<CkVarioPair>& out, bool* overflow) {
    unsigned cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, h->vg_count, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *overflow = cnt > h->vg_list_cap;
    if (*overflow) {
        (void)hipFree(h->vg_list);
        h->vg_list = nullptr;
        const uint64_t want = (uint64_t)cnt + cnt / 4 + 1024;
        if (want > 0xfffffff0ull) return fail("variogram: too many pairs on a bin edge for the host list");
        h->vg_list_cap = (unsigned)want;
        HIPCHK(hipMalloc((void**)&h->vg_list, (size_t)h->vg_list_cap * sizeof(CkVarioPair)));
        return 0;
    }
    out.resize(cnt);
    if (cnt) HIPCHK(hipMemcpy(out.data(), h->vg_list, (size_t)cnt * sizeof(CkVarioPair), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int ck_vario_extent(ck_handle* h, double max_dist, double* lo, double* hi, int64_t* n_positive) {
    CHKH(h);
    if (!h->vg_iu) return fail("ck_vario_begin has not been called");
    if (!(max_dist >= 0.0)) return fail("max_dist must be >= 0");
    *lo = *hi = NAN;
    *n_positive = 0;
    const int metric = h->metric;
    const double qcap0 = ck_host_vario_q_of_dist(metric, max_dist);
    double cap = qcap0 + ck_host_vario_band(metric, qcap0);   // pairs up to here may still have d <= max_dist
    bool have_lo = false, have_hi = false;
    double best_lo = INFINITY, best_hi = -1.0;
    // Round 0 normally settles both extremes.  Further rounds only when every pair within the band of the largest
    // q <= cap turns out to lie beyond max_dist: the cap then moves below them.
    // best_lo / best_hi over the candidates of one list, by the reference's own arithmetic (ck_host.cpp)
    auto decide = [&](const std::vector<CkVarioPair>& cand) {
        ck_host_vario_decide_extent(metric, h->vg_ci.data(), h->vg_cj.data(), cand.data(), (int64_t)cand.size(), max_dist,
                                    &best_lo, &best_hi);
        have_hi = best_hi >= 0.0;
        have_lo = best_lo < INFINITY;
    };
    for (int round = 0; round < 64 && !have_hi; ++round) {
        // The pass lists, on its way, every pair in a thin window under the cap: with dense data the largest retained q
        // lies inside it, and the candidates for the largest distance are then complete without a second pass.
        const double win = fmax(1e-9 * cap, 8.0 * ck_host_vario_band(metric, cap));
        const double qwin_lo = cap - win;
        HIPCHK(hipMemsetAsync(h->vg_count, 0, sizeof(unsigned), h->stream));
        ck_launch_vario_extent(h->stream, h->vg_bgrid, h->vg_same, h->vg_iu, h->vg_ni, h->vg_ju, h->vg_nj, cap, h->vg_part,
                               h->rank, h->world, h->vg_ib64, h->vg_jb, h->vg_jbsub, ck_host_vario_cmax(cap), h->vg_best, qwin_lo,
                               h->vg_list, h->vg_count, h->vg_list_cap);
        HIPCHK(hipGetLastError());
        std::vector<CkVarioExt> part(h->vg_bgrid);
        HIPCHK(hipMemcpyAsync(part.data(), h->vg_part, h->vg_bgrid * sizeof(CkVarioExt), hipMemcpyDeviceToHost, h->stream));
        std::vector<CkVarioPair> wcand;
        bool woverflow = false;
        if (vario_fetch_list(h, wcand, &woverflow)) return -1;   // synchronises the stream
        CkVarioExt best = part[0];
        for (int g = 1; g < h->vg_bgrid; ++g) {
            if (part[g].rmin < best.rmin) {
                best.rmin = part[g].rmin;
                best.imin = part[g].imin;
            }
            if (part[g].rmax > best.rmax) {
                best.rmax = part[g].rmax;
                best.imax = part[g].imax;
            }
        }
        if (best.imax < 0) break;   // no pair with q <= cap at all
        // every pair whose q is within the band of an extreme is a candidate; the reference's formula decides
        double qtop_lo = best.rmax - 2.0 * ck_host_vario_band(metric, best.rmax);
        const double qbot_hi = (!have_lo && best.imin >= 0) ? best.rmin + 2.0 * ck_host_vario_band(metric, best.rmin) : -1.0;
        bool top_done = false;
        if (!woverflow && qtop_lo >= qwin_lo) {   // the window holds every top candidate
            h->vg_stats[0] += (int64_t)wcand.size();
            decide(wcand);
            top_done = have_hi;   // else: all of them beyond max_dist -> the full candidate pass below, then a lower cap
        }
        if (!top_done || qbot_hi > 0.0) {
            std::vector<CkVarioPair> cand;
            const double top_from = top_done ? INFINITY : qtop_lo;   // top candidates already decided: the bottom ones only
            for (int pass = 0; pass < 3; ++pass) {
                HIPCHK(hipMemsetAsync(h->vg_count, 0, sizeof(unsigned), h->stream));
                ck_launch_vario_collect(h->stream, h->vg_bgrid, h->vg_same, h->vg_iu, h->vg_ni, h->vg_ju, h->vg_nj, top_from, cap,
                                        qbot_hi, h->vg_list, h->vg_count, h->vg_list_cap, h->rank, h->world, h->vg_ib64, h->vg_jb,
                                        h->vg_jbsub);
                HIPCHK(hipGetLastError());
                bool overflow = false;
                if (vario_fetch_list(h, cand, &overflow)) return -1;
                if (!overflow) break;
                if (pass == 2) return fail("variogram: candidate list kept overflowing");
            }
            h->vg_stats[0] += (int64_t)cand.size();
            decide(cand);
        }
        if (round) h->vg_stats[3] += 1;
        if (!have_hi) {
            if (!(qtop_lo > 0.0)) break;
            cap = nextafter(qtop_lo, 0.0);   // everything from qtop_lo up is beyond max_dist
        }
    }
    if (have_hi) *hi = best_hi;
    if (have_lo) *lo = best_lo;
    *n_positive = (have_lo && have_hi) ? 1 : 0;
    return 0;
}

extern "C" int ck_vario_bin(ck_handle* h, double max_dist, const double* edges, int n_edges, int covariogram,
                            double* sums, int64_t* counts) {
    CHKH(h);
    if (!h->vg_iu) return fail("ck_vario_begin has not been called");
    const int nb = n_edges - 1;
    if (nb < 1 || nb > CK_VG_MAXBINS) return fail("n_bins must be between 1 and " + std::to_string(CK_VG_MAXBINS));
    for (int b = 0; b < nb; ++b)
        if (!(edges[b + 1] > edges[b])) return fail("bin edges must increase");
    if (edges[0] != 0.0) return fail("first bin edge must be 0 (src/fields.py:402)");
    if (!(max_dist > 0.0)) return fail("max_dist must be positive");
    const int metric = h->metric;
    // levels, their rounding bands and the clusters of levels whose bands overlap: ck_host.cpp
    CkVarioLevels lv;
    if (ck_host_vario_levels(metric, max_dist, edges, nb, &lv)) return -1;
    const int E = lv.E, EC = lv.EC;
    const double *cxa = lv.cxa, *cxb = lv.cxb, *cthr = lv.cthr;
    const int* clast = lv.clast;
    const double q_reach = lv.q_reach;
    double* d_xa = h->vg_out;
    double* d_xb = h->vg_out + (CK_VG_MAXBINS + 2);
    double* d_dthr = h->vg_out + 2 * (CK_VG_MAXBINS + 2);
    double* d_sums = h->vg_out + 3 * (CK_VG_MAXBINS + 2);
    long long* d_cnt = (long long*)(d_sums + CK_VG_MAXBINS);
    void* d_args = (void*)(d_cnt + CK_VG_MAXBINS + 1);
    HIPCHK(hipMemcpyAsync(d_xa, cxa, (EC + 1) * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_xb, cxb, (EC + 1) * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_dthr, cthr, (EC + 1) * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));   // the arrays are on the stack
    std::vector<CkVarioPair> fix;
    for (int pass = 0; pass < 3; ++pass) {
        HIPCHK(hipMemsetAsync(h->vg_count, 0, sizeof(unsigned), h->stream));
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        ck_launch_vario_bin(h->stream, metric, h->vg_same, covariogram ? 1 : 0, h->vg_iu, h->vg_iv, h->vg_ni, h->vg_ju,
                            h->vg_jv, h->vg_nj, EC, d_xa, d_xb, d_dthr, ck_host_vario_cmax(q_reach), h->vg_ib64, h->vg_jb, h->vg_jbsub,
                            h->vg_bgrid, h->vg_psum, h->vg_pcnt, h->vg_list, h->vg_count, h->vg_list_cap, h->rank, h->world,
                            EC, d_sums, d_cnt, d_args);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        bool overflow = false;
        if (vario_fetch_list(h, fix, &overflow)) return -1;
        if (!overflow) break;
        if (pass == 2) return fail("variogram: edge-pair list kept overflowing");
    }
    long long dcn[CK_VG_MAXBINS + 1], cnt[CK_VG_MAXBINS + 1];
    std::vector<double> dsm(CK_VG_MAXBINS, 0.0), sm(CK_VG_MAXBINS + 1, 0.0);
    HIPCHK(hipMemcpyAsync(dsm.data(), d_sums, EC * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(dcn, d_cnt, (CK_VG_MAXBINS + 1) * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int b = 0; b <= CK_VG_MAXBINS; ++b) cnt[b] = 0;
    for (int k = 0; k < EC; ++k) {   // device bin k -> real bin clast[k] (< E: bins from the cap up hold nothing)
        sm[(size_t)clast[k]] = dsm[(size_t)k];
        cnt[clast[k]] = dcn[k];
    }
    // The pairs inside the band of a level (cluster) were binned below it; the reference's formula on libm decides.
    ck_host_vario_fix(metric, h->vg_ci.data(), h->vg_cj.data(), h->vg_vi.data(), h->vg_vj.data(), fix.data(), (int64_t)fix.size(),
                      lv, covariogram ? 1 : 0, sm.data(), cnt);
    for (int b = E; b < nb; ++b) {   // bins above the cap hold nothing
        sm[(size_t)b] = 0.0;
        cnt[b] = 0;
    }
    for (int b = 0; b < nb; ++b) {
        sums[b] = sm[(size_t)b];
        counts[b] = cnt[b];
    }
    h->vg_stats[1] = (int64_t)fix.size();
    h->vg_stats[2] = dcn[CK_VG_MAXBINS];
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->t_ms[9] = ms;
    return 0;
}

extern "C" int ck_vario_stats(ck_handle* h, int64_t* out, int n) {
    CHKH(h);
    for (int k = 0; k < n && k < 4; ++k) out[k] = h->vg_stats[k];
    return 0;
}

extern "C" int ck_vario_end(ck_handle* h) {
    CHKH(h);
    vario_free(h);
    return 0;
}

// ---------------------------------------------------------------------------------------
// diagnostics
// ---------------------------------------------------------------------------------------
extern "C" int ck_debug_site_order(ck_handle* h, int k, int64_t* perm_out, int64_t n_k) {
    CHKH(h);
    if (k < 0 || k >= h->n_procs) return fail("process index out of range");
    if (ensure_layout(h)) return -1;
    if (n_k != h->n[k]) return fail("n_k must equal the number of observations of process k");
    for (int64_t j = 0; j < n_k; ++j) perm_out[j] = h->perm[k][(size_t)j];
    return 0;
}

extern "C" int ck_debug_get_lower(ck_handle* h, double* out, int64_t n) {
    CHKH(h);
    if (!h->assembled) return fail("nothing assembled");
    if (n != h->N) return fail("n must equal the number of observations");
    std::vector<double> buf;
    memset(out, 0, (size_t)n * n * 8);
    const int64_t gap = h->n0p - h->n[0];
    auto ext = [&](int64_t g) -> int64_t {   // internal 0-based -> external 0-based, -1 for padding
        if (g < h->n[0]) return g;
        if (g >= h->n0p && g < h->nend) return g - gap;
        return -1;
    };
    for (int K = 0; K < h->nK; ++K) {
        if (!h->sig[K]) continue;
        const int64_t rows = h->Npad - (int64_t)K * CK_NB;
        buf.resize(rows * CK_NB);
        HIPCHK(hipMemcpyAsync(buf.data(), h->sig[K], rows * CK_NB * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int64_t r = 0; r < rows; ++r) {
            const int64_t er = ext((int64_t)K * CK_NB + r);
            if (er < 0) continue;
            for (int64_t c = 0; c < CK_NB; ++c) {
                const int64_t gc = (int64_t)K * CK_NB + c;
                if (gc > (int64_t)K * CK_NB + r) break;
                const int64_t ec = ext(gc);
                if (ec < 0) continue;
                out[er * n + ec] = buf[r * CK_NB + c];
            }
        }
    }
    return 0;
}

// out[e] = entry (rows[e], cols[e]) of the lower triangle held in the packed panels (internal, padded indices; r >= c)
__global__ void k_gather_entries(double* const* __restrict__ sigptr, const long long* __restrict__ rows,
                                 const long long* __restrict__ cols, long n, double* __restrict__ out) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const long r = rows[e], c = cols[e];
    const long K = c / CK_NB;
    out[e] = sigptr[K][(r - K * CK_NB) * CK_NB + (c - K * CK_NB)];
}

extern "C" int ck_debug_get_entries(ck_handle* h, const int64_t* rows, const int64_t* cols, int64_t n, double* out) {
    CHKH(h);
    if (!h->assembled) return fail("ck_assemble_joint has not been called");
    if (h->world != 1) return fail("ck_debug_get_entries is the single-process form");
    if (n < 0 || (n > 0 && (!rows || !cols || !out))) return fail("bad arguments");
    if (n == 0) return 0;
    // caller's stacked index (process 0 sites, then process 1) -> internal padded index
    std::vector<int64_t> inv((size_t)h->N);
    for (int k = 0; k < h->n_procs; ++k) {
        const int64_t off_c = k == 0 ? 0 : h->n[0], off_i = k == 0 ? 0 : h->n0p;
        for (int64_t j = 0; j < h->n[k]; ++j) inv[(size_t)(off_c + h->perm[k][(size_t)j])] = off_i + j;
    }
    std::vector<long long> hr((size_t)n), hc((size_t)n);
    for (int64_t e = 0; e < n; ++e) {
        if (rows[e] < 0 || rows[e] >= h->N || cols[e] < 0 || cols[e] >= h->N) return fail("entry index out of range");
        long long r = inv[(size_t)rows[e]], c = inv[(size_t)cols[e]];
        if (r < c) std::swap(r, c);   // symmetric: only the lower triangle is stored
        hr[(size_t)e] = r;
        hc[(size_t)e] = c;
    }
    DevTemps tmp;
    long long *dr = nullptr, *dc = nullptr;
    double* dout = nullptr;
    HIPCHK(tmp.get(&dr, (size_t)n * 8));
    HIPCHK(tmp.get(&dc, (size_t)n * 8));
    HIPCHK(tmp.get(&dout, (size_t)n * 8));
    HIPCHK(hipMemcpyAsync(dr, hr.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dc, hc.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    k_gather_entries<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream>>>(h->d_sigptr, dr, dc, (long)n, dout);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// Phase profile of the 64 x 64 diagonal-block kernel on a well-conditioned block: out[0..5] = microseconds of
// load | factorisation | scaling + store | inverse of the diagonal 16 x 16 blocks | off-diagonal blocks | store of the
// inverse (shader clock / 100 MHz reference), out[6] = the whole kernel by HIP events, out[7] = the same for the product
// kernel (k_potrf64), average over `iters` back-to-back launches.
extern "C" int ck_debug_potrf_profile(ck_handle* h, int iters, double* out8) {
    CHKH(h);
    if (!out8 || iters < 1) return fail("bad arguments");
    DevTemps tmp;
    double *dA = nullptr, *dL = nullptr;
    long long *dprof = nullptr, *dinfo = nullptr;
    HIPCHK(tmp.get(&dA, 64 * 512 * 8));
    HIPCHK(tmp.get(&dL, 64 * 64 * 8));
    HIPCHK(tmp.get(&dprof, 16 * 8));
    HIPCHK(tmp.get(&dinfo, 8));
    std::vector<double> A((size_t)64 * 512, 0.0);
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j <= i; ++j) A[(size_t)i * 512 + j] = (i == j ? 70.0 : 0.0) + 1.0 / (1.0 + i + j);
    HIPCHK(hipMemsetAsync(dinfo, 0, 8, h->stream));
    HIPCHK(hipMemsetAsync(dprof, 0, 16 * 8, h->stream));
    double clk_mhz = 100.0;   // provisional unit; rescaled below so that the phases add up to the instrumented launch
    long long hp[16];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int it = 0; it < 4; ++it) {
        HIPCHK(hipMemcpyAsync(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice, h->stream));
        ck_launch_potrf64_prof(h->stream, dA, 512, dinfo, dL, dprof);
        HIPCHK(hipMemcpyAsync(hp, dprof, sizeof(hp), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (it)
            for (int k = 0; k < 6; ++k) acc[k] += (double)(hp[k + 1] - hp[k]) / clk_mhz / 3.0;
    }
    for (int k = 0; k < 6; ++k) out8[k] = acc[k];
    for (int which = 0; which < 2; ++which) {
        HIPCHK(hipMemcpyAsync(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        for (int it = 0; it < iters; ++it) {   // refactoring the factor is harmless for timing (it stays positive definite enough)
            if (which == 0)
                ck_launch_potrf64_prof(h->stream, dA, 512, dinfo, dL, dprof);
            else
                ck_launch_potrf64(h->stream, dA, 512, 0, dinfo, dL);
        }
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        HIPCHK(hipEventSynchronize(h->ev1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        out8[6 + which] = (double)ms * 1e3 / iters;
    }
    {   // s_memtime ticks at the shader clock: scale the phases so that they add up to the instrumented launch's duration
        double sum = 0;
        for (int k = 0; k < 6; ++k) sum += out8[k];
        if (sum > 0)
            for (int k = 0; k < 6; ++k) out8[k] *= out8[6] / sum;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// Where a link of the cooperative panel step's chain spends its time: k_panel_coop on a well-conditioned rows x 512 panel;
// out[8 * b + k], b = 1 .. 7: microseconds from the moment chunk b saw the pivot chunk's "rows final" flag (k = 0) to
// k = 1 accumulation done | 2 inverse flag seen | 3 rows solved and stored | 4 drained + rows flag set | 5 diagonal block
// updated | 6 factored + inverse stored | 7 drained + flag set; out[0] = the whole launch by HIP events, out[1] = shader MHz.
extern "C" int ck_debug_coop_profile(ck_handle* h, int64_t rows, double* out64) {
    CHKH(h);
    if (!out64 || rows < CK_NB || rows % 64) return fail("bad arguments");
    DevTemps tmp;
    double *dP = nullptr;
    long long *dprof = nullptr, *dinfo = nullptr;
    HIPCHK(tmp.get(&dP, (size_t)(rows * CK_NB + CK_PANEL_TAIL) * 8));
    HIPCHK(tmp.get(&dprof, 64 * 8));
    HIPCHK(tmp.get(&dinfo, 8));
    std::vector<double> A((size_t)rows * CK_NB);
    for (int64_t i = 0; i < rows; ++i)
        for (int j = 0; j < CK_NB; ++j) A[(size_t)i * CK_NB + j] = (i == j ? 600.0 : 0.0) + 1.0 / (1.0 + (double)(i % 977) + j);
    long long hp[64];
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        HIPCHK(hipMemcpyAsync(dP, A.data(), A.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemsetAsync(dinfo, 0, 8, h->stream));
        HIPCHK(hipMemsetAsync(dprof, 0, 64 * 8, h->stream));
        h->coop_seq += 1;
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        ck_launch_panel_coop_prof(h->stream, dP, rows, dP + rows * CK_NB, 0, dinfo, h->d_coop, h->coop_seq, h->d_coop + 16, dprof);
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        HIPCHK(hipMemcpyAsync(hp, dprof, sizeof(hp), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    }
    HIPCHK(hipGetLastError());
    // the chain from link 1's first stamp to link 7's last takes (launch - head - tail); the stamps tick at the shader clock:
    // calibrate it on the span of the whole chain against the event time is too coarse -- report ticks / 2 400 MHz
    const double mhz = 2400.0;
    for (int k = 0; k < 64; ++k) out64[k] = 0.0;
    for (int b = 1; b <= 7; ++b)
        for (int k = 1; k < 8; ++k) out64[8 * b + k] = (double)(hp[8 * b + k] - hp[8 * b]) / mhz;
    for (int b = 2; b <= 7; ++b) out64[8 * b] = (double)(hp[8 * b] - hp[8 * (b - 1)]) / mhz;   // link-to-link period
    out64[0] = (double)ms * 1e3;
    out64[1] = mhz;
    return 0;
}

// Diagnostic: do small dependent kernels on the high-priority side stream run WHILE a chip-filling trailing update is being
// dispatched on the main stream, or only after its grid has been placed?  (What a look-ahead of the panel chain under
// the bulk of the previous group's update needs.)  On an assembled, unfactored handle of >= 8 panels (Sigma is destroyed:
// assemble again afterwards): mode 0: the update alone | 1: n_side cooperative panel steps on a scratch panel of `rows`
// rows alone | 2: both -- the first side kernel submitted in front of the update and released by the same event, the others
// behind it in stream order.  out[0] = update ms (mode 1: 0), out[1 + i] = end of side kernel i after the common start, ms.
extern "C" int ck_debug_stream_overlap(ck_handle* h, int mode, int64_t rows, int n_side, double* out) {
    CHKH(h);
    if (!h->assembled || h->factored || h->nK < 8 || h->world != 1) return fail("needs an assembled, unfactored single-process handle of >= 8 panels");
    if (!out || rows < CK_NB || rows % 64 || n_side < 0 || n_side > 16 || mode < 0 || mode > 2) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    DevTemps tmp;
    double* dP = nullptr;
    long long* dinfo = nullptr;
    HIPCHK(tmp.get(&dP, (size_t)(rows * CK_NB + CK_PANEL_TAIL) * 8));
    HIPCHK(tmp.get(&dinfo, 8));
    std::vector<double> A((size_t)rows * CK_NB);
    for (int64_t i = 0; i < rows; ++i)
        for (int j = 0; j < CK_NB; ++j) A[(size_t)i * CK_NB + j] = (i == j ? 600.0 : 0.0) + 1.0 / (1.0 + (double)(i % 977) + j);
    HIPCHK(hipMemcpyAsync(dP, A.data(), A.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(dinfo, 0, 8, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    struct Events {   // destroyed on every return path (the HIPCHK early returns used to leak them)
        std::vector<hipEvent_t> v;
        ~Events() {
            for (hipEvent_t e : v)
                if (e) (void)hipEventDestroy(e);
        }
    } evs;
    evs.v.assign((size_t)n_side + 2, nullptr);
    std::vector<hipEvent_t>& ev = evs.v;
    for (auto& e : ev) HIPCHK(hipEventCreate(&e));
    hipStream_t M = h->stream, S = h->side;
    HIPCHK(hipEventRecord(ev[0], M));
    if (mode >= 1) {
        HIPCHK(hipStreamWaitEvent(S, ev[0], 0));
        for (int i = 0; i < n_side; ++i) {
            h->coop_seq += 1;
            ck_launch_panel_coop(S, dP, rows, dP + rows * CK_NB, 0, dinfo, h->d_coop, h->coop_seq, h->d_coop + 16);
            HIPCHK(hipEventRecord(ev[2 + i], S));
        }
    }
    if (mode != 1) syrk_update(h, M, 0, 3, 3, 1, h->nK - 3);
    HIPCHK(hipEventRecord(ev[1], M));
    HIPCHK(hipStreamSynchronize(M));
    HIPCHK(hipStreamSynchronize(S));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev[0], ev[1]));
    out[0] = mode != 1 ? ms : 0.0;
    for (int i = 0; i < n_side; ++i) {
        out[1 + i] = 0.0;
        if (mode >= 1) {
            HIPCHK(hipEventElapsedTime(&ms, ev[0], ev[2 + i]));
            out[1 + i] = ms;
        }
    }
    h->assembled = false;
    return 0;
}

extern "C" int ck_debug_mfma_probe(ck_handle* h, int32_t* out) {
    CHKH(h);
    int32_t* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, 64 * 4 * 3 * sizeof(int32_t)));
    ck_launch_mfma_probe(h->stream, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d, 64 * 4 * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    (void)hipFree(d);
    return 0;
}

extern "C" int ck_estimate_bytes(ck_handle* h, int64_t m, int64_t* out) {
    CHKH(h);
    for (int k = 0; k < h->n_procs; ++k)
        if (!h->data_set[k]) return fail("ck_set_data missing for process " + std::to_string(k));
    const int64_t n1 = h->n_procs == 2 ? h->n[1] : 0;
    const int64_t Np = roundup((n1 > 0 ? roundup(h->n[0], 64) : h->n[0]) + n1, CK_NB);
    const int nK = (int)(Np / CK_NB);
    auto al = [](int64_t b) { return (b + 255) & ~(int64_t)255; };
    int64_t tot = 2 * al(3 * Np * 8) + al(Np * 8);
    for (int K = h->rank; K < nK; K += h->world) tot += al(((Np - (int64_t)K * CK_NB) * CK_NB + CK_PANEL_TAIL) * 8 + CK_PANEL_SLACK_BYTES);
    tot += 2 * al((int64_t)nK * sizeof(double*));   // d_sigptr, d_panelptr
    if (h->world > 1) tot += (int64_t)h->recv_slots * al((Np * CK_NB + CK_PANEL_TAIL) * 8 + CK_PANEL_SLACK_BYTES);
    const int64_t mpad = roundup(m + 1, CK_AUX_ALIGN);
    tot += al(mpad * Np * 8) + 2 * al(3 * mpad * 8) + 2 * al(2 * mpad * 8);
    *out = tot + 4096;
    return 0;
}

extern "C" int ck_debug_cu_probe(ck_handle* h, const uint32_t* cu_mask8, int n_wg, uint32_t* out) {
    CHKH(h);
    if (n_wg <= 0 || n_wg > 65536 || !out) return fail("bad n_wg");
    hipStream_t st = nullptr;
    if (cu_mask8)
        HIPCHK(hipExtStreamCreateWithCUMask(&st, 8, cu_mask8));
    else
        HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)n_wg * 4));
    HIPCHK(hipMemsetAsync(d, 0xff, (size_t)n_wg * 4, st));
    ck_launch_cu_probe(st, d, n_wg, 4);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d, (size_t)n_wg * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(d);
    (void)hipStreamDestroy(st);
    return 0;
}

extern "C" int ck_debug_mfma_peak(ck_handle* h, int waves_per_simd, int iters, double* out3) {
    CHKH(h);
    if (waves_per_simd < 1 || waves_per_simd > 8 || iters < 1) return fail("bad arguments");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, h->device));
    const int cus = prop.multiProcessorCount;
    double* sink = nullptr;
    HIPCHK(hipMalloc((void**)&sink, 32));
    const int threads = 256, blocks = cus * waves_per_simd;   // 4 waves per block = one per SIMD
    ck_launch_mfma_peak(h->stream, blocks, waves_per_simd, 10, sink);   // warm-up
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    const int nacc = ck_launch_mfma_peak(h->stream, blocks, waves_per_simd, iters, sink);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    const double flops = (double)blocks * (threads / 64) * (double)iters * nacc * 2048.0;
    double hs[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(hs, sink, 32, hipMemcpyDeviceToHost));
    out3[0] = flops / (ms * 1e-3) / 1e12;                           // TFLOP/s
    out3[1] = hs[2] > 0 ? hs[1] / hs[2] * 100.0 : 0.0;              // in-kernel shader clock, MHz
    out3[2] = hs[1] / ((double)iters * nacc);   // cycles per MFMA of one wave
    (void)hipFree(sink);
    return 0;
}

// Diagnostic: the clock the chip holds under the trailing-update kernel on the data of the last factorisation.  With
// option "gemm_stamps" = 1 every workgroup of k_syrk_group_d leaves the shader cycles and 100 MHz ticks of its
// lifetime; out6 = median / 5 % / 95 % quantile of the shader clock in MHz over the stamped workgroups, their number,
// the median lifetime in shader cycles and in microseconds.
extern "C" int ck_debug_gemm_clock(ck_handle* h, double* out6) {
    CHKH(h);
    if (!h->d_stamps) return fail("ck_debug_gemm_clock: set option gemm_stamps = 1 and factor first");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> st(4 * h->n_stamps);
    HIPCHK(hipMemcpy(st.data(), h->d_stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> mhz, cyc;
    for (size_t b = 0; b < h->n_stamps; ++b)
        if (st[4 * b + 1] > 1000) {   // > 10 us: a workgroup that ran a tile
            mhz.push_back((double)st[4 * b] / (double)st[4 * b + 1] * 100.0);
            cyc.push_back((double)st[4 * b]);
        }
    for (int i = 0; i < 6; ++i) out6[i] = 0;
    if (mhz.empty()) return 0;
    std::sort(mhz.begin(), mhz.end());
    std::sort(cyc.begin(), cyc.end());
    out6[0] = mhz[mhz.size() / 2];
    out6[1] = mhz[mhz.size() / 20];
    out6[2] = mhz[mhz.size() - 1 - mhz.size() / 20];
    out6[3] = (double)mhz.size();
    out6[4] = cyc[cyc.size() / 2];
    out6[5] = out6[4] / out6[0];
    return 0;
}

// raw stamps of the last stamped launch: out[4 b .. 4 b + 3] = shader cycles, 100 MHz ticks of workgroup b's lifetime (0, 0
// if it returned at once), its start in 100 MHz ticks, XCC_ID << 32 | HW_ID; grid4 = grid x, y, first block column, panels
extern "C" int ck_debug_gemm_stamps(ck_handle* h, uint64_t* out_host, int64_t n_words, int64_t* grid4) {
    CHKH(h);
    if (!h->d_stamps) return fail("ck_debug_gemm_stamps: set option gemm_stamps and factor first");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t n = std::min<size_t>((size_t)std::max<int64_t>(n_words, 0), 4 * h->n_stamps);
    HIPCHK(hipMemcpy(out_host, h->d_stamps, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) grid4[i] = h->stamp_grid[i];
    return 0;
}

extern "C" int ck_set_option(ck_handle* h, const char* name, int64_t value) {
    CHKH(h);
    if (!name) return fail("null option name");
    if (!strcmp(name, "gemm_stamps")) {   // diagnostic, see ck_debug_gemm_clock; after the first assemble
        if (h->d_stamps) {
            (void)hipFree(h->d_stamps);
            h->d_stamps = nullptr;
            h->n_stamps = 0;
        }
        if (value != 0) {
            if (!h->layout_ready) return fail("gemm_stamps: assemble first");
            HIPCHK(hipSetDevice(h->device));
            h->n_stamps = (size_t)(h->Npad / 128) * (CK_NB / 128) * (size_t)(h->Npad / CK_NB);
            HIPCHK(hipMalloc((void**)&h->d_stamps, 4 * h->n_stamps * sizeof(unsigned long long)));
            HIPCHK(hipMemset(h->d_stamps, 0, 4 * h->n_stamps * sizeof(unsigned long long)));
        }
        h->stamp_sel = (int)value;
        return 0;
    }
    if (!strcmp(name, "time_gemm")) {
        if (value < 0 || value > 2) return fail("time_gemm must be 0, 1 or 2");
        h->time_gemm = (int)value;
        return 0;
    }
    if (!strcmp(name, "fused_sweeps")) {
        if (value < -1 || value > 1) return fail("fused_sweeps must be -1 (automatic), 0 or 1");
        h->fused_sweeps_opt = (int)value;
        return 0;
    }
    if (!strcmp(name, "fused_prio")) {   // see ck_handle::fused_prio
        if (value < 0 || value > 2) return fail("fused_prio must be 0, 1 or 2");
        h->fused_prio = (int)value;
        return 0;
    }
    if (!strcmp(name, "coop_spins")) {
        if (value < 1000 || value > 2000000000LL) return fail("coop_spins must be in [1000, 2e9]");
        h->coop_spins = (unsigned)value;
        return 0;
    }
    if (!strcmp(name, "coop_inject_panel")) {   // tests: the bounded wait of k_panel_coop must trip and be recovered from
        if (value < -1) return fail("coop_inject_panel must be -1 (off) or a panel index");
        h->coop_inject_panel = (int)value;
        return 0;
    }
    if (!strcmp(name, "group_first") || !strcmp(name, "group_tail") || !strcmp(name, "group_tail_panels")) {   // see group_plan()
        if (value < (strcmp(name, "group_first") ? 0 : -1) || value > 1024) return fail("group_first (-1 = automatic) / group_tail / group_tail_panels must be in [0, 1024]");
        (!strcmp(name, "group_first") ? h->group_first : !strcmp(name, "group_tail") ? h->group_tail : h->group_tail_panels) = (int)value;
        return 0;
    }
    if (!strcmp(name, "assemble_queue")) {   // see ck_handle::assemble_queue
        if (value < -1 || value > 4096) return fail("assemble_queue must be in [-1, 4096]");
        h->assemble_queue = (int)value;
        return 0;
    }
    if (!strcmp(name, "local_left")) {   // see ck_handle::local_left
        if (value < 0 || value > 1) return fail("local_left must be 0 or 1");
        h->local_left = (int)value;
        return 0;
    }
    if (!strcmp(name, "tall_split")) {
        if (value < 0 || value > 2) return fail("tall_split must be 0, 1 or 2");
        h->tall_split = (int)value;
        return 0;
    }
    if (!strcmp(name, "solve_la")) {
        if (value < -1 || value > 1) return fail("solve_la must be -1, 0 or 1");
        h->solve_la = (int)value;
        return 0;
    }
    if (!strcmp(name, "tall_b2_stream")) {
        h->tall_b2_stream = value != 0;
        return 0;
    }
    if (!strcmp(name, "tall_thin")) {
        if (value < 0 || value > 1) return fail("tall_thin must be 0 or 1");
        h->tall_thin = (int)value;
        return 0;
    }
    if (!strcmp(name, "tall_split_rows")) {
        if (value < 0) return fail("tall_split_rows must be >= 0");
        h->tall_split_rows = (int)std::min<int64_t>(value, 1 << 30);
        return 0;
    }
    if (!strcmp(name, "tall_sweep")) {   // see ck_handle::tall_sweep
        if (value < 0 || value > 1) return fail("tall_sweep must be 0 or 1");
        h->tall_sweep = (int)value;
        return 0;
    }
    if (!strcmp(name, "fused_la")) {
        if (value < -1 || value > 1) return fail("fused_la must be -1 (automatic), 0 or 1");
        h->fused_la = (int)value;
        return 0;
    }
    if (!strcmp(name, "fused_group")) {
        if (value < 0 || value > 16) return fail("fused_group must be in [0, 16]");
        h->fused_group = (int)value;
        return 0;
    }
    if (!strcmp(name, "lookahead")) {   // see ck_handle::lookahead
        if (value < -1 || value > 1) return fail("lookahead must be -1 (automatic), 0 or 1");
        h->lookahead = (int)value;
        return 0;
    }
    if (!strcmp(name, "local_slab_mb")) {
        if (value < 0) return fail("local_slab_mb must be >= 0");
        h->local_slab_mb = value;
        return 0;
    }
    if (!strcmp(name, "panel_fused")) {   // see ck_handle::panel_fused
        if (value < 0 || value > 31) return fail("panel_fused must be in [0, 31]");
        h->panel_fused = (int)value;
        return 0;
    }
    if (!strcmp(name, "local_group")) {
        if (value < 1 || value > CK_LT_NINV) return fail("local_group must be in [1, 8]");
        h->local_group = (int)value;
        return 0;
    }
    if (!strcmp(name, "local_tile_min")) {   // see ck_handle::local_tile_min
        if (value < 0) return fail("local_tile_min must be >= 0");
        h->local_tile_min = (int)std::min<int64_t>(value, 1 << 30);
        return 0;
    }
    if (!strcmp(name, "panel_group")) {   // panels per trailing update (1 = after every panel)
        if (value < 0 || value > 16) return fail("panel_group must be in [0, 16] (0 = automatic)");
        h->panel_group = (int)value;
        return 0;
    }
    if (!strcmp(name, "site_order")) {   // 0: caller's order | 1: Hilbert order (see ck_handle::site_order)
        if (h->layout_ready && (value != 0) != (h->site_order != 0)) {
            // same sizes, other order: the sites are laid out again on the next assemble (what ck_factor does on its own
            // to report a failing minor in the caller's numbering; the step-wise driver does it through this option)
            h->layout_ready = false;
            h->assembled = h->factored = false;
            h->aux_state = 0;
        }
        h->site_order = value != 0;
        return 0;
    }
    if (!strcmp(name, "recv_slots")) {   // see ck_handle::recv_slots; before the first assemble / ck_estimate_bytes
        if (value < 2 || value > 64) return fail("recv_slots must be in [2, 64]");
        if (h->layout_ready || !h->sig.empty()) return fail("recv_slots must be set before the panels are allocated");
        h->recv_slots = (int)value;
        return 0;
    }
    if (!strcmp(name, "exact_cov")) {   // 1: per-entry Bessel evaluation instead of the tables
        h->exact_cov = value != 0;
        return 0;
    }
    return fail(std::string("unknown option ") + name);
}

extern "C" int ck_table_info(ck_handle* h, int block, int* enabled, int* n_intervals, double* q_lo, double* q_hi,
                             double* max_rel_err) {
    CHKH(h);
    if (block < 0 || block > 2) return fail("block must be 0 (11), 1 (12) or 2 (22)");
    if (ensure_layout(h)) return -1;
    const CkTable& T = h->tab[block];
    if (enabled) *enabled = T.enabled;
    if (n_intervals) *n_intervals = T.n_int;
    if (q_lo) *q_lo = T.q_lo;
    if (q_hi) *q_hi = T.q_hi;
    if (max_rel_err) *max_rel_err = T.max_rel_err;
    return 0;
}

extern "C" int ck_table_fallbacks(ck_handle* h, int reset, int64_t* count) {
    CHKH(h);
    if (count) *count = h->fallback_total;
    if (reset) h->fallback_total = 0;
    return 0;
}

extern "C" int ck_timings(ck_handle* h, double* out, int n) {
    CHKH(h);
    for (int k = 0; k < n && k < 16; ++k) out[k] = h->t_ms[k];
    return 0;
}

extern "C" int ck_dev_gemm_nt(ck_handle* h, double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                              int64_t ldb, int64_t M, int64_t N, int64_t K, int lower) {
    CHKH(h);
    if (M % CK_BM || N % 64 || K % 16) return fail("ck_dev_gemm_nt: M % 256, N % 64, K % 16 must be 0");
    ck_launch_gemm_nt(h->stream, C, ldc, A, lda, B, ldb, M, N, K, lower, 0, 1, 0, 0, 0);
    HIPCHK(hipGetLastError());
    return 0;
}
