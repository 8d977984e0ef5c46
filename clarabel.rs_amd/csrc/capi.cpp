// capi.cpp -- the C ABI of include/clarabel_hip.h.
//   chip_ldl_*  : DirectLDLSolver<f64>   (quasidef/mod.rs:14-26; behaviour of ldlsolvers/qdldl.rs)
//   chip_kkt_*  : KKTSolver<f64> as implemented by DirectLDLKKTSolver
//                 (quasidef/directldlkktsolver.rs:18-405), device resident
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "engine.hpp"
#include "../../include/clarabel_hip_testing.h"

using namespace chip;

namespace {

thread_local std::string g_last;

int fail(int code, const std::string &msg) {
    set_error(msg);
    return code;
}

const i64 *as_i64(const uint64_t *p) { return reinterpret_cast<const i64 *>(p); }

void fill_info(const Engine &E, chip_info *info) {
    std::memset(info, 0, sizeof(*info));
    std::strncpy(info->name, "hip", sizeof(info->name) - 1);
    info->threads = 1;
    info->direct = 1;
    info->nnzA = E.nnzK;
    info->nnzL = E.nnzL;
    info->n = E.N;
    info->n_levels = E.tree_depth;
    info->amd_lnz = E.amd.lnz;
    info->amd_ndiv = E.amd.ndiv;
    info->amd_nmultsubs_ldl = E.amd.nmultsubs_ldl;
    info->regularize_count = E.last_regularize_count;
    info->positive_inertia = -1;
}

int count_positive(Engine &E, i64 *out) {
    std::vector<double> d((size_t)E.N);
    if (E.N) CHIP_HIP(hipMemcpy(d.data(), E.D, (size_t)E.N * sizeof(double), hipMemcpyDeviceToHost));
    i64 c = 0;
    for (double v : d) c += v > 0.0;
    *out = c;
    return CHIP_OK;
}

} // namespace

// ===========================================================================
struct chip_ldl {
    Engine E;
    std::vector<double> hK; // host mirror of the caller's K.nzval (update/scale/offset land here)
    bool dirty = true;
    double *d_b = nullptr, *d_x = nullptr, *d_y = nullptr;
    // ---- fast path of the strict drop-in (round 5) ----
    // registered index sets: the reference's update_values / scale_values / offset_values always come with one of a few
    // FIXED index vectors of the LDLDataMap (Hsblocks, diag_full, the sparse cones' u / v / D: datamaps.rs:350-362);
    // registered once, a set lives on the device as positions in the value store and an update moves 8 bytes per entry
    struct IndexSet {
        int *pos = nullptr;      // device: position of entry t in the device's value order
        int8_t *signs = nullptr; // device (or nullptr): the signs of offset_values
        i64 k = 0;
    };
    std::vector<IndexSet> sets;
    double *d_vals = nullptr; // staging of one update's values
    i64 d_vals_cap = 0;
    // once an update went to the device copy directly, THAT copy is the caller's K.nzval (the host mirror is stale and
    // chip_ldl_refactor no longer uploads it); chip_ldl_set_values makes the host mirror current again
    bool dev_values = false;
    // device-resident refinement (chip_ldl_solve_refined): permuted right-hand side, iterate, residual, correction
    double *r_bp = nullptr, *r_x = nullptr, *r_e = nullptr, *r_w = nullptr;
    std::vector<void *> pinned; // caller buffers registered with chip_ldl_pin_buffer
    ~chip_ldl() {
        for (void *p : pinned) (void)hipHostUnregister(p);
    }
};

struct chip_kkt {
    KktLayout K;
    Engine E;
    // int32 device copies of the LDLDataMap pieces the kernels index with
    int *mapHs = nullptr, *diag_full = nullptr, *mapP = nullptr, *mapA = nullptr;
    // cones
    int nn_count = 0, zero_count = 0;
    int *nn_rows = nullptr, *nn_hsidx = nullptr, *zero_rows = nullptr;
    dev::SocView soc{};
    dev::Ns3View ns3{};      // Exponential / Power cones
    dev::GpwView gpw{};      // generalised power cones
    std::vector<int> gpw_cone_index, gpw_state_off, gpw_dim1; // host copies (alpha setter)
    dev::PsdView psd{};      // PSD triangle cones (any matrix side)
    bool has_hostHs = false; // cones whose Hs must come from the host (none of the SupportedConeT kinds any more)
    double *d_s = nullptr, *d_z = nullptr, *d_w = nullptr, *d_lam = nullptr;
    double *d_rhs = nullptr, *d_lhs = nullptr; // n+m staging
    double *bp = nullptr, *x = nullptr, *e = nullptr, *dx = nullptr; // N, permuted numbering
    double *bp2 = nullptr, *x2 = nullptr, *e2 = nullptr, *dx2 = nullptr; // ... of the second solve of a pair (chip_kkt_solve2_dev_enqueue)
    int last_ir2 = 0;
    double *d_tmp = nullptr;                                         // max(N, nHs, nnzP, nnzA) staging
    size_t tmp_len = 0;
    int last_ir = 0;
    double last_eps = 0;
    bool scaling_pending_check = false;
    bool psd_rows_all_blocks = false; // every dense diagonal block of the top is a PSD cone's Hs block (k_psd_write_hs_rows)
    int scaling_gen = 0; // generation of the last update_scaling: a failing cone writes it into mailbox.soc_fail
    bool x_holds_b = false; // x was initialised with the rhs by setrhs (skips a D2D copy)
    double static_diag_max = 0.0; // max |P_ii|: the diagonal entries of K that no cone kernel writes
    // fused solve path (Engine::ir_fused): setrhs only notes the caller's device buffers, the solve kernel
    // permutes them in; results of enqueued solves / updates not yet collected
    const double *rhs_x = nullptr, *rhs_z = nullptr;
    bool rhs_deferred = false;
    int *ir_run_ptr = nullptr, *ir_runs = nullptr; // run-length form of the permutation inside the bundles (dev::IrView)
    // k_bundle_irs takes this handle's fused solves (checked once at creation: every bundle qualifies, the grid is
    // co-resident).  That kernel does NOT leave the permuted right-hand side in bp: bp_stale says so, and a solve() that
    // follows without a new setrhs() reads the noted vectors (rhs_x / rhs_z) again -- they stay borrowed until the next
    // setrhs (include/clarabel_hip.h)
    bool ir_sf = false, bp_stale = false;
    int pend_update = 0;             // 1: an update has been enqueued and its verdict not read; 2: read, kept
    int pend_update_ok = 1;
    std::vector<int> pend_slots;     // ring slots of the solves enqueued since the last collect
    // arguments of the fused launches by ring slot: a launch whose grid barrier timed out (its workgroups were not
    // co-resident: another long-running kernel held their slots) is repeated on the one-kernel-per-phase path
    struct FusedArgs {
        const double *rx, *rz;
        double *lx, *lz;
    } fused_args[Engine::IR_RING] = {};
    int fused_fallbacks = 0;
    #ifdef CHIP_TESTING
    bool ir_test_drop = switches().ir_test_drop; // (tests, read when the handle is created)
#else
    static constexpr bool ir_test_drop = false;
#endif
    int world = 1;               // ranks sharing the problem (chip_kkt_attach_comm)
    // the exchange of the sharded path runs on the communicator's stream, overlapped with what this handle enqueues next:
    // the cone update and the factorisation may run beside its kernels, a PERSISTENT solve launch may not -- its grid fills
    // every register slot of the chip, and a foreign wave that is resident while it starts leaves the register file
    // fragmented, so that some of its workgroups can never become resident (measured: a one-workgroup kernel of 60 us next
    // to k_gstep_solve made every solve run into its wait budget).  The next fused launch waits for this event on the device.
    double *d_partial = nullptr; // per-block partial minima / sums of the cone reductions
    int partial_cap = 0;
    std::vector<double> h_partial;
};

extern "C" {

void chip_settings_default(chip_settings *s) {
    std::memset(s, 0, sizeof(*s));
    s->static_regularization_enable = 1;
    s->static_regularization_constant = 1e-8;
    s->static_regularization_proportional = 2.220446049250313e-16 * 2.220446049250313e-16;
    s->dynamic_regularization_enable = 1;
    s->dynamic_regularization_eps = 1e-13;
    s->dynamic_regularization_delta = 2e-7;
    s->iterative_refinement_enable = 1;
    s->iterative_refinement_reltol = 1e-13;
    s->iterative_refinement_abstol = 1e-12;
    s->iterative_refinement_max_iter = 10;
    s->iterative_refinement_stop_ratio = 5.0;
    s->device = -1;
    s->amd_dense_scale = 1.5;
    s->use_graph = 0;
    s->linesearch_backtrack_step = 0.8;
    s->min_terminate_step_length = 1e-4;
}

int32_t chip_auto_select(double lnz, double n_div, double n_mult_subs_ldl) {
    const double flops = n_div + n_mult_subs_ldl, thresh = 40.0; // auto.rs:72-79
    return (flops / lnz) < thresh ? 0 : 1;
}

const char *chip_last_error(void) {
    g_last = get_error();
    return g_last.c_str();
}

int32_t chip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t chip_amd_order(int64_t n, const uint64_t *colptr, const uint64_t *rowval, double dense_scale,
                       uint64_t *perm, uint64_t *iperm, double *info3) {
    if (n < 0 || !colptr || !perm) return fail(CHIP_ERR_ARG, "chip_amd_order: bad argument");
    for (i64 c = 0; c < n; c++)
        for (uint64_t p = colptr[c]; p < colptr[c + 1]; p++)
            if (rowval[p] > (uint64_t)c) return fail(CHIP_ERR_NOT_TRIU, "matrix is not upper triangular");
    std::vector<i64> p;
    AmdInfo info;
    int rc = amd_order(n, as_i64(colptr), as_i64(rowval), dense_scale, p, &info);
    if (rc) return fail(CHIP_ERR_ARG, "amd_order failed");
    for (i64 k = 0; k < n; k++) {
        perm[k] = (uint64_t)p[k];
        if (iperm) iperm[p[k]] = (uint64_t)k;
    }
    if (info3) {
        info3[0] = info.lnz;
        info3[1] = info.ndiv;
        info3[2] = info.nmultsubs_ldl;
    }
    return CHIP_OK;
}

// ---------------------------------------------------------------------------
// workgroups the device keeps resident for the fused solve kernel (4 x 256 threads per CU): a forest with fewer
// bundles than that is cut finer by the analysis (grouped fold).  Host-only handles: an MI355X's 256 CUs.
static int target_workgroups(const chip_settings &st) {
    if (switches().has_target_wg) return switches().target_wg; // (tests; 0 = never refine)
    if (st.device == CHIP_DEVICE_HOST_ONLY) return 1024;
    int dev = st.device, cus = 0;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return 1024;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        return 1024;
    }
    return 4 * cus;
}

// L1
// ---------------------------------------------------------------------------
int32_t chip_ldl_create(chip_ldl **out, int64_t n, const uint64_t *colptr, const uint64_t *rowval,
                        const double *nzval, const int8_t *dsigns, const uint64_t *perm_or_null,
                        const chip_settings *settings) {
    if (!out || n < 0 || !colptr || !rowval || !nzval) return fail(CHIP_ERR_ARG, "chip_ldl_create: bad argument");
    *out = nullptr;
    switches_reload(); // the CHIP_* diagnostic switches are read from the environment here, never in a launch loop
    chip_settings st;
    if (settings) st = *settings;
    else chip_settings_default(&st);
    std::vector<i64> perm0;
    if (perm_or_null) perm0.assign(as_i64(perm_or_null), as_i64(perm_or_null) + n);
    Symbolic S;
    // (target_wg = 0: L1 handles never take the fused launches, so a forest is not cut finer for them -- the grouped
    // fold's extra top nodes would only add level-scheduled launches to every solve)
    int rc = analyse(n, as_i64(colptr), as_i64(rowval), dsigns, perm0, st.amd_dense_scale, S, 0);
    if (rc) return rc;
    std::unique_ptr<chip_ldl> h(new chip_ldl());
    h->hK.assign(nzval, nzval + S.nnzK);
    if (st.device == CHIP_DEVICE_HOST_ONLY) {
        h->E.init_host_only(S, st);
        *out = h.release();
        return CHIP_OK;
    }
    rc = h->E.init(S, st);
    if (rc) return rc;
    h->dirty = true;
    if ((rc = h->E.alloc(&h->d_b, (size_t)n))) return rc;
    if ((rc = h->E.alloc(&h->d_x, (size_t)n))) return rc;
    if ((rc = h->E.alloc(&h->d_y, (size_t)n))) return rc;
    *out = h.release();
    return CHIP_OK;
}

void chip_ldl_destroy(chip_ldl *h) { delete h; }

static int ldl_plain_on_device(chip_ldl *h, int kind, const uint64_t *index, const double *values, double scalar,
                               const int8_t *signs, int64_t k);
int32_t chip_ldl_update_values(chip_ldl *h, const uint64_t *index, const double *values, int64_t k) {
    if (!h) return CHIP_ERR_ARG;
    if (h->dev_values) return ldl_plain_on_device(h, 0, index, values, 0.0, nullptr, k);
    for (i64 i = 0; i < k; i++) {
        if (index[i] >= (uint64_t)h->E.nnzK) return fail(CHIP_ERR_ARG, "update_values: index out of range");
        h->hK[index[i]] = values[i];
    }
    h->dirty = true;
    return CHIP_OK;
}
int32_t chip_ldl_scale_values(chip_ldl *h, const uint64_t *index, double scale, int64_t k) {
    if (!h) return CHIP_ERR_ARG;
    if (h->dev_values) return ldl_plain_on_device(h, 1, index, nullptr, scale, nullptr, k);
    for (i64 i = 0; i < k; i++) {
        if (index[i] >= (uint64_t)h->E.nnzK) return fail(CHIP_ERR_ARG, "scale_values: index out of range");
        h->hK[index[i]] *= scale;
    }
    h->dirty = true;
    return CHIP_OK;
}
int32_t chip_ldl_offset_values(chip_ldl *h, const uint64_t *index, double offset, const int8_t *signs,
                               int64_t k) {
    if (!h) return CHIP_ERR_ARG;
    if (h->dev_values) return ldl_plain_on_device(h, 2, index, nullptr, offset, signs, k);
    for (i64 i = 0; i < k; i++) {
        if (index[i] >= (uint64_t)h->E.nnzK) return fail(CHIP_ERR_ARG, "offset_values: index out of range");
        if (signs[i] > 0) h->hK[index[i]] += offset;
        else if (signs[i] < 0) h->hK[index[i]] -= offset;
    }
    h->dirty = true;
    return CHIP_OK;
}
int32_t chip_ldl_set_values(chip_ldl *h, const double *kkt_nzval) {
    if (!h || !kkt_nzval) return CHIP_ERR_ARG;
    std::memcpy(h->hK.data(), kkt_nzval, (size_t)h->E.nnzK * sizeof(double));
    h->dirty = true;
    h->dev_values = false; // (the host mirror is the caller's K.nzval again; the next refactor uploads it)
    return CHIP_OK;
}
extern "C++" {
namespace chip {
int kkt_device(const ::chip_kkt *h) { return h->E.device; }
hipStream_t kkt_stream(::chip_kkt *h) { return h->E.host_only ? nullptr : h->E.stream; }
void kkt_set_world(::chip_kkt *h, int world) { h->world = world; }
// The exchange's completion as an event the HANDLE owns (recorded here on the communicator's stream): a communicator
// destroyed before the handle's next persistent launch takes nothing the handle waits on with it.
int kkt_note_exchange(::chip_kkt *h, hipStream_t comm_stream) {
    Engine &E = h->E;
    if (!E.exch_event && hipEventCreateWithFlags(&E.exch_event, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        E.exch_event = nullptr;
        return CHIP_ERR_HIP;
    }
    if (hipEventRecord(E.exch_event, comm_stream) != hipSuccess) {
        (void)hipGetLastError();
        return CHIP_ERR_HIP;
    }
    E.exch_pending = true;
    return CHIP_OK;
}
bool kkt_host_only(const ::chip_kkt *h) { return h->E.host_only; }
} // namespace chip
}

#define NEED_DEVICE(E) \
    if ((E).host_only) return fail(CHIP_ERR_NO_DEVICE, "host-only handle: no numeric work without a GPU")

int32_t chip_ldl_get_symbolic(const chip_ldl *h, uint64_t *etree, uint64_t *Lp, uint64_t *Li,
                              uint64_t *lvlptr) {
    if (!h) return CHIP_ERR_ARG;
    return h->E.get_symbolic(etree, Lp, Li, lvlptr);
}

int32_t chip_ldl_refactor(chip_ldl *h) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    if (h->dirty && !h->dev_values && E.nnzK) {
        int rc = E.upload_values(h->hK.data());
        if (rc) return rc;
    }
    h->dirty = false;
    return E.refactor(false, nullptr);
}
int32_t chip_ldl_solve_dev(chip_ldl *h, double *x_dev, const double *b_dev) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    if (!E.factored) return fail(CHIP_ERR_NOT_FACTORED, "solve() before the first refactor()");
    CHIP_HIP(hipSetDevice(E.device));
    dev::permute_in(E.stream, h->d_y, b_dev, E.perm, E.N);
    E.enqueue_solve_inplace(h->d_y);
    dev::permute_out(E.stream, x_dev, h->d_y, E.perm, E.N);
    CHIP_HIP(hipGetLastError());
    return CHIP_OK;
}
int32_t chip_ldl_solve(chip_ldl *h, double *x, const double *b) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    if (!E.factored) return fail(CHIP_ERR_NOT_FACTORED, "solve() before the first refactor()");
    CHIP_HIP(hipSetDevice(E.device));
    const size_t bytes = (size_t)E.N * sizeof(double);
    if (E.N) CHIP_HIP(hipMemcpyAsync(h->d_b, b, bytes, hipMemcpyHostToDevice, E.stream));
    int rc = chip_ldl_solve_dev(h, h->d_x, h->d_b);
    if (rc) return rc;
    if (E.N) CHIP_HIP(hipMemcpyAsync(x, h->d_x, bytes, hipMemcpyDeviceToHost, E.stream));
    CHIP_HIP(hipStreamSynchronize(E.stream));
    return CHIP_OK;
}
int32_t chip_ldl_info(const chip_ldl *h, chip_info *info) {
    if (!h || !info) return CHIP_ERR_ARG;
    fill_info(h->E, info);
    if (h->E.factored && !h->E.host_only) {
        i64 c = 0;
        int rc = count_positive(const_cast<Engine &>(h->E), &c);
        if (rc) return rc;
        info->positive_inertia = c;
    }
    return CHIP_OK;
}
int32_t chip_ldl_get_perm(const chip_ldl *h, uint64_t *perm) {
    if (!h || !perm) return CHIP_ERR_ARG;
    for (int i = 0; i < h->E.N; i++) perm[i] = (uint64_t)h->E.h_perm[i];
    return CHIP_OK;
}
int32_t chip_ldl_get_factors(chip_ldl *h, uint64_t *Lp, uint64_t *Li, double *Lx, double *D, double *Dinv) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    CHIP_HIP(hipStreamSynchronize(E.stream));
    const size_t n = (size_t)E.N, nl = (size_t)E.nnzL;
    std::vector<int> tmp;
    if (Lp) {
        tmp.resize(n + 1);
        CHIP_HIP(hipMemcpy(tmp.data(), E.Lp, (n + 1) * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t i = 0; i <= n; i++) Lp[i] = (uint64_t)tmp[i];
    }
    if (Li && nl) {
        tmp.resize(nl);
        CHIP_HIP(hipMemcpy(tmp.data(), E.Li, nl * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nl; i++) Li[i] = (uint64_t)tmp[i];
    }
    if (Lx && nl) CHIP_HIP(hipMemcpy(Lx, E.Lx, nl * sizeof(double), hipMemcpyDeviceToHost));
    if (D && n) CHIP_HIP(hipMemcpy(D, E.D, n * sizeof(double), hipMemcpyDeviceToHost));
    if (Dinv && n) CHIP_HIP(hipMemcpy(Dinv, E.Dinv, n * sizeof(double), hipMemcpyDeviceToHost));
    return CHIP_OK;
}

// ---------------------------------------------------------------------------
// L2
// ---------------------------------------------------------------------------
static std::vector<int> narrow_plain(const std::vector<i64> &v, size_t n) {
    std::vector<int> o(n);
    for (size_t i = 0; i < n; i++) o[i] = (int)v[i];
    return o;
}

int32_t chip_kkt_create(chip_kkt **out, int64_t n, int64_t m, const uint64_t *Pcolptr,
                        const uint64_t *Prowval, const double *Pnzval, const uint64_t *Acolptr,
                        const uint64_t *Arowval, const double *Anzval, int64_t ncones,
                        const int32_t *cone_tags, const int64_t *cone_dims, const int64_t *cone_dims2,
                        const double *cone_alphas_or_null, const chip_settings *settings,
                        const uint64_t *perm_or_null) {
    if (!out || n < 0 || m < 0 || !Pcolptr || !Acolptr) return fail(CHIP_ERR_ARG, "chip_kkt_create: bad argument");
    *out = nullptr;
    switches_reload(); // the CHIP_* diagnostic switches are read from the environment here, never in a launch loop
    chip_settings st;
    if (settings) st = *settings;
    else chip_settings_default(&st);
    std::unique_ptr<chip_kkt> h(new chip_kkt());
    KktLayout &K = h->K;
    i64 mm = 0;
    PhaseClock clk;
    clk.tag = "kkt_create";
    int rc = build_cone_specs(ncones, cone_tags, cone_dims, cone_dims2, K.cones, mm, K.p, K.nHs);
    if (rc) return CHIP_ERR_ARG;
    if (mm != m) return fail(CHIP_ERR_DIM, "cone dimensions do not add up to m");
    rc = assemble_kkt_triu(n, m, as_i64(Pcolptr), as_i64(Prowval), Pnzval, as_i64(Acolptr), as_i64(Arowval),
                           Anzval, K);
    if (rc) return rc;
    clk("KKT assembly");
    std::vector<i64> perm0;
    if (perm_or_null) perm0.assign(as_i64(perm_or_null), as_i64(perm_or_null) + K.N);
    Symbolic S;
    // dense cone blocks (the Hs block of a cone that is not diagonal: its rows are a clique of K) enter the ordering as
    // one weighted node each -- worth it from a few dozen rows per block (PSD cones; not the 3 x 3 / 4 x 4 blocks)
    std::vector<i32> clique_of;
    {
        i32 g = 0;
        for (const ConeSpec &c : K.cones)
            if (!c.hs_diag && c.numel >= 16) {
                if (clique_of.empty()) clique_of.assign((size_t)K.N, -1);
                for (i64 r = 0; r < c.numel; r++) clique_of[(size_t)(n + c.start + r)] = g;
                g++;
            }
    }
    rc = analyse(K.N, K.colptr.data(), K.rowval.data(), K.dsigns.data(), perm0, st.amd_dense_scale, S, target_workgroups(st),
                 clique_of.empty() ? nullptr : &clique_of);
    if (rc) return rc;
    clk("analysis (total)");
    Engine &E = h->E;
    if (st.device == CHIP_DEVICE_HOST_ONLY) {
        E.init_host_only(S, st);
        *out = h.release();
        return CHIP_OK;
    }
    rc = E.init(S, st);
    if (rc) return rc;
    clk("engine init (uploads)");
    if (K.nnz) { // the device keeps K.nzval in T order (host.hpp: Symbolic::k2v)
        rawvec<double> vx((size_t)K.nnz);
        run_threads(K.nnz >= (i64)1 << 22 ? host_threads() : 1, [&](int t, int TT) {
            for (i64 u = K.nnz * t / TT; u < K.nnz * (t + 1) / TT; u++) vx[(size_t)u] = K.nzval[(size_t)S.v2k[(size_t)u]];
        });
        CHIP_HIP(hipMemcpy(E.Kx, vx.data(), (size_t)K.nnz * sizeof(double), hipMemcpyHostToDevice));
    }
    // every LDLDataMap index the kernels use is translated ONCE into a position of that store
    const bigvec &k2v = S.k2v;
    auto narrow = [&k2v](const auto &v, size_t cnt) { // (by host threads when the map is long: config 5's mapHs has 1.6e8 entries)
        rawvec<int> o(cnt);
        run_threads(cnt >= (size_t)1 << 22 ? host_threads() : 1, [&](int t, int TT) {
            for (size_t i = cnt * (size_t)t / (size_t)TT; i < cnt * (size_t)(t + 1) / (size_t)TT; i++) o[i] = k2v[(size_t)v[i]];
        });
        return o;
    };
    {
        double mx = 0.0; // the part of diag K no cone kernel writes: diag P (kkt_assembly.rs:120-121)
        for (i64 i = 0; i < K.n; i++) {
            const double a = K.nzval[(size_t)K.diagP[(size_t)i]];
            if (a != a) mx = a;
            else if (mx == mx) mx = std::max(mx, std::fabs(a));
        }
        h->static_diag_max = mx;
    }

    // ---- index maps as int32 ------------------------------------------------
    if ((rc = E.upload(&h->mapHs, narrow(K.mapHs, (size_t)K.nHs), (size_t)K.nHs))) return rc;
    if ((rc = E.upload(&h->diag_full, narrow(K.diag_full, (size_t)K.N), (size_t)K.N))) return rc;
    const size_t nnzP = (size_t)Pcolptr[n], nnzA = (size_t)Acolptr[n];
    if ((rc = E.upload(&h->mapP, narrow(K.mapP, nnzP), nnzP))) return rc;
    if ((rc = E.upload(&h->mapA, narrow(K.mapA, nnzA), nnzA))) return rc;

    // ---- cone work lists ----------------------------------------------------
    std::vector<int> nn_rows, nn_hs, zero_rows, s_start, s_dim, s_hs, s_sidx, s_ptr, mapU, mapV, mapD;
    std::vector<int> n3_start, n3_hs, n3_tag, pd_start, pd_dim, pd_hs, pd_off;
    std::vector<int> gp_start, gp_d1, gp_d2, gp_hs, gp_off, gp_mapptr, gp_map, gp_mapD;
    i64 gp_state = 0;
    i64 pd_state = 0;
    int pd_max = 0;
    std::vector<double> n3_alpha;
    for (const ConeSpec &c : K.cones) {
        if (c.tag == CHIP_CONE_NONNEGATIVE) {
            for (i64 k = 0; k < c.numel; k++) {
                nn_rows.push_back((int)(c.start + k));
                nn_hs.push_back((int)(c.block_start + k));
            }
        } else if (c.tag == CHIP_CONE_ZERO) {
            for (i64 k = 0; k < c.numel; k++) zero_rows.push_back((int)(c.start + k));
        } else if (c.tag == CHIP_CONE_SECONDORDER) {
            s_start.push_back((int)c.start);
            s_dim.push_back((int)c.numel);
            s_hs.push_back((int)c.block_start);
            s_sidx.push_back((int)c.sparse_idx);
        } else if (c.tag == CHIP_CONE_EXPONENTIAL || c.tag == CHIP_CONE_POWER) {
            const size_t ci = (size_t)(&c - K.cones.data());
            n3_start.push_back((int)c.start);
            n3_hs.push_back((int)c.block_start);
            n3_tag.push_back((int)c.tag);
            const double al = cone_alphas_or_null ? cone_alphas_or_null[ci] : 0.5;
            if (c.tag == CHIP_CONE_POWER && !(al > 0.0 && al < 1.0))
                return fail(CHIP_ERR_ARG, "PowerConeT exponent must lie in (0,1)");
            n3_alpha.push_back(al);
        } else if (c.tag == CHIP_CONE_GENPOWER) {
            gp_start.push_back((int)c.start);
            gp_d1.push_back((int)c.dim);
            gp_d2.push_back((int)c.dim2);
            gp_hs.push_back((int)c.block_start);
            gp_off.push_back((int)gp_state);
            gp_state += 6 * c.dim + 4 * c.dim2 + 3;
            gp_mapptr.push_back((int)gp_map.size());
            const i64 sidx = c.sparse_idx;
            for (i64 k = 0; k < c.dim; k++) gp_map.push_back(k2v[(size_t)K.sp_q[K.sp_q_ptr[sidx] + k]]);
            for (i64 k = 0; k < c.dim2; k++) gp_map.push_back(k2v[(size_t)K.sp_r[K.sp_r_ptr[sidx] + k]]);
            for (i64 k = 0; k < c.numel; k++) gp_map.push_back(k2v[(size_t)K.sp_u[K.sp_ptr[sidx] + k]]);
            for (int k = 0; k < 3; k++) gp_mapD.push_back(k2v[(size_t)K.sp_D[3 * sidx + k]]);
            h->gpw_cone_index.push_back((int)(&c - K.cones.data()));
        } else if (c.tag == CHIP_CONE_PSDTRIANGLE) {
            pd_start.push_back((int)c.start);
            pd_dim.push_back((int)c.dim);
            pd_hs.push_back((int)c.block_start);
            pd_off.push_back((int)pd_state);
            pd_state += 3 * c.dim * c.dim + 2 * c.dim; // B | lambda | lambda^-1/2 | R | Rinv
            pd_max = std::max<int>(pd_max, (int)c.dim);
        } else {
            h->has_hostHs = true;
        }
    }
    {
        dev::GpwView &gv = h->gpw;
        gv.ncones = (int)gp_start.size();
        int *g1, *g2, *g3, *g4, *g5, *g6, *g7, *g8;
        double *gs;
        if ((rc = E.upload(&g1, gp_start, gp_start.size()))) return rc;
        if ((rc = E.upload(&g2, gp_d1, gp_d1.size()))) return rc;
        if ((rc = E.upload(&g3, gp_d2, gp_d2.size()))) return rc;
        if ((rc = E.upload(&g4, gp_hs, gp_hs.size()))) return rc;
        if ((rc = E.upload(&g5, gp_off, gp_off.size()))) return rc;
        if ((rc = E.upload(&g6, gp_mapptr, gp_mapptr.size()))) return rc;
        if ((rc = E.upload(&g7, gp_map, gp_map.size()))) return rc;
        if ((rc = E.upload(&g8, gp_mapD, gp_mapD.size()))) return rc;
        // state: alpha defaults to 1/dim1 with psi = 1 / sum alpha^2 until chip_kkt_set_genpow_alpha
        std::vector<double> st0((size_t)(gp_state ? gp_state : 1), 0.0);
        for (size_t c = 0; c < gp_start.size(); c++) {
            double *st = st0.data() + gp_off[c];
            for (int k = 0; k < gp_d1[c]; k++) st[k] = 1.0 / gp_d1[c];
            st[6 * gp_d1[c] + 4 * gp_d2[c] + 1] = 1.0;               // mu
            st[6 * gp_d1[c] + 4 * gp_d2[c] + 2] = (double)gp_d1[c]; // psi
        }
        if ((rc = E.upload(&gs, st0, st0.size()))) return rc;
        gv.start = g1;
        gv.dim1 = g2;
        gv.dim2 = g3;
        gv.hs_start = g4;
        gv.state_off = g5;
        gv.map_ptr = g6;
        gv.mapQRP = g7;
        gv.mapD = g8;
        gv.mapHs = h->mapHs;
        gv.state = gs;
        h->gpw_state_off = gp_off;
        h->gpw_dim1 = gp_d1;
    }
    {
        dev::PsdView &pv = h->psd;
        pv.ncones = (int)pd_start.size();
        pv.maxdim = pd_max;
        int *q1, *q2, *q3, *q4;
        double *qs;
        if ((rc = E.upload(&q1, pd_start, pd_start.size()))) return rc;
        if ((rc = E.upload(&q2, pd_dim, pd_dim.size()))) return rc;
        if ((rc = E.upload(&q3, pd_hs, pd_hs.size()))) return rc;
        if ((rc = E.upload(&q4, pd_off, pd_off.size()))) return rc;
        if ((rc = E.alloc(&qs, (size_t)(pd_state ? pd_state : 1)))) return rc;
        pv.scratch = nullptr;
        pv.scratch_stride = 0;
        if (pd_max > 64) { // beyond the LDS budget of three / four n x n matrices: work matrices in HBM
            pv.scratch_stride = 4LL * pd_max * pd_max + 4LL * pd_max + 16;
            if ((rc = E.alloc(&pv.scratch, (size_t)pv.scratch_stride * pd_start.size()))) return rc;
        }
        pv.start = q1;
        pv.dim = q2;
        pv.hs_start = q3;
        pv.state_off = q4;
        pv.state = qs;
        pv.mapHs = h->mapHs;
        pv.fail = &E.mb_dev->soc_fail;
        // every cone's Hs block one of the dense diagonal blocks of the top?  Then Hs is written row by row of the
        // device's value order (k_psd_write_hs_rows) instead of through mapHs
        if (pv.ncones > 0 && pd_max <= 64 && E.dblk.nblk > 0) {
            const int nblk = E.dblk.nblk;
            std::vector<i32> blk_cone((size_t)nblk, -1), row_ij(E.h_dblk_node.size(), 0);
            std::vector<char> covered(pd_start.size(), 0);
            size_t nc = 0;
            for (int b = 0, o = 0; b < nblk; o += E.h_dblk_m[(size_t)b], b++) {
                const i32 mb = E.h_dblk_m[(size_t)b];
                const i64 z0 = (i64)E.h_perm[(size_t)E.h_dblk_node[(size_t)o]] - n;
                if (z0 < 0) continue;
                // the cone that holds z0 (cones ascend by start)
                const size_t c = (size_t)(std::upper_bound(pd_start.begin(), pd_start.end(), (i32)z0) - pd_start.begin());
                if (c == 0) continue;
                const i32 st = pd_start[c - 1], nd = pd_dim[c - 1], numel = nd * (nd + 1) / 2;
                if (mb != numel || covered[c - 1]) continue;
                bool ok = true;
                for (i32 a = 0; a < mb && ok; a++) {
                    const i64 t = (i64)E.h_perm[(size_t)E.h_dblk_node[(size_t)(o + a)]] - n - st;
                    if (t < 0 || t >= numel) {
                        ok = false;
                        break;
                    }
                    i32 j = (i32)((std::sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
                    while ((i64)j * (j + 1) / 2 > t) j--;
                    while ((i64)(j + 1) * (j + 2) / 2 <= t) j++;
                    row_ij[(size_t)(o + a)] = (i32)(t - (i64)j * (j + 1) / 2) | (j << 16);
                }
                if (!ok) continue;
                blk_cone[(size_t)b] = (i32)(c - 1);
                covered[c - 1] = 1;
                nc++;
            }
            if (nc == pd_start.size()) {
                // (every dense block belongs to a cone: the blocks may be written into L directly, Engine::hs_direct_begin)
                h->psd_rows_all_blocks = std::all_of(blk_cone.begin(), blk_cone.end(), [](i32 c) { return c >= 0; });
                int *r1, *r2;
                if ((rc = E.upload(&r1, blk_cone, blk_cone.size()))) return rc;
                if ((rc = E.upload(&r2, row_ij, row_ij.size()))) return rc;
                pv.rows_nblk = nblk;
                pv.blk_cone = r1;
                pv.row_ij = r2;
                pv.blk_m = E.dblk.m;
                pv.blk_rowbase = E.dblk.rowbase;
                pv.blk_start = E.dblk.start;
            }
        }
    }
    {
        dev::Ns3View &nv = h->ns3;
        nv.ncones = (int)n3_start.size();
        int *p1, *p2, *p3;
        double *pa, *pst;
        if ((rc = E.upload(&p1, n3_start, n3_start.size()))) return rc;
        if ((rc = E.upload(&p2, n3_hs, n3_hs.size()))) return rc;
        if ((rc = E.upload(&p3, n3_tag, n3_tag.size()))) return rc;
        if ((rc = E.upload(&pa, n3_alpha, n3_alpha.size()))) return rc;
        if ((rc = E.alloc(&pst, (size_t)(nv.ncones ? nv.ncones : 1) * 18))) return rc;
        CHIP_HIP(hipMemset(pst, 0, (size_t)(nv.ncones ? nv.ncones : 1) * 18 * sizeof(double)));
        nv.start = p1;
        nv.hs_start = p2;
        nv.tag = p3;
        nv.alpha = pa;
        nv.state = pst;
        nv.mapHs = h->mapHs;
    }
    const size_t nsp = K.sp_ptr.size() ? K.sp_ptr.size() - 1 : 0;
    s_ptr = narrow_plain(K.sp_ptr, nsp + 1);
    {
        const auto nu = narrow(K.sp_u, (size_t)(nsp ? K.sp_ptr[nsp] : 0)), nv_ = narrow(K.sp_v, (size_t)(nsp ? K.sp_ptr[nsp] : 0));
        mapU.assign(nu.begin(), nu.end());
        mapV.assign(nv_.begin(), nv_.end());
    }
    for (size_t s = 0; s < nsp; s++) {
        mapD.push_back(k2v[(size_t)K.sp_D[3 * s]]);
        mapD.push_back(k2v[(size_t)K.sp_D[3 * s + 1]]);
    }
    h->nn_count = (int)nn_rows.size();
    h->zero_count = (int)zero_rows.size();
    if ((rc = E.upload(&h->nn_rows, nn_rows, nn_rows.size()))) return rc;
    if ((rc = E.upload(&h->nn_hsidx, nn_hs, nn_hs.size()))) return rc;
    if ((rc = E.upload(&h->zero_rows, zero_rows, zero_rows.size()))) return rc;
    dev::SocView &sv = h->soc;
    sv.ncones = (int)s_start.size();
    int *p_start, *p_dim, *p_hs, *p_sidx, *p_ptr, *p_mu, *p_mv, *p_md;
    if ((rc = E.upload(&p_start, s_start, s_start.size()))) return rc;
    if ((rc = E.upload(&p_dim, s_dim, s_dim.size()))) return rc;
    if ((rc = E.upload(&p_hs, s_hs, s_hs.size()))) return rc;
    if ((rc = E.upload(&p_sidx, s_sidx, s_sidx.size()))) return rc;
    if ((rc = E.upload(&p_ptr, s_ptr, s_ptr.size()))) return rc;
    if ((rc = E.upload(&p_mu, mapU, mapU.size()))) return rc;
    if ((rc = E.upload(&p_mv, mapV, mapV.size()))) return rc;
    if ((rc = E.upload(&p_md, mapD, mapD.size()))) return rc;
    sv.start = p_start;
    sv.dim = p_dim;
    sv.hs_start = p_hs;
    sv.sparse_idx = p_sidx;
    sv.sp_ptr = p_ptr;
    sv.mapHs = h->mapHs;
    sv.mapU = p_mu;
    sv.mapV = p_mv;
    sv.mapD = p_md;
    if ((rc = E.alloc(&h->d_s, (size_t)m))) return rc;
    if ((rc = E.alloc(&h->d_z, (size_t)m))) return rc;
    if ((rc = E.alloc(&h->d_w, (size_t)m))) return rc;
    if ((rc = E.alloc(&h->d_lam, (size_t)m))) return rc;
    CHIP_HIP(hipMemset(h->d_w, 0, (size_t)(m ? m : 1) * sizeof(double)));
    CHIP_HIP(hipMemset(h->d_lam, 0, (size_t)(m ? m : 1) * sizeof(double)));
    sv.w = h->d_w;
    sv.lam = h->d_lam;
    double *st8;
    if ((rc = E.alloc(&st8, (size_t)(sv.ncones ? sv.ncones : 1) * 8))) return rc;
    CHIP_HIP(hipMemset(st8, 0, (size_t)(sv.ncones ? sv.ncones : 1) * 8 * sizeof(double)));
    sv.eta = st8;
    sv.d = st8;
    sv.fail = &E.mb_dev->soc_fail;
    // ---- vectors -------------------------------------------------------------
    if ((rc = E.alloc(&h->d_rhs, (size_t)(n + m)))) return rc;
    if ((rc = E.alloc(&h->d_lhs, (size_t)(n + m)))) return rc;
    if ((rc = E.alloc(&h->bp, (size_t)K.N))) return rc;
    if ((rc = E.alloc(&h->x, (size_t)K.N))) return rc;
    if ((rc = E.alloc(&h->e, (size_t)K.N))) return rc;
    if ((rc = E.alloc(&h->dx, (size_t)K.N))) return rc;
    h->tmp_len = std::max<size_t>({(size_t)K.N, (size_t)K.nHs, nnzP, nnzA, 1});
    if ((rc = E.alloc(&h->d_tmp, h->tmp_len))) return rc;
    CHIP_HIP(hipMemset(h->bp, 0, (size_t)(K.N ? K.N : 1) * sizeof(double)));
    if (E.ir_fused) {
        // the permutation inside every bundle as maximal ascending runs that stay inside one of the ranges
        // [0, n) (rhsx), [n, n + m) (rhsz), [n + m, N) (zeros); used when they are long on average
        std::vector<int> rp(1, 0), runs;
        const std::vector<i32> &pm = E.h_perm;
        const i64 n1 = K.n, n2 = K.n + K.m;
        auto range_of = [&](i64 o) { return o < n1 ? 0 : (o < n2 ? 1 : 2); };
        for (int b = 0; b < E.bundles.nb; b++) {
            const int s0 = S.bundle_ptr[(size_t)b], s1 = S.bundle_ptr[(size_t)b + 1];
            int t = s0;
            while (t < s1) {
                int e = t + 1;
                while (e < s1 && pm[(size_t)e] == pm[(size_t)e - 1] + 1 && range_of(pm[(size_t)e]) == range_of(pm[(size_t)t])) e++;
                runs.push_back(t - s0);
                runs.push_back(pm[(size_t)t]);
                runs.push_back(e - t);
                t = e;
            }
            rp.push_back((int)(runs.size() / 3));
        }
        if (!runs.empty() && (i64)(runs.size() / 3) * 64 <= (i64)E.NF) {
            if ((rc = E.upload(&h->ir_run_ptr, rp, rp.size()))) return rc;
            if ((rc = E.upload(&h->ir_runs, runs, runs.size()))) return rc;
            // (arrow systems only: a forest without top rows whose bundles qualify did not occur in any test problem -- block-diagonal
            // problems become grouped folds --, so that path of the kernel stays unreachable rather than untested)
            bool sf = E.gfold.ng == 0 && E.fold.k >= 1 && E.ir_tw == 256 && E.bundles.symv_split && E.bundles.nb == E.ir_grid &&
                      !switches().no_ir_sf && !switches().no_flat;
            for (int b = 0; b < E.bundles.nb && sf; b++) {
                const int s0 = S.bundle_ptr[(size_t)b], nloc = S.bundle_ptr[(size_t)b + 1] - s0;
                const int nleaf = S.blvl[(size_t)S.blvl_ptr[(size_t)b] + 1] - s0;
                const int nlev = S.blvl_ptr[(size_t)b + 1] - S.blvl_ptr[(size_t)b] - 1;
                sf = dev::irs_bundle_ok(nloc, nleaf, nlev, rp[(size_t)b + 1] - rp[(size_t)b]);
            }
            h->ir_sf = sf && dev::bundle_irs_capacity_ok(E.bundles);
        }
    }
    clk("maps, cone tables");
    *out = h.release();
    return CHIP_OK;
}

void chip_kkt_destroy(chip_kkt *h) { delete h; }

int32_t chip_kkt_dims(const chip_kkt *h, int64_t out[8]) {
    if (!h) return CHIP_ERR_ARG;
    out[0] = h->K.n;
    out[1] = h->K.m;
    out[2] = h->K.p;
    out[3] = h->K.N;
    out[4] = h->K.nnz;
    out[5] = h->K.nHs;
    out[6] = h->E.NF;
    out[7] = h->E.nnzU;
    return CHIP_OK;
}
// CompositeCone::degree (compositecone.rs:106-108): zerocone.rs:36-38 (0), nonnegativecone.rs:39-41 (dim),
// socone.rs:74-77 (1), psdtrianglecone.rs:77-79 (n), expcone.rs:55-57 / powcone.rs:48-50 (3),
// genpowcone.rs:84-86 (dim1 + 1)
int32_t chip_kkt_degree(const chip_kkt *h, int64_t *degree) {
    if (!h || !degree) return CHIP_ERR_ARG;
    i64 d = 0;
    for (const ConeSpec &c : h->K.cones) {
        switch (c.tag) {
        case CHIP_CONE_ZERO: break;
        case CHIP_CONE_NONNEGATIVE: d += c.dim; break;
        case CHIP_CONE_SECONDORDER: d += 1; break;
        case CHIP_CONE_PSDTRIANGLE: d += c.dim; break;
        case CHIP_CONE_EXPONENTIAL:
        case CHIP_CONE_POWER: d += 3; break;
        case CHIP_CONE_GENPOWER: d += c.dim + 1; break;
        default: break;
        }
    }
    *degree = d;
    return CHIP_OK;
}
// diagnostics: the chain supernodes chosen by the symbolic analysis (host.hpp: Symbolic::sn_*)
static int32_t get_supernodes(const Engine &E, int64_t *count, uint64_t *ptr, uint64_t *cols) {
    if (!count) return CHIP_ERR_ARG;
    *count = E.nsn > 0 ? E.nsn : 0;
    if (ptr)
        for (size_t i = 0; i < E.h_sn_ptr.size(); i++) ptr[i] = (uint64_t)E.h_sn_ptr[i];
    if (cols)
        for (size_t i = 0; i < E.h_sn_col.size(); i++) cols[i] = (uint64_t)E.h_sn_col[i];
    return CHIP_OK;
}
int32_t chip_kkt_get_supernodes(const chip_kkt *h, int64_t *count, uint64_t *ptr, uint64_t *cols) {
    return h ? get_supernodes(h->E, count, ptr, cols) : CHIP_ERR_ARG;
}
int32_t chip_ldl_get_supernodes(const chip_ldl *h, int64_t *count, uint64_t *ptr, uint64_t *cols) {
    return h ? get_supernodes(h->E, count, ptr, cols) : CHIP_ERR_ARG;
}
int32_t chip_kkt_get_matrix(const chip_kkt *h, uint64_t *colptr, uint64_t *rowval, double *nzval) {
    if (!h) return CHIP_ERR_ARG;
    const KktLayout &K = h->K;
    if (colptr)
        for (i64 i = 0; i <= K.N; i++) colptr[i] = (uint64_t)K.colptr[i];
    if (rowval)
        for (i64 i = 0; i < K.nnz; i++) rowval[i] = (uint64_t)K.rowval[i];
    if (nzval) std::memcpy(nzval, K.nzval.data(), (size_t)K.nnz * sizeof(double));
    return CHIP_OK;
}
int32_t chip_kkt_get_map(const chip_kkt *h, uint64_t *mapP, uint64_t *mapA, uint64_t *mapHs, uint64_t *diagP,
                         uint64_t *diag_full, int8_t *dsigns) {
    if (!h) return CHIP_ERR_ARG;
    const KktLayout &K = h->K;
    auto cp = [](uint64_t *dst, const auto &src, size_t n) {
        if (dst)
            for (size_t i = 0; i < n; i++) dst[i] = (uint64_t)src[i];
    };
    cp(mapP, K.mapP, K.mapP.size() - 1);
    cp(mapA, K.mapA, K.mapA.size() - 1);
    cp(mapHs, K.mapHs, (size_t)K.nHs);
    cp(diagP, K.diagP, (size_t)K.n);
    cp(diag_full, K.diag_full, (size_t)K.N);
    if (dsigns) std::memcpy(dsigns, K.dsigns.data(), (size_t)K.N);
    return CHIP_OK;
}

int32_t chip_kkt_update_scaling_dev(chip_kkt *h, const double *s_dev, const double *z_dev, double mu,
                                    int32_t strategy) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    h->scaling_gen += 1; // (no memset of the flag: a failure is recognised by its generation)
    h->soc.fail_gen = h->scaling_gen;
    h->psd.fail_gen = h->scaling_gen;
    dev::sym_update_scaling(E.stream, h->soc, h->nn_rows, h->nn_count, s_dev, z_dev, h->d_w, h->d_lam);
    dev::ns3_update_scaling(E.stream, h->ns3, s_dev, z_dev, mu, strategy);
    dev::gpw_update_scaling(E.stream, h->gpw, z_dev, mu);
    dev::psd_update_scaling(E.stream, h->psd, s_dev, z_dev);
    CHIP_HIP(hipGetLastError());
    h->scaling_pending_check = h->soc.ncones > 0 || h->psd.ncones > 0; // verdict folded into the next update()
    return 1;
}
int32_t chip_kkt_update_scaling(chip_kkt *h, const double *s, const double *z, double mu, int32_t strategy) {
    if (!h || !s || !z) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    const size_t bytes = (size_t)h->K.m * sizeof(double);
    if (bytes) {
        CHIP_HIP(hipMemcpyAsync(h->d_s, s, bytes, hipMemcpyHostToDevice, E.stream));
        CHIP_HIP(hipMemcpyAsync(h->d_z, z, bytes, hipMemcpyHostToDevice, E.stream));
    }
    int rc = chip_kkt_update_scaling_dev(h, h->d_s, h->d_z, mu, strategy);
    if (rc < 0) return rc;
    if (h->soc.ncones || h->psd.ncones) {
        rc = E.read_mailbox();
        if (rc) return rc;
        h->scaling_pending_check = false;
        return E.mb_host->soc_fail == h->scaling_gen ? 0 : 1;
    }
    CHIP_HIP(hipStreamSynchronize(E.stream));
    return 1;
}

static int update_enqueue(chip_kkt *h, const double *hsblocks_or_null) {
    Engine &E = h->E;
    const KktLayout &K = h->K;
    CHIP_HIP(hipSetDevice(E.device));
    if (h->has_hostHs) {
        if (!hsblocks_or_null)
            return fail(CHIP_ERR_ARG, "update: Hs blocks are required for PSD cones");
        for (const ConeSpec &c : K.cones) {
            if (c.tag <= CHIP_CONE_PSDTRIANGLE) continue; // (held on the device)
            CHIP_HIP(hipMemcpyAsync(h->d_tmp + c.block_start, hsblocks_or_null + c.block_start,
                                    (size_t)c.block_len * sizeof(double), hipMemcpyHostToDevice, E.stream));
            dev::scatter_values(E.stream, E.Kx, h->mapHs + c.block_start, h->d_tmp + c.block_start,
                                (int)c.block_len, -1.0);
        }
    }
    // static regulariser eps = c + prop * max|diag K| (directldlkktsolver.rs:324-329): with Zero /
    // Nonnegative / SecondOrder cones only, the kernels that write the diagonal entries leave their maxima in
    // slots (no pass over the N diagonal entries); other cone kinds take the explicit reduction
    const bool slot_eps = !(h->has_hostHs || h->ns3.ncones || h->gpw.ncones || h->psd.ncones);
    unsigned long long *dslots = slot_eps && E.st.static_regularization_enable ? E.diag_slots() : nullptr;
    dev::sym_write_kkt(E.stream, h->soc, h->nn_rows, h->nn_hsidx, h->nn_count, h->d_w, h->mapHs, E.Kx, dslots);
    dev::ns3_write_hs(E.stream, h->ns3, E.Kx);
    dev::gpw_write_kkt(E.stream, h->gpw, E.Kx);
    // PSD blocks that are dense diagonal blocks of the top, contiguous in L too: written into K AND L by the one kernel
    const bool direct = h->psd_rows_all_blocks && dev::psd_write_hs_rows_active(h->psd) && E.hs_direct_begin();
    dev::psd_write_hs(E.stream, h->psd, E.Kx, direct ? E.Lx : nullptr, direct ? E.dblk_l0 : nullptr);
    return E.refactor_enqueue(h->E.st.static_regularization_enable != 0, slot_eps ? nullptr : h->diag_full,
                              h->static_diag_max);
}
// the verdict of the last enqueued update, from the mailbox (Engine::refactor_collect has copied it)
static int update_verdict(chip_kkt *h, int ok) {
    Engine &E = h->E;
    if (ok < 0) return ok;
    h->last_eps = E.st.static_regularization_enable ? E.mb_host->eps : 0.0;
    if (h->scaling_pending_check) {
        h->scaling_pending_check = false;
        if (E.mb_host->soc_fail == h->scaling_gen) return 0;
    }
    return ok;
}
int32_t chip_kkt_update(chip_kkt *h, const double *hsblocks_or_null) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    int rc = update_enqueue(h, hsblocks_or_null);
    if (rc) return rc;
    h->pend_update = 0;
    return update_verdict(h, E.refactor_collect());
}
int32_t chip_kkt_update_enqueue(chip_kkt *h, const double *hsblocks_or_null) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    int rc = update_enqueue(h, hsblocks_or_null);
    if (rc) return rc;
    h->pend_update = 1;
    E.factored = true; // provisionally: the verdict arrives with chip_kkt_collect
    return CHIP_OK;
}

// solver.rs:334-352 as ONE enqueue: cones.update_scaling(s, z, mu, strategy), then kktsystem.update's KKT part
// (get_Hs scatter, static regularisation, numeric refactor).  With Zero / Nonnegative / SecondOrder cones only, the
// scaling and the Hs / sparse-cone writes of a cone run in one launch (dev::sym_scale_write) and -- where the bundle
// factorisation can take over the preparation work (Engine::fast_prep_ok) -- nothing else is launched ahead of it.
int32_t chip_kkt_update_scaled_enqueue(chip_kkt *h, const double *s_dev, const double *z_dev, double mu, int32_t strategy,
                                       const double *hsblocks_or_null) {
    if (!h || !s_dev || !z_dev) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    const bool sym_only = !(h->has_hostHs || h->ns3.ncones || h->gpw.ncones || h->psd.ncones);
    if (!sym_only || !E.st.static_regularization_enable || switches().no_step_kernel) {
        // (PSD blocks that go into L directly: the clear of L's fill-in range starts now, beside the scaling kernels)
        if (h->psd_rows_all_blocks && !h->has_hostHs && dev::psd_write_hs_rows_active(h->psd)) E.hs_direct_prefill_async();
        int rc = chip_kkt_update_scaling_dev(h, s_dev, z_dev, mu, strategy);
        if (rc < 0) return rc;
        return chip_kkt_update_enqueue(h, hsblocks_or_null);
    }
    h->scaling_gen += 1; // (no memset of the flag: a failure is recognised by its generation)
    h->soc.fail_gen = h->scaling_gen;
    h->psd.fail_gen = h->scaling_gen;
    h->scaling_pending_check = h->soc.ncones > 0;
    // the cone launch clears the status words of the refactor -- when it launches anything at all (no SOC cones and no
    // Nonnegative rows: an empty grid) and when the factor kernel that reads the slots is the one that will run
    const bool cone_clears = E.fast_prep_ok && !switches().no_fast_prep && (h->soc.ncones + h->nn_count) > 0 &&
                             (E.gstep_factor_on || !switches().no_factor_flat);
    dev::sym_scale_write(E.stream, h->soc, h->nn_rows, h->nn_hsidx, h->nn_count, s_dev, z_dev, h->d_w, h->d_lam, h->mapHs, E.Kx,
                         E.diag_slots(), cone_clears ? E.mb_dev->status : nullptr);
    CHIP_HIP(hipGetLastError());
    int rc = E.refactor_enqueue(true, nullptr, h->static_diag_max, cone_clears);
    if (rc) return rc;
    h->pend_update = 1;
    E.factored = true; // provisionally: the verdict arrives with chip_kkt_collect
    return CHIP_OK;
}

int32_t chip_kkt_setrhs_dev(chip_kkt *h, const double *rhsx_dev, const double *rhsz_dev) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    if (E.ir_fused) { // the solve kernel reads (and permutes) the caller's buffers itself, see the header
        h->rhs_x = rhsx_dev;
        h->rhs_z = rhsz_dev;
        h->rhs_deferred = true;
        h->x_holds_b = true;
        return CHIP_OK;
    }
    int rc = E.zero_norm_sets();
    if (rc) return rc;
    // one pass writes the permuted rhs twice: bp (kept for the residuals) and x (solved in
    // place), and folds ||b||inf into norm set 0 -- no separate copy / norm launches
    dev::setrhs_perm(E.stream, h->bp, h->x, rhsx_dev, rhsz_dev, E.perm, (int)h->K.n, (int)h->K.m, E.N,
                     E.norm_set(0), E.norm_nan(0));
    h->x_holds_b = true;
    CHIP_HIP(hipGetLastError());
    return CHIP_OK;
}
int32_t chip_kkt_setrhs(chip_kkt *h, const double *rhsx, const double *rhsz) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    const size_t n = (size_t)h->K.n, m = (size_t)h->K.m;
    if (n) CHIP_HIP(hipMemcpyAsync(h->d_rhs, rhsx, n * sizeof(double), hipMemcpyHostToDevice, E.stream));
    if (m) CHIP_HIP(hipMemcpyAsync(h->d_rhs + n, rhsz, m * sizeof(double), hipMemcpyHostToDevice, E.stream));
    int rc = chip_kkt_setrhs_dev(h, h->d_rhs, h->d_rhs + n);
    if (rc) return rc;
    CHIP_HIP(hipStreamSynchronize(E.stream)); // the caller may reuse rhsx/rhsz (they were copied to d_rhs)
    return CHIP_OK;
}

// x <- K^-1 bp with iterative refinement (directldlkktsolver.rs:168-189, :266-321);
// everything in the engine's permuted numbering.  Returns the reference's bool.
//
// Buffer rotation instead of copies: x (solution), e (residual, then solved IN PLACE into the
// correction, then turned into the candidate x + dx), w (next residual).  Accepting a round
// renames (x, e, w) <- (e, w, x): the reference's mem::swap(x, dx) without moving data.
// (the refinement loop itself, shared by the L2 handle and by chip_ldl_solve_refined: xio / eio / wio are the three work
// vectors and come back renamed; x holds K^-1 bp's first approximation on entry)
// (in two halves, so that two independent solves can be enqueued on two streams before either is waited for)
static void refine_begin(Engine &E, const double *bp, double *x, double *e, double *w) {
    const chip_settings &st = E.st;
    if (!st.iterative_refinement_enable) {
        dev::norm_inf(E.stream, x, E.N, E.norm_set(1), E.norm_nan(1));
        return;
    }
    // The first refinement round is enqueued SPECULATIVELY together with the initial residual,
    // so that one host synchronisation (one D2H copy of norm sets 0..2) serves both decisions of
    // directldlkktsolver.rs:288-318; if ||e0|| already meets the tolerance the speculative
    // candidate is simply never looked at.  Decisions are exactly the reference's.
    E.enqueue_residual(e, bp, x, 1);
    if (st.iterative_refinement_max_iter >= 1) {
        // w <- e0 is needed if the round is rejected?  No: a rejected round leaves x untouched and e
        // is dead afterwards, so e is solved in place.
        E.enqueue_solve_inplace(e, x); // e <- x + K^-1 e0   (the candidate; "+ x" fused into the sweep)
        E.enqueue_residual(w, bp, e, 2);
    }
}
static int refine_finish(Engine &E, const double *bp, double *&xio, double *&eio, double *&wio, int &last_ir) {
    const chip_settings &st = E.st;
    int rc;
    if (!st.iterative_refinement_enable) {
        double nx;
        if ((rc = E.read_norm(1, &nx))) return rc;
        return std::isfinite(nx) ? 1 : 0; // x.is_finite(), directldlkktsolver.rs:180
    }
    double *x = xio, *e = eio, *w = wio;
    const double abstol = st.iterative_refinement_abstol, reltol = st.iterative_refinement_reltol;
    const double stopratio = st.iterative_refinement_stop_ratio;
    const int maxiter = st.iterative_refinement_max_iter;
    double nn[3] = {0, 0, 0};
    if ((rc = E.read_norms(0, maxiter >= 1 ? 3 : 2, nn))) return rc;
    const double normb = nn[0];
    double norme = nn[1];
    if (!std::isfinite(norme)) return 0;
    int set = 2;
    for (int it = 0; it < maxiter; it++) {
        if (norme <= abstol + reltol * normb) break;
        const double lastnorme = norme;
        if (it == 0) {
            norme = nn[2]; // already computed above
        } else {
            E.enqueue_solve_inplace(e, x);
            set += 1;
            if (set >= NRM_SETS) set = 3;
            if (it + 2 >= NRM_SETS) // the set is being reused: clear it first
                CHIP_HIP(hipMemsetAsync(E.norm_set(set), 0, NRM_SET_WORDS * sizeof(unsigned long long), E.stream));
            E.enqueue_residual(w, bp, e, set);
            if ((rc = E.read_norm(set, &norme))) return rc;
        }
        last_ir += 1;
        if (!std::isfinite(norme)) return 0;
        const double improved = lastnorme / norme;
        const bool accept = !(improved < stopratio) || improved > 1.0;
        if (accept) { // (x, e, w) <- (candidate, its residual, free)
            double *t = x;
            x = e;
            e = w;
            w = t;
        }
        if (improved < stopratio) break;
    }
    xio = x;
    eio = e;
    wio = w;
    return 1;
}
static int refine_core(Engine &E, const double *bp, double *&xio, double *&eio, double *&wio, int &last_ir) {
    refine_begin(E, bp, xio, eio, wio);
    return refine_finish(E, bp, xio, eio, wio, last_ir);
}
static int solve_core(chip_kkt *h) {
    Engine &E = h->E;
    const int N = E.N;
    if (!E.factored) return fail(CHIP_ERR_NOT_FACTORED, "solve() before the first update()");
    h->last_ir = 0;
    int rc;
    if (h->bp_stale && !h->rhs_deferred) h->rhs_deferred = true; // (the last fused solve left no permuted copy of b)
    if (h->rhs_deferred) { // (fused path not taken for this solve: stage the noted right-hand side now)
        h->rhs_deferred = false;
        h->bp_stale = false;
        if ((rc = E.zero_norm_sets())) return rc;
        dev::setrhs_perm(E.stream, h->bp, h->x, h->rhs_x, h->rhs_z, E.perm, (int)h->K.n, (int)h->K.m, E.N,
                         E.norm_set(0), E.norm_nan(0));
        h->x_holds_b = true;
    }
    if (!h->x_holds_b) { // solve() again on the same right-hand side, or a full-N rhs in bp
        if ((rc = E.zero_norm_sets())) return rc;
        CHIP_HIP(hipMemcpyAsync(h->x, h->bp, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, E.stream));
        dev::norm_inf(E.stream, h->bp, N, E.norm_set(0), E.norm_nan(0));
    }
    h->x_holds_b = false;
    E.enqueue_solve_inplace(h->x);
    int ok = refine_core(E, h->bp, h->x, h->e, h->dx, h->last_ir);
    if (ok == 0 && E.sweeps_after_failure()) {
        // non-finite with persistent sweeps in use: possibly a level barrier that timed out -- once more on the per-level
        // launches (a genuinely non-finite system fails again, at the price of one more solve)
        h->last_ir = 0;
        if ((rc = E.zero_norm_sets())) return rc;
        CHIP_HIP(hipMemcpyAsync(h->x, h->bp, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, E.stream));
        dev::norm_inf(E.stream, h->bp, N, E.norm_set(0), E.norm_nan(0));
        E.enqueue_solve_inplace(h->x);
        ok = refine_core(E, h->bp, h->x, h->e, h->dx, h->last_ir);
    }
    return ok;
}

// ---- L1 fast path (round 5): registered index sets, device-resident refinement, pinned caller buffers -----------------
// The strict drop-in moved 352 MB over PCIe per interior-point iteration of config 3 (all of K.nzval per refactor, b and x
// of every LDL' solve of every refinement round).  What the reference's DirectLDLKKTSolver actually changes per
// iteration are the entries of a few fixed index vectors (directldlkktsolver.rs:143,245; datamaps.rs:213-219), and its
// refinement (:266-321) needs nothing from the host but b.
static int ldl_values_to_device(chip_ldl *h) { // first direct update: the device copy becomes the caller's K.nzval
    Engine &E = h->E;
    if (h->dev_values) return CHIP_OK;
    if (h->dirty && E.nnzK) {
        int rc = E.upload_values(h->hK.data());
        if (rc) return rc;
    }
    h->dev_values = true;
    return CHIP_OK;
}
static int ldl_stage(chip_ldl *h, i64 k) {
    if (k <= h->d_vals_cap) return CHIP_OK;
    int rc = h->E.alloc(&h->d_vals, (size_t)k); // (the engine frees it with the handle; earlier, smaller ones too)
    if (rc) return rc;
    h->d_vals_cap = k;
    return CHIP_OK;
}
// a plain (index-carrying) update on a handle whose device copy is authoritative: index translated and shipped each time
static int ldl_plain_on_device(chip_ldl *h, int kind, const uint64_t *index, const double *values, double scalar,
                               const int8_t *signs, int64_t k) {
    Engine &E = h->E;
    NEED_DEVICE(E);
    if (k <= 0) return CHIP_OK;
    CHIP_HIP(hipSetDevice(E.device));
    std::vector<i32> pos((size_t)k);
    for (i64 i = 0; i < k; i++) {
        if (index[i] >= (uint64_t)E.nnzK) return fail(CHIP_ERR_ARG, "values update: index out of range");
        pos[(size_t)i] = E.h_k2v[(size_t)index[i]];
    }
    int *dpos = nullptr;
    int8_t *dsg = nullptr;
    int rc = CHIP_OK;
    // (every HIP call checked; the two temporaries are released on every path)
    auto hipok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == CHIP_OK) rc = fail(CHIP_ERR_HIP, hip_err(e, what));
        return e == hipSuccess;
    };
    if (hipok(hipMalloc((void **)&dpos, (size_t)k * sizeof(int)), "values update: index buffer") &&
        hipok(hipMemcpyAsync(dpos, pos.data(), (size_t)k * sizeof(int), hipMemcpyHostToDevice, E.stream), "values update: index copy")) {
        if (kind == 0) {
            if ((rc = ldl_stage(h, k)) == CHIP_OK &&
                hipok(hipMemcpyAsync(h->d_vals, values, (size_t)k * sizeof(double), hipMemcpyHostToDevice, E.stream), "values update: value copy"))
                dev::scatter_values(E.stream, E.Kx, dpos, h->d_vals, (int)k, 1.0);
        } else if (kind == 1) {
            dev::scale_values(E.stream, E.Kx, dpos, (int)k, scalar);
        } else if (hipok(hipMalloc((void **)&dsg, (size_t)k), "values update: sign buffer") &&
                   hipok(hipMemcpyAsync(dsg, signs, (size_t)k, hipMemcpyHostToDevice, E.stream), "values update: sign copy")) {
            dev::offset_values(E.stream, E.Kx, dpos, dsg, (int)k, scalar);
        }
        (void)hipok(hipGetLastError(), "values update: launch");
    }
    (void)hipok(hipStreamSynchronize(E.stream), "values update: synchronise");
    if (dpos) (void)hipFree(dpos);
    if (dsg) (void)hipFree(dsg);
    E.sx_valid = false;
    h->dirty = true;
    return rc;
}
int32_t chip_ldl_register_index(chip_ldl *h, const uint64_t *index, int64_t k, const int8_t *signs_or_null, int32_t *id_out) {
    if (!h || !id_out || k < 0 || (k > 0 && !index)) return fail(CHIP_ERR_ARG, "chip_ldl_register_index: bad argument");
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    std::vector<i32> pos((size_t)k);
    for (i64 i = 0; i < k; i++) {
        if (index[i] >= (uint64_t)E.nnzK) return fail(CHIP_ERR_ARG, "chip_ldl_register_index: index out of range");
        pos[(size_t)i] = E.h_k2v[(size_t)index[i]];
    }
    chip_ldl::IndexSet st;
    st.k = k;
    int rc;
    if ((rc = E.upload(&st.pos, pos, (size_t)k))) return rc;
    if (signs_or_null) {
        std::vector<int8_t> sg(signs_or_null, signs_or_null + k);
        if ((rc = E.upload(&st.signs, sg, (size_t)k))) return rc;
    }
    h->sets.push_back(st);
    *id_out = (int32_t)h->sets.size() - 1;
    return CHIP_OK;
}
static int ldl_set_of(chip_ldl *h, int32_t id, chip_ldl::IndexSet **st) {
    if (!h || id < 0 || (size_t)id >= h->sets.size()) return fail(CHIP_ERR_ARG, "unknown index set");
    *st = &h->sets[(size_t)id];
    return CHIP_OK;
}
int32_t chip_ldl_update_values_id(chip_ldl *h, int32_t id, const double *values) {
    chip_ldl::IndexSet *st;
    int rc = ldl_set_of(h, id, &st);
    if (rc) return rc;
    Engine &E = h->E;
    NEED_DEVICE(E);
    if (st->k == 0) return CHIP_OK;
    if (!values) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(E.device));
    if ((rc = ldl_values_to_device(h))) return rc;
    if ((rc = ldl_stage(h, st->k))) return rc;
    CHIP_HIP(hipMemcpyAsync(h->d_vals, values, (size_t)st->k * sizeof(double), hipMemcpyHostToDevice, E.stream));
    dev::scatter_values(E.stream, E.Kx, st->pos, h->d_vals, (int)st->k, 1.0);
    // (the caller may reuse `values` at once: the reference's update_values copies; a pinned source is read by the DMA
    // engine asynchronously, so wait for the copy -- not for the kernel)
    CHIP_HIP(hipStreamSynchronize(E.stream));
    E.sx_valid = false;
    h->dirty = true;
    return CHIP_OK;
}
int32_t chip_ldl_scale_values_id(chip_ldl *h, int32_t id, double scale) {
    chip_ldl::IndexSet *st;
    int rc = ldl_set_of(h, id, &st);
    if (rc) return rc;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    if ((rc = ldl_values_to_device(h))) return rc;
    dev::scale_values(E.stream, E.Kx, st->pos, (int)st->k, scale);
    E.sx_valid = false;
    h->dirty = true;
    return CHIP_OK;
}
int32_t chip_ldl_offset_values_id(chip_ldl *h, int32_t id, double offset) {
    chip_ldl::IndexSet *st;
    int rc = ldl_set_of(h, id, &st);
    if (rc) return rc;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    if ((rc = ldl_values_to_device(h))) return rc;
    dev::offset_values(E.stream, E.Kx, st->pos, st->signs, (int)st->k, offset);
    E.sx_valid = false;
    h->dirty = true;
    return CHIP_OK;
}
int32_t chip_ldl_pin_buffer(chip_ldl *h, void *ptr, uint64_t bytes) {
    if (!h || !ptr || !bytes) return CHIP_ERR_ARG;
    NEED_DEVICE(h->E);
    CHIP_HIP(hipSetDevice(h->E.device));
    for (void *p : h->pinned)
        if (p == ptr) return CHIP_OK;
    CHIP_HIP(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault));
    h->pinned.push_back(ptr);
    return CHIP_OK;
}
// solve + iterative refinement of directldlkktsolver.rs:266-321 with the handle's CURRENT values as K (the caller has
// restored the unregularised diagonal after refactor, :255-261): b in, x out, everything in between on the device
int32_t chip_ldl_solve_refined(chip_ldl *h, double *x, const double *b, const chip_settings *ir_settings, int32_t *iterations) {
    if (!h || !x || !b) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    if (!E.factored) return fail(CHIP_ERR_NOT_FACTORED, "solve() before the first refactor()");
    CHIP_HIP(hipSetDevice(E.device));
    int rc;
    const size_t N = (size_t)E.N;
    if (!h->r_bp) {
        if ((rc = E.alloc(&h->r_bp, N))) return rc;
        if ((rc = E.alloc(&h->r_x, N))) return rc;
        if ((rc = E.alloc(&h->r_e, N))) return rc;
        if ((rc = E.alloc(&h->r_w, N))) return rc;
    }
    if (h->dirty && !h->dev_values && E.nnzK) { // plain updates since the refactor (the restored diagonal): the residual reads them
        if ((rc = E.upload_values(h->hK.data()))) return rc;
        E.sx_valid = false;
        h->dirty = false;
    }
    struct RestoreSettings { // (every return below leaves the handle with its own refinement settings)
        Engine &E;
        chip_settings saved;
        ~RestoreSettings() { E.st = saved; }
    } restore{E, E.st};
    if (ir_settings) {
        E.st.iterative_refinement_enable = ir_settings->iterative_refinement_enable;
        E.st.iterative_refinement_reltol = ir_settings->iterative_refinement_reltol;
        E.st.iterative_refinement_abstol = ir_settings->iterative_refinement_abstol;
        E.st.iterative_refinement_max_iter = ir_settings->iterative_refinement_max_iter;
        E.st.iterative_refinement_stop_ratio = ir_settings->iterative_refinement_stop_ratio;
    }
    if (N) CHIP_HIP(hipMemcpyAsync(h->d_b, b, N * sizeof(double), hipMemcpyHostToDevice, E.stream));
    if ((rc = E.zero_norm_sets())) return rc;
    dev::permute_in(E.stream, h->r_bp, h->d_b, E.perm, E.N);
    if (N) CHIP_HIP(hipMemcpyAsync(h->r_x, h->r_bp, N * sizeof(double), hipMemcpyDeviceToDevice, E.stream));
    dev::norm_inf(E.stream, h->r_bp, E.N, E.norm_set(0), E.norm_nan(0));
    E.enqueue_solve_inplace(h->r_x);
    int its = 0;
    const int ok = refine_core(E, h->r_bp, h->r_x, h->r_e, h->r_w, its);
    if (ok == 0) (void)E.sweeps_after_failure(); // (stale barrier words must not outlive a failed solve)
    if (iterations) *iterations = its;
    if (ok != 1) return ok;
    dev::permute_out(E.stream, h->d_x, h->r_x, E.perm, E.N);
    if (N) CHIP_HIP(hipMemcpyAsync(x, h->d_x, N * sizeof(double), hipMemcpyDeviceToHost, E.stream));
    CHIP_HIP(hipStreamSynchronize(E.stream));
    return 1;
}

// the whole solve (setrhs permutation, LDL' solve, refinement with its decisions, getlhs) as ONE
// persistent launch; *slot = where the verdict will appear in the result ring
static int fused_enqueue(chip_kkt *h, double *lhsx_dev, double *lhsz_dev, int *slot) {
    Engine &E = h->E;
    int rc0;
    if (!E.factored) return fail(CHIP_ERR_NOT_FACTORED, "solve() before the first update()");
    const chip_settings &st = E.st;
    *slot = E.ir_next;
    E.ir_next = (E.ir_next + 1) % Engine::IR_RING;
    dev::IrView ir{};
    ir.rx = h->rhs_x;
    ir.rz = h->rhs_z;
    ir.n = (int)h->K.n;
    ir.m = (int)h->K.m;
    ir.N = E.N;
    ir.perm = E.perm;
    ir.run_ptr = h->ir_run_ptr;
    ir.runs = h->ir_runs;
    ir.bp = h->bp;
    ir.xa = h->x;
    ir.xb = h->e;
    ir.ebuf = h->dx;
    ir.lhsx = lhsx_dev;
    ir.lhsz = lhsz_dev;
    ir.part = E.ir_part;
    ir.ctl = E.ir_ctl;
    ir.rel = E.ir_rel;
    ir.epoch = (E.ir_epoch = (E.ir_epoch + 1) & 0x3fffff); // (tags are epoch << 8 | barrier number)
    ir.res = E.ir_res + 4 * *slot;
    ir.abstol = st.iterative_refinement_abstol;
    ir.reltol = st.iterative_refinement_reltol;
    ir.stopratio = st.iterative_refinement_stop_ratio;
    ir.maxiter = st.iterative_refinement_max_iter;
    ir.ir_enable = st.iterative_refinement_enable;
    static long long *dbg_dev = nullptr;
    const bool dbg_on = switches().ir_debug > 0;
    if (dbg_on && !dbg_dev) {
        (void)hipMalloc((void **)&dbg_dev, 256 * sizeof(long long));
    }
    if (dbg_on) (void)hipMemsetAsync(dbg_dev, 0, 256 * sizeof(long long), E.stream);
    ir.dbg = dbg_on ? dbg_dev : nullptr;
    // CHIP_IR_DEBUG=2: stamps of EVERY workgroup, dumped to the file CHIP_IR_DEBUG_FILE (default /tmp/chip_ir_stamps.bin:
    // int32 G, then G x 32 int64) after each launch -- tools/ir_skew.py turns them into per-phase statistics
    const bool dbg_all_on = switches().ir_debug >= 2;
    static long long *dbg_all_dev = nullptr;
    static size_t dbg_all_len = 0;
    ir.dbg_all = nullptr;
    if (dbg_all_on) {
        const size_t need = (size_t)E.ir_grid * 32;
        if (need > dbg_all_len) {
            if (dbg_all_dev) (void)hipFree(dbg_all_dev);
            (void)hipMalloc((void **)&dbg_all_dev, need * sizeof(long long));
            dbg_all_len = need;
        }
        (void)hipMemsetAsync(dbg_all_dev, 0, need * sizeof(long long), E.stream);
        ir.dbg_all = dbg_all_dev;
    }
    ir.test_drop = h->ir_test_drop ? 1 : 0;
    ir.flat = switches().no_flat ? 0 : 1;
    const bool step_kernel = E.gstep_solve_on && !switches().no_step_kernel && ir.maxiter <= 30;
    ir.sf = (h->ir_sf && !step_kernel) ? 1 : 0;
    if (ir.sf) ir.bp = nullptr; // (k_bundle_irs reads b through the runs every time; see chip_kkt::bp_stale)
    {
        // (the residual of the last round still reads the right-hand side when that round's candidate is written)
        auto overlap = [](const double *a, size_t na, const double *b_, size_t nb_) {
            return a && b_ && a < b_ + nb_ && b_ < a + na;
        };
        const size_t n = (size_t)h->K.n, m = (size_t)h->K.m;
        const int flags = switches().irs_flags < 0 ? 3 : switches().irs_flags;
        ir.sf_flags = flags;
        ir.spec_out = (flags & 2) && !(overlap(lhsx_dev, n, ir.rx, n) || overlap(lhsx_dev, n, ir.rz, m) || overlap(lhsz_dev, m, ir.rx, n) ||
                        overlap(lhsz_dev, m, ir.rz, m));
    }
    h->bp_stale = ir.sf != 0;
    if ((rc0 = E.wait_for_exchange())) return rc0; // (no foreign kernel of ours beside a persistent launch)
    h->fused_args[*slot] = {h->rhs_x, h->rhs_z, lhsx_dev, lhsz_dev};
    h->rhs_deferred = false;
    h->x_holds_b = false;
    E.prof_begin(PF_IR);
    int rc;
    if (step_kernel) {
        // grouped fold with small bundles: the register-resident form of the same launch (bundle_gstep.hip)
        E.gstep.epoch += 1;
        dev::GStepView gsv = E.gstep;
        if (!E.gstep_vals_valid) gsv.gsl = gsv.gsu = nullptr; // (gather L and K through the source positions instead)
        rc = dev::gstep_solve(E.stream, E.view(), E.bundles, ir, E.gfold, gsv);
    } else {
        rc = dev::bundle_ir(E.stream, E.view(), E.bundles, E.fold, ir, E.ir_grid, E.ir_tw, E.gfold);
    }
    E.prof_end(PF_IR);
    if (rc) return fail(CHIP_ERR_HIP, hip_err((hipError_t)rc, "fused solve launch"));
    if (dbg_all_on) {
        (void)hipStreamSynchronize(E.stream);
        std::vector<long long> t((size_t)E.ir_grid * 32);
        (void)hipMemcpy(t.data(), dbg_all_dev, t.size() * sizeof(long long), hipMemcpyDeviceToHost);
        const std::string &path = switches().ir_debug_file;
        if (FILE *f = std::fopen(path.empty() ? "/tmp/chip_ir_stamps.bin" : path.c_str(), "wb")) {
            const int g = E.ir_grid;
            std::fwrite(&g, sizeof(int), 1, f);
            std::fwrite(t.data(), sizeof(long long), t.size(), f);
            std::fclose(f);
        }
    } else if (dbg_on) {
        long long t[256];
        (void)hipStreamSynchronize(E.stream);
        (void)hipMemcpy(t, dbg_dev, sizeof(t), hipMemcpyDeviceToHost);
        for (int w = 0; w < 2; w++) {
            std::fprintf(stderr, "k_bundle_ir wg%d phases (us):", w);
            for (int i = 1; i < 64 && t[64 * w + i]; i++) std::fprintf(stderr, " %.1f", (t[64 * w + i] - t[64 * w + i - 1]) * 0.01);
            std::fprintf(stderr, "\n");
        }
    }
    return CHIP_OK;
}
// verdict of ring slot `slot` (after the stream has been synchronised and the ring copied to the host);
// FUSED_TIMEOUT: the grid barrier timed out (not all workgroups were resident), the counters have been cleared
constexpr int FUSED_TIMEOUT = -1000;
static int fused_verdict(chip_kkt *h, int slot) {
    Engine &E = h->E;
    const int *r = E.ir_res_host + 4 * slot;
    if (r[2] || r[0] == 0) {
        (void)hipMemsetAsync(E.ir_ctl, 0, E.ir_ctl_len * sizeof(int), E.stream);
        return FUSED_TIMEOUT;
    }
    h->last_ir = r[1];
    return r[0] > 0 ? 1 : 0;
}
static int solve_core(chip_kkt *h);
// the solve of ring slot `slot` once more, one kernel per phase with the refinement decisions on the host
// (same results up to rounding); the slot's right-hand side buffers must still hold what was enqueued
static int fused_retry_unfused(chip_kkt *h, int slot) {
    Engine &E = h->E;
    const chip_kkt::FusedArgs &a = h->fused_args[slot];
    h->fused_fallbacks += 1;
    h->rhs_x = a.rx;
    h->rhs_z = a.rz;
    h->rhs_deferred = true;
    h->x_holds_b = true;
    const int ok = solve_core(h);
    if (ok != 1) return ok;
    dev::getlhs_perm(E.stream, a.lx, a.lz, h->x, E.iperm, (int)h->K.n, (int)h->K.m);
    CHIP_HIP(hipGetLastError());
    CHIP_HIP(hipStreamSynchronize(E.stream));
    return 1;
}
static int fused_read_ring(chip_kkt *h) { return h->E.read_mailbox(); } // (the ring lives in the mailbox)

int32_t chip_kkt_solve_dev(chip_kkt *h, double *lhsx_dev, double *lhsz_dev) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    if (h->bp_stale && !h->rhs_deferred) { // (solve() again on the same right-hand side: the noted vectors once more)
        h->rhs_deferred = true;
        h->x_holds_b = true;
    }
    if (E.ir_fused && h->rhs_deferred && (E.prof_family == PF_NONE || E.prof_family >= PF_IR)) {
        int slot = 0;
        int rc = fused_enqueue(h, lhsx_dev, lhsz_dev, &slot);
        if (rc) return rc;
        if ((rc = fused_read_ring(h))) return rc;
        rc = fused_verdict(h, slot);
        return rc == FUSED_TIMEOUT ? fused_retry_unfused(h, slot) : rc;
    }
    int ok = solve_core(h);
    if (ok != 1) return ok;
    dev::getlhs_perm(E.stream, lhsx_dev, lhsz_dev, h->x, E.iperm, (int)h->K.n, (int)h->K.m);
    CHIP_HIP(hipGetLastError());
    return 1;
}
int32_t chip_kkt_solve_dev_enqueue(chip_kkt *h, double *lhsx_dev, double *lhsz_dev) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    if ((int)h->pend_slots.size() >= Engine::IR_RING) return fail(CHIP_ERR_ARG, "solve_dev_enqueue: 16 solves pending, collect first");
    if (h->bp_stale && !h->rhs_deferred) { // (solve() again on the same right-hand side: the noted vectors once more)
        h->rhs_deferred = true;
        h->x_holds_b = true;
    }
    if (E.ir_fused && h->rhs_deferred && (E.prof_family == PF_NONE || E.prof_family >= PF_IR)) {
        int slot = 0;
        int rc = fused_enqueue(h, lhsx_dev, lhsz_dev, &slot);
        if (rc) return rc;
        h->pend_slots.push_back(slot);
        return CHIP_OK;
    }
    // no fused launch for this system: the refinement decisions need the host -- a pending update's verdict is
    // read first (the solve must not run on a failed factorisation), then the solve runs synchronously
    if (h->pend_update) {
        h->pend_update = 2; // collected, verdict kept
        h->pend_update_ok = update_verdict(h, E.refactor_collect());
        if (h->pend_update_ok != 1) {
            h->pend_slots.push_back(-1 - 0);
            return CHIP_OK;
        }
    }
    const int ok = chip_kkt_solve_dev(h, lhsx_dev, lhsz_dev);
    if (ok < 0) return ok;
    h->pend_slots.push_back(-1 - ok); // -1 = failed, -2 = succeeded (already known)
    return CHIP_OK;
}
// Two INDEPENDENT solves of one interior-point iteration -- K x = [-q; b] of kktsystem.rs:108-125 (it does not depend on
// the residuals) and the affine direction of core/solver.rs:351-361 -- as one call.  Systems whose top is level-scheduled
// (a sweep = a chain of small launches, refinement decisions on the host) run them on two streams: both chains are
// enqueued before either is waited for, and the device overlaps them; the third solve of the iteration (the combined
// direction) depends on the affine result and stays a call of its own.  Fused handles (one persistent launch per solve,
// which fills the chip) run the two launches one after the other.  Same results as two chip_kkt_solve_dev_enqueue calls;
// two verdicts are appended for chip_kkt_collect.
int32_t chip_kkt_solve2_dev_enqueue(chip_kkt *h, const double *rhsx_a, const double *rhsz_a, double *lhsx_a, double *lhsz_a,
                                    const double *rhsx_b, const double *rhsz_b, double *lhsx_b, double *lhsz_b) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    if ((int)h->pend_slots.size() + 2 > Engine::IR_RING) return fail(CHIP_ERR_ARG, "solve2_dev_enqueue: 16 solves pending, collect first");
    int rc;
    if (E.ir_fused || !E.pair_ok() || switches().no_solve_pair || E.prof_family != PF_NONE) { // (one after the other)
        if ((rc = chip_kkt_setrhs_dev(h, rhsx_a, rhsz_a))) return rc;
        if ((rc = chip_kkt_solve_dev_enqueue(h, lhsx_a, lhsz_a))) return rc;
        if ((rc = chip_kkt_setrhs_dev(h, rhsx_b, rhsz_b))) return rc;
        return chip_kkt_solve_dev_enqueue(h, lhsx_b, lhsz_b);
    }
    if (h->pend_update) { // the refinement decisions need the host: the pending update's verdict first
        h->pend_update = 2;
        h->pend_update_ok = update_verdict(h, E.refactor_collect());
        if (h->pend_update_ok != 1) {
            h->pend_slots.push_back(-1);
            h->pend_slots.push_back(-1);
            return CHIP_OK;
        }
    }
    if (!E.factored) return fail(CHIP_ERR_NOT_FACTORED, "solve() before the first update()");
    const size_t N = (size_t)E.N;
    if (!h->bp2) {
        if ((rc = E.alloc(&h->bp2, N))) return rc;
        if ((rc = E.alloc(&h->x2, N))) return rc;
        if ((rc = E.alloc(&h->e2, N))) return rc;
        if ((rc = E.alloc(&h->dx2, N))) return rc;
    }
    if ((rc = E.pair_begin())) return rc;
    const int n = (int)h->K.n, m = (int)h->K.m;
    h->last_ir = h->last_ir2 = 0;
    h->rhs_deferred = false;
    h->x_holds_b = false;
    h->bp_stale = false;
    // ---- both chains enqueued: A on the engine's stream, B on the second one
    if (E.pair_lockstep_ok()) {
        // (wide chain supernodes with a form for two right-hand sides: the two chains walk the levels together, the wide
        // levels as ONE launch that streams the panels for both vectors -- Engine::enqueue_solve_pair)
        if ((rc = E.zero_norm_sets())) return rc;
        dev::setrhs_perm(E.stream, h->bp, h->x, rhsx_a, rhsz_a, E.perm, n, m, E.N, E.norm_set(0), E.norm_nan(0));
        E.swap_ctx();
        rc = E.zero_norm_sets();
        if (!rc) dev::setrhs_perm(E.stream, h->bp2, h->x2, rhsx_b, rhsz_b, E.perm, n, m, E.N, E.norm_set(0), E.norm_nan(0));
        E.swap_ctx();
        if (rc) return rc;
        E.enqueue_solve_pair(h->x, nullptr, h->x2, nullptr);
        const chip_settings &st = E.st;
        if (!st.iterative_refinement_enable) {
            dev::norm_inf(E.stream, h->x, E.N, E.norm_set(1), E.norm_nan(1));
            E.swap_ctx();
            dev::norm_inf(E.stream, h->x2, E.N, E.norm_set(1), E.norm_nan(1));
            E.swap_ctx();
        } else { // (refine_begin for both: the residuals, then the first round enqueued ahead of its decision)
            E.enqueue_residual_pair(h->e, h->bp, h->x, h->e2, h->bp2, h->x2, 1);
            if (st.iterative_refinement_max_iter >= 1) {
                E.enqueue_solve_pair(h->e, h->x, h->e2, h->x2);
                E.enqueue_residual_pair(h->dx, h->bp, h->e, h->dx2, h->bp2, h->e2, 2);
            }
        }
    } else {
    if ((rc = E.zero_norm_sets())) return rc;
    dev::setrhs_perm(E.stream, h->bp, h->x, rhsx_a, rhsz_a, E.perm, n, m, E.N, E.norm_set(0), E.norm_nan(0));
    E.enqueue_solve_inplace(h->x);
    refine_begin(E, h->bp, h->x, h->e, h->dx);
    E.swap_ctx();
    rc = E.zero_norm_sets();
    if (!rc) {
        dev::setrhs_perm(E.stream, h->bp2, h->x2, rhsx_b, rhsz_b, E.perm, n, m, E.N, E.norm_set(0), E.norm_nan(0));
        E.enqueue_solve_inplace(h->x2);
        refine_begin(E, h->bp2, h->x2, h->e2, h->dx2);
    }
    E.swap_ctx();
    if (rc) return rc;
    }
    // ---- decisions (and any further rounds) of A, then of B
    const int oka = refine_finish(E, h->bp, h->x, h->e, h->dx, h->last_ir);
    if (oka == 1) dev::getlhs_perm(E.stream, lhsx_a, lhsz_a, h->x, E.iperm, n, m);
    E.swap_ctx();
    const int okb = refine_finish(E, h->bp2, h->x2, h->e2, h->dx2, h->last_ir2);
    if (oka == 0 || okb == 0) (void)E.sweeps_after_failure(); // (stale barrier words must not outlive a failed solve)
    if (okb == 1) dev::getlhs_perm(E.stream, lhsx_b, lhsz_b, h->x2, E.iperm, n, m);
    const hipError_t se = hipStreamSynchronize(E.stream); // (the second stream: its results are complete when this call returns)
    E.swap_ctx();
    if (se != hipSuccess) return fail(CHIP_ERR_HIP, hip_err(se, "second solve stream"));
    if (oka < 0) return oka;
    if (okb < 0) return okb;
    h->pend_slots.push_back(-1 - oka);
    h->pend_slots.push_back(-1 - okb);
    return CHIP_OK;
}
int32_t chip_kkt_collect(chip_kkt *h, int32_t *update_ok, int32_t *nsolves, int32_t solves_ok[16]) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    int rc;
    int uok = 1;
    bool ring = false;
    for (int sl : h->pend_slots) ring = ring || sl >= 0;
    // ONE device-to-host copy (the mailbox: refactor status, cone-scaling verdict, the solves' verdict ring)
    if (h->pend_update == 1) uok = update_verdict(h, E.refactor_collect());
    else {
        if (h->pend_update == 2) uok = h->pend_update_ok;
        if (ring && (rc = E.read_mailbox())) return rc;
        if (!ring) CHIP_HIP(hipStreamSynchronize(E.stream));
    }
    h->pend_update = 0;
    if (uok < 0) return uok;
    if (update_ok) *update_ok = uok;
    int n = 0;
    rc = CHIP_OK;
    bool suspect = false; // a timed-out launch leaves the barrier counters dirty for the launches queued behind it
    for (int sl : h->pend_slots) {
        int v;
        bool repeated = false;
        if (sl < 0) v = -1 - sl;
        else {
            v = fused_verdict(h, sl);
            if (v == FUSED_TIMEOUT || suspect) {
                suspect = true;
                repeated = true;
                v = fused_retry_unfused(h, sl);
            }
            if (v < 0) rc = v;
        }
        // (2: the lhs was garbage until this repeat -- whatever consumed it on the device must be re-issued)
        if (solves_ok && n < 16) solves_ok[n] = v > 0 ? (repeated ? 2 : 1) : 0;
        n++;
    }
    h->pend_slots.clear();
    if (nsolves) *nsolves = n;
    return rc;
}
int32_t chip_kkt_solve(chip_kkt *h, double *lhsx, double *lhsz) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    const size_t n = (size_t)h->K.n, m = (size_t)h->K.m;
    int ok = chip_kkt_solve_dev(h, h->d_lhs, h->d_lhs + n);
    if (ok != 1) return ok;
    if (lhsx && n) CHIP_HIP(hipMemcpyAsync(lhsx, h->d_lhs, n * sizeof(double), hipMemcpyDeviceToHost, E.stream));
    if (lhsz && m) CHIP_HIP(hipMemcpyAsync(lhsz, h->d_lhs + n, m * sizeof(double), hipMemcpyDeviceToHost, E.stream));
    CHIP_HIP(hipStreamSynchronize(E.stream));
    return 1;
}
int32_t chip_kkt_set_settings(chip_kkt *h, const chip_settings *s) {
    if (!h || !s) return CHIP_ERR_ARG;
    chip_settings &d = h->E.st;
    d.static_regularization_enable = s->static_regularization_enable;
    d.static_regularization_constant = s->static_regularization_constant;
    d.static_regularization_proportional = s->static_regularization_proportional;
    d.dynamic_regularization_enable = s->dynamic_regularization_enable;
    d.dynamic_regularization_eps = s->dynamic_regularization_eps;
    d.dynamic_regularization_delta = s->dynamic_regularization_delta;
    d.iterative_refinement_enable = s->iterative_refinement_enable;
    d.iterative_refinement_reltol = s->iterative_refinement_reltol;
    d.iterative_refinement_abstol = s->iterative_refinement_abstol;
    d.iterative_refinement_max_iter = s->iterative_refinement_max_iter;
    d.iterative_refinement_stop_ratio = s->iterative_refinement_stop_ratio;
    d.linesearch_backtrack_step = s->linesearch_backtrack_step;
    d.min_terminate_step_length = s->min_terminate_step_length;
    return CHIP_OK;
}
int32_t chip_kkt_scaling_ok(chip_kkt *h) {
    if (!h) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    if (!h->scaling_pending_check) return 1;
    CHIP_HIP(hipSetDevice(E.device));
    int rc = E.read_mailbox();
    if (rc) return rc;
    h->scaling_pending_check = false;
    return E.mb_host->soc_fail == h->scaling_gen ? 0 : 1;
}
int32_t chip_kkt_solve_full(chip_kkt *h, double *x, const double *b) {
    if (!h || !x || !b) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    const size_t bytes = (size_t)E.N * sizeof(double);
    CHIP_HIP(hipMemcpyAsync(h->d_tmp, b, bytes, hipMemcpyHostToDevice, E.stream));
    dev::permute_in(E.stream, h->bp, h->d_tmp, E.perm, E.N);
    h->x_holds_b = false;
    // bp was written directly: a right-hand side noted by an earlier setrhs() (borrowed pointers) is void
    h->rhs_deferred = false;
    h->bp_stale = false;
    h->rhs_x = h->rhs_z = nullptr;
    int ok = solve_core(h);
    if (ok != 1) return ok;
    dev::permute_out(E.stream, h->d_tmp, h->x, E.perm, E.N);
    CHIP_HIP(hipMemcpyAsync(x, h->d_tmp, bytes, hipMemcpyDeviceToHost, E.stream));
    CHIP_HIP(hipStreamSynchronize(E.stream));
    return 1;
}

static int update_block(chip_kkt *h, const int *map, const std::vector<i64> &hmap, const double *vals, size_t k) {
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    if (!k) return CHIP_OK;
    for (size_t i = 0; i < k; i++) h->K.nzval[(size_t)hmap[i]] = vals[i]; // host mirror (chip_kkt_get_matrix)
    CHIP_HIP(hipMemcpyAsync(h->d_tmp, vals, k * sizeof(double), hipMemcpyHostToDevice, E.stream));
    dev::scatter_values(E.stream, E.Kx, map, h->d_tmp, (int)k, 1.0);
    CHIP_HIP(hipStreamSynchronize(E.stream));
    return CHIP_OK;
}
int32_t chip_kkt_update_P(chip_kkt *h, const double *Pnzval) {
    if (!h || !Pnzval) return CHIP_ERR_ARG;
    int rc = update_block(h, h->mapP, h->K.mapP, Pnzval, h->K.mapP.size() - 1);
    double mx = 0.0;
    for (i64 i = 0; i < h->K.n; i++) {
        const double a = h->K.nzval[(size_t)h->K.diagP[(size_t)i]];
        if (a != a) mx = a;
        else if (mx == mx) mx = std::max(mx, std::fabs(a));
    }
    h->static_diag_max = mx;
    return rc;
}
int32_t chip_kkt_update_A(chip_kkt *h, const double *Anzval) {
    if (!h || !Anzval) return CHIP_ERR_ARG;
    return update_block(h, h->mapA, h->K.mapA, Anzval, h->K.mapA.size() - 1);
}
int32_t chip_kkt_mul_Hs_dev(chip_kkt *h, double *y_dev, const double *x_dev) {
    if (!h) return CHIP_ERR_ARG;
    if (h->has_hostHs)
        return fail(CHIP_ERR_UNSUPPORTED, "mul_Hs: a cone kind that is not held on the device");
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    dev::cones_mul_Hs(E.stream, h->nn_rows, h->nn_count, h->soc, h->zero_rows, h->zero_count, y_dev, x_dev);
    dev::ns3_mul_hs(E.stream, h->ns3, y_dev, x_dev);
    dev::gpw_mul_hs(E.stream, h->gpw, y_dev, x_dev);
    dev::psd_mul_hs(E.stream, h->psd, y_dev, x_dev);
    CHIP_HIP(hipGetLastError());
    return CHIP_OK;
}
#define NEED_SYMMETRIC(h)                                                                      \
    if ((h)->has_hostHs || (h)->ns3.ncones || (h)->gpw.ncones)                                 \
    return fail(CHIP_ERR_UNSUPPORTED, "margins / scaled_unit_shift: symmetric, device-held cones only")
#define NEED_STEP_OPS(h)                                                                       \
    if ((h)->has_hostHs)                                                                       \
    return fail(CHIP_ERR_UNSUPPORTED, "cone step operations: a cone kind that is not held on the device")

int32_t chip_kkt_scaled_unit_shift_dev(chip_kkt *h, double *z_dev, double alpha, int32_t primal_cone) {
    if (!h || !z_dev) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    NEED_SYMMETRIC(h);
    CHIP_HIP(hipSetDevice(E.device));
    dev::cone_unit_shift(E.stream, h->nn_rows, h->nn_count, h->zero_rows, h->zero_count, h->soc, z_dev, alpha,
                         primal_cone ? 1 : 0);
    dev::psd_unit_shift(E.stream, h->psd, z_dev, alpha);
    CHIP_HIP(hipGetLastError());
    return CHIP_OK;
}
int32_t chip_kkt_affine_ds_dev(chip_kkt *h, double *ds_dev, const double *s_dev) {
    if (!h || !ds_dev) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    NEED_STEP_OPS(h);
    if ((h->ns3.ncones || h->gpw.ncones) && !s_dev) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(E.device));
    dev::cone_affine_ds(E.stream, h->nn_rows, h->nn_count, h->zero_rows, h->zero_count, h->soc, ds_dev);
    dev::ns3_affine_ds(E.stream, h->ns3, ds_dev, s_dev);
    dev::gpw_copy(E.stream, h->gpw, ds_dev, s_dev);
    dev::psd_affine_ds(E.stream, h->psd, ds_dev);
    CHIP_HIP(hipGetLastError());
    return CHIP_OK;
}
int32_t chip_kkt_combined_ds_shift_dev(chip_kkt *h, double *shift_dev, double *step_z_dev, double *step_s_dev,
                                       double sigma_mu) {
    if (!h || !shift_dev || !step_z_dev || !step_s_dev) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    NEED_STEP_OPS(h);
    CHIP_HIP(hipSetDevice(E.device));
    dev::cone_combined_ds_shift(E.stream, h->nn_rows, h->nn_count, h->zero_rows, h->zero_count, h->soc, shift_dev,
                                step_z_dev, step_s_dev, sigma_mu);
    dev::ns3_combined_ds_shift(E.stream, h->ns3, shift_dev, step_z_dev, step_s_dev, sigma_mu);
    dev::gpw_combined_ds_shift(E.stream, h->gpw, shift_dev, sigma_mu);
    dev::psd_combined_ds_shift(E.stream, h->psd, shift_dev, step_z_dev, step_s_dev, sigma_mu);
    CHIP_HIP(hipGetLastError());
    return CHIP_OK;
}
int32_t chip_kkt_ds_from_dz_offset_dev(chip_kkt *h, double *out_dev, const double *ds_dev, const double *z_dev) {
    if (!h || !out_dev || !ds_dev || !z_dev) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    NEED_STEP_OPS(h);
    CHIP_HIP(hipSetDevice(E.device));
    dev::cone_ds_from_dz_offset(E.stream, h->nn_rows, h->nn_count, h->zero_rows, h->zero_count, h->soc, out_dev,
                                ds_dev, z_dev);
    dev::ns3_ds_from_dz_offset(E.stream, h->ns3, out_dev, ds_dev);
    dev::gpw_copy(E.stream, h->gpw, out_dev, ds_dev);
    dev::psd_ds_from_dz_offset(E.stream, h->psd, out_dev, ds_dev);
    CHIP_HIP(hipGetLastError());
    return CHIP_OK;
}
static int ensure_partials(chip_kkt *h) {
    if (h->d_partial) return CHIP_OK;
    h->partial_cap = 1024 + h->soc.ncones + h->psd.ncones + h->gpw.ncones + (h->ns3.ncones + 255) / 256 + 8;
    int rc = h->E.alloc(&h->d_partial, (size_t)h->partial_cap * 2);
    if (rc) return rc;
    h->h_partial.resize((size_t)h->partial_cap * 2);
    return CHIP_OK;
}
int32_t chip_kkt_step_length_dev(chip_kkt *h, const double *dz_dev, const double *ds_dev, const double *z_dev,
                                 const double *s_dev, double alpha_max, double *alpha_out) {
    if (!h || !alpha_out) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    NEED_STEP_OPS(h);
    CHIP_HIP(hipSetDevice(E.device));
    int rc = ensure_partials(h);
    if (rc) return rc;
    // symmetric cones first (compositecone.rs:326-327)
    int used = dev::cone_step_length(E.stream, h->nn_rows, h->nn_count, h->soc, dz_dev, ds_dev, z_dev, s_dev,
                                     alpha_max, h->d_partial, 1024);
    used += dev::psd_step_length(E.stream, h->psd, dz_dev, ds_dev, alpha_max, h->d_partial + used);
    double a = alpha_max;
    if (used) {
        CHIP_HIP(hipMemcpyAsync(h->h_partial.data(), h->d_partial, (size_t)used * sizeof(double),
                                hipMemcpyDeviceToHost, E.stream));
        CHIP_HIP(hipStreamSynchronize(E.stream));
        for (int i = 0; i < used; i++) a = std::min(a, h->h_partial[i]); // T::min: NaN-ignoring like f64::min
    }
    if (h->ns3.ncones || h->gpw.ncones) { // back off from the boundary, then the nonsymmetric cones (:329-337)
        a = std::min(a, 1.0 - std::sqrt(2.220446049250313e-16));
        used = dev::ns3_step_length(E.stream, h->ns3, dz_dev, ds_dev, z_dev, s_dev, a,
                                    E.st.min_terminate_step_length, E.st.linesearch_backtrack_step, h->d_partial);
        used += dev::gpw_step_length(E.stream, h->gpw, dz_dev, ds_dev, z_dev, s_dev, a,
                                     E.st.min_terminate_step_length, E.st.linesearch_backtrack_step,
                                     h->d_partial + used);
        CHIP_HIP(hipMemcpyAsync(h->h_partial.data(), h->d_partial, (size_t)used * sizeof(double),
                                hipMemcpyDeviceToHost, E.stream));
        CHIP_HIP(hipStreamSynchronize(E.stream));
        for (int i = 0; i < used; i++) a = std::min(a, h->h_partial[i]);
    }
    *alpha_out = a;
    return CHIP_OK;
}
int32_t chip_kkt_set_genpow_alpha(chip_kkt *h, int64_t cone_index, const double *alpha) {
    if (!h || !alpha) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    for (size_t g = 0; g < h->gpw_cone_index.size(); g++) {
        if (h->gpw_cone_index[g] != cone_index) continue;
        const int d1 = h->gpw_dim1[g];
        double sum = 0.0, sq = 0.0;
        for (int k = 0; k < d1; k++) {
            if (!(alpha[k] > 0.0)) return fail(CHIP_ERR_ARG, "GenPowerConeT: powers must be positive"); // genpowcone.rs:27
            sum += alpha[k];
            sq += alpha[k] * alpha[k];
        }
        if (std::fabs(1.0 - sum) >= 2.220446049250313e-16 * d1 * 0.5 + 1e-15)
            return fail(CHIP_ERR_ARG, "GenPowerConeT: powers must sum to one"); // genpowcone.rs:28
        CHIP_HIP(hipSetDevice(E.device));
        double *st = h->gpw.state + h->gpw_state_off[g];
        const int d2 = (int)h->K.cones[(size_t)cone_index].dim2;
        const double psi = 1.0 / sq;
        CHIP_HIP(hipMemcpyAsync(st, alpha, (size_t)d1 * sizeof(double), hipMemcpyHostToDevice, E.stream));
        CHIP_HIP(hipMemcpyAsync(st + 6 * d1 + 4 * d2 + 2, &psi, sizeof(double), hipMemcpyHostToDevice, E.stream));
        CHIP_HIP(hipStreamSynchronize(E.stream));
        return CHIP_OK;
    }
    return fail(CHIP_ERR_ARG, "cone_index is not a GenPowerConeT");
}
int32_t chip_kkt_compute_barrier_dev(chip_kkt *h, const double *z_dev, const double *s_dev, const double *dz_dev,
                                     const double *ds_dev, double alpha, double *barrier_out) {
    if (!h || !barrier_out || !z_dev || !s_dev || !dz_dev || !ds_dev) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    NEED_STEP_OPS(h);
    CHIP_HIP(hipSetDevice(E.device));
    int rc = ensure_partials(h);
    if (rc) return rc;
    int used = dev::cone_barrier(E.stream, h->nn_rows, h->nn_count, h->soc, h->ns3, z_dev, s_dev, dz_dev, ds_dev,
                                 alpha, h->d_partial);
    used += dev::psd_barrier(E.stream, h->psd, z_dev, s_dev, dz_dev, ds_dev, alpha, h->d_partial + used);
    // scratch for the GenPow primal gradients: the cones' own slices of the (otherwise NN / SOC) w state
    used += dev::gpw_barrier(E.stream, h->gpw, z_dev, s_dev, dz_dev, ds_dev, alpha, h->d_partial + used, h->d_w);
    double b = 0.0;
    if (used) {
        CHIP_HIP(hipMemcpyAsync(h->h_partial.data(), h->d_partial, (size_t)used * sizeof(double),
                                hipMemcpyDeviceToHost, E.stream));
        CHIP_HIP(hipStreamSynchronize(E.stream));
        for (int i = 0; i < used; i++) b += h->h_partial[i];
    }
    *barrier_out = b;
    return CHIP_OK;
}
int32_t chip_kkt_unit_initialization_dev(chip_kkt *h, double *z_dev, double *s_dev) {
    if (!h || !z_dev || !s_dev) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    NEED_STEP_OPS(h);
    CHIP_HIP(hipSetDevice(E.device));
    dev::cone_unit_initialization(E.stream, h->nn_rows, h->nn_count, h->soc, h->ns3, z_dev, s_dev, (int)h->K.m);
    dev::psd_unit_initialization(E.stream, h->psd, z_dev, s_dev);
    dev::gpw_unit_initialization(E.stream, h->gpw, z_dev, s_dev);
    CHIP_HIP(hipGetLastError());
    return CHIP_OK;
}
int32_t chip_kkt_margins_dev(chip_kkt *h, const double *z_dev, double *alpha_out, double *beta_out) {
    if (!h || !alpha_out || !beta_out) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    NEED_SYMMETRIC(h);
    CHIP_HIP(hipSetDevice(E.device));
    int rc = ensure_partials(h);
    if (rc) return rc;
    double *pmin = h->d_partial, *psum = h->d_partial + h->partial_cap;
    int used = dev::cone_margins(E.stream, h->nn_rows, h->nn_count, h->soc, z_dev, pmin, psum, 1024);
    used += dev::psd_margins(E.stream, h->psd, z_dev, pmin + used, psum + used);
    double a = 1.7976931348623157e308, b = 0.0; // T::max_value(), compositecone.rs:198
    if (used) {
        CHIP_HIP(hipMemcpyAsync(h->h_partial.data(), pmin, (size_t)used * sizeof(double), hipMemcpyDeviceToHost,
                                E.stream));
        CHIP_HIP(hipMemcpyAsync(h->h_partial.data() + h->partial_cap, psum, (size_t)used * sizeof(double),
                                hipMemcpyDeviceToHost, E.stream));
        CHIP_HIP(hipStreamSynchronize(E.stream));
        for (int i = 0; i < used; i++) {
            a = std::min(a, h->h_partial[i]);
            b += h->h_partial[h->partial_cap + i];
        }
    }
    *alpha_out = a;
    *beta_out = b;
    return CHIP_OK;
}
int32_t chip_kkt_info(const chip_kkt *h, chip_info *info) {
    if (!h || !info) return CHIP_ERR_ARG;
    fill_info(h->E, info);
    info->threads = h->world;
    info->last_ir_iterations = h->last_ir;
    info->last_regularizer = h->last_eps;
    if (h->E.factored && !h->E.host_only) {
        i64 c = 0;
        int rc = count_positive(const_cast<Engine &>(h->E), &c);
        if (rc) return rc;
        info->positive_inertia = c;
    }
    return CHIP_OK;
}
int32_t chip_kkt_get_perm(const chip_kkt *h, uint64_t *perm) {
    if (!h || !perm) return CHIP_ERR_ARG;
    for (int i = 0; i < h->E.N; i++) perm[i] = (uint64_t)h->E.h_perm[i];
    return CHIP_OK;
}
int32_t chip_kkt_get_symbolic(const chip_kkt *h, uint64_t *etree, uint64_t *Lp, uint64_t *Li,
                              uint64_t *lvlptr) {
    if (!h) return CHIP_ERR_ARG;
    return h->E.get_symbolic(etree, Lp, Li, lvlptr);
}
int32_t chip_kkt_get_values(chip_kkt *h, double *nzval) {
    if (!h || !nzval) return CHIP_ERR_ARG;
    Engine &E = h->E;
    NEED_DEVICE(E);
    CHIP_HIP(hipSetDevice(E.device));
    return E.download_values(nzval); // back in the caller's K.nzval order
}
int32_t chip_kkt_synchronize(chip_kkt *h) {
    if (!h) return CHIP_ERR_ARG;
    NEED_DEVICE(h->E);
    CHIP_HIP(hipStreamSynchronize(h->E.stream));
    return CHIP_OK;
}
void *chip_kkt_stream(chip_kkt *h) { return (h && !h->E.host_only) ? (void *)h->E.stream : nullptr; }
int32_t chip_kkt_profile(chip_kkt *h, int32_t family) {
    if (!h) return CHIP_ERR_ARG;
    h->E.prof_collect();
    h->E.prof_family = family;
    h->E.prof_ms_total = 0;
    h->E.prof_launches = 0;
    return CHIP_OK;
}
int32_t chip_kkt_profile_read(chip_kkt *h, double out[8]) {
    if (!h) return CHIP_ERR_ARG;
    h->E.prof_collect();
    std::memset(out, 0, 8 * sizeof(double));
    out[0] = (double)h->E.prof_launches;
    out[1] = h->E.prof_ms_total;
    out[2] = (double)h->E.prof_family;
    return CHIP_OK;
}
int32_t chip_kkt_step_kernels(const chip_kkt *h) {
    if (!h) return CHIP_ERR_ARG;
    const bool on = !switches().no_step_kernel;
    return (on && h->E.gstep_solve_on ? 1 : 0) | (on && h->E.gstep_factor_on ? 2 : 0) | (h->ir_sf && !(on && h->E.gstep_solve_on) ? 4 : 0);
}
int32_t chip_kkt_fused_fallbacks(const chip_kkt *h) { return h ? h->fused_fallbacks : CHIP_ERR_ARG; }
int32_t chip_kkt_work_model(const chip_kkt *h, double out[8]) {
    if (!h || !out) return CHIP_ERR_ARG;
    std::memcpy(out, h->E.sn_model, 8 * sizeof(double));
    out[5] = h->E.gfold.ng;    // groups of a grouped fold in use (0: none)
    out[6] = h->E.bundles.nb;  // subtree bundles
    out[7] = h->E.ir_fused ? h->E.ir_tw : 0; // threads per workgroup of the fused solve launch (0: not fused)
    return CHIP_OK;
}
int32_t chip_kkt_sweep_model(const chip_kkt *h, double out[4]) {
    if (!h || !out) return CHIP_ERR_ARG;
    const Engine &E = h->E;
    int gl = 0, sl = 0;
    for (char f : E.sn_lvl_g) gl += f != 0;
    for (size_t l = 0; l + 1 < E.sn_lvl_ptr.size(); l++) sl += E.sn_lvl_ptr[l + 1] > E.sn_lvl_ptr[l];
    out[0] = E.sn_g_ntasks > 0 ? E.sn_g_entries : 0.0;
    out[1] = E.sn_g_ntasks > 0 ? gl : 0;
    out[2] = sl;
    out[3] = E.sn_g_ntasks > 0 ? 1 : 0;
    return CHIP_OK;
}
#ifdef CHIP_TESTING
// test hooks (include/clarabel_hip_testing.h): a kernel that only spins, on a stream of its own (co-residency tests of
// the persistent launches); a switch of csrc/switches.hpp set or cleared by name
int32_t chip_debug_spin(int32_t device, int32_t blocks, int32_t threads, int32_t lds_bytes, double usec) {
    static hipStream_t spin_stream[16] = {};
    if (device < 0 || device >= 16 || blocks < 0 || threads <= 0 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024)
        return CHIP_ERR_ARG;
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Restore { // the caller's current device is left as it was
        int d;
        ~Restore() {
            if (d >= 0) (void)hipSetDevice(d);
        }
    } restore{prev};
    CHIP_HIP(hipSetDevice(device));
    if (blocks == 0) { // wait for the spinners launched so far
        if (spin_stream[device]) CHIP_HIP(hipStreamSynchronize(spin_stream[device]));
        return CHIP_OK;
    }
    if (!spin_stream[device]) CHIP_HIP(hipStreamCreateWithFlags(&spin_stream[device], hipStreamNonBlocking));
    dev::debug_spin(spin_stream[device], blocks, threads, lds_bytes, usec);
    CHIP_HIP(hipGetLastError());
    return CHIP_OK;
}
int32_t chip_debug_set_switch(const char *name, const char *value_or_null) {
    return switches_set(name, value_or_null) ? CHIP_OK : fail(CHIP_ERR_ARG, "chip_debug_set_switch: unknown switch");
}
int32_t chip_debug_counter(const void *kkt_handle, const char *name, double *out) {
    const chip_kkt *h = (const chip_kkt *)kkt_handle;
    if (!h || !name || !out) return CHIP_ERR_ARG;
    const std::string k(name);
    const Engine &E = h->E;
    if (k == "dense_blocks") *out = E.dblk.nblk;
    else if (k == "dense_block_rows") *out = E.dblk.nrows;
    else if (k == "nnzS") *out = (double)E.nnzS;
    else if (k == "psd_hs_row_blocks") *out = h->psd.rows_nblk;
    else if (k == "assembled_levels") {
        int c = 0;
        for (size_t l = 0; l + 1 < E.asm_lvl_ptr.size(); l++) c += E.asm_lvl_ptr[l + 1] > E.asm_lvl_ptr[l];
        *out = c;
    } else if (k == "assembled_targets") *out = E.asm_lvl_ptr.empty() ? 0 : E.asm_lvl_ptr.back();
    else if (k == "g_levels") { // unit levels whose supernodes use the one-pass substitution matrices (snode_g.hip)
        int c = 0;
        for (char f : E.sn_lvl_g) c += f != 0;
        *out = E.sn_g_ntasks > 0 ? c : 0;
    } else if (k == "sn_levels") {
        int c = 0;
        for (size_t l = 0; l + 1 < E.sn_lvl_ptr.size(); l++) c += E.sn_lvl_ptr[l + 1] > E.sn_lvl_ptr[l];
        *out = c;
    } else if (k == "g_entries") *out = E.sn_g_entries;
    else if (k == "gsweep_runs") *out = (double)E.gs_runs.size(); // runs of unit levels taken by one persistent launch per sweep (after the first solve)
    else if (k == "gsweep_launches") *out = E.gs_launches;
    else if (k == "gsweep_recoveries") *out = E.gs_recoveries;
    else if (k == "tri2_launches") *out = E.tri2_launches;
    else if (k == "dblk2_launches") *out = (double)E.dblk2_launches;
    else if (k == "dblk_blocks") *out = E.dblk.nblk;
    else if (k == "hs_direct_refactors") *out = (double)E.hs_direct_refactors;
    else if (k == "gsweep_levels") {
        int c = 0;
        for (const auto &r : E.gs_runs) c += r.nlev;
        *out = c;
    }
    else return fail(CHIP_ERR_ARG, "chip_debug_counter: unknown name");
    return CHIP_OK;
}
#endif

} // extern "C"
