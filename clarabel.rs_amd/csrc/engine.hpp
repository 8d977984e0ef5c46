// engine.hpp -- device-resident LDL' engine: owns the HBM copies of the symbolic
// structures and values and enqueues the level-scheduled kernel sequences.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/clarabel_hip.h"
#include "host.hpp"
#include "kernels.hpp"

namespace chip {

struct DeviceLists {
    std::vector<i32> t_ptr, w_ptr, b_ptr, br_ptr; // host copies of the level pointers
    int *t_idx = nullptr, *w_idx = nullptr, *b_row = nullptr, *b_beg = nullptr, *b_end = nullptr,
        *br_idx = nullptr;
    int *d_t_ptr = nullptr, *d_w_ptr = nullptr; // device copies of the level pointers (chain kernel)
    // chain_end[l] > l + 1: levels [l, chain_end[l]) form a run of narrow levels handled by one
    // single-workgroup launch; otherwise l + 1
    std::vector<i32> chain_end, chain_begin;
    dev::ListView T(int l) const { return {t_idx + t_ptr[l], t_ptr[l + 1] - t_ptr[l]}; }
    dev::ListView W(int l) const { return {w_idx + w_ptr[l], w_ptr[l + 1] - w_ptr[l]}; }
    dev::ListView BR(int l) const { return {br_idx + br_ptr[l], br_ptr[l + 1] - br_ptr[l]}; }
    dev::ChunkView B(int l) const {
        return {b_row + b_ptr[l], b_beg + b_ptr[l], b_end + b_ptr[l], b_ptr[l + 1] - b_ptr[l]};
    }
    dev::ChunkView Brange(int l0, int l1) const { // the chunks of levels [l0, l1): contiguous
        return {b_row + b_ptr[l0], b_beg + b_ptr[l0], b_end + b_ptr[l0], b_ptr[l1] - b_ptr[l0]};
    }
};

// small device<->host mailbox (one 256-byte D2H copy per decision point)
struct Mailbox {
    double eps;                  // static regulariser of the last refactor
    unsigned long long tmpmax;   // scratch of diag_absmax_eps
    unsigned long long tmpnan;
    int nan[16];                 // NaN sightings per norm set (set 0 = ||b||, 1.. = residual rounds)
    int status[4];               // non-finite pivot, zero pivot, regularize_count, (unused)
    int soc_fail;                // generation (see chip_kkt::scaling_gen) of the last failed cone scaling, 0 = none
    int pad[9];
    int ring[64];                // verdict quads of the last 16 fused solves (Engine::ir_res points here): the
                                 // refactor status and the solves' verdicts travel in ONE device-to-host copy
};
constexpr int NRM_SETS = 16;
// u64 words per set: NRM_SLOTS slotted maxima (one per 128-byte line) + one line for the NaN flag
constexpr int NRM_SET_WORDS = (dev::NRM_SLOTS + 1) * dev::NRM_STRIDE;
static_assert(sizeof(Mailbox) <= 512, "mailbox");

// profile families (hipEvent pairs around each launch of ONE selected family)
enum ProfFamily {
    PF_NONE = 0, PF_SYMV_T = 1, PF_BWD_T = 2, PF_FWD_T = 3, PF_FACTOR_T = 4, PF_IR = 5, PF_BFACTOR = 6,
    // supernode kernels (ids shared with the launchers in snode.hip: dev::PFK_*)
    PF_SN_UPDATE = dev::PFK_SN_UPDATE, PF_SN_DIAG = dev::PFK_SN_DIAG, PF_SN_ROWS = dev::PFK_SN_ROWS,
    PF_SN_EXTEND = dev::PFK_SN_EXTEND, PF_SN_TRI = dev::PFK_SN_TRI,
    PF_SN_GATHER = 12, // k_gather_merged launches of the supernode substitution path
    PF_COUNT
};

struct Engine {
    int device = 0;
    hipStream_t stream = nullptr;
    chip_settings st{};
    int N = 0, nlevels = 0;
    i64 nnzK = 0, nnzL = 0, nnzS = 0;
    i64 nnzR = 0;
    // symbolic (device)
    int *v2l = nullptr, *Lp = nullptr, *Li = nullptr, *Rp = nullptr, *Rcol = nullptr, *Rpos = nullptr,
        *Tpos = nullptr, *perm = nullptr, *iperm = nullptr, *Sp = nullptr, *Scol = nullptr, *Smap = nullptr,
        *Up = nullptr, *Ucol = nullptr;
    i64 nnzU = 0;
    double *Ux = nullptr; // == Kx: the U rows are the first nnzU entries of the value store
    // Kx holds the caller's K.nzval in T order V (host.hpp: Symbolic::k2v): h_k2v[p] = position of K.nzval[p];
    // d_v2k (device, allocated on first use) serves wholesale uploads in the caller's order (L1 boundary)
    bigvec h_k2v, h_v2k;
    int *d_v2k = nullptr;
    double *d_stage = nullptr;
    int8_t *dsigns = nullptr;
    // values (device)
    double *Kx = nullptr, *Lx = nullptr, *Rx = nullptr, *D = nullptr, *Dinv = nullptr, *Sx = nullptr;
    DeviceLists fac, fwd, bwd, smv;
    // chain supernodes (host.hpp: Symbolic::sn_*): factor lists `fac` and `snx` are indexed by UNIT level
    int nfaclevels = 0, nsn = 0, sn_wmax = 0;
    DeviceLists snx, snb, fwu, bwu;
    // snb beside the supernode chain (refactor_enqueue): groups of unit levels [first, second), the event each group's launch
    // on the second stream records, the group that starts at a level (-1: none)
    std::vector<std::pair<int, int>> snb_groups;
    std::vector<hipEvent_t> snb_events;
    std::vector<i32> snb_group_at;
    // Dense diagonal blocks of the top whose rows are CONTIGUOUS in L as well (row a of the stored upper triangle = column
    // a of L's panel, entry for entry): the kernel that writes such a block's values into K writes them into L too
    // (dblk_l0[row] = L position of the row's first entry), and the refactor scatters only the REST of K's top entries
    // (rest_idx) instead of reading all of them back -- config 5: 1.6e8 of 1.6e8.  hs_direct_begin() clears the fill-in range
    // ahead of that kernel and arms the next refactor_enqueue().
    int *dblk_l0 = nullptr, *rest_idx = nullptr;
    int nrest = 0;
    bool hs_direct_ok = false, hs_direct_armed = false;
    long long hs_direct_refactors = 0;
    bool hs_direct_begin();
    // the same clear started EARLY on the second stream (beside the cones' scaling kernels, which do not touch L): the main
    // stream only waits for it in hs_direct_begin()
    void hs_direct_prefill_async();
    hipEvent_t hs_ev[2] = {nullptr, nullptr};
    bool hs_prefill_pending = false;
    hipEvent_t snb_ready = nullptr;
    double *Rfx = nullptr; // values of L at the filtered row lists (refreshed per refactor)
    int nRf = 0, sn_nbmax = 0;
    int *sn_geo = nullptr, *sn_cb = nullptr, *sn_ptr = nullptr, *sn_col = nullptr, *sn_order = nullptr, *Rf_p = nullptr, *Rf_col = nullptr,
        *Rf_pos = nullptr, *upd_slot = nullptr;
    long long *upd_ptr = nullptr;
    double *sn_d = nullptr; // pivots of the supernode members, packed (dev::SnodeView::sn_d)
    int *sn_cnt = nullptr;  // dev::SnodeView::sn_cnt
    dev::DblkView dblk;       // dense diagonal blocks of the top in the residual (host.hpp: Symbolic::dblk_*)
    std::vector<i32> h_dblk_node, h_dblk_m; // (host copies: the cone layer matches its Hs blocks against them)
    double *bt_view = nullptr; // b minus the blocks' products (enqueue_residual)
    // ancestor updates assembled per target column (dev::SnodeAsmView; host.hpp: Symbolic::asm_*)
    int *asm_tgt = nullptr, *asm_src_ptr = nullptr, *asm_doff = nullptr;
    long long *asm_src = nullptr, *asm_uoff = nullptr;
    double *asm_U = nullptr, *asm_Ud = nullptr;
    std::vector<i32> asm_lvl_ptr;
    int8_t *sn_sg = nullptr; // dev::SnodeView::sn_sg
    // one-pass substitution matrices of the supernodes of moderate width (snode_g.hip): per unit level whether its
    // supernodes take the path (all of them or none), the matrices, the build's task list, the forward sweep's output
    std::vector<char> sn_lvl_g;
    std::vector<i32> sn_lvl_wmax;
    double *sn_Gx = nullptr, *sn_yt = nullptr;
    long long *sn_g_off = nullptr;
    int *sn_g_tasks = nullptr;
    int sn_g_ntasks = 0;
    double sn_g_entries = 0; // doubles of G (what one sweep through these supernodes streams)
    // runs of consecutive unit levels on that path as ONE persistent launch per sweep (k_snode_gsweep); built on the first solve
    struct GRun {
        int nlev, off, grid; // levels of the run (sweep order), first entry in gs_lv, co-resident grid
        size_t lds;
    };
    std::vector<GRun> gs_runs;
    std::vector<i32> gs_run_f, gs_run_b; // per unit level: the run that STARTS there in the forward / backward sweep, or -1
    dev::GSweepLevel *gs_lv = nullptr;
    int *gs_ctl = nullptr; // the grid barrier's counters (per solve context)
    bool gs_built = false;
    int gs_launches = 0; // (tests: persistent launches enqueued so far)
    // A persistent sweep whose level barrier timed out (a launch that was not co-resident) raises the solve's
    // non-finite flag and leaves the barrier words dirty.  After a solve that ended non-finite: clear the words of both
    // solve contexts and keep this handle to the per-level launches (gs_off); true: the caller may repeat the solve.
    bool gs_off = false;
    int gs_recoveries = 0; // (tests)
    bool sweeps_after_failure();
    int build_gsweeps();
    std::vector<i32> sn_lvl_ptr, sn_lvl_nblk, sn_lvl_hmax, sn_lvl_nbmax, h_sn_ptr, h_sn_col;
    // pipelined substitution through wide supernodes (dev::SnodeTriView): one flag per 64-column block
    int *sn_blk_ptr = nullptr, *sn_flags = nullptr;
    int *xperm = nullptr;      // supernode-contiguous order of the nodes (residual over the top rows)
    double *xs_view = nullptr; // x in that order
    int sn_epoch = 0;
    dev::BundleView bundles{}; // subtree bundles (device arrays)
    dev::FoldView fold{};      // few dense top rows folded into the bundle kernels (k == 0: not used)
    dev::GFoldView gfold{};    // grouped fold (a forest of small trees with tops of <= 8 nodes; ng == 0: not used):
                               // active only on handles whose solve is the fused launch
    size_t ir_ctl_len = 0;     // ints of ir_ctl (the grid barrier's counters + one line per group of a grouped fold)
    dev::TopBlkView topblk{};  // blocked substitution of a tall top (nblocks == 0: level-scheduled top)
    int NF = 0, tree_depth = 0;
    std::vector<i32> h_level;
    Mailbox *mb_dev = nullptr, *mb_host = nullptr;
    unsigned long long *nrm_dev = nullptr, *nrm_host = nullptr; // NRM_SETS slotted inf-norm accumulators
    int *fill_idx = nullptr;
    int nfill = 0;
    long long fill_from = -1; // >= 0: Lx[fill_from .. nnzL) is cleared as a range before K's entries are scattered (host.hpp)
    std::vector<i32> h_perm, h_lvlptr, h_etree;
    bool host_only = false;          // CHIP_DEVICE_HOST_ONLY: symbolic results only
    std::vector<i32> h_Lp;           // kept only for host-only handles
    bigvec h_Li;
    AmdInfo amd;
    bool factored = false;
    i64 last_regularize_count = 0;
    std::vector<void *> allocs;
    // settings.use_graph: the launch sequence of one LDL' solve captured once per (vector, addend)
    // pair and replayed as a hipGraph (hundreds of launches for a tall top)
    struct SolveGraph {
        double *xp;
        const double *addv;
        hipGraphExec_t exec;
    };
    std::vector<SolveGraph> graphs;
    // fused solve + refinement (dev::bundle_ir): available when the system is subtree bundles plus at most a
    // folded top, and all its workgroups are co-resident
    static constexpr int IR_RING = 16; // result quads of the last IR_RING enqueued solves
    bool ir_fused = false;
    // who reads Rx: the row-per-thread forward sweep of the bundles, and tops without chain supernodes (their row gathers,
    // the blocked substitution); a top with supernodes gathers from its filtered copy Rfx
    bool rx_needed() const {
        const bool flat_sweeps = sLi16 && bundles.max_levels <= 64 && !fold.k && !switches().no_bundle_flat_sweep && !switches().deterministic;
        return !(nsn > 0 && flat_sweeps);
    }
    bool rx_valid = false; // fused handles: the row-major copy Rx of L is only refreshed when the one-kernel-per-phase path runs
    bool sx_valid = false; // grouped fold: the full rows Sx of the top likewise (the fused launch reads K's own values)
    unsigned short *Li16 = nullptr, *Ucol16 = nullptr, *Lj16 = nullptr, *Urow16 = nullptr, *Rk16 = nullptr, *Ro16 = nullptr;
    unsigned short *sLi16 = nullptr, *sLj16 = nullptr; // dev::LdlView::sLi16
    unsigned short *fu_rec = nullptr, *fu_slot = nullptr;
    int *fu_ptr = nullptr;
    // grouped fold with small bundles: the step kernels (dev::gstep_solve / gstep_factor); gstep.desc == nullptr: not used
    dev::GStepView gstep{};
    bool gstep_solve_on = false, gstep_factor_on = false;
    bool gstep_vals_valid = false; // the last refactor went through k_gstep_factor: gstep.gsl / gsu are current
    int factor_lds_doubles = 0; // > 0: the bundle factorisation keeps its values in LDS (k_bundle_factor_lds)
    int ir_grid = 0, ir_next = 0, ir_tw = 256;
    int *ir_ctl = nullptr, *ir_res = nullptr, *ir_res_host = nullptr;
    int *ir_rel = nullptr;     // k_bundle_irs: release records of the grid barrier (dev::ir_rel_ints())
    int ir_epoch = 0;          // launches of k_bundle_irs so far (tags of the messages)
    double *ir_part = nullptr;
    // profiling
    int prof_family = PF_NONE;
    std::vector<hipEvent_t> prof_events; // pairs
    size_t prof_used = 0;
    double prof_ms_total = 0;
    i64 prof_launches = 0;

    ~Engine();
    int init(const Symbolic &S, const chip_settings &settings);
    void init_host_only(const Symbolic &S, const chip_settings &settings);
    int get_symbolic(uint64_t *etree, uint64_t *Lp, uint64_t *Li, uint64_t *lvlptr) const;
    template <typename T, typename A> int upload(T **dst, const std::vector<T, A> &src, size_t n);
    template <typename T> int alloc(T **dst, size_t n);
    // chain_max_w: most W rows a level may have to count as "narrow" (runs of narrow levels are chained)
    int upload_lists(DeviceLists &D, const LevelLists &L, int chain_max_w = 64);

    dev::LdlView view() const;
    // enqueue: (optional static regularisation) -> scatter -> level-scheduled factor
    // -> refresh of the symv values; then reads the status mailbox (one sync).
    // returns 1 ok / 0 numerical failure / <0 error.  static_reg: eps = c + prop * max|diag K| where the
    // maximum over the diagonal is either reduced here (diag_idx_dev = positions of the N diagonal entries in
    // Kx) or, with diag_idx_dev == nullptr, taken from the slotted maxima the cone kernels left in
    // diag_slots() combined with static_diag_max (the part of the diagonal no cone kernel writes)
    int refactor(bool static_reg, const int *diag_idx_dev, double static_diag_max = 0.0);
    // status_cleared: the caller's cone kernel has cleared the status words (dev::sym_scale_write); together with
    // fast_prep_ok and slotted maxima this skips the preparation launches (eps, scatter) -- "fast preparation"
    int refactor_enqueue(bool static_reg, const int *diag_idx_dev, double static_diag_max, bool status_cleared = false);
    int refactor_collect();
    // the set of slots the cone kernels of the NEXT update write and its refactor reads (two sets: with the fast
    // preparation the bundle factorisation itself reduces the current set and clears the other one)
    unsigned long long *diag_slots() const { return dslot_dev + (size_t)slot_parity * NRM_SET_WORDS; }
    unsigned long long *dslot_dev = nullptr; // 2 x (NRM_SLOTS slotted maxima of |diag| + one NaN flag line)
    int slot_parity = 0;
    bool fast_prep_ok = false; // the bundle factorisation can do the preparation launches' work itself (engine.cpp)
    int *fold_cnt = nullptr;   // arrival counter of the flat bundle factorisation's last-arriver top pivot
    int h_top_sign = 1;        // Dsigns of the single folded top column
    // values in the caller's order <-> the device's T order
    int upload_values(const double *host_nzval);
    int download_values(double *host_nzval);
    // xp <- K^-1 xp (permuted numbering); with addv the result is xp <- K^-1 xp + addv
    void enqueue_solve_inplace(double *xp, const double *addv = nullptr);
    void enqueue_solve_direct(double *xp, const double *addv);
    // two independent solves (context A = the active one, context B = alt; pair_begin() has run): systems whose wide chain
    // supernodes have a form for two right-hand sides walk the levels ONCE, every other stage twice (a launch per context
    // on its own stream), the wide levels as one launch that streams the panels for both vectors; everything else: two chains
    void enqueue_solve_pair(double *xa, const double *addva, double *xb, const double *addvb);
    bool pair_lockstep_ok();
    int tri2_launches = 0; // (tests: two-right-hand-side launches of k_snode_tri so far)
    // e = b - K x (permuted numbering); ||e||inf is folded into norm set `set` (>= 0)
    void enqueue_residual(double *e, const double *b, const double *x, int set, int phases = 7);
    void enqueue_residual_pair(double *eA, const double *bA, const double *xA, double *eB, const double *bB, const double *xB, int set);
    bool dblk_pair_ok = false; // k_dblk_symv<2> fits the LDS
    long long dblk2_launches = 0;
    int zero_norm_sets();                                                    // enqueue
    unsigned long long *norm_set(int set) const { return nrm_dev + (size_t)set * NRM_SET_WORDS; }
    int *norm_nan(int set) const { return (int *)(norm_set(set) + dev::NRM_SLOTS * dev::NRM_STRIDE); }
    int read_norm(int set, double *out);                                     // D2H + sync; NaN propagating
    int read_norms(int first, int count, double *out);                       // `count` <= 3 contiguous sets, ONE D2H copy
    int read_mailbox();                                                      // D2H + sync
    void prof_begin(int family);
    void prof_end(int family);
    void prof_pair(int family, hipEvent_t *ev0, hipEvent_t *ev1);
    void prof_collect();
    dev::SnodeView snode_view() const;
    // ---- a second solve context (round 5): two INDEPENDENT solves of an interior-point iteration (the constant right-hand side
    // of kktsystem.rs:108-125 and the affine direction of core/solver.rs:351-361) enqueued on two streams.  On systems whose
    // top is level-scheduled a sweep is a chain of small launches that leave most of the chip idle; two such chains overlap.
    // Everything a solve writes besides its own vectors lives here and is swapped with the engine's members by swap_ctx().
    struct SolveCtx {
        hipStream_t stream = nullptr;
        double *sn_yt = nullptr, *xs_view = nullptr, *bt_view = nullptr, *dblk_P = nullptr;
        unsigned long long *nrm_dev = nullptr, *nrm_host = nullptr;
        int *sn_flags = nullptr, *gs_ctl = nullptr;
    };
    SolveCtx alt;
    bool alt_ready = false, alt_active = false;
    bool alt_failed = false; // the second solve context could not be set up once: not tried again
    // the exchange of the sharded path (comm.cpp) as an event of this handle: a persistent launch (the fused solve, the
    // persistent sweeps) needs every workgroup co-resident and must not start beside the collective's kernels
    hipEvent_t exch_event = nullptr;
    bool exch_pending = false;
    int wait_for_exchange();
    hipEvent_t pair_event = nullptr, pair_ev_a = nullptr, pair_ev_b = nullptr;
    bool pair_ok() const { return !ir_fused && fold.k == 0 && gfold.ng == 0 && topblk.nblocks == 0; } // (no shared accumulators)
    int ensure_alt();
    int ensure_alt_once();
    void swap_ctx();
    // makes the second stream wait for everything enqueued on the first one so far (the refactor), after the value mirrors a
    // solve reads lazily (Rx, Sx) have been refreshed on the first stream
    int pair_begin();
    dev::LaunchProf launch_prof(); // hook handed to the launchers in snode.hip (nullptr-equivalent when off)
    // work model of the chain supernodes, per refactor / per sweep (host.hpp: Symbolic::sn_*), for the roofline
    // figures of bench.py: [0] multiply-add flops of k_snode_update (2 per multiply-add, useful part of the
    // tiles: rows >= the block's first row), [1] entries of the dense trapezoids (what one sweep streams),
    // [2] flops of k_snode_extend, [3] flops of k_snode_diag + k_snode_rows, [4] #supernodes
    double sn_model[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

std::string hip_err(hipError_t e, const char *what);
// for the L3 layer (kktsystem.cpp), which otherwise only uses the public chip_kkt_* entry points
int kkt_device(const ::chip_kkt *h);
bool kkt_host_only(const ::chip_kkt *h);

#define CHIP_HIP(expr)                                                   \
    do {                                                                 \
        hipError_t _e = (expr);                                          \
        if (_e != hipSuccess) {                                          \
            chip::set_error(chip::hip_err(_e, #expr));                   \
            return CHIP_ERR_HIP;                                         \
        }                                                                \
    } while (0)

} // namespace chip
