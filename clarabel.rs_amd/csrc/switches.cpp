// switches.cpp -- see switches.hpp
#include "switches.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>

namespace chip {
namespace {

struct Entry {
    const char *name;
    enum Kind { FLAG, INT, LONG, STR } kind;
    size_t off;
    size_t given_off; // FLAG-typed "was given" companion of an INT entry, or 0
};
#define SW(field) offsetof(Switches, field)
const Entry TABLE[] = {
    {"CHIP_TIMING", Entry::FLAG, SW(timing), 0},
    {"CHIP_HOST_THREADS", Entry::INT, SW(host_threads), 0},
    {"CHIP_HOST_PAR_MIN", Entry::LONG, SW(host_par_min), 0},
    {"CHIP_AMD_RESCAN", Entry::FLAG, SW(amd_rescan), 0},
    {"CHIP_NO_COMPONENTS", Entry::FLAG, SW(no_components), 0},
    {"CHIP_NO_CLIQUE_ORDER", Entry::FLAG, SW(no_clique_order), 0},
    {"CHIP_NO_BUNDLES", Entry::FLAG, SW(no_bundles), 0},
    {"CHIP_BUNDLE_MAX_WORK", Entry::LONG, SW(bundle_max_work), 0},
    {"CHIP_SNB_CHUNK", Entry::LONG, SW(snb_chunk), 0},
    {"CHIP_NO_GROUPFOLD", Entry::FLAG, SW(no_groupfold), 0},
    {"CHIP_GROUPFOLD_MIN", Entry::LONG, SW(groupfold_min), 0},
    {"CHIP_TARGET_WG", Entry::INT, SW(target_wg), SW(has_target_wg)},
    {"CHIP_NO_SNODE", Entry::FLAG, SW(no_snode), 0},
    {"CHIP_NO_TOPFOLD", Entry::FLAG, SW(no_topfold), 0},
    {"CHIP_NO_FACTOR_FLAT", Entry::FLAG, SW(no_factor_flat), 0},
    {"CHIP_NO_TOPBLK", Entry::FLAG, SW(no_topblk), 0},
    {"CHIP_NO_SNX_HOIST", Entry::FLAG, SW(no_snx_hoist), 0},
    {"CHIP_NO_PSD_MFMA", Entry::FLAG, SW(no_psd_mfma), 0},
    {"CHIP_NO_PSD_ROWS", Entry::FLAG, SW(no_psd_rows), 0},
    {"CHIP_NO_DENSE_SYMV", Entry::FLAG, SW(no_dense_symv), 0},
    {"CHIP_DENSE_SYMV_MIN", Entry::LONG, SW(dense_symv_min), 0},
    {"CHIP_NO_FUSED_IR", Entry::FLAG, SW(no_fused_ir), 0},
    {"CHIP_NO_FACTOR_LDS", Entry::FLAG, SW(no_factor_lds), 0},
    {"CHIP_NO_FACTOR_CHAIN", Entry::FLAG, SW(no_factor_chain), 0},
    {"CHIP_NO_SNODE_TRI", Entry::FLAG, SW(no_snode_tri), 0},
    {"CHIP_NO_BUNDLE_FLAT_SWEEP", Entry::FLAG, SW(no_bundle_flat_sweep), 0},
    {"CHIP_NO_FLAT", Entry::FLAG, SW(no_flat), 0},
    {"CHIP_NO_IR_SF", Entry::FLAG, SW(no_ir_sf), 0},
    {"CHIP_IRS_FLAGS", Entry::INT, SW(irs_flags), 0},
    {"CHIP_IR_TEST_DROP", Entry::FLAG, SW(ir_test_drop), 0},
    {"CHIP_IR_DEBUG", Entry::INT, SW(ir_debug), 0},
    {"CHIP_IR_DEBUG_FILE", Entry::STR, SW(ir_debug_file), 0},
    {"CHIP_NO_STEP_KERNEL", Entry::FLAG, SW(no_step_kernel), 0},
    {"CHIP_NO_FAST_PREP", Entry::FLAG, SW(no_fast_prep), 0},
    {"CHIP_SN_XB_CAP", Entry::INT, SW(sn_xb_cap), 0},
    {"CHIP_SN_DEBUG", Entry::INT, SW(sn_debug), 0},
    {"CHIP_NO_SPLITK", Entry::FLAG, SW(no_splitk), 0},
    {"CHIP_NO_SNODE_PANEL", Entry::FLAG, SW(no_snode_panel), 0},
    {"CHIP_SN_PANEL_SLOTS", Entry::INT, SW(sn_panel_slots), 0},
    {"CHIP_NO_PANEL_OVERLAP", Entry::FLAG, SW(no_panel_overlap), 0},
    {"CHIP_NO_PANEL_UNIFORM", Entry::FLAG, SW(no_panel_uniform), 0},
    {"CHIP_NO_PANEL_MFMA", Entry::FLAG, SW(no_panel_mfma), 0},
    {"CHIP_NO_PANEL_DIAG_MFMA", Entry::FLAG, SW(no_panel_diag_mfma), 0},
    {"CHIP_NO_EXTEND_ASM", Entry::FLAG, SW(no_extend_asm), 0},
    {"CHIP_EXTEND_ASM_MIN", Entry::INT, SW(extend_asm_min), 0},
    {"CHIP_NO_DBLK_PAIR", Entry::FLAG, SW(no_dblk_pair), 0},
    {"CHIP_FILL_RANGE_MIN", Entry::LONG, SW(fill_range_min), 0},
    {"CHIP_NO_HS_PREFILL_ASYNC", Entry::FLAG, SW(no_hs_prefill_async), 0},
    {"CHIP_NO_HS_DIRECT", Entry::FLAG, SW(no_hs_direct), 0},
    {"CHIP_NO_SN_WIDE", Entry::FLAG, SW(no_sn_wide), 0},
    {"CHIP_SN_WIDE_MIN_COUNT", Entry::INT, SW(sn_wide_min_count), 0},
    {"CHIP_SN_WIDE_WAVES", Entry::INT, SW(sn_wide_waves), 0},
    {"CHIP_SN_ASM_CAP", Entry::INT, SW(sn_asm_cap), 0},
    {"CHIP_NO_FACTOR_OVERLAP", Entry::FLAG, SW(no_factor_overlap), 0},
    {"CHIP_NO_SOLVE_PAIR", Entry::FLAG, SW(no_solve_pair), 0},
    {"CHIP_NO_PAIR_LOCKSTEP", Entry::FLAG, SW(no_pair_lockstep), 0},
    {"CHIP_NO_SWEEP_MERGE", Entry::FLAG, SW(no_sweep_merge), 0},
    {"CHIP_NO_SWEEP_PERSIST", Entry::FLAG, SW(no_sweep_persist), 0},
    {"CHIP_GS_TEST_DROP", Entry::FLAG, SW(gs_test_drop), 0},
    {"CHIP_GSWEEP_GRID", Entry::INT, SW(gsweep_grid), 0},
    {"CHIP_NO_SNODE_G", Entry::FLAG, SW(no_snode_g), 0},
    {"CHIP_SN_G_MAXW", Entry::INT, SW(sn_g_maxw), 0},
    {"CHIP_DETERMINISTIC", Entry::FLAG, SW(deterministic), 0},
};
#undef SW

// Handles may live on several threads: a reader gets a pointer to an IMMUTABLE snapshot; a reload (every handle
// creation, test hooks) parses into a new snapshot and publishes it with one atomic store.  Old snapshots stay
// alive (a deque never moves its elements; a snapshot is ~300 bytes per handle creation), so a reference taken
// before a reload remains valid.
std::deque<Switches> g_snapshots;
std::atomic<const Switches *> g_cur{nullptr};
std::mutex g_mu;

void parse_locked() {
    Switches s; // defaults
    char *base = reinterpret_cast<char *>(&s);
    for (const Entry &e : TABLE) {
        const char *v = std::getenv(e.name);
        if (!v) continue;
        switch (e.kind) {
        case Entry::FLAG: *reinterpret_cast<bool *>(base + e.off) = true; break;
        case Entry::INT: *reinterpret_cast<int *>(base + e.off) = std::atoi(v); break;
        case Entry::LONG: *reinterpret_cast<long long *>(base + e.off) = std::atoll(v); break;
        case Entry::STR: *reinterpret_cast<std::string *>(base + e.off) = v; break;
        }
        if (e.given_off) *reinterpret_cast<bool *>(base + e.given_off) = true;
    }
    g_snapshots.push_back(std::move(s));
    g_cur.store(&g_snapshots.back(), std::memory_order_release);
}

} // namespace

const Switches &switches() {
    const Switches *p = g_cur.load(std::memory_order_acquire);
    if (!p) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_cur.load(std::memory_order_relaxed)) parse_locked();
        p = g_cur.load(std::memory_order_relaxed);
    }
    return *p;
}
void switches_reload() {
    std::lock_guard<std::mutex> lk(g_mu);
    parse_locked();
}
bool switches_set(const char *name, const char *value) {
    if (!name) return false;
    bool known = false;
    for (const Entry &e : TABLE) known = known || std::strcmp(e.name, name) == 0;
    if (!known) return false;
    if (value) setenv(name, value, 1);
    else unsetenv(name);
    switches_reload();
    return true;
}

} // namespace chip
