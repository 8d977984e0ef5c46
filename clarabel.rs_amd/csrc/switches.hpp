// switches.hpp -- the CHIP_* diagnostic switches, parsed from the environment ONCE into a struct.
//
// The switches exist so that every code path stays covered by the tests (older forms of a kernel, mechanisms
// turned off one at a time) and for the profiling tools; none of them is part of the C ABI's contract.  They
// are read from the environment when a handle is created (chip_ldl_create / chip_kkt_create call
// switches_reload()) and never inside a launch loop: launchers and engine code read the parsed struct.
// Tests flip a switch either by setting the environment variable before they create a handle or, for a handle
// that already exists, through chip_debug_set_switch (include/clarabel_hip.h; CHIP_TESTING builds only).
#pragma once
#include <string>

namespace chip {

struct Switches {
    // ---- host analysis (symbolic.cpp, amd_order.cpp) ----
    bool timing = false;            // CHIP_TIMING: wall-clock of the analysis phases on stderr
    int host_threads = 0;           // CHIP_HOST_THREADS (0: min(16, hardware threads))
    long long host_par_min = 2000000; // CHIP_HOST_PAR_MIN: smallest pass that is split over threads
    bool amd_rescan = false;        // CHIP_AMD_RESCAN: every member of a new element rescans its lists
    bool no_components = false;     // CHIP_NO_COMPONENTS: order the whole graph, not one component per pattern
    bool no_clique_order = false;   // CHIP_NO_CLIQUE_ORDER: dense cone blocks enter the ordering row by row, not as one node
    bool no_bundles = false;        // CHIP_NO_BUNDLES
    long long bundle_max_work = 0;  // CHIP_BUNDLE_MAX_WORK (0: default)
    bool no_groupfold = false;      // CHIP_NO_GROUPFOLD
    long long groupfold_min = 0;    // CHIP_GROUPFOLD_MIN (0: default)
    bool has_target_wg = false;     // CHIP_TARGET_WG given
    int target_wg = 0;
    bool no_snode = false;          // CHIP_NO_SNODE
    bool no_topfold = false;        // CHIP_NO_TOPFOLD
    bool no_factor_flat = false;    // CHIP_NO_FACTOR_FLAT (also read by the launcher of the bundle factorisation)
    bool no_topblk = false;         // CHIP_NO_TOPBLK
    long long snb_chunk = 0;        // CHIP_SNB_CHUNK: fewest updates of a chunk of the bundle columns' contributions into a supernode member (0: default)
    bool no_snx_hoist = false;      // CHIP_NO_SNX_HOIST: the bundle columns' contributions into supernode members stay in the launches of the members' unit levels
    bool no_psd_mfma = false;       // CHIP_NO_PSD_MFMA: the n x n products of the PSD cone kernels as scalar dot products, not on the matrix cores
    bool no_psd_rows = false;       // CHIP_NO_PSD_ROWS: the Hs blocks of PSD cones written through mapHs (caller's order), not row by row
    long long dense_symv_min = 0;   // CHIP_DENSE_SYMV_MIN: fewest block entries for which the blocks leave S (tests; 0: 2^20)
    bool no_dense_symv = false;     // CHIP_NO_DENSE_SYMV: dense diagonal blocks of the top stay in the full rows S of the residual
    // ---- engine / launchers ----
    bool no_fused_ir = false;       // CHIP_NO_FUSED_IR: one kernel per phase, refinement control on the host
    bool no_factor_lds = false;     // CHIP_NO_FACTOR_LDS
    bool no_factor_chain = false;   // CHIP_NO_FACTOR_CHAIN
    bool no_snode_tri = false;      // CHIP_NO_SNODE_TRI
    bool no_bundle_flat_sweep = false; // CHIP_NO_BUNDLE_FLAT_SWEEP: the stand-alone bundle sweeps keep the row- / column-per-thread form (k_bundle_fwd / bwd), no entry-parallel form (k_bundle_sweep_flat)
    bool no_flat = false;           // CHIP_NO_FLAT: column-per-thread sweeps inside k_bundle_ir
    int irs_flags = -1;             // CHIP_IRS_FLAGS: experiment bits of k_bundle_irs (-1: the defaults; kernels.hpp: IrView::sf)
    bool no_ir_sf = false;          // CHIP_NO_IR_SF: the fused solve stays on k_bundle_ir also where k_bundle_irs (candidate on chip, no permuted copy of b) applies
    bool ir_test_drop = false;      // CHIP_IR_TEST_DROP (tests: a fused launch that cannot complete its barrier)
    int ir_debug = 0;               // CHIP_IR_DEBUG: 1 = stamps of two workgroups on stderr, 2 = all workgroups -> file
    std::string ir_debug_file;      // CHIP_IR_DEBUG_FILE
    bool no_step_kernel = false;    // CHIP_NO_STEP_KERNEL: the grouped-fold step kernels of round 4 off
    bool no_fast_prep = false;      // CHIP_NO_FAST_PREP: the refactor keeps its preparation launches (eps, scatter, top pivot)
    // ---- supernode kernels (snode.hip) ----
    int sn_xb_cap = 0;              // CHIP_SN_XB_CAP (0: default)
    int sn_debug = 0;               // CHIP_SN_DEBUG
    bool no_splitk = false;         // CHIP_NO_SPLITK
    bool no_snode_panel = false;    // CHIP_NO_SNODE_PANEL: separate diag / rows launches
    int sn_panel_slots = 0;         // CHIP_SN_PANEL_SLOTS: workgroups of one k_snode_panel launch beyond which a workgroup walks several
                                    // groups of 256 rows (0: 256; tests: 1 -> every workgroup walks all groups)
    bool no_panel_overlap = false;  // CHIP_NO_PANEL_OVERLAP: k_snode_panel (block factorisation, then the rows) instead of k_snode_panel2
    bool no_panel_uniform = false;  // CHIP_NO_PANEL_UNIFORM: the block team of k_snode_panel2 broadcasts pivots / coefficients by v_readlane (round 4) instead of factoring 8 x 8 sub-blocks in every lane
    bool no_panel_mfma = false;     // CHIP_NO_PANEL_MFMA
    bool no_panel_diag_mfma = false; // CHIP_NO_PANEL_DIAG_MFMA
    bool no_extend_asm = false;     // CHIP_NO_EXTEND_ASM: ancestor updates always by fp64 atomics (k_snode_extend), never assembled
    int extend_asm_min = 0;         // CHIP_EXTEND_ASM_MIN: fewest supernodes of a level whose updates are assembled (0: never, unless CHIP_DETERMINISTIC)
    bool no_dblk_pair = false;      // CHIP_NO_DBLK_PAIR: the dense diagonal blocks' residual products per vector, not one launch for a pair
    long long fill_range_min = 1 << 20; // CHIP_FILL_RANGE_MIN: fewest fill-in slots of the top from which they are cleared as a range (tests: 0 = always)
    bool no_hs_prefill_async = false; // CHIP_NO_HS_PREFILL_ASYNC: that clear on the main stream, after the scaling kernels
    bool no_hs_direct = false;      // CHIP_NO_HS_DIRECT: the PSD blocks reach L through the refactor's scatter of K, not from the kernel that writes them
    bool no_sn_wide = false;        // CHIP_NO_SN_WIDE: no 128 x 256 tiles for the ancestors' update (k_snode_extend_wide)
    int sn_wide_waves = 4;          // CHIP_SN_WIDE_WAVES: waves per workgroup of k_snode_extend_wide (4: two workgroups per CU; 8: one)
    int sn_wide_min_count = 8;      // CHIP_SN_WIDE_MIN_COUNT: supernodes of a level from which they are used (tests: 1)
    int sn_asm_cap = 0;             // CHIP_SN_ASM_CAP: rows of a target column per LDS window of k_snode_assemble (tests; 0: 4096)
    bool no_factor_overlap = false; // CHIP_NO_FACTOR_OVERLAP: the bundle columns' contributions into supernode members all ahead of the supernode chain, none beside it on the second stream
    bool no_pair_lockstep = false;  // CHIP_NO_PAIR_LOCKSTEP: the paired solves keep two independent chains of launches, also where the wide supernodes have a two-right-hand-side form (k_snode_tri<.., 2>)
    bool no_solve_pair = false;     // CHIP_NO_SOLVE_PAIR: chip_kkt_solve2_dev_enqueue runs its two solves one after the other on every handle
    bool no_sweep_merge = false;    // CHIP_NO_SWEEP_MERGE: the row gathers of a unit level in their own launch, also next to supernodes on the one-pass matrices
    bool no_sweep_persist = false;  // CHIP_NO_SWEEP_PERSIST: a launch per unit level on the one-pass matrices, no persistent launch per run of levels (k_snode_gsweep)
    bool gs_test_drop = false;      // CHIP_GS_TEST_DROP (tests: the last workgroup of a persistent sweep leaves at once, so its level barrier times out)
    int gsweep_grid = 0;            // CHIP_GSWEEP_GRID: workgroups of a persistent sweep (0: half the compute units; tests: 1, 3)
    bool no_snode_g = false;        // CHIP_NO_SNODE_G: no one-pass substitution matrices G = [I; L_B] T^-1 (snode_g.hip): every supernode keeps the pipelined substitution
    int sn_g_maxw = 0;              // CHIP_SN_G_MAXW: widest supernode that takes the G path (0: the kernels' limit, 512)
    bool deterministic = false;     // CHIP_DETERMINISTIC (read when a handle is created): no k-split of the update tiles, ancestor updates assembled in a fixed order
};

// the parsed switches (first call parses the environment)
const Switches &switches();
// read the environment again (called when a handle is created)
void switches_reload();
// set (value != nullptr) or clear one switch by its environment name and re-parse; false: unknown name
bool switches_set(const char *name, const char *value);

} // namespace chip
