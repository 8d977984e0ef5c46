// kernels.hpp -- launch wrappers of the hand-written gfx950 kernels (algebra / bundle_factor / bundle_solve / bundle_ir / snode / cones .hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace chip {
namespace dev {

// device-side view of the symbolic structures (all int32 / fp64, HBM resident)
struct LdlView {
    int N;
    int nnzL;
    const int *Lp, *Li;          // L by columns, ascending rows
    const int *Rp, *Rcol, *Rpos; // L by rows, Rpos = CSC slot of the entry
    const int *Tpos;             // CSC slot -> CSR slot
    double *Lx, *Rx;             // values in CSC / CSR order
    double *D, *Dinv;
    const int8_t *dsigns;
    int *status; // [0]=non-finite pivot seen, [1]=zero pivot seen, [2]=regularize_count, [3]=positive inertia
    double reg_eps, reg_delta;
    // K by rows of the smaller index for the bundle nodes (row i = diagonal first, then the entries to
    // ancestors ascending): the bundle factorisation takes the initial values of column i from here
    const int *Up, *Ucol;
    const double *Ux;
    const double *eps_ptr; // static regulariser (device scalar) applied to the diagonal while it is read; nullptr: none
    // ... or, eps_slots != nullptr (the refactor's "fast preparation": no eps / scatter launches ahead of the bundle
    // factorisation), computed by every workgroup itself from the slotted maxima of |diag K| the cone kernels left:
    // eps = eps_c + eps_prop * max(eps_static_max, slots); workgroup 0 stores it to eps_out and clears the OTHER set
    // of slots (eps_clear) for the next update
    const unsigned long long *eps_slots;
    unsigned long long *eps_clear;
    double eps_c, eps_prop, eps_static_max;
    double *eps_out;
    // bundles + folded top only (else nullptr): 16-bit bundle-local row indices parallel to Li / Ucol for the
    // bundle part (host.hpp: Symbolic::Li16); mirror_rows: the factorisation keeps the row-major copy Rx
    // up to date (the fused solve kernel does not read it)
    const unsigned short *Li16, *Ucol16;
    const unsigned short *Lj16, *Urow16; // bundle-local column of an L entry / row of a U entry (flat sweeps)
    const unsigned short *Rk16, *Ro16;   // rows of L inside the bundles: bundle-local column, offset inside that column
    const unsigned short *fu_rec, *fu_slot; // entry-parallel bundle factorisation (host.hpp: Symbolic::fu_rec), or nullptr
    const int *fu_ptr;
    int mirror_rows;
    // systems with a level-scheduled top: bundle-local row (0xFFFF: a top row) / column of the entries of the bundle columns,
    // for the entry-parallel stand-alone sweeps (k_bundle_sweep_flat; host.hpp: Symbolic::sLi16), or nullptr
    const unsigned short *sLi16, *sLj16;
};

// subtree bundles: bundle b = nodes [bundle_ptr[b], bundle_ptr[b+1]); its level boundaries are
// blvl[blvl_ptr[b] .. blvl_ptr[b+1])
struct BundleView {
    int nb;
    const int *bundle_ptr, *blvl_ptr, *blvl;
    int max_nodes;
    int max_entries; // most entries of L in one bundle's columns (the step kernels' LDS layout)
    // k_bundle_ir only: doubles of dynamic LDS per workgroup (0: max_nodes) and whether the residual runs in its
    // "split" form (bundle_symv_split: x of the non-leaf nodes AND the residual in LDS, no gathers from global
    // memory), which needs nloc + max(0, nloc - 2 nleaf) doubles for every bundle
    int ir_lds_doubles, symv_split;
    int max_levels; // most levels of a bundle
    long long *fdbg; // diagnostics (CHIP_IR_DEBUG=3): 32 words per workgroup of k_bundle_factor_flat ([0] hardware id, [1..] stamps), or nullptr
};

// Few dense top rows folded into the bundle kernels (host.hpp: Symbolic::nfold); k == 0: unused
struct FoldView {
    int k, NF;
    const int *rseg;               // (bundle, top row) -> CSR slot range of L
    const int *tt;                 // k x k CSC slots of the top-top entries of L
    const int *sp, *scol, *sslot;  // top-top entries of K (slots in Sx)
    // slotted accumulators (zero between uses): 3 kinds (0 forward sweep, 1 residual, 2 pivot) x 8 top rows
    // x FOLD_SLOTS slots, one 128-byte line per slot -- a thousand bundles adding to ONE address would
    // serialise at ~13 ns each right at the tail of the launch
    double *acc;
    // fast preparation, k == 1: the top's diagonal entry of K (its initial pivot; sign top_sign) is read by the LAST
    // workgroup of the bundle factorisation to arrive at cnt, which then applies the pivot rule itself
    const double *top_k;
    int *cnt;
    int top_sign;
};
// GROUPED fold (host.hpp: Symbolic::gf_*): a forest of small trees, each with a top of at most 8 nodes that is
// folded into the bundle kernels of ITS tree.  ng == 0: unused.  Only k_bundle_ir, k_bundle_factor and
// k_gfold_top_factor know about groups; every other kernel sees an ordinary level-scheduled top.
struct GFoldView {
    int ng;
    const int *ptr, *node;        // group g: top nodes node[ptr[g] .. ptr[g+1]) (ascending = topological)
    const int *bptr;              // its bundles [bptr[g], bptr[g+1]); bundles >= bptr[ng] have no top
    const int *bgrp;              // group of every bundle, -1 = none
    const int *tt;                // [g*64 + i*8 + j]: CSC slot of L(top_i, top_j), i > j, -1 = structurally zero
    const int *sp, *scol, *sslot; // K entries among a group's top rows (row index ptr[g] + i): column inside the group, position in V
    double *fsh;                  // [nb*8]    forward sweep: the bundles' shares of the top rows
    double *rsh;                  // [2][nb*8] residual: the bundles' shares of (K x)[top]
    double *rec;                  // [ng][2][32] published by a group's last arriver: [0,8) dx_top, [8,16) candidate x_top,
                                  //             [16,24) residual of the top rows, [24] its max |.|, [25] max |b_top|
    int *gcnt;                    // [ng*32] arrival counters, one 128-byte line per group, zero between launches
    double *fac;                  // [nb*36] factorisation: a bundle's contribution to the Schur complement of its group's
                                  //             top, packed lower triangle (i >= j at i (i + 1) / 2 + j)
};
constexpr int FOLD_SLOTS = 16, FOLD_STRIDE = 16;
inline __host__ __device__ int fold_acc_index(int kind, int row, int slot) {
    return ((kind * 8 + row) * FOLD_SLOTS + slot) * FOLD_STRIDE;
}
// Blocked substitution over a tall top (host.hpp: Symbolic::topblk): block b = rows
// [NF + b*w, min(N, NF + (b+1)*w)); T = per block the strictly-lower part of (I + L_bb)^-1, packed by
// rows ((i, k), k < i at i(i-1)/2 + k), w(w-1)/2 doubles per block
struct TopBlkView {
    int nblocks, w, NF, N;
    const int *Rsplit, *Lsplit; // indexed by j - NF
    double *T;
    double *ys;    // nblocks * w: the blocks' intermediate right-hand sides
    int *counters; // nblocks arrival tickets (zero between sweeps)
};
struct ListView {
    const int *idx; // T or W rows
    int count;
};
struct ChunkView {
    const int *row, *beg, *end;
    int count;
};

// ---- value plumbing -----------------------------------------------------------
// fill_idx: the slots of L that no entry of K maps to (structural fill-in) -- zeroed here,
// so no memset of the whole factor is needed; status (4 ints) is cleared by the same launch.
// (the entries with both ends in the top: Kx + nnzU, v2l, count = nnzK - nnzU)
void scatter_init(hipStream_t s, const double *Kx, const int *a2l, int nnzK, int nnzL, double *Lx,
                  double *D, const int8_t *dsigns, const double *eps_or_null, const int *fill_idx,
                  int nfill, int *status);
// eps = c + prop * max(static_max, slotted maxima of |diag K| left by the cone kernels); the slots are
// cleared for the next update.  scal[0] = eps out.
void eps_from_slots(hipStream_t s, unsigned long long *slots, double c, double prop, double static_max,
                    double *scal);
void scatter_rest(hipStream_t s, const double *Kx, const int *a2l, const int *rest, int nrest, int nnzL, double *Lx, double *D,
                  const int8_t *dsigns, const double *eps, int *status);
void gather_values(hipStream_t s, double *Sx, const double *Kx, const int *Smap, int nnzS);
// Dense diagonal blocks of the top in the residual (host.hpp: Symbolic::dblk_*): block b = the m[b] nodes
// rownode[rowbase[b] ..], its strict upper triangle row by row in the device's K values, row a starting at
// start[rowbase[b] + a].
struct DblkView {
    int nblk = 0, split = 1, mmax = 0, nrows = 0;
    const int *p0 = nullptr, *m = nullptr, *rowbase = nullptr; // per block
    const int *start = nullptr;                                // per block row
    const int *rownode = nullptr;                              // per block row: its node
    double *P = nullptr;                                       // nrows x split partial products
};
int dblk_attributes(int mmax);
// bt[i] = b[i] - (H x)[i] for the rows of the blocks (bt holds a copy of b on entry); every entry of a block is read once
void dblk_symv(hipStream_t s, const DblkView &d, const double *Kx, const double *x, double *bt);
void dblk_symv2(hipStream_t s, const DblkView &d, const double *Kx, const double *x0, const double *x1, double *P1, double *bt0);
void dblk_finish(hipStream_t s, const DblkView &d, double *bt); // (d.P: the context's own partial sums)
void diag_absmax_eps(hipStream_t s, const double *Kx, const int *diag_idx, int N, double c,
                     double prop, double *scal /*[0]=eps out, uses [1] as scratch*/);
void scatter_values(hipStream_t s, double *Kx, const int *map, const double *vals, int k, double scale);
// Kx[map[t]] += sign[t] * offset (signs == nullptr: +offset) / Kx[map[t]] *= scale over an index set held on the device
void offset_values(hipStream_t s, double *Kx, const int *map, const int8_t *signs, int k, double offset);
void scale_values(hipStream_t s, double *Kx, const int *map, int k, double scale);

// ---- numeric LDL' -------------------------------------------------------------
// fold.k == 1: every bundle also subtracts its share of the single top column's pivot; fold_top_pivot
// then applies the pivot rule to it (the top needs no factor launches of its own)
// lds_doubles > 0: k_bundle_factor_lds, the bundle's L and D values resident in LDS (lds_doubles = the largest
// bundle's entries + nodes; bundle_factor_lds_ok says whether the handle qualifies)
// returns hipSuccess (0) or the launch error
int bundle_factor(hipStream_t s, const LdlView &v, const BundleView &bv, const FoldView &fold, int lds_doubles = 0);
bool bundle_factor_lds_ok(int lds_doubles);
void fold_top_pivot(hipStream_t s, const LdlView &v, const FoldView &fold);
// fold.k > 0: every bundle also subtracts its part of the k top rows of L from x[NF + i]
void bundle_fwd(hipStream_t s, const LdlView &v, const BundleView &bv, double *x, const FoldView &fold);
// the k x k top-top part of the forward sweep, D^-1, and the backward sweep, in one tiny launch
void fold_top_solve(hipStream_t s, const LdlView &v, const FoldView &fold, double *x);
// e[NF + i] = b - (bundle parts accumulated in fold.tsum) - (top-top part of K) x; ||.||inf folded in
void fold_top_residual(hipStream_t s, const FoldView &fold, const double *Sx, const double *x, const double *b,
                       double *e, unsigned long long *nrm, int *nan);
// addv != nullptr: the bundle rows leave as x + addv (refinement candidate), see Engine::enqueue_solve_inplace
void bundle_bwd(hipStream_t s, const LdlView &v, const BundleView &bv, double *x, const double *addv);
// e[bundle rows] = b - K x with K stored once (U: row i = diagonal + entries to ancestors)
void bundle_symv(hipStream_t s, const BundleView &bv, const int *Up, const int *Ucol, const double *Ux,
                 const double *x, const double *b, double *e, unsigned long long *nrm, int *nan, const FoldView &fold, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// ---- whole solve + iterative refinement in one persistent launch (k_bundle_ir) ----------------------
struct IrView {
    const double *rx, *rz; // right-hand side in the caller's order: entry o < n from rx, n <= o < n + m from rz,
                           // the sparse-cone rows are zero (directldlkktsolver.rs:160-166)
    int n, m, N;
    const int *perm;       // perm[new] = old
    // optional run-length form of perm inside the bundles (nullptr: per-element): bundle b owns runs
    // [run_ptr[b], run_ptr[b+1]); run r = {first local index, first original index, length}: local index
    // l0 + t <-> original index o0 + t, all inside ONE of the ranges [0, n), [n, n + m), [n + m, N)
    const int *run_ptr, *runs;
    double *bp;            // N: the permuted right-hand side, kept for the residuals
    double *xa, *xb;       // N each: accepted iterate / candidate (the roles swap)
    double *ebuf;          // N: residual spill, used only when a workgroup owns several bundles
    double *lhsx, *lhsz;   // outputs in the caller's order (either may be nullptr)
    double *part;          // ir_part_doubles(nb, k) doubles of partial results
    int *ctl;              // ir_ctl_ints() ints, zero at launch: the grid barrier's counters (a launch that runs
                           // to its end leaves them zero again; after a barrier timeout the host clears them)
    int *res;              // 4 ints written by the launch: [0] 1 ok / -1 numerical failure (0: did not finish),
                           // [1] refinement rounds, [2] barrier timeout, [3] the accepted x is in xb
    double abstol, reltol, stopratio;
    int maxiter, ir_enable;
    long long *dbg;        // diagnostics: 128 time stamps of two workgroups, or nullptr
    long long *dbg_all;    // diagnostics (CHIP_IR_DEBUG=2): 32 words per workgroup: [0] hardware id, [1..] time stamps
    int test_drop;         // tests: the last workgroup leaves at once, so every grid barrier times out
    int flat;              // entry-parallel sweeps / residual (bundle_sweep_flat, bundle_symv_flat); 0: column per thread
    int *rel;              // k_bundle_irs: ir_rel_ints() ints, the barrier's release records (tagged messages; any content at launch)
    int epoch;             // k_bundle_irs: distinguishes this launch's messages from an earlier launch's (the host counts)
    int sf_flags;          // k_bundle_irs, experiment bits (CHIP_IRS_FLAGS): 1 = round 0's iterate stays in registers
    int spec_out;          // k_bundle_irs: lhsx / lhsz do not overlap rx / rz -- the last candidate may be written before its verdict
    int sf;                // k_bundle_irs (one bundle per workgroup, the candidate in registers; bp may be nullptr: not written)
};
int ir_rel_ints();
int ir_ctl_ints();                 // (+ 32 per group of a grouped fold, appended: GFoldView::gcnt)
size_t ir_part_doubles(int nb, int k);
// largest co-resident grid of k_bundle_ir for these bundles (0: the kernel cannot run) and the workgroup
// size (*tw: 256 or 512 threads) it is to be launched with
int bundle_ir_capacity(const BundleView &bv, int *tw);
// returns hipSuccess (0) or the launch error; grid <= bundle_ir_capacity, grid >= nb when fold.k > 0
int bundle_ir(hipStream_t s, const LdlView &v, const BundleView &bv, const FoldView &fold, const IrView &ir, int grid,
              int tw, const GFoldView &gf);
// k_bundle_irs: whether a co-resident grid of one 256-thread workgroup per bundle exists / whether a bundle qualifies
bool bundle_irs_capacity_ok(const BundleView &bv);
bool irs_bundle_ok(int nloc, int nleaf, int nlevels, int nruns);
// grouped fold, after bundle_factor: every bundle's contribution to the Schur complement of its group's top
// (k_gfold_schur -> gf.fac), then the k x k LDL' of every group's top (k_gfold_top_factor)
void gfold_top_factor(hipStream_t s, const LdlView &v, const BundleView &bv, const GFoldView &gf);

// ---- grouped fold, small bundles: the "step" kernels (bundle_gstep.hip) ------------------------------------
// The entries of a bundle's columns of L and of its U rows in the order the kernels keep them in registers ("gs order"):
// L: the entries inside the bundle level by level (level of the column), then the entries in the group's top rows
// sorted by (top row, column); U: the entries that are not in top columns in their stored order, then those in top
// columns sorted by (top column, row).  All arrays are parallel to the bundle columns of L / the bundles' U rows.
constexpr int GS_MAXL = 14;          // elimination levels inside a bundle the step kernels handle
constexpr int GS_LST = GS_MAXL + 2;
constexpr int GS_DESC = 64;          // ints per bundle descriptor (two 128-byte lines)
constexpr int GS_GTOP = 152;         // ints per group: top table
constexpr int GS_MV = 12;            // messages per (bundle, phase slot): 8 shares of the top rows, the bundle's ||e||inf, ||b||inf
struct GStepView {
    int lr, ur, nr;               // register slots a thread needs: ceil(max L entries / 256), ceil(max U entries / 256),
                                  // ceil(max nodes / 256)
    // per bundle, one 128-byte line: [0] first node s0, [1] nodes, [2] first CSC slot of its columns of L, [3] entries of L,
    // [4] first entry of its U rows, [5] entries of U, [6] levels nl, [7] group (-1: none), [8] first top row of the group
    // in GFoldView::node, [9] top rows k, [10] first bundle of the group, [11] bundles of the group, [12] gs position of
    // the first U entry in a top column, [13] BundleView::blvl_ptr of the bundle, [14] index of the group's leader among
    // the leaders (the first bundle of every group; a bundle without group leads itself: [10] = itself, [11] = 1), [15] leaders; [16 .. 16 + nl + 1]: gs positions where
    // levels 0 .. nl begin (level nl = the top-row entries), then the bundle's entry count; [32 .. 32 + nl]: first update
    // record (LdlView::fu_rec) of every level, then the end of the last level's records; [48 .. 48 + nl]: first node of every
    // level (bundle-local), then the node count
    const int *desc;
    // per group: [0, 8) node of top row t (final numbering, -1 beyond k), [8, 16) its index in the caller's order,
    // [16, 80) CSC slot of L(top_i, top_j) at i * 8 + j (-1: structurally zero / j >= i), [80, 144) position in V of
    // K(top_i, top_c) at i * 8 + c (-1: zero), [144, 152) Dsigns of the top rows
    const int *gtop;
    const unsigned short *lsrc;   // gs position -> CSC slot of the value, relative to the bundle's first slot
    const unsigned int *lij;      // gs position -> (row16 << 16) | column16, bundle-local; row >= nloc: top row nloc + t
    const unsigned short *usrc;   // gs position -> position inside the bundle's U range the value comes from
    const unsigned int *uij;      // gs position -> (row16 << 16) | column16; column >= nloc: top column nloc + t
    const unsigned short *ufs;    // gs position of a U entry -> where it lands in the factorisation's value store (LdlView::fu_slot)
    // values in gs order, written by k_gstep_factor (nullptr in the view handed to k_gstep_solve: gather through lsrc / usrc)
    double *gsl, *gsu;
    int *msg;                     // k_gstep_solve: [nb][4 phase slots][GS_MV] messages of 16 bytes (value, tag)
    int *lmsg;                    // [leaders][2][2]: a group's ||e||inf, ||b||inf, published by its leader, polled by every workgroup
    int *fmsg;                    // k_gstep_factor: [nb][36] messages (a bundle's Schur share, packed lower triangle)
    int epoch;                    // tags of this launch's messages (the host counts launches)
    long long *dbg;               // diagnostics (CHIP_IR_DEBUG=3): 32 words per workgroup of k_gstep_factor, or nullptr
};
// largest co-resident grid of k_gstep_solve for these bundles (0: cannot run)
int gstep_solve_capacity(const BundleView &bv, const GStepView &gs);
// one launch = setrhs + LDL' solve + iterative refinement with its decisions + getlhs (as bundle_ir); grid = bv.nb
int gstep_solve(hipStream_t s, const LdlView &v, const BundleView &bv, const IrView &ir, const GFoldView &gf,
                const GStepView &gs);
// bundle factorisation + Schur shares + the groups' k x k tops in one launch (needs LdlView::fu_rec and
// BundleView::max_entries)
bool gstep_factor_ok(const BundleView &bv);
int gstep_factor(hipStream_t s, const LdlView &v, const BundleView &bv, const GFoldView &gf, const GStepView &gs);

void factor_T(hipStream_t s, const LdlView &v, ListView cols);
void factor_W(hipStream_t s, const LdlView &v, ListView cols);
void factor_B(hipStream_t s, const LdlView &v, ChunkView chunks);
// chain supernodes (host.hpp: Symbolic::sn_*)
struct SnodeView {
    const int *sn_ptr, *sn_col;
    const long long *upd_ptr; // per supernode: offset of its packed strict lower triangle of B x B in upd_slot
    const int *upd_slot;      // CSC slot of L(B[r], B[c]), r > c  (nullptr: no dense ancestor updates)
    const int *sn_geo;        // per supernode: (last member column e, rows of B = |struct(e)|) -- saves two dependent loads
    const int *sn_cb;         // per member (parallel to sn_col): Lp[c_t] - t - 1, the base of panel column t (entry (i, t) at cb + i)
    double *sn_d;             // per member: its pivot d_t, written by k_snode_diag next to D[c_t] (the update tiles read the
                              // pivots of a run of members: contiguous here, cols -> D there)
    const int8_t *sn_sg;      // per member: dsigns[c_t] (the block factorisation reads the signs of a run of members)
    long long *dbg;           // CHIP_SN_DEBUG: phase stamps of workgroup 0 of the launch (64 slots), else nullptr
    int *sn_cnt;              // per supernode: workgroups of k_snode_panel that have finished with the unfactored diagonal
                              // block (zero between launches: the last one resets it)
    // ancestor updates ASSEMBLED (SnodeAsmView): k_snode_extend stores supernode s's update matrix at U + asm_uoff[s]
    // (packed like upd_slot) and its diagonal at Ud + asm_doff[s] instead of subtracting it with atomics (nullptr)
    double *U = nullptr, *Ud = nullptr;
    const long long *asm_uoff = nullptr;
    const int *asm_doff = nullptr;
    // one-pass substitution matrices (snode_g.hip): supernode s keeps G = [I; L_B] T^-1, (w + nb) x w, column-major with
    // leading dimension snode_g_ld(w + nb), at Gx + g_off[s] (g_off[s] < 0: the supernode keeps the pipelined substitution)
    double *Gx = nullptr;
    const long long *g_off = nullptr;
};
// One unit level's update matrices summed per TARGET column (host.hpp: Symbolic::asm_*): workgroup t owns node
// tgt[t0 + t] and subtracts its sources src_ptr[..] one after the other -- a fixed order, no atomics.  Source q =
// src[3 q .. 3 q + 2] = (offset in U, offset in upd_slot, entries | offset in Ud << 32).
struct SnodeAsmView {
    const int *tgt, *src_ptr;
    const long long *src;
    int t0 = 0, nt = 0;
};
int snode_kernel_attributes(int wmax, int nbmax);

// optional per-launch hook: hipEvent pairs around the launches of ONE selected kernel family (engine.hpp:
// ProfFamily; the ids of the supernode kernels are fixed here because the launchers live in snode.hip)
enum { PFK_SN_UPDATE = 7, PFK_SN_DIAG = 8, PFK_SN_ROWS = 9, PFK_SN_EXTEND = 10, PFK_SN_TRI = 11 };
struct LaunchProf {
    void (*begin)(void *ctx, int family);
    void (*end)(void *ctx, int family);
    void *ctx;
};
void factor_snodes(hipStream_t s, const LdlView &v, const SnodeView &sv, const int *order, int count, int wmax_all,
                   int nblk, int hmax, int nbmax, const LaunchProf *lp = nullptr, const SnodeAsmView *av = nullptr);
void factor_finalize(hipStream_t s, const LdlView &v, ListView cols);

// ---- triangular solves + symv (row-gather family) -------------------------------
enum GatherMode { FWD = 0, BWD = 1, SYMV = 2, SPMV = 3 };
struct GatherArgs {
    const int *ptr, *idx; // CSR-like: row r owns [ptr[r], ptr[r+1])
    const double *val;
    const double *xin;  // gathered vector
    double *out;        // FWD/BWD: in-place accumulator (== xin); SYMV: e
    const double *aux;  // BWD: Dinv; SYMV: b
    // SYMV only: ||out||inf folded into the launch.  NRM_SLOTS words, spread over
    // distinct cache lines so the ~12 ns/atomic same-address serialisation of a
    // device-scope atomicMax is divided by the slot count; the host takes the max.
    unsigned long long *nrm;
    int *nan;
    double alpha;       // SPMV only: out = (aux ? aux : 0) + alpha * (M xin)
};
constexpr int NRM_SLOTS = 64;
constexpr int NRM_STRIDE = 16; // in u64 words: one slot per 128-byte line
void gather_Bprep(hipStream_t s, GatherMode m, const GatherArgs &a, ListView rows);
// levels [l0, l1) of a chain-like stretch in ONE single-workgroup launch (FWD ascending, BWD
// descending); t_idx/w_idx are the FULL list arrays, t_ptr/w_ptr their per-level pointers (device)
// invert the diagonal blocks of L (after a refactor) / sweep the top block by block
void solve_kernel_attributes(); // once per process, before the first (possibly captured) solve
void topblk_build(hipStream_t s, const LdlView &v, const TopBlkView &tb);
void topblk_solve(hipStream_t s, GatherMode m, const LdlView &v, const TopBlkView &tb, double *x);
// levels [l0, l1) of a chain-like stretch of the factorisation in ONE single-workgroup launch
void factor_chain(hipStream_t s, const LdlView &v, const int *t_idx, const int *t_ptr, const int *w_idx,
                  const int *w_ptr, int l0, int l1);
void gather_chain(hipStream_t s, GatherMode m, const GatherArgs &a, const int *t_idx, const int *t_ptr,
                  const int *w_idx, const int *w_ptr, int l0, int l1);
// T + W + B lists of one level in a single launch (B rows still need gather_Bprep first)
void gather_merged(hipStream_t s, GatherMode m, const GatherArgs &a, ListView t, ListView w, ChunkView c);
// pipelined multi-workgroup substitution through wide supernodes (k_snode_tri): blk_ptr[sn] = first flag of
// supernode sn (one per 64-column block), epoch = the value a finished block's flag carries in this sweep
struct SnodeTriView {
    const int *blk_ptr;
    int *msg;   // per 64-column block of every supernode: 64 messages of 16 bytes (x value, sweep epoch twice)
    int epoch;
    int *timeout_flag;
};
// forward / backward substitution through the supernodes order[0..count) of one unit level
void solve_snodes(hipStream_t s, GatherMode m, const LdlView &v, const SnodeView &sv, const int *order, int count,
                  int wmax_all, int nbmax_all, int wlvl, int nblvl, double *x, const SnodeTriView *tri = nullptr,
                  const LaunchProf *lp = nullptr, double *x2 = nullptr, const SnodeTriView *tri2 = nullptr);
// whether solve_snodes takes the pipelined multi-workgroup path for a level of this width (the only one with a form for
// two right-hand sides: x2 / tri2 -- the second vector and ITS message buffer -- are ignored elsewhere)
bool solve_snodes_is_tri(const SnodeTriView *tri, int wlvl);
// ---- supernodes of moderate width: substitutions in one pass over G = [I; L_B] T^-1 (snode_g.hip) ----
int snode_g_max_width();            // widest supernode that may take this path
long long snode_g_ld(int h);        // leading dimension of a supernode's G
int snode_g_attributes(int hmax);   // once per handle (dynamic LDS of the kernels)
// after the numeric factorisation: G of every supernode on the path; tasks = (record in order_all, first row) pairs,
// 256 rows of G per workgroup
void snode_ginv(hipStream_t s, const LdlView &v, const SnodeView &sv, const int *order_all, const int *tasks, int ntasks);
// one unit level's supernodes, forward (x_S(new) -> yt, x_B -= M x_S) or backward (x_S <- G' [D^-1 yt_S; -x_B]);
// wlvl / hlvl: the level's largest width / height
// ga != nullptr: the launch also takes the row gathers (t, w, c) of ga -- forward: those of the NEXT unit level (every final
// store an atomic subtraction), backward: the ordinary columns of THIS level (c must be empty) -- see snode_g.hip: SweepGather
void solve_snodes_g(hipStream_t s, GatherMode m, const LdlView &v, const SnodeView &sv, const int *order, int count, int wlvl,
                    int hlvl, double *x, double *yt, const LaunchProf *lp = nullptr, const GatherArgs *ga = nullptr,
                    ListView t = ListView{nullptr, 0}, ListView w = ListView{nullptr, 0}, ChunkView c = ChunkView{nullptr, nullptr, nullptr, 0});
// a RUN of consecutive unit levels on the one-pass matrices in ONE persistent launch (k_snode_gsweep): the levels in sweep
// order, each with the row gathers that ride along (forward: the next level's; backward: the level's ordinary columns)
struct GSweepLevel {
    int off, count;              // the level's records: order_all + 8 * off, count supernodes
    int gx;                      // 64-row (forward) / 64-column (backward) blocks per supernode
    int tcount, wcount, ccount;  // the riding gathers' lists (SweepGather)
    int pad0, pad1;
    const int *trows, *wrows, *crow, *cbeg, *cend;
};
// largest co-resident grid of the persistent sweep with `lds` bytes of dynamic LDS (0: cannot run)
int snode_gsweep_capacity(GatherMode m, size_t lds);
size_t snode_gsweep_lds(GatherMode m, int hmax);
// ctl: ir_ctl_ints() zeros (zero again when the launch ends); fail: raised when a barrier times out
void solve_snodes_gsweep(hipStream_t s, GatherMode m, const LdlView &v, const SnodeView &sv, const int *order_all, double *x,
                         double *yt, const GSweepLevel *lv, int nlev, int grid, size_t lds, const GatherArgs &ga, int *ctl,
                         int *fail, const LaunchProf *lp = nullptr);
// diagnostics / tests: a kernel of `blocks` x `threads` that only spins for `usec` microseconds on stream s
// (co-residency tests of the persistent launches)
void debug_spin(hipStream_t s, int blocks, int threads, int lds_bytes, double usec);
// ||v[rows]||inf of a short row list into the slots (the B rows of a SYMV)
void norm_rows(hipStream_t s, const double *v, ListView rows, unsigned long long *nrm, int *nan);

// ---- vectors ------------------------------------------------------------------
void permute_in(hipStream_t s, double *y, const double *b, const int *perm, int N);
void permute_out(hipStream_t s, double *x, const double *y, const int *perm, int N);
// bperm (kept for the refinement residuals) and xinit (solved in place) both receive the
// permuted right-hand side; ||b||inf is folded into nrm/nan (slot layout as above)
void setrhs_perm(hipStream_t s, double *bperm, double *xinit, const double *rhsx, const double *rhsz,
                 const int *perm, int n, int m, int N, unsigned long long *nrm, int *nan);
void getlhs_perm(hipStream_t s, double *lhsx, double *lhsz, const double *xperm, const int *iperm,
                 int n, int m);
void add_vec(hipStream_t s, double *dx, const double *x, int N); // dx = x + dx
// w = a x + b y (y == nullptr: w = a x); w may alias x or y
void waxpby(hipStream_t s, double *w, double a, const double *x, double b, const double *y, int n);
// *out = a . b, deterministic (fixed partition + tree); scratch: dot_scratch_doubles() doubles
int dot_scratch_doubles();
void dot(hipStream_t s, const double *a, const double *b, int n, double *out, double *scratch);
// up to DOT_BATCH_MAX dot products in one launch pair: out[spec.slot] = a . b with the same fixed
// partition as dot(); scratch: multi_dot_scratch_doubles() doubles
constexpr int DOT_BATCH_MAX = 12;
struct DotSpec {
    const double *a, *b;
    int n, slot;
};
struct DotBatch {
    DotSpec s[DOT_BATCH_MAX];
    int count;
};
int multi_dot_scratch_doubles();
void multi_dot(hipStream_t s, const DotBatch &bt, double *out, double *scratch);
// w = a x + b y + c z
void lin3(hipStream_t s, double *w, double a, const double *x, double b, const double *y, double c, const double *z,
          int n);
// *out = sum (s + alpha ds)(z + alpha dz)   (vecmath.rs:87-99)
void dot_shifted(hipStream_t s, const double *z, const double *sv, const double *dz, const double *ds, double alpha,
                 int n, double *out, double *scratch);
// slots (NRM_SLOTS x NRM_STRIDE) <- bit patterns of partial max|v|; *nanflag |= any NaN
void norm_inf(hipStream_t s, const double *v, int N, unsigned long long *out, int *nanflag);

// ---- cones --------------------------------------------------------------------
struct SocView {
    int ncones;
    const int *start, *dim;      // rows in [0,m)
    const int *hs_start;         // start of the cone's Hs block in mapHs
    const int *sparse_idx;       // -1 for dense (dim<=4) cones
    const int *sp_ptr;           // offsets into mapU / mapV
    const int *mapHs, *mapU, *mapV, *mapD; // K.nzval indices
    double *w, *lam;             // m-sized state
    double *eta, *d;             // per-cone state
    int *fail;                   // a cone that leaves its interior stores fail_gen here (no clearing between updates)
    int fail_gen;
};
// Exponential / Power cones (3-dimensional, non-symmetric): per-cone state of 18 doubles
// = Hs[6] | H_dual[6] | grad[3] | z[3]
struct Ns3View {
    int ncones;
    const int *start;    // rows in [0,m)
    const int *hs_start; // start of the cone's 6-entry Hs block in mapHs
    const int *tag;      // 3 = Exponential, 4 = Power
    const double *alpha; // PowerConeT exponent
    double *state;
    const int *mapHs;
};
// Generalised power cones (genpowcone.rs): one workgroup per cone.  state per cone (doubles) at
// state_off: alpha[d1] | q[d1] | d1[d1] | r[d2] | p[d1+d2] | grad[d1+d2] | z[d1+d2] | d2, mu, psi
struct GpwView {
    int ncones;
    const int *start, *dim1, *dim2, *hs_start, *state_off;
    const int *map_ptr;               // per cone: offset of its q / r / p index runs in mapQRP (q, then r, then p)
    const int *mapQRP, *mapD, *mapHs; // K.nzval indices (mapD: 3 per cone)
    double *state;
};
// PSD triangle cones held on the device: state per cone = B (n*n, the NT scaling matrix R R') | lambda (n) |
// lambda^-1/2 (n) | R | Rinv.  maxdim <= 64: the kernels' work matrices live in LDS (scratch == nullptr);
// larger cones: in this cone's slice of `scratch` (scratch_stride doubles per cone, >= 4 n^2 + 4 n + 16)
struct PsdView {
    int ncones;
    int maxdim;
    double *scratch;
    long long scratch_stride;
    const int *start, *dim, *hs_start, *state_off;
    double *state;
    const int *mapHs;
    int *fail;
    int fail_gen;
    // Hs written row by row of the device's value order (k_psd_write_hs_rows): set when every PSD cone's block of K is
    // one of the dense diagonal blocks of the top (DblkView) -- block b belongs to cone blk_cone[b] (-1: not a PSD
    // cone's), its i-th row is the svec entry (row_ij & 0xffff, row_ij >> 16) of the cone
    // cones larger than 64 (work matrices in HBM scratch): doubles of dynamic LDS the launch provides for the JACOBI phases
    // -- the eigenvalue iteration of step_length / margins (one n x n matrix) and the one-sided SVD of update_scaling (two)
    // are hundreds of rounds, each a few barriers apart, every access a round trip to the L2 when the matrix sits in
    // scratch; staged in LDS they run as for the small cones.  0: not staged (set by the launchers in cones.hip).
    int jacobi_lds = 0;
    int rows_nblk = 0;
    const int *blk_cone = nullptr, *row_ij = nullptr, *blk_m = nullptr, *blk_rowbase = nullptr, *blk_start = nullptr;
};
void psd_update_scaling(hipStream_t s, const PsdView &v, const double *sv, const double *zv);
// (Lx / l0: k_psd_write_hs_rows writes the blocks into L as well -- only when the rows form runs, see psd_write_hs_rows_active)
void psd_write_hs(hipStream_t s, const PsdView &v, double *Kx, double *Lx = nullptr, const int *l0 = nullptr);
bool psd_write_hs_rows_active(const PsdView &v);
// PSD cone operations either side of the solve (psdtrianglecone.rs:104-303, symmetric_common.rs:53-95)
void psd_mul_hs(hipStream_t s, const PsdView &v, double *y, const double *x);
void psd_affine_ds(hipStream_t s, const PsdView &v, double *ds);
void psd_combined_ds_shift(hipStream_t s, const PsdView &v, double *shift, double *step_z, double *step_s,
                           double sigma_mu);
void psd_ds_from_dz_offset(hipStream_t s, const PsdView &v, double *out, const double *ds);
int psd_step_length(hipStream_t s, const PsdView &v, const double *dz, const double *ds, double amax,
                    double *partial);
int psd_margins(hipStream_t s, const PsdView &v, const double *z, double *pmin, double *psum);
int psd_barrier(hipStream_t s, const PsdView &v, const double *z, const double *sv, const double *dz,
                const double *ds, double alpha, double *partial);
void psd_unit_shift(hipStream_t s, const PsdView &v, double *z, double alpha);
void psd_unit_initialization(hipStream_t s, const PsdView &v, double *z, double *sv);
void ns3_update_scaling(hipStream_t s, const Ns3View &v, const double *sv, const double *zv, double mu,
                        int strategy);
void ns3_write_hs(hipStream_t s, const Ns3View &v, double *Kx);
void ns3_mul_hs(hipStream_t s, const Ns3View &v, double *y, const double *x);
// Nonnegative rows + second-order cones, one launch each.  Scaling (nonnegativecone.rs:77-90, socone.rs:134-211)
// and the Hs values / sparse-cone columns written into Kx (get_Hs negated, datamaps.rs:199-220).
// dslots (may be nullptr): slotted maxima of |diagonal entries written| for the static regulariser
// (NRM_SLOTS words of stride NRM_STRIDE + one NaN flag word at NRM_SLOTS * NRM_STRIDE)
void sym_update_scaling(hipStream_t s, const SocView &v, const int *nn_rows, int nn, const double *sv,
                        const double *zv, double *w, double *lam);
void sym_write_kkt(hipStream_t s, const SocView &v, const int *nn_rows, const int *nn_hsidx, int nn, const double *w,
                   const int *mapHs, double *Kx, unsigned long long *dslots);
// both in one launch (solver.rs:334-352: cones.update_scaling, then the KKT update's get_Hs scatter): every workgroup
// scales its cone / slab and writes its K entries at once; status_or_null: the 4 status words of the refactor that
// follows are cleared here (its own preparation launch is skipped)
void sym_scale_write(hipStream_t s, const SocView &v, const int *nn_rows, const int *nn_hsidx, int nn, const double *sv,
                     const double *zv, double *w, double *lam, const int *mapHs, double *Kx, unsigned long long *dslots,
                     int *status_or_null);
// step / rhs operations of the symmetric cones (Zero rows, Nonnegative rows, SecondOrder cones)
// Exponential / Power cones either side of the solve (expcone.rs:129-181, powcone.rs:128-180)
void ns3_affine_ds(hipStream_t s, const Ns3View &v, double *ds, const double *sv);
void ns3_combined_ds_shift(hipStream_t s, const Ns3View &v, double *shift, const double *step_z,
                           const double *step_s, double sigma_mu);
void ns3_ds_from_dz_offset(hipStream_t s, const Ns3View &v, double *out, const double *ds);
// per-block minima of the backtracking line searches started at alpha -> partial; returns #partials
int ns3_step_length(hipStream_t s, const Ns3View &v, const double *dz, const double *ds, const double *z,
                    const double *sv, double alpha, double alpha_min, double step, double *partial);
// compute_barrier of the composite cone (compositecone.rs:342-352): partial sums -> partial; returns #partials
int cone_barrier(hipStream_t s, const int *nn_rows, int nn, const SocView &soc, const Ns3View &v, const double *z,
                 const double *sv, const double *dz, const double *ds, double alpha, double *partial);
// unit_initialization of the composite cone (compositecone.rs:208-214): z, s of length m
void cone_unit_initialization(hipStream_t s, const int *nn_rows, int nn, const SocView &soc, const Ns3View &v,
                              double *z, double *sv, int m);
// GenPow cones: update_scaling (dual scaling only, genpowcone.rs:141-157,361-401), the sparse KKT
// expansion update (datamaps.rs:322-343) and the operations either side of the solve (:163-250)
void gpw_update_scaling(hipStream_t s, const GpwView &v, const double *zv, double mu);
void gpw_write_kkt(hipStream_t s, const GpwView &v, double *Kx);
void gpw_mul_hs(hipStream_t s, const GpwView &v, double *y, const double *x);
void gpw_copy(hipStream_t s, const GpwView &v, double *out, const double *in);          // affine_ds, ds_from_dz_offset
void gpw_combined_ds_shift(hipStream_t s, const GpwView &v, double *shift, double sigma_mu);
int gpw_step_length(hipStream_t s, const GpwView &v, const double *dz, const double *ds, const double *z,
                    const double *sv, double alpha, double alpha_min, double step, double *partial);
// work: an m-vector of scratch (the primal gradient of every cone)
int gpw_barrier(hipStream_t s, const GpwView &v, const double *z, const double *sv, const double *dz,
                const double *ds, double alpha, double *partial, double *work);
void gpw_unit_initialization(hipStream_t s, const GpwView &v, double *z, double *sv);
void cone_unit_shift(hipStream_t s, const int *nn_rows, int nn, const int *zero_rows, int nz, const SocView &v,
                     double *z, double alpha, int primal);
void cone_affine_ds(hipStream_t s, const int *nn_rows, int nn, const int *zero_rows, int nz, const SocView &v,
                    double *ds);
void cone_combined_ds_shift(hipStream_t s, const int *nn_rows, int nn, const int *zero_rows, int nz,
                            const SocView &v, double *shift, double *step_z, double *step_s, double sigma_mu);
void cone_ds_from_dz_offset(hipStream_t s, const int *nn_rows, int nn, const int *zero_rows, int nz,
                            const SocView &v, double *out, const double *ds, const double *z);
// both return the number of partial results written (to be min-/sum-reduced by the host)
int cone_step_length(hipStream_t s, const int *nn_rows, int nn, const SocView &v, const double *dz,
                     const double *ds, const double *z, const double *sv, double amax, double *partial,
                     int partial_cap);
int cone_margins(hipStream_t s, const int *nn_rows, int nn, const SocView &v, const double *z, double *pmin,
                 double *psum, int partial_cap);
void cones_mul_Hs(hipStream_t s, const int *nn_rows, int nn_count, const SocView &v,
                  const int *zero_rows, int zero_count, double *y, const double *x);

} // namespace dev
} // namespace chip
