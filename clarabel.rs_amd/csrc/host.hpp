// host.hpp -- host-side (C++17) pieces of the MI355X KKT backend: fill-reducing
// ordering, symbolic analysis (runs ONCE per problem, on the host, as
// north_star prescribes) and KKT assembly in Clarabel's exact CSC layout.
// Reference citations are relative to /root/reference/src.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <functional>
#include <string>
#include <memory>
#include <utility>
#include <vector>

#include "switches.hpp"

namespace chip {

using i64 = int64_t;
using i32 = int32_t;

// ---------------------------------------------------------------------------
// ordering (amd_order.cpp)
// ---------------------------------------------------------------------------
struct AmdInfo {
    double lnz = 0, ndiv = 0, nmultsubs_ldl = 0;
    i64 ndense = 0;
};
// Approximate minimum degree (Amestoy/Davis/Duff 1996) on the pattern of
// A + A' given triu(A) in CSC.  perm[k] = k-th eliminated original index.
// Stands in for crate `amd 0.2.2` (qdldl.rs:905-917; dense = 10*dense_scale*sqrt(n)).
int amd_order(i64 n, const i64 *Ap, const i64 *Ai, double dense_scale, std::vector<i64> &perm,
              AmdInfo *info);
// the same by connected components: identical patterns ordered once, distinct ones in parallel (amd_order.cpp)
int amd_order_components(i64 n, const i64 *Ap, const i64 *Ai, double dense_scale, std::vector<i64> &perm,
                         AmdInfo *info);
// the same with dense cone blocks entering as one weighted node each (group[i]: block of node i, -1: none)
int amd_order_grouped(i64 n, const i64 *Ap, const i64 *Ai, double dense_scale, const i32 *group, std::vector<i64> &perm,
                      AmdInfo *info);

// ---------------------------------------------------------------------------
// symbolic analysis (symbolic.cpp)
// ---------------------------------------------------------------------------
// Work-list classes used by every level-scheduled kernel family.
//   T : one thread per row/column        (short)
//   W : one 256-thread workgroup per row (medium)
//   B : rows split in chunks over many workgroups, partial sums combined with
//       fp64 atomics (the 10^6-long budget row of the portfolio SOCP)
struct LevelLists {
    // per level l: T rows = t_idx[t_ptr[l]..t_ptr[l+1]), W rows likewise,
    // B chunks = (b_row, b_beg, b_end) triples in b_ptr ranges.
    std::vector<i32> t_ptr, t_idx, w_ptr, w_idx, b_ptr, b_row, b_beg, b_end;
    // rows that own at least one B chunk, per level (need a pre/post pass)
    std::vector<i32> br_ptr, br_idx;
};

// std::vector whose resize() does NOT value-initialise (trivial element types): the arrays of the analysis that have
// one entry per entry of K or L are filled by the threads right after they are sized -- a zero fill first is a second,
// single-threaded pass over gigabytes (config 5: ~9 GB of them).  assign(n, v) still fills.
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind {
        using other = NoInitAlloc<U>;
    };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U> void construct(U *p) { ::new ((void *)p) U; }
    template <class U, class... Args> void construct(U *p, Args &&...args) { ::new ((void *)p) U(std::forward<Args>(args)...); }
};
using bigvec = std::vector<i32, NoInitAlloc<i32>>;
template <class T> using rawvec = std::vector<T, NoInitAlloc<T>>; // (sized, then filled by threads: no zero fill in between)

struct Symbolic {
    i32 N = 0;
    i64 nnzK = 0; // nnz(triu K)
    i64 nnzL = 0;
    i64 nnzR = 0; // entries of the row view Rp/Rcol/Rpos (= nnzL without chain supernodes; else only non-member columns)
    std::vector<i32> perm, iperm; // final elimination order (level-major), perm[new] = old
    std::vector<int8_t> dsigns;   // permuted D signs
    // The device keeps K.nzval in "T order" V: row-wise by the smaller permuted index lo, diagonal first,
    // then the entries (lo, hi) to ancestors hi ascending.  V[u] = K.nzval[v2k[u]], k2v its inverse; row lo
    // owns V[Vp[lo] .. Vp[lo+1]).  Rows lo < NF (bundle nodes) are the U rows below; for the rows of the top
    // v2l[u - Vp[NF]] says where V[u] lands in the factor's storage: < nnzL -> Lx position (CSC of L),
    // otherwise nnzL + j -> D[j].
    bigvec k2v, v2k, v2l;
    std::vector<i32> Vp;
    // slots of L in TOP columns (CSC positions) that no entry of K maps to: structural fill-in
    std::vector<i32> fill_idx;
    // ... or, when most of the top's slots are fill-in (the dense trapezoids of chain supernodes: config 5 has 3.9e8 such
    // slots, 1.6 GB of indices), no list at all: the whole range [fill_from, nnz(L)) is cleared before K's entries land
    i64 fill_from = -1;
    // L, CSC with ascending rows (structure only; values live on the device)
    std::vector<i32> Lp;
    bigvec Li;
    // L, CSR (row j: columns k ascending), Rpos = CSC position of the entry,
    // Tpos = CSR position of each CSC entry
    std::vector<i32> Rp;
    bigvec Rcol, Rpos, Tpos;
    std::vector<i32> etree; // parent in the final numbering, -1 = root
    std::vector<i32> level; // elimination-tree level of every node (leaves = 0)
    i32 tree_depth = 0;
    // Final numbering = [bundle 0 | bundle 1 | ... | top].  A bundle is a set of complete
    // elimination subtrees small enough for ONE workgroup (nodes level-major inside);
    // nodes [NF, N) are the remaining ancestors ("top"), level-major, processed by the
    // level-scheduled kernels: top level l = columns [lvlptr[l], lvlptr[l+1]), lvlptr[0] = NF.
    i32 NF = 0;
    std::vector<i32> bundle_ptr;      // nb + 1
    std::vector<i32> blvl_ptr, blvl;  // per bundle: boundaries of its levels (absolute node ids)
    i32 max_bundle_nodes = 0;
    i32 nlevels = 0;                  // number of TOP levels
    // Blocked substitution for a TALL top (chain-like trees: thousands of sequential levels).  The
    // top rows [NF, N) are cut into blocks of TOPBLK consecutive rows (the numbering is topological,
    // so any consecutive range works); the unit-lower diagonal block of L of every block is inverted
    // once per refactor, and a sweep then needs one dependent step per BLOCK instead of per level.
    // topblk = 0: not used.  Rsplit[j - NF]: first CSR slot of row j whose column lies in j's block
    // (the external prefix ends there); Lsplit[j - NF]: first CSC slot of column j whose row lies
    // beyond j's block.
    i32 topblk = 0;
    std::vector<i32> Rsplit, Lsplit;
    // "Arrow" KKTs: a top of at most TOPFOLD_MAX rows (config 3: the single budget row coupling all
    // cones).  Their long rows are then accumulated by the BUNDLE workgroups while the bundle's slice
    // of the vector is in LDS, and a one-workgroup kernel finishes the tiny top-top part.
    //   fold_rseg[(b*k + i)*2 + {0,1}] : CSR slots of row NF+i of L whose columns lie in bundle b
    //   fold_tt[i*k + j] (i > j)        : CSC slot of L(NF+i, NF+j), -1 if structurally zero
    //   fold_sp / fold_scol / fold_sslot: per top row the entries of K with both ends in the top
    //                                     (column index relative to NF, position in V)
    i32 nfold = 0;
    std::vector<i32> fold_rseg, fold_tt, fold_sp, fold_scol, fold_sslot;
    // GROUPED fold: a forest of many small trees whose tops are each at most TOPFOLD_MAX nodes (BASELINE
    // config 4 when a GPU holds only a share of the trees: the forest cut is refined until about `target_wg`
    // bundles exist, several per tree, and the few ancestors of a tree's bundles -- its top -- are folded into
    // the bundle kernels per tree).  Group g = one elimination tree with a non-empty top:
    //   gf_ptr[g] .. gf_ptr[g+1]   : its top nodes gf_node[.] (final numbering, ascending = topological)
    //   gf_bptr[g] .. gf_bptr[g+1] : its bundles (contiguous ids); bundles beyond gf_bptr[ng] belong to no group
    //   gf_tt[g*64 + i*8 + j]      : CSC slot of L(top_i, top_j), i > j, -1 if structurally zero
    //   gf_sp / gf_scol / gf_sslot : per top row (index gf_ptr[g] + i) the entries of K with both ends in the
    //                                group's top: column (index inside the group), position in V
    // Li16 / Ucol16 then encode a top row as nloc + its index inside the bundle's group.  nfold stays 0: every
    // kernel that does not know about groups treats the top as an ordinary level-scheduled top.
    i32 gf_ng = 0;
    std::vector<i32> gf_ptr, gf_node, gf_bptr, gf_tt, gf_sp, gf_scol, gf_sslot;
    // bundles + (at most) a folded top only: 16-bit bundle-local row indices of the entries of the bundle
    // columns of L (Li16, parallel to Li[0 .. Lp[NF])) and of the U rows (Ucol16); an index >= the
    // bundle's node count nloc stands for top row NF + (index - nloc).  Empty otherwise.
    std::vector<uint16_t> Li16, Ucol16;
    // ... or, for systems with a level-scheduled top (no fold), only what the entry-parallel sweeps of the stand-alone bundle
    // kernels need (k_bundle_sweep_flat): bundle-local row (0xFFFF: a top row) and column of every entry of the bundle
    // columns of L, and the most levels of a bundle.  Empty when Li16 is not (the fused solve covers those systems).
    std::vector<uint16_t> sLi16, sLj16;
    i32 max_bundle_levels = 0;
    // ... and, for the entry-parallel ("flat") sweeps and residual of k_bundle_ir, the bundle-local COLUMN of every
    // entry of the bundle columns of L (Lj16) and the bundle-local ROW of every entry of the U rows (Urow16)
    std::vector<uint16_t> Lj16, Urow16;
    // ... and, for the factorisation with the bundle's values resident in LDS (k_bundle_factor_lds), the row lists of
    // the bundle rows in 16 bits: contribution t (t < Rp[NF]) of row j comes from bundle-local column Rk16[t], whose
    // entry (j, k) is the Ro16[t]-th of column k
    std::vector<uint16_t> Rk16, Ro16;
    // ... and the factorisation's UPDATE RECORDS for its entry-parallel form (k_bundle_factor_flat): when column k of a
    // bundle is final, every pair (a >= b) of its entries whose lower row r_b lies in the bundle updates entry
    // (r_a, r_b) -- or the pivot of r_b when a == b -- by l_ak d_k l_bk.  Record = 4 x 16 bits {slot of (r_a, k),
    // slot of (r_b, k), k, target}, slots and k bundle-local, target = local slot of (r_a, r_b) or nE + r_b for a pivot;
    // per (bundle, level of k) one contiguous range fu_ptr[blvl_ptr[b] + l] .., sorted by target inside.
    // fu_slot[u]: where U entry u lands in the bundle's value store (0xFFFF = the diagonal).  Empty when some bundle
    // has nE + nloc >= 65535.
    std::vector<uint16_t> fu_rec, fu_slot;
    std::vector<i32> fu_ptr;
    std::vector<i32> lvlptr;
    // Chain supernodes of the top (symbolic.cpp): supernode s = columns sn_col[sn_ptr[s] .. sn_ptr[s+1])
    // (ascending, each the parent of the previous one); all its columns are padded to the dense
    // trapezoid {later members} + struct(last).  sn_of[j] = supernode of column j or -1.
    // The numeric factorisation schedules UNITS (supernodes and the remaining single top columns) by
    // unit level (fac lists below are indexed by unit level, nfaclevels of them): per level first the
    // ordinary columns (fac), then the contributions from OUTSIDE columns into the members of the
    // level's supernodes (snx: chunks over the filtered row lists Rf_*), then the level's supernodes
    // block column by block column (sn_lvl_ptr ranges of sn_order) and their dense updates of the
    // ancestors' columns (upd_slot).
    std::vector<i32> sn_of, sn_ptr, sn_col;
    // row lists (CSR of L: column, CSC slot) of the top rows WITHOUT supernode-member columns; empty
    // when there are no supernodes
    std::vector<i32> Rf_p, Rf_col, Rf_pos;
    std::vector<i64> upd_ptr;   // per supernode: its range of upd_slot (packed strict lower triangle of B x B)
    std::vector<i32> upd_slot;  // CSC slot of L(B[r], B[c])
    // The same updates ASSEMBLED instead of scattered with atomics: the supernodes of a unit level write their
    // update matrices to a private buffer (supernode s at asm_uoff[s], packed like upd_slot; its diagonal at
    // asm_doff[s]) and one workgroup per TARGET column sums its sources in a fixed order (k_snode_assemble).
    // Per level l: targets asm_lvl_ptr[l] .. asm_lvl_ptr[l+1]; target t = node asm_tgt[t] with the sources
    // asm_src_ptr[t] .. asm_src_ptr[t+1], source = (offset in the level's buffer, offset in upd_slot, entries,
    // offset of its diagonal entry).  Levels with fewer than two contributing supernodes have no targets.
    std::vector<i64> asm_uoff;
    std::vector<i32> asm_doff;
    std::vector<i32> asm_lvl_ptr, asm_tgt, asm_src_ptr;
    struct AsmSrc {
        i64 uo, so;
        i32 cnt, dofs;
    };
    std::vector<AsmSrc> asm_src;
    i64 asm_usize = 0; // doubles: the largest level's buffer
    i32 asm_dsize = 0;
    std::vector<i32> sn_lvl_nblk, sn_lvl_hmax, sn_lvl_nbmax; // per unit level: maxima over its supernodes
    std::vector<i32> sn_lvl_ptr, sn_order;      // supernodes by unit level
    i32 nfaclevels = 0;
    // substitutions with supernodes, by unit level: fwu = rows of L restricted to non-member columns
    // (all top rows; the members' dense parts and the supernodes' pushes to their B rows are done by
    // k_snode_fwd), bwu = columns of the single top columns (the members' columns: k_snode_bwd)
    LevelLists fwu, bwu;
    LevelLists snx;                             // per unit level: B chunks (b_row = member column, ranges into Rf_*)
    LevelLists snb;                             // per unit level of the target: the chunks of the members over the bundle columns of
                                                // their lists (they depend on nothing in the top; snx then keeps the top columns only)
    std::vector<long long> snb_work;            // ... and their work (multiply-adds) per level
    // K for the residual e = b - K x, permuted numbering:
    //   U : rows i < NF (bundle nodes): diagonal + entries to ancestors, each K entry ONCE -- these ARE
    //       the first nnzU entries of V (no copy)
    //   S : rows i >= NF (top nodes): the full row (both triangles); Sp has N+1 entries, empty rows for
    //       i < NF; Smap = position in V (values refreshed by a gather at every refactor)
    //   dense diagonal blocks of the top (dblk_m[b] nodes joined pairwise, first node dblk_p0[b]; nodes and the position
    //       in V of each node's first block entry in dblk_node / dblk_start, block after block): multiplied from V
    //       directly, their off-diagonal entries are NOT in S (symbolic.cpp; k_dblk_symv)
    i64 nnzS = 0, nnzU = 0;
    std::vector<i32> dblk_p0, dblk_m, dblk_start, dblk_node;
    std::vector<i32> Sp;
    bigvec Scol, Smap;
    // with chain supernodes: Scol refers to positions of a copy of x in which the members of a supernode are
    // consecutive (xs[i] = x[xperm[i]]; empty = Scol holds node indices)
    std::vector<i32> xperm;
    std::vector<i32> Up;
    bigvec Ucol;
    // kernel work lists
    LevelLists fac, fwd, bwd; // factor (by column), forward solve (rows of L), backward (columns)
    LevelLists smv;           // symv: a single pseudo-level over all rows
    AmdInfo amd;
};

// `perm0` empty => AMD.  Returns 0 or a negative chip_status.
// target_wg: workgroups the device keeps resident for the fused solve kernel (CUs x 4): a forest with fewer
// bundles than that is cut finer (grouped fold, above); 0 = never refine.
// clique_of (may be nullptr): node -> dense cone block it belongs to (-1: none); the ordering then takes every block as
// one weighted node (amd_order_grouped)
int analyse(i64 n, const i64 *Ap, const i64 *Ai, const int8_t *dsigns_or_null,
            const std::vector<i64> &perm0, double amd_dense_scale, Symbolic &S, i32 target_wg = 1024,
            const std::vector<i32> *clique_of = nullptr);

// ---------------------------------------------------------------------------
// KKT assembly (kkt_assembly.cpp)  -- kkt_assembly.rs:20-183, datamaps.rs
// ---------------------------------------------------------------------------
struct ConeSpec {
    i32 tag;
    i64 dim, dim2;
    i64 numel;
    bool hs_diag, sparse;
    i64 pdim;
    i64 start;       // rng_cones[i].start
    i64 block_start; // rng_blocks[i].start
    i64 block_len;
    i64 sparse_col;  // first extra KKT column of a sparse-expandable cone, else -1
    i64 sparse_idx;  // ordinal among sparse cones
};

struct KktLayout {
    i64 n = 0, m = 0, p = 0, N = 0, nnz = 0;
    std::vector<ConeSpec> cones;
    i64 nHs = 0;
    // triu K
    // (rowval / nzval / mapHs: one entry per entry of K or of the Hs blocks -- config 5: 1.6e8 each --, every slot written exactly
    // once by assemble_kkt_triu, the cone blocks by several threads: no zero fill)
    std::vector<i64> colptr;
    rawvec<i64> rowval;
    rawvec<double> nzval;
    // LDLDataMap (datamaps.rs:350-362)
    std::vector<i64> mapP, mapA, diagP, diag_full;
    rawvec<i64> mapHs;
    // sparse maps, flattened: for sparse cone s the u-entries are
    // sp_u[sp_ptr[s] .. sp_ptr[s]+numel), same for v; D indices sp_D[3*s..]
    std::vector<i64> sp_ptr, sp_u, sp_v, sp_D;
    // GenPow only: q (len dim1) and r (len dim2) offsets
    std::vector<i64> sp_q_ptr, sp_q, sp_r_ptr, sp_r;
    std::vector<int8_t> dsigns;
};

int build_cone_specs(i64 ncones, const i32 *tags, const i64 *dims, const i64 *dims2,
                     std::vector<ConeSpec> &out, i64 &m, i64 &p, i64 &nHs);
int assemble_kkt_triu(i64 n, i64 m, const i64 *Pp, const i64 *Pi, const double *Px, const i64 *Ap,
                      const i64 *Ai, const double *Ax, KktLayout &K);

void set_error(const std::string &msg);

// CHIP_TIMING=1: wall-clock of the analysis phases on stderr
struct PhaseClock {
    const char *tag = "analyse";
    bool on = switches().timing;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void operator()(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[chip %s] %-28s %8.3f s\n", tag, what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

// ---- host threads for the analysis passes (std::thread, no runtime dependency) ----------------
// CHIP_HOST_THREADS overrides the default min(hardware threads, 16); 1 = everything on the caller.
int host_threads();
// body(t, T) on T threads (t = 0..T-1), the caller being thread 0; returns when all are done.
void run_threads(int T, const std::function<void(int, int)> &body);
// cut [0, n) into T contiguous ranges of about equal WEIGHT given the prefix sums ptr[0..n]
// (ptr[0] need not be 0): cuts[t] .. cuts[t+1] is the share of thread t.
template <class P> std::vector<int64_t> balanced_cuts(const P *ptr, int64_t n, int T) {
    std::vector<int64_t> cuts((size_t)T + 1, n);
    cuts[0] = 0;
    int64_t j = 0;
    for (int t = 1; t < T; t++) {
        const double target = (double)ptr[0] + ((double)ptr[n] - (double)ptr[0]) * t / T;
        while (j < n && (double)ptr[j] < target) j++;
        cuts[t] = j;
    }
    return cuts;
}
// Scratch array WITHOUT value initialisation: std::vector zero-fills on the constructing thread -- 650 MB per
// array of config 5's 1.6e8-entry passes, one thread, before the threads that own the data ever touch it (that,
// not the passes themselves, was most of the "T / C2 orders" stage) -- here the first touch is the parallel fill.
template <class T> struct RawBuf {
    std::unique_ptr<T[]> p;
    explicit RawBuf(size_t k) : p(new T[k]) {}
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T *data() { return p.get(); }
    const T *data() const { return p.get(); }
};
// Stable parallel bucket placement (the counting pass of a radix sort): the source is T ordered chunks, chunk t being
// the t-th part of the source order; scan(t, T, f) calls f(key, payload) for the items of chunk t in order.  Every
// item is placed at ptr[key] + (its rank among the items of its key, in source order) -- the result does not depend
// on T.  Work: two scans of the source plus O(T x nkeys) counters (the destination-owning passes this replaces had
// every thread scan the WHOLE source: T x nnz reads per pass, 84 GB per pass for config 5's 1.6e8 entries of K).
template <class Scan, class Place>
void stable_buckets(int T, i32 nkeys, std::vector<i32> &ptr, bool compute_ptr, Scan scan, Place place) {
    if (T < 1) T = 1;
    std::vector<std::vector<i32>> cnt((size_t)T);
    run_threads(T, [&](int t, int TT) {
        std::vector<i32> &c = cnt[(size_t)t];
        c.assign((size_t)nkeys + 1, 0);
        scan(t, TT, [&](i32 key, i64) { c[(size_t)key]++; });
    });
    if (compute_ptr) ptr.assign((size_t)nkeys + 1, 0);
    run_threads(T, [&](int t, int TT) { // per key: the chunks' counts become their offsets inside the key's range
        for (i64 key = (i64)nkeys * t / TT; key < (i64)nkeys * (t + 1) / TT; key++) {
            i32 run = 0;
            for (int c = 0; c < TT; c++) {
                const i32 k = cnt[(size_t)c][(size_t)key];
                cnt[(size_t)c][(size_t)key] = run;
                run += k;
            }
            if (compute_ptr) ptr[(size_t)key + 1] = run;
        }
    });
    if (compute_ptr)
        for (i32 key = 0; key < nkeys; key++) ptr[(size_t)key + 1] += ptr[(size_t)key];
    run_threads(T, [&](int t, int TT) {
        std::vector<i32> &c = cnt[(size_t)t];
        scan(t, TT, [&](i32 key, i64 payload) { place(key, payload, ptr[(size_t)key] + c[(size_t)key]++); });
    });
}
const char *get_error();

} // namespace chip
