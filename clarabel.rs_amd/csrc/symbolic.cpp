// symbolic.cpp -- the once-per-problem host analysis of the quasidefinite KKT
// matrix: ordering, symmetric permutation, elimination tree, pattern of L,
// level sets and the per-level work lists the HIP kernels consume.
//
// What the reference does at this point (qdldl.rs:230-295: AMD -> permute_symmetric
// -> _etree -> logical _factor) is kept in meaning, not in form:
//   * the elimination order is the AMD (or user) order re-sorted LEVEL-MAJOR:
//     nodes of elimination-tree level 0 first, then level 1, ...  A parent is
//     always at a strictly higher level than its children, so this is a
//     topological order of the same tree: same fill, same tree, same pivots up
//     to rounding (SURVEY.md App. E "what a GPU design may change freely").
//     It makes every level a contiguous index range -> coalesced per-level
//     kernels over D, Dinv, Lp, x.
//   * L is stored once as CSC (ascending rows) and indexed a second time by
//     rows (CSR) so that forward substitution, backward substitution and the
//     left-looking numeric factorisation are all pure gathers.
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

#include "host.hpp"

namespace chip {

static thread_local std::string g_err; // per thread, like errno: handles may live on different threads
void set_error(const std::string &msg) { g_err = msg; }
const char *get_error() { return g_err.c_str(); }

int host_threads() {
    int t = (int)std::thread::hardware_concurrency();
    if (t < 1) t = 1;
    if (t > 16) t = 16;
    if (switches().host_threads > 0) t = switches().host_threads;
    return t;
}

void run_threads(int T, const std::function<void(int, int)> &body) {
    if (T <= 1) {
        body(0, 1);
        return;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)T - 1);
    for (int t = 1; t < T; t++) th.emplace_back([&body, t, T] { body(t, T); });
    body(0, T);
    for (auto &x : th) x.join();
}

namespace {


// kernel work-list thresholds (see bundle_factor.hip, bundle_solve.hip, snode.hip)
constexpr i32 T_MAX = 32;      // <= T_MAX entries: one thread per row
constexpr i32 B_MIN = 16384;   // >  B_MIN entries: split over workgroups
constexpr i32 B_CHUNK = 4096;  // entries per B chunk
constexpr i32 DBLK_MIN_M = 64;          // smallest dense diagonal block of the top that the residual multiplies from V directly
constexpr i32 DBLK_MAX_M = 6144;        // largest (k_dblk_symv keeps x and y of the block in LDS)
constexpr i64 DBLK_MIN_ENTRIES = 1 << 20; // fewer block entries than this in total: the blocks stay in the full rows S
constexpr i32 TOPFOLD_MAX = 8;    // at most this many top rows are folded into the bundle kernels (kernels.hpp)
constexpr i32 TOPBLK = 128;       // rows per block of the blocked top substitution (bundle_solve.hip: TOPBLK)
constexpr i64 F_MIN_WORK = 16384;    // factor: a column with more (contribution, tail entry) updates than this
constexpr i64 F_CHUNK_MIN = 1024, F_CHUNK_MAX = 4096, F_CHUNK_PARTS = 96;  // ... is split over workgroups in chunks of about this many updates
constexpr i32 FAC_T_ROW = 8;   // factor: thread-per-column if contributions <= this
constexpr i32 FAC_T_COL = 48;  // ... and column length <= this
// chain supernodes of the top (dense trapezoids factored by one workgroup each, snode.hip)
constexpr i32 SN_MIN_W = 16;       // shorter chains stay ordinary columns
constexpr i32 SN_MAX_W = 4096;     // longer chains are cut
constexpr i32 SN_PAD_ABS = 16;     // explicit zeros tolerated per column: max(SN_PAD_ABS, SN_PAD_REL * padded length)
constexpr double SN_PAD_REL = 0.5;
constexpr i64 SN_UPD_BUDGET = 600000000; // entries of the ancestor-update slot maps (4 bytes each)
// subtree bundles (one workgroup each; the vector slice of a bundle is staged in LDS)
constexpr i64 BUNDLE_MAX_NODES = 6144;     // 48 KiB of fp64 in LDS
constexpr i64 BUNDLE_MAX_ENTRIES = 131072; // nnz(L rows + cols) one workgroup should stream
constexpr i64 BUNDLE_TARGET_COUNT = 2048;  // aim for >= 8 workgroups per CU
constexpr i32 BUNDLE_MAX_COL = 512;        // columns longer than this are not bundled
constexpr i64 BUNDLE_MAX_WORK = 1000000;   // sum of (column length)^2 one workgroup should factor with per-entry gathers (config 2: 8e6 -> 1e6 took the update from 29.8 to 24.1 ms)

// threads worth starting for a pass over `items` entries
// (CHIP_HOST_PAR_MIN: the smallest pass that is split, 2e6 entries by default; tests set it to 0)
inline int par_threads(i64 items) {
    const i64 min_items = switches().host_par_min;
    return items >= min_items ? host_threads() : 1;
}

// upper-triangular pattern of P K P' (row = min, col = max), columns unsorted,
// plus (optionally) for each source entry its destination slot.
void permuted_triu(i64 n, const i64 *Ap, const i64 *Ai, const std::vector<i32> &iperm,
                   std::vector<i64> &Cp, bigvec &Ci) {
    // Threads own ranges of DESTINATION columns: every thread scans all of K (sequential reads) and
    // places the entries of its own columns in source order, so the result does not depend on the
    // number of threads.
    const i64 nnz = n > 0 ? Ap[n] : 0;
    const int T = par_threads(nnz);
    Cp.assign((size_t)n + 1, 0);
    if (T == 1) {
        for (i64 c = 0; c < n; c++) {
            const i32 pc = iperm[c];
            for (i64 p = Ap[c]; p < Ap[c + 1]; p++) {
                const i32 pr = iperm[Ai[p]];
                Cp[(pr > pc ? pr : pc) + 1]++;
            }
        }
        for (i64 c = 0; c < n; c++) Cp[c + 1] += Cp[c];
        Ci.resize((size_t)Cp[n] + 1);
        std::vector<i64> nextp(Cp.begin(), Cp.end() - 1);
        for (i64 c = 0; c < n; c++) {
            const i32 pc = iperm[c];
            for (i64 p = Ap[c]; p < Ap[c + 1]; p++) {
                const i32 pr = iperm[Ai[p]];
                const i32 col = pr > pc ? pr : pc, row = pr > pc ? pc : pr;
                Ci[nextp[col]++] = row;
            }
        }
        return;
    }
    // the permuted (row, col) of every entry once, by source chunks, then one stable bucket pass by column
    // (host.hpp: stable_buckets -- the entries of a column keep their source order for any thread count)
    RawBuf<i32> erow((size_t)nnz + 1), ecol((size_t)nnz + 1);
    {
        const std::vector<int64_t> ccuts = balanced_cuts(Ap, n, T);
        run_threads(T, [&](int t, int) {
            for (i64 c = ccuts[t]; c < ccuts[t + 1]; c++) {
                const i32 pc = iperm[c];
                for (i64 p = Ap[c]; p < Ap[c + 1]; p++) {
                    const i32 pr = iperm[Ai[p]];
                    ecol[p] = pr > pc ? pr : pc;
                    erow[p] = pr > pc ? pc : pr;
                }
            }
        });
    }
    Ci.resize((size_t)nnz + 1);
    Ci[(size_t)nnz] = 0;
    std::vector<i32> cp32;
    stable_buckets(T, (i32)n, cp32, true,
                   [&](int t, int TT, auto f) {
                       for (i64 p = nnz * t / TT; p < nnz * (t + 1) / TT; p++) f(ecol[p], p);
                   },
                   [&](i32, i64 p, i32 u) { Ci[(size_t)u] = erow[p]; });
    for (i64 c = 0; c <= n; c++) Cp[(size_t)c] = cp32[(size_t)c];
}

// elimination tree (parent, -1 = root) and strictly-lower column counts of L.
void etree_counts(i64 n, const std::vector<i64> &Cp, const bigvec &Ci,
                  std::vector<i32> &parent, std::vector<i32> &cnt, std::vector<i32> *rowcnt = nullptr) {
    parent.assign((size_t)n, -1);
    cnt.assign((size_t)n, 0);
    if (rowcnt) rowcnt->assign((size_t)n, 0);
    std::vector<i32> stamp((size_t)n, -1);
    for (i32 j = 0; j < n; j++) {
        stamp[j] = j;
        i32 rc = 0;
        for (i64 p = Cp[j]; p < Cp[j + 1]; p++) {
            i32 i = Ci[p];
            while (stamp[i] != j) {
                if (parent[i] < 0) parent[i] = j;
                cnt[i]++;
                rc++;
                stamp[i] = j;
                i = parent[i];
            }
        }
        if (rowcnt) (*rowcnt)[j] = rc;
    }
}

struct ListBuilder {
    LevelLists &L;
    explicit ListBuilder(LevelLists &l) : L(l) {
        L = LevelLists();
        L.t_ptr.push_back(0);
        L.w_ptr.push_back(0);
        L.b_ptr.push_back(0);
        L.br_ptr.push_back(0);
    }
    void add_T(i32 r) { L.t_idx.push_back(r); }
    void add_W(i32 r) { L.w_idx.push_back(r); }
    void add_B(i32 r, i64 beg, i64 end) {
        L.br_idx.push_back(r);
        for (i64 b = beg; b < end; b += B_CHUNK) {
            L.b_row.push_back(r);
            L.b_beg.push_back((i32)b);
            L.b_end.push_back((i32)std::min<i64>(end, b + B_CHUNK));
        }
    }
    // chunks of a column of the factorisation balanced by WORK: contribution t of column j updates
    // tail(t) = Lp[k+1] - (Rpos[t]+1) entries; a chunk closes at B_CHUNK contributions or at its
    // share of the column's updates (about 1/96 of them, between 1024 and 4096: a 33k-update column
    // of config 2 becomes ~32 chunks, an 800k-update dense PSD column of config 5 ~200)
    // (chunk_min / chunk_max: other bounds of a chunk's updates, 0: the defaults above)
    void add_B_work(i32 r, i64 beg, i64 end, const i32 *Rcol, const i32 *Rpos, const std::vector<i32> &Lp,
                    i64 total_work, i64 chunk_min = 0, i64 chunk_max = 0) {
        L.br_idx.push_back(r);
        const i64 F_CHUNK_WORK = std::min(chunk_max > 0 ? chunk_max : F_CHUNK_MAX,
                                          std::max(chunk_min > 0 ? chunk_min : F_CHUNK_MIN, total_work / F_CHUNK_PARTS));
        i64 b = beg, work = 0;
        for (i64 t = beg; t < end; t++) {
            work += Lp[Rcol[t] + 1] - (Rpos[t] + 1) + 1;
            if (t + 1 == end || t + 1 - b >= B_CHUNK || work >= F_CHUNK_WORK) {
                L.b_row.push_back(r);
                L.b_beg.push_back((i32)b);
                L.b_end.push_back((i32)(t + 1));
                b = t + 1;
                work = 0;
            }
        }
    }
    void close_level() {
        L.t_ptr.push_back((i32)L.t_idx.size());
        L.w_ptr.push_back((i32)L.w_idx.size());
        L.b_ptr.push_back((i32)L.b_row.size());
        L.br_ptr.push_back((i32)L.br_idx.size());
    }
};

} // namespace

int analyse(i64 n, const i64 *Ap, const i64 *Ai, const int8_t *dsigns, const std::vector<i64> &perm0,
            double amd_dense_scale, Symbolic &S, i32 target_wg, const std::vector<i32> *clique_of) {
    S = Symbolic();
    if (n < 0 || n >= (i64)1 << 31) {
        set_error("KKT dimension out of int32 range");
        return -1;
    }
    // ---- check_structure (qdldl.rs:213-228): triu, no empty column ----------
    for (i64 c = 0; c < n; c++) {
        for (i64 p = Ap[c]; p < Ap[c + 1]; p++)
            if (Ai[p] > c || Ai[p] < 0) {
                set_error("matrix is not upper triangular");
                return -3;
            }
    }
    for (i64 c = 0; c < n; c++)
        if (!(Ap[c] < Ap[c + 1])) {
            set_error("matrix has an empty column");
            return -2;
        }
    const i64 nnzK = n > 0 ? Ap[n] : 0;
    if (nnzK >= ((i64)1 << 31) - 8) {
        set_error("nnz(K) out of int32 range");
        return -1;
    }
    S.N = (i32)n;
    S.nnzK = nnzK;

    PhaseClock clk;
    // ---- ordering -----------------------------------------------------------
    std::vector<i64> p0 = perm0;
    if (p0.empty() && n > 0) {
        const bool grouped = clique_of && (i64)clique_of->size() == n && !switches().no_clique_order;
        int rc = grouped ? amd_order_grouped(n, Ap, Ai, amd_dense_scale, clique_of->data(), p0, &S.amd)
                         : amd_order_components(n, Ap, Ai, amd_dense_scale, p0, &S.amd);
        if (rc) {
            set_error("amd_order failed");
            return rc;
        }
    }
    std::vector<i32> ip0((size_t)n, -1);
    for (i64 k = 0; k < n; k++) {
        const i64 v = p0[k];
        if (v < 0 || v >= n || ip0[v] != -1) {
            set_error("invalid permutation");
            return -5;
        }
        ip0[v] = (i32)k;
    }
    clk("ordering");
    std::vector<i64> Cp;
    bigvec Ci;
    std::vector<i32> parent, cnt;
    std::vector<i32> rowcnt;
    // ---- shallower tree, same fill: reorder inside chains ----------------------
    // A chain j -> parent(j) -> ... in which every column is its parent's column plus the parent
    // (count[j] == count[parent]+1) is a clique with one common outer structure: its members may be
    // eliminated in ANY order without new fill, provided every column c below the chain still comes
    // before all chain members it touches.  Minimum degree orders such members by degree ties, not by
    // height: on config 3 it pivots a cone's (u, v) columns before the cone's last two rows, which then
    // sit on the critical path (8 levels instead of 6).  Here each member gets its release level
    // r = 1 + max level of the outside columns that touch it (one row-pattern walk over the tree, the
    // same walk that counts), the members of a chain are re-sequenced by ascending r, and the whole
    // order is re-sorted by the resulting levels.  Pass A below analyses the new order from scratch.
    // (pass A runs first; when members move, the new order is analysed and, its fill being no larger, kept)
    permuted_triu(n, Ap, Ai, ip0, Cp, Ci);
    etree_counts(n, Cp, Ci, parent, cnt, &rowcnt);
    i64 nnzL0 = 0;
    for (i32 j = 0; j < n; j++) nnzL0 += cnt[j];
    // (the walk costs one more pass over nnz(L): skipped for factors beyond 1e8 entries -- dense fronts, whose
    // depth is a matter of the supernode kernels, not of single columns)
    if (perm0.empty() && n > 0 && nnzL0 <= 100000000) {
        std::vector<i32> chain((size_t)n), nlev((size_t)n, 0), rel((size_t)n, 0), stamp((size_t)n, -1);
        std::vector<i32> cfirst((size_t)n, -1), cnext((size_t)n, -1); // members of a chain, linked in old order
        i32 nchains = 0;
        // a node opens a chain unless a child already claimed it as the continuation of the child's chain
        std::vector<i32> claimed((size_t)n, -1);
        std::vector<i32> ctail((size_t)n, -1);
        std::vector<i32> mem, srt;
        bool moved = false;
        for (i32 j = 0; j < n; j++) {
            const i32 c = claimed[j] >= 0 ? claimed[j] : nchains++;
            chain[j] = c;
            if (cfirst[c] < 0) cfirst[c] = j;
            else cnext[ctail[c]] = j;
            ctail[c] = j;
            // release level of j: walk row j's pattern (all columns that hold j)
            stamp[j] = j;
            i32 r = 0;
            for (i64 p = Cp[j]; p < Cp[j + 1]; p++) {
                i32 i = Ci[p];
                while (stamp[i] != j) {
                    if (chain[i] != c && nlev[i] + 1 > r) r = nlev[i] + 1;
                    stamp[i] = j;
                    i = parent[i];
                }
            }
            rel[j] = r;
            const i32 pj = parent[j];
            const bool cont = pj >= 0 && cnt[j] == cnt[pj] + 1 && claimed[pj] < 0;
            if (cont) {
                claimed[pj] = c;
                continue;
            }
            // j closes its chain: sequence the members by release level
            if (cfirst[c] == j) {
                nlev[j] = r;
                continue;
            }
            mem.clear();
            for (i32 t = cfirst[c]; t >= 0; t = cnext[t]) mem.push_back(t);
            srt = mem;
            std::stable_sort(srt.begin(), srt.end(), [&](i32 x, i32 y) { return rel[x] < rel[y]; });
            if (srt != mem) moved = true;
            i32 lv = -1;
            for (i32 t : srt) {
                lv = std::max(lv + 1, rel[t]);
                nlev[t] = lv;
            }
        }
        if (moved) {
            // counting sort of the nodes by new level (stable: ties keep the old order)
            i32 maxl = 0;
            for (i32 j = 0; j < n; j++) maxl = std::max(maxl, nlev[j]);
            std::vector<i32> lp((size_t)maxl + 2, 0);
            for (i32 j = 0; j < n; j++) lp[nlev[j] + 1]++;
            for (i32 l = 0; l <= maxl; l++) lp[l + 1] += lp[l];
            std::vector<i64> p1((size_t)n);
            std::vector<i32> ip1((size_t)n);
            for (i32 j = 0; j < n; j++) {
                const i32 t = lp[nlev[j]]++;
                p1[t] = p0[j];
                ip1[p1[t]] = t;
            }
            std::vector<i64> Cp1;
            bigvec Ci1;
            std::vector<i32> parent1, cnt1, rowcnt1;
            permuted_triu(n, Ap, Ai, ip1, Cp1, Ci1);
            etree_counts(n, Cp1, Ci1, parent1, cnt1, &rowcnt1);
            i64 nnzL1 = 0;
            for (i32 j = 0; j < n; j++) nnzL1 += cnt1[j];
            if (nnzL1 <= nnzL0) { // (always, by the clique argument; checked because it is free)
                p0.swap(p1);
                ip0.swap(ip1);
                Cp.swap(Cp1);
                Ci.swap(Ci1);
                parent.swap(parent1);
                cnt.swap(cnt1);
                rowcnt.swap(rowcnt1);
            }
        }
    }
    std::vector<i32> level((size_t)n, 0);
    i32 depth = n > 0 ? 1 : 0;
    for (i32 j = 0; j < n; j++) {
        const i32 pj = parent[j];
        if (pj >= 0 && level[pj] < level[j] + 1) level[pj] = level[j] + 1;
        if (level[j] + 1 > depth) depth = level[j] + 1;
    }
    clk("pass A (etree, counts, chains)");
    // ---- cut the forest: a node whose whole subtree is small goes (with that subtree)
    //      into a "bundle" that ONE workgroup factors / solves start to finish; the
    //      remaining ancestors form the "top", processed level by level by the whole GPU.
    std::vector<i64> sub_nodes((size_t)n, 1), sub_ent((size_t)n, 0), sub_work((size_t)n, 0);
    for (i32 j = 0; j < n; j++) {
        sub_ent[j] = (i64)cnt[j] + rowcnt[j];
        sub_work[j] = (i64)cnt[j] * cnt[j]; // ~ multiply-adds of column j's updates
    }
    for (i32 j = 0; j < n; j++)
        if (parent[j] >= 0) {
            sub_nodes[parent[j]] += sub_nodes[j];
            sub_ent[parent[j]] += sub_ent[j];
            sub_work[parent[j]] += sub_work[j];
        }
    const bool no_bundles = switches().no_bundles;
    i64 max_work = BUNDLE_MAX_WORK; // (CHIP_BUNDLE_MAX_WORK: tuning / tests)
    if (switches().bundle_max_work > 0) max_work = switches().bundle_max_work;
    std::vector<char> top((size_t)n, 0);
    i64 NF = 0, maxsub = 0;
    auto cut_at = [&](i64 cap_nodes, std::vector<char> &tp) {
        i64 nf = 0;
        tp.assign((size_t)n, 0);
        for (i32 j = 0; j < n; j++) {
            // (a long column belongs to a dense front near the root: it is left to the top -- chain
            // supernodes -- even when its subtree is small; the top stays closed under "parent of")
            if (no_bundles || sub_nodes[j] > cap_nodes || sub_ent[j] > BUNDLE_MAX_ENTRIES || cnt[j] > BUNDLE_MAX_COL ||
                sub_work[j] > max_work)
                tp[j] = 1;
            if (tp[j] && parent[j] >= 0) tp[parent[j]] = 1;
            if (!tp[j]) nf++;
        }
        return nf;
    };
    NF = cut_at(BUNDLE_MAX_NODES, top);
    // ---- a finer cut for forests that would leave most of the device idle (grouped fold, host.hpp) ----
    // Tree root of every node; the top nodes of one tree form its GROUP.  While the cut yields fewer than 3/4 of
    // `target_wg` bundles and every group stays within TOPFOLD_MAX nodes, the node cap is halved: a tree whose
    // root region moves into the top falls apart into its subtrees, which are packed into several bundles per
    // tree below.  (An arrow -- ONE tree with a dense top, config 3 -- stops at once: its top would grow past 8.)
    std::vector<i32> root_of((size_t)n, 0);
    for (i32 j = (i32)n - 1; j >= 0; j--) root_of[j] = parent[j] < 0 ? j : root_of[parent[j]];
    bool grouped = false; // >= 2 trees with a non-empty top of <= TOPFOLD_MAX nodes each, no other top nodes
    {
        std::vector<i32> gcount((size_t)n, 0);
        auto groups_of = [&](const std::vector<char> &tp, i32 &ng, i32 &gmax) {
            ng = 0;
            gmax = 0;
            for (i32 j = 0; j < n; j++) gcount[j] = 0;
            for (i32 j = 0; j < n; j++)
                if (tp[j]) {
                    if (gcount[root_of[j]]++ == 0) ng++;
                    gmax = std::max(gmax, gcount[root_of[j]]);
                }
        };
        // bundles the packing below will make of a cut: per group about target * (its share of the forest
        // nodes), at least one per group / per ungrouped tree
        auto estimate_bundles = [&](const std::vector<char> &tp, i64 nf) {
            i64 nbe = 0;
            std::vector<i64> gn((size_t)n, 0), gsubs((size_t)n, 0), gmaxs((size_t)n, 0);
            for (i32 j = 0; j < n; j++)
                if (!tp[j]) {
                    gn[root_of[j]]++;
                    if (parent[j] < 0 || tp[parent[j]]) { // root of a complete subtree: indivisible
                        gsubs[root_of[j]]++;
                        gmaxs[root_of[j]] = std::max(gmaxs[root_of[j]], sub_nodes[j]);
                    }
                }
            for (i32 r = 0; r < n; r++)
                if (gn[r] > 0) {
                    const i64 want = (i64)target_wg * gn[r] / std::max<i64>(nf, 1);
                    const i64 fit = gn[r] / std::max<i64>(gmaxs[r], 256);
                    nbe += std::max<i64>(1, std::min<i64>(std::min<i64>(want, gsubs[r]), std::max<i64>(fit, 1)));
                }
            return nbe;
        };
        i32 ng = 0, gmax = 0;
        groups_of(top, ng, gmax);
        const bool refine_ok = target_wg > 0 && !no_bundles && perm0.empty() && !switches().no_groupfold;
        // (measured on an MI355X, config 4's shares: 128 trees -> 8 bundles per tree 0.48 ms against 0.53 ms per step
        // as whole trees; 256 trees -> 4 per tree 0.66 against 0.61; 512 -> 2 per tree 1.25 against 0.98: the finer cut
        // pays once a tree can be cut into about eight bundles -- CHIP_GROUPFOLD_MIN overrides the factor)
        i64 min_factor = 8;
        if (switches().groupfold_min > 0) min_factor = switches().groupfold_min;
        if (refine_ok && gmax <= TOPFOLD_MAX && estimate_bundles(top, NF) * min_factor <= (i64)target_wg) {
            i64 cap_nodes = BUNDLE_MAX_NODES;
            std::vector<char> cand;
            while (estimate_bundles(top, NF) * 4 < (i64)target_wg * 3 && cap_nodes >= 256) {
                cap_nodes /= 2;
                const i64 nf2 = cut_at(cap_nodes, cand);
                i32 ng2 = 0, gmax2 = 0;
                groups_of(cand, ng2, gmax2);
                if (gmax2 > TOPFOLD_MAX || nf2 == 0) break;
                if (nf2 != NF) {
                    top.swap(cand);
                    NF = nf2;
                    ng = ng2;
                    gmax = gmax2;
                }
            }
        }
        grouped = ng >= 2 && gmax <= TOPFOLD_MAX && refine_ok;
    }
    // subtree id of every forest node (roots = forest nodes whose parent is top or absent)
    std::vector<i32> sub((size_t)n, -1);
    i32 nsub = 0;
    for (i32 j = (i32)n - 1; j >= 0; j--) {
        if (top[j]) continue;
        if (parent[j] < 0 || top[parent[j]]) {
            sub[j] = nsub++;
            maxsub = std::max(maxsub, sub_nodes[j]);
        } else {
            sub[j] = sub[parent[j]];
        }
    }
    // subtrees in order of their first node (keeps the ordering's locality), packed greedily
    std::vector<i32> first((size_t)nsub, -1), sub_size((size_t)nsub, 0);
    std::vector<i64> sub_w((size_t)nsub, 0); // factorisation work of each subtree
    i64 forest_work = 0, maxsub_work = 0;
    for (i32 j = 0; j < n; j++)
        if (sub[j] >= 0) {
            if (first[sub[j]] < 0) first[sub[j]] = j;
            sub_size[sub[j]]++;
            sub_w[sub[j]] += (i64)cnt[j] * cnt[j];
            forest_work += (i64)cnt[j] * cnt[j];
        }
    for (i32 t = 0; t < nsub; t++) maxsub_work = std::max(maxsub_work, sub_w[t]);
    std::vector<i32> sorder((size_t)nsub);
    std::iota(sorder.begin(), sorder.end(), 0);
    std::sort(sorder.begin(), sorder.end(), [&](i32 a, i32 b) { return first[a] < first[b]; });
    i64 cap = (NF + BUNDLE_TARGET_COUNT - 1) / BUNDLE_TARGET_COUNT;
    cap = std::max<i64>(cap, maxsub);
    cap = std::max<i64>(cap, 256);
    cap = std::min<i64>(cap, BUNDLE_MAX_NODES);
    // ... and balanced by WORK as well: one workgroup factors a bundle from start to finish, so the launch
    // lasts as long as its heaviest bundle (config 2: a few bundles full of heavy subtrees took 17.8 ms)
    const i64 wcap = std::max<i64>(std::max<i64>(forest_work / BUNDLE_TARGET_COUNT, maxsub_work), 4096);
    std::vector<i32> bundle_of_sub((size_t)nsub, 0);
    i32 nb = 0;
    // grouped fold: group id of every tree with a non-empty top (by first top node), -1 otherwise
    std::vector<i32> grp_of_root;
    i32 ngroups = 0;
    std::vector<i32> grp_first_bundle; // gf_bptr
    if (grouped) {
        grp_of_root.assign((size_t)n, -1);
        for (i32 j = 0; j < n; j++)
            if (top[j] && grp_of_root[root_of[j]] < 0) grp_of_root[root_of[j]] = ngroups++;
        // subtrees by group (stable in `sorder`), the ungrouped ones last; per group its own node cap:
        // about target_wg * (share of the forest nodes) bundles, never below 256 nodes
        std::vector<i64> gnodes((size_t)ngroups, 0), gmaxsub((size_t)ngroups, 0), gwork((size_t)ngroups, 0);
        auto grp_of_sub = [&](i32 s) { return grp_of_root[root_of[first[s]]]; };
        for (i32 s = 0; s < nsub; s++) {
            const i32 g = grp_of_sub(s);
            if (g >= 0) {
                gnodes[g] += sub_size[s];
                gwork[g] += sub_w[s];
                gmaxsub[g] = std::max<i64>(gmaxsub[g], sub_size[s]);
            }
        }
        std::vector<std::vector<i32>> by_group((size_t)ngroups + 1);
        for (i32 s : sorder) {
            const i32 g = grp_of_sub(s);
            by_group[g >= 0 ? g : ngroups].push_back(s);
        }
        grp_first_bundle.assign((size_t)ngroups + 1, 0);
        for (i32 g = 0; g <= ngroups; g++) {
            if (g < ngroups) grp_first_bundle[g] = nb;
            i64 gcap = cap, gwcap = wcap;
            if (g < ngroups) {
                const i64 want = std::max<i64>(1, (i64)target_wg * gnodes[g] / std::max<i64>(NF, 1));
                // (+ the largest subtree: the greedy packing then fills every bundle beyond nodes / want, so
                // that at most `want` bundles come out)
                gcap = std::max<i64>((gnodes[g] + want - 1) / want + gmaxsub[g], 256);
                gcap = std::min<i64>(gcap, BUNDLE_MAX_NODES);
                gwcap = std::max<i64>(wcap, 2 * ((gwork[g] + want - 1) / want));
            }
            i64 cur = 0, curw = 0;
            for (i32 s : by_group[g]) {
                if (cur > 0 && (cur + sub_size[s] > gcap || curw + sub_w[s] > gwcap)) {
                    nb++;
                    cur = 0;
                    curw = 0;
                }
                bundle_of_sub[s] = nb;
                cur += sub_size[s];
                curw += sub_w[s];
            }
            if (cur > 0) nb++;
            if (g + 1 == ngroups) grp_first_bundle[ngroups] = nb;
        }
        if (ngroups == 0) grp_first_bundle[0] = 0;
    } else {
        i64 cur = 0, curw = 0;
        for (i32 s : sorder) {
            if (cur > 0 && (cur + sub_size[s] > cap || curw + sub_w[s] > wcap)) {
                nb++;
                cur = 0;
                curw = 0;
            }
            bundle_of_sub[s] = nb;
            cur += sub_size[s];
            curw += sub_w[s];
        }
        if (nsub > 0) nb++;
    }
    // final order: bundles first (level-major inside each), then the top level-major.
    // Two stable counting sorts: by level, then by group (bundle id, or nb for top nodes).
    std::vector<i32> bylevel((size_t)n);
    {
        std::vector<i32> cntl((size_t)depth + 1, 0);
        for (i32 j = 0; j < n; j++) cntl[level[j] + 1]++;
        for (i32 l = 0; l < depth; l++) cntl[l + 1] += cntl[l];
        for (i32 j = 0; j < n; j++) bylevel[cntl[level[j]]++] = j;
    }
    std::vector<i32> order((size_t)n);
    std::vector<i32> gptr((size_t)nb + 2, 0);
    {
        auto group = [&](i32 j) { return top[j] ? nb : bundle_of_sub[sub[j]]; };
        for (i32 j = 0; j < n; j++) gptr[group(j) + 1]++;
        for (i32 g = 0; g <= nb; g++) gptr[g + 1] += gptr[g];
        std::vector<i32> pos(gptr.begin(), gptr.end() - 1);
        for (i32 t = 0; t < n; t++) {
            const i32 j = bylevel[t];
            order[pos[group(j)]++] = j;
        }
    }
    // inside a bundle the nodes of one level are independent: order them by their ORIGINAL index, so that the
    // permutation becomes a handful of long ascending runs per bundle (config 3: the x block, the
    // Nonnegative rows and the second-order-cone rows of a block) -- the fused solve kernel then stages
    // its right-hand side and writes its result with coalesced copies instead of per-element gathers
    for (i32 g = 0; g < nb; g++) {
        i32 t = gptr[g];
        const i32 tend = gptr[g + 1];
        while (t < tend) {
            i32 e = t + 1;
            while (e < tend && level[order[e]] == level[order[t]]) e++;
            std::sort(order.begin() + t, order.begin() + e, [&](i32 a, i32 b) { return p0[a] < p0[b]; });
            t = e;
        }
    }
    S.perm.resize((size_t)n);
    S.iperm.resize((size_t)n);
    S.level.resize((size_t)n);
    for (i32 t = 0; t < n; t++) {
        S.perm[t] = (i32)p0[order[t]];
        S.level[t] = level[order[t]];
    }
    for (i32 j = 0; j < n; j++) S.iperm[S.perm[j]] = j;
    S.tree_depth = depth;
    S.NF = (i32)NF;
    S.bundle_ptr.assign(gptr.begin(), gptr.begin() + nb + 1);
    S.max_bundle_nodes = 0;
    S.blvl_ptr.assign((size_t)nb + 1, 0);
    for (i32 b = 0; b < nb; b++) {
        const i32 s0 = S.bundle_ptr[b], s1 = S.bundle_ptr[b + 1];
        S.max_bundle_nodes = std::max(S.max_bundle_nodes, s1 - s0);
        // boundaries of the (ascending) levels present in this bundle; level 0 always first
        i32 lv = 0;
        S.blvl.push_back(s0);
        for (i32 t = s0; t < s1; t++)
            while (S.level[t] > lv) {
                S.blvl.push_back(t);
                lv++;
            }
        S.blvl.push_back(s1);
        S.blvl_ptr[b + 1] = (i32)S.blvl.size();
    }
    // top levels (compressed: only the levels that occur among top nodes)
    std::vector<i32> lvlptr;
    lvlptr.push_back((i32)NF);
    for (i32 t = (i32)NF + 1; t < n; t++)
        if (S.level[t] != S.level[t - 1]) lvlptr.push_back(t);
    if (NF < n) lvlptr.push_back((i32)n);
    const i32 nlevels = (i32)lvlptr.size() - 1;
    S.nlevels = nlevels;
    S.lvlptr = lvlptr;
    S.dsigns.resize((size_t)n);
    for (i32 j = 0; j < n; j++) S.dsigns[j] = dsigns ? dsigns[S.perm[j]] : (int8_t)1;

    clk("forest cut, renumbering");
    // ---- pass B: final pattern ----------------------------------------------
    // (the final order is a topological re-sort of pass A's tree: same filled graph, so the tree and the
    // column counts are pass A's, renumbered; only the permuted pattern of K is built afresh)
    permuted_triu(n, Ap, Ai, S.iperm, Cp, Ci);
    {
        std::vector<i32> inv((size_t)n), parentB((size_t)n), cntB((size_t)n);
        for (i32 t = 0; t < n; t++) inv[order[t]] = t;
        for (i32 t = 0; t < n; t++) {
            const i32 j = order[t];
            parentB[t] = parent[j] >= 0 ? inv[parent[j]] : -1;
            cntB[t] = cnt[j];
        }
        parent.swap(parentB);
        cnt.swap(cntB);
    }
    S.etree = parent;
    clk("pass B (pattern of K, renumbered tree)");
    // ---- chain supernodes among the top columns ---------------------------------
    // A chain j -> parent(j) -> ... of top columns in which every node is the heaviest child of its
    // parent has nested structures: struct(j) is contained in {later chain nodes} + struct(last).
    // Padding every column of a chain segment to exactly that set (explicit zeros) turns the segment
    // into a dense trapezoid -- rows = [later members..., struct(last)...] in ascending order, so the
    // entry (panel row i, member t) sits at Lp[c_t] + i - t - 1 -- which k_factor_snode factors with
    // dense block operations instead of per-entry index gathers.  Only the pattern changes (a superset
    // of the true one; the added entries are exact zeros), so every other kernel works unchanged.
    std::vector<char> sn_skip((size_t)n, 0); // member that is not the last column of its supernode
    S.sn_of.assign((size_t)n, -1);
    S.sn_ptr.assign(1, 0);
    if (!switches().no_snode && S.NF < n) {
        const i32 NFi = S.NF;
        std::vector<i32> best((size_t)n, -1);
        for (i32 j = NFi; j < n; j++) {
            const i32 pj = parent[j];
            if (pj >= 0 && (best[pj] < 0 || cnt[j] > cnt[best[pj]])) best[pj] = j;
        }
        std::vector<char> linked((size_t)n, 0);
        for (i32 pj = NFi; pj < n; pj++)
            if (best[pj] >= 0) linked[best[pj]] = 1;
        // dense cone blocks (clique_of): a supernode never runs across a block's boundary.  The rows of a block are a
        // leaf subtree of the elimination tree; merged with the overlap / coupling variables above them -- which are the
        // parents of the NEIGHBOURING blocks' rows too -- a block's supernode would sit above its neighbours' supernodes
        // and the blocks, all independent, would be factored one unit level after the other.
        std::vector<i32> cqf;
        if (clique_of && (i64)clique_of->size() == n && !switches().no_clique_order) {
            cqf.resize((size_t)n);
            for (i32 t = 0; t < n; t++) cqf[(size_t)t] = (*clique_of)[(size_t)S.perm[(size_t)t]];
        }
        std::vector<i32> chain;
        std::vector<std::pair<i32, std::vector<i32>>> found; // (first column, columns ascending)
        for (i32 h = NFi; h < n; h++) {
            if (linked[h]) continue; // not the top end of a chain
            chain.clear();
            for (i32 j = h; j >= NFi; j = best[j]) {
                chain.push_back(j); // descending: last column first
                if (best[j] < 0) break;
            }
            size_t a = 0;
            while (a < chain.size()) {
                const i32 e = chain[a];
                size_t b = a + 1;
                while (b < chain.size() && (i32)(b - a) < SN_MAX_W) {
                    if (!cqf.empty() && cqf[(size_t)chain[b]] != cqf[(size_t)chain[b - 1]]) break;
                    const i64 padded = (i64)cnt[e] + (i64)(b - a);
                    const i64 zeros = padded - cnt[chain[b]];
                    if (zeros > std::max<i64>(SN_PAD_ABS, (i64)(SN_PAD_REL * (double)padded))) break;
                    b++;
                }
                if ((i32)(b - a) >= SN_MIN_W) {
                    std::vector<i32> cols(chain.begin() + a, chain.begin() + b);
                    std::reverse(cols.begin(), cols.end());
                    found.emplace_back(cols.front(), std::move(cols));
                }
                a = b;
            }
        }
        std::sort(found.begin(), found.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
        i64 upd_budget = SN_UPD_BUDGET;
        for (auto &f : found) {
            const i32 id = (i32)S.sn_ptr.size() - 1;
            const std::vector<i32> &cols = f.second;
            const i32 w = (i32)cols.size(), e = cols.back();
            const i64 upd = (i64)cnt[e] * (cnt[e] - 1) / 2; // slots of its update of the ancestors
            if (upd > upd_budget) continue;                  // (stays a run of ordinary columns)
            upd_budget -= upd;
            for (i32 t = 0; t < w; t++) {
                S.sn_of[cols[t]] = id;
                S.sn_col.push_back(cols[t]);
                if (t + 1 < w) {
                    sn_skip[cols[t]] = 1;
                    cnt[cols[t]] = cnt[e] + (w - 1 - t); // padded length
                }
            }
            S.sn_ptr.push_back((i32)S.sn_col.size());
        }
    }
    S.Lp.assign((size_t)n + 1, 0);
    i64 nnzL = 0;
    for (i32 j = 0; j < n; j++) {
        nnzL += cnt[j];
        if (nnzL >= ((i64)1 << 31) - n - 8) {
            set_error("nnz(L) out of int32 range");
            return -1;
        }
        S.Lp[j + 1] = (i32)nnzL;
    }
    S.nnzL = nnzL;
    S.Li.resize((size_t)nnzL + 1); // (bigvec: not zero filled; every entry is written below)
    S.Li[(size_t)nnzL] = 0;
    {
        // rows of L by row-subtree traversal; k ascending => ascending rows per column
        // (a supernode's members form a path of the tree and only its LAST column is written here -- the others
        // get the padded pattern below --, so the walk steps over a supernode in one hop: from any member
        // straight to the last one; rows that are themselves members of that supernode stop there.  Config 5:
        // the walk shrinks from nnz(L) = 3.9e8 steps to the compressed structure)
        std::vector<i32> nextp(S.Lp.begin(), S.Lp.end() - 1), stamp((size_t)n, -1), hop((size_t)n);
        for (i32 j = 0; j < n; j++) hop[j] = j;
        for (size_t sn = 0; sn + 1 < S.sn_ptr.size(); sn++) {
            const i32 e = S.sn_col[S.sn_ptr[sn + 1] - 1];
            for (i32 t = S.sn_ptr[sn]; t < S.sn_ptr[sn + 1]; t++) hop[S.sn_col[t]] = e;
        }
        for (i32 k = 0; k < n; k++) {
            stamp[hop[k]] = k;
            for (i64 p = Cp[k]; p < Cp[k + 1]; p++) {
                i32 i = Ci[p];
                while (i >= 0 && i < k) {
                    const i32 r = hop[i];
                    if (r >= k || stamp[r] == k) break;
                    stamp[r] = k;
                    S.Li[nextp[r]++] = k;
                    i = parent[r];
                }
            }
        }
        // padded members: later members of the supernode, then the structure of its last column
        for (size_t sn = 0; sn + 1 < S.sn_ptr.size(); sn++) {
            const i32 *cols = S.sn_col.data() + S.sn_ptr[sn];
            const i32 w = S.sn_ptr[sn + 1] - S.sn_ptr[sn], e = cols[w - 1];
            const i32 nb = S.Lp[e + 1] - S.Lp[e];
            for (i32 t = 0; t + 1 < w; t++) {
                i32 q = S.Lp[cols[t]];
                for (i32 u = t + 1; u < w; u++) S.Li[q++] = cols[u];
                for (i32 r = 0; r < nb; r++) S.Li[q++] = S.Li[S.Lp[e] + r];
            }
        }
    }
    clk("supernodes + pattern of L");
    // ---- CSR view of L ------------------------------------------------------
    // With chain supernodes only the entries in columns that are NOT supernode members are ever looked at
    // by rows (bundle columns, ordinary top columns: the left-looking column kernels and the forward gathers;
    // member columns act through the dense supernode kernels), so only those are listed -- config 5: a few
    // 10^5 of 3.9e8 entries.  Tpos of a member column's entry is unused.
    {
        const bool filter = S.sn_ptr.size() > 1;
        auto listed = [&](i32 k) { return !filter || S.sn_of[k] < 0; };
        const int T = par_threads(nnzL);
        S.Rp.assign((size_t)n + 1, 0);
        run_threads(T, [&](int t, int TT) {
            const i32 k0 = (i32)(n * t / TT), k1 = (i32)(n * (t + 1) / TT);
            for (i32 k = 0; k < n; k++) {
                if (!listed(k)) continue;
                for (i32 q = S.Lp[k]; q < S.Lp[k + 1]; q++) {
                    const i32 r = S.Li[q];
                    if (r >= k0 && r < k1) S.Rp[r + 1]++;
                }
            }
        });
        for (i32 j = 0; j < n; j++) S.Rp[j + 1] += S.Rp[j];
        S.nnzR = S.Rp[n];
        S.Rcol.resize((size_t)S.nnzR + 1);
        S.Rpos.resize((size_t)S.nnzR + 1);
        S.Tpos.resize((size_t)nnzL + 1); // (entries of member columns stay unwritten: nothing reads them)
        S.Rcol[(size_t)S.nnzR] = S.Rpos[(size_t)S.nnzR] = S.Tpos[(size_t)nnzL] = 0;
        std::vector<i32> nextp(S.Rp.begin(), S.Rp.end() - 1);
        const std::vector<int64_t> cuts = balanced_cuts(S.Rp.data(), n, T);
        run_threads(T, [&](int t, int) { // threads own ranges of rows; columns scanned in order by all
            const i32 k0 = (i32)cuts[t], k1 = (i32)cuts[t + 1];
            if (k0 >= k1) return;
            for (i32 k = 0; k < n; k++) {
                if (!listed(k)) continue;
                for (i32 q = S.Lp[k]; q < S.Lp[k + 1]; q++) {
                    const i32 r = S.Li[q];
                    if (r < k0 || r >= k1) continue;
                    const i32 u = nextp[r]++;
                    S.Rcol[u] = k;
                    S.Rpos[u] = q;
                    S.Tpos[q] = u;
                }
            }
        });
    }
    clk("CSR view of L");
    // ---- the permuted upper triangle of K, twice sorted (three counting passes, no comparisons):
    //      T : by smaller index lo, larger index hi ascending;  C2 : by hi, lo ascending; each entry
    //      carries its position p in the caller's K.nzval.  Feeds the scatter map and the residual rows.
    std::vector<i32> Tp((size_t)n + 1, 0), C2p((size_t)n + 1, 0);
    RawBuf<i32> Thi((size_t)nnzK + 1), Tsrc((size_t)nnzK + 1), C2lo((size_t)nnzK + 1), C2src((size_t)nnzK + 1); // (host.hpp: no zero fill)
    {
        // (each pass: threads own ranges of destination keys and scan the whole source in order; the
        // permuted (lo, hi) of every entry is computed once, by source chunks, so that the scans are streams)
        const int T = par_threads(nnzK);
        std::vector<i32> hp((size_t)n + 1, 0);
        RawBuf<i32> hlo((size_t)nnzK + 1), hsrc((size_t)nnzK + 1), elo((size_t)nnzK + 1), ehi((size_t)nnzK + 1);
        {
            const std::vector<int64_t> ccuts = balanced_cuts(Ap, n, T);
            run_threads(T, [&](int t, int) {
                for (i64 c = ccuts[t]; c < ccuts[t + 1]; c++) {
                    const i32 pc = S.iperm[c];
                    for (i64 p = Ap[c]; p < Ap[c + 1]; p++) {
                        const i32 pr = S.iperm[Ai[p]];
                        elo[p] = pr < pc ? pr : pc;
                        ehi[p] = pr < pc ? pc : pr;
                    }
                }
            });
        }
        // three stable bucket passes (host.hpp: stable_buckets), each over chunks of its source in order:
        //   1. by hi (source order)        -> (hlo, hsrc) grouped by hi
        //   2. by lo, from the hi groups   -> T  : by lo, hi ascending
        //   3. by hi, from T               -> C2 : by hi, lo ascending
        stable_buckets(T, (i32)n, hp, true,
                       [&](int t, int TT, auto f) {
                           for (i64 p = nnzK * t / TT; p < nnzK * (t + 1) / TT; p++) f(ehi[p], p);
                       },
                       [&](i32, i64 p, i32 u) {
                           hlo[u] = elo[p];
                           hsrc[u] = (i32)p;
                       });
        {
            const std::vector<int64_t> hcuts = balanced_cuts(hp.data(), n, T);
            stable_buckets(T, (i32)n, Tp, true,
                           [&](int t, int, auto f) {
                               for (i32 hi = (i32)hcuts[t]; hi < (i32)hcuts[t + 1]; hi++)
                                   for (i32 w = hp[hi]; w < hp[hi + 1]; w++) f(hlo[w], ((i64)hi << 32) | (uint32_t)w);
                           },
                           [&](i32, i64 pl, i32 u) {
                               Thi[u] = (i32)(pl >> 32);
                               Tsrc[u] = hsrc[(uint32_t)pl];
                           });
        }
        C2p = hp;
        {
            const std::vector<int64_t> tcuts = balanced_cuts(Tp.data(), n, T);
            stable_buckets(T, (i32)n, C2p, false,
                           [&](int t, int, auto f) {
                               for (i32 lo = (i32)tcuts[t]; lo < (i32)tcuts[t + 1]; lo++)
                                   for (i32 u = Tp[lo]; u < Tp[lo + 1]; u++) f(Thi[u], ((i64)lo << 32) | (uint32_t)u);
                           },
                           [&](i32, i64 pl, i32 w) {
                               C2lo[w] = (i32)(pl >> 32);
                               C2src[w] = (i32)(uint32_t)pl; // position in V (T order)
                           });
        }
    }
    clk("T / C2 orders of K");
    // ---- the device's value store V = K.nzval in T order (row-wise by the smaller permuted index,
    //      diagonal first, ancestors ascending): V[u] = K.nzval[Tsrc[u]].  Rows lo < NF of V are the U
    //      rows the bundle kernels stream (residual AND the initial values of the bundle columns of the
    //      factorisation); rows lo >= NF (entries with both ends in the top) are scattered into the top
    //      columns of L by v2l: merge of row lo of T with column lo of L ----------
    S.k2v.resize((size_t)nnzK + 1);
    S.k2v[(size_t)nnzK] = 0;
    run_threads(par_threads(nnzK), [&](int t, int TT) {
        for (i64 u = nnzK * t / TT; u < nnzK * (t + 1) / TT; u++) S.k2v[Tsrc[u]] = (i32)u;
    });
    S.v2k.assign(Tsrc.data(), Tsrc.data() + nnzK);
    S.Vp = Tp;
    {
        const i32 NFi = S.NF;
        const i32 u0 = Tp[NFi];
        S.v2l.resize((size_t)(nnzK - u0) + 1);
        S.v2l[(size_t)(nnzK - u0)] = 0;
        // (only the rows of the top: the bundle columns merge their U rows inside k_bundle_factor)
        const i64 q0 = S.Lp[NFi];
        RawBuf<char> covered((size_t)(nnzL - q0) + 1); // (zeroed by the threads: 3.9e8 flags for config 5)
        {
            const i64 span = nnzL - q0 + 1;
            run_threads(par_threads(span), [&](int t, int TT) {
                std::fill(covered.data() + span * t / TT, covered.data() + span * (t + 1) / TT, (char)0);
            });
        }
        const int T = par_threads(nnzK - u0);
        std::vector<int64_t> cuts = balanced_cuts(Tp.data() + NFi, (i64)n - NFi, T); // rows NF.., by their entries
        std::vector<int> bad((size_t)T, 0);
        run_threads(T, [&](int t, int) {
            for (i32 lo = NFi + (i32)cuts[t]; lo < NFi + (i32)cuts[t + 1]; lo++) {
                i32 q = S.Lp[lo];
                const i32 qe = S.Lp[lo + 1];
                for (i32 u = Tp[lo]; u < Tp[lo + 1]; u++) {
                    const i32 hi = Thi[u];
                    if (hi == lo) {
                        S.v2l[u - u0] = (i32)(nnzL + lo);
                        continue;
                    }
                    while (q < qe && S.Li[q] < hi) q++;
                    if (q >= qe || S.Li[q] != hi) {
                        bad[t] = 1;
                        return;
                    }
                    S.v2l[u - u0] = q;
                    covered[(size_t)(q - q0)] = 1;
                }
            }
        });
        for (int t = 0; t < T; t++)
            if (bad[t]) {
                set_error("internal: K entry missing from the pattern of L");
                return -9;
            }
        // fill slots of the TOP columns (the bundle kernels zero their own while they merge the U rows)
        i64 nuncovered = 0;
        {
            const int Tc = par_threads(nnzL - q0);
            std::vector<i64> cnt((size_t)Tc, 0);
            run_threads(Tc, [&](int t, int TT) {
                const i64 span = nnzL - q0;
                i64 c = 0;
                for (i64 q = q0 + span * t / TT; q < q0 + span * (t + 1) / TT; q++) c += covered[(size_t)(q - q0)] ? 0 : 1;
                cnt[(size_t)t] = c;
            });
            for (int t = 0; t < Tc; t++) nuncovered += cnt[(size_t)t];
        }
        // (CHIP_FILL_RANGE_MIN, tests: 0 = always the range form, whatever the share of fill-in)
        const long long fmin = switches().fill_range_min;
        if (fmin == 0 || (nuncovered * 2 > nnzL - q0 && nuncovered > (i64)fmin)) {
            S.fill_from = q0; // (mostly fill-in: the range is cleared as a whole, no index list)
        } else {
            const int Tf = par_threads(nnzL - q0);
            std::vector<std::vector<i32>> parts((size_t)Tf);
            run_threads(Tf, [&](int t, int TT) {
                const i64 span = nnzL - q0;
                for (i64 q = q0 + span * t / TT; q < q0 + span * (t + 1) / TT; q++)
                    if (!covered[(size_t)(q - q0)]) parts[(size_t)t].push_back((i32)q);
            });
            for (int t = 0; t < Tf; t++) S.fill_idx.insert(S.fill_idx.end(), parts[(size_t)t].begin(), parts[(size_t)t].end());
        }
    }
    clk("v2l map, fill slots");
    // ---- K for the refinement residual e = b - K x (permuted numbering) ------
    // Every nonzero K_ij (i < j in the final numbering) joins a node to one of its ANCESTORS,
    // so i and j are in the same bundle or j is a top node.  Bundle rows therefore use the
    // symmetric matrix stored ONCE, row-wise by the smaller index (U: row i holds its diagonal
    // and its entries to ancestors); one workgroup per bundle applies each entry in both
    // directions inside LDS.  Only the (few) top rows keep a full row-wise copy (S).
    {
        const i32 NFi = S.NF;
        // U: rows lo < NF = rows of T (diagonal first, ancestors ascending)
        S.Up.assign(Tp.begin(), Tp.begin() + NFi + 1);
        S.nnzU = S.Up[NFi];
        S.Ucol.assign(Thi.data(), Thi.data() + S.nnzU);
        S.Ucol.push_back(0);
        // Dense diagonal blocks of the top: top nodes p_0 < p_1 < ... < p_{m-1} that K joins pairwise (a PSD or dense
        // second-order cone's Hs block) and whose rows of T list each other FIRST: row p_a = (diagonal, p_{a+1}, ...,
        // p_{m-1}, then whatever else).  The block's strict upper triangle then sits in V row by row, each row contiguous
        // -- the residual multiplies it straight from there (k_dblk_symv: every entry read ONCE, no index arrays), and
        // the full-row copy S below leaves these entries out.  Config 5: 200 blocks of 1275 hold 3.2e8 of S's 3.3e8
        // entries; S cost 12 bytes per entry per product and an uncoalesced gather per refactor.
        std::vector<i32> dblk_of((size_t)n, -1);
        S.dblk_p0.clear();
        S.dblk_m.clear();
        S.dblk_start.clear();
        S.dblk_node.clear();
        if (!switches().no_dense_symv) {
            for (i32 r = NFi; r < n; r++) {
                if (dblk_of[(size_t)r] >= 0) continue;
                i32 m = std::min<i32>(Tp[r + 1] - Tp[r], DBLK_MAX_M); // (the diagonal comes first: row length = 1 + others)
                if (m < DBLK_MIN_M) continue;
                const i32 *row0 = &Thi[Tp[r]];                        // row0[a] = p_a for a >= 1
                for (i32 a = 1; a < m; a++) {
                    const i32 q = row0[a];
                    if (dblk_of[(size_t)q] >= 0) {
                        m = a;
                        break;
                    }
                    const i32 want = m - 1 - a, have = std::min<i32>(Tp[q + 1] - Tp[q] - 1, want);
                    const i32 *rq = &Thi[Tp[q] + 1];
                    i32 k = 0;
                    while (k < have && rq[k] == row0[a + 1 + k]) k++;
                    if (k < want) m = a + 1 + k;
                }
                if (m < DBLK_MIN_M) continue;
                for (i32 a = 0; a < m; a++) {
                    const i32 q = a ? row0[a] : r;
                    dblk_of[(size_t)q] = (i32)S.dblk_p0.size();
                    S.dblk_node.push_back(q);
                    S.dblk_start.push_back(Tp[q] + 1);
                }
                S.dblk_p0.push_back(r);
                S.dblk_m.push_back(m);
            }
            i64 ent = 0;
            for (i32 m : S.dblk_m) ent += (i64)m * (m - 1) / 2;
            if (ent < (switches().dense_symv_min > 0 ? switches().dense_symv_min : DBLK_MIN_ENTRIES)) { // (not worth two more launches per product)
                S.dblk_p0.clear();
                S.dblk_m.clear();
                S.dblk_start.clear();
                S.dblk_node.clear();
                std::fill(dblk_of.begin(), dblk_of.end(), -1);
            }
            if (switches().timing)
                std::fprintf(stderr, "[chip analyse] dense diagonal blocks of the top: %d (%lld entries leave the full rows)\n",
                             (int)S.dblk_p0.size(), (long long)(S.dblk_p0.empty() ? 0 : ent));
        }
        auto in_block = [&](i32 r, i32 c) { return dblk_of[(size_t)r] >= 0 && dblk_of[(size_t)r] == dblk_of[(size_t)c] && r != c; };
        // S: full rows r >= NF = column r of C2 (lo <= r ascending, diagonal last) then row r of T without its diagonal
        S.Sp.assign((size_t)n + 1, 0);
        run_threads(par_threads((i64)Tp[n] - Tp[NFi]), [&](int t, int TT) {
            const i64 span = (i64)n - NFi;
            for (i32 r = NFi + (i32)(span * t / TT); r < NFi + (i32)(span * (t + 1) / TT); r++) {
                i32 cntr = 0;
                for (i32 w = C2p[r]; w < C2p[r + 1]; w++) cntr += !in_block(r, C2lo[w]);
                for (i32 u = Tp[r]; u < Tp[r + 1]; u++) cntr += Thi[u] != r && !in_block(r, Thi[u]);
                S.Sp[r + 1] = cntr;
            }
        });
        for (i32 j = 0; j < n; j++) S.Sp[j + 1] += S.Sp[j];
        S.nnzS = S.Sp[n];
        S.Scol.resize((size_t)S.nnzS + 1);
        S.Smap.resize((size_t)S.nnzS + 1);
        S.Scol[(size_t)S.nnzS] = S.Smap[(size_t)S.nnzS] = 0;
        const int T = par_threads(S.nnzS);
        const std::vector<int64_t> cuts = balanced_cuts(S.Sp.data() + NFi, (i64)n - NFi, T);
        run_threads(T, [&](int t, int) {
            for (i32 r = NFi + (i32)cuts[t]; r < NFi + (i32)cuts[t + 1]; r++) {
                i32 o = S.Sp[r];
                for (i32 w = C2p[r]; w < C2p[r + 1]; w++) {
                    if (in_block(r, C2lo[w])) continue;
                    S.Scol[o] = C2lo[w];
                    S.Smap[o++] = C2src[w];
                }
                for (i32 u = Tp[r]; u < Tp[r + 1]; u++)
                    if (Thi[u] != r && !in_block(r, Thi[u])) {
                        S.Scol[o] = Thi[u];
                        S.Smap[o++] = u;
                    }
            }
        });
    }
    clk("U / S rows of K");
    // ---- few dense top rows folded into the bundle kernels ----------------------
    {
        const i32 ntop = (i32)n - S.NF;
        const i32 nbun = S.bundle_ptr.empty() ? 0 : (i32)S.bundle_ptr.size() - 1;
        if (ntop >= 1 && ntop <= TOPFOLD_MAX && nbun > 0 && !switches().no_topfold) {
            const i32 k = ntop;
            S.nfold = k;
            S.fold_rseg.assign((size_t)nbun * k * 2, 0);
            for (i32 i = 0; i < k; i++) {
                const i32 r = S.NF + i;
                const i32 *rb = S.Rcol.data() + S.Rp[r], *re = S.Rcol.data() + S.Rp[r + 1];
                for (i32 b = 0; b < nbun; b++) {
                    S.fold_rseg[((size_t)b * k + i) * 2] = (i32)(std::lower_bound(rb, re, S.bundle_ptr[b]) - S.Rcol.data());
                    S.fold_rseg[((size_t)b * k + i) * 2 + 1] =
                        (i32)(std::lower_bound(rb, re, S.bundle_ptr[b + 1]) - S.Rcol.data());
                }
            }
            S.fold_tt.assign((size_t)k * k, -1);
            for (i32 j = 0; j < k; j++)
                for (i32 q = S.Lp[S.NF + j]; q < S.Lp[S.NF + j + 1]; q++) S.fold_tt[(size_t)(S.Li[q] - S.NF) * k + j] = q;
            S.fold_sp.assign((size_t)k + 1, 0);
            for (i32 i = 0; i < k; i++) {
                const i32 r = S.NF + i;
                for (i32 t = S.Sp[r]; t < S.Sp[r + 1]; t++)
                    if (S.Scol[t] >= S.NF) {
                        S.fold_scol.push_back(S.Scol[t] - S.NF);
                        S.fold_sslot.push_back(S.Smap[t]); // position in V
                    }
                S.fold_sp[i + 1] = (i32)S.fold_scol.size();
            }
        }
    }
    // ---- grouped fold (host.hpp): per tree with a non-empty top ------------------------------
    std::vector<i32> gpos; // index of a top node inside its group (by node - NF), grouped fold only
    if (grouped && S.nfold == 0 && ngroups >= 2 && !switches().no_topfold) {
        const i32 nbun = (i32)S.bundle_ptr.size() - 1;
        bool ok = nbun > 0 && S.dblk_p0.empty(); // (dense blocks of the top are not in the rows S the groups read)
        for (i32 g = 0; g < ngroups && ok; g++) ok = grp_first_bundle[g + 1] > grp_first_bundle[g];
        // group of every top node through its tree root (root_of / grp_of_root are in pass A's numbering:
        // order[t] = pass-A node of final node t)
        std::vector<i32> gof((size_t)(n - S.NF), -1);
        if (ok) {
            S.gf_ptr.assign((size_t)ngroups + 1, 0);
            for (i32 t = S.NF; t < n; t++) {
                const i32 g = grp_of_root[root_of[order[t]]];
                if (g < 0) {
                    ok = false;
                    break;
                }
                gof[t - S.NF] = g;
                S.gf_ptr[g + 1]++;
            }
        }
        if (ok) {
            for (i32 g = 0; g < ngroups; g++) S.gf_ptr[g + 1] += S.gf_ptr[g];
            S.gf_node.assign((size_t)(n - S.NF), 0);
            gpos.assign((size_t)(n - S.NF), 0);
            std::vector<i32> nx(S.gf_ptr.begin(), S.gf_ptr.end() - 1);
            for (i32 t = S.NF; t < n; t++) {
                const i32 g = gof[t - S.NF];
                gpos[t - S.NF] = nx[g] - S.gf_ptr[g];
                S.gf_node[nx[g]++] = t;
            }
            // every bundle column / U row may only reach the top nodes of its OWN group
            for (i32 g = 0; g < ngroups && ok; g++)
                for (i32 b = grp_first_bundle[g]; b < grp_first_bundle[g + 1] && ok; b++)
                    for (i32 j = S.bundle_ptr[b]; j < S.bundle_ptr[b + 1] && ok; j++)
                        for (i32 q = S.Lp[j]; q < S.Lp[j + 1]; q++)
                            if (S.Li[q] >= S.NF && gof[S.Li[q] - S.NF] != g) {
                                ok = false;
                                break;
                            }
            for (i32 b = grp_first_bundle[ngroups]; b < nbun && ok; b++)
                for (i32 j = S.bundle_ptr[b]; j < S.bundle_ptr[b + 1] && ok; j++)
                    if (S.Lp[j + 1] > S.Lp[j] && S.Li[S.Lp[j + 1] - 1] >= S.NF) ok = false;
        }
        if (ok) {
            S.gf_ng = ngroups;
            S.gf_bptr = grp_first_bundle;
            S.gf_tt.assign((size_t)ngroups * 64, -1);
            S.gf_sp.assign((size_t)(n - S.NF) + 1, 0);
            for (i32 g = 0; g < ngroups; g++) {
                const i32 k = S.gf_ptr[g + 1] - S.gf_ptr[g];
                for (i32 j = 0; j < k; j++) {
                    const i32 cj = S.gf_node[S.gf_ptr[g] + j];
                    for (i32 q = S.Lp[cj]; q < S.Lp[cj + 1]; q++) S.gf_tt[(size_t)g * 64 + (size_t)gpos[S.Li[q] - S.NF] * 8 + j] = q;
                }
                for (i32 i = 0; i < k; i++) {
                    const i32 r = S.gf_node[S.gf_ptr[g] + i];
                    for (i32 t = S.Sp[r]; t < S.Sp[r + 1]; t++)
                        if (S.Scol[t] >= S.NF) {
                            S.gf_scol.push_back(gpos[S.Scol[t] - S.NF]);
                            S.gf_sslot.push_back(S.Smap[t]); // position in V
                        }
                    S.gf_sp[(size_t)S.gf_ptr[g] + i + 1] = (i32)S.gf_scol.size();
                }
            }
        } else {
            S.gf_ptr.clear();
            S.gf_node.clear();
            gpos.clear();
        }
    }
    if (clk.on)
        std::fprintf(stderr, "[chip analyse] bundles %d (max %d nodes), top %d nodes, fold k = %d, grouped fold: %d groups\n",
                     (int)S.bundle_ptr.size() - 1, (int)S.max_bundle_nodes, (int)(n - S.NF), (int)S.nfold, (int)S.gf_ng);
    // ---- 16-bit bundle-local indices for the fused solve kernel (k_bundle_ir) ----------------
    // Systems that are bundles plus at most a folded top: row index i of an entry of a bundle column /
    // U row becomes i - s0 (its bundle starts at s0) when i lies in the bundle, nloc + (i - NF) for the
    // (at most TOPFOLD_MAX) top rows -- 2 bytes per entry instead of 4 in all three sweeps.
    {
        const i32 nbun = S.bundle_ptr.empty() ? 0 : (i32)S.bundle_ptr.size() - 1;
        const bool eligible = nbun > 0 && (S.nfold > 0 || S.gf_ng > 0 || S.NF == n) && S.max_bundle_nodes + TOPFOLD_MAX < 65535;
        // (grouped fold: a top row is addressed by its index inside the bundle's group)
        auto top_index = [&](i32 i) { return S.gf_ng > 0 ? gpos[i - S.NF] : i - S.NF; };
        if (eligible) {
            const i64 nLb = S.Lp[S.NF];
            S.Li16.resize((size_t)nLb + 1);
            S.Ucol16.resize((size_t)S.nnzU + 1);
            S.Lj16.resize((size_t)nLb + 1);
            S.Urow16.resize((size_t)S.nnzU + 1);
            S.Rk16.resize((size_t)S.Rp[S.NF] + 1);
            S.Ro16.resize((size_t)S.Rp[S.NF] + 1);
            for (i32 b = 0; b < nbun; b++) {
                const i32 s0 = S.bundle_ptr[b], s1 = S.bundle_ptr[b + 1], nloc = s1 - s0;
                for (i32 j = s0; j < s1; j++) {
                    for (i32 q = S.Lp[j]; q < S.Lp[j + 1]; q++) {
                        const i32 i = S.Li[q];
                        S.Li16[q] = (uint16_t)(i < s1 ? i - s0 : nloc + top_index(i));
                        S.Lj16[q] = (uint16_t)(j - s0);
                    }
                    for (i32 t = S.Rp[j]; t < S.Rp[j + 1]; t++) { // (row j of L: columns k of the same bundle)
                        const i32 k = S.Rcol[t];
                        S.Rk16[t] = (uint16_t)(k - s0);
                        S.Ro16[t] = (uint16_t)std::min<i32>(S.Rpos[t] - S.Lp[k], 65535);
                    }
                    for (i32 u = S.Up[j]; u < S.Up[j + 1]; u++) {
                        const i32 i = S.Ucol[u];
                        S.Ucol16[u] = (uint16_t)(i < s1 ? i - s0 : nloc + top_index(i));
                        S.Urow16[u] = (uint16_t)(j - s0);
                    }
                }
            }
        }
    }
    // ---- ... and for the entry-parallel sweeps of the stand-alone bundle kernels (systems with a level-scheduled top) ----
    {
        const i32 nbun = S.bundle_ptr.empty() ? 0 : (i32)S.bundle_ptr.size() - 1;
        S.max_bundle_levels = 0;
        for (i32 b = 0; b < nbun; b++) S.max_bundle_levels = std::max(S.max_bundle_levels, S.blvl_ptr[b + 1] - S.blvl_ptr[b] - 1);
        if (nbun > 0 && S.Li16.empty() && S.nfold == 0 && S.gf_ng == 0 && S.max_bundle_nodes < 65535 && !switches().no_bundle_flat_sweep) {
            const i64 nLb = S.Lp[S.NF];
            S.sLi16.resize((size_t)nLb + 1);
            S.sLj16.resize((size_t)nLb + 1);
            S.Urow16.resize((size_t)S.nnzU + 1); // (the entry-parallel factorisation finds a diagonal entry's node here)
            const int T = par_threads(nLb);
            run_threads(T, [&](int t, int TT) {
                for (i32 b = t; b < nbun; b += TT) {
                    const i32 s0 = S.bundle_ptr[b], s1 = S.bundle_ptr[b + 1];
                    for (i32 j = s0; j < s1; j++) {
                        for (i32 q = S.Lp[j]; q < S.Lp[j + 1]; q++) {
                            const i32 i = S.Li[q];
                            S.sLi16[q] = (uint16_t)(i < s1 ? i - s0 : 0xFFFF);
                            S.sLj16[q] = (uint16_t)(j - s0);
                        }
                        for (i32 u = S.Up[j]; u < S.Up[j + 1]; u++) S.Urow16[u] = (uint16_t)(j - s0);
                    }
                }
            });
        }
    }
    // ---- update records of the entry-parallel bundle factorisation (host.hpp: fu_rec) ----------------
    // (systems with a level-scheduled top take the records too when it has chain supernodes -- nothing there reads the
    // row-major mirror of the bundle columns that this form of the factorisation does not keep -- and they stay small)
    const bool flat_unfused = S.Li16.empty() && !S.sLi16.empty() && S.sn_ptr.size() > 1;
    if ((!S.Li16.empty() || flat_unfused) && !switches().no_factor_flat) {
        const i32 nbun = (i32)S.bundle_ptr.size() - 1;
        bool ok = true;
        for (i32 b = 0; b < nbun && ok; b++) {
            const i32 s0 = S.bundle_ptr[b], s1 = S.bundle_ptr[b + 1];
            ok = (i64)(S.Lp[s1] - S.Lp[s0]) + (s1 - s0) < 65535;
        }
        if (ok) {
            // per bundle: count, then fill (threads over bundles: every bundle owns its range of the arrays)
            std::vector<i64> bcount((size_t)nbun + 1, 0);
            const int T = par_threads(S.Lp[S.NF]);
            auto pairs_of = [&](i32 k, i32 s1) { // pairs (a >= b) of column k with r_b inside the bundle
                i64 c = 0;
                const i32 cb = S.Lp[k], ce = S.Lp[k + 1];
                i32 nin = 0;
                while (cb + nin < ce && S.Li[cb + nin] < s1) nin++;
                c = (i64)nin * (ce - cb) - (i64)nin * (nin - 1) / 2; // b over the nin in-bundle rows, a from b to the end
                if (S.nfold == 1 && cb + nin < ce) c += 1; // (the single folded top row: its pivot share l_tk^2 d_k)
                return c;
            };
            run_threads(T, [&](int t, int TT) {
                for (i32 b = t; b < nbun; b += TT) {
                    i64 c = 0;
                    for (i32 k = S.bundle_ptr[b]; k < S.bundle_ptr[b + 1]; k++) c += pairs_of(k, S.bundle_ptr[b + 1]);
                    bcount[b + 1] = c;
                }
            });
            for (i32 b = 0; b < nbun; b++) bcount[b + 1] += bcount[b];
            if (bcount[nbun] < (flat_unfused ? (i64)48 << 20 : (i64)1 << 30)) {
                S.fu_rec.resize((size_t)bcount[nbun] * 4 + 4);
                S.fu_ptr.assign(S.blvl.size() + 1, 0);
                S.fu_slot.assign((size_t)S.nnzU + 1, 0xFFFF);
                run_threads(T, [&](int t, int TT) {
                    std::vector<std::array<uint16_t, 4>> lvl;
                    for (i32 b = t; b < nbun; b += TT) {
                        const i32 s0 = S.bundle_ptr[b], s1 = S.bundle_ptr[b + 1], e0 = S.Lp[s0], nE = S.Lp[s1] - e0;
                        i64 out = bcount[b];
                        const i32 lb0 = S.blvl_ptr[b], nl = S.blvl_ptr[b + 1] - lb0 - 1;
                        for (i32 l = 0; l < nl; l++) {
                            S.fu_ptr[lb0 + l] = (i32)out;
                            lvl.clear();
                            for (i32 k = S.blvl[lb0 + l]; k < S.blvl[lb0 + l + 1]; k++) {
                                const i32 cb = S.Lp[k], ce = S.Lp[k + 1];
                                for (i32 qb = cb; qb < ce && S.Li[qb] < s1; qb++) {
                                    const i32 j = S.Li[qb]; // target column r_b (in the bundle)
                                    const i32 *jb = S.Li.data() + S.Lp[j], *je = S.Li.data() + S.Lp[j + 1];
                                    for (i32 qa = qb; qa < ce; qa++) {
                                        uint16_t tgt;
                                        if (qa == qb) tgt = (uint16_t)(nE + (j - s0));
                                        else tgt = (uint16_t)((std::lower_bound(jb, je, S.Li[qa]) - S.Li.data()) - e0);
                                        lvl.push_back({(uint16_t)(qa - e0), (uint16_t)(qb - e0), (uint16_t)(k - s0), tgt});
                                    }
                                }
                                // one folded top row (the last entry of a column that reaches it): its pivot takes
                                // l_tk (l_tk d_k) like any other target -- slot nE + nloc, behind the bundle's pivots
                                if (S.nfold == 1 && ce > cb && S.Li[ce - 1] >= s1)
                                    lvl.push_back({(uint16_t)(ce - 1 - e0), (uint16_t)(ce - 1 - e0), (uint16_t)(k - s0),
                                                   (uint16_t)(nE + (s1 - s0))});
                            }
                            std::stable_sort(lvl.begin(), lvl.end(),
                                             [](const std::array<uint16_t, 4> &x, const std::array<uint16_t, 4> &y) { return x[3] < y[3]; });
                            for (const auto &r : lvl) {
                                uint16_t *dst = S.fu_rec.data() + (size_t)out * 4;
                                dst[0] = r[0];
                                dst[1] = r[1];
                                dst[2] = r[2];
                                dst[3] = r[3];
                                out++;
                            }
                        }
                        S.fu_ptr[lb0 + nl] = (i32)out; // (== bcount[b + 1]: also the first range of the next bundle)
                        // where the U entries land: row j of U = (diagonal, entries (j, i) to ancestors i) -> slot of row i in column j
                        for (i32 j = s0; j < s1; j++) {
                            const i32 *jb = S.Li.data() + S.Lp[j], *je = S.Li.data() + S.Lp[j + 1];
                            for (i32 u = S.Up[j] + 1; u < S.Up[j + 1]; u++)
                                S.fu_slot[u] = (uint16_t)((std::lower_bound(jb, je, S.Ucol[u]) - S.Li.data()) - e0);
                        }
                    }
                });
            }
        }
    }
    // ---- blocked substitution for tall tops -----------------------------------
    {
        const i32 ntop = (i32)n - S.NF;
        const bool off = switches().no_topblk;
        // worthwhile only for chain-like tops: a block step costs ~3x a level step, so the number of
        // blocks must be well below the number of levels (config 2: 272 blocks for 4383 levels;
        // config 5's wide levels -- 200 rows each -- stay level scheduled)
        const i64 nblk = (ntop + TOPBLK - 1) / TOPBLK;
        const bool has_sn = S.sn_ptr.size() > 1; // (with chain supernodes the top is solved through them)
        if (!off && !has_sn && ntop >= 4 * TOPBLK && nlevels >= 256 && 3 * nblk < (i64)nlevels) {
            S.topblk = TOPBLK;
            S.Rsplit.resize((size_t)ntop);
            S.Lsplit.resize((size_t)ntop);
            for (i32 j = S.NF; j < n; j++) {
                const i32 b = (j - S.NF) / TOPBLK, r0 = S.NF + b * TOPBLK;
                const i32 r1 = std::min<i32>((i32)n, r0 + TOPBLK);
                const i32 *rb = S.Rcol.data() + S.Rp[j], *re = S.Rcol.data() + S.Rp[j + 1];
                S.Rsplit[j - S.NF] = (i32)(std::lower_bound(rb, re, r0) - S.Rcol.data());
                const i32 *cb = S.Li.data() + S.Lp[j], *ce = S.Li.data() + S.Lp[j + 1];
                S.Lsplit[j - S.NF] = (i32)(std::lower_bound(cb, ce, r1) - S.Li.data());
            }
        }
    }
    clk("fold / topblk");
    // ---- per-level work lists ----------------------------------------------
    {
        ListBuilder fwd(S.fwd), bwd(S.bwd);
        for (i32 l = 0; l < nlevels; l++) {
            for (i32 j = lvlptr[l]; j < lvlptr[l + 1]; j++) {
                const i32 rj = S.Rp[j + 1] - S.Rp[j], cj = S.Lp[j + 1] - S.Lp[j];
                // forward substitution: row j of L
                if (rj > B_MIN) fwd.add_B(j, S.Rp[j], S.Rp[j + 1]);
                else if (rj > T_MAX) fwd.add_W(j);
                else if (rj > 0) fwd.add_T(j);
                // backward substitution: column j of L
                if (cj > B_MIN) bwd.add_B(j, S.Lp[j], S.Lp[j + 1]);
                else if (cj > T_MAX) bwd.add_W(j);
                else bwd.add_T(j);
            }
            fwd.close_level();
            bwd.close_level();
        }
        // numeric factorisation: UNITS (supernodes, single columns) by unit level.  Without
        // supernodes the unit levels are the top levels above.
        const i32 NFi = S.NF;
        const i32 nsn = (i32)S.sn_ptr.size() - 1;
        std::vector<i32> ulev((size_t)n, 0), childmax((size_t)n, -1);
        std::vector<i32> sn_level((size_t)std::max(nsn, 0), 0);
        i32 nfl = 0;
        for (i32 j = NFi; j < n; j++) {
            const i32 sn = S.sn_of[j];
            i32 lv;
            if (sn < 0) {
                lv = childmax[j] + 1;
                ulev[j] = lv;
            } else {
                if (j != S.sn_col[S.sn_ptr[sn + 1] - 1]) continue; // decided at the last member
                lv = 0;
                for (i32 t = S.sn_ptr[sn]; t < S.sn_ptr[sn + 1]; t++) lv = std::max(lv, childmax[S.sn_col[t]] + 1);
                for (i32 t = S.sn_ptr[sn]; t < S.sn_ptr[sn + 1]; t++) ulev[S.sn_col[t]] = lv;
                sn_level[sn] = lv;
            }
            nfl = std::max(nfl, lv + 1);
            const i32 pj = S.etree[j];
            if (pj >= 0) childmax[pj] = std::max(childmax[pj], lv);
        }
        S.nfaclevels = nfl;
        // row lists of the top without the contributions of supernode members: those arrive as dense
        // updates (k_snode_update inside a supernode, k_snode_extend from a descendant supernode)
        if (nsn > 0) {
            S.Rf_p.assign((size_t)n + 1, 0);
            for (i32 j = NFi; j < n; j++) {
                i32 c = 0;
                for (i32 q = S.Rp[j]; q < S.Rp[j + 1]; q++) c += S.sn_of[S.Rcol[q]] < 0;
                S.Rf_p[j + 1] = c;
            }
            for (i32 j = 0; j < n; j++) S.Rf_p[j + 1] += S.Rf_p[j];
            S.Rf_col.resize((size_t)S.Rf_p[n] + 1);
            S.Rf_pos.resize((size_t)S.Rf_p[n] + 1);
            for (i32 j = NFi; j < n; j++) {
                i32 o = S.Rf_p[j];
                for (i32 q = S.Rp[j]; q < S.Rp[j + 1]; q++)
                    if (S.sn_of[S.Rcol[q]] < 0) {
                        S.Rf_col[o] = S.Rcol[q];
                        S.Rf_pos[o] = S.Rpos[q];
                        o++;
                    }
            }
            // slots of the ancestor updates: supernode s subtracts L_B D L_B' at (B[r], B[c]), r > c
            S.upd_ptr.assign((size_t)nsn + 1, 0);
            for (i32 sn = 0; sn < nsn; sn++) {
                const i32 e = S.sn_col[S.sn_ptr[sn + 1] - 1];
                const i64 nb = S.Lp[e + 1] - S.Lp[e];
                S.upd_ptr[sn + 1] = S.upd_ptr[sn] + nb * (nb - 1) / 2;
            }
            S.upd_slot.resize((size_t)S.upd_ptr[nsn] + 1);
            for (i32 sn = 0; sn < nsn; sn++) {
                const i32 e = S.sn_col[S.sn_ptr[sn + 1] - 1];
                const i32 *B = S.Li.data() + S.Lp[e];
                const i32 nb = S.Lp[e + 1] - S.Lp[e];
                i64 u = S.upd_ptr[sn];
                for (i32 c = 0; c < nb; c++) {
                    const i32 j = B[c];
                    i32 q = S.Lp[j];
                    const i32 qe = S.Lp[j + 1];
                    for (i32 r = c + 1; r < nb; r++) {
                        while (q < qe && S.Li[q] < B[r]) q++;
                        if (q >= qe || S.Li[q] != B[r]) {
                            set_error("internal: supernode update target missing from the pattern of L");
                            return -9;
                        }
                        S.upd_slot[u++] = q;
                    }
                }
            }
        }
        const std::vector<i32> &FRp = nsn > 0 ? S.Rf_p : S.Rp;
        const i32 *FRcol = nsn > 0 ? S.Rf_col.data() : S.Rcol.data(), *FRpos = nsn > 0 ? S.Rf_pos.data() : S.Rpos.data();
        // buckets by unit level
        std::vector<i32> lcount((size_t)nfl + 1, 0), bucket((size_t)std::max<i64>(n - NFi, 0));
        for (i32 j = NFi; j < n; j++)
            if (S.sn_of[j] < 0) lcount[ulev[j] + 1]++;
        for (i32 l = 0; l < nfl; l++) lcount[l + 1] += lcount[l];
        {
            std::vector<i32> pos(lcount.begin(), lcount.end() - 1);
            for (i32 j = NFi; j < n; j++)
                if (S.sn_of[j] < 0) bucket[pos[ulev[j]]++] = j;
        }
        S.sn_order.resize((size_t)std::max(nsn, 0));
        std::iota(S.sn_order.begin(), S.sn_order.end(), 0);
        std::stable_sort(S.sn_order.begin(), S.sn_order.end(), [&](i32 a, i32 b) { return sn_level[a] < sn_level[b]; });
        S.sn_lvl_ptr.assign((size_t)nfl + 1, 0);
        for (i32 sn = 0; sn < nsn; sn++) S.sn_lvl_ptr[sn_level[sn] + 1]++;
        for (i32 l = 0; l < nfl; l++) S.sn_lvl_ptr[l + 1] += S.sn_lvl_ptr[l];
        S.sn_lvl_nblk.assign((size_t)nfl, 0);
        S.sn_lvl_hmax.assign((size_t)nfl, 0);
        S.sn_lvl_nbmax.assign((size_t)nfl, 0);
        // ancestor updates assembled per target column (host.hpp: asm_*)
        if (nsn > 0) {
            S.asm_uoff.assign((size_t)nsn, 0);
            S.asm_doff.assign((size_t)nsn, 0);
            S.asm_lvl_ptr.assign((size_t)nfl + 1, 0);
            struct Trip {
                i32 node, sn, cB;
            };
            std::vector<Trip> trips;
            for (i32 l = 0; l < nfl; l++) {
                trips.clear();
                i64 uo = 0;
                i32 dofs = 0, contributing = 0;
                for (i32 u = S.sn_lvl_ptr[l]; u < S.sn_lvl_ptr[l + 1]; u++) {
                    const i32 sn = S.sn_order[u];
                    const i32 e = S.sn_col[S.sn_ptr[sn + 1] - 1];
                    const i32 nb = S.Lp[e + 1] - S.Lp[e];
                    S.asm_uoff[sn] = uo;
                    S.asm_doff[sn] = dofs;
                    uo += (i64)nb * (nb - 1) / 2;
                    dofs += nb;
                    contributing += nb > 0;
                }
                if (contributing >= 2) {
                    for (i32 u = S.sn_lvl_ptr[l]; u < S.sn_lvl_ptr[l + 1]; u++) {
                        const i32 sn = S.sn_order[u];
                        const i32 e = S.sn_col[S.sn_ptr[sn + 1] - 1];
                        const i32 *B = S.Li.data() + S.Lp[e];
                        const i32 nb = S.Lp[e + 1] - S.Lp[e];
                        for (i32 c = 0; c < nb; c++) trips.push_back({B[c], sn, c});
                    }
                    std::stable_sort(trips.begin(), trips.end(), [](const Trip &a, const Trip &b) { return a.node < b.node; });
                    for (size_t k = 0; k < trips.size(); k++) {
                        const Trip &t = trips[k];
                        if (k == 0 || trips[k - 1].node != t.node) {
                            S.asm_tgt.push_back(t.node);
                            S.asm_src_ptr.push_back((i32)S.asm_src.size());
                        }
                        const i32 e = S.sn_col[S.sn_ptr[t.sn + 1] - 1];
                        const i64 nb = S.Lp[e + 1] - S.Lp[e];
                        const i64 tri = (i64)t.cB * nb - (i64)t.cB * (t.cB + 1) / 2;
                        S.asm_src.push_back({S.asm_uoff[t.sn] + tri, S.upd_ptr[t.sn] + tri, (i32)(nb - t.cB - 1), S.asm_doff[t.sn] + t.cB});
                    }
                    S.asm_usize = std::max(S.asm_usize, uo);
                    S.asm_dsize = std::max(S.asm_dsize, dofs);
                }
                S.asm_lvl_ptr[l + 1] = (i32)S.asm_tgt.size();
            }
            S.asm_src_ptr.push_back((i32)S.asm_src.size());
        }
        // Forward substitution, supernode members: a member's row restricted to NON-member columns (bundle
        // columns and ordinary top columns; member columns of any supernode reach it through the dense pushes)
        // can be gathered as soon as the ordinary top columns it refers to are final -- usually long before its own
        // unit level.  Each member row therefore joins the gather launch of level 1 + (highest unit level of an
        // ordinary top column in its list), 0 when it only refers to bundle columns: the gathers and the pushes
        // are both subtractions from the same entry, their order is free.  Config 2: 27 gather launches per
        // sweep become a handful.  
        std::vector<std::vector<i32>> hoisted((size_t)nfl);
        if (nsn > 0) {
            const bool no_hoist = false;
            for (i32 sn = 0; sn < nsn; sn++)
                for (i32 t = S.sn_ptr[sn]; t < S.sn_ptr[sn + 1]; t++) {
                    const i32 j = S.sn_col[t];
                    if (FRp[j + 1] == FRp[j]) continue;
                    i32 gl = 0;
                    if (no_hoist) gl = sn_level[sn];
                    else
                        for (i32 q = FRp[j]; q < FRp[j + 1]; q++)
                            if (FRcol[q] >= NFi) gl = std::max(gl, ulev[FRcol[q]] + 1);
                    hoisted[(size_t)std::min(gl, sn_level[sn])].push_back(j);
                }
        }
        ListBuilder fac(S.fac), snx(S.snx), fwu(S.fwu), bwu(S.bwu);
        for (i32 l = 0; l < nfl; l++) {
            for (i32 u = lcount[l]; u < lcount[l + 1]; u++) {
                const i32 j = bucket[u];
                const i32 rj = FRp[j + 1] - FRp[j], cj = S.Lp[j + 1] - S.Lp[j];
                // factor: column j gathers rj contributions into cj targets; heavy columns (many
                // contributions, or long tails: dense fronts) are spread over workgroups
                i64 work = 0;
                if (rj > B_MIN || (rj > FAC_T_ROW && (i64)rj * cj > F_MIN_WORK))
                    for (i32 t = FRp[j]; t < FRp[j + 1]; t++) work += S.Lp[FRcol[t] + 1] - (FRpos[t] + 1) + 1;
                if (rj > B_MIN || work > F_MIN_WORK)
                    fac.add_B_work(j, FRp[j], FRp[j + 1], FRcol, FRpos, S.Lp, work);
                else if (rj <= FAC_T_ROW && cj <= FAC_T_COL) fac.add_T(j);
                else fac.add_W(j);
                if (nsn > 0) { // substitutions by unit level (used instead of fwd / bwd when there are supernodes)
                    if (rj > B_MIN) fwu.add_B(j, FRp[j], FRp[j + 1]);
                    else if (rj > T_MAX) fwu.add_W(j);
                    else if (rj > 0) fwu.add_T(j);
                    if (cj > B_MIN) bwu.add_B(j, S.Lp[j], S.Lp[j + 1]);
                    else if (cj > T_MAX) bwu.add_W(j);
                    else bwu.add_T(j);
                }
            }
            for (i32 o = S.sn_lvl_ptr[l]; o < S.sn_lvl_ptr[l + 1]; o++) {
                const i32 sn = S.sn_order[o];
                const i32 w = S.sn_ptr[sn + 1] - S.sn_ptr[sn], e = S.sn_col[S.sn_ptr[sn + 1] - 1];
                const i32 nbel = S.Lp[e + 1] - S.Lp[e];
                S.sn_lvl_nblk[l] = std::max(S.sn_lvl_nblk[l], (w + 63) / 64);
                S.sn_lvl_hmax[l] = std::max(S.sn_lvl_hmax[l], w + nbel);
                S.sn_lvl_nbmax[l] = std::max(S.sn_lvl_nbmax[l], nbel);
                for (i32 t = S.sn_ptr[sn]; t < S.sn_ptr[sn + 1]; t++) {
                    const i32 j = S.sn_col[t];
                    i32 eb = FRp[j];
                    const i32 ee = FRp[j + 1];
                    // the contributions of BUNDLE columns (the head of the list: bundle columns are numbered first) depend
                    // on nothing in the top: they are taken by ONE launch ahead of all levels (snb, below); the level's
                    // own launch keeps the contributions of ordinary top columns (CHIP_NO_SNX_HOIST: everything)
                    if (!switches().no_snx_hoist)
                        while (eb < ee && FRcol[eb] < NFi) eb++;
                    if (ee == eb) continue;
                    i64 work = 0;
                    for (i32 q = eb; q < ee; q++) work += S.Lp[FRcol[q] + 1] - (FRpos[q] + 1) + 1;
                    snx.add_B_work(j, eb, ee, FRcol, FRpos, S.Lp, work);
                }
            }
            for (const i32 j : hoisted[(size_t)l]) { // member rows whose gather belongs to this level's launch
                const i32 eb = FRp[j], ee = FRp[j + 1], rj = ee - eb;
                if (rj > B_MIN) fwu.add_B(j, eb, ee);
                else if (rj > T_MAX) fwu.add_W(j);
                else fwu.add_T(j);
            }
            fac.close_level();
            snx.close_level();
            fwu.close_level();
            bwu.close_level();
        }
        {
            // contributions of bundle columns into supernode members (see above), one list per unit level of the TARGET's
            // supernode: the engine takes the first levels' lists ahead of the supernode chain and the rest beside it
            ListBuilder snb(S.snb);
            // (chunks of >= 16384 updates: a member column is rarely split -- with the general bounds, 1024 - 4096, config 2's
            // 184 M updates were 170 000 workgroups of ~4 updates per thread, each paying the staging of its column, a scan and
            // one global atomic per row of the column: 0.4 ms per step)
            const i64 snb_chunk = switches().snb_chunk > 0 ? switches().snb_chunk : 16384;
            for (i32 l = 0; l < nfl; l++) {
                i64 lwork = 0, lcontrib = 0, lcols = 0, lrows = 0;
                if (nsn > 0 && !switches().no_snx_hoist)
                    for (i32 o = S.sn_lvl_ptr[l]; o < S.sn_lvl_ptr[l + 1]; o++) {
                        const i32 sn = S.sn_order[o];
                        for (i32 t = S.sn_ptr[sn]; t < S.sn_ptr[sn + 1]; t++) {
                            const i32 j = S.sn_col[t];
                            const i32 eb = FRp[j];
                            i32 mid = eb;
                            while (mid < FRp[j + 1] && FRcol[mid] < NFi) mid++;
                            if (mid == eb) continue;
                            i64 work = 0;
                            for (i32 q = eb; q < mid; q++) work += S.Lp[FRcol[q] + 1] - (FRpos[q] + 1) + 1;
                            snb.add_B_work(j, eb, mid, FRcol, FRpos, S.Lp, work, snb_chunk, 4 * snb_chunk);
                            lwork += work;
                            lcontrib += mid - eb;
                            lcols++;
                            lrows += S.Lp[j + 1] - S.Lp[j];
                        }
                    }
                snb.close_level();
                S.snb_work.push_back(lwork);
                if (switches().timing && lwork)
                    fprintf(stderr, "[chip] snb level %d: work %lld, %lld contributions into %lld columns (%lld rows), %d chunks\n", (int)l,
                            (long long)lwork, (long long)lcontrib, (long long)lcols, (long long)lrows,
                            (int)(S.snb.b_ptr.empty() ? 0 : S.snb.b_ptr.back()));
            }
        }
        ListBuilder smv(S.smv);
        for (i32 j = S.NF; j < n; j++) {
            const i32 len = S.Sp[j + 1] - S.Sp[j];
            if (len > B_MIN) smv.add_B(j, S.Sp[j], S.Sp[j + 1]);
            else if (len > T_MAX) smv.add_W(j);
            else smv.add_T(j);
        }
        smv.close_level();
    }
    clk("work lists, update slots");
    // ---- residual over the top rows: a supernode-contiguous view of x ----------------------
    // The final numbering is level major, so the members of a chain supernode -- one per level -- lie a whole
    // level apart: a top row's entries to a dense front gather x with a stride of hundreds of bytes (config 5:
    // every lane its own 64-byte sector, the residual over the top rows ran at 1.8 TB/s, bound by L2 sectors,
    // not HBM).  The residual therefore reads a copy of x in which every supernode's members are consecutive
    // (one N-element gather per residual) and Scol is renumbered to match.
    if (S.sn_ptr.size() > 1 && S.nfold == 0) {
        const i32 NFi = S.NF;
        S.xperm.resize((size_t)n);
        std::vector<i32> xinv((size_t)n);
        i32 pos = 0;
        for (; pos < NFi; pos++) S.xperm[pos] = pos;
        for (size_t sn = 0; sn + 1 < S.sn_ptr.size(); sn++)
            for (i32 t = S.sn_ptr[sn]; t < S.sn_ptr[sn + 1]; t++) S.xperm[pos++] = S.sn_col[t];
        for (i32 j = NFi; j < n; j++)
            if (S.sn_of[j] < 0) S.xperm[pos++] = j;
        for (i32 i = 0; i < n; i++) xinv[S.xperm[i]] = i;
        run_threads(par_threads(S.nnzS), [&](int t, int TT) {
            for (i64 u = (i64)S.nnzS * t / TT; u < (i64)S.nnzS * (t + 1) / TT; u++) S.Scol[u] = xinv[S.Scol[u]];
        });
        clk("supernode-contiguous x view");
    }
    if (switches().timing && S.bundle_ptr.size() > 1) { // the bundle geometry the kernels see
        const i32 nb = (i32)S.bundle_ptr.size() - 1;
        i64 mx_n = 0, mx_e = 0, mx_u = 0, mx_l = 0, sum_n = 0, sum_e = 0, sum_u = 0, mx_top = 0, mx_utop = 0;
        for (i32 b = 0; b < nb; b++) {
            const i32 s0 = S.bundle_ptr[b], s1 = S.bundle_ptr[b + 1];
            const i64 ne = S.Lp[s1] - S.Lp[s0], nu = S.Up.empty() ? 0 : S.Up[s1] - S.Up[s0];
            i64 ntop = 0, nutop = 0;
            if (!S.Li16.empty())
                for (i64 q = S.Lp[s0]; q < S.Lp[s1]; q++) ntop += S.Li16[(size_t)q] >= s1 - s0;
            if (!S.Ucol16.empty())
                for (i64 q = S.Up[s0]; q < S.Up[s1]; q++) nutop += S.Ucol16[(size_t)q] >= s1 - s0;
            mx_n = std::max<i64>(mx_n, s1 - s0), mx_e = std::max(mx_e, ne), mx_u = std::max(mx_u, nu);
            mx_l = std::max<i64>(mx_l, S.blvl_ptr[b + 1] - S.blvl_ptr[b] - 1);
            mx_top = std::max(mx_top, ntop), mx_utop = std::max(mx_utop, nutop);
            sum_n += s1 - s0, sum_e += ne, sum_u += nu;
        }
        i64 mx_g = 0, mx_k = 0;
        for (i32 g = 0; g < S.gf_ng; g++) {
            mx_g = std::max<i64>(mx_g, S.gf_bptr[g + 1] - S.gf_bptr[g]);
            mx_k = std::max<i64>(mx_k, S.gf_ptr[g + 1] - S.gf_ptr[g]);
        }
        std::fprintf(stderr,
                     "[chip analyse] bundles %d: nodes max %lld mean %.0f, L entries max %lld mean %.0f (to top rows max %lld), "
                     "U entries max %lld mean %.0f (to top columns max %lld), levels max %lld; NF %d of %lld; fold k %d; groups %d "
                     "(bundles per group max %lld, top nodes max %lld)\n",
                     nb, (long long)mx_n, (double)sum_n / nb, (long long)mx_e, (double)sum_e / nb, (long long)mx_top,
                     (long long)mx_u, (double)sum_u / nb, (long long)mx_utop, (long long)mx_l, S.NF, (long long)n, S.nfold,
                     S.gf_ng, (long long)mx_g, (long long)mx_k);
    }
    return 0;
}

} // namespace chip
