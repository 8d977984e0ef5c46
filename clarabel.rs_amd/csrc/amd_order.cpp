// amd_order.cpp -- approximate minimum degree ordering, written from the
// published algorithm (Amestoy, Davis, Duff: "An approximate minimum degree
// ordering algorithm", SIMAX 1996): quotient graph with element absorption,
// approximate external degrees, aggressive absorption, mass elimination and
// hash-based supervariable detection, plus up-front removal of dense rows.
//
// It stands in for the un-vendored crate `amd = "0.2.2"` that the reference
// calls at qdldl.rs:905-917 / ldlsolvers/mod.rs:15-23 / auto.rs:69 with
// control.dense = 10 * 1.5.  The ordering itself is "parity unpinned" (only a
// 4x4 KAT exists, qdldl/test.rs:123-129); every engine accepts an injected
// permutation, and solutions do not depend on the ordering beyond rounding.
//
// Storage differs from SuiteSparse AMD on purpose: each variable's list is
// compacted in place inside its original adjacency slot (it can never grow),
// while new element lists are appended to a separate growing pool, so no
// garbage collection pass is needed (pool size is bounded by the symbolic
// front sizes, i.e. O(nnz(L)) integers, allocated once per problem).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "host.hpp"

namespace chip {

namespace {

// I = index type of the quotient graph: int32 whenever the symmetric adjacency (2 x off-diagonal
// entries) fits -- the ordering is memory bound, half-width indices are ~1.5x faster -- else int64
// wgt (may be nullptr): initial supervariable sizes -- node i stands for wgt[i] >= 1 indistinguishable variables of the
// original graph (amd_order_grouped: the rows of a dense cone block enter as one node); degrees, the dense-row test and
// the fill statistics count weights
template <typename I>
int amd_order_impl(i64 n, const i64 *Ap, const i64 *Ai, double dense_scale, std::vector<i64> &perm,
                   AmdInfo *info, i64 dense_n, const i64 *wgt = nullptr) {
    constexpr I NONE = -1;
    perm.assign((size_t)n, 0);
    AmdInfo st;
    if (n == 0) {
        if (info) *info = st;
        return 0;
    }
    const bool timing = switches().timing;
    const auto t_begin = std::chrono::steady_clock::now();
    // ---- symmetric adjacency without the diagonal ---------------------------
    // (threads own ranges of nodes and scan all of K in order: the lists come out the same for any count)
    const i64 nnzA = Ap[n];
    const i64 par_min = switches().host_par_min;
    const int T = nnzA >= par_min ? host_threads() : 1;
    std::vector<I> ast((size_t)n + 1, 0);
    std::vector<int> bad((size_t)T, 0);
    RawBuf<I> adj((size_t)2 * (size_t)nnzA + 1); // (no zero fill; sized for every entry off the diagonal)
    if (sizeof(I) == 4) {
        // one stable bucket pass (host.hpp): entry (r, c) of K goes to the lists of r and of c, source order kept
        {
            const std::vector<int64_t> ccuts = balanced_cuts(Ap, n, T);
            run_threads(T, [&](int t, int) {
                for (i64 c = ccuts[t]; c < ccuts[t + 1] && !bad[t]; c++)
                    for (i64 p = Ap[c]; p < Ap[c + 1]; p++)
                        if (Ai[p] < 0 || Ai[p] >= n) {
                            bad[t] = 1;
                            break;
                        }
            });
            for (int t = 0; t < T; t++)
                if (bad[t]) return -9;
            std::vector<i32> ptr32;
            stable_buckets(T, (i32)n, ptr32, true,
                           [&](int t, int, auto f) {
                               for (i64 c = ccuts[t]; c < ccuts[t + 1]; c++)
                                   for (i64 p = Ap[c]; p < Ap[c + 1]; p++) {
                                       const i64 r = Ai[p];
                                       if (r == c) continue;
                                       f((i32)r, c);
                                       f((i32)c, r);
                                   }
                           },
                           [&](i32, i64 other, i32 u) { adj[(size_t)u] = (I)other; });
            for (i64 i = 0; i <= n; i++) ast[(size_t)i] = (I)ptr32[(size_t)i];
        }
    } else {
    run_threads(T, [&](int t, int TT) {
        const I k0 = (I)(n * t / TT), k1 = (I)(n * (t + 1) / TT);
        for (I c = 0; c < n; c++)
            for (I p = Ap[c]; p < Ap[c + 1]; p++) {
                const I r = Ai[p];
                if (r < 0 || r >= n) {
                    bad[t] = 1;
                    return;
                }
                if (r != c) {
                    if (r >= k0 && r < k1) ast[r + 1]++;
                    if (c >= k0 && c < k1) ast[c + 1]++;
                }
            }
    });
    for (int t = 0; t < T; t++)
        if (bad[t]) return -9;
    for (I i = 0; i < n; i++) ast[i + 1] += ast[i];
    {
        std::vector<I> fillp(ast.begin(), ast.end() - 1);
        const std::vector<int64_t> cuts = balanced_cuts(ast.data(), n, T);
        run_threads(T, [&](int t, int) {
            const I k0 = (I)cuts[t], k1 = (I)cuts[t + 1];
            if (k0 >= k1) return;
            for (I c = 0; c < n; c++)
                for (I p = Ap[c]; p < Ap[c + 1]; p++) {
                    const I r = Ai[p];
                    if (r == c) continue;
                    if (r >= k0 && r < k1) adj[fillp[r]++] = c;
                    if (c >= k0 && c < k1) adj[fillp[c]++] = r;
                }
        });
    }
    }
    std::vector<I> alen((size_t)n), aelen((size_t)n, 0);
    for (I i = 0; i < n; i++) alen[i] = ast[i + 1] - ast[i];

    // ---- node state ---------------------------------------------------------
    std::vector<I> nv((size_t)n, 1);      // supervariable size; 0 = not a live variable
    i64 wtot = n;                         // total weight = largest possible degree
    if (wgt) {
        wtot = 0;
        for (I i = 0; i < n; i++) {
            nv[i] = (I)wgt[i];
            wtot += wgt[i];
        }
    }
    std::vector<I> degree((size_t)n, 0);  // variables: approx external degree; elements: |Le|
    std::vector<i64> w((size_t)n, 1);     // 0 = dead element; otherwise pass stamps
    std::vector<I> head((size_t)wtot + 1, NONE), nxt((size_t)n, NONE), prv((size_t)n, NONE);
    std::vector<I> vparent((size_t)n, NONE); // absorbed variable -> variable/pivot it joined
    std::vector<I> est((size_t)n, 0), elen((size_t)n, 0);
    std::vector<I> epool;
    epool.reserve((size_t)ast[n] / 2 + 16);
    std::vector<I> hhead((size_t)n, NONE), hnext((size_t)n, NONE), lasthash((size_t)n, 0);
    std::vector<I> pivots;
    pivots.reserve((size_t)n);
    std::vector<char> is_dense((size_t)n, 0);

    // ---- dense rows are pulled out and ordered last -------------------------
    // (dense_n: the dimension the dense-row threshold is computed from -- the whole matrix when one connected
    // component of it is being ordered)
    double dth = 10.0 * dense_scale * std::sqrt((double)(dense_n > 0 ? dense_n : n));
    if (dth < 16.0) dth = 16.0;
    // (the degrees compared against dth are WEIGHTED when dense blocks enter as one node each: the clamp is then the
    // total weight, not the compressed node count -- a node next to a block of >= 16 rows has a weighted degree
    // above n on small problems and would be pulled out as "dense")
    if (!wgt && dth > (double)n) dth = (double)n;
    I ndense = 0;
    i64 wdense = 0;
    if (dth > (double)wtot) dth = (double)wtot;
    for (I i = 0; i < n; i++) {
        double di = (double)alen[i];
        if (wgt) { // (weighted degree)
            di = 0.0;
            for (I p = ast[i]; p < ast[i] + alen[i]; p++) di += (double)wgt[adj[p]];
        }
        if (di > dth) {
            is_dense[i] = 1;
            wdense += nv[i];
            nv[i] = 0;
            ndense++;
        }
    }
    st.ndense = wgt ? wdense : ndense;
    const I nlive = (I)(wtot - wdense);
    for (I i = 0; i < n; i++) {
        if (is_dense[i]) continue;
        I d = 0;
        for (I p = ast[i]; p < ast[i] + alen[i]; p++)
            if (!is_dense[adj[p]]) d += wgt ? (I)wgt[adj[p]] : 1;
        degree[i] = d;
    }
    std::vector<I> tail((size_t)wtot + 1, NONE);
    auto dl_insert = [&](I i, I d) {
        nxt[i] = head[d];
        prv[i] = NONE;
        if (head[d] != NONE) prv[head[d]] = i;
        else tail[d] = i;
        head[d] = i;
    };
    // Ties among the variables of a freshly formed element: last-in-first-out like the classic
    // implementations (default), or first-in-first-out (measured and dropped: a variable that has been
    // waiting at this degree goes before one that just joined it, which spreads equal-degree pivots
    // over independent subtrees; same kind of fill, often a shallower tree, sometimes a deeper one)
    const bool lifo = true; // (last-in-first-out ties in the degree lists; the other order was measured and dropped)
    auto dl_insert_tail = [&](I i, I d) {
        if (lifo) return dl_insert(i, d);
        prv[i] = tail[d];
        nxt[i] = NONE;
        if (tail[d] != NONE) nxt[tail[d]] = i;
        else head[d] = i;
        tail[d] = i;
    };
    auto dl_remove = [&](I i, I d) {
        if (prv[i] != NONE) nxt[prv[i]] = nxt[i];
        else head[d] = nxt[i];
        if (nxt[i] != NONE) prv[nxt[i]] = prv[i];
        else tail[d] = prv[i];
    };
    // insert in reverse so that ties are broken by ascending index
    for (I i = n - 1; i >= 0; i--)
        if (!is_dense[i]) dl_insert(i, degree[i]);

    const auto t_adj = std::chrono::steady_clock::now();
    I nelim = 0, mindeg = 0, lemax = 0;
    i64 wflg = 2;

    double tph[5] = {0, 0, 0, 0, 0};
    long long c_lme = 0, c_p1 = 0, c_p2e = 0, c_p2v = 0; // (work counters, printed with CHIP_TIMING)
    auto tick = [&](int k, std::chrono::steady_clock::time_point &t0) {
        if (!timing) return;
        const auto t1 = std::chrono::steady_clock::now();
        tph[k] += std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
    };
    // Adjacency list of a live variable i: [ast[i], +avlen[i]) its variable neighbours, then aelen[i] elements
    // (alen = both).  Variables first, so that the (long) variable part can stay where it is while the element
    // part changes at every pivot that has i in its element:
    //
    // the approximate-degree update rescans the variable list of every member of the new element Lme to drop the
    // entries that are themselves in Lme (or dead) and to sum the weights of the rest.  For a variable that was
    // in the PREVIOUS pivot's element too and was cleaned then, the only entries that can need dropping now are
    // the members that are new to Lme -- and every new member scans its own list anyway and meets the edge from
    // its side.  So: members that scan flag the Lme members they find (`touched`); a member that was in the
    // previous element, has not been flagged and has no merged neighbour (`dirty`) keeps its list, its weight
    // sum and its hash (degA / hashA) without looking at it.  Config 5: 800 coupling variables with ~500-entry
    // lists sit in the element of every one of 1.8e5 pivots -- 22 of the ordering's 27 s were these rescans.
    std::vector<I> avlen(alen), degA((size_t)n, 0), lastscan((size_t)n, -2), touched((size_t)n, -1);
    std::vector<uint64_t> hashA((size_t)n, 0);
    std::vector<char> dirty((size_t)n, 0), cand;
    I seq = 0;
    const bool rescan_all = switches().amd_rescan; // every member rescans (the textbook update; tests)
    // Grouped ordering (wgt): MULTIPLE elimination with a degree tolerance.  A stage eliminates an independent set of
    // nodes whose degree is within (1 + tol) of the stage's minimum -- a node adjacent to one of the stage's pivots (a
    // member of its element) waits for the next stage.  Plain minimum degree walks a chain of cliques (BASELINE config
    // 5: 200 PSD blocks tied pairwise by overlap variables) from its ends inward, one block after the other: minimal
    // fill, but an elimination tree that IS the chain -- every block swallows the overlap variables towards its
    // predecessor and becomes that block's parent: 200 dependent levels of dense fronts, nothing for the device to run
    // side by side.  With the stages the blocks leave odd-even: the even ones as leaves, the odd ones with the overlap
    // variables of both sides as their parents -- depth 2 instead of 200 for the same fill.
    std::vector<I> blocked(wgt ? (size_t)n : (size_t)0, 0);
    I stage = 1, stage_thr = -1;
    const double stage_tol = 1.0; // degree tolerance of a stage of the grouped ordering
    while (nelim < nlive) {
        auto tp0 = std::chrono::steady_clock::now();
        while (mindeg <= wtot && head[mindeg] == NONE) mindeg++;
        I me = head[mindeg], medeg = mindeg;
        if (wgt) {
            for (;;) {
                if (stage_thr < 0) stage_thr = (I)std::min<double>((double)wtot, (double)mindeg * (1.0 + stage_tol) + 1.0);
                me = NONE;
                for (I d = mindeg; d <= stage_thr && me == NONE; d++)
                    for (I c = head[d]; c != NONE; c = nxt[c])
                        if (blocked[c] != stage) {
                            me = c;
                            medeg = d;
                            break;
                        }
                if (me != NONE) break;
                stage++; // nothing eligible is left: the next stage starts from the current minimum degree
                stage_thr = -1;
            }
        }
        dl_remove(me, medeg);
        I nvpiv = nv[me];
        nelim += nvpiv;
        nv[me] = -nvpiv;
        seq++;

        // ---- form the new element: union of me's variables and the variables
        //      of every element adjacent to me (those elements are absorbed)
        if (sizeof(I) == 4 && epool.size() > (size_t)2000000000) return -77; // retry with 64-bit indices
        const I mstart = (I)epool.size();
        I degme = 0;
        for (I p = ast[me] + avlen[me]; p < ast[me] + alen[me]; p++) {
            const I e = adj[p];
            if (w[e] == 0) continue;
            const I q0 = est[e], q1 = est[e] + elen[e];
            for (I q = q0; q < q1; q++) {
                const I i = epool[q];
                const I nvi = nv[i];
                if (nvi > 0) {
                    degme += nvi;
                    nv[i] = -nvi;
                    epool.push_back(i);
                    dl_remove(i, degree[i]);
                }
            }
            w[e] = 0; // absorbed into me
        }
        for (I p = ast[me]; p < ast[me] + avlen[me]; p++) {
            const I i = adj[p];
            const I nvi = nv[i];
            if (nvi != 0) touched[i] = seq; // me is in i's variable list: i has to clean it
            if (nvi > 0) {
                degme += nvi;
                nv[i] = -nvi;
                epool.push_back(i);
                dl_remove(i, degree[i]);
            }
        }
        est[me] = mstart;
        elen[me] = (I)epool.size() - mstart;
        if (wgt) // (this stage's pivots are independent at distance 2: the members of the new element wait, and so do their
                 // variable neighbours -- two cone blocks tied by a handful of overlap variables are not eliminated in the
                 // same stage, or the second would swallow the variables between them and become the first one's parent)
            for (I q = mstart; q < (I)epool.size(); q++) {
                const I i = epool[q];
                blocked[i] = stage;
                for (I p = ast[i]; p < ast[i] + avlen[i]; p++) blocked[adj[p]] = stage;
                // ... and the nodes it meets in its elements: a later pivot of this stage next to i would swallow i
                // (mass elimination) once i's other neighbours are gone, and become this pivot's parent
                for (I p = ast[i] + avlen[i]; p < ast[i] + alen[i]; p++) {
                    const I e = adj[p];
                    if (w[e] == 0) continue;
                    for (I t = est[e]; t < est[e] + elen[e]; t++) blocked[epool[t]] = stage;
                }
            }
        c_lme += elen[me];
        const I mend = mstart + elen[me];
        w[me] = 1;
        lemax = std::max(lemax, degme);

        tick(0, tp0);
        // ---- pass 1: w[e] - wflg = |Le \ Lme| for every element touching Lme
        for (I q = mstart; q < mend; q++) {
            const I i = epool[q];
            const I nvi = -nv[i];
            c_p1 += alen[i] - avlen[i];
            for (I p = ast[i] + avlen[i]; p < ast[i] + alen[i]; p++) {
                const I e = adj[p];
                const i64 we = w[e];
                if (we >= wflg) w[e] = we - nvi;
                else if (we != 0) w[e] = degree[e] + wflg - nvi;
            }
        }
        tick(1, tp0);
        // ---- pass 2: prune each member's lists, approximate degree, hash
        auto pass2_one = [&](I i, bool scan) {
            const I p1 = ast[i], avl = avlen[i], eb = p1 + avl, ee = p1 + alen[i];
            I deg = 0, pn = eb;
            uint64_t hash = 0;
            c_p2e += ee - eb;
            if (scan) c_p2v += avl;
            for (I p = eb; p < ee; p++) { // elements, compacted in place
                const I e = adj[p];
                const i64 we = w[e];
                if (we == 0) continue;
                const i64 dext = we - wflg;
                if (dext > 0) {
                    deg += (I)dext;
                    adj[pn++] = e;
                    hash += (uint64_t)e;
                } else {
                    w[e] = 0; // Le is a subset of Lme: aggressive absorption
                }
            }
            const I ke = pn - eb;
            if (scan) {
                I pv = p1, dA = 0;
                uint64_t hA = 0;
                for (I p = p1; p < eb; p++) {
                    const I j = adj[p];
                    const I nvj = nv[j];
                    if (nvj > 0) {
                        dA += nvj;
                        adj[pv++] = j;
                        hA += (uint64_t)j;
                    } else if (nvj < 0) {
                        touched[j] = seq; // j is in Lme as well: it has to drop i from its list
                    }
                }
                const I avn = pv - p1;
                if (avn != avl)
                    for (I t = 0; t < ke; t++) adj[p1 + avn + t] = adj[eb + t]; // elements follow the variables
                avlen[i] = avn;
                degA[i] = dA;
                hashA[i] = hA;
                dirty[i] = 0;
            }
            deg += degA[i];
            hash += hashA[i];
            lastscan[i] = seq;
            if (ke == 0 && avlen[i] == 0) {
                lasthash[i] = NONE; // nothing outside the new element: eliminated together with me (below)
            } else {
                degree[i] = std::min(degree[i], deg);
                adj[p1 + avlen[i] + ke] = me; // (fits: i lost me from its variables or an absorbed element)
                aelen[i] = ke + 1;
                alen[i] = avlen[i] + ke + 1;
                lasthash[i] = (I)(hash % (uint64_t)n);
            }
        };
        {
            const I nm = mend - mstart;
            cand.assign((size_t)nm, 0);
            for (I q = mstart; q < mend; q++) {
                const I i = epool[q];
                cand[(size_t)(q - mstart)] = !rescan_all && lastscan[i] == seq - 1 && !dirty[i] && touched[i] != seq;
            }
            for (I q = mstart; q < mend; q++) // members that are new to the element (or flagged) scan and flag
                if (!cand[(size_t)(q - mstart)]) pass2_one(epool[q], true);
            for (I q = mstart; q < mend; q++) // the others scan only if a scanning member found them
                if (cand[(size_t)(q - mstart)]) pass2_one(epool[q], touched[epool[q]] == seq);
        }
        for (I q = mstart; q < mend; q++) { // bookkeeping in member order
            const I i = epool[q];
            const I nvi = -nv[i];
            if (lasthash[i] == NONE) {
                vparent[i] = me;
                nvpiv += nvi;
                nelim += nvi;
                degme -= nvi;
                nv[i] = 0;
                aelen[i] = -1;
            } else {
                const I h = lasthash[i];
                hnext[i] = hhead[h];
                hhead[h] = i;
            }
        }
        degree[me] = degme;
        wflg += lemax;

        tick(2, tp0);
        // ---- supervariable detection among the members of Lme ----------------
        for (I q = mstart; q < mend; q++) {
            const I i0 = epool[q];
            if (nv[i0] >= 0) continue;
            const I h = lasthash[i0];
            I a = hhead[h];
            if (a == NONE) continue;
            hhead[h] = NONE;
            for (; a != NONE && hnext[a] != NONE; a = hnext[a]) {
                const I ln = alen[a], avl = avlen[a];
                for (I p = ast[a]; p < ast[a] + ln - 1; p++) w[adj[p]] = wflg; // (the last entry is me)
                I pb = a, b2 = hnext[a];
                while (b2 != NONE) {
                    bool same = (alen[b2] == ln && avlen[b2] == avl);
                    for (I p = ast[b2]; same && p < ast[b2] + ln - 1; p++)
                        if (w[adj[p]] != wflg) same = false;
                    if (same) {
                        vparent[b2] = a;
                        nv[a] += nv[b2]; // both negative here
                        nv[b2] = 0;
                        if (wgt && blocked[b2] == stage) blocked[a] = stage; // (the merged variable waits if a part of it does)
                        aelen[b2] = -1;
                        // the neighbours' cached weight sums still hold (a took b2's weight), their lists and
                        // hashes do not (b2 is dead): they rescan when they next meet an element
                        for (I p = ast[b2]; p < ast[b2] + avl; p++) dirty[adj[p]] = 1;
                        hnext[pb] = hnext[b2];
                        b2 = hnext[b2];
                    } else {
                        pb = b2;
                        b2 = hnext[b2];
                    }
                }
                wflg++;
            }
        }
        tick(3, tp0);
        // ---- finalise: restore nv, final degrees, compact the element -------
        {
            const I nleft = nlive - nelim;
            I pd = mstart;
            for (I q = mstart; q < mend; q++) {
                const I i = epool[q];
                const I nvi = -nv[i];
                if (nvi <= 0) continue;
                nv[i] = nvi;
                I deg = degree[i] + degme - nvi;
                deg = std::min(deg, nleft - nvi);
                if (deg < 0) deg = 0;
                degree[i] = deg;
                dl_insert_tail(i, deg);
                mindeg = std::min(mindeg, deg);
                epool[pd++] = i;
            }
            elen[me] = pd - mstart;
            epool.resize((size_t)pd);
            if (elen[me] == 0) w[me] = 0;
        }
        tick(4, tp0);
        nv[me] = 0;
        pivots.push_back(me);
        if (timing && wgt && n < 4000 && nvpiv >= 8)
            std::fprintf(stderr, "[chip amd] pivot %d weight %d degree %d stage %d\n", (int)me, (int)nvpiv, (int)degme, (int)stage);
        // fill statistics in the style of amd::Info (used by ldlsolvers/auto.rs:69-77)
        {
            const double f = (double)nvpiv, r = (double)(degme + (wgt ? (I)wdense : ndense));
            const double lnzme = f * r + (f - 1) * f / 2.0;
            st.lnz += lnzme;
            st.ndiv += lnzme;
            const double s = f * r * r + r * (f - 1) * f + (f - 1) * f * (2 * f - 1) / 6.0;
            st.nmultsubs_ldl += (s + lnzme) / 2.0;
        }
    }
    if (ndense > 0) {
        const double f = (double)(wgt ? wdense : (i64)ndense);
        const double lnzme = (f - 1) * f / 2.0;
        st.lnz += lnzme;
        st.ndiv += lnzme;
        const double s = (f - 1) * f * (2 * f - 1) / 6.0;
        st.nmultsubs_ldl += (s + lnzme) / 2.0;
    }

    if (timing) {
        std::fprintf(stderr, "[chip amd] adjacency %.3f s, elimination %.3f s (%lld pivots)\n",
                     std::chrono::duration<double>(t_adj - t_begin).count(),
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t_adj).count(), (long long)pivots.size());
        std::fprintf(stderr, "[chip amd]   element %.3f, pass 1 %.3f, pass 2 %.3f, supervariables %.3f, finalise %.3f s\n",
                     tph[0], tph[1], tph[2], tph[3], tph[4]);
        std::fprintf(stderr, "[chip amd]   sum |Lme| %lld, element-list entries visited: pass 1 %lld, pass 2 %lld, variable-list entries rescanned %lld\n",
                     c_lme, c_p1, c_p2e, c_p2v);
    }
    // ---- expand supervariables: every absorbed variable follows its pivot ---
    std::vector<char> is_pivot((size_t)n, 0);
    for (I v : pivots) is_pivot[v] = 1;
    std::vector<I> root((size_t)n, NONE);
    std::vector<I> stack;
    for (I i = 0; i < n; i++) {
        if (is_dense[i] || is_pivot[i] || root[i] != NONE) continue;
        stack.clear();
        I c = i;
        while (!is_pivot[c] && root[c] == NONE) {
            stack.push_back(c);
            c = vparent[c];
            if (c == NONE) return -9; // cannot happen
        }
        const I r = is_pivot[c] ? c : root[c];
        for (I s : stack) root[s] = r;
    }
    std::vector<I> mcount((size_t)n + 1, 0);
    for (I i = 0; i < n; i++)
        if (root[i] != NONE) mcount[root[i] + 1]++;
    for (I i = 0; i < n; i++) mcount[i + 1] += mcount[i];
    std::vector<I> members((size_t)mcount[n] + 1), mfill(mcount.begin(), mcount.end() - 1);
    for (I i = 0; i < n; i++)
        if (root[i] != NONE) members[mfill[root[i]]++] = i;
    I k = 0;
    for (I v : pivots) {
        perm[k++] = v;
        for (I q = mcount[v]; q < mcount[v + 1]; q++) perm[k++] = members[q];
    }
    // dense rows last, lightest first
    {
        std::vector<I> dn;
        for (I i = 0; i < n; i++)
            if (is_dense[i]) dn.push_back(i);
        std::stable_sort(dn.begin(), dn.end(), [&](I a, I b) { return alen[a] < alen[b]; });
        for (I v : dn) perm[k++] = v;
    }
    if (k != n) return -9;
    if (info) *info = st;
    return 0;
}

} // namespace

static int amd_order_dn(i64 n, const i64 *Ap, const i64 *Ai, double dense_scale, std::vector<i64> &perm,
                        AmdInfo *info, i64 dense_n, const i64 *wgt = nullptr) {
    const i64 nnz = n > 0 ? Ap[n] : 0;
    if (n < ((i64)1 << 30) && 2 * nnz < ((i64)1 << 31) - 16 && dense_n < ((i64)1 << 30)) {
        const int rc = amd_order_impl<i32>(n, Ap, Ai, dense_scale, perm, info, dense_n, wgt);
        if (rc != -77) return rc;
    }
    return amd_order_impl<i64>(n, Ap, Ai, dense_scale, perm, info, dense_n, wgt);
}
// Ordering with dense cone blocks entering as ONE weighted node each (group[i] >= 0: node i belongs to that group; -1:
// on its own).  The rows of a PSD cone's Hs block are mutually adjacent: once the first of them is eliminated the others
// inherit its outside neighbours, so minimum degree orders them consecutively anyway -- but it gets there by carrying a
// 1275-row clique through its quotient graph (BASELINE config 5: 1.6e8 entries of K, 8 s of elimination; the compressed
// graph has 2e4 nodes).  The group's node has the union of its members' outside neighbours and their count as weight;
// members leave consecutively, ascending.
int amd_order_grouped(i64 n, const i64 *Ap, const i64 *Ai, double dense_scale, const i32 *group, std::vector<i64> &perm,
                      AmdInfo *info) {
    std::vector<i64> cid((size_t)n, -1), gfirst;
    i64 ngroups = 0;
    for (i64 i = 0; i < n; i++)
        if (group[i] >= 0) ngroups = std::max<i64>(ngroups, group[i] + 1);
    gfirst.assign((size_t)ngroups, -1);
    i64 nc = 0;
    std::vector<i64> wgt;
    for (i64 i = 0; i < n; i++) {
        const i32 g = group[i];
        if (g < 0) {
            cid[i] = nc++;
            wgt.push_back(1);
        } else if (gfirst[g] < 0) {
            gfirst[g] = nc;
            cid[i] = nc++;
            wgt.push_back(1);
        } else {
            cid[i] = gfirst[g];
            wgt[(size_t)gfirst[g]]++;
        }
    }
    if (nc == n) return amd_order_components(n, Ap, Ai, dense_scale, perm, info);
    // edges of the compressed graph (upper triangle), deduplicated
    std::vector<uint64_t> keys;
    for (i64 c = 0; c < n; c++)
        for (i64 p = Ap[c]; p < Ap[c + 1]; p++) {
            const i64 r = Ai[p];
            if (r < 0 || r >= n) return -9;
            const i64 a = cid[r], b = cid[c];
            if (a == b) continue;
            keys.push_back((uint64_t)std::max(a, b) * (uint64_t)nc + (uint64_t)std::min(a, b)); // (column-major)
        }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    std::vector<i64> cp((size_t)nc + 1, 0), ci(keys.size());
    for (size_t k = 0; k < keys.size(); k++) {
        cp[(size_t)(keys[k] / (uint64_t)nc) + 1]++;
        ci[k] = (i64)(keys[k] % (uint64_t)nc);
    }
    for (i64 c = 0; c < nc; c++) cp[c + 1] += cp[c];
    std::vector<i64> cperm;
    const int rc = amd_order_dn(nc, cp.data(), ci.data(), dense_scale, cperm, info, n, wgt.data());
    if (rc) return rc;
    // expand: a group's members in ascending order at their node's place
    std::vector<i64> mptr((size_t)nc + 1, 0), mem((size_t)n);
    for (i64 i = 0; i < n; i++) mptr[cid[i] + 1]++;
    for (i64 c = 0; c < nc; c++) mptr[c + 1] += mptr[c];
    {
        std::vector<i64> fill(mptr.begin(), mptr.end() - 1);
        for (i64 i = 0; i < n; i++) mem[(size_t)fill[cid[i]]++] = i;
    }
    perm.assign((size_t)n, 0);
    i64 out = 0;
    for (i64 k = 0; k < nc; k++)
        for (i64 q = mptr[cperm[k]]; q < mptr[cperm[k] + 1]; q++) perm[out++] = mem[(size_t)q];
    if (switches().timing) {
        std::fprintf(stderr, "[chip amd] %lld groups compressed: %lld -> %lld nodes, %zu edges\n", (long long)ngroups, (long long)n,
                     (long long)nc, keys.size());
        if (nc < 4000) { // (debug: the order in which heavy nodes leave, light ones as run lengths)
            std::fprintf(stderr, "[chip amd] order:");
            i64 run = 0;
            for (i64 k = 0; k < nc; k++) {
                if (wgt[(size_t)cperm[k]] > 1) {
                    if (run) std::fprintf(stderr, " (%lld)", (long long)run);
                    run = 0;
                    std::fprintf(stderr, " G%lld", (long long)mem[(size_t)mptr[cperm[k]]]);
                } else run++;
            }
            std::fprintf(stderr, " (%lld)\n", (long long)run);
        }
    }
    return out == n ? 0 : -9;
}
int amd_order(i64 n, const i64 *Ap, const i64 *Ai, double dense_scale, std::vector<i64> &perm,
              AmdInfo *info) {
    return amd_order_dn(n, Ap, Ai, dense_scale, perm, info, 0);
}

// Block-diagonal matrices (BASELINE config 4: 1024 independent SOCPs; any problem made of independent blocks):
// the graph falls into connected components, which minimum degree orders independently of each other anyway.
// Components with IDENTICAL patterns (same size, same relative indices: compared exactly, found by hash) take
// the ordering of one representative; the distinct patterns are ordered in parallel on the host threads.  The
// dense-row threshold stays that of the whole matrix.  One component: the plain call.
int amd_order_components(i64 n, const i64 *Ap, const i64 *Ai, double dense_scale, std::vector<i64> &perm,
                         AmdInfo *info) {
    if (n <= 1 || switches().no_components) return amd_order(n, Ap, Ai, dense_scale, perm, info);
    // ---- connected components (union-find with path halving; the smaller index stays the root) ----
    std::vector<i64> root((size_t)n);
    for (i64 i = 0; i < n; i++) root[i] = i;
    auto find = [&](i64 x) {
        while (root[x] != x) {
            root[x] = root[root[x]];
            x = root[x];
        }
        return x;
    };
    i64 ncomp = n; // (one component left: the whole graph is ordered at once, the rest of the scan is not needed)
    for (i64 c = 0; c < n && ncomp > 1; c++)
        for (i64 p = Ap[c]; p < Ap[c + 1]; p++) {
            const i64 r = Ai[p];
            if (r < 0 || r >= n) return -9;
            if (r == c) continue;
            const i64 a = find(r), b = find(c);
            if (a != b) {
                root[a > b ? a : b] = a > b ? b : a;
                ncomp--;
            }
        }
    if (ncomp == 1) return amd_order(n, Ap, Ai, dense_scale, perm, info);
    std::vector<i64> comp((size_t)n, -1), csize;
    for (i64 i = 0; i < n; i++) {
        const i64 r = find(i);
        if (comp[r] < 0) { // (r <= i: the root is the component's smallest node)
            comp[r] = (i64)csize.size();
            csize.push_back(0);
        }
        comp[i] = comp[r];
        csize[comp[i]]++;
    }
    const i64 nc = (i64)csize.size();
    if (nc == 1) return amd_order(n, Ap, Ai, dense_scale, perm, info);
    // nodes by component (ascending inside: the local triangle stays upper), local ids
    std::vector<i64> cptr((size_t)nc + 1, 0), nodes((size_t)n), lid((size_t)n);
    for (i64 q = 0; q < nc; q++) cptr[q + 1] = cptr[q] + csize[q];
    {
        std::vector<i64> fill(cptr.begin(), cptr.end() - 1);
        for (i64 i = 0; i < n; i++) {
            lid[i] = fill[comp[i]] - cptr[comp[i]];
            nodes[fill[comp[i]]++] = i;
        }
    }
    // local patterns, all in one pair of arrays: component q owns columns cptr[q] .. cptr[q+1] of (Lp, Li)
    std::vector<i64> Lp((size_t)n + 1, 0), Li((size_t)Ap[n]);
    for (i64 t = 0; t < n; t++) Lp[t + 1] = Lp[t] + (Ap[nodes[t] + 1] - Ap[nodes[t]]);
    for (i64 t = 0; t < n; t++) {
        i64 o = Lp[t];
        for (i64 p = Ap[nodes[t]]; p < Ap[nodes[t] + 1]; p++) Li[o++] = lid[Ai[p]];
    }
    // hash of every component's local pattern; identical patterns share a representative
    std::vector<uint64_t> hsh((size_t)nc);
    for (i64 q = 0; q < nc; q++) {
        uint64_t h = 1469598103934665603ull ^ (uint64_t)csize[q];
        for (i64 t = cptr[q]; t < cptr[q + 1]; t++) {
            h = (h ^ (uint64_t)(Lp[t + 1] - Lp[t])) * 1099511628211ull;
            for (i64 p = Lp[t]; p < Lp[t + 1]; p++) h = (h ^ (uint64_t)Li[p]) * 1099511628211ull;
        }
        hsh[q] = h;
    }
    std::vector<i64> order((size_t)nc), rep((size_t)nc, -1), reps;
    for (i64 q = 0; q < nc; q++) order[q] = q;
    std::stable_sort(order.begin(), order.end(), [&](i64 a, i64 b) { return hsh[a] < hsh[b]; });
    auto same = [&](i64 a, i64 b) {
        if (csize[a] != csize[b]) return false;
        const i64 na = csize[a];
        if (Lp[cptr[a] + na] - Lp[cptr[a]] != Lp[cptr[b] + na] - Lp[cptr[b]]) return false;
        for (i64 t = 0; t < na; t++)
            if (Lp[cptr[a] + t + 1] - Lp[cptr[a] + t] != Lp[cptr[b] + t + 1] - Lp[cptr[b] + t]) return false;
        const i64 pa = Lp[cptr[a]], pb = Lp[cptr[b]], len = Lp[cptr[a] + na] - pa;
        for (i64 k = 0; k < len; k++)
            if (Li[pa + k] != Li[pb + k]) return false;
        return true;
    };
    for (i64 k = 0; k < nc;) {
        i64 e = k;
        while (e < nc && hsh[order[e]] == hsh[order[k]]) e++;
        std::vector<i64> grp; // representatives among the components with this hash (collisions: several)
        for (i64 t = k; t < e; t++) {
            const i64 q = order[t];
            for (i64 r : grp)
                if (same(r, q)) {
                    rep[q] = r;
                    break;
                }
            if (rep[q] < 0) {
                rep[q] = q;
                grp.push_back(q);
                reps.push_back(q);
            }
        }
        k = e;
    }
    // order the representatives (in parallel: each call is single threaded at these sizes)
    const i64 nr = (i64)reps.size();
    std::vector<std::vector<i64>> rperm((size_t)nr);
    std::vector<AmdInfo> rinfo((size_t)nr);
    std::vector<int> rrc((size_t)nr, 0);
    std::vector<i64> rep_slot((size_t)nc, -1);
    for (i64 k = 0; k < nr; k++) rep_slot[reps[k]] = k;
    const int T = (int)std::min<i64>(nr, host_threads());
    run_threads(T, [&](int t, int TT) {
        std::vector<i64> lp;
        for (i64 k = t; k < nr; k += TT) {
            const i64 q = reps[k], nq = csize[q], base = Lp[cptr[q]];
            lp.assign((size_t)nq + 1, 0);
            for (i64 j = 0; j <= nq; j++) lp[j] = Lp[cptr[q] + j] - base;
            rrc[k] = amd_order_dn(nq, lp.data(), Li.data() + base, dense_scale, rperm[k], &rinfo[k], n);
        }
    });
    for (i64 k = 0; k < nr; k++)
        if (rrc[k]) return rrc[k];
    perm.assign((size_t)n, 0);
    AmdInfo tot;
    i64 out = 0;
    for (i64 q = 0; q < nc; q++) {
        const i64 k = rep_slot[rep[q]];
        const std::vector<i64> &pr = rperm[k];
        for (i64 j = 0; j < csize[q]; j++) perm[out++] = nodes[cptr[q] + pr[j]];
        tot.lnz += rinfo[k].lnz;
        tot.ndiv += rinfo[k].ndiv;
        tot.nmultsubs_ldl += rinfo[k].nmultsubs_ldl;
        tot.ndense += rinfo[k].ndense;
    }
    if (switches().timing)
        std::fprintf(stderr, "[chip amd] %lld connected components, %lld distinct patterns\n", (long long)nc, (long long)nr);
    if (info) *info = tot;
    return 0;
}

} // namespace chip
