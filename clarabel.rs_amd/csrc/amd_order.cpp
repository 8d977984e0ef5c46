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
template <typename I>
int amd_order_impl(i64 n, const i64 *Ap, const i64 *Ai, double dense_scale, std::vector<i64> &perm,
                   AmdInfo *info) {
    constexpr I NONE = -1;
    perm.assign((size_t)n, 0);
    AmdInfo st;
    if (n == 0) {
        if (info) *info = st;
        return 0;
    }
    const bool timing = std::getenv("CHIP_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    // ---- symmetric adjacency without the diagonal ---------------------------
    // (threads own ranges of nodes and scan all of K in order: the lists come out the same for any count)
    const i64 nnzA = Ap[n];
    i64 par_min = 2000000;
    if (const char *e = std::getenv("CHIP_HOST_PAR_MIN")) par_min = std::atoll(e);
    const int T = nnzA >= par_min ? host_threads() : 1;
    std::vector<I> ast((size_t)n + 1, 0);
    std::vector<int> bad((size_t)T, 0);
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
    std::vector<I> adj((size_t)ast[n] + 1);
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
    std::vector<I> alen((size_t)n), aelen((size_t)n, 0);
    for (I i = 0; i < n; i++) alen[i] = ast[i + 1] - ast[i];

    // ---- node state ---------------------------------------------------------
    std::vector<I> nv((size_t)n, 1);      // supervariable size; 0 = not a live variable
    std::vector<I> degree((size_t)n, 0);  // variables: approx external degree; elements: |Le|
    std::vector<i64> w((size_t)n, 1);     // 0 = dead element; otherwise pass stamps
    std::vector<I> head((size_t)n + 1, NONE), nxt((size_t)n, NONE), prv((size_t)n, NONE);
    std::vector<I> vparent((size_t)n, NONE); // absorbed variable -> variable/pivot it joined
    std::vector<I> est((size_t)n, 0), elen((size_t)n, 0);
    std::vector<I> epool;
    epool.reserve((size_t)ast[n] / 2 + 16);
    std::vector<I> hhead((size_t)n, NONE), hnext((size_t)n, NONE), lasthash((size_t)n, 0);
    std::vector<I> pivots;
    pivots.reserve((size_t)n);
    std::vector<char> is_dense((size_t)n, 0);

    // ---- dense rows are pulled out and ordered last -------------------------
    double dth = 10.0 * dense_scale * std::sqrt((double)n);
    if (dth < 16.0) dth = 16.0;
    if (dth > (double)n) dth = (double)n;
    I ndense = 0;
    for (I i = 0; i < n; i++)
        if ((double)alen[i] > dth) {
            is_dense[i] = 1;
            nv[i] = 0;
            ndense++;
        }
    st.ndense = ndense;
    const I nlive = n - ndense;
    for (I i = 0; i < n; i++) {
        if (is_dense[i]) continue;
        I d = 0;
        for (I p = ast[i]; p < ast[i] + alen[i]; p++)
            if (!is_dense[adj[p]]) d++;
        degree[i] = d;
    }
    std::vector<I> tail((size_t)n + 1, NONE);
    auto dl_insert = [&](I i, I d) {
        nxt[i] = head[d];
        prv[i] = NONE;
        if (head[d] != NONE) prv[head[d]] = i;
        else tail[d] = i;
        head[d] = i;
    };
    // Ties among the variables of a freshly formed element: last-in-first-out like the classic
    // implementations (default), or first-in-first-out (CHIP_AMD_FIFO: a variable that has been
    // waiting at this degree goes before one that just joined it, which spreads equal-degree pivots
    // over independent subtrees; same kind of fill, often a shallower tree, sometimes a deeper one)
    static const bool lifo = getenv("CHIP_AMD_FIFO") == nullptr;
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
    auto tick = [&](int k, std::chrono::steady_clock::time_point &t0) {
        if (!timing) return;
        const auto t1 = std::chrono::steady_clock::now();
        tph[k] += std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
    };
    while (nelim < nlive) {
        auto tp0 = std::chrono::steady_clock::now();
        while (mindeg <= n && head[mindeg] == NONE) mindeg++;
        const I me = head[mindeg];
        dl_remove(me, mindeg);
        I nvpiv = nv[me];
        nelim += nvpiv;
        nv[me] = -nvpiv;

        // ---- form the new element: union of me's variables and the variables
        //      of every element adjacent to me (those elements are absorbed)
        if (sizeof(I) == 4 && epool.size() > (size_t)2000000000) return -77; // retry with 64-bit indices
        const I mstart = (I)epool.size();
        I degme = 0;
        for (I p = ast[me]; p < ast[me] + aelen[me]; p++) {
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
        for (I p = ast[me] + aelen[me]; p < ast[me] + alen[me]; p++) {
            const I i = adj[p];
            const I nvi = nv[i];
            if (nvi > 0) {
                degme += nvi;
                nv[i] = -nvi;
                epool.push_back(i);
                dl_remove(i, degree[i]);
            }
        }
        est[me] = mstart;
        elen[me] = (I)epool.size() - mstart;
        const I mend = mstart + elen[me];
        w[me] = 1;
        lemax = std::max(lemax, degme);

        tick(0, tp0);
        // ---- pass 1: w[e] - wflg = |Le \ Lme| for every element touching Lme
        for (I q = mstart; q < mend; q++) {
            const I i = epool[q];
            const I nvi = -nv[i];
            for (I p = ast[i]; p < ast[i] + aelen[i]; p++) {
                const I e = adj[p];
                const i64 we = w[e];
                if (we >= wflg) w[e] = we - nvi;
                else if (we != 0) w[e] = degree[e] + wflg - nvi;
            }
        }
        tick(1, tp0);
        // ---- pass 2: prune each variable's list, approximate degree, hash
        for (I q = mstart; q < mend; q++) {
            const I i = epool[q];
            const I nvi = -nv[i];
            const I p1 = ast[i], p2 = p1 + aelen[i], pe = p1 + alen[i];
            I pn = p1, deg = 0;
            uint64_t hash = 0;
            for (I p = p1; p < p2; p++) {
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
            const I p3 = pn;
            for (I p = p2; p < pe; p++) {
                const I j = adj[p];
                const I nvj = nv[j];
                if (nvj > 0) {
                    deg += nvj;
                    adj[pn++] = j;
                    hash += (uint64_t)j;
                }
            }
            if (pn == p1) {
                // nothing outside the new element: eliminate i together with me
                vparent[i] = me;
                nvpiv += nvi;
                nelim += nvi;
                degme -= nvi;
                nv[i] = 0;
                aelen[i] = -1;
            } else {
                degree[i] = std::min(degree[i], deg);
                adj[pn] = adj[p3];
                adj[p3] = adj[p1];
                adj[p1] = me;
                alen[i] = pn - p1 + 1;
                aelen[i] = p3 - p1 + 1;
                const I h = (I)(hash % (uint64_t)n);
                lasthash[i] = h;
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
                const I ln = alen[a], eln = aelen[a];
                for (I p = ast[a] + 1; p < ast[a] + ln; p++) w[adj[p]] = wflg;
                I pb = a, b = hnext[a];
                while (b != NONE) {
                    bool same = (alen[b] == ln && aelen[b] == eln);
                    for (I p = ast[b] + 1; same && p < ast[b] + ln; p++)
                        if (w[adj[p]] != wflg) same = false;
                    if (same) {
                        vparent[b] = a;
                        nv[a] += nv[b]; // both negative here
                        nv[b] = 0;
                        aelen[b] = -1;
                        hnext[pb] = hnext[b];
                        b = hnext[b];
                    } else {
                        pb = b;
                        b = hnext[b];
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
        // fill statistics in the style of amd::Info (used by ldlsolvers/auto.rs:69-77)
        {
            const double f = (double)nvpiv, r = (double)(degme + ndense);
            const double lnzme = f * r + (f - 1) * f / 2.0;
            st.lnz += lnzme;
            st.ndiv += lnzme;
            const double s = f * r * r + r * (f - 1) * f + (f - 1) * f * (2 * f - 1) / 6.0;
            st.nmultsubs_ldl += (s + lnzme) / 2.0;
        }
    }
    if (ndense > 0) {
        const double f = (double)ndense;
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

int amd_order(i64 n, const i64 *Ap, const i64 *Ai, double dense_scale, std::vector<i64> &perm,
              AmdInfo *info) {
    const i64 nnz = n > 0 ? Ap[n] : 0;
    if (n < ((i64)1 << 30) && 2 * nnz < ((i64)1 << 31) - 16) {
        const int rc = amd_order_impl<i32>(n, Ap, Ai, dense_scale, perm, info);
        if (rc != -77) return rc;
    }
    return amd_order_impl<i64>(n, Ap, Ai, dense_scale, perm, info);
}

} // namespace chip
