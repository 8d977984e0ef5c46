// bundle_ir.hip -- k_bundle_ir: a whole KKT solve with iterative refinement in one persistent launch (bundles + folded top)
// (one of the translation units behind kernels.hpp; the design rules and the reference citations are in
// dev_common.hpp)
#include "dev_common.hpp"
#include "bundle_symv.hpp"
#include "grid_sync.hpp"

namespace chip {
namespace dev {

namespace {

// ---------------------------------------------------------------------------
// k_bundle_ir: a WHOLE KKT solve with iterative refinement (directldlkktsolver.rs:168-189, :266-321) in ONE
// persistent launch, for systems that consist of subtree bundles plus at most TOPFOLD_MAX folded top rows
// (config 3: 1000 bundles + the budget row; config 4: a forest of bundles, no top).  Per refinement
// round every workgroup does, for its bundle with the vector slice in LDS throughout:
//     forward sweep -> [grid barrier: top rows] -> backward sweep -> candidate x (+ dx) -> residual
//     e = b - K x into the SAME LDS array -> [grid barrier: ||e||inf, top rows of e] -> decision
// so a solve + r refinement rounds costs 1 launch and no host round trip instead of 6 (r + 1) launches,
// 4 (r + 1) one-thread kernels and a device-to-host copy of the norms; x never makes the
// forward -> backward -> residual -> forward trips through HBM, and the right-hand side permutation
// (setrhs) and the un-permutation of the result (getlhs) happen in the staging pass and the final write.
// The refinement decisions are taken ON THE DEVICE, identically by every workgroup: each one reduces the
// bundles' partial norms / top-row shares -- plain stores, read back in a fixed order after the barrier,
// so the result is run-to-run reproducible -- and evaluates the reference's tests.
// Requires all workgroups to be co-resident (cooperative launch; the host checks the occupancy) when the
// top is folded; a forest without top only synchronises for the norms.
// ---------------------------------------------------------------------------
// entries per shot of the three phases inside k_bundle_ir
constexpr int IR_SH_FWD = 3, IR_SH_BWD = 3, IR_SH_SYMV = 3;
// k_bundle_ir runs 256-thread workgroups, four per CU = 4 waves per SIMD: 128 vector registers per thread
// instead of the 64 of the stand-alone bundle kernels (512 threads, 8 waves per SIMD) -- the fused kernel
// carries ~35 pointers plus the software pipeline of the sweeps (entries of the next level in registers), and
// under a 64-register budget it spilled into scratch inside the hot loops.  Each thread takes IR_RPT columns
// of a chunk at a time.
constexpr int IR_RPT = 2;
constexpr int IR_FATCAP = 256; // long rows per level handled cooperatively (more: serially, still correct)

// Forward / backward substitution of bundle b over the slice xs staged in LDS, both streaming the COLUMNS of
// L (Lp, 16-bit local row indices Li16, Lx: no row-major copy of L is needed), level by level with one
// __syncthreads() per level:
//   forward  (qdldl.rs:708-719, x[Li] -= Lx * x[i], column oriented like the reference): a node whose value
//            is final pushes it into the rows of its column with LDS fp64 atomics.  The long rows at the
//            top of a subtree (the u / v columns of a sparse SOC: a thousand entries each) thus receive their
//            contributions from all threads as the wide levels below them complete -- wave-uniform targets
//            are reduced in registers first (lds_scatter_add) -- instead of one cooperative pass per row on a
//            serial chain of one-node levels; pushes into the folded top rows (row index >= nloc) are this
//            bundle's shares of those rows (tacc[0..k), zeroed here);
//   backward (qdldl.rs:737-752): x_j = y_j / d_j - sum over column j of l_ij x_i, ancestors inside the bundle
//            from LDS, the folded top rows from xt; 1 / d_j travels through the pipeline with the column pointers.
// A sweep is a chain of dependent round trips (column pointers -> entries -> LDS), and the entries do NOT
// depend on x.  The levels are therefore walked in CHUNKS of RPT x TW columns (RPT per thread) through a
// software pipeline: while chunk c is processed, the first SH entries of the columns of chunk c + 1 and the
// column pointers of chunk c + 2 are in flight, whatever level they belong to -- in the steady state a chunk
// costs LDS work only (measured before: 5-7 us per 1000-node level, one exposed round trip each).  Columns
// longer than SH take their remaining entries in place; columns longer than THIN_MAX (backward) are shared
// by a wave after the level's last chunk.
template <bool FWDMODE, int SH, int RPT, int TW>
__device__ __forceinline__ void bundle_sweep_cols(const LdlView &v, const BundleView &bv, int b, double *xs,
                                                  const double *xt, double *tacc, int k, int *fat, int &nfat,
                                                  const double *__restrict__ dinv = nullptr) {
    constexpr int CH = RPT * TW;
    const int s0 = bv.bundle_ptr[b], nloc = bv.bundle_ptr[b + 1] - s0;
    const int *lv = bv.blvl + bv.blvl_ptr[b];
    const int nl = bv.blvl_ptr[b + 1] - bv.blvl_ptr[b] - 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (FWDMODE && (int)threadIdx.x < 8) tacc[threadIdx.x] = 0.0;
    double tpart = 0.0; // forward, k == 1 (the usual arrow): the single top row's share in registers
    // chunk iterator: (level step, offset inside the level); step == nl: past the end
    struct Chunk {
        int step, off;
    };
    auto level_of = [&](int step) { return FWDMODE ? step : nl - 1 - step; };
    auto advance = [&](Chunk c) {
        if (c.step >= nl) return c;
        const int l = level_of(c.step);
        if (lv[l] + c.off + CH < lv[l + 1]) return Chunk{c.step, c.off + CH};
        return Chunk{c.step + 1, 0};
    };
    int cb[RPT], ce[RPT];   // pointers of the chunk whose entries are (being) fetched
    int p1b[RPT], p1e[RPT]; // pointers of the chunk after it
    double cd[RPT], p1d[RPT]; // backward: 1 / d of the same columns (dinv == nullptr: xs already holds y / d)
    int ei[RPT][SH];
    double ev[RPT][SH];
    auto request_ptrs = [&](Chunk c, int (&pb)[RPT], int (&pe)[RPT], double (&pd)[RPT]) {
        const int l = level_of(c.step < nl ? c.step : nl - 1);
        const int lb = lv[l] + c.off, le = c.step < nl ? lv[l + 1] : 0;
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int j = lb + (int)threadIdx.x + u * TW;
            pb[u] = j < le ? v.Lp[j] : 0;
            pe[u] = j < le ? v.Lp[j + 1] : 0;
            pd[u] = (!FWDMODE && dinv && j < le) ? dinv[j] : 1.0;
        }
    };
    auto request_entries = [&](const int (&pb)[RPT], const int (&pe)[RPT]) {
#pragma unroll
        for (int u = 0; u < RPT; ++u)
#pragma unroll
            for (int e = 0; e < SH; ++e) {
                const unsigned t = (unsigned)(pb[u] + e);
                const bool ok = (int)t < pe[u];
                ei[u][e] = ok ? (int)v.Li16[t] : -1;
                ev[u][e] = ok ? v.Lx[t] : 0.0;
            }
    };
    Chunk cur{0, 0};
    Chunk nx1 = advance(cur), nx2 = advance(nx1);
    request_ptrs(cur, cb, ce, cd);
    request_entries(cb, ce);
    request_ptrs(nx1, p1b, p1e, p1d);
    while (cur.step < nl) {
        const int l = level_of(cur.step);
        const int lb = lv[l], le = lv[l + 1];
        const int c0 = lb + cur.off;
        const bool level_begins = cur.off == 0, level_ends = c0 + CH >= le;
        // this chunk's data out of the pipeline
        int ci[RPT][SH], tb[RPT], te[RPT];
        double cv[RPT][SH], dj[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            tb[u] = cb[u];
            te[u] = ce[u];
            dj[u] = cd[u];
#pragma unroll
            for (int e = 0; e < SH; ++e) {
                ci[u][e] = ei[u][e];
                cv[u][e] = ev[u][e];
            }
        }
        // refill: entries of the next chunk (its pointers arrived a chunk ago), pointers of the one after
#pragma unroll
        for (int u = 0; u < RPT; ++u) cb[u] = p1b[u], ce[u] = p1e[u], cd[u] = p1d[u];
        request_entries(cb, ce);
        request_ptrs(nx2, p1b, p1e, p1d);
        if (level_begins) {
            __syncthreads(); // forward: every push into this level's nodes has landed; backward: its ancestors are final
            if (!FWDMODE) {
                if (threadIdx.x == 0) nfat = 0;
                if (le - lb > 1) __syncthreads();
            }
        }
        {
            int jr[RPT];
            double yj[RPT], sum[RPT];
            int maxlen = 0;
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                const int j = c0 + (int)threadIdx.x + u * TW;
                jr[u] = j < le ? j : -1;
                sum[u] = 0.0;
                yj[u] = (FWDMODE && j < le) ? xs[j - s0] : 0.0;
                if (!FWDMODE && te[u] - tb[u] > THIN_MAX) { // a long column: shared by a wave below
                    const int slot = atomicAdd(&nfat, 1);
                    if (slot < IR_FATCAP) {
                        fat[slot] = j;
                        xs[j - s0] *= dj[u]; // (its 1 / d now; the wave subtracts the column's sum later)
                        jr[u] = -1;
                        te[u] = tb[u];
                    }
                }
                maxlen = max(maxlen, te[u] - tb[u]);
            }
            // forward: wave-uniform trip count (cross-lane operations below): as long as ANY lane has entries left
            for (int kk = 0; FWDMODE ? (__ballot(kk < maxlen) != 0ull) : (kk < maxlen); kk += SH) {
                int ii[RPT][SH];
                double vv[RPT][SH];
                if (kk == 0) { // (prefetched)
#pragma unroll
                    for (int u = 0; u < RPT; ++u)
#pragma unroll
                        for (int e = 0; e < SH; ++e) {
                            ii[u][e] = (tb[u] + e < te[u]) ? ci[u][e] : -1;
                            vv[u][e] = cv[u][e];
                        }
                } else {
#pragma unroll
                    for (int u = 0; u < RPT; ++u)
#pragma unroll
                        for (int e = 0; e < SH; ++e) {
                            const unsigned t = (unsigned)(tb[u] + kk + e);
                            const bool ok = (int)t < te[u];
                            ii[u][e] = ok ? (int)v.Li16[t] : -1;
                            vv[u][e] = ok ? v.Lx[t] : 0.0;
                        }
                }
#pragma unroll
                for (int u = 0; u < RPT; ++u)
#pragma unroll
                    for (int e = 0; e < SH; ++e) {
                        const int i = ii[u][e];
                        if (FWDMODE) {
                            const double val = vv[u][e] * yj[u];
                            int tgt = -1;
                            if (i >= 0) {
                                if (i < nloc) tgt = i;
                                else if (k == 1) tpart += val;
                                else atomicAdd(&tacc[i - nloc], val);
                            }
                            lds_scatter_add(xs, tgt, -val);
                        } else if (i >= 0) {
                            sum[u] += vv[u][e] * (i < nloc ? xs[i] : xt[i - nloc]);
                        }
                    }
            }
            if (!FWDMODE) {
#pragma unroll
                for (int u = 0; u < RPT; ++u)
                    if (jr[u] >= 0) xs[jr[u] - s0] = xs[jr[u] - s0] * dj[u] - sum[u]; // qdldl.rs:737-752
            }
        }
        if (!FWDMODE && level_ends) { // the level's long columns, one wave each
            __syncthreads();
            const int nf = min(nfat, IR_FATCAP);
            for (int f = wv; f < nf; f += TW / 64) {
                const int j = fat[f];
                double sacc = 0.0;
                for (int t = v.Lp[j] + lane; t < v.Lp[j + 1]; t += 64) {
                    const int i = (int)v.Li16[t];
                    sacc += v.Lx[t] * (i < nloc ? xs[i] : xt[i - nloc]);
                }
                sacc = wave_sum(sacc);
                if (lane == 0) xs[j - s0] -= sacc;
            }
        }
        cur = nx1;
        nx1 = nx2;
        nx2 = advance(nx2);
    }
    __syncthreads();
    if (FWDMODE && k == 1) {
        tpart = wave_sum_all(tpart);
        if (lane == 0 && tpart != 0.0) atomicAdd(&tacc[0], tpart);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Entry-parallel ("flat") sweeps of k_bundle_ir.  The columns of a bundle are numbered level-major, so the entries of
// one level's columns are ONE contiguous range of the CSC arrays of L: the threads stride over that range -- every
// entry knows its row (Li16) and its column (Lj16) -- with FLAT_U independent, coalesced (index, index, value) loads in
// flight per thread and no pointer chase at all; the updates go to the LDS slice as fp64 atomics.  A level then costs one
// round trip plus a barrier whatever its columns look like, where the column-per-thread form (bundle_sweep_cols) walked
// pointer -> entries -> update chains a few columns at a time: measured on config 3 (1000 bundles of 3003 nodes, 256
// threads) a sweep's time grew by 4.4 us per 250 nodes, 1.7 TB/s marginal -- latency times trips, not bandwidth.
//   forward : x_i -= l_ij y_j     for the entries of the columns j of level l, l ascending (qdldl.rs:708-719)
//   backward: x_j -= l_ij x_i     after x_j *= 1 / d_j for the whole slice, l descending          (qdldl.rs:737-752)
// ---------------------------------------------------------------------------
constexpr int FLAT_U = 4;
constexpr int FLAT_MAXLEV = 64;  // levels of a bundle the flat sweeps keep entry pointers for (more: column per thread)
constexpr int FLAT_MIN_NODES = 512; // smaller bundles keep the column-per-thread form (a level must fill the workgroup)
// lev_e[0 .. nl]: first entry of every level's columns (LDS, filled once per launch by flat_level_table)
__device__ __forceinline__ void flat_level_table(const LdlView &v, const BundleView &bv, int b, int *lev_e) {
    const int *lv = bv.blvl + bv.blvl_ptr[b];
    const int nl = bv.blvl_ptr[b + 1] - bv.blvl_ptr[b] - 1;
    if ((int)threadIdx.x <= nl && nl <= FLAT_MAXLEV) lev_e[threadIdx.x] = v.Lp[lv[threadIdx.x]];
}
template <bool FWDMODE, int TW>
__device__ __forceinline__ void bundle_sweep_flat(const LdlView &v, const BundleView &bv, int b, double *xs,
                                                  const double *xt, double *tacc, int k, const int *lev_e) {
    const int s0 = bv.bundle_ptr[b], nloc = bv.bundle_ptr[b + 1] - s0;
    const int nl = bv.blvl_ptr[b + 1] - bv.blvl_ptr[b] - 1;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid)); // (no hoisting of this phase's address arithmetic out of the caller's round loop)
    const int lane = tid & 63;
    if (FWDMODE && tid < 8) tacc[tid] = 0.0;
    if (!FWDMODE)
        for (int i = tid; i < nloc; i += TW) xs[i] *= v.Dinv[s0 + i];
    double tpart = 0.0;
    // a stream of batches of TW * FLAT_U entries, level after level; the next batch's loads are issued before the
    // current one is consumed -- across a level boundary too (the entries do not depend on x), so a level costs its
    // barrier plus the LDS work, not a round trip
    int step = 0, base = 0, ee = 0;
    auto level_range = [&](int st_, int &eb_, int &ee_) {
        const int l = FWDMODE ? st_ : nl - 1 - st_;
        eb_ = lev_e[l];
        ee_ = lev_e[l + 1];
    };
    auto skip_empty = [&]() { // -> first non-empty level at or after `step`
        while (step < nl) {
            level_range(step, base, ee);
            if (base < ee) return;
            ++step;
        }
    };
    lds_barrier(); // (lev_e, the scaled slice)
    if (TW == 512) {
        // (80 registers per thread in the 512-thread variant: no second batch in flight)
        for (int st_ = 0; st_ < nl; ++st_) {
            int eb_, ee_;
            level_range(st_, eb_, ee_);
            for (int bs = eb_; bs < ee_; bs += TW * FLAT_U) {
                int ii[FLAT_U], jj[FLAT_U];
                double vv[FLAT_U];
#pragma unroll
                for (int u = 0; u < FLAT_U; ++u) {
                    const int t = bs + u * TW + tid;
                    const bool ok = t < ee_;
                    ii[u] = ok ? (int)v.Li16[t] : -1;
                    jj[u] = ok ? (int)v.Lj16[t] : 0;
                    vv[u] = ok ? v.Lx[t] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < FLAT_U; ++u) {
                    const int i = ii[u];
                    if (i < 0) continue;
                    if (FWDMODE) {
                        const double val = vv[u] * xs[jj[u]];
                        if (i < nloc) atomicAdd(&xs[i], -val);
                        else if (k == 1) tpart += val;
                        else atomicAdd(&tacc[i - nloc], val);
                    } else {
                        atomicAdd(&xs[jj[u]], -(vv[u] * (i < nloc ? xs[i] : xt[i - nloc])));
                    }
                }
            }
            lds_barrier();
        }
        if (FWDMODE && k == 1) {
            tpart = wave_sum_all(tpart);
            if (lane == 0 && tpart != 0.0) atomicAdd(&tacc[0], tpart);
        }
        lds_barrier();
        return;
    }
    skip_empty();
    int ci[FLAT_U], cj[FLAT_U], ni[FLAT_U], nj[FLAT_U];
    double cv[FLAT_U], nv[FLAT_U];
    auto request = [&](int bs, int en, int *ii, int *jj, double *vv) {
#pragma unroll
        for (int u = 0; u < FLAT_U; ++u) {
            const int t = bs + u * TW + tid;
            const bool ok = t < en;
            ii[u] = ok ? (int)v.Li16[t] : -1;
            jj[u] = ok ? (int)v.Lj16[t] : 0;
            vv[u] = ok ? v.Lx[t] : 0.0;
        }
    };
    if (step < nl) request(base, ee, ci, cj, cv);
    while (step < nl) {
        // the batch after this one
        int nstep = step, nbase = base + TW * FLAT_U, nee = ee;
        if (nbase >= nee) {
            nstep = step + 1;
            while (nstep < nl) {
                level_range(nstep, nbase, nee);
                if (nbase < nee) break;
                ++nstep;
            }
        }
        if (nstep < nl) request(nbase, nee, ni, nj, nv);
#pragma unroll
        for (int u = 0; u < FLAT_U; ++u) {
            const int i = ci[u];
            if (i < 0) continue;
            if (FWDMODE) {
                const double val = cv[u] * xs[cj[u]];
                if (i < nloc) atomicAdd(&xs[i], -val);
                else if (k == 1) tpart += val;
                else atomicAdd(&tacc[i - nloc], val);
            } else {
                atomicAdd(&xs[cj[u]], -(cv[u] * (i < nloc ? xs[i] : xt[i - nloc])));
            }
        }
        if (nstep != step) lds_barrier(); // the level is complete
        step = nstep;
        base = nbase;
        ee = nee;
#pragma unroll
        for (int u = 0; u < FLAT_U; ++u) {
            ci[u] = ni[u];
            cj[u] = nj[u];
            cv[u] = nv[u];
        }
    }
    if (FWDMODE && k == 1) {
        tpart = wave_sum_all(tpart);
        if (lane == 0 && tpart != 0.0) atomicAdd(&tacc[0], tpart);
    }
    lds_barrier();
}

// The residual of k_bundle_ir in the split LDS layout of bundle_symv_split, entry-parallel: the rows of the non-leaf
// nodes are walked as ONE flat range of U entries (row Urow16, column Ucol16), both directions of every entry as LDS
// atomics; the leaf rows (their e has no place in LDS) four rows per thread at once, their result to the spill vector.
template <int TW>
__device__ __forceinline__ void bundle_symv_flat(const LdlView &v, const BundleView &bv, const double *x,
                                                 const double *__restrict__ b, double *spill, double *xs, double *red,
                                                 int k, int bid, const double *xt, double *out_norm, double *out_share) {
    const int *__restrict__ Up = v.Up;
    const unsigned short *__restrict__ Ucol16 = v.Ucol16, *__restrict__ Urow16 = v.Urow16;
    const double *__restrict__ Ux = v.Ux;
    const int s0 = bv.bundle_ptr[bid], nloc = bv.bundle_ptr[bid + 1] - s0;
    const int nleaf = bv.blvl[bv.blvl_ptr[bid] + 1] - s0, nin = nloc - nleaf;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid)); // (no hoisting of this phase's address arithmetic out of the caller's round loop)
    auto xpos = [&](int t) { return t < nleaf ? t : nloc + (t - nleaf); }; // x of non-leaf t
    // leaf rows: pointers of this thread's first four rows are requested before the staging pass
    constexpr int LR = 4, LS = 3;
    int tb[LR], te[LR];
#pragma unroll
    for (int u = 0; u < LR; ++u) {
        const int i = tid + u * TW;
        tb[u] = i < nleaf ? Up[s0 + i] : 0;
        te[u] = i < nleaf ? Up[s0 + i + 1] : 0;
    }
    const int fb = Up[s0 + nleaf], fe = Up[s0 + nloc]; // the flat range: rows of the non-leaf nodes
    for (int t = tid; t < nin; t += TW) {
        const double xv = x[s0 + nleaf + t], bv_ = b[s0 + nleaf + t];
        xs[xpos(t)] = xv;
        xs[nleaf + t] = bv_;
    }
    double tpart = 0.0;
    __shared__ double tacc3[8];
    if (k > 1 && tid < 8) tacc3[tid] = 0.0;
    double mleaf = 0.0;
    bool nan = false;
    __syncthreads();
    // ---- leaf rows ----
    for (int w0 = 0; w0 < nleaf; w0 += LR * TW) {
        int jj[LR][LS];
        double vv[LR][LS], xi[LR], bi[LR], acc[LR];
#pragma unroll
        for (int u = 0; u < LR; ++u) {
            const int i = w0 + tid + u * TW;
            xi[u] = i < nleaf ? x[s0 + i] : 0.0;
            bi[u] = i < nleaf ? b[s0 + i] : 0.0;
            acc[u] = 0.0;
#pragma unroll
            for (int q = 0; q < LS; ++q) {
                const int t = tb[u] + q;
                const bool ok = t < te[u];
                jj[u][q] = ok ? (int)Ucol16[t] : -1;
                vv[u][q] = ok ? Ux[t] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < LR; ++u) {
            const int i = w0 + tid + u * TW;
            auto apply = [&](int j, double val) {
                if (j >= nloc) {
                    acc[u] += val * xt[j - nloc];
                    if (k == 1) tpart += val * xi[u];
                    else atomicAdd(&tacc3[j - nloc], val * xi[u]);
                } else if (j == i) {
                    acc[u] += val * xi[u];
                } else {
                    acc[u] += val * xs[xpos(j - nleaf)];
                    atomicAdd(&xs[j], -(val * xi[u]));
                }
            };
#pragma unroll
            for (int q = 0; q < LS; ++q)
                if (jj[u][q] >= 0) apply(jj[u][q], vv[u][q]);
            for (int t = tb[u] + LS; t < te[u]; ++t) apply((int)Ucol16[t], Ux[t]); // (a leaf with a long row: rare)
            if (i < nleaf) {
                const double val = bi[u] - acc[u];
                spill[s0 + i] = val;
                if (val != val) nan = true;
                else mleaf = fmax(mleaf, fabs(val));
            }
            const int in = i + LR * TW;
            tb[u] = in < nleaf ? Up[s0 + in] : 0;
            te[u] = in < nleaf ? Up[s0 + in + 1] : 0;
        }
    }
    // ---- rows of the non-leaf nodes: flat over their entries, the next batch in flight while this one is consumed ----
    int ii[FLAT_U], jj[FLAT_U], ni[FLAT_U], nj[FLAT_U];
    double vv[FLAT_U], nv[FLAT_U];
    auto request = [&](int bs, int *pi, int *pj, double *pv) {
#pragma unroll
        for (int u = 0; u < FLAT_U; ++u) {
            const int t = bs + u * TW + tid;
            const bool ok = t < fe;
            pi[u] = ok ? (int)Urow16[t] : -1;
            pj[u] = ok ? (int)Ucol16[t] : 0;
            pv[u] = ok ? Ux[t] : 0.0;
        }
    };
    if (fb < fe) request(fb, ii, jj, vv);
    for (int base = fb; base < fe; base += TW * FLAT_U) {
        if (base + TW * FLAT_U < fe) request(base + TW * FLAT_U, ni, nj, nv);
        else {
#pragma unroll
            for (int u = 0; u < FLAT_U; ++u) ni[u] = -1;
        }
#pragma unroll
        for (int u = 0; u < FLAT_U; ++u) {
            const int i = ii[u], j = jj[u];
            if (i < 0) continue;
            const double xi = xs[xpos(i - nleaf)];
            if (j >= nloc) {
                atomicAdd(&xs[i], -(vv[u] * xt[j - nloc]));
                if (k == 1) tpart += vv[u] * xi;
                else atomicAdd(&tacc3[j - nloc], vv[u] * xi);
            } else if (j == i) {
                atomicAdd(&xs[i], -(vv[u] * xi));
            } else {
                atomicAdd(&xs[i], -(vv[u] * xs[xpos(j - nleaf)]));
                atomicAdd(&xs[j], -(vv[u] * xi));
            }
        }
#pragma unroll
        for (int u = 0; u < FLAT_U; ++u) {
            ii[u] = ni[u];
            jj[u] = nj[u];
            vv[u] = nv[u];
        }
    }
    __syncthreads();
    double m = mleaf;
    for (int i = nleaf + tid; i < nloc; i += TW) {
        const double val = xs[i];
        if (val != val) nan = true;
        else m = fmax(m, fabs(val));
    }
    for (int i = tid; i < nleaf; i += TW) xs[i] = spill[s0 + i]; // (every thread re-reads what it wrote itself)
    m = block_max(m, red);
    const bool anynan = __syncthreads_or(nan);
    if (tid == 0)
        __hip_atomic_store(out_norm, anynan ? __longlong_as_double(0x7ff8000000000000ll) : m, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (k == 1) {
        tpart = block_sum(tpart, red);
        if (tid == 0) __hip_atomic_store(out_share, tpart, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (k > 1) {
        __syncthreads();
        if (tid == 0)
            for (int i = 0; i < k; ++i)
                __hip_atomic_store(out_share + i, tacc3[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

constexpr int IR_MAXRUNS = 32;
// per-workgroup state of k_bundle_ir, kept in LDS so that nothing but loop counters stays in registers
// across the sweeps (their inner loops need the whole 64-register budget of 8 waves per SIMD)
struct IrState {
    double normb, norme, lastnorme;
    int rounds, ok, done, sel, par, gen, pad;
    double btop[8], rtop[8], dxt[8], curt[8], candt[8];
    double dinvt[8], ltt[64], ktt[64]; // constants of the folded top: 1/d, L(top, top), K(top, top) (full rows)
    double tacc[8];                    // this bundle's shares of the top rows in the forward sweep
    double pubv[20];                   // k_bundle_irs: a barrier's results: [0, k) forward sums, [k] ||e||, [k + 1] ||b||, [k + 2, 2k + 2) residual sums
    int runs[3 * IR_MAXRUNS];          // run-length form of the bundle's slice of the permutation
};

// TW threads per workgroup: 256 when four workgroups fit a CU (4 waves per SIMD, 128 registers: config 3's
// 3003-node bundles), 512 for bundles whose LDS slice only lets three in (config 4's 6007-node bundles: 6
// waves per SIMD, 80 registers -- with 256 threads only 12 of a CU's 32 wave slots would be used)
// GR: grouped fold (GFoldView): every workgroup owns ONE bundle (nb <= gridDim.x); the top of a bundle's tree is
// solved by the LAST of the tree's workgroups to arrive at the group's counter, right before it arrives at the grid
// barrier -- the others find the result in the group's record after the barrier.
template <int TW, bool GR>
__global__ __launch_bounds__(TW) __attribute__((amdgpu_waves_per_eu(TW == 512 ? 6 : 4, TW == 512 ? 6 : 4)))
void k_bundle_ir(LdlView v, BundleView bv, FoldView fold, IrView ir, GFoldView gf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *xs = (double *)smem;
    __shared__ double red[16];
    __shared__ int fat[IR_FATCAP];
    __shared__ int nfat;
    __shared__ int lev_e[FLAT_MAXLEV + 1];
    __shared__ IrState st;
    const int nb = bv.nb, G = gridDim.x, tid = threadIdx.x;
    const int grp = GR ? gf.bgrp[blockIdx.x] : -1;
    const int gbase = (GR && grp >= 0) ? gf.ptr[grp] : 0;
    const int k = GR ? (grp >= 0 ? gf.ptr[grp + 1] - gbase : 0) : fold.k; // (GR: differs between workgroups)
    const bool folded = GR || k > 0;                                       // a barrier in the middle of every round
    const int NF = GR ? 0 : (k ? fold.NF : ir.N);
    const int gnb = (GR && grp >= 0) ? gf.bptr[grp + 1] - gf.bptr[grp] : 0; // workgroups of this group
    const bool gfirst = GR && grp >= 0 && (int)blockIdx.x == gf.bptr[grp];
    int gph = 0;                                                            // group phases passed so far
    auto topnode = [&](int i) { return GR ? gf.node[gbase + i] : NF + i; };
    double *grec = (GR && grp >= 0) ? gf.rec + (size_t)(grp < 0 ? 0 : grp) * 64 : nullptr;  // [2][32]
    FoldView lfold = fold; // what the residual body needs to know: the number of folded rows of THIS workgroup
    lfold.k = k;
    if (ir.test_drop && (int)blockIdx.x == G - 1 && G > 1) return; // (tests: a launch that is not co-resident)
    const bool single = nb <= G; // one bundle per workgroup: its residual never leaves LDS
    // partial results: device-coherent stores before a barrier, reduced in a fixed order by its last arriver
    double *pnb = ir.part;                  // [nb]       ||b||inf of the bundles' rows
    double *pn = pnb + nb;                  // [2][nb]    ||e||inf of the bundles' rows
    const int kp = GR ? 0 : k;              // (GR: the shares live in gf.fsh / gf.rsh, k differs between workgroups)
    double *shf = pn + 2 * nb;              // [nb*k]     forward sweep: shares of the top rows
    double *shs = shf + (size_t)nb * kp;    // [2][nb*k]  residual: shares of (K x)[top]
    double *pub = shs + 2 * (size_t)nb * kp; // [2][32]   published reductions: [0..8) forward sums, [8] ||e||,
                                            //            [9] ||b||, [16..24) residual sums
    auto rhs_at = [&](int j) { // permuted right-hand side entry j (directldlkktsolver.rs:160-166)
        const int o = ir.perm[j];
        return o < ir.n ? ir.rx[o] : (o < ir.n + ir.m ? ir.rz[o - ir.n] : 0.0);
    };
    if (tid == 0) {
        st.normb = st.norme = st.lastnorme = 0.0;
        st.rounds = 0;
        st.ok = 1;
        st.done = 0;
        st.sel = 0;
        st.par = 0;
        st.gen = 0;
        if (blockIdx.x == 0) {
            ir.res[0] = 0; // "did not finish" until the verdict is written at the very end
            ir.res[2] = 0;
        }
    }
    if (tid < 64) {
        st.ltt[tid] = 0.0;
        st.ktt[tid] = 0.0;
    }
    __syncthreads();
    // constants of the folded top (chains of dependent loads): fetched by three different waves right before
    // the first grid barrier, where the workgroup waits anyway -- not ahead of the staging pass
    auto load_top_constants = [&]() {
        if (tid < 8) {
            st.btop[tid] = tid < k ? rhs_at(topnode(tid)) : 0.0;
            // (bp must hold the WHOLE permuted right-hand side afterwards: a second solve() without a new
            // setrhs() restarts from bp, directldlkktsolver.rs:168-175 keeps self.b)
            if (tid < k && (GR ? gfirst : blockIdx.x == 0)) ir.bp[topnode(tid)] = st.btop[tid];
            st.curt[tid] = 0.0;
            st.dinvt[tid] = tid < k ? v.Dinv[topnode(tid)] : 0.0;
        } else if (tid >= 64 && tid < 64 + k * k) {
            const int ti = (tid - 64) / k, tj = (tid - 64) % k;
            const int q = GR ? gf.tt[grp * 64 + ti * 8 + tj] : fold.tt[tid - 64];
            if (q >= 0) st.ltt[ti * 8 + tj] = v.Lx[q];
        } else if (tid >= 128 && tid < 128 + k) {
            const int i = tid - 128;
            const int *sp = GR ? gf.sp + gbase : fold.sp, *scol = GR ? gf.scol : fold.scol, *sslot = GR ? gf.sslot : fold.sslot;
            for (int t = sp[i]; t < sp[i + 1]; ++t) st.ktt[i * 8 + scol[t]] += v.Ux[sslot[t]];
        }
    };
    // GR: sums of the group's shares sh[q * 8 + i] over its bundles q, in a fixed order, by wave 0 (lane = (q mod 8, i));
    // the totals land in st.tacc (the forward sweep's own shares have been stored by then)
    auto group_sums = [&](const double *sh) {
        if (tid < 64) {
            const int i = tid & 7, q8 = tid >> 3;
            double part = 0.0;
            for (int q = gf.bptr[grp] + q8; q < gf.bptr[grp + 1]; q += 8) part += ir_load(&sh[(size_t)q * 8 + i]);
            part += __shfl_xor(part, 8, 64);
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            if (tid < 8) st.tacc[tid] = part;
        }
        __syncthreads();
    };
    if (!GR && k == 0) load_top_constants(); // (a forest: only btop / curt are cleared)
    int dbgn = 0;
    auto stamp = [&]() { // diagnostics (CHIP_IR_DEBUG): phase boundaries of workgroups 0 and G/2 on the 100 MHz clock
        if (ir.dbg_all) {
            if (tid == 0 && dbgn < 31) {
                if (dbgn == 0) // HW_REG_HW_ID (4) in the low word, HW_REG_XCC_ID (20) in the high word
                    ir.dbg_all[(size_t)blockIdx.x * 32] =
                        (long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4) |
                        ((long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 20) << 32);
                ir.dbg_all[(size_t)blockIdx.x * 32 + 1 + dbgn++] = wall_clock64();
            }
        } else if (ir.dbg && tid == 0 && (blockIdx.x == 0 || (int)blockIdx.x == G / 2) && dbgn < 64)
            ir.dbg[(blockIdx.x ? 64 : 0) + dbgn++] = wall_clock64();
    };
    // the last arriver of a barrier: fixed-order reductions of what the workgroups stored before arriving
    auto reduce_forward = [&](int par) {
        for (int i = 0; i < k; ++i) {
            double part = 0.0;
            for (int q = tid; q < nb; q += TW) part += ir_load(&shf[(size_t)q * k + i]);
            part = block_sum(part, red);
            if (tid == 0) ir_store(&pub[par * 32 + i], part);
        }
    };
    auto reduce_residual = [&](int par, bool first) { // norms NaN propagating; every load is issued before any reduction
        double mb = 0.0, m = 0.0, part0 = 0.0; // (no private array: it would live in scratch memory)
        for (int q = tid; q < nb; q += TW) {
            if (first) mb = nanmax(mb, ir_load(&pnb[q]));
            m = nanmax(m, ir_load(&pn[(size_t)par * nb + q]));
            if (ir.ir_enable && !GR && k >= 1) part0 += ir_load(&shs[(size_t)par * nb * k + (size_t)q * k]);
        }
        if (GR) // the top rows of every group: reduced by the groups' last arrivers, published in their records
            for (int q = tid; q < gf.ng; q += TW) {
                if (first) mb = nanmax(mb, ir_load(&gf.rec[(size_t)q * 64 + par * 32 + 25]));
                m = nanmax(m, ir_load(&gf.rec[(size_t)q * 64 + par * 32 + 24]));
            }
        if (first) {
            mb = block_nanmax(mb, red);
            if (tid == 0) ir_store(&pub[par * 32 + 9], mb);
        }
        m = block_nanmax(m, red);
        if (tid == 0) ir_store(&pub[par * 32 + 8], m);
        if (!GR)
            for (int i = 0; i < k; ++i) {
                double part = part0;
                if (i > 0) {
                    part = 0.0;
                    if (ir.ir_enable)
                        for (int q = tid; q < nb; q += TW) part += ir_load(&shs[(size_t)par * nb * k + (size_t)q * k + i]);
                }
                const double tot = block_sum(part, red);
                if (tid == 0) ir_store(&pub[par * 32 + 16 + i], tot);
            }
    };
    // the reference's decisions about the candidate of round `round`, whose residual sums were published with
    // parity `par`: thread 0 of every workgroup alike, state in LDS
    auto decide = [&](int round, int par) {
        if (tid == 0) {
            double m = ir_load(&pub[par * 32 + 8]);
            if (round == 0) {
                double nbm = ir_load(&pub[par * 32 + 9]);
                if (!GR)
                    for (int i = 0; i < k; ++i) nbm = nanmax(nbm, fabs(st.btop[i]));
                st.normb = nbm;
            }
            for (int i = 0; i < (GR ? 0 : k); ++i) { // (GR: the groups' last arrivers did this, see the residual phase)
                double sacc = ir_load(&pub[par * 32 + 16 + i]);
                for (int c = 0; c < k; ++c) sacc += st.ktt[i * 8 + c] * st.candt[c];
                // (no refinement: the top entries of x take part in the finiteness test instead)
                st.rtop[i] = ir.ir_enable ? st.btop[i] - sacc : st.candt[i];
                m = nanmax(m, fabs(st.rtop[i]));
            }
            // (rtop = top rows of the candidate's residual: consumed only if it is accepted and a round follows)
            const double newnorm = m, tol = ir.abstol + ir.reltol * st.normb;
            bool accept, done = false;
            if (round == 0) {
                accept = true;
                st.norme = newnorm;
                if (!(newnorm - newnorm == 0.0)) { // non-finite (:284-286; without refinement: x.is_finite(), :180)
                    st.ok = 0;
                    done = true;
                } else if (!ir.ir_enable || ir.maxiter <= 0 || newnorm <= tol) {
                    done = true;
                }
            } else {
                st.rounds += 1;
                if (!(newnorm - newnorm == 0.0)) { // :305-307
                    st.ok = 0;
                    accept = false;
                    done = true;
                } else {
                    const double improved = st.lastnorme / newnorm;
                    accept = !(improved < ir.stopratio) || improved > 1.0; // :309-318
                    if (improved < ir.stopratio) done = true;
                    if (accept) st.norme = newnorm;
                }
            }
            if (accept) {
                st.sel ^= 1;
                for (int i = 0; i < k; ++i) st.curt[i] = st.candt[i];
            }
            if (!done && (st.rounds >= ir.maxiter || st.norme <= tol)) done = true; // :288-293
            st.lastnorme = st.norme;
            st.done = done ? 1 : 0;
        }
        __syncthreads();
    };
    stamp();
    // Round 0 solves for x from b, round r > 0 for the correction dx from the residual of the accepted x.
    // With a folded top the verdict on round r - 1's candidate is taken at the barrier in the MIDDLE of
    // round r (whose forward sweep has then run speculatively on the residual in LDS): one barrier per
    // round instead of two; only the last possible round ends with a barrier of its own.
    bool pending = false; // a candidate whose residual partials have been stored but not yet reduced
    for (int round = 0;; ++round) {
        const int par = round & 1;
        bool stop = false;
        for (int b = blockIdx.x; b < nb; b += G) {
            const int s0 = bv.bundle_ptr[b], nloc = bv.bundle_ptr[b + 1] - s0;
            __syncthreads();
            const int nruns = ir.runs ? ir.run_ptr[b + 1] - ir.run_ptr[b] : 0;
            if (round == 0 && nruns > 0 && nruns <= IR_MAXRUNS) {
                // the permutation as a few contiguous runs (descriptors in LDS): one round trip, coalesced
                if (tid < 3 * nruns) st.runs[tid] = ir.runs[3 * ir.run_ptr[b] + tid];
                __syncthreads();
                double mx = 0.0;
                bool nan = false;
                int r = 0;
                for (int i = tid; i < nloc; i += TW) {
                    while (i >= st.runs[3 * r] + st.runs[3 * r + 2]) ++r; // (runs ascend in the local index)
                    const int o = st.runs[3 * r + 1] + (i - st.runs[3 * r]);
                    const double val = o < ir.n ? ir.rx[o] : (o < ir.n + ir.m ? ir.rz[o - ir.n] : 0.0);
                    xs[i] = val;
                    ir.bp[s0 + i] = val;
                    if (val != val) nan = true;
                    else mx = fmax(mx, fabs(val));
                }
                mx = block_max(mx, red);
                const bool anynan = __syncthreads_or(nan);
                if (tid == 0) ir_store(&pnb[b], anynan ? __longlong_as_double(0x7ff8000000000000ll) : mx);
            } else if (round == 0) {
                double mx = 0.0;
                bool nan = false;
                for (int i0 = tid; i0 < nloc; i0 += 4 * TW) { // four independent perm -> rhs chains in flight
                    int o[4];
                    double val[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) o[u] = i0 + u * TW < nloc ? ir.perm[s0 + i0 + u * TW] : -1;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        val[u] = o[u] < 0 ? 0.0 : (o[u] < ir.n ? ir.rx[o[u]] : (o[u] < ir.n + ir.m ? ir.rz[o[u] - ir.n] : 0.0));
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (o[u] >= 0) {
                            const int i = i0 + u * TW;
                            xs[i] = val[u];
                            ir.bp[s0 + i] = val[u];
                            if (val[u] != val[u]) nan = true;
                            else mx = fmax(mx, fabs(val[u]));
                        }
                }
                mx = block_max(mx, red);
                const bool anynan = __syncthreads_or(nan);
                if (tid == 0) ir_store(&pnb[b], anynan ? __longlong_as_double(0x7ff8000000000000ll) : mx);
            } else if (!single) {
                for (int i = tid; i < nloc; i += TW) xs[i] = ir.ebuf[s0 + i];
            } // (single: xs still holds this bundle's residual)
            __syncthreads();
            stamp();
            // (entry-parallel sweeps for bundles whose levels fill the workgroup; decided per bundle, the same way in
            // every phase)
            const bool flat = ir.flat && nloc >= FLAT_MIN_NODES && bv.blvl_ptr[b + 1] - bv.blvl_ptr[b] - 1 <= FLAT_MAXLEV;
            if (flat && (round == 0 || !single)) {
                __syncthreads(); // (!single: the previous bundle's sweeps are done with the table)
                flat_level_table(v, bv, b, lev_e);
            }
            if (flat) bundle_sweep_flat<true, TW>(v, bv, b, xs, nullptr, st.tacc, k, lev_e);
            else bundle_sweep_cols<true, IR_SH_FWD, IR_RPT, TW>(v, bv, b, xs, nullptr, st.tacc, k, fat, nfat);
            stamp();
            // this bundle's shares of the top rows of L (accumulated by the pushes)
            if (!GR && tid < k) ir_store(&shf[(size_t)b * k + tid], st.tacc[tid]);
            if (GR && tid < 8) ir_store(&gf.fsh[(size_t)b * 8 + tid], tid < k ? st.tacc[tid] : 0.0);
            if (folded) {
                stamp();
                if (round == 0) load_top_constants();
                if (GR && grp >= 0) {
                    // the last of the group's workgroups to arrive sums the shares and solves the group's k x k top
                    // part of both sweeps; the record is complete before this workgroup arrives at the grid barrier
                    ++gph;
                    if (ir_group_arrive(gf.gcnt + grp * 32, gnb * gph)) {
                        group_sums(gf.fsh);
                        if (tid == 0) {
                            // (y in LDS, st.dxt -- rewritten from the group's record after the barrier: a private array
                            // indexed by a loop variable lives in scratch memory)
                            double nt = 0.0, nbt = 0.0;
                            for (int i = 0; i < k; ++i) {
                                const double rhs_i = round == 0 ? st.btop[i] : ir_load(&grec[(par ^ 1) * 32 + 16 + i]);
                                double sacc = rhs_i - st.tacc[i];
                                for (int j = 0; j < i; ++j) sacc -= st.ltt[i * 8 + j] * st.dxt[j];
                                st.dxt[i] = sacc;
                            }
                            for (int i = k - 1; i >= 0; --i) {
                                double sacc = st.dxt[i] * st.dinvt[i];
                                for (int j = i + 1; j < k; ++j) sacc -= st.ltt[j * 8 + i] * st.dxt[j];
                                st.dxt[i] = sacc;
                            }
                            // (the candidate x_top + dx_top is formed AFTER the barrier: whether the previous
                            // candidate was accepted is decided there)
                            for (int i = 0; i < k; ++i) {
                                ir_store(&grec[par * 32 + i], st.dxt[i]);
                                nt = nanmax(nt, fabs(st.dxt[i]));
                                nbt = nanmax(nbt, fabs(st.btop[i]));
                            }
                            if (!ir.ir_enable) { // no refinement: the top entries take part in x.is_finite() (:180)
                                ir_store(&grec[par * 32 + 24], nt);
                                ir_store(&grec[par * 32 + 25], nbt);
                            }
                            // the record is complete: release the group's other workgroups
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            __builtin_amdgcn_s_waitcnt(0);
                            __hip_atomic_store(gf.gcnt + grp * 32 + 1, gph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        __syncthreads();
                    } else if (ir_wait_word(gf.gcnt + grp * 32 + 1, gph) == IR_TIMEOUT) {
                        if (tid == 0) ir.res[2] = 1;
                        return;
                    }
                }
                if (GR) {
                    // no grid-wide wait between the sweeps: only the group's.  The verdict on the previous round's
                    // candidate -- every workgroup arrived for it at the end of that round -- is awaited here, with
                    // a forward sweep and the group's top solve between arrival and wait
                    stamp();
                    if (pending) {
                        if (ir_wait_word(ir.ctl + 32 * (1 + IR_NSUB + (int)(blockIdx.x % IR_NSUB)), st.gen) == IR_TIMEOUT) {
                            if (tid == 0) ir.res[2] = 1;
                            return;
                        }
                        decide(round - 1, par ^ 1);
                        pending = false;
                        if (__builtin_amdgcn_readfirstlane(st.done)) {
                            stop = true; // (this round's forward sweep was speculative)
                            break;
                        }
                    }
                    if (tid < 8 && grp >= 0) {
                        const double y = tid < k ? ir_load(&grec[par * 32 + tid]) : 0.0;
                        st.dxt[tid] = y;
                        st.candt[tid] = round == 0 ? y : 1.0 * st.curt[tid] + 1.0 * y;
                    }
                } else {
                if (tid == 0) st.gen += 1;
                const int state = ir_arrive_wait(ir.ctl, st.gen, G);
                if (state == IR_TIMEOUT) {
                    if (tid == 0) ir.res[2] = 1;
                    return;
                }
                stamp();
                if (state == IR_LAST) {
                    if (!GR) reduce_forward(par);
                    if (pending) reduce_residual(par ^ 1, round == 1);
                    ir_release(ir.ctl, st.gen, G);
                }
                __syncthreads();
                if (pending) { // the verdict on the previous round's candidate
                    decide(round - 1, par ^ 1);
                    pending = false;
                    if (__builtin_amdgcn_readfirstlane(st.done)) {
                        stop = true; // (this round's forward sweep was speculative)
                        break;
                    }
                }
                if (tid == 0) {
                    // the k x k top part of both sweeps, by every workgroup alike (k <= 8)
                    // (y in LDS, st.dxt: a private array indexed by a loop variable lives in scratch memory, and its
                    // reloads sat on the critical path right behind the barrier)
                    for (int i = 0; i < k; ++i) {
                        double sacc = (round == 0 ? st.btop[i] : st.rtop[i]) - ir_load(&pub[par * 32 + i]);
                        for (int j = 0; j < i; ++j) sacc -= st.ltt[i * 8 + j] * st.dxt[j];
                        st.dxt[i] = sacc;
                    }
                    for (int i = k - 1; i >= 0; --i) {
                        double sacc = st.dxt[i] * st.dinvt[i];
                        for (int j = i + 1; j < k; ++j) sacc -= st.ltt[j * 8 + i] * st.dxt[j];
                        st.dxt[i] = sacc;
                    }
                    for (int i = 0; i < k; ++i)
                        st.candt[i] = round == 0 ? st.dxt[i] : 1.0 * st.curt[i] + 1.0 * st.dxt[i];
                }
                } // (!GR)
            }
            __syncthreads();
            stamp();
            // D^-1 of the backward sweep (qdldl.rs:737-752): through the sweep's pipeline with 128 registers; as a
            // pass of its own in the 80-register variant (three more pipeline registers per column spill there)
            if (flat) {
                bundle_sweep_flat<false, TW>(v, bv, b, xs, st.dxt, nullptr, k, lev_e);
            } else {
                if (TW != 256) {
                    for (int i = tid; i < nloc; i += TW) xs[i] *= v.Dinv[s0 + i];
                    __syncthreads();
                }
                bundle_sweep_cols<false, IR_SH_BWD, IR_RPT, TW>(v, bv, b, xs, st.dxt, nullptr, k, fat, nfat,
                                                                TW == 256 ? v.Dinv : nullptr);
            }
            stamp();
            {
                // the candidate: x (round 0) or x + dx (directldlkktsolver.rs:300 axpby(1, x, 1))
                const int sel = __builtin_amdgcn_readfirstlane(st.sel);
                const double *cur = sel ? ir.xb : ir.xa;
                double *alt = sel ? ir.xa : ir.xb;
                if (round == 0) {
                    for (int i = tid; i < nloc; i += TW) alt[s0 + i] = xs[i];
                } else {
                    for (int i = tid; i < nloc; i += TW) alt[s0 + i] = 1.0 * cur[s0 + i] + 1.0 * xs[i];
                }
                if (!ir.ir_enable) { // no refinement: only x.is_finite() is asked for (:180)
                    double mx = 0.0;
                    bool nan = false;
                    for (int i = tid; i < nloc; i += TW) {
                        const double val = xs[i];
                        if (val != val) nan = true;
                        else mx = fmax(mx, fabs(val));
                    }
                    mx = block_max(mx, red);
                    const bool anynan = __syncthreads_or(nan);
                    if (tid == 0) ir_store(&pn[(size_t)par * nb + b], anynan ? __longlong_as_double(0x7ff8000000000000ll) : mx);
                    continue;
                }
                __syncthreads(); // the candidate's slice is visible workgroup-wide
                stamp();
                double *share_out = GR ? &gf.rsh[((size_t)par * nb + b) * 8] : &shs[(size_t)par * nb * k + (size_t)b * k];
                if (single && bv.symv_split && flat)
                    bundle_symv_flat<TW>(v, bv, alt, ir.bp, ir.ebuf, xs, red, k, b, st.candt, &pn[(size_t)par * nb + b], share_out);
                else if (single && bv.symv_split)
                    bundle_symv_split<IR_SH_SYMV, TW, 2>(bv, v.Up, v.Ucol16, v.Ux, alt, ir.bp, ir.ebuf, xs, red, k, b, st.candt,
                                                         &pn[(size_t)par * nb + b], share_out);
                else
                    bundle_symv_body<true, IR_SH_SYMV, TW, 2>(bv, v.Up, (const int *)v.Ucol16, v.Ux, alt, ir.bp, single ? nullptr : ir.ebuf, nullptr, nullptr,
                                       xs, red, lfold, b, st.candt, &pn[(size_t)par * nb + b], share_out);
                if (GR && grp >= 0) {
                    // the top rows of the candidate's residual (and their norm): by the group's last arriver
                    ++gph;
                    if (ir_group_arrive(gf.gcnt + grp * 32, gnb * gph)) {
                        group_sums(gf.rsh + (size_t)par * nb * 8);
                        if (tid == 0) {
                            double m = 0.0, mb = 0.0;
                            for (int i = 0; i < k; ++i) {
                                double sacc = st.tacc[i];
                                for (int c = 0; c < k; ++c) sacc += st.ktt[i * 8 + c] * st.candt[c];
                                const double rt = st.btop[i] - sacc;
                                ir_store(&grec[par * 32 + 16 + i], rt);
                                m = nanmax(m, fabs(rt));
                                mb = nanmax(mb, fabs(st.btop[i]));
                            }
                            ir_store(&grec[par * 32 + 24], m);
                            ir_store(&grec[par * 32 + 25], mb);
                        }
                    }
                }
            }
        }
        if (stop) break;
        pending = true;
        // A further round is possible only with refinement on, rounds left, and (a forest without top, or
        // several bundles per workgroup) ... the verdict then rides on that round's barrier; otherwise the
        // round ends with a barrier of its own.
        const bool more_possible = ir.ir_enable && round < ir.maxiter;
        if (GR && more_possible) {
            // arrival for the verdict on this round's candidate; it is awaited in the middle of the next round
            if (tid == 0) st.gen += 1;
            if (ir_arrive_nowait(ir.ctl, st.gen, G) == IR_LAST) {
                reduce_residual(par, round == 0);
                ir_release(ir.ctl, st.gen, G);
            }
            continue;
        }
        if (folded && more_possible) continue;
        stamp();
        if (tid == 0) st.gen += 1;
        const int state = ir_arrive_wait(ir.ctl, st.gen, G);
        if (state == IR_TIMEOUT) {
            if (tid == 0) ir.res[2] = 1;
            return;
        }
        stamp();
        if (state == IR_LAST) {
            reduce_residual(par, round == 0);
            ir_release(ir.ctl, st.gen, G);
        }
        __syncthreads();
        decide(round, par);
        pending = false;
        if (__builtin_amdgcn_readfirstlane(st.done)) break;
    }
    stamp();
    // ---- getlhs (directldlkktsolver.rs:205-215): the accepted x, un-permuted ----
    const int ok = __builtin_amdgcn_readfirstlane(st.ok);
    if (ok) {
        double *cur = __builtin_amdgcn_readfirstlane(st.sel) ? ir.xb : ir.xa;
        auto put = [&](int j, double val) {
            const int o = ir.perm[j];
            if (o < ir.n) {
                if (ir.lhsx) ir.lhsx[o] = val;
            } else if (o < ir.n + ir.m) {
                if (ir.lhsz) ir.lhsz[o - ir.n] = val;
            }
        };
        for (int b = blockIdx.x; b < nb; b += G) {
            const int s0 = bv.bundle_ptr[b], nloc = bv.bundle_ptr[b + 1] - s0;
            const int nruns = ir.runs ? ir.run_ptr[b + 1] - ir.run_ptr[b] : 0;
            if (nruns > 0 && nruns <= IR_MAXRUNS) {
                __syncthreads();
                if (tid < 3 * nruns) st.runs[tid] = ir.runs[3 * ir.run_ptr[b] + tid];
                __syncthreads();
                int r = 0;
                for (int i = tid; i < nloc; i += TW) {
                    while (i >= st.runs[3 * r] + st.runs[3 * r + 2]) ++r;
                    const int o = st.runs[3 * r + 1] + (i - st.runs[3 * r]);
                    const double val = cur[s0 + i];
                    if (o < ir.n) {
                        if (ir.lhsx) ir.lhsx[o] = val;
                    } else if (o < ir.n + ir.m) {
                        if (ir.lhsz) ir.lhsz[o - ir.n] = val;
                    }
                }
            } else {
                for (int i = tid; i < nloc; i += TW) put(s0 + i, cur[s0 + i]);
            }
        }
        if ((GR ? gfirst : blockIdx.x == 0) && tid < k) {
            put(topnode(tid), st.curt[tid]);
            cur[topnode(tid)] = st.curt[tid];
        }
    }
    // (every workgroup is past the last barrier that follows a group phase: the group's counter is free again)
    if (GR && gfirst && tid == 0) {
        __hip_atomic_store(gf.gcnt + grp * 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gf.gcnt + grp * 32 + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x == 0 && tid == 0) {
        ir.res[0] = ok ? 1 : -1; // (0 = the kernel never got here)
        ir.res[1] = st.rounds;
        ir.res[3] = st.sel;
        pub[64] = st.normb;
        pub[65] = st.norme;
    }
    stamp();
    ir_grid_exit(ir.ctl, __builtin_amdgcn_readfirstlane(st.gen) + 1, G);
}


// ---------------------------------------------------------------------------
// k_bundle_irs: the fused solve for the shape the bench workload has (round 6) -- EVERY workgroup owns exactly one
// bundle (grid == nb), every bundle takes the entry-parallel sweeps and the split residual, its slice of the
// permutation is a few runs, at most IRS_NPT nodes per thread.  Same phases, barriers and decisions as k_bundle_ir
// (directldlkktsolver.rs:168-189, :266-321); what differs is where the VECTORS live.  k_bundle_ir moved 863 MB per
// launch on config 3, a third of it vectors: the permuted right-hand side written to bp and read back by both
// residuals, every candidate written to xa / xb and gathered back by its residual, the leaves' residual spilled and
// re-read, the accepted iterate read once more for getlhs.  Here a thread KEEPS its IRS_NPT entries of the candidate
// in registers from the end of the backward sweep on (c[]): the residual takes x of the thread's own rows from there
// and stages the non-leaf entries into LDS from there; the leaves' residual returns through the same registers; the
// right-hand side is read through the runs from the caller's vectors each time (never written: bp is the HOST's
// business now, capi.cpp: chip_kkt::bp_stale); a candidate goes to HBM only when a further round can follow (it is
// that round's accepted iterate), and the result is written from the registers when the last candidate is the accepted
// one.  Config 3, one solve + one round: 24 (b) + 2 x 24 (b again) + 24 + 24 (round 0's x out and back) + 24 (result)
// MB of vectors instead of ~310.
// ---------------------------------------------------------------------------
constexpr int IRS_NPT = 12;
// The thread id behind an empty asm: whatever is computed from it cannot be hoisted out of the loop over the refinement
// rounds.  Without it the compiler moves every phase's address arithmetic (a dozen arrays x tid) in front of that loop,
// where it stays live across ALL phases: the 128-register budget then spills inside the loop (measured: 85 spilled
// registers, the phases 10 - 20 % slower), although no phase needs more than 76 registers by itself.
__device__ __forceinline__ int opaque_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// original index of bundle-local node i through the runs (LDS); successive calls with ascending i: one monotone walk
struct RunWalk {
    const int *runs;
    int r;
    __device__ __forceinline__ int orig(int i) {
        while (i >= runs[3 * r] + runs[3 * r + 2]) ++r; // (runs ascend in the local index)
        return runs[3 * r + 1] + (i - runs[3 * r]);
    }
};

// e = b - K c for the rows of the workgroup's bundle, c = the candidate in the threads' registers; split LDS layout of
// bundle_symv_split (e of the non-leaf rows at xs[nleaf .. nloc), x of the non-leaf nodes in the space around it).
// keep_e: a further round can follow -- the leaves' residual is put into xs[0 .. nleaf) at the end, so xs holds the whole
// residual; otherwise only the norms leave.
template <int TW, int NPT, int NLP>
__device__ __forceinline__ void irs_symv(const LdlView &v, const IrView &ir, const int *runs, const double (&c)[NPT],
                                         bool keep_e, double *xs, double *red, double *tacc3, int k, int s0,
                                         int nloc, int nleaf, const double *xt, double *out_norm, double *out_share) {
    double cl[NLP]; // the candidate at the thread's leaf nodes, then (keep_e) their residual
    const int *__restrict__ Up = v.Up;
    const unsigned short *__restrict__ Ucol16 = v.Ucol16, *__restrict__ Urow16 = v.Urow16;
    const double *__restrict__ Ux = v.Ux;
    const int tid = opaque_tid();
    auto xpos = [&](int t) { return t < nleaf ? t : nloc + (t - nleaf); }; // x of non-leaf t
    auto rhs_of = [&](int o) { return o < ir.n ? ir.rx[o] : (o < ir.n + ir.m ? ir.rz[o - ir.n] : 0.0); };
    constexpr int LR = 4, LS = 3;
    static_assert(NLP % LR == 0 && NLP <= NPT, "leaf passes");
    int tb[LR], te[LR];
#pragma unroll
    for (int u = 0; u < LR; ++u) {
        const int i = tid + u * TW;
        tb[u] = i < nleaf ? Up[s0 + i] : 0;
        te[u] = i < nleaf ? Up[s0 + i + 1] : 0;
    }
    const int fb = Up[s0 + nleaf], fe = Up[s0 + nloc]; // the flat range: rows of the non-leaf nodes
    {
        // the non-leaf rows' b and x into LDS (b through the runs from the caller's vectors)
        RunWalk rw{runs, 0};
        double bq[NPT];
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const int i = tid + u * TW;
            bq[u] = (i >= nleaf && i < nloc) ? rhs_of(rw.orig(i)) : 0.0;
        }
        lds_barrier(); // (every thread has taken its candidate entries out of xs)
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const int i = tid + u * TW;
            if (i >= nleaf && i < nloc) {
                xs[xpos(i - nleaf)] = c[u];
                xs[i] = bq[u];
            }
        }
#pragma unroll
        for (int u = 0; u < NLP; ++u) cl[u] = c[u];
    }
    double tpart = 0.0;
    if (k > 1 && tid < 8) tacc3[tid] = 0.0;
    double mleaf = 0.0;
    bool nan = false;
    lds_barrier();
    // ---- leaf rows: the thread's own, four per pass; x_i from the registers ----
    RunWalk rl{runs, 0};
#pragma unroll
    for (int p = 0; p < NLP / LR; ++p) {
        if (p * LR * TW < nleaf) {
            int jj[LR][LS];
            double vv[LR][LS], bi[LR], acc[LR];
#pragma unroll
            for (int u = 0; u < LR; ++u) {
                const int i = (p * LR + u) * TW + tid;
                bi[u] = i < nleaf ? rhs_of(rl.orig(i)) : 0.0;
                acc[u] = 0.0;
#pragma unroll
                for (int q = 0; q < LS; ++q) {
                    const int t = tb[u] + q;
                    const bool ok = t < te[u];
                    jj[u][q] = ok ? (int)Ucol16[t] : -1;
                    vv[u][q] = ok ? Ux[t] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < LR; ++u) {
                const int i = (p * LR + u) * TW + tid;
                const double xi = cl[p * LR + u];
                auto apply = [&](int j, double val) {
                    if (j >= nloc) {
                        acc[u] += val * xt[j - nloc];
                        if (k == 1) tpart += val * xi;
                        else atomicAdd(&tacc3[j - nloc], val * xi);
                    } else if (j == i) {
                        acc[u] += val * xi;
                    } else {
                        acc[u] += val * xs[xpos(j - nleaf)];
                        atomicAdd(&xs[j], -(val * xi));
                    }
                };
#pragma unroll
                for (int q = 0; q < LS; ++q)
                    if (jj[u][q] >= 0) apply(jj[u][q], vv[u][q]);
                for (int t = tb[u] + LS; t < te[u]; ++t) apply((int)Ucol16[t], Ux[t]); // (a leaf with a long row: rare)
                if (i < nleaf) {
                    const double val = bi[u] - acc[u];
                    if (keep_e) cl[p * LR + u] = val;
                    if (val != val) nan = true;
                    else mleaf = fmax(mleaf, fabs(val));
                }
                const int in = i + LR * TW;
                tb[u] = in < nleaf ? Up[s0 + in] : 0;
                te[u] = in < nleaf ? Up[s0 + in + 1] : 0;
            }
        }
    }
    // ---- rows of the non-leaf nodes: flat over their entries, the next batch in flight while this one is consumed ----
    {
        int ii[FLAT_U], jj[FLAT_U], ni[FLAT_U], nj[FLAT_U];
        double vv[FLAT_U], nv[FLAT_U];
        auto request = [&](int bs, int *pi, int *pj, double *pv) {
#pragma unroll
            for (int u = 0; u < FLAT_U; ++u) {
                const int t = bs + u * TW + tid;
                const bool ok = t < fe;
                pi[u] = ok ? (int)Urow16[t] : -1;
                pj[u] = ok ? (int)Ucol16[t] : 0;
                pv[u] = ok ? Ux[t] : 0.0;
            }
        };
        if (fb < fe) request(fb, ii, jj, vv);
        for (int base = fb; base < fe; base += TW * FLAT_U) {
            if (base + TW * FLAT_U < fe) request(base + TW * FLAT_U, ni, nj, nv);
            else {
#pragma unroll
                for (int u = 0; u < FLAT_U; ++u) ni[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < FLAT_U; ++u) {
                const int i = ii[u], j = jj[u];
                if (i < 0) continue;
                const double xi = xs[xpos(i - nleaf)];
                if (j >= nloc) {
                    atomicAdd(&xs[i], -(vv[u] * xt[j - nloc]));
                    if (k == 1) tpart += vv[u] * xi;
                    else atomicAdd(&tacc3[j - nloc], vv[u] * xi);
                } else if (j == i) {
                    atomicAdd(&xs[i], -(vv[u] * xi));
                } else {
                    atomicAdd(&xs[i], -(vv[u] * xs[xpos(j - nleaf)]));
                    atomicAdd(&xs[j], -(vv[u] * xi));
                }
            }
#pragma unroll
            for (int u = 0; u < FLAT_U; ++u) {
                ii[u] = ni[u];
                jj[u] = nj[u];
                vv[u] = nv[u];
            }
        }
    }
    lds_barrier();
    double m = mleaf;
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const int i = tid + u * TW;
        if (i >= nleaf && i < nloc) {
            const double val = xs[i];
            if (val != val) nan = true;
            else m = fmax(m, fabs(val));
        }
    }
    if (keep_e) { // (the x of the non-leaf nodes that lived in xs[0 .. nleaf) has been consumed)
#pragma unroll
        for (int u = 0; u < NLP; ++u)
            if (tid + u * TW < nleaf) xs[tid + u * TW] = cl[u];
    }
    m = block_max(m, red);
    const bool anynan = __syncthreads_or(nan);
    if (tid == 0)
        __hip_atomic_store(out_norm, anynan ? __longlong_as_double(0x7ff8000000000000ll) : m, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (k == 1) {
        tpart = block_sum(tpart, red);
        if (tid == 0) __hip_atomic_store(out_share, tpart, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (k > 1) {
        lds_barrier();
        if (tid == 0)
            for (int i = 0; i < k; ++i)
                __hip_atomic_store(out_share + i, tacc3[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int TW, int NLP>
__global__ __launch_bounds__(TW) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_bundle_irs(LdlView v, BundleView bv, FoldView fold, IrView ir) {
    constexpr int NPT = IRS_NPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *xs = (double *)smem;
    __shared__ double red[16];
    __shared__ int lev_e[FLAT_MAXLEV + 1];
    __shared__ double tacc3[8];
    __shared__ IrState st;
    const int nb = bv.nb, G = gridDim.x, tid = threadIdx.x, b = blockIdx.x;
    const int k = fold.k;
    const bool folded = k > 0; // a barrier in the middle of every round
    const int NF = k ? fold.NF : ir.N;
    if (ir.test_drop && b == G - 1 && G > 1) return; // (tests: a launch that is not co-resident)
    const int s0 = bv.bundle_ptr[b], nloc = bv.bundle_ptr[b + 1] - s0;
    const int nleaf = bv.blvl[bv.blvl_ptr[b] + 1] - s0;
    // partial results: device-coherent stores before a barrier, reduced in a fixed order by its last arriver
    double *pnb = ir.part;                  // [nb]       ||b||inf of the bundles' rows
    double *pn = pnb + nb;                  // [2][nb]    ||e||inf of the bundles' rows
    double *shf = pn + 2 * nb;              // [nb*k]     forward sweep: shares of the top rows
    double *shs = shf + (size_t)nb * k;     // [2][nb*k]  residual: shares of (K x)[top]
    double *pub = shs + 2 * (size_t)nb * k; // [2][32]    published reductions
    auto rhs_of = [&](int o) { return o < ir.n ? ir.rx[o] : (o < ir.n + ir.m ? ir.rz[o - ir.n] : 0.0); };
    if (tid == 0) {
        st.normb = st.norme = st.lastnorme = 0.0;
        st.rounds = 0;
        st.ok = 1;
        st.done = 0;
        st.sel = 0;
        st.par = 0;
        st.gen = 0;
        st.pad = 0; // (1: the last verdict accepted its candidate)
        if (b == 0) {
            ir.res[0] = 0; // "did not finish" until the verdict is written at the very end
            ir.res[2] = 0;
        }
    }
    if (tid < 64) {
        st.ltt[tid] = 0.0;
        st.ktt[tid] = 0.0;
    }
    {
        const int r0 = ir.run_ptr[b], nruns = ir.run_ptr[b + 1] - r0;
        if (tid < 3 * nruns) st.runs[tid] = ir.runs[3 * r0 + tid];
    }
    flat_level_table(v, bv, b, lev_e);
    __syncthreads();
    auto load_top_constants = [&]() {
        if (tid < 8) {
            st.btop[tid] = tid < k ? rhs_of(ir.perm[NF + tid]) : 0.0;
            if (ir.bp && tid < k && b == 0) ir.bp[NF + tid] = st.btop[tid];
            st.curt[tid] = 0.0;
            st.dinvt[tid] = tid < k ? v.Dinv[NF + tid] : 0.0;
        } else if (tid >= 64 && tid < 64 + k * k) {
            const int ti = (tid - 64) / k, tj = (tid - 64) % k;
            const int q = fold.tt[tid - 64];
            if (q >= 0) st.ltt[ti * 8 + tj] = v.Lx[q];
        } else if (tid >= 128 && tid < 128 + k) {
            const int i = tid - 128;
            for (int t = fold.sp[i]; t < fold.sp[i + 1]; ++t) st.ktt[i * 8 + fold.scol[t]] += v.Ux[fold.sslot[t]];
        }
    };
    if (k == 0) load_top_constants(); // (a forest: only btop / curt are cleared)
    int dbgn = 0;
    auto stamp = [&]() { // diagnostics (CHIP_IR_DEBUG): phase boundaries on the 100 MHz clock
        if (ir.dbg_all) {
            if (tid == 0 && dbgn < 31) {
                if (dbgn == 0)
                    ir.dbg_all[(size_t)b * 32] = (long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4) |
                                                 ((long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 20) << 32);
                ir.dbg_all[(size_t)b * 32 + 1 + dbgn++] = wall_clock64();
            }
        } else if (ir.dbg && tid == 0 && (b == 0 || b == G / 2) && dbgn < 64)
            ir.dbg[(b ? 64 : 0) + dbgn++] = wall_clock64();
    };
    __shared__ double redm[4 * 16];
    // The grid barrier with its reductions.  Every workgroup has stored its partial results (forward shares shf, norms
    // pn / pnb, residual shares shs) with device-coherent stores; the LAST arriver sums them in a fixed order -- all
    // loads in flight at once, ONE pass of workgroup reductions for the four numbers of the usual single top row -- and
    // publishes the results inside the release messages (grid_sync.hpp: ir_publish); everyone else finds them in its
    // sub-group's record.  Afterwards st.pubv holds them in every workgroup.
    //   fwd: this round's forward sums;  resid: norms / residual sums of the candidate whose partials have parity rpar
    const int nmsg = 2 * k + 2;
    const int tagbase = ir.epoch << 8;
    auto barrier = [&](bool fwd, bool resid, int rpar, bool first) -> int {
        if (tid == 0) st.gen += 1;
        const int state = ir_arrive_nowait(ir.ctl, st.gen, G);
        const int tag = tagbase | (__builtin_amdgcn_readfirstlane(st.gen) & 0xff);
        if (state == IR_LAST) {
            double f0 = 0.0, r0 = 0.0, mb = 0.0, m = 0.0;
            for (int q = tid; q < nb; q += TW) {
                if (fwd && k >= 1) f0 += ir_load(&shf[(size_t)q * k]);
                if (resid) {
                    if (first) mb = nanmax(mb, ir_load(&pnb[q]));
                    m = nanmax(m, ir_load(&pn[(size_t)rpar * nb + q]));
                    if (ir.ir_enable && k >= 1) r0 += ir_load(&shs[(size_t)rpar * nb * k + (size_t)q * k]);
                }
            }
            f0 = wave_sum(f0);
            r0 = wave_sum(r0);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mb = nanmax(mb, __shfl_down(mb, o, 64));
                m = nanmax(m, __shfl_down(m, o, 64));
            }
            const int lane = tid & 63, wv = tid >> 6;
            if (lane == 0) {
                redm[wv * 4 + 0] = f0;
                redm[wv * 4 + 1] = r0;
                redm[wv * 4 + 2] = mb;
                redm[wv * 4 + 3] = m;
            }
            __syncthreads();
            if (tid == 0) {
                double tf = redm[0], tr = redm[1], tmb = redm[2], tm = redm[3];
                for (int w = 1; w < TW / 64; ++w) {
                    tf += redm[w * 4 + 0];
                    tr += redm[w * 4 + 1];
                    tmb = nanmax(tmb, redm[w * 4 + 2]);
                    tm = nanmax(tm, redm[w * 4 + 3]);
                }
                if (k >= 1) {
                    st.pubv[0] = tf;
                    st.pubv[k + 2] = tr;
                }
                st.pubv[k] = tm;
                if (first || !resid) st.pubv[k + 1] = tmb;
            }
            for (int i = 1; i < k; ++i) { // (further top rows: a pass each)
                double pf = 0.0, pr = 0.0;
                for (int q = tid; q < nb; q += TW) {
                    if (fwd) pf += ir_load(&shf[(size_t)q * k + i]);
                    if (resid && ir.ir_enable) pr += ir_load(&shs[(size_t)rpar * nb * k + (size_t)q * k + i]);
                }
                pf = block_sum(pf, red);
                pr = block_sum(pr, red);
                if (tid == 0) {
                    st.pubv[i] = pf;
                    st.pubv[k + 2 + i] = pr;
                }
            }
            __syncthreads();
            ir_publish(ir.rel, st.pubv, nmsg, tag, G);
            return IR_LAST;
        }
        return ir_wait_record(ir.rel, st.pubv, nmsg, tag);
    };
    // the reference's decisions about the candidate of round `round` (k_bundle_ir: decide)
    auto decide = [&](int round, int par) {
        if (tid == 0) {
            double m = st.pubv[k];
            if (round == 0) {
                double nbm = st.pubv[k + 1];
                for (int i = 0; i < k; ++i) nbm = nanmax(nbm, fabs(st.btop[i]));
                st.normb = nbm;
            }
            for (int i = 0; i < k; ++i) {
                double sacc = st.pubv[k + 2 + i];
                for (int cc = 0; cc < k; ++cc) sacc += st.ktt[i * 8 + cc] * st.candt[cc];
                st.rtop[i] = ir.ir_enable ? st.btop[i] - sacc : st.candt[i];
                m = nanmax(m, fabs(st.rtop[i]));
            }
            const double newnorm = m, tol = ir.abstol + ir.reltol * st.normb;
            bool accept, done = false;
            if (round == 0) {
                accept = true;
                st.norme = newnorm;
                if (!(newnorm - newnorm == 0.0)) { // non-finite (:284-286; without refinement: x.is_finite(), :180)
                    st.ok = 0;
                    done = true;
                } else if (!ir.ir_enable || ir.maxiter <= 0 || newnorm <= tol) {
                    done = true;
                }
            } else {
                st.rounds += 1;
                if (!(newnorm - newnorm == 0.0)) { // :305-307
                    st.ok = 0;
                    accept = false;
                    done = true;
                } else {
                    const double improved = st.lastnorme / newnorm;
                    accept = !(improved < ir.stopratio) || improved > 1.0; // :309-318
                    if (improved < ir.stopratio) done = true;
                    if (accept) st.norme = newnorm;
                }
            }
            st.pad = accept ? 1 : 0;
            if (accept) {
                st.sel ^= 1;
                for (int i = 0; i < k; ++i) st.curt[i] = st.candt[i];
            }
            if (!done && (st.rounds >= ir.maxiter || st.norme <= tol)) done = true; // :288-293
            st.lastnorme = st.norme;
            st.done = done ? 1 : 0;
        }
        __syncthreads();
    };
    stamp();
    // ---- setrhs (directldlkktsolver.rs:160-166): the bundle's slice of the permuted right-hand side into LDS ----
    {
        double c[NPT];
        RunWalk rw{st.runs, 0};
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const int i = tid + u * TW;
            c[u] = i < nloc ? rhs_of(rw.orig(i)) : 0.0;
        }
        double mx = 0.0;
        bool nan = false;
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const int i = tid + u * TW;
            if (i < nloc) {
                xs[i] = c[u];
                if (ir.bp) ir.bp[s0 + i] = c[u]; // (only when the host asks for the permuted copy)
                if (c[u] != c[u]) nan = true;
                else mx = fmax(mx, fabs(c[u]));
            }
        }
        mx = block_max(mx, red);
        const bool anynan = __syncthreads_or(nan);
        if (tid == 0) ir_store(&pnb[b], anynan ? __longlong_as_double(0x7ff8000000000000ll) : mx);
    }
    bool pending = false;   // a candidate whose residual partials have been stored but not yet reduced
    // Where the iterates live.  Round 0's candidate x0 stays in the threads' REGISTERS (xk) when a round can follow: it
    // is that round's accepted iterate (round 0 is always accepted), x1 = x0 + dx is formed from there.  The candidate of
    // the LAST possible round is written straight to the caller's lhs vectors (spec_out: they do not overlap the
    // right-hand side, which the residual still reads) -- if the verdict then rejects it, the accepted iterate is
    // written over it.  Only candidates that further rounds may build on (round >= 1 with rounds left) go to xa / xb.
    double xk[NPT];
    bool xk_valid = false;  // x0 lives in xk only
    bool spec_done = false; // the last candidate has been written to lhs
    for (int round = 0;; ++round) {
        const int par = round & 1;
        __syncthreads();
        stamp();
        bundle_sweep_flat<true, TW>(v, bv, b, xs, nullptr, st.tacc, k, lev_e);
        stamp();
        if (tid < k) ir_store(&shf[(size_t)b * k + tid], st.tacc[tid]);
        if (folded) {
            stamp();
            if (round == 0) load_top_constants();
            if (barrier(true, pending, par ^ 1, round == 1) == IR_TIMEOUT) {
                if (tid == 0) ir.res[2] = 1;
                return;
            }
            stamp();
            if (pending) { // the verdict on the previous round's candidate
                decide(round - 1, par ^ 1);
                pending = false;
                if (__builtin_amdgcn_readfirstlane(st.done)) break; // (this round's forward sweep was speculative)
            }
            if (tid == 0) {
                // the k x k top part of both sweeps, by every workgroup alike (k <= 8)
                // (y in LDS, st.dxt: a private array indexed by a loop variable lives in scratch memory, and its
                // reloads sat on the critical path right behind the barrier)
                for (int i = 0; i < k; ++i) {
                    double sacc = (round == 0 ? st.btop[i] : st.rtop[i]) - st.pubv[i];
                    for (int j = 0; j < i; ++j) sacc -= st.ltt[i * 8 + j] * st.dxt[j];
                    st.dxt[i] = sacc;
                }
                for (int i = k - 1; i >= 0; --i) {
                    double sacc = st.dxt[i] * st.dinvt[i];
                    for (int j = i + 1; j < k; ++j) sacc -= st.ltt[j * 8 + i] * st.dxt[j];
                    st.dxt[i] = sacc;
                }
                for (int i = 0; i < k; ++i)
                    st.candt[i] = round == 0 ? st.dxt[i] : 1.0 * st.curt[i] + 1.0 * st.dxt[i];
            }
        }
        __syncthreads();
        stamp();
        bundle_sweep_flat<false, TW>(v, bv, b, xs, st.dxt, nullptr, k, lev_e);
        stamp();
        // the candidate: x (round 0) or x + dx (directldlkktsolver.rs:300 axpby(1, x, 1)), into the registers
        const bool more_possible = ir.ir_enable && round < ir.maxiter;
        double c[NPT];
        {
            const int tid = opaque_tid();
            const int sel = __builtin_amdgcn_readfirstlane(st.sel);
            double *cur = sel ? ir.xb : ir.xa;
            double *alt = sel ? ir.xa : ir.xb;
            if (round == 0) {
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    const int i = tid + u * TW;
                    c[u] = i < nloc ? xs[i] : 0.0;
                }
            } else {
                if (!xk_valid) {
#pragma unroll
                    for (int u = 0; u < NPT; ++u) {
                        const int i = tid + u * TW;
                        xk[u] = i < nloc ? cur[s0 + i] : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    const int i = tid + u * TW;
                    c[u] = i < nloc ? 1.0 * xk[u] + 1.0 * xs[i] : 0.0;
                }
            }
            if (more_possible && round == 0 && (ir.sf_flags & 1)) {
#pragma unroll
                for (int u = 0; u < NPT; ++u) xk[u] = c[u];
                xk_valid = true;
            } else if (more_possible || !ir.spec_out) {
                // (a later round may build on it / the result vectors overlap the right-hand side: through xa / xb)
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    const int i = tid + u * TW;
                    if (i < nloc) {
                        alt[s0 + i] = c[u];
                        if (xk_valid) cur[s0 + i] = xk[u]; // (x0 leaves the registers: later rounds read it from there)
                    }
                }
                xk_valid = false;
            } else {
                RunWalk rw{st.runs, 0};
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    const int i = tid + u * TW;
                    if (i < nloc) {
                        const int o = rw.orig(i);
                        if (o < ir.n) {
                            if (ir.lhsx) ir.lhsx[o] = c[u];
                        } else if (o < ir.n + ir.m) {
                            if (ir.lhsz) ir.lhsz[o - ir.n] = c[u];
                        }
                    }
                }
                spec_done = true;
            }
        }
        if (!ir.ir_enable) { // no refinement: only x.is_finite() is asked for (:180)
            double mx = 0.0;
            bool nan = false;
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                if (tid + u * TW < nloc) {
                    if (c[u] != c[u]) nan = true;
                    else mx = fmax(mx, fabs(c[u]));
                }
            }
            mx = block_max(mx, red);
            const bool anynan = __syncthreads_or(nan);
            if (tid == 0) ir_store(&pn[(size_t)par * nb + b], anynan ? __longlong_as_double(0x7ff8000000000000ll) : mx);
        } else {
            stamp();
            irs_symv<TW, NPT, NLP>(v, ir, st.runs, c, more_possible, xs, red, tacc3, k, s0, nloc, nleaf, st.candt,
                                   &pn[(size_t)par * nb + b], &shs[(size_t)par * nb * k + (size_t)b * k]);
        }
        pending = true;
        if (folded && more_possible) continue; // (the verdict rides on the next round's barrier)
        stamp();
        if (barrier(false, true, par, round == 0) == IR_TIMEOUT) {
            if (tid == 0) ir.res[2] = 1;
            return;
        }
        stamp();
        decide(round, par);
        pending = false;
        if (__builtin_amdgcn_readfirstlane(st.done)) break;
    }
    stamp();
    // ---- getlhs (directldlkktsolver.rs:205-215): the accepted x, un-permuted ----
    const int ok = __builtin_amdgcn_readfirstlane(st.ok);
    if (ok) {
        // the last candidate is on its way already (spec_done) and the verdict accepted it: nothing to do; else the
        // accepted iterate from the registers (x0) or from xa / xb
        const bool accepted_last = __builtin_amdgcn_readfirstlane(st.pad) != 0;
        if (!(spec_done && accepted_last)) {
            const int tid = opaque_tid();
            const double *cur = __builtin_amdgcn_readfirstlane(st.sel) ? ir.xb : ir.xa;
            if (!xk_valid) {
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    const int i = tid + u * TW;
                    xk[u] = i < nloc ? cur[s0 + i] : 0.0;
                }
            }
            RunWalk rw{st.runs, 0};
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const int i = tid + u * TW;
                if (i < nloc) {
                    const int o = rw.orig(i);
                    if (o < ir.n) {
                        if (ir.lhsx) ir.lhsx[o] = xk[u];
                    } else if (o < ir.n + ir.m) {
                        if (ir.lhsz) ir.lhsz[o - ir.n] = xk[u];
                    }
                }
            }
        }
        if (b == 0 && tid < k) {
            const int o = ir.perm[NF + tid];
            if (o < ir.n) {
                if (ir.lhsx) ir.lhsx[o] = st.curt[tid];
            } else if (o < ir.n + ir.m) {
                if (ir.lhsz) ir.lhsz[o - ir.n] = st.curt[tid];
            }
        }
    }
    if (b == 0 && tid == 0) {
        ir.res[0] = ok ? 1 : -1; // (0 = the kernel never got here)
        ir.res[1] = st.rounds;
        ir.res[3] = st.sel;
        pub[64] = st.normb;
        pub[65] = st.norme;
    }
    stamp();
    ir_grid_exit(ir.ctl, __builtin_amdgcn_readfirstlane(st.gen) + 1, G);
}

} // namespace

int ir_ctl_ints() { return IR_CTL_INTS; }
int ir_rel_ints() { return IR_REL_INTS; }
size_t ir_part_doubles(int nb, int k) { return (size_t)nb * (3 + 3 * (size_t)k) + 72; }
// workgroup size of k_bundle_ir for these bundles and the largest co-resident grid (0: the kernel cannot run)
static size_t bundle_ir_lds(const BundleView &bv) {
    return ((size_t)std::max(bv.max_nodes, bv.ir_lds_doubles) * sizeof(double) + 15) & ~(size_t)15;
}
template <int TW> static int bundle_ir_capacity_tw(const BundleView &bv) {
    const size_t lds = bundle_ir_lds(bv);
    if (raise_dynamic_lds((const void *)k_bundle_ir<TW, false>, (size_t)lds) != hipSuccess ||
        raise_dynamic_lds((const void *)k_bundle_ir<TW, true>, (size_t)lds) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    int per_cu_g = 0; // (the grouped variant may differ by a register or two: the smaller count holds for both)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_bundle_ir<TW, false>, TW, lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_g, (const void *)k_bundle_ir<TW, true>, TW, lds) != hipSuccess ||
        hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    per_cu = std::min(per_cu, per_cu_g);
    // cross-check with the LDS budget (static + dynamic, 1 KB allocation granularity assumed) and the wave slots
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void *)k_bundle_ir<TW, true>) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    const size_t per_wg = ((fa.sharedSizeBytes + lds + 1023) / 1024) * 1024;
    const int by_lds = (int)(prop.maxSharedMemoryPerMultiProcessor / per_wg);
    const int by_waves = (TW == 512 ? 24 : 16) / (TW / 64); // waves per CU the kernel was compiled for
    per_cu = std::min(per_cu, std::min(by_lds, by_waves));
    return per_cu * prop.multiProcessorCount;
}
int bundle_ir_capacity(const BundleView &bv, int *tw) {
    if (!bv.nb) return 0;
    // four 256-thread workgroups per CU when the LDS slices allow it, else 512-thread workgroups
    const int c256 = bundle_ir_capacity_tw<256>(bv);
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (c256 >= 4 * prop.multiProcessorCount) {
        *tw = 256;
        return c256;
    }
    // few large bundles (a batched problem's share of one GPU of eight): one 1024-thread workgroup per CU --
    // a bundle's sweeps are latency chains, twice the threads take a level's columns in half the passes
    const bool no1024 = false;
    if (!no1024 && bv.nb <= prop.multiProcessorCount) {
        const int c1024 = bundle_ir_capacity_tw<1024>(bv);
        if (c1024 >= bv.nb) {
            *tw = 1024;
            return c1024;
        }
    }
    *tw = 512;
    return bundle_ir_capacity_tw<512>(bv);
}
// k_bundle_irs can take this handle's solves: co-resident grid of nb workgroups of 256 threads (the per-bundle conditions
// -- flat sweeps, nodes per thread, runs of the permutation -- are the caller's to check, see irs_bundle_ok)
bool bundle_irs_capacity_ok(const BundleView &bv) {
    if (!bv.nb || !bv.symv_split) return false;
    const size_t lds = bundle_ir_lds(bv);
    if (raise_dynamic_lds((const void *)k_bundle_irs<256, 4>, lds) != hipSuccess) return false;
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    hipFuncAttributes fa;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_bundle_irs<256, 4>, 256, lds) != hipSuccess ||
        hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipFuncGetAttributes(&fa, (const void *)k_bundle_irs<256, 4>) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    const size_t per_wg = ((fa.sharedSizeBytes + lds + 1023) / 1024) * 1024;
    per_cu = std::min(per_cu, std::min((int)(prop.maxSharedMemoryPerMultiProcessor / per_wg), 4));
    return per_cu * prop.multiProcessorCount >= bv.nb;
}
bool irs_bundle_ok(int nloc, int nleaf, int nlevels, int nruns) {
    return nloc >= FLAT_MIN_NODES && nloc <= IRS_NPT * 256 && nleaf <= 4 * 256 && nlevels <= FLAT_MAXLEV && nruns > 0 &&
           nruns <= IR_MAXRUNS;
}
int bundle_ir(hipStream_t s, const LdlView &v, const BundleView &bv, const FoldView &fold, const IrView &ir, int grid,
              int tw, const GFoldView &gf) {
    // grid <= bundle_ir_capacity(): every workgroup is resident on an otherwise idle device, and a grid
    // barrier that cannot complete times out instead of hanging
    const size_t lds = bundle_ir_lds(bv);
    if (ir.sf) { // (the caller has checked bundle_irs_capacity_ok / irs_bundle_ok)
        k_bundle_irs<256, 4><<<bv.nb, 256, lds, s>>>(v, bv, fold, ir);
        return (int)hipGetLastError();
    }
    if (gf.ng > 0) {
        if (tw == 256) k_bundle_ir<256, true><<<grid, 256, lds, s>>>(v, bv, fold, ir, gf);
        else if (tw == 1024) k_bundle_ir<1024, true><<<grid, 1024, lds, s>>>(v, bv, fold, ir, gf);
        else k_bundle_ir<512, true><<<grid, 512, lds, s>>>(v, bv, fold, ir, gf);
    } else {
        if (tw == 256) k_bundle_ir<256, false><<<grid, 256, lds, s>>>(v, bv, fold, ir, gf);
        else if (tw == 1024) k_bundle_ir<1024, false><<<grid, 1024, lds, s>>>(v, bv, fold, ir, gf);
        else k_bundle_ir<512, false><<<grid, 512, lds, s>>>(v, bv, fold, ir, gf);
    }
    return (int)hipGetLastError();
}

} // namespace dev
} // namespace chip
