// bundle_solve.hip -- substitutions and residual, one kernel per phase: bundle sweeps, the row-gather family, chains, blocked tops, folded tops
// (one of the translation units behind kernels.hpp; the design rules and the reference citations are in
// dev_common.hpp)
#include "dev_common.hpp"
#include "bundle_symv.hpp"

namespace chip {
namespace dev {

constexpr int BSF_MAXLEV = 64; // levels of a bundle the entry-parallel stand-alone sweeps keep entry pointers for

namespace {

// one row (forward: row of L, all inside the bundle; backward: column of L, ancestors inside the
// bundle from LDS, top ancestors -- final before the launch -- from x), strided over `stride`
// threads starting at `first`; xs = the bundle's slice of x in LDS
template <bool FWDMODE>
__device__ __forceinline__ double bundle_row_dot(const LdlView &v, const double *xs, const double *x, int s0,
                                                 int s1, int r, int first, int stride) {
    double s = 0.0;
    if (FWDMODE) {
        for (int t = v.Rp[r] + first; t < v.Rp[r + 1]; t += stride) s += v.Rx[t] * xs[v.Rcol[t] - s0];
    } else {
        for (int q = v.Lp[r] + first; q < v.Lp[r + 1]; q += stride) {
            const int i = v.Li[q];
            s += v.Lx[q] * (i < s1 ? xs[i - s0] : x[i]);
        }
    }
    return s;
}

// forward (rows of L, descendants only -> all inside the bundle) or backward (columns of L)
// sweep of a bundle with its slice of x staged in LDS, one __syncthreads()-separated level at a
// time.  A workgroup's sweep is a chain of dependent global loads per level (row pointers ->
// entries -> gathers), so the number of sequential round trips is what is minimised:
//  * backward: x_j = x_j / d_j - sum_i l_ij x_i (qdldl.rs:737-752).  The scaling by 1/d_j is
//    applied while staging (coalesced, off the per-level path).  (Folding the top-ancestor tail of
//    every column into the staging pass as well was measured and dropped: the tails share cache
//    lines with the in-bundle entries, so L was streamed twice -- 266 MB instead of 160 MB.)
//  * the row pointers of the NEXT level's first sweep are requested before the current level is
//    processed.
//  * thin rows: two rows per thread, FOUR entries of each row per shot -- rows of <= 4 entries
//    (nearly all rows of a block-arrow KKT) cost one round trip instead of one per entry.
constexpr int ESHOT = 4;     // forward: all gathers come from LDS
constexpr int ESHOT_BWD = 2; // backward: entries of top ancestors are gathered from global memory (64-bit addresses)
template <bool FWDMODE>
__device__ __forceinline__ void bundle_solve_body(const LdlView &v, const BundleView &bv, double *x,
                                                  const double *__restrict__ addv, double *xs, double *red,
                                                  int *fat, int &nfat, const FoldView &fold) {
    const int b = blockIdx.x;
    const int s0 = bv.bundle_ptr[b], s1 = bv.bundle_ptr[b + 1], nloc = s1 - s0;
    const int *lv = bv.blvl + bv.blvl_ptr[b];
    const int nl = bv.blvl_ptr[b + 1] - bv.blvl_ptr[b] - 1;
    const int *pbeg = FWDMODE ? v.Rp : v.Lp;          // first slot of a row
    const int *pend = FWDMODE ? v.Rp + 1 : v.Lp + 1;  // one past its last slot
    const int *cidx = FWDMODE ? v.Rcol : v.Li;
    const double *cval = FWDMODE ? v.Rx : v.Lx;
    const int nsteps = FWDMODE ? nl - 1 : nl;         // forward: level 0 has no descendants
    auto level_of = [&](int step) { return FWDMODE ? step + 1 : nl - 1 - step; };
    // row pointers of the first sweep (2 rows per thread) of a level
    int ntb[2] = {0, 0}, nte[2] = {0, 0};
    auto request_ptrs = [&](int step) {
        const int l = level_of(step);
        const int lb = lv[l], le = lv[l + 1];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lb + (int)threadIdx.x + u * BWG;
            ntb[u] = j < le ? pbeg[j] : 0;
            nte[u] = j < le ? pend[j] : 0;
        }
    };
    if (nsteps > 0) request_ptrs(0);
    // folded top row 0: this thread's first entries of the bundle's segment are requested now -- they
    // do not depend on x -- and consumed in the epilogue
    constexpr int FPF = 2;
    int fj[FPF];
    double fv[FPF];
    int ftb = 0, fte = 0;
    if (FWDMODE && fold.k > 0) {
        ftb = fold.rseg[(b * fold.k) * 2];
        fte = fold.rseg[(b * fold.k) * 2 + 1];
#pragma unroll
        for (int q = 0; q < FPF; ++q) {
            const int t = ftb + (int)threadIdx.x + q * BWG;
            fj[q] = t < fte ? v.Rcol[t] : -1;
            fv[q] = t < fte ? v.Rx[t] : 0.0;
        }
    }
    if (FWDMODE) {
        for (int i = threadIdx.x; i < nloc; i += BWG) xs[i] = x[s0 + i];
    } else {
        for (int i = threadIdx.x; i < nloc; i += BWG) xs[i] = x[s0 + i] * v.Dinv[s0 + i];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // one row by the whole workgroup; the result lands in xs[r - s0] (visible after a barrier)
    auto coop_row = [&](int r) {
        double s = bundle_row_dot<FWDMODE>(v, xs, x, s0, s1, r, threadIdx.x, BWG);
        s = block_sum(s, red);
        if (threadIdx.x == 0) xs[r - s0] -= s;
    };
    for (int step = 0; step < nsteps; ++step) {
        const int l = level_of(step);
        const int lb = lv[l], le = lv[l + 1];
        int ftb[2] = {ntb[0], ntb[1]}, fte[2] = {nte[0], nte[1]}; // this level's first sweep
        if (step + 1 < nsteps) request_ptrs(step + 1);            // in flight while this level runs
        __syncthreads(); // the previous level is final in xs; its fat list is no longer read
        if (le - lb == 1) { // a level of its own: no classification pass
            coop_row(lb);
            continue;
        }
        if (threadIdx.x == 0) nfat = 0;
        __syncthreads();
        for (int j0 = lb + threadIdx.x; j0 < le; j0 += 2 * BWG) {
            int jr[2], tb[2], te[2];
            double sum[2];
            const bool first = j0 < lb + BWG;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = j0 + u * BWG;
                jr[u] = j < le ? j : -1;
                tb[u] = first ? ftb[u] : (j < le ? pbeg[j] : 0);
                te[u] = first ? fte[u] : (j < le ? pend[j] : 0);
                sum[u] = 0.0;
                if (te[u] - tb[u] > THIN_MAX) {
                    const int slot = atomicAdd(&nfat, 1);
                    if (slot < FATCAP) {
                        fat[slot] = j;
                        jr[u] = -1; // handled cooperatively below
                        te[u] = tb[u];
                    }
                }
            }
            const int maxlen = max(te[0] - tb[0], te[1] - tb[1]);
            constexpr int SH = FWDMODE ? ESHOT : ESHOT_BWD;
            for (int k = 0; k < maxlen; k += SH) {
                int ii[2][SH];
                double vv[2][SH];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < SH; ++e) {
                        const unsigned t = (unsigned)(tb[u] + k + e);
                        const bool ok = (int)t < te[u];
                        ii[u][e] = ok ? cidx[t] : -1;
                        vv[u][e] = ok ? cval[t] : 0.0;
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < SH; ++e)
                        if (ii[u][e] >= 0)
                            sum[u] += vv[u][e] * ((FWDMODE || ii[u][e] < s1) ? xs[ii[u][e] - s0] : x[ii[u][e]]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (jr[u] >= 0) xs[jr[u] - s0] -= sum[u];
        }
        __syncthreads();
        const int nf = min(nfat, FATCAP);
        if (nf <= 2) {
            // the separators at the top of a subtree: one long row at a time, all 8 waves on it
            for (int f = 0; f < nf; ++f) coop_row(fat[f]);
        } else {
            for (int f = wv; f < nf; f += BWG / 64) {
                const int r = fat[f];
                double s = bundle_row_dot<FWDMODE>(v, xs, x, s0, s1, r, lane, 64);
                s = wave_sum(s);
                if (lane == 0) xs[r - s0] -= s;
            }
        }
    }
    __syncthreads();
    // addv: the refinement step x + dx folded into the final write of the backward sweep
    if (addv)
        for (int i = threadIdx.x; i < nloc; i += BWG) x[s0 + i] = xs[i] + addv[s0 + i];
    else
        for (int i = threadIdx.x; i < nloc; i += BWG) x[s0 + i] = xs[i];
    if (FWDMODE && fold.k > 0) {
        // the few dense top rows (an "arrow"): this bundle's columns of each of them, gathered from
        // the slice that is still in LDS; one global atomic per (bundle, top row) into the slotted
        // accumulators that k_fold_top_solve subtracts from the right-hand side entry x[top]
        for (int i = 0; i < fold.k; ++i) {
            const int tb = fold.rseg[(b * fold.k + i) * 2], te = fold.rseg[(b * fold.k + i) * 2 + 1];
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int t = tb + (int)threadIdx.x;
            if (i == 0) { // prefetched part
#pragma unroll
                for (int q = 0; q < FPF; ++q)
                    if (fj[q] >= 0) a0 += fv[q] * xs[fj[q] - s0];
                t += FPF * BWG;
            }
            for (; t + 3 * BWG < te; t += 4 * BWG) {
                const int j0 = v.Rcol[t], j1 = v.Rcol[t + BWG], j2 = v.Rcol[t + 2 * BWG], j3 = v.Rcol[t + 3 * BWG];
                const double v0 = v.Rx[t], v1 = v.Rx[t + BWG], v2 = v.Rx[t + 2 * BWG], v3 = v.Rx[t + 3 * BWG];
                a0 += v0 * xs[j0 - s0];
                a1 += v1 * xs[j1 - s0];
                a2 += v2 * xs[j2 - s0];
                a3 += v3 * xs[j3 - s0];
            }
            for (; t < te; t += BWG) a0 += v.Rx[t] * xs[v.Rcol[t] - s0];
            const double sum = block_sum((a0 + a1) + (a2 + a3), red);
            if (threadIdx.x == 0 && te > tb) atomicAdd(&fold.acc[fold_acc_index(0, i, b % FOLD_SLOTS)], sum);
        }
    }
}
__global__ __launch_bounds__(BWG) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_bundle_fwd(LdlView v, BundleView bv, double *x, FoldView fold) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[16];
    __shared__ int fat[FATCAP];
    __shared__ int nfat;
    bundle_solve_body<true>(v, bv, x, nullptr, (double *)smem, red, fat, nfat, fold);
}
__global__ __launch_bounds__(BWG) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_bundle_bwd(LdlView v, BundleView bv, double *x, const double *__restrict__ addv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[16];
    __shared__ int fat[FATCAP];
    __shared__ int nfat;
    bundle_solve_body<false>(v, bv, x, addv, (double *)smem, red, fat, nfat, FoldView{});
}


// ---------------------------------------------------------------------------
// row-gather family: forward substitution (rows of L), backward substitution
// fused with D^-1 (columns of L = rows of L'), and the residual e = b - K x.
//   FWD : out[r]  = out[r] - sum val[t] * xin[idx[t]]           (qdldl.rs:708-719)
//   BWD : out[r]  = out[r]*Dinv[r] - sum ...                    (qdldl.rs:737-752)
//   SYMV: out[r]  = b[r] - sum ...                              (directldlkktsolver.rs:334-347)
//   SPMV: out[r]  = aux[r] + alpha * sum ...   (sparse gemv / symv of the IPM residuals and
//                                               RHS algebra, csc/matrix_math.rs:178-343)
// ---------------------------------------------------------------------------
// returns the stored value (SYMV: the residual entry, folded into the inf-norm by the caller)
template <int MODE>
__device__ __forceinline__ double store_row(const GatherArgs &a, int r, double s) {
    double v;
    if (MODE == FWD) v = a.out[r] - s;
    else if (MODE == BWD) v = a.out[r] * a.aux[r] - s;
    else if (MODE == SPMV) v = (a.aux ? a.aux[r] : 0.0) + a.alpha * s;
    else v = a.aux[r] - s;
    a.out[r] = v;
    return v;
}

// ---------------------------------------------------------------------------
// The same two sweeps ENTRY-parallel (round 5), for systems whose top is level-scheduled (BASELINE config 2: 645 bundles
// of ~360 nodes and up to 39 levels): the row- / column-per-thread form above pays one dependent round trip per shot of
// a row's entries on every level (2.7 us per level, 106 us per sweep, twelve sweeps per step).  The entries of a bundle's
// columns are contiguous level by level and do not depend on x: they are walked as ONE stream of batches of BWG * BSF_U
// (row, column, value) triples, the next batch requested before the current one is consumed -- across level boundaries
// too --, every entry one LDS atomic: a level costs its barrier plus the LDS work (as bundle_sweep_flat of k_bundle_ir).
//   forward : x_i -= l_ij x_j for the rows i INSIDE the bundle (the top rows gather from the bundle columns in their own
//             launches, as before); row 0xFFFF = a top row: skipped.
//   backward: the top rows' share  x_j -= l_ij x_i (i in the top: final)  is independent of the bundle's own unknowns and
//             taken by a flat prologue over all entries; the level loop then handles the rows inside the bundle only.
// The sums meet in ds_add_f64 in arrival order (like the fused kernels'); CHIP_DETERMINISTIC / CHIP_NO_BUNDLE_FLAT_SWEEP
// keep the form above.
// ---------------------------------------------------------------------------
constexpr int BSF_U = 4;
template <bool FWDMODE>
__global__ __launch_bounds__(BWG) void k_bundle_sweep_flat(LdlView v, BundleView bv, double *x, const double *__restrict__ addv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *xs = (double *)smem;
    __shared__ int lev_e[BSF_MAXLEV + 1];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int s0 = bv.bundle_ptr[b], nloc = bv.bundle_ptr[b + 1] - s0;
    const int *lv = bv.blvl + bv.blvl_ptr[b];
    const int nl = bv.blvl_ptr[b + 1] - bv.blvl_ptr[b] - 1;
    const unsigned short *__restrict__ Li16 = v.sLi16, *__restrict__ Lj16 = v.sLj16;
    if (tid <= nl) lev_e[tid] = v.Lp[lv[tid]];
    if (FWDMODE) {
        for (int i = tid; i < nloc; i += BWG) xs[i] = x[s0 + i];
    } else {
        for (int i = tid; i < nloc; i += BWG) xs[i] = x[s0 + i] * v.Dinv[s0 + i];
    }
    __syncthreads();
    if (!FWDMODE) {
        // the top rows' share: eight entries per thread in flight, the gathers from x behind them
        const int e0 = lev_e[0], e1 = lev_e[nl];
        for (int base = e0; base < e1; base += 8 * BWG) {
            int gi[8], jj[8];
            double vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = base + u * BWG + tid;
                const bool top = t < e1 && Li16[min(t, e1 - 1)] == 0xFFFFu;
                gi[u] = top ? v.Li[t] : -1;
                jj[u] = top ? (int)Lj16[t] : 0;
                vv[u] = top ? v.Lx[t] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (gi[u] >= 0) atomicAdd(&xs[jj[u]], -(vv[u] * x[gi[u]]));
        }
        __syncthreads();
    }
    // levels: forward ascending (leaves first), backward descending
    int step = 0, base = 0, ee = 0;
    auto level_range = [&](int st_, int &eb_, int &ee_) {
        const int l = FWDMODE ? st_ : nl - 1 - st_;
        eb_ = lev_e[l];
        ee_ = lev_e[l + 1];
    };
    while (step < nl) { // first non-empty level
        level_range(step, base, ee);
        if (base < ee) break;
        ++step;
    }
    int ci[BSF_U], cj[BSF_U], ni[BSF_U], nj[BSF_U];
    double cv[BSF_U], nv[BSF_U];
    auto request = [&](int bs, int en, int *ii, int *jj, double *vv) {
#pragma unroll
        for (int u = 0; u < BSF_U; ++u) {
            const int t = bs + u * BWG + tid;
            const bool ok = t < en;
            ii[u] = ok ? (int)Li16[t] : 0xFFFF;
            jj[u] = ok ? (int)Lj16[t] : 0;
            vv[u] = ok ? v.Lx[t] : 0.0;
        }
    };
    if (step < nl) request(base, ee, ci, cj, cv);
    while (step < nl) {
        int nstep = step, nbase = base + BWG * BSF_U, nee = ee; // the batch after this one
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
        for (int u = 0; u < BSF_U; ++u) {
            const int i = ci[u];
            if (i >= nloc) continue; // (a top row, or beyond the batch)
            if (FWDMODE) atomicAdd(&xs[i], -(cv[u] * xs[cj[u]]));
            else atomicAdd(&xs[cj[u]], -(cv[u] * xs[i]));
        }
        if (nstep != step) __syncthreads(); // the level is complete
        step = nstep;
        base = nbase;
        ee = nee;
#pragma unroll
        for (int u = 0; u < BSF_U; ++u) {
            ci[u] = ni[u];
            cj[u] = nj[u];
            cv[u] = nv[u];
        }
    }
    __syncthreads();
    // addv: the refinement step x + dx folded into the final write of the backward sweep
    if (addv)
        for (int i = tid; i < nloc; i += BWG) x[s0 + i] = xs[i] + addv[s0 + i];
    else
        for (int i = tid; i < nloc; i += BWG) x[s0 + i] = xs[i];
}

template <int MODE>
__global__ __launch_bounds__(WG) void k_gather_Bprep(GatherArgs a, const int *__restrict__ rows, int count) {
    const int t = blockIdx.x * WG + threadIdx.x;
    if (t >= count) return;
    const int r = rows[t];
    if (MODE == BWD) a.out[r] = a.out[r] * a.aux[r];
    else if (MODE == SYMV) a.out[r] = a.aux[r];
    else if (MODE == SPMV) a.out[r] = a.aux ? a.aux[r] : 0.0;
}
// T, W and B work of one level in ONE launch: the three classes are independent, so their
// blocks simply coexist in the grid (long B chunks first, then wave-per-row, then the
// thread-per-row slab with its XCD-aware mapping).  off8 = first T block, a multiple of 8.
template <int MODE>
__global__ __launch_bounds__(WG) void k_gather_merged(GatherArgs a, const int *__restrict__ trows, int tcount,
                                                      const int *__restrict__ wrows, int wcount,
                                                      const int *__restrict__ crow,
                                                      const int *__restrict__ cbeg,
                                                      const int *__restrict__ cend, int ccount, int off8) {
    __shared__ double red[16];
    const int bid = blockIdx.x;
    if (bid >= off8) {
        const int lb0 = bid - off8, per = (gridDim.x - off8) >> 3;
        const int lb = (lb0 & 7) * per + (lb0 >> 3);
        const int tid = lb * WG + threadIdx.x;
        double v = 0.0;
        if (tid < tcount) {
            const int r = trows[tid];
            const int b = a.ptr[r], e = a.ptr[r + 1];
            double s = 0.0;
            for (int t = b; t < e; ++t) s += a.val[t] * a.xin[a.idx[t]];
            v = store_row<MODE>(a, r, s);
        }
        if (MODE == SYMV && a.nrm) {
            const bool nan = v != v;
            const double m = block_max(nan ? 0.0 : fabs(v), red);
            if (__syncthreads_or(nan)) {
                if (threadIdx.x == 0) *a.nan = 1;
            }
            if (threadIdx.x == 0) fold_norm(a.nrm, a.nan, m, false, lb);
        }
    } else if (bid < ccount) {
        // 4 independent gathers in flight per thread (the chunk is one long dot product)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        const int ce = cend[bid];
        int t = cbeg[bid] + threadIdx.x;
        for (; t + 3 * WG < ce; t += 4 * WG) {
            const int i0 = a.idx[t], i1 = a.idx[t + WG], i2 = a.idx[t + 2 * WG], i3 = a.idx[t + 3 * WG];
            const double v0 = a.val[t], v1 = a.val[t + WG], v2 = a.val[t + 2 * WG], v3 = a.val[t + 3 * WG];
            s0 += v0 * a.xin[i0];
            s1 += v1 * a.xin[i1];
            s2 += v2 * a.xin[i2];
            s3 += v3 * a.xin[i3];
        }
        for (; t < ce; t += WG) s0 += a.val[t] * a.xin[a.idx[t]];
        double s = (s0 + s1) + (s2 + s3);
        s = block_sum(s, red);
        if (threadIdx.x == 0) atomicAdd(&a.out[crow[bid]], MODE == SPMV ? a.alpha * s : -s);
    } else {
        const int wid = (bid - ccount) * 4 + (threadIdx.x >> 6);
        if (wid >= wcount) return;
        const int lane = threadIdx.x & 63;
        const int r = wrows[wid];
        const int b = a.ptr[r], e = a.ptr[r + 1];
        // four independent (index, value) -> gather chains in flight per lane (a 2000-entry row of a dense front
        // is 33 rounds of two dependent round trips otherwise: config 5's residual over the top rows ran at 1.8 TB/s)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int t = b + lane;
        for (; t + 192 < e; t += 256) {
            const int i0 = a.idx[t], i1 = a.idx[t + 64], i2 = a.idx[t + 128], i3 = a.idx[t + 192];
            const double v0 = a.val[t], v1 = a.val[t + 64], v2 = a.val[t + 128], v3 = a.val[t + 192];
            s0 += v0 * a.xin[i0];
            s1 += v1 * a.xin[i1];
            s2 += v2 * a.xin[i2];
            s3 += v3 * a.xin[i3];
        }
        for (; t < e; t += 64) s0 += a.val[t] * a.xin[a.idx[t]];
        const double s = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) {
            const double v = store_row<MODE>(a, r, s);
            if (MODE == SYMV && a.nrm) fold_norm(a.nrm, a.nan, v != v ? 0.0 : fabs(v), v != v, wid);
        }
    }
}

__global__ __launch_bounds__(BWG) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_bundle_symv(BundleView bv, const int *__restrict__ Up, const int *__restrict__ Ucol,
                   const double *__restrict__ Ux, const double *__restrict__ x,
                   const double *__restrict__ b, double *e, unsigned long long *nrm, int *nanflag,
                   FoldView fold) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[16];
    bundle_symv_body(bv, Up, Ucol, Ux, x, b, e, nrm, nanflag, (double *)smem, red, fold);
}
// the k x k top-top part of both sweeps of a folded top (k <= 8): forward with the bundle parts already
// subtracted from x[top], D^-1, backward
__global__ void k_fold_top_solve(LdlView v, FoldView fold, double *x) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int k = fold.k, NF = fold.NF;
    double y[8];
    for (int i = 0; i < k; ++i) {
        double s = x[NF + i];
        for (int q = 0; q < FOLD_SLOTS; ++q) { // the bundles' shares of row i (reset for the next sweep)
            double *a = &fold.acc[fold_acc_index(0, i, q)];
            s -= *a;
            *a = 0.0;
        }
        for (int j = 0; j < i; ++j) {
            const int q = fold.tt[i * k + j];
            if (q >= 0) s -= v.Lx[q] * y[j];
        }
        y[i] = s;
    }
    for (int i = k - 1; i >= 0; --i) {
        double s = y[i] * v.Dinv[NF + i];
        for (int j = i + 1; j < k; ++j) {
            const int q = fold.tt[j * k + i];
            if (q >= 0) s -= v.Lx[q] * y[j];
        }
        y[i] = s; // y now holds x for rows >= i
    }
    for (int i = 0; i < k; ++i) x[NF + i] = y[i];
}
// residual of the folded top rows: bundle shares from fold.tsum (reset here), top-top entries from S
__global__ void k_fold_top_residual(FoldView fold, const double *__restrict__ Sx, const double *__restrict__ x,
                                    const double *__restrict__ b, double *e, unsigned long long *nrm,
                                    int *nanflag) {
    const int i = threadIdx.x;
    if (blockIdx.x != 0 || i >= fold.k) return;
    double s = 0.0;
    for (int q = 0; q < FOLD_SLOTS; ++q) {
        double *a = &fold.acc[fold_acc_index(1, i, q)];
        s += *a;
        *a = 0.0;
    }
    for (int t = fold.sp[i]; t < fold.sp[i + 1]; ++t) s += Sx[fold.sslot[t]] * x[fold.NF + fold.scol[t]];
    const double val = b[fold.NF + i] - s;
    e[fold.NF + i] = val;
    if (nrm) fold_norm(nrm, nanflag, val != val ? 0.0 : fabs(val), val != val, i);
}

// A run of consecutive NARROW levels (a chain-like stretch of the elimination tree: a handful
// of rows per level) handled by ONE 1024-thread workgroup that walks the levels with
// __syncthreads() in between -- a few us per level instead of one launch per level.  A level
// is a string of dependent L2 round trips (level pointers -> row ids -> row pointers -> entries
// -> gathers), so: the level pointers of the whole run sit in LDS, the row id and row pointers of
// a wavefront group's row on the NEXT level are requested while the current level is computed,
// few rows share the 16 wavefronts, and every lane keeps 4 entries in flight.
constexpr int CHAIN_CAP = 4096; // levels per launch (LDS copy of their T / W pointers)
template <int MODE>
__global__ __launch_bounds__(1024) void k_chain(GatherArgs a, const int *__restrict__ t_idx,
                                                const int *__restrict__ t_ptr,
                                                const int *__restrict__ w_idx,
                                                const int *__restrict__ w_ptr, int l0, int l1) {
    __shared__ double part[16];
    __shared__ int tp[CHAIN_CAP + 1], wp[CHAIN_CAP + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nlev = l1 - l0; // <= CHAIN_CAP
    for (int i = tid; i <= nlev; i += 1024) {
        tp[i] = t_ptr[l0 + i];
        wp[i] = w_ptr[l0 + i];
    }
    __syncthreads();
    // this wavefront's role on a level with nw W rows: `share` wavefronts per row
    auto role = [&](int nw, int &share, int &grp, int &sub) {
        share = 1;
        while (share < 16 && nw * share * 2 <= 16) share *= 2;
        grp = wv / share;
        sub = wv % share;
    };
    auto level_at = [&](int step) { return (MODE == FWD) ? step : nlev - 1 - step; };
    // prefetched row of this wavefront group for the current level: id, first and last+1 slot
    // ... and the row's own entry (rhs value, times 1/d in the backward sweep): only row r's level
    // writes out[r], so it can be read a level ahead
    int nr = -1, nb = 0, ne = 0;
    double nown = 0.0;
    auto own_of = [&](int r) { return MODE == BWD ? a.out[r] * a.aux[r] : a.out[r]; };
    {
        const int ll = level_at(0), nw = wp[ll + 1] - wp[ll];
        int share, grp, sub;
        role(nw, share, grp, sub);
        if (grp < nw) {
            nr = w_idx[wp[ll] + grp];
            nb = a.ptr[nr];
            ne = a.ptr[nr + 1];
            nown = own_of(nr);
        }
    }
    for (int step = 0; step < nlev; ++step) {
        const int ll = level_at(step);
        const int r0 = nr, b0 = nb, e0 = ne;
        const double own0 = nown;
        // request the next level's row id now; its pointers are read at the end of this level
        int nxt = -1;
        if (step + 1 < nlev) {
            const int ln = level_at(step + 1), nwn = wp[ln + 1] - wp[ln];
            int share, grp, sub;
            role(nwn, share, grp, sub);
            if (grp < nwn) nxt = w_idx[wp[ln] + grp];
        }
        for (int i = tp[ll] + tid; i < tp[ll + 1]; i += 1024) {
            const int r = t_idx[i];
            double s = 0.0;
            for (int t = a.ptr[r]; t < a.ptr[r + 1]; ++t) s += a.val[t] * a.xin[a.idx[t]];
            store_row<MODE>(a, r, s);
        }
        const int nw = wp[ll + 1] - wp[ll];
        if (nw > 0) {
            int share, grp, sub;
            role(nw, share, grp, sub);
            const int ngrp = 16 / share, stride = 64 * share;
            for (int i0 = 0; i0 < nw; i0 += ngrp) {
                const int i = i0 + grp;
                double s = 0.0;
                int r = -1;
                if (i < nw) {
                    int t, e;
                    if (i0 == 0) {
                        r = r0;
                        t = b0;
                        e = e0;
                    } else {
                        r = w_idx[wp[ll] + i];
                        t = a.ptr[r];
                        e = a.ptr[r + 1];
                    }
                    t += sub * 64 + lane;
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    for (; t + 3 * stride < e; t += 4 * stride) {
                        const int j0 = a.idx[t], j1 = a.idx[t + stride], j2 = a.idx[t + 2 * stride],
                                  j3 = a.idx[t + 3 * stride];
                        const double v0 = a.val[t], v1 = a.val[t + stride], v2 = a.val[t + 2 * stride],
                                     v3 = a.val[t + 3 * stride];
                        s0 += v0 * a.xin[j0];
                        s1 += v1 * a.xin[j1];
                        s2 += v2 * a.xin[j2];
                        s3 += v3 * a.xin[j3];
                    }
                    for (; t < e; t += stride) s0 += a.val[t] * a.xin[a.idx[t]];
                    s = wave_sum((s0 + s1) + (s2 + s3));
                }
                if (share == 1) {
                    if (lane == 0 && r >= 0) {
                        if (i0 == 0) a.out[r] = own0 - s;
                        else store_row<MODE>(a, r, s);
                    }
                } else {
                    if (lane == 0) part[wv] = s;
                    __syncthreads();
                    if (lane == 0 && sub == 0 && r >= 0) {
                        double tot = 0.0;
                        for (int q = 0; q < share; ++q) tot += part[grp * share + q];
                        if (i0 == 0) a.out[r] = own0 - tot;
                        else store_row<MODE>(a, r, tot);
                    }
                    __syncthreads();
                }
            }
        }
        nr = nxt;
        nb = nxt >= 0 ? a.ptr[nxt] : 0;
        ne = nxt >= 0 ? a.ptr[nxt + 1] : 0;
        nown = nxt >= 0 ? own_of(nxt) : 0.0;
        __syncthreads(); // level final and visible workgroup-wide
    }
}

// ---------------------------------------------------------------------------
// Blocked substitution over a tall top (chain-like elimination trees: config 2 has ~4400
// sequential top levels).  L = [L_11 0; L_21 L_22] with unit-lower diagonal blocks of TOPBLK rows:
//   forward   y_b = (I + L_bb)^-1 (b_b - L_b,<b y_<b)
//   backward  x_b = (I + L_bb)^-T (D_b^-1 y_b - L_>b,b' x_>b)
// T_b = (I + L_bb)^-1 is formed once per refactor, so a sweep has one dependent step per block
// of 128 rows instead of one per elimination-tree level.
// ---------------------------------------------------------------------------
constexpr int TOPBLK = 128;
constexpr int TOPBLK_PACK = TOPBLK * (TOPBLK - 1) / 2;
__device__ __forceinline__ int tri_idx(int i, int k) { return i * (i - 1) / 2 + k; } // k < i
// one workgroup per block: M = strictly-lower part of L_bb (dense, packed) in LDS, then
// T = (I + M)^-1 column by column (columns are independent: no barriers), packed by rows
__global__ __launch_bounds__(WG) void k_topblk_build(LdlView v, TopBlkView tb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *M = (double *)smem, *T = M + TOPBLK_PACK;
    const int b = blockIdx.x, r0 = tb.NF + b * tb.w, w = min(tb.w, tb.N - r0), tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < TOPBLK_PACK; i += WG) M[i] = 0.0;
    __syncthreads();
    for (int i = wv; i < w; i += WG / 64) {
        const int j = r0 + i;
        for (int t = tb.Rsplit[j - tb.NF] + lane; t < v.Rp[j + 1]; t += 64) M[tri_idx(i, v.Rcol[t] - r0)] = v.Rx[t];
    }
    __syncthreads();
    // column c of T: T[i][c] = -(M[i][c] + sum_{c < k < i} M[i][k] T[k][c])
    for (int c = tid; c < w; c += WG) {
        for (int i = c + 1; i < w; ++i) {
            double s = M[tri_idx(i, c)];
            for (int k = c + 1; k < i; ++k) s += M[tri_idx(i, k)] * T[tri_idx(k, c)];
            T[tri_idx(i, c)] = -s;
        }
    }
    __syncthreads();
    double *out = tb.T + (size_t)b * TOPBLK_PACK;
    const int np = w * (w - 1) / 2;
    for (int i = tid; i < np; i += WG) out[i] = T[i];
}
// One launch per block, in sweep order on the stream.  The rows of the block are spread over
// workgroups (a wavefront per row: the external part of the row, 4 entries per lane in flight) so
// that the whole GPU streams them; the workgroup that finishes last (agent-scope ticket) applies
// the inverted diagonal block from LDS and publishes the block's slice of x.
template <int MODE>
__global__ __launch_bounds__(1024) void k_topblk_step(LdlView v, TopBlkView tb, double *x, int b, double *ysg,
                                                    int *counters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *T = (double *)smem, *ys = T + TOPBLK_PACK;
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r0 = tb.NF + b * tb.w, w = min(tb.w, tb.N - r0), np = w * (w - 1) / 2;
    const int i = blockIdx.x * 16 + wv; // 16 wavefronts = 16 rows per workgroup
    if (i < w) {
        const int j = r0 + i;
        int t, e;
        const int *idx;
        const double *val;
        if (MODE == FWD) {
            t = v.Rp[j];
            e = tb.Rsplit[j - tb.NF];
            idx = v.Rcol;
            val = v.Rx;
        } else {
            t = tb.Lsplit[j - tb.NF];
            e = v.Lp[j + 1];
            idx = v.Li;
            val = v.Lx;
        }
        t += lane;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (; t + 192 < e; t += 256) {
            const int j0 = idx[t], j1 = idx[t + 64], j2 = idx[t + 128], j3 = idx[t + 192];
            const double v0 = val[t], v1 = val[t + 64], v2 = val[t + 128], v3 = val[t + 192];
            s0 += v0 * x[j0];
            s1 += v1 * x[j1];
            s2 += v2 * x[j2];
            s3 += v3 * x[j3];
        }
        for (; t < e; t += 64) s0 += val[t] * x[idx[t]];
        const double s = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) ysg[b * TOPBLK + i] = (MODE == FWD ? x[j] : x[j] * v.Dinv[j]) - s;
    }
    __threadfence(); // release this workgroup's rows
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&counters[b], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence(); // acquire every other workgroup's rows
    const double *Tg = tb.T + (size_t)b * TOPBLK_PACK;
    for (int k = tid; k < np; k += 1024) T[k] = Tg[k];
    for (int k = tid; k < w; k += 1024) ys[k] = __hip_atomic_load(&ysg[b * TOPBLK + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // x_b = T y (forward: lower triangle, unit diagonal) or T' y (backward); 8 threads per row
    {
        const int r = tid >> 3, sub = tid & 7;
        double s = 0.0;
        if (r < w) {
            if (MODE == FWD) {
                for (int k = sub; k < r; k += 8) s += T[tri_idx(r, k)] * ys[k];
            } else {
                for (int k = r + 1 + sub; k < w; k += 8) s += T[tri_idx(k, r)] * ys[k];
            }
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (r < w && sub == 0) x[r0 + r] = ys[r] + s;
    }
    if (tid == 0) counters[b] = 0; // ready for the next sweep
}

__global__ __launch_bounds__(WG) void k_norm_rows(const double *__restrict__ vv, const int *__restrict__ rows,
                                                  int count, unsigned long long *nrm, int *nan) {
    const int t = blockIdx.x * WG + threadIdx.x;
    if (t >= count) return;
    const double a = vv[rows[t]];
    fold_norm(nrm, nan, a != a ? 0.0 : fabs(a), a != a, t);
}


} // namespace

static size_t bundle_lds(const BundleView &bv) { return ((size_t)bv.max_nodes * sizeof(double) + 15) & ~(size_t)15; }
// the entry-parallel form: the system has the 16-bit local indices, every bundle's level table fits, no fold
static bool bundle_sweeps_flat(const LdlView &v, const BundleView &bv, const FoldView *fold) {
    return v.sLi16 && v.sLj16 && bv.max_levels <= BSF_MAXLEV && !(fold && fold->k) && !switches().no_bundle_flat_sweep &&
           !switches().deterministic;
}
void bundle_fwd(hipStream_t s, const LdlView &v, const BundleView &bv, double *x, const FoldView &fold) {
    if (!bv.nb) return;
    if (bundle_sweeps_flat(v, bv, &fold)) k_bundle_sweep_flat<true><<<bv.nb, BWG, bundle_lds(bv), s>>>(v, bv, x, nullptr);
    else k_bundle_fwd<<<bv.nb, BWG, bundle_lds(bv), s>>>(v, bv, x, fold);
}
void bundle_bwd(hipStream_t s, const LdlView &v, const BundleView &bv, double *x, const double *addv) {
    if (!bv.nb) return;
    if (bundle_sweeps_flat(v, bv, nullptr)) k_bundle_sweep_flat<false><<<bv.nb, BWG, bundle_lds(bv), s>>>(v, bv, x, addv);
    else k_bundle_bwd<<<bv.nb, BWG, bundle_lds(bv), s>>>(v, bv, x, addv);
}
void fold_top_solve(hipStream_t s, const LdlView &v, const FoldView &fold, double *x) {
    if (fold.k) k_fold_top_solve<<<1, 64, 0, s>>>(v, fold, x);
}
void fold_top_residual(hipStream_t s, const FoldView &fold, const double *Sx, const double *x, const double *b,
                       double *e, unsigned long long *nrm, int *nan) {
    if (fold.k) k_fold_top_residual<<<1, 64, 0, s>>>(fold, Sx, x, b, e, nrm, nan);
}
void bundle_symv(hipStream_t s, const BundleView &bv, const int *Up, const int *Ucol, const double *Ux,
                 const double *x, const double *b, double *e, unsigned long long *nrm, int *nan,
                 const FoldView &fold, hipEvent_t ev0, hipEvent_t ev1) {
    if (!bv.nb) return;
    const size_t lds = ((size_t)bv.max_nodes * sizeof(double) + 15) & ~(size_t)15; // the e slice only
    if (ev0 && ev1) // profiling: the command processor stamps the events right around THIS kernel
        hipExtLaunchKernelGGL(k_bundle_symv, dim3(bv.nb), dim3(BWG), lds, s, ev0, ev1, 0, bv, Up, Ucol, Ux, x, b, e, nrm,
                              nan, fold);
    else k_bundle_symv<<<bv.nb, BWG, lds, s>>>(bv, Up, Ucol, Ux, x, b, e, nrm, nan, fold);
}

#define DISPATCH_MODE(KERNEL, GRID, ...)                                   \
    switch (m) {                                                           \
    case FWD: KERNEL<FWD><<<GRID, WG, 0, s>>>(__VA_ARGS__); break;         \
    case BWD: KERNEL<BWD><<<GRID, WG, 0, s>>>(__VA_ARGS__); break;         \
    case SPMV: KERNEL<SPMV><<<GRID, WG, 0, s>>>(__VA_ARGS__); break;       \
    default: KERNEL<SYMV><<<GRID, WG, 0, s>>>(__VA_ARGS__); break;         \
    }

void gather_merged(hipStream_t s, GatherMode m, const GatherArgs &a, ListView t, ListView w, ChunkView c) {
    if (!t.count && !w.count && !c.count) return;
    const int nbW = (w.count + 3) / 4;
    const int off8 = (c.count + nbW + 7) & ~7;
    const int nbT = t.count ? grid_for(t.count) : 0;
    const int grid = off8 + nbT;
    DISPATCH_MODE(k_gather_merged, grid, a, t.idx, t.count, w.idx, w.count, c.row, c.beg, c.end, c.count, off8)
}
void topblk_build(hipStream_t s, const LdlView &v, const TopBlkView &tb) {
    if (!tb.nblocks) return;
    const size_t lds = (size_t)2 * TOPBLK_PACK * sizeof(double);
    (void)raise_dynamic_lds((const void *)k_topblk_build, (size_t)lds);
    k_topblk_build<<<tb.nblocks, WG, lds, s>>>(v, tb);
}
// kernels of the solve sequence that need more than 64 KB of dynamic LDS: allowed once per process,
// outside any stream capture
void solve_kernel_attributes() {
    const size_t lds = (size_t)(TOPBLK_PACK + TOPBLK) * sizeof(double);
    (void)raise_dynamic_lds((const void *)k_topblk_step<FWD>, (size_t)lds);
    (void)raise_dynamic_lds((const void *)k_topblk_step<BWD>, (size_t)lds);
}
void topblk_solve(hipStream_t s, GatherMode m, const LdlView &v, const TopBlkView &tb, double *x) {
    if (!tb.nblocks) return;
    const size_t lds = (size_t)(TOPBLK_PACK + TOPBLK) * sizeof(double);
    for (int step = 0; step < tb.nblocks; ++step) {
        const int b = m == FWD ? step : tb.nblocks - 1 - step;
        const int w = std::min(tb.w, tb.N - (tb.NF + b * tb.w));
        const int grid = (w + 15) / 16;
        if (m == FWD) k_topblk_step<FWD><<<grid, 1024, lds, s>>>(v, tb, x, b, tb.ys, tb.counters);
        else k_topblk_step<BWD><<<grid, 1024, lds, s>>>(v, tb, x, b, tb.ys, tb.counters);
    }
}
void gather_chain(hipStream_t s, GatherMode m, const GatherArgs &a, const int *t_idx, const int *t_ptr,
                  const int *w_idx, const int *w_ptr, int l0, int l1) {
    // at most CHAIN_CAP levels per launch, in sweep order
    if (m == FWD) {
        for (int b = l0; b < l1; b += CHAIN_CAP)
            k_chain<FWD><<<1, 1024, 0, s>>>(a, t_idx, t_ptr, w_idx, w_ptr, b, std::min(l1, b + CHAIN_CAP));
    } else {
        for (int e = l1; e > l0; e -= CHAIN_CAP)
            k_chain<BWD><<<1, 1024, 0, s>>>(a, t_idx, t_ptr, w_idx, w_ptr, std::max(l0, e - CHAIN_CAP), e);
    }
}
void gather_Bprep(hipStream_t s, GatherMode m, const GatherArgs &a, ListView r) {
    if (!r.count || m == FWD) return;
    DISPATCH_MODE(k_gather_Bprep, (r.count + WG - 1) / WG, a, r.idx, r.count)
}
void norm_rows(hipStream_t s, const double *v, ListView rows, unsigned long long *nrm, int *nan) {
    if (rows.count) k_norm_rows<<<(rows.count + WG - 1) / WG, WG, 0, s>>>(v, rows.idx, rows.count, nrm, nan);
}

} // namespace dev
} // namespace chip
