// bundle_factor.hip -- numeric LDL': top columns by level (k_factor_T / W / B / chain), subtree bundles (k_bundle_factor, _lds, _flat), grouped-fold tops
// (one of the translation units behind kernels.hpp; the design rules and the reference citations are in
// dev_common.hpp)
#include "dev_common.hpp"

namespace chip {
namespace dev {

namespace {

// ---------------------------------------------------------------------------
// numeric LDL': left-looking by columns, one launch per elimination-tree level
//
//   c_ij = a_ij - sum_{k in rowstruct(j)} l_ik d_k l_jk   (i in colstruct(j))
//   d_j  = a_jj - sum_k l_jk^2 d_k ;  pivot rule ;  l_ij = c_ij / d_j
//
// Every k in rowstruct(j) is a descendant of j (lower level, finished in an
// earlier launch), so a level's columns are independent.  Column j's slots
// hold a_ij on entry (k_scatter_init) and l_ij on exit, also mirrored into
// the row-major copy Rx that the forward substitution streams.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void finish_column_serial(const LdlView &v, int j, int cb, int ce, double d) {
    const double dinv = pivot_rule(v, j, d);
    for (int q = cb; q < ce; ++q) {
        const double l = v.Lx[q] * dinv;
        v.Lx[q] = l;
        if (v.mirror_rows) v.Rx[v.Tpos[q]] = l;
    }
}

// diagonal of bundle column j as the factorisation starts: K_jj (first entry of U row j) shifted by the
// static regulariser (directldlkktsolver.rs:233-245: +eps where Dsigns = +1, -eps otherwise)
__device__ __forceinline__ double diag_from_U(const LdlView &v, int j) {
    const double val = v.Ux[v.Up[j]];
    if (!v.eps_ptr) return val;
    const double eps = v.eps_ptr[0];
    return v.dsigns[j] == 1 ? val + eps : val - eps;
}

// one thread factors column j (few contributions, short column).  INIT_U: the column's initial values
// are merged from U row j (its entries to ancestors: a subset of the column's rows, both ascending) --
// nothing has been scattered into Lx / D beforehand; otherwise they are found in Lx / D.
template <bool INIT_U>
__device__ __forceinline__ void factor_col_thread(const LdlView &v, int j) {
    double d = INIT_U ? diag_from_U(v, j) : v.D[j];
    const int cb = v.Lp[j], ce = v.Lp[j + 1];
    const int rb = v.Rp[j], re = v.Rp[j + 1];
    if (ce - cb <= 4) {
        // tiny column (the bulk of block-arrow KKTs): its row ids and running values live in
        // registers -- no search loads, no read-modify-write round trips through L2
        const int cn = ce - cb;
        const int r0 = cn > 0 ? v.Li[cb] : -1, r1 = cn > 1 ? v.Li[cb + 1] : -1;
        const int r2 = cn > 2 ? v.Li[cb + 2] : -1;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        if (INIT_U) {
            const int ub = v.Up[j] + 1, ue = v.Up[j + 1];
            int hi[4];
            double hv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hi[q] = ub + q < ue ? v.Ucol[ub + q] : -2;
                hv[q] = ub + q < ue ? v.Ux[ub + q] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (hi[q] == r0) a0 = hv[q];
                else if (hi[q] == r1) a1 = hv[q];
                else if (hi[q] == r2) a2 = hv[q];
                else if (hi[q] >= 0) a3 = hv[q];
            }
        } else {
            a0 = cn > 0 ? v.Lx[cb] : 0.0;
            a1 = cn > 1 ? v.Lx[cb + 1] : 0.0;
            a2 = cn > 2 ? v.Lx[cb + 2] : 0.0;
            a3 = cn > 3 ? v.Lx[cb + 3] : 0.0;
        }
        for (int t = rb; t < re; ++t) {
            const int k = v.Rcol[t], p = v.Rpos[t];
            const double ljk = v.Lx[p];
            const double w = ljk * v.D[k];
            d -= ljk * w;
            const int pe = v.Lp[k + 1];
            for (int pp = p + 1; pp < pe; ++pp) {
                const int i = v.Li[pp];
                const double u = v.Lx[pp] * w;
                if (i == r0) a0 -= u;
                else if (i == r1) a1 -= u;
                else if (i == r2) a2 -= u;
                else a3 -= u;
            }
        }
        const double dinv = pivot_rule(v, j, d);
        const double a[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < cn) {
                const double l = a[q] * dinv;
                v.Lx[cb + q] = l;
                if (v.mirror_rows) v.Rx[v.Tpos[cb + q]] = l;
            }
        return;
    }
    if (INIT_U) { // merge U row j into the column (slots without an entry of K: fill-in, zero)
        int u = v.Up[j] + 1;
        const int ue = v.Up[j + 1];
        for (int q = cb; q < ce; ++q) {
            double val = 0.0;
            if (u < ue && v.Ucol[u] == v.Li[q]) val = v.Ux[u++];
            v.Lx[q] = val;
        }
    }
    for (int t = rb; t < re; ++t) {
        const int k = v.Rcol[t], p = v.Rpos[t];
        const double ljk = v.Lx[p];
        const double w = ljk * v.D[k];
        d -= ljk * w;
        const int pe = v.Lp[k + 1];
        int q = cb;
        for (int pp = p + 1; pp < pe; ++pp) {
            const int i = v.Li[pp];
            while (v.Li[q] != i) ++q; // rows below j of column k are a subset of column j
            v.Lx[q] -= v.Lx[pp] * w;
            ++q;
        }
    }
    finish_column_serial(v, j, cb, ce, d);
}

// T: one thread per column
__global__ __launch_bounds__(WG) void k_factor_T(LdlView v, const int *__restrict__ cols, int count) {
    const int tid = logical_block() * WG + threadIdx.x;
    if (tid >= count) return;
    factor_col_thread<false>(v, cols[tid]);
}

constexpr int RCAP = 512;      // contributions per flattened batch (scan needs blockDim >= RCAP/2)
constexpr int W_LDS_CAP = 2048; // column values + row ids kept in LDS (16 + 8 KiB of 160 KiB)
constexpr int B_LDS_CAP = 2048; // the same for the chunked column kernel (padded supernode columns are long)

__device__ __forceinline__ int find_row(const int *__restrict__ Li, int lo, int hi, int row) {
    // first q in [lo,hi) with Li[q] >= row (the row is known to be present)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (Li[mid] < row) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// A whole 256-thread workgroup factors column j.  Threads stride over the
// contributing columns k; column j's running values live in LDS and receive
// LDS fp64 atomics (ds_add_f64).  Narrow columns (<= 4 rows: the u/v and
// budget-like separators of block-arrow KKTs) take per-thread register
// partials + one block reduction instead of hammering 4 LDS addresses.
// Must be called by all threads of the workgroup; ends un-synchronised.
template <bool INIT_U>
__device__ __forceinline__ void factor_col_block(const LdlView &v, int j, double *acc, int *rows, int *cst,
                                                 double *cw, int *coff, double *red, double *s_dinv) {
    const int cb = v.Lp[j], cn = v.Lp[j + 1] - cb;
    const int rb = v.Rp[j], rn = v.Rp[j + 1] - rb;
    const int tid = threadIdx.x;
    double dpart = 0.0;
    if (cn <= 4) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const int r0 = cn > 0 ? v.Li[cb] : -1, r1 = cn > 1 ? v.Li[cb + 1] : -1;
        const int r2 = cn > 2 ? v.Li[cb + 2] : -1;
        for (int t = tid; t < rn; t += blockDim.x) {
            const int k = v.Rcol[rb + t], p = v.Rpos[rb + t];
            const double ljk = v.Lx[p];
            const double w = ljk * v.D[k];
            dpart += ljk * w;
            const int pe = v.Lp[k + 1];
            for (int pp = p + 1; pp < pe; ++pp) {
                const int i = v.Li[pp];
                const double u = v.Lx[pp] * w;
                if (i == r0) a0 += u;
                else if (i == r1) a1 += u;
                else if (i == r2) a2 += u;
                else a3 += u;
            }
        }
        a0 = block_sum(a0, red);
        a1 = block_sum(a1, red);
        a2 = block_sum(a2, red);
        a3 = block_sum(a3, red);
        dpart = block_sum(dpart, red);
        if (tid == 0) {
            const double dinv = pivot_rule(v, j, (INIT_U ? diag_from_U(v, j) : v.D[j]) - dpart);
            const double a[4] = {a0, a1, a2, a3};
            double k0[4] = {0.0, 0.0, 0.0, 0.0}; // the column's initial values
            if (INIT_U) {
                int u = v.Up[j] + 1;
                const int ue = v.Up[j + 1];
                for (int q = 0; q < cn; ++q)
                    if (u < ue && v.Ucol[u] == v.Li[cb + q]) k0[q] = v.Ux[u++];
            } else {
                for (int q = 0; q < cn; ++q) k0[q] = v.Lx[cb + q];
            }
            for (int q = 0; q < cn; ++q) {
                const double l = (k0[q] - a[q]) * dinv;
                v.Lx[cb + q] = l;
                if (v.mirror_rows) v.Rx[v.Tpos[cb + q]] = l;
            }
        }
        return;
    }
    const bool lds = cn <= W_LDS_CAP;
    if (lds)
        for (int q = tid; q < cn; q += blockDim.x) {
            acc[q] = INIT_U ? 0.0 : v.Lx[cb + q];
            rows[q] = v.Li[cb + q];
        }
    __syncthreads();
    if (INIT_U) { // U row j -> its slots of the column (bundle columns are short: always the LDS path)
        const int ub = v.Up[j] + 1, ue = v.Up[j + 1];
        for (int u = ub + tid; u < ue; u += blockDim.x) {
            const int hi = v.Ucol[u];
            int l2 = 0, h2 = cn;
            while (l2 < h2) {
                const int mid = (l2 + h2) >> 1;
                if (rows[mid] < hi) l2 = mid + 1;
                else h2 = mid;
            }
            acc[l2] = v.Ux[u];
        }
        __syncthreads();
    }
    if (cn >= 24) {
        // General-fill columns: the contributing columns have long tails of very different
        // lengths.  The (contribution, tail entry) pairs are FLATTENED: per batch of up to RCAP
        // contributions an exclusive scan of the tail lengths is built in LDS, then the threads
        // stride over the flat update index u -- adjacent lanes read adjacent entries of a tail
        // (coalesced), every iteration's loads are independent of the previous one, and both
        // lookups (owner of u, slot of the row in column j) are binary searches in LDS.
        for (int base = 0; base < rn; base += RCAP) {
            const int nbt = min(RCAP, rn - base);
            __syncthreads(); // previous batch fully consumed
            for (int t = tid; t < nbt; t += blockDim.x) {
                const int k = v.Rcol[rb + base + t], p = v.Rpos[rb + base + t];
                const double ljk = v.Lx[p];
                const double w = ljk * v.D[k];
                dpart += ljk * w;
                cst[t] = p + 1;
                cw[t] = w;
                coff[t + 1] = v.Lp[k + 1] - (p + 1);
            }
            if (tid == 0) coff[0] = 0;
            __syncthreads();
            // inclusive scan of coff[1..nbt] (Hillis-Steele, <= 10 rounds)
            for (int off = 1; off < nbt; off <<= 1) {
                int add0 = 0, add1 = 0;
                const int i0 = tid + 1, i1 = tid + 1 + (int)blockDim.x;
                if (i0 <= nbt && i0 - off >= 1) add0 = coff[i0 - off];
                if (i1 <= nbt && i1 - off >= 1) add1 = coff[i1 - off];
                __syncthreads();
                if (i0 <= nbt) coff[i0] += add0;
                if (i1 <= nbt) coff[i1] += add1;
                __syncthreads();
            }
            const int total = coff[nbt];
            for (int u = tid; u < total; u += blockDim.x) {
                int lo = 0, hi = nbt; // last t with coff[t] <= u
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (coff[mid] <= u) lo = mid;
                    else hi = mid;
                }
                const int pp = cst[lo] + (u - coff[lo]);
                const int i = v.Li[pp];
                const double val = v.Lx[pp] * cw[lo];
                if (lds) {
                    int l2 = 0, h2 = cn;
                    while (l2 < h2) {
                        const int mid = (l2 + h2) >> 1;
                        if (rows[mid] < i) l2 = mid + 1;
                        else h2 = mid;
                    }
                    atomicAdd(&acc[l2], -val);
                } else { // column too long for LDS: same flattened walk, L2-resident lookups + atomics
                    atomicAdd(&v.Lx[find_row(v.Li, cb, cb + cn, i)], -val);
                }
            }
        }
    } else {
        for (int t = tid; t < rn; t += blockDim.x) {
            const int k = v.Rcol[rb + t], p = v.Rpos[rb + t];
            const double ljk = v.Lx[p];
            const double w = ljk * v.D[k];
            dpart += ljk * w;
            const int pe = v.Lp[k + 1];
            int q = cb;
            for (int pp = p + 1; pp < pe; ++pp) {
                q = find_row(v.Li, q, cb + cn, v.Li[pp]);
                const double u = -(v.Lx[pp] * w);
                if (lds) atomicAdd(&acc[q - cb], u);
                else atomicAdd(&v.Lx[q], u);
                ++q;
            }
        }
    }
    if (!lds) __threadfence();
    dpart = block_sum(dpart, red);
    if (tid == 0) *s_dinv = pivot_rule(v, j, (INIT_U ? diag_from_U(v, j) : v.D[j]) - dpart);
    __syncthreads();
    const double dinv = *s_dinv;
    for (int q = tid; q < cn; q += blockDim.x) {
        const double c = lds ? acc[q]
                             : __hip_atomic_load(&v.Lx[cb + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double l = c * dinv;
        v.Lx[cb + q] = l;
        if (v.mirror_rows) v.Rx[v.Tpos[cb + q]] = l;
    }
}

// W: one workgroup per column
__global__ __launch_bounds__(1024) void k_factor_W(LdlView v, const int *__restrict__ cols, int count) {
    __shared__ double acc[W_LDS_CAP];
    __shared__ int rows[W_LDS_CAP];
    __shared__ int cst[RCAP];
    __shared__ double cw[RCAP];
    __shared__ int coff[RCAP + 1];
    __shared__ double red[16];
    __shared__ double s_dinv;
    if ((int)blockIdx.x >= count) return;
    factor_col_block<false>(v, cols[blockIdx.x], acc, rows, cst, cw, coff, red, &s_dinv);
}

// A run of consecutive NARROW top levels of the factorisation (a chain-like stretch of the
// elimination tree: a banded matrix's separator chain has one or two columns per level) walked by
// ONE 1024-thread workgroup with __syncthreads() between levels, instead of one ~7 us launch
// sequence per level.  Columns of a level: thread-per-column ones together, the others one after
// the other by the whole workgroup.
__global__ __launch_bounds__(1024) void k_factor_chain(LdlView v, const int *__restrict__ t_idx,
                                                       const int *__restrict__ t_ptr,
                                                       const int *__restrict__ w_idx,
                                                       const int *__restrict__ w_ptr, int l0, int l1) {
    __shared__ double acc[W_LDS_CAP];
    __shared__ int rows[W_LDS_CAP];
    __shared__ int cst[RCAP];
    __shared__ double cw[RCAP];
    __shared__ int coff[RCAP + 1];
    __shared__ double red[16];
    __shared__ double s_dinv;
    for (int l = l0; l < l1; ++l) {
        for (int i = t_ptr[l] + threadIdx.x; i < t_ptr[l + 1]; i += 1024) factor_col_thread<false>(v, t_idx[i]);
        for (int i = w_ptr[l]; i < w_ptr[l + 1]; ++i) {
            factor_col_block<false>(v, w_idx[i], acc, rows, cst, cw, coff, red, &s_dinv);
            __syncthreads();
        }
        __syncthreads(); // level l final and visible workgroup-wide
    }
}

// ---------------------------------------------------------------------------
// Subtree bundles: ONE workgroup factors / solves a bundle of complete
// elimination subtrees start to finish, level by level, with __syncthreads()
// between levels -- all the cross-level traffic of the bottom of the tree stays
// inside a CU (the vector slice of the bundle is staged in LDS for the solves),
// and ~N/bundle_size workgroups run concurrently in a single launch instead of
// one launch per level.  Only the few ancestors above the cut ("top") still go
// through the level-scheduled kernels.
// ---------------------------------------------------------------------------


// Per level: every thread sweeps the level's thin columns (strided, no barriers in between,
// so many independent gathers are in flight), parking fat columns in an LDS list that the
// whole workgroup then works through cooperatively.
__global__ __launch_bounds__(BWG) void k_bundle_factor(LdlView v, BundleView bv, FoldView fold) {
    __shared__ double acc[W_LDS_CAP];
    __shared__ int rows[W_LDS_CAP];
    __shared__ int cst[RCAP];
    __shared__ double cw[RCAP];
    __shared__ int coff[RCAP + 1];
    __shared__ double red[16];
    __shared__ double s_dinv;
    __shared__ int fat[FATCAP];
    __shared__ int nfat;
    const int b = blockIdx.x;
    const int *lv = bv.blvl + bv.blvl_ptr[b];
    const int nl = bv.blvl_ptr[b + 1] - bv.blvl_ptr[b] - 1;
    for (int l = 0; l < nl; ++l) {
        const int lb = lv[l], le = lv[l + 1];
        if (threadIdx.x == 0) nfat = 0;
        __syncthreads();
        for (int j = lb + threadIdx.x; j < le; j += BWG) {
            const int rj = v.Rp[j + 1] - v.Rp[j], cj = v.Lp[j + 1] - v.Lp[j];
            bool thin = rj <= FAC_THIN_ROW && cj <= FAC_THIN_COL;
            if (!thin) {
                const int slot = atomicAdd(&nfat, 1);
                if (slot < FATCAP) fat[slot] = j;
                else thin = true; // list full: fall back to the serial path (correct, slower)
            }
            if (thin) factor_col_thread<true>(v, j);
        }
        __syncthreads();
        const int nf = min(nfat, FATCAP);
        for (int f = 0; f < nf; ++f) {
            factor_col_block<true>(v, fat[f], acc, rows, cst, cw, coff, red, &s_dinv);
            __syncthreads();
        }
        // level l is final (global writes visible workgroup-wide) before level l+1
    }
    if (fold.k == 1) {
        // a single dense top row (the arrow's shaft): its pivot d_t = a_tt - sum_k l_tk^2 d_k gets this
        // bundle's share here (the l_tk were just computed as the last entries of the bundle's columns);
        // k_fold_top_pivot then applies the pivot rule
        __syncthreads();
        const int tb = fold.rseg[b * 2], te = fold.rseg[b * 2 + 1];
        double s = 0.0;
        for (int t = tb + (int)threadIdx.x; t < te; t += BWG) {
            const double l = v.Lx[v.Rpos[t]];
            s += l * (l * v.D[v.Rcol[t]]);
        }
        s = block_sum(s, red);
        if (threadIdx.x == 0 && te > tb) atomicAdd(&fold.acc[fold_acc_index(2, 0, b % FOLD_SLOTS)], s);
    }
}
// grouped fold: a bundle's contribution to the Schur complement of its group's top (k <= 8 rows),
// S[i][j] = sum over the bundle's columns c of l_ic d_c l_jc -- the entries of a column in the top rows are its
// LAST ones (16-bit local index >= nloc).  A launch of its own behind k_bundle_factor (its 44 accumulator registers
// would halve that kernel's occupancy): per-thread register accumulators over the packed lower triangle, reduced wave
// by wave in a fixed order; k_gfold_top_factor subtracts the shares of a group's bundles from K_tt and factors the
// k x k block.
constexpr int GSWG = 256;
__global__ __launch_bounds__(GSWG) void k_gfold_schur(LdlView v, BundleView bv, GFoldView gf) {
    __shared__ double wsum[(GSWG / 64) * 36];
    const int b = blockIdx.x;
    if (gf.bgrp[b] < 0) return;
    {
        const int s0 = bv.bundle_ptr[b], s1 = bv.bundle_ptr[b + 1], nloc = s1 - s0;
        double sa[36];
#pragma unroll
        for (int p = 0; p < 36; ++p) sa[p] = 0.0;
        for (int j = s0 + (int)threadIdx.x; j < s1; j += GSWG) {
            const int cb = v.Lp[j], ce = v.Lp[j + 1];
            double vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) vv[i] = 0.0;
            bool any = false;
            // (the last 8 entries of the column requested at once: the top rows sort behind the bundle's own)
            int ti8[8];
            double va8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int q = ce - 1 - e;
                ti8[e] = q >= cb ? (int)v.Li16[q] - nloc : -1;
                va8[e] = q >= cb ? v.Lx[q] : 0.0;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (ti8[e] >= 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (ti8[e] == i) vv[i] = va8[e];
                    any = true;
                }
            if (any) {
                const double dj = v.D[j];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double wi = vv[i] * dj;
#pragma unroll
                    for (int jj = 0; jj <= i; ++jj) sa[i * (i + 1) / 2 + jj] += wi * vv[jj];
                }
            }
        }
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            const double t = wave_sum(sa[p]);
            if (lane == 0) wsum[wv * 36 + p] = t;
        }
        __syncthreads();
        if (threadIdx.x < 36) {
            double t = 0.0;
            for (int w = 0; w < GSWG / 64; ++w) t += wsum[w * 36 + threadIdx.x];
            gf.fac[(size_t)b * 36 + threadIdx.x] = t;
        }
    }
}
// grouped fold: the k x k block of every group's top -- K_tt (scattered into D / the top-top slots of Lx by
// k_scatter_init, static regulariser included) minus the Schur contributions of the group's bundles, then LDL' with
// the pivot rule of qdldl.rs:645-665; one thread per group
__global__ __launch_bounds__(64) void k_gfold_top_factor(LdlView v, GFoldView gf) {
    // one wave per group: lane p < k (k + 1) / 2 owns entry p of the packed lower triangle -- its initial value and
    // the shares of the group's bundles (summed in a fixed order: run-to-run reproducible) -- then lane 0 factors the
    // k x k block from LDS
    __shared__ double A[36];
    const int g = blockIdx.x, lane = threadIdx.x;
    const int base = gf.ptr[g], k = gf.ptr[g + 1] - base, np = k * (k + 1) / 2;
    if (lane < np) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= lane) ++i;
        const int j = lane - i * (i + 1) / 2;
        double a;
        if (i == j) a = v.D[gf.node[base + i]];
        else {
            const int q = gf.tt[g * 64 + i * 8 + j];
            a = q >= 0 ? v.Lx[q] : 0.0;
        }
        for (int b = gf.bptr[g]; b < gf.bptr[g + 1]; ++b) a -= gf.fac[(size_t)b * 36 + lane];
        A[lane] = a;
    }
    __syncthreads();
    if (lane != 0) return;
    for (int j = 0; j < k; ++j) {
        const int nj = gf.node[base + j];
        const double dinv = pivot_rule(v, nj, A[j * (j + 1) / 2 + j]);
        for (int i = j + 1; i < k; ++i) {
            const double aij = A[i * (i + 1) / 2 + j];
            for (int i2 = j + 1; i2 <= i; ++i2) A[i * (i + 1) / 2 + i2] -= aij * (A[i2 * (i2 + 1) / 2 + j] * dinv);
        }
        for (int i = j + 1; i < k; ++i) {
            const double lij = A[i * (i + 1) / 2 + j] * dinv;
            A[i * (i + 1) / 2 + j] = lij;
            const int q = gf.tt[g * 64 + i * 8 + j];
            if (q >= 0) {
                v.Lx[q] = lij;
                if (v.mirror_rows) v.Rx[v.Tpos[q]] = lij;
            }
        }
    }
}
// ---------------------------------------------------------------------------
// The same factorisation with the bundle's VALUES resident in LDS (fused handles whose bundles have < 65535 entries
// and fit two workgroups per CU).  k_bundle_factor streams ~2.2 x the algorithmic bytes (PMC: 486 MB on config 3)
// because the left-looking form re-reads what the workgroup wrote a level earlier -- l_jk, d_k, the tails of the
// contributing columns -- through 32-bit row lists; here
//   Ls[0 .. nE)  : the entries of the bundle's columns (CSC order, local slot = global slot - Lp[s0])
//   Ds[0 .. nloc): the pivots
// live in LDS from the merge of the U rows (initial values) to ONE coalesced write of L at the end; the level loop
// reads only index data: Rp, the 16-bit row lists (Rk16 / Ro16), Lp of the contributing columns, Li16 of the tails.
// Same arithmetic per entry as k_bundle_factor (sums in the same order for thin columns).
// ---------------------------------------------------------------------------
constexpr int FLWG = 512;   // k_bundle_factor_lds
constexpr int FFWG = 1024;  // k_bundle_factor_flat: twice the threads take a level's records in half the passes (120 -> 90 us on config 3)
__device__ __forceinline__ double pivot_rule_local(const LdlView &v, int j, double d, double *dout) {
    const double sign = (double)v.dsigns[j];
    if (d * sign < v.reg_eps) {
        d = v.reg_delta * sign;
        atomicAdd(&v.status[2], 1); // rare
    }
    if (d == 0.0) v.status[1] = 1;
    const double dinv = 1.0 / d;
    if (!isfinite(dinv)) v.status[0] = 1;
    v.D[j] = d;
    v.Dinv[j] = dinv;
    *dout = d;
    return dinv;
}
__global__ __launch_bounds__(FLWG) void k_bundle_factor_lds(LdlView v, BundleView bv, FoldView fold, int lds_doubles) {
    extern __shared__ __attribute__((aligned(16))) char fl_smem[];
    __shared__ double red[16];
    __shared__ int fat[256];
    __shared__ int nfat;
    __shared__ double s_dinv;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int s0 = bv.bundle_ptr[b], s1 = bv.bundle_ptr[b + 1], nloc = s1 - s0;
    const int e0 = v.Lp[s0], nE = v.Lp[s1] - e0;
    double *Ls = (double *)fl_smem, *Ds = Ls + (lds_doubles - bv.max_nodes); // (Ds behind the largest bundle's entries)
    Ds = Ls + nE;
    const double eps = v.eps_ptr ? v.eps_ptr[0] : 0.0;
    // ---- initial values: U row j (diagonal first, then its entries to ancestors) merged into column j ----
    for (int j = s0 + tid; j < s1; j += FLWG) {
        const int cb = v.Lp[j] - e0, ce = v.Lp[j + 1] - e0;
        int u = v.Up[j];
        const int ue = v.Up[j + 1];
        const double dg = v.Ux[u];
        Ds[j - s0] = v.eps_ptr ? (v.dsigns[j] == 1 ? dg + eps : dg - eps) : dg;
        ++u;
        for (int q = cb; q < ce; ++q) {
            double val = 0.0;
            if (u < ue && v.Ucol16[u] == v.Li16[e0 + q]) val = v.Ux[u++];
            Ls[q] = val;
        }
    }
    __syncthreads();
    const int *lv = bv.blvl + bv.blvl_ptr[b];
    const int nl = bv.blvl_ptr[b + 1] - bv.blvl_ptr[b] - 1;
    // one contribution t of row j: column k (local), l_jk at local slot p; returns w = l_jk d_k and the tail range
    auto contribution = [&](int t, int &p, int &pe, double &ljk) {
        const int k = (int)v.Rk16[t];
        const int kb = v.Lp[s0 + k] - e0;
        pe = v.Lp[s0 + k + 1] - e0;
        p = kb + (int)v.Ro16[t];
        ljk = Ls[p];
        return ljk * Ds[k];
    };
    for (int l = 0; l < nl; ++l) {
        const int lb = lv[l], le = lv[l + 1];
        if (tid == 0) nfat = 0;
        __syncthreads();
        for (int j = lb + tid; j < le; j += FLWG) {
            const int rb = v.Rp[j], re = v.Rp[j + 1];
            const int cb = v.Lp[j] - e0, cn = v.Lp[j + 1] - e0 - cb;
            if (re - rb > FAC_THIN_ROW || cn > FAC_THIN_COL) {
                const int slot = atomicAdd(&nfat, 1);
                if (slot < 256) {
                    fat[slot] = j;
                    continue;
                } // (list full: serial path below, correct but slower)
            }
            double d = Ds[j - s0];
            if (cn <= 4) { // the bulk of block-arrow KKTs: row ids and running values in registers
                const int r0 = cn > 0 ? (int)v.Li16[e0 + cb] : -1, r1 = cn > 1 ? (int)v.Li16[e0 + cb + 1] : -1;
                const int r2 = cn > 2 ? (int)v.Li16[e0 + cb + 2] : -1;
                double a0 = cn > 0 ? Ls[cb] : 0.0, a1 = cn > 1 ? Ls[cb + 1] : 0.0, a2 = cn > 2 ? Ls[cb + 2] : 0.0,
                       a3 = cn > 3 ? Ls[cb + 3] : 0.0;
                for (int t = rb; t < re; ++t) {
                    int p, pe;
                    double ljk;
                    const double w = contribution(t, p, pe, ljk);
                    d -= ljk * w;
                    for (int pp = p + 1; pp < pe; ++pp) {
                        const int i = (int)v.Li16[e0 + pp];
                        const double uu = Ls[pp] * w;
                        if (i == r0) a0 -= uu;
                        else if (i == r1) a1 -= uu;
                        else if (i == r2) a2 -= uu;
                        else a3 -= uu;
                    }
                }
                double dd;
                const double dinv = pivot_rule_local(v, j, d, &dd);
                Ds[j - s0] = dd;
                if (cn > 0) Ls[cb] = a0 * dinv;
                if (cn > 1) Ls[cb + 1] = a1 * dinv;
                if (cn > 2) Ls[cb + 2] = a2 * dinv;
                if (cn > 3) Ls[cb + 3] = a3 * dinv;
            } else {
                for (int t = rb; t < re; ++t) {
                    int p, pe;
                    double ljk;
                    const double w = contribution(t, p, pe, ljk);
                    d -= ljk * w;
                    int q = cb;
                    for (int pp = p + 1; pp < pe; ++pp) {
                        const unsigned short i = v.Li16[e0 + pp];
                        while (v.Li16[e0 + q] != i) ++q; // rows below j of column k are a subset of column j
                        Ls[q] -= Ls[pp] * w;
                        ++q;
                    }
                }
                double dd;
                const double dinv = pivot_rule_local(v, j, d, &dd);
                Ds[j - s0] = dd;
                for (int q = cb; q < cb + cn; ++q) Ls[q] *= dinv;
            }
        }
        __syncthreads();
        // columns with many contributions (the separators at the top of a subtree): the whole workgroup on one column
        const int nf = min(nfat, 256);
        for (int f = 0; f < nf; ++f) {
            const int j = fat[f];
            const int rb = v.Rp[j], rn = v.Rp[j + 1] - rb;
            const int cb = v.Lp[j] - e0, cn = v.Lp[j + 1] - e0 - cb;
            double dpart = 0.0;
            if (cn <= 4) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                const int r0 = cn > 0 ? (int)v.Li16[e0 + cb] : -1, r1 = cn > 1 ? (int)v.Li16[e0 + cb + 1] : -1;
                const int r2 = cn > 2 ? (int)v.Li16[e0 + cb + 2] : -1;
                for (int t = tid; t < rn; t += FLWG) {
                    int p, pe;
                    double ljk;
                    const double w = contribution(rb + t, p, pe, ljk);
                    dpart += ljk * w;
                    for (int pp = p + 1; pp < pe; ++pp) {
                        const int i = (int)v.Li16[e0 + pp];
                        const double uu = Ls[pp] * w;
                        if (i == r0) a0 += uu;
                        else if (i == r1) a1 += uu;
                        else if (i == r2) a2 += uu;
                        else a3 += uu;
                    }
                }
                a0 = block_sum(a0, red);
                a1 = block_sum(a1, red);
                a2 = block_sum(a2, red);
                a3 = block_sum(a3, red);
                dpart = block_sum(dpart, red);
                if (tid == 0) {
                    double dd;
                    const double dinv = pivot_rule_local(v, j, Ds[j - s0] - dpart, &dd);
                    Ds[j - s0] = dd;
                    const double a[4] = {a0, a1, a2, a3};
                    for (int q = 0; q < cn; ++q) Ls[cb + q] = (Ls[cb + q] - a[q]) * dinv;
                }
            } else {
                for (int t = tid; t < rn; t += FLWG) {
                    int p, pe;
                    double ljk;
                    const double w = contribution(rb + t, p, pe, ljk);
                    dpart += ljk * w;
                    int q = cb;
                    for (int pp = p + 1; pp < pe; ++pp) {
                        const unsigned short i = v.Li16[e0 + pp];
                        int lo = q, hi = cb + cn; // first slot of column j with row >= i (present by construction)
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (v.Li16[e0 + mid] < i) lo = mid + 1;
                            else hi = mid;
                        }
                        q = lo;
                        atomicAdd(&Ls[q], -(Ls[pp] * w));
                        ++q;
                    }
                }
                dpart = block_sum(dpart, red);
                if (tid == 0) {
                    double dd;
                    s_dinv = pivot_rule_local(v, j, Ds[j - s0] - dpart, &dd);
                    Ds[j - s0] = dd;
                }
                __syncthreads();
                const double dinv = s_dinv;
                for (int q = cb + tid; q < cb + cn; q += FLWG) Ls[q] *= dinv;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (fold.k == 1) {
        // a single dense top row: d_t -= sum l_tc^2 d_c over this bundle's columns (l_tc is the LAST entry of a column
        // that reaches the top row); k_fold_top_pivot applies the pivot rule
        double sacc = 0.0;
        for (int j = s0 + tid; j < s1; j += FLWG) {
            const int ce = v.Lp[j + 1] - e0;
            if (ce > v.Lp[j] - e0 && (int)v.Li16[e0 + ce - 1] >= nloc) {
                const double lt = Ls[ce - 1];
                sacc += lt * (lt * Ds[j - s0]);
            }
        }
        sacc = block_sum(sacc, red);
        if (tid == 0 && sacc != 0.0) atomicAdd(&fold.acc[fold_acc_index(2, 0, b % FOLD_SLOTS)], sacc);
    }
    // ---- the factor's values, once, coalesced ----
    for (int q = tid; q < nE; q += FLWG) v.Lx[e0 + q] = Ls[q];
}
// ---------------------------------------------------------------------------
// ... and entry-parallel (right-looking): no pointer is chased inside the level loop.  When the columns of a level are
// final, every pair of entries of such a column updates one later entry or pivot -- the symbolic phase lists these
// updates as 8-byte records {slot a, slot b, column k, target}, one contiguous range per (bundle, level), sorted by
// target -- and the threads stride over the range: target -= l_a (l_b d_k), an LDS atomic (runs of one target reduced
// in registers first: the pivot of a separator column takes a thousand updates).  Per level: pivots + scaling of the
// level's columns (thread per column, LDS only), a barrier, the records, a barrier.  The initial values come from a
// flat pass over the bundle's U entries (fu_slot says where each lands).
// ---------------------------------------------------------------------------

// Round 6: everything the launch reads is REQUESTED UP FRONT -- the U entries with their landing slots, the first
// batch of update records, the column pointers and signs of the thread's own columns (CPT per thread, level-major numbering:
// thread t owns columns t, t + 1024, ...), the slotted maxima of the regulariser -- so the level loop works on LDS and
// registers only.  Before, a workgroup walked a chain of ~9 dependent round trips (U pass with a dependent load for the
// diagonal rows, per level the records and the columns' pointers, a closing pass over Lp -> Li16 for the top row's
// share): ~43 us per workgroup for 300 KB, two rounds of workgroups per launch.  The single folded top row's pivot
// share sum l_tc^2 d_c now arrives through the records as well (target nE + nloc, symbolic.cpp), no closing pass.
template <int CPT>
__global__ __launch_bounds__(FFWG) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_bundle_factor_flat(LdlView v, BundleView bv, FoldView fold) {
    extern __shared__ __attribute__((aligned(16))) char ff_smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int s0 = bv.bundle_ptr[b], s1 = bv.bundle_ptr[b + 1], nloc = s1 - s0;
    const int e0 = v.Lp[s0], nE = v.Lp[s1] - e0;
    double *Ls = (double *)ff_smem, *Ds = Ls + nE; // (contiguous: a record's target addresses either; Ds[nloc]: the top row's share)
    const int *lv = bv.blvl + bv.blvl_ptr[b];
    const int nl = bv.blvl_ptr[b + 1] - bv.blvl_ptr[b] - 1;
    const int *tp = v.fu_ptr + bv.blvl_ptr[b];
    typedef unsigned short fu_v4 __attribute__((ext_vector_type(4)));
    const fu_v4 *rec = (const fu_v4 *)v.fu_rec;
    constexpr int FU = 8; // records in flight per thread
    constexpr int UF = 8; // U entries per thread and pass
    int dbgn = 0;
    auto stamp = [&]() { // diagnostics (CHIP_IR_DEBUG=3): phase boundaries of every workgroup on the 100 MHz clock
        if (bv.fdbg && tid == 0 && dbgn < 31) {
            if (dbgn == 0)
                bv.fdbg[(size_t)b * 32] = (long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4) |
                                          ((long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 20) << 32);
            bv.fdbg[(size_t)b * 32 + 1 + dbgn++] = wall_clock64();
        }
    };
    stamp();
    // ---- requests: own columns, first pass of U entries, first batch of records ----
    // (packed: 64 registers per thread -- two workgroups per CU -- must hold all of it: column = first slot | length << 16,
    // signs as a bit mask, a U entry's slot | row << 16)
    unsigned ccol[CPT];
    unsigned csgm = 0;
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int j = tid + q * FFWG;
        const bool ok = j < nloc;
        const int cb = ok ? v.Lp[s0 + j] - e0 : 0, ce = ok ? v.Lp[s0 + j + 1] - e0 : 0;
        ccol[q] = (unsigned)cb | ((unsigned)(ce - cb) << 16);
        if (ok && v.dsigns[s0 + j] == 1) csgm |= 1u << q;
    }
    const int ub = v.Up[s0], ue = v.Up[s1];
    unsigned usr[UF];
    double uval[UF];
    auto request_u = [&](int base) {
#pragma unroll
        for (int q = 0; q < UF; ++q) {
            const int u = base + q * FFWG + tid;
            const bool ok = u < ue;
            usr[q] = ok ? ((unsigned)v.fu_slot[u] | ((unsigned)v.Urow16[u] << 16)) : 0xFFFEu; // (slot 0xFFFE: nothing)
            uval[q] = ok ? v.Ux[u] : 0.0;
        }
    };
    request_u(ub);
    const int rend = tp[nl];
    fu_v4 r[FU];
    auto request = [&](int base) {
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int t = base + u * FFWG + tid;
            if (t < rend) r[u] = rec[t];
            else r[u] = fu_v4{0, 0, 0, 0xFFFF};
        }
    };
    int base = tp[0];
    bool eps_on;
    __shared__ double s_eps;
    (void)static_eps(v, &eps_on, &s_eps);
    for (int q = tid; q <= nE + nloc; q += FFWG) Ls[q] = 0.0; // (fill-in slots stay zero; Ds[nloc] collects the top row's share)
    lds_barrier();
    stamp();
    const double eps = s_eps;
    // ---- initial values: the U entries to their slots ----
    for (int ubase = ub;;) {
#pragma unroll
        for (int q = 0; q < UF; ++q) {
            const unsigned slot = usr[q] & 0xFFFFu;
            if (slot == 0xFFFFu) Ds[usr[q] >> 16] = uval[q];
            else if (slot != 0xFFFEu) Ls[slot] = uval[q];
        }
        ubase += UF * FFWG;
        if (ubase >= ue) break;
        request_u(ubase);
    }
    request(base); // (the first batch of records: in flight under the regulariser pass and the first level's pivots)
    lds_barrier();
    stamp();
    if (eps_on) { // static regularisation of the diagonal (directldlkktsolver.rs:217-250), before any update lands on it
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int j = tid + q * FFWG;
            if (j < nloc) Ds[j] = ((csgm >> q) & 1u) ? Ds[j] + eps : Ds[j] - eps;
        }
    }
    for (int l = 0; l < nl; ++l) {
        // the level's columns are final: pivot rule (qdldl.rs:645-665), scale -- the thread's own columns of this level
        const int lb = lv[l] - s0, le = lv[l + 1] - s0;
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int j = tid + q * FFWG;
            if (j >= lb && j < le) {
                double d = Ds[j];
                const double sign = ((csgm >> q) & 1u) ? 1.0 : -1.0;
                if (d * sign < v.reg_eps) {
                    d = v.reg_delta * sign;
                    atomicAdd(&v.status[2], 1); // rare
                }
                if (d == 0.0) v.status[1] = 1;
                const double dinv = 1.0 / d;
                if (!isfinite(dinv)) v.status[0] = 1;
                Ds[j] = d; // (D and 1 / D leave in one coalesced pass at the end)
                const int cb = (int)(ccol[q] & 0xFFFFu), cn = (int)(ccol[q] >> 16);
                for (int t = cb; t < cb + cn; ++t) Ls[t] *= dinv;
            }
        }
        lds_barrier();
        stamp();
        const int rb = tp[l], re = tp[l + 1];
        for (;;) { // (wave-uniform bounds: lds_scatter_add is cross-lane)
            // (the level's records sit in the register slots ulo .. uhi of the batch: the others are skipped by a
            // uniform branch -- these phases are bound by the instruction count, 8 waves per SIMD)
            const int ulo = rb > base ? (rb - base) / FFWG : 0, uhi = (min(re, base + FU * FFWG) - 1 - base) / FFWG;
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                if (u < ulo || u > uhi) continue;
                const int t = base + u * FFWG + tid;
                const bool ok = t >= rb && t < re;
                const double val = ok ? Ls[r[u].x] * (Ls[r[u].y] * Ds[r[u].z]) : 0.0;
                lds_scatter_add(Ls, ok ? (int)r[u].w : -1, -val);
            }
            if (base + FU * FFWG >= re) break; // (the batch reaches into the next level: that level goes on with it)
            base += FU * FFWG;
            request(base);
        }
        if (base + FU * FFWG == re && re < rend) { // (the batch ended with the level: the next one, under the barrier)
            base += FU * FFWG;
            request(base);
        }
        lds_barrier();
        stamp();
    }
    if (fold.k == 1 && tid == 0) {
        // a single dense top row: d_t -= sum l_tc^2 d_c over this bundle's columns -- collected by the records of the
        // (top, top) pairs in Ds[nloc] (as its negative); k_fold_top_pivot applies the pivot rule
        const double sacc = -Ds[nloc];
        if (sacc != 0.0) atomicAdd(&fold.acc[fold_acc_index(2, 0, b % FOLD_SLOTS)], sacc);
    }
    for (int q = tid; q < nE; q += FFWG) v.Lx[e0 + q] = Ls[q];
    for (int j = tid; j < nloc; j += FFWG) {
        const double d = Ds[j];
        v.D[s0 + j] = d;
        v.Dinv[s0 + j] = 1.0 / d;
    }
    stamp();
    static_eps_epilogue(v, eps);
}
__global__ void k_fold_top_pivot(LdlView v, FoldView fold) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // the top column's initial pivot: as k_scatter_init left it in D, or -- fast preparation (fold.top_k) -- K's diagonal
    // entry shifted by the static regulariser the bundle factorisation published (LdlView::eps_out).
    // (Folding this into the bundle kernel as a last-arriver epilogue was measured: inlined it cost that kernel 40 %,
    // out of line 170 %, although one thread runs it once.)
    double d;
    if (fold.top_k) {
        d = fold.top_k[0];
        if (v.eps_out) d = fold.top_sign == 1 ? d + v.eps_out[0] : d - v.eps_out[0];
    } else {
        d = v.D[fold.NF];
    }
    for (int q = 0; q < FOLD_SLOTS; ++q) {
        double *a = &fold.acc[fold_acc_index(2, 0, q)];
        d -= *a;
        *a = 0.0;
    }
    (void)pivot_rule(v, fold.NF, d);
}

// B: a column with a huge row count (> 16384 contributions).  Each workgroup
// folds one chunk of contributions and meets the others in global fp64
// atomics on the column's slots / D[j]; k_factor_finalize then pivots+scales.
__global__ __launch_bounds__(WG) void k_factor_B(LdlView v, const int *__restrict__ crow,
                                                 const int *__restrict__ cbeg,
                                                 const int *__restrict__ cend, int count) {
    __shared__ double red[16];
    __shared__ double acc[B_LDS_CAP];
    __shared__ int rows[B_LDS_CAP];
    __shared__ int cst[RCAP];
    __shared__ double cw[RCAP];
    __shared__ int coff[RCAP + 1];
    if ((int)blockIdx.x >= count) return;
    const int j = crow[blockIdx.x];
    const int cb = v.Lp[j], ce = v.Lp[j + 1], cn = ce - cb;
    const int tb = cbeg[blockIdx.x], te = cend[blockIdx.x], tid = threadIdx.x;
    double dpart = 0.0;
    if (cn >= 24 && cn <= B_LDS_CAP) {
        // dense-front column: this workgroup folds its slice of the contributions into a private
        // LDS copy of the column (flattened (contribution, tail entry) pairs, as factor_col_block)
        // and meets the other slices with ONE global atomic per row at the end
        for (int q = tid; q < cn; q += WG) {
            acc[q] = 0.0;
            rows[q] = v.Li[cb + q];
        }
        for (int base = tb; base < te; base += RCAP) {
            const int nbt = min(RCAP, te - base);
            __syncthreads();
            for (int t = tid; t < nbt; t += WG) {
                const int k = v.Rcol[base + t], p = v.Rpos[base + t];
                const double ljk = v.Lx[p];
                const double w = ljk * v.D[k];
                dpart += ljk * w;
                cst[t] = p + 1;
                cw[t] = w;
                coff[t + 1] = v.Lp[k + 1] - (p + 1);
            }
            if (tid == 0) coff[0] = 0;
            __syncthreads();
            for (int off = 1; off < nbt; off <<= 1) { // inclusive scan of coff[1..nbt]
                int add0 = 0, add1 = 0;
                const int i0 = tid + 1, i1 = tid + 1 + WG;
                if (i0 <= nbt && i0 - off >= 1) add0 = coff[i0 - off];
                if (i1 <= nbt && i1 - off >= 1) add1 = coff[i1 - off];
                __syncthreads();
                if (i0 <= nbt) coff[i0] += add0;
                if (i1 <= nbt) coff[i1] += add1;
                __syncthreads();
            }
            const int total = coff[nbt];
            // four updates per thread in lockstep: the owner searches, then the four (row id, value)
            // loads, then the slot searches are independent chains -> 4x the loads in flight
            for (int u0 = tid; u0 < total; u0 += 4 * WG) {
                int own[4], pp[4], ri[4];
                double val[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int u = u0 + a * WG;
                    int lo = 0, hi = nbt;
                    if (u < total)
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (coff[mid] <= u) lo = mid;
                            else hi = mid;
                        }
                    own[a] = lo;
                    pp[a] = u < total ? cst[lo] + (u - coff[lo]) : -1;
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    ri[a] = pp[a] >= 0 ? v.Li[pp[a]] : 0;
                    val[a] = pp[a] >= 0 ? v.Lx[pp[a]] * cw[own[a]] : 0.0;
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (pp[a] < 0) continue;
                    int l2 = 0, h2 = cn;
                    while (l2 < h2) {
                        const int mid = (l2 + h2) >> 1;
                        if (rows[mid] < ri[a]) l2 = mid + 1;
                        else h2 = mid;
                    }
                    atomicAdd(&acc[l2], -val[a]);
                }
            }
        }
        __syncthreads();
        for (int q = tid; q < cn; q += WG)
            if (acc[q] != 0.0) atomicAdd(&v.Lx[cb + q], acc[q]);
    } else {
        for (int t = tb + tid; t < te; t += WG) {
            const int k = v.Rcol[t], p = v.Rpos[t];
            const double ljk = v.Lx[p];
            const double w = ljk * v.D[k];
            dpart += ljk * w;
            const int pe = v.Lp[k + 1];
            int q = cb;
            for (int pp = p + 1; pp < pe; ++pp) {
                q = find_row(v.Li, q, ce, v.Li[pp]);
                atomicAdd(&v.Lx[q], -(v.Lx[pp] * w));
                ++q;
            }
        }
    }
    dpart = block_sum(dpart, red);
    if (threadIdx.x == 0) atomicAdd(&v.D[j], -dpart);
}
__global__ __launch_bounds__(WG) void k_factor_finalize(LdlView v, const int *__restrict__ cols, int count) {
    __shared__ double s_dinv;
    if ((int)blockIdx.x >= count) return;
    const int j = cols[blockIdx.x];
    const int cb = v.Lp[j], ce = v.Lp[j + 1];
    if (threadIdx.x == 0) s_dinv = pivot_rule(v, j, v.D[j]);
    __syncthreads();
    const double dinv = s_dinv;
    for (int q = cb + threadIdx.x; q < ce; q += WG) {
        const double l = v.Lx[q] * dinv;
        v.Lx[q] = l;
        v.Rx[v.Tpos[q]] = l;
    }
}


} // namespace

void factor_T(hipStream_t s, const LdlView &v, ListView c) {
    if (c.count) k_factor_T<<<grid_for(c.count), WG, 0, s>>>(v, c.idx, c.count);
}
void factor_W(hipStream_t s, const LdlView &v, ListView c) {
    if (c.count) k_factor_W<<<c.count, 1024, 0, s>>>(v, c.idx, c.count);
}
void fold_top_pivot(hipStream_t s, const LdlView &v, const FoldView &fold) {
    if (fold.k == 1) k_fold_top_pivot<<<1, 64, 0, s>>>(v, fold);
}
static size_t factor_lds_bytes(int lds_doubles) { return ((size_t)lds_doubles * sizeof(double) + 15) & ~(size_t)15; }
bool bundle_factor_lds_ok(int lds_doubles) {
    if (lds_doubles <= 0) return false;
    const size_t lds = factor_lds_bytes(lds_doubles);
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void *)k_bundle_factor_lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (fa.sharedSizeBytes + lds > 80 * 1024 - 512) return false; // two workgroups per CU
    if (raise_dynamic_lds((const void *)k_bundle_factor_lds, (size_t)lds) != hipSuccess ||
        raise_dynamic_lds((const void *)k_bundle_factor_flat<4>, (size_t)lds) != hipSuccess ||
        raise_dynamic_lds((const void *)k_bundle_factor_flat<8>, (size_t)lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}
int bundle_factor(hipStream_t s, const LdlView &v, const BundleView &bv, const FoldView &fold, int lds_doubles) {
    if (!bv.nb) return 0;
    const bool no_flat = switches().no_factor_flat;
    if (lds_doubles > 0 && v.fu_rec && !no_flat && bv.max_nodes <= 8 * FFWG) {
        if (bv.max_nodes <= 4 * FFWG) k_bundle_factor_flat<4><<<bv.nb, FFWG, factor_lds_bytes(lds_doubles), s>>>(v, bv, fold);
        else k_bundle_factor_flat<8><<<bv.nb, FFWG, factor_lds_bytes(lds_doubles), s>>>(v, bv, fold);
    }
    else if (lds_doubles > 0 && v.Li16) k_bundle_factor_lds<<<bv.nb, FLWG, factor_lds_bytes(lds_doubles), s>>>(v, bv, fold, lds_doubles); // (needs the 16-bit row lists)
    else k_bundle_factor<<<bv.nb, BWG, 0, s>>>(v, bv, fold);
    return (int)hipGetLastError(); // (a rejected launch would leave the factor stale)
}
void gfold_top_factor(hipStream_t s, const LdlView &v, const BundleView &bv, const GFoldView &gf) {
    if (gf.ng <= 0) return;
    k_gfold_schur<<<bv.nb, GSWG, 0, s>>>(v, bv, gf);
    k_gfold_top_factor<<<gf.ng, 64, 0, s>>>(v, gf);
}
void factor_B(hipStream_t s, const LdlView &v, ChunkView c) {
    if (c.count) k_factor_B<<<c.count, WG, 0, s>>>(v, c.row, c.beg, c.end, c.count);
}
void factor_finalize(hipStream_t s, const LdlView &v, ListView c) {
    if (c.count) k_factor_finalize<<<c.count, WG, 0, s>>>(v, c.idx, c.count);
}
void factor_chain(hipStream_t s, const LdlView &v, const int *t_idx, const int *t_ptr, const int *w_idx,
                  const int *w_ptr, int l0, int l1) {
    if (l1 > l0) k_factor_chain<<<1, 1024, 0, s>>>(v, t_idx, t_ptr, w_idx, w_ptr, l0, l1);
}

} // namespace dev
} // namespace chip
