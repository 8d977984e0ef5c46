// kernels.hip -- hand-written CDNA4 (gfx950) kernels of the KKT hot path.
//
// The sparse part is HBM-bound gather/scatter over fp64 values and int32 indices
// (SURVEY.md 8d); the dense chain supernodes of the top use the f64 matrix cores
// (v_mfma_f64_16x16x4_f64, k_snode_*).  Design rules applied throughout
// (/opt/skills/guides/cdna_hip_programming.md):
//   * 64-wide wavefronts: wave reductions use 64-lane shuffles, workgroups
//     are 256 threads = 4 waves, "wave per row" kernels pack 4 rows per group;
//   * the elimination order is level-major (symbolic.cpp), so the thread-per-
//     row kernels of one level read D/Dinv/ptr/x over a contiguous index range
//     (coalesced) and the blockIdx -> slab map is XCD-aware: hardware block b
//     runs on XCD b%8, so logical block (b%8)*per + b/8 gives every XCD (own
//     L2) one contiguous slab of rows;
//   * the bottom of the elimination tree is cut into subtree bundles: ONE
//     workgroup factors / solves a bundle start to finish with its vector slice
//     in LDS and __syncthreads() between levels; only the remaining ancestors
//     (the "top") resolve dependencies by kernel boundaries (one launch per
//     level, ~1.5us each -- cheaper than any grid barrier on this part,
//     MI355X_MICROARCH.md price list), by single-workgroup chain kernels over
//     runs of narrow levels, or block by block with inverted diagonal blocks;
//   * rows / columns too heavy for one workgroup (the 10^6-entry budget row,
//     dense-front columns) are split in work-balanced chunks over many
//     workgroups whose partial results meet in one fp64 atomic per chunk or
//     per row, never one global atomic per entry.
//
// Reference semantics restated (citations relative to /root/reference/src):
//   numeric LDL' + pivot rule   qdldl/qdldl.rs:469-669  (rule :645-651)
//   L / D L' solves             qdldl/qdldl.rs:708-768
//   symv for refinement         algebra/csc/matrix_math.rs:178-208
//   cone scalings, Hs, step     solver/core/cones/{nonnegative,so,exp,pow,genpow,
//   operations, barriers        psdtriangle}cone.rs, symmetric_common.rs,
//                               nonsymmetric_common.rs, compositecone.rs
//   sparse gemv, dots, waxpby   algebra/csc/matrix_math.rs:258-343, vecmath.rs
#include "kernels.hpp"
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>

namespace chip {
namespace dev {

namespace {

constexpr int WG = 256;

// XCD-aware logical block id; grids are launched with a multiple of 8 blocks.
__device__ __forceinline__ int logical_block() {
    const int per = gridDim.x >> 3;
    return (blockIdx.x & 7) * per + (blockIdx.x >> 3);
}
inline int grid_for(int count) {
    int nb = (count + WG - 1) / WG;
    nb = (nb + 7) & ~7;
    return nb < 8 ? 8 : nb;
}

__device__ __forceinline__ double wave_sum_all(double v);
__device__ __forceinline__ double wave_max_all(double v);
// sum / max over the 64 lanes of a wavefront (every lane receives the result)
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_all(v); }
// The same sum by data-parallel-primitive moves inside the vector ALU (row_shr 1, 2, 4, 8 inside the rows of 16
// lanes, then row_bcast 15 / 31 across the rows: the classic gfx9 reduction) instead of six dependent trips
// through the LDS crossbar (ds_bpermute, what __shfl_down compiles to): ~80 instead of ~700 cycles.  The total
// is returned to EVERY lane.  Used where a reduction sits on the critical path of a sweep.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double v) { // value of the lane selected by CTRL, 0 where there is none
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_all(double v) {
    v += dpp_fetch<0x111, 0xf>(v); // row_shr:1
    v += dpp_fetch<0x112, 0xf>(v); // row_shr:2
    v += dpp_fetch<0x114, 0xf>(v); // row_shr:4
    v += dpp_fetch<0x118, 0xf>(v); // row_shr:8   -> lane 15 of every row holds the row's sum
    v += dpp_fetch<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
    v += dpp_fetch<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// (lanes without a source keep their own value: old = the lane's own value, bound_ctrl off)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch_self(double v) {
    const int l0 = __double2loint(v), h0 = __double2hiint(v);
    const int lo = __builtin_amdgcn_update_dpp(l0, l0, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(h0, h0, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max_all(double v) {
    v = fmax(v, dpp_fetch_self<0x111, 0xf>(v));
    v = fmax(v, dpp_fetch_self<0x112, 0xf>(v));
    v = fmax(v, dpp_fetch_self<0x114, 0xf>(v));
    v = fmax(v, dpp_fetch_self<0x118, 0xf>(v));
    v = fmax(v, dpp_fetch_self<0x142, 0xa>(v));
    v = fmax(v, dpp_fetch_self<0x143, 0xc>(v));
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v) { return wave_max_all(v); }
// the former reduction order (a butterfly through lane shuffles, lane 0 receives the result): kept for the
// vector algebra and the cone kernels of the caller's side, see the note above "vectors" below
__device__ __forceinline__ double wave_sum_tree(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max_tree(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ double block_sum_tree(double v, double *red) {
    v = wave_sum_tree(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double t = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}
__device__ __forceinline__ double block_max_tree(double v, double *red) {
    v = wave_max_tree(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double t = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = fmax(t, red[i]);
    return t;
}
// sum over a workgroup of up to 16 waves (red[16]), result broadcast to every thread
__device__ __forceinline__ double block_sum(double v, double *red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double t = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}
__device__ __forceinline__ double block_max(double v, double *red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double t = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = fmax(t, red[i]);
    return t;
}

// qdldl.rs:645-665: sign-based dynamic regularisation, then invert.
__device__ __forceinline__ double pivot_rule(const LdlView &v, int j, double d) {
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
    return dinv;
}

// ---------------------------------------------------------------------------
// value plumbing
// ---------------------------------------------------------------------------
// K.nzval (caller's order) -> initial values of the factorisation: off-diagonal
// entry (r,c) lands in its slot of column min(pr,pc) of L, a diagonal entry in
// D[.], optionally shifted by the static regulariser +-eps
// (directldlkktsolver.rs:233-245).  Lx is zeroed beforehand (fill-in slots).
__global__ __launch_bounds__(WG) void k_scatter_init(const double *__restrict__ Kx,
                                                     const int *__restrict__ a2l, int nnzK, int nnzL,
                                                     double *Lx, double *D,
                                                     const int8_t *__restrict__ dsigns,
                                                     const double *eps_ptr,
                                                     const int *__restrict__ fill_idx, int nfill,
                                                     int *status) {
    const double eps = eps_ptr ? eps_ptr[0] : 0.0;
    if (blockIdx.x == 0 && threadIdx.x < 4) status[threadIdx.x] = 0;
    for (int t = logical_block() * WG + threadIdx.x; t < nfill; t += gridDim.x * WG) Lx[fill_idx[t]] = 0.0;
    for (int t = logical_block() * WG + threadIdx.x; t < nnzK; t += gridDim.x * WG) {
        const int tgt = a2l[t];
        const double val = Kx[t];
        if (tgt >= nnzL) {
            const int j = tgt - nnzL;
            D[j] = eps_ptr ? (dsigns[j] == 1 ? val + eps : val - eps) : val;
        } else {
            Lx[tgt] = val;
        }
    }
}
__global__ __launch_bounds__(WG) void k_gather_values(double *__restrict__ Sx,
                                                      const double *__restrict__ Kx,
                                                      const int *__restrict__ Smap, int nnzS) {
    for (int t = logical_block() * WG + threadIdx.x; t < nnzS; t += gridDim.x * WG) Sx[t] = Kx[Smap[t]];
}
__global__ __launch_bounds__(WG) void k_scatter_values(double *Kx, const int *__restrict__ map,
                                                       const double *__restrict__ vals, int k,
                                                       double scale) {
    for (int t = blockIdx.x * WG + threadIdx.x; t < k; t += gridDim.x * WG) Kx[map[t]] = vals[t] * scale;
}
// max |K[diag]| -> bits in scal[1] (u64 compare is monotone for non-negative doubles);
// NaN propagates like vecmath.rs:132-142 through the flag in scal[2].
__global__ __launch_bounds__(WG) void k_diag_absmax(const double *__restrict__ Kx,
                                                    const int *__restrict__ didx, int N,
                                                    unsigned long long *scal) {
    __shared__ double red[16];
    double m = 0.0;
    bool nan = false;
    for (int t = blockIdx.x * WG + threadIdx.x; t < N; t += gridDim.x * WG) {
        const double a = Kx[didx[t]];
        if (a != a) nan = true;
        else m = fmax(m, fabs(a));
    }
    m = block_max(m, red);
    // one same-address atomic per workgroup serialises (~13 ns each): skip it when the running
    // maximum (a possibly stale read -- the maximum only grows) already covers this block
    if (threadIdx.x == 0) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
        if (bits > __hip_atomic_load(&scal[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&scal[1], bits);
    }
    if (nan) scal[2] = 1ull;
}
__global__ void k_eps_from_max(double c, double prop, unsigned long long *scal) {
    double m = __longlong_as_double((long long)scal[1]);
    if (scal[2]) m = __longlong_as_double(0x7ff8000000000000ll);
    ((double *)scal)[0] = c + prop * m; // directldlkktsolver.rs:324-329
}

// eps = c + prop * max|diag K| from the slotted maxima the cone kernels accumulated while they wrote
// their diagonal entries (no pass over the diagonal); clears the slots for the next update
__global__ void k_eps_from_slots(unsigned long long *slots, double c, double prop, double static_max,
                                 double *scal) {
    const int lane = threadIdx.x;
    unsigned long long *sl = slots + (size_t)lane * NRM_STRIDE;
    double m = lane < NRM_SLOTS ? __longlong_as_double((long long)*sl) : 0.0;
    if (lane < NRM_SLOTS) *sl = 0ull;
    m = wave_max(m);
    if (lane == 0) {
        m = fmax(m, static_max);
        int *nanflag = (int *)(slots + (size_t)NRM_SLOTS * NRM_STRIDE);
        if (*nanflag || static_max != static_max) m = __longlong_as_double(0x7ff8000000000000ll);
        *nanflag = 0;
        scal[0] = c + prop * m; // directldlkktsolver.rs:324-329
    }
}

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
constexpr int FAC_THIN_ROW = 8, FAC_THIN_COL = 48, THIN_MAX = 32;
constexpr int BWG = 512;      // bundle workgroup: 8 waves -> more loads in flight per subtree
constexpr int FATCAP = 1024;  // per-level list of rows/columns that need cooperative handling

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
__device__ __forceinline__ void lds_scatter_add(double *acc, int tgt, double val);
__global__ __launch_bounds__(FFWG) void k_bundle_factor_flat(LdlView v, BundleView bv, FoldView fold) {
    extern __shared__ __attribute__((aligned(16))) char ff_smem[];
    __shared__ double red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int s0 = bv.bundle_ptr[b], s1 = bv.bundle_ptr[b + 1], nloc = s1 - s0;
    const int e0 = v.Lp[s0], nE = v.Lp[s1] - e0;
    double *Ls = (double *)ff_smem, *Ds = Ls + nE; // (contiguous: a record's target addresses either)
    const double eps = v.eps_ptr ? v.eps_ptr[0] : 0.0;
    const int *lv = bv.blvl + bv.blvl_ptr[b];
    const int nl = bv.blvl_ptr[b + 1] - bv.blvl_ptr[b] - 1;
    const int *tp = v.fu_ptr + bv.blvl_ptr[b];
    for (int q = tid; q < nE; q += FFWG) Ls[q] = 0.0; // (fill-in slots stay zero)
    __syncthreads();
    {
        const int ub = v.Up[s0], ue = v.Up[s1];
        for (int u = ub + tid; u < ue; u += FFWG) {
            const unsigned short slot = v.fu_slot[u];
            const double val = v.Ux[u];
            if (slot == 0xFFFFu) {
                const int j = (int)v.Urow16[u];
                Ds[j] = v.eps_ptr ? (v.dsigns[s0 + j] == 1 ? val + eps : val - eps) : val;
            } else {
                Ls[slot] = val;
            }
        }
    }
    __syncthreads();
    typedef unsigned short fu_v4 __attribute__((ext_vector_type(4)));
    const fu_v4 *rec = (const fu_v4 *)v.fu_rec;
    constexpr int FU = 8; // records in flight per thread
    for (int l = 0; l < nl; ++l) {
        const int rb = tp[l], re = tp[l + 1];
        // the level's first records are requested BEFORE its columns are finalised: they are index data
        fu_v4 r[FU];
        auto request = [&](int base) {
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int t = base + u * FFWG + tid;
                if (t < re) r[u] = rec[t];
                else r[u] = fu_v4{0, 0, 0, 0xFFFF};
            }
        };
        request(rb);
        // the level's columns are final: pivot rule, scale
        for (int j = lv[l] + tid; j < lv[l + 1]; j += FFWG) {
            const int cb = v.Lp[j] - e0, ce = v.Lp[j + 1] - e0;
            double dd;
            const double dinv = pivot_rule_local(v, j, Ds[j - s0], &dd);
            Ds[j - s0] = dd;
            for (int q = cb; q < ce; ++q) Ls[q] *= dinv;
        }
        __syncthreads();
        for (int base = rb; base < re; base += FFWG * FU) { // (wave-uniform bounds: lds_scatter_add is cross-lane)
            if (base != rb) request(base);
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const bool ok = r[u].w != 0xFFFFu;
                const double val = ok ? Ls[r[u].x] * (Ls[r[u].y] * Ds[r[u].z]) : 0.0;
                lds_scatter_add(Ls, ok ? (int)r[u].w : -1, -val);
            }
        }
        __syncthreads();
    }
    if (fold.k == 1) {
        double sacc = 0.0;
        for (int j = s0 + tid; j < s1; j += FFWG) {
            const int ce = v.Lp[j + 1] - e0;
            if (ce > v.Lp[j] - e0 && (int)v.Li16[e0 + ce - 1] >= nloc) {
                const double lt = Ls[ce - 1];
                sacc += lt * (lt * Ds[j - s0]);
            }
        }
        sacc = block_sum(sacc, red);
        if (tid == 0 && sacc != 0.0) atomicAdd(&fold.acc[fold_acc_index(2, 0, b % FOLD_SLOTS)], sacc);
    }
    for (int q = tid; q < nE; q += FFWG) v.Lx[e0 + q] = Ls[q];
}
__global__ void k_fold_top_pivot(LdlView v, FoldView fold) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double d = v.D[fold.NF];
    for (int q = 0; q < FOLD_SLOTS; ++q) {
        double *a = &fold.acc[fold_acc_index(2, 0, q)];
        d -= *a;
        *a = 0.0;
    }
    (void)pivot_rule(v, fold.NF, d);
}

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

// ---------------------------------------------------------------------------
// Chain supernodes (host.hpp: Symbolic::sn_*): w columns c_0 < ... < c_{w-1}, each the parent of the
// previous one, all padded to the dense trapezoid  rows(c_t) = [c_{t+1}, ..., c_{w-1}, B...]  with
// B = struct(c_{w-1}), so that panel entry (i, t), i > t, is Lx[Lp[c_t] + i - t - 1] (panel rows i < w
// are the members, rows >= w the nb rows of B).  On entry Lx / D of the members hold K minus the
// contributions of all columns that are NOT supernode members (the sparse column kernels over the
// filtered row lists) and minus the dense updates of descendant supernodes (k_snode_extend).
// All supernodes of a unit level advance together, one block column of SN_NB at a time, kernel
// boundaries acting as the grid-wide synchronisation:
//   k_snode_update(b): A'[i, J_b] -= sum_{k < j0} L[i,k] d_k L[j,k] for all panel rows i >= j0 = 64 b,
//      left-looking, 16 x 64 tiles on the f64 matrix cores (v_mfma_f64_16x16x4_f64, four per A
//      operand); the (d_k L[j,k]) operand is staged in LDS SN_KC columns at a time, the L[i,k] operand is
//      streamed from the finished columns in 128-byte runs, SN_U requests in flight per lane;
//   k_snode_diag(b): the 64 x 64 diagonal block column by column with the sign-based dynamic
//      regularisation of qdldl.rs:645-665 in LDS;  k_snode_rows(b): the rows below it, one thread per
//      row, by forward substitution against the block;
//   k_snode_extend: once a supernode is complete, its update of the ANCESTORS' columns, the
//      nb x nb matrix L_B D L_B' (the multifrontal "update matrix"), computed with the same tiles
//      and subtracted at precomputed slots (upd_slot) with fp64 atomics.
// ~64 flops per streamed double instead of the ~1/8 of the per-entry gathers of the column kernels.
// ---------------------------------------------------------------------------
constexpr int SN_NB = 64;
constexpr int SN_KC = 128;  // 64 * 128 * 8 = 64 KiB
constexpr int SN_U = 4;     // k-groups of A operands in flight per lane (x 2 tiles; 8 needs more than 128 registers)
constexpr int SN_WST = 4;   // k rows (entry + pivot) in flight per thread while the LDS operand is staged
constexpr int SN_WG = 512;
constexpr int SN_ROWS = 256; // panel rows per workgroup of the update kernels: 8 waves x 2 tiles of 16
typedef double snode_v4d __attribute__((ext_vector_type(4)));
typedef double snode_v2d __attribute__((ext_vector_type(2)));

// broadcast of lane `src` (a compile-time constant after unrolling) without the LDS crossbar
__device__ __forceinline__ double readlane_f64(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                            __builtin_amdgcn_readlane(__double2loint(v), src));
}
// x <- (I + T)^-1 x (FWDMODE) or (I + T)^-T x for one SN_NB block held by ONE wave, lane = row.  Tt is the block
// in LDS with the LANE index fastest (forward: Tt[jj * SN_NB + row] = T(row, jj); backward: Tt[jj * SN_NB +
// row] = T(jj, row)), zero outside the strict triangle, so a narrow last block needs no bounds.  The 64
// coefficients of a lane do not depend on x: they are read up front (conflict-free), and the 64 dependent
// steps are a v_readlane + v_fma each (with __shfl through LDS and a 512-byte-stride read per step the same
// loop took ~3 us of a ~9 us pipeline stage of k_snode_tri).
template <bool FWDMODE> __device__ __forceinline__ double snode_block_solve(const double *Tt, double xv, int lane) {
    double t[SN_NB];
#pragma unroll
    for (int jj = 0; jj < SN_NB; ++jj) t[jj] = Tt[jj * SN_NB + lane];
    if (FWDMODE) {
#pragma unroll
        for (int jj = 0; jj < SN_NB - 1; ++jj) xv -= t[jj] * readlane_f64(xv, jj);
    } else {
#pragma unroll
        for (int jj = SN_NB - 1; jj > 0; --jj) xv -= t[jj] * readlane_f64(xv, jj);
    }
    return xv;
}

// CHIP_SN_DEBUG: wall-clock stamps (10 ns ticks) of ONE workgroup of a supernode launch at its phase boundaries;
// `drain` first waits for the loads in flight, so that a phase owns the latency of what it requested
__device__ __forceinline__ void sn_stamp(long long *dbg, bool me, int slot, bool drain = false) {
    if (!dbg) return;
    if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (me) dbg[slot] = wall_clock64();
}
struct SnodeGeom {
    const int *cols;
    const int *cb; // column bases of the panel (host-computed: Lp[cols[t]] - t - 1)
    double *d;     // pivots of the members, packed (k_snode_diag)
    const int8_t *sg; // signs of the members, packed
    int w, nb, h, e;
};
// The supernodes of a unit level come as RECORDS in level order (`order` points at the level's first record):
// (supernode id, first member p0, width w, last member column e, rows of B, -, -, -) -- one 32-byte read where the
// kernels of rounds 1-2 chased order -> sn -> sn_ptr / sn_geo (three dependent loads at the head of every launch).
constexpr int SN_REC = 8;
__device__ __forceinline__ SnodeGeom snode_geom(const SnodeView &sv, const int *__restrict__ order, int idx, int &sn) {
    typedef int rec_v4i __attribute__((ext_vector_type(4)));
    const rec_v4i r0 = *(const rec_v4i *)(order + SN_REC * idx);
    const int nb = order[SN_REC * idx + 4];
    SnodeGeom g;
    sn = r0.x;
    g.cols = sv.sn_col + r0.y;
    g.cb = sv.sn_cb + r0.y;
    g.d = sv.sn_d + r0.y;
    g.sg = sv.sn_sg + r0.y;
    g.w = r0.z;
    g.e = r0.w;
    g.nb = nb;
    g.h = g.w + g.nb;
    return g;
}

// acc[2][4] += L[rows of this wave's two tiles, k0..kend) * (d L[jrow0.., k])' ; emit per element.
// EXTEND = false: targets are the supernode's own block column (each element owned by one lane unless the k
// range is split: atomic_emit); true: the ancestors' columns through upd_slot (atomics).
//
// Round 3 (the launch ran at 0.17 of the f64 matrix peak, one workgroup per CU):
//  * 128 registers per lane, so that TWO workgroups share a CU (four waves per SIMD): one stages its (d L)'
//    operand while the other multiplies.  The A operands are no longer double buffered in registers -- the
//    entries of k-group g + 1 are requested INTO the registers of group g right after the matrix instructions
//    that read them, SN_U groups of loads in flight per lane;
//  * the A loads are unconditional: rows beyond the panel and columns beyond the chunk are clamped to valid
//    entries (their products meet stored zeros of the LDS operand, or rows that are never emitted) -- the
//    predicated form compiled to a branch and an LDS round trip per load;
//  * the result leaves through LDS, transposed: the matrix instruction leaves a lane with ONE column of the tile
//    (16 columns per instruction, 32 bytes of each cache line), the panel is column-major, so every emitted
//    instruction touched 16 lines -- now a lane owns a ROW, an instruction covers four columns x 16 consecutive
//    rows (128-byte runs), and the slot indices of the ancestor update are read the same way.
template <bool EXTEND>
__device__ __forceinline__ void snode_tiles(const LdlView &v, const SnodeView &sv, const SnodeGeom &g, int sn,
                                            const int *colbase, double *Wl, int jrow0, int ncols, int kend,
                                            int row_begin, int kbeg = 0, bool atomic_emit = false) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, l15 = lane & 15;
    const int i0[2] = {row_begin + wave * 32, row_begin + wave * 32 + 16};
    // (clamped: a row beyond the panel reads the last row, its results are not emitted)
    const int irc[2] = {min(i0[0] + l15, g.h - 1), min(i0[1] + l15, g.h - 1)};
    snode_v4d acc[2][SN_NB / 16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < SN_NB / 16; ++c) acc[t][c] = snode_v4d{0.0, 0.0, 0.0, 0.0};
    // the A operands are one stream of k-groups (4 SN_U columns each) over the whole k range, chunk boundaries
    // included: group g + 1 is requested into the registers of group g as they are released, unconditionally
    // (beyond the range: clamped, never used), so that the number of loads in flight is a constant the
    // compiler can count (a conditional request made it wait for ALL loads at the head of every group)
    double a[2][SN_U];
    const bool wave_live = i0[0] < g.h; // (wave uniform)
    auto request = [&](int u, int kabs) { // two independent 4 x 128-byte runs
        const int cb = colbase[min(kabs + 4 * u + kq, kend - 1)];
        a[0][u] = v.Lx[cb + irc[0]];
        a[1][u] = v.Lx[cb + irc[1]];
    };
    const bool dbgme = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0;
    sn_stamp(sv.dbg, dbgme, 16);
    __syncthreads(); // (the caller has just filled colbase)
    sn_stamp(sv.dbg, dbgme, 17, true);
    if (wave_live && kbeg < kend) {
#pragma unroll
        for (int u = 0; u < SN_U; ++u) request(u, kbeg);
    }
    for (int kc0 = kbeg; kc0 < kend; kc0 += SN_KC) {
        const int kcn = min(SN_KC, kend - kc0);
        const int kcnu = (kcn + 4 * SN_U - 1) / (4 * SN_U) * (4 * SN_U); // whole groups: the tail rows of the operand are zeros
        __syncthreads(); // the previous chunk has been consumed
        // the (d_k L[j,k]) operand: a wave stages whole k rows -- lane = column of the block --, SN_WST rows in
        // flight; the column base is a wave-uniform LDS read and the pivot a wave-uniform load from the packed
        // pivots (SnodeView::sn_d), issued together with the entry it scales: ONE global round trip per batch
        // (round 2: cols -> D -> LDS, a barrier, then the entries)
        for (int kr = wave; kr < kcnu; kr += SN_WST * (SN_WG / 64)) {
            double wv[SN_WST], dv[SN_WST];
#pragma unroll
            for (int r = 0; r < SN_WST; ++r) {
                const int kk = min(kr + r * (SN_WG / 64), kcn - 1); // (clamped: no branch per load)
                wv[r] = v.Lx[colbase[kc0 + kk] + jrow0 + min(lane, ncols - 1)];
                dv[r] = g.d[kc0 + kk];
            }
#pragma unroll
            for (int r = 0; r < SN_WST; ++r) {
                const int kk = kr + r * (SN_WG / 64);
                if (kk < kcnu) Wl[kk * SN_NB + lane] = (kk < kcn && lane < ncols) ? wv[r] * dv[r] : 0.0;
            }
        }
        __syncthreads();
        if (kc0 == kbeg) sn_stamp(sv.dbg, dbgme, 18);
        if (!wave_live) continue; // (after the barriers: the whole wave is beyond the panel)
        for (int kk = 0; kk < kcnu; kk += 4 * SN_U) {
            const int knext = kk + 4 * SN_U < kcnu ? kc0 + kk + 4 * SN_U : kc0 + SN_KC; // (the next chunk's first group)
#pragma unroll
            for (int u = 0; u < SN_U; ++u) {
                const int kl = kk + 4 * u + kq;
#pragma unroll
                for (int c = 0; c < SN_NB / 16; ++c) {
                    const double bw = Wl[kl * SN_NB + 16 * c + l15];
                    acc[0][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][u], bw, acc[0][c], 0, 0, 0);
                    acc[1][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][u], bw, acc[1][c], 0, 0, 0);
                }
                request(u, knext);
            }
        }
    }
    sn_stamp(sv.dbg, dbgme, 19);
    // ---- emit through LDS: this wave's 16 x 64 tile in its own 8 KiB of the (now free) operand buffer, element
    //      (row rr, column jj) at jj * 16 + (rr ^ (jj & 15)) -- the swizzle keeps both the column-per-lane writes
    //      and the row-per-lane reads off common banks
    __syncthreads();
    if (i0[0] >= g.h) return;
    double *Tw = Wl + wave * (16 * SN_NB);
    const int *Bn = v.Li + v.Lp[g.e]; // node ids of the rows of B
    const long long ubase = EXTEND ? sv.upd_ptr[sn] : 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t == 1) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < SN_NB / 16; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jj = l15 + 16 * c, rr = kq + 4 * r;
                Tw[jj * 16 + (rr ^ l15)] = acc[t][c][r];
            }
        __builtin_amdgcn_wave_barrier();
        const int i = i0[t] + l15; // lane = row of the tile; an instruction covers columns 4 m + kq
        if (i0[t] >= g.h) break;
        const int rB = i - g.w;
#pragma unroll
        for (int m0 = 0; m0 < 16; m0 += 8) { // (eight at a time: registers)
            double val[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int jj = 4 * (m0 + m) + kq;
                val[m] = Tw[jj * 16 + (l15 ^ (jj & 15))];
            }
            if (i >= g.h) continue;
            if (!EXTEND) {
                if (atomic_emit) { // split-k: several workgroups share the element
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int jj = 4 * (m0 + m) + kq, j = jrow0 + jj;
                        if (jj >= ncols) continue;
                        if (i > j) atomicAdd(&v.Lx[colbase[j] + i], -val[m]);
                        else if (i == j) atomicAdd(&v.D[g.cols[j]], -val[m]);
                    }
                } else {
                    double cur[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) { // (the eight reads together, then the eight writes)
                        const int jj = 4 * (m0 + m) + kq, j = jrow0 + jj;
                        cur[m] = (jj < ncols && i > j) ? v.Lx[colbase[j] + i] : 0.0;
                    }
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int jj = 4 * (m0 + m) + kq, j = jrow0 + jj;
                        if (jj >= ncols) continue;
                        if (i > j) v.Lx[colbase[j] + i] = cur[m] - val[m];
                        else if (i == j) v.D[g.cols[j]] -= val[m];
                    }
                }
            } else {
                int slot[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) { // (consecutive rows of one column: consecutive slots)
                    const int jj = 4 * (m0 + m) + kq, cB = jrow0 + jj - g.w;
                    const bool lower = jj < ncols && rB > cB;
                    slot[m] = lower ? sv.upd_slot[ubase + (long long)cB * g.nb - (long long)cB * (cB + 1) / 2 + (rB - cB - 1)] : -1;
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int jj = 4 * (m0 + m) + kq, cB = jrow0 + jj - g.w;
                    if (slot[m] >= 0) atomicAdd(&v.Lx[slot[m]], -val[m]);
                    else if (jj < ncols && rB == cB) atomicAdd(&v.D[Bn[cB]], -val[m]);
                }
            }
        }
    }
    sn_stamp(sv.dbg, dbgme, 20, true);
    if (sv.dbg && tid == 0) atomicMax((unsigned long long *)&sv.dbg[21], (unsigned long long)wall_clock64()); // last workgroup's end
}

__device__ __forceinline__ int *snode_lds(char *smem, double *&Wl) {
    Wl = (double *)smem;
    return (int *)(Wl + SN_KC * SN_NB);
}
// grid (row groups, supernodes of the level, k splits): with few workgroups in flight (the narrow
// levels near the root) the finished columns are divided among gridDim.z workgroups per tile group,
// which then meet in fp64 atomics
__global__ __launch_bounds__(SN_WG) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_snode_update(LdlView v, SnodeView sv, const int *__restrict__ order,
                                                        int b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Wl;
    int *colbase = snode_lds(smem, Wl);
    int sn;
    const SnodeGeom g = snode_geom(sv, order, (int)blockIdx.y, sn);
    const int j0 = b * SN_NB;
    if (j0 >= g.w) return;
    const int row_begin = j0 + (int)blockIdx.x * SN_ROWS;
    if (row_begin >= g.h) return;
    // this split's share of the k range, in units of one block column (j0 is a multiple of SN_NB)
    const int nunits = j0 / SN_NB, ns = (int)gridDim.z;
    const int c0 = (int)(((long long)nunits * blockIdx.z) / ns), c1 = (int)(((long long)nunits * (blockIdx.z + 1)) / ns);
    if (c0 >= c1) return;
    for (int t = threadIdx.x; t < g.w; t += SN_WG) colbase[t] = g.cb[t];
    snode_tiles<false>(v, sv, g, sn, colbase, Wl, j0, min(SN_NB, g.w - j0), c1 * SN_NB, row_begin, c0 * SN_NB, ns > 1);
}
// grid (row groups, column blocks of B, supernodes of the level)
__global__ __launch_bounds__(SN_WG) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_snode_extend(LdlView v, SnodeView sv, const int *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Wl;
    int *colbase = snode_lds(smem, Wl);
    int sn;
    const SnodeGeom g = snode_geom(sv, order, (int)blockIdx.z, sn);
    const int c0 = (int)blockIdx.y * SN_NB;
    if (c0 >= g.nb) return;
    const int row_begin = g.w + c0 + (int)blockIdx.x * SN_ROWS;
    if (row_begin >= g.h) return;
    for (int t = threadIdx.x; t < g.w; t += SN_WG) colbase[t] = g.cb[t];
    snode_tiles<true>(v, sv, g, sn, colbase, Wl, g.w + c0, min(SN_NB, g.nb - c0), g.w, row_begin);
}
// grid (supernodes of the level): the SN_NB x SN_NB diagonal block of block column b, right-looking, the
// block in REGISTERS: thread (row i = lane, column quarter q = wave) holds T[i][16 q .. 16 q + 15]; the loop
// over the 64 columns is fully unrolled, so every register index is a compile-time constant.  Per column ONE
// barrier: the wave that owns the column publishes it UNSCALED together with the pivot candidate (the running
// diagonal entry of that row) in LDS -- double buffered --, then every thread evaluates the pivot rule of
// qdldl.rs:645-665 itself, scales (l = c / d, as the reference: c * (1/d)) and applies the rank-1 update to
// its 16 entries and to its row's running diagonal.  (The previous version kept the block in LDS with two
// barriers and a div/mod-indexed trailing update per column: 91 us per block column; this step is the
// sequential part of every supernode's factorisation.)  Leaves the scaled block in Lx, (d, 1/d) in D / Dinv.
constexpr int SN_DWG = 256;
__global__ __launch_bounds__(SN_DWG) void k_snode_diag(LdlView v, SnodeView sv, const int *__restrict__ order,
                                                       int b) {
    __shared__ __attribute__((aligned(16))) double lcol[2][SN_NB];
    __shared__ double piv[2];
    __shared__ double sgn[SN_NB];
    __shared__ int colbase[SN_NB];
    int sn;
    const SnodeGeom g = snode_geom(sv, order, (int)blockIdx.x, sn);
    const int j0 = b * SN_NB;
    if (j0 >= g.w) return;
    const int nbw = min(SN_NB, g.w - j0), tid = threadIdx.x, i = tid & 63, q = tid >> 6;
    const bool live = i < nbw;
    if (tid < SN_NB) {
        colbase[tid] = tid < nbw ? g.cb[j0 + tid] : 0;
        sgn[tid] = tid < nbw ? (double)g.sg[j0 + tid] : 1.0; // (rows beyond a narrow last block: an identity)
    }
    __syncthreads();
    const int ci = live ? g.cols[j0 + i] : 0;
    double di = live ? v.D[ci] : 1.0; // running diagonal entry of row i (kept by all four threads of the row)
    double T[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const int j = 16 * q + cc;
        T[cc] = (live && j < nbw && i > j) ? v.Lx[colbase[j] + j0 + i] : 0.0;
    }
    double dfin = 1.0, dinvfin = 1.0; // pivot of row i, recorded by the threads of wave 0
    int nreg = 0, bad = 0;
    // column jj = 16 qq + c: the quarter loop stays rolled, the 16 columns of a quarter are unrolled, so T[c]
    // is a fixed register.  The owner publishes the strictly-lower part of the column (zeros from the
    // diagonal up) and the pivot candidate separately: every product below is then unconditional -- the 16
    // column entries a thread needs come in as eight 16-byte LDS reads, no per-entry branches (the first
    // version's `j2 > jj ? lcol[j2] : 0` compiled to 16 serialised conditional LDS round trips per column,
    // 53 of its 62 us).
    for (int qq = 0; qq < SN_NB / 16; ++qq) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int jj = 16 * qq + c, buf = c & 1;
            if (q == qq) {
                lcol[buf][i] = i > jj ? T[c] : 0.0;
                if (i == jj) piv[buf] = di;
            }
            __syncthreads();
            double d = piv[buf];
            const double sg = sgn[jj];
            const bool reg = d * sg < v.reg_eps;
            if (reg) d = v.reg_delta * sg;
            const double dinv = 1.0 / d;
            if (q == 0 && i == jj) {
                dfin = d;
                dinvfin = dinv;
                if (reg) nreg = 1;
                if (d == 0.0) bad |= 2;
                if (!isfinite(dinv)) bad |= 1;
            }
            const double l = lcol[buf][i] * dinv; // 0 for i <= jj
            if (q == qq) T[c] = i > jj ? l : T[c];
            const double w = l * d;
            di -= w * l;
            const snode_v2d *lc = (const snode_v2d *)&lcol[buf][16 * q];
#pragma unroll
            for (int c2 = 0; c2 < 8; ++c2) {
                const snode_v2d pr = lc[c2];
                T[2 * c2] -= w * (pr.x * dinv);
                T[2 * c2 + 1] -= w * (pr.y * dinv);
            }
        }
    }
    if (q == 0 && live) {
        v.D[ci] = dfin;
        g.d[j0 + i] = dfin;
        v.Dinv[ci] = dinvfin;
        if (nreg) atomicAdd(&v.status[2], 1);
        if (bad & 2) v.status[1] = 1;
        if (bad & 1) v.status[0] = 1;
    }
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const int j = 16 * q + cc;
        if (live && j < nbw && i > j) v.Lx[colbase[j] + j0 + i] = T[cc];
    }
}
// grid (row groups of SN_DWG rows, supernodes of the level): the rows below the diagonal block of
// block column b, one thread per row, forward substitution against the (finished) block
// (one WAVE per workgroup: the 2016 products of a row each read a coefficient from LDS -- broadcast reads, bound
// by the LDS issue rate of the CU -- so 64 rows per CU over many CUs beat 256 rows on a few)
constexpr int SN_RWG = 64;
__global__ __launch_bounds__(SN_RWG) void k_snode_rows(LdlView v, SnodeView sv, const int *__restrict__ order,
                                                       int b) {
    __shared__ __attribute__((aligned(16))) double dT[SN_NB * SN_NB];
    __shared__ double dinvl[SN_NB], dl[SN_NB];
    __shared__ int colbase[SN_NB];
    int sn;
    const SnodeGeom g = snode_geom(sv, order, (int)blockIdx.y, sn);
    const int j0 = b * SN_NB;
    if (j0 >= g.w) return;
    const int nbw = min(SN_NB, g.w - j0), tid = threadIdx.x;
    const int i = j0 + nbw + (int)blockIdx.x * SN_RWG + tid;
    if (j0 + nbw + (int)blockIdx.x * SN_RWG >= g.h) return;
    {
        const bool in = tid < nbw; // (a narrow last block is padded with an identity)
        const int c = in ? g.cols[j0 + tid] : 0;
        colbase[tid] = in ? g.cb[j0 + tid] : 0;
        dinvl[tid] = in ? v.Dinv[c] : 1.0;
        dl[tid] = in ? v.D[c] : 0.0;
    }
    __syncthreads();
    // dTt[q * SN_NB + jj] = d_q L_JJ(jj, q) for jj > q, else 0 (column q of the block contiguous); the pivots come
    // from LDS (v.D[g.cols[..]] inside this loop was a chain of two dependent global loads per round)
    for (int base = 0; base < SN_NB * SN_NB; base += 32 * SN_RWG) {
        double tv[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int idx = base + tid + r * SN_RWG, q = idx / SN_NB, jj = idx % SN_NB;
            tv[r] = (jj > q && jj < nbw) ? v.Lx[colbase[q] + j0 + jj] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int idx = base + tid + r * SN_RWG;
            dT[idx] = tv[r] * dl[idx / SN_NB];
        }
    }
    double x[SN_NB];
    const bool rowok = i < g.h;
    const int ic = rowok ? i : g.h - 1; // (unconditional loads from a valid row: no branch per entry)
#pragma unroll
    for (int jj = 0; jj < SN_NB; ++jj) x[jj] = v.Lx[colbase[jj < nbw ? jj : 0] + ic];
#pragma unroll
    for (int jj = 0; jj < SN_NB; ++jj) x[jj] = jj < nbw ? x[jj] : 0.0;
    __syncthreads();
    if (!rowok) return;
    // right-looking: x_q is final after the updates of columns < q; it then updates every later entry of the
    // row.  The products of one q are independent (the left-looking form chained 2016 FMAs on one accumulator
    // behind serialised LDS reads: 48 us); per entry the subtractions still happen in the order q = 0, 1, ...
    // of qdldl.rs:708-719.  Coefficients come in as 16-byte pairs (jj even, jj + 1); the pair that straddles q
    // multiplies a stored zero.
#pragma unroll
    for (int q = 0; q < SN_NB; ++q) {
        const double xq = x[q] * dinvl[q];
        x[q] = xq;
        // (an opaque zero that "depends" on x_q ties this column's LDS reads to its place in the chain: left
        // alone the compiler reads all 1024 coefficient pairs before the first product and spills them)
        int zoff;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zoff) : "v"(__double2hiint(xq)));
        const snode_v2d *cf = (const snode_v2d *)&dT[q * SN_NB + zoff];
#pragma unroll
        for (int p2 = (q + 1) / 2; p2 < SN_NB / 2; ++p2) {
            const snode_v2d cc = cf[p2];
            x[2 * p2] -= xq * cc.x;
            x[2 * p2 + 1] -= xq * cc.y;
        }
    }
#pragma unroll
    for (int jj = 0; jj < SN_NB; ++jj)
        if (jj < nbw) v.Lx[colbase[jj] + i] = x[jj];
    // (no row-major mirror Rx for supernode columns: with supernodes the forward sweep of the top reads
    // the filtered lists Rfx -- non-member columns only -- and k_snode_fwd reads Lx itself)
}

// ---------------------------------------------------------------------------
// Block column b of every supernode of a unit level in ONE launch (round 3; k_snode_diag + k_snode_rows were
// two launches of ~28 + ~33 us whatever the size, most of it latency: 64 pivots one barrier apart, then the
// coefficient block fetched again by every workgroup of the rows kernel).  Grid (groups of SNP_WG panel rows
// below the block, supernodes), SNP_WG = 256 threads:
//   1. EVERY workgroup factors the 64 x 64 diagonal block itself (redundantly: latency, not throughput, is what
//      counts here; only group 0 writes it back).  Thread (row i = lane, quarter q = wave) holds the 16 entries
//      (i, 16 q ..) of its row, unscaled (u), and a copy of the row's running diagonal.  The columns go in
//      groups of SNP_CB = 4: the wave that owns a group factors its four columns by itself -- pivots and the
//      six l(c', c) it needs across lanes by v_readlane, no barrier --, publishes their unscaled entries u
//      (double buffered) and scaled entries l (Ll, which stays) plus the pivots in LDS; after ONE barrier
//      every thread applies the four columns to its entries right of the group: 16 barriers per block instead
//      of 64.  Per entry the subtractions are those of the reference's row solve (qdldl.rs:610-640), in its
//      order: u_ic -= l_ck * u_ik for k = 0, 1, ... (the SCALED entry of row c times the UNSCALED entry of the
//      own row); d_i = a_ii - sum_k u_ik l_ik (qdldl.rs:634); the sign rule of qdldl.rs:645-665 by all lanes;
//   2. then its SNP_WG rows below the block, one row per thread, by the same recurrence against Ll (u_Rq is
//      final after the columns < q; l_Rq = u_Rq / d_q on the way out).
// ---------------------------------------------------------------------------
constexpr int SNP_WG = 256;
constexpr int SNP_CB = 4;
constexpr int SNP_XLD = 17; // row stride of a wave's head block in LDS (odd: lanes = rows hit distinct banks)
__global__ __launch_bounds__(SNP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_snode_panel(LdlView v, SnodeView sv, const int *__restrict__ order, int b,
                                                                                                  int rows_mfma) {
    __shared__ __attribute__((aligned(16))) double Ll[SN_NB * SN_NB];        // Ll[k * 64 + i] = l(i, k), 0 for i <= k
    __shared__ __attribute__((aligned(16))) double ucol[2][SNP_CB][SN_NB];   // unscaled entries of a group's columns (0 for i <= c)
    __shared__ double xh[SNP_WG / 64][64 * SNP_XLD];                       // rows phase (matrix-core form): a wave's head block, [row][column]
    __shared__ double dinvl[SN_NB], sgn[SN_NB];
    __shared__ int colbase[SN_NB];
    __shared__ int s_nreg, s_bad;
    int sn;
    const SnodeGeom g = snode_geom(sv, order, (int)blockIdx.y, sn);
    const int j0 = b * SN_NB;
    if (j0 >= g.w) return;
    const int nbw = min(SN_NB, g.w - j0), tid = threadIdx.x, i = tid & 63, q = tid >> 6;
    const bool dbgme = blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
    sn_stamp(sv.dbg, dbgme, 0);
    const int row0 = j0 + nbw + (int)blockIdx.x * SNP_WG; // first of this group's rows below the block
    if (blockIdx.x > 0 && row0 >= g.h) return;
    const bool live = i < nbw;
    if (tid < SN_NB) {
        colbase[tid] = g.cb[j0 + min(tid, nbw - 1)];
        sgn[tid] = tid < nbw ? (double)g.sg[j0 + tid] : 1.0; // (columns beyond a narrow last block: an identity)
    }
    if (tid == 0) {
        s_nreg = 0;
        s_bad = 0;
    }
    __syncthreads();
    sn_stamp(sv.dbg, dbgme, 1);
    const int ci = live ? g.cols[j0 + i] : 0;
    double di = live ? v.D[ci] : 1.0; // running diagonal entry of row i (kept by all four threads of the row)
    const double sgl = sgn[i];        // lane j: sign of column j (read by v_readlane: an LDS read per pivot sat on the chain)
    double T[16];
    double dfin = 1.0, dinvfin = 1.0; // pivot of row i, recorded by the owner of column i
    int nreg = 0, bad = 0;
    if (rows_mfma & 2) {
        // ---- block factorisation, matrix-core form (round 3): wave q owns column quarter q of the 64 x 64 block, ALL
        //      64 rows of it, in the accumulator layout (lane = column 16 q + l15, registers = rows: 4 tiles x 4).
        //      Quarter kb is factored by its owner alone in the lane = row form (through the wave's LDS slice): the
        //      diagonal entries live IN the block (entry (c, c) is the pivot candidate), 16 pivots with the
        //      in-quarter updates by v_readlane; it publishes the unscaled panel (negated, [row][k]) and the scaled
        //      one (Ll); after ONE barrier the quarters behind it subtract (64 x 16) x (16 x 16) on the matrix cores
        //      (16 instructions per wave).  Four barriers per block (sixteen before, sixty-four in round 2).
        const int l15 = i & 15, kq = i >> 4;
        double *xw = xh[q];
        snode_v4d Aq[4];
        {
            const int j = 16 * q + l15; // this lane's column
            const bool jok = j < nbw;
            const int cb = colbase[j];
            const double djj = jok ? v.D[g.cols[j0 + min(j, nbw - 1)]] : 1.0;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * t + kq + 4 * r;
                    double val = (jok && row < nbw && row > j) ? v.Lx[cb + j0 + row] : 0.0;
                    if (row == j) val = djj;
                    Aq[t][r] = val;
                }
        }
        sn_stamp(sv.dbg, dbgme, 2, true);
#pragma unroll 1
        for (int kb = 0; kb < SN_NB / 16; ++kb) {
            if (q == kb) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xw[(16 * t + kq + 4 * r) * SNP_XLD + l15] = Aq[t][r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int c = 0; c < 16; ++c) T[c] = xw[i * SNP_XLD + c];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int c = 16 * kb + t;
                    double d = readlane_f64(T[t], c); // entry (c, c): (c is wave uniform)
                    const double sg = readlane_f64(sgl, c);
                    const bool reg = d * sg < v.reg_eps;
                    if (reg) d = v.reg_delta * sg;
                    const double dinv = 1.0 / d;
                    if (i == c) {
                        dfin = d;
                        dinvfin = dinv;
                        if (reg) nreg = 1;
                        if (d == 0.0) bad |= 2;
                        if (!isfinite(dinv)) bad |= 1;
                    }
                    const double uc = i > c ? T[t] : 0.0;
                    const double l = uc * dinv;
                    T[t] = l;
                    xw[i * SNP_XLD + t] = -uc;
                    Ll[c * SN_NB + i] = l;
                    if (i == 0) dinvl[c] = dinv;
#pragma unroll
                    for (int t2 = t + 1; t2 < 16; ++t2) // entry (i, c2) -= l(c2, c) u(i, c), the diagonal entries included
                        T[t2] -= readlane_f64(l, 16 * kb + t2) * uc;
                }
            }
            __syncthreads();
            if (q > kb) {
                const double *xo = xh[kb];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const double bv = Ll[(16 * kb + 4 * s4 + kq) * SN_NB + 16 * q + l15]; // B[k][n] = l(16 q + n, 16 kb + k)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double av = xo[(16 * t + l15) * SNP_XLD + 4 * s4 + kq]; // A[m][k] = -u(16 t + m, 16 kb + k)
                        Aq[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, Aq[t], 0, 0, 0);
                    }
                }
            }
        }
    } else {
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const int j = 16 * q + cc;
        T[cc] = (live && j < nbw && i > j) ? v.Lx[colbase[j] + j0 + i] : 0.0;
    }
    sn_stamp(sv.dbg, dbgme, 2, true);
    // the quarter loop stays rolled; the 16 columns of a quarter are unrolled, so T[..] are fixed registers
    for (int qq = 0; qq < SN_NB / 16; ++qq) {
#pragma unroll
        for (int gc = 0; gc < 16; gc += SNP_CB) {
            const int c0 = 16 * qq + gc, buf = (gc / SNP_CB) & 1;
            if (q == qq) { // the owner: the group's columns among themselves
#pragma unroll
                for (int t = 0; t < SNP_CB; ++t) {
                    const int c = c0 + t;
                    double d = readlane_f64(di, c); // (c is wave uniform)
                    const double sg = readlane_f64(sgl, c);
                    const bool reg = d * sg < v.reg_eps;
                    if (reg) d = v.reg_delta * sg;
                    const double dinv = 1.0 / d;
                    if (i == c) {
                        dfin = d;
                        dinvfin = dinv;
                        if (reg) nreg = 1;
                        if (d == 0.0) bad |= 2;
                        if (!isfinite(dinv)) bad |= 1;
                    }
                    const double uc = i > c ? T[gc + t] : 0.0;
                    const double l = uc * dinv;
                    di -= uc * l;
                    T[gc + t] = i > c ? l : T[gc + t];
                    ucol[buf][t][i] = uc;
                    Ll[c * SN_NB + i] = l;
                    if (i == 0) dinvl[c] = dinv;
#pragma unroll
                    for (int t2 = t + 1; t2 < SNP_CB; ++t2) // entry (i, c0 + t2) -= l(c0 + t2, c) u(i, c)
                        T[gc + t2] -= readlane_f64(l, c0 + t2) * uc;
                }
            }
            __syncthreads();
            // everybody: the group's columns applied to the own entries right of the group (and to the copy of the
            // running diagonal, which the owner has already updated)
#pragma unroll
            for (int t = 0; t < SNP_CB; ++t) {
                const int c = c0 + t;
                const double uc = ucol[buf][t][i];
                if (q != qq) di -= uc * Ll[c * SN_NB + i];
                if (q >= qq) { // (wave uniform; earlier quarters are finished)
                    const snode_v2d *lr = (const snode_v2d *)&Ll[c * SN_NB + 16 * q];
#pragma unroll
                    for (int c2 = 0; c2 < 8; ++c2) {
                        const snode_v2d pr = lr[c2]; // l(16 q + 2 c2, c), l(16 q + 2 c2 + 1, c): zero up to the diagonal
                        if (q > qq || 2 * c2 >= gc + SNP_CB) T[2 * c2] -= pr.x * uc;
                        if (q > qq || 2 * c2 + 1 >= gc + SNP_CB) T[2 * c2 + 1] -= pr.y * uc;
                    }
                }
            }
        }
    }
    } // (round-2 form of the block factorisation)
    sn_stamp(sv.dbg, dbgme, 3);
    if (nreg) atomicAdd(&s_nreg, 1);
    if (bad) atomicOr(&s_bad, bad);
    __syncthreads(); // (also: Ll and dinvl are complete, and every entry of the unfactored block has been consumed)
    // The factored block goes back IN PLACE, and the workgroups of a large launch do not all run at the same time:
    // a workgroup that starts late must still find the UNFACTORED block.  So the block is written by the last of
    // the supernode's workgroups to get here -- all of them hold the same result -- which needs no waiting: a
    // counter per supernode, reset by the one that finds it complete.
    if (tid == 0) {
        const int nwg = max(1, (g.h - (j0 + nbw) + SNP_WG - 1) / SNP_WG); // workgroups of this supernode that got past the early return
        const int old = __hip_atomic_fetch_add(&sv.sn_cnt[sn], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_nreg = old == nwg - 1 ? (s_nreg | 0x40000000) : s_nreg;
        if (old == nwg - 1) __hip_atomic_store(&sv.sn_cnt[sn], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool writer = (s_nreg & 0x40000000) != 0;
    if (writer) {
        if (q == i / 16 && live) { // the owner of column i
            v.D[ci] = dfin;
            v.Dinv[ci] = dinvfin;
            g.d[j0 + i] = dfin;
        }
        if (tid == 0) {
            if (s_nreg & 0xffff) atomicAdd(&v.status[2], s_nreg & 0xffff);
            if (s_bad & 2) v.status[1] = 1;
            if (s_bad & 1) v.status[0] = 1;
        }
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const int j = 16 * q + cc;
            if (live && j < nbw && i > j) v.Lx[colbase[j] + j0 + i] = T[cc];
        }
    }
    sn_stamp(sv.dbg, dbgme, 4, true);
    if (row0 >= g.h) return;
    if (rows_mfma & 1) {
        // ---- the rows below the block, blocked by 16 columns (round 3): a wave owns 64 rows.  Block kb of every row is
        //      finished by the recurrence in the lane = row form (120 products per row instead of 2016), and its effect
        //      on the blocks behind it is a (64 x 16) x (16 x 16) product on the f64 matrix cores -- 96 instructions per
        //      wave for all six block pairs.  The blocks behind the head live in the accumulator layout (lane = column,
        //      registers = rows), the head block crosses between the two forms through the wave's own LDS slice.
        //      Per entry the subtractions still go k = 0, 1, ...; inside a matrix instruction they are fused
        //      multiply-adds (as in the update tiles).
        const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
        double *xw = xh[wave];
        const int R0w = row0 + 64 * wave;         // first row of the wave
        const int R = R0w + lane;                  // lane = row form
        const bool rowok = R < g.h;
        const int Rc = min(R, g.h - 1);
        if (R0w >= g.h) return;                   // (whole wave beyond the panel; no workgroup barrier below)
        // blocks 1..3 in the accumulator layout: acc[jb - 1][t][r] = X[row 16 t + kq + 4 r][column 16 jb + l15]
        snode_v4d acc[3][4];
#pragma unroll
        for (int jb = 1; jb < 4; ++jb) {
            const int jj = 16 * jb + l15;
            const int cb = colbase[jj];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = min(R0w + 16 * t + kq + 4 * r, g.h - 1);
                    const double xv = v.Lx[cb + row];
                    acc[jb - 1][t][r] = jj < nbw ? xv : 0.0;
                }
        }
        double h[16]; // the head block of the own row, lane = row form
#pragma unroll
        for (int c = 0; c < 16; ++c) h[c] = v.Lx[colbase[c] + Rc];
#pragma unroll
        for (int c = 0; c < 16; ++c) h[c] = c < nbw ? h[c] : 0.0;
        sn_stamp(sv.dbg, dbgme, 5, true);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kb > 0) { // the head block leaves the accumulator layout: [row][column] in LDS, then a row per lane
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xw[(16 * t + kq + 4 * r) * SNP_XLD + l15] = acc[kb - 1][t][r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int c = 0; c < 16; ++c) h[c] = xw[lane * SNP_XLD + c];
                __builtin_amdgcn_wave_barrier();
            }
            // the block's own triangle: u_c -= l(16 kb + c, 16 kb + kk) u_kk
            const double *Lb = Ll + (16 * kb) * SN_NB + 16 * kb;
#pragma unroll
            for (int kk = 0; kk < 15; ++kk) {
                const double uq = h[kk];
                int zoff; // (ties the column's LDS reads to its place in the chain, see below)
                asm volatile("v_mov_b32 %0, 0" : "=v"(zoff) : "v"(__double2hiint(uq)));
                const snode_v2d *cf = (const snode_v2d *)(Lb + kk * SN_NB + zoff);
#pragma unroll
                for (int p2 = (kk + 1) / 2; p2 < 8; ++p2) {
                    const snode_v2d cc = cf[p2]; // (the pair that straddles kk meets a stored zero)
                    h[2 * p2] -= cc.x * uq;
                    h[2 * p2 + 1] -= cc.y * uq;
                }
            }
            if (rowok) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const int jj = 16 * kb + c;
                    if (jj < nbw) v.Lx[colbase[jj] + R] = h[c] * dinvl[jj];
                }
            }
            if (kb == 3) break;
            // the finished block, negated, as the A operand of the products: [row][column] in LDS
#pragma unroll
            for (int c = 0; c < 16; ++c) xw[lane * SNP_XLD + c] = -h[c];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                double a4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) a4[t] = xw[(16 * t + l15) * SNP_XLD + 4 * s4 + kq]; // A[m = l15][k = 4 s4 + kq] of tile t
#pragma unroll
                for (int jb = kb + 1; jb < 4; ++jb) {
                    const double bv = Ll[(16 * kb + 4 * s4 + kq) * SN_NB + 16 * jb + l15]; // B[k][n] = l(16 jb + n, 16 kb + k)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[jb - 1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[t], bv, acc[jb - 1][t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        sn_stamp(sv.dbg, dbgme, 6);
        sn_stamp(sv.dbg, dbgme, 7, true);
        if (sv.dbg && lane == 0) atomicMax((unsigned long long *)&sv.dbg[8], (unsigned long long)wall_clock64());
        return;
    }
    // ---- the rows below the block: thread = row, its 64 entries in registers, right-looking (the products of one
    //      column are independent; per entry the subtractions happen in the order k = 0, 1, ... of qdldl.rs:610-640).
    //      (Measured and dropped: the same recurrence blocked by 16 columns with one rolled code body -- 30 KB of
    //      code instead of 75 KB --: 57.5 us per launch against 55; the launch is not bound by instruction fetch.)
    const int R = row0 + tid;
    const bool rowok = R < g.h;
    const int Rc = rowok ? R : g.h - 1; // (unconditional loads from a valid row)
    double x[SN_NB];
#pragma unroll
    for (int jj = 0; jj < SN_NB; ++jj) x[jj] = v.Lx[colbase[jj] + Rc];
#pragma unroll
    for (int jj = 0; jj < SN_NB; ++jj) x[jj] = jj < nbw ? x[jj] : 0.0;
    sn_stamp(sv.dbg, dbgme, 5, true);
#pragma unroll
    for (int k = 0; k < SN_NB; ++k) {
        const double uq = x[k];
        // (an opaque zero that "depends" on u_q ties this column's LDS reads to its place in the chain: left
        // alone the compiler reads all 1024 coefficient pairs before the first product and spills them)
        int zoff;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zoff) : "v"(__double2hiint(uq)));
        const snode_v2d *cf = (const snode_v2d *)&Ll[k * SN_NB + zoff];
#pragma unroll
        for (int p2 = (k + 1) / 2; p2 < SN_NB / 2; ++p2) {
            const snode_v2d cc = cf[p2];
            x[2 * p2] -= cc.x * uq;
            x[2 * p2 + 1] -= cc.y * uq;
        }
    }
    sn_stamp(sv.dbg, dbgme, 6);
    if (!rowok) return;
#pragma unroll
    for (int jj = 0; jj < SN_NB; ++jj)
        if (jj < nbw) v.Lx[colbase[jj] + R] = x[jj] * dinvl[jj];
    sn_stamp(sv.dbg, dbgme, 7, true);
    if (sv.dbg && tid == 0) atomicMax((unsigned long long *)&sv.dbg[8], (unsigned long long)wall_clock64()); // last workgroup's end
}

// Substitutions through a chain supernode, one workgroup per supernode of the unit level, the
// members' slice of x (and the nb entries of the rows of B) in LDS, block columns of SN_NB:
//   forward  (qdldl.rs:708-719): x_S <- (I + L_SS)^-1 x_S block by block -- the 64 unknowns of a block
//            by one wave (values in registers, broadcast by lane shuffles), then every row below the
//            block subtracts its 64 products (columns streamed, one row per thread) -- and finally
//            x_B -= L_BS x_S is pushed to the ancestors' entries with one atomic per row;
//   backward (qdldl.rs:737-752): x_S <- D^-1 x_S - L_BS' x_B - L_SS' x_S, last block first: one wave per
//            column reduces the rows below the block, then the block itself backwards in one wave.
// The rows' contributions from columns that are not supernode members are gathered beforehand by the
// row-gather kernels over the filtered lists (Engine: fwu / bwu).
constexpr int SN_XB_CAP = 4096; // rows of B kept in LDS (beyond: global atomics / loads)
struct SnodeSolveLds {
    double *xs, *xB, *Tl, *csum;
    int *colbase;
};
__device__ __forceinline__ SnodeSolveLds snode_solve_lds(char *smem, int wmax, int nbcap) {
    SnodeSolveLds L;
    L.xs = (double *)smem;
    L.xB = L.xs + wmax;
    L.Tl = L.xB + nbcap;
    L.csum = L.Tl + SN_NB * SN_NB;
    L.colbase = (int *)(L.csum + SN_NB);
    return L;
}
// with_B = 0: the rows of B are left to k_snode_push / k_snode_pull (their own multi-workgroup launches)
__global__ __launch_bounds__(SN_WG) void k_snode_fwd(LdlView v, SnodeView sv, const int *__restrict__ order,
                                                     double *x, int wmax, int nbcap, int with_B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SnodeSolveLds L = snode_solve_lds(smem, wmax, nbcap);
    int sn;
    const SnodeGeom g = snode_geom(sv, order, (int)blockIdx.x, sn);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int *Bn = v.Li + v.Lp[g.e];
    const bool ldsB = g.nb <= nbcap;
    const int hrows = with_B ? g.h : g.w;
    for (int t = tid; t < g.w; t += SN_WG) {
        L.colbase[t] = g.cb[t];
        L.xs[t] = x[g.cols[t]];
    }
    if (ldsB && with_B)
        for (int r = tid; r < g.nb; r += SN_WG) L.xB[r] = 0.0;
    __syncthreads();
    for (int j0 = 0; j0 < g.w; j0 += SN_NB) {
        const int nbw = min(SN_NB, g.w - j0);
        for (int idx = tid; idx < SN_NB * SN_NB; idx += SN_WG) { // Tl[col * SN_NB + row] (snode_block_solve)
            const int jj = idx / SN_NB, ii = idx % SN_NB;
            L.Tl[idx] = (ii > jj && ii < nbw) ? v.Lx[L.colbase[j0 + jj] + j0 + ii] : 0.0;
        }
        __syncthreads();
        if (wave == 0) { // the block's unknowns: lane = row, values in registers
            double xv = lane < nbw ? L.xs[j0 + lane] : 0.0;
            xv = snode_block_solve<true>(L.Tl, xv, lane);
            if (lane < nbw) L.xs[j0 + lane] = xv;
        }
        __syncthreads();
        // rows below the block
        for (int i = j0 + nbw + tid; i < hrows; i += SN_WG) {
            double sacc = 0.0;
            if (nbw == SN_NB) { // 32 column runs in flight per thread
#pragma unroll
                for (int j2 = 0; j2 < SN_NB; j2 += 32) {
                    double lv[32];
#pragma unroll
                    for (int q = 0; q < 32; ++q) lv[q] = v.Lx[L.colbase[j0 + j2 + q] + i];
#pragma unroll
                    for (int q = 0; q < 32; ++q) sacc += lv[q] * L.xs[j0 + j2 + q];
                }
            } else {
#pragma unroll 8
                for (int jj = 0; jj < nbw; ++jj) sacc += v.Lx[L.colbase[j0 + jj] + i] * L.xs[j0 + jj];
            }
            if (i < g.w) L.xs[i] -= sacc;
            else if (ldsB) L.xB[i - g.w] -= sacc;
            else atomicAdd(&x[Bn[i - g.w]], -sacc);
        }
        __syncthreads();
    }
    for (int t = tid; t < g.w; t += SN_WG) x[g.cols[t]] = L.xs[t];
    if (ldsB && with_B)
        for (int r = tid; r < g.nb; r += SN_WG) atomicAdd(&x[Bn[r]], L.xB[r]);
}
// x_B -= L_BS x_S after k_snode_fwd(with_B = 0): grid (groups of SN_WG rows of B, chunks of SN_PCH member
// columns, supernodes); one thread per row, the chunk of x_S in LDS, one atomic per (row, chunk)
constexpr int SN_PCH = 256;
__global__ __launch_bounds__(SN_WG) void k_snode_push(LdlView v, SnodeView sv, const int *__restrict__ order,
                                                      double *x) {
    __shared__ double xs[SN_PCH];
    __shared__ int cb[SN_PCH];
    int sn;
    const SnodeGeom g = snode_geom(sv, order, (int)blockIdx.z, sn);
    const int t0 = (int)blockIdx.y * SN_PCH;
    const int r0 = (int)blockIdx.x * SN_WG;
    if (t0 >= g.w || r0 >= g.nb) return;
    const int nt = min(SN_PCH, g.w - t0), tid = threadIdx.x;
    if (tid < nt) {
        const int c = g.cols[t0 + tid];
        xs[tid] = x[c];
        cb[tid] = g.cb[t0 + tid] + g.w; // + panel row w + r
    }
    __syncthreads();
    const int r = r0 + tid;
    if (r >= g.nb) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int t = 0;
    for (; t + 7 < nt; t += 8) { // eight column runs in flight per thread
        const double l0 = v.Lx[cb[t] + r], l1 = v.Lx[cb[t + 1] + r], l2 = v.Lx[cb[t + 2] + r], l3 = v.Lx[cb[t + 3] + r];
        const double l4 = v.Lx[cb[t + 4] + r], l5 = v.Lx[cb[t + 5] + r], l6 = v.Lx[cb[t + 6] + r],
                     l7 = v.Lx[cb[t + 7] + r];
        s0 += l0 * xs[t] + l4 * xs[t + 4];
        s1 += l1 * xs[t + 1] + l5 * xs[t + 5];
        s2 += l2 * xs[t + 2] + l6 * xs[t + 6];
        s3 += l3 * xs[t + 3] + l7 * xs[t + 7];
    }
    for (; t < nt; ++t) s0 += v.Lx[cb[t] + r] * xs[t];
    const int *Bn = v.Li + v.Lp[g.e];
    atomicAdd(&x[Bn[r]], -((s0 + s1) + (s2 + s3)));
}
// x_S <- D^-1 x_S - L_BS' x_B before k_snode_bwd(with_B = 0): grid (groups of 64 member columns,
// supernodes); one wave per column over the nb rows of B (x_B in LDS)
__global__ __launch_bounds__(SN_WG) void k_snode_pull(LdlView v, SnodeView sv, const int *__restrict__ order,
                                                      double *x, int nbcap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *xB = (double *)smem;
    int sn;
    const SnodeGeom g = snode_geom(sv, order, (int)blockIdx.y, sn);
    const int t0 = (int)blockIdx.x * SN_NB;
    if (t0 >= g.w) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int *Bn = v.Li + v.Lp[g.e];
    const bool ldsB = g.nb <= nbcap;
    if (ldsB)
        for (int r = tid; r < g.nb; r += SN_WG) xB[r] = x[Bn[r]];
    __syncthreads();
    const int nt = min(SN_NB, g.w - t0);
    for (int tt = wave; tt < nt; tt += SN_WG / 64) {
        const int t = t0 + tt, c = g.cols[t];
        const int base = g.cb[t] + g.w;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        auto xat = [&](int r) { return ldsB ? xB[r] : x[Bn[r]]; };
        int r = lane;
        for (; r + 192 < g.nb; r += 256) {
            const double l0 = v.Lx[base + r], l1 = v.Lx[base + r + 64], l2 = v.Lx[base + r + 128], l3 = v.Lx[base + r + 192];
            s0 += l0 * xat(r);
            s1 += l1 * xat(r + 64);
            s2 += l2 * xat(r + 128);
            s3 += l3 * xat(r + 192);
        }
        for (; r < g.nb; r += 64) s0 += v.Lx[base + r] * xat(r);
        const double tot = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) x[c] = x[c] * v.Dinv[c] - tot;
    }
}
__global__ __launch_bounds__(SN_WG) void k_snode_bwd(LdlView v, SnodeView sv, const int *__restrict__ order,
                                                     double *x, int wmax, int nbcap, int with_B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SnodeSolveLds L = snode_solve_lds(smem, wmax, nbcap);
    int sn;
    const SnodeGeom g = snode_geom(sv, order, (int)blockIdx.x, sn);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int *Bn = v.Li + v.Lp[g.e];
    const bool ldsB = g.nb <= nbcap;
    const int hrows = with_B ? g.h : g.w;
    for (int t = tid; t < g.w; t += SN_WG) {
        const int c = g.cols[t];
        L.colbase[t] = g.cb[t];
        L.xs[t] = with_B ? x[c] * v.Dinv[c] : x[c]; // (k_snode_pull has applied D^-1 already)
    }
    if (ldsB && with_B)
        for (int r = tid; r < g.nb; r += SN_WG) L.xB[r] = x[Bn[r]];
    __syncthreads();
    const int nblk = (g.w + SN_NB - 1) / SN_NB;
    for (int b = nblk - 1; b >= 0; --b) {
        const int j0 = b * SN_NB, nbw = min(SN_NB, g.w - j0), j1 = j0 + nbw;
        for (int idx = tid; idx < SN_NB * SN_NB; idx += SN_WG) {
            const int ii = idx / SN_NB, jj = idx % SN_NB;
            L.Tl[idx] = (ii > jj && ii < nbw) ? v.Lx[L.colbase[j0 + jj] + j0 + ii] : 0.0;
        }
        // rows below the block (finished members, then B): each wave owns the columns wave, wave + 8, ...
        // of the block and walks them together, lanes along the columns (8 x 2 runs in flight per lane)
        {
            constexpr int CPW = SN_NB / (SN_WG / 64); // columns per wave
            auto xat = [&](int i) { return i < g.w ? L.xs[i] : (ldsB ? L.xB[i - g.w] : x[Bn[i - g.w]]); };
            double sc[CPW];
            int cbq[CPW];
#pragma unroll
            for (int q = 0; q < CPW; ++q) {
                sc[q] = 0.0;
                const int jj = wave + q * (SN_WG / 64);
                cbq[q] = jj < nbw ? L.colbase[j0 + jj] : INT_MIN; // (colbase itself may be -1)
            }
            int i = j1 + lane;
            for (; i + 64 < hrows; i += 128) {
                double l0[CPW], l1[CPW];
#pragma unroll
                for (int q = 0; q < CPW; ++q) {
                    l0[q] = cbq[q] != INT_MIN ? v.Lx[cbq[q] + i] : 0.0;
                    l1[q] = cbq[q] != INT_MIN ? v.Lx[cbq[q] + i + 64] : 0.0;
                }
                const double x0 = xat(i), x1 = xat(i + 64);
#pragma unroll
                for (int q = 0; q < CPW; ++q) sc[q] += l0[q] * x0 + l1[q] * x1;
            }
            for (; i < hrows; i += 64) {
                const double x0 = xat(i);
#pragma unroll
                for (int q = 0; q < CPW; ++q) sc[q] += (cbq[q] != INT_MIN ? v.Lx[cbq[q] + i] : 0.0) * x0;
            }
#pragma unroll
            for (int q = 0; q < CPW; ++q) {
                const double tot = wave_sum(sc[q]);
                const int jj = wave + q * (SN_WG / 64);
                if (lane == 0 && jj < nbw) L.csum[jj] = tot;
            }
        }
        __syncthreads();
        if (wave == 0) {
            double xv = lane < nbw ? L.xs[j0 + lane] - L.csum[lane] : 0.0;
            xv = snode_block_solve<false>(L.Tl, xv, lane);
            if (lane < nbw) L.xs[j0 + lane] = xv;
        }
        __syncthreads();
    }
    for (int t = tid; t < g.w; t += SN_WG) x[g.cols[t]] = L.xs[t];
}

// ---------------------------------------------------------------------------
// Substitutions through WIDE chain supernodes with several workgroups per supernode (config 5: 830..1750
// member columns; one workgroup streaming a 6 MB triangle is latency bound at ~30 GB/s).  The triangle
// (I + L_SS) is cut into 64 x 64 blocks; workgroup (r, s) owns block row r of supernode s of the level:
//   forward :  x_r <- (I + L_rr)^-1 (x_r - sum_{c < r} L_rc x_c)
//   backward:  x_r <- (I + L_rr)^-T (x_r - sum_{c > r} L_cr' x_c)      (x_r already scaled by D^-1 and
//                                                                        with L_BS' x_B taken off: k_snode_pull)
// as a pipeline INSIDE one launch: a workgroup consumes block c as soon as the flag of x_c shows this
// sweep's epoch, its own 64 x 64 products accumulated in registers (lane = row of the block, every wave a
// quarter of the columns; the cross-lane / cross-wave reduction happens once at the end), then solves its
// diagonal block in one wave and publishes x_r and its flag.  A workgroup only ever waits for workgroups
// with a SMALLER linear block id (the backward launch numbers the block rows in reverse), which the
// dispatcher starts first -- the usual synchronisation-free sparse triangular solve -- and the wait times
// out rather than hang.  x_c and the flags cross workgroups inside the launch: device-coherent atomic
// stores / loads, no agent-scope fence (see ir_arrive_wait).  The rows of B are handled by k_snode_push /
// k_snode_pull in their own launches.
// ---------------------------------------------------------------------------
// One 16-byte message per unknown: (value lo, epoch, value hi, epoch) written by ONE dwordx4 store and read by ONE dwordx4
// load, both device coherent (sc1, what the compiler emits for agent-scope atomics) -- a consumer that sees
// this sweep's epoch has the value with it, in one round trip; value and flag as two stores needed the
// producer to wait for the first to be acknowledged and the consumer to load twice (~2 of ~4.5 us per hop).
typedef int msg_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void msg_store(int *slot, double val, int tag) {
    msg_v4i m;
    // (val_lo, tag, val_hi, tag): each 8-byte half carries its own tag, so a store that the memory system
    // splits at 8-byte granularity can never pair a fresh tag with a stale half of the value
    m.x = __double2loint(val);
    m.y = tag;
    m.z = __double2hiint(val);
    m.w = tag;
    // (s_nop: a store of more than 8 bytes reads its data registers a few cycles after issue; the compiler pads
    // that hazard for its own stores, not for inline assembly -- without it the next VALU write clobbered the tags)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" ::"v"(slot), "v"(m) : "memory");
}
__device__ __forceinline__ msg_v4i msg_load(const int *slot) {
    msg_v4i m;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(m) : "v"(slot) : "memory");
    return m;
}
constexpr int SN2_WG = 256;
constexpr int SN2_WMAX = 4096; // widest supernode (symbolic.cpp: SN_MAX_W); its column bases are kept in LDS
template <bool FWDMODE>
__global__ __launch_bounds__(SN2_WG) void k_snode_tri(LdlView v, SnodeView sv, const int *__restrict__ order,
                                                      const int *__restrict__ blk_ptr, int *msg, int epoch,
                                                      double *x, int *timeout_flag) {
    __shared__ double Tl[SN_NB * SN_NB];
    __shared__ double part[SN2_WG / 64][SN_NB];
    __shared__ double pulled[SN_NB]; // backward: L_B,r' x_B of the own columns; forward: the finished x_r
    __shared__ int colbase[SN2_WMAX]; // forward: of all earlier columns; backward: of the own block only
    int sn;
    const SnodeGeom g = snode_geom(sv, order, (int)blockIdx.y, sn);
    const int nblk = (g.w + SN_NB - 1) / SN_NB;
    if ((int)blockIdx.x >= nblk) return;
    const int r = FWDMODE ? (int)blockIdx.x : nblk - 1 - (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = r * SN_NB, nbw = min(SN_NB, g.w - j0);
    int *mb = msg + (size_t)blk_ptr[sn] * 256; // this supernode's message slots: block c, lane l at (c * 64 + l) * 4
    constexpr int CPW = SN_NB / (SN2_WG / 64); // columns per wave: of block c (forward) / of the own block (backward)
    // nothing below depends on x: column bases, the diagonal block and the own entries are requested first
    const int cb_lo = FWDMODE ? 0 : j0, cb_hi = j0 + nbw;
    for (int t = cb_lo + tid; t < cb_hi; t += SN2_WG) colbase[t - cb_lo] = g.cb[t];
    double xown = (wave == 0 && lane < nbw) ? x[g.cols[j0 + lane]] : 0.0; // (last written before this launch)
    __syncthreads();
    const int *cbr = colbase + (FWDMODE ? j0 : 0); // column bases of the own block
    const int *Bn = v.Li + v.Lp[g.e];              // node ids of the rows of B
    if (!FWDMODE) {
        // the block's own columns first take D^-1 and the rows of B: x_j <- x_j / d_j - sum_r L(B_r, j) x(B_r)
        // (what k_snode_pull did in a launch of its own; here every block does it while it would otherwise
        // wait for the flags of the later blocks).  Rows of B along the lanes, this wave's CPW columns together.
        double pacc[CPW];
#pragma unroll
        for (int q = 0; q < CPW; ++q) pacc[q] = 0.0;
        for (int r0 = 0; r0 < g.nb; r0 += 64) {
            const int r = r0 + lane;
            const bool rok = r < g.nb;
            const double xb = rok ? x[Bn[r]] : 0.0;
            double lq[CPW];
#pragma unroll
            for (int q = 0; q < CPW; ++q) {
                const int j = wave * CPW + q;
                lq[q] = (rok && j < nbw) ? v.Lx[cbr[j] + g.w + r] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < CPW; ++q) pacc[q] += lq[q] * xb;
        }
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            const double tot = wave_sum(pacc[q]);
            if (lane == 0) pulled[wave * CPW + q] = tot;
        }
        __syncthreads();
        if (wave == 0 && lane < nbw) xown = xown * v.Dinv[g.cols[j0 + lane]] - pulled[lane];
    }
    // the diagonal block with the solve's lane index fastest (snode_block_solve): forward Tl[col * SN_NB + row],
    // backward Tl[row * SN_NB + col]
    for (int idx = tid; idx < SN_NB * SN_NB; idx += SN2_WG) {
        const int hi = idx / SN_NB, lo = idx % SN_NB;
        const int ii = FWDMODE ? lo : hi, jj = FWDMODE ? hi : lo;
        Tl[idx] = (ii > jj && ii < nbw) ? v.Lx[cbr[jj] + j0 + ii] : 0.0;
    }
    double acc[CPW];
#pragma unroll
    for (int q = 0; q < CPW; ++q) acc[q] = 0.0;
    const int nsteps = FWDMODE ? r : nblk - 1 - r;
    // every WAVE runs the pipeline on its own (no workgroup barrier per step): it requests the L entries of
    // the next step, waits for the flag of x_c, reads x_c (one entry per lane) and accumulates
    double lv[CPW], ln[CPW];
    auto request = [&](double(&dst)[CPW], int step) {
        const int c = FWDMODE ? step : nblk - 1 - step;
        const int c0 = c * SN_NB, ncw = min(SN_NB, g.w - c0);
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            const int j = wave * CPW + q;
            if (FWDMODE) dst[q] = (lane < nbw && j < ncw) ? v.Lx[colbase[c0 + j] + j0 + lane] : 0.0; // L(j0 + lane, c0 + j)
            else dst[q] = (lane < ncw && j < nbw) ? v.Lx[cbr[j] + c0 + lane] : 0.0;                  // L(c0 + lane, j0 + j)
        }
    };
    if (nsteps > 0) request(ln, 0);
    bool ok = true;
    for (int step = 0; step < nsteps && ok; ++step) {
        const int c = FWDMODE ? step : nblk - 1 - step;
        const int c0 = c * SN_NB, ncw = min(SN_NB, g.w - c0);
#pragma unroll
        for (int q = 0; q < CPW; ++q) lv[q] = ln[q];
        if (step + 1 < nsteps) request(ln, step + 1);
        // every lane polls the message of "its" unknown of block c
        msg_v4i mm;
        for (long long spins = 0;; ++spins) {
            mm = msg_load(mb + (c * 64 + lane) * 4);
            const bool got = lane >= ncw || (mm.y == epoch && mm.w == epoch);
            if (__all(got)) break;
            __builtin_amdgcn_s_sleep(1);
            if (spins > (1ll << 18)) {
                ok = false;
                if (lane == 0) *timeout_flag = 1;
                break;
            }
        }
        if (!ok) break;
        const double xcv = lane < ncw ? __hiloint2double(mm.z, mm.x) : 0.0;
        if (FWDMODE) {
#pragma unroll
            for (int q = 0; q < CPW; ++q) acc[0] += lv[q] * __shfl(xcv, wave * CPW + q, 64);
        } else {
#pragma unroll
            for (int q = 0; q < CPW; ++q) acc[q] += lv[q] * xcv;
        }
    }
    // reduce: forward across the waves (each holds its columns' share of every row), backward across lanes
    if (FWDMODE) {
        part[wave][lane] = acc[0];
    } else {
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            const double tot = wave_sum(acc[q]);
            if (lane == 0) part[0][wave * CPW + q] = tot;
        }
    }
    __syncthreads(); // (also: Tl is complete)
    if (wave == 0 && ok) {
        double xv = xown;
        if (lane < nbw) {
            if (FWDMODE) {
#pragma unroll
                for (int w = 0; w < SN2_WG / 64; ++w) xv -= part[w][lane];
            } else {
                xv -= part[0][lane];
            }
        }
        xv = snode_block_solve<FWDMODE>(Tl, xv, lane);
        if (lane < nbw) {
            msg_store(mb + (r * 64 + lane) * 4, xv, epoch); // to the other blocks of this sweep
            x[g.cols[j0 + lane]] = xv;                      // to the launches that follow
        }
        if (FWDMODE) pulled[lane] = lane < nbw ? xv : 0.0;
    }
    if (FWDMODE && g.nb > 0) {
        // after the flag (off the pipeline's critical path): this block's share of x_B -= L_BS x_S, one row of B
        // per thread, one atomic per (row, block) -- what k_snode_push did in a launch of its own
        __syncthreads();
        if (!ok) return;
        for (int rb = tid; rb < g.nb; rb += SN2_WG) {
            double sacc = 0.0;
#pragma unroll
            for (int j2 = 0; j2 < SN_NB; j2 += 16) {
                double lv2[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) lv2[q] = (j2 + q < nbw) ? v.Lx[cbr[j2 + q] + g.w + rb] : 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) sacc += lv2[q] * pulled[j2 + q];
            }
            atomicAdd(&x[Bn[rb]], -sacc);
        }
    }
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
// fold a partial max (and NaN sighting) into the slotted inf-norm accumulator
__device__ __forceinline__ void fold_norm(unsigned long long *nrm, int *nan, double m, bool sawnan,
                                          int slot_seed) {
    if (sawnan) *nan = 1;
    if (m > 0.0) {
        unsigned long long *slot = nrm + (slot_seed & (NRM_SLOTS - 1)) * NRM_STRIDE;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
        if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bits) atomicMax(slot, bits);
    }
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

// scatter-add of one value per lane into the LDS row accumulators (tgt < 0: nothing to add).
// Must be called wave-converged.  When every active target of the wavefront is the SAME row
// (consecutive rows that all couple to one separator column: the u / v columns of a sparse SOC,
// a budget row) the 64 contributions are reduced in registers and ONE ds_add_f64 is issued
// instead of 64 serialised same-address atomics.
__device__ __forceinline__ void lds_scatter_add(double *acc, int tgt, double val) {
    const unsigned long long live = __ballot(tgt >= 0);
    if (live == 0ull) return;
    const int lead = __builtin_amdgcn_readfirstlane(__ffsll((long long)live) - 1);
    const int t0 = __builtin_amdgcn_readlane(tgt, lead); // (scalar lane select: no trip through the LDS crossbar)
    if (__popcll(live) > 1 && __ballot(tgt >= 0 && tgt != t0) == 0ull) {
        const double sum = wave_sum_all(tgt >= 0 ? val : 0.0);
        if ((threadIdx.x & 63) == 0) atomicAdd(&acc[t0], sum);
    } else if (tgt >= 0) {
        atomicAdd(&acc[tgt], val);
    }
}

// Residual e = b - K x for the rows of one bundle, with the symmetric matrix read ONCE:
// U row i = diagonal + entries (i, j) to ancestors j > i.  Every entry is applied in both
// directions: gathered into row i's own sum, and scattered (LDS fp64 atomic) into row j when j
// is in the bundle; rows j in the top are produced by the level-scheduled gather over their full
// rows instead.  Only the e slice lives in LDS (it takes the atomics); x is gathered from global
// memory -- a bundle's slice is a few tens of KB and stays in L1/L2 -- so that FOUR workgroups
// fit a CU (two LDS slices of a 3000-node bundle would cap it at three and push the 1000
// bundles of config 3 into a second round).  Two rows per thread, SSHOT entries of each per shot:
// rows of <= SSHOT entries cost one round trip.  ||e||inf of the bundle is folded into the slots.
constexpr int SSHOT_DEFAULT = 3; // entries of a row per shot (registers: 2 rows x SSHOT x (index, value, x address))
// FUSED (k_bundle_ir): the bundle id is passed in, x of the top rows comes from LDS (xt), the residual
// stays in es (e == nullptr) and the bundle's partial results -- ||e||inf of its rows (NaN when it saw one)
// and its shares of (K x)[top rows] -- are STORED to out_norm / out_share[0..k) instead of being added to
// shared accumulators: the consumers reduce them in a fixed order after a grid-wide barrier.
template <bool FUSED = false, int SSHOT = SSHOT_DEFAULT, int TW = BWG, int NR = 2>
__device__ __forceinline__ void bundle_symv_body(const BundleView &bv, const int *__restrict__ Up,
                                                 const int *__restrict__ Ucol, const double *__restrict__ Ux,
                                                 const double *x, const double *__restrict__ b, double *e,
                                                 unsigned long long *nrm, int *nanflag, double *es, double *red,
                                                 const FoldView &fold, int bid = blockIdx.x,
                                                 const double *xt = nullptr, double *out_norm = nullptr,
                                                 double *out_share = nullptr) {
    const int s0 = bv.bundle_ptr[bid], s1 = bv.bundle_ptr[bid + 1], nloc = s1 - s0;
    const int lane = threadIdx.x & 63, wbase = threadIdx.x - lane;
    // row pointers of the first sweep are requested BEFORE the b slice is staged
    int tb[NR], te[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int i = threadIdx.x + u * TW;
        tb[u] = i < nloc ? Up[s0 + i] : 0;
        te[u] = i < nloc ? Up[s0 + i + 1] : 0;
    }
    for (int i = threadIdx.x; i < nloc; i += TW) es[i] = b[s0 + i];
    // folded top rows: this bundle's share of (K x)[top], per thread, reduced at the end
    double tpart = 0.0; // fold.k == 1 (the usual arrow): registers
    __shared__ double tacc[8];
    if (fold.k > 1 && threadIdx.x < 8) tacc[threadIdx.x] = 0.0;
    __syncthreads();
    // loop bounds are kept wave-uniform (lds_scatter_add uses cross-lane operations)
    for (int w0 = wbase; w0 < nloc; w0 += NR * TW) {
        const int i0 = w0 + lane;
        double acc[NR], xi[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) acc[u] = 0.0;
        int maxlen = 0;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = i0 + u * TW;
            xi[u] = i < nloc ? x[s0 + i] : 0.0;
            maxlen = max(maxlen, te[u] - tb[u]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
        for (int k = 0; k < maxlen; k += SSHOT) {
            int jj[NR][SSHOT];
            double vv[NR][SSHOT];
#pragma unroll
            for (int u = 0; u < NR; ++u)
#pragma unroll
                for (int q = 0; q < SSHOT; ++q) {
                    const unsigned t = (unsigned)(tb[u] + k + q); // unsigned offset -> sgpr-base addressing
                    const bool ok = (int)t < te[u];
                    // (FUSED: Ucol points at the 16-bit bundle-local indices, >= nloc for the top rows)
                    jj[u][q] = ok ? (FUSED ? (int)((const unsigned short *)Ucol)[t] : Ucol[t]) : -1;
                    vv[u][q] = ok ? Ux[t] : 0.0;
                }
#pragma unroll
            for (int u = 0; u < NR; ++u)
#pragma unroll
                for (int q = 0; q < SSHOT; ++q) {
                    const int j = jj[u][q];
                    int tgt = -1;
                    if (FUSED) {
                        if (j >= 0) {
                            acc[u] += vv[u][q] * (j >= nloc ? xt[j - nloc] : x[s0 + j]);
                            if (j < nloc) {
                                if (j != i0 + u * TW) tgt = j;
                            } else if (fold.k == 1) {
                                tpart += vv[u][q] * xi[u];
                            } else {
                                atomicAdd(&tacc[j - nloc], vv[u][q] * xi[u]);
                            }
                        }
                    } else if (j >= 0) {
                        acc[u] += vv[u][q] * x[j];
                        if (j < s1) {
                            if (j != s0 + i0 + u * TW) tgt = j - s0;
                        } else if (fold.k == 1) {
                            tpart += vv[u][q] * xi[u];
                        } else if (fold.k > 1) {
                            atomicAdd(&tacc[j - fold.NF], vv[u][q] * xi[u]);
                        }
                    }
                    lds_scatter_add(es, tgt, -(vv[u][q] * xi[u]));
                }
        }
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = i0 + u * TW;
            if (i < nloc) atomicAdd(&es[i], -acc[u]);
            const int in = i + NR * TW; // the next sweep's row pointers
            tb[u] = in < nloc ? Up[s0 + in] : 0;
            te[u] = in < nloc ? Up[s0 + in + 1] : 0;
        }
    }
    __syncthreads();
    double m = 0.0;
    bool nan = false;
    for (int i = threadIdx.x; i < nloc; i += TW) {
        const double val = es[i];
        if (e) e[s0 + i] = val;
        if (val != val) nan = true;
        else m = fmax(m, fabs(val));
    }
    if (FUSED) {
        m = block_max(m, red);
        const bool anynan = __syncthreads_or(nan);
        // (device-coherent stores: read by another workgroup inside the same launch, see ir_arrive_wait)
        if (threadIdx.x == 0)
            __hip_atomic_store(out_norm, anynan ? __longlong_as_double(0x7ff8000000000000ll) : m, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        if (fold.k == 1) {
            tpart = block_sum(tpart, red);
            if (threadIdx.x == 0) __hip_atomic_store(out_share, tpart, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (fold.k > 1) {
            __syncthreads();
            if (threadIdx.x == 0)
                for (int i = 0; i < fold.k; ++i)
                    __hip_atomic_store(out_share + i, tacc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (nrm) {
        m = block_max(m, red);
        if (nan) *nanflag = 1;
        if (threadIdx.x == 0) fold_norm(nrm, nanflag, m, false, bid);
    }
    if (fold.k == 1) {
        tpart = block_sum(tpart, red);
        if (threadIdx.x == 0 && tpart != 0.0) atomicAdd(&fold.acc[fold_acc_index(1, 0, bid % FOLD_SLOTS)], tpart);
    } else if (fold.k > 1) {
        __syncthreads();
        if ((int)threadIdx.x < fold.k && tacc[threadIdx.x] != 0.0)
            atomicAdd(&fold.acc[fold_acc_index(1, threadIdx.x, bid % FOLD_SLOTS)], tacc[threadIdx.x]);
    }
}
// The residual of k_bundle_ir without a single gather from global memory ("split" form; one bundle per workgroup).
// A row's entries point at ANCESTORS, which are never leaves (level 0 of the bundle), and a leaf's own x is read by
// its own row only.  So during the residual the LDS slice holds, instead of one full vector:
//   e of the non-leaf nodes at their natural places xs[nleaf .. nloc) (they take the scatter-adds),
//   x of the non-leaf nodes in the space that is left: non-leaf t at xs[t] (t < nleaf) or xs[nloc + t - nleaf],
// nloc + max(0, nloc - 2 nleaf) doubles in all; the leaves' e goes straight to the spill vector (coalesced) and comes
// back into xs[0 .. nleaf) once the gathers are done.  The old form gathered x from the L2 (a 24 KB window per
// workgroup, written just before): two dependent round trips per sweep of rows, 38 of the launch's 235 us on config 3.
template <int SSHOT, int TW, int NR>
__device__ __forceinline__ void bundle_symv_split(const BundleView &bv, const int *__restrict__ Up,
                                                  const unsigned short *__restrict__ Ucol16,
                                                  const double *__restrict__ Ux, const double *x,
                                                  const double *__restrict__ b, double *spill, double *xs, double *red,
                                                  int k, int bid, const double *xt, double *out_norm, double *out_share) {
    const int s0 = bv.bundle_ptr[bid], nloc = bv.bundle_ptr[bid + 1] - s0;
    const int nleaf = bv.blvl[bv.blvl_ptr[bid] + 1] - s0, nin = nloc - nleaf;
    const int lane = threadIdx.x & 63, wbase = threadIdx.x - lane;
    int tb[NR], te[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int i = threadIdx.x + u * TW;
        tb[u] = i < nloc ? Up[s0 + i] : 0;
        te[u] = i < nloc ? Up[s0 + i + 1] : 0;
    }
    auto xpos = [&](int t) { return t < nleaf ? t : nloc + (t - nleaf); }; // x of non-leaf t
    for (int t = threadIdx.x; t < nin; t += TW) {
        const double xv = x[s0 + nleaf + t], bv_ = b[s0 + nleaf + t];
        xs[xpos(t)] = xv;
        xs[nleaf + t] = bv_;
    }
    double tpart = 0.0;
    __shared__ double tacc2[8];
    if (k > 1 && threadIdx.x < 8) tacc2[threadIdx.x] = 0.0;
    double mleaf = 0.0;
    bool nan = false;
    __syncthreads();
    for (int w0 = wbase; w0 < nloc; w0 += NR * TW) {
        const int i0 = w0 + lane;
        double acc[NR], xi[NR], bi[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) acc[u] = 0.0;
        int maxlen = 0;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = i0 + u * TW;
            xi[u] = i < nloc ? x[s0 + i] : 0.0;
            bi[u] = i < nleaf ? b[s0 + i] : 0.0;
            maxlen = max(maxlen, te[u] - tb[u]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
        for (int kk = 0; kk < maxlen; kk += SSHOT) {
            int jj[NR][SSHOT];
            double vv[NR][SSHOT];
#pragma unroll
            for (int u = 0; u < NR; ++u)
#pragma unroll
                for (int q = 0; q < SSHOT; ++q) {
                    const unsigned t = (unsigned)(tb[u] + kk + q);
                    const bool ok = (int)t < te[u];
                    jj[u][q] = ok ? (int)Ucol16[t] : -1;
                    vv[u][q] = ok ? Ux[t] : 0.0;
                }
#pragma unroll
            for (int u = 0; u < NR; ++u)
#pragma unroll
                for (int q = 0; q < SSHOT; ++q) {
                    const int j = jj[u][q], i = i0 + u * TW;
                    int tgt = -1;
                    if (j >= 0) {
                        if (j >= nloc) {
                            acc[u] += vv[u][q] * xt[j - nloc];
                            if (k == 1) tpart += vv[u][q] * xi[u];
                            else atomicAdd(&tacc2[j - nloc], vv[u][q] * xi[u]);
                        } else if (j == i) {
                            acc[u] += vv[u][q] * xi[u];
                        } else {
                            acc[u] += vv[u][q] * xs[xpos(j - nleaf)];
                            tgt = j;
                        }
                    }
                    lds_scatter_add(xs, tgt, -(vv[u][q] * xi[u]));
                }
        }
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = i0 + u * TW;
            if (i < nleaf) {
                const double val = bi[u] - acc[u];
                spill[s0 + i] = val;
                if (val != val) nan = true;
                else mleaf = fmax(mleaf, fabs(val));
            } else if (i < nloc) {
                atomicAdd(&xs[i], -acc[u]);
            }
            const int in = i + NR * TW; // the next sweep's row pointers
            tb[u] = in < nloc ? Up[s0 + in] : 0;
            te[u] = in < nloc ? Up[s0 + in + 1] : 0;
        }
    }
    __syncthreads();
    double m = mleaf;
    for (int i = nleaf + (int)threadIdx.x; i < nloc; i += TW) {
        const double val = xs[i];
        if (val != val) nan = true;
        else m = fmax(m, fabs(val));
    }
    // the leaves' residual back into the slice (every thread re-reads what it wrote itself)
    for (int i = threadIdx.x; i < nleaf; i += TW) xs[i] = spill[s0 + i];
    m = block_max(m, red);
    const bool anynan = __syncthreads_or(nan);
    if (threadIdx.x == 0)
        __hip_atomic_store(out_norm, anynan ? __longlong_as_double(0x7ff8000000000000ll) : m, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (k == 1) {
        tpart = block_sum(tpart, red);
        if (threadIdx.x == 0) __hip_atomic_store(out_share, tpart, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (k > 1) {
        __syncthreads();
        if (threadIdx.x == 0)
            for (int i = 0; i < k; ++i)
                __hip_atomic_store(out_share + i, tacc2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
constexpr int IR_NSUB = 32;         // sub-counters / release words of the grid barrier, one 128-byte line each
constexpr int IR_CTL_INTS = 32 * (1 + 2 * IR_NSUB);

// Grid barrier with a reduction slot: every workgroup ARRIVES (hierarchical counters: ctl[32 (1 + s)] =
// sub-counter s, ctl[0] = master, all monotonic over the launch and zero at its start); the workgroup
// whose arrival completes the count is told so (IR_LAST) -- it alone reduces the partial results the
// others stored before arriving, publishes the few reduced numbers and then RELEASES the barrier by
// writing the generation into the release words ctl[32 (1 + IR_NSUB + s)], one per sub-group, which the
// waiting workgroups poll (~30 pollers per cache line, with back-off).  1000 workgroups that all re-read
// 1000 partials after a plain barrier would put 10^6 L2 requests behind every barrier.
// A wait that cannot complete (a launch that is not co-resident) times out: IR_TIMEOUT.
// NO agent-scope fence anywhere: on this part a release / acquire at agent scope writes back / invalidates
// the XCD's whole L2 (the eight L2s are not coherent with each other), and a polling loop of acquire loads
// keeps invalidating it under the workgroups that still compute (measured: 200-300 us per barrier).
// Everything that crosses workgroups -- partial results, published reductions, counters, release words --
// is therefore written and read with agent-scope ATOMIC stores / loads, which are performed at the device's
// coherence point; the issuing thread waits for its own stores to complete (workgroup-scope release =
// s_waitcnt) before it arrives.
enum { IR_TIMEOUT = 0, IR_WAITED = 1, IR_LAST = 2 };
__device__ __forceinline__ int ir_arrive_wait(int *ctl, int gen, int nwg) {
    __shared__ int s_state;
    __syncthreads();
    if (threadIdx.x == 0) {
        // this thread's atomic stores of the partial results have completed (been acknowledged) before the
        // arrival is issued
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        const int sub = blockIdx.x % IR_NSUB;
        const int members = nwg / IR_NSUB + (sub < nwg % IR_NSUB ? 1 : 0);
        int state = IR_WAITED;
        if (atomicAdd(ctl + 32 * (1 + sub), 1) + 1 == members * gen) {
            if (atomicAdd(ctl, 1) + 1 == min(IR_NSUB, nwg) * gen) state = IR_LAST;
        }
        if (state != IR_LAST) {
            const int *rel = ctl + 32 * (1 + IR_NSUB + sub);
            long long spins = 0;
            while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1ll << 21)) {
                    state = IR_TIMEOUT;
                    break;
                }
            }
        }
        s_state = state;
    }
    __syncthreads();
    return s_state;
}
__device__ __forceinline__ void ir_release(int *ctl, int gen, int nwg) {
    __syncthreads();
    // (the published results were stored by thread 0; it orders them before the release words)
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        for (int q = 0; q < min(IR_NSUB, nwg); ++q)
            __hip_atomic_store(ctl + 32 * (1 + IR_NSUB + q), gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// the last use of the counters in a launch: arrive without waiting; whoever completes the count zeroes
// them (every other workgroup is done with them), so the next launch on the stream needs no memset
__device__ __forceinline__ void ir_grid_exit(int *ctl, int gen, int nwg) {
    if (threadIdx.x != 0) return;
    const int sub = blockIdx.x % IR_NSUB;
    const int members = nwg / IR_NSUB + (sub < nwg % IR_NSUB ? 1 : 0);
    if (atomicAdd(ctl + 32 * (1 + sub), 1) + 1 == members * gen) {
        if (atomicAdd(ctl, 1) + 1 == min(IR_NSUB, nwg) * gen) {
            for (int q = 0; q < IR_NSUB; ++q) {
                __hip_atomic_store(ctl + 32 * (1 + q), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ctl + 32 * (1 + IR_NSUB + q), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(ctl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// the two halves of ir_arrive_wait for the grouped fold, whose verdict on a round rides on the NEXT round without a
// grid-wide wait in between: arrival without waiting (IR_LAST for the workgroup that completes the count: it reduces
// and releases), and the wait for the release of generation `gen` (IR_TIMEOUT / IR_WAITED)
__device__ __forceinline__ int ir_arrive_nowait(int *ctl, int gen, int nwg) {
    __shared__ int s_state2;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        const int sub = blockIdx.x % IR_NSUB;
        const int members = nwg / IR_NSUB + (sub < nwg % IR_NSUB ? 1 : 0);
        int state = IR_WAITED;
        if (atomicAdd(ctl + 32 * (1 + sub), 1) + 1 == members * gen) {
            if (atomicAdd(ctl, 1) + 1 == min(IR_NSUB, nwg) * gen) state = IR_LAST;
        }
        s_state2 = state;
    }
    __syncthreads();
    return s_state2;
}
__device__ __forceinline__ int ir_wait_word(const int *word, int gen) {
    __shared__ int s_state3;
    __syncthreads(); // (every thread has read the verdict of a previous call)
    if (threadIdx.x == 0) {
        int state = IR_WAITED;
        long long spins = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1ll << 22)) {
                state = IR_TIMEOUT;
                break;
            }
        }
        s_state3 = state;
    }
    __syncthreads();
    return s_state3;
}
// grouped fold: non-blocking arrival at a group's counter (monotonic over the launch); true for the workgroup whose
// arrival completes `expect` -- it alone then reduces what the group's other workgroups stored before arriving
__device__ __forceinline__ bool ir_group_arrive(int *cnt, int expect) {
    __shared__ int s_glast;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        s_glast = (atomicAdd(cnt, 1) + 1 == expect) ? 1 : 0;
    }
    __syncthreads();
    return s_glast != 0;
}
// values that cross workgroups inside the launch: device-coherent atomic accesses (see above)
__device__ __forceinline__ double ir_load(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ir_store(double *p, double val) {
    __hip_atomic_store(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double nanmax(double a, double b) { return (a != a || b != b) ? (a != a ? a : b) : fmax(a, b); }
// NaN-propagating max over the workgroup, broadcast
__device__ __forceinline__ double block_nanmax(double v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nanmax(v, __shfl_down(v, o, 64));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double t = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = nanmax(t, red[i]);
    return t;
}

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
    const int tid = threadIdx.x, lane = tid & 63;
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
    __syncthreads(); // (lev_e, the scaled slice)
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
            __syncthreads();
        }
        if (FWDMODE && k == 1) {
            tpart = wave_sum_all(tpart);
            if (lane == 0 && tpart != 0.0) atomicAdd(&tacc[0], tpart);
        }
        __syncthreads();
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
        if (nstep != step) __syncthreads(); // the level is complete
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
    __syncthreads();
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
    const int tid = threadIdx.x;
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
        double mb = 0.0, m = 0.0, part[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int q = tid; q < nb; q += TW) {
            if (first) mb = nanmax(mb, ir_load(&pnb[q]));
            m = nanmax(m, ir_load(&pn[(size_t)par * nb + q]));
            if (ir.ir_enable && !GR) {
                if (k == 1) part[0] += ir_load(&shs[(size_t)par * nb + q]);
                else
                    for (int i = 0; i < k; ++i) part[i] += ir_load(&shs[(size_t)par * nb * k + (size_t)q * k + i]);
            }
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
                const double tot = block_sum(part[i], red);
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
                            double y[8], nt = 0.0, nbt = 0.0;
                            for (int i = 0; i < k; ++i) {
                                const double rhs_i = round == 0 ? st.btop[i] : ir_load(&grec[(par ^ 1) * 32 + 16 + i]);
                                double sacc = rhs_i - st.tacc[i];
                                for (int j = 0; j < i; ++j) sacc -= st.ltt[i * 8 + j] * y[j];
                                y[i] = sacc;
                            }
                            for (int i = k - 1; i >= 0; --i) {
                                double sacc = y[i] * st.dinvt[i];
                                for (int j = i + 1; j < k; ++j) sacc -= st.ltt[j * 8 + i] * y[j];
                                y[i] = sacc;
                            }
                            // (the candidate x_top + dx_top is formed AFTER the barrier: whether the previous
                            // candidate was accepted is decided there)
                            for (int i = 0; i < k; ++i) {
                                ir_store(&grec[par * 32 + i], y[i]);
                                nt = nanmax(nt, fabs(y[i]));
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
                    double y[8];
                    for (int i = 0; i < k; ++i) {
                        double sacc = (round == 0 ? st.btop[i] : st.rtop[i]) - ir_load(&pub[par * 32 + i]);
                        for (int j = 0; j < i; ++j) sacc -= st.ltt[i * 8 + j] * y[j];
                        y[i] = sacc;
                    }
                    for (int i = k - 1; i >= 0; --i) {
                        double sacc = y[i] * st.dinvt[i];
                        for (int j = i + 1; j < k; ++j) sacc -= st.ltt[j * 8 + i] * y[j];
                        y[i] = sacc;
                    }
                    for (int i = 0; i < k; ++i) {
                        st.dxt[i] = y[i];
                        st.candt[i] = round == 0 ? y[i] : 1.0 * st.curt[i] + 1.0 * y[i];
                    }
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

// ---------------------------------------------------------------------------
// vectors
// ---------------------------------------------------------------------------
// From here to the end of the device code (vector algebra, dot products, cone kernels): the reductions keep
// their former butterfly order.  Their sums feed the interior-point loop's line searches and checkpoints,
// whose decisions the end-to-end tests compare ITERATE FOR ITERATE with the oracle-backed loop; the
// summation order is part of what was validated there.  (The factorisation / substitution / residual kernels
// above use the DPP reductions: their results are compared by value.)
#define wave_sum wave_sum_tree
#define wave_max wave_max_tree
#define block_sum block_sum_tree
#define block_max block_max_tree
__global__ __launch_bounds__(WG) void k_permute_in(double *__restrict__ y, const double *__restrict__ b,
                                                   const int *__restrict__ perm, int N) {
    for (int j = logical_block() * WG + threadIdx.x; j < N; j += gridDim.x * WG) y[j] = b[perm[j]];
}
__global__ __launch_bounds__(WG) void k_permute_out(double *__restrict__ x, const double *__restrict__ y,
                                                    const int *__restrict__ perm, int N) {
    for (int j = logical_block() * WG + threadIdx.x; j < N; j += gridDim.x * WG) x[perm[j]] = y[j];
}
// directldlkktsolver.rs:160-166 in the permuted numbering
__global__ __launch_bounds__(WG) void k_setrhs_perm(double *__restrict__ bp, double *__restrict__ xi,
                                                    const double *__restrict__ rx,
                                                    const double *__restrict__ rz,
                                                    const int *__restrict__ perm, int n, int m, int N,
                                                    unsigned long long *nrm, int *nanflag) {
    __shared__ double red[16];
    double mx = 0.0;
    bool nan = false;
    for (int j = logical_block() * WG + threadIdx.x; j < N; j += gridDim.x * WG) {
        const int o = perm[j];
        const double val = o < n ? rx[o] : (o < n + m ? rz[o - n] : 0.0);
        bp[j] = val;
        xi[j] = val;
        if (val != val) nan = true;
        else mx = fmax(mx, fabs(val));
    }
    mx = block_max(mx, red);
    if (nan) *nanflag = 1;
    if (threadIdx.x == 0) fold_norm(nrm, nanflag, mx, false, blockIdx.x);
}
// directldlkktsolver.rs:205-215
__global__ __launch_bounds__(WG) void k_getlhs_perm(double *lx, double *lz, const double *__restrict__ xp,
                                                    const int *__restrict__ iperm, int n, int m) {
    for (int i = logical_block() * WG + threadIdx.x; i < n + m; i += gridDim.x * WG) {
        const double val = xp[iperm[i]];
        if (i < n) {
            if (lx) lx[i] = val;
        } else if (lz) lz[i - n] = val;
    }
}
// ---- dense vector algebra of the caller either side of the solve (vecmath.rs:83-85,
//      :186-204): w = a x + b y, deterministic two-stage dot products -------------------------
__global__ __launch_bounds__(WG) void k_waxpby(double *w, double a, const double *x, double b, const double *y,
                                               int n) {
    for (int i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG)
        w[i] = y ? a * x[i] + b * y[i] : a * x[i];
}
constexpr int DOT_BLOCKS = 512;
__global__ __launch_bounds__(WG) void k_dot_partial(const double *__restrict__ a, const double *__restrict__ b,
                                                    int n, double *partials) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) s += a[i] * b[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(WG) void k_dot_final(const double *__restrict__ partials, int nb, double *out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += WG) s += partials[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = s;
}
// several dot products per launch pair (each with exactly the partition of the single k_dot_partial
// / k_dot_final pair, so the values do not depend on how they are batched)
__global__ __launch_bounds__(WG) void k_multi_dot_partial(DotBatch bt, double *partials) {
    __shared__ double red[16];
    const DotSpec sp = bt.s[blockIdx.y];
    const int nb = sp.n > 0 ? min(DOT_BLOCKS, (sp.n + WG - 1) / WG) : 0;
    if ((int)blockIdx.x >= nb) return;
    const double *__restrict__ a = sp.a;
    const double *__restrict__ b = sp.b;
    double acc = 0.0;
    for (int i = blockIdx.x * WG + threadIdx.x; i < sp.n; i += nb * WG) acc += a[i] * b[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.y * DOT_BLOCKS + blockIdx.x] = acc;
}
__global__ __launch_bounds__(WG) void k_multi_dot_final(DotBatch bt, const double *__restrict__ partials,
                                                        double *out) {
    __shared__ double red[16];
    const DotSpec sp = bt.s[blockIdx.x];
    const int nb = sp.n > 0 ? min(DOT_BLOCKS, (sp.n + WG - 1) / WG) : 0;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += WG) acc += partials[blockIdx.x * DOT_BLOCKS + i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) out[sp.slot] = acc;
}
// w = a x + b y + c z
__global__ __launch_bounds__(WG) void k_lin3(double *w, double a, const double *x, double b, const double *y,
                                             double c, const double *z, int n) {
    for (int i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) w[i] = a * x[i] + b * y[i] + c * z[i];
}
// vecmath.rs:87-99  dot_shifted: sum (s + a ds)(z + a dz), same two-stage reduction
__global__ __launch_bounds__(WG) void k_dot_shifted_partial(const double *__restrict__ z, const double *__restrict__ sv,
                                                            const double *__restrict__ dz,
                                                            const double *__restrict__ ds, double alpha, int n,
                                                            double *partials) {
    __shared__ double red[16];
    double acc = 0.0;
    for (int i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) {
        const double si = sv[i] + alpha * ds[i];
        const double zi = z[i] + alpha * dz[i];
        acc += si * zi;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
__global__ __launch_bounds__(WG) void k_add_vec(double *__restrict__ dx, const double *__restrict__ x, int N) {
    for (int i = logical_block() * WG + threadIdx.x; i < N; i += gridDim.x * WG) dx[i] = 1.0 * x[i] + 1.0 * dx[i];
}
__global__ __launch_bounds__(WG) void k_norm_inf(const double *__restrict__ vv, int N,
                                                 unsigned long long *out, int *nanflag) {
    __shared__ double red[16];
    double m = 0.0;
    bool nan = false;
    for (int i = logical_block() * WG + threadIdx.x; i < N; i += gridDim.x * WG) {
        const double a = vv[i];
        if (a != a) nan = true;
        else m = fmax(m, fabs(a));
    }
    m = block_max(m, red);
    if (nan) *nanflag = 1;
    if (threadIdx.x == 0) fold_norm(out, nanflag, m, false, blockIdx.x);
}

// ---------------------------------------------------------------------------
// cones: scaling update + Hs blocks fused into the KKT value update
// ---------------------------------------------------------------------------

// overflow-safe 2-norm of x[1..n) over a workgroup (vecmath.rs:206-226 computes the
// same scale*sqrt(sum (x/scale)^2) with a running scale)
__device__ __forceinline__ double block_norm_tail(const double *x, int n, double *red) {
    double amax = 0.0;
    for (int i = 1 + threadIdx.x; i < n; i += WG) amax = fmax(amax, fabs(x[i]));
    amax = block_max(amax, red);
    if (amax == 0.0) return 0.0;
    double ss = 0.0;
    for (int i = 1 + threadIdx.x; i < n; i += WG) {
        const double r = fabs(x[i]) / amax;
        ss += r * r;
    }
    ss = block_sum(ss, red);
    return amax * sqrt(ss);
}

// per-cone state layout in v.eta/v.d plus the rank-2 coefficients
//   st[8*c + 0..7] = eta, d, u0, u1, v1, (unused)
// socone.rs:134-211, one workgroup per cone
// one launch for the scalings of the Nonnegative rows and the second-order cones: workgroups [0, ncones) take
// one cone each, the following ones a slab of Nonnegative rows (nonnegativecone.rs:77-90)
__device__ __forceinline__ void soc_update_scaling_body(const SocView &v, const double *__restrict__ sv,
                                                        const double *__restrict__ zv, int c, double *red);
__global__ __launch_bounds__(WG) void k_sym_update_scaling(SocView v, const int *__restrict__ nn_rows, int nn,
                                                           const double *__restrict__ sv,
                                                           const double *__restrict__ zv, double *w, double *lam) {
    __shared__ double red[16];
    if ((int)blockIdx.x < v.ncones) {
        soc_update_scaling_body(v, sv, zv, blockIdx.x, red);
        return;
    }
    const int nblk = gridDim.x - v.ncones, blk = blockIdx.x - v.ncones;
    for (int t = blk * WG + threadIdx.x; t < nn; t += nblk * WG) {
        const int r = nn_rows[t];
        const double s = sv[r], z = zv[r];
        lam[r] = sqrt(s * z);
        w[r] = sqrt(s / z);
    }
}
__device__ __forceinline__ void soc_update_scaling_body(const SocView &v, const double *__restrict__ sv,
                                                        const double *__restrict__ zv, int c, double *red) {
    const int n = v.dim[c];
    const double *s = sv + v.start[c], *z = zv + v.start[c];
    double *w = v.w + v.start[c], *lam = v.lam + v.start[c];
    const int tid = threadIdx.x;
    const double z0 = z[0], s0 = s[0];
    const double z1n = block_norm_tail(z, n, red);
    const double s1n = block_norm_tail(s, n, red);
    const double zres = (z0 - z1n) * (z0 + z1n), sres = (s0 - s1n) * (s0 + s1n);
    const double zscale = zres > 0.0 ? sqrt(zres) : 0.0;
    const double sscale = sres > 0.0 ? sqrt(sres) : 0.0;
    if (zscale == 0.0 || sscale == 0.0) {
        if (tid == 0) *v.fail = v.fail_gen;
        return;
    }
    const double eta = sqrt(sscale / zscale);
    const double rs = 1.0 / sscale, mrz = -(1.0 / zscale);
    for (int i = tid; i < n; i += WG) {
        double wi = s[i] * rs;
        if (i == 0) wi += z0 / zscale;
        else wi = mrz * z[i] + 1.0 * wi;
        w[i] = wi;
    }
    __syncthreads();
    const double w0a = w[0];
    const double w1n = block_norm_tail(w, n, red);
    const double wres = (w0a - w1n) * (w0a + w1n);
    const double wscale = wres > 0.0 ? sqrt(wres) : 0.0;
    if (wscale == 0.0) {
        if (tid == 0) *v.fail = v.fail_gen;
        return;
    }
    const double rw = 1.0 / wscale;
    double sq = 0.0;
    for (int i = tid; i < n; i += WG) {
        const double wi = w[i] * rw;
        w[i] = wi;
        if (i > 0) sq += wi * wi;
    }
    const double w1sq = block_sum(sq, red);
    const double w0 = sqrt(1.0 + w1sq);
    // lambda, socone.rs:174-184
    const double gamma = 0.5 * wscale;
    const double ca = (gamma + z0 / zscale) / sscale, cb = (gamma + s0 / sscale) / zscale;
    const double sc = 1.0 / (s0 / sscale + z0 / zscale + 2.0 * gamma);
    const double sqz = sqrt(sscale * zscale);
    for (int i = tid; i < n; i += WG) {
        if (i == 0) lam[0] = gamma * sqz;
        else lam[i] = ((ca * s[i] + cb * z[i]) * sc) * sqz;
    }
    if (tid == 0) {
        w[0] = w0;
        double *st = v.eta + 8 * c;
        st[0] = eta;
        // rank-2 terms, socone.rs:187-208
        const double alpha = 2.0 * w0;
        const double wsq = w0 * w0 + w1sq;
        const double wsqinv = 1.0 / wsq;
        const double d = 0.5 * wsqinv;
        const double u0 = sqrt(wsq - d);
        st[1] = d;
        st[2] = u0;
        st[3] = alpha / u0;
        st[4] = sqrt(2.0 * (2.0 + wsqinv) / (2.0 * wsq - wsqinv));
    }
}

// get_Hs (socone.rs:217-246) negated, and the sparse expansion columns
// (datamaps.rs:199-220): u, v scaled by -eta^2, D = [-eta^2, +eta^2].
__device__ __forceinline__ void soc_write_kkt_body(const SocView &v, double *Kx, unsigned long long *dslots, int c);
// one launch for the Hs values of the second-order cones (workgroups [0, ncones)) and of the Nonnegative rows
// (the following workgroups; get_Hs nonnegativecone.rs:96-101, negated and scattered)
__global__ __launch_bounds__(WG) void k_sym_write_kkt(SocView v, const int *__restrict__ nn_rows,
                                                      const int *__restrict__ nn_hsidx, int nn,
                                                      const double *__restrict__ w, const int *__restrict__ mapHs,
                                                      double *Kx, unsigned long long *dslots) {
    __shared__ double red[16];
    if ((int)blockIdx.x < v.ncones) {
        soc_write_kkt_body(v, Kx, dslots, blockIdx.x);
        return;
    }
    const int nblk = gridDim.x - v.ncones, blk = blockIdx.x - v.ncones;
    double mx = 0.0;
    bool nan = false;
    for (int t = blk * WG + threadIdx.x; t < nn; t += nblk * WG) {
        const double wi = w[nn_rows[t]];
        const double h = wi * wi;
        Kx[mapHs[nn_hsidx[t]]] = -h;
        if (h != h) nan = true;
        else mx = fmax(mx, h);
    }
    if (dslots) {
        mx = block_max(mx, red);
        int *nanflag = (int *)(dslots + (size_t)NRM_SLOTS * NRM_STRIDE);
        if (threadIdx.x == 0) fold_norm(dslots, nanflag, mx, false, blockIdx.x);
        if (nan) *nanflag = 1;
    }
}
__device__ __forceinline__ void soc_write_kkt_body(const SocView &v, double *Kx, unsigned long long *dslots, int c) {
    const int n = v.dim[c];
    const double *w = v.w + v.start[c];
    const double *st = v.eta + 8 * c;
    const double eta2 = st[0] * st[0];
    if (dslots && threadIdx.x == 0) {
        // the diagonal entries this cone writes: sparse form eta^2 d, eta^2 (Hs) and -+eta^2 (D); dense
        // form the diagonal of eta^2 (2 w w' - J)
        double mx;
        if (v.sparse_idx[c] >= 0) {
            mx = fmax(fabs(eta2 * st[1]), fabs(eta2));
        } else {
            const double s2 = 1.4142135623730951;
            mx = fabs(((s2 * w[0] - 1.0) * (s2 * w[0] + 1.0)) * eta2);
            for (int col = 1; col < n; ++col) mx = fmax(mx, fabs((2.0 * w[col] * w[col] + 1.0) * eta2));
        }
        int *nanflag = (int *)(dslots + (size_t)NRM_SLOTS * NRM_STRIDE);
        fold_norm(dslots, nanflag, mx != mx ? 0.0 : mx, mx != mx, c);
    }
    const int *mh = v.mapHs + v.hs_start[c];
    const int sidx = v.sparse_idx[c];
    if (sidx >= 0) {
        const double d = st[1], u0 = st[2], u1 = st[3], v1 = st[4];
        const int *mu = v.mapU + v.sp_ptr[sidx], *mv = v.mapV + v.sp_ptr[sidx];
        for (int i = threadIdx.x; i < n; i += WG) {
            const double h = (i == 0) ? eta2 * d : eta2;
            Kx[mh[i]] = -h;
            const double ui = (i == 0) ? u0 : u1 * w[i];
            const double vi = (i == 0) ? 0.0 : v1 * w[i];
            Kx[mu[i]] = ui * (-eta2);
            Kx[mv[i]] = vi * (-eta2);
        }
        if (threadIdx.x == 0) {
            Kx[v.mapD[2 * sidx]] = -eta2;
            Kx[v.mapD[2 * sidx + 1]] = eta2;
        }
    } else if (threadIdx.x == 0) {
        // dense packed triu of eta^2 (2 w w' - J), dim <= 4
        const double s2 = 1.4142135623730951;
        double h = (s2 * w[0] - 1.0) * (s2 * w[0] + 1.0);
        Kx[mh[0]] = -(h * eta2);
        int k = 1;
        for (int col = 1; col < n; ++col) {
            const double wc = w[col];
            for (int row = 0; row <= col; ++row) {
                h = 2.0 * w[row] * wc;
                if (row == col) h += 1.0;
                Kx[mh[k++]] = -(h * eta2);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Exponential / Power cones: 3x3 closed forms, one thread per cone.
// state per cone (18 doubles): Hs[6] | H_dual[6] | grad[3] | z[3]; packed triu
// order [00,01,11,02,12,22] (dense3x3/core.rs) == the KKT dense-triangle fill order.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double logsafe(double v) { return v <= 0.0 ? -INFINITY : log(v); }
__device__ __forceinline__ void sym3_mul(const double *H, double *y, const double *x) {
    y[0] = (H[0] * x[0]) + (H[1] * x[1]) + (H[3] * x[2]);
    y[1] = (H[1] * x[0]) + (H[2] * x[1]) + (H[4] * x[2]);
    y[2] = (H[3] * x[0]) + (H[4] * x[1]) + (H[5] * x[2]);
}
// expcone.rs:396-458 (Wright omega, two refinement sweeps)
__device__ double wright_omega(double z) {
    double p, w;
    if (z < 1.0 + 3.141592653589793) {
        const double zm1 = z - 1.0;
        p = zm1;
        w = 1.0 + p * 0.5;
        p *= zm1;
        w += p * (1. / 16.0);
        p *= zm1;
        w -= p * (1. / 192.0);
        p *= zm1;
        w -= p * (1. / 3072.0);
        p *= zm1;
        w += p * (13. / 61440.0);
    } else {
        const double logz = logsafe(z), zinv = 1.0 / z;
        w = z - logz;
        double q = logz * zinv;
        w += q;
        q *= zinv;
        w += q * (logz / 2.0 - 1.0);
        q *= zinv;
        w += q * (logz * logz / 3.0 - logz * 1.5 + 1.0);
    }
    double r = z - w - logsafe(w);
    for (int it = 0; it < 2; it++) {
        const double wp1 = w + 1.0;
        const double t = wp1 * (wp1 + (r * 2.0) / 3.0);
        w *= 1.0 + (r / wp1) * (t - r * 0.5) / (t - r);
        const double r4 = r * r * r * r;
        const double wp16 = wp1 * wp1 * wp1 * wp1 * wp1 * wp1;
        r = (w * w * 2.0 - w * 8.0 - 1.0) / (wp16 * 72.0) * r4;
    }
    return w;
}
// powcone.rs:447-491 + nonsymmetric_common.rs:193-219
__device__ double pow_newton_raphson(double s3, double phi, double a) {
    const double eps = 2.220446049250313e-16;
    double x = -1.0 / s3 + (s3 * 2.0 + sqrt((phi * phi) / (s3 * s3) + phi * 3.0)) / (phi - s3 * s3);
    const double t0 = -2.0 * a * logsafe(a) - 2.0 * (1.0 - a) * logsafe(1.0 - a);
    for (int iter = 0; iter < 100; iter++) {
        const double t1 = x * x, t2 = (2.0 * x) / s3;
        const double dfdx = (a * a * 2.0) / (a * x + (1.0 + a) / s3) +
                            ((1.0 - a) * 2.0) * (1.0 - a) / ((1.0 - a) * x + (2.0 - a) / s3) -
                            ((x + 1.0 / s3) * 2.0) / (t1 + t2);
        const double t2b = (x * 2.0) / s3;
        const double f = 2.0 * a * logsafe(2.0 * a * t1 + (1.0 + a) * t2b) +
                         2.0 * (1.0 - a) * logsafe(2.0 * (1.0 - a) * t1 + (2.0 - a) * t2b) - logsafe(phi) -
                         logsafe(t1 + t2b) - 2.0 * logsafe(t2b) + t0;
        const double dx = -f / dfdx;
        if (dx < eps || fabs(dx / x) < sqrt(eps) || fabs(dfdx) < eps) break;
        x += dx;
    }
    return x;
}
// update_scaling of expcone.rs:106-124 / powcone.rs:99-117 with update_Hs of
// nonsymmetric_common.rs:53-143; strategy 0 = PrimalDual, 1 = Dual
__global__ __launch_bounds__(WG) void k_ns3_update_scaling(Ns3View v, const double *__restrict__ sv,
                                                           const double *__restrict__ zv, double mu_in,
                                                           int strategy) {
    const int c = blockIdx.x * WG + threadIdx.x;
    if (c >= v.ncones) return;
    const double eps = 2.220446049250313e-16;
    const double *s = sv + v.start[c], *z = zv + v.start[c];
    double *Hs = v.state + 18 * c, *Hd = Hs + 6, *grad = Hs + 12, *zc = Hs + 15;
    const double a = v.alpha[c];
    const bool isexp = v.tag[c] == 3;
    double zt[3];
    if (isexp) { // expcone.rs:330-353, 361-373
        const double l = logsafe(-z[2] / z[0]);
        const double r = -z[0] * l - z[0] + z[1];
        const double c2 = 1.0 / r;
        grad[0] = c2 * l - 1.0 / z[0];
        grad[1] = -c2;
        grad[2] = (c2 * z[0] - 1.0) / z[2];
        Hd[0] = (r * r - z[0] * r + l * l * z[0] * z[0]) / (r * z[0] * z[0] * r);
        Hd[1] = -l / (r * r);
        Hd[2] = 1.0 / (r * r);
        Hd[3] = (z[1] - z[0]) / (r * r * z[2]);
        Hd[4] = -z[0] / (r * r * z[2]);
        Hd[5] = (r * r - z[0] * r + z[0] * z[0]) / (r * r * z[2] * z[2]);
        const double om = wright_omega(1.0 - s[0] / s[1] - logsafe(s[1] / s[2]));
        zt[0] = 1.0 / ((om - 1.0) * s[1]);
        zt[1] = zt[0] + zt[0] * logsafe(om * s[1] / s[2]) - 1.0 / s[1];
        zt[2] = om / ((1.0 - om) * s[2]);
    } else { // powcone.rs:353-386, 394-420
        const double phi = pow(z[0] / a, 2.0 * a) * pow(z[1] / (1.0 - a), 2.0 - 2.0 * a);
        const double psi = phi - z[2] * z[2];
        double g0 = 2.0 * a * phi / (z[0] * psi);
        double g1 = 2.0 * (1.0 - a) * phi / (z[1] * psi);
        double g2 = -2.0 * z[2] / psi;
        Hd[0] = g0 * g0 - 2.0 * a * (2.0 * a - 1.0) * phi / (z[0] * z[0] * psi) + (1.0 - a) / (z[0] * z[0]);
        Hd[1] = g0 * g1 - 4.0 * a * (1.0 - a) * phi / (z[0] * z[1] * psi);
        Hd[2] = g1 * g1 - 2.0 * (1.0 - a) * (1.0 - 2.0 * a) * phi / (z[1] * z[1] * psi) + a / (z[1] * z[1]);
        Hd[3] = g0 * g2;
        Hd[4] = g1 * g2;
        Hd[5] = g2 * g2 + 2.0 / psi;
        grad[0] = -2.0 * a * phi / (z[0] * psi) - (1.0 - a) / z[0];
        grad[1] = -2.0 * (1.0 - a) * phi / (z[1] * psi) - a / z[1];
        grad[2] = 2.0 * z[2] / psi;
        const double phis = pow(s[0], 2.0 * a) * pow(s[1], 2.0 - a * 2.0);
        const double abs_s = fabs(s[2]);
        if (abs_s > eps) {
            zt[2] = pow_newton_raphson(abs_s, phis, a);
            if (s[2] < 0.0) zt[2] = -zt[2];
            zt[0] = -(a * zt[2] * s[2] + 1.0 + a) / s[0];
            zt[1] = -((1.0 - a) * zt[2] * s[2] + 2.0 - a) / s[1];
        } else {
            zt[2] = 0.0;
            zt[0] = -(1.0 + a) / s[0];
            zt[1] = -(2.0 - a) / s[1];
        }
    }
    zc[0] = z[0];
    zc[1] = z[1];
    zc[2] = z[2];
    if (strategy == 1) {
        for (int i = 0; i < 6; i++) Hs[i] = mu_in * Hd[i];
        return;
    }
    const double *st = grad;
    const double dot_sz = s[0] * z[0] + s[1] * z[1] + s[2] * z[2];
    const double mu = dot_sz / 3.0;
    const double mut = (st[0] * zt[0] + st[1] * zt[1] + st[2] * zt[2]) / 3.0;
    double ds[3], dz[3], tmp[3];
    for (int i = 0; i < 3; i++) {
        ds[i] = s[i] + mu * st[i];
        dz[i] = z[i] + mu * zt[i];
    }
    const double dot_dsz = ds[0] * dz[0] + ds[1] * dz[1] + ds[2] * dz[2];
    const double de1 = mu * mut - 1.0;
    double q0 = zt[0] * (Hd[0] * zt[0] + Hd[1] * zt[1] + Hd[3] * zt[2]);
    q0 += zt[1] * (Hd[1] * zt[0] + Hd[2] * zt[1] + Hd[4] * zt[2]);
    q0 += zt[2] * (Hd[3] * zt[0] + Hd[4] * zt[1] + Hd[5] * zt[2]);
    const double de2 = q0 - 3.0 * mut * mut;
    if (fabs(de1) > sqrt(eps) && fabs(de2) > eps && dot_sz > 0.0 && dot_dsz > 0.0) {
        sym3_mul(Hd, tmp, zt);
        for (int i = 0; i < 3; i++) tmp[i] = mut * st[i] - tmp[i];
        const int IDX[3][3] = {{0, 1, 3}, {1, 2, 4}, {3, 4, 5}};
        double W6[6];
        for (int i = 0; i < 6; i++) W6[i] = Hd[i];
        for (int i = 0; i < 3; i++)
            for (int j = i; j < 3; j++) W6[IDX[i][j]] -= st[i] * st[j] / 3.0 + tmp[i] * tmp[j] / de2;
        double sumsq = 0.0;
        sumsq += W6[0] * W6[0] + W6[2] * W6[2] + W6[5] * W6[5];
        sumsq += (W6[1] * W6[1] + W6[3] * W6[3] + W6[4] * W6[4]) * 2.0;
        const double t = mu * sqrt(sumsq);
        double ax[3];
        ax[0] = z[1] * zt[2] - z[2] * zt[1];
        ax[1] = z[2] * zt[0] - z[0] * zt[2];
        ax[2] = z[0] * zt[1] - z[1] * zt[0];
        // stable 2-norm (vecmath.rs:206-226), sequential as in the reference
        double scale = 0.0, ssq = 1.0;
        for (int i = 0; i < 3; i++) {
            if (ax[i] == 0.0) continue;
            const double aa = fabs(ax[i]);
            if (scale < aa) {
                const double rr = scale / aa;
                ssq = 1.0 + ssq * rr * rr;
                scale = aa;
            } else {
                const double rr = aa / scale;
                ssq = ssq + rr * rr;
            }
        }
        const double nrm = scale * sqrt(ssq);
        if (nrm != 0.0) {
            const double rn = 1.0 / nrm;
            for (int i = 0; i < 3; i++) ax[i] *= rn;
        }
        for (int i = 0; i < 3; i++)
            for (int j = i; j < 3; j++)
                Hs[IDX[i][j]] = s[i] * s[j] / dot_sz + ds[i] * ds[j] / dot_dsz + t * ax[i] * ax[j];
    } else {
        for (int i = 0; i < 6; i++) Hs[i] = mu * Hd[i];
    }
}
// get_Hs (expcone.rs:130-133) negated + scattered
__global__ __launch_bounds__(WG) void k_ns3_write_hs(Ns3View v, double *Kx) {
    const int t = blockIdx.x * WG + threadIdx.x;
    if (t >= v.ncones * 6) return;
    const int c = t / 6, k = t - 6 * c;
    Kx[v.mapHs[v.hs_start[c] + k]] = -v.state[18 * c + k];
}
// mul_Hs (expcone.rs:135-137)
__global__ __launch_bounds__(WG) void k_ns3_mul_hs(Ns3View v, double *y, const double *__restrict__ x) {
    const int c = blockIdx.x * WG + threadIdx.x;
    if (c >= v.ncones) return;
    sym3_mul(v.state + 18 * c, y + v.start[c], x + v.start[c]);
}

// ---------------------------------------------------------------------------
// PSD triangle cone (psdtrianglecone.rs:144-212, 467-509): one workgroup per cone,
// all dense work (two Cholesky factors, an SVD, three small GEMMs) in LDS.
//   S = L1 L1', Z = L2 L2', M = L2' L1 = U Sigma V'
//   R = L1 V Sigma^-1/2,  B = R R' (the NT scaling matrix),  Hs = B (x)_s B
// The SVD is a one-sided (Hestenes) Jacobi iteration on the columns of M: it accumulates V
// and leaves sigma_p = ||m_p||; B does not depend on the order / signs of the singular pairs.
// n <= PSD_MAX_DIM (three n x n fp64 matrices in LDS).
// ---------------------------------------------------------------------------

// in-place lower Cholesky of the column-major n x n matrix A (upper part ignored); returns false
// (uniformly) when a pivot is not positive -> update_scaling fails like ?potrf (psdtrianglecone.rs:165-169)
__device__ bool lds_cholesky(double *A, int n, int *flag) {
    const int tid = threadIdx.x;
    for (int k = 0; k < n; ++k) {
        if (tid == 0) {
            const double p = A[k + k * n];
            if (!(p > 0.0)) *flag = 1;
            else A[k + k * n] = sqrt(p);
        }
        __syncthreads();
        if (*flag) return false;
        const double d = A[k + k * n];
        for (int i = k + 1 + tid; i < n; i += WG) A[i + k * n] /= d;
        __syncthreads();
        const int r = n - k - 1;
        for (int idx = tid; idx < r * r; idx += WG) {
            const int i = k + 1 + idx % r, j = k + 1 + idx / r;
            if (j <= i) A[i + j * n] -= A[i + k * n] * A[j + k * n];
        }
        __syncthreads();
    }
    return true;
}

// state of one PSD cone in HBM (3 n^2 + 2 n doubles): B = R R' (n*n) | lambda (n) | lambda^-1/2 (n) | R (n*n) | Rinv (n*n)
// GS = false: the four n x n work matrices live in LDS (n <= 64); GS = true: in this cone's slice of a scratch
// buffer in HBM (L2 resident: 4 n^2 doubles = 0.5 MB at n = 128) -- the same algorithm for cones of any size
// (the reference calls LAPACK and has no limit, psdtrianglecone.rs:144-204)
template <bool GS>
__global__ __launch_bounds__(WG) void k_psd_update_scaling(PsdView v, const double *__restrict__ sv,
                                                           const double *__restrict__ zv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int flag, rotated;
    const int c = blockIdx.x;
    if (c >= v.ncones) return;
    const int n = v.dim[c], tid = threadIdx.x;
    // A: S -> L1, Bm: Z -> L2, Cm: M = L2' L1 -> U Sigma, Vm: V
    double *A = GS ? v.scratch + (size_t)c * v.scratch_stride : (double *)smem;
    double *Bm = A + n * n, *Cm = Bm + n * n, *Vm = Cm + n * n;
    double *sig = Vm + n * n, *sgn = sig + n;
    int *rank = (int *)(sgn + n);
    const double *s = sv + v.start[c], *z = zv + v.start[c];
    const double isq2 = 0.7071067811865476;
    if (tid == 0) flag = 0;
    // svec -> symmetric matrices (dense/matrix_math.rs:165-205): packed triu, column major
    for (int idx = tid; idx < n * n; idx += WG) {
        const int i = idx % n, j = idx / n;
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int t = hi * (hi + 1) / 2 + lo;
        const double sc = (i == j) ? 1.0 : isq2;
        A[idx] = s[t] * sc;
        Bm[idx] = z[t] * sc;
    }
    __syncthreads();
    if (!lds_cholesky(A, n, &flag) || !lds_cholesky(Bm, n, &flag)) {
        if (tid == 0) *v.fail = v.fail_gen;
        return;
    }
    // M = L2' L1 ; V = I
    for (int idx = tid; idx < n * n; idx += WG) {
        const int a = idx % n, b = idx / n;
        double acc = 0.0;
        for (int i = (a > b ? a : b); i < n; ++i) acc += Bm[i + a * n] * A[i + b * n];
        Cm[idx] = acc;
        Vm[idx] = (a == b) ? 1.0 : 0.0;
    }
    __syncthreads();
    // one-sided Jacobi, round-robin pairing over np players (np even), EIGHT lanes per pair: each lane takes every
    // eighth row of the two columns (dot products as 8 partial sums + three butterfly steps inside the 8-lane
    // group, then its share of the rotation) -- with one thread per pair 25 of the 256 threads worked and a round
    // was ~350 dependent LDS round trips long: 4.0 ms per n = 50 cone, the longest single kernel of config 5's step
    const int np = (n + 1) & ~1;
    constexpr int JG = 8;
    const int jg = tid & (JG - 1);
    for (int sweep = 0; sweep < 30; ++sweep) {
        if (tid == 0) rotated = 0;
        __syncthreads();
        for (int r = 0; r < np - 1; ++r) {
            for (int pr0 = 0; pr0 < np / 2; pr0 += WG / JG) { // (every lane runs the shuffles: no early exit)
                const int pr = pr0 + tid / JG;
                int p = 0, q = 0;
                if (pr == 0) {
                    p = np - 1;
                    q = r;
                } else {
                    p = (r + pr) % (np - 1);
                    q = (r - pr + np - 1) % (np - 1);
                }
                const bool valid = pr < np / 2 && p < n && q < n;
                double *mp = Cm + (valid ? p : 0) * n, *mq = Cm + (valid ? q : 0) * n;
                double al = 0.0, be = 0.0, ga = 0.0;
                if (valid)
                    for (int i = jg; i < n; i += JG) {
                        const double a0 = mp[i], b0 = mq[i];
                        al += a0 * a0;
                        be += b0 * b0;
                        ga += a0 * b0;
                    }
#pragma unroll
                for (int off = 1; off < JG; off <<= 1) {
                    al += __shfl_xor(al, off, 64);
                    be += __shfl_xor(be, off, 64);
                    ga += __shfl_xor(ga, off, 64);
                }
                // (the butterfly adds in a different order on every lane: all eight take the leader's sums, so
                // that they agree bit for bit on the rotation and on whether to rotate at all)
                const int lead = (tid & 63) & ~(JG - 1);
                al = __shfl(al, lead, 64);
                be = __shfl(be, lead, 64);
                ga = __shfl(ga, lead, 64);
                if (valid && fabs(ga) > 1e-15 * sqrt(al * be) && ga != 0.0) {
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                    double *vp = Vm + p * n, *vq = Vm + q * n;
                    for (int i = jg; i < n; i += JG) {
                        const double a0 = mp[i], b0 = mq[i];
                        mp[i] = cs * a0 - sn * b0;
                        mq[i] = sn * a0 + cs * b0;
                        const double a1 = vp[i], b1 = vq[i];
                        vp[i] = cs * a1 - sn * b1;
                        vq[i] = sn * a1 + cs * b1;
                    }
                    if (jg == 0) rotated = 1;
                }
            }
            __syncthreads();
        }
        if (!rotated) break;
        __syncthreads();
    }
    __syncthreads();
    // sigma_p = ||m_p||; conventions LAPACK leaves open, fixed like the oracle: singular values in
    // descending order, each right singular vector signed so that its largest entry is positive
    for (int p = tid; p < n; p += WG) {
        double a = 0.0, big = 0.0, sg = 1.0;
        for (int i = 0; i < n; ++i) {
            a += Cm[i + p * n] * Cm[i + p * n];
            const double vv = Vm[i + p * n];
            if (fabs(vv) > big) {
                big = fabs(vv);
                sg = vv < 0.0 ? -1.0 : 1.0;
            }
        }
        sig[p] = sqrt(a);
        sgn[p] = sg;
    }
    __syncthreads();
    for (int p = tid; p < n; p += WG) {
        int r = 0;
        for (int q = 0; q < n; ++q) r += (sig[q] > sig[p]) || (sig[q] == sig[p] && q < p);
        rank[p] = r;
    }
    __syncthreads();
    double *st = v.state + v.state_off[c];
    double *Bout = st, *lam = st + n * n, *lis = lam + n, *Rout = lis + n, *Riout = Rout + n * n;
    for (int p = tid; p < n; p += WG) {
        lam[rank[p]] = sig[p];
        lis[rank[p]] = 1.0 / sqrt(sig[p]);
    }
    // R = L1 V Sigma^-1/2 (column p -> rank[p]);  Rinv = Sigma^-1/2 U' L2' with U = (M V) Sigma^-1
    for (int idx = tid; idx < n * n; idx += WG) {
        const int i = idx % n, p = idx / n;
        double acc = 0.0;
        for (int k = 0; k <= i; ++k) acc += A[i + k * n] * Vm[k + p * n];
        const double lsq = 1.0 / sqrt(sig[p]);
        Rout[i + rank[p] * n] = acc * sgn[p] * lsq;
        double acc2 = 0.0; // Rinv[p, i] = lsq/sig * sum_k Cm[k,p] L2[i,k]
        for (int k = 0; k <= i; ++k) acc2 += Cm[k + p * n] * Bm[i + k * n];
        Riout[rank[p] + i * n] = (acc2 / sig[p]) * sgn[p] * lsq;
    }
    __syncthreads();
    __threadfence_block();
    // B = R R' (invariant under the conventions above)
    for (int idx = tid; idx < n * n; idx += WG) {
        const int i = idx % n, j = idx / n;
        double acc = 0.0;
        for (int p = 0; p < n; ++p) acc += Rout[i + p * n] * Rout[j + p * n];
        Bout[idx] = acc;
    }
}

// ---- PSD cone operations either side of the solve: one workgroup per cone, matrices in LDS ----
__device__ __forceinline__ void psd_svec_to_mat(double *M, const double *x, int n, double scale_in = 1.0) {
    const double isq2 = 0.7071067811865476;
    for (int idx = threadIdx.x; idx < n * n; idx += WG) {
        const int i = idx % n, j = idx / n;
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        M[idx] = x[hi * (hi + 1) / 2 + lo] * ((i == j) ? 1.0 : isq2) * scale_in;
    }
}
// dense/matrix_math.rs:186-205
__device__ __forceinline__ void psd_mat_to_svec(double *y, const double *M, int n) {
    const double isq2 = 0.7071067811865476;
    const int numel = n * (n + 1) / 2;
    for (int t = threadIdx.x; t < numel; t += WG) {
        int col = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while (col * (col + 1) / 2 > t) --col;
        while ((col + 1) * (col + 2) / 2 <= t) ++col;
        const int row = t - col * (col + 1) / 2;
        y[t] = row == col ? M[row + col * n] : (M[row + col * n] + M[col + row * n]) * isq2;
    }
}
// C = op(A) op(B), all n x n column major; A / B may live in LDS or (L2-resident) global memory
template <bool TA, bool TB>
__device__ __forceinline__ void psd_gemm(double *C, const double *A, const double *B, int n) {
    for (int idx = threadIdx.x; idx < n * n; idx += WG) {
        const int i = idx % n, j = idx / n;
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc += (TA ? A[k + i * n] : A[i + k * n]) * (TB ? B[j + k * n] : B[k + j * n]);
        C[idx] = acc;
    }
}
// Y = Rx' X Rx (transpose == false: W x, W^-1 x) or Rx X Rx' (true: W' x, W^-T x), psdtrianglecone.rs:340-396
__device__ __forceinline__ void psd_mul_Wx(double *Y, double *T, const double *X, const double *Rx, int n,
                                           bool transpose) {
    if (transpose) {
        psd_gemm<false, true>(T, X, Rx, n); // T = X Rx'
        __syncthreads();
        psd_gemm<false, false>(Y, Rx, T, n); // Y = Rx T
    } else {
        psd_gemm<true, false>(T, Rx, X, n); // T = Rx' X
        __syncthreads();
        psd_gemm<false, false>(Y, T, Rx, n); // Y = T Rx
    }
    __syncthreads();
}
// eigenvalues of the symmetric matrix A (LDS, destroyed) by parallel two-sided Jacobi: per round the
// n/2 disjoint pairs of a round-robin schedule are rotated together (columns, then rows).  Returns
// the smallest eigenvalue and (psum) the sum of the positive ones to every thread.
__device__ double psd_eig_min(double *A, int n, double *cs, double *red, int *flag, double *psum) {
    const int np = (n + 1) & ~1, half = np / 2, tid = threadIdx.x;
    double *cc = cs, *ss = cs + half;
    int *pp = (int *)(cs + 2 * half), *qq = pp + half;
    for (int sweep = 0; sweep < 40; ++sweep) {
        if (tid == 0) *flag = 0;
        __syncthreads();
        for (int r = 0; r < np - 1; ++r) {
            for (int pr = tid; pr < half; pr += WG) {
                int p, q;
                if (pr == 0) {
                    p = np - 1;
                    q = r;
                } else {
                    p = (r + pr) % (np - 1);
                    q = (r - pr + np - 1) % (np - 1);
                }
                if (p > q) {
                    const int t = p;
                    p = q;
                    q = t;
                }
                double c = 1.0, sn = 0.0;
                if (q < n) {
                    const double apq = A[p + q * n], app = A[p + p * n], aqq = A[q + q * n];
                    if (fabs(apq) > 1e-17 * sqrt(fabs(app * aqq)) && apq != 0.0) {
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
                        c = 1.0 / sqrt(1.0 + t * t);
                        sn = t * c;
                        if (fabs(apq) > 1e-15 * (fabs(app) + fabs(aqq))) *flag = 1;
                    }
                } else {
                    p = -1;
                }
                pp[pr] = p;
                qq[pr] = q;
                cc[pr] = c;
                ss[pr] = sn;
            }
            __syncthreads();
            // columns: A <- A J
            for (int w = tid; w < half * n; w += WG) {
                const int k = w / n, i = w % n;
                const int p = pp[k], q = qq[k];
                if (p < 0) continue;
                const double c = cc[k], sn = ss[k];
                const double a = A[i + p * n], b = A[i + q * n];
                A[i + p * n] = c * a - sn * b;
                A[i + q * n] = sn * a + c * b;
            }
            __syncthreads();
            // rows: A <- J' A
            for (int w = tid; w < half * n; w += WG) {
                const int k = w / n, j = w % n;
                const int p = pp[k], q = qq[k];
                if (p < 0) continue;
                const double c = cc[k], sn = ss[k];
                const double a = A[p + j * n], b = A[q + j * n];
                A[p + j * n] = c * a - sn * b;
                A[q + j * n] = sn * a + c * b;
            }
            __syncthreads();
        }
        if (!*flag) break;
        __syncthreads();
    }
    double mn = INFINITY, sp = 0.0;
    for (int i = tid; i < n; i += WG) {
        const double e = A[i + i * n];
        mn = fmin(mn, e);
        sp += fmax(e, 0.0);
    }
    mn = -block_max(-mn, red);
    sp = block_sum(sp, red);
    if (psum) *psum = sp;
    return mn;
}
// state offsets (layout above k_psd_update_scaling)
struct PsdState {
    const double *B, *lam, *lis, *R, *Ri;
};
__device__ __forceinline__ PsdState psd_state(const PsdView &v, int c, int n) {
    const double *st = v.state + v.state_off[c];
    return {st, st + n * n, st + n * n + n, st + n * n + 2 * n, st + 2 * n * n + 2 * n};
}
//   OP 0 mul_Hs: o0 = svec(B X B)  (== W'(W x), psdtrianglecone.rs:214-218)
//   OP 1 affine_ds: o0 = svec(diag(lambda^2))  (:220-225)
//   OP 2 combined_ds_shift (symmetric_common.rs:53-84): o1 <- W o1, o2 <- W^-T o2, o0 = o2 o o1 - sc e
//   OP 3 ds_from_dz_offset (symmetric_common.rs:89-95): o0 = W'(lambda \ i0)
//   OP 4 step_length (:235-279, 437-463) with i0 = dz, i1 = ds, alpha_max = sc -> partial[c]
//   OP 5 margins (:104-121) of i0 -> partial[c] (min eig), partial2[c] (sum of positive eigs)
//   OP 6 barrier (:281-303) at (i0, i1) + sc (i2, i3) -> partial[c]
template <int OP, bool GS>
__global__ __launch_bounds__(WG) void k_psd_ops(PsdView v, double *o0, double *o1, double *o2,
                                                const double *__restrict__ i0, const double *__restrict__ i1,
                                                const double *__restrict__ i2, const double *__restrict__ i3,
                                                double sc, double *partial, double *partial2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[16];
    __shared__ int flag;
    const int c = blockIdx.x;
    if (c >= v.ncones) return;
    const int n = v.dim[c], off = v.start[c], tid = threadIdx.x;
    double *X = GS ? v.scratch + (size_t)c * v.scratch_stride : (double *)smem; // (see k_psd_update_scaling)
    double *Y = X + n * n, *T = Y + n * n, *cs = T + n * n;
    const PsdState st = psd_state(v, c, n);
    const int numel = n * (n + 1) / 2;
    if (OP == 0) {
        psd_svec_to_mat(X, i0 + off, n);
        __syncthreads();
        psd_gemm<false, false>(T, st.B, X, n);
        __syncthreads();
        psd_gemm<false, false>(Y, T, st.B, n);
        __syncthreads();
        psd_mat_to_svec(o0 + off, Y, n);
    } else if (OP == 1) {
        for (int t = tid; t < numel; t += WG) o0[off + t] = 0.0;
        __syncthreads();
        for (int k = tid; k < n; k += WG) o0[off + k * (k + 1) / 2 + k] = st.lam[k] * st.lam[k];
    } else if (OP == 2) {
        psd_svec_to_mat(X, o1 + off, n);
        __syncthreads();
        psd_mul_Wx(Y, T, X, st.R, n, false); // Y = W dz
        psd_mat_to_svec(o1 + off, Y, n);
        psd_svec_to_mat(X, o2 + off, n);
        __syncthreads();
        double *Z = T; // reuse after the product below is done with T
        psd_mul_Wx(X, T, X, st.Ri, n, true); // X = W^-T ds (T = X Ri' is complete before X is overwritten)
        psd_mat_to_svec(o2 + off, X, n);
        __syncthreads();
        // shift = (X Y + Y X) / 2 - sc I
        for (int idx = tid; idx < n * n; idx += WG) {
            const int i = idx % n, j = idx / n;
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc += X[i + k * n] * Y[k + j * n] + Y[i + k * n] * X[k + j * n];
            Z[idx] = 0.5 * acc - (i == j ? sc : 0.0);
        }
        __syncthreads();
        psd_mat_to_svec(o0 + off, Z, n);
    } else if (OP == 3) {
        psd_svec_to_mat(X, i0 + off, n);
        __syncthreads();
        for (int idx = tid; idx < n * n; idx += WG) {
            const int i = idx % n, j = idx / n;
            X[idx] = (2.0 * X[idx]) / (st.lam[i] + st.lam[j]);
        }
        __syncthreads();
        psd_mul_Wx(Y, T, X, st.R, n, true);
        psd_mat_to_svec(o0 + off, Y, n);
    } else if (OP == 4) {
        double amin = sc;
        for (int pass = 0; pass < 2; ++pass) {
            psd_svec_to_mat(X, (pass == 0 ? i0 : i1) + off, n);
            __syncthreads();
            psd_mul_Wx(Y, T, X, pass == 0 ? st.R : st.Ri, n, pass == 1);
            for (int idx = tid; idx < n * n; idx += WG) Y[idx] *= st.lis[idx % n] * st.lis[idx / n]; // lrscale
            __syncthreads();
            const double g = psd_eig_min(Y, n, cs, red, &flag, nullptr);
            if (g < 0.0) amin = fmin(amin, fmin(-(1.0 / g), sc));
            __syncthreads();
        }
        if (tid == 0) partial[c] = amin;
    } else if (OP == 5) {
        psd_svec_to_mat(X, i0 + off, n);
        __syncthreads();
        double sp;
        const double mn = psd_eig_min(X, n, cs, red, &flag, &sp);
        if (tid == 0) {
            partial[c] = mn;
            partial2[c] = sp;
        }
    } else if (OP == 6) {
        double bar = 0.0;
        for (int pass = 0; pass < 2; ++pass) {
            const double *xa = (pass == 0 ? i0 : i1) + off, *xb = (pass == 0 ? i2 : i3) + off;
            const double isq2 = 0.7071067811865476;
            for (int idx = tid; idx < n * n; idx += WG) {
                const int i = idx % n, j = idx / n;
                const int lo = i < j ? i : j, hi = i < j ? j : i;
                const int t = hi * (hi + 1) / 2 + lo;
                X[idx] = (1.0 * xa[t] + sc * xb[t]) * ((i == j) ? 1.0 : isq2);
            }
            if (tid == 0) flag = 0;
            __syncthreads();
            if (!lds_cholesky(X, n, &flag)) {
                bar = INFINITY;
            } else {
                double ld = 0.0;
                for (int i = tid; i < n; i += WG) ld += log(X[i + i * n]);
                ld = block_sum(ld, red);
                bar -= 2.0 * ld;
            }
            __syncthreads();
        }
        if (tid == 0) partial[c] = bar;
    }
}
// scaled_unit_shift / unit_initialization of the PSD cones (:123-137): diagonal svec entries
__global__ __launch_bounds__(WG) void k_psd_diag(PsdView v, double *z, double *s2, double alpha, int init) {
    const int c = blockIdx.x;
    if (c >= v.ncones) return;
    const int n = v.dim[c], off = v.start[c];
    if (init) {
        const int numel = n * (n + 1) / 2;
        for (int t = threadIdx.x; t < numel; t += WG) {
            z[off + t] = 0.0;
            s2[off + t] = 0.0;
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < n; k += WG) {
        const int t = off + k * (k + 1) / 2 + k;
        if (init) {
            z[t] = 1.0;
            s2[t] = 1.0;
        } else {
            z[t] += alpha;
        }
    }
}

// get_Hs = pack_triu(skron(B)) (psdtrianglecone.rs:210-212, 467-509), negated and scattered into K.
// Packed column-major triu: entry t <-> (row, col), row <= col; row <-> (i, j), col <-> (k, l).
template <bool GS>
__global__ __launch_bounds__(WG) void k_psd_write_hs(PsdView v, double *Kx, int blocks_per_cone) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int c = blockIdx.x / blocks_per_cone, part = blockIdx.x % blocks_per_cone;
    if (c >= v.ncones) return;
    const int n = v.dim[c];
    const double *Bin = v.state + v.state_off[c];
    const double *B = GS ? Bin : (const double *)smem; // large cones read B = R R' where it lives (L2 resident)
    if (!GS) {
        double *Bl = (double *)smem;
        for (int idx = threadIdx.x; idx < n * n; idx += WG) Bl[idx] = Bin[idx];
        __syncthreads();
    }
    const int numel = n * (n + 1) / 2;
    const long long total = (long long)numel * (numel + 1) / 2;
    const int *mh = v.mapHs + v.hs_start[c];
    const double sqrt2 = 1.4142135623730951;
    for (long long t = (long long)part * WG + threadIdx.x; t < total; t += (long long)blocks_per_cone * WG) {
        // col = largest cc with cc(cc+1)/2 <= t
        long long cc = (long long)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while (cc * (cc + 1) / 2 > t) --cc;
        while ((cc + 1) * (cc + 2) / 2 <= t) ++cc;
        const int col = (int)cc, row = (int)(t - cc * (cc + 1) / 2);
        int l = (int)((sqrt(8.0 * col + 1.0) - 1.0) * 0.5);
        while (l * (l + 1) / 2 > col) --l;
        while ((l + 1) * (l + 2) / 2 <= col) ++l;
        const int k = col - l * (l + 1) / 2;
        int j = (int)((sqrt(8.0 * row + 1.0) - 1.0) * 0.5);
        while (j * (j + 1) / 2 > row) --j;
        while ((j + 1) * (j + 2) / 2 <= row) ++j;
        const int i = row - j * (j + 1) / 2;
        const double Ajl = B[j + l * n], Ajk = B[j + k * n];
        double h;
        if (i != j && k != l) h = B[i + k * n] * Ajl + B[i + l * n] * Ajk;
        else if (i == j && k != l) h = sqrt2 * Ajl * Ajk;
        else if (i != j && k == l) h = sqrt2 * B[i + l * n] * Ajk;
        else h = Ajl * Ajl;
        Kx[mh[t]] = -h;
    }
}

// mul_Hs: nonnegativecone.rs:103-108, zerocone.rs:98-100
__global__ __launch_bounds__(WG) void k_nn_mul_hs(const int *__restrict__ rows, int count,
                                                  const double *__restrict__ w, double *y,
                                                  const double *__restrict__ x, int zero) {
    for (int t = logical_block() * WG + threadIdx.x; t < count; t += gridDim.x * WG) {
        const int r = rows[t];
        y[r] = zero ? 0.0 : w[r] * (w[r] * x[r]);
    }
}
// socone.rs:248-256
__global__ __launch_bounds__(WG) void k_soc_mul_hs(SocView v, double *y, const double *__restrict__ x) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    if (c >= v.ncones) return;
    const int n = v.dim[c];
    const double *w = v.w + v.start[c];
    const double *xc = x + v.start[c];
    double *yc = y + v.start[c];
    double dp = 0.0;
    for (int i = threadIdx.x; i < n; i += WG) dp += w[i] * xc[i];
    const double cc = block_sum(dp, red) * 2.0;
    const double eta = v.eta[8 * c];
    const double e2 = eta * eta;
    for (int i = threadIdx.x; i < n; i += WG) {
        const double base = (i == 0) ? -xc[0] : xc[i];
        yc[i] = (cc * w[i] + 1.0 * base) * e2;
    }
}

// ---------------------------------------------------------------------------
// step / right-hand-side operations of the symmetric cones:
// affine_ds, combined_ds_shift, ds_from_dz_offset, step_length, margins
// ---------------------------------------------------------------------------
// Nonnegative cone, elementwise (nonnegativecone.rs:110-153, symmetric_common.rs:53-84)
//   OP 0: ds = lam*lam            OP 1: combined shift (dz <- w dz, ds <- ds/w, shift = ds*dz - sm)
//   OP 2: out = ds / z            OP 3: zero fill (Zero cone rows)
template <int OP>
__global__ __launch_bounds__(WG) void k_nn_step_ops(const int *__restrict__ rows, int count,
                                                    const double *__restrict__ w,
                                                    const double *__restrict__ lam, double *o0, double *o1,
                                                    double *o2, const double *__restrict__ i0, double sm) {
    for (int t = logical_block() * WG + threadIdx.x; t < count; t += gridDim.x * WG) {
        const int r = rows[t];
        if (OP == 0) o0[r] = lam[r] * lam[r];
        else if (OP == 1) {
            const double dz = 1.0 * (o1[r] * w[r]);
            const double dsv = 1.0 * (o2[r] / w[r]);
            o1[r] = dz;
            o2[r] = dsv;
            o0[r] = dsv * dz + (-sm);
        } else if (OP == 2) o0[r] = i0[r] / o1[r];
        else o0[r] = 0.0;
    }
}
// scaled_unit_shift (compositecone.rs:208-214): z += alpha * e per cone -- every row of a
// nonnegative cone (nonnegativecone.rs:64-66), the head of a second-order cone (socone.rs:110-112);
// Zero cone rows are zeroed for the PRIMAL cone only (zerocone.rs:63-69)
__global__ __launch_bounds__(WG) void k_unit_shift(const int *__restrict__ nn_rows, int nn,
                                                   const int *__restrict__ zero_rows, int nz,
                                                   const int *__restrict__ soc_start, int nsoc, double *z,
                                                   double alpha, int primal) {
    const int total = nn + nz + nsoc;
    for (int t = blockIdx.x * WG + threadIdx.x; t < total; t += gridDim.x * WG) {
        if (t < nn) z[nn_rows[t]] += alpha;
        else if (t < nn + nz) {
            if (primal) z[zero_rows[t - nn]] = 0.0;
        } else z[soc_start[t - nn - nz]] += alpha;
    }
}
// per-block partial minima of the NN step lengths (nonnegativecone.rs:128-153)
__global__ __launch_bounds__(WG) void k_nn_step_length(const int *__restrict__ rows, int count,
                                                       const double *__restrict__ dz,
                                                       const double *__restrict__ ds,
                                                       const double *__restrict__ z,
                                                       const double *__restrict__ s, double amax,
                                                       double *partial) {
    __shared__ double red[16];
    double a = amax;
    for (int t = blockIdx.x * WG + threadIdx.x; t < count; t += gridDim.x * WG) {
        const int r = rows[t];
        if (dz[r] < 0.0) a = fmin(a, -z[r] / dz[r]);
        if (ds[r] < 0.0) a = fmin(a, -s[r] / ds[r]);
    }
    a = -block_max(-a, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = a;
}
// per-block partial (min z, sum max(z,0)) of NN rows (nonnegativecone.rs:58-62)
__global__ __launch_bounds__(WG) void k_nn_margins(const int *__restrict__ rows, int count,
                                                   const double *__restrict__ z, double *pmin, double *psum) {
    __shared__ double red[16];
    double a = 1.7976931348623157e308, b = 0.0;
    for (int t = blockIdx.x * WG + threadIdx.x; t < count; t += gridDim.x * WG) {
        const double zi = z[rows[t]];
        a = fmin(a, zi);
        b += fmax(zi, 0.0);
    }
    a = -block_max(-a, red);
    b = block_sum(b, red);
    if (threadIdx.x == 0) {
        pmin[blockIdx.x] = a;
        psum[blockIdx.x] = b;
    }
}

__device__ __forceinline__ double block_dot_tail(const double *a, const double *b, int n, double *red) {
    double s = 0.0;
    for (int i = 1 + threadIdx.x; i < n; i += WG) s += a[i] * b[i];
    return block_sum(s, red);
}
// socone.rs:421-495 on quantities already reduced by the workgroup
__device__ __forceinline__ double soc_step_roots(double x0, double y0, double x1n, double y1n, double x1y1,
                                                 double amax) {
    if (x0 >= 0.0 && y0 < 0.0) amax = fmin(amax, -x0 / y0);
    const double a = (y0 - y1n) * (y0 + y1n);
    const double b = 2.0 * (x0 * y0 - x1y1);
    const double cres = (x0 - x1n) * (x0 + x1n);
    const double c = cres > 0.0 ? cres : 0.0;
    const double d = b * b - 4.0 * a * c;
    if ((a > 0.0 && b > 0.0) || d < 0.0) return amax;
    if (a == 0.0) return amax;
    if (c == 0.0) return a >= 0.0 ? amax : 0.0;
    const double t = (b >= 0.0) ? (-b - sqrt(d)) : (-b + sqrt(d));
    double r1 = (2.0 * c) / t, r2 = t / (2.0 * a);
    if (r1 < 0.0) r1 = INFINITY;
    if (r2 < 0.0) r2 = INFINITY;
    return fmin(amax, fmin(r1, r2));
}
// one workgroup per second-order cone.
//   OP 0 affine_ds (socone.rs:258-260,360-367)      OP 1 combined_ds_shift (symmetric_common.rs:53-84,
//   OP 2 ds_from_dz_offset (socone.rs:266-287)           socone.rs:504-530)
//   OP 3 step_length -> partial[c] (socone.rs:289-302,421-495)
//   OP 4 margins -> pmin[c] = z0 - ||z1||, psum[c] = max(0, .) (socone.rs:104-108)
template <int OP>
__global__ __launch_bounds__(WG) void k_soc_step_ops(SocView v, double *o0, double *o1, double *o2,
                                                     const double *__restrict__ i0,
                                                     const double *__restrict__ i1,
                                                     const double *__restrict__ i2,
                                                     const double *__restrict__ i3, double sc,
                                                     double *partial, double *partial2) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    if (c >= v.ncones) return;
    const int n = v.dim[c], off = v.start[c], tid = threadIdx.x;
    const double *w = v.w + off, *lam = v.lam + off;
    const double eta = v.eta[8 * c];
    if (OP == 0) {
        double *ds = o0 + off;
        double dd = 0.0;
        for (int i = tid; i < n; i += WG) dd += lam[i] * lam[i];
        dd = block_sum(dd, red);
        const double l0 = lam[0];
        for (int i = tid; i < n; i += WG) ds[i] = (i == 0) ? dd : l0 * lam[i] + l0 * lam[i];
    } else if (OP == 1) {
        double *sh = o0 + off, *dz = o1 + off, *dsv = o2 + off;
        // dz <- W dz
        const double zeta = block_dot_tail(w, dz, n, red);
        const double x0 = dz[0];
        const double cw = x0 + zeta / (1.0 + w[0]);
        // ds <- W^-1 ds
        const double zeti = block_dot_tail(w, dsv, n, red);
        const double s0 = dsv[0];
        const double ci = -s0 + zeti / (1.0 + w[0]);
        __syncthreads();
        for (int i = tid; i < n; i += WG) {
            double a, b;
            if (i == 0) {
                a = (1.0 * eta) * (w[0] * x0 + zeta);
                b = (1.0 / eta) * (w[0] * s0 - zeti);
            } else {
                a = (1.0 * eta * cw) * w[i];
                a = (1.0 * eta) * dz[i] + 1.0 * a;
                b = (1.0 / eta * ci) * w[i];
                b = (1.0 / eta) * dsv[i] + 1.0 * b;
            }
            dz[i] = a;
            dsv[i] = b;
        }
        __syncthreads();
        // shift = ds o dz, shift[0] -= sigma*mu
        double dd = 0.0;
        for (int i = tid; i < n; i += WG) dd += dsv[i] * dz[i];
        dd = block_sum(dd, red);
        const double y0 = dsv[0], z0 = dz[0];
        for (int i = tid; i < n; i += WG) sh[i] = (i == 0) ? dd + (-sc) : y0 * dz[i] + z0 * dsv[i];
    } else if (OP == 2) {
        double *out = o0 + off;
        const double *d = i0 + off, *z = i1 + off;
        const double z1n = block_norm_tail(z, n, red);
        const double resz = (z[0] - z1n) * (z[0] + z1n);
        const double l1d1 = block_dot_tail(lam, d, n, red);
        const double w1d1 = block_dot_tail(w, d, n, red);
        const double cc = lam[0] * d[0] - l1d1;
        const double scale = cc / resz;
        const double rl = 1.0 / lam[0];
        for (int i = tid; i < n; i += WG) {
            double o = (i == 0) ? z[0] : -z[i];
            o *= scale;
            if (i == 0) o += eta * w1d1;
            else o += eta * (d[i] + w1d1 / (1.0 + w[0]) * w[i]);
            out[i] = o * rl;
        }
    } else if (OP == 3) {
        const double *dz = i0 + off, *dsv = i1 + off, *z = i2 + off, *s = i3 + off;
        const double z1n = block_norm_tail(z, n, red), dz1n = block_norm_tail(dz, n, red);
        const double zdz = block_dot_tail(z, dz, n, red);
        const double s1n = block_norm_tail(s, n, red), ds1n = block_norm_tail(dsv, n, red);
        const double sds = block_dot_tail(s, dsv, n, red);
        if (tid == 0) {
            const double az = soc_step_roots(z[0], dz[0], z1n, dz1n, zdz, sc);
            const double as = soc_step_roots(s[0], dsv[0], s1n, ds1n, sds, sc);
            partial[c] = fmin(az, as);
        }
    } else {
        const double *z = i0 + off;
        const double z1n = block_norm_tail(z, n, red);
        if (tid == 0) {
            const double a = z[0] - z1n;
            partial[c] = a;
            partial2[c] = fmax(0.0, a);
        }
    }
}

// ---------------------------------------------------------------------------
// Exponential / Power cones: the step operations either side of the solve, one thread per
// cone (expcone.rs:129-328, powcone.rs:128-337, nonsymmetric_common.rs:164-192) on the state
// written by k_ns3_update_scaling: Hs[6] | H_dual[6] | grad[3] | z[3]
// ---------------------------------------------------------------------------
// dense3x3/cholesky.rs:13-57 on the packed triu [00,01,11,02,12,22]
__device__ __forceinline__ bool chol3_factor(double *L, const double *A) {
    double t = A[0];
    if (t <= 0.0) return false;
    L[0] = sqrt(t);
    L[1] = A[1] / L[0];
    t = A[2] - L[1] * L[1];
    if (t <= 0.0) return false;
    L[2] = sqrt(t);
    L[3] = A[3] / L[0];
    L[4] = (A[4] - L[1] * L[3]) / L[2];
    t = A[5] - L[3] * L[3] - L[4] * L[4];
    if (t <= 0.0) return false;
    L[5] = sqrt(t);
    return true;
}
__device__ __forceinline__ void chol3_solve(const double *L, double *x, const double *b) {
    const double c0 = b[0] / L[0];
    const double c1 = (b[1] - L[1] * c0) / L[2];
    const double c2 = (b[2] - L[3] * c0 - L[4] * c1) / L[5];
    x[2] = c2 / L[5];
    x[1] = (c1 - L[4] * x[2]) / L[2];
    x[0] = (c0 - L[1] * x[1] - L[3] * x[2]) / L[0];
}
__device__ __forceinline__ double dot3(const double *a, const double *b) {
    return ((0.0 + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2];
}
__device__ bool ns3_feasible(bool isexp, bool dual, double a, const double *q) {
    if (isexp) {
        if (!dual) { // expcone.rs:189-203
            if (q[2] > 0.0 && q[1] > 0.0) return q[1] * logsafe(q[2] / q[1]) - q[0] > 0.0;
            return false;
        }
        if (q[2] > 0.0 && q[0] < 0.0) return q[1] - q[0] - q[0] * logsafe(-q[2] / q[0]) > 0.0; // :205-220
        return false;
    }
    if (!(q[0] > 0.0 && q[1] > 0.0)) return false;
    if (!dual) // powcone.rs:188-203
        return exp(2.0 * a * logsafe(q[0]) + 2.0 * (1.0 - a) * logsafe(q[1])) - q[2] * q[2] > 0.0;
    return exp((a * 2.0) * logsafe(q[0] / a) + (1.0 - a) * logsafe(q[1] / (1.0 - a)) * 2.0) - q[2] * q[2] > 0.0;
}
__device__ double ns3_backtrack(bool isexp, bool dual, double a, const double *dq, const double *q, double alpha,
                                double amin, double step) {
    for (;;) {
        const double w[3] = {1.0 * q[0] + alpha * dq[0], 1.0 * q[1] + alpha * dq[1], 1.0 * q[2] + alpha * dq[2]};
        if (ns3_feasible(isexp, dual, a, w)) break;
        alpha *= step;
        if (alpha < amin) return 0.0;
    }
    return alpha;
}
__device__ void ns3_higher_correction(bool isexp, double a, const double *Hd, const double *z, double *eta,
                                      const double *ds, const double *v) {
    double L[6], u[3];
    if (!chol3_factor(L, Hd)) {
        eta[0] = eta[1] = eta[2] = 0.0;
        return;
    }
    chol3_solve(L, u, ds);
    if (isexp) { // expcone.rs:254-308
        eta[1] = 1.0;
        eta[2] = -z[0] / z[2];
        eta[0] = logsafe(eta[2]);
        const double psi = z[0] * eta[0] - z[0] + z[1];
        const double dpu = dot3(u, eta), dpv = dot3(v, eta);
        const double coef =
            ((u[0] * (v[0] / z[0] - v[2] / z[2]) + u[2] * (z[0] * v[2] / z[2] - v[0]) / z[2]) * psi -
             2.0 * dpu * dpv) / (psi * psi * psi);
        for (int i = 0; i < 3; i++) eta[i] *= coef;
        const double ip2 = 1.0 / (psi * psi);
        eta[0] += (1.0 / psi - 2.0 / z[0]) * u[0] * v[0] / (z[0] * z[0]) - u[2] * v[2] / (z[2] * z[2]) / psi +
                  dpu * ip2 * (v[0] / z[0] - v[2] / z[2]) + dpv * ip2 * (u[0] / z[0] - u[2] / z[2]);
        eta[2] += 2.0 * (z[0] / psi - 1.0) * u[2] * v[2] / (z[2] * z[2] * z[2]) -
                  (u[2] * v[0] + u[0] * v[2]) / (z[2] * z[2]) / psi +
                  dpu * ip2 * (z[0] * v[2] / (z[2] * z[2]) - v[0] / z[2]) +
                  dpv * ip2 * (z[0] * u[2] / (z[2] * z[2]) - u[0] / z[2]);
    } else { // powcone.rs:260-337
        double Hp[6], Hv[3], Hu[3];
        const double phi = pow(z[0] / a, 2.0 * a) * pow(z[1] / (1.0 - a), 2.0 - 2.0 * a);
        const double psi = phi - z[2] * z[2];
        eta[0] = 2.0 * a * phi / z[0];
        eta[1] = 2.0 * (1.0 - a) * phi / z[1];
        eta[2] = -2.0 * z[2];
        Hp[1] = 4.0 * a * (1.0 - a) * phi / (z[0] * z[1]);
        Hp[0] = 2.0 * a * (2.0 * a - 1.0) * phi / (z[0] * z[0]);
        Hp[3] = 0.0;
        Hp[2] = 2.0 * (1.0 - a) * (1.0 - 2.0 * a) * phi / (z[1] * z[1]);
        Hp[4] = 0.0;
        Hp[5] = -2.0;
        const double dpu = dot3(u, eta), dpv = dot3(v, eta);
        sym3_mul(Hp, Hv, v);
        const double coef = (dot3(u, Hv) * psi - 2.0 * dpu * dpv) / (psi * psi * psi);
        const double coef2 = 4.0 * a * (2.0 * a - 1.0) * (1.0 - a) * phi * (u[0] / z[0] - u[1] / z[1]) *
                             (v[0] / z[0] - v[1] / z[1]) / psi;
        const double ip2 = 1.0 / (psi * psi);
        eta[0] = coef * eta[0] - 2.0 * (1.0 - a) * u[0] * v[0] / (z[0] * z[0] * z[0]) + coef2 / z[0] +
                 Hv[0] * dpu * ip2;
        eta[1] = coef * eta[1] - 2.0 * a * u[1] * v[1] / (z[1] * z[1] * z[1]) - coef2 / z[1] + Hv[1] * dpu * ip2;
        eta[2] = coef * eta[2] + Hv[2] * dpu * ip2;
        sym3_mul(Hp, Hu, u);
        for (int i = 0; i < 3; i++) eta[i] = (dpv * ip2) * Hu[i] + 1.0 * eta[i];
    }
    for (int i = 0; i < 3; i++) eta[i] *= 0.5;
}
__device__ double ns3_barrier(bool isexp, double a, const double *z, const double *s) {
    if (isexp) { // expcone.rs:222-252
        const double l = logsafe(-z[2] / z[0]);
        const double bd = -logsafe(-z[2] * z[0]) - logsafe(z[1] - z[0] - z[0] * l);
        double om = wright_omega(1.0 - s[0] / s[1] - logsafe(s[1] / s[2]));
        om = (om - 1.0) * (om - 1.0) / om;
        const double bp = -logsafe(om) - logsafe(s[1]) * 2.0 - logsafe(s[2]) - 3.0;
        return (0.0 + bd) + bp;
    }
    // powcone.rs:223-258 (primal gradient of :394-420)
    const double eps = 2.220446049250313e-16;
    const double arg1 = pow(z[0] / a, 2.0 * a) * pow(z[1] / (1.0 - a), 2.0 - 2.0 * a) - z[2] * z[2];
    const double bd = -logsafe(arg1) - (1.0 - a) * logsafe(z[0]) - a * logsafe(z[1]);
    double g[3];
    const double phis = pow(s[0], 2.0 * a) * pow(s[1], 2.0 - a * 2.0);
    const double abs_s = fabs(s[2]);
    if (abs_s > eps) {
        g[2] = pow_newton_raphson(abs_s, phis, a);
        if (s[2] < 0.0) g[2] = -g[2];
        g[0] = -(a * g[2] * s[2] + 1.0 + a) / s[0];
        g[1] = -((1.0 - a) * g[2] * s[2] + 2.0 - a) / s[1];
    } else {
        g[2] = 0.0;
        g[0] = -(1.0 + a) / s[0];
        g[1] = -(2.0 - a) / s[1];
    }
    double bp = 0.0;
    bp += logsafe(pow(-g[0] / a, 2.0 * a) * pow(-g[1] / (1.0 - a), 2.0 - a * 2.0) - g[2] * g[2]);
    bp += (1.0 - a) * logsafe(-g[0]);
    bp += a * logsafe(-g[1]) - 3.0;
    return (0.0 + bd) + bp;
}
//   OP 0 affine_ds: o0 = i0 (= s)               OP 1 combined_ds_shift: o0 = grad*sm - eta(ds = i1, v = i0)
//   OP 2 ds_from_dz_offset: o0 = i0 (= ds)      OP 3 step_length from sc -> partial[block] (min)
//   OP 4 barrier at (z, s) + sc*(dz, ds) -> partial[block] (sum)        OP 5 unit_initialization (o0 = z, o1 = s)
template <int OP>
__global__ __launch_bounds__(WG) void k_ns3_step_ops(Ns3View v, double *o0, double *o1,
                                                     const double *__restrict__ i0,
                                                     const double *__restrict__ i1,
                                                     const double *__restrict__ i2,
                                                     const double *__restrict__ i3, double sc, double amin,
                                                     double step, double *partial) {
    __shared__ double red[16];
    const int c = blockIdx.x * WG + threadIdx.x;
    const bool live = c < v.ncones;
    const int off = live ? v.start[c] : 0;
    const bool isexp = live ? v.tag[c] == 3 : true;
    const double a = live ? v.alpha[c] : 0.5;
    const double *st = v.state + 18 * (live ? c : 0);
    double out = OP == 3 ? sc : 0.0;
    if (live) {
        if (OP == 0 || OP == 2) {
            for (int k = 0; k < 3; k++) o0[off + k] = i0[off + k];
        } else if (OP == 1) {
            double eta[3];
            const double vz[3] = {i0[off], i0[off + 1], i0[off + 2]}, dsv[3] = {i1[off], i1[off + 1], i1[off + 2]};
            ns3_higher_correction(isexp, a, st + 6, st + 15, eta, dsv, vz);
            for (int k = 0; k < 3; k++) o0[off + k] = st[12 + k] * sc - eta[k];
        } else if (OP == 3) {
            const double dz[3] = {i0[off], i0[off + 1], i0[off + 2]}, dsv[3] = {i1[off], i1[off + 1], i1[off + 2]};
            const double z[3] = {i2[off], i2[off + 1], i2[off + 2]}, s[3] = {i3[off], i3[off + 1], i3[off + 2]};
            const double az = ns3_backtrack(isexp, true, a, dz, z, sc, amin, step);
            const double as = ns3_backtrack(isexp, false, a, dsv, s, sc, amin, step);
            out = fmin(az, as);
        } else if (OP == 4) {
            double cz[3], cs[3];
            for (int k = 0; k < 3; k++) {
                cz[k] = i0[off + k] + sc * i2[off + k];
                cs[k] = i1[off + k] + sc * i3[off + k];
            }
            out = ns3_barrier(isexp, a, cz, cs);
        } else if (OP == 5) {
            double u[3];
            if (isexp) { // expcone.rs:87-93
                u[0] = -1.051383945322714;
                u[1] = 0.556409619469370;
                u[2] = 1.258967884768947;
            } else { // powcone.rs:79-87
                u[0] = sqrt(1.0 + a);
                u[1] = sqrt(1.0 + (1.0 - a));
                u[2] = 0.0;
            }
            for (int k = 0; k < 3; k++) o0[off + k] = o1[off + k] = u[k];
        }
    }
    if (OP == 3) {
        out = -block_max(-out, red);
        if (threadIdx.x == 0) partial[blockIdx.x] = out;
    } else if (OP == 4) {
        out = block_sum(out, red);
        if (threadIdx.x == 0) partial[blockIdx.x] = out;
    }
}
// barrier of the nonnegative rows (nonnegativecone.rs:155-166): per-block partial sums
__global__ __launch_bounds__(WG) void k_nn_barrier(const int *__restrict__ rows, int count,
                                                   const double *__restrict__ z, const double *__restrict__ s,
                                                   const double *__restrict__ dz,
                                                   const double *__restrict__ ds, double alpha, double *partial) {
    __shared__ double red[16];
    double b = 0.0;
    for (int t = blockIdx.x * WG + threadIdx.x; t < count; t += gridDim.x * WG) {
        const int r = rows[t];
        b -= logsafe((s[r] + alpha * ds[r]) * (z[r] + alpha * dz[r]));
    }
    b = block_sum(b, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = b;
}
// barrier of one second-order cone per workgroup (socone.rs:304-314, 410-417)
__global__ __launch_bounds__(WG) void k_soc_barrier(SocView v, const double *__restrict__ zv,
                                                    const double *__restrict__ sv,
                                                    const double *__restrict__ dzv,
                                                    const double *__restrict__ dsv, double alpha,
                                                    double *partial) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    if (c >= v.ncones) return;
    const int n = v.dim[c], off = v.start[c];
    const double *z = zv + off, *s = sv + off, *dz = dzv + off, *ds = dsv + off;
    double qs = 0.0, qz = 0.0, ms = 0.0, mz = 0.0;
    for (int i = 1 + threadIdx.x; i < n; i += WG) {
        ms = fmax(ms, fabs(s[i] + alpha * ds[i]));
        mz = fmax(mz, fabs(z[i] + alpha * dz[i]));
    }
    ms = block_max(ms, red);
    mz = block_max(mz, red);
    for (int i = 1 + threadIdx.x; i < n; i += WG) { // scaled sums of squares (norm_shifted is overflow safe)
        const double xs = ms > 0.0 ? (s[i] + alpha * ds[i]) / ms : 0.0, xz = mz > 0.0 ? (z[i] + alpha * dz[i]) / mz : 0.0;
        qs += xs * xs;
        qz += xz * xz;
    }
    qs = block_sum(qs, red);
    qz = block_sum(qz, red);
    if (threadIdx.x == 0) {
        const double s1 = ms * sqrt(qs), z1 = mz * sqrt(qz);
        const double s0 = s[0] + alpha * ds[0], z0 = z[0] + alpha * dz[0];
        const double res_s = (s0 - s1) * (s0 + s1), res_z = (z0 - z1) * (z0 + z1);
        partial[c] = (res_s > 0.0 && res_z > 0.0) ? -logsafe(res_s * res_z) * 0.5 : INFINITY;
    }
}
// unit_initialization of the symmetric cones (zerocone.rs:71-74, nonnegativecone.rs:68-71, socone.rs:114-119)
__global__ __launch_bounds__(WG) void k_sym_unit_init(const int *__restrict__ nn_rows, int nn,
                                                      const int *__restrict__ soc_start, int nsoc, double *z,
                                                      double *s) {
    const int total = nn + nsoc;
    for (int t = blockIdx.x * WG + threadIdx.x; t < total; t += gridDim.x * WG) {
        const int r = t < nn ? nn_rows[t] : soc_start[t - nn];
        z[r] = 1.0;
        s[r] = 1.0;
    }
}

// ---------------------------------------------------------------------------
// Generalised power cone (genpowcone.rs), one workgroup per cone
// ---------------------------------------------------------------------------
__device__ __forceinline__ double block_prod(double v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v *= __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double t = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t *= red[i];
    return t;
}
struct GpwState {
    double *alpha, *q, *d1, *r, *p, *grad, *z, *sc; // sc: d2, mu, psi
};
__device__ __forceinline__ GpwState gpw_state(const GpwView &v, int c) {
    const int a = v.dim1[c], b = v.dim2[c];
    double *st = v.state + v.state_off[c];
    return {st, st + a, st + 2 * a, st + 3 * a, st + 3 * a + b, st + 4 * a + 2 * b, st + 5 * a + 3 * b,
            st + 6 * a + 4 * b};
}
// genpowcone.rs:361-401
__global__ __launch_bounds__(WG) void k_gpw_update_scaling(GpwView v, const double *__restrict__ zv, double mu) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    if (c >= v.ncones) return;
    const int a = v.dim1[c], b = v.dim2[c], tid = threadIdx.x;
    const double *z = zv + v.start[c];
    const GpwState st = gpw_state(v, c);
    double pr = 1.0, sq = 0.0;
    for (int k = tid; k < a; k += WG) pr *= pow(z[k] / st.alpha[k], 2.0 * st.alpha[k]);
    for (int k = tid; k < b; k += WG) sq += z[a + k] * z[a + k];
    const double phi = block_prod(pr, red);
    const double norm2w = block_sum(sq, red);
    const double zeta = phi - norm2w;
    const double p0 = sqrt(phi * (phi + norm2w) / 2.0);
    const double p1 = -2.0 * phi / p0;
    const double q0 = sqrt(zeta * phi / 2.0);
    const double r1 = 2.0 * sqrt(zeta / (phi + norm2w));
    for (int k = tid; k < a; k += WG) {
        const double al = st.alpha[k], zk = z[k];
        const double tau = 2.0 * al / zk;
        st.grad[k] = -tau * phi / zeta - (1.0 - al) / zk;
        st.d1[k] = tau * phi / (zeta * zk) + (1.0 - al) / (zk * zk);
        st.p[k] = (p0 / zeta) * tau;
        st.q[k] = tau * (q0 / zeta);
        st.z[k] = zk;
    }
    for (int k = tid; k < b; k += WG) {
        const double wk = z[a + k];
        st.grad[a + k] = (2.0 / zeta) * wk;
        st.p[a + k] = (p1 / zeta) * wk;
        st.r[k] = (r1 / zeta) * wk;
        st.z[a + k] = wk;
    }
    if (tid == 0) {
        st.sc[0] = 2.0 / zeta;
        st.sc[1] = mu;
    }
}
// get_Hs (:163-171) negated into K + csc_update_sparsecone (datamaps.rs:322-343)
__global__ __launch_bounds__(WG) void k_gpw_write_kkt(GpwView v, double *Kx) {
    const int c = blockIdx.x;
    if (c >= v.ncones) return;
    const int a = v.dim1[c], b = v.dim2[c], tid = threadIdx.x;
    const GpwState st = gpw_state(v, c);
    const double mu = st.sc[1], d2 = st.sc[0], sm = -sqrt(mu);
    const int *mh = v.mapHs + v.hs_start[c], *mq = v.mapQRP + v.map_ptr[c], *mr = mq + a, *mp = mr + b;
    for (int k = tid; k < a; k += WG) {
        Kx[mh[k]] = -(mu * st.d1[k]);
        Kx[mq[k]] = st.q[k] * sm;
    }
    for (int k = tid; k < b; k += WG) {
        Kx[mh[a + k]] = -(mu * d2);
        Kx[mr[k]] = st.r[k] * sm;
    }
    for (int k = tid; k < a + b; k += WG) Kx[mp[k]] = st.p[k] * sm;
    if (tid < 3) Kx[v.mapD[3 * c + tid]] = tid == 2 ? 1.0 : -1.0;
}
// feasibility of q (+ al dq) in the primal (dual == false) or dual cone, genpowcone.rs:279-317
__device__ bool gpw_feasible(const GpwState &st, int a, int b, bool dual, const double *q, const double *dq,
                             double al, double *red) {
    int bad = 0;
    double res = 0.0, sq = 0.0;
    for (int k = threadIdx.x; k < a; k += WG) {
        const double x = dq ? 1.0 * q[k] + al * dq[k] : q[k];
        if (!(x > 0.0)) bad = 1;
        res += 2.0 * st.alpha[k] * logsafe(dual ? x / st.alpha[k] : x);
    }
    for (int k = threadIdx.x; k < b; k += WG) {
        const double x = dq ? 1.0 * q[a + k] + al * dq[a + k] : q[a + k];
        sq += x * x;
    }
    if (__syncthreads_or(bad)) return false;
    res = block_sum(res, red);
    sq = block_sum(sq, red);
    return exp(res) - sq > 0.0;
}
// dual barrier (:333-356) of the vector sgn * (q + al dq)
__device__ double gpw_barrier_dual(const GpwState &st, int a, int b, const double *q, const double *dq, double al,
                                   double sgn, double *red) {
    double res = 0.0, sq = 0.0, lg = 0.0;
    for (int k = threadIdx.x; k < a; k += WG) {
        const double x = sgn * (dq ? 1.0 * q[k] + al * dq[k] : q[k]);
        res += 2.0 * st.alpha[k] * logsafe(x / st.alpha[k]);
        lg += logsafe(x) * (1.0 - st.alpha[k]);
    }
    for (int k = threadIdx.x; k < b; k += WG) {
        const double x = dq ? 1.0 * q[a + k] + al * dq[a + k] : q[a + k];
        sq += x * x;
    }
    res = block_sum(res, red);
    sq = block_sum(sq, red);
    lg = block_sum(lg, red);
    return -logsafe(exp(res) - sq) - lg;
}
//   OP 0 mul_Hs (:173-193)                 OP 1 copy (affine_ds :195-197, ds_from_dz_offset :206-208)
//   OP 2 combined_ds_shift = grad * sm (:199-204)
//   OP 3 step_length from sc (:210-233) -> partial[c]      OP 4 barrier at (z, s) + sc (dz, ds) (:235-250)
//   OP 5 unit_initialization (:127-135)
template <int OP>
__global__ __launch_bounds__(WG) void k_gpw_ops(GpwView v, double *o0, double *o1, const double *__restrict__ i0,
                                                const double *__restrict__ i1, const double *__restrict__ i2,
                                                const double *__restrict__ i3, double sc, double amin,
                                                double step, double *partial, double *work) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    if (c >= v.ncones) return;
    const int a = v.dim1[c], b = v.dim2[c], n = a + b, off = v.start[c], tid = threadIdx.x;
    const GpwState st = gpw_state(v, c);
    if (OP == 0) {
        const double *x = i0 + off;
        double cp = 0.0, cq = 0.0, cr = 0.0;
        for (int k = tid; k < n; k += WG) cp += st.p[k] * x[k];
        for (int k = tid; k < a; k += WG) cq += st.q[k] * x[k];
        for (int k = tid; k < b; k += WG) cr += st.r[k] * x[a + k];
        cp = block_sum(cp, red);
        cq = block_sum(cq, red);
        cr = block_sum(cr, red);
        const double mu = st.sc[1], d2 = st.sc[0];
        for (int k = tid; k < n; k += WG) {
            const double y = k < a ? st.d1[k] * x[k] - cq * st.q[k] : d2 * x[k] - cr * st.r[k - a];
            o0[off + k] = (cp * st.p[k] + 1.0 * y) * mu;
        }
    } else if (OP == 1) {
        for (int k = tid; k < n; k += WG) o0[off + k] = i0[off + k];
    } else if (OP == 2) {
        for (int k = tid; k < n; k += WG) o0[off + k] = st.grad[k] * sc;
    } else if (OP == 3) {
        double amin_z = sc, amin_s = sc;
        for (int pass = 0; pass < 2; ++pass) {
            const double *dq = (pass == 0 ? i0 : i1) + off, *q = (pass == 0 ? i2 : i3) + off;
            double al = sc;
            for (;;) {
                if (gpw_feasible(st, a, b, pass == 0, q, dq, al, red)) break;
                al *= step;
                if (al < amin) {
                    al = 0.0;
                    break;
                }
            }
            if (pass == 0) amin_z = al;
            else amin_s = al;
        }
        if (tid == 0) partial[c] = fmin(amin_z, amin_s);
    } else if (OP == 4) {
        // dual part, then the primal barrier = -f*(-g(s)) - degree with g from a Newton iteration (:409-485)
        const double bd = gpw_barrier_dual(st, a, b, i0 + off, i2 + off, sc, 1.0, red);
        const double *s = i1 + off, *ds = i3 + off;
        double *g = work + off; // scratch: the cone's slice of an m-vector
        double pr = 1.0, sq = 0.0;
        for (int k = tid; k < a; k += WG) pr *= pow(1.0 * s[k] + sc * ds[k], 2.0 * st.alpha[k]);
        double mx = 0.0;
        for (int k = tid; k < b; k += WG) mx = fmax(mx, fabs(1.0 * s[a + k] + sc * ds[a + k]));
        const double phi = block_prod(pr, red);
        mx = block_max(mx, red);
        for (int k = tid; k < b; k += WG) {
            const double x = mx > 0.0 ? (1.0 * s[a + k] + sc * ds[a + k]) / mx : 0.0;
            sq += x * x;
        }
        const double norm_r = mx * sqrt(block_sum(sq, red));
        const double eps = 2.220446049250313e-16;
        if (norm_r > eps) {
            const double psi = st.sc[2];
            double x = -(1.0 / norm_r) +
                       (psi * norm_r + sqrt((phi / norm_r / norm_r + psi * psi - 1.0) * phi)) / (phi - norm_r * norm_r);
            for (int iter = 0; iter < 100; iter++) {
                double df = 0.0, f = 0.0;
                for (int k = tid; k < a; k += WG) {
                    const double al = st.alpha[k], pk = 1.0 * s[k] + sc * ds[k];
                    df += 2.0 * al * norm_r / (norm_r * x + (1.0 + al) / al);
                    f += 2.0 * al * (logsafe(x * norm_r + (1.0 + al) / al) - logsafe(pk));
                }
                df = block_sum(df, red) + -(2.0 * x + 2.0 / norm_r) / (x * x + 2.0 * x / norm_r);
                f = block_sum(f, red) + -logsafe(2.0 * x / norm_r + x * x);
                const double dx = -f / df;
                if (dx < eps || fabs(dx / x) < sqrt(eps) || fabs(df) < eps) break;
                x += dx;
            }
            for (int k = tid; k < b; k += WG) g[a + k] = (x / norm_r) * st.r[k];
            for (int k = tid; k < a; k += WG)
                g[k] = -(1.0 + st.alpha[k] + st.alpha[k] * x * norm_r) / (1.0 * s[k] + sc * ds[k]);
        } else {
            for (int k = tid; k < b; k += WG) g[a + k] = 0.0;
            for (int k = tid; k < a; k += WG) g[k] = -(1.0 + st.alpha[k]) / (1.0 * s[k] + sc * ds[k]);
        }
        __syncthreads();
        const double bp = -gpw_barrier_dual(st, a, b, g, nullptr, 0.0, -1.0, red) - (double)(a + 1);
        if (tid == 0) partial[c] = (0.0 + bp) + bd;
    } else if (OP == 5) {
        for (int k = tid; k < n; k += WG) {
            const double u = k < a ? sqrt(1.0 + st.alpha[k]) : 0.0;
            o0[off + k] = u;
            o1[off + k] = u;
        }
    }
}

#undef wave_sum
#undef wave_max
#undef block_sum
#undef block_max

} // namespace

// ===========================================================================
// launch wrappers
// ===========================================================================
void cone_unit_shift(hipStream_t s, const int *nn_rows, int nn, const int *zero_rows, int nz, const SocView &v,
                     double *z, double alpha, int primal) {
    const int total = nn + nz + v.ncones;
    if (total) k_unit_shift<<<std::min(grid_for(total), 2048), WG, 0, s>>>(nn_rows, nn, zero_rows, nz, v.start, v.ncones, z, alpha, primal);
}
void cone_affine_ds(hipStream_t s, const int *nn_rows, int nn, const int *zero_rows, int nz, const SocView &v,
                    double *ds) {
    if (nn) k_nn_step_ops<0><<<grid_for(nn) > 2048 ? 2048 : grid_for(nn), WG, 0, s>>>(nn_rows, nn, v.w, v.lam, ds, nullptr, nullptr, nullptr, 0.0);
    if (nz) k_nn_step_ops<3><<<grid_for(nz) > 2048 ? 2048 : grid_for(nz), WG, 0, s>>>(zero_rows, nz, v.w, v.lam, ds, nullptr, nullptr, nullptr, 0.0);
    if (v.ncones) k_soc_step_ops<0><<<v.ncones, WG, 0, s>>>(v, ds, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, nullptr, nullptr);
}
void cone_combined_ds_shift(hipStream_t s, const int *nn_rows, int nn, const int *zero_rows, int nz,
                            const SocView &v, double *shift, double *step_z, double *step_s, double sigma_mu) {
    if (nn) k_nn_step_ops<1><<<grid_for(nn) > 2048 ? 2048 : grid_for(nn), WG, 0, s>>>(nn_rows, nn, v.w, v.lam, shift, step_z, step_s, nullptr, sigma_mu);
    if (nz) k_nn_step_ops<3><<<grid_for(nz) > 2048 ? 2048 : grid_for(nz), WG, 0, s>>>(zero_rows, nz, v.w, v.lam, shift, nullptr, nullptr, nullptr, 0.0);
    if (v.ncones) k_soc_step_ops<1><<<v.ncones, WG, 0, s>>>(v, shift, step_z, step_s, nullptr, nullptr, nullptr, nullptr, sigma_mu, nullptr, nullptr);
}
void cone_ds_from_dz_offset(hipStream_t s, const int *nn_rows, int nn, const int *zero_rows, int nz,
                            const SocView &v, double *out, const double *ds, const double *z) {
    if (nn) k_nn_step_ops<2><<<grid_for(nn) > 2048 ? 2048 : grid_for(nn), WG, 0, s>>>(nn_rows, nn, v.w, v.lam, out, const_cast<double *>(z), nullptr, ds, 0.0);
    if (nz) k_nn_step_ops<3><<<grid_for(nz) > 2048 ? 2048 : grid_for(nz), WG, 0, s>>>(zero_rows, nz, v.w, v.lam, out, nullptr, nullptr, nullptr, 0.0);
    if (v.ncones) k_soc_step_ops<2><<<v.ncones, WG, 0, s>>>(v, out, nullptr, nullptr, ds, z, nullptr, nullptr, 0.0, nullptr, nullptr);
}
int cone_step_length(hipStream_t s, const int *nn_rows, int nn, const SocView &v, const double *dz,
                     const double *ds, const double *z, const double *sv, double amax, double *partial,
                     int partial_cap) {
    int used = 0;
    if (nn) {
        int nb = (nn + WG - 1) / WG;
        if (nb > 1024) nb = 1024;
        if (nb > partial_cap) nb = partial_cap;
        k_nn_step_length<<<nb, WG, 0, s>>>(nn_rows, nn, dz, ds, z, sv, amax, partial);
        used = nb;
    }
    if (v.ncones) {
        k_soc_step_ops<3><<<v.ncones, WG, 0, s>>>(v, nullptr, nullptr, nullptr, dz, ds, z, sv, amax, partial + used, nullptr);
        used += v.ncones;
    }
    return used;
}
int cone_margins(hipStream_t s, const int *nn_rows, int nn, const SocView &v, const double *z, double *pmin,
                 double *psum, int partial_cap) {
    int used = 0;
    if (nn) {
        int nb = (nn + WG - 1) / WG;
        if (nb > 1024) nb = 1024;
        if (nb > partial_cap) nb = partial_cap;
        k_nn_margins<<<nb, WG, 0, s>>>(nn_rows, nn, z, pmin, psum);
        used = nb;
    }
    if (v.ncones) {
        k_soc_step_ops<4><<<v.ncones, WG, 0, s>>>(v, nullptr, nullptr, nullptr, z, nullptr, nullptr, nullptr, 0.0, pmin + used, psum + used);
        used += v.ncones;
    }
    return used;
}
void gpw_update_scaling(hipStream_t s, const GpwView &v, const double *zv, double mu) {
    if (v.ncones) k_gpw_update_scaling<<<v.ncones, WG, 0, s>>>(v, zv, mu);
}
void gpw_write_kkt(hipStream_t s, const GpwView &v, double *Kx) {
    if (v.ncones) k_gpw_write_kkt<<<v.ncones, WG, 0, s>>>(v, Kx);
}
void gpw_mul_hs(hipStream_t s, const GpwView &v, double *y, const double *x) {
    if (v.ncones) k_gpw_ops<0><<<v.ncones, WG, 0, s>>>(v, y, nullptr, x, nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, nullptr, nullptr);
}
void gpw_copy(hipStream_t s, const GpwView &v, double *out, const double *in) {
    if (v.ncones) k_gpw_ops<1><<<v.ncones, WG, 0, s>>>(v, out, nullptr, in, nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, nullptr, nullptr);
}
void gpw_combined_ds_shift(hipStream_t s, const GpwView &v, double *shift, double sigma_mu) {
    if (v.ncones) k_gpw_ops<2><<<v.ncones, WG, 0, s>>>(v, shift, nullptr, nullptr, nullptr, nullptr, nullptr, sigma_mu, 0.0, 0.0, nullptr, nullptr);
}
int gpw_step_length(hipStream_t s, const GpwView &v, const double *dz, const double *ds, const double *z,
                    const double *sv, double alpha, double alpha_min, double step, double *partial) {
    if (!v.ncones) return 0;
    k_gpw_ops<3><<<v.ncones, WG, 0, s>>>(v, nullptr, nullptr, dz, ds, z, sv, alpha, alpha_min, step, partial, nullptr);
    return v.ncones;
}
int gpw_barrier(hipStream_t s, const GpwView &v, const double *z, const double *sv, const double *dz,
                const double *ds, double alpha, double *partial, double *work) {
    if (!v.ncones) return 0;
    k_gpw_ops<4><<<v.ncones, WG, 0, s>>>(v, nullptr, nullptr, z, sv, dz, ds, alpha, 0.0, 0.0, partial, work);
    return v.ncones;
}
void gpw_unit_initialization(hipStream_t s, const GpwView &v, double *z, double *sv) {
    if (v.ncones) k_gpw_ops<5><<<v.ncones, WG, 0, s>>>(v, z, sv, nullptr, nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, nullptr, nullptr);
}
static int ns3_blocks(const Ns3View &v) { return (v.ncones + WG - 1) / WG; }
void ns3_affine_ds(hipStream_t s, const Ns3View &v, double *ds, const double *sv) {
    if (v.ncones) k_ns3_step_ops<0><<<ns3_blocks(v), WG, 0, s>>>(v, ds, nullptr, sv, nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, nullptr);
}
void ns3_combined_ds_shift(hipStream_t s, const Ns3View &v, double *shift, const double *step_z,
                           const double *step_s, double sigma_mu) {
    if (v.ncones) k_ns3_step_ops<1><<<ns3_blocks(v), WG, 0, s>>>(v, shift, nullptr, step_z, step_s, nullptr, nullptr, sigma_mu, 0.0, 0.0, nullptr);
}
void ns3_ds_from_dz_offset(hipStream_t s, const Ns3View &v, double *out, const double *ds) {
    if (v.ncones) k_ns3_step_ops<2><<<ns3_blocks(v), WG, 0, s>>>(v, out, nullptr, ds, nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, nullptr);
}
int ns3_step_length(hipStream_t s, const Ns3View &v, const double *dz, const double *ds, const double *z,
                    const double *sv, double alpha, double alpha_min, double step, double *partial) {
    if (!v.ncones) return 0;
    k_ns3_step_ops<3><<<ns3_blocks(v), WG, 0, s>>>(v, nullptr, nullptr, dz, ds, z, sv, alpha, alpha_min, step, partial);
    return ns3_blocks(v);
}
int cone_barrier(hipStream_t s, const int *nn_rows, int nn, const SocView &soc, const Ns3View &v, const double *z,
                 const double *sv, const double *dz, const double *ds, double alpha, double *partial) {
    int used = 0;
    if (nn) {
        int nb = (nn + WG - 1) / WG;
        if (nb > 1024) nb = 1024;
        k_nn_barrier<<<nb, WG, 0, s>>>(nn_rows, nn, z, sv, dz, ds, alpha, partial);
        used = nb;
    }
    if (soc.ncones) {
        k_soc_barrier<<<soc.ncones, WG, 0, s>>>(soc, z, sv, dz, ds, alpha, partial + used);
        used += soc.ncones;
    }
    if (v.ncones) {
        k_ns3_step_ops<4><<<ns3_blocks(v), WG, 0, s>>>(v, nullptr, nullptr, z, sv, dz, ds, alpha, 0.0, 0.0, partial + used);
        used += ns3_blocks(v);
    }
    return used;
}
void cone_unit_initialization(hipStream_t s, const int *nn_rows, int nn, const SocView &soc, const Ns3View &v,
                              double *z, double *sv, int m) {
    if (m) {
        (void)hipMemsetAsync(z, 0, (size_t)m * sizeof(double), s);
        (void)hipMemsetAsync(sv, 0, (size_t)m * sizeof(double), s);
    }
    const int total = nn + soc.ncones;
    if (total) k_sym_unit_init<<<std::min(grid_for(total), 2048), WG, 0, s>>>(nn_rows, nn, soc.start, soc.ncones, z, sv);
    if (v.ncones) k_ns3_step_ops<5><<<ns3_blocks(v), WG, 0, s>>>(v, z, sv, nullptr, nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, nullptr);
}
void scatter_init(hipStream_t s, const double *Kx, const int *a2l, int nnzK, int nnzL, double *Lx,
                  double *D, const int8_t *dsigns, const double *eps, const int *fill_idx, int nfill,
                  int *status) {
    int nb = grid_for(nnzK > 0 ? nnzK : 1);
    if (nb > 4096) nb = 4096;
    k_scatter_init<<<nb, WG, 0, s>>>(Kx, a2l, nnzK, nnzL, Lx, D, dsigns, eps, fill_idx, nfill, status);
}
void gather_values(hipStream_t s, double *Sx, const double *Kx, const int *Smap, int nnzS) {
    if (nnzS == 0) return;
    int nb = grid_for(nnzS);
    if (nb > 4096) nb = 4096;
    k_gather_values<<<nb, WG, 0, s>>>(Sx, Kx, Smap, nnzS);
}
void scatter_values(hipStream_t s, double *Kx, const int *map, const double *vals, int k, double scale) {
    if (k == 0) return;
    int nb = (k + WG - 1) / WG;
    if (nb > 4096) nb = 4096;
    k_scatter_values<<<nb, WG, 0, s>>>(Kx, map, vals, k, scale);
}
void diag_absmax_eps(hipStream_t s, const double *Kx, const int *didx, int N, double c, double prop,
                     double *scal) {
    (void)hipMemsetAsync(scal, 0, 3 * sizeof(double), s);
    if (N > 0) {
        int nb = (N + WG - 1) / WG;
        if (nb > 1024) nb = 1024;
        k_diag_absmax<<<nb, WG, 0, s>>>(Kx, didx, N, (unsigned long long *)scal);
    }
    k_eps_from_max<<<1, 1, 0, s>>>(c, prop, (unsigned long long *)scal);
}

void eps_from_slots(hipStream_t s, unsigned long long *slots, double c, double prop, double static_max,
                    double *scal) {
    k_eps_from_slots<<<1, 64, 0, s>>>(slots, c, prop, static_max, scal);
}

void factor_T(hipStream_t s, const LdlView &v, ListView c) {
    if (c.count) k_factor_T<<<grid_for(c.count), WG, 0, s>>>(v, c.idx, c.count);
}
void factor_W(hipStream_t s, const LdlView &v, ListView c) {
    if (c.count) k_factor_W<<<c.count, 1024, 0, s>>>(v, c.idx, c.count);
}
static size_t bundle_lds(const BundleView &bv) { return ((size_t)bv.max_nodes * sizeof(double) + 15) & ~(size_t)15; }
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
    if (hipFuncSetAttribute((const void *)k_bundle_factor_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_bundle_factor_flat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}
void bundle_factor(hipStream_t s, const LdlView &v, const BundleView &bv, const FoldView &fold, int lds_doubles) {
    if (!bv.nb) return;
    static const bool no_flat = std::getenv("CHIP_NO_FACTOR_FLAT") != nullptr;
    if (lds_doubles > 0 && v.fu_rec && !no_flat) k_bundle_factor_flat<<<bv.nb, FFWG, factor_lds_bytes(lds_doubles), s>>>(v, bv, fold);
    else if (lds_doubles > 0) k_bundle_factor_lds<<<bv.nb, FLWG, factor_lds_bytes(lds_doubles), s>>>(v, bv, fold, lds_doubles);
    else k_bundle_factor<<<bv.nb, BWG, 0, s>>>(v, bv, fold);
}
void gfold_top_factor(hipStream_t s, const LdlView &v, const BundleView &bv, const GFoldView &gf) {
    if (gf.ng <= 0) return;
    k_gfold_schur<<<bv.nb, GSWG, 0, s>>>(v, bv, gf);
    k_gfold_top_factor<<<gf.ng, 64, 0, s>>>(v, gf);
}
void bundle_fwd(hipStream_t s, const LdlView &v, const BundleView &bv, double *x, const FoldView &fold) {
    if (bv.nb) k_bundle_fwd<<<bv.nb, BWG, bundle_lds(bv), s>>>(v, bv, x, fold);
}
void bundle_bwd(hipStream_t s, const LdlView &v, const BundleView &bv, double *x, const double *addv) {
    if (bv.nb) k_bundle_bwd<<<bv.nb, BWG, bundle_lds(bv), s>>>(v, bv, x, addv);
}
int ir_ctl_ints() { return IR_CTL_INTS; }
size_t ir_part_doubles(int nb, int k) { return (size_t)nb * (3 + 3 * (size_t)k) + 72; }
// workgroup size of k_bundle_ir for these bundles and the largest co-resident grid (0: the kernel cannot run)
static size_t bundle_ir_lds(const BundleView &bv) {
    return ((size_t)std::max(bv.max_nodes, bv.ir_lds_doubles) * sizeof(double) + 15) & ~(size_t)15;
}
template <int TW> static int bundle_ir_capacity_tw(const BundleView &bv) {
    const size_t lds = bundle_ir_lds(bv);
    if (hipFuncSetAttribute((const void *)k_bundle_ir<TW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_bundle_ir<TW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
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
    static const bool no1024 = std::getenv("CHIP_NO_IR1024") != nullptr;
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
int bundle_ir(hipStream_t s, const LdlView &v, const BundleView &bv, const FoldView &fold, const IrView &ir, int grid,
              int tw, const GFoldView &gf) {
    // grid <= bundle_ir_capacity(): every workgroup is resident on an otherwise idle device, and a grid
    // barrier that cannot complete times out instead of hanging
    const size_t lds = bundle_ir_lds(bv);
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
void factor_B(hipStream_t s, const LdlView &v, ChunkView c) {
    if (c.count) k_factor_B<<<c.count, WG, 0, s>>>(v, c.row, c.beg, c.end, c.count);
}
static size_t snode_solve_lds_bytes(int wmax, int nbcap) {
    return (size_t)(wmax + nbcap + SN_NB * SN_NB + SN_NB) * sizeof(double) + (size_t)wmax * sizeof(int);
}
static size_t snode_lds_bytes(int wmax) { return (size_t)(SN_KC * SN_NB) * sizeof(double) + (size_t)wmax * sizeof(int); }
int snode_kernel_attributes(int wmax, int nbmax) {
    const int lds = (int)snode_lds_bytes(wmax);
    int rc = (int)hipFuncSetAttribute((const void *)k_snode_update, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (!rc) rc = (int)hipFuncSetAttribute((const void *)k_snode_extend, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int lds2 = (int)snode_solve_lds_bytes(wmax, std::min(nbmax, SN_XB_CAP));
    if (!rc) rc = (int)hipFuncSetAttribute((const void *)k_snode_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    if (!rc) rc = (int)hipFuncSetAttribute((const void *)k_snode_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    return rc;
}
// wlvl / nblvl: maxima over the supernodes of this launch.  Levels with a large B part run it in
// separate multi-workgroup launches: one workgroup per supernode is latency bound.
void solve_snodes(hipStream_t s, GatherMode m, const LdlView &v, const SnodeView &sv, const int *order, int count,
                  int wmax_all, int nbmax_all, int wlvl, int nblvl, double *x, const SnodeTriView *tri,
                  const LaunchProf *lp) {
    if (!count) return;
    if (tri && tri->msg && wlvl > 2 * SN_NB) {
        if (lp) lp->begin(lp->ctx, PFK_SN_TRI);
        // wide supernodes: the triangle by several workgroups per supernode (k_snode_tri), the rows of B by
        // their own multi-workgroup launches
        const int nblkmax = (wlvl + SN_NB - 1) / SN_NB;
        if (m == FWD) {
            k_snode_tri<true><<<dim3(nblkmax, count), SN2_WG, 0, s>>>(v, sv, order, tri->blk_ptr, tri->msg, tri->epoch, x,
                                                                    tri->timeout_flag);
        } else {
            k_snode_tri<false><<<dim3(nblkmax, count), SN2_WG, 0, s>>>(v, sv, order, tri->blk_ptr, tri->msg, tri->epoch, x,
                                                                     tri->timeout_flag);
        }
        (void)nblvl;
        if (lp) lp->end(lp->ctx, PFK_SN_TRI);
        return;
    }
    int cap = SN_XB_CAP;
    if (const char *e = std::getenv("CHIP_SN_XB_CAP")) cap = std::max(1, std::min(SN_XB_CAP, std::atoi(e))); // tests
    const int nbcap = std::min(nbmax_all, cap);
    const size_t lds = snode_solve_lds_bytes(wmax_all, nbcap);
    const bool split = nblvl >= 256;
    if (m == FWD) {
        k_snode_fwd<<<count, SN_WG, lds, s>>>(v, sv, order, x, wmax_all, nbcap, split ? 0 : 1);
        if (split)
            k_snode_push<<<dim3((nblvl + SN_WG - 1) / SN_WG, (wlvl + SN_PCH - 1) / SN_PCH, count), SN_WG, 0, s>>>(
                v, sv, order, x);
    } else {
        if (split)
            k_snode_pull<<<dim3((wlvl + SN_NB - 1) / SN_NB, count), SN_WG, (size_t)nbcap * sizeof(double), s>>>(
                v, sv, order, x, nbcap);
        k_snode_bwd<<<count, SN_WG, lds, s>>>(v, sv, order, x, wmax_all, nbcap, split ? 0 : 1);
    }
}
// all supernodes order[0..count) of one unit level: block columns one after the other, then their
// updates of the ancestors.  nblk / hmax / nbmax: maxima over these supernodes.
namespace {
// CHIP_SN_DEBUG=1: every launch of the update tiles / the panel kernel is followed by a synchronisation and the
// stamps of its workgroup 0 are accumulated; the per-phase means are printed when the process ends
struct SnDebug {
    long long *dev = nullptr;
    double sum[2][8] = {};
    long n[2] = {0, 0};
    bool on = false;
    SnDebug() {
        on = std::getenv("CHIP_SN_DEBUG") != nullptr;
        mode = on ? std::atoi(std::getenv("CHIP_SN_DEBUG")) : 0;
        if (on) {
            (void)hipMalloc((void **)&dev, (64 + (size_t)RING * 32) * sizeof(long long));
            (void)hipMemset(dev, 0, (64 + (size_t)RING * 32) * sizeof(long long));
        }
    }
    void collect(hipStream_t s, int kind) { // kind 0: panel (slots 0..7), 1: update tiles (slots 16..20)
        long long t[64];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(t, dev, sizeof(t), hipMemcpyDeviceToHost);
        const int base = kind ? 16 : 0, cnt = kind ? 5 : 8;
        for (int i = 1; i < cnt; i++)
            if (t[base + i] && t[base + i - 1]) sum[kind][i] += (t[base + i] - t[base + i - 1]) * 0.01;
        n[kind]++;
        (void)hipMemset(dev, 0, 64 * sizeof(long long));
    }
    // CHIP_SN_DEBUG=2: no synchronisation; launch k stamps into its own 32 slots of a ring, and at the end the time
    // between the LAST stamp of a launch and the FIRST stamp of the next one (what a kernel boundary costs) is printed
    static constexpr int RING = 2048;
    int mode = 0;
    long nring = 0;
    std::vector<int> kinds;
    long long *ring_slot(int kind) {
        if ((nring % RING) == 0 && nring) flush_ring();
        kinds.push_back(kind);
        return dev + 64 + (size_t)(nring++ % RING) * 32;
    }
    double gap_sum[2] = {0, 0}, in_sum[2] = {0, 0}, skew_sum[2] = {0, 0};
    long gap_n[2] = {0, 0};
    void flush_ring() {
        (void)hipDeviceSynchronize();
        std::vector<long long> t((size_t)RING * 32);
        (void)hipMemcpy(t.data(), dev + 64, t.size() * sizeof(long long), hipMemcpyDeviceToHost);
        const long cnt = (long)kinds.size();
        long long prev_last = 0;
        for (long k = 0; k < cnt; k++) {
            const long long *r = t.data() + (size_t)k * 32;
            const int base = kinds[k] ? 16 : 0, hi = kinds[k] ? 20 : 7;
            long long first = r[base], last = 0;
            for (int i = base; i <= hi; i++) last = std::max(last, r[i]);
            const long long wg0_last = last;
            last = std::max(last, r[kinds[k] ? 21 : 8]); // (the end of the launch's LAST workgroup)
            if (first) skew_sum[kinds[k]] += (last - wg0_last) * 0.01;
            if (first && prev_last && first > prev_last && first - prev_last < 100000 && k > 0 && kinds[k - 1] != kinds[k]) { // (update <-> panel pairs of one block column)
                gap_sum[kinds[k]] += (first - prev_last) * 0.01;
                in_sum[kinds[k]] += (last - first) * 0.01;
                gap_n[kinds[k]]++;
                for (int i = base + 1; i <= hi; i++)
                    if (r[i] && r[i - 1]) sum[kinds[k]][i - base] += (r[i] - r[i - 1]) * 0.01;
                n[kinds[k]]++;
            }
            prev_last = last;
        }
        kinds.clear();
        (void)hipMemset(dev + 64, 0, (size_t)RING * 32 * sizeof(long long));
    }
    ~SnDebug() {
        if (on && mode == 2) {
            flush_ring();
            for (int k = 0; k < 2; k++)
                if (gap_n[k])
                    std::fprintf(stderr, "[chip sn debug] %s: mean over %ld launches: %.2f us between the END of the previous launch's last workgroup and this one's first stamp, %.2f us from there to the end of its last workgroup (workgroup 0 ends %.2f us before the last one)\n",
                                 k ? "k_snode_update" : "k_snode_panel", gap_n[k], gap_sum[k] / gap_n[k], in_sum[k] / gap_n[k], skew_sum[k] / std::max(1L, n[k]));
        }
        if (!on) return;
        const char *pn[8] = {"", "geometry+colbase", "block loads", "block factorisation", "write-back", "row loads", "rows recurrence", "stores"};
        const char *un[5] = {"", "colbase", "first operand staged", "matrix instructions (all chunks)", "emit"};
        if (n[0]) {
            std::fprintf(stderr, "[chip sn debug] k_snode_panel, workgroup 0, mean over %ld launches (us):", n[0]);
            for (int i = 1; i < 8; i++) std::fprintf(stderr, " %s %.2f;", pn[i], sum[0][i] / n[0]);
            std::fprintf(stderr, "\n");
        }
        if (n[1]) {
            std::fprintf(stderr, "[chip sn debug] k_snode_update, workgroup 0, mean over %ld launches (us):", n[1]);
            for (int i = 1; i < 5; i++) std::fprintf(stderr, " %s %.2f;", un[i], sum[1][i] / n[1]);
            std::fprintf(stderr, "\n");
        }
    }
};
SnDebug &sn_debug() {
    static SnDebug d;
    return d;
}
} // namespace
void factor_snodes(hipStream_t s, const LdlView &v, const SnodeView &sv_in, const int *order, int count, int wmax_all,
                   int nblk, int hmax, int nbmax, const LaunchProf *lp) {
    if (!count) return;
    SnodeView sv = sv_in;
    SnDebug &dbg = sn_debug();
    if (dbg.on) sv.dbg = dbg.dev;
    const size_t lds = snode_lds_bytes(wmax_all);
    auto pb = [&](int f) { if (lp) lp->begin(lp->ctx, f); };
    auto pe = [&](int f) { if (lp) lp->end(lp->ctx, f); };
    for (int b = 0; b < nblk; ++b) {
        if (b > 0) {
            const int rows = hmax - b * SN_NB;
            if (rows > 0) {
                const int groups = (rows + SN_ROWS - 1) / SN_ROWS;
                int ksplit = 1; // fill the chip when the level has few supernodes: the finished columns in shares of whole block columns
                // (CHIP_NO_SPLITK: no split -> no fp64 atomics between the splits, a fixed summation order)
                static const bool no_splitk = std::getenv("CHIP_NO_SPLITK") != nullptr;
                static const int split_target = std::getenv("CHIP_SN_SPLIT_TARGET") ? std::atoi(std::getenv("CHIP_SN_SPLIT_TARGET")) : 256;
                static const int split_max = std::getenv("CHIP_SN_SPLIT_MAX") ? std::atoi(std::getenv("CHIP_SN_SPLIT_MAX")) : 8;
                static const int split_unit = std::getenv("CHIP_SN_SPLIT_UNIT") ? std::atoi(std::getenv("CHIP_SN_SPLIT_UNIT")) : 1; // block columns per share, at least
                while (!no_splitk && ksplit < split_max && ksplit * 2 * split_unit <= b && groups * count * ksplit < split_target) ksplit *= 2;
                pb(PFK_SN_UPDATE);
                if (dbg.mode == 2) sv.dbg = dbg.ring_slot(1) - 16 + 16; // (slots 16..20 of the launch's 32)
                k_snode_update<<<dim3(groups, count, ksplit), SN_WG, lds, s>>>(v, sv, order, b);
                pe(PFK_SN_UPDATE);
                if (dbg.on && dbg.mode != 2) dbg.collect(s, 1);
            }
        }
        const bool no_panel = std::getenv("CHIP_NO_SNODE_PANEL") != nullptr; // (read per call: the tests switch forms inside one process)
        if (!no_panel) { // the diagonal block and the rows below it in one launch of one-wave workgroups
            const int below = hmax - b * SN_NB - 1;
            pb(PFK_SN_DIAG);
            if (dbg.mode == 2) sv.dbg = dbg.ring_slot(0);
            // (bit 0: rows phase on the matrix cores; bit 1: block factorisation on the matrix cores)
            const int panel_mode = (std::getenv("CHIP_NO_PANEL_MFMA") ? 0 : 1) | (std::getenv("CHIP_NO_PANEL_DIAG_MFMA") ? 0 : 2);
            k_snode_panel<<<dim3(std::max(1, (below + SNP_WG - 1) / SNP_WG), count), SNP_WG, 0, s>>>(v, sv, order, b, panel_mode);
            pe(PFK_SN_DIAG);
            if (dbg.on && dbg.mode != 2) dbg.collect(s, 0);
            continue;
        }
        pb(PFK_SN_DIAG);
        k_snode_diag<<<count, SN_DWG, 0, s>>>(v, sv, order, b);
        pe(PFK_SN_DIAG);
        const int below = hmax - b * SN_NB - 1; // (a narrow last block leaves more rows below it)
        if (below > 0) {
            pb(PFK_SN_ROWS);
            k_snode_rows<<<dim3((below + SN_RWG - 1) / SN_RWG, count), SN_RWG, 0, s>>>(v, sv, order, b);
            pe(PFK_SN_ROWS);
        }
    }
    if (nbmax > 0 && sv.upd_slot) {
        pb(PFK_SN_EXTEND);
        k_snode_extend<<<dim3((nbmax + SN_ROWS - 1) / SN_ROWS, (nbmax + SN_NB - 1) / SN_NB, count), SN_WG, lds, s>>>(
            v, sv_in, order);
        pe(PFK_SN_EXTEND);
    }
}
__global__ void k_debug_spin(long long ticks) {
    extern __shared__ char spin_lds[];
    if (ticks < 0) spin_lds[threadIdx.x] = 0; // (keeps the dynamic LDS allocation alive)
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
void debug_spin(hipStream_t s, int blocks, int threads, int lds_bytes, double usec) {
    if (lds_bytes > 65536)
        (void)hipFuncSetAttribute((const void *)k_debug_spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    k_debug_spin<<<blocks, threads, (size_t)lds_bytes, s>>>((long long)(usec * 100.0)); // 100 MHz clock
}
void factor_finalize(hipStream_t s, const LdlView &v, ListView c) {
    if (c.count) k_factor_finalize<<<c.count, WG, 0, s>>>(v, c.idx, c.count);
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
    (void)hipFuncSetAttribute((const void *)k_topblk_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_topblk_build<<<tb.nblocks, WG, lds, s>>>(v, tb);
}
// kernels of the solve sequence that need more than 64 KB of dynamic LDS: allowed once per process,
// outside any stream capture
void solve_kernel_attributes() {
    const size_t lds = (size_t)(TOPBLK_PACK + TOPBLK) * sizeof(double);
    (void)hipFuncSetAttribute((const void *)k_topblk_step<FWD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)k_topblk_step<BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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
void factor_chain(hipStream_t s, const LdlView &v, const int *t_idx, const int *t_ptr, const int *w_idx,
                  const int *w_ptr, int l0, int l1) {
    if (l1 > l0) k_factor_chain<<<1, 1024, 0, s>>>(v, t_idx, t_ptr, w_idx, w_ptr, l0, l1);
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
static int stream_grid(int N) {
    int nb = grid_for(N);
    return nb > 2048 ? 2048 : nb;
}
void permute_in(hipStream_t s, double *y, const double *b, const int *perm, int N) {
    if (N) k_permute_in<<<stream_grid(N), WG, 0, s>>>(y, b, perm, N);
}
void permute_out(hipStream_t s, double *x, const double *y, const int *perm, int N) {
    if (N) k_permute_out<<<stream_grid(N), WG, 0, s>>>(x, y, perm, N);
}
void setrhs_perm(hipStream_t s, double *bp, double *xi, const double *rx, const double *rz, const int *perm,
                 int n, int m, int N, unsigned long long *nrm, int *nan) {
    if (N) k_setrhs_perm<<<stream_grid(N), WG, 0, s>>>(bp, xi, rx, rz, perm, n, m, N, nrm, nan);
}
void norm_rows(hipStream_t s, const double *v, ListView rows, unsigned long long *nrm, int *nan) {
    if (rows.count) k_norm_rows<<<(rows.count + WG - 1) / WG, WG, 0, s>>>(v, rows.idx, rows.count, nrm, nan);
}
void getlhs_perm(hipStream_t s, double *lx, double *lz, const double *xp, const int *iperm, int n, int m) {
    if (n + m) k_getlhs_perm<<<stream_grid(n + m), WG, 0, s>>>(lx, lz, xp, iperm, n, m);
}
void waxpby(hipStream_t s, double *w, double a, const double *x, double b, const double *y, int n) {
    if (n) k_waxpby<<<stream_grid(n), WG, 0, s>>>(w, a, x, b, y, n);
}
int dot_scratch_doubles() { return DOT_BLOCKS; }
void dot(hipStream_t s, const double *a, const double *b, int n, double *out, double *scratch) {
    const int nb = n > 0 ? std::min(DOT_BLOCKS, (n + WG - 1) / WG) : 0;
    if (nb) k_dot_partial<<<nb, WG, 0, s>>>(a, b, n, scratch);
    k_dot_final<<<1, WG, 0, s>>>(scratch, nb, out);
}
int multi_dot_scratch_doubles() { return DOT_BATCH_MAX * DOT_BLOCKS; }
void multi_dot(hipStream_t s, const DotBatch &bt, double *out, double *scratch) {
    if (bt.count <= 0) return;
    int nbmax = 0;
    for (int k = 0; k < bt.count; k++)
        if (bt.s[k].n > 0) nbmax = std::max(nbmax, std::min(DOT_BLOCKS, (bt.s[k].n + WG - 1) / WG));
    if (nbmax) k_multi_dot_partial<<<dim3(nbmax, bt.count), WG, 0, s>>>(bt, scratch);
    k_multi_dot_final<<<bt.count, WG, 0, s>>>(bt, scratch, out);
}
void lin3(hipStream_t s, double *w, double a, const double *x, double b, const double *y, double c, const double *z,
          int n) {
    if (n) k_lin3<<<stream_grid(n), WG, 0, s>>>(w, a, x, b, y, c, z, n);
}
void dot_shifted(hipStream_t s, const double *z, const double *sv, const double *dz, const double *ds, double alpha,
                 int n, double *out, double *scratch) {
    const int nb = n > 0 ? std::min(DOT_BLOCKS, (n + WG - 1) / WG) : 0;
    if (nb) k_dot_shifted_partial<<<nb, WG, 0, s>>>(z, sv, dz, ds, alpha, n, scratch);
    k_dot_final<<<1, WG, 0, s>>>(scratch, nb, out);
}
void add_vec(hipStream_t s, double *dx, const double *x, int N) {
    if (N) k_add_vec<<<stream_grid(N), WG, 0, s>>>(dx, x, N);
}
void norm_inf(hipStream_t s, const double *v, int N, unsigned long long *out, int *nanflag) {
    if (N) k_norm_inf<<<stream_grid(N), WG, 0, s>>>(v, N, out, nanflag);
}

// Nonnegative rows + second-order cones in one launch each (scaling; Hs values)
static int nn_blocks(int count) { return count ? std::min((count + WG - 1) / WG, 2048) : 0; }
void sym_update_scaling(hipStream_t s, const SocView &v, const int *nn_rows, int nn, const double *sv,
                        const double *zv, double *w, double *lam) {
    const int grid = v.ncones + nn_blocks(nn);
    if (grid) k_sym_update_scaling<<<grid, WG, 0, s>>>(v, nn_rows, nn, sv, zv, w, lam);
}
void sym_write_kkt(hipStream_t s, const SocView &v, const int *nn_rows, const int *nn_hsidx, int nn, const double *w,
                   const int *mapHs, double *Kx, unsigned long long *dslots) {
    const int grid = v.ncones + nn_blocks(nn);
    if (grid) k_sym_write_kkt<<<grid, WG, 0, s>>>(v, nn_rows, nn_hsidx, nn, w, mapHs, Kx, dslots);
}
void psd_update_scaling(hipStream_t s, const PsdView &v, const double *sv, const double *zv) {
    if (!v.ncones) return;
    if (v.scratch) { // cones too large for LDS: work matrices in HBM scratch
        k_psd_update_scaling<true><<<v.ncones, WG, 0, s>>>(v, sv, zv);
        return;
    }
    const size_t lds = ((size_t)(4 * v.maxdim * v.maxdim + 3 * v.maxdim) * sizeof(double) + 15) & ~(size_t)15;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_psd_update_scaling<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_psd_update_scaling<false><<<v.ncones, WG, lds, s>>>(v, sv, zv);
}
void psd_write_hs(hipStream_t s, const PsdView &v, double *Kx) {
    if (!v.ncones) return;
    if (v.scratch) {
        const int bpc = 64; // numel^2 / 2 entries per cone: 3.4e7 at n = 128
        k_psd_write_hs<true><<<v.ncones * bpc, WG, 0, s>>>(v, Kx, bpc);
        return;
    }
    const int bpc = 16;
    const size_t lds = ((size_t)(v.maxdim * v.maxdim) * sizeof(double) + 15) & ~(size_t)15;
    k_psd_write_hs<false><<<v.ncones * bpc, WG, lds, s>>>(v, Kx, bpc);
}
static size_t psd_ops_lds(const PsdView &v) {
    return ((size_t)(3 * v.maxdim * v.maxdim + 4 * v.maxdim + 8) * sizeof(double) + 15) & ~(size_t)15;
}
template <typename K> static void psd_allow_lds(K kernel, size_t lds) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
#define PSD_LAUNCH(OP, ...)                                                          \
    do {                                                                             \
        if (v.scratch) {                                                             \
            k_psd_ops<OP, true><<<v.ncones, WG, 0, s>>>(v, __VA_ARGS__);             \
        } else {                                                                     \
            const size_t lds_ = psd_ops_lds(v);                                      \
            psd_allow_lds(k_psd_ops<OP, false>, lds_);                               \
            k_psd_ops<OP, false><<<v.ncones, WG, lds_, s>>>(v, __VA_ARGS__);         \
        }                                                                            \
    } while (0)
void psd_mul_hs(hipStream_t s, const PsdView &v, double *y, const double *x) {
    if (v.ncones) PSD_LAUNCH(0, y, nullptr, nullptr, x, nullptr, nullptr, nullptr, 0.0, nullptr, nullptr);
}
void psd_affine_ds(hipStream_t s, const PsdView &v, double *ds) {
    if (v.ncones) PSD_LAUNCH(1, ds, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, nullptr, nullptr);
}
void psd_combined_ds_shift(hipStream_t s, const PsdView &v, double *shift, double *step_z, double *step_s,
                           double sigma_mu) {
    if (v.ncones) PSD_LAUNCH(2, shift, step_z, step_s, nullptr, nullptr, nullptr, nullptr, sigma_mu, nullptr, nullptr);
}
void psd_ds_from_dz_offset(hipStream_t s, const PsdView &v, double *out, const double *ds) {
    if (v.ncones) PSD_LAUNCH(3, out, nullptr, nullptr, ds, nullptr, nullptr, nullptr, 0.0, nullptr, nullptr);
}
int psd_step_length(hipStream_t s, const PsdView &v, const double *dz, const double *ds, double amax,
                    double *partial) {
    if (!v.ncones) return 0;
    PSD_LAUNCH(4, nullptr, nullptr, nullptr, dz, ds, nullptr, nullptr, amax, partial, nullptr);
    return v.ncones;
}
int psd_margins(hipStream_t s, const PsdView &v, const double *z, double *pmin, double *psum) {
    if (!v.ncones) return 0;
    PSD_LAUNCH(5, nullptr, nullptr, nullptr, z, nullptr, nullptr, nullptr, 0.0, pmin, psum);
    return v.ncones;
}
int psd_barrier(hipStream_t s, const PsdView &v, const double *z, const double *sv, const double *dz,
                const double *ds, double alpha, double *partial) {
    if (!v.ncones) return 0;
    PSD_LAUNCH(6, nullptr, nullptr, nullptr, z, sv, dz, ds, alpha, partial, nullptr);
    return v.ncones;
}
void psd_unit_shift(hipStream_t s, const PsdView &v, double *z, double alpha) {
    if (v.ncones) k_psd_diag<<<v.ncones, WG, 0, s>>>(v, z, nullptr, alpha, 0);
}
void psd_unit_initialization(hipStream_t s, const PsdView &v, double *z, double *sv) {
    if (v.ncones) k_psd_diag<<<v.ncones, WG, 0, s>>>(v, z, sv, 0.0, 1);
}
void ns3_update_scaling(hipStream_t s, const Ns3View &v, const double *sv, const double *zv, double mu,
                        int strategy) {
    if (v.ncones) k_ns3_update_scaling<<<(v.ncones + WG - 1) / WG, WG, 0, s>>>(v, sv, zv, mu, strategy);
}
void ns3_write_hs(hipStream_t s, const Ns3View &v, double *Kx) {
    if (v.ncones) k_ns3_write_hs<<<(v.ncones * 6 + WG - 1) / WG, WG, 0, s>>>(v, Kx);
}
void ns3_mul_hs(hipStream_t s, const Ns3View &v, double *y, const double *x) {
    if (v.ncones) k_ns3_mul_hs<<<(v.ncones + WG - 1) / WG, WG, 0, s>>>(v, y, x);
}
void cones_mul_Hs(hipStream_t s, const int *nn_rows, int nn_count, const SocView &v,
                  const int *zero_rows, int zero_count, double *y, const double *x) {
    if (nn_count) k_nn_mul_hs<<<stream_grid(nn_count), WG, 0, s>>>(nn_rows, nn_count, v.w, y, x, 0);
    if (zero_count) k_nn_mul_hs<<<stream_grid(zero_count), WG, 0, s>>>(zero_rows, zero_count, v.w, y, x, 1);
    if (v.ncones) k_soc_mul_hs<<<v.ncones, WG, 0, s>>>(v, y, x);
}

} // namespace dev
} // namespace chip
