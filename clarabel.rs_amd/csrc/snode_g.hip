// snode_g.hip -- chain supernodes of moderate width: substitutions WITHOUT a dependency chain inside a supernode
// (one of the translation units behind kernels.hpp; geometry and helpers in snode_common.hpp).
//
// The substitutions through a supernode S (w member columns, nb rows of B below; qdldl.rs:708-768 restricted to the
// dense trapezoid) are
//     forward :  x_S <- T^-1 x_S ,   x_B <- x_B - L_B x_S          T = I + strict lower triangle of L_SS
//     backward:  x_S <- T^-T (D^-1 x_S - L_B' x_B)
// k_snode_tri walks T block by block: one hop of ~4 us per 64 columns, 129 dependent hops per sweep on BASELINE
// config 2, twelve sweeps per step.  Here the chain is paid ONCE per refactorisation: k_snode_ginv forms
//     G = [ I ; L_B ] T^-1        ((w + nb) x w:  T^-1 on top,  M = L_B T^-1 below)
// and a sweep through S is ONE pass over G with no order among its rows / columns:
//     forward :  [ x_S ; dx_B ] = G x_S(old)      x_B -= dx_B                       (k_snode_gfwd: a row per lane)
//     backward:  x_S = G' [ D^-1 x_S ; -x_B ]                                       (k_snode_gbwd: a column per wave)
// Every ROW of G is independent of the others (row i solves y T = p with p = e_i or row i of L_B): the build is
// embarrassingly parallel over rows -- 256 rows per workgroup, column blocks from the last to the first:
//     Y[:, J_c] = ( P[:, J_c] - sum_{k > c} Y[:, J_k] T[J_k, J_c] ) T_cc^-1
// the sum on the f64 matrix cores (the finished column blocks of G streamed as the A operand, T staged in LDS), the
// 64 x 64 triangular part by the 16-blocked recurrence of the panel kernels, mirrored (columns in reverse order).
// Results of a sweep differ from the substitution's in rounding only (the reference's order of subtractions is that
// of the substitution, qdldl.rs:708-752); the refinement of directldlkktsolver.rs:266-321 runs on top as before and
// parity is asserted on the refined solution.  Supernodes wider than SG_WMAX keep the pipelined substitution.
#include "dev_common.hpp"
#include "snode_common.hpp"
#include "grid_sync.hpp"

namespace chip {
namespace dev {

namespace {

constexpr int SG_WG = 256;        // 4 waves: 256 rows of G per workgroup of the build
constexpr int SG_KC = 128;        // rows of T staged per chunk
constexpr int SG_KLD = SG_KC + 1; // stride of a staged column (odd: the 16 lanes of a matrix-core operand read hit distinct banks)
constexpr int SG_U = 4;           // k-groups of A operands in flight per lane
constexpr int SG_XLD = 17;        // row stride of a wave's head block in LDS
constexpr int SG_WMAX = 512;      // widest supernode that takes this path (column bases / x_S in LDS)

__device__ __forceinline__ int sg_ldg(int h) { return (h + 7) & ~7; }

// This wave's 64 x 64 tile in the accumulator layout of v_mfma_f64_16x16x4_f64:
//     acc[jb][t][r] = X[row 16 t + kq + 4 r][column 16 jb + l15]          (l15 = lane & 15, kq = lane >> 4).
// For every row: x_c -= sum_{k < c} coef(c, k) x_k, c = 0 .. 63, with coef(c, k) = Ll[k * 64 + c] (zero for c <= k),
// blocked by 16 as in the panel kernels (snode.hip: snq_rows): head block kb leaves the accumulator layout through
// the wave's LDS slice xw ([row][column], stride SG_XLD), is finished in the lane = row form -- 120 products per row
// --, handed to store(kb, h) and applied to the blocks behind it as a (64 x 16) x (16 x 16) product on the matrix
// cores.  Per entry the subtractions go k = 0, 1, ...; inside a matrix instruction they are fused multiply-adds.
template <class Store>
__device__ __forceinline__ void sg_block_recur(snode_v4d (&acc)[4][4], const double *Ll, double *xw, int lane, Store store) {
    const int l15 = lane & 15, kq = lane >> 4;
    double h[16];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) xw[(16 * t + kq + 4 * r) * SG_XLD + l15] = acc[kb][t][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) h[c] = xw[lane * SG_XLD + c];
        __builtin_amdgcn_wave_barrier();
        const double *Lb = Ll + (16 * kb) * SN_NB + 16 * kb;
#pragma unroll
        for (int kk = 0; kk < 15; ++kk) {
            const double uq = h[kk];
            int zoff; // (an opaque zero that "depends" on u_q ties this column's LDS reads to its place in the chain, see k_snode_panel)
            asm volatile("v_mov_b32 %0, 0" : "=v"(zoff) : "v"(__double2hiint(uq)));
            const snode_v2d *cf = (const snode_v2d *)(Lb + kk * SN_NB + zoff);
#pragma unroll
            for (int p2 = (kk + 1) / 2; p2 < 8; ++p2) {
                const snode_v2d cc = cf[p2]; // (the pair that straddles kk meets a stored zero)
                h[2 * p2] -= cc.x * uq;
                h[2 * p2 + 1] -= cc.y * uq;
            }
        }
        store(kb, h);
        if (kb == 3) break;
#pragma unroll
        for (int c = 0; c < 16; ++c) xw[lane * SG_XLD + c] = -h[c];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            double a4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a4[t] = xw[(16 * t + l15) * SG_XLD + 4 * s4 + kq]; // A[m = l15][k = 4 s4 + kq] of row tile t
#pragma unroll
            for (int jb = kb + 1; jb < 4; ++jb) {
                const double bv = Ll[(16 * kb + 4 * s4 + kq) * SN_NB + 16 * jb + l15]; // B[k][n] = coef(16 jb + n, 16 kb + k)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[jb][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[t], bv, acc[jb][t], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// One workgroup per task = (record of the supernode in `order_all`, first of its 256 rows of G).  Inside a column
// block the columns are taken in REVERSE order (column' = 63 - column), which turns y T = z -- y_c = z_c - sum_{k > c}
// y_k T[k][c], last column first -- into the forward recurrence sg_block_recur knows.
__global__ __launch_bounds__(SG_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_snode_ginv(LdlView v, SnodeView sv, const int *__restrict__ order_all,
                                                      const int *__restrict__ tasks) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    double *Wt = (double *)gsm;          // products: Wt[column' * SG_KLD + k] = -T[kc0 + k][j0 + 63 - column']
    double *Ll = (double *)gsm;          // recurrence (after the products; the same bytes): Ll[k' * 64 + c']
    double *xwb = Ll + SN_NB * SN_NB;    // ... and the four waves' head blocks
    __shared__ int colbase[SG_WMAX];
    const int rec = tasks[2 * blockIdx.x], R0 = tasks[2 * blockIdx.x + 1];
    int sn;
    const SnodeGeom g = snode_geom(sv, order_all, rec, sn);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int ldg = sg_ldg(g.h);
    double *G = sv.Gx + g.goff;
    for (int t = tid; t < g.w; t += SG_WG) colbase[t] = g.cb[t];
    const int nblk = (g.w + SN_NB - 1) / SN_NB;
    const int Rlast = min(R0 + SG_WG, g.h) - 1;                      // last row of this workgroup
    const int cstart = Rlast < g.w ? Rlast / SN_NB : nblk - 1;       // rows of T^-1 end at their diagonal block
    const int kend = min(g.w, SN_NB * (cstart + 1));                 // columns of G this workgroup's rows own
    const int R0w = R0 + 64 * wave;
    const bool wlive = R0w < g.h;
    int rowA[4]; // rows of the A operand (clamped: a row beyond the panel reads the last one, its results are not stored)
#pragma unroll
    for (int t = 0; t < 4; ++t) rowA[t] = min(R0w + 16 * t + l15, g.h - 1);
    __syncthreads();
    for (int c = cstart; c >= 0; --c) {
        const int j0 = SN_NB * c, ncw = min(SN_NB, g.w - j0);
        const int kbeg = j0 + SN_NB;
        // a wave whose rows all lie above the block (rows of T^-1) owns zeros only: G is zero-filled once, nothing to do
        const bool wact = wlive && R0w + 63 >= j0;
        snode_v4d acc[4][4];
        // ---- P = [I; L_B], this block's columns reversed
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int jj = 63 - (16 * jb + l15);
            const int cbj = colbase[j0 + min(jj, ncw - 1)];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = R0w + 16 * t + kq + 4 * r;
                    double val = 0.0;
                    if (wact && jj < ncw) {
                        if (row < g.w) val = row == j0 + jj ? 1.0 : 0.0;
                        else if (row < g.h) val = v.Lx[cbj + row];
                    }
                    acc[jb][t][r] = val;
                }
        }
        // ---- minus the finished column blocks times T[J_k, J_c]
        if (kbeg < kend) { // (workgroup uniform)
            double a[SG_U][4];
            auto request = [&](int u, int kabs) {
                const size_t col = (size_t)min(kabs + 4 * u + kq, kend - 1);
#pragma unroll
                for (int t = 0; t < 4; ++t) a[u][t] = G[col * ldg + rowA[t]];
            };
            if (wact) {
#pragma unroll
                for (int u = 0; u < SG_U; ++u) request(u, kbeg);
            }
            for (int kc0 = kbeg; kc0 < kend; kc0 += SG_KC) {
                const int kcn = min(SG_KC, kend - kc0);
                const int kcnu = (kcn + 4 * SG_U - 1) / (4 * SG_U) * (4 * SG_U); // whole groups: the tail rows of the operand are zeros
                __syncthreads(); // the previous chunk (or the previous block's recurrence) is done with these bytes
                // lanes along k: 512-byte runs of a column of L, eight requests in flight per thread
                for (int i0 = tid; i0 < SN_NB * SG_KC; i0 += 8 * SG_WG) {
                    double tv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int idx = i0 + q * SG_WG, kk = idx & (SG_KC - 1), jjp = idx >> 7, jj = 63 - jjp;
                        tv[q] = v.Lx[colbase[j0 + min(jj, ncw - 1)] + kc0 + min(kk, kcn - 1)]; // (clamped: unconditional loads)
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int idx = i0 + q * SG_WG, kk = idx & (SG_KC - 1), jjp = idx >> 7, jj = 63 - jjp;
                        if (kk < kcnu) Wt[jjp * SG_KLD + kk] = (kk < kcn && jj < ncw) ? -tv[q] : 0.0;
                    }
                }
                __syncthreads();
                if (!wact) continue; // (after the barriers)
                for (int kk = 0; kk < kcnu; kk += 4 * SG_U) {
                    const int knext = kk + 4 * SG_U < kcnu ? kc0 + kk + 4 * SG_U : kc0 + SG_KC; // (the next chunk's first group)
#pragma unroll
                    for (int u = 0; u < SG_U; ++u) {
                        const int kl = kk + 4 * u + kq;
#pragma unroll
                        for (int jb = 0; jb < 4; ++jb) {
                            const double bw = Wt[(16 * jb + l15) * SG_KLD + kl];
#pragma unroll
                            for (int t = 0; t < 4; ++t) acc[jb][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][t], bw, acc[jb][t], 0, 0, 0);
                        }
                        request(u, knext);
                    }
                }
            }
        }
        // ---- times T_cc^-1: coef(c', k') = T[j0 + 63 - k'][j0 + 63 - c'], k' < c'
        __syncthreads();
        for (int idx = tid; idx < SN_NB * SN_NB; idx += SG_WG) {
            const int kp = idx >> 6, cp = idx & 63, jr = 63 - kp, jc = 63 - cp;
            double val = 0.0;
            if (cp > kp && jr < ncw) val = v.Lx[colbase[j0 + jc] + j0 + jr]; // (jc < jr < ncw)
            Ll[idx] = val;
        }
        __syncthreads();
        if (wact) {
            const int row = R0w + lane;
            sg_block_recur(acc, Ll, xwb + wave * (64 * SG_XLD), lane, [&](int kb, const double(&hh)[16]) {
                if (row >= g.h) return;
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) {
                    const int jj = 63 - (16 * kb + cc);
                    if (jj < ncw) G[(size_t)(j0 + jj) * ldg + row] = hh[cc];
                }
            });
        }
        __syncthreads(); // (the next block stages its operand over Ll / the head blocks; this block's columns of G are visible)
    }
}

// ---- the sparse rows / columns of a unit level inside the same launch (round 5) -------------------------------------
// A sweep used to alternate two launches per unit level: the row gathers over the columns that are NOT supernode members
// (k_gather_merged over the filtered lists) and the supernodes' kernel.  Forward, the gathers of level l + 1 need nothing
// the supernodes of level l write (those push into ancestors; the gathers read bundle columns and ordinary top columns
// of levels <= l, final since the previous launch) and both only SUBTRACT from rows of later levels: with the gathers'
// final store an atomic as well, the two run side by side in one launch.  Backward, the ordinary columns of level l and
// the supernodes of level l both depend on higher levels only.  The extra blocks of the grid: first the chunks of long
// rows (forward only; one atomic per chunk), then the wave-per-row list (16 rows per block), then the thread-per-row list.
struct SweepGather {
    GatherArgs a;
    const int *trows, *wrows, *crow, *cbeg, *cend;
    int tcount, wcount, ccount;
};
constexpr int SGS_WG_ = 1024;
__host__ __device__ inline int sweep_gather_blocks(int tcount, int wcount, int ccount) {
    return ccount + (wcount + 15) / 16 + (tcount + SGS_WG_ - 1) / SGS_WG_;
}
// COH (the persistent sweeps, k_snode_gsweep): the vector is written by other workgroups EARLIER IN THE SAME LAUNCH --
// every access to it goes to the device's coherence point (grid_sync.hpp), never through this XCD's L2
template <bool COH> __device__ __forceinline__ double sg_xload(const double *p) { return COH ? ir_load(p) : *p; }
template <bool COH> __device__ __forceinline__ void sg_xstore(double *p, double val) {
    if (COH) ir_store(p, val);
    else *p = val;
}
template <int MODE, bool COH>
__device__ __forceinline__ void sg_gather_block(const SweepGather &q, int gb, double *red) {
    const GatherArgs &a = q.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nW = (q.wcount + 15) / 16;
    if (gb < q.ccount) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        const int ce = q.cend[gb];
        int t = q.cbeg[gb] + tid;
        for (; t + 3 * SGS_WG_ < ce; t += 4 * SGS_WG_) {
            const int i0 = a.idx[t], i1 = a.idx[t + SGS_WG_], i2 = a.idx[t + 2 * SGS_WG_], i3 = a.idx[t + 3 * SGS_WG_];
            const double v0 = a.val[t], v1 = a.val[t + SGS_WG_], v2 = a.val[t + 2 * SGS_WG_], v3 = a.val[t + 3 * SGS_WG_];
            s0 += v0 * sg_xload<COH>(&a.xin[i0]);
            s1 += v1 * sg_xload<COH>(&a.xin[i1]);
            s2 += v2 * sg_xload<COH>(&a.xin[i2]);
            s3 += v3 * sg_xload<COH>(&a.xin[i3]);
        }
        for (; t < ce; t += SGS_WG_) s0 += a.val[t] * sg_xload<COH>(&a.xin[a.idx[t]]);
        const double ws = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) red[wave] = ws;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < SGS_WG_ / 64; ++w) tot += red[w];
            atomicAdd(&a.out[q.crow[gb]], -tot);
        }
        return;
    }
    if (gb < q.ccount + nW) {
        const int wid = (gb - q.ccount) * 16 + wave;
        if (wid >= q.wcount) return;
        const int r = q.wrows[wid];
        const int b = a.ptr[r], e = a.ptr[r + 1];
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int t = b + lane;
        for (; t + 192 < e; t += 256) {
            const int i0 = a.idx[t], i1 = a.idx[t + 64], i2 = a.idx[t + 128], i3 = a.idx[t + 192];
            const double v0 = a.val[t], v1 = a.val[t + 64], v2 = a.val[t + 128], v3 = a.val[t + 192];
            s0 += v0 * sg_xload<COH>(&a.xin[i0]);
            s1 += v1 * sg_xload<COH>(&a.xin[i1]);
            s2 += v2 * sg_xload<COH>(&a.xin[i2]);
            s3 += v3 * sg_xload<COH>(&a.xin[i3]);
        }
        for (; t < e; t += 64) s0 += a.val[t] * sg_xload<COH>(&a.xin[a.idx[t]]);
        const double sm = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) {
            if (MODE == FWD) atomicAdd(&a.out[r], -sm);
            else sg_xstore<COH>(&a.out[r], sg_xload<COH>(&a.out[r]) * a.aux[r] - sm);
        }
        return;
    }
    const int k = (gb - q.ccount - nW) * SGS_WG_ + tid;
    if (k >= q.tcount) return;
    const int r = q.trows[k];
    const int b = a.ptr[r], e = a.ptr[r + 1];
    double sm = 0.0;
    for (int t = b; t < e; ++t) sm += a.val[t] * sg_xload<COH>(&a.xin[a.idx[t]]);
    if (MODE == FWD) atomicAdd(&a.out[r], -sm);
    else sg_xstore<COH>(&a.out[r], sg_xload<COH>(&a.out[r]) * a.aux[r] - sm);
}

// forward: grid (64-row blocks of G, supernodes of the unit level), sixteen waves.  Lane = row; wave q takes the columns
// q, q + 16, ...: their entries are REQUESTED FIRST -- they depend on nothing but the record -- and x_S is staged while
// they are in flight (a launch is a handful of dependent memory round trips, not bandwidth); the sixteen partial sums
// meet in LDS.  Member rows go to yt (the other workgroups of the supernode still read x_S(old) from x), the rows of B
// leave as one atomic per (row, supernode).
constexpr int SGS_WG = SGS_WG_;
constexpr int SGS_NW = SGS_WG / 64;
constexpr size_t SGS_FWD_LDS = (size_t)(SG_WMAX + SGS_NW * 64) * sizeof(double); // x_S + the sixteen partial sums
// one (64-row block bx, supernode by of the level) of the forward pass, in two halves: sg_fwd_prepare requests everything
// that does not depend on the vector (the record, the first sixteen columns' entries of this lane's row of G, the node
// ids of x_S) -- the persistent sweep issues it BEFORE it waits at the level's barrier --, sg_fwd_finish stages x_S and
// does the arithmetic; xs[SG_WMAX], part[SGS_NW * 64] in LDS
struct SgFwdPrep {
    SnodeGeom g;
    double l[16];
    int r0, cidx;
    bool active;
};
__device__ __forceinline__ int sg_fwd_ncols(const SnodeGeom &g, int r0) { return r0 < g.w ? min(g.w, r0 + 64) : g.w; } // (rows of T^-1 end at their diagonal block)
__device__ __forceinline__ void sg_fwd_prepare(SgFwdPrep &p, const LdlView &v, const SnodeView &sv, const int *__restrict__ order,
                                               int by, int bx) {
    int sn;
    p.g = snode_geom(sv, order, by, sn);
    const SnodeGeom &g = p.g;
    p.r0 = 64 * bx;
    p.active = p.r0 < g.h;
    if (!p.active) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncols = sg_fwd_ncols(g, p.r0), ldg = sg_ldg(g.h);
    const double *Gr = sv.Gx + g.goff + min(p.r0 + lane, g.h - 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) p.l[q] = Gr[(size_t)min(wave + SGS_NW * q, ncols - 1) * ldg]; // (clamped: unconditional)
    p.cidx = g.cols[min(tid, ncols - 1)]; // (SGS_WG >= SG_WMAX: one member per thread)
}
template <bool COH>
__device__ __forceinline__ void sg_fwd_finish(SgFwdPrep &p, const LdlView &v, const SnodeView &sv, double *x, double *yt,
                                              double *xs, double *part) {
    if (!p.active) return;
    const SnodeGeom &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncols = sg_fwd_ncols(g, p.r0), ldg = sg_ldg(g.h), i = p.r0 + lane;
    const double *Gr = sv.Gx + g.goff + min(i, g.h - 1);
    double(&l)[16] = p.l;
    if (tid < ncols) xs[tid] = sg_xload<COH>(&x[p.cidx]);
    __syncthreads();
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q += 4) {
        const int j = wave + SGS_NW * q;
        s0 += j < ncols ? l[q] * xs[j] : 0.0;
        s1 += j + SGS_NW < ncols ? l[q + 1] * xs[j + SGS_NW] : 0.0;
        s2 += j + 2 * SGS_NW < ncols ? l[q + 2] * xs[j + 2 * SGS_NW] : 0.0;
        s3 += j + 3 * SGS_NW < ncols ? l[q + 3] * xs[j + 3 * SGS_NW] : 0.0;
    }
    for (int jb = 16 * SGS_NW; jb < ncols; jb += 16 * SGS_NW) { // (supernodes wider than 256)
#pragma unroll
        for (int q = 0; q < 16; ++q) l[q] = Gr[(size_t)min(jb + wave + SGS_NW * q, ncols - 1) * ldg];
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
            const int j = jb + wave + SGS_NW * q;
            s0 += j < ncols ? l[q] * xs[j] : 0.0;
            s1 += j + SGS_NW < ncols ? l[q + 1] * xs[j + SGS_NW] : 0.0;
            s2 += j + 2 * SGS_NW < ncols ? l[q + 2] * xs[j + 2 * SGS_NW] : 0.0;
            s3 += j + 3 * SGS_NW < ncols ? l[q + 3] * xs[j + 3 * SGS_NW] : 0.0;
        }
    }
    part[wave * 64 + lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && i < g.h) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < SGS_NW; ++q) tot += part[q * 64 + lane];
        if (i < g.w) yt[g.cols[i]] = tot; // (read by the backward sweep: another launch)
        else atomicAdd(&x[v.Li[g.bn0 + i - g.w]], -tot);
    }
}
template <bool COH>
__device__ __forceinline__ void sg_fwd_task(const LdlView &v, const SnodeView &sv, const int *__restrict__ order, double *x,
                                            double *yt, int by, int bx, double *xs, double *part) {
    SgFwdPrep p;
    sg_fwd_prepare(p, v, sv, order, by, bx);
    sg_fwd_finish<COH>(p, v, sv, x, yt, xs, part);
}
__global__ __launch_bounds__(SGS_WG) void k_snode_gfwd(LdlView v, SnodeView sv, const int *__restrict__ order, double *x,
                                                       double *yt, int count, SweepGather sg) {
    __shared__ double xs[SG_WMAX];
    __shared__ double part[SGS_NW * 64];
    if ((int)blockIdx.y >= count) { // the next level's row gathers ride along (see SweepGather)
        const int gb = ((int)blockIdx.y - count) * (int)gridDim.x + (int)blockIdx.x;
        if (gb < sweep_gather_blocks(sg.tcount, sg.wcount, sg.ccount)) sg_gather_block<FWD, false>(sg, gb, part);
        return;
    }
    sg_fwd_task<false>(v, sv, order, x, yt, (int)blockIdx.y, (int)blockIdx.x, xs, part);
}
// backward: grid (64-column blocks, supernodes of the unit level), sixteen waves of four columns each.  A wave's first
// 4 x 4 x 64 entries are requested first, s = [D^-1 y_S ; -x_B] (from the block's first row on) is staged in LDS while
// they are in flight; rows along the lanes, fixed order of summation, no atomics.
// (in the same two halves: the record, the wave's first 4 x 4 x 64 entries of G, and this thread's first entry of
// s -- complete for a member row, the node id for a row of B -- do not depend on the vector)
struct SgBwdPrep {
    SnodeGeom g;
    double l[4][4];
    double sval;
    int j0, sidx;
    bool active;
};
__device__ __forceinline__ const double *sg_bwd_col(const SnodeView &sv, const SnodeGeom &g, int j0, int j) {
    return sv.Gx + g.goff + (size_t)min(j, g.w - 1) * sg_ldg(g.h) + j0;
}
__device__ __forceinline__ void sg_bwd_prepare(SgBwdPrep &p, const LdlView &v, const SnodeView &sv, const int *__restrict__ order,
                                               const double *yt, int by, int bx) {
    int sn;
    p.g = snode_geom(sv, order, by, sn);
    const SnodeGeom &g = p.g;
    p.j0 = 64 * bx;
    p.active = p.j0 < g.w;
    if (!p.active) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nr = g.h - p.j0, jf = p.j0 + 4 * wave; // this wave's columns jf .. jf + 3
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const double *Gc = sg_bwd_col(sv, g, p.j0, jf + u);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) p.l[u][rr] = Gc[min(lane + 64 * rr, nr - 1)]; // (entries above the diagonal inside the block are stored zeros)
    }
    const int i = p.j0 + tid;
    p.sidx = -1;
    p.sval = 0.0;
    if (i < g.w) {
        const int c = g.cols[i];
        p.sval = yt[c] * v.Dinv[c];
    } else if (i < g.h) {
        p.sidx = v.Li[g.bn0 + i - g.w];
    }
}
template <bool COH>
__device__ __forceinline__ void sg_bwd_finish(SgBwdPrep &p, const LdlView &v, const SnodeView &sv, double *x, const double *yt,
                                              double *ss) {
    if (!p.active) return;
    const SnodeGeom &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int j0 = p.j0, nr = g.h - p.j0, jf = p.j0 + 4 * (tid >> 6);
    double(&l)[4][4] = p.l;
    const int *Bn = v.Li + g.bn0;
    if (j0 + tid < g.h) ss[tid] = p.sidx >= 0 ? -sg_xload<COH>(&x[p.sidx]) : p.sval;
    for (int i = j0 + tid + SGS_WG; i < g.h; i += SGS_WG) {
        double val;
        if (i < g.w) {
            const int c = g.cols[i];
            val = yt[c] * v.Dinv[c];
        } else {
            val = -sg_xload<COH>(&x[Bn[i - g.w]]);
        }
        ss[i - j0] = val;
    }
    __syncthreads();
    if (jf >= g.w) return; // (wave uniform; after the barrier)
    double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int i = lane + 64 * rr;
        const double sa = i < nr ? ss[i] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += l[u][rr] * sa;
    }
    for (int ib = 256; ib < nr; ib += 256) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int u = 0; u < 4; ++u) l[u][rr] = sg_bwd_col(sv, g, j0, jf + u)[min(ib + lane + 64 * rr, nr - 1)];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = ib + lane + 64 * rr;
            const double sa = i < nr ? ss[i] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += l[u][rr] * sa;
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const double tot = wave_sum(a[u]);
        if (lane == 0 && jf + u < g.w) sg_xstore<COH>(&x[g.cols[jf + u]], tot);
    }
}
template <bool COH>
__device__ __forceinline__ void sg_bwd_task(const LdlView &v, const SnodeView &sv, const int *__restrict__ order, double *x,
                                            const double *yt, int by, int bx, double *ss) {
    SgBwdPrep p;
    sg_bwd_prepare(p, v, sv, order, yt, by, bx);
    sg_bwd_finish<COH>(p, v, sv, x, yt, ss);
}
__global__ __launch_bounds__(SGS_WG) void k_snode_gbwd(LdlView v, SnodeView sv, const int *__restrict__ order, double *x,
                                                       const double *yt, int count, SweepGather sg) {
    extern __shared__ __attribute__((aligned(16))) char bsm[];
    double *ss = (double *)bsm;
    if ((int)blockIdx.y >= count) { // this level's ordinary columns ride along (see SweepGather)
        const int gb = ((int)blockIdx.y - count) * (int)gridDim.x + (int)blockIdx.x;
        if (gb < sweep_gather_blocks(sg.tcount, sg.wcount, sg.ccount)) sg_gather_block<BWD, false>(sg, gb, ss);
        return;
    }
    sg_bwd_task<false>(v, sv, order, x, yt, (int)blockIdx.y, (int)blockIdx.x, ss);
}

// ---- a RUN of consecutive unit levels in one persistent launch (round 5) ---------------------------------------------
// A sweep through k unit levels on the one-pass matrices was k launches of a handful of dependent memory round trips
// each (8 - 9 us of kernel + the boundary: config 2 has 27 such levels, twelve sweeps per step).  k_snode_gsweep walks
// the levels of a run itself: the tasks of a level (its supernodes' blocks + the row gathers that ride along) are spread
// over a co-resident grid, a grid barrier (grid_sync.hpp) separates the levels.  The vector crosses workgroups inside
// the launch: read and written at the device's coherence point (COH above); G, L, the index lists and yt do not.
// A barrier that cannot complete (the grid was not co-resident) raises *fail -- the solve reports a non-finite result.
// the sweep's own barrier: ONE arrival counter (ctl[0], monotonic over the launch) and eight release words a cache line
// apart (ctl[32 (1 + q)], polled by the workgroups with blockIdx = q mod 8) -- at most half the chip takes part, the
// two-level counters of grid_sync.hpp (built for 1000 workgroups) would add a dependent round trip per level.  (Waiting
// workgroups polling the arrival counter itself, one round trip less on paper, was measured: the polls queue behind the
// arrivals on that one address and the gain of the persistent launch all but disappears.)
__device__ __forceinline__ bool gs_barrier(int *ctl, int gen, int nwg) {
    __shared__ int s_ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        int ok = 1;
        if (atomicAdd(ctl, 1) + 1 == nwg * gen) {
            for (int q = 0; q < 8; ++q) __hip_atomic_store(ctl + 32 * (1 + q), gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const int *rel = ctl + 32 * (1 + ((int)blockIdx.x & 7));
            long long spins = 0;
            while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1ll << 22)) {
                    ok = 0;
                    break;
                }
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}
// (the last use of the words in a launch: whoever completes the count zeroes them -- every workgroup is past the last wait)
__device__ __forceinline__ void gs_exit(int *ctl, int gen, int nwg) {
    if (threadIdx.x != 0) return;
    if (atomicAdd(ctl, 1) + 1 == nwg * gen) {
        for (int q = 0; q < 8; ++q) __hip_atomic_store(ctl + 32 * (1 + q), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ctl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int MODE>
__global__ __launch_bounds__(SGS_WG) void k_snode_gsweep(LdlView v, SnodeView sv, const int *__restrict__ order_all, double *x,
                                                         double *yt, const GSweepLevel *__restrict__ lv, int nlev, GatherArgs ga,
                                                         int *ctl, int *fail, int test_drop) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    double *sm = (double *)gsm;
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    if (test_drop && bid == G - 1 && G > 1 && nlev > 1) return; // (tests: a launch whose level barrier cannot complete)
    SgFwdPrep pf;
    SgBwdPrep pb;
    pf.active = false;
    pb.active = false;
    GSweepLevel L = lv[0];
    // this workgroup's FIRST task of a level is prepared ahead of the level's barrier: what it needs besides the vector
    auto prepare_first = [&](const GSweepLevel &Lq) {
        const bool mine = bid < Lq.gx * Lq.count;
        const int by = mine ? bid / Lq.gx : 0, bx = mine ? bid - by * Lq.gx : 0;
        if (MODE == FWD) {
            pf.active = false;
            if (mine) sg_fwd_prepare(pf, v, sv, order_all + 8 * (size_t)Lq.off, by, bx);
        } else {
            pb.active = false;
            if (mine) sg_bwd_prepare(pb, v, sv, order_all + 8 * (size_t)Lq.off, yt, by, bx);
        }
    };
    prepare_first(L);
    for (int li = 0; li < nlev; ++li) {
        SweepGather sg;
        sg.a = ga;
        sg.trows = L.trows;
        sg.wrows = L.wrows;
        sg.crow = L.crow;
        sg.cbeg = L.cbeg;
        sg.cend = L.cend;
        sg.tcount = L.tcount;
        sg.wcount = L.wcount;
        sg.ccount = L.ccount;
        const int nsn_tasks = L.gx * L.count, ntasks = nsn_tasks + sweep_gather_blocks(L.tcount, L.wcount, L.ccount);
        if (bid < ntasks) { // the task prepared ahead of the barrier
            if (bid < nsn_tasks) {
                if (MODE == FWD) sg_fwd_finish<true>(pf, v, sv, x, yt, sm, sm + SG_WMAX);
                else sg_bwd_finish<true>(pb, v, sv, x, yt, sm);
            } else {
                sg_gather_block<MODE, true>(sg, bid - nsn_tasks, sm);
            }
            __syncthreads(); // (the next task stages over the same LDS)
        }
        for (int task = bid + G; task < ntasks; task += G) {
            if (task < nsn_tasks) {
                const int by = task / L.gx, bx = task - by * L.gx;
                if (MODE == FWD) sg_fwd_task<true>(v, sv, order_all + 8 * (size_t)L.off, x, yt, by, bx, sm, sm + SG_WMAX);
                else sg_bwd_task<true>(v, sv, order_all + 8 * (size_t)L.off, x, yt, by, bx, sm);
            } else {
                sg_gather_block<MODE, true>(sg, task - nsn_tasks, sm);
            }
            __syncthreads();
        }
        if (li + 1 < nlev) {
            L = lv[li + 1];
            prepare_first(L);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's stores and atomics are performed
            if (!gs_barrier(ctl, li + 1, G)) {
                if (threadIdx.x == 0) *fail = 1;
                return;
            }
        }
    }
    gs_exit(ctl, nlev, G);
}

} // namespace

static size_t ginv_lds_bytes() {
    const size_t a = (size_t)SN_NB * SG_KLD * sizeof(double);
    const size_t b = (size_t)(SN_NB * SN_NB + (SG_WG / 64) * 64 * SG_XLD) * sizeof(double);
    return std::max(a, b);
}
int snode_g_max_width() { return SG_WMAX; }
int snode_g_attributes(int hmax) {
    int rc = (int)raise_dynamic_lds((const void *)k_snode_ginv, ginv_lds_bytes());
    if (!rc && (size_t)hmax * sizeof(double) > 48 * 1024)
        rc = (int)raise_dynamic_lds((const void *)k_snode_gbwd, (size_t)hmax * sizeof(double));
    return rc;
}
long long snode_g_ld(int h) { return (h + 7) & ~7; }
void snode_ginv(hipStream_t s, const LdlView &v, const SnodeView &sv, const int *order_all, const int *tasks, int ntasks) {
    if (ntasks <= 0) return;
    k_snode_ginv<<<ntasks, SG_WG, ginv_lds_bytes(), s>>>(v, sv, order_all, tasks);
}
void solve_snodes_g(hipStream_t s, GatherMode m, const LdlView &v, const SnodeView &sv, const int *order, int count, int wlvl,
                    int hlvl, double *x, double *yt, const LaunchProf *lp, const GatherArgs *ga, ListView t, ListView w, ChunkView c) {
    if (!count) return;
    SweepGather sg{};
    int extra = 0;
    if (ga) {
        sg.a = *ga;
        sg.trows = t.idx;
        sg.tcount = t.count;
        sg.wrows = w.idx;
        sg.wcount = w.count;
        sg.crow = c.row;
        sg.cbeg = c.beg;
        sg.cend = c.end;
        sg.ccount = c.count;
        extra = sweep_gather_blocks(t.count, w.count, c.count);
    }
    if (lp) lp->begin(lp->ctx, PFK_SN_TRI);
    const int gx = m == FWD ? (hlvl + 63) / 64 : (wlvl + 63) / 64;
    const dim3 grid(gx, count + (extra + gx - 1) / gx);
    if (m == FWD) k_snode_gfwd<<<grid, SGS_WG, 0, s>>>(v, sv, order, x, yt, count, sg);
    else k_snode_gbwd<<<grid, SGS_WG, std::max((size_t)hlvl, (size_t)16) * sizeof(double), s>>>(v, sv, order, x, yt, count, sg);
    if (lp) lp->end(lp->ctx, PFK_SN_TRI);
}

size_t snode_gsweep_lds(GatherMode m, int hmax) {
    return m == FWD ? SGS_FWD_LDS : std::max((size_t)hmax, (size_t)64) * sizeof(double);
}
int snode_gsweep_capacity(GatherMode m, size_t lds) {
    const void *k = m == FWD ? (const void *)k_snode_gsweep<FWD> : (const void *)k_snode_gsweep<BWD>;
    if (lds > 48 * 1024 && raise_dynamic_lds(k, lds) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    const hipError_t e = m == FWD ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_snode_gsweep<FWD>, SGS_WG, lds)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_snode_gsweep<BWD>, SGS_WG, lds);
    if (e != hipSuccess || hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return per_cu * prop.multiProcessorCount;
}
void solve_snodes_gsweep(hipStream_t s, GatherMode m, const LdlView &v, const SnodeView &sv, const int *order_all, double *x,
                         double *yt, const GSweepLevel *lv, int nlev, int grid, size_t lds, const GatherArgs &ga, int *ctl,
                         int *fail, const LaunchProf *lp) {
    if (nlev <= 0 || grid <= 0) return;
    if (lp) lp->begin(lp->ctx, PFK_SN_TRI);
    const int drop = switches().gs_test_drop ? 1 : 0;
    if (m == FWD) k_snode_gsweep<FWD><<<grid, SGS_WG, lds, s>>>(v, sv, order_all, x, yt, lv, nlev, ga, ctl, fail, drop);
    else k_snode_gsweep<BWD><<<grid, SGS_WG, lds, s>>>(v, sv, order_all, x, yt, lv, nlev, ga, ctl, fail, drop);
    if (lp) lp->end(lp->ctx, PFK_SN_TRI);
}

} // namespace dev
} // namespace chip
