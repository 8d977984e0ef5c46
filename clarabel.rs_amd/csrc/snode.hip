// snode.hip -- chain supernodes: MFMA update tiles, panel factorisation, ancestor updates, substitutions (k_snode_*)
// (one of the translation units behind kernels.hpp; the design rules and the reference citations are in
// dev_common.hpp)
#include "dev_common.hpp"
#include "grid_sync.hpp"
#include "snode_common.hpp"
#include <type_traits>

namespace chip {
namespace dev {

namespace {

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

// the same for NR vectors at once: the coefficients are read once and the NR dependent chains interleave (the second
// vector's 64 steps hide in the first one's latencies instead of doubling a hop of k_snode_tri's pipeline)
template <bool FWDMODE, int NR> __device__ __forceinline__ void snode_block_solve_n(const double *Tt, double (&xv)[NR], int lane) {
    double t[SN_NB];
#pragma unroll
    for (int jj = 0; jj < SN_NB; ++jj) t[jj] = Tt[jj * SN_NB + lane];
    if (FWDMODE) {
#pragma unroll
        for (int jj = 0; jj < SN_NB - 1; ++jj)
#pragma unroll
            for (int k = 0; k < NR; ++k) xv[k] -= t[jj] * readlane_f64(xv[k], jj);
    } else {
#pragma unroll
        for (int jj = SN_NB - 1; jj > 0; --jj)
#pragma unroll
            for (int k = 0; k < NR; ++k) xv[k] -= t[jj] * readlane_f64(xv[k], jj);
    }
}

// CHIP_SN_DEBUG: wall-clock stamps (10 ns ticks) of ONE workgroup of a supernode launch at its phase boundaries;
// `drain` first waits for the loads in flight, so that a phase owns the latency of what it requested
__device__ __forceinline__ void sn_stamp(long long *dbg, bool me, int slot, bool drain = false) {
    if (!dbg) return;
    if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (me) dbg[slot] = wall_clock64();
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
    lds_barrier(); // (the caller has just filled colbase; LDS only: see dev_common.hpp)
    sn_stamp(sv.dbg, dbgme, 17, true);
    // The (d_k L[j,k]) operand of the 64 target columns, double buffered (round 6): two buffers of SN_KC / 2 k rows; a wave
    // stages whole k rows -- lane = column of the block.  The entries of the NEXT chunk are requested before this chunk's
    // matrix instructions and written into the other buffer behind them: one LDS barrier per chunk and no global round
    // trip between two chunks (rounds 2-5: four batches of loads per 128 k rows between two barriers -- ~7 us per chunk
    // against 7-14 us of matrix instructions, hidden only while the CU's other workgroup happened to multiply).  The
    // pivots of a wave's rows come as ONE load (lane r holds row r's) and are broadcast when the entries are written.
    constexpr int KCB = SN_KC / 2, SR = KCB / (SN_WG / 64);
    double wv[SR], dvec = 0.0;
    auto stage_request = [&](int kc0) {
        dvec = g.d[min(kc0 + wave + (lane & (SR - 1)) * (SN_WG / 64), kend - 1)];
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const int k = min(kc0 + wave + r * (SN_WG / 64), kend - 1); // (clamped: no branch per load)
            wv[r] = v.Lx[colbase[k] + jrow0 + min(lane, ncols - 1)];
        }
    };
    auto stage_commit = [&](double *Wb, int kc0) {
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const int kl = wave + r * (SN_WG / 64);
            Wb[kl * SN_NB + lane] = (kc0 + kl < kend && lane < ncols) ? wv[r] * readlane_f64(dvec, r) : 0.0;
        }
    };
    if (kbeg < kend) {
        stage_request(kbeg);
        if (wave_live) {
#pragma unroll
            for (int u = 0; u < SN_U; ++u) request(u, kbeg);
        }
        stage_commit(Wl, kbeg);
    }
    lds_barrier();
    sn_stamp(sv.dbg, dbgme, 18);
    int buf = 0;
    for (int kc0 = kbeg; kc0 < kend; kc0 += KCB, buf ^= 1) {
        const bool more = kc0 + KCB < kend;
        if (more) stage_request(kc0 + KCB);
        const int kcn = min(KCB, kend - kc0);
        const int kcnu = (kcn + 4 * SN_U - 1) / (4 * SN_U) * (4 * SN_U); // whole groups: the tail rows of the operand are zeros
        const double *Wb = Wl + buf * (KCB * SN_NB);
        if (wave_live) { // (else: the whole wave is beyond the panel; it still stages)
            for (int kk = 0; kk < kcnu; kk += 4 * SN_U) {
                const int knext = kk + 4 * SN_U < kcnu ? kc0 + kk + 4 * SN_U : kc0 + KCB; // (the next chunk's first group)
#pragma unroll
                for (int u = 0; u < SN_U; ++u) {
                    const int kl = kk + 4 * u + kq;
#pragma unroll
                    for (int c = 0; c < SN_NB / 16; ++c) {
                        const double bw = Wb[kl * SN_NB + 16 * c + l15];
                        acc[0][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][u], bw, acc[0][c], 0, 0, 0);
                        acc[1][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][u], bw, acc[1][c], 0, 0, 0);
                    }
                    request(u, knext);
                }
            }
        }
        if (more) stage_commit(Wl + (buf ^ 1) * (KCB * SN_NB), kc0 + KCB);
        lds_barrier(); // (LDS only: the A operands requested for the next chunk stay in flight)
    }
    sn_stamp(sv.dbg, dbgme, 19);
    // ---- emit through LDS: this wave's 16 x 64 tile in its own 8 KiB of the (now free) operand buffer, element
    //      (row rr, column jj) at jj * 16 + (rr ^ (jj & 15)) -- the swizzle keeps both the column-per-lane writes
    //      and the row-per-lane reads off common banks
    lds_barrier();
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
        if (!EXTEND && !atomic_emit) {
            // (round 5: the tile's sixteen read-modify-writes per lane as ONE batch of loads and one of stores -- in two
            // batches of eight the emit was four dependent round trips per workgroup, 7 of the 13 us of a config 2 launch)
            // (the tile's values stay in LDS until the stores: sixteen loads in flight cost sixteen registers pairs, not 32)
            if (i >= g.h) continue;
            double cur[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int jj = 4 * m + kq, j = jrow0 + jj;
                cur[m] = (jj < ncols && i > j) ? v.Lx[colbase[j] + i] : 0.0;
            }
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int jj = 4 * m + kq, j = jrow0 + jj;
                if (jj >= ncols) continue;
                const double val = Tw[jj * 16 + (l15 ^ (jj & 15))];
                if (i > j) v.Lx[colbase[j] + i] = cur[m] - val;
                else if (i == j) v.D[g.cols[j]] -= val;
            }
            continue;
        }
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
            } else if (sv.U) {
                // assembled per target column afterwards (k_snode_assemble): plain stores, consecutive rows of one
                // column are consecutive entries of the supernode's packed triangle
                double *Us = sv.U + sv.asm_uoff[sn];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int jj = 4 * (m0 + m) + kq, cB = jrow0 + jj - g.w;
                    if (jj >= ncols) continue;
                    if (rB > cB) Us[(long long)cB * g.nb - (long long)cB * (cB + 1) / 2 + (rB - cB - 1)] = val[m];
                    else if (rB == cB) sv.Ud[sv.asm_doff[sn] + cB] = val[m];
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
// grid (row groups, column blocks of B, supernodes of the level) -- or, xcd != 0, ONE dimension that is decoded so that
// the tiles of a supernode share an XCD: workgroups go to the eight XCDs round-robin by their linear id, every XCD has
// its own L2, and the 15 column blocks x 4 row groups of a config 5 clique all stream the same 9 MB panel, k chunk by
// k chunk and roughly in step.  Spread over the chip each tile fetched its operands through the fabric (23 GB per
// launch of the leaf level, ~2.9 TB/s: the launch was bound by that, not by the matrix cores); with a supernode per
// XCD a k chunk (~1 MB) is fetched once and the other tiles find it in the L2.  id = 8 q + r: XCD r works on supernode
// 8 (q / T) + r, tile q % T of T = gx * gy.
// ks > 1 (round 5; levels with few supernodes -- BASELINE config 2's 26 deep levels ran 3 workgroups per supernode for
// ~56 us): the member columns k are divided among ks workgroups per tile, which meet in the fp64 atomics the tiles
// leave through anyway (not with the assembled form, whose stores are plain).
__global__ __launch_bounds__(SN_WG) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_snode_extend(LdlView v, SnodeView sv, const int *__restrict__ order,
                                                                                                   int xcd, int gx, int gy, int count, int ks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Wl;
    int *colbase = snode_lds(smem, Wl);
    int bx = (int)blockIdx.x % gx, ksi = (int)blockIdx.x / gx, by = (int)blockIdx.y, bz = (int)blockIdx.z;
    if (xcd) {
        const int id = (int)blockIdx.x, r = id & 7, q = id >> 3, T = gx * gy * ks, tile = q % T, rem = tile % (gx * gy);
        bz = 8 * (q / T) + r;
        if (bz >= count) return;
        ksi = tile / (gx * gy);
        by = rem / gx; // (column blocks outermost: the tiles of one column block share the staged operand as well)
        bx = rem % gx;
    }
    int sn;
    const SnodeGeom g = snode_geom(sv, order, bz, sn);
    const int c0 = by * SN_NB;
    if (c0 >= g.nb) return;
    const int row_begin = g.w + c0 + bx * SN_ROWS;
    if (row_begin >= g.h) return;
    // this split's share of the member columns, in units of one block column
    const int nunits = (g.w + SN_NB - 1) / SN_NB;
    const int kbeg = (int)(((long long)nunits * ksi) / ks) * SN_NB, kend = min(g.w, (int)(((long long)nunits * (ksi + 1)) / ks) * SN_NB);
    if (kbeg >= kend) return;
    for (int t = threadIdx.x; t < g.w; t += SN_WG) colbase[t] = g.cb[t];
    snode_tiles<true>(v, sv, g, sn, colbase, Wl, g.w + c0, min(SN_NB, g.nb - c0), kend, row_begin, kbeg);
}
// ---------------------------------------------------------------------------
// Wide tiles for the ancestors' update of levels with many supernodes (round 6): k_snode_extend_wide.
//
// A wave owns 16 rows x 256 columns (sixteen accumulator tiles, 128 registers): an A operand (an entry of the panel,
// read from HBM) feeds 16 matrix instructions instead of 4.  The (d L)' operand of the 256 target columns is staged in
// chunks of four k rows per wave of the workgroup and DOUBLE BUFFERED in LDS by the workgroup itself -- the next chunk's
// entries are requested before the current chunk's matrix instructions and written behind them, one LDS barrier per
// chunk.  NW = 4 waves (64 x 256 tiles, 2 x 32 KiB of LDS, ~230 registers): two workgroups per CU, one's matrix
// instructions cover the other's first loads, barriers and the way its tiles leave; NW = 8 (128 x 256, 2 x 64 KiB): one
// per CU -- 36.9 against 36.3 ms per step of config 5.  All tiles of a supernode stage the same operand: the launch
// places a supernode's workgroups on one XCD so that they find it in its L2.  tools/micro/mfma_f64_probe.hip: with two
// waves per SIMD, 16 accumulators and one ds_read_b64 per matrix instruction the matrix cores reach 77.5 of their 78.6
// TFLOP/s.
// Config 5's leaf level (200 supernodes, 903 rows of B, 1275 member columns): 5.8 ms against 8.5 with the 64-column
// tiles; the computed tiles cover 1.09 x the triangle's useful area (1.28 x with 256 x 64 tiles that are not skipped
// above the diagonal) -- per executed flop 0.50 against 0.40 of the matrix peak.  The same form for the update of a
// panel's own block columns (the columns before a group of four applied to all four, k_snode_update with the group's own
// columns in between) was measured and not kept: 8.6 + 3.6 ms against 11.1 -- the short launches for the columns inside
// a group cost what the wide tiles save.
// ---------------------------------------------------------------------------
constexpr int SNW_NC = 256;   // target columns of a wide tile
// (k rows per LDS buffer: four per wave of the workgroup -- 8 waves: 32 * 256 * 8 = 64 KiB, 4 waves: 32 KiB)
constexpr int SNW_U = 4;      // k-groups of A operands in flight per lane
constexpr int SNW_SR = 4;     // k rows a wave stages per chunk

// element (k row kl, column j) of a staged buffer: odd k rows swap the halves of every 32-column group, so that the
// four k rows a matrix instruction's B operand takes (lanes 16 q .. 16 q + 15 = row kq) hit disjoint banks pairwise
__device__ __forceinline__ int snw_at(int kl, int j) { return kl * SNW_NC + (j ^ ((kl & 1) << 4)); }

// (kl = KOFF + kq with KOFF even: the swizzle bit is the lane's kq & 1, so the sixteen operands of a k-group sit at
// compile-time offsets from two per-lane pointers -- pe for even c, po for odd c)
template <int NCW, int KOFF>
__device__ __forceinline__ void snw_group(snode_v4d (&acc)[SNW_NC / 16], double a, const double *pe, const double *po) {
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
        const double bw = ((c & 1) ? po : pe)[KOFF * SNW_NC + 16 * c];
        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bw, acc[c], 0, 0, 0);
    }
}

// one 16 x 64 slice of a wave's tile leaves through the wave's 8 KiB of LDS, transposed (see snode_tiles)
template <bool EXTEND>
__device__ __forceinline__ void snw_emit_slice(const LdlView &v, const SnodeView &sv, const SnodeGeom &g, int sn,
                                               const int *colbase, double *Tw, const snode_v4d *acc4, int jbase,
                                               int nc, int i0, bool atomic_emit, int lane) {
    const int kq = lane >> 4, l15 = lane & 15;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jj = l15 + 16 * c, rr = kq + 4 * r;
            Tw[jj * 16 + (rr ^ l15)] = acc4[c][r];
        }
    __builtin_amdgcn_wave_barrier();
    const int i = i0 + l15; // lane = row of the tile; an instruction covers columns 4 m + kq
    if (i < g.h) {
        if (!EXTEND && !atomic_emit) {
            double cur[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int jj = 4 * m + kq, j = jbase + jj;
                cur[m] = (jj < nc && i > j) ? v.Lx[colbase[j] + i] : 0.0;
            }
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int jj = 4 * m + kq, j = jbase + jj;
                if (jj >= nc) continue;
                const double val = Tw[jj * 16 + (l15 ^ (jj & 15))];
                if (i > j) v.Lx[colbase[j] + i] = cur[m] - val;
                else if (i == j) v.D[g.cols[j]] -= val;
            }
        } else if (!EXTEND) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int jj = 4 * m + kq, j = jbase + jj;
                if (jj >= nc) continue;
                const double val = Tw[jj * 16 + (l15 ^ (jj & 15))];
                if (i > j) atomicAdd(&v.Lx[colbase[j] + i], -val);
                else if (i == j) atomicAdd(&v.D[g.cols[j]], -val);
            }
        } else {
            const int rB = i - g.w;
            const int *Bn = v.Li + v.Lp[g.e];
            if (sv.U) {
                double *Us = sv.U + sv.asm_uoff[sn];
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    const int jj = 4 * m + kq, cB = jbase + jj - g.w;
                    if (jj >= nc) continue;
                    const double val = Tw[jj * 16 + (l15 ^ (jj & 15))];
                    if (rB > cB) Us[(long long)cB * g.nb - (long long)cB * (cB + 1) / 2 + (rB - cB - 1)] = val;
                    else if (rB == cB) sv.Ud[sv.asm_doff[sn] + cB] = val;
                }
            } else {
                const long long ubase = sv.upd_ptr[sn];
#pragma unroll
                for (int m0 = 0; m0 < 16; m0 += 8) {
                    int slot[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int jj = 4 * (m0 + m) + kq, cB = jbase + jj - g.w;
                        const bool lower = jj < nc && rB > cB;
                        slot[m] = lower ? sv.upd_slot[ubase + (long long)cB * g.nb - (long long)cB * (cB + 1) / 2 + (rB - cB - 1)] : -1;
                    }
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int jj = 4 * (m0 + m) + kq, cB = jbase + jj - g.w;
                        const double val = Tw[jj * 16 + (l15 ^ (jj & 15))];
                        if (slot[m] >= 0) atomicAdd(&v.Lx[slot[m]], -val);
                        else if (jj < nc && rB == cB) atomicAdd(&v.D[Bn[cB]], -val);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier(); // (the next slice overwrites Tw)
}

// acc[16 rows of this wave, 256 columns from jrow0] += L[rows, kbeg..kend) (d L[jrow0.., k])'; emitted per element.
// Wl: two buffers of SNW_KC x SNW_NC doubles.
template <bool EXTEND, int NW>
__device__ __forceinline__ void snode_tiles_wide(const LdlView &v, const SnodeView &sv, const SnodeGeom &g, int sn,
                                                 const int *colbase, double *Wl, int jrow0, int ncols, int kend,
                                                 int row_begin, int kbeg, bool atomic_emit) {
    constexpr int SNW_KC = SNW_SR * NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, l15 = lane & 15;
    const int i0 = row_begin + wave * 16;
    const int irc = min(i0 + l15, g.h - 1); // (clamped: a row beyond the panel reads the last row, never emitted)
    const bool wave_live = i0 < g.h;
    // tiles at or below the diagonal only: columns jrow0 + 16 c <= i0 + 15
    const int ncw = min((ncols + 15) / 16, (i0 + 15 - jrow0) / 16 + 1);
    snode_v4d acc[SNW_NC / 16];
#pragma unroll
    for (int c = 0; c < SNW_NC / 16; ++c) acc[c] = snode_v4d{0.0, 0.0, 0.0, 0.0};
    double a[SNW_U];
    auto request = [&](int u, int kabs) { a[u] = v.Lx[colbase[min(kabs + kq, kend - 1)] + irc]; };
    // the staged operand: this wave's SNW_SR k rows of a chunk, the 256 columns as four runs of 64 lanes
    double wv[SNW_SR][4], dv[SNW_SR];
    auto stage_request = [&](int kc0) {
#pragma unroll
        for (int r = 0; r < SNW_SR; ++r) {
            const int k = min(kc0 + wave + r * NW, kend - 1); // (clamped: no branch per load)
            const int cb = colbase[k] + jrow0;
            dv[r] = g.d[k];
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[r][q] = v.Lx[cb + min(lane + 64 * q, ncols - 1)];
        }
    };
    auto stage_commit = [&](double *Wb, int kc0) {
#pragma unroll
        for (int r = 0; r < SNW_SR; ++r) {
            const int kl = wave + r * NW;
            const bool kok = kc0 + kl < kend;
#pragma unroll
            for (int q = 0; q < 4; ++q) Wb[snw_at(kl, lane + 64 * q)] = (kok && lane + 64 * q < ncols) ? wv[r][q] * dv[r] : 0.0;
        }
    };
    lds_barrier(); // (the caller has just filled colbase)
    if (kbeg < kend) {
        stage_request(kbeg);
        if (wave_live) {
#pragma unroll
            for (int u = 0; u < SNW_U; ++u) request(u, kbeg + 4 * u);
        }
        stage_commit(Wl, kbeg);
    }
    lds_barrier();
    int buf = 0;
    for (int kc0 = kbeg; kc0 < kend; kc0 += SNW_KC, buf ^= 1) {
        const bool more = kc0 + SNW_KC < kend;
        if (more) stage_request(kc0 + SNW_KC);
        const double *Wb = Wl + buf * (SNW_KC * SNW_NC);
        if (wave_live) {
            // (whole chunks: rows beyond kend are stored zeros, the A operands there clamped to valid entries)
            const int swz = (kq & 1) << 4;
            const double *pe = Wb + kq * SNW_NC + l15 + swz, *po = Wb + kq * SNW_NC + l15 - swz;
            auto chunk = [&](auto nc) {
                constexpr int NCW = decltype(nc)::value;
                static_assert(SNW_KC == 4 * SNW_U || SNW_KC == 8 * SNW_U, "one or two rounds of the operand ring per chunk");
#define SNW_G(KK, UU) snw_group<NCW, KK + 4 * UU>(acc, a[UU], pe, po); request(UU, kc0 + KK + 4 * UU + 4 * SNW_U);
                SNW_G(0, 0) SNW_G(0, 1) SNW_G(0, 2) SNW_G(0, 3)
                if (SNW_KC == 8 * SNW_U) { SNW_G(16, 0) SNW_G(16, 1) SNW_G(16, 2) SNW_G(16, 3) }
#undef SNW_G
            };
            if (ncw > 12) chunk(std::integral_constant<int, 16>{});
            else if (ncw > 8) chunk(std::integral_constant<int, 12>{});
            else if (ncw > 4) chunk(std::integral_constant<int, 8>{});
            else chunk(std::integral_constant<int, 4>{});
        }
        if (more) stage_commit(Wl + (buf ^ 1) * (SNW_KC * SNW_NC), kc0 + SNW_KC);
        lds_barrier();
    }
    if (!wave_live) return;
    double *Tw = Wl + wave * (16 * SN_NB);
#pragma unroll
    for (int cs = 0; cs < SNW_NC / SN_NB; ++cs) {
        if (4 * cs >= ncw) break;
        snw_emit_slice<EXTEND>(v, sv, g, sn, colbase, Tw, &acc[4 * cs], jrow0 + cs * SN_NB,
                               min(SN_NB, ncols - cs * SN_NB), i0, atomic_emit, lane);
    }
}

template <int NW> __device__ __forceinline__ int *snode_lds_wide(char *smem, double *&Wl) {
    Wl = (double *)smem;
    return (int *)(Wl + 2 * SNW_SR * NW * SNW_NC);
}
// the ancestors' update in (16 NW) x 256 tiles: gx row groups x gy column blocks x ks shares of the member columns.
// NW = 8: one workgroup per CU (2 x 64 KiB of LDS); NW = 4: two (2 x 32 KiB each), the other one's matrix instructions
// cover a workgroup's first loads, its barriers and the way its tiles leave.
template <int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_snode_extend_wide(LdlView v, SnodeView sv, const int *__restrict__ order, int gx, int gy, int count, int ks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Wl;
    int *colbase = snode_lds_wide<NW>(smem, Wl);
    const int id = (int)blockIdx.x, r = id & 7, q = id >> 3, T = gx * gy * ks, tile = q % T, rem = tile % (gx * gy);
    const int bz = 8 * (q / T) + r;
    if (bz >= count) return;
    const int ksi = tile / (gx * gy), by = rem / gx, bx = rem % gx;
    int sn;
    const SnodeGeom g = snode_geom(sv, order, bz, sn);
    const int c0 = by * SNW_NC;
    if (c0 >= g.nb) return;
    const int row_begin = g.w + bx * (16 * NW);
    if (row_begin >= g.h || row_begin + 16 * NW <= g.w + c0) return; // (beyond the panel / above the diagonal)
    const int nunits = (g.w + SN_NB - 1) / SN_NB;
    const int kbeg = (int)(((long long)nunits * ksi) / ks) * SN_NB, kend = min(g.w, (int)(((long long)nunits * (ksi + 1)) / ks) * SN_NB);
    if (kbeg >= kend) return;
    for (int t = threadIdx.x; t < g.w; t += 64 * NW) colbase[t] = g.cb[t];
    snode_tiles_wide<true, NW>(v, sv, g, sn, colbase, Wl, g.w + c0, min(SNW_NC, g.nb - c0), kend, row_begin, kbeg, false);
}

// grid (target columns of the level): the update matrices the level's supernodes left in U, summed per target column
// in LDS -- source after source, a barrier between them: two sources may hit the same row from different threads --
// and subtracted from the column once.  The sums have a fixed order (the sources are sorted by supernode on the host):
// what the fp64 atomics of k_snode_extend cannot promise, and ~200 supernodes hammering the same ~4e5 addresses
// (config 5's leaf level) was the slowest part of that launch.  The headers of up to SNA_WG sources are staged in LDS
// together and the entries of source k + 1 are requested before those of source k are added; a column longer than the
// LDS buffer is done in windows of SNA_CAP rows (its sources are read once per window).
constexpr int SNA_WG = 256;
constexpr int SNA_CAP = 4096; // doubles of LDS per target column (32 KiB: four workgroups per CU)
constexpr int SNA_PF = 4;     // entries per thread of one source, requested together
__global__ __launch_bounds__(SNA_WG) void k_snode_assemble(LdlView v, SnodeView sv, SnodeAsmView av, int cap) {
    __shared__ double acc[SNA_CAP];
    __shared__ long long h_uo[SNA_WG], h_so[SNA_WG];
    __shared__ int h_cnt[SNA_WG], h_do[SNA_WG];
    const int t = av.t0 + (int)blockIdx.x, tid = threadIdx.x;
    const int c = av.tgt[t];
    const int q0 = av.src_ptr[t], q1 = av.src_ptr[t + 1];
    const int base = v.Lp[c], len = v.Lp[c + 1] - base;
    for (int w0 = 0; w0 == 0 || w0 < len; w0 += cap) {
        const int nl = max(0, min(len - w0, cap));
        for (int i = tid; i < nl; i += SNA_WG) acc[i] = 0.0;
        double dacc = 0.0; // thread 0, first window: the diagonal entries, in source order
        for (int qc = q0; qc < q1; qc += SNA_WG) {
            const int nq = min(SNA_WG, q1 - qc);
            __syncthreads(); // (the previous chunk's headers have been consumed; acc is zeroed)
            if (tid < nq) {
                const long long *hq = av.src + 3 * (long long)(qc + tid);
                const long long cd = hq[2];
                h_uo[tid] = hq[0];
                h_so[tid] = hq[1];
                h_cnt[tid] = (int)(cd & 0xffffffffll);
                h_do[tid] = (int)(cd >> 32);
            }
            __syncthreads();
            int slot[SNA_PF], slotn[SNA_PF];
            double val[SNA_PF], valn[SNA_PF], dn = 0.0;
            auto request = [&](int k, int i0, int(&sl)[SNA_PF], double(&vl)[SNA_PF]) {
                const long long uo = h_uo[k], so = h_so[k];
                const int cnt = h_cnt[k];
#pragma unroll
                for (int u = 0; u < SNA_PF; ++u) {
                    const int i = max(0, min(i0 + u * SNA_WG, cnt - 1)); // (clamped: unconditional loads; cnt = 0: entry 0 of the
                    sl[u] = sv.upd_slot[so + i];                          //  next source or the array's spare element, never used)
                    vl[u] = sv.U[uo + i];
                }
            };
            auto apply = [&](int k, int i0, const int(&sl)[SNA_PF], const double(&vl)[SNA_PF]) {
                const int cnt = h_cnt[k];
#pragma unroll
                for (int u = 0; u < SNA_PF; ++u) {
                    const int p = sl[u] - base - w0;
                    if (i0 + u * SNA_WG < cnt && p >= 0 && p < nl) acc[p] += vl[u]; // (the rows of one source are distinct)
                }
            };
            request(0, tid, slotn, valn);
            if (tid == 0 && w0 == 0) dn = sv.Ud[h_do[0]];
            for (int k = 0; k < nq; ++k) {
#pragma unroll
                for (int u = 0; u < SNA_PF; ++u) slot[u] = slotn[u], val[u] = valn[u];
                const double dk = dn;
                if (k + 1 < nq) {
                    request(k + 1, tid, slotn, valn);
                    if (tid == 0 && w0 == 0) dn = sv.Ud[h_do[k + 1]];
                }
                dacc += dk;
                apply(k, tid, slot, val);
                for (int i0 = tid + SNA_WG * SNA_PF; i0 < h_cnt[k]; i0 += SNA_WG * SNA_PF) { // (more than 1024 rows of B)
                    request(k, i0, slot, val);
                    apply(k, i0, slot, val);
                }
                __syncthreads();
            }
        }
        __syncthreads();
        for (int i = tid; i < nl; i += SNA_WG) {
            const double a = acc[i];
            if (a != 0.0) v.Lx[base + w0 + i] -= a;
        }
        if (tid == 0 && w0 == 0) v.D[c] -= dacc;
        __syncthreads();
    }
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
    // the rows below the block in groups of SNP_WG; workgroup x takes the groups x, x + gridDim.x, ... (a launch with
    // many supernodes gets fewer workgroups per supernode than groups: each factors the diagonal block once and then
    // walks several groups, instead of nine workgroups per supernode all repeating the 23 us block factorisation)
    const int row_first = j0 + nbw + (int)blockIdx.x * SNP_WG; // first row of this workgroup's first group
    if (blockIdx.x > 0 && row_first >= g.h) return;
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
        const int nwg = max(1, min((int)gridDim.x, (g.h - (j0 + nbw) + SNP_WG - 1) / SNP_WG)); // workgroups of this supernode that got past the early return
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
    for (int row0 = row_first; row0 < g.h; row0 += (int)gridDim.x * SNP_WG) {
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
        if (R0w >= g.h) continue;                 // (whole wave beyond the panel; no workgroup barrier below)
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
        continue;
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
    if (!rowok) continue;
#pragma unroll
    for (int jj = 0; jj < SN_NB; ++jj)
        if (jj < nbw) v.Lx[colbase[jj] + R] = x[jj] * dinvl[jj];
    sn_stamp(sv.dbg, dbgme, 7, true);
    if (sv.dbg && tid == 0) atomicMax((unsigned long long *)&sv.dbg[8], (unsigned long long)wall_clock64()); // last workgroup's end
    } // (groups of rows)
}

// ---------------------------------------------------------------------------
// k_snode_panel with the two phases OVERLAPPED (round 4).  In k_snode_panel the block factorisation goes quarter by
// quarter -- one wave does 16 pivots while the other three wait, ~5.8 us per quarter -- and only then do the rows below
// the block start their own four quarter steps (~4.8 us each): 23 + 19 us one after the other, on the critical path of
// every one of config 2's 129 block columns.  Rows step kb needs nothing but quarter kb of the factored block.  Here a
// workgroup has EIGHT waves: waves 0-3 factor the block exactly as before, waves 4-7 hold the first group of 256 rows
// in registers and take step kb right after the barrier that publishes quarter kb, while the block team is busy with
// quarter kb + 1.  Afterwards all eight waves walk the workgroup's remaining row groups (512 rows per round).  Both
// phases in their matrix-core forms only (the scalar forms stay in k_snode_panel); Ll and the waves' head blocks in
// dynamic LDS (~104 KiB).
// ---------------------------------------------------------------------------
constexpr int SNQ_WG = 512;
constexpr int SNQ_GS = 8; // pivots a lane factors by itself at a time (k_snode_panel2<true>)
__device__ __forceinline__ void snq_rows(const LdlView &v, const SnodeGeom &g, const double *Ll, const double *dinvl,
                                         const int *colbase, double *xw, int R0w, int nbw, int lane, int kb_from, int kb_to,
                                         snode_v4d (&acc)[3][4], double (&h)[16], bool load) {
    // rows R0w .. R0w + 63 of the panel (lane = row in the head-block form): steps kb_from .. kb_to - 1 of the blocked
    // recurrence of k_snode_panel; `load`: fetch the rows first (step 0 only)
    const int l15 = lane & 15, kq = lane >> 4;
    const int R = R0w + lane;
    const bool rowok = R < g.h;
    const int Rc = min(R, g.h - 1);
    if (load) {
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
#pragma unroll
        for (int c = 0; c < 16; ++c) h[c] = v.Lx[colbase[c] + Rc];
#pragma unroll
        for (int c = 0; c < 16; ++c) h[c] = c < nbw ? h[c] : 0.0;
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        if (kb < kb_from || kb >= kb_to) continue;
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
        const double *Lb = Ll + (16 * kb) * SN_NB + 16 * kb;
#pragma unroll
        for (int kk = 0; kk < 15; ++kk) {
            const double uq = h[kk];
            int zoff; // (ties the column's LDS reads to its place in the chain, see k_snode_panel)
            asm volatile("v_mov_b32 %0, 0" : "=v"(zoff) : "v"(__double2hiint(uq)));
            const snode_v2d *cf = (const snode_v2d *)(Lb + kk * SN_NB + zoff);
#pragma unroll
            for (int p2 = (kk + 1) / 2; p2 < 8; ++p2) {
                const snode_v2d cc = cf[p2];
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
#pragma unroll
        for (int c = 0; c < 16; ++c) xw[lane * SNP_XLD + c] = -h[c];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            double a4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a4[t] = xw[(16 * t + l15) * SNP_XLD + 4 * s4 + kq];
#pragma unroll
            for (int jb = kb + 1; jb < 4; ++jb) {
                const double bv = Ll[(16 * kb + 4 * s4 + kq) * SN_NB + 16 * jb + l15];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[jb - 1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[t], bv, acc[jb - 1][t], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// UNI (round 5): the owner wave factors its quarter in groups of SNQ_GS pivots WITHOUT cross-lane traffic on the pivot
// chain.  A group's SNQ_GS x SNQ_GS diagonal sub-block goes through LDS to EVERY lane, which factors it by itself (the
// divisions, the multiply-adds, the sign rule) and applies the group's pivots to its own row with coefficients it now
// holds; the quarter's columns behind the group take the group's pivots as one block (coefficients: the scaled
// entries of the rows behind the sub-block, through LDS).  The readlane form broadcast the pivot and every
// coefficient l(c', c) from its lane -- two v_readlane per product on the chain of all 64 pivots, ~360 ns per pivot.
// Per entry the same operations in the same order: the factors are bitwise those of the readlane form
// (CHIP_NO_PANEL_UNIFORM).
template <bool UNI>
__global__ __launch_bounds__(SNQ_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_snode_panel2(LdlView v, SnodeView sv, const int *__restrict__ order,
                                                                                                   int b) {
    extern __shared__ __attribute__((aligned(16))) char qsm[];
    double *Ll = (double *)qsm;                 // Ll[k * 64 + i] = l(i, k), 0 for i <= k
    double *xhb = Ll + SN_NB * SN_NB;           // eight head blocks, [row][column], stride SNP_XLD
    __shared__ double dinvl[SN_NB], sgn[SN_NB];
    __shared__ __attribute__((aligned(16))) double sdl[64 + 16 * SNQ_GS]; // UNI: a group's sub-block | the coefficients of the cross update
    __shared__ int colbase[SN_NB];
    __shared__ int s_nreg, s_bad;
    int sn;
    const SnodeGeom g = snode_geom(sv, order, (int)blockIdx.y, sn);
    const int j0 = b * SN_NB;
    if (j0 >= g.w) return;
    const int nbw = min(SN_NB, g.w - j0), tid = threadIdx.x, i = tid & 63, wave = tid >> 6;
    const bool teamD = wave < 4;
    const int q = wave & 3;
    // row groups of 256 below the block; workgroup x takes the groups x, x + gridDim.x, ...: the first one by the rows
    // team alongside the block factorisation, the others by all eight waves
    const int row_first = j0 + nbw + (int)blockIdx.x * SNP_WG;
    if (blockIdx.x > 0 && row_first >= g.h) return;
    const bool live = i < nbw;
    if (tid < SN_NB) {
        colbase[tid] = g.cb[j0 + min(tid, nbw - 1)];
        sgn[tid] = tid < nbw ? (double)g.sg[j0 + tid] : 1.0;
    }
    if (tid == 0) {
        s_nreg = 0;
        s_bad = 0;
    }
    __syncthreads();
    const int ci = live ? g.cols[j0 + i] : 0;
    const double sgl = sgn[i];
    double dfin = 1.0, dinvfin = 1.0;
    int nreg = 0, bad = 0;
    const int l15 = i & 15, kq = i >> 4;
    double *xw = xhb + wave * (64 * SNP_XLD);
    // A wave belongs to ONE team for the block column's first pass, and the two teams run their own loops over the quarters
    // (wave-uniform branches; both execute the same four barriers): the block team's quarter, its lane = row copy and the
    // sub-blocks it factors are then never live together with the rows team's 64 x 64 tile -- in one common loop the
    // register allocator had to keep both sets (more than 256 registers, spills on the pivot chain).
    double T[16]; // block team: the owner's quarter in the lane = row form (scaled on the way out: written back below)
    const int R0first = row_first + 64 * q; // rows team: this wave's 64 rows of the first group
    if (teamD) {
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
#pragma unroll 1
        for (int kb = 0; kb < SN_NB / 16; ++kb) {
            if (q == kb) { // (as in k_snode_panel: the owner factors its quarter in the lane = row form)
    #pragma unroll
                for (int t = 0; t < 4; ++t)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) xw[(16 * t + kq + 4 * r) * SNP_XLD + l15] = Aq[t][r];
                __builtin_amdgcn_wave_barrier();
    #pragma unroll
                for (int c = 0; c < 16; ++c) T[c] = xw[i * SNP_XLD + c];
                __builtin_amdgcn_wave_barrier();
                if (UNI) {
                    const int cbase = 16 * kb;
                    constexpr int GS = SNQ_GS; // pivots per group
    #pragma unroll
                    for (int gq = 0; gq < 16 / GS; ++gq) {
                        const int a_me = i - cbase - GS * gq; // this lane's row inside the group's GS x GS sub-block (if 0 <= a_me < GS)
                        if (a_me >= 0 && a_me < GS) {
    #pragma unroll
                            for (int bb = 0; bb < GS; ++bb) sdl[a_me * GS + bb] = T[GS * gq + bb];
                        }
                        __builtin_amdgcn_wave_barrier();
                        double S[GS][GS];
    #pragma unroll
                        for (int a = 0; a < GS; ++a)
    #pragma unroll
                            for (int bb = 0; bb <= a; ++bb) S[a][bb] = sdl[a * GS + bb];
                        __builtin_amdgcn_wave_barrier();
                        double ug[GS]; // this row's unscaled entries of the group's columns
    #pragma unroll
                        for (int t = 0; t < GS; ++t) {
                            const int c = cbase + GS * gq + t;
                            double d = S[t][t];
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
                            double la[GS];
    #pragma unroll
                            for (int a = t + 1; a < GS; ++a) la[a] = S[a][t] * dinv; // l(c + a - t, c), known to every lane
    #pragma unroll
                            for (int a = t + 1; a < GS; ++a)
    #pragma unroll
                                for (int bb = t + 1; bb <= a; ++bb) S[a][bb] -= la[bb] * S[a][t];
                            const double uc = i > c ? T[GS * gq + t] : 0.0;
                            const double l = uc * dinv;
                            ug[t] = uc;
                            T[GS * gq + t] = l;
                            xw[i * SNP_XLD + GS * gq + t] = -uc;
                            Ll[c * SN_NB + i] = l;
                            if (i == 0) dinvl[c] = dinv;
    #pragma unroll
                            for (int t2 = t + 1; t2 < GS; ++t2) T[GS * gq + t2] -= la[t2] * uc;
                        }
                        constexpr int NX_MAX = 16 - GS;
                        const int nx = 16 - GS * (gq + 1); // columns of the quarter behind the group
                        if (nx > 0) {
                            // they take the group's pivots, t = 0 .. GS - 1 in order: the coefficients l(cbase + GS (gq + 1) + b, c_t) are
                            // the scaled entries of the rows behind the sub-block -- those lanes' T[GS gq + t]
                            const int b_me = i - cbase - GS * (gq + 1);
                            if (b_me >= 0 && b_me < nx) {
    #pragma unroll
                                for (int t = 0; t < GS; ++t) sdl[64 + t * 16 + b_me] = T[GS * gq + t]; // [pivot][row]: a pivot's coefficients contiguous
                            }
                            __builtin_amdgcn_wave_barrier();
    #pragma unroll
                            for (int t = 0; t < GS; ++t) {
                                // (an opaque zero that depends on the previous pivot's first result keeps the compiler from
                                // requesting all coefficients at once ahead of the first product)
                                int zoff;
                                asm volatile("v_mov_b32 %0, 0" : "=v"(zoff) : "v"(__double2hiint(T[GS * (gq + 1)])));
                                const snode_v2d *cf = (const snode_v2d *)(sdl + 64 + t * 16 + zoff);
    #pragma unroll
                                for (int p2 = 0; p2 < NX_MAX / 2; ++p2) {
                                    if (2 * p2 >= nx) continue; // (compile time after unrolling)
                                    const snode_v2d cc = cf[p2];
                                    T[GS * (gq + 1) + 2 * p2] -= cc.x * ug[t];
                                    T[GS * (gq + 1) + 2 * p2 + 1] -= cc.y * ug[t];
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                } else {
    #pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int c = 16 * kb + t;
                    double d = readlane_f64(T[t], c);
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
                    for (int t2 = t + 1; t2 < 16; ++t2) T[t2] -= readlane_f64(l, 16 * kb + t2) * uc;
                }
                }
            }

            __syncthreads(); // quarter kb of Ll, its pivots and the owner's negated panel are published
            if (q > kb) {
                const double *xo = xhb + kb * (64 * SNP_XLD);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const double bv = Ll[(16 * kb + 4 * s4 + kq) * SN_NB + 16 * q + l15];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double av = xo[(16 * t + l15) * SNP_XLD + 4 * s4 + kq];
                        Aq[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, Aq[t], 0, 0, 0);
                    }
                }
            }
        }
    } else {
        const bool rlive = R0first < g.h;
        snode_v4d racc[3][4];
        double rh[16];
        if (rlive) snq_rows(v, g, Ll, dinvl, colbase, xw, R0first, nbw, i, 0, 0, racc, rh, true); // (loads only)
#pragma unroll 1
        for (int kb = 0; kb < SN_NB / 16; ++kb) {
            __syncthreads(); // (the block team's barrier of quarter kb)
            if (rlive) snq_rows(v, g, Ll, dinvl, colbase, xw, R0first, nbw, i, kb, kb + 1, racc, rh, false);
        }
    }
    if (nreg) atomicAdd(&s_nreg, 1);
    if (bad) atomicOr(&s_bad, bad);
    __syncthreads();
    // the factored block goes back in place, written by the LAST of the supernode's workgroups to get here (see k_snode_panel)
    if (tid == 0) {
        const int nwg = max(1, min((int)gridDim.x, (g.h - (j0 + nbw) + SNP_WG - 1) / SNP_WG));
        const int old = __hip_atomic_fetch_add(&sv.sn_cnt[sn], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_nreg = old == nwg - 1 ? (s_nreg | 0x40000000) : s_nreg;
        if (old == nwg - 1) __hip_atomic_store(&sv.sn_cnt[sn], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool writer = (s_nreg & 0x40000000) != 0;
    if (writer && teamD) {
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
        // (T holds the scaled quarter in the lane = row form only for the quarter's owner: wave q owns columns 16 q ..)
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const int j = 16 * q + cc;
            if (live && j < nbw && i > j) v.Lx[colbase[j] + j0 + i] = T[cc];
        }
    }
    // the workgroup's other row groups: all eight waves, two groups (512 rows) per round
    const int gstride = (int)gridDim.x * SNP_WG;
    for (int base = row_first + gstride; base < g.h; base += 2 * gstride) {
        const int R0w = (teamD ? base : base + gstride) + 64 * q;
        snode_v4d racc[3][4];
        double rh[16];
        if (R0w < g.h) snq_rows(v, g, Ll, dinvl, colbase, xw, R0w, nbw, i, 0, 4, racc, rh, true);
    }
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
// Eight waves per block (round 4; four before): a step's 64 columns are eight per wave instead of sixteen -- half the
// streamed entries per lane and step --, the diagonal block is fetched and the rows of B are pulled / pushed by twice
// the threads (that part sits in front of the first hop of every backward launch and at the tail of every forward
// one).  Config 2: 25.0 -> 23.6 ms per step, config 5: 47.4 -> 45.8.  (Sixteen waves leave 128 registers per lane: the
// wave that solves the diagonal block holds its 64 coefficients in registers and needs more.)
constexpr int SN2_WG = 512;
// NR right-hand sides per pass (round 6): the two independent solves of an interior-point iteration (kktsystem.rs:108-125,
// core/solver.rs:351-361) and their refinement rounds walk the SAME entries of L -- with NR = 2 a launch streams the
// panel once and applies it to both vectors (their messages live in two buffers, one per solve context).  These sweeps
// run at ~0.5 of the HBM peak: the bytes are what is left to save.
template <bool FWDMODE, int NR>
__global__ __launch_bounds__(SN2_WG) void k_snode_tri(LdlView v, SnodeView sv, const int *__restrict__ order,
                                                      const int *__restrict__ blk_ptr, int *msg0, int *msg1, int epoch0,
                                                      int epoch1, double *x0, double *x1, int *timeout_flag, int *timeout_flag1) {
    __shared__ double Tl[SN_NB * SN_NB];
    __shared__ double part[NR][SN2_WG / 64][SN_NB];
    __shared__ double pulled[NR][SN_NB]; // backward: L_B,r' x_B of the own columns; forward: the finished x_r
    // (dynamic: as many column bases as the level's widest supernode has columns -- a fixed SN2_WMAX array cost 16 KB and,
    // with the second right-hand side's partial sums, the third workgroup of a CU)
    extern __shared__ int colbase[]; // forward: of all earlier columns; backward: of the own block only
    int sn;
    const SnodeGeom g = snode_geom(sv, order, (int)blockIdx.y, sn);
    const int nblk = (g.w + SN_NB - 1) / SN_NB;
    if ((int)blockIdx.x >= nblk) return;
    const int r = FWDMODE ? (int)blockIdx.x : nblk - 1 - (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = r * SN_NB, nbw = min(SN_NB, g.w - j0);
    double *const xv_[2] = {x0, NR > 1 ? x1 : x0};
    const int ep_[2] = {epoch0, NR > 1 ? epoch1 : epoch0};
    // this supernode's message slots: block c, lane l at (c * 64 + l) * 4
    int *const mb_[2] = {msg0 + (size_t)blk_ptr[sn] * 256, (NR > 1 ? msg1 : msg0) + (size_t)blk_ptr[sn] * 256};
    constexpr int CPW = SN_NB / (SN2_WG / 64); // columns per wave: of block c (forward) / of the own block (backward)
    // nothing below depends on x: column bases, the diagonal block and the own entries are requested first
    const int cb_lo = FWDMODE ? 0 : j0, cb_hi = j0 + nbw;
    for (int t = cb_lo + tid; t < cb_hi; t += SN2_WG) colbase[t - cb_lo] = g.cb[t];
    double xown[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) xown[k] = (wave == 0 && lane < nbw) ? xv_[k][g.cols[j0 + lane]] : 0.0; // (last written before this launch)
    __syncthreads();
    const int *cbr = colbase + (FWDMODE ? j0 : 0); // column bases of the own block
    const int *Bn = v.Li + v.Lp[g.e];              // node ids of the rows of B
    if (!FWDMODE) {
        // the block's own columns first take D^-1 and the rows of B: x_j <- x_j / d_j - sum_r L(B_r, j) x(B_r)
        // (what k_snode_pull did in a launch of its own; here every block does it while it would otherwise
        // wait for the flags of the later blocks).  Rows of B along the lanes, this wave's CPW columns together.
        double pacc[NR][CPW];
#pragma unroll
        for (int k = 0; k < NR; ++k)
#pragma unroll
            for (int q = 0; q < CPW; ++q) pacc[k][q] = 0.0;
        for (int r0 = 0; r0 < g.nb; r0 += 64) {
            const int rr = r0 + lane;
            const bool rok = rr < g.nb;
            const int node = rok ? Bn[rr] : 0;
            double xb[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) xb[k] = rok ? xv_[k][node] : 0.0;
            double lq[CPW];
#pragma unroll
            for (int q = 0; q < CPW; ++q) {
                const int j = wave * CPW + q;
                lq[q] = (rok && j < nbw) ? v.Lx[cbr[j] + g.w + rr] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < NR; ++k)
#pragma unroll
                for (int q = 0; q < CPW; ++q) pacc[k][q] += lq[q] * xb[k];
        }
#pragma unroll
        for (int k = 0; k < NR; ++k)
#pragma unroll
            for (int q = 0; q < CPW; ++q) {
                const double tot = wave_sum(pacc[k][q]);
                if (lane == 0) pulled[k][wave * CPW + q] = tot;
            }
        __syncthreads();
        if (wave == 0 && lane < nbw) {
            const double dinv = v.Dinv[g.cols[j0 + lane]];
#pragma unroll
            for (int k = 0; k < NR; ++k) xown[k] = xown[k] * dinv - pulled[k][lane];
        }
    }
    // the diagonal block with the solve's lane index fastest (snode_block_solve): forward Tl[col * SN_NB + row],
    // backward Tl[row * SN_NB + col]
    for (int idx = tid; idx < SN_NB * SN_NB; idx += SN2_WG) {
        const int hi = idx / SN_NB, lo = idx % SN_NB;
        const int ii = FWDMODE ? lo : hi, jj = FWDMODE ? hi : lo;
        Tl[idx] = (ii > jj && ii < nbw) ? v.Lx[cbr[jj] + j0 + ii] : 0.0;
    }
    double acc[NR][CPW];
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
        for (int q = 0; q < CPW; ++q) acc[k][q] = 0.0;
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
        // every lane polls the message(s) of "its" unknown of block c
        msg_v4i mm[NR];
        for (long long spins = 0;; ++spins) {
            bool got = true;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                mm[k] = msg_load(mb_[k] + (c * 64 + lane) * 4);
                got = got && (lane >= ncw || (mm[k].y == ep_[k] && mm[k].w == ep_[k]));
            }
            if (__all(got)) break;
            __builtin_amdgcn_s_sleep(1);
            if (spins > (1ll << 18)) {
                ok = false;
                if (lane == 0) {
                    *timeout_flag = 1;
                    if (NR > 1) *timeout_flag1 = 1; // (both solves see the failure)
                }
                break;
            }
        }
        if (!ok) break;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const double xcv = lane < ncw ? __hiloint2double(mm[k].z, mm[k].x) : 0.0;
            if (FWDMODE) {
#pragma unroll
                for (int q = 0; q < CPW; ++q) acc[k][0] += lv[q] * readlane_f64(xcv, wave * CPW + q); // (a wave-uniform lane: v_readlane, not a trip through the LDS crossbar)
            } else {
#pragma unroll
                for (int q = 0; q < CPW; ++q) acc[k][q] += lv[q] * xcv;
            }
        }
    }
    // reduce: forward across the waves (each holds its columns' share of every row), backward across lanes
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        if (FWDMODE) {
            part[k][wave][lane] = acc[k][0];
        } else {
#pragma unroll
            for (int q = 0; q < CPW; ++q) {
                const double tot = wave_sum(acc[k][q]);
                if (lane == 0) part[k][0][wave * CPW + q] = tot;
            }
        }
    }
    __syncthreads(); // (also: Tl is complete)
    if (wave == 0 && ok) {
        double xr[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            double xv = xown[k];
            if (lane < nbw) {
                if (FWDMODE) {
#pragma unroll
                    for (int w = 0; w < SN2_WG / 64; ++w) xv -= part[k][w][lane];
                } else {
                    xv -= part[k][0][lane];
                }
            }
            xr[k] = xv;
        }
        snode_block_solve_n<FWDMODE, NR>(Tl, xr, lane);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            if (lane < nbw) {
                msg_store(mb_[k] + (r * 64 + lane) * 4, xr[k], ep_[k]); // to the other blocks of this sweep
                xv_[k][g.cols[j0 + lane]] = xr[k];                      // to the launches that follow
            }
            if (FWDMODE) pulled[k][lane] = lane < nbw ? xr[k] : 0.0;
        }
    }
    if (FWDMODE && g.nb > 0) {
        // after the flag (off the pipeline's critical path): this block's share of x_B -= L_BS x_S, one row of B
        // per thread, one atomic per (row, block, vector) -- what k_snode_push did in a launch of its own
        __syncthreads();
        if (!ok) return;
        for (int rb = tid; rb < g.nb; rb += SN2_WG) {
            double sacc[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) sacc[k] = 0.0;
#pragma unroll 1
            for (int j2 = 0; j2 < SN_NB; j2 += 16) { // (not unrolled: 64 entries in flight per thread took 255 registers -- one workgroup per CU for the whole kernel)
                double lv2[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) lv2[q] = (j2 + q < nbw) ? v.Lx[cbr[j2 + q] + g.w + rb] : 0.0;
#pragma unroll
                for (int k = 0; k < NR; ++k)
#pragma unroll
                    for (int q = 0; q < 16; ++q) sacc[k] += lv2[q] * pulled[k][j2 + q];
            }
            const int node = Bn[rb];
#pragma unroll
            for (int k = 0; k < NR; ++k) atomicAdd(&xv_[k][node], -sacc[k]);
        }
    }
}


} // namespace

static size_t snode_solve_lds_bytes(int wmax, int nbcap) {
    return (size_t)(wmax + nbcap + SN_NB * SN_NB + SN_NB) * sizeof(double) + (size_t)wmax * sizeof(int);
}
constexpr size_t SNW_LDS_MAX = (size_t)160 * 1024;
static size_t g_wide_lds[2] = {0, 0}; // the largest LDS the wide kernels (4 / 8 waves) have been granted, 0: none
static size_t snode_lds_wide_bytes(int wmax, int nw) { return (size_t)2 * SNW_SR * nw * SNW_NC * sizeof(double) + (size_t)wmax * sizeof(int); }
static size_t snode_lds_bytes(int wmax) { return (size_t)(SN_KC * SN_NB) * sizeof(double) + (size_t)wmax * sizeof(int); }
int snode_kernel_attributes(int wmax, int nbmax) {
    const int lds = (int)snode_lds_bytes(wmax);
    int rc = (int)raise_dynamic_lds((const void *)k_snode_update, (size_t)lds);
    if (!rc) rc = (int)raise_dynamic_lds((const void *)k_snode_extend, (size_t)lds);
    // (the wide tiles are an option: a device that refuses their LDS keeps the 64-column tiles)
    if (!rc) {
        if (snode_lds_wide_bytes(wmax, 8) <= SNW_LDS_MAX && raise_dynamic_lds((const void *)k_snode_extend_wide<8>, snode_lds_wide_bytes(wmax, 8)) == hipSuccess)
            g_wide_lds[1] = std::max(g_wide_lds[1], snode_lds_wide_bytes(wmax, 8)); // (only ever raised, like the attribute)
        if (snode_lds_wide_bytes(wmax, 4) <= SNW_LDS_MAX && raise_dynamic_lds((const void *)k_snode_extend_wide<4>, snode_lds_wide_bytes(wmax, 4)) == hipSuccess)
            g_wide_lds[0] = std::max(g_wide_lds[0], snode_lds_wide_bytes(wmax, 4));
        (void)hipGetLastError();
    }
    const int lds2 = (int)snode_solve_lds_bytes(wmax, std::min(nbmax, SN_XB_CAP));
    if (!rc) rc = (int)raise_dynamic_lds((const void *)k_snode_panel2<true>, (size_t)(SN_NB * SN_NB + (SNQ_WG / 64) * 64 * SNP_XLD) * sizeof(double));
    if (!rc) rc = (int)raise_dynamic_lds((const void *)k_snode_panel2<false>, (size_t)(SN_NB * SN_NB + (SNQ_WG / 64) * 64 * SNP_XLD) * sizeof(double));
    if (!rc) rc = (int)raise_dynamic_lds((const void *)k_snode_fwd, (size_t)lds2);
    if (!rc) rc = (int)raise_dynamic_lds((const void *)k_snode_bwd, (size_t)lds2);
    return rc;
}
// wlvl / nblvl: maxima over the supernodes of this launch.  Levels with a large B part run it in
// separate multi-workgroup launches: one workgroup per supernode is latency bound.
bool solve_snodes_is_tri(const SnodeTriView *tri, int wlvl) { return tri && tri->msg && wlvl > 2 * SN_NB; }
void solve_snodes(hipStream_t s, GatherMode m, const LdlView &v, const SnodeView &sv, const int *order, int count,
                  int wmax_all, int nbmax_all, int wlvl, int nblvl, double *x, const SnodeTriView *tri,
                  const LaunchProf *lp, double *x2, const SnodeTriView *tri2) {
    if (!count) return;
    if (tri && tri->msg && wlvl > 2 * SN_NB) {
        if (lp) lp->begin(lp->ctx, PFK_SN_TRI);
        // wide supernodes: the triangle by several workgroups per supernode (k_snode_tri), the rows of B by
        // their own multi-workgroup launches
        const int nblkmax = (wlvl + SN_NB - 1) / SN_NB;
        const size_t cbytes = (size_t)((m == FWD ? wlvl : SN_NB) + 64) * sizeof(int); // (column bases: all earlier columns / the own block)
        if (x2 && tri2 && tri2->msg) { // two right-hand sides, one pass over the panels
            if (m == FWD)
                k_snode_tri<true, 2><<<dim3(nblkmax, count), SN2_WG, cbytes, s>>>(v, sv, order, tri->blk_ptr, tri->msg, tri2->msg, tri->epoch,
                                                                           tri2->epoch, x, x2, tri->timeout_flag, tri2->timeout_flag);
            else
                k_snode_tri<false, 2><<<dim3(nblkmax, count), SN2_WG, cbytes, s>>>(v, sv, order, tri->blk_ptr, tri->msg, tri2->msg, tri->epoch,
                                                                            tri2->epoch, x, x2, tri->timeout_flag, tri2->timeout_flag);
        } else if (m == FWD) {
            k_snode_tri<true, 1><<<dim3(nblkmax, count), SN2_WG, cbytes, s>>>(v, sv, order, tri->blk_ptr, tri->msg, tri->msg, tri->epoch,
                                                                       tri->epoch, x, x, tri->timeout_flag, tri->timeout_flag);
        } else {
            k_snode_tri<false, 1><<<dim3(nblkmax, count), SN2_WG, cbytes, s>>>(v, sv, order, tri->blk_ptr, tri->msg, tri->msg, tri->epoch,
                                                                        tri->epoch, x, x, tri->timeout_flag, tri->timeout_flag);
        }
        (void)nblvl;
        if (lp) lp->end(lp->ctx, PFK_SN_TRI);
        return;
    }
    int cap = SN_XB_CAP;
    if (switches().sn_xb_cap > 0) cap = std::min(SN_XB_CAP, switches().sn_xb_cap); // tests
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
        mode = switches().sn_debug;
        on = mode > 0;
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
                   int nblk, int hmax, int nbmax, const LaunchProf *lp, const SnodeAsmView *av) {
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
                const bool no_splitk = switches().no_splitk || switches().deterministic;
                // (256 workgroups wanted, at most 8 shares of at least one block column: 128 / 512 wanted or 16 shares
                // change config 5's step by less than the noise, round 6)
                const int split_target = 256, split_max = 8, split_unit = 1;
                while (!no_splitk && ksplit < split_max && ksplit * 2 * split_unit <= b && groups * count * ksplit < split_target) ksplit *= 2;
                pb(PFK_SN_UPDATE);
                if (dbg.mode == 2) sv.dbg = dbg.ring_slot(1) - 16 + 16; // (slots 16..20 of the launch's 32)
                k_snode_update<<<dim3(groups, count, ksplit), SN_WG, lds, s>>>(v, sv, order, b);
                pe(PFK_SN_UPDATE);
                if (dbg.on && dbg.mode != 2) dbg.collect(s, 1);
            }
        }
        const bool no_panel = switches().no_snode_panel;
        if (!no_panel) { // the diagonal block and the rows below it in one launch of one-wave workgroups
            const int below = hmax - b * SN_NB - 1;
            pb(PFK_SN_DIAG);
            if (dbg.mode == 2) sv.dbg = dbg.ring_slot(0);
            // (bit 0: rows phase on the matrix cores; bit 1: block factorisation on the matrix cores)
            const int panel_mode = (switches().no_panel_mfma ? 0 : 1) | (switches().no_panel_diag_mfma ? 0 : 2);
            // (measured on config 5: 256 / 512 / 768 / 1024 workgroups per launch -> 50.4 / 51.0 / 50.6 / 50.8 ms per step; one group per workgroup: 52.2)
            const int groups = std::max(1, (below + SNP_WG - 1) / SNP_WG);
            const int slots = switches().sn_panel_slots > 0 ? switches().sn_panel_slots : 256;
            const int gx = std::max(1, std::min(groups, slots / std::max(1, count)));
            if (panel_mode == 3 && !dbg.on && !switches().no_panel_overlap) { // both phases on the matrix cores, overlapped
                const size_t qlds = (size_t)(SN_NB * SN_NB + (SNQ_WG / 64) * 64 * SNP_XLD) * sizeof(double);
                if (switches().no_panel_uniform) k_snode_panel2<false><<<dim3(gx, count), SNQ_WG, qlds, s>>>(v, sv, order, b);
                else k_snode_panel2<true><<<dim3(gx, count), SNQ_WG, qlds, s>>>(v, sv, order, b);
            } else {
                k_snode_panel<<<dim3(gx, count), SNP_WG, 0, s>>>(v, sv, order, b, panel_mode);
            }
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
        SnodeView se = sv_in;
        if (!av || !av->nt) se.U = nullptr; // (this level scatters with atomics)
        // levels with at least one supernode per XCD and more than one wide column block: 128 x 256 tiles
        const int nw = switches().sn_wide_waves == 8 ? 8 : 4;
        const size_t lds_wide = snode_lds_wide_bytes(wmax_all, nw);
        if (lds_wide <= g_wide_lds[nw == 8 ? 1 : 0] && count >= switches().sn_wide_min_count && nbmax > SNW_NC && !switches().no_sn_wide) {
            const int gx = (nbmax + 16 * nw - 1) / (16 * nw), gy = (nbmax + SNW_NC - 1) / SNW_NC;
            const dim3 grid((unsigned)(8 * ((count + 7) / 8) * gx * gy));
            if (nw == 8) k_snode_extend_wide<8><<<grid, 512, lds_wide, s>>>(v, se, order, gx, gy, count, 1);
            else k_snode_extend_wide<4><<<grid, 256, lds_wide, s>>>(v, se, order, gx, gy, count, 1);
        } else {
            const int gx = (nbmax + SN_ROWS - 1) / SN_ROWS, gy = (nbmax + SN_NB - 1) / SN_NB;
            const bool xcd = count >= 8; // (fewer supernodes than XCDs: spread the tiles; with them spread always: 794 against 564 us)
            int ks = 1; // k-split of the tiles while the launch would leave most of the chip idle (atomics only: not with the assembled form)
            if (!se.U && !switches().no_splitk && !switches().deterministic)
                while (ks < 8 && ks * 2 <= nblk && (long long)gx * gy * count * ks * 2 <= 512) ks *= 2;
            if (xcd) k_snode_extend<<<dim3((unsigned)(8 * ((count + 7) / 8) * gx * gy * ks)), SN_WG, lds, s>>>(v, se, order, 1, gx, gy, count, ks);
            else k_snode_extend<<<dim3(gx * ks, gy, count), SN_WG, lds, s>>>(v, se, order, 0, gx, gy, count, ks);
        }
        if (se.U) {
            const int cap = switches().sn_asm_cap > 0 ? std::min(switches().sn_asm_cap, SNA_CAP) : SNA_CAP; // (tests: short windows)
            k_snode_assemble<<<av->nt, SNA_WG, 0, s>>>(v, se, *av, cap);
        }
        pe(PFK_SN_EXTEND);
    }
}

} // namespace dev
} // namespace chip
