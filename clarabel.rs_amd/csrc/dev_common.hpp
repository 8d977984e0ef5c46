// dev_common.hpp -- device-side helpers shared by the kernel translation units (wave / workgroup reductions,
// the pivot rule, LDS scatter-add, slotted norms) and the design notes of the kernel set.
//
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
#pragma once
#include "kernels.hpp"
#include "switches.hpp"
#include <hip/hip_ext.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>
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
inline int stream_grid(int N) {
    int nb = grid_for(N);
    return nb > 2048 ? 2048 : nb;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel, not of a handle: it is only ever RAISED
// (a later handle with smaller bundles must not lower the limit under an earlier handle's launches)
inline hipError_t raise_dynamic_lds(const void *fn, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> cur;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    size_t &c = cur[{dev, fn}];
    if (bytes <= c) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) c = bytes;
    else (void)hipGetLastError();
    return e;
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

// the static regulariser of this refactor (directldlkktsolver.rs:324-329): the device scalar of the preparation launch,
// or -- fast preparation -- reduced by every wavefront itself from the 64 slotted maxima (+ NaN flag) of |diag K|
// (fast preparation: wave 0 alone reads the slots -- a thousand workgroups x all their waves on the same 64 lines is a hot
// spot -- and hands eps to the others through *share, an LDS word; the caller's next __syncthreads() publishes it: use
// static_eps_get() after that barrier)
__device__ __forceinline__ double static_eps(const LdlView &v, bool *on, double *share) {
    if (v.eps_slots) {
        *on = true;
        if (threadIdx.x >= 64) return 0.0;
        const int lane = threadIdx.x & 63;
        // (plain loads: the slots were written by an earlier launch on the stream)
        double m = lane < NRM_SLOTS ? __longlong_as_double((long long)v.eps_slots[(size_t)lane * NRM_STRIDE]) : 0.0;
        const int nanflag = *(const int *)(v.eps_slots + (size_t)NRM_SLOTS * NRM_STRIDE);
        m = wave_max_all(m);
        m = fmax(m, v.eps_static_max);
        if (nanflag || v.eps_static_max != v.eps_static_max) m = __longlong_as_double(0x7ff8000000000000ll);
        const double eps = v.eps_c + v.eps_prop * m;
        if (lane == 0) *share = eps;
        return eps;
    }
    *on = v.eps_ptr != nullptr;
    const double eps = v.eps_ptr ? v.eps_ptr[0] : 0.0;
    if (threadIdx.x == 0) *share = eps;
    return eps;
}
// workgroup 0 of the bundle factorisation, fast preparation: publish eps, clear the other set of slots
__device__ __forceinline__ void static_eps_epilogue(const LdlView &v, double eps) {
    if (!v.eps_slots || blockIdx.x != 0) return;
    if (threadIdx.x == 0 && v.eps_out) *v.eps_out = eps;
    if (v.eps_clear && threadIdx.x <= NRM_SLOTS) v.eps_clear[(size_t)threadIdx.x * NRM_STRIDE] = 0ull;
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

// Barrier for data exchanged through LDS only: waits for this wave's LDS operations (lgkmcnt), NOT for its outstanding
// global loads and stores -- __syncthreads() is a workgroup-scope fence + barrier and drains both (s_waitcnt vmcnt(0)),
// which pulls the wait for operands requested early and needed late to the next barrier: a "prefetch across the level
// barrier" behind __syncthreads() is no prefetch at all (round 6: measured on the bundle factorisation, whose record
// loads were meant to fly under the pivots of a level and instead held every barrier of the level loop).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// bundle kernels: thin / fat thresholds shared by the factorisation and the substitutions
constexpr int FAC_THIN_ROW = 8, FAC_THIN_COL = 48, THIN_MAX = 32;
constexpr int BWG = 512;      // bundle workgroup: 8 waves -> more loads in flight per subtree
constexpr int FATCAP = 1024;  // per-level list of rows/columns that need cooperative handling

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

__device__ __forceinline__ double nanmax(double a, double b) { return (a != a || b != b) ? (a != a ? a : b) : fmax(a, b); }

} // namespace

} // namespace dev
} // namespace chip
