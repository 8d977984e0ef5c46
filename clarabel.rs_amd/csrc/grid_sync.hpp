// grid_sync.hpp -- device-side synchronisation of the persistent launches (k_bundle_ir, k_gstep_*): a grid barrier with a
// reduction slot built on hierarchical arrival counters, non-blocking arrival / later wait, device-coherent loads and
// stores for values that cross workgroups inside a launch.
#pragma once
#include "dev_common.hpp"

namespace chip {
namespace dev {

namespace {

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

// (lds_barrier: dev_common.hpp)
// NaN-propagating max over the workgroup through LDS (red: one double per wave), broadcast; `bad`: this thread saw a NaN
__device__ __forceinline__ double lds_block_nanmax(double m, bool bad, double *red) {
    m = wave_max_all(m);
    const bool wbad = __ballot(bad) != 0ull;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    lds_barrier(); // (red may still be read from an earlier call)
    if (lane == 0) red[wv] = wbad ? __longlong_as_double(0x7ff8000000000000ll) : m;
    lds_barrier();
    double t = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = nanmax(t, red[i]);
    return t;
}

// one value + the epoch it belongs to as ONE 16-byte device-coherent message (k_snode_tri's unknowns, the group
// exchange of k_gstep_*): the reader polls the slot until both tags carry the epoch it waits for
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

// two messages in one round trip
__device__ __forceinline__ void msg_load2(const int *slot_a, const int *slot_b, msg_v4i &a, msg_v4i &b) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(slot_a), "v"(slot_b)
                 : "memory");
}
__device__ __forceinline__ void msg_load3(const int *slot_a, const int *slot_b, const int *slot_c, msg_v4i &a, msg_v4i &b,
                                          msg_v4i &c) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c)
                 : "v"(slot_a), "v"(slot_b), "v"(slot_c)
                 : "memory");
}
__device__ __forceinline__ bool msg_ready(const msg_v4i &m, int tag) { return m.y == tag && m.w == tag; }
__device__ __forceinline__ double msg_value(const msg_v4i &m) { return __hiloint2double(m.z, m.x); }

// Release of the grid barrier WITH its results (k_bundle_irs, round 6): the last arriver writes the few reduced numbers
// as tagged 16-byte messages (value + the barrier's tag in one store), one copy per sub-group of workgroups; a waiting
// workgroup polls the messages of its sub-group with wave 0 -- lane j owns message j -- until all carry the tag, and has
// the values at that moment.  The older form (ir_release) stored the numbers, waited for the stores to be acknowledged,
// wrote the release words, and every workgroup then loaded the numbers: two more trips through the fabric per barrier.
constexpr int IR_REC_MSGS = 32; // message slots per sub-group record (512 bytes)
constexpr int IR_REL_INTS = IR_NSUB * IR_REC_MSGS * 4;
__device__ __forceinline__ void ir_publish(int *rel, const double *vals, int nmsg, int tag, int nwg) {
    const int nsub = min(IR_NSUB, nwg);
    for (int t = threadIdx.x; t < nsub * nmsg; t += blockDim.x) {
        const int sgrp = t / nmsg, j = t - sgrp * nmsg;
        msg_store(rel + (sgrp * IR_REC_MSGS + j) * 4, vals[j], tag);
    }
}
// vals[0 .. nmsg) <- the record of this workgroup's sub-group once every message carries `tag`; IR_TIMEOUT / IR_WAITED
__device__ __forceinline__ int ir_wait_record(const int *rel, double *vals, int nmsg, int tag) {
    __shared__ int s_state4;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int *slot = rel + ((int)(blockIdx.x % IR_NSUB) * IR_REC_MSGS + (lane < nmsg ? lane : 0)) * 4;
        int state = IR_WAITED;
        long long spins = 0;
        msg_v4i m;
        for (;;) {
            m = msg_load(slot);
            if (__all(msg_ready(m, tag) ? 1 : 0)) break;
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1ll << 21)) {
                state = IR_TIMEOUT;
                break;
            }
        }
        if (state != IR_TIMEOUT && lane < nmsg) vals[lane] = msg_value(m);
        if (lane == 0) s_state4 = state;
    }
    __syncthreads();
    return s_state4;
}

} // namespace

} // namespace dev
} // namespace chip
