// bundle_symv.hpp -- e = b - K x over the rows of one bundle with K read once (device bodies shared by
// k_bundle_symv in bundle_solve.hip and the fused solve k_bundle_ir in bundle_ir.hip)
#pragma once
#include "dev_common.hpp"

namespace chip {
namespace dev {

namespace {

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

} // namespace

} // namespace dev
} // namespace chip
