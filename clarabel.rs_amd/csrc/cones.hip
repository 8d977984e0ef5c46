// cones.hip -- cone scalings fused into the KKT value update, Hs products, step operations, barriers (NN / SOC / Exp / Pow / GenPow / PSD)
// (one of the translation units behind kernels.hpp; the design rules and the reference citations are in
// dev_common.hpp)
#include <atomic>
#include "dev_common.hpp"

namespace chip {
namespace dev {

namespace {

#define wave_sum wave_sum_tree
#define wave_max wave_max_tree
#define block_sum block_sum_tree
#define block_max block_max_tree
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
template <bool KKT>
__device__ __forceinline__ bool soc_scaling_reg(const SocView &v, const double *__restrict__ sv, const double *__restrict__ zv,
                                                int c, double *red, double *Kx, unsigned long long *dslots);
__global__ __launch_bounds__(WG) void k_sym_update_scaling(SocView v, const int *__restrict__ nn_rows, int nn,
                                                           const double *__restrict__ sv,
                                                           const double *__restrict__ zv, double *w, double *lam) {
    __shared__ double red[16];
    if ((int)blockIdx.x < v.ncones) {
        // (cones that fit the registers of a workgroup: one round trip for s, z instead of six passes through memory)
        if (!soc_scaling_reg<false>(v, sv, zv, blockIdx.x, red, nullptr, nullptr)) soc_update_scaling_body(v, sv, zv, blockIdx.x, red);
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

// The same scaling for cones of dimension <= 1 + SOC_RPT * WG with s, z, w of the cone in REGISTERS: the body above walks
// s and z through memory six times behind seven workgroup reductions -- for a 1001-row cone that is a chain of ~13 memory
// round trips (30 us for config 3's thousand cones); here one round trip loads the cone, every later pass is register
// work.  Thread t holds the tail elements 1 + t + u * WG (the mapping of block_norm_tail, so its partial sums are the
// same); element 0 is everybody's.  KKT: also get_Hs negated and the sparse expansion columns (soc_write_kkt_body) from
// the registers, the K positions requested with the cone's data.  Returns false when the cone does not fit (dim too
// large: the caller takes the memory-walking bodies).
constexpr int SOC_RPT = 4;
__device__ __forceinline__ double reg_norm_tail(const double (&x)[SOC_RPT], const bool (&in)[SOC_RPT], double *red) {
    double amax = 0.0;
#pragma unroll
    for (int u = 0; u < SOC_RPT; ++u)
        if (in[u]) amax = fmax(amax, fabs(x[u]));
    amax = block_max(amax, red);
    if (amax == 0.0) return 0.0;
    double ss = 0.0;
#pragma unroll
    for (int u = 0; u < SOC_RPT; ++u)
        if (in[u]) {
            const double r = fabs(x[u]) / amax;
            ss += r * r;
        }
    ss = block_sum(ss, red);
    return amax * sqrt(ss);
}
template <bool KKT>
__device__ __forceinline__ bool soc_scaling_reg(const SocView &v, const double *__restrict__ sv, const double *__restrict__ zv,
                                                int c, double *red, double *Kx, unsigned long long *dslots) {
    const int n = v.dim[c];
    if (n > 1 + SOC_RPT * WG) return false;
    const int start = v.start[c], tid = threadIdx.x;
    const double *s = sv + start, *z = zv + start;
    double *w = v.w + start, *lam = v.lam + start;
    double sr[SOC_RPT], zr[SOC_RPT], wr[SOC_RPT];
    bool in[SOC_RPT];
    int mh[SOC_RPT], mu[SOC_RPT], mv[SOC_RPT];
    const int sidx = v.sparse_idx[c];
    const int *pmh = v.mapHs + v.hs_start[c];
    const int *pmu = (KKT && sidx >= 0) ? v.mapU + v.sp_ptr[sidx] : pmh, *pmv = (KKT && sidx >= 0) ? v.mapV + v.sp_ptr[sidx] : pmh;
#pragma unroll
    for (int u = 0; u < SOC_RPT; ++u) {
        const int i = 1 + tid + u * WG;
        in[u] = i < n;
        const int ic = in[u] ? i : 0; // (clamped, unconditional loads)
        sr[u] = s[ic];
        zr[u] = z[ic];
        if (KKT) {
            mh[u] = pmh[sidx >= 0 ? ic : 0];
            mu[u] = pmu[sidx >= 0 ? ic : 0];
            mv[u] = pmv[sidx >= 0 ? ic : 0];
        }
    }
    const double z0 = z[0], s0 = s[0];
    const double z1n = reg_norm_tail(zr, in, red);
    const double s1n = reg_norm_tail(sr, in, red);
    const double zres = (z0 - z1n) * (z0 + z1n), sres = (s0 - s1n) * (s0 + s1n);
    const double zscale = zres > 0.0 ? sqrt(zres) : 0.0;
    const double sscale = sres > 0.0 ? sqrt(sres) : 0.0;
    if (zscale == 0.0 || sscale == 0.0) {
        if (tid == 0) *v.fail = v.fail_gen;
        return true;
    }
    const double eta = sqrt(sscale / zscale);
    const double rs = 1.0 / sscale, mrz = -(1.0 / zscale);
#pragma unroll
    for (int u = 0; u < SOC_RPT; ++u) wr[u] = mrz * zr[u] + 1.0 * (sr[u] * rs);
    const double w0a = s0 * rs + z0 / zscale;
    const double w1n = reg_norm_tail(wr, in, red);
    const double wres = (w0a - w1n) * (w0a + w1n);
    const double wscale = wres > 0.0 ? sqrt(wres) : 0.0;
    if (wscale == 0.0) {
        if (tid == 0) *v.fail = v.fail_gen;
        return true;
    }
    const double rw = 1.0 / wscale;
    double sq = 0.0;
#pragma unroll
    for (int u = 0; u < SOC_RPT; ++u) {
        wr[u] = wr[u] * rw;
        if (in[u]) sq += wr[u] * wr[u];
    }
    const double w1sq = block_sum(sq, red);
    const double w0 = sqrt(1.0 + w1sq);
    // lambda, socone.rs:174-184
    const double gamma = 0.5 * wscale;
    const double ca = (gamma + z0 / zscale) / sscale, cb = (gamma + s0 / sscale) / zscale;
    const double sc = 1.0 / (s0 / sscale + z0 / zscale + 2.0 * gamma);
    const double sqz = sqrt(sscale * zscale);
    // rank-2 terms, socone.rs:187-208
    const double alpha = 2.0 * w0;
    const double wsq = w0 * w0 + w1sq;
    const double wsqinv = 1.0 / wsq;
    const double d = 0.5 * wsqinv;
    const double u0 = sqrt(wsq - d);
    const double u1 = alpha / u0;
    const double v1 = sqrt(2.0 * (2.0 + wsqinv) / (2.0 * wsq - wsqinv));
#pragma unroll
    for (int u = 0; u < SOC_RPT; ++u)
        if (in[u]) {
            const int i = 1 + tid + u * WG;
            w[i] = wr[u];
            lam[i] = ((ca * sr[u] + cb * zr[u]) * sc) * sqz;
        }
    if (tid == 0) {
        w[0] = w0;
        lam[0] = gamma * sqz;
        double *st = v.eta + 8 * c;
        st[0] = eta;
        st[1] = d;
        st[2] = u0;
        st[3] = u1;
        st[4] = v1;
    }
    if (!KKT) return true;
    // ---- get_Hs negated + the sparse expansion columns (socone.rs:217-246, datamaps.rs:199-220) ----
    const double eta2 = eta * eta;
    if (sidx >= 0) {
        if (dslots && tid == 0) {
            const double mx = fmax(fabs(eta2 * d), fabs(eta2));
            int *nanflag = (int *)(dslots + (size_t)NRM_SLOTS * NRM_STRIDE);
            fold_norm(dslots, nanflag, mx != mx ? 0.0 : mx, mx != mx, c);
        }
#pragma unroll
        for (int u = 0; u < SOC_RPT; ++u)
            if (in[u]) {
                Kx[mh[u]] = -eta2;
                Kx[mu[u]] = (u1 * wr[u]) * (-eta2);
                Kx[mv[u]] = (v1 * wr[u]) * (-eta2);
            }
        if (tid == 0) {
            Kx[pmh[0]] = -(eta2 * d);
            Kx[pmu[0]] = u0 * (-eta2);
            Kx[pmv[0]] = 0.0 * (-eta2);
            Kx[v.mapD[2 * sidx]] = -eta2;
            Kx[v.mapD[2 * sidx + 1]] = eta2;
        }
    } else {
        __syncthreads(); // (dense form, dim <= 4: thread 0 reads w back)
        soc_write_kkt_body(v, Kx, dslots, c);
    }
    return true;
}

// update_scaling and the KKT update's get_Hs scatter of the same rows in ONE launch (solver.rs:334-352 calls them back to
// back): a cone's workgroup scales it and writes its K entries while w is hot; a slab of Nonnegative rows reads s, z once
// and writes lambda, w and -w^2.  status: the refactor's 4 status words, cleared here (its preparation launch is skipped).
__global__ __launch_bounds__(WG) void k_sym_scale_write(SocView v, const int *__restrict__ nn_rows,
                                                        const int *__restrict__ nn_hsidx, int nn,
                                                        const double *__restrict__ sv, const double *__restrict__ zv,
                                                        double *w, double *lam, const int *__restrict__ mapHs, double *Kx,
                                                        unsigned long long *dslots, int *status) {
    __shared__ double red[16];
    if (status && blockIdx.x == 0 && threadIdx.x < 4) status[threadIdx.x] = 0;
    if ((int)blockIdx.x < v.ncones) {
        if (soc_scaling_reg<true>(v, sv, zv, blockIdx.x, red, Kx, dslots)) return;
        soc_update_scaling_body(v, sv, zv, blockIdx.x, red);
        __syncthreads(); // (w and the cone's state, written by this workgroup, are read back below)
        soc_write_kkt_body(v, Kx, dslots, blockIdx.x);
        return;
    }
    const int nblk = gridDim.x - v.ncones, blk = blockIdx.x - v.ncones;
    double mx = 0.0;
    bool nan = false;
    for (int t = blk * WG + threadIdx.x; t < nn; t += nblk * WG) {
        const int r = nn_rows[t];
        const double s = sv[r], z = zv[r];
        lam[r] = sqrt(s * z);
        const double wi = sqrt(s / z);
        w[r] = wi;
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
// C = op(A) op(B), all n x n column major; A / B may live in LDS or (L2-resident) global memory.
// n >= 16 (and not switched off: CHIP_NO_PSD_MFMA): 16 x 16 tiles of C on the f64 matrix cores, a wave per tile, k in
// steps of four (v_mfma_f64_16x16x4_f64: lane (l15, kq) supplies A(row l15, k kq) and B(k kq, column l15) and receives
// C(rows kq + 4 r, column l15)); rows / columns / k beyond n are clamped loads multiplied by zero.  One thread per
// element of C with an n-long dot product -- the form below -- issues 2 n loads and n multiply-adds per element: at
// n = 96 the three n x n products of an operation took most of its time.  The matrix instruction fuses its
// multiply-adds and sums k in another order than the scalar loop: results agree to rounding (the cone tests' tolerances).
__device__ int g_psd_no_mfma = 0;
template <bool TA, bool TB>
__device__ __forceinline__ void psd_gemm(double *C, const double *A, const double *B, int n) {
    if (n >= 16 && !(g_psd_no_mfma & 1)) {
        typedef double gemm_v4d __attribute__((ext_vector_type(4)));
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kq = lane >> 4;
        const int nt = (n + 15) / 16;
        for (int tile = wave; tile < nt * nt; tile += WG / 64) {
            const int i0 = 16 * (tile % nt), j0 = 16 * (tile / nt);
            const int ia = min(i0 + l15, n - 1), jb = min(j0 + l15, n - 1);
            const double am = i0 + l15 < n ? 1.0 : 0.0, bm = j0 + l15 < n ? 1.0 : 0.0;
            gemm_v4d acc = {0.0, 0.0, 0.0, 0.0};
            for (int k0 = 0; k0 < n; k0 += 4) {
                const int kk = min(k0 + kq, n - 1);
                const double km = k0 + kq < n ? 1.0 : 0.0;
                const double a = (TA ? A[kk + ia * n] : A[ia + kk * n]) * (am * km);
                const double b = (TB ? B[jb + kk * n] : B[kk + jb * n]) * bm;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + kq + 4 * r, col = j0 + l15;
                if (row < n && col < n) C[row + col * n] = acc[r];
            }
        }
        return;
    }
    for (int idx = threadIdx.x; idx < n * n; idx += WG) {
        const int i = idx % n, j = idx / n;
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc += (TA ? A[k + i * n] : A[i + k * n]) * (TB ? B[j + k * n] : B[k + j * n]);
        C[idx] = acc;
    }
}
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
    // M = L2' L1 ; V = I  (the factors' strict upper triangles still hold the symmetric input: zeroed, so that the
    // products below are plain n x n products for the matrix cores)
    for (int idx = tid; idx < n * n; idx += WG) {
        const int a = idx % n, b = idx / n;
        if (a < b) {
            A[idx] = 0.0;
            Bm[idx] = 0.0;
        }
        Vm[idx] = (a == b) ? 1.0 : 0.0;
    }
    __syncthreads();
    psd_gemm<true, false>(Cm, Bm, A, n);
    __syncthreads();
    // (round 5) cones beyond 64: M and V of the Jacobi iteration staged in LDS when the launch provided room for both --
    // ~30 sweeps x (n - 1) rounds, a barrier apart, were each a round trip to the L2-resident scratch (15 ms at n = 96)
    double *Cg = Cm, *Vg = Vm;
    // (columns padded to a leading dimension = 8 mod 32, so that the pairs of a half wave start 16 banks apart, were
    // measured too: no change at n = 96 -- the rounds are bound by their dependent chains, not by LDS bank conflicts)
    const bool staged = GS && v.jacobi_lds >= 2 * n * n;
    if (staged) {
        __threadfence_block();
        double *Cl = (double *)smem, *Vl = Cl + n * n;
        for (int idx = tid; idx < n * n; idx += WG) {
            Cl[idx] = Cm[idx];
            Vl[idx] = Vm[idx];
        }
        Cm = Cl;
        Vm = Vl;
        __syncthreads();
    }
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
                // (a butterfly of commutative adds: the eight lanes of a pair end with bit-identical sums, so they agree
                // on the rotation and on whether to rotate at all)
#pragma unroll
                for (int off = 1; off < JG; off <<= 1) {
                    al += __shfl_xor(al, off, 64);
                    be += __shfl_xor(be, off, 64);
                    ga += __shfl_xor(ga, off, 64);
                }
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
    if (staged) { // back to the scratch slice: the products below read them from there
        for (int idx = tid; idx < n * n; idx += WG) {
            Cg[idx] = Cm[idx];
            Vg[idx] = Vm[idx];
        }
        Cm = Cg;
        Vm = Vg;
        __syncthreads();
        __threadfence_block();
    }
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
    // R = L1 V Sigma^-1/2 (column p -> rank[p]);  Rinv = Sigma^-1/2 U' L2' with U = (M V) Sigma^-1: the two products
    // L1 V and L2 (M V) through Bout as scratch (it receives B = R R' last), then scaled and permuted into place
    psd_gemm<false, false>(Bout, A, Vm, n);
    __syncthreads();
    __threadfence_block(); // (Bout is global memory: written by one wave, read by another)
    for (int idx = tid; idx < n * n; idx += WG) {
        const int i = idx % n, p = idx / n;
        const double lsq = 1.0 / sqrt(sig[p]);
        Rout[i + rank[p] * n] = Bout[idx] * sgn[p] * lsq;
    }
    __syncthreads();
    psd_gemm<false, false>(Bout, Bm, Cm, n); // (L2 (M V))[i, p] = sum_k L2[i,k] Cm[k,p]
    __syncthreads();
    __threadfence_block();
    for (int idx = tid; idx < n * n; idx += WG) {
        const int i = idx % n, p = idx / n;
        const double lsq = 1.0 / sqrt(sig[p]);
        Riout[rank[p] + i * n] = (Bout[idx] / sig[p]) * sgn[p] * lsq;
    }
    __syncthreads();
    __threadfence_block();
    // B = R R' (invariant under the conventions above)
    psd_gemm<false, true>(Bout, Rout, Rout, n);
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
// Round 6: for n >= 12 the eigenvalues come from a Householder reduction to tridiagonal form (n - 2 reflections, each a
// matrix-vector product and a rank-2 update of the trailing block: 4/3 n^3 flops in all) and Sturm-count bisection on the
// tridiagonal matrix, one eigenvalue per thread -- what LAPACK's syevr does for the reference
// (psdtrianglecone.rs:437-463), instead of ~8 sweeps x (n - 1) rounds of dependent rotation steps (n = 96: 3.6 ms per
// decomposition on 256 threads; the rounds are chains of barriers, not arithmetic).  Only the VALUES are needed here
// (smallest one, sum of the positive ones).  cs: 3 n doubles of work space (w, then d and e^2).
__device__ double psd_eig_tridiag(double *A, int n, double *cs, double *red, double *psum) {
    const int tid = threadIdx.x;
    double *w = cs, *dd = cs + n, *e2 = cs + 2 * n;
    for (int k = 0; k + 2 < n; ++k) {
        const int m = n - k - 1;          // x = A[k + 1 .. n, k]: the column below the diagonal (contiguous)
        double *x = A + (k + 1) + (size_t)k * n;
        double s2 = 0.0;
        for (int i = 1 + tid; i < m; i += WG) s2 += x[i] * x[i];
        s2 = block_sum(s2, red);
        const double x0 = x[0];
        if (s2 == 0.0) continue;          // (already tridiagonal in this column; uniform: s2 is broadcast)
        const double nrm = sqrt(x0 * x0 + s2), alpha = x0 >= 0.0 ? -nrm : nrm;
        const double u0 = x0 - alpha, tau = 2.0 / (s2 + u0 * u0); // H = I - tau u u', u = x - alpha e_1
        __syncthreads();                  // (every thread has read x[0])
        if (tid == 0) x[0] = u0;
        __syncthreads();
        // p = tau A22 u (A22 = the trailing m x m block, full symmetric storage: row i is read along a column stride)
        for (int i = tid; i < m; i += WG) {
            const double *row = A + (k + 1 + i) + (size_t)(k + 1) * n;
            double acc = 0.0;
            for (int j = 0; j < m; ++j) acc += row[(size_t)j * n] * x[j];
            w[i] = tau * acc;
        }
        __syncthreads();
        double kk = 0.0;
        for (int i = tid; i < m; i += WG) kk += x[i] * w[i];
        kk = 0.5 * tau * block_sum(kk, red);
        for (int i = tid; i < m; i += WG) w[i] -= kk * x[i]; // w = p - (tau / 2)(u'p) u
        __syncthreads();
        // A22 <- A22 - u w' - w u'
        for (int idx = tid; idx < m * m; idx += WG) {
            const int i = idx % m, j = idx / m;
            A[(k + 1 + i) + (size_t)(k + 1 + j) * n] -= x[i] * w[j] + w[i] * x[j];
        }
        __syncthreads();
        if (tid == 0) x[0] = alpha;       // the subdiagonal entry e_k
        __syncthreads();
    }
    // d (diagonal) and e^2 (squared subdiagonal); Gershgorin bounds
    double lo = INFINITY, hi = -INFINITY;
    for (int i = tid; i < n; i += WG) {
        const double di = A[i + (size_t)i * n];
        const double el = i > 0 ? A[i + (size_t)(i - 1) * n] : 0.0, er = i + 1 < n ? A[(i + 1) + (size_t)i * n] : 0.0;
        dd[i] = di;
        e2[i] = el * el; // e2[i] = e_{i-1}^2 (e2[0] = 0)
        const double rad = fabs(el) + fabs(er);
        lo = fmin(lo, di - rad);
        hi = fmax(hi, di + rad);
    }
    lo = -block_max(-lo, red);
    hi = block_max(hi, red);
    __syncthreads();
    // eigenvalue number t (ascending) by bisection on the number of eigenvalues below a shift (Sturm count of the LDL'
    // recurrence q_i = d_i - s - e_{i-1}^2 / q_{i-1}; a zero pivot is replaced by a tiny one)
    double ev = 0.0;
    if (tid < n) {
        double a = lo, b = hi;
        const double scale = fmax(fabs(lo), fabs(hi));
        const double tiny = 2.2250738585072014e-308 / 2.220446049250313e-16;
        for (int it = 0; it < 120; ++it) {
            const double mid = 0.5 * (a + b);
            if (mid <= a || mid >= b || b - a <= 4.440892098500626e-16 * scale) break;
            int cnt = 0;
            double q = 1.0;
            for (int i = 0; i < n; ++i) {
                q = dd[i] - mid - (i > 0 ? e2[i] / q : 0.0);
                if (fabs(q) < tiny) q = -tiny;
                cnt += q < 0.0 ? 1 : 0;
            }
            if (cnt > tid) b = mid;
            else a = mid;
        }
        ev = 0.5 * (a + b);
    }
    double mn = tid == 0 ? ev : INFINITY; // (thread 0 holds the smallest eigenvalue)
    mn = -block_max(-mn, red);
    const double sp = block_sum(tid < n ? fmax(ev, 0.0) : 0.0, red);
    if (psum) *psum = sp;
    return mn;
}
__device__ double psd_eig_min(double *A, int n, double *cs, double *red, int *flag, double *psum) {
    if (n >= 12 && n <= WG && !(g_psd_no_mfma & 2)) return psd_eig_tridiag(A, n, cs, red, psum);
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
                    q = -1; // (odd n: the player that sits this round out -- no partner, the identity rotation)
                }
                pp[pr] = p;
                qq[pr] = q;
                cc[pr] = c;
                ss[pr] = sn;
            }
            __syncthreads();
            // A <- J' A J, both sides in ONE pass (round 5): the 2 x 2 block (p1, q1) x (p2, q2) of the pairs k1, k2 takes
            // the column rotation of k2 on its two rows, then the row rotation of k1 on its two columns -- the arithmetic
            // of a column pass followed by a row pass, element for element, in half the LDS traffic and one barrier less;
            // lanes run over k1 (p1 ascending, q1 descending: consecutive words), where the row pass read with stride n
            // (n = 96: every lane on one bank)
            {
                int k1 = tid % half, k2 = tid / half;
                const int d1 = WG % half, d2 = WG / half;
                for (int w = tid; w < half * half; w += WG) {
                    const int p1 = pp[k1], q1 = qq[k1], p2 = pp[k2], q2 = qq[k2];
                    const bool h1 = q1 >= 0, h2 = q2 >= 0; // (false: the player that sits this round out)
                    const double c1 = cc[k1], s1 = ss[k1], c2 = cc[k2], s2 = ss[k2];
                    const double app = A[p1 + p2 * n], apq = h2 ? A[p1 + q2 * n] : 0.0;
                    const double aqp = h1 ? A[q1 + p2 * n] : 0.0, aqq = (h1 && h2) ? A[q1 + q2 * n] : 0.0;
                    const double bpp = c2 * app - s2 * apq, bpq = s2 * app + c2 * apq;
                    const double bqp = c2 * aqp - s2 * aqq, bqq = s2 * aqp + c2 * aqq;
                    A[p1 + p2 * n] = c1 * bpp - s1 * bqp;
                    if (h1) A[q1 + p2 * n] = s1 * bpp + c1 * bqp;
                    if (h2) A[p1 + q2 * n] = c1 * bpq - s1 * bqq;
                    if (h1 && h2) A[q1 + q2 * n] = s1 * bpq + c1 * bqq;
                    k1 += d1;
                    k2 += d2;
                    if (k1 >= half) {
                        k1 -= half;
                        ++k2;
                    }
                }
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
        // shift = (X Y + Y X) / 2 - sc I = (Z + Z') / 2 - sc I with Z = X Y (X and Y are symmetric: Y X = (X Y)')
        psd_gemm<false, false>(Z, X, Y, n);
        __syncthreads();
        for (int idx = tid; idx < n * n; idx += WG) {
            const int i = idx % n, j = idx / n;
            if (i > j) continue;
            const double sym = 0.5 * (Z[i + j * n] + Z[j + i * n]) - (i == j ? sc : 0.0);
            Z[i + j * n] = sym;
            Z[j + i * n] = sym;
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
            double *Ye = Y, *cse = cs;
            if (GS && v.jacobi_lds >= n * n + 3 * n + 16) { // (round 5: the eigenvalue iteration on an LDS copy, see PsdView::jacobi_lds)
                __threadfence_block();
                Ye = (double *)smem;
                cse = Ye + n * n;
                for (int idx = tid; idx < n * n; idx += WG) Ye[idx] = Y[idx];
                __syncthreads();
            }
            const double g = psd_eig_min(Ye, n, cse, red, &flag, nullptr);
            if (g < 0.0) amin = fmin(amin, fmin(-(1.0 / g), sc));
            __syncthreads();
        }
        if (tid == 0) partial[c] = amin;
    } else if (OP == 5) {
        psd_svec_to_mat(X, i0 + off, n);
        __syncthreads();
        double sp;
        double *Xe = X, *cse = cs;
        if (GS && v.jacobi_lds >= n * n + 3 * n + 16) {
            __threadfence_block();
            Xe = (double *)smem;
            cse = Xe + n * n;
            for (int idx = tid; idx < n * n; idx += WG) Xe[idx] = X[idx];
            __syncthreads();
        }
        const double mn = psd_eig_min(Xe, n, cse, red, &flag, &sp);
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

// The same entries written in the order they have in the device's value store: the Hs block of a cone is one of the
// dense diagonal blocks of the top (kernels.hpp: DblkView), whose strict upper triangle sits there row by row, every
// row contiguous -- a workgroup takes rows a = x, x + gridDim.x, ... of the block, threads run along a row: coalesced
// 8-byte stores, no index array, no square roots (the svec pair (i, j) of every row comes packed from the host).
// k_psd_write_hs walks the entries in the CALLER's order and scatters them through mapHs: 1.6e8 uncoalesced stores
// per update on config 5 (2.2-2.6 ms; this form: see DESIGN 4.1).
// Lx / l0 (or nullptr): the same values into the rows' places in L (Engine::dblk_l0: contiguous there as well) -- the
// refactor then does not read them back from K
__global__ __launch_bounds__(WG) void k_psd_write_hs_rows(PsdView v, double *Kx, double *Lx, const int *__restrict__ l0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y, c = v.blk_cone[b];
    if (c < 0) return;
    const int n = v.dim[c], m = v.blk_m[b], rb = v.blk_rowbase[b];
    double *Bl = (double *)smem;
    int *ijl = (int *)(Bl + n * n);
    const double *Bin = v.state + v.state_off[c];
    for (int idx = threadIdx.x; idx < n * n; idx += WG) Bl[idx] = Bin[idx];
    for (int a = threadIdx.x; a < m; a += WG) ijl[a] = v.row_ij[rb + a];
    __syncthreads();
    const double sqrt2 = 1.4142135623730951;
    auto entry = [&](int i, int j, int k, int l) {
        const double Ajl = Bl[j + l * n], Ajk = Bl[j + k * n];
        if (i != j && k != l) return Bl[i + k * n] * Ajl + Bl[i + l * n] * Ajk;
        if (i == j && k != l) return sqrt2 * Ajl * Ajk;
        if (i != j && k == l) return sqrt2 * Bl[i + l * n] * Ajk;
        return Ajl * Ajl;
    };
    for (int a = blockIdx.x; a < m; a += gridDim.x) {
        const int ij = ijl[a], i = ij & 0xffff, j = ij >> 16;
        const int st = v.blk_start[rb + a]; // first entry right of the diagonal; the diagonal itself sits just before
        if (threadIdx.x == 0) Kx[st - 1] = -entry(i, j, i, j);
        const int ls = Lx ? l0[rb + a] : 0;
        for (int bb = a + 1 + threadIdx.x; bb < m; bb += WG) {
            const int kl = ijl[bb];
            const double val = -entry(i, j, kl & 0xffff, kl >> 16);
            Kx[st + bb - a - 1] = val;
            if (Lx) Lx[ls + bb - a - 1] = val;
        }
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
static int nn_blocks(int count) { return count ? std::min((count + WG - 1) / WG, 2048) : 0; }
void sym_update_scaling(hipStream_t s, const SocView &v, const int *nn_rows, int nn, const double *sv,
                        const double *zv, double *w, double *lam) {
    const int grid = v.ncones + nn_blocks(nn);
    if (grid) k_sym_update_scaling<<<grid, WG, 0, s>>>(v, nn_rows, nn, sv, zv, w, lam);
}
void sym_scale_write(hipStream_t s, const SocView &v, const int *nn_rows, const int *nn_hsidx, int nn, const double *sv,
                     const double *zv, double *w, double *lam, const int *mapHs, double *Kx, unsigned long long *dslots,
                     int *status_or_null) {
    const int grid = v.ncones + nn_blocks(nn);
    if (grid) k_sym_scale_write<<<grid, WG, 0, s>>>(v, nn_rows, nn_hsidx, nn, sv, zv, w, lam, mapHs, Kx, dslots, status_or_null);
}
void sym_write_kkt(hipStream_t s, const SocView &v, const int *nn_rows, const int *nn_hsidx, int nn, const double *w,
                   const int *mapHs, double *Kx, unsigned long long *dslots) {
    const int grid = v.ncones + nn_blocks(nn);
    if (grid) k_sym_write_kkt<<<grid, WG, 0, s>>>(v, nn_rows, nn_hsidx, nn, w, mapHs, Kx, dslots);
}
// CHIP_NO_PSD_MFMA -> the device-side flag psd_gemm reads (set when it changes; the launches that follow on `s` see it)
constexpr size_t PSD_JACOBI_LDS_MAX = 158 * 1024; // dynamic LDS a launch may ask for beside the kernels' small static arrays (160 KiB per CU)
static void psd_sync_switch(hipStream_t s) {
    // g_psd_no_mfma is a per-DEVICE symbol: the cached state is kept per device (one process may drive several GPUs)
    static std::atomic<int> cur[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 ? 0 : dev % 64;
    const int want = switches().no_psd_mfma ? 1 : 0; // (bit 1 of g_psd_no_mfma: psd_eig_min keeps the Jacobi iteration -- measured, not selectable any more)
    if (want != cur[dev].load(std::memory_order_acquire)) {
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_psd_no_mfma), &want, sizeof(int), 0, hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);
        cur[dev].store(want, std::memory_order_release);
    }
}
void psd_update_scaling(hipStream_t s, const PsdView &v, const double *sv, const double *zv) {
    if (!v.ncones) return;
    psd_sync_switch(s);
    if (v.scratch) { // cones too large for LDS: work matrices in HBM scratch, the Jacobi iteration's two in LDS when they fit
        PsdView vs = v;
        const size_t need = (size_t)2 * v.maxdim * v.maxdim * sizeof(double);
        size_t lds = 0;
        if (need <= PSD_JACOBI_LDS_MAX && raise_dynamic_lds((const void *)k_psd_update_scaling<true>, need) == hipSuccess) lds = need;
        vs.jacobi_lds = (int)(lds / sizeof(double));
        k_psd_update_scaling<true><<<v.ncones, WG, lds, s>>>(vs, sv, zv);
        return;
    }
    const size_t lds = ((size_t)(4 * v.maxdim * v.maxdim + 3 * v.maxdim) * sizeof(double) + 15) & ~(size_t)15;
    if (lds > 64 * 1024) (void)raise_dynamic_lds((const void *)k_psd_update_scaling<false>, (size_t)lds);
    k_psd_update_scaling<false><<<v.ncones, WG, lds, s>>>(v, sv, zv);
}
bool psd_write_hs_rows_active(const PsdView &v) { return v.ncones > 0 && !v.scratch && v.rows_nblk > 0 && !switches().no_psd_rows; }
void psd_write_hs(hipStream_t s, const PsdView &v, double *Kx, double *Lx, const int *l0) {
    if (!v.ncones) return;
    if (v.scratch) {
        const int bpc = 64; // numel^2 / 2 entries per cone: 3.4e7 at n = 128
        k_psd_write_hs<true><<<v.ncones * bpc, WG, 0, s>>>(v, Kx, bpc);
        return;
    }
    const int bpc = 16;
    if (v.rows_nblk > 0 && !switches().no_psd_rows) { // every cone's block is a dense block of the top: row by row of the value store
        const int numel = v.maxdim * (v.maxdim + 1) / 2;
        const size_t lds2 = ((size_t)(v.maxdim * v.maxdim) * sizeof(double) + (size_t)numel * sizeof(int) + 15) & ~(size_t)15;
        if (lds2 > 64 * 1024) (void)raise_dynamic_lds((const void *)k_psd_write_hs_rows, lds2);
        k_psd_write_hs_rows<<<dim3(bpc, v.rows_nblk), WG, lds2, s>>>(v, Kx, Lx, l0);
        return;
    }
    const size_t lds = ((size_t)(v.maxdim * v.maxdim) * sizeof(double) + 15) & ~(size_t)15;
    k_psd_write_hs<false><<<v.ncones * bpc, WG, lds, s>>>(v, Kx, bpc);
}
static size_t psd_ops_lds(const PsdView &v) {
    return ((size_t)(3 * v.maxdim * v.maxdim + 4 * v.maxdim + 8) * sizeof(double) + 15) & ~(size_t)15;
}
template <typename K> static void psd_allow_lds(K kernel, size_t lds) {
    if (lds > 64 * 1024) (void)raise_dynamic_lds((const void *)kernel, (size_t)lds);
}
#define PSD_LAUNCH(OP, ...)                                                          \
    do {                                                                             \
        psd_sync_switch(s);                                                          \
        if (v.scratch) {                                                             \
            PsdView vs_ = v;                                                         \
            size_t jl_ = 0;                                                          \
            if (OP == 4 || OP == 5) { /* the eigenvalue iterations on an LDS copy */   \
                const size_t need_ = ((size_t)v.maxdim * v.maxdim + 3 * (size_t)v.maxdim + 16) * sizeof(double); \
                if (need_ <= PSD_JACOBI_LDS_MAX && raise_dynamic_lds((const void *)k_psd_ops<OP, true>, need_) == hipSuccess) jl_ = need_; \
            }                                                                        \
            vs_.jacobi_lds = (int)(jl_ / sizeof(double));                            \
            k_psd_ops<OP, true><<<v.ncones, WG, jl_, s>>>(vs_, __VA_ARGS__);         \
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
