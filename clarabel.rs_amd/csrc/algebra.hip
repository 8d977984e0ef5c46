// algebra.hip -- value plumbing (K values -> factor storage, static regulariser), vector algebra, dot products, norms
// (one of the translation units behind kernels.hpp; the design rules and the reference citations are in
// dev_common.hpp)
#include "dev_common.hpp"

namespace chip {
namespace dev {

namespace {

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
// the same scatter for a LIST of K's top entries (rest[q] indexes Kx / a2l): what is left when the dense diagonal blocks'
// strict triangles have been written into L by the kernel that produced them (Engine::hs_direct_begin)
__global__ __launch_bounds__(WG) void k_scatter_rest(const double *__restrict__ Kx, const int *__restrict__ a2l,
                                                     const int *__restrict__ rest, int nrest, int nnzL, double *Lx, double *D,
                                                     const int8_t *__restrict__ dsigns, const double *eps_ptr, int *status) {
    const double eps = eps_ptr ? eps_ptr[0] : 0.0;
    if (blockIdx.x == 0 && threadIdx.x < 4) status[threadIdx.x] = 0;
    for (int q = logical_block() * WG + threadIdx.x; q < nrest; q += gridDim.x * WG) {
        const int t = rest[q], tgt = a2l[t];
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
// ---------------------------------------------------------------------------
// Dense diagonal blocks of the top in the residual e = b - K x (kernels.hpp: DblkView).  Block b is symmetric m x m (its
// i-th row and column belong to node rownode[rowbase + i]) with only its strict upper triangle stored, row a
// contiguous: entry (a, i), i > a, at start[a] + i - a - 1.  Workgroup
// (s, b) owns the rows a = s, s + split, ... ("columns" below: by symmetry row a of the upper triangle is column a of the
// lower one) and reads each of their entries ONCE: entry h = H(i, a) serves y_i += h x_a (a register of the lane that
// owns row i) and y_a += h x_i (a wave reduction per column).  Rows go in tiles of 512 -- lane l of every wave holds rows
// T0 + 64 k + l, k < 8, and their x in registers --, a wave walks its columns through the tile with eight loads in
// flight per lane, and at the end of the tile the eight waves' row sums are added in a fixed order.  No atomics: the
// sum of every y_i has the same order in every run.  P[(rowbase + i) * split + s] = this workgroup's share of y_i.
// ---------------------------------------------------------------------------
constexpr int DB_WG = 512, DB_RT = 8, DB_TILE = 64 * DB_RT;
// NR = 2 (round 6): the residuals of the two solves of a pair in ONE pass over the blocks' entries -- every entry is read
// once and serves both vectors; per vector the same operations in the same order as the one-vector launch (its sums are
// bitwise the same).  Vector 1's partial sums go to P1 (the second solve context's buffer).
template <int NR>
__global__ __launch_bounds__(DB_WG) void k_dblk_symv(DblkView d, const double *__restrict__ Kx, const double *__restrict__ x0,
                                                      const double *__restrict__ x1, double *__restrict__ P1, int mpad) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    double *xs = (double *)dsm, *yl = xs + NR * mpad, *part = yl + NR * mpad; // xs[k][i], yl[k][i]
    int *sts = (int *)(part + (DB_WG / 64) * DB_TILE);
    const int b = (int)blockIdx.y, s = (int)blockIdx.x, split = (int)gridDim.x;
    const int m = d.m[b], rb = d.rowbase[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < m; i += DB_WG) {
        const int node = d.rownode[rb + i];
        xs[i] = x0[node];
        yl[i] = 0.0;
        if (NR > 1) {
            xs[mpad + i] = x1[node];
            yl[mpad + i] = 0.0;
        }
        sts[i] = d.start[rb + i] - i - 1; // entry (i', i) of column i at sts[i] + i'
    }
    __syncthreads();
    for (int T0 = 0; T0 < m; T0 += DB_TILE) {
        double xr[NR][DB_RT], acc[NR][DB_RT];
#pragma unroll
        for (int v = 0; v < NR; ++v)
#pragma unroll
            for (int k = 0; k < DB_RT; ++k) {
                const int i = T0 + 64 * k + lane;
                xr[v][k] = i < m ? xs[v * mpad + i] : 0.0;
                acc[v][k] = 0.0;
            }
        const int aend = min(m - 1, T0 + DB_TILE - 1); // columns with a row in this tile (column a has the rows a + 1 .. m - 1)
        for (int a = s + split * wave; a < aend; a += split * (DB_WG / 64)) {
            const int base = sts[a];
            double xa[NR];
#pragma unroll
            for (int v = 0; v < NR; ++v) xa[v] = xs[v * mpad + a];
            double h[DB_RT];
#pragma unroll
            for (int k = 0; k < DB_RT; ++k) {
                const int i = T0 + 64 * k + lane;
                const bool ok = i > a && i < m;
                h[k] = Kx[base + (ok ? i : a + 1)]; // (clamped: unconditional loads)
                h[k] = ok ? h[k] : 0.0;
            }
#pragma unroll
            for (int v = 0; v < NR; ++v) {
                double dsum = 0.0;
#pragma unroll
                for (int k = 0; k < DB_RT; ++k) {
                    acc[v][k] += h[k] * xa[v];
                    dsum += h[k] * xr[v][k];
                }
                dsum = wave_sum(dsum);
                if (lane == 0) yl[v * mpad + a] += dsum; // (column a belongs to this wave alone)
            }
        }
#pragma unroll
        for (int v = 0; v < NR; ++v) { // (one vector after the other through the same buffer)
#pragma unroll
            for (int k = 0; k < DB_RT; ++k) part[wave * DB_TILE + 64 * k + lane] = acc[v][k];
            __syncthreads();
            if (T0 + tid < m) {
                double r = 0.0;
#pragma unroll
                for (int w = 0; w < DB_WG / 64; ++w) r += part[w * DB_TILE + tid];
                yl[v * mpad + T0 + tid] += r;
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < m; i += DB_WG) {
        d.P[(size_t)(rb + i) * split + s] = yl[i];
        if (NR > 1) P1[(size_t)(rb + i) * split + s] = yl[mpad + i];
    }
}
__global__ __launch_bounds__(WG) void k_dblk_finish(DblkView d, double *__restrict__ bt) {
    const int j = blockIdx.x * WG + threadIdx.x;
    if (j >= d.nrows) return;
    double r = 0.0;
    for (int s = 0; s < d.split; ++s) r += d.P[(size_t)j * d.split + s];
    bt[d.rownode[j]] -= r;
}
__global__ __launch_bounds__(WG) void k_scatter_values(double *Kx, const int *__restrict__ map,
                                                       const double *__restrict__ vals, int k,
                                                       double scale) {
    for (int t = blockIdx.x * WG + threadIdx.x; t < k; t += gridDim.x * WG) Kx[map[t]] = vals[t] * scale;
}
// Kx[map[t]] += sign[t] * offset (sign 0: untouched; signs == nullptr: +offset) / Kx[map[t]] *= scale -- the L1 boundary's
// offset_values / scale_values on a registered index set (capi.cpp: chip_ldl_*_values_id)
__global__ __launch_bounds__(WG) void k_offset_values(double *Kx, const int *__restrict__ map, const int8_t *__restrict__ signs,
                                                      int k, double offset) {
    for (int t = blockIdx.x * WG + threadIdx.x; t < k; t += gridDim.x * WG) {
        const int sg = signs ? (int)signs[t] : 1;
        if (sg > 0) Kx[map[t]] += offset;
        else if (sg < 0) Kx[map[t]] -= offset;
    }
}
__global__ __launch_bounds__(WG) void k_scale_values(double *Kx, const int *__restrict__ map, int k, double scale) {
    for (int t = blockIdx.x * WG + threadIdx.x; t < k; t += gridDim.x * WG) Kx[map[t]] *= scale;
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

#undef wave_sum
#undef wave_max
#undef block_sum
#undef block_max

__global__ void k_debug_spin(long long ticks) {
    extern __shared__ char spin_lds[];
    if (ticks < 0) spin_lds[threadIdx.x] = 0; // (keeps the dynamic LDS allocation alive)
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

} // namespace

void scatter_init(hipStream_t s, const double *Kx, const int *a2l, int nnzK, int nnzL, double *Lx,
                  double *D, const int8_t *dsigns, const double *eps, const int *fill_idx, int nfill,
                  int *status) {
    int nb = grid_for(nnzK > 0 ? nnzK : 1);
    if (nb > 4096) nb = 4096;
    k_scatter_init<<<nb, WG, 0, s>>>(Kx, a2l, nnzK, nnzL, Lx, D, dsigns, eps, fill_idx, nfill, status);
}
void scatter_rest(hipStream_t s, const double *Kx, const int *a2l, const int *rest, int nrest, int nnzL, double *Lx, double *D,
                  const int8_t *dsigns, const double *eps, int *status) {
    int nb = grid_for(nrest > 0 ? nrest : 1);
    if (nb > 4096) nb = 4096;
    k_scatter_rest<<<nb, WG, 0, s>>>(Kx, a2l, rest, nrest, nnzL, Lx, D, dsigns, eps, status);
}
void gather_values(hipStream_t s, double *Sx, const double *Kx, const int *Smap, int nnzS) {
    if (nnzS == 0) return;
    int nb = grid_for(nnzS);
    if (nb > 4096) nb = 4096;
    k_gather_values<<<nb, WG, 0, s>>>(Sx, Kx, Smap, nnzS);
}
static size_t dblk_lds_bytes(int mpad, int nr) { return (size_t)(2 * nr * mpad + (DB_WG / 64) * DB_TILE) * sizeof(double) + (size_t)mpad * sizeof(int); }
// bit 0 of the result: the two-vector form fits the LDS as well (dblk_symv2); < 0: error
int dblk_attributes(int mmax) {
    const int mpad = (mmax + 63) / 64 * 64;
    if (raise_dynamic_lds((const void *)k_dblk_symv<1>, dblk_lds_bytes(mpad, 1)) != hipSuccess) return -1;
    if (dblk_lds_bytes(mpad, 2) > (size_t)160 * 1024 || raise_dynamic_lds((const void *)k_dblk_symv<2>, dblk_lds_bytes(mpad, 2)) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return 1;
}
void dblk_symv(hipStream_t s, const DblkView &d, const double *Kx, const double *x, double *bt) {
    if (!d.nblk) return;
    const int mpad = (d.mmax + 63) / 64 * 64;
    k_dblk_symv<1><<<dim3(d.split, d.nblk), DB_WG, dblk_lds_bytes(mpad, 1), s>>>(d, Kx, x, x, nullptr, mpad);
    k_dblk_finish<<<(d.nrows + WG - 1) / WG, WG, 0, s>>>(d, bt);
}
// two vectors, one pass: the products of both (d.P / P1); bt of the FIRST vector is finished here, the second one's by
// dblk_finish on its own stream once this launch is done
void dblk_symv2(hipStream_t s, const DblkView &d, const double *Kx, const double *x0, const double *x1, double *P1, double *bt0) {
    if (!d.nblk) return;
    const int mpad = (d.mmax + 63) / 64 * 64;
    k_dblk_symv<2><<<dim3(d.split, d.nblk), DB_WG, dblk_lds_bytes(mpad, 2), s>>>(d, Kx, x0, x1, P1, mpad);
    k_dblk_finish<<<(d.nrows + WG - 1) / WG, WG, 0, s>>>(d, bt0);
}
void dblk_finish(hipStream_t s, const DblkView &d, double *bt) {
    if (!d.nblk) return;
    k_dblk_finish<<<(d.nrows + WG - 1) / WG, WG, 0, s>>>(d, bt);
}
void scatter_values(hipStream_t s, double *Kx, const int *map, const double *vals, int k, double scale) {
    if (k == 0) return;
    int nb = (k + WG - 1) / WG;
    if (nb > 4096) nb = 4096;
    k_scatter_values<<<nb, WG, 0, s>>>(Kx, map, vals, k, scale);
}
void offset_values(hipStream_t s, double *Kx, const int *map, const int8_t *signs, int k, double offset) {
    if (k == 0) return;
    k_offset_values<<<std::min((k + WG - 1) / WG, 4096), WG, 0, s>>>(Kx, map, signs, k, offset);
}
void scale_values(hipStream_t s, double *Kx, const int *map, int k, double scale) {
    if (k == 0) return;
    k_scale_values<<<std::min((k + WG - 1) / WG, 4096), WG, 0, s>>>(Kx, map, k, scale);
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

void debug_spin(hipStream_t s, int blocks, int threads, int lds_bytes, double usec) {
    if (lds_bytes > 65536)
        (void)raise_dynamic_lds((const void *)k_debug_spin, (size_t)lds_bytes);
    k_debug_spin<<<blocks, threads, (size_t)lds_bytes, s>>>((long long)(usec * 100.0)); // 100 MHz clock
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

} // namespace dev
} // namespace chip
