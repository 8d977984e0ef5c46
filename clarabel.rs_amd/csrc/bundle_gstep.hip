// bundle_gstep.hip -- the "step" kernels of a grouped fold: a forest of small elimination trees, each cut into a few
// bundles whose tree's top (at most 8 nodes) is folded into the bundle kernels (BASELINE config 4 when a GPU holds only
// a share of the trees: 128 trees = 1024 bundles of ~750 nodes).  For bundles this small a phase of k_bundle_ir is a
// chain of dependent memory round trips, not bandwidth: per launch the 1500 entries of a bundle's L were streamed four
// times and its 2000 entries of K twice, every stream a pointer -> index -> value chain of its own, and the group's
// top was solved by a last arriver behind four device-scope round trips.  Here
//   * the bundle's entries of L and of K live in REGISTERS for the whole launch (a 256-thread workgroup has 128 KB of
//     them: LR + UR entries per thread, value + packed 16-bit (row, column)), loaded once through a host-made order in
//     which the entries of one elimination level are contiguous and the entries in the rows / columns of the group's
//     top come last, sorted by top row -- the sweeps and the residual touch LDS only (the three vector slices);
//   * the group exchange is all-to-all: every bundle PUBLISHES its shares of the top rows as 16-byte tagged messages
//     and every workgroup of the group polls its mates' messages and solves the k x k top itself, redundantly, in a
//     fixed order -- one store and one polling load on the critical path;
//   * the refinement decisions (directldlkktsolver.rs:266-321) are those of k_bundle_ir: grid-wide norms through the
//     hierarchical arrival counters of grid_sync.hpp, the verdict on a round's candidate awaited in the middle of the
//     next round, whose forward sweep runs speculatively (a rejected candidate always ends the refinement).
// k_gstep_factor is the bundle factorisation of k_bundle_factor_flat with the bundle's Schur contribution to its
// group's top taken from LDS and the k x k top factored by the LAST of the group's workgroups to deposit its share --
// one launch instead of three.
// Reference semantics: qdldl.rs:469-669 (factor, pivot rule :645-665), :708-768 (solves),
// directldlkktsolver.rs:160-189, 205-215, 266-347 (setrhs / getlhs, refinement, residual against the unregularised K).
#include "dev_common.hpp"
#include "grid_sync.hpp"

namespace chip {
namespace dev {

namespace {

constexpr int GS_TW = 256;
constexpr int GS_MAXRUNS = 32;

// per-workgroup state shared through LDS (written by thread 0 / wave 0, read after a barrier)
struct GsState {
    double normb, norme, lastnorme;
    int rounds, ok, done, sel, gen, accept;
    double btop[8], dinvt[8], ltt[64], ktt[64];
    double fsum[8], rsum[8];         // the group's totals of the published shares
    double dxt[8], acct[8], candt[8], rtop[8];
    double tacc[8];                  // this bundle's shares (forward sweep / residual)
    double mtop;                     // max |top rows of the residual| (the group's first workgroup)
    int lev[GS_MAXL + 2];
    int runs[3 * GS_MAXRUNS];
    int timeout;
};

// Poll the messages of phase slot `slot` of the bundles [gb0, gb0 + gnb) until all carry `tag`; out[t], t < 8 = the sum
// over the bundles of value t, in a fixed order (every workgroup of the group computes the same bits).  Wave 0 only.
// Returns false on a timeout.
__device__ __forceinline__ bool gs_poll_sum(const int *msg, int gb0, int gnb, int k, int slot, int tag, double *out) {
    const int lane = threadIdx.x & 63, q = lane >> 3, t = lane & 7;
    double tot = 0.0;
    for (int base = 0; base < gnb; base += 8) {
        const bool valid = base + q < gnb && t < k;
        const int *addr = msg + ((size_t)((gb0 + base + (valid ? q : 0)) * 4 + slot) * 8 + t) * 4;
        msg_v4i m;
        long long spins = 0;
        for (;;) {
            m = msg_load(addr);
            const bool ready = !valid || (m.y == tag && m.w == tag);
            if (__all(ready)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1ll << 20)) return false; // (~2 s)
        }
        double val = valid ? __hiloint2double(m.z, m.x) : 0.0;
        val += __shfl_xor(val, 8, 64);
        val += __shfl_xor(val, 16, 64);
        val += __shfl_xor(val, 32, 64);
        tot += val;
    }
    if (lane < 8) out[lane] = tot;
    return true;
}

// LR / UR: entries of L / of K a thread keeps in registers; NR: nodes per thread (right-hand side, 1 / d)
template <int LR, int UR, int NR>
__global__ __launch_bounds__(GS_TW) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_gstep_solve(LdlView v, BundleView bv, IrView ir, GFoldView gf, GStepView gs) {
    constexpr int TW = GS_TW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[16];
    __shared__ GsState st;
    const int nb = bv.nb, G = gridDim.x, tid = threadIdx.x, b = blockIdx.x;
    const int s0 = bv.bundle_ptr[b], nloc = bv.bundle_ptr[b + 1] - s0;
    const int nmax = bv.max_nodes;
    double *xs = (double *)smem;         // work: right-hand side -> y -> dx, then the residual
    double *A0 = xs + nmax, *A1 = A0 + nmax; // accepted iterate / candidate (roles swap)
    double *bs = A1 + nmax, *dis = bs + nmax; // the right-hand side slice, 1 / d
    const int grp = gf.bgrp[b];
    const int gbase = grp >= 0 ? gf.ptr[grp] : 0;
    const int k = grp >= 0 ? gf.ptr[grp + 1] - gbase : 0;
    const int gb0 = grp >= 0 ? gf.bptr[grp] : 0, gnb = grp >= 0 ? gf.bptr[grp + 1] - gb0 : 0;
    const bool gfirst = grp >= 0 && b == gb0;
    const int e0 = v.Lp[s0], nE = v.Lp[s0 + nloc] - e0;
    const int ub = v.Up[s0], nU = v.Up[s0 + nloc] - ub;
    const int nl = bv.blvl_ptr[b + 1] - bv.blvl_ptr[b] - 1;
    double *pnb = ir.part, *pn = pnb + nb, *pub = pn + 2 * nb; // partial norms / published reductions (as k_bundle_ir)
    if (ir.test_drop && b == G - 1 && G > 1) return; // (tests: a launch that is not co-resident)

    // ---- the bundle's entries of L and K into registers; 1 / d; the folded top's constants ----
    unsigned lij[LR], uij[UR];
    double lv[LR], uv[UR];
    {
        int lsrc[LR], usrc[UR];
#pragma unroll
        for (int u = 0; u < LR; ++u) {
            const int p = tid + u * TW;
            lsrc[u] = p < nE ? (int)gs.lsrc[e0 + p] : -1;
            lij[u] = p < nE ? gs.lij[e0 + p] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int p = tid + u * TW;
            usrc[u] = p < nU ? (int)gs.usrc[ub + p] : -1;
            uij[u] = p < nU ? gs.uij[ub + p] : 0xFFFFFFFFu;
        }
        for (int i = tid; i < nloc; i += TW) dis[i] = v.Dinv[s0 + i];
        if (tid <= nl + 1) st.lev[tid] = gs.lptr[(size_t)b * GS_LST + tid];
        if (tid == 0) {
            st.normb = st.norme = st.lastnorme = 0.0;
            st.rounds = 0;
            st.ok = 1;
            st.done = 0;
            st.sel = 0;
            st.gen = 0;
            st.accept = 0;
            st.timeout = 0;
            if (b == 0) {
                ir.res[0] = 0; // "did not finish" until the verdict is written at the very end
                ir.res[2] = 0;
            }
        }
        if (tid >= 64 && tid < 128) {
            st.ltt[tid - 64] = 0.0;
            st.ktt[tid - 64] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < LR; ++u) lv[u] = lsrc[u] >= 0 ? v.Lx[e0 + lsrc[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < UR; ++u) uv[u] = usrc[u] >= 0 ? v.Ux[ub + usrc[u]] : 0.0;
    }
    auto topnode = [&](int i) { return gf.node[gbase + i]; };
    auto rhs_of = [&](int o) { return o < ir.n ? ir.rx[o] : (o < ir.n + ir.m ? ir.rz[o - ir.n] : 0.0); };
    __syncthreads();
    if (tid < 8) {
        const double bt = tid < k ? rhs_of(ir.perm[topnode(tid)]) : 0.0;
        st.btop[tid] = bt;
        if (tid < k && gfirst) ir.bp[topnode(tid)] = bt; // (bp holds the whole permuted right-hand side afterwards)
        st.dinvt[tid] = tid < k ? v.Dinv[topnode(tid)] : 0.0;
        st.acct[tid] = st.candt[tid] = st.dxt[tid] = st.rtop[tid] = 0.0;
    } else if (tid >= 64 && tid < 64 + k * k) {
        const int ti = (tid - 64) / k, tj = (tid - 64) % k;
        const int q = gf.tt[grp * 64 + ti * 8 + tj];
        if (q >= 0) st.ltt[ti * 8 + tj] = v.Lx[q];
    } else if (tid >= 128 && tid < 128 + k) {
        const int i = tid - 128;
        const int *sp = gf.sp + gbase;
        for (int t = sp[i]; t < sp[i + 1]; ++t) st.ktt[i * 8 + gf.scol[t]] += v.Ux[gf.sslot[t]];
    }
    // ---- setrhs (directldlkktsolver.rs:160-166): the bundle's slice of the permuted right-hand side ----
    {
        const int nruns = ir.runs ? ir.run_ptr[b + 1] - ir.run_ptr[b] : 0;
        double mx = 0.0, breg[NR];
        bool nan = false;
        if (nruns > 0 && nruns <= GS_MAXRUNS) {
            for (int q = tid; q < 3 * nruns; q += TW) st.runs[q] = ir.runs[3 * ir.run_ptr[b] + q];
            __syncthreads();
            int r = 0;
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const int i = tid + u * TW;
                double val = 0.0;
                if (i < nloc) {
                    while (i >= st.runs[3 * r] + st.runs[3 * r + 2]) ++r; // (runs ascend in the local index)
                    val = rhs_of(st.runs[3 * r + 1] + (i - st.runs[3 * r]));
                }
                breg[u] = val;
            }
        } else {
            int o[NR];
#pragma unroll
            for (int u = 0; u < NR; ++u) o[u] = tid + u * TW < nloc ? ir.perm[s0 + tid + u * TW] : -1;
#pragma unroll
            for (int u = 0; u < NR; ++u) breg[u] = o[u] >= 0 ? rhs_of(o[u]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = tid + u * TW;
            if (i < nloc) {
                xs[i] = breg[u];
                bs[i] = breg[u];
                ir.bp[s0 + i] = breg[u];
                if (breg[u] != breg[u]) nan = true;
                else mx = fmax(mx, fabs(breg[u]));
            }
        }
        mx = block_max(mx, red);
        const bool anynan = __syncthreads_or(nan);
        if (tid == 0) {
            double part = anynan ? __longlong_as_double(0x7ff8000000000000ll) : mx;
            if (gfirst)
                for (int i = 0; i < k; ++i) part = nanmax(part, fabs(st.btop[i]));
            ir_store(&pnb[b], part);
        }
    }
    __syncthreads();
    int dbgn = 0;
    auto stamp = [&]() { // diagnostics (CHIP_IR_DEBUG=2): phase boundaries of every workgroup on the 100 MHz clock
        if (ir.dbg_all && tid == 0 && dbgn < 31) {
            if (dbgn == 0)
                ir.dbg_all[(size_t)b * 32] = (long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4) |
                                             ((long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 20) << 32);
            ir.dbg_all[(size_t)b * 32 + 1 + dbgn++] = wall_clock64();
        }
    };
    stamp();
    auto bail = [&]() { // a wait that cannot complete: report the timeout (the host repeats the solve unfused)
        if (tid == 0) ir.res[2] = 1;
    };
    // ---- sweeps over the register-resident entries ----
    // (the packed indices are made opaque at the head of every phase: the compiler otherwise hoists their unpacked
    // forms and the LDS addresses derived from them out of the round loop -- two to four registers per entry instead of
    // one -- and spills)
    auto opaque_l = [&]() {
#pragma unroll
        for (int u = 0; u < LR; ++u) asm volatile("" : "+v"(lij[u]));
    };
    auto opaque_u = [&]() {
#pragma unroll
        for (int u = 0; u < UR; ++u) asm volatile("" : "+v"(uij[u]));
    };
    auto forward = [&]() { // qdldl.rs:708-719: x_i -= l_ij y_j, level by level; the top rows' shares -> st.tacc
        opaque_l();
        if (tid < 8) st.tacc[tid] = 0.0;
        if (nl == 0) __syncthreads();
        for (int l = 0; l < nl; ++l) {
            const int pa = st.lev[l], pb = st.lev[l + 1];
#pragma unroll
            for (int u = 0; u < LR; ++u) {
                const int p = tid + u * TW;
                if (p >= pa && p < pb) atomicAdd(&xs[lij[u] >> 16], -(lv[u] * xs[lij[u] & 0xFFFFu]));
            }
            __syncthreads();
        }
        const int pa = st.lev[nl], pb = st.lev[nl + 1];
#pragma unroll
        for (int u = 0; u < LR; ++u) { // (entries in the top rows, sorted by row: wave-uniform targets mostly)
            const int p = tid + u * TW;
            const bool in = p >= pa && p < pb;
            lds_scatter_add(st.tacc, in ? (int)(lij[u] >> 16) - nloc : -1, in ? lv[u] * xs[lij[u] & 0xFFFFu] : 0.0);
        }
        __syncthreads();
    };
    auto backward = [&]() { // qdldl.rs:737-752: x_j = y_j / d_j - sum_i l_ij x_i, the top rows' unknowns from st.dxt
        opaque_l();
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = tid + u * TW;
            if (i < nloc) xs[i] *= dis[i];
        }
        __syncthreads();
        {
            const int pa = st.lev[nl], pb = st.lev[nl + 1];
#pragma unroll
            for (int u = 0; u < LR; ++u) {
                const int p = tid + u * TW;
                if (p >= pa && p < pb) atomicAdd(&xs[lij[u] & 0xFFFFu], -(lv[u] * st.dxt[(lij[u] >> 16) - nloc]));
            }
        }
        __syncthreads();
        for (int l = nl - 1; l >= 0; --l) {
            const int pa = st.lev[l], pb = st.lev[l + 1];
#pragma unroll
            for (int u = 0; u < LR; ++u) {
                const int p = tid + u * TW;
                if (p >= pa && p < pb) atomicAdd(&xs[lij[u] & 0xFFFFu], -(lv[u] * xs[lij[u] >> 16]));
            }
            __syncthreads();
        }
    };
    // e = b - K x (unregularised K, csc/matrix_math.rs:178-208: every off-diagonal entry used twice) for the candidate in
    // `cnd` / st.candt; e -> xs, the shares of (K x)[top rows] -> st.tacc; returns the bundle's ||e||inf (NaN propagating)
    auto residual = [&](const double *cnd) {
        opaque_u();
        if (tid < 8) st.tacc[tid] = 0.0;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = tid + u * TW;
            if (i < nloc) xs[i] = bs[i];
        }
        __syncthreads();
        const int ntop0 = gs.untop[b]; // entries at or beyond this position lie in the top columns
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int p = tid + u * TW;
            const int i = (int)(uij[u] >> 16), j = (int)(uij[u] & 0xFFFFu);
            const bool live = p < nU, top = live && p >= ntop0;
            if (live && !top) {
                if (j == i) atomicAdd(&xs[i], -(uv[u] * cnd[i]));
                else {
                    atomicAdd(&xs[i], -(uv[u] * cnd[j]));
                    atomicAdd(&xs[j], -(uv[u] * cnd[i]));
                }
            } else if (top) {
                atomicAdd(&xs[i], -(uv[u] * st.candt[j - nloc]));
            }
            lds_scatter_add(st.tacc, top ? j - nloc : -1, top ? uv[u] * cnd[i] : 0.0);
        }
        __syncthreads();
        double m = 0.0;
        bool nan = false;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = tid + u * TW;
            if (i < nloc) {
                const double val = xs[i];
                if (val != val) nan = true;
                else m = fmax(m, fabs(val));
            }
        }
        m = block_max(m, red);
        const bool anynan = __syncthreads_or(nan);
        return anynan ? __longlong_as_double(0x7ff8000000000000ll) : m;
    };
    auto publish = [&](int slot, int tag) { // st.tacc[0 .. k) as tagged messages
        if (tid < k) msg_store(gs.msg + ((size_t)(b * 4 + slot) * 8 + tid) * 4, st.tacc[tid], tag);
    };
    // fixed-order reductions of the last arriver of a barrier, and the reference's decisions (as k_bundle_ir)
    auto reduce_norms = [&](int par, bool first) {
        double mb = 0.0, m = 0.0;
        for (int q = tid; q < nb; q += TW) {
            if (first) mb = nanmax(mb, ir_load(&pnb[q]));
            m = nanmax(m, ir_load(&pn[(size_t)par * nb + q]));
        }
        if (first) {
            mb = block_nanmax(mb, red);
            if (tid == 0) ir_store(&pub[par * 32 + 9], mb);
        }
        m = block_nanmax(m, red);
        if (tid == 0) ir_store(&pub[par * 32 + 8], m);
    };
    auto decide = [&](int round, int par) {
        if (tid == 0) {
            const double newnorm = ir_load(&pub[par * 32 + 8]);
            if (round == 0) st.normb = ir_load(&pub[par * 32 + 9]);
            const double tol = ir.abstol + ir.reltol * st.normb;
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
                st.sel ^= 1; // the candidate becomes the accepted iterate
                for (int i = 0; i < 8; ++i) st.acct[i] = st.candt[i];
            }
            if (!done && (st.rounds >= ir.maxiter || st.norme <= tol)) done = true; // :288-293
            st.lastnorme = st.norme;
            st.done = done ? 1 : 0;
        }
        __syncthreads();
    };
    // the k x k top part of both sweeps (every workgroup of the group alike): rhs - (the group's forward shares)
    auto top_solve = [&](int round) {
        if (tid == 0) {
            double y[8];
            for (int i = 0; i < k; ++i) {
                double sacc = (round == 0 ? st.btop[i] : st.rtop[i]) - st.fsum[i];
                for (int j = 0; j < i; ++j) sacc -= st.ltt[i * 8 + j] * y[j];
                y[i] = sacc;
            }
            for (int i = k - 1; i >= 0; --i) {
                double sacc = y[i] * st.dinvt[i];
                for (int j = i + 1; j < k; ++j) sacc -= st.ltt[j * 8 + i] * y[j];
                y[i] = sacc;
            }
            for (int i = 0; i < k; ++i) st.dxt[i] = y[i];
        }
    };
    // top rows of the residual of the candidate in st.candt from the group's residual shares st.rsum -> st.rtop;
    // returns max |rtop| (NaN propagating).  Thread 0.
    auto top_residual = [&]() {
        double m = 0.0;
        for (int i = 0; i < k; ++i) {
            double sacc = st.rsum[i];
            for (int c = 0; c < k; ++c) sacc += st.ktt[i * 8 + c] * st.candt[c];
            const double rt = st.btop[i] - sacc;
            st.rtop[i] = rt;
            m = nanmax(m, fabs(rt));
        }
        return m;
    };
    bool pending = false;
    for (int round = 0;; ++round) {
        const int par = round & 1;
        // tags of this launch's messages: forward shares of round r, residual shares of round r
        const int tagF = gs.epoch * 64 + 2 * round, tagR = tagF + 1, tagRprev = tagF - 1;
        forward();
        stamp();
        if (grp >= 0) {
            publish(par, tagF);
            if (tid < 64) {
                bool ok = gs_poll_sum(gs.msg, gb0, gnb, k, par, tagF, st.fsum);
                if (ok && round > 0) ok = gs_poll_sum(gs.msg, gb0, gnb, k, 2 + (par ^ 1), tagRprev, st.rsum);
                if (!ok && tid == 0) st.timeout = 1;
            }
            __syncthreads();
            if (st.timeout) return bail();
            if (tid == 0 && round > 0) (void)top_residual(); // -> st.rtop: the top rows' right-hand side of this round
            __syncthreads();
            top_solve(round);
            __syncthreads();
        }
        stamp();
        backward();
        stamp();
        if (pending) { // the verdict on the previous round's candidate (arrived for at the end of that round)
            if (ir_wait_word(ir.ctl + 32 * (1 + IR_NSUB + (b % IR_NSUB)), st.gen) == IR_TIMEOUT) return bail();
            decide(round - 1, par ^ 1);
            pending = false;
            if (__builtin_amdgcn_readfirstlane(st.done)) break; // (this round's sweeps were speculative)
        }
        stamp();
        // the candidate: x (round 0) or x + dx (directldlkktsolver.rs:300 axpby(1, x, 1))
        const int sel = __builtin_amdgcn_readfirstlane(st.sel);
        const double *acc = sel ? A1 : A0;
        double *cnd = sel ? A0 : A1;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = tid + u * TW;
            if (i < nloc) cnd[i] = round == 0 ? xs[i] : 1.0 * acc[i] + 1.0 * xs[i];
        }
        if (tid < 8) st.candt[tid] = round == 0 ? st.dxt[tid] : 1.0 * st.acct[tid] + 1.0 * st.dxt[tid];
        __syncthreads();
        double mine;
        if (!ir.ir_enable) { // no refinement: only x.is_finite() is asked for (:180)
            double mx = 0.0;
            bool nan = false;
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const int i = tid + u * TW;
                if (i < nloc) {
                    const double val = cnd[i];
                    if (val != val) nan = true;
                    else mx = fmax(mx, fabs(val));
                }
            }
            mx = block_max(mx, red);
            const bool anynan = __syncthreads_or(nan);
            mine = anynan ? __longlong_as_double(0x7ff8000000000000ll) : mx;
            if (gfirst)
                for (int i = 0; i < k; ++i) mine = nanmax(mine, fabs(st.candt[i]));
        } else {
            mine = residual(cnd);
            stamp();
            if (grp >= 0) {
                publish(2 + par, tagR);
                if (gfirst) { // the top rows of the residual belong to the group's first workgroup's partial norm
                    if (tid < 64 && !gs_poll_sum(gs.msg, gb0, gnb, k, 2 + par, tagR, st.rsum) && tid == 0) st.timeout = 1;
                    __syncthreads();
                    if (st.timeout) return bail();
                    // (st.rtop is written here and recomputed from the same numbers at the next group wait)
                    if (tid == 0) st.mtop = top_residual();
                    __syncthreads();
                    mine = nanmax(mine, st.mtop);
                }
            }
        }
        if (tid == 0) ir_store(&pn[(size_t)par * nb + b], mine);
        pending = true;
        const bool more_possible = ir.ir_enable && round < ir.maxiter;
        if (tid == 0) st.gen += 1;
        if (more_possible) {
            // arrival for the verdict on this round's candidate; it is awaited in the middle of the next round
            if (ir_arrive_nowait(ir.ctl, st.gen, G) == IR_LAST) {
                reduce_norms(par, round == 0);
                ir_release(ir.ctl, st.gen, G);
            }
            stamp();
            continue;
        }
        stamp();
        const int state = ir_arrive_wait(ir.ctl, st.gen, G);
        if (state == IR_TIMEOUT) return bail();
        if (state == IR_LAST) {
            reduce_norms(par, round == 0);
            ir_release(ir.ctl, st.gen, G);
        }
        __syncthreads();
        stamp();
        decide(round, par);
        pending = false;
        break;
    }
    stamp();
    // ---- getlhs (directldlkktsolver.rs:205-215): the accepted x, un-permuted ----
    const int ok = __builtin_amdgcn_readfirstlane(st.ok);
    if (ok) {
        const double *acc = __builtin_amdgcn_readfirstlane(st.sel) ? A1 : A0;
        auto put = [&](int o, double val) {
            if (o < ir.n) {
                if (ir.lhsx) ir.lhsx[o] = val;
            } else if (o < ir.n + ir.m) {
                if (ir.lhsz) ir.lhsz[o - ir.n] = val;
            }
        };
        const int nruns = ir.runs ? ir.run_ptr[b + 1] - ir.run_ptr[b] : 0;
        if (nruns > 0 && nruns <= GS_MAXRUNS) {
            int r = 0;
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const int i = tid + u * TW;
                if (i < nloc) {
                    while (i >= st.runs[3 * r] + st.runs[3 * r + 2]) ++r;
                    put(st.runs[3 * r + 1] + (i - st.runs[3 * r]), acc[i]);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const int i = tid + u * TW;
                if (i < nloc) put(ir.perm[s0 + i], acc[i]);
            }
        }
        if (gfirst && tid < k) put(ir.perm[topnode(tid)], st.acct[tid]);
    }
    if (b == 0 && tid == 0) {
        ir.res[0] = ok ? 1 : -1; // (0 = the kernel never got here)
        ir.res[1] = st.rounds;
        ir.res[3] = 0;
        pub[64] = st.normb;
        pub[65] = st.norme;
    }
    stamp();
    ir_grid_exit(ir.ctl, __builtin_amdgcn_readfirstlane(st.gen) + 1, G);
}

// ---------------------------------------------------------------------------
// k_gstep_factor: numeric LDL' of every bundle as k_bundle_factor_flat (entry-parallel, right-looking, the bundle's
// values in LDS, update records walked flat), then the bundle's contribution to the Schur complement of its group's top
// S[i][j] = sum over the bundle's columns c of l_ic d_c l_jc straight from LDS (the entries of a column in the top
// rows carry 16-bit row indices >= nloc), published as tagged messages; the LAST workgroup of a group to deposit its
// share (one arrival counter per group, no waiting) sums the shares in a fixed order, subtracts them from K_tt and
// factors the k x k block with the pivot rule of qdldl.rs:645-665.
// ---------------------------------------------------------------------------
constexpr int GF_TW = 512;
__global__ __launch_bounds__(GF_TW) void k_gstep_factor(LdlView v, BundleView bv, GFoldView gf, GStepView gs) {
    constexpr int TW = GF_TW;
    extern __shared__ __attribute__((aligned(16))) char ff_smem[];
    __shared__ double A[36], wsum[(GF_TW / 64) * 36];
    __shared__ int s_last;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int s0 = bv.bundle_ptr[b], s1 = bv.bundle_ptr[b + 1], nloc = s1 - s0;
    const int e0 = v.Lp[s0], nE = v.Lp[s1] - e0;
    double *Ls = (double *)ff_smem, *Ds = Ls + nE; // (contiguous: a record's target addresses either)
    const double eps = v.eps_ptr ? v.eps_ptr[0] : 0.0;
    const int *lv = bv.blvl + bv.blvl_ptr[b];
    const int nl = bv.blvl_ptr[b + 1] - bv.blvl_ptr[b] - 1;
    const int *tp = v.fu_ptr + bv.blvl_ptr[b];
    for (int q = tid; q < nE; q += TW) Ls[q] = 0.0; // (fill-in slots stay zero)
    if (tid < 36) A[tid] = 0.0;
    __syncthreads();
    {
        const int ub = v.Up[s0], ue = v.Up[s1];
        for (int u = ub + tid; u < ue; u += TW) {
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
        fu_v4 r[FU];
        auto request = [&](int base) {
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int t = base + u * TW + tid;
                if (t < re) r[u] = rec[t];
                else r[u] = fu_v4{0, 0, 0, 0xFFFF};
            }
        };
        request(rb);
        // the level's columns are final: pivot rule, scale
        for (int j = lv[l] + tid; j < lv[l + 1]; j += TW) {
            const int cb = v.Lp[j] - e0, ce = v.Lp[j + 1] - e0;
            double d = Ds[j - s0];
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
            Ds[j - s0] = d;
            for (int q = cb; q < ce; ++q) Ls[q] *= dinv;
        }
        __syncthreads();
        for (int base = rb; base < re; base += TW * FU) { // (wave-uniform bounds: lds_scatter_add is cross-lane)
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
    for (int q = tid; q < nE; q += TW) v.Lx[e0 + q] = Ls[q];
    const int grp = gf.bgrp[b];
    if (grp < 0) return;
    // ---- this bundle's share of the Schur complement of its group's top: a column's entries in the top rows are its
    // last ones (16-bit row index >= nloc); per-thread register accumulators over the packed lower triangle, reduced
    // wave by wave in a fixed order (as k_gfold_schur, with L and D read from LDS) ----
    const int gbase = gf.ptr[grp], k = gf.ptr[grp + 1] - gbase;
    const int np = k * (k + 1) / 2;
    {
        double sa[36];
#pragma unroll
        for (int p = 0; p < 36; ++p) sa[p] = 0.0;
        for (int j = s0 + tid; j < s1; j += TW) {
            const int cb = v.Lp[j] - e0, ce = v.Lp[j + 1] - e0;
            double vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) vv[i] = 0.0;
            bool any = false;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int q = ce - 1 - e;
                const int ti = q >= cb ? (int)v.Li16[e0 + q] - nloc : -1;
                if (ti >= 0) {
                    const double val = Ls[q];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (ti == i) vv[i] = val;
                    any = true;
                }
            }
            if (any) {
                const double dj = Ds[j - s0];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double wi = vv[i] * dj;
#pragma unroll
                    for (int jj = 0; jj <= i; ++jj) sa[i * (i + 1) / 2 + jj] += wi * vv[jj];
                }
            }
        }
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int p = 0; p < 36; ++p)
            if (p < np) {
                const double t = wave_sum(sa[p]);
                if (lane == 0) wsum[wv * 36 + p] = t;
            }
        __syncthreads();
        if (tid < np) {
            double t = 0.0;
            for (int w = 0; w < TW / 64; ++w) t += wsum[w * 36 + tid];
            A[tid] = t;
        }
    }
    __syncthreads();
    if (tid < np) msg_store(gs.fmsg + ((size_t)b * 36 + tid) * 4, A[tid], gs.epoch);
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_s_waitcnt(0); // (this workgroup's messages have been acknowledged before it arrives)
        const int gnb = gf.bptr[grp + 1] - gf.bptr[grp];
        const int old = atomicAdd(gf.gcnt + grp * 32 + 2, 1);
        s_last = (old + 1 == gnb) ? 1 : 0;
        if (s_last) __hip_atomic_store(gf.gcnt + grp * 32 + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    // ---- the last arriver: K_tt (scattered into D / the top-top slots of Lx by k_scatter_init, static regulariser
    // included) minus the shares in bundle order, then the k x k LDL' ----
    if (tid < 64) {
        const int lane = tid;
        if (lane < np) {
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= lane) ++i;
            const int j = lane - i * (i + 1) / 2;
            double a;
            if (i == j) a = v.D[gf.node[gbase + i]];
            else {
                const int q = gf.tt[grp * 64 + i * 8 + j];
                a = q >= 0 ? v.Lx[q] : 0.0;
            }
            for (int q = gf.bptr[grp]; q < gf.bptr[grp + 1]; ++q) {
                const int *addr = gs.fmsg + ((size_t)q * 36 + lane) * 4;
                msg_v4i m = msg_load(addr);
                long long spins = 0;
                while ((m.y != gs.epoch || m.w != gs.epoch) && ++spins < (1ll << 20)) m = msg_load(addr); // (never spins:
                // every mate's messages were acknowledged before its arrival; the bound only guards a broken launch)
                a -= __hiloint2double(m.z, m.x);
            }
            A[lane] = a;
        }
    }
    __syncthreads();
    if (tid != 0) return;
    for (int j = 0; j < k; ++j) {
        const int nj = gf.node[gbase + j];
        const double dinv = pivot_rule(v, nj, A[j * (j + 1) / 2 + j]);
        for (int i = j + 1; i < k; ++i) {
            const double aij = A[i * (i + 1) / 2 + j];
            for (int i2 = j + 1; i2 <= i; ++i2) A[i * (i + 1) / 2 + i2] -= aij * (A[i2 * (i2 + 1) / 2 + j] * dinv);
        }
        for (int i = j + 1; i < k; ++i) {
            const double lij = A[i * (i + 1) / 2 + j] * dinv;
            A[i * (i + 1) / 2 + j] = lij;
            const int q = gf.tt[grp * 64 + i * 8 + j];
            if (q >= 0) {
                v.Lx[q] = lij;
                if (v.mirror_rows) v.Rx[v.Tpos[q]] = lij;
            }
        }
    }
}

} // namespace

// ===========================================================================
// launch wrappers
// ===========================================================================
static size_t gstep_solve_lds(const BundleView &bv) { return ((size_t)5 * bv.max_nodes * sizeof(double) + 15) & ~(size_t)15; }
static size_t gstep_factor_lds(int lds_doubles) { return ((size_t)lds_doubles * sizeof(double) + 15) & ~(size_t)15; }
template <int LR, int UR, int NR> static int gstep_capacity_of(const BundleView &bv) {
    const size_t lds = gstep_solve_lds(bv);
    const void *fn = (const void *)k_gstep_solve<LR, UR, NR>;
    if (raise_dynamic_lds(fn, (size_t)lds) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    hipFuncAttributes fa;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, GS_TW, lds) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
        hipGetDeviceProperties(&prop, dev) != hipSuccess || hipFuncGetAttributes(&fa, fn) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    // cross-check with the LDS budget (static + dynamic, 1 KB allocation granularity assumed) and the wave slots
    const size_t per_wg = ((fa.sharedSizeBytes + lds + 1023) / 1024) * 1024;
    per_cu = std::min(per_cu, std::min((int)(prop.maxSharedMemoryPerMultiProcessor / per_wg), 16 / (GS_TW / 64)));
    return per_cu * prop.multiProcessorCount;
}
// the register-slot variants compiled: {LR, UR, NR}
static int gstep_variant(const GStepView &gs) {
    if (gs.lr <= 6 && gs.ur <= 8 && gs.nr <= 3) return 0;
    if (gs.lr <= 8 && gs.ur <= 10 && gs.nr <= 4) return 1;
    return -1;
}
int gstep_solve_capacity(const BundleView &bv, const GStepView &gs) {
    switch (gstep_variant(gs)) {
    case 0: return gstep_capacity_of<6, 8, 3>(bv);
    case 1: return gstep_capacity_of<8, 10, 4>(bv);
    default: return 0;
    }
}
int gstep_solve(hipStream_t s, const LdlView &v, const BundleView &bv, const IrView &ir, const GFoldView &gf,
                const GStepView &gs) {
    const size_t lds = gstep_solve_lds(bv);
    switch (gstep_variant(gs)) {
    case 0: k_gstep_solve<6, 8, 3><<<bv.nb, GS_TW, lds, s>>>(v, bv, ir, gf, gs); break;
    case 1: k_gstep_solve<8, 10, 4><<<bv.nb, GS_TW, lds, s>>>(v, bv, ir, gf, gs); break;
    default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}
bool gstep_factor_ok(int lds_doubles) {
    const size_t lds = gstep_factor_lds(lds_doubles);
    if (lds > 36 * 1024) return false; // four workgroups per CU
    if (raise_dynamic_lds((const void *)k_gstep_factor, (size_t)lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}
int gstep_factor(hipStream_t s, const LdlView &v, const BundleView &bv, const GFoldView &gf, const GStepView &gs,
                 int lds_doubles) {
    k_gstep_factor<<<bv.nb, GF_TW, gstep_factor_lds(lds_doubles), s>>>(v, bv, gf, gs);
    return (int)hipGetLastError();
}

} // namespace dev
} // namespace chip
