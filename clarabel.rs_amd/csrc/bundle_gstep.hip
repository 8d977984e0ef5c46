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

// per-workgroup state shared through LDS (written by thread 0 / wave 0, read after a barrier)
struct GsState {
    double normb, norme, lastnorme;
    int rounds, ok, done, sel, gen, accept;
    double btop[8], dinvt[8], ltt[64], ktt[64];
    double fsum[8], rsum[8];         // the group's totals of the published shares
    double dxt[8], acct[8], candt[8], rtop[8];
    double tacc[8];                  // this bundle's shares (forward sweep / residual)
    double gnorm[2];                 // max over the group's bundles of ||e||inf / ||b||inf of their rows
    double vn, vb;                   // the verdict's inputs: ||e||inf, ||b||inf over the whole system
    int lev[GS_MAXL + 2];
    int tnode[8], torig[8];          // node / original index of the group's top rows
    int timeout;
};

// Poll the messages of the bundles [gb0, gb0 + gnb) until all carry their tag, one round trip per poll for everything:
//   FWD:   the forward shares in phase slot slot_f (tag_f): out_f[t], t < 8 = their sum over the bundles;
//   RES:   the residual shares and norms in phase slot slot_r (tag_r): out_r[t] = the sum of share t, out_max[0 / 1] = the
//          NaN-propagating max of values 8 / 9 (a bundle's ||e||inf and ||b||inf);
// sums in a fixed order (every workgroup of the group computes the same bits).  Wave 0 only; lane = (bundle q < 8,
// value t < 8).  Returns false on a timeout.
template <bool FWD, bool RES>
__device__ __forceinline__ bool gs_poll_group(const int *msg, int gb0, int gnb, int k, int slot_f, int tag_f, int slot_r,
                                              int tag_r, double *out_f, double *out_r, double *out_max) {
    const int lane = threadIdx.x & 63, q = lane >> 3, t = lane & 7;
    double totf = 0.0, totr = 0.0, mx = 0.0;
    for (int base = 0; base < gnb; base += 8) {
        const bool mate = base + q < gnb;
        const bool vs = mate && t < k, vn = RES && mate && t < 2;
        const int *recf = msg + (size_t)((gb0 + base + (mate ? q : 0)) * 4 + slot_f) * GS_MV * 4;
        const int *recr = msg + (size_t)((gb0 + base + (mate ? q : 0)) * 4 + slot_r) * GS_MV * 4;
        msg_v4i mf, mr, mn;
        long long spins = 0;
        for (;;) {
            if (FWD && RES) msg_load3(recf + t * 4, recr + t * 4, recr + (8 + (t & 1)) * 4, mf, mr, mn);
            else if (RES) msg_load2(recr + t * 4, recr + (8 + (t & 1)) * 4, mr, mn);
            else mf = msg_load(recf + t * 4);
            bool ready = true;
            if (FWD) ready = ready && (!vs || msg_ready(mf, tag_f));
            if (RES) ready = ready && (!vs || msg_ready(mr, tag_r)) && (!vn || msg_ready(mn, tag_r));
            if (__all(ready)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1ll << 20)) return false; // (~2 s)
        }
        auto gsum = [&](double val) {
            val += __shfl_xor(val, 8, 64);
            val += __shfl_xor(val, 16, 64);
            val += __shfl_xor(val, 32, 64);
            return val;
        };
        if (FWD) totf += gsum(vs ? msg_value(mf) : 0.0);
        if (RES) {
            totr += gsum(vs ? msg_value(mr) : 0.0);
            double nv = vn ? msg_value(mn) : 0.0;
            nv = nanmax(nv, __shfl_xor(nv, 8, 64));
            nv = nanmax(nv, __shfl_xor(nv, 16, 64));
            nv = nanmax(nv, __shfl_xor(nv, 32, 64));
            mx = nanmax(mx, nv);
        }
    }
    if (FWD && lane < 8) out_f[lane] = totf;
    if (RES && lane < 8) out_r[lane] = totr;
    if (RES && lane < 2) out_max[lane] = mx;
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
    const int G = gridDim.x, tid = threadIdx.x, b = blockIdx.x;
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
    // ---- stage A: the bundle's descriptor (one 128-byte line: no chain of dependent pointer loads) ----
    const int *dsc = gs.desc + (size_t)b * GS_DESC;
    const int s0 = dsc[0], nloc = dsc[1], e0 = dsc[2], nE = dsc[3], ub = dsc[4], nU = dsc[5], nl = dsc[6], grp = dsc[7];
    const int k = dsc[9], gb0 = dsc[10], gnb = dsc[11], ntop0 = dsc[12], lead = dsc[14], nlead = dsc[15];
    const bool gfirst = b == gb0; // the group's leader (a bundle without group is a group of one)
    const int nmax = bv.max_nodes;
    double *xs = (double *)smem;             // work: right-hand side -> y -> dx, then the residual
    double *A0 = xs + nmax, *A1 = A0 + nmax; // accepted iterate / candidate (roles swap)
    double *bs = A1 + nmax, *dis = bs + nmax; // the right-hand side slice, 1 / d
    int *og = (int *)(dis + nmax);            // original index of every node (setrhs / getlhs)
    if (ir.test_drop && b == G - 1 && G > 1) return; // (tests: a launch that is not co-resident)

    // ---- stage B: every index the launch needs (one round trip, all loads independent) ----
    // (every load is UNCONDITIONAL with a clamped index and its result masked afterwards: a predicated load compiles to
    // a branch with a wait of its own, which would turn the round trip into a chain of them; the host pads the tables)
    unsigned lij[LR], uij[UR];
    double lv[LR], uv[UR];
    double normb_mine = 0.0; // ||b||inf over this bundle's rows
    {
        // (issue order = order of first use: the right-hand side and the top's tables, then L; the entries of K are
        // requested behind the first forward sweep -- their 28 KB per bundle would only delay everybody's L, and they are
        // not needed before the first residual: the requests travel in the shadow of the group exchange)
        int lsrc[LR], orig[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) orig[u] = ir.perm[s0 + min(tid + u * TW, nloc - 1)];
        // the folded top's tables (per group, host made): lane t < 8: node / original index of top row t; lanes
        // 64..127: CSC slot of L(top_i, top_j); lanes 128..191: position in V of K(top_i, top_c)
        const int *gt = gs.gtop + (size_t)(grp < 0 ? 0 : grp) * GS_GTOP;
        const bool topt = grp >= 0 && tid < k, toptt = grp >= 0 && tid >= 64 && tid < 192;
        int tnode = gt[tid & 7], torig = gt[8 + (tid & 7)], tslot = gt[16 + ((tid - 64) & 127)];
        const bool gsv = gs.gsl != nullptr; // the last refactor left L's and K's values in gs order: no gather
#pragma unroll
        for (int u = 0; u < LR; ++u) {
            const int p = min(tid + u * TW, max(nE - 1, 0));
            lsrc[u] = gsv ? p : (int)gs.lsrc[e0 + p];
            lij[u] = gs.lij[e0 + p];
        }
        if (tid <= nl + 1) st.lev[tid] = dsc[16 + tid];
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
        // ---- stage C: the values (second round trip), in the same order ----
        tnode = topt ? tnode : -1;
        torig = topt ? torig : -1;
        tslot = toptt ? tslot : -1;
        double breg[NR], dreg[NR];
        auto rhs_ptr = [&](int o) { // (rows beyond n + m -- the sparse cones' extra rows -- read a valid address and are masked)
            return o < ir.n ? ir.rx + o : ir.rz + (o < ir.n + ir.m ? o - ir.n : 0);
        };
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            breg[u] = *rhs_ptr(orig[u]);
            dreg[u] = v.Dinv[s0 + min(tid + u * TW, nloc - 1)];
        }
        const double tv0 = *rhs_ptr(max(torig, 0));
        const double tv1 = v.Dinv[max(tnode, 0)];
        const double tv2 = (tid < 128 ? v.Lx : v.Ux)[max(tslot, 0)];
        const double *lvals = gsv ? gs.gsl : v.Lx;
#pragma unroll
        for (int u = 0; u < LR; ++u) lv[u] = lvals[e0 + lsrc[u]];
        // ---- masks ----
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const bool in = tid + u * TW < nloc;
            breg[u] = (in && orig[u] < ir.n + ir.m) ? breg[u] : 0.0;
            if (in) og[tid + u * TW] = orig[u];
        }
#pragma unroll
        for (int u = 0; u < LR; ++u) {
            const bool in = tid + u * TW < nE;
            lij[u] = in ? lij[u] : 0xFFFFFFFFu;
            lv[u] = in ? lv[u] : 0.0;
        }
        double tval = 0.0, tdinv = 0.0;
        if (tid < 8) {
            tval = (torig >= 0 && torig < ir.n + ir.m) ? tv0 : 0.0;
            tdinv = tnode >= 0 ? tv1 : 0.0;
        } else if (tid >= 64 && tid < 192) {
            tval = tslot >= 0 ? tv2 : 0.0;
        }
        // ---- setrhs (directldlkktsolver.rs:160-166): the slices into LDS, ||b||inf of the bundle's rows ----
        double mx = 0.0;
        bool nan = false;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = tid + u * TW;
            if (i < nloc) {
                xs[i] = breg[u];
                bs[i] = breg[u];
                dis[i] = dreg[u];
                ir.bp[s0 + i] = breg[u];
                if (breg[u] != breg[u]) nan = true;
                else mx = fmax(mx, fabs(breg[u]));
            }
        }
        if (tid < 8) {
            st.btop[tid] = tval;
            if (tnode >= 0 && gfirst) ir.bp[tnode] = tval; // (bp holds the whole permuted right-hand side afterwards)
            st.dinvt[tid] = tdinv;
            st.acct[tid] = st.candt[tid] = st.dxt[tid] = st.rtop[tid] = 0.0;
            st.tnode[tid] = tnode;
            st.torig[tid] = torig;
        } else if (tid >= 64 && tid < 128) {
            st.ltt[tid - 64] = tval;
        } else if (tid >= 128 && tid < 192) {
            st.ktt[tid - 128] = tval;
        }
        normb_mine = lds_block_nanmax(mx, nan, red);
    }
    lds_barrier();
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
        if (nl == 0) lds_barrier();
        for (int l = 0; l < nl; ++l) {
            const int pa = st.lev[l], pb = st.lev[l + 1];
#pragma unroll
            for (int u = 0; u < LR; ++u) {
                const int p = tid + u * TW;
                if (p >= pa && p < pb) atomicAdd(&xs[lij[u] >> 16], -(lv[u] * xs[lij[u] & 0xFFFFu]));
            }
            lds_barrier();
        }
        const int pa = st.lev[nl], pb = st.lev[nl + 1];
#pragma unroll
        for (int u = 0; u < LR; ++u) { // (entries in the top rows, sorted by row: wave-uniform targets mostly)
            const int p = tid + u * TW;
            const bool in = p >= pa && p < pb;
            lds_scatter_add(st.tacc, in ? (int)(lij[u] >> 16) - nloc : -1, in ? lv[u] * xs[lij[u] & 0xFFFFu] : 0.0);
        }
        lds_barrier();
    };
    auto backward = [&]() { // qdldl.rs:737-752: x_j = y_j / d_j - sum_i l_ij x_i, the top rows' unknowns from st.dxt
        opaque_l();
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = tid + u * TW;
            if (i < nloc) xs[i] *= dis[i];
        }
        lds_barrier();
        {
            const int pa = st.lev[nl], pb = st.lev[nl + 1];
#pragma unroll
            for (int u = 0; u < LR; ++u) {
                const int p = tid + u * TW;
                if (p >= pa && p < pb) atomicAdd(&xs[lij[u] & 0xFFFFu], -(lv[u] * st.dxt[(lij[u] >> 16) - nloc]));
            }
        }
        lds_barrier();
        for (int l = nl - 1; l >= 0; --l) {
            const int pa = st.lev[l], pb = st.lev[l + 1];
#pragma unroll
            for (int u = 0; u < LR; ++u) {
                const int p = tid + u * TW;
                if (p >= pa && p < pb) atomicAdd(&xs[lij[u] & 0xFFFFu], -(lv[u] * xs[lij[u] >> 16]));
            }
            lds_barrier();
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
        lds_barrier();
        // (entries at or beyond position ntop0 lie in the top columns)
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
        lds_barrier();
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
        return lds_block_nanmax(m, nan, red);
    };
    // st.tacc[0 .. k) as tagged messages; with_norms: also this bundle's ||e||inf and ||b||inf (values 8, 9)
    int oz = 0;
    auto publish = [&](int slot, int tag, bool with_norms, double nrm) {
        int *recm = gs.msg + oz + (size_t)(b * 4 + slot) * GS_MV * 4;
        if (tid < k) msg_store(recm + tid * 4, st.tacc[tid], tag);
        else if (with_norms && (tid == 8 || tid == 9)) msg_store(recm + tid * 4, tid == 8 ? nrm : normb_mine, tag);
    };
    // the reference's decisions (as k_bundle_ir) from the system-wide norms in st.vn / st.vb
    auto decide = [&](int round) {
        if (tid == 0) {
            const double newnorm = st.vn;
            if (round == 0) st.normb = st.vb;
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
        lds_barrier();
    };
    // the k x k top part of both sweeps (every workgroup of the group alike): rhs - (the group's forward shares)
    auto top_solve = [&](int round) {
        if (tid == 0) {
            // (y in LDS, st.dxt: a private array indexed by a loop variable lives in scratch memory -- its reloads sat on
            // the critical path of every round; round 6, found on k_bundle_irs)
            for (int i = 0; i < k; ++i) {
                double sacc = (round == 0 ? st.btop[i] : st.rtop[i]) - st.fsum[i];
                for (int j = 0; j < i; ++j) sacc -= st.ltt[i * 8 + j] * st.dxt[j];
                st.dxt[i] = sacc;
            }
            for (int i = k - 1; i >= 0; --i) {
                double sacc = st.dxt[i] * st.dinvt[i];
                for (int j = i + 1; j < k; ++j) sacc -= st.ltt[j * 8 + i] * st.dxt[j];
                st.dxt[i] = sacc;
            }
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
    // The group's view of a candidate whose residual shares / norms were published in slot `slot` with `tag`: totals of
    // the shares -> st.rsum, top rows of the residual -> st.rtop, and the group's contribution to the system-wide norms
    // (its bundles' rows and its top rows), which the group's leader publishes to the other leaders.  All threads.
    auto group_collect = [&](int slot_f, int tag_f, bool with_f, int slot, int tag, int par) {
        if (tid < 64) {
            const bool ok = with_f ? gs_poll_group<true, true>(gs.msg + oz, gb0, gnb, k, slot_f, tag_f, slot, tag, st.fsum, st.rsum, st.gnorm)
                                   : gs_poll_group<false, true>(gs.msg + oz, gb0, gnb, k, 0, 0, slot, tag, nullptr, st.rsum, st.gnorm);
            if (!ok && tid == 0) st.timeout = 1;
        }
        lds_barrier();
        if (st.timeout) return false;
        if (tid == 0) {
            double gn = st.gnorm[0], gb = st.gnorm[1];
            if (ir.ir_enable) gn = nanmax(gn, top_residual());
            else // no refinement: the top entries of x take part in the finiteness test instead (:180)
                for (int i = 0; i < k; ++i) gn = nanmax(gn, fabs(st.candt[i]));
            for (int i = 0; i < k; ++i) gb = nanmax(gb, fabs(st.btop[i]));
            st.gnorm[0] = gn;
            st.gnorm[1] = gb;
        }
        lds_barrier();
        if (gfirst && tid < 2) msg_store(gs.lmsg + oz + ((size_t)(lead * 2 + par) * 2 + tid) * 4, st.gnorm[tid], tag);
        return true;
    };
    // The system-wide norms of that candidate -> st.vn / st.vb: EVERY workgroup polls all leaders' messages (no counter,
    // no last arriver, nobody reduces for the others, no second hop from a leader to its group: 16 bytes per leader and
    // poll, a few KB per workgroup) and reduces them itself, NaN propagating.
    auto verdict_inputs = [&](int tag, int par) {
        double vn = 0.0, vb = 0.0;
        bool ok = true;
        for (int q = tid; q < nlead && ok; q += TW) {
            const int *recl = gs.lmsg + oz + (size_t)(q * 2 + par) * 2 * 4;
            msg_v4i ma, mb;
            long long spins = 0;
            for (;;) {
                msg_load2(recl, recl + 4, ma, mb);
                if (msg_ready(ma, tag) && msg_ready(mb, tag)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1ll << 20)) {
                    ok = false;
                    break;
                }
            }
            vn = nanmax(vn, msg_value(ma));
            vb = nanmax(vb, msg_value(mb));
        }
        if (!ok) st.timeout = 1;
        const double n1 = lds_block_nanmax(vn != vn ? 0.0 : vn, vn != vn, red);
        const double n2 = lds_block_nanmax(vb != vb ? 0.0 : vb, vb != vb, red);
        if (tid == 0) {
            st.vn = n1;
            st.vb = n2;
        }
        lds_barrier();
        return st.timeout == 0;
    };
    // ---- round 0's forward sweep and shares; then the entries of K (see above): indices and values as two dependent
    // requests while the shares travel to the group's other workgroups ----
    forward();
    stamp();
    if (k > 0) publish(0, gs.epoch * 64, false, 0.0);
    {
        int usrc[UR];
        const bool gsv = gs.gsu != nullptr;
        const double *uvals = gsv ? gs.gsu : v.Ux;
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int p = min(tid + u * TW, max(nU - 1, 0));
            usrc[u] = gsv ? p : (int)gs.usrc[ub + p];
            uij[u] = gs.uij[ub + p];
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) uv[u] = uvals[ub + usrc[u]];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const bool in = tid + u * TW < nU;
            uij[u] = in ? uij[u] : 0xFFFFFFFFu;
            uv[u] = in ? uv[u] : 0.0;
        }
    }
    bool pending = false;
    for (int round = 0;; ++round) {
        const int par = round & 1;
        // tags of this launch's messages: forward shares of round r, residual shares / norms of round r
        const int tagF = gs.epoch * 64 + 2 * round, tagR = tagF + 1, tagRprev = tagF - 1;
        // (an opaque zero added to the message bases: their lane addresses are recomputed where they are used instead of
        // being kept in registers across the whole loop)
        oz = 0;
        asm volatile("" : "+v"(oz));
        if (round > 0) { // (round 0's sweep and shares: ahead of the loop)
            forward();
            stamp();
            if (k > 0) publish(par, tagF, false, 0.0);
        }
        // the group's forward shares; and what the previous round's candidate left: its residual's top rows (this
        // round's right-hand side there) and the group's norms, which go out to the other leaders
        if (pending) {
            if (!group_collect(par, tagF, k > 0, 2 + (par ^ 1), tagRprev, par ^ 1)) return bail();
        } else if (k > 0) {
            if (tid < 64 && !gs_poll_group<true, false>(gs.msg + oz, gb0, gnb, k, par, tagF, 0, 0, st.fsum, nullptr, nullptr) && tid == 0)
                st.timeout = 1;
            lds_barrier();
            if (st.timeout) return bail();
        }
        top_solve(round);
        lds_barrier();
        stamp();
        backward();
        stamp();
        if (pending) { // the verdict on the previous round's candidate
            if (!verdict_inputs(tagRprev, par ^ 1)) return bail();
            decide(round - 1);
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
        lds_barrier();
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
            mine = lds_block_nanmax(mx, nan, red);
            if (tid < 8) st.tacc[tid] = 0.0;
            lds_barrier();
        } else {
            mine = residual(cnd);
        }
        stamp();
        publish(2 + par, tagR, true, mine);
        pending = true;
        stamp();
        if (ir.ir_enable && round < ir.maxiter) continue; // (the verdict is awaited in the middle of the next round)
        if (!group_collect(0, 0, false, 2 + par, tagR, par)) return bail();
        stamp();
        if (!verdict_inputs(tagR, par)) return bail();
        stamp();
        decide(round);
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
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = tid + u * TW;
            if (i < nloc) put(og[i], acc[i]);
        }
        if (gfirst && tid < k) put(st.torig[tid], st.acct[tid]);
    }
    if (b == 0 && tid == 0) {
        ir.res[0] = ok ? 1 : -1; // (0 = the kernel never got here)
        ir.res[1] = st.rounds;
        ir.res[3] = 0;
    }
    stamp();
}

// ---------------------------------------------------------------------------
// k_gstep_factor: numeric LDL' of every bundle, entry-parallel and right-looking with the bundle's values in LDS and the
// symbolic phase's update records walked flat (as k_bundle_factor_flat), then the bundle's contribution to the Schur
// complement of its group's top S[i][j] = sum over the bundle's columns c of l_ic d_c l_jc straight from LDS, published
// as tagged messages; the LAST workgroup of a group to deposit its share (one arrival counter per group, nobody waits)
// sums the shares in a fixed order, subtracts them from K_tt and factors the k x k block with the pivot rule of
// qdldl.rs:645-665.  Built for bundles so small that the launch is a chain of latencies, not bandwidth: every index the
// launch needs (column pointers, 16-bit rows / columns of the entries, signs, the levels' first update records) is
// requested in ONE round trip behind the bundle's descriptor and parked in LDS / registers; a level then costs three
// barriers of LDS work -- pivots (thread per column), scaling (thread per ENTRY: a separator column has hundreds), updates.
// ---------------------------------------------------------------------------
constexpr int GF_TW = 256; // (four workgroups per CU at <= 128 registers: every bundle of a 1024-bundle share is resident at once)
constexpr int GF_FU = 8; // update records in flight per thread
struct GfLds { // layout of the dynamic LDS of k_gstep_factor (doubles first)
    double *Ls, *Ds, *Di;
    int *lp;
    unsigned short *li, *lj;
    signed char *sg;
};
__host__ __device__ inline size_t gf_lds_bytes(int nE, int nloc) {
    size_t b = (size_t)(nE + 2 * nloc) * sizeof(double) + (size_t)(nloc + 1) * sizeof(int);
    b += (size_t)2 * ((nE + 3) & ~3) * sizeof(unsigned short) + (size_t)((nloc + 7) & ~7);
    return (b + 15) & ~(size_t)15;
}
// nE / nloc: the launch's maxima (the same layout in every workgroup); nE_own: this bundle's entry count -- its pivots
// follow its entries directly, because an update record addresses a pivot as target nE_own + row
__device__ __forceinline__ GfLds gf_lds(char *smem, int nE, int nloc, int nE_own) {
    GfLds l;
    l.Ls = (double *)smem;
    l.Ds = l.Ls + nE_own;
    l.Di = l.Ls + nE + nloc;
    l.lp = (int *)(l.Di + nloc);
    l.li = (unsigned short *)(l.lp + nloc + 1);
    l.lj = l.li + ((nE + 3) & ~3);
    l.sg = (signed char *)(l.lj + ((nE + 3) & ~3));
    return l;
}
__global__ __launch_bounds__(GF_TW) void k_gstep_factor(LdlView v, BundleView bv, GFoldView gf, GStepView gs) {
    constexpr int TW = GF_TW, FU = GF_FU;
    extern __shared__ __attribute__((aligned(16))) char ff_smem[];
    __shared__ double A[36], wsum[(GF_TW / 64) * 36];
    __shared__ int s_last, s_tp[GS_MAXL + 2], s_ln[GS_MAXL + 2];
    const int b = blockIdx.x, tid = threadIdx.x;
    // ---- stage A: the bundle's descriptor ----
    const int *dsc = gs.desc + (size_t)b * GS_DESC;
    const int s0 = dsc[0], nloc = dsc[1], e0 = dsc[2], nE = dsc[3], ub = dsc[4], nU = dsc[5], nl = dsc[6], grp = dsc[7];
    const int gbase = dsc[8], k = dsc[9], gb0 = dsc[10], gnb = dsc[11];
    const int np = k * (k + 1) / 2;
    const GfLds L = gf_lds(ff_smem, bv.max_entries, bv.max_nodes, nE);
    double *Ls = L.Ls, *Ds = L.Ds, *Di = L.Di;
    typedef unsigned short fu_v4 __attribute__((ext_vector_type(4)));
    const fu_v4 *rec = (const fu_v4 *)v.fu_rec;
    int dbgn = 0;
    auto stamp = [&]() { // diagnostics (CHIP_IR_DEBUG=3): phase boundaries of every workgroup on the 100 MHz clock
        if (gs.dbg && tid == 0 && dbgn < 31) {
            if (dbgn == 0)
                gs.dbg[(size_t)b * 32] = (long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4) |
                                         ((long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 20) << 32);
            gs.dbg[(size_t)b * 32 + 1 + dbgn++] = wall_clock64();
        }
    };
    stamp();
    // ---- stage B: every index and value the launch needs, in one round trip ----
    // The bundle's U entries (K's values) are read through the gs order of the solve kernel and written back in that
    // order (gs.gsu): the solves then load them, and L (gs.gsl, below), fully coalesced -- the gather through the 16-bit
    // source positions is paid once per refactor here instead of once per solve there.
    constexpr int UF = 8; // U entries per thread held in registers (more: streamed)
    constexpr int LF = 8; // gs positions of L per thread whose source slot is prefetched (more: loaded at the end)
    unsigned short uslot[UF], urow[UF], usrc[UF], lsrcr[LF];
    double uval[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
        const int t = min(tid + u * TW, max(nU - 1, 0));
        usrc[u] = gs.usrc[ub + t];
        uslot[u] = gs.ufs[ub + t];
        urow[u] = (unsigned short)(gs.uij[ub + t] >> 16);
    }
#pragma unroll
    for (int u = 0; u < LF; ++u) lsrcr[u] = gs.lsrc[e0 + min(tid + u * TW, max(nE - 1, 0))];
    for (int i = tid; i <= nloc; i += TW) L.lp[i] = v.Lp[s0 + i] - e0;
    for (int i = tid; i < nloc; i += TW) L.sg[i] = v.dsigns[s0 + i];
    for (int q = tid; q < nE; q += TW) {
        L.li[q] = v.Li16[e0 + q];
        L.lj[q] = v.Lj16[e0 + q];
        Ls[q] = 0.0; // (fill-in slots stay zero)
    }
    if (tid <= nl) {
        s_tp[tid] = dsc[32 + tid]; // first update record of every level, then the end
        s_ln[tid] = dsc[48 + tid]; // first node of every level (bundle-local), then the node count
    }
    bool eps_on;
    __shared__ double s_eps;
    (void)static_eps(v, &eps_on, &s_eps);
#pragma unroll
    for (int u = 0; u < UF; ++u) {
        uval[u] = v.Ux[ub + usrc[u]];
        if (tid + u * TW >= nU) uslot[u] = (unsigned short)0xFFFEu;
    }
    // the level-0 records, and K_tt for whoever turns out to be the group's last arriver (lane p < np of wave 0)
    fu_v4 r[FU];
    auto request = [&](int base, int re) {
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int t = base + u * TW + tid;
            if (t < re) r[u] = rec[t];
            else r[u] = fu_v4{0, 0, 0, 0xFFFF};
        }
    };
    request(dsc[32], dsc[33]);
    int tq = -2, tnode = -1, ti = 0, tj = 0, tks = -1, tsg = 0;
    if (grp >= 0 && tid < np) {
        while ((ti + 1) * (ti + 2) / 2 <= tid) ++ti;
        tj = tid - ti * (ti + 1) / 2;
        const int *gt = gs.gtop + (size_t)grp * GS_GTOP;
        tnode = gt[ti];
        tq = ti == tj ? -2 : gt[16 + ti * 8 + tj];
        tks = gt[80 + ti * 8 + tj]; // position of K(top_i, top_j) in V, -1: structurally zero
        tsg = gt[144 + ti];
    }
    // K_tt entry: fast preparation -- straight from K (+- eps on the diagonal); otherwise as k_scatter_init left it in
    // D / the top-top slots of Lx (static regulariser included)
    double ktt0 = 0.0;
    if (grp >= 0 && tid < np) {
        if (v.eps_slots) {
            ktt0 = tks >= 0 ? v.Ux[tks] : 0.0; // (+- eps on the diagonal: applied by the last arriver, below)
        } else {
            ktt0 = ti == tj ? v.D[tnode] : (tq >= 0 ? v.Lx[tq] : 0.0);
        }
    }
    if (tid < 36) A[tid] = 0.0;
    __syncthreads();
    const double eps = s_eps;
    stamp();
    // ---- the bundle's U entries (initial values of its columns) into LDS ----
    auto place = [&](unsigned short slot, unsigned short row, double val) {
        if (slot == 0xFFFFu) Ds[row] = eps_on ? (L.sg[row] == 1 ? val + eps : val - eps) : val;
        else if (slot != 0xFFFEu) Ls[slot] = val;
    };
#pragma unroll
    for (int u = 0; u < UF; ++u) {
        place(uslot[u], urow[u], uval[u]);
        if (tid + u * TW < nU) gs.gsu[ub + tid + u * TW] = uval[u];
    }
    for (int t = tid + UF * TW; t < nU; t += TW) {
        const double val = v.Ux[ub + gs.usrc[ub + t]];
        place(gs.ufs[ub + t], (unsigned short)(gs.uij[ub + t] >> 16), val);
        gs.gsu[ub + t] = val;
    }
    __syncthreads();
    stamp();
    for (int l = 0; l < nl; ++l) {
        const int rb = s_tp[l], re = s_tp[l + 1];
        // the level's columns are final: pivot rule (thread per column) ...
        for (int j = s_ln[l] + tid; j < s_ln[l + 1]; j += TW) {
            double d = Ds[j];
            const double sign = (double)L.sg[j];
            if (d * sign < v.reg_eps) {
                d = v.reg_delta * sign;
                atomicAdd(&v.status[2], 1); // rare
            }
            if (d == 0.0) v.status[1] = 1;
            const double dinv = 1.0 / d;
            if (!isfinite(dinv)) v.status[0] = 1;
            v.D[s0 + j] = d;
            v.Dinv[s0 + j] = dinv;
            Ds[j] = d;
            Di[j] = dinv;
        }
        __syncthreads();
        // ... scale (thread per entry of the level's columns: one contiguous range of the CSC arrays)
        for (int q = L.lp[s_ln[l]] + tid; q < L.lp[s_ln[l + 1]]; q += TW) Ls[q] *= Di[L.lj[q]];
        __syncthreads();
        // ... and the updates the level's columns cause: target -= l_a (l_b d_k)
        for (int base = rb; base < re; base += TW * FU) { // (wave-uniform bounds: lds_scatter_add is cross-lane)
            if (base != rb) request(base, re);
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const bool ok = r[u].w != 0xFFFFu;
                const double val = ok ? Ls[r[u].x] * (Ls[r[u].y] * Ds[r[u].z]) : 0.0;
                lds_scatter_add(Ls, ok ? (int)r[u].w : -1, -val);
            }
        }
        if (l + 1 < nl) request(s_tp[l + 1], s_tp[l + 2]); // (index data: in flight across the barrier)
        __syncthreads();
        stamp();
    }
    for (int q = tid; q < nE; q += TW) v.Lx[e0 + q] = Ls[q];
#pragma unroll
    for (int u = 0; u < LF; ++u)
        if (tid + u * TW < nE) gs.gsl[e0 + tid + u * TW] = Ls[lsrcr[u]];
    for (int p = tid + LF * TW; p < nE; p += TW) gs.gsl[e0 + p] = Ls[gs.lsrc[e0 + p]];
    static_eps_epilogue(v, eps);
    stamp();
    if (grp < 0) return;
    // ---- this bundle's share of the Schur complement of its group's top: a column's entries in the top rows are its
    // last ones (16-bit row index >= nloc); per-thread register accumulators over the packed lower triangle, reduced
    // wave by wave in a fixed order (as k_gfold_schur, with L, D and the indices read from LDS) ----
    {
        double sa[36];
#pragma unroll
        for (int p = 0; p < 36; ++p) sa[p] = 0.0;
        for (int j = tid; j < nloc; j += TW) {
            const int cb = L.lp[j], ce = L.lp[j + 1];
            double vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) vv[i] = 0.0;
            bool any = false;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int q = ce - 1 - e;
                const int tt = q >= cb ? (int)L.li[q] - nloc : -1;
                if (tt >= 0) {
                    const double val = Ls[q];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (tt == i) vv[i] = val;
                    any = true;
                }
            }
            if (any) {
                const double dj = Ds[j];
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
            msg_store(gs.fmsg + ((size_t)b * 36 + tid) * 4, t, gs.epoch);
        }
    }
    stamp();
    if (tid == 0) {
        __builtin_amdgcn_s_waitcnt(0); // (this workgroup's messages -- all from wave 0 -- have been acknowledged)
        const int old = atomicAdd(gf.gcnt + grp * 32 + 2, 1);
        s_last = (old + 1 == gnb) ? 1 : 0;
        if (s_last) __hip_atomic_store(gf.gcnt + grp * 32 + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    stamp();
    if (!s_last) return;
    // ---- the last arriver: K_tt minus the shares in bundle order, then the k x k LDL' ----
    if (tid < np) {
        double a = ktt0;
        if (v.eps_slots && ti == tj && eps_on) a = tsg == 1 ? a + eps : a - eps;
        for (int q = gb0; q < gb0 + gnb; ++q) {
            const int *addr = gs.fmsg + ((size_t)q * 36 + tid) * 4;
            msg_v4i m = msg_load(addr);
            long long spins = 0; // (never spins: every mate's messages were acknowledged before its arrival)
            while ((m.y != gs.epoch || m.w != gs.epoch) && ++spins < (1ll << 20)) m = msg_load(addr);
            a -= __hiloint2double(m.z, m.x);
        }
        A[tid] = a;
    }
    __syncthreads();
    if (tid != 0) return;
    const int *gt = gs.gtop + (size_t)grp * GS_GTOP;
    for (int j = 0; j < k; ++j) {
        const double dinv = pivot_rule(v, gt[j], A[j * (j + 1) / 2 + j]);
        for (int i = j + 1; i < k; ++i) {
            const double aij = A[i * (i + 1) / 2 + j];
            for (int i2 = j + 1; i2 <= i; ++i2) A[i * (i + 1) / 2 + i2] -= aij * (A[i2 * (i2 + 1) / 2 + j] * dinv);
        }
        for (int i = j + 1; i < k; ++i) {
            const double lij = A[i * (i + 1) / 2 + j] * dinv;
            A[i * (i + 1) / 2 + j] = lij;
            const int q = gt[16 + i * 8 + j];
            if (q >= 0) {
                v.Lx[q] = lij;
                if (v.mirror_rows) v.Rx[v.Tpos[q]] = lij;
            }
        }
    }
    stamp();
    (void)gbase;
}

} // namespace

// ===========================================================================
// launch wrappers
// ===========================================================================
static size_t gstep_solve_lds(const BundleView &bv) {
    return ((size_t)bv.max_nodes * (5 * sizeof(double) + sizeof(int)) + 15) & ~(size_t)15;
}
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
bool gstep_factor_ok(const BundleView &bv) {
    const size_t lds = gf_lds_bytes(bv.max_entries, bv.max_nodes);
    if (lds > 37 * 1024) return false; // four workgroups per CU
    return raise_dynamic_lds((const void *)k_gstep_factor, lds) == hipSuccess;
}
int gstep_factor(hipStream_t s, const LdlView &v, const BundleView &bv, const GFoldView &gf, const GStepView &gs) {
    k_gstep_factor<<<bv.nb, GF_TW, gf_lds_bytes(bv.max_entries, bv.max_nodes), s>>>(v, bv, gf, gs);
    return (int)hipGetLastError();
}

} // namespace dev
} // namespace chip
