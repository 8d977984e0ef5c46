// engine.cpp -- device-resident numeric LDL' / solve / residual sequences.
// One HIP stream per handle; dependencies between elimination-tree levels are
// kernel boundaries on that stream (see dev_common.hpp for the rationale).
#include "engine.hpp"

#include <cstdlib>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>

namespace chip {

std::string hip_err(hipError_t e, const char *what) {
    return std::string(what) + ": " + hipGetErrorString(e);
}

Engine::~Engine() {
    if (stream) (void)hipStreamSynchronize(stream);
    for (hipEvent_t ev : prof_events) (void)hipEventDestroy(ev);
    for (SolveGraph &g : graphs) (void)hipGraphExecDestroy(g.exec);
    for (void *p : allocs) (void)hipFree(p);
    if (mb_host) (void)hipHostFree(mb_host);
    if (alt_active) swap_ctx();
    if (nrm_host) (void)hipHostFree(nrm_host);
    if (alt.nrm_host) (void)hipHostFree(alt.nrm_host);
    if (alt.stream) (void)hipStreamDestroy(alt.stream);
    if (pair_event) (void)hipEventDestroy(pair_event);
    if (pair_ev_a) (void)hipEventDestroy(pair_ev_a);
    if (pair_ev_b) (void)hipEventDestroy(pair_ev_b);
    if (exch_event) (void)hipEventDestroy(exch_event);
    if (snb_ready) (void)hipEventDestroy(snb_ready);
    for (hipEvent_t e : hs_ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t ev : snb_events)
        if (ev) (void)hipEventDestroy(ev);
    if (stream) (void)hipStreamDestroy(stream);
}

template <typename T> int Engine::alloc(T **dst, size_t n) {
    void *p = nullptr;
    CHIP_HIP(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    allocs.push_back(p);
    *dst = (T *)p;
    return CHIP_OK;
}
template <typename T, typename A> int Engine::upload(T **dst, const std::vector<T, A> &src, size_t n) {
    int rc = alloc(dst, n);
    if (rc) return rc;
    if (n) CHIP_HIP(hipMemcpy(*dst, src.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return CHIP_OK;
}
int Engine::upload_lists(DeviceLists &Dl, const LevelLists &L, int chain_max_w) {
    Dl.t_ptr = L.t_ptr;
    Dl.w_ptr = L.w_ptr;
    Dl.b_ptr = L.b_ptr;
    Dl.br_ptr = L.br_ptr;
    int rc;
    if ((rc = upload(&Dl.t_idx, L.t_idx, L.t_idx.size()))) return rc;
    if ((rc = upload(&Dl.w_idx, L.w_idx, L.w_idx.size()))) return rc;
    if ((rc = upload(&Dl.b_row, L.b_row, L.b_row.size()))) return rc;
    if ((rc = upload(&Dl.b_beg, L.b_beg, L.b_beg.size()))) return rc;
    if ((rc = upload(&Dl.b_end, L.b_end, L.b_end.size()))) return rc;
    if ((rc = upload(&Dl.br_idx, L.br_idx, L.br_idx.size()))) return rc;
    if ((rc = upload(&Dl.d_t_ptr, L.t_ptr, L.t_ptr.size()))) return rc;
    if ((rc = upload(&Dl.d_w_ptr, L.w_ptr, L.w_ptr.size()))) return rc;
    // runs of narrow levels (no B chunks, few rows): candidates for the chain kernel
    const int nl = (int)L.t_ptr.size() - 1;
    Dl.chain_end.assign((size_t)(nl > 0 ? nl : 0), 0);
    Dl.chain_begin.assign((size_t)(nl > 0 ? nl : 0), 0);
    auto narrow = [&](int l) {
        return L.b_ptr[l + 1] == L.b_ptr[l] && L.t_ptr[l + 1] - L.t_ptr[l] <= 2048 &&
               L.w_ptr[l + 1] - L.w_ptr[l] <= chain_max_w;
    };
    for (int l = 0; l < nl;) {
        int e = l;
        while (e < nl && narrow(e)) e++;
        if (e - l >= 2) {
            for (int k = l; k < e; k++) {
                Dl.chain_end[k] = e;
                Dl.chain_begin[k] = l;
            }
            l = e;
        } else {
            Dl.chain_end[l] = l + 1;
            Dl.chain_begin[l] = l;
            l++;
        }
    }
    return CHIP_OK;
}

void Engine::init_host_only(const Symbolic &S, const chip_settings &settings) {
    st = settings;
    host_only = true;
    N = S.N;
    nlevels = S.nlevels;
    nnzK = S.nnzK;
    nnzL = S.nnzL;
    nnzS = S.nnzS;
    nnzU = S.nnzU;
    h_perm = S.perm;
    h_lvlptr = S.lvlptr;
    h_etree = S.etree;
    h_level = S.level;
    NF = S.NF;
    tree_depth = S.tree_depth;
    h_Lp = S.Lp;
    h_Li = S.Li;
    amd = S.amd;
    h_sn_ptr = S.sn_ptr;
    h_sn_col = S.sn_col;
    nsn = (int)S.sn_ptr.size() - 1;
    nfaclevels = S.nfaclevels;
}

int Engine::get_symbolic(uint64_t *etree, uint64_t *oLp, uint64_t *oLi, uint64_t *lvlptr) const {
    if (etree)
        for (int i = 0; i < N; i++) etree[i] = h_etree[i] < 0 ? UINT64_MAX : (uint64_t)h_etree[i];
    if (lvlptr)
        for (int i = 0; i < N; i++) lvlptr[i] = (uint64_t)h_level[i];
    if (oLp || oLi) {
        std::vector<i32> tp, ti;
        const i32 *pLp = h_Lp.data(), *pLi = h_Li.data();
        if (!host_only) {
            tp.resize((size_t)N + 1);
            ti.resize((size_t)nnzL + 1);
            CHIP_HIP(hipMemcpy(tp.data(), Lp, ((size_t)N + 1) * sizeof(int), hipMemcpyDeviceToHost));
            if (nnzL) CHIP_HIP(hipMemcpy(ti.data(), Li, (size_t)nnzL * sizeof(int), hipMemcpyDeviceToHost));
            pLp = tp.data();
            pLi = ti.data();
        }
        if (oLp)
            for (int i = 0; i <= N; i++) oLp[i] = (uint64_t)pLp[i];
        if (oLi)
            for (i64 i = 0; i < nnzL; i++) oLi[i] = (uint64_t)pLi[i];
    }
    return CHIP_OK;
}

int Engine::init(const Symbolic &S, const chip_settings &settings) {
    st = settings;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (the product has no CPU fallback)");
        return CHIP_ERR_NO_DEVICE;
    }
    if (st.device >= 0) {
        CHIP_HIP(hipSetDevice(st.device));
        device = st.device;
    } else {
        CHIP_HIP(hipGetDevice(&device));
    }
    CHIP_HIP(hipStreamCreate(&stream));
    dev::solve_kernel_attributes();
    N = S.N;
    nlevels = S.nlevels;
    nnzK = S.nnzK;
    nnzL = S.nnzL;
    nnzS = S.nnzS;
    h_perm = S.perm;
    h_lvlptr = S.lvlptr;
    h_etree = S.etree;
    h_level = S.level;
    NF = S.NF;
    tree_depth = S.tree_depth;
    amd = S.amd;
    int rc;
    const size_t n = (size_t)N;
    nnzU = S.nnzU;
    if ((rc = upload(&v2l, S.v2l, (size_t)(nnzK - nnzU)))) return rc;
    h_k2v = S.k2v;
    h_v2k = S.v2k;
    nfill = (int)S.fill_idx.size();
    fill_from = S.fill_from;
    if ((rc = upload(&fill_idx, S.fill_idx, S.fill_idx.size()))) return rc;
    if ((rc = upload(&Lp, S.Lp, n + 1))) return rc;
    if ((rc = upload(&Li, S.Li, (size_t)nnzL))) return rc;
    if ((rc = upload(&Rp, S.Rp, n + 1))) return rc;
    nnzR = S.nnzR;
    if ((rc = upload(&Rcol, S.Rcol, (size_t)nnzR))) return rc;
    if ((rc = upload(&Rpos, S.Rpos, (size_t)nnzR))) return rc;
    if ((rc = upload(&Tpos, S.Tpos, (size_t)nnzL))) return rc;
    if ((rc = upload(&perm, S.perm, n))) return rc;
    if ((rc = upload(&iperm, S.iperm, n))) return rc;
    if ((rc = upload(&Sp, S.Sp, n + 1))) return rc;
    if ((rc = upload(&Scol, S.Scol, (size_t)nnzS))) return rc;
    if (!S.xperm.empty()) { // (the residual over the top rows reads x through a supernode-contiguous copy)
        if ((rc = upload(&xperm, S.xperm, S.xperm.size()))) return rc;
        if ((rc = alloc(&xs_view, n))) return rc;
    }
    if ((rc = upload(&Smap, S.Smap, (size_t)nnzS))) return rc;
    if (!S.dblk_p0.empty()) {
        const int nblk = (int)S.dblk_p0.size();
        std::vector<i32> rowbase((size_t)nblk, 0);
        const std::vector<i32> &rownode = S.dblk_node;
        int mmax = 0;
        for (int b = 0, o = 0; b < nblk; b++) {
            rowbase[(size_t)b] = o;
            o += S.dblk_m[(size_t)b];
            mmax = std::max(mmax, (int)S.dblk_m[(size_t)b]);
        }
        int *dp0 = nullptr, *dm = nullptr, *drb = nullptr, *dst = nullptr, *drn = nullptr;
        if ((rc = upload(&dp0, S.dblk_p0, S.dblk_p0.size()))) return rc;
        if ((rc = upload(&dm, S.dblk_m, S.dblk_m.size()))) return rc;
        if ((rc = upload(&drb, rowbase, rowbase.size()))) return rc;
        if ((rc = upload(&dst, S.dblk_start, S.dblk_start.size()))) return rc;
        if ((rc = upload(&drn, rownode, rownode.size()))) return rc;
        h_dblk_node = S.dblk_node;
        h_dblk_m = S.dblk_m;
        dblk.nblk = nblk;
        dblk.split = std::max(1, std::min(8, 1024 / nblk)); // (enough workgroups to fill the chip; a block's rows interleaved)
        dblk.mmax = mmax;
        dblk.nrows = (int)rownode.size();
        dblk.p0 = dp0;
        dblk.m = dm;
        dblk.rowbase = drb;
        dblk.start = dst;
        dblk.rownode = drn;
        if ((rc = alloc(&dblk.P, (size_t)dblk.nrows * dblk.split))) return rc;
        if ((rc = alloc(&bt_view, n))) return rc;
        { // rows of the blocks that are contiguous in L too -> dblk_l0, and the rest of K's top entries -> rest_idx
            const size_t nrows = rownode.size();
            const i64 ntop = (i64)nnzK - (i64)nnzU;
            std::vector<i32> l0(nrows, -1), rowlen(nrows, 0);
            for (int b = 0, o = 0; b < nblk; o += S.dblk_m[(size_t)b], b++)
                for (i32 a = 0; a < S.dblk_m[(size_t)b]; a++) rowlen[(size_t)(o + a)] = S.dblk_m[(size_t)b] - a - 1;
            std::atomic<int> bad{0};
            run_threads(host_threads(), [&](int t, int TT) {
                for (size_t r = (size_t)t; r < nrows; r += (size_t)TT) {
                    const i64 u0 = (i64)S.dblk_start[r] - (i64)nnzU, len = rowlen[r];
                    if (len <= 0) continue;
                    if (u0 < 0 || u0 + len > ntop) {
                        bad = 1;
                        continue;
                    }
                    const i64 base = S.v2l[(size_t)u0];
                    if (base < 0 || base + len > (i64)nnzL) bad = 1;
                    for (i64 q = 1; q < len && !bad; q++)
                        if ((i64)S.v2l[(size_t)(u0 + q)] != base + q) bad = 1;
                    l0[r] = (i32)base;
                }
            });
            i64 covered = 0;
            for (size_t r = 0; r < nrows; r++) covered += rowlen[r];
            // (worth it -- and the list of the other entries short -- only where the blocks are most of K's top entries)
            if (!bad && S.fill_from >= 0 && ntop < (i64)2000000000 && covered * 2 > ntop) {
                std::vector<std::pair<i64, i64>> iv;
                iv.reserve(nrows);
                for (size_t r = 0; r < nrows; r++)
                    if (rowlen[r] > 0) iv.emplace_back((i64)S.dblk_start[r] - (i64)nnzU, (i64)rowlen[r]);
                std::sort(iv.begin(), iv.end());
                std::vector<i32> rest;
                i64 pos = 0;
                bool overlap = false;
                for (const auto &pr : iv) {
                    if (pr.first < pos) overlap = true;
                    for (i64 u = pos; u < pr.first; u++) rest.push_back((i32)u);
                    pos = std::max(pos, pr.first + pr.second);
                }
                for (i64 u = pos; u < ntop; u++) rest.push_back((i32)u);
                if (!overlap) {
                    if ((rc = upload(&dblk_l0, l0, l0.size()))) return rc;
                    if ((rc = upload(&rest_idx, rest, rest.size()))) return rc;
                    nrest = (int)rest.size();
                    hs_direct_ok = true;
                }
            }
        }
        const int da = dev::dblk_attributes(mmax);
        if (da < 0) {
            set_error("k_dblk_symv: dynamic LDS attribute");
            return CHIP_ERR_HIP;
        }
        dblk_pair_ok = (da & 1) != 0;
    }
    if ((rc = upload(&Up, S.Up, S.Up.size()))) return rc;
    if ((rc = upload(&Ucol, S.Ucol, (size_t)nnzU))) return rc;
    if ((rc = upload(&dsigns, S.dsigns, n))) return rc;
    if ((rc = alloc(&Kx, (size_t)nnzK))) return rc;
    Ux = Kx; // the U rows are the head of the value store (T order)
    if ((rc = alloc(&Lx, (size_t)nnzL))) return rc;
    if ((rc = alloc(&Rx, (size_t)nnzR))) return rc;
    if ((rc = alloc(&D, n))) return rc;
    if ((rc = alloc(&Dinv, n))) return rc;
    if ((rc = alloc(&Sx, (size_t)nnzS))) return rc;
    if ((rc = upload_lists(fac, S.fac, 2))) return rc; // factor: W columns of a chained level run one after the other
    if ((rc = upload_lists(fwd, S.fwd))) return rc;
    if ((rc = upload_lists(bwd, S.bwd))) return rc;
    if ((rc = upload_lists(smv, S.smv))) return rc;
    nfaclevels = S.nfaclevels;
    nsn = (int)S.sn_ptr.size() - 1;
    h_sn_ptr = S.sn_ptr;
    h_sn_col = S.sn_col;
    if (nsn > 0) {
        if ((rc = upload_lists(snx, S.snx))) return rc;
        if ((rc = upload_lists(snb, S.snb))) return rc;
        { // groups of unit levels whose bundle contributions run beside the supernode chain (refactor_enqueue): level 0
          // alone, then levels until a group holds >= 8 % of the work
            long long total = 0;
            for (long long w : S.snb_work) total += w;
            snb_group_at.assign((size_t)S.nfaclevels + 1, -1);
            const int nl = std::min((int)S.snb_work.size(), S.nfaclevels);
            for (int l = 0; l < nl && total > 0;) {
                const int l0 = l;
                long long acc = 0;
                do acc += S.snb_work[(size_t)l++];
                while (l0 > 0 && l < nl && acc * 100 < total * 8);
                snb_group_at[(size_t)l0] = (i32)snb_groups.size();
                snb_groups.push_back({l0, l});
            }
        }
        if ((rc = upload_lists(fwu, S.fwu))) return rc;
        if ((rc = upload_lists(bwu, S.bwu))) return rc;
        nRf = S.Rf_p.empty() ? 0 : S.Rf_p.back();
        if ((rc = alloc(&Rfx, (size_t)nRf))) return rc;
        if ((rc = upload(&sn_ptr, S.sn_ptr, S.sn_ptr.size()))) return rc;
        if ((rc = upload(&sn_col, S.sn_col, S.sn_col.size()))) return rc;
        {
            // the supernodes in level order as records (snode.hip: SN_REC = 8 ints): id, first member, width, last
            // member column, rows of B
            std::vector<i32> rec(S.sn_order.size() * 8 + 8, 0);
            for (size_t k = 0; k < S.sn_order.size(); k++) {
                const i32 sn = S.sn_order[k], p0 = S.sn_ptr[sn], e = S.sn_col[S.sn_ptr[sn + 1] - 1];
                rec[8 * k] = sn;
                rec[8 * k + 1] = p0;
                rec[8 * k + 2] = S.sn_ptr[sn + 1] - p0;
                rec[8 * k + 3] = e;
                rec[8 * k + 4] = S.Lp[e + 1] - S.Lp[e];
                rec[8 * k + 5] = S.Lp[e];
                rec[8 * k + 6] = -1; // (offset of G: filled in below for the supernodes that keep one)
                rec[8 * k + 7] = -1;
            }
            if ((rc = upload(&sn_order, rec, rec.size()))) return rc;
        }
        {
            std::vector<i32> geo((size_t)2 * nsn + 2, 0);
            for (int sn = 0; sn < nsn; sn++) {
                const i32 e = S.sn_col[S.sn_ptr[sn + 1] - 1];
                geo[2 * sn] = e;
                geo[2 * sn + 1] = S.Lp[e + 1] - S.Lp[e];
            }
            if ((rc = upload(&sn_geo, geo, geo.size()))) return rc;
            // column bases of the dense panels (snode.hip: SnodeGeom::cb): every supernode kernel started with
            // Lp[cols[t]] - t - 1 for its columns, two dependent loads before the first useful one
            std::vector<i32> cb(S.sn_col.size() + 1, 0);
            for (int sn = 0; sn < nsn; sn++)
                for (i32 p = S.sn_ptr[sn]; p < S.sn_ptr[sn + 1]; p++) cb[p] = S.Lp[S.sn_col[p]] - (p - S.sn_ptr[sn]) - 1;
            if ((rc = upload(&sn_cb, cb, cb.size()))) return rc;
            if ((rc = alloc(&sn_d, S.sn_col.size() + 1))) return rc;
            {
                std::vector<int8_t> sg(S.sn_col.size() + 1, 1);
                for (size_t q = 0; q < S.sn_col.size(); q++) sg[q] = S.dsigns[S.sn_col[q]];
                if ((rc = upload(&sn_sg, sg, sg.size()))) return rc;
            }
            if ((rc = alloc(&sn_cnt, (size_t)nsn + 1))) return rc;
            CHIP_HIP(hipMemset(sn_cnt, 0, ((size_t)nsn + 1) * sizeof(int)));
        }
        if ((rc = upload(&Rf_p, S.Rf_p, S.Rf_p.size()))) return rc;
        if ((rc = upload(&Rf_col, S.Rf_col, S.Rf_col.size()))) return rc;
        if ((rc = upload(&Rf_pos, S.Rf_pos, S.Rf_pos.size()))) return rc;
        if ((rc = upload(&upd_slot, S.upd_slot, S.upd_slot.size()))) return rc;
        {
            std::vector<long long> up(S.upd_ptr.begin(), S.upd_ptr.end());
            if ((rc = upload(&upd_ptr, up, up.size()))) return rc;
        }
        asm_lvl_ptr = S.asm_lvl_ptr;
        // (the assembled form of the ancestor updates is a mode -- CHIP_DETERMINISTIC / CHIP_EXTEND_ASM_MIN, read when the
        // handle is created --: without it neither its tables nor its buffer, 0.9 GB on config 5, are put on the device)
        if (!S.asm_tgt.empty() && (switches().deterministic || switches().extend_asm_min > 0)) {
            std::vector<long long> src(S.asm_src.size() * 3), uoff(S.asm_uoff.begin(), S.asm_uoff.end());
            for (size_t q = 0; q < S.asm_src.size(); q++) {
                src[3 * q] = S.asm_src[q].uo;
                src[3 * q + 1] = S.asm_src[q].so;
                src[3 * q + 2] = (long long)(unsigned)S.asm_src[q].cnt | ((long long)S.asm_src[q].dofs << 32);
            }
            if ((rc = upload(&asm_src, src, src.size()))) return rc;
            if ((rc = upload(&asm_uoff, uoff, uoff.size()))) return rc;
            if ((rc = upload(&asm_doff, S.asm_doff, S.asm_doff.size()))) return rc;
            if ((rc = upload(&asm_tgt, S.asm_tgt, S.asm_tgt.size()))) return rc;
            if ((rc = upload(&asm_src_ptr, S.asm_src_ptr, S.asm_src_ptr.size()))) return rc;
            if ((rc = alloc(&asm_U, (size_t)S.asm_usize + 1))) return rc;
            if ((rc = alloc(&asm_Ud, (size_t)S.asm_dsize + 1))) return rc;
            CHIP_HIP(hipMemset(asm_U, 0, ((size_t)S.asm_usize + 1) * sizeof(double)));
        }
        {
            std::vector<i32> bp((size_t)nsn + 1, 0);
            for (int sn = 0; sn < nsn; sn++) bp[sn + 1] = bp[sn] + (S.sn_ptr[sn + 1] - S.sn_ptr[sn] + 63) / 64;
            if ((rc = upload(&sn_blk_ptr, bp, bp.size()))) return rc;
            // message slots of the pipelined substitution (k_snode_tri): 64 x 16 bytes per block, epoch 0 = never written
            if ((rc = alloc(&sn_flags, ((size_t)bp[nsn] + 1) * 256))) return rc;
            CHIP_HIP(hipMemset(sn_flags, 0, ((size_t)bp[nsn] + 1) * 256 * sizeof(int)));
        }
        for (int sn = 0; sn < nsn; sn++) {
            const i32 e = S.sn_col[S.sn_ptr[sn + 1] - 1];
            const double w = S.sn_ptr[sn + 1] - S.sn_ptr[sn], nb = S.Lp[e + 1] - S.Lp[e], h = w + nb;
            for (int b = 1; b * 64 < (int)w; b++) {
                const double j0 = 64.0 * b, nc = std::min(64.0, w - j0);
                sn_model[0] += 2.0 * (h - j0) * nc * j0;
            }
            sn_model[1] += w * (w - 1) / 2 + w * nb;
            sn_model[2] += 2.0 * (nb * (nb + 1) / 2) * w;
            for (int b = 0; b * 64 < (int)w; b++) {
                const double j0 = 64.0 * b, nc = std::min(64.0, w - j0);
                sn_model[3] += nc * nc * nc / 3.0 + (h - j0 - nc) * nc * nc;
            }
        }
        sn_model[4] = nsn;
        sn_lvl_ptr = S.sn_lvl_ptr;
        sn_lvl_nblk = S.sn_lvl_nblk;
        sn_lvl_hmax = S.sn_lvl_hmax;
        sn_lvl_nbmax = S.sn_lvl_nbmax;
        sn_wmax = 0;
        for (int sn = 0; sn < nsn; sn++) sn_wmax = std::max(sn_wmax, S.sn_ptr[sn + 1] - S.sn_ptr[sn]);
        sn_nbmax = 0;
        for (int l = 0; l < S.nfaclevels; l++) sn_nbmax = std::max(sn_nbmax, S.sn_lvl_nbmax[l]);
        if (dev::snode_kernel_attributes(sn_wmax, sn_nbmax) != 0) {
            set_error("k_factor_snode: dynamic LDS size rejected");
            return CHIP_ERR_HIP;
        }
        // ---- one-pass substitution matrices G = [I; L_B] T^-1 (snode_g.hip) for the unit levels whose supernodes are
        //      all of moderate width: a sweep through such a level is one launch without a chain of block hops
        sn_lvl_g.assign((size_t)S.nfaclevels, 0);
        sn_lvl_wmax.assign((size_t)S.nfaclevels, 0);
        {
            const int gmaxw = std::min(dev::snode_g_max_width(), switches().sn_g_maxw > 0 ? switches().sn_g_maxw : dev::snode_g_max_width());
            const int gmaxh = 6144; // (rows: the backward kernel keeps [D^-1 y_S; -x_B] in LDS)
            std::vector<long long> goff((size_t)nsn + 1, -1);
            std::vector<i32> tasks;
            long long total = 0;
            int ghmax = 0;
            for (int l = 0; l < S.nfaclevels && !switches().no_snode_g; l++) {
                bool ok = S.sn_lvl_ptr[l + 1] > S.sn_lvl_ptr[l];
                int wl = 0;
                for (int u = S.sn_lvl_ptr[l]; u < S.sn_lvl_ptr[l + 1]; u++) {
                    const i32 sn = S.sn_order[u], e = S.sn_col[S.sn_ptr[sn + 1] - 1];
                    const int w = S.sn_ptr[sn + 1] - S.sn_ptr[sn], h = w + (S.Lp[e + 1] - S.Lp[e]);
                    ok = ok && w <= gmaxw && h <= gmaxh;
                    wl = std::max(wl, w);
                }
                sn_lvl_wmax[l] = wl;
                if (!ok) continue;
                sn_lvl_g[l] = 1;
                for (int u = S.sn_lvl_ptr[l]; u < S.sn_lvl_ptr[l + 1]; u++) {
                    const i32 sn = S.sn_order[u], e = S.sn_col[S.sn_ptr[sn + 1] - 1];
                    const int w = S.sn_ptr[sn + 1] - S.sn_ptr[sn], h = w + (S.Lp[e + 1] - S.Lp[e]);
                    goff[sn] = total;
                    total += dev::snode_g_ld(h) * (long long)w;
                    ghmax = std::max(ghmax, h);
                    for (int r0 = 0; r0 < h; r0 += 256) {
                        tasks.push_back(u); // (the record's index in sn_order)
                        tasks.push_back(r0);
                    }
                }
            }
            size_t mem_free = 0, mem_total = 0;
            if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) {
                (void)hipGetLastError();
                mem_free = 0;
            }
            // (16 GB of G, or more than a third of what is free on the device -- a second solve context and the vectors
            // still have to fit: keep the pipelined substitution)
            if (total > (1ll << 31) || (mem_free > 0 && (unsigned long long)total * sizeof(double) > mem_free / 3)) {
                std::fill(sn_lvl_g.begin(), sn_lvl_g.end(), 0);
                std::fill(goff.begin(), goff.end(), -1);
                tasks.clear();
                total = 0;
            }
            if (total > 0) {
                // the widest (longest-running) rows of the build first
                const size_t nt = tasks.size() / 2;
                std::vector<size_t> ord(nt);
                for (size_t k = 0; k < nt; k++) ord[k] = k;
                auto wof = [&](size_t k) {
                    const i32 sn = S.sn_order[(size_t)tasks[2 * k]];
                    return S.sn_ptr[sn + 1] - S.sn_ptr[sn];
                };
                std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return wof(a) > wof(b); });
                std::vector<i32> sorted(tasks.size());
                for (size_t k = 0; k < nt; k++) {
                    sorted[2 * k] = tasks[2 * ord[k]];
                    sorted[2 * k + 1] = tasks[2 * ord[k] + 1];
                }
                if ((rc = upload(&sn_g_tasks, sorted, sorted.size()))) return rc;
                sn_g_ntasks = (int)nt;
                if ((rc = upload(&sn_g_off, goff, goff.size()))) return rc;
                { // the offsets also travel in the supernodes' records (snode_common.hpp: SnodeGeom::goff)
                    std::vector<i32> rec(S.sn_order.size() * 8 + 8, 0);
                    CHIP_HIP(hipMemcpy(rec.data(), sn_order, rec.size() * sizeof(i32), hipMemcpyDeviceToHost));
                    for (size_t k = 0; k < S.sn_order.size(); k++) {
                        const long long o = goff[(size_t)S.sn_order[k]];
                        rec[8 * k + 6] = (i32)(unsigned)(o & 0xffffffffll);
                        rec[8 * k + 7] = (i32)(unsigned)((unsigned long long)o >> 32);
                    }
                    CHIP_HIP(hipMemcpy(sn_order, rec.data(), rec.size() * sizeof(i32), hipMemcpyHostToDevice));
                }
                if (alloc(&sn_Gx, (size_t)total + 8) != CHIP_OK) {
                    // no room for G after all: the handle works without it (pipelined substitution everywhere)
                    (void)hipGetLastError();
                    sn_Gx = nullptr;
                    std::fill(sn_lvl_g.begin(), sn_lvl_g.end(), 0);
                    sn_g_ntasks = 0;
                    std::vector<i32> rec(S.sn_order.size() * 8 + 8, 0);
                    CHIP_HIP(hipMemcpy(rec.data(), sn_order, rec.size() * sizeof(i32), hipMemcpyDeviceToHost));
                    for (size_t k = 0; k < S.sn_order.size(); k++) rec[8 * k + 6] = rec[8 * k + 7] = -1;
                    CHIP_HIP(hipMemcpy(sn_order, rec.data(), rec.size() * sizeof(i32), hipMemcpyHostToDevice));
                    total = 0;
                }
                if (total > 0) {
                CHIP_HIP(hipMemset(sn_Gx, 0, ((size_t)total + 8) * sizeof(double))); // (entries above a row's diagonal block are never written)
                if ((rc = alloc(&sn_yt, n))) return rc;
                CHIP_HIP(hipMemset(sn_yt, 0, n * sizeof(double)));
                if ((rc = alloc(&gs_ctl, (size_t)dev::ir_ctl_ints()))) return rc;
                CHIP_HIP(hipMemset(gs_ctl, 0, (size_t)dev::ir_ctl_ints() * sizeof(int)));
                sn_g_entries = (double)total;
                if (dev::snode_g_attributes(ghmax) != 0) {
                    set_error("k_snode_ginv: dynamic LDS size rejected");
                    return CHIP_ERR_HIP;
                }
                } // (G allocated)
            }
        }
    }
    {
        int *bp = nullptr, *lp = nullptr, *lv = nullptr;
        if ((rc = upload(&bp, S.bundle_ptr, S.bundle_ptr.size()))) return rc;
        if ((rc = upload(&lp, S.blvl_ptr, S.blvl_ptr.size()))) return rc;
        if ((rc = upload(&lv, S.blvl, S.blvl.size()))) return rc;
        bundles.nb = S.bundle_ptr.empty() ? 0 : (int)S.bundle_ptr.size() - 1;
        bundles.bundle_ptr = bp;
        bundles.blvl_ptr = lp;
        bundles.blvl = lv;
        bundles.max_nodes = S.max_bundle_nodes;
        bundles.max_levels = S.max_bundle_levels;
        if (!S.sLi16.empty()) {
            if ((rc = upload(&sLi16, S.sLi16, S.sLi16.size()))) return rc;
            if ((rc = upload(&sLj16, S.sLj16, S.sLj16.size()))) return rc;
            // ... and the entry-parallel bundle factorisation (k_bundle_factor_flat) when its records were built and a
            // bundle's L and D fit the LDS of half a CU (config 2: 645 bundles, 23 M records)
            if (!S.fu_rec.empty() && !switches().no_factor_lds) {
                int need = 0;
                for (int b = 0; b < bundles.nb; b++) {
                    const int s0 = S.bundle_ptr[b], s1 = S.bundle_ptr[b + 1];
                    need = std::max(need, (int)(S.Lp[s1] - S.Lp[s0]) + (s1 - s0) + 1); // (+ 1: the folded top row's share)
                }
                if (dev::bundle_factor_lds_ok(need)) {
                    if ((rc = upload(&fu_rec, S.fu_rec, S.fu_rec.size()))) return rc;
                    if ((rc = upload(&fu_slot, S.fu_slot, S.fu_slot.size()))) return rc;
                    if ((rc = upload(&fu_ptr, S.fu_ptr, S.fu_ptr.size()))) return rc;
                    if ((rc = upload(&Urow16, S.Urow16, S.Urow16.size()))) return rc;
                    factor_lds_doubles = need;
                }
            }
        }
    }
    if (S.nfold > 0) {
        int *a1 = nullptr, *a2 = nullptr, *a3 = nullptr, *a4 = nullptr, *a5 = nullptr;
        double *ts = nullptr;
        if ((rc = upload(&a1, S.fold_rseg, S.fold_rseg.size()))) return rc;
        if ((rc = upload(&a2, S.fold_tt, S.fold_tt.size()))) return rc;
        if ((rc = upload(&a3, S.fold_sp, S.fold_sp.size()))) return rc;
        if ((rc = upload(&a4, S.fold_scol, S.fold_scol.size()))) return rc;
        if ((rc = upload(&a5, S.fold_sslot, S.fold_sslot.size()))) return rc;
        const size_t nacc = (size_t)dev::fold_acc_index(3, 0, 0);
        if ((rc = alloc(&ts, nacc))) return rc;
        CHIP_HIP(hipMemset(ts, 0, nacc * sizeof(double)));
        fold.k = S.nfold;
        fold.NF = S.NF;
        fold.rseg = a1;
        fold.tt = a2;
        fold.sp = a3;
        fold.scol = a4;
        fold.sslot = a5;
        fold.acc = ts;
        if (S.nfold == 1 && !S.dsigns.empty()) h_top_sign = (int)S.dsigns[(size_t)S.NF];
    }
    if (S.topblk > 0) {
        int *rs = nullptr, *ls = nullptr;
        double *T = nullptr;
        if ((rc = upload(&rs, S.Rsplit, S.Rsplit.size()))) return rc;
        if ((rc = upload(&ls, S.Lsplit, S.Lsplit.size()))) return rc;
        topblk.w = S.topblk;
        topblk.NF = S.NF;
        topblk.N = S.N;
        topblk.nblocks = (S.N - S.NF + S.topblk - 1) / S.topblk;
        if ((rc = alloc(&T, (size_t)topblk.nblocks * S.topblk * (S.topblk - 1) / 2))) return rc;
        topblk.Rsplit = rs;
        topblk.Lsplit = ls;
        topblk.T = T;
        double *ysb = nullptr;
        int *cnt = nullptr;
        if ((rc = alloc(&ysb, (size_t)topblk.nblocks * S.topblk))) return rc;
        if ((rc = alloc(&cnt, (size_t)topblk.nblocks))) return rc;
        CHIP_HIP(hipMemset(cnt, 0, (size_t)topblk.nblocks * sizeof(int)));
        topblk.ys = ysb;
        topblk.counters = cnt;
    }
    bool need_ring = false;
    const bool grouped = S.gf_ng > 0;
    if (bundles.nb > 0 && nsn == 0 && topblk.nblocks == 0 && (fold.k > 0 || grouped || NF == N) && !S.Li16.empty() &&
        !switches().no_fused_ir) {
        int cap = dev::bundle_ir_capacity(bundles, &ir_tw);
        if (cap > 0) {
            // the residual's "split" form (bundle_symv.hpp: bundle_symv_split) needs nloc + max(0, nloc - 2 nleaf) doubles
            // of LDS per bundle: taken when the larger slice leaves the co-resident grid and the workgroup size as
            // they are
            int need = 0;
            for (int b = 0; b < bundles.nb; b++) {
                const int s0 = S.bundle_ptr[b], nloc = S.bundle_ptr[b + 1] - s0, nleaf = S.blvl[S.blvl_ptr[b] + 1] - s0;
                need = std::max(need, nloc + std::max(0, nloc - 2 * nleaf));
            }
            dev::BundleView trial = bundles;
            trial.ir_lds_doubles = need;
            trial.symv_split = 1;
            int tw2 = 0;
            const int cap2 = dev::bundle_ir_capacity(trial, &tw2);
            if (tw2 == ir_tw && cap2 >= std::min(cap, bundles.nb)) {
                bundles = trial;
                cap = cap2;
            } else {
                cap = dev::bundle_ir_capacity(bundles, &ir_tw); // (restores the kernels' LDS attribute)
            }
        }
        if (cap > 0 && ((fold.k == 0 && !grouped) || bundles.nb <= cap)) {
            ir_fused = true;
            if ((rc = upload(&Li16, S.Li16, S.Li16.size()))) return rc;
            if ((rc = upload(&Ucol16, S.Ucol16, S.Ucol16.size()))) return rc;
            if ((rc = upload(&Lj16, S.Lj16, S.Lj16.size()))) return rc;
            if ((rc = upload(&Urow16, S.Urow16, S.Urow16.size()))) return rc;
            if ((rc = upload(&Rk16, S.Rk16, S.Rk16.size()))) return rc;
            if ((rc = upload(&Ro16, S.Ro16, S.Ro16.size()))) return rc;
            if (!switches().no_factor_lds) {
                // the bundle factorisation with its L and D values in LDS: every bundle's entries must be addressable
                // in 16 bits and two workgroups must fit a CU
                int need = 0;
                bool ok = true;
                for (int b = 0; b < bundles.nb && ok; b++) {
                    const int s0 = S.bundle_ptr[b], s1 = S.bundle_ptr[b + 1];
                    const int ne = S.Lp[s1] - S.Lp[s0];
                    ok = ne < 65535;
                    need = std::max(need, ne + (s1 - s0) + 1); // (+ 1: the folded top row's share, k_bundle_factor_flat)
                }
                if (ok && dev::bundle_factor_lds_ok(need)) factor_lds_doubles = need;
                if (factor_lds_doubles > 0 && !S.fu_rec.empty()) {
                    if ((rc = upload(&fu_rec, S.fu_rec, S.fu_rec.size()))) return rc;
                    if ((rc = upload(&fu_slot, S.fu_slot, S.fu_slot.size()))) return rc;
                    if ((rc = upload(&fu_ptr, S.fu_ptr, S.fu_ptr.size()))) return rc;
                }
            }
            ir_grid = std::min(bundles.nb, cap);
            ir_ctl_len = (size_t)dev::ir_ctl_ints() + (grouped ? (size_t)32 * S.gf_ng : 0);
            if ((rc = alloc(&ir_ctl, ir_ctl_len))) return rc;
            CHIP_HIP(hipMemset(ir_ctl, 0, ir_ctl_len * sizeof(int)));
            if ((rc = alloc(&ir_rel, (size_t)dev::ir_rel_ints()))) return rc;
            CHIP_HIP(hipMemset(ir_rel, 0, (size_t)dev::ir_rel_ints() * sizeof(int)));
            need_ring = true; // (ir_res = the ring inside the mailbox, set once that is allocated)
            if ((rc = alloc(&ir_part, dev::ir_part_doubles(bundles.nb, fold.k)))) return rc;
            if (grouped) {
                // grouped fold: known to k_bundle_ir / k_bundle_factor / k_gfold_top_factor only (everything else
                // treats the top as an ordinary level-scheduled top: the unfused path stays valid as a fallback)
                int *a1 = nullptr, *a2 = nullptr, *a3 = nullptr, *a4 = nullptr, *a5 = nullptr, *a6 = nullptr, *a7 = nullptr,
                    *a8 = nullptr;
                std::vector<i32> bgrp((size_t)bundles.nb, -1);
                for (int g = 0; g < S.gf_ng; g++)
                    for (int b = S.gf_bptr[g]; b < S.gf_bptr[g + 1]; b++) bgrp[b] = g;
                if ((rc = upload(&a1, S.gf_ptr, S.gf_ptr.size()))) return rc;
                if ((rc = upload(&a2, S.gf_node, S.gf_node.size()))) return rc;
                if ((rc = upload(&a3, S.gf_bptr, S.gf_bptr.size()))) return rc;
                if ((rc = upload(&a4, bgrp, bgrp.size()))) return rc;
                if ((rc = upload(&a5, S.gf_tt, S.gf_tt.size()))) return rc;
                if ((rc = upload(&a6, S.gf_sp, S.gf_sp.size()))) return rc;
                if ((rc = upload(&a7, S.gf_scol, S.gf_scol.size()))) return rc;
                if ((rc = upload(&a8, S.gf_sslot, S.gf_sslot.size()))) return rc;
                double *fsh = nullptr, *rsh = nullptr, *rec = nullptr, *fac = nullptr;
                if ((rc = alloc(&fsh, (size_t)bundles.nb * 8))) return rc;
                if ((rc = alloc(&rsh, (size_t)bundles.nb * 16))) return rc;
                if ((rc = alloc(&rec, (size_t)S.gf_ng * 64))) return rc;
                if ((rc = alloc(&fac, (size_t)bundles.nb * 36))) return rc;
                CHIP_HIP(hipMemset(rec, 0, (size_t)S.gf_ng * 64 * sizeof(double)));
                CHIP_HIP(hipMemset(fac, 0, (size_t)bundles.nb * 36 * sizeof(double)));
                gfold.ng = S.gf_ng;
                gfold.ptr = a1;
                gfold.node = a2;
                gfold.bptr = a3;
                gfold.bgrp = a4;
                gfold.tt = a5;
                gfold.sp = a6;
                gfold.scol = a7;
                gfold.sslot = a8;
                gfold.fsh = fsh;
                gfold.rsh = rsh;
                gfold.rec = rec;
                gfold.fac = fac;
                gfold.gcnt = ir_ctl + dev::ir_ctl_ints();
            }
            if (grouped && !switches().no_step_kernel) {
                // the step kernels (bundle_gstep.hip): the bundles' entries of L and K in "gs order" (kernels.hpp:
                // GStepView), kept in registers by the kernels; taken when every bundle fits the register slots compiled
                const int nbn = bundles.nb;
                i64 mxE = 0, mxU = 0, mxN = 0, mxL = 0;
                for (int b = 0; b < nbn; b++) {
                    const int s0 = S.bundle_ptr[b], s1 = S.bundle_ptr[b + 1];
                    mxE = std::max<i64>(mxE, S.Lp[s1] - S.Lp[s0]);
                    mxU = std::max<i64>(mxU, S.Up[s1] - S.Up[s0]);
                    mxN = std::max<i64>(mxN, s1 - s0);
                    mxL = std::max<i64>(mxL, S.blvl_ptr[b + 1] - S.blvl_ptr[b] - 1);
                }
                dev::GStepView g{};
                g.lr = (int)((mxE + 255) / 256);
                g.ur = (int)((mxU + 255) / 256);
                g.nr = (int)((mxN + 255) / 256);
                if (mxL <= dev::GS_MAXL && mxE < 65535 && mxU < 65535 && g.lr <= 8 && g.ur <= 10 && g.nr <= 4) {
                    const i32 NFn = S.NF;
                    std::vector<i32> desc((size_t)nbn * dev::GS_DESC, 0), gtop((size_t)std::max(1, (int)S.gf_ng) * dev::GS_GTOP, -1);
                    std::vector<i32> bgrp_h((size_t)nbn, -1);
                    for (int g = 0; g < S.gf_ng; g++) {
                        for (int b = S.gf_bptr[g]; b < S.gf_bptr[g + 1]; b++) bgrp_h[(size_t)b] = g;
                        i32 *gt = gtop.data() + (size_t)g * dev::GS_GTOP;
                        const int base = S.gf_ptr[g], kk = S.gf_ptr[g + 1] - base;
                        for (int t = 0; t < kk; t++) {
                            gt[t] = S.gf_node[(size_t)base + t];
                            gt[8 + t] = h_perm[(size_t)gt[t]];
                            gt[144 + t] = S.dsigns.empty() ? 1 : (int)S.dsigns[(size_t)gt[t]];
                            for (int j = 0; j < 8; j++) gt[16 + t * 8 + j] = S.gf_tt[(size_t)g * 64 + t * 8 + j];
                            for (i32 q = S.gf_sp[(size_t)base + t]; q < S.gf_sp[(size_t)base + t + 1]; q++)
                                gt[80 + t * 8 + S.gf_scol[(size_t)q]] = S.gf_sslot[(size_t)q];
                        }
                    }
                    // (+ 8: the kernels clamp their indices instead of predicating their loads)
                    std::vector<uint16_t> lsrc((size_t)S.Lp[NFn] + 8, 0), usrc((size_t)S.Up[NFn] + 8, 0), ufs((size_t)S.Up[NFn] + 8, 0);
                    std::vector<uint32_t> lij((size_t)S.Lp[NFn] + 8, 0), uij((size_t)S.Up[NFn] + 8, 0);
                    std::vector<std::pair<uint32_t, i32>> tops; // (top index << 16 | column or row, source)
                    int nlead = 0;
                    for (int b = 0; b < nbn; b++) {
                        const int s0 = S.bundle_ptr[b], s1 = S.bundle_ptr[b + 1], nloc = s1 - s0;
                        const i32 e0 = S.Lp[s0], ub = S.Up[s0];
                        const i32 *lv = S.blvl.data() + S.blvl_ptr[b];
                        const int nl = S.blvl_ptr[b + 1] - S.blvl_ptr[b] - 1;
                        i32 *dsc = desc.data() + (size_t)b * dev::GS_DESC;
                        const int g = bgrp_h[(size_t)b];
                        dsc[0] = s0, dsc[1] = nloc, dsc[2] = e0, dsc[3] = S.Lp[s1] - e0, dsc[4] = ub, dsc[5] = S.Up[s1] - ub;
                        dsc[6] = nl, dsc[7] = g;
                        dsc[8] = g >= 0 ? S.gf_ptr[g] : 0, dsc[9] = g >= 0 ? S.gf_ptr[g + 1] - S.gf_ptr[g] : 0;
                        dsc[10] = g >= 0 ? S.gf_bptr[g] : b, dsc[11] = g >= 0 ? S.gf_bptr[g + 1] - S.gf_bptr[g] : 1;
                        if (dsc[10] == b) nlead++; // (a group's first bundle -- or a bundle without group -- leads)
                        dsc[14] = nlead - 1;
                        dsc[13] = S.blvl_ptr[b];
                        for (int l = 0; l <= nl; l++) dsc[48 + l] = lv[l] - s0;
                        if (!S.fu_ptr.empty())
                            for (int l = 0; l <= nl; l++) dsc[32 + l] = S.fu_ptr[(size_t)S.blvl_ptr[b] + l];
                        i32 p = 0;
                        tops.clear();
                        for (int l = 0; l < nl; l++) {
                            dsc[16 + l] = p;
                            for (i32 j = lv[l]; j < lv[l + 1]; j++)
                                for (i32 q = S.Lp[j]; q < S.Lp[j + 1]; q++) {
                                    const uint32_t r16 = S.Li16[(size_t)q];
                                    if ((int)r16 < nloc) {
                                        lsrc[(size_t)e0 + p] = (uint16_t)(q - e0);
                                        lij[(size_t)e0 + p] = (r16 << 16) | (uint32_t)(j - s0);
                                        p++;
                                    } else {
                                        tops.push_back({((r16 - (uint32_t)nloc) << 16) | (uint32_t)(j - s0), q});
                                    }
                                }
                        }
                        dsc[16 + nl] = p;
                        std::sort(tops.begin(), tops.end());
                        for (const auto &t : tops) {
                            lsrc[(size_t)e0 + p] = (uint16_t)(t.second - e0);
                            lij[(size_t)e0 + p] = (((t.first >> 16) + (uint32_t)nloc) << 16) | (t.first & 0xFFFFu);
                            p++;
                        }
                        dsc[16 + nl + 1] = p; // == S.Lp[s1] - e0
                        // the U rows: entries outside the top columns in their stored order, then the top columns'
                        i32 pu = 0;
                        tops.clear();
                        for (i32 u = ub; u < S.Up[s1]; u++) {
                            const uint32_t r16 = S.Urow16[(size_t)u], c16 = S.Ucol16[(size_t)u];
                            if ((int)c16 < nloc) {
                                usrc[(size_t)ub + pu] = (uint16_t)(u - ub);
                                uij[(size_t)ub + pu] = (r16 << 16) | c16;
                                pu++;
                            } else {
                                tops.push_back({((c16 - (uint32_t)nloc) << 16) | r16, u});
                            }
                        }
                        dsc[12] = pu;
                        std::sort(tops.begin(), tops.end());
                        for (const auto &t : tops) {
                            usrc[(size_t)ub + pu] = (uint16_t)(t.second - ub);
                            uij[(size_t)ub + pu] = ((t.first & 0xFFFFu) << 16) | ((t.first >> 16) + (uint32_t)nloc);
                            pu++;
                        }
                        if (!S.fu_slot.empty())
                            for (i32 q = 0; q < pu; q++) ufs[(size_t)ub + q] = S.fu_slot[(size_t)ub + usrc[(size_t)ub + q]];
                    }
                    for (int b = 0; b < nbn; b++) desc[(size_t)b * dev::GS_DESC + 15] = nlead;
                    int *d_desc = nullptr, *d_gtop = nullptr, *d_msg = nullptr, *d_fmsg = nullptr, *d_lmsg = nullptr;
                    unsigned short *d_lsrc = nullptr, *d_usrc = nullptr, *d_ufs = nullptr;
                    double *d_gsl = nullptr, *d_gsu = nullptr;
                    unsigned int *d_lij = nullptr, *d_uij = nullptr;
                    if ((rc = upload(&d_desc, desc, desc.size()))) return rc;
                    if ((rc = upload(&d_gtop, gtop, gtop.size()))) return rc;
                    if ((rc = upload(&d_lsrc, lsrc, lsrc.size()))) return rc;
                    if ((rc = upload(&d_usrc, usrc, usrc.size()))) return rc;
                    if ((rc = upload(&d_ufs, ufs, ufs.size()))) return rc;
                    if ((rc = alloc(&d_gsl, lsrc.size()))) return rc;
                    if ((rc = alloc(&d_gsu, usrc.size()))) return rc;
                    if ((rc = upload(&d_lij, lij, lij.size()))) return rc;
                    if ((rc = upload(&d_uij, uij, uij.size()))) return rc;
                    const size_t nmsg = (size_t)nbn * 4 * dev::GS_MV * 4, nfmsg = (size_t)nbn * 36 * 4, nlmsg = (size_t)nlead * 2 * 2 * 4;
                    if ((rc = alloc(&d_msg, nmsg))) return rc;
                    if ((rc = alloc(&d_fmsg, nfmsg))) return rc;
                    CHIP_HIP(hipMemset(d_msg, 0, nmsg * sizeof(int)));
                    CHIP_HIP(hipMemset(d_fmsg, 0, nfmsg * sizeof(int)));
                    if ((rc = alloc(&d_lmsg, nlmsg))) return rc;
                    CHIP_HIP(hipMemset(d_lmsg, 0, nlmsg * sizeof(int)));
                    g.desc = d_desc;
                    g.gtop = d_gtop;
                    g.lsrc = d_lsrc;
                    g.usrc = d_usrc;
                    g.ufs = d_ufs;
                    g.gsl = d_gsl;
                    g.gsu = d_gsu;
                    g.lij = d_lij;
                    g.uij = d_uij;
                    g.msg = d_msg;
                    g.fmsg = d_fmsg;
                    g.lmsg = d_lmsg;
                    g.epoch = 0;
                    gstep = g;
                    gstep_solve_on = dev::gstep_solve_capacity(bundles, gstep) >= bundles.nb;
                    bundles.max_entries = (int)mxE;
                    gstep_factor_on = fu_rec != nullptr && dev::gstep_factor_ok(bundles);
                }
            }
        }
    }
    if ((rc = alloc(&mb_dev, 1))) return rc;
    CHIP_HIP(hipMemset(mb_dev, 0, sizeof(Mailbox)));
    CHIP_HIP(hipHostMalloc((void **)&mb_host, sizeof(Mailbox), hipHostMallocDefault));
    std::memset(mb_host, 0, sizeof(Mailbox));
    if (need_ring) {
        ir_res = mb_dev->ring; // (address arithmetic only)
        ir_res_host = mb_host->ring;
    }
    if ((rc = alloc(&dslot_dev, (size_t)2 * NRM_SET_WORDS))) return rc;
    CHIP_HIP(hipMemset(dslot_dev, 0, (size_t)2 * NRM_SET_WORDS * sizeof(unsigned long long)));
    // fast preparation (refactor_enqueue): the grouped-fold step factorisation, or an arrow with ONE top column whose
    // only top-top entry of K is its diagonal, factored by the flat bundle kernel
    fast_prep_ok = gstep_factor_on || (ir_fused && fold.k == 1 && factor_lds_doubles > 0 && fu_rec != nullptr && nfill == 0 && fill_from < 0 &&
                                       nnzK - nnzU == 1 && nsn == 0);
    if (fast_prep_ok && fold.k == 1) {
        if ((rc = alloc(&fold_cnt, (size_t)32))) return rc;
        CHIP_HIP(hipMemset(fold_cnt, 0, 32 * sizeof(int)));
    }
    if ((rc = alloc(&nrm_dev, (size_t)NRM_SETS * NRM_SET_WORDS))) return rc;
    CHIP_HIP(hipMemset(nrm_dev, 0, (size_t)NRM_SETS * NRM_SET_WORDS * sizeof(unsigned long long)));
    CHIP_HIP(hipHostMalloc((void **)&nrm_host, 3 * NRM_SET_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
    return CHIP_OK;
}

dev::LdlView Engine::view() const {
    dev::LdlView v{};
    v.N = N;
    v.nnzL = (int)nnzL;
    v.Lp = Lp;
    v.Li = Li;
    v.Rp = Rp;
    v.Rcol = Rcol;
    v.Rpos = Rpos;
    v.Tpos = Tpos;
    v.Lx = Lx;
    v.Rx = Rx;
    v.D = D;
    v.Dinv = Dinv;
    v.dsigns = dsigns;
    v.status = mb_dev->status; // address arithmetic only, never dereferenced on the host
    v.reg_eps = st.dynamic_regularization_eps;
    v.reg_delta = st.dynamic_regularization_delta;
    v.Up = Up;
    v.Ucol = Ucol;
    v.Ux = Ux;
    v.eps_ptr = nullptr;
    v.Li16 = Li16;
    v.Lj16 = Lj16;
    v.Urow16 = Urow16;
    v.Rk16 = Rk16;
    v.fu_rec = fu_rec;
    v.fu_slot = fu_slot;
    v.fu_ptr = fu_ptr;
    v.Ro16 = Ro16;
    v.Ucol16 = Ucol16;
    v.mirror_rows = ir_fused ? 0 : 1;
    v.sLi16 = sLi16;
    v.sLj16 = sLj16;
    return v;
}

void Engine::prof_begin(int family) {
    if (family != prof_family) return;
    if (prof_used + 2 > prof_events.size()) {
        for (int i = 0; i < 64; i++) {
            hipEvent_t ev;
            if (hipEventCreate(&ev) != hipSuccess) return;
            prof_events.push_back(ev);
        }
    }
    (void)hipEventRecord(prof_events[prof_used], stream);
}
// the next event pair of the active family for a launch that stamps its own events
// (hipExtLaunchKernelGGL: no launch gap inside the measured interval); nullptrs when inactive
void Engine::prof_pair(int family, hipEvent_t *ev0, hipEvent_t *ev1) {
    if (family != prof_family) return;
    if (prof_used >= 8192) prof_collect();
    if (prof_used + 2 > prof_events.size()) {
        for (int i = 0; i < 64; i++) {
            hipEvent_t ev;
            if (hipEventCreate(&ev) != hipSuccess) return;
            prof_events.push_back(ev);
        }
    }
    *ev0 = prof_events[prof_used];
    *ev1 = prof_events[prof_used + 1];
    prof_used += 2;
}
void Engine::prof_end(int family) {
    if (family != prof_family) return;
    if (prof_used + 2 > prof_events.size()) return;
    (void)hipEventRecord(prof_events[prof_used + 1], stream);
    prof_used += 2;
    if (prof_used >= 8192) prof_collect();
}
dev::LaunchProf Engine::launch_prof() {
    return dev::LaunchProf{[](void *c, int f) { ((Engine *)c)->prof_begin(f); },
                           [](void *c, int f) { ((Engine *)c)->prof_end(f); }, this};
}
void Engine::prof_collect() {
    if (!prof_used) return;
    (void)hipEventSynchronize(prof_events[prof_used - 1]);
    for (size_t i = 0; i + 1 < prof_used; i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, prof_events[i], prof_events[i + 1]) == hipSuccess) {
            prof_ms_total += ms;
            prof_launches++;
        }
    }
    prof_used = 0;
}

bool Engine::sweeps_after_failure() {
    if (!gs_ctl || gs_off || gs_launches == 0) return false;
    (void)hipDeviceSynchronize();
    (void)hipMemset(gs_ctl, 0, (size_t)dev::ir_ctl_ints() * sizeof(int));
    if (alt.gs_ctl) (void)hipMemset(alt.gs_ctl, 0, (size_t)dev::ir_ctl_ints() * sizeof(int));
    (void)hipGetLastError();
    gs_off = true;
    gs_recoveries += 1;
    return true;
}
// a device-side wait that cannot be enqueued becomes a host-side one: the chain of block columns must not read member
// columns while the second stream still adds into them
static void wait_event_or_sync(hipStream_t s, hipEvent_t ev) {
    if (hipStreamWaitEvent(s, ev, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipEventSynchronize(ev);
    }
}
int Engine::wait_for_exchange() {
    if (!exch_pending) return CHIP_OK;
    exch_pending = false;
    if (hipStreamWaitEvent(stream, exch_event, 0) != hipSuccess) { // (fall back to the host: the exchange must be over)
        (void)hipGetLastError();
        CHIP_HIP(hipEventSynchronize(exch_event));
    }
    return CHIP_OK;
}
// (restartable: the stream and the event are created once; a failed set-up is remembered and not repeated on every
// refactor -- the buffers it did allocate stay with the handle's allocation list)
int Engine::ensure_alt() {
    if (alt_ready) return CHIP_OK;
    if (alt_failed) return CHIP_ERR_HIP;
    const int rc_all = ensure_alt_once();
    if (rc_all != CHIP_OK) alt_failed = true;
    return rc_all;
}
int Engine::ensure_alt_once() {
    int rc;
    if (!alt.stream) CHIP_HIP(hipStreamCreateWithFlags(&alt.stream, hipStreamNonBlocking));
    if (!pair_event) CHIP_HIP(hipEventCreateWithFlags(&pair_event, hipEventDisableTiming));
    if (!pair_ev_a) CHIP_HIP(hipEventCreateWithFlags(&pair_ev_a, hipEventDisableTiming));
    if (!pair_ev_b) CHIP_HIP(hipEventCreateWithFlags(&pair_ev_b, hipEventDisableTiming));
    const size_t n = (size_t)N;
    if (sn_yt) {
        if ((rc = alloc(&alt.sn_yt, n))) return rc;
        CHIP_HIP(hipMemset(alt.sn_yt, 0, n * sizeof(double)));
    }
    if (gs_ctl) {
        if ((rc = alloc(&alt.gs_ctl, (size_t)dev::ir_ctl_ints()))) return rc;
        CHIP_HIP(hipMemset(alt.gs_ctl, 0, (size_t)dev::ir_ctl_ints() * sizeof(int)));
    }
    if (xs_view && (rc = alloc(&alt.xs_view, n))) return rc;
    if (bt_view && (rc = alloc(&alt.bt_view, n))) return rc;
    if (dblk.P && (rc = alloc(&alt.dblk_P, (size_t)dblk.nrows * dblk.split))) return rc;
    if ((rc = alloc(&alt.nrm_dev, (size_t)NRM_SETS * NRM_SET_WORDS))) return rc;
    CHIP_HIP(hipMemset(alt.nrm_dev, 0, (size_t)NRM_SETS * NRM_SET_WORDS * sizeof(unsigned long long)));
    CHIP_HIP(hipHostMalloc((void **)&alt.nrm_host, 3 * NRM_SET_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
    if (sn_flags && nsn > 0) {
        size_t blocks = 0;
        for (int sn = 0; sn < nsn; sn++) blocks += (size_t)(h_sn_ptr[sn + 1] - h_sn_ptr[sn] + 63) / 64;
        if ((rc = alloc(&alt.sn_flags, (blocks + 1) * 256))) return rc;
        CHIP_HIP(hipMemset(alt.sn_flags, 0, (blocks + 1) * 256 * sizeof(int)));
    }
    alt_ready = true;
    return CHIP_OK;
}
void Engine::swap_ctx() {
    std::swap(stream, alt.stream);
    std::swap(sn_yt, alt.sn_yt);
    std::swap(xs_view, alt.xs_view);
    std::swap(bt_view, alt.bt_view);
    std::swap(dblk.P, alt.dblk_P);
    std::swap(nrm_dev, alt.nrm_dev);
    std::swap(nrm_host, alt.nrm_host);
    std::swap(sn_flags, alt.sn_flags);
    std::swap(gs_ctl, alt.gs_ctl);
    alt_active = !alt_active;
}
int Engine::pair_begin() {
    int rc = ensure_alt();
    if (rc) return rc;
    if (!rx_valid && rx_needed()) {
        dev::gather_values(stream, Rx, Lx, Rpos, (int)nnzR);
        rx_valid = true;
    }
    if (!fold.k && !sx_valid) {
        dev::gather_values(stream, Sx, Kx, Smap, (int)nnzS);
        sx_valid = true;
    }
    CHIP_HIP(hipEventRecord(pair_event, stream));
    CHIP_HIP(hipStreamWaitEvent(alt.stream, pair_event, 0));
    return CHIP_OK;
}

// Runs of >= 2 consecutive unit levels on the one-pass matrices: one persistent launch per run and sweep (snode_g.hip:
// k_snode_gsweep).  Forward, level l carries the row gathers of level l + 1; backward, a level qualifies when its ordinary
// columns need no chunk preparation and carries them itself -- exactly what the per-level launches do (enqueue_solve_direct).
// The grid is at most HALF the chip's compute units (one 1024-thread workgroup each): the two solves of a pair run their
// sweeps side by side on two streams, and two persistent grids must be co-resident together whatever order their
// workgroups are dispatched in.
int Engine::build_gsweeps() {
    gs_built = true;
    gs_runs.clear();
    gs_run_f.assign((size_t)nfaclevels, -1);
    gs_run_b.assign((size_t)nfaclevels, -1);
    if (sn_g_ntasks <= 0 || !gs_ctl || switches().no_sweep_merge || switches().no_sweep_persist) return CHIP_OK;
    std::vector<dev::GSweepLevel> tab;
    int ncu = 0, devid = 0;
    {
        hipDeviceProp_t prop;
        CHIP_HIP(hipGetDevice(&devid));
        CHIP_HIP(hipGetDeviceProperties(&prop, devid));
        ncu = prop.multiProcessorCount;
    }
    const int gmax = switches().gsweep_grid > 0 ? switches().gsweep_grid : std::max(1, ncu / 2);
    // a level with more blocks than two rounds of the grid stays a launch of its own: there the chip-wide grid of the plain
    // launch wins (config 5's leaf level: 200 supernodes; measured 0.3 - 0.6 ms per step slower inside a run)
    const int task_cap = switches().gsweep_grid > 0 ? (1 << 30) : 2 * gmax;
    auto is_g = [&](int l, bool fwd_dir) {
        if (!(sn_lvl_g[l] && sn_lvl_ptr[l + 1] > sn_lvl_ptr[l])) return false;
        const int gx = ((fwd_dir ? sn_lvl_hmax[l] : sn_lvl_wmax[l]) + 63) / 64;
        return (long long)gx * (sn_lvl_ptr[l + 1] - sn_lvl_ptr[l]) <= task_cap;
    };
    auto close_run = [&](dev::GatherMode m, size_t first, int start_level, int hmax) {
        const int nlev = (int)(tab.size() - first);
        if (nlev < 2) {
            tab.resize(first);
            return;
        }
        const size_t lds = dev::snode_gsweep_lds(m, hmax);
        int most = 1;
        for (size_t k = first; k < tab.size(); k++) {
            const dev::GSweepLevel &L = tab[k];
            most = std::max(most, L.gx * L.count + L.ccount + (L.wcount + 15) / 16 + (L.tcount + 1023) / 1024);
        }
        const int grid = std::min(std::min(gmax, dev::snode_gsweep_capacity(m, lds)), most);
        if (grid <= 0) {
            tab.resize(first);
            return;
        }
        (m == dev::FWD ? gs_run_f : gs_run_b)[(size_t)start_level] = (i32)gs_runs.size();
        gs_runs.push_back({nlev, (int)first, grid, lds});
    };
    for (int l = 0; l < nfaclevels;) { // forward
        if (!is_g(l, true)) {
            l++;
            continue;
        }
        const size_t first = tab.size();
        const int l0 = l;
        for (; l < nfaclevels && is_g(l, true); l++) {
            dev::GSweepLevel L{};
            L.off = sn_lvl_ptr[l];
            L.count = sn_lvl_ptr[l + 1] - sn_lvl_ptr[l];
            L.gx = (sn_lvl_hmax[l] + 63) / 64;
            if (l + 1 < nfaclevels) {
                const dev::ListView t = fwu.T(l + 1), w = fwu.W(l + 1);
                const dev::ChunkView c = fwu.B(l + 1);
                L.trows = t.idx, L.tcount = t.count, L.wrows = w.idx, L.wcount = w.count;
                L.crow = c.row, L.cbeg = c.beg, L.cend = c.end, L.ccount = c.count;
            }
            tab.push_back(L);
        }
        close_run(dev::FWD, first, l0, 0);
    }
    auto bwd_ok = [&](int l) { return is_g(l, false) && bwu.B(l).count == 0; };
    for (int l = nfaclevels - 1; l >= 0;) { // backward
        if (!bwd_ok(l)) {
            l--;
            continue;
        }
        const size_t first = tab.size();
        const int l0 = l;
        int hmax = 0;
        for (; l >= 0 && bwd_ok(l); l--) {
            dev::GSweepLevel L{};
            L.off = sn_lvl_ptr[l];
            L.count = sn_lvl_ptr[l + 1] - sn_lvl_ptr[l];
            L.gx = (sn_lvl_wmax[l] + 63) / 64;
            const dev::ListView t = bwu.T(l), w = bwu.W(l);
            L.trows = t.idx, L.tcount = t.count, L.wrows = w.idx, L.wcount = w.count;
            hmax = std::max(hmax, sn_lvl_hmax[l]);
            tab.push_back(L);
        }
        close_run(dev::BWD, first, l0, hmax);
    }
    if (tab.empty()) return CHIP_OK;
    void *p = nullptr;
    CHIP_HIP(hipMalloc(&p, tab.size() * sizeof(dev::GSweepLevel)));
    allocs.push_back(p);
    CHIP_HIP(hipMemcpy(p, tab.data(), tab.size() * sizeof(dev::GSweepLevel), hipMemcpyHostToDevice));
    gs_lv = (dev::GSweepLevel *)p;
    return CHIP_OK;
}

dev::SnodeView Engine::snode_view() const {
    dev::SnodeView sv{sn_ptr, sn_col, upd_ptr, upd_slot, sn_geo, sn_cb, sn_d, sn_sg, nullptr, sn_cnt};
    sv.Gx = sn_Gx;
    sv.g_off = sn_g_off;
    return sv;
}

void Engine::hs_direct_prefill_async() {
    if (!hs_direct_ok || fill_from < 0 || switches().no_hs_direct || switches().no_hs_prefill_async || (long long)nnzL <= fill_from ||
        hs_prefill_pending || alt_active || prof_family != PF_NONE)
        return;
    if (ensure_alt() != CHIP_OK) return;
    for (hipEvent_t &e : hs_ev)
        if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            e = nullptr;
            return;
        }
    // (L's last readers -- the previous step's solves -- are on the main stream: the clear waits for them)
    if (hipEventRecord(hs_ev[0], stream) != hipSuccess || hipStreamWaitEvent(alt.stream, hs_ev[0], 0) != hipSuccess ||
        hipMemsetAsync(Lx + fill_from, 0, (size_t)((long long)nnzL - fill_from) * sizeof(double), alt.stream) != hipSuccess ||
        hipEventRecord(hs_ev[1], alt.stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(alt.stream); // (whatever was enqueued there is done before the main stream goes on)
        return;
    }
    hs_prefill_pending = true;
}
bool Engine::hs_direct_begin() {
    if (hs_prefill_pending) { // the clear is under way on the second stream
        hs_prefill_pending = false;
        wait_event_or_sync(stream, hs_ev[1]);
        hs_direct_armed = true;
        return true;
    }
    if (!hs_direct_ok || fill_from < 0 || switches().no_hs_direct || (long long)nnzL <= fill_from) return false;
    // (the fill-in range holds the blocks' L entries: cleared BEFORE they are written, not by the refactor)
    if (hipMemsetAsync(Lx + fill_from, 0, (size_t)((long long)nnzL - fill_from) * sizeof(double), stream) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    hs_direct_armed = true;
    return true;
}
int Engine::read_mailbox() {
    CHIP_HIP(hipMemcpyAsync(mb_host, mb_dev, sizeof(Mailbox), hipMemcpyDeviceToHost, stream));
    CHIP_HIP(hipStreamSynchronize(stream));
    return CHIP_OK;
}

// qdldl.rs:188-200 (refactor) on the device.  With static_reg the +-eps shift of
// directldlkktsolver.rs:217-250 is applied while the values are scattered into
// the factor's storage; Kx itself stays unregularised (that is what the
// refinement residual must see, :255-261), so nothing has to be "restored".
int Engine::refactor(bool static_reg, const int *diag_idx_dev, double static_diag_max) {
    int rc = refactor_enqueue(static_reg, diag_idx_dev, static_diag_max);
    if (rc) return rc;
    return refactor_collect();
}
int Engine::refactor_enqueue(bool static_reg, const int *diag_idx_dev, double static_diag_max, bool status_cleared) {
    dev::LdlView v = view();
    const double *eps_ptr = nullptr;
    // fast preparation: eps from the slots inside the bundle factorisation, the top's initial values read from K there,
    // status words cleared by the cone kernel -- no eps / scatter launches (and no pivot launch for a single top column)
    // (fold.k == 1 handles: only k_bundle_factor_flat reads the slots -- with CHIP_NO_FACTOR_FLAT the launcher falls back
    // to kernels that look at eps_ptr, so the preparation launches must stay)
    const bool fast = fast_prep_ok && static_reg && !diag_idx_dev && status_cleared && !switches().no_step_kernel && !switches().no_fast_prep &&
                      (gstep_factor_on || !switches().no_factor_flat);
    dev::FoldView ffold = fold;
    if (fast) {
        v.eps_slots = diag_slots();
        v.eps_clear = dslot_dev + (size_t)(slot_parity ^ 1) * NRM_SET_WORDS;
        v.eps_c = st.static_regularization_constant;
        v.eps_prop = st.static_regularization_proportional;
        v.eps_static_max = static_diag_max;
        v.eps_out = &mb_dev->eps;
        slot_parity ^= 1; // (the next update's cone kernels write the set this launch clears)
        if (fold.k == 1) {
            ffold.top_k = Kx + nnzU;
            ffold.top_sign = h_top_sign;
        }
    } else if (static_reg) {
        if (diag_idx_dev)
            dev::diag_absmax_eps(stream, Kx, diag_idx_dev, N, st.static_regularization_constant,
                                 st.static_regularization_proportional, (double *)mb_dev);
        else // the cone kernels left the maxima of the diagonal entries they wrote in the CURRENT set of slots (the fast
             // preparation alternates between two sets: after an odd number of fast refactors that is set 1)
            dev::eps_from_slots(stream, diag_slots(), st.static_regularization_constant,
                                st.static_regularization_proportional, static_diag_max, (double *)mb_dev);
        eps_ptr = (const double *)mb_dev;
    }
    v.eps_ptr = eps_ptr;
    // entries with both ends in the top -> the top columns of L / D; clears the status words.  The bundle
    // columns take their initial values straight from the U rows inside k_bundle_factor.
    if (hs_prefill_pending) { // (a clear started for a write that did not happen: joined, and the refactor clears again)
        hs_prefill_pending = false;
        wait_event_or_sync(stream, hs_ev[1]);
    }
    const bool direct = hs_direct_armed && !fast; // (one refactor per armed write: L no longer holds the blocks afterwards)
    hs_direct_armed = false;
    if (!fast && !direct && fill_from >= 0 && (long long)nnzL > fill_from)
        CHIP_HIP(hipMemsetAsync(Lx + fill_from, 0, (size_t)((long long)nnzL - fill_from) * sizeof(double), stream));
    if (direct) hs_direct_refactors++;
    if (direct)
        dev::scatter_rest(stream, Kx + nnzU, v2l, rest_idx, nrest, (int)nnzL, Lx, D, dsigns, eps_ptr, mb_dev->status);
    else if (!fast)
        dev::scatter_init(stream, Kx + nnzU, v2l, (int)(nnzK - nnzU), (int)nnzL, Lx, D, dsigns, eps_ptr, fill_idx, nfill,
                          mb_dev->status);
    const bool top_folded = fold.k == 1 || gfold.ng > 0;
    prof_begin(PF_BFACTOR);
    if (gstep_factor_on && !switches().no_step_kernel) {
        // grouped fold with small bundles: bundle columns, Schur shares and the groups' k x k tops in ONE launch
        gstep.epoch += 1;
        long long *fdbg = nullptr;
        if (switches().ir_debug >= 3) { // stamps of every workgroup of the factor launch -> <CHIP_IR_DEBUG_FILE>.factor
            (void)hipMalloc((void **)&fdbg, (size_t)bundles.nb * 32 * sizeof(long long));
            (void)hipMemsetAsync(fdbg, 0, (size_t)bundles.nb * 32 * sizeof(long long), stream);
        }
        gstep.dbg = fdbg;
        const int lrc = dev::gstep_factor(stream, v, bundles, gfold, gstep);
        gstep.dbg = nullptr;
        gstep_vals_valid = true; // (gs.gsl / gs.gsu hold this factorisation's L and K in gs order)
        if (fdbg) {
            (void)hipStreamSynchronize(stream);
            std::vector<long long> t((size_t)bundles.nb * 32);
            (void)hipMemcpy(t.data(), fdbg, t.size() * sizeof(long long), hipMemcpyDeviceToHost);
            (void)hipFree(fdbg);
            const std::string path = (switches().ir_debug_file.empty() ? std::string("/tmp/chip_ir_stamps.bin") : switches().ir_debug_file) + ".factor";
            if (FILE *f = std::fopen(path.c_str(), "wb")) {
                const int g = bundles.nb;
                std::fwrite(&g, sizeof(int), 1, f);
                std::fwrite(t.data(), sizeof(long long), t.size(), f);
                std::fclose(f);
            }
        }
        prof_end(PF_BFACTOR);
        if (lrc) {
            set_error(hip_err((hipError_t)lrc, "k_gstep_factor launch"));
            return CHIP_ERR_HIP;
        }
    } else {
        gstep_vals_valid = false;
        long long *fdbg = nullptr;
        if (switches().ir_debug >= 3) { // stamps of every workgroup of the factor launch -> <CHIP_IR_DEBUG_FILE>.factor
            (void)hipMalloc((void **)&fdbg, (size_t)bundles.nb * 32 * sizeof(long long));
            (void)hipMemsetAsync(fdbg, 0, (size_t)bundles.nb * 32 * sizeof(long long), stream);
        }
        dev::BundleView bdbg = bundles;
        bdbg.fdbg = fdbg;
        const int lrc = dev::bundle_factor(stream, v, bdbg, ffold, factor_lds_doubles); // everything below the cut: one launch
        prof_end(PF_BFACTOR);
        if (fdbg) {
            (void)hipStreamSynchronize(stream);
            std::vector<long long> t((size_t)bundles.nb * 32);
            (void)hipMemcpy(t.data(), fdbg, t.size() * sizeof(long long), hipMemcpyDeviceToHost);
            (void)hipFree(fdbg);
            const std::string path = (switches().ir_debug_file.empty() ? std::string("/tmp/chip_ir_stamps.bin") : switches().ir_debug_file) + ".factor";
            if (FILE *f = std::fopen(path.c_str(), "wb")) {
                const int g = bundles.nb;
                std::fwrite(&g, sizeof(int), 1, f);
                std::fwrite(t.data(), sizeof(long long), t.size(), f);
                std::fclose(f);
            }
        }
        if (lrc) {
            set_error(hip_err((hipError_t)lrc, "bundle factorisation launch"));
            return CHIP_ERR_HIP;
        }
        // single top column: pivot accumulated by the bundles; grouped fold: the k x k tops from the bundles' Schur shares
        if (fold.k == 1) dev::fold_top_pivot(stream, v, ffold); // (fast preparation: reads the top's K entry itself)
        if (gfold.ng > 0) dev::gfold_top_factor(stream, v, bundles, gfold);
    }
    const bool use_chain = !switches().no_factor_chain;
    // units (single columns, chain supernodes) by unit level; a level's supernodes run after its
    // single columns: first the contributions of outside columns into their members (the chunked
    // column kernel over the external lists, no pivots), then one workgroup per supernode
    dev::LdlView vf = v; // top-level column kernels: row lists without supernode-member columns
    if (nsn > 0) {
        vf.Rp = Rf_p;
        vf.Rcol = Rf_col;
        vf.Rpos = Rf_pos;
    }
    dev::SnodeView sview = snode_view();
    sview.U = asm_U;
    sview.Ud = asm_Ud;
    sview.asm_uoff = asm_uoff;
    sview.asm_doff = asm_doff;
    auto has_sn = [&](int l) { return nsn > 0 && sn_lvl_ptr[l + 1] > sn_lvl_ptr[l]; };
    const dev::LaunchProf lprof = launch_prof();
    // a level's ancestor updates can be assembled per target column (no atomics, a fixed order of summation): on request
    // only (CHIP_DETERMINISTIC, or CHIP_EXTEND_ASM_MIN = fewest supernodes of a level).  Measured: the atomics are NOT
    // what bounds k_snode_extend -- with the assembly config 5 is 0.5 ms per step slower (49.2 against 48.7), config 2 0.2 ms
    const int asm_min = switches().deterministic ? 2 : switches().extend_asm_min > 0 ? switches().extend_asm_min : (1 << 30);
    size_t snb_waited = 1; // groups whose event the stream has waited for (group 0 runs on the stream itself)
    bool snb_beside = false;
    auto run_supernodes = [&](int l) {
        if (!has_sn(l)) return;
        if (snb_beside)
            for (; snb_waited < snb_groups.size() && snb_groups[snb_waited].first <= l; snb_waited++)
                wait_event_or_sync(stream, snb_events[snb_waited]);
        dev::factor_B(stream, vf, snx.B(l));
        const int count = sn_lvl_ptr[l + 1] - sn_lvl_ptr[l];
        dev::SnodeAsmView av{asm_tgt, asm_src_ptr, asm_src, 0, 0};
        if (asm_U && !switches().no_extend_asm && count >= asm_min) {
            av.t0 = asm_lvl_ptr[l];
            av.nt = asm_lvl_ptr[l + 1] - asm_lvl_ptr[l];
        }
        dev::factor_snodes(stream, v, sview, sn_order + 8 * sn_lvl_ptr[l], count, sn_wmax, sn_lvl_nblk[l], sn_lvl_hmax[l],
                           sn_lvl_nbmax[l], prof_family >= PF_SN_UPDATE ? &lprof : nullptr, &av);
    };
    // the bundle columns' contributions into the supernode members of ALL unit levels: one launch (they depend on nothing
    // in the top; the chunks meet the other contributions in the same fp64 atomics)
    // Beside the chain (round 5): the chain of block columns that follows is a sequence of small dependent launches that
    // leaves most of the chip idle, and only level 0's supernodes need their contributions before it starts -- the lists
    // of the later levels run on the second stream, group by group, each group's event awaited ahead of its first level.
    // Everything that lands in a member column while the chain runs is an fp64 atomic (these chunks, k_snode_extend);
    // not with assembled updates / CHIP_DETERMINISTIC (plain read-modify-writes), not while launches are being timed.
    const int nsl = (int)snb.b_ptr.size() - 1;
    bool beside = nsn > 0 && !top_folded && snb_groups.size() > 1 && !switches().no_factor_overlap && asm_min == (1 << 30) &&
                  prof_family == PF_NONE && !alt_active;
    if (beside && ensure_alt() != CHIP_OK) beside = false;
    if (beside) {
        if (!snb_ready) CHIP_HIP(hipEventCreateWithFlags(&snb_ready, hipEventDisableTiming));
        while (snb_events.size() < snb_groups.size()) {
            hipEvent_t ev = nullptr;
            CHIP_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            snb_events.push_back(ev);
        }
        dev::factor_B(stream, vf, snb.Brange(snb_groups[0].first, snb_groups[0].second));
        CHIP_HIP(hipEventRecord(snb_ready, stream));
        CHIP_HIP(hipStreamWaitEvent(alt.stream, snb_ready, 0));
        for (size_t g = 1; g < snb_groups.size(); g++) {
            dev::factor_B(alt.stream, vf, snb.Brange(snb_groups[g].first, snb_groups[g].second));
            CHIP_HIP(hipEventRecord(snb_events[g], alt.stream));
        }
    } else if (nsn > 0 && !top_folded && nsl >= 1) {
        dev::factor_B(stream, vf, snb.Brange(0, nsl));
    }
    snb_beside = beside;
    for (int l = top_folded ? nfaclevels : 0; l < nfaclevels;) {
        int e = fac.chain_end[l];
        for (int k = l; k < e; k++)
            if (has_sn(k)) { // a level with supernodes ends the run: they must precede the next level
                e = k + 1;
                break;
            }
        if (use_chain && e - l >= 4) { // a chain-like stretch: one single-workgroup launch for levels [l, e)
            dev::factor_chain(stream, vf, fac.t_idx, fac.d_t_ptr, fac.w_idx, fac.d_w_ptr, l, e);
            run_supernodes(e - 1);
            l = e;
            continue;
        }
        prof_begin(PF_FACTOR_T);
        dev::factor_T(stream, vf, fac.T(l));
        prof_end(PF_FACTOR_T);
        dev::factor_W(stream, vf, fac.W(l));
        const dev::ChunkView b = fac.B(l);
        if (b.count) {
            dev::factor_B(stream, vf, b);
            dev::factor_finalize(stream, v, fac.BR(l));
        }
        run_supernodes(l);
        l++;
    }
    if (snb_beside)
        for (; snb_waited < snb_groups.size(); snb_waited++) wait_event_or_sync(stream, snb_events[snb_waited]);
    if (nsn > 0) dev::gather_values(stream, Rfx, Lx, Rf_pos, nRf); // L at the filtered row lists (forward sweep)
    if (sn_g_ntasks > 0) { // the substitution matrices of the supernodes of moderate width, all of them in one launch
        dev::SnodeView sg = snode_view();
        dev::snode_ginv(stream, v, sg, sn_order, sn_g_tasks, sn_g_ntasks);
    }
    if (nsn <= 0) dev::topblk_build(stream, v, topblk); // inverses of the diagonal blocks of a tall top
    // full rows of the top for the residual of the one-kernel-per-phase path; a handle whose fused launch folds the
    // top per group never reads them unless that launch falls back: refreshed on demand (enqueue_residual)
    sx_valid = false;
    if (!fold.k && !(ir_fused && gfold.ng > 0)) {
        dev::gather_values(stream, Sx, Kx, Smap, (int)nnzS);
        sx_valid = true;
    }
    // (the entry-parallel bundle factorisation keeps no row-major mirror either)
    rx_valid = !ir_fused && !(factor_lds_doubles > 0 && fu_rec && !switches().no_factor_flat);
    return CHIP_OK;
}
int Engine::refactor_collect() {
    int rc = read_mailbox();
    if (rc) return rc;
    // a failed refactor leaves garbage in L / D / Dinv: the handle goes back to "not factored" so that
    // a later solve() fails loudly (the reference panics on ZeroPivot, ldlsolvers/qdldl.rs:104)
    factored = !mb_host->status[0] && !mb_host->status[1];
    last_regularize_count = mb_host->status[2];
    if (mb_host->status[1]) {
        // the reference adapter unwrap()s QDLDLError::ZeroPivot (ldlsolvers/qdldl.rs:104)
        // i.e. panics; we report it as a numerical failure instead.
        set_error("zero pivot");
        return 0;
    }
    return mb_host->status[0] ? 0 : 1;
}

// qdldl.rs:755-768 in the permuted numbering, in place
void Engine::enqueue_solve_inplace(double *xp, const double *addv) {
    if (!gs_built && nsn > 0) (void)build_gsweeps(); // (outside any capture: allocates)
    // (a fused handle taking the one-kernel-per-phase path: these kernels stream L by rows.  Refreshed here,
    // ahead of the graph lookup and outside any capture: a replayed graph contains no gather)
    if (!rx_valid && rx_needed()) {
        dev::gather_values(stream, Rx, Lx, Rpos, (int)nnzR);
        rx_valid = true;
    }
    if (st.use_graph && prof_family == PF_NONE) {
        for (const SolveGraph &g : graphs)
            if (g.xp == xp && g.addv == addv) {
                if (hipGraphLaunch(g.exec, stream) == hipSuccess) return;
                break;
            }
        if (graphs.size() < 16) { // the refinement rotates three vectors: a handful of pairs at most
            hipGraph_t gr = nullptr;
            hipGraphExec_t ex = nullptr;
            if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                enqueue_solve_direct(xp, addv);
                const bool ok = hipStreamEndCapture(stream, &gr) == hipSuccess && gr &&
                                hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0) == hipSuccess;
                if (gr) (void)hipGraphDestroy(gr);
                if (ok && hipGraphLaunch(ex, stream) == hipSuccess) {
                    graphs.push_back({xp, addv, ex});
                    return;
                }
                if (ex) (void)hipGraphExecDestroy(ex);
                (void)hipGetLastError();
            }
        }
    }
    enqueue_solve_direct(xp, addv);
}
bool Engine::pair_lockstep_ok() {
    if (nsn <= 0 || fold.k || st.use_graph || switches().no_snode_tri || switches().no_pair_lockstep || prof_family != PF_NONE ||
        !sn_flags || !alt.sn_flags)
        return false;
    for (int l = 0; l < nfaclevels; l++) {
        const bool g = sn_g_ntasks > 0 && sn_lvl_g[l];
        if (sn_lvl_ptr[l + 1] > sn_lvl_ptr[l] && !g && sn_lvl_nblk[l] * 64 > 2 * 64) return true; // (a level k_snode_tri takes)
    }
    return false;
}
void Engine::enqueue_solve_pair(double *xa, const double *addva, double *xb, const double *addvb) {
    if (!pair_lockstep_ok()) {
        enqueue_solve_inplace(xa, addva);
        swap_ctx();
        enqueue_solve_inplace(xb, addvb);
        swap_ctx();
        return;
    }
    if (!gs_built && nsn > 0) (void)build_gsweeps();
    if (!rx_valid && rx_needed()) {
        dev::gather_values(stream, Rx, Lx, Rpos, (int)nnzR);
        rx_valid = true;
    }
    // The level loop of enqueue_solve_direct's supernode branch, once, for both contexts: `each` runs a stage for A on
    // A's stream, swaps the contexts (stream, scratch vectors, barrier words, message buffers), runs it for B, swaps back.
    double *xs[2] = {xa, xb};
    const double *addvs[2] = {addva, addvb};
    int ctx = 0;
    auto each = [&](auto &&fn) {
        ctx = 0;
        fn(xs[0], addvs[0]);
        swap_ctx();
        ctx = 1;
        fn(xs[1], addvs[1]);
        swap_ctx();
        ctx = 0;
    };
    hipStream_t sA = stream, sB = alt.stream;
    const dev::LdlView v = view();
    const dev::SnodeView sview = snode_view();
    each([&](double *x, const double *) { dev::bundle_fwd(stream, v, bundles, x, fold); });
    const bool merge = sn_g_ntasks > 0 && !switches().no_sweep_merge;
    auto is_g = [&](int l) { return sn_g_ntasks > 0 && sn_lvl_g[l] && sn_lvl_ptr[l + 1] > sn_lvl_ptr[l]; };
    const bool persist = merge && gs_lv && !switches().no_sweep_persist && !gs_off;
    if (persist && exch_pending) (void)wait_for_exchange();
    int epoch = ++sn_epoch;
    // the wide levels: B's stream hands its vector over (event), A's stream runs the launch for both, B's stream waits for it
    auto tri_level = [&](dev::GatherMode m, int l) {
        dev::SnodeTriView ta{sn_blk_ptr, sn_flags, epoch, norm_nan(1)};
        swap_ctx();
        dev::SnodeTriView tb{sn_blk_ptr, sn_flags, epoch, norm_nan(1)};
        swap_ctx();
        (void)hipEventRecord(pair_ev_b, sB);
        if (hipStreamWaitEvent(sA, pair_ev_b, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipEventSynchronize(pair_ev_b);
        }
        dev::solve_snodes(sA, m, v, sview, sn_order + 8 * sn_lvl_ptr[l], sn_lvl_ptr[l + 1] - sn_lvl_ptr[l], sn_wmax, sn_nbmax,
                          sn_lvl_nblk[l] * 64, sn_lvl_nbmax[l], xs[0], &ta, nullptr, xs[1], &tb);
        tri2_launches++;
        (void)hipEventRecord(pair_ev_a, sA);
        if (hipStreamWaitEvent(sB, pair_ev_a, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipEventSynchronize(pair_ev_a);
        }
    };
    bool gathered = false;
    for (int l = 0; l < nfaclevels; l++) {
        if (!gathered)
            each([&](double *x, const double *) {
                dev::GatherArgs f{Rf_p, Rf_col, Rfx, x, x, nullptr, nullptr, nullptr};
                dev::gather_merged(stream, dev::FWD, f, fwu.T(l), fwu.W(l), fwu.B(l));
            });
        gathered = false;
        if (persist && gs_run_f[(size_t)l] >= 0) {
            const GRun &r = gs_runs[(size_t)gs_run_f[(size_t)l]];
            each([&](double *x, const double *) {
                dev::GatherArgs f{Rf_p, Rf_col, Rfx, x, x, nullptr, nullptr, nullptr};
                dev::solve_snodes_gsweep(stream, dev::FWD, v, sview, sn_order, x, sn_yt, gs_lv + r.off, r.nlev, r.grid, r.lds, f, gs_ctl,
                                         norm_nan(1), nullptr);
                gs_launches++;
            });
            l += r.nlev - 1;
            gathered = l + 1 < nfaclevels;
            continue;
        }
        if (is_g(l)) {
            const bool ride = merge && l + 1 < nfaclevels;
            each([&](double *x, const double *) {
                dev::GatherArgs f{Rf_p, Rf_col, Rfx, x, x, nullptr, nullptr, nullptr};
                dev::solve_snodes_g(stream, dev::FWD, v, sview, sn_order + 8 * sn_lvl_ptr[l], sn_lvl_ptr[l + 1] - sn_lvl_ptr[l],
                                    sn_lvl_wmax[l], sn_lvl_hmax[l], x, sn_yt, nullptr, ride ? &f : nullptr,
                                    ride ? fwu.T(l + 1) : dev::ListView{nullptr, 0}, ride ? fwu.W(l + 1) : dev::ListView{nullptr, 0},
                                    ride ? fwu.B(l + 1) : dev::ChunkView{nullptr, nullptr, nullptr, 0});
            });
            gathered = ride;
        } else if (sn_lvl_ptr[l + 1] > sn_lvl_ptr[l] && sn_lvl_nblk[l] * 64 > 2 * 64) {
            tri_level(dev::FWD, l);
        } else {
            each([&](double *x, const double *) {
                dev::SnodeTriView t{sn_blk_ptr, sn_flags, epoch, norm_nan(1)};
                dev::solve_snodes(stream, dev::FWD, v, sview, sn_order + 8 * sn_lvl_ptr[l], sn_lvl_ptr[l + 1] - sn_lvl_ptr[l], sn_wmax,
                                  sn_nbmax, sn_lvl_nblk[l] * 64, sn_lvl_nbmax[l], x, &t, nullptr);
            });
        }
    }
    epoch = ++sn_epoch;
    for (int l = nfaclevels - 1; l >= 0; l--) {
        if (persist && gs_run_b[(size_t)l] >= 0) {
            const GRun &r = gs_runs[(size_t)gs_run_b[(size_t)l]];
            each([&](double *x, const double *) {
                dev::GatherArgs g{Lp, Li, Lx, x, x, Dinv, nullptr, nullptr};
                dev::solve_snodes_gsweep(stream, dev::BWD, v, sview, sn_order, x, sn_yt, gs_lv + r.off, r.nlev, r.grid, r.lds, g, gs_ctl,
                                         norm_nan(1), nullptr);
                gs_launches++;
            });
            l -= r.nlev - 1;
            continue;
        }
        const dev::ChunkView b = bwu.B(l);
        bool rode = false;
        if (is_g(l)) {
            const bool ride = merge && b.count == 0;
            each([&](double *x, const double *) {
                dev::GatherArgs g{Lp, Li, Lx, x, x, Dinv, nullptr, nullptr};
                dev::solve_snodes_g(stream, dev::BWD, v, sview, sn_order + 8 * sn_lvl_ptr[l], sn_lvl_ptr[l + 1] - sn_lvl_ptr[l],
                                    sn_lvl_wmax[l], sn_lvl_hmax[l], x, sn_yt, nullptr, ride ? &g : nullptr,
                                    ride ? bwu.T(l) : dev::ListView{nullptr, 0}, ride ? bwu.W(l) : dev::ListView{nullptr, 0});
            });
            rode = ride;
        } else if (sn_lvl_ptr[l + 1] > sn_lvl_ptr[l] && sn_lvl_nblk[l] * 64 > 2 * 64) {
            tri_level(dev::BWD, l);
        } else {
            each([&](double *x, const double *) {
                dev::SnodeTriView t{sn_blk_ptr, sn_flags, epoch, norm_nan(1)};
                dev::solve_snodes(stream, dev::BWD, v, sview, sn_order + 8 * sn_lvl_ptr[l], sn_lvl_ptr[l + 1] - sn_lvl_ptr[l], sn_wmax,
                                  sn_nbmax, sn_lvl_nblk[l] * 64, sn_lvl_nbmax[l], x, &t, nullptr);
            });
        }
        if (rode) continue;
        each([&](double *x, const double *) {
            dev::GatherArgs g{Lp, Li, Lx, x, x, Dinv, nullptr, nullptr};
            if (b.count) dev::gather_Bprep(stream, dev::BWD, g, bwu.BR(l));
            dev::gather_merged(stream, dev::BWD, g, bwu.T(l), bwu.W(l), b);
        });
    }
    each([&](double *x, const double *addv) {
        dev::bundle_bwd(stream, v, bundles, x, addv);
        if (addv && N > NF) dev::add_vec(stream, x + NF, addv + NF, N - NF);
    });
}
void Engine::enqueue_solve_direct(double *xp, const double *addv) {
    const dev::LdlView v = view();
    dev::bundle_fwd(stream, v, bundles, xp, fold);
    if (fold.k) { // an "arrow": the bundles have already folded the top rows; finish the k x k part
        dev::fold_top_solve(stream, v, fold, xp);
        dev::bundle_bwd(stream, v, bundles, xp, addv);
        if (addv && N > NF) dev::add_vec(stream, xp + NF, addv + NF, N - NF);
        return;
    }
    if (nsn > 0) {
        // chain supernodes: units by unit level.  Forward: every top row first gathers from the columns
        // that are not supernode members, then the level's supernodes solve their dense triangles and
        // push L_BS x_S to their ancestors' entries; backward: the reverse, column oriented.
        const dev::SnodeView sview = snode_view();
        // wide supernodes: several workgroups per supernode, pipelined through per-block flags that carry this
        // sweep's epoch (not inside a captured graph: a replay would meet its own flags)
        const bool use_tri = !switches().no_snode_tri && !st.use_graph;
        dev::SnodeTriView tri{sn_blk_ptr, sn_flags, 0, norm_nan(1)};
        dev::GatherArgs f{Rf_p, Rf_col, Rfx, xp, xp, nullptr, nullptr, nullptr};
        tri.epoch = ++sn_epoch;
        const dev::LaunchProf lprof = launch_prof();
        const dev::LaunchProf *lp = prof_family >= PF_SN_UPDATE ? &lprof : nullptr;
        // levels on the one-pass matrices take the NEXT level's row gathers (forward) / their own ordinary columns (backward)
        // into the supernodes' launch (snode_g.hip: SweepGather); CHIP_NO_SWEEP_MERGE keeps the two launches per level
        const bool merge = sn_g_ntasks > 0 && !switches().no_sweep_merge;
        auto is_g = [&](int l) { return sn_g_ntasks > 0 && sn_lvl_g[l] && sn_lvl_ptr[l + 1] > sn_lvl_ptr[l]; };
        const bool persist = merge && gs_lv && !switches().no_sweep_persist && !gs_off; // runs of such levels: one persistent launch each
        if (persist && exch_pending) (void)wait_for_exchange(); // (a persistent sweep beside the collective's kernels may not be co-resident)
        bool gathered = false; // level l's gathers already ran inside the previous level's launch
        for (int l = 0; l < nfaclevels; l++) {
            if (!gathered) {
                prof_begin(PF_SN_GATHER);
                dev::gather_merged(stream, dev::FWD, f, fwu.T(l), fwu.W(l), fwu.B(l));
                prof_end(PF_SN_GATHER);
            }
            gathered = false;
            if (persist && gs_run_f[(size_t)l] >= 0) { // a run of such levels: one persistent launch
                const GRun &r = gs_runs[(size_t)gs_run_f[(size_t)l]];
                dev::solve_snodes_gsweep(stream, dev::FWD, v, sview, sn_order, xp, sn_yt, gs_lv + r.off, r.nlev, r.grid, r.lds, f,
                                         gs_ctl, norm_nan(1), lp);
                gs_launches++;
                l += r.nlev - 1;
                gathered = l + 1 < nfaclevels; // (the run's last level carried them)
                continue;
            }
            if (is_g(l)) { // one pass over G, no hops (x_S(new) -> sn_yt, the rows of B subtracted in place)
                const bool ride = merge && l + 1 < nfaclevels;
                dev::solve_snodes_g(stream, dev::FWD, v, sview, sn_order + 8 * sn_lvl_ptr[l], sn_lvl_ptr[l + 1] - sn_lvl_ptr[l],
                                    sn_lvl_wmax[l], sn_lvl_hmax[l], xp, sn_yt, lp, ride ? &f : nullptr,
                                    ride ? fwu.T(l + 1) : dev::ListView{nullptr, 0}, ride ? fwu.W(l + 1) : dev::ListView{nullptr, 0},
                                    ride ? fwu.B(l + 1) : dev::ChunkView{nullptr, nullptr, nullptr, 0});
                gathered = ride;
            } else {
                dev::solve_snodes(stream, dev::FWD, v, sview, sn_order + 8 * sn_lvl_ptr[l], sn_lvl_ptr[l + 1] - sn_lvl_ptr[l],
                                  sn_wmax, sn_nbmax, sn_lvl_nblk[l] * 64, sn_lvl_nbmax[l], xp, use_tri ? &tri : nullptr, lp);
            }
        }
        tri.epoch = ++sn_epoch;
        dev::GatherArgs g{Lp, Li, Lx, xp, xp, Dinv, nullptr, nullptr};
        for (int l = nfaclevels - 1; l >= 0; l--) {
            if (persist && gs_run_b[(size_t)l] >= 0) {
                const GRun &r = gs_runs[(size_t)gs_run_b[(size_t)l]];
                dev::solve_snodes_gsweep(stream, dev::BWD, v, sview, sn_order, xp, sn_yt, gs_lv + r.off, r.nlev, r.grid, r.lds, g,
                                         gs_ctl, norm_nan(1), lp);
                gs_launches++;
                l -= r.nlev - 1;
                continue;
            }
            const dev::ChunkView b = bwu.B(l);
            if (is_g(l)) {
                const bool ride = merge && b.count == 0; // (chunked columns need their preparation pass first)
                dev::solve_snodes_g(stream, dev::BWD, v, sview, sn_order + 8 * sn_lvl_ptr[l], sn_lvl_ptr[l + 1] - sn_lvl_ptr[l],
                                    sn_lvl_wmax[l], sn_lvl_hmax[l], xp, sn_yt, lp, ride ? &g : nullptr,
                                    ride ? bwu.T(l) : dev::ListView{nullptr, 0}, ride ? bwu.W(l) : dev::ListView{nullptr, 0});
                if (ride) continue;
            } else {
                dev::solve_snodes(stream, dev::BWD, v, sview, sn_order + 8 * sn_lvl_ptr[l], sn_lvl_ptr[l + 1] - sn_lvl_ptr[l],
                                  sn_wmax, sn_nbmax, sn_lvl_nblk[l] * 64, sn_lvl_nbmax[l], xp, use_tri ? &tri : nullptr, lp);
            }
            if (b.count) dev::gather_Bprep(stream, dev::BWD, g, bwu.BR(l));
            prof_begin(PF_SN_GATHER);
            dev::gather_merged(stream, dev::BWD, g, bwu.T(l), bwu.W(l), b);
            prof_end(PF_SN_GATHER);
        }
        dev::bundle_bwd(stream, v, bundles, xp, addv);
        if (addv && N > NF) dev::add_vec(stream, xp + NF, addv + NF, N - NF);
        return;
    }
    if (topblk.nblocks) { // tall top: one dependent step per block of rows instead of per level
        dev::topblk_solve(stream, dev::FWD, v, topblk, xp);
        dev::topblk_solve(stream, dev::BWD, v, topblk, xp);
        dev::bundle_bwd(stream, v, bundles, xp, addv);
        if (addv && N > NF) dev::add_vec(stream, xp + NF, addv + NF, N - NF);
        return;
    }
    dev::GatherArgs f{Rp, Rcol, Rx, xp, xp, nullptr, nullptr, nullptr};
    for (int l = 0; l < nlevels;) {
        const int e = fwd.chain_end[l];
        if (e > l + 1) { // a chain-like stretch: one single-workgroup launch for levels [l, e)
            dev::gather_chain(stream, dev::FWD, f, fwd.t_idx, fwd.d_t_ptr, fwd.w_idx, fwd.d_w_ptr, l, e);
            l = e;
            continue;
        }
        prof_begin(PF_FWD_T);
        dev::gather_merged(stream, dev::FWD, f, fwd.T(l), fwd.W(l), fwd.B(l));
        prof_end(PF_FWD_T);
        l++;
    }
    dev::GatherArgs g{Lp, Li, Lx, xp, xp, Dinv, nullptr, nullptr};
    for (int l = nlevels - 1; l >= 0;) {
        const int b0 = bwd.chain_begin[l]; // first level of the run of narrow levels that ends at l
        if (bwd.chain_end[l] == l + 1 && l + 1 - b0 >= 2) {
            dev::gather_chain(stream, dev::BWD, g, bwd.t_idx, bwd.d_t_ptr, bwd.w_idx, bwd.d_w_ptr, b0, l + 1);
            l = b0 - 1;
            continue;
        }
        const dev::ChunkView b = bwd.B(l);
        if (b.count) dev::gather_Bprep(stream, dev::BWD, g, bwd.BR(l));
        prof_begin(PF_BWD_T);
        dev::gather_merged(stream, dev::BWD, g, bwd.T(l), bwd.W(l), b);
        prof_end(PF_BWD_T);
        l--;
    }
    dev::bundle_bwd(stream, v, bundles, xp, addv);
    // the top rows feed the bundles' backward sweeps, so their share of "+ addv" comes last
    if (addv && N > NF) dev::add_vec(stream, xp + NF, addv + NF, N - NF);
}

// e = b - K x with the UNregularised K (directldlkktsolver.rs:334-347)
// phases (enqueue_residual_pair runs them separately): 1 = what precedes the dense diagonal blocks' products, 2 = those
// products for this vector alone, 4 = everything after them
void Engine::enqueue_residual(double *e, const double *b, const double *x, int set, int phases) {
    dev::GatherArgs a{Sp, Scol, Sx, x, e, b, nullptr, nullptr};
    if (set >= 0) {
        a.nrm = norm_set(set);
        a.nan = norm_nan(set);
    }
    if (fold.k) { // top rows: bundle shares accumulated by the bundle kernel, finished by one tiny launch
        if (!(phases & 4)) return;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        prof_pair(PF_SYMV_T, &ev0, &ev1);
        dev::bundle_symv(stream, bundles, Up, Ucol, Ux, x, b, e, a.nrm, a.nan, fold, ev0, ev1);
        dev::fold_top_residual(stream, fold, Kx, x, b, e, a.nrm, a.nan); // (top-top entries by their position in Kx)
        return;
    }
    if (phases & 1) {
        if (!sx_valid) { // (K's values have not changed since the refactor: every write to them is followed by one)
            dev::gather_values(stream, Sx, Kx, Smap, (int)nnzS);
            sx_valid = true;
        }
        if (xperm) dev::gather_values(stream, xs_view, x, xperm, N);
        // dense diagonal blocks of the top: multiplied from K's values directly, taken off b beforehand
        if (dblk.nblk) (void)hipMemcpyAsync(bt_view, b, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, stream);
    }
    if (xperm) a.xin = xs_view;
    if (dblk.nblk) {
        if (phases & 2) dev::dblk_symv(stream, dblk, Kx, x, bt_view);
        a.aux = bt_view;
    }
    if (!(phases & 4)) return;
    const dev::ChunkView bc = smv.B(0);
    if (bc.count) dev::gather_Bprep(stream, dev::SYMV, a, smv.BR(0));
    dev::gather_merged(stream, dev::SYMV, a, smv.T(0), smv.W(0), bc); // the top rows (full rows)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    prof_pair(PF_SYMV_T, &ev0, &ev1);
    dev::bundle_symv(stream, bundles, Up, Ucol, Ux, x, b, e, a.nrm, a.nan, dev::FoldView{}, ev0, ev1); // everything else
    if (bc.count && set >= 0) dev::norm_rows(stream, e, smv.BR(0), a.nrm, a.nan);
}
// The residuals of the two solves of a pair (A: this context, B: the second one).  With dense diagonal blocks in the top
// their products are ONE launch for both vectors (k_dblk_symv<2>: every entry of the blocks read once) on A's stream:
// B's stream hands its vector over (event), waits for the launch and finishes its own sums.
void Engine::enqueue_residual_pair(double *eA, const double *bA, const double *xA, double *eB, const double *bB, const double *xB, int set) {
    const bool two = dblk.nblk > 0 && dblk_pair_ok && !fold.k && pair_ev_a && pair_ev_b && !switches().no_dblk_pair && prof_family == PF_NONE;
    if (!two) {
        enqueue_residual(eA, bA, xA, set);
        swap_ctx();
        enqueue_residual(eB, bB, xB, set);
        swap_ctx();
        return;
    }
    hipStream_t sA = stream, sB = alt.stream;
    enqueue_residual(eA, bA, xA, set, 1);
    swap_ctx();
    enqueue_residual(eB, bB, xB, set, 1);
    double *PB = dblk.P, *btB = bt_view;
    swap_ctx();
    (void)hipEventRecord(pair_ev_b, sB);
    if (hipStreamWaitEvent(sA, pair_ev_b, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipEventSynchronize(pair_ev_b);
    }
    dev::dblk_symv2(sA, dblk, Kx, xA, xB, PB, bt_view);
    dblk2_launches++;
    (void)hipEventRecord(pair_ev_a, sA);
    if (hipStreamWaitEvent(sB, pair_ev_a, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipEventSynchronize(pair_ev_a);
    }
    enqueue_residual(eA, bA, xA, set, 4);
    swap_ctx();
    dev::dblk_finish(stream, dblk, btB); // (this context's partial sums, written by the launch on the other stream)
    enqueue_residual(eB, bB, xB, set, 4);
    swap_ctx();
}

int Engine::zero_norm_sets() {
    CHIP_HIP(hipMemsetAsync(nrm_dev, 0, (size_t)NRM_SETS * NRM_SET_WORDS * sizeof(unsigned long long), stream));
    return CHIP_OK;
}

int Engine::read_norms(int first, int count, double *out) {
    CHIP_HIP(hipMemcpyAsync(nrm_host, norm_set(first), (size_t)count * NRM_SET_WORDS * sizeof(unsigned long long),
                            hipMemcpyDeviceToHost, stream));
    CHIP_HIP(hipStreamSynchronize(stream));
    for (int k = 0; k < count; k++) {
        unsigned long long m = 0;
        const unsigned long long *base = nrm_host + (size_t)k * NRM_SET_WORDS;
        for (int i = 0; i < dev::NRM_SLOTS; i++) m = std::max(m, base[(size_t)i * dev::NRM_STRIDE]);
        double d;
        std::memcpy(&d, &m, sizeof(d));
        const int nanflag = *(const int *)(base + (size_t)dev::NRM_SLOTS * dev::NRM_STRIDE);
        out[k] = nanflag ? std::nan("") : d; // vecmath.rs:132-142 NaN propagation
    }
    return CHIP_OK;
}
int Engine::read_norm(int set, double *out) { return read_norms(set, 1, out); }

// the caller's K.nzval order -> the device's T order (one staging copy + one gather kernel)
int Engine::upload_values(const double *host_nzval) {
    if (!nnzK) return CHIP_OK;
    int rc;
    if (!d_v2k) {
        if ((rc = upload(&d_v2k, h_v2k, (size_t)nnzK))) return rc;
        if ((rc = alloc(&d_stage, (size_t)nnzK))) return rc;
    }
    CHIP_HIP(hipMemcpyAsync(d_stage, host_nzval, (size_t)nnzK * sizeof(double), hipMemcpyHostToDevice, stream));
    dev::gather_values(stream, Kx, d_stage, d_v2k, (int)nnzK);
    return CHIP_OK;
}
int Engine::download_values(double *host_nzval) {
    if (!nnzK) return CHIP_OK;
    std::vector<double> tmp((size_t)nnzK);
    CHIP_HIP(hipStreamSynchronize(stream));
    CHIP_HIP(hipMemcpy(tmp.data(), Kx, (size_t)nnzK * sizeof(double), hipMemcpyDeviceToHost));
    for (i64 p = 0; p < nnzK; p++) host_nzval[p] = tmp[(size_t)h_k2v[(size_t)p]];
    return CHIP_OK;
}

// the C ABI layer allocates / uploads these element types through the engine
template int Engine::alloc<int>(int **, size_t);
template int Engine::alloc<double>(double **, size_t);
template int Engine::alloc<int8_t>(int8_t **, size_t);
template int Engine::upload<int>(int **, const std::vector<int> &, size_t);
template int Engine::upload<double>(double **, const std::vector<double> &, size_t);
template int Engine::upload<int8_t>(int8_t **, const std::vector<int8_t> &, size_t);
template int Engine::upload<unsigned short>(unsigned short **, const std::vector<unsigned short> &, size_t);

} // namespace chip
