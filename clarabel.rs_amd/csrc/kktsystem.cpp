// kktsystem.cpp -- L3 of the C ABI: DefaultKKTSystem (default/kktsystem.rs:16-292) and
// DefaultResiduals::update (default/residuals.rs:69-111) with every vector resident in HBM.
// Built on the public chip_kkt_* entry points (L2) and on the handle's stream; the only new
// device work is sparse gemv / symv (the row-gather family in SPMV mode), w = a x + b y and
// deterministic dot products.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#include "engine.hpp"

using namespace chip;

namespace {

int failk(int code, const std::string &msg) {
    set_error(msg);
    return code;
}

constexpr int T_MAX = 32, B_MIN = 16384, B_CHUNK = 4096; // same classes as symbolic.cpp

// CSR-like sparse operator y = aux + alpha * M x with its row work lists
struct SpMat {
    int rows = 0;
    int *ptr = nullptr, *idx = nullptr, *map = nullptr; // map: position in the caller's nzval
    double *val = nullptr;
    size_t nnz = 0;
    int *t_idx = nullptr, *w_idx = nullptr, *b_row = nullptr, *b_beg = nullptr, *b_end = nullptr, *br_idx = nullptr;
    int nt = 0, nw = 0, nbc = 0, nbr = 0;
};

struct DevPool {
    std::vector<void *> allocs;
    template <typename T> int alloc(T **dst, size_t n) {
        *dst = nullptr;
        void *p = nullptr;
        CHIP_HIP(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
        allocs.push_back(p);
        *dst = (T *)p;
        return CHIP_OK;
    }
    template <typename T> int upload(T **dst, const std::vector<T> &src) {
        int rc = alloc(dst, src.size());
        if (rc) return rc;
        if (!src.empty()) CHIP_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
        return CHIP_OK;
    }
    ~DevPool() {
        for (void *p : allocs) (void)hipFree(p);
    }
};

int build_spmat(DevPool &pool, SpMat &M, int rows, const std::vector<int> &ptr, const std::vector<int> &idx,
                const std::vector<int> &map) {
    M.rows = rows;
    M.nnz = idx.size();
    std::vector<int> t, w, brow, bbeg, bend, br;
    for (int r = 0; r < rows; r++) {
        const int len = ptr[r + 1] - ptr[r];
        if (len > B_MIN) {
            br.push_back(r);
            for (int b = ptr[r]; b < ptr[r + 1]; b += B_CHUNK) {
                brow.push_back(r);
                bbeg.push_back(b);
                bend.push_back(std::min(ptr[r + 1], b + B_CHUNK));
            }
        } else if (len > T_MAX) {
            w.push_back(r);
        } else {
            t.push_back(r); // includes empty rows: they must still receive aux (or 0)
        }
    }
    int rc;
    if ((rc = pool.upload(&M.ptr, ptr))) return rc;
    if ((rc = pool.upload(&M.idx, idx))) return rc;
    if ((rc = pool.upload(&M.map, map))) return rc;
    if ((rc = pool.alloc(&M.val, idx.size()))) return rc;
    if ((rc = pool.upload(&M.t_idx, t))) return rc;
    if ((rc = pool.upload(&M.w_idx, w))) return rc;
    if ((rc = pool.upload(&M.b_row, brow))) return rc;
    if ((rc = pool.upload(&M.b_beg, bbeg))) return rc;
    if ((rc = pool.upload(&M.b_end, bend))) return rc;
    if ((rc = pool.upload(&M.br_idx, br))) return rc;
    M.nt = (int)t.size();
    M.nw = (int)w.size();
    M.nbc = (int)brow.size();
    M.nbr = (int)br.size();
    return CHIP_OK;
}

} // namespace

struct chip_kktsystem {
    chip_kkt *kkt = nullptr;
    hipStream_t stream = nullptr;
    int device = 0;
    int n = 0, m = 0;
    size_t nnzP = 0, nnzA = 0;
    DevPool pool;
    SpMat Psym, Arow, Acol; // P as full symmetric rows; A by rows (A x); A by columns (A' z)
    double *q = nullptr, *b = nullptr, *src = nullptr; // src: staging for value refreshes
    double *x1 = nullptr, *z1 = nullptr, *x2 = nullptr, *z2 = nullptr, *workx = nullptr, *workz = nullptr,
           *work_conic = nullptr, *wn = nullptr, *workx2 = nullptr, *wn2 = nullptr, *negq = nullptr;
    dev::DotBatch batch{}; // dot products queued for ONE launch pair (flushed by read_dots)
    double *dots = nullptr, *scratch = nullptr; // 16 result slots + reduction scratch
    double hdots[16];

    void spmv(const SpMat &M, double *y, const double *aux, double alpha, const double *x) {
        dev::GatherArgs a{M.ptr, M.idx, M.val, x, y, aux, nullptr, nullptr, alpha};
        if (M.nbr) dev::gather_Bprep(stream, dev::SPMV, a, dev::ListView{M.br_idx, M.nbr});
        dev::gather_merged(stream, dev::SPMV, a, dev::ListView{M.t_idx, M.nt}, dev::ListView{M.w_idx, M.nw},
                           dev::ChunkView{M.b_row, M.b_beg, M.b_end, M.nbc});
    }
    int refresh(SpMat &M, const double *host_vals, size_t nsrc) {
        if (nsrc) CHIP_HIP(hipMemcpyAsync(src, host_vals, nsrc * sizeof(double), hipMemcpyHostToDevice, stream));
        dev::gather_values(stream, M.val, src, M.map, (int)M.nnz);
        CHIP_HIP(hipStreamSynchronize(stream)); // src is reused by the next refresh
        return CHIP_OK;
    }
    // queue a . b -> hdots[slot]; the operands must stay untouched until read_dots()
    void dot(int slot, const double *a, const double *bb, int len) {
        if (batch.count == dev::DOT_BATCH_MAX) flush_dots();
        batch.s[batch.count++] = dev::DotSpec{a, bb, len, slot};
    }
    void flush_dots() {
        dev::multi_dot(stream, batch, dots, scratch);
        batch.count = 0;
    }
    int read_dots() {
        flush_dots();
        CHIP_HIP(hipMemcpyAsync(hdots, dots, sizeof(hdots), hipMemcpyDeviceToHost, stream));
        CHIP_HIP(hipStreamSynchronize(stream));
        return CHIP_OK;
    }
    int copy(double *dst, const double *srcv, int len) {
        if (len) CHIP_HIP(hipMemcpyAsync(dst, srcv, (size_t)len * sizeof(double), hipMemcpyDeviceToDevice, stream));
        return CHIP_OK;
    }
    int zero(double *dst, int len) {
        if (len) CHIP_HIP(hipMemsetAsync(dst, 0, (size_t)len * sizeof(double), stream));
        return CHIP_OK;
    }
    // cached per update(): q.x2, b.z2, x2'Px2 (the reference recomputes the same values per solve)
    double qx2 = 0, bz2 = 0, x2Px2 = 0;
    int solve_constant_rhs();
};

// new values on the fixed patterns (create, and data_updating.rs:98-133)
static int refresh_data(chip_kktsystem *h, const double *P, const double *A, const double *q, const double *b) {
    int rc;
    if (P && h->nnzP) {
        if ((rc = h->refresh(h->Psym, P, h->nnzP))) return rc;
    }
    if (A && h->nnzA) {
        if ((rc = h->refresh(h->Arow, A, h->nnzA))) return rc;
        if ((rc = h->refresh(h->Acol, A, h->nnzA))) return rc;
    }
    if (q && h->n) CHIP_HIP(hipMemcpyAsync(h->q, q, (size_t)h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (q && h->n) dev::waxpby(h->stream, h->negq, -1.0, h->q, 0.0, nullptr, h->n); // the constant rhs -q
    if (b && h->m) CHIP_HIP(hipMemcpyAsync(h->b, b, (size_t)h->m * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CHIP_HIP(hipStreamSynchronize(h->stream));
    return CHIP_OK;
}

int32_t chip_kktsystem_create(chip_kktsystem **out, chip_kkt *kkt, const uint64_t *Pcolptr, const uint64_t *Prowval,
                              const double *Pnzval, const uint64_t *Acolptr, const uint64_t *Arowval,
                              const double *Anzval, const double *q, const double *b) {
    if (!out || !kkt || !Pcolptr || !Acolptr) return CHIP_ERR_ARG;
    *out = nullptr;
    if (kkt_host_only(kkt)) return failk(CHIP_ERR_NO_DEVICE, "host-only handle: no numeric work without a GPU");
    int64_t dims[8];
    int rc = chip_kkt_dims(kkt, dims);
    if (rc) return rc;
    const int64_t n = dims[0], m = dims[1];
    const uint64_t nnzP = Pcolptr[n], nnzA = Acolptr[n];
    if (nnzP >= (1ull << 31) || nnzA >= (1ull << 31)) return failk(CHIP_ERR_DIM, "nnz out of int32 range");
    if ((nnzP && (!Prowval || !Pnzval)) || (nnzA && (!Arowval || !Anzval)) || (n && !q) || (m && !b))
        return CHIP_ERR_ARG;
    std::unique_ptr<chip_kktsystem> h(new chip_kktsystem());
    h->kkt = kkt;
    h->stream = (hipStream_t)chip_kkt_stream(kkt);
    h->device = kkt_device(kkt);
    h->n = (int)n;
    h->m = (int)m;
    h->nnzP = nnzP;
    h->nnzA = nnzA;
    CHIP_HIP(hipSetDevice(h->device));
    // ---- P as full symmetric rows (csc/matrix_math.rs:178-208 applies each off-diagonal twice)
    {
        std::vector<int> ptr((size_t)n + 1, 0);
        for (int64_t c = 0; c < n; c++)
            for (uint64_t p = Pcolptr[c]; p < Pcolptr[c + 1]; p++) {
                const int64_t r = (int64_t)Prowval[p];
                if (r > c || r < 0) return failk(CHIP_ERR_NOT_TRIU, "P is not upper triangular");
                ptr[r + 1]++;
                if (r != c) ptr[c + 1]++;
            }
        for (int64_t i = 0; i < n; i++) ptr[i + 1] += ptr[i];
        std::vector<int> idx((size_t)ptr[n]), map((size_t)ptr[n]), next(ptr.begin(), ptr.end() - 1);
        for (int64_t c = 0; c < n; c++)
            for (uint64_t p = Pcolptr[c]; p < Pcolptr[c + 1]; p++) {
                const int64_t r = (int64_t)Prowval[p];
                idx[next[r]] = (int)c;
                map[next[r]++] = (int)p;
                if (r != c) {
                    idx[next[c]] = (int)r;
                    map[next[c]++] = (int)p;
                }
            }
        if ((rc = build_spmat(h->pool, h->Psym, (int)n, ptr, idx, map))) return rc;
    }
    // ---- A by rows (A x) and by columns (A' z)
    {
        std::vector<int> ptr((size_t)m + 1, 0);
        for (uint64_t p = 0; p < nnzA; p++) {
            if (Arowval[p] >= (uint64_t)m) return failk(CHIP_ERR_DIM, "A row index out of range");
            ptr[Arowval[p] + 1]++;
        }
        for (int64_t i = 0; i < m; i++) ptr[i + 1] += ptr[i];
        std::vector<int> idx((size_t)nnzA), map((size_t)nnzA), next(ptr.begin(), ptr.end() - 1);
        for (int64_t c = 0; c < n; c++)
            for (uint64_t p = Acolptr[c]; p < Acolptr[c + 1]; p++) {
                const int t = next[Arowval[p]]++;
                idx[t] = (int)c;
                map[t] = (int)p;
            }
        if ((rc = build_spmat(h->pool, h->Arow, (int)m, ptr, idx, map))) return rc;
        std::vector<int> cptr((size_t)n + 1), cidx((size_t)nnzA), cmap((size_t)nnzA);
        for (int64_t c = 0; c <= n; c++) cptr[c] = (int)Acolptr[c];
        for (uint64_t p = 0; p < nnzA; p++) {
            cidx[p] = (int)Arowval[p];
            cmap[p] = (int)p;
        }
        if ((rc = build_spmat(h->pool, h->Acol, (int)n, cptr, cidx, cmap))) return rc;
    }
    DevPool &pl = h->pool;
    if ((rc = pl.alloc(&h->src, std::max<size_t>({(size_t)nnzP, (size_t)nnzA, (size_t)n, (size_t)m})))) return rc;
    if ((rc = pl.alloc(&h->q, n)) || (rc = pl.alloc(&h->b, m))) return rc;
    if ((rc = pl.alloc(&h->x1, n)) || (rc = pl.alloc(&h->x2, n)) || (rc = pl.alloc(&h->workx, n)) ||
        (rc = pl.alloc(&h->wn, n)) || (rc = pl.alloc(&h->workx2, n)) || (rc = pl.alloc(&h->wn2, n)) ||
        (rc = pl.alloc(&h->negq, n)))
        return rc;
    if ((rc = pl.alloc(&h->z1, m)) || (rc = pl.alloc(&h->z2, m)) || (rc = pl.alloc(&h->workz, m)) ||
        (rc = pl.alloc(&h->work_conic, m)))
        return rc;
    if ((rc = pl.alloc(&h->dots, 16)) || (rc = pl.alloc(&h->scratch, (size_t)dev::multi_dot_scratch_doubles()))) return rc;
    CHIP_HIP(hipMemset(h->dots, 0, 16 * sizeof(double)));
    if ((rc = refresh_data(h.get(), nnzP ? Pnzval : nullptr, nnzA ? Anzval : nullptr, q, b))) return rc;
    *out = h.release();
    return CHIP_OK;
}

void chip_kktsystem_destroy(chip_kktsystem *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    delete h;
}

int32_t chip_kktsystem_update_data(chip_kktsystem *h, const double *P, const double *A, const double *q,
                                   const double *b) {
    if (!h) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    int rc = refresh_data(h, P, A, q, b);
    if (rc) return rc;
    if (P && (rc = chip_kkt_update_P(h->kkt, P))) return rc; // directldlkktsolver.rs:191-197
    if (A && (rc = chip_kkt_update_A(h->kkt, A))) return rc;
    return CHIP_OK;
}

// kktsystem.rs:264-279
int chip_kktsystem::solve_constant_rhs() {
    int rc = chip_kkt_setrhs_dev(kkt, negq, b);
    if (rc) return rc;
    rc = chip_kkt_solve_dev(kkt, x2, z2);
    if (rc != 1) return rc;
    // scalars of the tau denominator that only depend on (x2, z2)
    dot(0, q, x2, n);
    dot(1, b, z2, m);
    if (nnzP) {
        spmv(Psym, wn, nullptr, 1.0, x2);
        dot(2, x2, wn, n);
    }
    if ((rc = read_dots())) return rc;
    qx2 = hdots[0];
    bz2 = hdots[1];
    x2Px2 = nnzP ? hdots[2] : 0.0;
    return 1;
}

int32_t chip_kktsystem_update(chip_kktsystem *h) {
    if (!h) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    int rc = chip_kkt_update(h->kkt, nullptr);
    if (rc != 1) return rc;
    return h->solve_constant_rhs();
}

int32_t chip_kktsystem_solve(chip_kktsystem *h, chip_vars *lhs, const chip_vars *rhs, const chip_vars *var,
                             int32_t step_direction) {
    if (!h || !lhs || !rhs || !var) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    const int n = h->n, m = h->m;
    hipStream_t s = h->stream;
    int rc;
    // the reference's workx = rhs.x is read in place; conic = the constant term of  Hs dz + ds = -c
    // (the affine step's copy of s is read in place as well)
    const double *conic = var->s;
    if (step_direction != CHIP_STEP_AFFINE) {
        if ((rc = chip_kkt_ds_from_dz_offset_dev(h->kkt, h->work_conic, rhs->s, var->z))) return rc;
        conic = h->work_conic;
    }
    dev::waxpby(s, h->workz, 1.0, conic, -1.0, rhs->z, m);
    if ((rc = chip_kkt_setrhs_dev(h->kkt, rhs->x, h->workz))) return rc;
    rc = chip_kkt_solve_dev(h->kkt, h->x1, h->z1);
    if (rc != 1) return rc;
    // tau: xi = x / tau ; numerator and denominator of kktsystem.rs:170-186 (all four dot
    // products in one launch pair; the P terms vanish identically for an LP / SOCP without P)
    const double tau = var->tau, kappa = var->kappa;
    h->dot(3, h->q, h->x1, n);
    h->dot(4, h->b, h->z1, m);
    if (h->nnzP) {
        double *xi = h->workx, *xd = h->workx2;
        dev::waxpby(s, xi, 1.0 / tau, var->x, 0.0, nullptr, n);
        h->spmv(h->Psym, h->wn, nullptr, 1.0, h->x1); // P x1
        h->dot(5, xi, h->wn, n);                       // xi' P x1
        dev::waxpby(s, xd, -1.0, h->x2, 1.0, xi, n);   // xi - x2
        h->spmv(h->Psym, h->wn2, nullptr, 1.0, xd);
        h->dot(6, xd, h->wn2, n);                      // (xi - x2)' P (xi - x2)
    }
    if ((rc = h->read_dots())) return rc;
    const double xiPx1 = h->nnzP ? h->hdots[5] : 0.0, dPd = h->nnzP ? h->hdots[6] : 0.0;
    const double tau_num = rhs->tau - rhs->kappa / tau + h->hdots[3] + h->hdots[4] + 2.0 * xiPx1;
    double tau_den = kappa / tau - h->qx2 - h->bz2;
    tau_den += dPd - h->x2Px2;
    const double ltau = tau_num / tau_den;
    lhs->tau = ltau;
    dev::waxpby(s, lhs->x, 1.0, h->x1, ltau, h->x2, n);
    dev::waxpby(s, lhs->z, 1.0, h->z1, ltau, h->z2, m);
    // ds = -(Hs dz + c)
    if ((rc = chip_kkt_mul_Hs_dev(h->kkt, lhs->s, lhs->z))) return rc;
    dev::waxpby(s, lhs->s, -1.0, conic, -1.0, lhs->s, m);
    lhs->kappa = -(rhs->kappa + kappa * ltau) / tau;
    CHIP_HIP(hipGetLastError());
    return 1;
}

int32_t chip_kktsystem_solve_initial_point(chip_kktsystem *h, chip_vars *var) {
    if (!h || !var) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    const int n = h->n, m = h->m;
    hipStream_t s = h->stream;
    int rc;
    if (h->nnzP == 0) { // LP initialisation: [0; b] -> (x, -s), then [-q; 0] -> z
        if ((rc = h->zero(h->workx, n))) return rc;
        if ((rc = chip_kkt_setrhs_dev(h->kkt, h->workx, h->b))) return rc;
        rc = chip_kkt_solve_dev(h->kkt, var->x, var->s);
        dev::waxpby(s, var->s, -1.0, var->s, 0.0, nullptr, m); // negate (also on failure, as the reference)
        if (rc != 1) return rc;
        if ((rc = h->zero(h->workz, m))) return rc;
        if ((rc = chip_kkt_setrhs_dev(h->kkt, h->negq, h->workz))) return rc;
        rc = chip_kkt_solve_dev(h->kkt, nullptr, var->z);
        return rc;
    }
    if ((rc = chip_kkt_setrhs_dev(h->kkt, h->negq, h->b))) return rc;
    rc = chip_kkt_solve_dev(h->kkt, var->x, var->z);
    dev::waxpby(s, var->s, -1.0, var->z, 0.0, nullptr, m);
    return rc;
}

int32_t chip_residuals_update(chip_kktsystem *h, const chip_vars *var, double *rx, double *rz, double *rx_inf,
                              double *rz_inf, double *Px, double out5[5]) {
    return chip_residuals_update_norms(h, var, rx, rz, rx_inf, rz_inf, Px, out5, nullptr);
}
int32_t chip_residuals_update_norms(chip_kktsystem *h, const chip_vars *var, double *rx, double *rz, double *rx_inf,
                                    double *rz_inf, double *Px, double out5[5], double *norms5) {
    if (!h || !var || !out5) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    const int n = h->n, m = h->m;
    hipStream_t s = h->stream;
    int rc;
    h->dot(8, h->q, var->x, n);
    h->dot(9, h->b, var->z, m);
    h->dot(10, var->s, var->z, m);
    if (h->nnzP) {
        h->spmv(h->Psym, Px, nullptr, 1.0, var->x); // Px = P x (P symmetric)
        h->dot(11, var->x, Px, n);
    } else if ((rc = h->zero(Px, n)))
        return rc;
    h->spmv(h->Acol, rx_inf, nullptr, -1.0, var->z); // rx_inf = -A' z
    h->spmv(h->Arow, rz_inf, var->s, 1.0, var->x);   // rz_inf = A x + s
    dev::lin3(s, rx, 1.0, rx_inf, -1.0, Px, -var->tau, h->q, n); // rx = rx_inf - Px - q tau
    dev::waxpby(s, rz, 1.0, rz_inf, -var->tau, h->b, m);         // rz = rz_inf - b tau
    if (norms5) { // ||x||, ||z||, ||s||, ||rz||, ||rx|| (default/info.rs:142-165) in the same launch pair
        h->dot(0, var->x, var->x, n);
        h->dot(1, var->z, var->z, m);
        h->dot(2, var->s, var->s, m);
        h->dot(3, rz, rz, m);
        h->dot(4, rx, rx, n);
    }
    if ((rc = h->read_dots())) return rc;
    if (norms5)
        for (int k = 0; k < 5; k++) norms5[k] = std::sqrt(h->hdots[k]);
    const double qx = h->hdots[8], bz = h->hdots[9], sz = h->hdots[10], xPx = h->nnzP ? h->hdots[11] : 0.0;
    out5[0] = qx + bz + var->kappa + xPx / var->tau;
    out5[1] = qx;
    out5[2] = bz;
    out5[3] = sz;
    out5[4] = xPx;
    return CHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// DefaultVariables (default/variables.rs:58-256) on device-resident vectors: the vector algebra of
// the step either side of the KKT solve, so that only scalars cross the boundary per iteration.
// ---------------------------------------------------------------------------------------------
int32_t chip_variables_calc_mu(chip_kktsystem *h, const chip_vars *var, double dot_sz, double *mu_out) {
    if (!h || !var || !mu_out) return CHIP_ERR_ARG;
    int64_t deg = 0;
    int rc = chip_kkt_degree(h->kkt, &deg);
    if (rc) return rc;
    *mu_out = (dot_sz + var->tau * var->kappa) / (double)(deg + 1); // variables.rs:63-66
    return CHIP_OK;
}

// variables.rs:68-79
int32_t chip_variables_affine_step_rhs(chip_kktsystem *h, chip_vars *d, const double *rx, const double *rz,
                                       double rtau, const chip_vars *var) {
    if (!h || !d || !var) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    int rc;
    if ((rc = h->copy(d->x, rx, h->n))) return rc;
    if ((rc = h->copy(d->z, rz, h->m))) return rc;
    if ((rc = chip_kkt_affine_ds_dev(h->kkt, d->s, var->s))) return rc;
    d->tau = rtau;
    d->kappa = var->tau * var->kappa;
    return CHIP_OK;
}

// variables.rs:81-118 (d.s must already hold affine_ds, as in the reference)
int32_t chip_variables_combined_step_rhs(chip_kktsystem *h, chip_vars *d, const double *rx, const double *rz,
                                         double rtau, const chip_vars *var, chip_vars *step, double sigma,
                                         double mu, double mscale) {
    if (!h || !d || !var || !step) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    const int n = h->n, m = h->m;
    hipStream_t s = h->stream;
    const double sigma_mu = sigma * mu;
    dev::waxpby(s, d->x, 1.0 - sigma, rx, 0.0, nullptr, n);
    d->tau = (1.0 - sigma) * rtau;
    d->kappa = -sigma_mu + mscale * step->tau * step->kappa + var->tau * var->kappa;
    if (mscale != 1.0) dev::waxpby(s, step->z, mscale, step->z, 0.0, nullptr, m);
    int rc = chip_kkt_combined_ds_shift_dev(h->kkt, d->z, step->z, step->s, sigma_mu); // d.z is work
    if (rc) return rc;
    dev::waxpby(s, d->s, 1.0, d->s, 1.0, d->z, m);
    dev::waxpby(s, d->z, 1.0 - sigma, rz, 0.0, nullptr, m);
    CHIP_HIP(hipGetLastError());
    return CHIP_OK;
}

// variables.rs:120-160
int32_t chip_variables_calc_step_length(chip_kktsystem *h, const chip_vars *var, const chip_vars *step,
                                        int32_t step_direction, double max_step_fraction, double *alpha_out) {
    if (!h || !var || !step || !alpha_out) return CHIP_ERR_ARG;
    const double inf = std::numeric_limits<double>::max();
    const double a_tau = step->tau < 0.0 ? -var->tau / step->tau : inf;
    const double a_kap = step->kappa < 0.0 ? -var->kappa / step->kappa : inf;
    double alpha = std::min(std::min(a_tau, a_kap), 1.0);
    int rc = chip_kkt_step_length_dev(h->kkt, step->z, step->s, var->z, var->s, alpha, &alpha);
    if (rc) return rc;
    if (step_direction == CHIP_STEP_COMBINED) alpha *= max_step_fraction;
    *alpha_out = alpha;
    return CHIP_OK;
}

// variables.rs:162-168
int32_t chip_variables_add_step(chip_kktsystem *h, chip_vars *var, const chip_vars *step, double alpha) {
    if (!h || !var || !step) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    dev::waxpby(s, var->x, alpha, step->x, 1.0, var->x, h->n);
    dev::waxpby(s, var->s, alpha, step->s, 1.0, var->s, h->m);
    dev::waxpby(s, var->z, alpha, step->z, 1.0, var->z, h->m);
    var->tau += alpha * step->tau;
    var->kappa += alpha * step->kappa;
    CHIP_HIP(hipGetLastError());
    return CHIP_OK;
}

// _shift_to_cone_interior, variables.rs:231-261
static int shift_to_cone_interior(chip_kktsystem *h, double *z, int primal) {
    double min_margin = 0.0, pos_margin = 0.0;
    int rc = chip_kkt_margins_dev(h->kkt, z, &min_margin, &pos_margin);
    if (rc) return rc;
    int64_t deg = 0;
    if ((rc = chip_kkt_degree(h->kkt, &deg))) return rc;
    const double target = std::max(1.0, (pos_margin * 0.1) / (double)deg);
    if (min_margin <= 0.0) {
        if ((rc = chip_kkt_scaled_unit_shift_dev(h->kkt, z, -min_margin, primal))) return rc;
        return chip_kkt_scaled_unit_shift_dev(h->kkt, z, target, primal);
    }
    if (min_margin < target) return chip_kkt_scaled_unit_shift_dev(h->kkt, z, target - min_margin, primal);
    return chip_kkt_scaled_unit_shift_dev(h->kkt, z, 0.0, primal);
}
// variables.rs:170-176
int32_t chip_variables_symmetric_initialization(chip_kktsystem *h, chip_vars *var) {
    if (!h || !var) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    int rc;
    if ((rc = shift_to_cone_interior(h, var->s, 1))) return rc;
    if ((rc = shift_to_cone_interior(h, var->z, 0))) return rc;
    var->tau = 1.0;
    var->kappa = 1.0;
    return CHIP_OK;
}
// variables.rs:178-184
int32_t chip_variables_unit_initialization(chip_kktsystem *h, chip_vars *var) {
    if (!h || !var) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    int rc = chip_kkt_unit_initialization_dev(h->kkt, var->z, var->s);
    if (rc) return rc;
    if ((rc = h->zero(var->x, h->n))) return rc;
    var->tau = 1.0;
    var->kappa = 1.0;
    return CHIP_OK;
}

// variables.rs:205-227
int32_t chip_variables_barrier(chip_kktsystem *h, const chip_vars *var, const chip_vars *step, double alpha,
                               double *barrier_out) {
    if (!h || !var || !step || !barrier_out) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    int64_t deg = 0;
    int rc = chip_kkt_degree(h->kkt, &deg);
    if (rc) return rc;
    const double coef = (double)(deg + 1);
    const double cur_tau = var->tau + alpha * step->tau, cur_kappa = var->kappa + alpha * step->kappa;
    dev::dot_shifted(h->stream, var->z, var->s, step->z, step->s, alpha, h->m, h->dots + 12, h->scratch);
    if ((rc = h->read_dots())) return rc;
    const double mu = (h->hdots[12] + cur_tau * cur_kappa) / coef;
    auto logsafe = [](double v) { return v <= 0.0 ? -std::numeric_limits<double>::infinity() : std::log(v); };
    double barrier = coef * logsafe(mu) - logsafe(cur_tau) - logsafe(cur_kappa);
    double cb = 0.0;
    if ((rc = chip_kkt_compute_barrier_dev(h->kkt, var->z, var->s, step->z, step->s, alpha, &cb))) return rc;
    *barrier_out = barrier + cb;
    return CHIP_OK;
}

// variables.rs:229-239
int32_t chip_variables_rescale(chip_kktsystem *h, chip_vars *var) {
    if (!h || !var) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    const double inv = 1.0 / std::max(var->tau, var->kappa);
    hipStream_t s = h->stream;
    dev::waxpby(s, var->x, inv, var->x, 0.0, nullptr, h->n);
    dev::waxpby(s, var->z, inv, var->z, 0.0, nullptr, h->m);
    dev::waxpby(s, var->s, inv, var->s, 0.0, nullptr, h->m);
    var->tau *= inv;
    var->kappa *= inv;
    CHIP_HIP(hipGetLastError());
    return CHIP_OK;
}

// Euclidean norms of up to 8 device vectors with ONE host synchronisation: what DefaultInfo::update
// (default/info.rs:142-165) needs of x, z, s, rx, rz, ... when they live in HBM.  sqrt of the
// deterministic two-stage sum of squares (the reference's stable_norm, vecmath.rs:115-118, rescales
// by the largest entry; identical up to rounding away from overflow/underflow).
int32_t chip_vec_norms(chip_kktsystem *h, int32_t count, const double *const *vecs, const int64_t *lens,
                       double *out) {
    if (!h || count < 0 || count > 8 || (count && (!vecs || !lens || !out))) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(h->device));
    for (int k = 0; k < count; k++) h->dot(k, vecs[k], vecs[k], (int)lens[k]);
    int rc = h->read_dots();
    if (rc) return rc;
    for (int k = 0; k < count; k++) out[k] = std::sqrt(h->hdots[k]);
    return CHIP_OK;
}
