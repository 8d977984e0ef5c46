/*
 * oracle/qdldl_oracle.c  --  TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * A plain-C restatement of the reference's native QDLDL engine
 * (/root/reference/src/qdldl/qdldl.rs, Clarabel.rs v0.11.1) used ONLY as the
 * checker for the HIP path (tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg).  Nothing in the product (clarabel.rs_amd/) may link,
 * import or call this file.
 *
 * Parity status: PINNED for everything except the fill-reducing ordering.
 *   - solves / factors / etree / permute_symmetric are checked against every
 *     known-answer test of src/qdldl/test.rs and faer_ldl.rs:352-409
 *     (tests/test_oracle_kats.py).
 *   - the AMD ordering lives in the un-vendored crate `amd = "0.2.2"`
 *     (Cargo.toml:18); it is NOT restated here: the oracle always takes an
 *     explicit permutation (qdldl.rs:36-38 `QDLDLSettings.perm`), i.e. the
 *     ordering itself is "parity unpinned" (see DESIGN.md).
 *
 * Index type is int64_t (the reference uses usize); values are double.
 * Each function cites the reference lines it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_UNKNOWN ((int64_t)-1) /* QDLDL_UNKNOWN = usize::MAX, qdldl.rs:426 */

/* error codes mirror QDLDLError, qdldl.rs:10-26 */
enum {
    ORC_OK = 0,
    ORC_ERR_DIM = 1,
    ORC_ERR_EMPTY_COLUMN = 2,
    ORC_ERR_NOT_TRIU = 3,
    ORC_ERR_ZERO_PIVOT = 4,
    ORC_ERR_BAD_PERM = 5,
    ORC_ERR_SYMBOLIC = 6 /* solve() on a logical-only factorisation panics, qdldl.rs:118 */
};

typedef struct {
    int64_t n;
    /* permutation, qdldl.rs:72-77 */
    int64_t *perm, *iperm;
    /* L (strictly lower, unit diagonal implied), D, Dinv  qdldl.rs:78-83 */
    int64_t *Lp, *Li;
    double *Lx, *D, *Dinv;
    /* workspace, qdldl.rs:298-328 */
    int64_t *etree, *Lnz, *iwork;
    unsigned char *bwork;
    double *fwork;
    int64_t positive_inertia;
    /* triu(P A P') and the map A -> PAPt */
    int64_t *Ap, *Ai;
    double *Ax;
    int64_t nnzA;
    int64_t *AtoPAPt;
    signed char *Dsigns;
    int regularize_enable;
    double regularize_eps, regularize_delta;
    int64_t regularize_count;
    int is_symbolic;
    int64_t nnzL;
} orc_qdldl;

/* qdldl.rs:771-782 (_invperm).  NB the reference's duplicate test
 * `b[*j] == 0` cannot see a repeated index whose first occurrence was at
 * position 0; we use a proper "seen" array -- accepts a superset of nothing:
 * every valid permutation passes in both, every index >= n fails in both. */
int orc_invperm(int64_t n, const int64_t *p, int64_t *ip) {
    unsigned char *seen = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int64_t i = 0; i < n; i++) {
        int64_t j = p[i];
        if (j < 0 || j >= n || seen[j]) {
            free(seen);
            return ORC_ERR_BAD_PERM;
        }
        seen[j] = 1;
        ip[j] = i;
    }
    free(seen);
    return ORC_OK;
}

/* qdldl.rs:789-794  x[i] = b[p[i]] */
void orc_permute(int64_t n, double *x, const double *b, const int64_t *p) {
    for (int64_t i = 0; i < n; i++) x[i] = b[p[i]];
}
/* qdldl.rs:796-801  x[p[i]] = b[i] */
void orc_ipermute(int64_t n, double *x, const double *b, const int64_t *p) {
    for (int64_t i = 0; i < n; i++) x[p[i]] = b[i];
}

/* qdldl.rs:213-228 check_structure (is_square / is_triu / no empty column) */
int orc_check_structure(int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai) {
    if (m != n) return ORC_ERR_DIM;
    for (int64_t c = 0; c < n; c++)
        for (int64_t k = Ap[c]; k < Ap[c + 1]; k++)
            if (Ai[k] > c) return ORC_ERR_NOT_TRIU;
    for (int64_t c = 0; c < n; c++)
        if (!(Ap[c] < Ap[c + 1])) return ORC_ERR_EMPTY_COLUMN;
    return ORC_OK;
}

/* qdldl.rs:806-903 permute_symmetric + _permute_symmetric_inner.
 * P = (PAP')_triu with unsorted columns; AtoPAPt[k] = destination of entry k. */
void orc_permute_symmetric(int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                           const int64_t *iperm, int64_t *Pc, int64_t *Pr, double *Pv,
                           int64_t *AtoPAPt) {
    int64_t *num_entries = (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t));
    /* 1. count entries per column of P (:848-860) */
    for (int64_t colA = 0; colA < n; colA++) {
        int64_t colP = iperm[colA];
        for (int64_t k = Ap[colA]; k < Ap[colA + 1]; k++) {
            int64_t rowA = Ai[k];
            int64_t rowP = iperm[rowA];
            if (rowA <= colA) {
                int64_t col_idx = rowP > colP ? rowP : colP;
                num_entries[col_idx] += 1;
            }
        }
    }
    /* 2. cumsum (:864-871) */
    Pc[0] = 0;
    int64_t acc = 0;
    for (int64_t c = 0; c < n; c++) {
        acc += num_entries[c];
        Pc[c + 1] = acc;
    }
    for (int64_t c = 0; c < n; c++) num_entries[c] = Pc[c]; /* row_starts */
    /* 3. place (:877-902) */
    for (int64_t colA = 0; colA < n; colA++) {
        int64_t colP = iperm[colA];
        for (int64_t k = Ap[colA]; k < Ap[colA + 1]; k++) {
            int64_t rowA = Ai[k];
            if (rowA <= colA) {
                int64_t rowP = iperm[rowA];
                int64_t col_idx = colP > rowP ? colP : rowP;
                int64_t dst = num_entries[col_idx];
                Pr[dst] = colP < rowP ? colP : rowP;
                Pv[dst] = Ax[k];
                AtoPAPt[k] = dst;
                num_entries[col_idx] += 1;
            }
        }
    }
    free(num_entries);
}

/* qdldl.rs:433-464 _etree: elimination tree + column counts of L */
void orc_etree(int64_t n, const int64_t *Ap, const int64_t *Ai, int64_t *work, int64_t *Lnz,
               int64_t *etree) {
    for (int64_t i = 0; i < n; i++) {
        work[i] = 0;
        Lnz[i] = 0;
        etree[i] = ORC_UNKNOWN;
    }
    for (int64_t j = 0; j < n; j++) {
        work[j] = j;
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            int64_t i = Ai[p];
            while (work[i] != j) {
                if (etree[i] == ORC_UNKNOWN) etree[i] = j;
                Lnz[i] += 1;
                work[i] = j;
                i = etree[i];
            }
        }
    }
}

/* qdldl.rs:469-669 _factor_inner: up-looking LDL' with sign-based dynamic
 * regularisation.  Returns ORC_OK or ORC_ERR_ZERO_PIVOT; *pos_count gets the
 * number of positive pivots. */
int orc_factor_inner(int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                     int64_t *Lp, int64_t *Li, double *Lx, double *D, double *Dinv,
                     const int64_t *Lnz, const int64_t *etree, unsigned char *bwork,
                     int64_t *iwork, double *fwork, int logical, const signed char *Dsigns,
                     int reg_enable, double reg_eps, double reg_delta, int64_t *reg_count,
                     int64_t *pos_count) {
    *reg_count = 0;
    int64_t positive = 0;
    unsigned char *y_markers = bwork;
    int64_t *y_idx = iwork;
    int64_t *elim_buffer = iwork + n;
    int64_t *next_colspace = iwork + 2 * n;
    double *y_vals = fwork;

    Lp[0] = 0; /* :501-506 */
    for (int64_t i = 0; i < n; i++) Lp[i + 1] = Lp[i] + Lnz[i];
    for (int64_t i = 0; i < n; i++) { /* :511-514 */
        y_markers[i] = 0;
        y_vals[i] = 0.0;
        D[i] = 0.0;
        next_colspace[i] = Lp[i];
    }
    if (n == 0) {
        *pos_count = 0;
        return ORC_OK;
    }
    if (!logical) { /* :516-534 */
        D[0] = Ax[0];
        if (reg_enable) {
            double sign = (double)Dsigns[0];
            if (D[0] * sign < reg_eps) {
                D[0] = reg_delta * sign;
                *reg_count += 1;
            }
        }
        if (D[0] == 0.0) return ORC_ERR_ZERO_PIVOT;
        if (D[0] > 0.0) positive += 1;
        Dinv[0] = 1.0 / D[0];
    }
    for (int64_t k = 1; k < n; k++) { /* :538-666 */
        int64_t nnz_y = 0;
        for (int64_t p = Ap[k]; p < Ap[k + 1]; p++) { /* :552-599 */
            int64_t bidx = Ai[p];
            if (bidx == k) {
                D[k] = Ax[p];
                continue;
            }
            y_vals[bidx] = Ax[p];
            int64_t next_idx = bidx;
            if (!y_markers[next_idx]) {
                y_markers[next_idx] = 1;
                elim_buffer[0] = next_idx;
                int64_t nnz_e = 1;
                next_idx = etree[bidx];
                while (next_idx != ORC_UNKNOWN && next_idx < k) {
                    if (y_markers[next_idx]) break;
                    y_markers[next_idx] = 1;
                    elim_buffer[nnz_e] = next_idx;
                    next_idx = etree[next_idx];
                    nnz_e += 1;
                }
                while (nnz_e != 0) {
                    nnz_e -= 1;
                    y_idx[nnz_y] = elim_buffer[nnz_e];
                    nnz_y += 1;
                }
            }
        }
        for (int64_t ii = nnz_y - 1; ii >= 0; ii--) { /* :602-641 */
            int64_t cidx = y_idx[ii];
            int64_t tmp_idx = next_colspace[cidx];
            if (!logical) {
                double y_c = y_vals[cidx];
                for (int64_t q = Lp[cidx]; q < tmp_idx; q++) y_vals[Li[q]] -= Lx[q] * y_c;
                double l = y_c * Dinv[cidx];
                Lx[tmp_idx] = l;
                D[k] -= y_c * l;
            }
            Li[tmp_idx] = k;
            next_colspace[cidx] += 1;
            y_vals[cidx] = 0.0;
            y_markers[cidx] = 0;
        }
        if (!logical) { /* :643-665 */
            if (reg_enable) {
                double sign = (double)Dsigns[k];
                if (D[k] * sign < reg_eps) {
                    D[k] = reg_delta * sign;
                    *reg_count += 1;
                }
            }
            if (D[k] == 0.0) return ORC_ERR_ZERO_PIVOT;
            if (D[k] > 0.0) positive += 1;
            Dinv[k] = 1.0 / D[k];
        }
    }
    *pos_count = positive;
    return ORC_OK;
}

/* qdldl.rs:708-719 (I+L) x = b in place */
void orc_lsolve(int64_t n, const int64_t *Lp, const int64_t *Li, const double *Lx, double *x) {
    for (int64_t i = 0; i < n; i++) {
        double xi = x[i];
        for (int64_t q = Lp[i]; q < Lp[i + 1]; q++) x[Li[q]] -= Lx[q] * xi;
    }
}
/* qdldl.rs:722-734 (I+L)' x = b in place */
void orc_ltsolve(int64_t n, const int64_t *Lp, const int64_t *Li, const double *Lx, double *x) {
    for (int64_t i = n - 1; i >= 0; i--) {
        double s = 0.0;
        for (int64_t q = Lp[i]; q < Lp[i + 1]; q++) s += Lx[q] * x[Li[q]];
        x[i] -= s;
    }
}
/* qdldl.rs:737-752 D (I+L)' x = b in place (fused) */
void orc_dltsolve(int64_t n, const int64_t *Lp, const int64_t *Li, const double *Lx,
                  const double *Dinv, double *x) {
    for (int64_t i = n - 1; i >= 0; i--) {
        double s = 0.0;
        for (int64_t q = Lp[i]; q < Lp[i + 1]; q++) s += Lx[q] * x[Li[q]];
        x[i] *= Dinv[i];
        x[i] -= s;
    }
}
/* qdldl.rs:755-768 */
void orc_solve_factors(int64_t n, const int64_t *Lp, const int64_t *Li, const double *Lx,
                       const double *Dinv, double *b) {
    orc_lsolve(n, Lp, Li, Lx, b);
    orc_dltsolve(n, Lp, Li, Lx, Dinv, b);
}

/* qdldl.rs:382-424 _factor */
static int orc_factor(orc_qdldl *f, int logical) {
    if (logical) {
        for (int64_t i = 0; i < f->nnzL; i++) f->Lx[i] = 1.0;
        for (int64_t i = 0; i < f->n; i++) {
            f->D[i] = 1.0;
            f->Dinv[i] = 1.0;
        }
    }
    int64_t pos = 0;
    int rc = orc_factor_inner(f->n, f->Ap, f->Ai, f->Ax, f->Lp, f->Li, f->Lx, f->D, f->Dinv,
                              f->Lnz, f->etree, f->bwork, f->iwork, f->fwork, logical,
                              f->Dsigns, f->regularize_enable, f->regularize_eps,
                              f->regularize_delta, &f->regularize_count, &pos);
    if (rc != ORC_OK) return rc;
    f->positive_inertia = pos;
    return ORC_OK;
}

void orc_qdldl_free(orc_qdldl *f) {
    if (!f) return;
    free(f->perm); free(f->iperm); free(f->Lp); free(f->Li); free(f->Lx); free(f->D);
    free(f->Dinv); free(f->etree); free(f->Lnz); free(f->iwork); free(f->bwork);
    free(f->fwork); free(f->Ap); free(f->Ai); free(f->Ax); free(f->AtoPAPt); free(f->Dsigns);
    free(f);
}

/* qdldl.rs:95-102 + 230-295 (QDLDLFactorisation::new / _qdldl_new).
 * perm must be given (see header); Dsigns may be NULL (all +1). */
int orc_qdldl_new(orc_qdldl **out, int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai,
                  const double *Ax, const int64_t *perm, const signed char *Dsigns, int logical,
                  int reg_enable, double reg_eps, double reg_delta) {
    *out = NULL;
    int rc = orc_check_structure(m, n, Ap, Ai);
    if (rc != ORC_OK) return rc;
    orc_qdldl *f = (orc_qdldl *)calloc(1, sizeof(orc_qdldl));
    size_t nn = (size_t)(n > 0 ? n : 1);
    int64_t nnzA = Ap[n];
    f->n = n;
    f->nnzA = nnzA;
    f->perm = (int64_t *)malloc(nn * sizeof(int64_t));
    f->iperm = (int64_t *)malloc(nn * sizeof(int64_t));
    memcpy(f->perm, perm, (size_t)n * sizeof(int64_t));
    rc = orc_invperm(n, perm, f->iperm);
    if (rc != ORC_OK) {
        orc_qdldl_free(f);
        return rc;
    }
    f->Ap = (int64_t *)malloc((nn + 1) * sizeof(int64_t));
    f->Ai = (int64_t *)malloc((size_t)(nnzA > 0 ? nnzA : 1) * sizeof(int64_t));
    f->Ax = (double *)malloc((size_t)(nnzA > 0 ? nnzA : 1) * sizeof(double));
    f->AtoPAPt = (int64_t *)malloc((size_t)(nnzA > 0 ? nnzA : 1) * sizeof(int64_t));
    orc_permute_symmetric(n, Ap, Ai, Ax, f->iperm, f->Ap, f->Ai, f->Ax, f->AtoPAPt);
    /* permuted signs, :257-261 */
    f->Dsigns = (signed char *)malloc(nn);
    for (int64_t i = 0; i < n; i++) f->Dsigns[i] = Dsigns ? Dsigns[perm[i]] : 1;
    f->regularize_enable = reg_enable;
    f->regularize_eps = reg_eps;
    f->regularize_delta = reg_delta;
    /* workspace, :334-379 */
    f->etree = (int64_t *)malloc(nn * sizeof(int64_t));
    f->Lnz = (int64_t *)malloc(nn * sizeof(int64_t));
    f->iwork = (int64_t *)malloc(3 * nn * sizeof(int64_t));
    f->bwork = (unsigned char *)malloc(nn);
    f->fwork = (double *)malloc(nn * sizeof(double));
    orc_etree(n, f->Ap, f->Ai, f->iwork, f->Lnz, f->etree);
    int64_t sumLnz = 0;
    for (int64_t i = 0; i < n; i++) sumLnz += f->Lnz[i];
    f->nnzL = sumLnz;
    f->Lp = (int64_t *)malloc((nn + 1) * sizeof(int64_t));
    f->Li = (int64_t *)malloc((size_t)(sumLnz > 0 ? sumLnz : 1) * sizeof(int64_t));
    f->Lx = (double *)malloc((size_t)(sumLnz > 0 ? sumLnz : 1) * sizeof(double));
    f->D = (double *)calloc(nn, sizeof(double));
    f->Dinv = (double *)calloc(nn, sizeof(double));
    rc = orc_factor(f, logical);
    if (rc != ORC_OK) {
        orc_qdldl_free(f);
        return rc;
    }
    f->is_symbolic = logical;
    *out = f;
    return ORC_OK;
}

/* qdldl.rs:116-138 solve in place */
int orc_qdldl_solve(orc_qdldl *f, double *b) {
    if (f->is_symbolic) return ORC_ERR_SYMBOLIC;
    double *tmp = f->fwork;
    orc_permute(f->n, tmp, b, f->perm);
    orc_solve_factors(f->n, f->Lp, f->Li, f->Lx, f->Dinv, tmp);
    orc_ipermute(f->n, b, tmp, f->perm);
    return ORC_OK;
}
/* qdldl.rs:142-149 */
void orc_qdldl_update_values(orc_qdldl *f, const int64_t *idx, const double *v, int64_t k) {
    for (int64_t i = 0; i < k; i++) f->Ax[f->AtoPAPt[idx[i]]] = v[i];
}
/* qdldl.rs:153-160 */
void orc_qdldl_scale_values(orc_qdldl *f, const int64_t *idx, double s, int64_t k) {
    for (int64_t i = 0; i < k; i++) f->Ax[f->AtoPAPt[idx[i]]] *= s;
}
/* qdldl.rs:166-183 */
void orc_qdldl_offset_values(orc_qdldl *f, const int64_t *idx, double off,
                             const signed char *signs, int64_t k) {
    for (int64_t i = 0; i < k; i++) {
        if (signs[i] > 0) f->Ax[f->AtoPAPt[idx[i]]] += off;
        else if (signs[i] < 0) f->Ax[f->AtoPAPt[idx[i]]] -= off;
    }
}
/* qdldl.rs:188-200 */
int orc_qdldl_refactor(orc_qdldl *f) {
    f->is_symbolic = 0;
    return orc_factor(f, 0);
}
/* ldlsolvers/qdldl.rs:100-106: success == all Dinv finite */
int orc_qdldl_dinv_is_finite(const orc_qdldl *f) {
    for (int64_t i = 0; i < f->n; i++)
        if (!isfinite(f->Dinv[i])) return 0;
    return 1;
}

/* accessors for the python test harness */
int64_t orc_qdldl_n(const orc_qdldl *f) { return f->n; }
int64_t orc_qdldl_nnzL(const orc_qdldl *f) { return f->nnzL; }
int64_t orc_qdldl_nnzA(const orc_qdldl *f) { return f->nnzA; }
int64_t orc_qdldl_positive_inertia(const orc_qdldl *f) { return f->positive_inertia; }
int64_t orc_qdldl_regularize_count(const orc_qdldl *f) { return f->regularize_count; }
const int64_t *orc_qdldl_perm(const orc_qdldl *f) { return f->perm; }
const int64_t *orc_qdldl_Lp(const orc_qdldl *f) { return f->Lp; }
const int64_t *orc_qdldl_Li(const orc_qdldl *f) { return f->Li; }
const double *orc_qdldl_Lx(const orc_qdldl *f) { return f->Lx; }
const double *orc_qdldl_D(const orc_qdldl *f) { return f->D; }
const double *orc_qdldl_Dinv(const orc_qdldl *f) { return f->Dinv; }
const int64_t *orc_qdldl_etree(const orc_qdldl *f) { return f->etree; }
const int64_t *orc_qdldl_Lnz(const orc_qdldl *f) { return f->Lnz; }
const int64_t *orc_qdldl_Ap(const orc_qdldl *f) { return f->Ap; }
const int64_t *orc_qdldl_Ai(const orc_qdldl *f) { return f->Ai; }
const double *orc_qdldl_Ax(const orc_qdldl *f) { return f->Ax; }
const int64_t *orc_qdldl_AtoPAPt(const orc_qdldl *f) { return f->AtoPAPt; }
