/* Test / measurement infrastructure only (see oracle/__init__.py) -- NOT the reference, NOT the product.
 *
 * A MULTI-THREADED host comparator for the KKT factor + solve path: the left-looking column LDL' of
 * src/qdldl/qdldl.rs:469-669 with its independent columns run in parallel -- the columns of one elimination-tree
 * level do not depend on each other -- on OpenMP threads, plus level-scheduled triangular solves (qdldl.rs:708-768)
 * and a two-pass symmetric matrix-vector product (csc/matrix_math.rs:178-208).  Same pivot rule (qdldl.rs:645-665),
 * same inputs (the oracle's permuted upper triangle, its pattern of L, its elimination tree), so its factors agree
 * with the oracle's up to rounding (checked by the caller).
 *
 * Why it exists: the reference's own multi-threaded engine (faer, ldlsolvers/faer_ldl.rs:99-157) cannot be built
 * here (no Rust toolchain), and bench.py's earlier stand-in (SuperLU) runs on one core.  This file says what the
 * host's cores can do on the SAME algorithm; it is labelled "port-mt" wherever it is quoted.  It has no supernodes:
 * on systems with dense fronts (configs 2, 5) a supernodal code such as faer would be far ahead of it.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

typedef struct {
    i64 n, nnzL;
    const i64 *Lp, *Li;      /* borrowed: pattern of L by columns (rows ascending) */
    i64 *Rp, *Rcol, *Rpos;   /* pattern of L by rows: column k, position of (j, k) in column k */
    i64 *lvl_ptr, *lvl_idx;  /* columns by elimination-tree level */
    i64 nlevels;
    double *Lx, *D, *Dinv;
    int nthreads;
    double **work;           /* per thread: dense accumulator of length n (all zero between columns) */
    i64 *Tp, *Ti;            /* row view of the upper triangle of A (A(j, i), i > j = initial value of L(i, j)) */
    i64 *Tpos;
} orc_mt;

void orc_mt_free(orc_mt *m) {
    if (!m) return;
    free(m->Rp); free(m->Rcol); free(m->Rpos); free(m->lvl_ptr); free(m->lvl_idx);
    free(m->Lx); free(m->D); free(m->Dinv); free(m->Tp); free(m->Ti); free(m->Tpos);
    if (m->work) for (int t = 0; t < m->nthreads; t++) free(m->work[t]);
    free(m->work);
    free(m);
}

/* Ap / Ai: the permuted upper triangle the factorisation will be given values for (pattern fixed) */
orc_mt *orc_mt_new(i64 n, const i64 *Lp, const i64 *Li, const i64 *etree, const i64 *Ap, const i64 *Ai, int nthreads) {
    orc_mt *m = (orc_mt *)calloc(1, sizeof(orc_mt));
    if (nthreads < 1) nthreads = 1;
    m->n = n; m->Lp = Lp; m->Li = Li; m->nnzL = Lp[n]; m->nthreads = nthreads;
    /* rows of L */
    m->Rp = (i64 *)calloc((size_t)n + 1, sizeof(i64));
    for (i64 q = 0; q < m->nnzL; q++) m->Rp[Li[q] + 1]++;
    for (i64 j = 0; j < n; j++) m->Rp[j + 1] += m->Rp[j];
    m->Rcol = (i64 *)malloc((size_t)(m->nnzL + 1) * sizeof(i64));
    m->Rpos = (i64 *)malloc((size_t)(m->nnzL + 1) * sizeof(i64));
    i64 *nx = (i64 *)malloc((size_t)(n + 1) * sizeof(i64));
    memcpy(nx, m->Rp, (size_t)n * sizeof(i64));
    for (i64 k = 0; k < n; k++)
        for (i64 q = Lp[k]; q < Lp[k + 1]; q++) {
            const i64 j = Li[q], t = nx[j]++;
            m->Rcol[t] = k;
            m->Rpos[t] = q;
        }
    /* levels: a node is one above its highest child; a column only needs columns of lower levels */
    i64 *lev = (i64 *)calloc((size_t)n + 1, sizeof(i64));
    i64 depth = n > 0 ? 1 : 0;
    for (i64 j = 0; j < n; j++) {
        const i64 p = etree[j];
        if (p >= 0 && p < n && lev[p] < lev[j] + 1) lev[p] = lev[j] + 1;
        if (lev[j] + 1 > depth) depth = lev[j] + 1;
    }
    m->nlevels = depth;
    m->lvl_ptr = (i64 *)calloc((size_t)depth + 1, sizeof(i64));
    m->lvl_idx = (i64 *)malloc((size_t)(n + 1) * sizeof(i64));
    for (i64 j = 0; j < n; j++) m->lvl_ptr[lev[j] + 1]++;
    for (i64 l = 0; l < depth; l++) m->lvl_ptr[l + 1] += m->lvl_ptr[l];
    memcpy(nx, m->lvl_ptr, (size_t)depth * sizeof(i64));
    for (i64 j = 0; j < n; j++) m->lvl_idx[nx[lev[j]]++] = j;
    free(lev);
    /* upper triangle of A by rows */
    const i64 nnzA = Ap[n];
    m->Tp = (i64 *)calloc((size_t)n + 1, sizeof(i64));
    for (i64 c = 0; c < n; c++)
        for (i64 p = Ap[c]; p < Ap[c + 1]; p++) m->Tp[Ai[p] + 1]++;
    for (i64 j = 0; j < n; j++) m->Tp[j + 1] += m->Tp[j];
    m->Ti = (i64 *)malloc((size_t)(nnzA + 1) * sizeof(i64));
    m->Tpos = (i64 *)malloc((size_t)(nnzA + 1) * sizeof(i64));
    memcpy(nx, m->Tp, (size_t)n * sizeof(i64));
    for (i64 c = 0; c < n; c++)
        for (i64 p = Ap[c]; p < Ap[c + 1]; p++) {
            const i64 t = nx[Ai[p]]++;
            m->Ti[t] = c;
            m->Tpos[t] = p;
        }
    free(nx);
    m->Lx = (double *)calloc((size_t)m->nnzL + 1, sizeof(double));
    m->D = (double *)calloc((size_t)n + 1, sizeof(double));
    m->Dinv = (double *)calloc((size_t)n + 1, sizeof(double));
    m->work = (double **)calloc((size_t)nthreads, sizeof(double *));
    for (int t = 0; t < nthreads; t++) m->work[t] = (double *)calloc((size_t)n + 1, sizeof(double));
    return m;
}

static inline void column(orc_mt *m, i64 j, const double *Ax, const int8_t *signs, double eps, double delta, double *y,
                          i64 *regcount, int *bad) {
    const i64 *Lp = m->Lp, *Li = m->Li;
    double d = 0.0;
    /* A(j, j) and A(j, i), i > j: row j of the upper triangle */
    for (i64 t = m->Tp[j]; t < m->Tp[j + 1]; t++) {
        const i64 c = m->Ti[t];
        if (c == j) d = Ax[m->Tpos[t]];
        else y[c] = Ax[m->Tpos[t]];
    }
    for (i64 t = m->Rp[j]; t < m->Rp[j + 1]; t++) {
        const i64 k = m->Rcol[t], p = m->Rpos[t];
        const double ljk = m->Lx[p], w = ljk * m->D[k];
        d -= ljk * w;
        for (i64 pp = p + 1; pp < Lp[k + 1]; pp++) y[Li[pp]] -= m->Lx[pp] * w;
    }
    const double s = (double)signs[j];
    if (d * s < eps) { /* qdldl.rs:645-651 */
        d = delta * s;
#pragma omp atomic
        (*regcount)++;
    }
    if (d == 0.0) *bad = 1;
    m->D[j] = d;
    const double dinv = 1.0 / d;
    m->Dinv[j] = dinv;
    for (i64 q = Lp[j]; q < Lp[j + 1]; q++) {
        m->Lx[q] = y[Li[q]] * dinv;
        y[Li[q]] = 0.0;
    }
}

/* a heavy column on a level with few columns: its contributions split over the threads, private accumulators */
static void column_split(orc_mt *m, i64 j, const double *Ax, const int8_t *signs, double eps, double delta, i64 *regcount,
                         int *bad) {
    const i64 *Lp = m->Lp, *Li = m->Li;
    const i64 rb = m->Rp[j], re = m->Rp[j + 1];
    double dsum = 0.0;
#pragma omp parallel num_threads(m->nthreads) reduction(+ : dsum)
    {
        double *y = m->work[omp_get_thread_num()];
#pragma omp for schedule(dynamic, 64)
        for (i64 t = rb; t < re; t++) {
            const i64 k = m->Rcol[t], p = m->Rpos[t];
            const double ljk = m->Lx[p], w = ljk * m->D[k];
            dsum += ljk * w;
            for (i64 pp = p + 1; pp < Lp[k + 1]; pp++) y[Li[pp]] -= m->Lx[pp] * w;
        }
    }
    double d = 0.0;
    double *y0 = m->work[0];
    for (i64 t = m->Tp[j]; t < m->Tp[j + 1]; t++) {
        const i64 c = m->Ti[t];
        if (c == j) d = Ax[m->Tpos[t]];
        else y0[c] += Ax[m->Tpos[t]];
    }
    d -= dsum;
    const double s = (double)signs[j];
    if (d * s < eps) {
        d = delta * s;
        (*regcount)++;
    }
    if (d == 0.0) *bad = 1;
    m->D[j] = d;
    const double dinv = 1.0 / d;
    m->Dinv[j] = dinv;
#pragma omp parallel for num_threads(m->nthreads) schedule(static)
    for (i64 q = Lp[j]; q < Lp[j + 1]; q++) {
        double a = 0.0;
        for (int t = 0; t < m->nthreads; t++) {
            a += m->work[t][Li[q]];
            m->work[t][Li[q]] = 0.0;
        }
        m->Lx[q] = a * dinv;
    }
}

/* returns 0 ok, 1 zero pivot / non-finite */
int orc_mt_factor(orc_mt *m, const double *Ax, const int8_t *signs, double eps, double delta, i64 *regcount_out) {
    i64 regcount = 0;
    int bad = 0;
    for (i64 l = 0; l < m->nlevels; l++) {
        const i64 b = m->lvl_ptr[l], e = m->lvl_ptr[l + 1];
        if (e - b >= 4 * (i64)m->nthreads || m->nthreads == 1) {
#pragma omp parallel num_threads(m->nthreads)
            {
                double *y = m->work[omp_get_thread_num()];
#pragma omp for schedule(dynamic, 256)
                for (i64 t = b; t < e; t++) column(m, m->lvl_idx[t], Ax, signs, eps, delta, y, &regcount, &bad);
            }
        } else {
            for (i64 t = b; t < e; t++) {
                const i64 j = m->lvl_idx[t];
                if (m->Rp[j + 1] - m->Rp[j] >= 512) column_split(m, j, Ax, signs, eps, delta, &regcount, &bad);
                else column(m, j, Ax, signs, eps, delta, m->work[0], &regcount, &bad);
            }
        }
    }
    for (i64 j = 0; j < m->n && !bad; j++)
        if (!isfinite(m->Dinv[j])) bad = 1;
    if (regcount_out) *regcount_out = regcount;
    return bad;
}

/* x <- (L D L')^-1 x, permuted numbering (qdldl.rs:755-768), level scheduled: forward by rows, backward by columns */
void orc_mt_solve(orc_mt *m, double *x) {
    const i64 *Lp = m->Lp, *Li = m->Li;
    for (i64 l = 1; l < m->nlevels; l++) {
        const i64 b = m->lvl_ptr[l], e = m->lvl_ptr[l + 1];
#pragma omp parallel for num_threads(m->nthreads) schedule(dynamic, 512) if (e - b >= 2048)
        for (i64 t = b; t < e; t++) {
            const i64 j = m->lvl_idx[t];
            double s = 0.0;
            for (i64 r = m->Rp[j]; r < m->Rp[j + 1]; r++) s += m->Lx[m->Rpos[r]] * x[m->Rcol[r]];
            x[j] -= s;
        }
    }
    for (i64 l = m->nlevels - 1; l >= 0; l--) {
        const i64 b = m->lvl_ptr[l], e = m->lvl_ptr[l + 1];
#pragma omp parallel for num_threads(m->nthreads) schedule(dynamic, 512) if (e - b >= 2048)
        for (i64 t = b; t < e; t++) {
            const i64 j = m->lvl_idx[t];
            double s = 0.0;
            for (i64 q = Lp[j]; q < Lp[j + 1]; q++) s += m->Lx[q] * x[Li[q]];
            x[j] = x[j] * m->Dinv[j] - s;
        }
    }
}

/* y = b - K x for the symmetric K given by its upper triangle (Ap, Ai, Ax) and the row view built in orc_mt_new */
void orc_mt_residual(orc_mt *m, const i64 *Ap, const i64 *Ai, const double *Ax, const double *x, const double *b, double *y) {
#pragma omp parallel for num_threads(m->nthreads) schedule(static)
    for (i64 j = 0; j < m->n; j++) {
        double s = 0.0;
        for (i64 p = Ap[j]; p < Ap[j + 1]; p++) s += Ax[p] * x[Ai[p]];          /* column j: rows i <= j */
        for (i64 t = m->Tp[j]; t < m->Tp[j + 1]; t++)                           /* row j: columns c > j   */
            if (m->Ti[t] != j) s += Ax[m->Tpos[t]] * x[m->Ti[t]];
        y[j] = b[j] - s;
    }
}

const double *orc_mt_Lx(const orc_mt *m) { return m->Lx; }
const double *orc_mt_D(const orc_mt *m) { return m->D; }
int orc_mt_max_threads(void) { return omp_get_max_threads(); }
