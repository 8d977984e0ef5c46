/*
 * oracle/kkt_oracle.c  --  TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the reference's KKT layer that sits above the LDL
 * engine (Clarabel.rs v0.11.1, paths relative to /root/reference/src):
 *   - CSC block count/fill helpers            algebra/csc/utils.rs:16-307
 *   - assemble_kkt_matrix + LDLDataMap        solver/core/kktsolvers/direct/quasidef/kkt_assembly.rs:20-183
 *                                             .../quasidef/datamaps.rs:112-405
 *   - triu/tril symv                          algebra/csc/matrix_math.rs:178-208
 *   - norm_inf (NaN propagating), stable norm algebra/vecmath.rs:132-142,206-226
 *   - DirectLDLKKTSolver update / regularise / solve / iterative refinement
 *                                             .../quasidef/directldlkktsolver.rs:134-405
 *   - Zero / Nonnegative / SecondOrder cone scaling + Hs blocks
 *                                             solver/core/cones/{zerocone,nonnegativecone,socone}.rs
 * Used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline.
 * The LDL engine underneath is oracle/qdldl_oracle.c.
 *
 * Parity status: pinned by kkt_assembly.rs:185-355 (triu AND tril patterns),
 * matrix.rs:250-286 (symv, quad_form), vector.rs norm KATs; the cone Hs
 * blocks have no reference KATs and are cross-checked by the
 * NT identities Hs*z = s in tests/test_oracle_kats.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- from qdldl_oracle.c ------------------------------------------------ */
typedef struct orc_qdldl orc_qdldl;
int orc_qdldl_new(orc_qdldl **out, int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai,
                  const double *Ax, const int64_t *perm, const signed char *Dsigns, int logical,
                  int reg_enable, double reg_eps, double reg_delta);
void orc_qdldl_free(orc_qdldl *f);
int orc_qdldl_solve(orc_qdldl *f, double *b);
void orc_qdldl_update_values(orc_qdldl *f, const int64_t *idx, const double *v, int64_t k);
void orc_qdldl_scale_values(orc_qdldl *f, const int64_t *idx, double s, int64_t k);
int orc_qdldl_refactor(orc_qdldl *f);
int orc_qdldl_dinv_is_finite(const orc_qdldl *f);

/* cone tags (shared numbering with include/clarabel_hip.h) */
enum { CONE_ZERO = 0, CONE_NONNEG = 1, CONE_SOC = 2, CONE_EXP = 3, CONE_POW = 4,
       CONE_GENPOW = 5, CONE_PSDTRI = 6 };
enum { SHAPE_TRIU = 0, SHAPE_TRIL = 1 };

#define SOC_NO_EXPANSION_MAX_SIZE 4 /* socone.rs:46 */

/* ------------------------------------------------------------------------ */
/* vector helpers                                                            */
/* ------------------------------------------------------------------------ */
/* vecmath.rs:132-142 */
double orc_norm_inf(const double *v, int64_t n) {
    double out = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (isnan(v[i])) return NAN;
        double a = fabs(v[i]);
        out = out > a ? out : a; /* T::max */
    }
    return out;
}
/* vecmath.rs:206-226 stable_norm */
double orc_norm2(const double *v, int64_t n) {
    double scale = 0.0, sumsq = 1.0;
    for (int64_t i = 0; i < n; i++) {
        double xi = v[i];
        if (xi == 0.0) continue;
        double a = fabs(xi);
        if (scale < a) {
            double r = scale / a;
            sumsq = 1.0 + sumsq * r * r;
            scale = a;
        } else {
            double r = a / scale;
            sumsq = sumsq + r * r;
        }
    }
    return scale * sqrt(sumsq);
}
static double dotp(const double *a, const double *b, int64_t n) {
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}
static int all_finite(const double *v, int64_t n) {
    for (int64_t i = 0; i < n; i++)
        if (!isfinite(v[i])) return 0;
    return 1;
}

/* matrix_math.rs:178-208 _csc_symv_unsafe:  y = a*A*x + b*y, A stored as one
 * triangle (works identically for triu or tril) */
void orc_symv(int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, double *y,
              const double *x, double a, double b) {
    for (int64_t i = 0; i < n; i++) y[i] *= b;
    for (int64_t col = 0; col < n; col++) {
        double xcol = x[col];
        for (int64_t p = Ap[col]; p < Ap[col + 1]; p++) {
            int64_t row = Ai[p];
            double Aij = Ax[p];
            y[row] += a * Aij * xcol;
            if (row != col) y[col] += a * Aij * x[row];
        }
    }
}
/* matrix_math.rs:212-257 _csc_quad_form (triu) */
double orc_quad_form_triu(int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                          const double *y, const double *x) {
    double out = 0.0;
    for (int64_t col = 0; col < n; col++) {
        double t1 = 0.0, t2 = 0.0;
        for (int64_t p = Ap[col]; p < Ap[col + 1]; p++) {
            int64_t row = Ai[p];
            if (row < col) {
                t1 += Ax[p] * x[row];
                t2 += Ax[p] * y[row];
            } else if (row == col) {
                out += Ax[p] * x[col] * y[col];
            }
        }
        out += t1 * y[col] + t2 * x[col];
    }
    return out;
}

/* ------------------------------------------------------------------------ */
/* cones                                                                     */
/* ------------------------------------------------------------------------ */
typedef struct {
    int tag;
    int64_t dim;   /* SOC/NN/Zero: numel; PSD: matrix side n; GenPow: dim1 */
    int64_t dim2;  /* GenPow dim2 */
    int64_t numel;
    int hs_diag;   /* Hs_is_diagonal */
    int sparse;    /* is_sparse_expandable */
    int64_t pdim;  /* 2 for sparse SOC, 3 for GenPow */
    int64_t cone_start;  /* rng_cones[i].start */
    int64_t block_start; /* rng_blocks[i].start */
    int64_t block_len;
    /* scaling state */
    double *w, *lam;
    double eta;
    double *u, *v;
    double d;
    /* Exp / Pow cones (expcone.rs:18-30, powcone.rs:8-22) */
    double alpha;
    double Hs3[6], Hdual[6], grad3[3], zc[3];
    int ns_valid; /* Hs3 computed by update_scaling (else get_Hs leaves the caller's values) */
    /* GenPow (genpowcone.rs:10-47): alpha[dim1]; p[numel], q[dim1], r[dim2], d1[dim1], d2, grad[numel], z */
    double *ga, *gp, *gq, *gr, *gd1, *ggrad, *gz;
    double gd2, gmu, gpsi;
    int gvalid;
} orc_cone;

typedef struct {
    int64_t ncones;
    orc_cone *c;
    int64_t numel;     /* m */
    int64_t nblockvals; /* rng_blocks.last().end */
    int64_t pdim_total;
    int64_t nsparse;
} orc_cones;

static int64_t tri_number(int64_t k) { return k * (k + 1) / 2; }

void orc_cones_free(orc_cones *cs) {
    if (!cs) return;
    for (int64_t i = 0; i < cs->ncones; i++) {
        free(cs->c[i].w); free(cs->c[i].lam); free(cs->c[i].u); free(cs->c[i].v);
        free(cs->c[i].ga); free(cs->c[i].gp); free(cs->c[i].gq); free(cs->c[i].gr); free(cs->c[i].gd1);
        free(cs->c[i].ggrad); free(cs->c[i].gz);
    }
    free(cs->c);
    free(cs);
}

/* compositecone.rs:36-128 (new / make_rng_cones / make_rng_blocks) */
orc_cones *orc_cones_new_ex(int64_t ncones, const int32_t *tags, const int64_t *dims,
                            const int64_t *dims2, const double *alphas);
orc_cones *orc_cones_new(int64_t ncones, const int32_t *tags, const int64_t *dims,
                         const int64_t *dims2) {
    return orc_cones_new_ex(ncones, tags, dims, dims2, NULL);
}
orc_cones *orc_cones_new_ex(int64_t ncones, const int32_t *tags, const int64_t *dims,
                            const int64_t *dims2, const double *alphas) {
    orc_cones *cs = (orc_cones *)calloc(1, sizeof(orc_cones));
    cs->ncones = ncones;
    cs->c = (orc_cone *)calloc((size_t)(ncones > 0 ? ncones : 1), sizeof(orc_cone));
    int64_t cstart = 0, bstart = 0;
    for (int64_t i = 0; i < ncones; i++) {
        orc_cone *c = &cs->c[i];
        c->tag = tags[i];
        c->dim = dims[i];
        c->dim2 = dims2 ? dims2[i] : 0;
        c->alpha = alphas ? alphas[i] : 0.5;
        switch (c->tag) {
        case CONE_ZERO: case CONE_NONNEG:
            c->numel = c->dim; c->hs_diag = 1; c->sparse = 0; break;
        case CONE_SOC:
            c->numel = c->dim;
            c->sparse = c->dim > SOC_NO_EXPANSION_MAX_SIZE; /* socone.rs:54-60 */
            c->hs_diag = c->sparse;                         /* socone.rs:213-215 */
            c->pdim = c->sparse ? 2 : 0;
            break;
        case CONE_EXP: case CONE_POW:
            c->numel = 3; c->hs_diag = 0; c->sparse = 0; break;
        case CONE_GENPOW:
            c->numel = c->dim + c->dim2; c->hs_diag = 1; c->sparse = 1; c->pdim = 3; break;
        case CONE_PSDTRI:
            c->numel = tri_number(c->dim); c->hs_diag = 0; c->sparse = 0; break;
        default: c->numel = 0;
        }
        c->cone_start = cstart;
        c->block_start = bstart;
        c->block_len = c->hs_diag ? c->numel : tri_number(c->numel);
        cstart += c->numel;
        bstart += c->block_len;
        if (c->sparse) { cs->pdim_total += c->pdim; cs->nsparse += 1; }
        if (c->tag == CONE_NONNEG || c->tag == CONE_SOC) {
            c->w = (double *)calloc((size_t)c->numel, sizeof(double));
            c->lam = (double *)calloc((size_t)c->numel, sizeof(double));
        }
        if (c->tag == CONE_SOC && c->sparse) {
            c->u = (double *)calloc((size_t)c->numel, sizeof(double));
            c->v = (double *)calloc((size_t)c->numel, sizeof(double));
            c->d = 1.0;
        }
        if (c->tag == CONE_GENPOW) {
            c->ga = (double *)calloc((size_t)c->dim + 1, sizeof(double));
            c->gq = (double *)calloc((size_t)c->dim + 1, sizeof(double));
            c->gd1 = (double *)calloc((size_t)c->dim + 1, sizeof(double));
            c->gr = (double *)calloc((size_t)c->dim2 + 1, sizeof(double));
            c->gp = (double *)calloc((size_t)c->numel + 1, sizeof(double));
            c->ggrad = (double *)calloc((size_t)c->numel + 1, sizeof(double));
            c->gz = (double *)calloc((size_t)c->numel + 1, sizeof(double));
            for (int64_t k = 0; k < c->dim; k++) c->ga[k] = 1.0 / (double)c->dim; /* until set_genpow_alpha */
            c->gmu = 1.0;
            c->gpsi = (double)c->dim;
        }
    }
    cs->numel = cstart;
    cs->nblockvals = bstart;
    return cs;
}
int64_t orc_cones_numel(const orc_cones *cs) { return cs->numel; }
int64_t orc_cones_nblockvals(const orc_cones *cs) { return cs->nblockvals; }
int64_t orc_cones_pdim(const orc_cones *cs) { return cs->pdim_total; }

/* socone.rs:388-406 */
static double soc_residual(const double *z, int64_t n) {
    double z1 = orc_norm2(z + 1, n - 1);
    return (z[0] - z1) * (z[0] + z1);
}
static double sqrt_soc_residual(const double *z, int64_t n) {
    double r = soc_residual(z, n);
    return r > 0.0 ? sqrt(r) : 0.0;
}

/* socone.rs:134-211 SecondOrderCone::update_scaling */
static int soc_update_scaling(orc_cone *c, const double *s, const double *z) {
    int64_t n = c->numel;
    double zscale = sqrt_soc_residual(z, n);
    double sscale = sqrt_soc_residual(s, n);
    if (zscale == 0.0 || sscale == 0.0) return 0;
    c->eta = sqrt(sscale / zscale);
    double *w = c->w;
    double rs = 1.0 / sscale; /* w.scale(sscale.recip()) */
    for (int64_t i = 0; i < n; i++) w[i] = s[i] * rs;
    w[0] += z[0] / zscale;
    double mrz = -(1.0 / zscale); /* axpby(-zscale.recip(), z1, 1) */
    for (int64_t i = 1; i < n; i++) w[i] = mrz * z[i] + 1.0 * w[i];
    double wscale = sqrt_soc_residual(w, n);
    if (wscale == 0.0) return 0;
    double rw = 1.0 / wscale;
    for (int64_t i = 0; i < n; i++) w[i] *= rw;
    double w1sq = dotp(w + 1, w + 1, n - 1);
    w[0] = sqrt(1.0 + w1sq);
    /* lambda, :174-184 */
    double gamma = 0.5 * wscale;
    double *lam = c->lam;
    lam[0] = gamma;
    double ca = (gamma + z[0] / zscale) / sscale;
    double cb = (gamma + s[0] / sscale) / zscale;
    for (int64_t i = 1; i < n; i++) lam[i] = ca * s[i] + cb * z[i];
    double sc = 1.0 / (s[0] / sscale + z[0] / zscale + 2.0 * gamma);
    for (int64_t i = 1; i < n; i++) lam[i] *= sc;
    double sq = sqrt(sscale * zscale);
    for (int64_t i = 0; i < n; i++) lam[i] *= sq;
    /* sparse expansion terms, :187-208 */
    if (c->sparse) {
        double alpha = 2.0 * w[0];
        double wsq = w[0] * w[0] + w1sq;
        double wsqinv = 1.0 / wsq;
        c->d = 0.5 * wsqinv;
        double u0 = sqrt(wsq - c->d);
        double u1 = alpha / u0;
        double v0 = 0.0;
        double v1 = sqrt(2.0 * (2.0 + wsqinv) / (2.0 * wsq - wsqinv));
        c->u[0] = u0;
        for (int64_t i = 1; i < n; i++) c->u[i] = u1 * w[i] + 0.0 * c->u[i];
        c->v[0] = v0;
        for (int64_t i = 1; i < n; i++) c->v[i] = v1 * w[i] + 0.0 * c->v[i];
    }
    return 1;
}

/* ---- Exponential / Power cones ------------------------------------------- */
static double logsafe(double v) { return v <= 0.0 ? -INFINITY : log(v); } /* scalarmath.rs:14-20 */
/* packed triu 3x3: [00,01,11,02,12,22] (dense3x3/core.rs:25-57) */
static void sym3_mul(const double *H, double *y, const double *x) {
    y[0] = (H[0] * x[0]) + (H[1] * x[1]) + (H[3] * x[2]);
    y[1] = (H[1] * x[0]) + (H[2] * x[1]) + (H[4] * x[2]);
    y[2] = (H[3] * x[0]) + (H[4] * x[1]) + (H[5] * x[2]);
}
static double sym3_quad(const double *H, const double *y, const double *x) {
    double out = 0.0;
    out += y[0] * (H[0] * x[0] + H[1] * x[1] + H[3] * x[2]);
    out += y[1] * (H[1] * x[0] + H[2] * x[1] + H[4] * x[2]);
    out += y[2] * (H[3] * x[0] + H[4] * x[1] + H[5] * x[2]);
    return out;
}
static double sym3_norm_fro(const double *d) {
    double sumsq = 0.0;
    sumsq += d[0] * d[0] + d[2] * d[2] + d[5] * d[5];
    sumsq += (d[1] * d[1] + d[3] * d[3] + d[4] * d[4]) * 2.0;
    return sqrt(sumsq);
}
static const int SYM3_IDX[3][3] = {{0, 1, 3}, {1, 2, 4}, {3, 4, 5}};

/* expcone.rs:396-458 */
double orc_wright_omega(double z) {
    double p, w;
    if (z < 1.0 + M_PI) {
        double zm1 = z - 1.0;
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
        double logz = logsafe(z), zinv = 1.0 / z;
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
        double wp1 = w + 1.0;
        double t = wp1 * (wp1 + (r * 2.0) / 3.0);
        w *= 1.0 + (r / wp1) * (t - r * 0.5) / (t - r);
        double r4 = r * r * r * r;
        double wp16 = wp1 * wp1 * wp1 * wp1 * wp1 * wp1;
        r = (w * w * 2.0 - w * 8.0 - 1.0) / (wp16 * 72.0) * r4;
    }
    return w;
}
/* expcone.rs:330-353 */
static void exp_update_dual_grad_H(orc_cone *c, const double *z) {
    double *grad = c->grad3, *H = c->Hdual;
    double l = logsafe(-z[2] / z[0]);
    double r = -z[0] * l - z[0] + z[1];
    double c2 = 1.0 / r;
    grad[0] = c2 * l - 1.0 / z[0];
    grad[1] = -c2;
    grad[2] = (c2 * z[0] - 1.0) / z[2];
    H[0] = (r * r - z[0] * r + l * l * z[0] * z[0]) / (r * z[0] * z[0] * r);
    H[1] = -l / (r * r);
    H[2] = 1.0 / (r * r);
    H[3] = (z[1] - z[0]) / (r * r * z[2]);
    H[4] = -z[0] / (r * r * z[2]);
    H[5] = (r * r - z[0] * r + z[0] * z[0]) / (r * r * z[2] * z[2]);
}
/* expcone.rs:361-373 */
static void exp_gradient_primal(const double *s, double *g) {
    double om = orc_wright_omega(1.0 - s[0] / s[1] - logsafe(s[1] / s[2]));
    g[0] = 1.0 / ((om - 1.0) * s[1]);
    g[1] = g[0] + g[0] * logsafe(om * s[1] / s[2]) - 1.0 / s[1];
    g[2] = om / ((1.0 - om) * s[2]);
}
/* powcone.rs:353-386 */
static void pow_update_dual_grad_H(orc_cone *c, const double *z) {
    double *H = c->Hdual, *g = c->grad3;
    double a = c->alpha;
    double phi = pow(z[0] / a, 2.0 * a) * pow(z[1] / (1.0 - a), 2.0 - 2.0 * a);
    double psi = phi - z[2] * z[2];
    g[0] = 2.0 * a * phi / (z[0] * psi);
    g[1] = 2.0 * (1.0 - a) * phi / (z[1] * psi);
    g[2] = -2.0 * z[2] / psi;
    H[0] = g[0] * g[0] - 2.0 * a * (2.0 * a - 1.0) * phi / (z[0] * z[0] * psi) + (1.0 - a) / (z[0] * z[0]);
    H[1] = g[0] * g[1] - 4.0 * a * (1.0 - a) * phi / (z[0] * z[1] * psi);
    H[2] = g[1] * g[1] - 2.0 * (1.0 - a) * (1.0 - 2.0 * a) * phi / (z[1] * z[1] * psi) + a / (z[1] * z[1]);
    H[3] = g[0] * g[2];
    H[4] = g[1] * g[2];
    H[5] = g[2] * g[2] + 2.0 / psi;
    g[0] = -2.0 * a * phi / (z[0] * psi) - (1.0 - a) / z[0];
    g[1] = -2.0 * (1.0 - a) * phi / (z[1] * psi) - a / z[1];
    g[2] = 2.0 * z[2] / psi;
}
/* powcone.rs:447-491 + nonsymmetric_common.rs:193-219 */
static double pow_newton_raphson(double s3, double phi, double a) {
    const double eps = 2.220446049250313e-16;
    double x = -1.0 / s3 + (s3 * 2.0 + sqrt((phi * phi) / (s3 * s3) + phi * 3.0)) / (phi - s3 * s3);
    double t0 = -2.0 * a * logsafe(a) - 2.0 * (1.0 - a) * logsafe(1.0 - a);
    for (int iter = 0; iter < 100; iter++) {
        double t1 = x * x, t2 = (2.0 * x) / s3;
        double dfdx = (a * a * 2.0) / (a * x + (1.0 + a) / s3) +
                      ((1.0 - a) * 2.0) * (1.0 - a) / ((1.0 - a) * x + (2.0 - a) / s3) -
                      ((x + 1.0 / s3) * 2.0) / (t1 + t2);
        double t2b = (x * 2.0) / s3;
        double f = 2.0 * a * logsafe(2.0 * a * t1 + (1.0 + a) * t2b) +
                   2.0 * (1.0 - a) * logsafe(2.0 * (1.0 - a) * t1 + (2.0 - a) * t2b) - logsafe(phi) -
                   logsafe(t1 + t2b) - 2.0 * logsafe(t2b) + t0;
        double dx = -f / dfdx;
        if (dx < eps || fabs(dx / x) < sqrt(eps) || fabs(dfdx) < eps) break;
        x += dx;
    }
    return x;
}
/* powcone.rs:394-420 */
static void pow_gradient_primal(const orc_cone *c, const double *s, double *g) {
    const double eps = 2.220446049250313e-16;
    double a = c->alpha;
    double phi = pow(s[0], 2.0 * a) * pow(s[1], 2.0 - a * 2.0);
    double abs_s = fabs(s[2]);
    if (abs_s > eps) {
        g[2] = pow_newton_raphson(abs_s, phi, a);
        if (s[2] < 0.0) g[2] = -g[2];
        g[0] = -(a * g[2] * s[2] + 1.0 + a) / s[0];
        g[1] = -((1.0 - a) * g[2] * s[2] + 2.0 - a) / s[1];
    } else {
        g[2] = 0.0;
        g[0] = -(1.0 + a) / s[0];
        g[1] = -(2.0 - a) / s[1];
    }
}
/* nonsymmetric_common.rs:53-143 update_Hs; strategy: 0 = PrimalDual, 1 = Dual (core/solver.rs:77-80) */
static void ns3_update_Hs(orc_cone *c, const double *s, const double *z, double mu_in, int strategy) {
    const double eps = 2.220446049250313e-16;
    double *Hd = c->Hdual, *Hs = c->Hs3;
    if (strategy == 1) {
        for (int i = 0; i < 6; i++) Hs[i] = mu_in * Hd[i];
        return;
    }
    double zt[3];
    if (c->tag == CONE_EXP) exp_gradient_primal(s, zt);
    else pow_gradient_primal(c, s, zt);
    const double *st = c->grad3;
    double ds[3], tmp[3], dz[3];
    double dot_sz = dotp(s, z, 3);
    double mu = dot_sz / 3.0;
    double mut = dotp(st, zt, 3) / 3.0;
    for (int i = 0; i < 3; i++) {
        ds[i] = s[i] + mu * st[i];
        dz[i] = z[i] + mu * zt[i];
    }
    double dot_dsz = dotp(ds, dz, 3);
    double de1 = mu * mut - 1.0;
    double de2 = sym3_quad(Hd, zt, zt) - 3.0 * mut * mut;
    if (fabs(de1) > sqrt(eps) && fabs(de2) > eps && dot_sz > 0.0 && dot_dsz > 0.0) {
        sym3_mul(Hd, tmp, zt);
        for (int i = 0; i < 3; i++) tmp[i] = mut * st[i] - tmp[i];
        for (int i = 0; i < 6; i++) Hs[i] = Hd[i];
        for (int i = 0; i < 3; i++)
            for (int j = i; j < 3; j++) Hs[SYM3_IDX[i][j]] -= st[i] * st[j] / 3.0 + tmp[i] * tmp[j] / de2;
        double t = mu * sym3_norm_fro(Hs);
        double ax[3];
        ax[0] = z[1] * zt[2] - z[2] * zt[1];
        ax[1] = z[2] * zt[0] - z[0] * zt[2];
        ax[2] = z[0] * zt[1] - z[1] * zt[0];
        double nrm = orc_norm2(ax, 3); /* normalize(), vecmath.rs:74-81 */
        if (nrm != 0.0) {
            double rn = 1.0 / nrm;
            for (int i = 0; i < 3; i++) ax[i] *= rn;
        }
        for (int i = 0; i < 3; i++)
            for (int j = i; j < 3; j++)
                Hs[SYM3_IDX[i][j]] = s[i] * s[j] / dot_sz + ds[i] * ds[j] / dot_dsz + t * ax[i] * ax[j];
    } else {
        for (int i = 0; i < 6; i++) Hs[i] = mu * Hd[i];
    }
}

/* ---- Generalised power cone (genpowcone.rs) -------------------------------------------------- */
/* GenPowerCone::new (genpowcone.rs:21-63): the powers alpha[dim1] of cone i; psi = 1 / sum alpha^2 */
int orc_cones_set_genpow_alpha(orc_cones *cs, int64_t i, const double *alpha) {
    if (i < 0 || i >= cs->ncones || cs->c[i].tag != CONE_GENPOW) return -1;
    orc_cone *c = &cs->c[i];
    double sq = 0.0;
    for (int64_t k = 0; k < c->dim; k++) { c->ga[k] = alpha[k]; sq += alpha[k] * alpha[k]; }
    c->gpsi = 1.0 / sq;
    return 0;
}
/* genpowcone.rs:361-401 */
static void genpow_update_dual_grad_H(orc_cone *c, const double *z) {
    const int64_t d1n = c->dim, d2n = c->dim2;
    double phi = 1.0;
    for (int64_t k = 0; k < d1n; k++) phi = phi * pow(z[k] / c->ga[k], 2.0 * c->ga[k]);
    double norm2w = 0.0;
    for (int64_t k = 0; k < d2n; k++) norm2w += z[d1n + k] * z[d1n + k];
    const double zeta = phi - norm2w;
    double *tau = c->gq;
    for (int64_t k = 0; k < d1n; k++) {
        tau[k] = 2.0 * c->ga[k] / z[k];
        c->ggrad[k] = -tau[k] * phi / zeta - (1.0 - c->ga[k]) / z[k];
    }
    for (int64_t k = 0; k < d2n; k++) c->ggrad[d1n + k] = (2.0 / zeta) * z[d1n + k];
    const double p0 = sqrt(phi * (phi + norm2w) / 2.0);
    const double p1 = -2.0 * phi / p0;
    const double q0 = sqrt(zeta * phi / 2.0);
    const double r1 = 2.0 * sqrt(zeta / (phi + norm2w));
    for (int64_t k = 0; k < d1n; k++)
        c->gd1[k] = tau[k] * phi / (zeta * z[k]) + (1.0 - c->ga[k]) / (z[k] * z[k]);
    c->gd2 = 2.0 / zeta;
    for (int64_t k = 0; k < d1n; k++) c->gp[k] = (p0 / zeta) * tau[k];
    for (int64_t k = 0; k < d2n; k++) c->gp[d1n + k] = (p1 / zeta) * z[d1n + k];
    for (int64_t k = 0; k < d1n; k++) c->gq[k] *= q0 / zeta;
    for (int64_t k = 0; k < d2n; k++) c->gr[k] = (r1 / zeta) * z[d1n + k];
}
/* genpowcone.rs:279-317 */
static int genpow_is_primal_feasible(const orc_cone *c, const double *s) {
    for (int64_t k = 0; k < c->dim; k++) if (!(s[k] > 0.0)) return 0;
    double res = 0.0;
    for (int64_t k = 0; k < c->dim; k++) res = res + 2.0 * c->ga[k] * logsafe(s[k]);
    double sq = 0.0;
    for (int64_t k = 0; k < c->dim2; k++) sq += s[c->dim + k] * s[c->dim + k];
    return exp(res) - sq > 0.0;
}
static int genpow_is_dual_feasible(const orc_cone *c, const double *z) {
    for (int64_t k = 0; k < c->dim; k++) if (!(z[k] > 0.0)) return 0;
    double res = 0.0;
    for (int64_t k = 0; k < c->dim; k++) res = res + 2.0 * c->ga[k] * logsafe(z[k] / c->ga[k]);
    double sq = 0.0;
    for (int64_t k = 0; k < c->dim2; k++) sq += z[c->dim + k] * z[c->dim + k];
    return exp(res) - sq > 0.0;
}
/* genpowcone.rs:333-356 */
static double genpow_barrier_dual(const orc_cone *c, const double *z) {
    double res = 0.0;
    for (int64_t k = 0; k < c->dim; k++) res += 2.0 * c->ga[k] * logsafe(z[k] / c->ga[k]);
    double sq = 0.0;
    for (int64_t k = 0; k < c->dim2; k++) sq += z[c->dim + k] * z[c->dim + k];
    res = exp(res) - sq;
    double barrier = -logsafe(res);
    for (int64_t k = 0; k < c->dim; k++) barrier -= logsafe(z[k]) * (1.0 - c->ga[k]);
    return barrier;
}
/* genpowcone.rs:409-485 (gradient_primal, _newton_raphson_genpowcone; nonsymmetric_common.rs:193-219).
 * NB the reference scales the w-part of the gradient with data.r (the dual-side vector), as written
 * at genpowcone.rs:427; restated as is */
static void genpow_gradient_primal(const orc_cone *c, double *g, const double *s) {
    const int64_t d1n = c->dim, d2n = c->dim2;
    const double eps = 2.220446049250313e-16;
    double phi = 1.0;
    for (int64_t k = 0; k < d1n; k++) phi = phi * pow(s[k], 2.0 * c->ga[k]);
    const double *pv = s, *rv = s + d1n;
    const double norm_r = orc_norm2(rv, d2n);
    if (norm_r > eps) {
        const double psi = c->gpsi;
        double x = -(1.0 / norm_r) + (psi * norm_r + sqrt((phi / norm_r / norm_r + psi * psi - 1.0) * phi)) /
                                         (phi - norm_r * norm_r);
        for (int iter = 0; iter < 100; iter++) {
            double dfdx = -(2.0 * x + 2.0 / norm_r) / (x * x + 2.0 * x / norm_r);
            for (int64_t k = 0; k < d1n; k++)
                dfdx = dfdx + 2.0 * c->ga[k] * norm_r / (norm_r * x + (1.0 + c->ga[k]) / c->ga[k]);
            double f = -logsafe(2.0 * x / norm_r + x * x);
            for (int64_t k = 0; k < d1n; k++)
                f = f + 2.0 * c->ga[k] * (logsafe(x * norm_r + (1.0 + c->ga[k]) / c->ga[k]) - logsafe(pv[k]));
            const double dx = -f / dfdx;
            if (dx < eps || fabs(dx / x) < sqrt(eps) || fabs(dfdx) < eps) break;
            x += dx;
        }
        const double g1 = x;
        for (int64_t k = 0; k < d2n; k++) g[d1n + k] = (g1 / norm_r) * c->gr[k];
        for (int64_t k = 0; k < d1n; k++) g[k] = -(1.0 + c->ga[k] + c->ga[k] * g1 * norm_r) / pv[k];
    } else {
        for (int64_t k = 0; k < d2n; k++) g[d1n + k] = 0.0;
        for (int64_t k = 0; k < d1n; k++) g[k] = -(1.0 + c->ga[k]) / pv[k];
    }
}
static double genpow_barrier_primal(const orc_cone *c, const double *s) {
    double *g = (double *)malloc((size_t)(c->numel + 1) * sizeof(double));
    genpow_gradient_primal(c, g, s);
    for (int64_t k = 0; k < c->numel; k++) g[k] = -g[k];
    const double out = -genpow_barrier_dual(c, g) - (double)(c->dim + 1);
    free(g);
    return out;
}
static double genpow_backtrack(const orc_cone *c, const double *dq, const double *q, double a_init, double a_min,
                               double step, int dual) {
    double alpha = a_init;
    double *w = (double *)malloc((size_t)(c->numel + 1) * sizeof(double));
    for (;;) {
        for (int64_t i = 0; i < c->numel; i++) w[i] = 1.0 * q[i] + alpha * dq[i];
        if (dual ? genpow_is_dual_feasible(c, w) : genpow_is_primal_feasible(c, w)) break;
        alpha *= step;
        if (alpha < a_min) { alpha = 0.0; break; }
    }
    free(w);
    return alpha;
}

/* compositecone.rs:226-243 update_scaling (Zero/NN/SOC only; other cone types
 * keep whatever Hs the caller stored with orc_kkt_set_hs_override) */
int orc_cones_update_scaling_ex(orc_cones *cs, const double *s, const double *z, double mu, int strategy);
int orc_cones_update_scaling(orc_cones *cs, const double *s, const double *z) {
    return orc_cones_update_scaling_ex(cs, s, z, 1.0, 0);
}
int orc_cones_update_scaling_ex(orc_cones *cs, const double *s, const double *z, double mu, int strategy) {
    for (int64_t i = 0; i < cs->ncones; i++) {
        orc_cone *c = &cs->c[i];
        const double *si = s + c->cone_start, *zi = z + c->cone_start;
        if (c->tag == CONE_NONNEG) { /* nonnegativecone.rs:77-90 */
            for (int64_t k = 0; k < c->numel; k++) {
                c->lam[k] = sqrt(si[k] * zi[k]);
                c->w[k] = sqrt(si[k] / zi[k]);
            }
        } else if (c->tag == CONE_SOC) {
            if (!soc_update_scaling(c, si, zi)) return 0;
        } else if (c->tag == CONE_EXP || c->tag == CONE_POW) { /* expcone.rs:106-124, powcone.rs:99-117 */
            if (c->tag == CONE_EXP) exp_update_dual_grad_H(c, zi);
            else pow_update_dual_grad_H(c, zi);
            ns3_update_Hs(c, si, zi, mu, strategy);
            c->zc[0] = zi[0]; c->zc[1] = zi[1]; c->zc[2] = zi[2];
            c->ns_valid = 1;
        } else if (c->tag == CONE_GENPOW) { /* genpowcone.rs:141-157 */
            genpow_update_dual_grad_H(c, zi);
            c->gmu = mu;
            memcpy(c->gz, zi, (size_t)c->numel * sizeof(double));
            c->gvalid = 1;
        }
    }
    return 1;
}

/* compositecone.rs:253-257 get_Hs + per-cone get_Hs */
void orc_cones_get_Hs(const orc_cones *cs, double *Hs) {
    for (int64_t i = 0; i < cs->ncones; i++) {
        const orc_cone *c = &cs->c[i];
        double *blk = Hs + c->block_start;
        if (c->tag == CONE_ZERO) { /* zerocone.rs:94-96 */
            for (int64_t k = 0; k < c->block_len; k++) blk[k] = 0.0;
        } else if (c->tag == CONE_NONNEG) { /* nonnegativecone.rs:96-101 */
            for (int64_t k = 0; k < c->numel; k++) blk[k] = c->w[k] * c->w[k];
        } else if (c->tag == CONE_SOC) { /* socone.rs:217-246 */
            double eta2 = c->eta * c->eta;
            if (c->sparse) {
                for (int64_t k = 0; k < c->numel; k++) blk[k] = eta2;
                blk[0] *= c->d;
            } else {
                const double *w = c->w;
                blk[0] = (M_SQRT2 * w[0] - 1.0) * (M_SQRT2 * w[0] + 1.0);
                int64_t h = 1;
                for (int64_t col = 1; col < c->numel; col++) {
                    double wcol = w[col];
                    for (int64_t row = 0; row <= col; row++) blk[h++] = 2.0 * w[row] * wcol;
                    blk[h - 1] += 1.0;
                }
                for (int64_t k = 0; k < c->block_len; k++) blk[k] *= eta2;
            }
        }
        else if ((c->tag == CONE_EXP || c->tag == CONE_POW) && c->ns_valid) { /* expcone.rs:130-133 */
            for (int64_t k = 0; k < 6; k++) blk[k] = c->Hs3[k];
        } else if (c->tag == CONE_GENPOW && c->gvalid) { /* genpowcone.rs:163-171 */
            for (int64_t k = 0; k < c->dim; k++) blk[k] = c->gmu * c->gd1[k];
            for (int64_t k = 0; k < c->dim2; k++) blk[c->dim + k] = c->gmu * c->gd2;
        }
        /* other cone types: left untouched (caller-provided) */
    }
}

/* mul_Hs: nonnegativecone.rs:103-108, socone.rs:248-256, zerocone.rs:98-100 */
void orc_cones_mul_Hs(const orc_cones *cs, double *y, const double *x) {
    for (int64_t i = 0; i < cs->ncones; i++) {
        const orc_cone *c = &cs->c[i];
        double *yi = y + c->cone_start;
        const double *xi = x + c->cone_start;
        if (c->tag == CONE_ZERO) {
            for (int64_t k = 0; k < c->numel; k++) yi[k] = 0.0;
        } else if (c->tag == CONE_NONNEG) {
            for (int64_t k = 0; k < c->numel; k++) yi[k] = c->w[k] * (c->w[k] * xi[k]);
        } else if (c->tag == CONE_SOC) {
            double cc = dotp(c->w, xi, c->numel) * 2.0;
            for (int64_t k = 0; k < c->numel; k++) yi[k] = xi[k];
            yi[0] = -xi[0];
            for (int64_t k = 0; k < c->numel; k++) yi[k] = cc * c->w[k] + 1.0 * yi[k];
            double e2 = c->eta * c->eta;
            for (int64_t k = 0; k < c->numel; k++) yi[k] *= e2;
        } else if (c->tag == CONE_EXP || c->tag == CONE_POW) { /* expcone.rs:135-137 */
            sym3_mul(c->Hs3, yi, xi);
        } else if (c->tag == CONE_GENPOW) { /* genpowcone.rs:173-193 */
            const int64_t d1n = c->dim, d2n = c->dim2;
            const double cp = dotp(c->gp, xi, c->numel), cq = dotp(c->gq, xi, d1n), cr = dotp(c->gr, xi + d1n, d2n);
            for (int64_t k = 0; k < d1n; k++) yi[k] = c->gd1[k] * xi[k] - cq * c->gq[k];
            for (int64_t k = 0; k < d2n; k++) yi[d1n + k] = c->gd2 * xi[d1n + k] - cr * c->gr[k];
            for (int64_t k = 0; k < c->numel; k++) yi[k] = cp * c->gp[k] + 1.0 * yi[k];
            for (int64_t k = 0; k < c->numel; k++) yi[k] *= c->gmu;
        }
    }
}
const double *orc_cone_Hs3(const orc_cones *cs, int64_t i) { return cs->c[i].Hs3; }
const double *orc_cone_Hdual(const orc_cones *cs, int64_t i) { return cs->c[i].Hdual; }
const double *orc_cone_grad3(const orc_cones *cs, int64_t i) { return cs->c[i].grad3; }
/* ---- step / right-hand-side operations of the symmetric cones -------- */
/* socone.rs:504-530 (fast W / W^-1 products of the ECOS paper) */
static void soc_mul_W(double *y, const double *x, double a, double b, const double *w, double eta, int64_t n) {
    double zeta = dotp(w + 1, x + 1, n - 1);
    double c = x[0] + zeta / (1.0 + w[0]);
    y[0] = (a * eta) * (w[0] * x[0] + zeta) + b * y[0];
    for (int64_t i = 1; i < n; i++) y[i] = (a * eta * c) * w[i] + b * y[i];
    for (int64_t i = 1; i < n; i++) y[i] = (a * eta) * x[i] + 1.0 * y[i];
}
static void soc_mul_Winv(double *y, const double *x, double a, double b, const double *w, double eta, int64_t n) {
    double zeta = dotp(w + 1, x + 1, n - 1);
    double c = -x[0] + zeta / (1.0 + w[0]);
    y[0] = (a / eta) * (w[0] * x[0] - zeta) + b * y[0];
    for (int64_t i = 1; i < n; i++) y[i] = (a / eta * c) * w[i] + b * y[i];
    for (int64_t i = 1; i < n; i++) y[i] = (a / eta) * x[i] + 1.0 * y[i];
}
/* socone.rs:360-367 */
static void soc_circ_op(double *x, const double *y, const double *z, int64_t n) {
    x[0] = dotp(y, z, n);
    double y0 = y[0], z0 = z[0];
    for (int64_t i = 1; i < n; i++) x[i] = y0 * z[i] + z0 * y[i];
}
/* ---- Exponential / Power cones: feasibility, barriers, 3rd-order correction ------------- */
/* dense3x3/cholesky.rs:13-57 on the packed triu [00,01,11,02,12,22]; L shares the packing */
static int chol3_factor(double *L, const double *A) {
    double t = A[0];
    if (t <= 0.0) return 0;
    L[0] = sqrt(t);
    L[1] = A[1] / L[0];
    t = A[2] - L[1] * L[1];
    if (t <= 0.0) return 0;
    L[2] = sqrt(t);
    L[3] = A[3] / L[0];
    L[4] = (A[4] - L[1] * L[3]) / L[2];
    t = A[5] - L[3] * L[3] - L[4] * L[4];
    if (t <= 0.0) return 0;
    L[5] = sqrt(t);
    return 1;
}
static void chol3_solve(const double *L, double *x, const double *b) {
    double c0 = b[0] / L[0];
    double c1 = (b[1] - L[1] * c0) / L[2];
    double c2 = (b[2] - L[3] * c0 - L[4] * c1) / L[5];
    x[2] = c2 / L[5];
    x[1] = (c1 - L[4] * x[2]) / L[2];
    x[0] = (c0 - L[1] * x[1] - L[3] * x[2]) / L[0];
}
/* expcone.rs:189-220 */
static int exp_is_primal_feasible(const double *s) {
    if (s[2] > 0.0 && s[1] > 0.0) {
        double res = s[1] * logsafe(s[2] / s[1]) - s[0];
        if (res > 0.0) return 1;
    }
    return 0;
}
static int exp_is_dual_feasible(const double *z) {
    if (z[2] > 0.0 && z[0] < 0.0) {
        double res = z[1] - z[0] - z[0] * logsafe(-z[2] / z[0]);
        if (res > 0.0) return 1;
    }
    return 0;
}
/* expcone.rs:222-252 */
static double exp_barrier_primal(const double *s) {
    double om = orc_wright_omega(1.0 - s[0] / s[1] - logsafe(s[1] / s[2]));
    om = (om - 1.0) * (om - 1.0) / om;
    return -logsafe(om) - logsafe(s[1]) * 2.0 - logsafe(s[2]) - 3.0;
}
static double exp_barrier_dual(const double *z) {
    double l = logsafe(-z[2] / z[0]);
    return -logsafe(-z[2] * z[0]) - logsafe(z[1] - z[0] - z[0] * l);
}
/* expcone.rs:254-308 */
static void exp_higher_correction(const orc_cone *c, double *eta, const double *ds, const double *v) {
    double L[6], u[3];
    const double *z = c->zc;
    if (!chol3_factor(L, c->Hdual)) { eta[0] = eta[1] = eta[2] = 0.0; return; }
    chol3_solve(L, u, ds);
    eta[1] = 1.0;
    eta[2] = -z[0] / z[2];
    eta[0] = logsafe(eta[2]);
    double psi = z[0] * eta[0] - z[0] + z[1];
    double dpu = dotp(u, eta, 3), dpv = dotp(v, eta, 3);
    double coef = ((u[0] * (v[0] / z[0] - v[2] / z[2]) + u[2] * (z[0] * v[2] / z[2] - v[0]) / z[2]) * psi -
                   2.0 * dpu * dpv) / (psi * psi * psi);
    for (int i = 0; i < 3; i++) eta[i] *= coef;
    double ip2 = 1.0 / (psi * psi);
    eta[0] += (1.0 / psi - 2.0 / z[0]) * u[0] * v[0] / (z[0] * z[0]) - u[2] * v[2] / (z[2] * z[2]) / psi +
              dpu * ip2 * (v[0] / z[0] - v[2] / z[2]) + dpv * ip2 * (u[0] / z[0] - u[2] / z[2]);
    eta[2] += 2.0 * (z[0] / psi - 1.0) * u[2] * v[2] / (z[2] * z[2] * z[2]) -
              (u[2] * v[0] + u[0] * v[2]) / (z[2] * z[2]) / psi +
              dpu * ip2 * (z[0] * v[2] / (z[2] * z[2]) - v[0] / z[2]) +
              dpv * ip2 * (z[0] * u[2] / (z[2] * z[2]) - u[0] / z[2]);
    for (int i = 0; i < 3; i++) eta[i] *= 0.5;
}
/* powcone.rs:188-221 */
static int pow_is_primal_feasible(const orc_cone *c, const double *s) {
    double a = c->alpha;
    if (s[0] > 0.0 && s[1] > 0.0) {
        double res = exp(2.0 * a * logsafe(s[0]) + 2.0 * (1.0 - a) * logsafe(s[1])) - s[2] * s[2];
        if (res > 0.0) return 1;
    }
    return 0;
}
static int pow_is_dual_feasible(const orc_cone *c, const double *z) {
    double a = c->alpha;
    if (z[0] > 0.0 && z[1] > 0.0) {
        double res = exp((a * 2.0) * logsafe(z[0] / a) + (1.0 - a) * logsafe(z[1] / (1.0 - a)) * 2.0) - z[2] * z[2];
        if (res > 0.0) return 1;
    }
    return 0;
}
/* powcone.rs:223-258 */
static double pow_barrier_primal(const orc_cone *c, const double *s) {
    double a = c->alpha, g[3];
    pow_gradient_primal(c, s, g);
    double out = 0.0;
    out += logsafe(pow(-g[0] / a, 2.0 * a) * pow(-g[1] / (1.0 - a), 2.0 - a * 2.0) - g[2] * g[2]);
    out += (1.0 - a) * logsafe(-g[0]);
    out += a * logsafe(-g[1]) - 3.0;
    return out;
}
static double pow_barrier_dual(const orc_cone *c, const double *z) {
    double a = c->alpha;
    double arg1 = pow(z[0] / a, 2.0 * a) * pow(z[1] / (1.0 - a), 2.0 - 2.0 * a) - z[2] * z[2];
    return -logsafe(arg1) - (1.0 - a) * logsafe(z[0]) - a * logsafe(z[1]);
}
/* powcone.rs:260-337 */
static void pow_higher_correction(const orc_cone *c, double *eta, const double *ds, const double *v) {
    double L[6], u[3], Hp[6], Hv[3], Hu[3];
    const double *z = c->zc;
    if (!chol3_factor(L, c->Hdual)) { eta[0] = eta[1] = eta[2] = 0.0; return; }
    chol3_solve(L, u, ds);
    double a = c->alpha;
    double phi = pow(z[0] / a, 2.0 * a) * pow(z[1] / (1.0 - a), 2.0 - 2.0 * a);
    double psi = phi - z[2] * z[2];
    eta[0] = 2.0 * a * phi / z[0];
    eta[1] = 2.0 * (1.0 - a) * phi / z[1];
    eta[2] = -2.0 * z[2];
    Hp[1] = 4.0 * a * (1.0 - a) * phi / (z[0] * z[1]);
    Hp[0] = 2.0 * a * (2.0 * a - 1.0) * phi / (z[0] * z[0]);
    Hp[3] = 0.0;
    Hp[2] = 2.0 * (1.0 - a) * (1.0 - 2.0 * a) * phi / (z[1] * z[1]);
    Hp[4] = 0.0;
    Hp[5] = -2.0;
    double dpu = dotp(u, eta, 3), dpv = dotp(v, eta, 3);
    sym3_mul(Hp, Hv, v);
    double coef = (dotp(u, Hv, 3) * psi - 2.0 * dpu * dpv) / (psi * psi * psi);
    double coef2 = 4.0 * a * (2.0 * a - 1.0) * (1.0 - a) * phi * (u[0] / z[0] - u[1] / z[1]) *
                   (v[0] / z[0] - v[1] / z[1]) / psi;
    double ip2 = 1.0 / (psi * psi);
    eta[0] = coef * eta[0] - 2.0 * (1.0 - a) * u[0] * v[0] / (z[0] * z[0] * z[0]) + coef2 / z[0] + Hv[0] * dpu * ip2;
    eta[1] = coef * eta[1] - 2.0 * a * u[1] * v[1] / (z[1] * z[1] * z[1]) - coef2 / z[1] + Hv[1] * dpu * ip2;
    eta[2] = coef * eta[2] + Hv[2] * dpu * ip2;
    sym3_mul(Hp, Hu, u);
    for (int i = 0; i < 3; i++) eta[i] = (dpv * ip2) * Hu[i] + 1.0 * eta[i];
    for (int i = 0; i < 3; i++) eta[i] *= 0.5;
}
/* nonsymmetric_common.rs:164-192 */
static double ns3_backtrack(const orc_cone *c, const double *dq, const double *q, double a_init, double a_min,
                            double step, int dual) {
    double alpha = a_init, w[3];
    for (;;) {
        for (int i = 0; i < 3; i++) w[i] = 1.0 * q[i] + alpha * dq[i];
        int ok = c->tag == CONE_EXP ? (dual ? exp_is_dual_feasible(w) : exp_is_primal_feasible(w))
                                    : (dual ? pow_is_dual_feasible(c, w) : pow_is_primal_feasible(c, w));
        if (ok) break;
        alpha *= step;
        if (alpha < a_min) { alpha = 0.0; break; }
    }
    return alpha;
}

/* compositecone.rs:266-272 affine_ds: nonnegativecone.rs:110-115, socone.rs:258-260, zerocone.rs:102-104,
 * expcone.rs:129-131 / powcone.rs:128-130 (ds = s) */
void orc_cones_affine_ds(const orc_cones *cs, double *ds, const double *s) {
    for (int64_t i = 0; i < cs->ncones; i++) {
        const orc_cone *c = &cs->c[i];
        double *d = ds + c->cone_start;
        if (c->tag == CONE_ZERO) {
            for (int64_t k = 0; k < c->numel; k++) d[k] = 0.0;
        } else if (c->tag == CONE_NONNEG) {
            for (int64_t k = 0; k < c->numel; k++) d[k] = c->lam[k] * c->lam[k];
        } else if (c->tag == CONE_SOC) {
            soc_circ_op(d, c->lam, c->lam, c->numel);
        } else if (c->tag == CONE_EXP || c->tag == CONE_POW || c->tag == CONE_GENPOW) { /* genpowcone.rs:195-197 */
            for (int64_t k = 0; k < c->numel; k++) d[k] = s[c->cone_start + k];
        }
    }
}
/* compositecone.rs:274-289 + symmetric_common.rs:53-84 (NN, SOC), zerocone.rs:106-110 */
void orc_cones_combined_ds_shift(const orc_cones *cs, double *shift, double *step_z, double *step_s,
                                 double sigma_mu) {
    for (int64_t i = 0; i < cs->ncones; i++) {
        const orc_cone *c = &cs->c[i];
        double *sh = shift + c->cone_start, *dz = step_z + c->cone_start, *dsv = step_s + c->cone_start;
        int64_t n = c->numel;
        if (c->tag == CONE_ZERO) {
            for (int64_t k = 0; k < n; k++) sh[k] = 0.0;
        } else if (c->tag == CONE_NONNEG) {
            for (int64_t k = 0; k < n; k++) sh[k] = dz[k];
            for (int64_t k = 0; k < n; k++) dz[k] = 1.0 * (sh[k] * c->w[k]) + 0.0 * dz[k]; /* mul_W :188-194 */
            for (int64_t k = 0; k < n; k++) sh[k] = dsv[k];
            for (int64_t k = 0; k < n; k++) dsv[k] = 1.0 * (sh[k] / c->w[k]) + 0.0 * dsv[k]; /* mul_Winv */
            for (int64_t k = 0; k < n; k++) sh[k] = dsv[k] * dz[k];                          /* circ_op */
            for (int64_t k = 0; k < n; k++) sh[k] = sh[k] + (-sigma_mu);                      /* translate */
        } else if (c->tag == CONE_SOC) {
            memcpy(sh, dz, (size_t)n * sizeof(double));
            soc_mul_W(dz, sh, 1.0, 0.0, c->w, c->eta, n);
            memcpy(sh, dsv, (size_t)n * sizeof(double));
            soc_mul_Winv(dsv, sh, 1.0, 0.0, c->w, c->eta, n);
            soc_circ_op(sh, dsv, dz, n);
            sh[0] += -sigma_mu; /* scaled_unit_shift, socone.rs:110-112 */
        } else if (c->tag == CONE_EXP || c->tag == CONE_POW) { /* expcone.rs:133-142, powcone.rs:132-141 */
            double eta[3];
            if (c->tag == CONE_EXP) exp_higher_correction(c, eta, dsv, dz);
            else pow_higher_correction(c, eta, dsv, dz);
            for (int k = 0; k < 3; k++) sh[k] = c->grad3[k] * sigma_mu - eta[k];
        } else if (c->tag == CONE_GENPOW) { /* genpowcone.rs:199-204: no higher-order correction */
            for (int64_t k = 0; k < n; k++) sh[k] = c->ggrad[k] * sigma_mu;
        }
    }
}
/* compositecone.rs:291-299: nonnegativecone.rs:122-126, socone.rs:266-287, zerocone.rs:112-114 */
void orc_cones_ds_from_dz_offset(const orc_cones *cs, double *out, const double *ds, const double *z) {
    for (int64_t i = 0; i < cs->ncones; i++) {
        const orc_cone *c = &cs->c[i];
        double *o = out + c->cone_start;
        const double *d = ds + c->cone_start, *zi = z + c->cone_start;
        int64_t n = c->numel;
        if (c->tag == CONE_ZERO) {
            for (int64_t k = 0; k < n; k++) o[k] = 0.0;
        } else if (c->tag == CONE_NONNEG) {
            for (int64_t k = 0; k < n; k++) o[k] = d[k] / zi[k];
        } else if (c->tag == CONE_SOC) {
            double resz = soc_residual(zi, n);
            double l1d1 = dotp(c->lam + 1, d + 1, n - 1);
            double w1d1 = dotp(c->w + 1, d + 1, n - 1);
            for (int64_t k = 0; k < n; k++) o[k] = -zi[k];
            o[0] = zi[0];
            double cc = c->lam[0] * d[0] - l1d1;
            double sc = cc / resz;
            for (int64_t k = 0; k < n; k++) o[k] *= sc;
            o[0] += c->eta * w1d1;
            for (int64_t k = 1; k < n; k++) o[k] += c->eta * (d[k] + w1d1 / (1.0 + c->w[0]) * c->w[k]);
            double rl = 1.0 / c->lam[0];
            for (int64_t k = 0; k < n; k++) o[k] *= rl;
        } else if (c->tag == CONE_EXP || c->tag == CONE_POW || c->tag == CONE_GENPOW) { /* expcone.rs:144-146 */
            for (int64_t k = 0; k < n; k++) o[k] = d[k];
        }
    }
}
/* socone.rs:421-495 */
static double soc_step_component(const double *x, const double *y, double amax, int64_t n) {
    if (x[0] >= 0.0 && y[0] < 0.0) {
        double t = -x[0] / y[0];
        amax = amax < t ? amax : t;
    }
    double a = soc_residual(y, n);
    double b = 2.0 * (x[0] * y[0] - dotp(x + 1, y + 1, n - 1));
    double cres = soc_residual(x, n);
    double c = cres > 0.0 ? cres : 0.0;
    double d = b * b - 4.0 * a * c;
    if ((a > 0.0 && b > 0.0) || d < 0.0) return amax;
    if (a == 0.0) return amax;
    if (c == 0.0) return a >= 0.0 ? amax : 0.0;
    double t = (b >= 0.0) ? (-b - sqrt(d)) : (-b + sqrt(d));
    double r1 = (2.0 * c) / t, r2 = t / (2.0 * a);
    if (r1 < 0.0) r1 = INFINITY;
    if (r2 < 0.0) r2 = INFINITY;
    double r = r1 < r2 ? r1 : r2;
    return amax < r ? amax : r;
}
/* compositecone.rs:300-340: symmetric cones first (nonnegativecone.rs:128-153, socone.rs:289-302,
 * zerocone.rs:116-127), then -- backed off to 1 - sqrt(eps) -- the nonsymmetric ones
 * (expcone.rs:148-168, powcone.rs:147-167 via backtrack_search).  Returns alpha (= alpha_z = alpha_s). */
double orc_cones_step_length_ex(const orc_cones *cs, const double *dz, const double *ds, const double *z,
                                const double *s, double amax, double backtrack_step, double alpha_min) {
    double alpha = amax;
    int all_symmetric = 1;
    for (int64_t i = 0; i < cs->ncones; i++) {
        const orc_cone *c = &cs->c[i];
        const double *dzi = dz + c->cone_start, *dsi = ds + c->cone_start;
        const double *zi = z + c->cone_start, *si = s + c->cone_start;
        double az = alpha, as = alpha;
        if (c->tag == CONE_EXP || c->tag == CONE_POW || c->tag == CONE_GENPOW) { all_symmetric = 0; continue; }
        if (c->tag == CONE_NONNEG) {
            for (int64_t k = 0; k < c->numel; k++) {
                if (dzi[k] < 0.0) { double t = -zi[k] / dzi[k]; az = az < t ? az : t; }
                if (dsi[k] < 0.0) { double t = -si[k] / dsi[k]; as = as < t ? as : t; }
            }
        } else if (c->tag == CONE_SOC) {
            az = soc_step_component(zi, dzi, alpha, c->numel);
            as = soc_step_component(si, dsi, alpha, c->numel);
        }
        double mn = az < as ? az : as;
        alpha = alpha < mn ? alpha : mn;
    }
    if (!all_symmetric) {
        double ceilv = 1.0 - sqrt(2.220446049250313e-16);
        alpha = alpha < ceilv ? alpha : ceilv;
        for (int64_t i = 0; i < cs->ncones; i++) {
            const orc_cone *c = &cs->c[i];
            if (c->tag != CONE_EXP && c->tag != CONE_POW && c->tag != CONE_GENPOW) continue;
            double az, as;
            if (c->tag == CONE_GENPOW) { /* genpowcone.rs:210-233 */
                az = genpow_backtrack(c, dz + c->cone_start, z + c->cone_start, alpha, alpha_min, backtrack_step, 1);
                as = genpow_backtrack(c, ds + c->cone_start, s + c->cone_start, alpha, alpha_min, backtrack_step, 0);
            } else {
                az = ns3_backtrack(c, dz + c->cone_start, z + c->cone_start, alpha, alpha_min, backtrack_step, 1);
                as = ns3_backtrack(c, ds + c->cone_start, s + c->cone_start, alpha, alpha_min, backtrack_step, 0);
            }
            double mn = az < as ? az : as;
            alpha = alpha < mn ? alpha : mn;
        }
    }
    return alpha;
}
double orc_cones_step_length(const orc_cones *cs, const double *dz, const double *ds, const double *z,
                             const double *s, double amax) {
    /* settings.rs defaults: linesearch_backtrack_step 0.8, min_terminate_step_length 1e-4 */
    return orc_cones_step_length_ex(cs, dz, ds, z, s, amax, 0.8, 1e-4);
}
/* vecmath.rs norm_shifted: stable norm of (z + alpha dz) */
static double norm2_shifted(const double *z, const double *dz, double alpha, int64_t n) {
    double scale = 0.0, sumsq = 1.0;
    for (int64_t i = 0; i < n; i++) {
        double xi = z[i] + alpha * dz[i];
        if (xi == 0.0) continue;
        double a = fabs(xi);
        if (scale < a) {
            double r = scale / a;
            sumsq = 1.0 + sumsq * r * r;
            scale = a;
        } else {
            double r = a / scale;
            sumsq = sumsq + r * r;
        }
    }
    return scale * sqrt(sumsq);
}
/* compositecone.rs:342-352 compute_barrier: nonnegativecone.rs:155-166, socone.rs:304-314,
 * zerocone.rs:129-131, expcone.rs:170-181, powcone.rs:169-180 */
double orc_cones_compute_barrier(const orc_cones *cs, const double *z, const double *s, const double *dz,
                                 const double *ds, double alpha) {
    double barrier = 0.0;
    for (int64_t i = 0; i < cs->ncones; i++) {
        const orc_cone *c = &cs->c[i];
        const double *zi = z + c->cone_start, *si = s + c->cone_start;
        const double *dzi = dz + c->cone_start, *dsi = ds + c->cone_start;
        if (c->tag == CONE_NONNEG) {
            double b = 0.0;
            for (int64_t k = 0; k < c->numel; k++) b -= logsafe((si[k] + alpha * dsi[k]) * (zi[k] + alpha * dzi[k]));
            barrier += b;
        } else if (c->tag == CONE_SOC) {
            double x0 = si[0] + alpha * dsi[0], x1 = norm2_shifted(si + 1, dsi + 1, alpha, c->numel - 1);
            double res_s = (x0 - x1) * (x0 + x1);
            x0 = zi[0] + alpha * dzi[0];
            x1 = norm2_shifted(zi + 1, dzi + 1, alpha, c->numel - 1);
            double res_z = (x0 - x1) * (x0 + x1);
            barrier += (res_s > 0.0 && res_z > 0.0) ? -logsafe(res_s * res_z) * 0.5 : INFINITY;
        } else if (c->tag == CONE_EXP || c->tag == CONE_POW) {
            double cz[3], csv[3];
            for (int k = 0; k < 3; k++) {
                cz[k] = zi[k] + alpha * dzi[k];
                csv[k] = si[k] + alpha * dsi[k];
            }
            if (c->tag == CONE_EXP) barrier += exp_barrier_dual(cz) + exp_barrier_primal(csv);
            else barrier += pow_barrier_dual(c, cz) + pow_barrier_primal(c, csv);
        } else if (c->tag == CONE_GENPOW) { /* genpowcone.rs:235-250 */
            double *w = (double *)malloc((size_t)(c->numel + 1) * sizeof(double));
            for (int64_t k = 0; k < c->numel; k++) w[k] = 1.0 * si[k] + alpha * dsi[k];
            barrier += genpow_barrier_primal(c, w);
            for (int64_t k = 0; k < c->numel; k++) w[k] = 1.0 * zi[k] + alpha * dzi[k];
            barrier += genpow_barrier_dual(c, w);
            free(w);
        }
    }
    return barrier;
}
/* compositecone.rs:208-214 unit_initialization: zerocone.rs:71-74, nonnegativecone.rs:68-71,
 * socone.rs:114-119, expcone.rs:87-93, powcone.rs:79-87 */
void orc_cones_unit_initialization(const orc_cones *cs, double *z, double *s) {
    for (int64_t i = 0; i < cs->ncones; i++) {
        const orc_cone *c = &cs->c[i];
        double *zi = z + c->cone_start, *si = s + c->cone_start;
        for (int64_t k = 0; k < c->numel; k++) zi[k] = si[k] = 0.0;
        if (c->tag == CONE_NONNEG) {
            for (int64_t k = 0; k < c->numel; k++) zi[k] = si[k] = 1.0;
        } else if (c->tag == CONE_SOC) {
            zi[0] = si[0] = 1.0;
        } else if (c->tag == CONE_EXP) {
            si[0] = -1.051383945322714; si[1] = 0.556409619469370; si[2] = 1.258967884768947;
            for (int k = 0; k < 3; k++) zi[k] = si[k];
        } else if (c->tag == CONE_POW) {
            si[0] = sqrt(1.0 + c->alpha); si[1] = sqrt(1.0 + (1.0 - c->alpha)); si[2] = 0.0;
            for (int k = 0; k < 3; k++) zi[k] = si[k];
        } else if (c->tag == CONE_GENPOW) { /* genpowcone.rs:127-135 */
            for (int64_t k = 0; k < c->dim; k++) zi[k] = si[k] = sqrt(1.0 + c->ga[k]);
        }
    }
}
int orc_cones_is_symmetric(const orc_cones *cs) {
    for (int64_t i = 0; i < cs->ncones; i++)
        if (cs->c[i].tag == CONE_EXP || cs->c[i].tag == CONE_POW || cs->c[i].tag == CONE_GENPOW) return 0;
    return 1;
}
/* CompositeCone::margins (compositecone.rs:130-152): (min over cones of alpha, sum of beta);
 * nonnegativecone.rs:58-62, socone.rs:104-108, zerocone.rs margins = (inf, 0) */
void orc_cones_margins(const orc_cones *cs, const double *z, double *alpha_out, double *beta_out) {
    double alpha = 1.7976931348623157e308 /* T::max_value() */, beta = 0.0;
    for (int64_t i = 0; i < cs->ncones; i++) {
        const orc_cone *c = &cs->c[i];
        const double *zi = z + c->cone_start;
        if (c->tag == CONE_NONNEG) {
            double a = INFINITY, b = 0.0;
            for (int64_t k = 0; k < c->numel; k++) {
                a = a < zi[k] ? a : zi[k];
                b += zi[k] > 0.0 ? zi[k] : 0.0;
            }
            alpha = alpha < a ? alpha : a;
            beta += b;
        } else if (c->tag == CONE_SOC) {
            double a = zi[0] - orc_norm2(zi + 1, c->numel - 1);
            alpha = alpha < a ? alpha : a;
            beta += a > 0.0 ? a : 0.0;
        }
    }
    *alpha_out = alpha;
    *beta_out = beta;
}

/* state accessors for tests */
double orc_cone_eta(const orc_cones *cs, int64_t i) { return cs->c[i].eta; }
double orc_cone_d(const orc_cones *cs, int64_t i) { return cs->c[i].d; }
const double *orc_cone_w(const orc_cones *cs, int64_t i) { return cs->c[i].w; }
const double *orc_cone_lambda(const orc_cones *cs, int64_t i) { return cs->c[i].lam; }
const double *orc_cone_u(const orc_cones *cs, int64_t i) { return cs->c[i].u; }
const double *orc_cone_v(const orc_cones *cs, int64_t i) { return cs->c[i].v; }

/* ------------------------------------------------------------------------ */
/* CSC block helpers: algebra/csc/utils.rs (K.colptr used as counters)        */
/* ------------------------------------------------------------------------ */
typedef struct {
    int64_t m, n;
    int64_t *colptr, *rowval;
    double *nzval;
    int64_t nnz;
} orc_csc;

static void colcount_dense_triangle(orc_csc *K, int64_t initcol, int64_t blockcols, int shape) {
    for (int64_t k = 0; k < blockcols; k++) /* utils.rs:16-33 */
        K->colptr[initcol + k] += (shape == SHAPE_TRIU) ? (k + 1) : (blockcols - k);
}
static void colcount_diag(orc_csc *K, int64_t initcol, int64_t blockcols) { /* :37-40 */
    for (int64_t k = 0; k < blockcols; k++) K->colptr[initcol + k] += 1;
}
static int missing_diag(const orc_csc *M, int64_t i) { /* :50-52 */
    return M->colptr[i] == M->colptr[i + 1] || M->rowval[M->colptr[i + 1] - 1] != i;
}
static void colcount_missing_diag(orc_csc *K, const orc_csc *M, int64_t initcol) { /* :45-57 */
    for (int64_t i = 0; i < M->n; i++)
        if (missing_diag(M, i)) K->colptr[i + initcol] += 1;
}
static void colcount_colvec(orc_csc *K, int64_t n, int64_t firstrow, int64_t firstcol) {
    (void)firstrow; /* :62-65 */
    K->colptr[firstcol] += n;
}
static void colcount_rowvec(orc_csc *K, int64_t n, int64_t firstrow, int64_t firstcol) {
    (void)firstrow; /* :70-76 */
    for (int64_t k = 0; k < n; k++) K->colptr[firstcol + k] += 1;
}
static void colcount_block(orc_csc *K, const orc_csc *M, int64_t initcol, int transpose) {
    if (transpose) { /* :80-94 */
        for (int64_t p = 0; p < M->nnz; p++) K->colptr[initcol + M->rowval[p]] += 1;
    } else {
        for (int64_t i = 0; i < M->n; i++) K->colptr[initcol + i] += M->colptr[i + 1] - M->colptr[i];
    }
}
static void fill_colvec(orc_csc *K, int64_t *vtoKKT, int64_t len, int64_t initrow, int64_t initcol) {
    for (int64_t i = 0; i < len; i++) { /* :98-106 */
        int64_t dest = K->colptr[initcol];
        K->rowval[dest] = initrow + i;
        K->nzval[dest] = 0.0;
        vtoKKT[i] = dest;
        K->colptr[initcol] += 1;
    }
}
static void fill_rowvec(orc_csc *K, int64_t *vtoKKT, int64_t len, int64_t initrow, int64_t initcol) {
    for (int64_t i = 0; i < len; i++) { /* :110-119 */
        int64_t col = initcol + i;
        int64_t dest = K->colptr[col];
        K->rowval[dest] = initrow;
        K->nzval[dest] = 0.0;
        vtoKKT[i] = dest;
        K->colptr[col] += 1;
    }
}
static void fill_block(orc_csc *K, const orc_csc *M, int64_t *MtoKKT, int64_t initrow,
                       int64_t initcol, int transpose) {
    for (int64_t i = 0; i < M->n; i++) { /* :124-157 */
        for (int64_t j = M->colptr[i]; j < M->colptr[i + 1]; j++) {
            int64_t col, row;
            if (transpose) {
                col = M->rowval[j] + initcol;
                row = i + initrow;
            } else {
                col = i + initcol;
                row = M->rowval[j] + initrow;
            }
            int64_t dest = K->colptr[col];
            K->rowval[dest] = row;
            K->nzval[dest] = M->nzval[j];
            MtoKKT[j] = dest;
            K->colptr[col] += 1;
        }
    }
}
static void fill_dense_triangle(orc_csc *K, int64_t *blocktoKKT, int64_t offset, int64_t blockdim,
                                int shape) {
    int64_t kidx = 0; /* :161-219 */
    if (shape == SHAPE_TRIU) {
        for (int64_t col = offset; col < offset + blockdim; col++)
            for (int64_t row = offset; row <= col; row++) {
                int64_t dest = K->colptr[col];
                K->rowval[dest] = row;
                K->nzval[dest] = 0.0;
                K->colptr[col] += 1;
                blocktoKKT[kidx++] = dest;
            }
    } else {
        for (int64_t row = offset; row < offset + blockdim; row++)
            for (int64_t col = offset; col <= row; col++) {
                int64_t dest = K->colptr[col];
                K->rowval[dest] = row;
                K->nzval[dest] = 0.0;
                K->colptr[col] += 1;
                blocktoKKT[kidx++] = dest;
            }
    }
}
static void fill_diag(orc_csc *K, int64_t *diagtoKKT, int64_t offset, int64_t blockdim) {
    for (int64_t i = 0; i < blockdim; i++) { /* :223-231 */
        int64_t col = offset + i;
        int64_t dest = K->colptr[col];
        K->rowval[dest] = col;
        K->nzval[dest] = 0.0;
        K->colptr[col] += 1;
        diagtoKKT[i] = dest;
    }
}
static void fill_missing_diag(orc_csc *K, const orc_csc *M, int64_t initcol) {
    for (int64_t i = 0; i < M->n; i++) /* :236-249 */
        if (missing_diag(M, i)) {
            int64_t dest = K->colptr[i + initcol];
            K->rowval[dest] = i + initcol;
            K->nzval[dest] = 0.0;
            K->colptr[i + initcol] += 1;
        }
}
static void colcount_to_colptr(orc_csc *K) { /* :253-260 */
    int64_t cur = 0;
    for (int64_t i = 0; i <= K->n; i++) {
        int64_t c = K->colptr[i];
        K->colptr[i] = cur;
        cur += c;
    }
}
static void backshift_colptrs(orc_csc *K) { /* :271-274 */
    for (int64_t i = K->n; i > 0; i--) K->colptr[i] = K->colptr[i - 1];
    K->colptr[0] = 0;
}
static int64_t count_diagonal_entries_triu(const orc_csc *M) { /* :276-291 */
    int64_t c = 0;
    for (int64_t i = 0; i < M->n; i++)
        if (M->colptr[i + 1] != M->colptr[i] && M->rowval[M->colptr[i + 1] - 1] == i) c++;
    return c;
}

/* ------------------------------------------------------------------------ */
/* LDLDataMap + assemble_kkt_matrix                                          */
/* ------------------------------------------------------------------------ */
typedef struct {
    int64_t n, m, p, nnzP, nnzA;
    int64_t *P, *A, *Hsblocks, *diagP, *diag_full;
    int64_t nHs;
    /* sparse maps, in cone order: for SOC  u[numel], v[numel], D[2];
     * for GenPow p[numel], q[dim1], r[dim2], D[3]  (datamaps.rs:112-116,227-232) */
    int64_t nsparse;
    int64_t **sp_u, **sp_v, **sp_q; /* SOC: u,v  GenPow: p->sp_u, q->sp_v, r->sp_q */
    int64_t (*sp_D)[3];
} orc_map;

typedef struct {
    orc_csc K;
    orc_map map;
    int shape;
} orc_kktmat;

static void map_free(orc_map *mp) {
    free(mp->P); free(mp->A); free(mp->Hsblocks); free(mp->diagP); free(mp->diag_full);
    for (int64_t i = 0; i < mp->nsparse; i++) {
        if (mp->sp_u) free(mp->sp_u[i]);
        if (mp->sp_v) free(mp->sp_v[i]);
        if (mp->sp_q) free(mp->sp_q[i]);
    }
    free(mp->sp_u); free(mp->sp_v); free(mp->sp_q); free(mp->sp_D);
}
void orc_kktmat_free(orc_kktmat *km) {
    if (!km) return;
    free(km->K.colptr); free(km->K.rowval); free(km->K.nzval);
    map_free(&km->map);
    free(km);
}

/* kkt_assembly.rs:20-183 */
orc_kktmat *orc_assemble_kkt(int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi,
                             const double *Px, const int64_t *Ap, const int64_t *Ai,
                             const double *Ax, const orc_cones *cones, int shape) {
    orc_kktmat *km = (orc_kktmat *)calloc(1, sizeof(orc_kktmat));
    km->shape = shape;
    orc_csc P = {n, n, (int64_t *)Pp, (int64_t *)Pi, (double *)Px, Pp[n]};
    orc_csc A = {m, n, (int64_t *)Ap, (int64_t *)Ai, (double *)Ax, Ap[n]};
    orc_map *map = &km->map;
    /* LDLDataMap::new, datamaps.rs:365-404 */
    int64_t p = cones->pdim_total;
    map->n = n; map->m = m; map->p = p; map->nnzP = P.nnz; map->nnzA = A.nnz;
    map->P = (int64_t *)calloc((size_t)(P.nnz > 0 ? P.nnz : 1), sizeof(int64_t));
    map->A = (int64_t *)calloc((size_t)(A.nnz > 0 ? A.nnz : 1), sizeof(int64_t));
    map->nHs = cones->nblockvals;
    map->Hsblocks = (int64_t *)calloc((size_t)(map->nHs > 0 ? map->nHs : 1), sizeof(int64_t));
    map->diagP = (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t));
    map->diag_full = (int64_t *)calloc((size_t)(n + m + p > 0 ? n + m + p : 1), sizeof(int64_t));
    map->nsparse = cones->nsparse;
    size_t ns = (size_t)(map->nsparse > 0 ? map->nsparse : 1);
    map->sp_u = (int64_t **)calloc(ns, sizeof(int64_t *));
    map->sp_v = (int64_t **)calloc(ns, sizeof(int64_t *));
    map->sp_q = (int64_t **)calloc(ns, sizeof(int64_t *));
    map->sp_D = (int64_t(*)[3])calloc(ns, sizeof(int64_t[3]));
    int64_t nnz_vec = 0;
    {
        int64_t si = 0;
        for (int64_t i = 0; i < cones->ncones; i++) {
            const orc_cone *c = &cones->c[i];
            if (!c->sparse) continue;
            if (c->tag == CONE_SOC) {
                map->sp_u[si] = (int64_t *)calloc((size_t)c->numel, sizeof(int64_t));
                map->sp_v[si] = (int64_t *)calloc((size_t)c->numel, sizeof(int64_t));
                nnz_vec += 2 * c->numel; /* datamaps.rs:131-133 */
            } else {                     /* GenPow, datamaps.rs:248-250 */
                map->sp_u[si] = (int64_t *)calloc((size_t)c->numel, sizeof(int64_t));            /* p */
                map->sp_v[si] = (int64_t *)calloc((size_t)(c->dim > 0 ? c->dim : 1), sizeof(int64_t));   /* q */
                map->sp_q[si] = (int64_t *)calloc((size_t)(c->dim2 > 0 ? c->dim2 : 1), sizeof(int64_t)); /* r */
                nnz_vec += c->numel + c->dim + c->dim2;
            }
            si++;
        }
    }
    /* kkt_assembly.rs:33-46 */
    int64_t nnz_diagP = count_diagonal_entries_triu(&P);
    int64_t nnzKKT = P.nnz + n - nnz_diagP + A.nnz + map->nHs + nnz_vec + p;
    int64_t N = m + n + p;
    orc_csc *K = &km->K;
    K->m = N; K->n = N; K->nnz = nnzKKT;
    K->colptr = (int64_t *)calloc((size_t)N + 1, sizeof(int64_t));
    K->rowval = (int64_t *)calloc((size_t)(nnzKKT > 0 ? nnzKKT : 1), sizeof(int64_t));
    K->nzval = (double *)calloc((size_t)(nnzKKT > 0 ? nnzKKT : 1), sizeof(double));

    /* _kkt_assemble_colcounts, :53-103 */
    if (shape == SHAPE_TRIU) {
        colcount_block(K, &P, 0, 0);
        colcount_missing_diag(K, &P, 0);
        colcount_block(K, &A, n, 1);
    } else {
        colcount_missing_diag(K, &P, 0);
        colcount_block(K, &P, 0, 1);
        colcount_block(K, &A, 0, 0);
    }
    {
        int64_t pcol = m + n, si = 0;
        for (int64_t i = 0; i < cones->ncones; i++) {
            const orc_cone *c = &cones->c[i];
            int64_t row = c->cone_start + n;
            if (c->hs_diag) colcount_diag(K, row, c->numel);
            else colcount_dense_triangle(K, row, c->numel, shape);
            if (c->sparse) {
                if (c->tag == CONE_SOC) { /* datamaps.rs:149-171 */
                    if (shape == SHAPE_TRIU) {
                        colcount_colvec(K, c->numel, row, pcol);
                        colcount_colvec(K, c->numel, row, pcol + 1);
                    } else {
                        colcount_rowvec(K, c->numel, pcol, row);
                        colcount_rowvec(K, c->numel, pcol + 1, row);
                    }
                } else { /* datamaps.rs:266-292 */
                    if (shape == SHAPE_TRIU) {
                        colcount_colvec(K, c->dim, row, pcol);
                        colcount_colvec(K, c->dim2, row + c->dim, pcol + 1);
                        colcount_colvec(K, c->numel, row, pcol + 2);
                    } else {
                        colcount_rowvec(K, c->dim, pcol, row);
                        colcount_rowvec(K, c->dim2, pcol + 1, row + c->dim);
                        colcount_rowvec(K, c->numel, pcol + 2, row);
                    }
                }
                colcount_diag(K, pcol, c->pdim);
                pcol += c->pdim;
                si++;
            }
        }
    }
    /* _kkt_assemble_fill, :105-183 */
    colcount_to_colptr(K);
    if (shape == SHAPE_TRIU) {
        fill_block(K, &P, map->P, 0, 0, 0);
        fill_missing_diag(K, &P, 0);
        fill_block(K, &A, map->A, 0, n, 1);
    } else {
        fill_missing_diag(K, &P, 0);
        fill_block(K, &P, map->P, 0, 0, 1);
        fill_block(K, &A, map->A, n, 0, 0);
    }
    {
        int64_t pcol = m + n, si = 0;
        for (int64_t i = 0; i < cones->ncones; i++) {
            const orc_cone *c = &cones->c[i];
            int64_t row = c->cone_start + n;
            int64_t *block = map->Hsblocks + c->block_start;
            if (c->hs_diag) fill_diag(K, block, row, c->numel);
            else fill_dense_triangle(K, block, row, c->numel, shape);
            if (c->sparse) {
                if (c->tag == CONE_SOC) { /* datamaps.rs:173-197: v first, then u */
                    if (shape == SHAPE_TRIU) {
                        fill_colvec(K, map->sp_v[si], c->numel, row, pcol);
                        fill_colvec(K, map->sp_u[si], c->numel, row, pcol + 1);
                    } else {
                        fill_rowvec(K, map->sp_v[si], c->numel, pcol, row);
                        fill_rowvec(K, map->sp_u[si], c->numel, pcol + 1, row);
                    }
                } else { /* datamaps.rs:294-319: q, r, p */
                    if (shape == SHAPE_TRIU) {
                        fill_colvec(K, map->sp_v[si], c->dim, row, pcol);
                        fill_colvec(K, map->sp_q[si], c->dim2, row + c->dim, pcol + 1);
                        fill_colvec(K, map->sp_u[si], c->numel, row, pcol + 2);
                    } else {
                        fill_rowvec(K, map->sp_v[si], c->dim, pcol, row);
                        fill_rowvec(K, map->sp_q[si], c->dim2, pcol + 1, row + c->dim);
                        fill_rowvec(K, map->sp_u[si], c->numel, pcol + 2, row);
                    }
                }
                fill_diag(K, map->sp_D[si], pcol, c->pdim);
                pcol += c->pdim;
                si++;
            }
        }
    }
    backshift_colptrs(K);
    if (shape == SHAPE_TRIU) { /* :165-182 */
        for (int64_t i = 0; i < N; i++) map->diag_full[i] = K->colptr[i + 1] - 1;
        for (int64_t i = 0; i < n; i++) map->diagP[i] = K->colptr[i + 1] - 1;
    } else {
        for (int64_t i = 0; i < N; i++) map->diag_full[i] = K->colptr[i];
        for (int64_t i = 0; i < n; i++) map->diagP[i] = K->colptr[i];
    }
    return km;
}
int64_t orc_kktmat_dim(const orc_kktmat *km) { return km->K.n; }
int64_t orc_kktmat_nnz(const orc_kktmat *km) { return km->K.nnz; }
const int64_t *orc_kktmat_colptr(const orc_kktmat *km) { return km->K.colptr; }
const int64_t *orc_kktmat_rowval(const orc_kktmat *km) { return km->K.rowval; }
double *orc_kktmat_nzval(orc_kktmat *km) { return km->K.nzval; }
const int64_t *orc_kktmat_map_P(const orc_kktmat *km) { return km->map.P; }
const int64_t *orc_kktmat_map_A(const orc_kktmat *km) { return km->map.A; }
const int64_t *orc_kktmat_map_Hs(const orc_kktmat *km) { return km->map.Hsblocks; }
const int64_t *orc_kktmat_map_diagP(const orc_kktmat *km) { return km->map.diagP; }
const int64_t *orc_kktmat_map_diag_full(const orc_kktmat *km) { return km->map.diag_full; }
int64_t orc_kktmat_nsparse(const orc_kktmat *km) { return km->map.nsparse; }
const int64_t *orc_kktmat_map_u(const orc_kktmat *km, int64_t i) { return km->map.sp_u[i]; }
const int64_t *orc_kktmat_map_v(const orc_kktmat *km, int64_t i) { return km->map.sp_v[i]; }
const int64_t *orc_kktmat_map_q(const orc_kktmat *km, int64_t i) { return km->map.sp_q[i]; }
const int64_t *orc_kktmat_map_D(const orc_kktmat *km, int64_t i) { return km->map.sp_D[i]; }

/* ------------------------------------------------------------------------ */
/* DirectLDLKKTSolver (with the QDLDL engine)                                */
/* ------------------------------------------------------------------------ */
typedef struct {
    /* CoreSettings fields consumed by the path, settings.rs:139-181 */
    int static_reg_enable;
    double static_reg_constant, static_reg_proportional;
    int dynamic_reg_enable; /* NB ignored by the qdldl adapter (ldlsolvers/qdldl.rs:38) */
    double dynamic_reg_eps, dynamic_reg_delta;
    int ir_enable;
    double ir_reltol, ir_abstol;
    int32_t ir_max_iter;
    double ir_stop_ratio;
} orc_settings;

void orc_settings_default(orc_settings *s) { /* settings.rs:139-181 defaults */
    s->static_reg_enable = 1;
    s->static_reg_constant = 1e-8;
    s->static_reg_proportional = 2.220446049250313e-16 * 2.220446049250313e-16;
    s->dynamic_reg_enable = 1;
    s->dynamic_reg_eps = 1e-13;
    s->dynamic_reg_delta = 2e-7;
    s->ir_enable = 1;
    s->ir_reltol = 1e-13;
    s->ir_abstol = 1e-12;
    s->ir_max_iter = 10;
    s->ir_stop_ratio = 5.0;
}

typedef struct {
    int64_t m, n, p, N;
    double *x, *b, *work1, *work2;
    orc_kktmat *km;
    signed char *dsigns;
    double *Hsblocks;
    orc_qdldl *ldl;
    orc_cones *cones; /* not owned */
    double diagonal_regularizer;
    int32_t last_ir_iters; /* oracle-only instrumentation */
} orc_kktsolver;

void orc_kktsolver_free(orc_kktsolver *ks) {
    if (!ks) return;
    free(ks->x); free(ks->b); free(ks->work1); free(ks->work2); free(ks->dsigns);
    free(ks->Hsblocks);
    if (ks->ldl) orc_qdldl_free(ks->ldl);
    orc_kktmat_free(ks->km);
    free(ks);
}

/* directldlkktsolver.rs:60-118 (+ _fill_signs :392-405), qdldl engine with an
 * explicit permutation (ldlsolvers/qdldl.rs:18-49: logical=true,
 * regularize_enable=true, eps/delta from settings). */
orc_kktsolver *orc_kktsolver_new(int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi,
                                 const double *Px, const int64_t *Ap, const int64_t *Ai,
                                 const double *Ax, orc_cones *cones, const orc_settings *st,
                                 const int64_t *perm, int *err) {
    orc_kktsolver *ks = (orc_kktsolver *)calloc(1, sizeof(orc_kktsolver));
    ks->km = orc_assemble_kkt(n, m, Pp, Pi, Px, Ap, Ai, Ax, cones, SHAPE_TRIU);
    int64_t p = ks->km->map.p, N = n + m + p;
    ks->m = m; ks->n = n; ks->p = p; ks->N = N;
    ks->cones = cones;
    size_t NN = (size_t)(N > 0 ? N : 1);
    ks->x = (double *)calloc(NN, sizeof(double));
    ks->b = (double *)calloc(NN, sizeof(double));
    ks->work1 = (double *)calloc(NN, sizeof(double));
    ks->work2 = (double *)calloc(NN, sizeof(double));
    ks->dsigns = (signed char *)malloc(NN);
    for (int64_t i = 0; i < N; i++) ks->dsigns[i] = 1;
    for (int64_t i = n; i < n + m; i++) ks->dsigns[i] = -1;
    {
        int64_t pp = m + n;
        for (int64_t i = 0; i < cones->ncones; i++) {
            const orc_cone *c = &cones->c[i];
            if (!c->sparse) continue;
            if (c->tag == CONE_SOC) { /* datamaps.rs:134-136 */
                ks->dsigns[pp] = -1; ks->dsigns[pp + 1] = 1;
            } else {                  /* datamaps.rs:251-253 */
                ks->dsigns[pp] = -1; ks->dsigns[pp + 1] = -1; ks->dsigns[pp + 2] = 1;
            }
            pp += c->pdim;
        }
    }
    ks->Hsblocks = (double *)calloc((size_t)(cones->nblockvals > 0 ? cones->nblockvals : 1), sizeof(double));
    int64_t *ident = NULL;
    if (!perm) {
        ident = (int64_t *)malloc(NN * sizeof(int64_t));
        for (int64_t i = 0; i < N; i++) ident[i] = i;
        perm = ident;
    }
    int rc = orc_qdldl_new(&ks->ldl, N, N, ks->km->K.colptr, ks->km->K.rowval, ks->km->K.nzval,
                           perm, ks->dsigns, 1, 1, st->dynamic_reg_eps, st->dynamic_reg_delta);
    free(ident);
    if (err) *err = rc;
    if (rc != 0) {
        orc_kktsolver_free(ks);
        return NULL;
    }
    return ks;
}

/* directldlkktsolver.rs:351-390 */
static void update_values(orc_kktsolver *ks, const int64_t *idx, const double *v, int64_t k) {
    for (int64_t i = 0; i < k; i++) ks->km->K.nzval[idx[i]] = v[i];
    orc_qdldl_update_values(ks->ldl, idx, v, k);
}
static void scale_values(orc_kktsolver *ks, const int64_t *idx, double s, int64_t k) {
    for (int64_t i = 0; i < k; i++) ks->km->K.nzval[idx[i]] *= s;
    orc_qdldl_scale_values(ks->ldl, idx, s, k);
}

/* directldlkktsolver.rs:217-264 + :324-329 */
static int regularize_and_refactor(orc_kktsolver *ks, const orc_settings *st) {
    orc_map *map = &ks->km->map;
    double *Kx = ks->km->K.nzval;
    double *diag_kkt = ks->work1, *diag_shifted = ks->work2;
    int64_t N = ks->N;
    if (st->static_reg_enable) {
        for (int64_t i = 0; i < N; i++) diag_kkt[i] = Kx[map->diag_full[i]];
        double maxdiag = orc_norm_inf(diag_kkt, N);
        double eps = st->static_reg_constant + st->static_reg_proportional * maxdiag;
        for (int64_t i = 0; i < N; i++)
            diag_shifted[i] = (ks->dsigns[i] == 1) ? diag_kkt[i] + eps : diag_kkt[i] - eps;
        update_values(ks, map->diag_full, diag_shifted, N);
        ks->diagonal_regularizer = eps;
    }
    int rc = orc_qdldl_refactor(ks->ldl); /* reference unwrap()s: zero pivot would panic */
    int ok = (rc == 0) && orc_qdldl_dinv_is_finite(ks->ldl);
    if (st->static_reg_enable)
        for (int64_t i = 0; i < N; i++) Kx[map->diag_full[i]] = diag_kkt[i];
    return ok;
}

/* directldlkktsolver.rs:134-158.  hs_override: optional full Hsblocks vector
 * whose entries are used for cone types the oracle does not restate
 * (Exp/Pow/GenPow/PSD); may be NULL. */
int orc_kktsolver_update(orc_kktsolver *ks, const orc_settings *st, const double *hs_override) {
    orc_map *map = &ks->km->map;
    orc_cones *cones = ks->cones;
    if (hs_override) memcpy(ks->Hsblocks, hs_override, (size_t)cones->nblockvals * sizeof(double));
    orc_cones_get_Hs(cones, ks->Hsblocks);
    for (int64_t i = 0; i < cones->nblockvals; i++) ks->Hsblocks[i] = -ks->Hsblocks[i];
    update_values(ks, map->Hsblocks, ks->Hsblocks, cones->nblockvals);
    int64_t si = 0;
    for (int64_t i = 0; i < cones->ncones; i++) {
        const orc_cone *c = &cones->c[i];
        if (!c->sparse) continue;
        if (c->tag == CONE_SOC) { /* datamaps.rs:199-220 */
            double eta2 = c->eta * c->eta;
            update_values(ks, map->sp_u[si], c->u, c->numel);
            update_values(ks, map->sp_v[si], c->v, c->numel);
            scale_values(ks, map->sp_u[si], -eta2, c->numel);
            scale_values(ks, map->sp_v[si], -eta2, c->numel);
            double dd[2] = {-eta2, eta2};
            update_values(ks, map->sp_D[si], dd, 2);
        } else if (c->tag == CONE_GENPOW) { /* datamaps.rs:322-343: maps p -> sp_u, q -> sp_v, r -> sp_q */
            const double sqrtmu = sqrt(c->gmu);
            update_values(ks, map->sp_v[si], c->gq, c->dim);
            update_values(ks, map->sp_q[si], c->gr, c->dim2);
            update_values(ks, map->sp_u[si], c->gp, c->numel);
            scale_values(ks, map->sp_v[si], -sqrtmu, c->dim);
            scale_values(ks, map->sp_q[si], -sqrtmu, c->dim2);
            scale_values(ks, map->sp_u[si], -sqrtmu, c->numel);
            double dd[3] = {-1.0, -1.0, 1.0};
            update_values(ks, map->sp_D[si], dd, 3);
        }
        si++;
    }
    return regularize_and_refactor(ks, st);
}

/* directldlkktsolver.rs:160-166 */
void orc_kktsolver_setrhs(orc_kktsolver *ks, const double *rhsx, const double *rhsz) {
    memcpy(ks->b, rhsx, (size_t)ks->n * sizeof(double));
    memcpy(ks->b + ks->n, rhsz, (size_t)ks->m * sizeof(double));
    for (int64_t i = ks->n + ks->m; i < ks->N; i++) ks->b[i] = 0.0;
}

/* ldlsolvers/qdldl.rs:91-95 */
static void ldl_solve(orc_kktsolver *ks, double *x, const double *b) {
    memcpy(x, b, (size_t)ks->N * sizeof(double));
    orc_qdldl_solve(ks->ldl, x);
}
/* directldlkktsolver.rs:334-347 */
static double get_refine_error(orc_kktsolver *ks, double *e, const double *b, const double *xi) {
    memcpy(e, b, (size_t)ks->N * sizeof(double));
    orc_symv(ks->N, ks->km->K.colptr, ks->km->K.rowval, ks->km->K.nzval, e, xi, -1.0, 1.0);
    return orc_norm_inf(e, ks->N);
}
/* directldlkktsolver.rs:266-321 */
static int iterative_refinement(orc_kktsolver *ks, const orc_settings *st) {
    double *x = ks->x, *b = ks->b, *e = ks->work1, *dx = ks->work2;
    int64_t N = ks->N;
    ks->last_ir_iters = 0;
    double normb = orc_norm_inf(b, N);
    double norme = get_refine_error(ks, e, b, x);
    if (!isfinite(norme)) return 0;
    for (int32_t it = 0; it < st->ir_max_iter; it++) {
        if (norme <= st->ir_abstol + st->ir_reltol * normb) break;
        double lastnorme = norme;
        ldl_solve(ks, dx, e);
        for (int64_t i = 0; i < N; i++) dx[i] = 1.0 * x[i] + 1.0 * dx[i];
        norme = get_refine_error(ks, e, b, dx);
        ks->last_ir_iters += 1;
        if (!isfinite(norme)) return 0;
        double improved = lastnorme / norme;
        if (improved < st->ir_stop_ratio) {
            if (improved > 1.0) { double *t = x; x = dx; dx = t; }
            break;
        }
        { double *t = x; x = dx; dx = t; }
    }
    /* mem::swap on Vec<T> swaps the buffers; keep ks->x pointing at the result */
    if (x != ks->x) { ks->work2 = ks->x; ks->x = x; }
    return 1;
}
/* directldlkktsolver.rs:168-189 + getlhs :205-215 */
int orc_kktsolver_solve(orc_kktsolver *ks, const orc_settings *st, double *lhsx, double *lhsz) {
    ldl_solve(ks, ks->x, ks->b);
    int ok = st->ir_enable ? iterative_refinement(ks, st) : all_finite(ks->x, ks->N);
    if (ok) {
        if (lhsx) memcpy(lhsx, ks->x, (size_t)ks->n * sizeof(double));
        if (lhsz) memcpy(lhsz, ks->x + ks->n, (size_t)ks->m * sizeof(double));
    }
    return ok;
}
int64_t orc_kktsolver_dim(const orc_kktsolver *ks) { return ks->N; }
int64_t orc_kktsolver_pdim(const orc_kktsolver *ks) { return ks->p; }
const double *orc_kktsolver_x(const orc_kktsolver *ks) { return ks->x; }
double *orc_kktsolver_b(orc_kktsolver *ks) { return ks->b; }
orc_kktmat *orc_kktsolver_kktmat(orc_kktsolver *ks) { return ks->km; }
orc_qdldl *orc_kktsolver_ldl(orc_kktsolver *ks) { return ks->ldl; }
const signed char *orc_kktsolver_dsigns(const orc_kktsolver *ks) { return ks->dsigns; }
int32_t orc_kktsolver_last_ir_iters(const orc_kktsolver *ks) { return ks->last_ir_iters; }
double orc_kktsolver_regularizer(const orc_kktsolver *ks) { return ks->diagonal_regularizer; }
/* full-N solve straight on the internal b (lets tests set sparse-cone rows too) */
int orc_kktsolver_solve_full(orc_kktsolver *ks, const orc_settings *st, const double *b, double *x) {
    memcpy(ks->b, b, (size_t)ks->N * sizeof(double));
    ldl_solve(ks, ks->x, ks->b);
    int ok = st->ir_enable ? iterative_refinement(ks, st) : all_finite(ks->x, ks->N);
    if (ok && x) memcpy(x, ks->x, (size_t)ks->N * sizeof(double));
    return ok;
}

/* ======================================================================== */
/* L3: DefaultKKTSystem (default/kktsystem.rs:108-292) and DefaultResiduals  */
/* (default/residuals.rs:69-111) -- the caller either side of the KKT solve. */
/* Test infrastructure only, like the rest of this file.                     */
/* ======================================================================== */
/* matrix_math.rs:258-299 _csc_axpby_N / :301-343 _csc_axpby_T */
void orc_gemv_N(int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, double *y,
                const double *x, double a, double b) {
    for (int64_t i = 0; i < m; i++) y[i] = b == 0.0 ? 0.0 : b * y[i];
    if (a == 0.0) return;
    for (int64_t j = 0; j < n; j++)
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            if (a == 1.0) y[Ai[p]] += Ax[p] * x[j];
            else if (a == -1.0) y[Ai[p]] -= Ax[p] * x[j];
            else y[Ai[p]] += a * Ax[p] * x[j];
        }
}
void orc_gemv_T(int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, double *y,
                const double *x, double a, double b) {
    (void)m;
    for (int64_t j = 0; j < n; j++) y[j] = b == 0.0 ? 0.0 : b * y[j];
    if (a == 0.0) return;
    for (int64_t j = 0; j < n; j++)
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            if (a == 1.0) y[j] += Ax[p] * x[Ai[p]];
            else if (a == -1.0) y[j] -= Ax[p] * x[Ai[p]];
            else y[j] += a * Ax[p] * x[Ai[p]];
        }
}
double orc_dot(const double *a, const double *b, int64_t n) { return dotp(a, b, n); }

typedef struct {
    orc_kktsolver *ks; /* borrowed */
    orc_cones *cones;  /* borrowed */
    int64_t n, m, nnzP, nnzA;
    int64_t *Pp, *Pi, *Ap, *Ai;
    double *Px, *Ax, *q, *b;
    double *x1, *z1, *x2, *z2, *workx, *workz, *work_conic;
} orc_kktsystem;

static void *dupmem(const void *src, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (bytes) memcpy(p, src, bytes);
    return p;
}
void orc_kktsystem_free(orc_kktsystem *s) {
    if (!s) return;
    free(s->Pp); free(s->Pi); free(s->Ap); free(s->Ai); free(s->Px); free(s->Ax); free(s->q); free(s->b);
    free(s->x1); free(s->z1); free(s->x2); free(s->z2); free(s->workx); free(s->workz); free(s->work_conic);
    free(s);
}
/* kktsystem.rs:38-88 (the KKTSolver itself is built by the caller and borrowed) */
orc_kktsystem *orc_kktsystem_new(orc_kktsolver *ks, orc_cones *cones, int64_t n, int64_t m, const int64_t *Pp,
                                 const int64_t *Pi, const double *Px, const int64_t *Ap, const int64_t *Ai,
                                 const double *Ax, const double *q, const double *b) {
    orc_kktsystem *s = (orc_kktsystem *)calloc(1, sizeof(*s));
    s->ks = ks; s->cones = cones; s->n = n; s->m = m;
    s->nnzP = Pp[n]; s->nnzA = Ap[n];
    s->Pp = (int64_t *)dupmem(Pp, (size_t)(n + 1) * 8); s->Pi = (int64_t *)dupmem(Pi, (size_t)s->nnzP * 8);
    s->Px = (double *)dupmem(Px, (size_t)s->nnzP * 8);
    s->Ap = (int64_t *)dupmem(Ap, (size_t)(n + 1) * 8); s->Ai = (int64_t *)dupmem(Ai, (size_t)s->nnzA * 8);
    s->Ax = (double *)dupmem(Ax, (size_t)s->nnzA * 8);
    s->q = (double *)dupmem(q, (size_t)n * 8); s->b = (double *)dupmem(b, (size_t)m * 8);
    s->x1 = (double *)calloc((size_t)n + 1, 8); s->x2 = (double *)calloc((size_t)n + 1, 8);
    s->workx = (double *)calloc((size_t)n + 1, 8);
    s->z1 = (double *)calloc((size_t)m + 1, 8); s->z2 = (double *)calloc((size_t)m + 1, 8);
    s->workz = (double *)calloc((size_t)m + 1, 8); s->work_conic = (double *)calloc((size_t)m + 1, 8);
    return s;
}
const double *orc_kktsystem_x2(const orc_kktsystem *s) { return s->x2; }
const double *orc_kktsystem_z2(const orc_kktsystem *s) { return s->z2; }

/* kktsystem.rs:264-279 */
static int kktsystem_solve_constant_rhs(orc_kktsystem *s, const orc_settings *st) {
    for (int64_t i = 0; i < s->n; i++) s->workx[i] = -1.0 * s->q[i]; /* axpby(-1, q, 0) */
    orc_kktsolver_setrhs(s->ks, s->workx, s->b);
    return orc_kktsolver_solve(s->ks, st, s->x2, s->z2);
}
/* kktsystem.rs:108-125 */
int orc_kktsystem_update(orc_kktsystem *s, const orc_settings *st, const double *hs_override) {
    if (!orc_kktsolver_update(s->ks, st, hs_override)) return 0;
    return kktsystem_solve_constant_rhs(s, st);
}
/* kktsystem.rs:127-209.  vars/rhs/lhs are (x[n], z[m], s[m], tau, kappa); lhs_tk = {tau, kappa} out.
 * step_direction: 0 = Affine, 1 = Combined */
int orc_kktsystem_solve(orc_kktsystem *s, const orc_settings *st, double *lhs_x, double *lhs_z, double *lhs_s,
                        double *lhs_tk, const double *rhs_x, const double *rhs_z, const double *rhs_s,
                        double rhs_tau, double rhs_kappa, const double *var_x, const double *var_z,
                        const double *var_s, double var_tau, double var_kappa, int step_direction) {
    const int64_t n = s->n, m = s->m;
    double *workx = s->workx, *workz = s->workz, *cterm = s->work_conic;
    memcpy(workx, rhs_x, (size_t)n * 8);
    if (step_direction == 0) memcpy(cterm, var_s, (size_t)m * 8);
    else orc_cones_ds_from_dz_offset(s->cones, cterm, rhs_s, var_z);
    for (int64_t i = 0; i < m; i++) workz[i] = 1.0 * cterm[i] + (-1.0) * rhs_z[i]; /* waxpby */
    orc_kktsolver_setrhs(s->ks, workx, workz);
    if (!orc_kktsolver_solve(s->ks, st, s->x1, s->z1)) return 0;
    /* solve for dtau */
    double *xi = workx;
    const double rtau = 1.0 / var_tau;
    for (int64_t i = 0; i < n; i++) xi[i] = rtau * var_x[i]; /* axpby(recip(tau), x, 0) */
    double tau_num = rhs_tau - rhs_kappa / var_tau + dotp(s->q, s->x1, n) + dotp(s->b, s->z1, m) +
                     2.0 * orc_quad_form_triu(n, s->Pp, s->Pi, s->Px, xi, s->x1);
    for (int64_t i = 0; i < n; i++) xi[i] = -1.0 * s->x2[i] + 1.0 * xi[i]; /* axpby(-1, x2, 1) */
    double tau_den = var_kappa / var_tau - dotp(s->q, s->x2, n) - dotp(s->b, s->z2, m);
    tau_den += orc_quad_form_triu(n, s->Pp, s->Pi, s->Px, xi, xi) -
               orc_quad_form_triu(n, s->Pp, s->Pi, s->Px, s->x2, s->x2);
    const double ltau = tau_num / tau_den;
    for (int64_t i = 0; i < n; i++) lhs_x[i] = 1.0 * s->x1[i] + ltau * s->x2[i];
    for (int64_t i = 0; i < m; i++) lhs_z[i] = 1.0 * s->z1[i] + ltau * s->z2[i];
    orc_cones_mul_Hs(s->cones, lhs_s, lhs_z);
    for (int64_t i = 0; i < m; i++) lhs_s[i] = -1.0 * cterm[i] + (-1.0) * lhs_s[i]; /* axpby(-1, c, -1) */
    lhs_tk[0] = ltau;
    lhs_tk[1] = -(rhs_kappa + var_kappa * ltau) / var_tau;
    return 1;
}
/* kktsystem.rs:211-258 */
int orc_kktsystem_solve_initial_point(orc_kktsystem *s, const orc_settings *st, double *var_x, double *var_s,
                                      double *var_z) {
    const int64_t n = s->n, m = s->m;
    int ok;
    if (s->nnzP == 0) {
        for (int64_t i = 0; i < n; i++) s->workx[i] = 0.0;
        memcpy(s->workz, s->b, (size_t)m * 8);
        orc_kktsolver_setrhs(s->ks, s->workx, s->workz);
        ok = orc_kktsolver_solve(s->ks, st, var_x, var_s);
        for (int64_t i = 0; i < m; i++) var_s[i] = -var_s[i];
        if (!ok) return ok;
        for (int64_t i = 0; i < n; i++) s->workx[i] = -1.0 * s->q[i];
        for (int64_t i = 0; i < m; i++) s->workz[i] = 0.0;
        orc_kktsolver_setrhs(s->ks, s->workx, s->workz);
        ok = orc_kktsolver_solve(s->ks, st, NULL, var_z);
    } else {
        for (int64_t i = 0; i < n; i++) s->workx[i] = -s->q[i];
        memcpy(s->workz, s->b, (size_t)m * 8);
        orc_kktsolver_setrhs(s->ks, s->workx, s->workz);
        ok = orc_kktsolver_solve(s->ks, st, var_x, var_z);
        for (int64_t i = 0; i < m; i++) var_s[i] = -var_z[i];
    }
    return ok;
}
/* residuals.rs:69-111.  out5 = {rtau, dot_qx, dot_bz, dot_sz, dot_xPx} */
void orc_residuals_update(const orc_kktsystem *s, const double *var_x, const double *var_z, const double *var_s,
                          double var_tau, double var_kappa, double *rx, double *rz, double *rx_inf,
                          double *rz_inf, double *Px, double *out5) {
    const int64_t n = s->n, m = s->m;
    const double qx = dotp(s->q, var_x, n), bz = dotp(s->b, var_z, m), sz = dotp(var_s, var_z, m);
    for (int64_t i = 0; i < n; i++) Px[i] = 0.0;
    orc_symv(n, s->Pp, s->Pi, s->Px, Px, var_x, 1.0, 0.0);
    const double xPx = dotp(var_x, Px, n);
    orc_gemv_T(m, n, s->Ap, s->Ai, s->Ax, rx_inf, var_z, -1.0, 0.0);
    memcpy(rz_inf, var_s, (size_t)m * 8);
    orc_gemv_N(m, n, s->Ap, s->Ai, s->Ax, rz_inf, var_x, 1.0, 1.0);
    for (int64_t i = 0; i < n; i++) rx[i] = -1.0 * Px[i] + (-var_tau) * s->q[i];
    for (int64_t i = 0; i < n; i++) rx[i] = 1.0 * rx_inf[i] + 1.0 * rx[i];
    for (int64_t i = 0; i < m; i++) rz[i] = 1.0 * rz_inf[i] + (-var_tau) * s->b[i];
    out5[0] = qx + bz + var_kappa + xPx / var_tau;
    out5[1] = qx; out5[2] = bz; out5[3] = sz; out5[4] = xPx;
}

/* compositecone.rs:208-214: nonnegativecone.rs:64-66, socone.rs:110-112, zerocone.rs:63-69 */
void orc_cones_scaled_unit_shift(const orc_cones *cs, double *z, double alpha, int primal_cone) {
    for (int64_t i = 0; i < cs->ncones; i++) {
        const orc_cone *c = &cs->c[i];
        double *zi = z + c->cone_start;
        if (c->tag == CONE_ZERO) {
            if (primal_cone)
                for (int64_t k = 0; k < c->numel; k++) zi[k] = 0.0;
        } else if (c->tag == CONE_NONNEG) {
            for (int64_t k = 0; k < c->numel; k++) zi[k] += alpha;
        } else if (c->tag == CONE_SOC) {
            zi[0] += alpha;
        }
    }
}
/* compositecone.rs:130-150 degree(): Zero 0, Nonnegative dim, SOC 1 */
int64_t orc_cones_degree(const orc_cones *cs) {
    int64_t d = 0;
    for (int64_t i = 0; i < cs->ncones; i++) {
        const orc_cone *c = &cs->c[i];
        if (c->tag == CONE_NONNEG) d += c->numel;
        else if (c->tag == CONE_SOC) d += 1;
        else if (c->tag == CONE_EXP || c->tag == CONE_POW) d += 3;
        else if (c->tag == CONE_PSDTRI) d += c->dim;
        else if (c->tag == CONE_GENPOW) d += c->dim + 1; /* genpowcone.rs:76-78 */
    }
    return d;
}

/* compositecone.rs:155-157: false as soon as one cone (GenPow, genpowcone.rs:96-98) disallows it */
int orc_cones_allows_primal_dual_scaling(const orc_cones *cs) {
    for (int64_t i = 0; i < cs->ncones; i++)
        if (cs->c[i].tag == CONE_GENPOW) return 0;
    return 1;
}
