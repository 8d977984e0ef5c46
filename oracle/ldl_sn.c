/* Test / measurement infrastructure only (see oracle/__init__.py) -- NOT the reference, NOT the product.
 *
 * A SUPERNODAL, MULTI-THREADED host comparator for the KKT factor + solve path on systems with dense fronts
 * (BASELINE configs 2 and 5), where the reference itself would not pick its scalar QDLDL engine but faer's supernodal
 * LDL' (ldlsolvers/auto.rs:60-88, ldlsolvers/faer_ldl.rs:99-157: symbolic analysis once, then
 * `factorize_numeric_ldlt` and `solve_in_place` on a rayon pool).  faer is a Rust crate and cannot be built here; this
 * file is a plain multifrontal LDL' written for the comparison and is labelled "port-supernodal, NOT faer" wherever it is
 * quoted (bench.py: cpu_baseline_mt of the c2 / c5 lines):
 *   - elimination tree of the permuted matrix, postordered (an equivalent reordering: same fill);
 *   - column structures by merging the children's (only the structure of a supernode's first column is kept);
 *   - fundamental supernodes, then relaxed amalgamation of a last child into its parent while few explicit zeros
 *     are introduced (the banded config 2 has narrow fundamental supernodes);
 *   - numeric phase: one dense frontal matrix per supernode, assembled from K and the children's update matrices,
 *     its pivot columns factored by a blocked right-looking LDL' with the sign rule of src/qdldl/qdldl.rs:645-665
 *     (a pivot with the wrong sign or below eps becomes sign * delta), the Schur complement passed up;
 *   - OpenMP: the supernodes of one height of the assembly tree in parallel while there are at least as many as
 *     threads, otherwise one after the other with the dense updates themselves threaded;
 *   - level-scheduled supernodal substitutions and a gather-only symmetric product for the refinement residual.
 * Its D and its solutions are validated against the scalar oracle's by the tests (tests/test_oracle_supernodal.py).
 * Compiled on the box it is timed on: gcc -O3 -march=native -fopenmp (fused multiply-adds allowed: this is a
 * performance comparator, its results are checked to a tolerance, not bit for bit).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef int64_t i64;

typedef struct {
    i64 n, nsn, nnzL, nlev;
    int nthreads;
    i64 *perm, *iperm;      /* final order: perm[new] = old */
    i64 *Lp_, *Li_;         /* lower triangle of the permuted matrix by columns (rows ascending), diagonal first */
    i64 *Lsrc;              /* position of each such entry in the caller's value array (upper CSC of the ORIGINAL matrix) */
    i64 *sn_first;          /* nsn + 1: first column of every supernode (columns consecutive) */
    i64 *sn_rptr, *sn_rows; /* rows below the supernode's columns (ascending) */
    i64 *sn_parent, *sn_of;
    i64 *ch_ptr, *ch_idx;   /* children of every supernode in the assembly tree */
    i64 *Tp, *Ti, *Tsrc;    /* strict upper triangle of the ORIGINAL matrix by rows (for the residual) */
    i64 *lev_ptr, *lev_idx; /* supernodes by height in the assembly tree (leaves first) */
    i64 *xoff;              /* offset of supernode s's panel (f x w, column-major, f = w + r) in Lx */
    double *Lx, *D, *Dinv;
    /* numeric work: update matrices live until the parent has consumed them */
    double **upd;
    double flops;
} sn_t;

static void *xmalloc(size_t b) {
    void *p = malloc(b ? b : 8);
    if (!p) abort();
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s);
    if (!p) abort();
    return p;
}

void sn_free(sn_t *S) {
    if (!S) return;
    free(S->perm); free(S->iperm); free(S->Lp_); free(S->Li_); free(S->Lsrc); free(S->sn_first); free(S->sn_rptr);
    free(S->sn_rows); free(S->sn_parent); free(S->sn_of); free(S->ch_ptr); free(S->ch_idx); free(S->Tp); free(S->Ti); free(S->Tsrc); free(S->lev_ptr); free(S->lev_idx); free(S->xoff);
    free(S->Lx); free(S->D); free(S->Dinv);
    if (S->upd) for (i64 s = 0; s < S->nsn; s++) free(S->upd[s]);
    free(S->upd);
    free(S);
}

/* lower triangle (by columns, rows ascending, diagonal first) of P A P' for upper-CSC A; src = position in A's values */
static void permuted_lower(i64 n, const i64 *Ap, const i64 *Ai, const i64 *iperm, i64 **Lp_out, i64 **Li_out, i64 **Ls_out) {
    i64 *Lp = (i64 *)xcalloc((size_t)n + 1, sizeof(i64));
    const i64 nnz = Ap[n];
    for (i64 c = 0; c < n; c++)
        for (i64 p = Ap[c]; p < Ap[c + 1]; p++) {
            const i64 a = iperm[Ai[p]], b = iperm[c];
            Lp[(a < b ? a : b) + 1]++; /* entry (max, min) of the lower triangle: column min */
        }
    for (i64 j = 0; j < n; j++) Lp[j + 1] += Lp[j];
    i64 *Li = (i64 *)xmalloc((size_t)nnz * sizeof(i64)), *Ls = (i64 *)xmalloc((size_t)nnz * sizeof(i64));
    i64 *nx = (i64 *)xmalloc((size_t)(n + 1) * sizeof(i64));
    memcpy(nx, Lp, (size_t)(n + 1) * sizeof(i64));
    /* rows ascending: walk the rows r = 0 .. n-1 of the lower triangle in order, i.e. transpose twice -- simpler: fill,
     * then sort each column by row (columns are short except dense ones; insertion into a counting sort by row) */
    for (i64 c = 0; c < n; c++)
        for (i64 p = Ap[c]; p < Ap[c + 1]; p++) {
            const i64 a = iperm[Ai[p]], b = iperm[c];
            const i64 col = a < b ? a : b, row = a < b ? b : a;
            const i64 q = nx[col]++;
            Li[q] = row;
            Ls[q] = p;
        }
    /* sort every column by row with one global counting pass: (row, col) pairs bucketed by row, then re-emitted */
    {
        i64 *cnt = (i64 *)xcalloc((size_t)n + 1, sizeof(i64));
        for (i64 q = 0; q < nnz; q++) cnt[Li[q] + 1]++;
        for (i64 j = 0; j < n; j++) cnt[j + 1] += cnt[j];
        i64 *bc = (i64 *)xmalloc((size_t)nnz * sizeof(i64)), *bs = (i64 *)xmalloc((size_t)nnz * sizeof(i64));
        for (i64 c = 0; c < n; c++)
            for (i64 q = Lp[c]; q < Lp[c + 1]; q++) {
                const i64 t = cnt[Li[q]]++;
                bc[t] = c;
                bs[t] = Ls[q];
            }
        /* cnt[r] is now the END of row r's bucket; buckets are in row order, columns ascending inside */
        memcpy(nx, Lp, (size_t)(n + 1) * sizeof(i64));
        i64 t = 0;
        for (i64 r = 0; r < n; r++)
            for (; t < cnt[r]; t++) {
                const i64 q = nx[bc[t]]++;
                Li[q] = r;
                Ls[q] = bs[t];
            }
        free(cnt); free(bc); free(bs);
    }
    free(nx);
    *Lp_out = Lp; *Li_out = Li; *Ls_out = Ls;
}

/* elimination tree from the lower triangle by columns == upper triangle by rows (Liu's algorithm with path compression) */
static void etree_of(i64 n, const i64 *Lp, const i64 *Li, i64 *parent) {
    /* needs, for every row i, the columns k < i with L(i,k) != 0: walk columns k ascending, rows i > k: for the classic
     * algorithm we process i ascending and its k's; with column access we use the equivalent "for k ascending, for i in
     * col k" only after transposing.  Build the transpose (rows of the lower triangle). */
    i64 *Rp = (i64 *)xcalloc((size_t)n + 1, sizeof(i64));
    for (i64 k = 0; k < n; k++)
        for (i64 q = Lp[k]; q < Lp[k + 1]; q++)
            if (Li[q] != k) Rp[Li[q] + 1]++;
    for (i64 j = 0; j < n; j++) Rp[j + 1] += Rp[j];
    i64 *Rk = (i64 *)xmalloc((size_t)Rp[n] * sizeof(i64)), *nx = (i64 *)xmalloc((size_t)(n + 1) * sizeof(i64));
    memcpy(nx, Rp, (size_t)(n + 1) * sizeof(i64));
    for (i64 k = 0; k < n; k++)
        for (i64 q = Lp[k]; q < Lp[k + 1]; q++)
            if (Li[q] != k) Rk[nx[Li[q]]++] = k;
    i64 *anc = (i64 *)xmalloc((size_t)n * sizeof(i64));
    for (i64 i = 0; i < n; i++) {
        parent[i] = -1;
        anc[i] = -1;
        for (i64 t = Rp[i]; t < Rp[i + 1]; t++) {
            i64 k = Rk[t];
            while (k != -1 && k < i) {
                const i64 nxt = anc[k];
                anc[k] = i;
                if (nxt == -1) parent[k] = i;
                k = nxt;
            }
        }
    }
    free(Rp); free(Rk); free(nx); free(anc);
}

/* postorder of a forest given parent[]; post[k] = k-th node */
static void postorder(i64 n, const i64 *parent, i64 *post) {
    i64 *head = (i64 *)xmalloc((size_t)n * sizeof(i64)), *next = (i64 *)xmalloc((size_t)n * sizeof(i64));
    i64 *stack = (i64 *)xmalloc((size_t)n * sizeof(i64));
    for (i64 i = 0; i < n; i++) head[i] = -1;
    for (i64 i = n - 1; i >= 0; i--)
        if (parent[i] != -1) {
            next[i] = head[parent[i]];
            head[parent[i]] = i;
        }
    i64 k = 0;
    for (i64 r = 0; r < n; r++) {
        if (parent[r] != -1) continue;
        i64 top = 0;
        stack[0] = r;
        while (top >= 0) {
            const i64 p = stack[top], c = head[p];
            if (c == -1) {
                post[k++] = p;
                top--;
            } else {
                head[p] = next[c];
                stack[++top] = c;
            }
        }
    }
    free(head); free(next); free(stack);
}

static int cmp_i64(const void *a, const void *b) {
    const i64 x = *(const i64 *)a, y = *(const i64 *)b;
    return x < y ? -1 : x > y;
}

/* Analysis.  Ap / Ai: upper triangle of the (original-order) KKT matrix by columns; perm_in[new] = old. */
sn_t *sn_new(i64 n, const i64 *Ap, const i64 *Ai, const i64 *perm_in, int nthreads, double relax) {
    sn_t *S = (sn_t *)xcalloc(1, sizeof(sn_t));
    const int timing = getenv("LDLSN_TIMING") != NULL;
    double t0 = omp_get_wtime();
#define TICK(what) do { if (timing) { const double t1 = omp_get_wtime(); fprintf(stderr, "[ldl_sn] %-28s %.3f s\n", what, t1 - t0); t0 = t1; } } while (0)
    S->n = n;
    S->nthreads = nthreads < 1 ? 1 : nthreads;
    i64 *ip0 = (i64 *)xmalloc((size_t)n * sizeof(i64));
    for (i64 k = 0; k < n; k++) ip0[perm_in[k]] = k;
    i64 *Lp0, *Li0, *Ls0;
    permuted_lower(n, Ap, Ai, ip0, &Lp0, &Li0, &Ls0);
    i64 *par0 = (i64 *)xmalloc((size_t)n * sizeof(i64)), *post = (i64 *)xmalloc((size_t)n * sizeof(i64));
    TICK("permuted lower");
    etree_of(n, Lp0, Li0, par0);
    postorder(n, par0, post);
    TICK("etree + postorder");
    free(Lp0); free(Li0); free(Ls0);
    S->perm = (i64 *)xmalloc((size_t)n * sizeof(i64));
    S->iperm = (i64 *)xmalloc((size_t)n * sizeof(i64));
    for (i64 k = 0; k < n; k++) S->perm[k] = perm_in[post[k]];
    for (i64 k = 0; k < n; k++) S->iperm[S->perm[k]] = k;
    free(ip0); free(par0); free(post);
    permuted_lower(n, Ap, Ai, S->iperm, &S->Lp_, &S->Li_, &S->Lsrc);
    i64 *parent = (i64 *)xmalloc((size_t)n * sizeof(i64));
    etree_of(n, S->Lp_, S->Li_, parent);
    TICK("second permutation + etree");
    /* column structures, children merged into parents; a column's structure is dropped once its parent has it */
    i64 **st = (i64 **)xcalloc((size_t)n, sizeof(i64 *));
    i64 *cc = (i64 *)xcalloc((size_t)n, sizeof(i64)); /* entries below the diagonal */
    i64 *mark = (i64 *)xmalloc((size_t)n * sizeof(i64));
    i64 *head = (i64 *)xmalloc((size_t)n * sizeof(i64)), *next = (i64 *)xmalloc((size_t)n * sizeof(i64));
    i64 *nchild = (i64 *)xcalloc((size_t)n, sizeof(i64));
    for (i64 i = 0; i < n; i++) mark[i] = -1, head[i] = -1;
    for (i64 i = n - 1; i >= 0; i--)
        if (parent[i] != -1) {
            next[i] = head[parent[i]];
            head[parent[i]] = i;
            nchild[parent[i]]++;
        }
    i64 *tmp = (i64 *)xmalloc((size_t)n * sizeof(i64));
    double nnzL = 0;
    for (i64 j = 0; j < n; j++) {
        i64 m = 0;
        mark[j] = j;
        for (i64 q = S->Lp_[j]; q < S->Lp_[j + 1]; q++) {
            const i64 i = S->Li_[q];
            if (i != j && mark[i] != j) mark[i] = j, tmp[m++] = i;
        }
        for (i64 c = head[j]; c != -1; c = next[c]) {
            for (i64 t = 0; t < cc[c]; t++) {
                const i64 i = st[c][t];
                if (mark[i] != j) mark[i] = j, tmp[m++] = i;
            }
        }
        cc[j] = m;
        st[j] = (i64 *)xmalloc((size_t)m * sizeof(i64));
        memcpy(st[j], tmp, (size_t)m * sizeof(i64));
        nnzL += (double)m;
        /* children whose structure is no longer needed: all but those that may start... (a child's structure is only
         * needed again if it is the first column of a supernode: decided below from counts, so keep the FIRST-column
         * candidates: a child c is a non-first column iff parent[c-1] == c and cc[c-1] == cc[c] + 1; first columns keep) */
        for (i64 c = head[j]; c != -1; c = next[c]) {
            const int first = !(c > 0 && parent[c - 1] == c && cc[c - 1] == cc[c] + 1);
            if (!first) {
                free(st[c]);
                st[c] = NULL;
            }
        }
    }
    free(tmp);
    TICK("column structures");
    S->nnzL = (i64)nnzL;
    /* fundamental supernodes (consecutive columns, each the parent of the previous one, nested structures) */
    i64 *first = (i64 *)xmalloc((size_t)(n + 1) * sizeof(i64));
    i64 nsn = 0;
    for (i64 j = 0; j < n; j++) {
        const int cont = j > 0 && parent[j - 1] == j && cc[j - 1] == cc[j] + 1;
        if (!cont) first[nsn++] = j;
    }
    first[nsn] = n;
    /* relaxed amalgamation: supernode s joins the one that follows it when that one starts at the parent of s's last
     * column (s is then its last child in the postorder) and the explicit zeros stay below `relax` of the merged panel */
    {
        i64 *keep = (i64 *)xmalloc((size_t)(nsn + 1) * sizeof(i64));
        i64 m = 0;
        i64 cur_first = nsn ? first[0] : 0;
        double cur_nz = 0; /* true entries of the current merged group's panel */
        for (i64 s = 0; s < nsn; s++) {
            const i64 a = first[s], b = first[s + 1], w = b - a;
            const double nz = (double)w * (w + 1) / 2 + (double)w * cc[b - 1];
            if (a == cur_first) cur_nz = nz;
            int merge = 0;
            if (s + 1 < nsn && parent[b - 1] == b) {
                const i64 b2 = first[s + 2], w2 = b2 - b, wm = b2 - cur_first;
                const double nz2 = (double)w2 * (w2 + 1) / 2 + (double)w2 * cc[b2 - 1];
                const double full = (double)wm * (wm + 1) / 2 + (double)wm * cc[b2 - 1];
                const double zeros = full - (cur_nz + nz2);
                if (zeros <= relax * full || wm <= 8) {
                    merge = 1;
                    cur_nz += nz2; /* (the zeros become stored entries of the merged panel, but are not counted as true) */
                }
            }
            if (!merge) {
                keep[m++] = cur_first;
                cur_first = b;
            }
        }
        keep[m] = n;
        /* structure of a merged supernode's first column = its own columns after it + the structure of the LAST one's
         * group; rebuilt below from the last fundamental member */
        S->nsn = m;
        S->sn_first = (i64 *)xmalloc((size_t)(m + 1) * sizeof(i64));
        memcpy(S->sn_first, keep, (size_t)(m + 1) * sizeof(i64));
        free(keep);
    }
    /* rows of every supernode: the structure of its LAST column (sorted) */
    S->sn_rptr = (i64 *)xcalloc((size_t)S->nsn + 1, sizeof(i64));
    S->sn_of = (i64 *)xmalloc((size_t)n * sizeof(i64));
    /* the structure of a last column may have been dropped (non-first columns are): recompute it from the first
     * column of ITS fundamental supernode: struct(last) = struct(first) minus the columns in between */
    {
        i64 *ffirst = (i64 *)xmalloc((size_t)n * sizeof(i64)); /* first column of the fundamental supernode of j */
        for (i64 j = 0; j < n; j++) ffirst[j] = (j > 0 && parent[j - 1] == j && cc[j - 1] == cc[j] + 1) ? ffirst[j - 1] : j;
        for (i64 s = 0; s < S->nsn; s++) S->sn_rptr[s + 1] = S->sn_rptr[s] + cc[S->sn_first[s + 1] - 1];
        S->sn_rows = (i64 *)xmalloc((size_t)S->sn_rptr[S->nsn] * sizeof(i64));
        for (i64 s = 0; s < S->nsn; s++) {
            const i64 last = S->sn_first[s + 1] - 1, f0 = ffirst[last];
            i64 *out = S->sn_rows + S->sn_rptr[s], m = 0;
            for (i64 t = 0; t < cc[f0]; t++)
                if (st[f0][t] > last) out[m++] = st[f0][t];
            qsort(out, (size_t)m, sizeof(i64), cmp_i64);
            for (i64 j = S->sn_first[s]; j <= last; j++) S->sn_of[j] = s;
        }
        free(ffirst);
    }
    TICK("supernodes + rows");
    for (i64 j = 0; j < n; j++) free(st[j]);
    free(st); free(mark); free(head); free(next); free(nchild);
    /* assembly tree, heights, panel offsets, flop count */
    S->sn_parent = (i64 *)xmalloc((size_t)S->nsn * sizeof(i64));
    i64 *height = (i64 *)xcalloc((size_t)S->nsn, sizeof(i64));
    S->xoff = (i64 *)xmalloc((size_t)(S->nsn + 1) * sizeof(i64));
    i64 off = 0, nlev = 0;
    for (i64 s = 0; s < S->nsn; s++) {
        const i64 last = S->sn_first[s + 1] - 1, w = S->sn_first[s + 1] - S->sn_first[s], r = S->sn_rptr[s + 1] - S->sn_rptr[s];
        S->sn_parent[s] = parent[last] == -1 ? -1 : S->sn_of[parent[last]];
        S->xoff[s] = off;
        off += (w + r) * w;
        for (i64 k = 0; k < w; k++) S->flops += (double)(w + r - k) * (w + r - k);
    }
    S->xoff[S->nsn] = off;
    for (i64 s = 0; s < S->nsn; s++) {
        const i64 p = S->sn_parent[s];
        if (p != -1 && height[p] < height[s] + 1) height[p] = height[s] + 1;
        if (height[s] + 1 > nlev) nlev = height[s] + 1;
    }
    /* (children precede parents in the postorder, so one ascending pass fixes the heights) */
    S->ch_ptr = (i64 *)xcalloc((size_t)S->nsn + 1, sizeof(i64));
    S->ch_idx = (i64 *)xmalloc((size_t)S->nsn * sizeof(i64));
    for (i64 s = 0; s < S->nsn; s++)
        if (S->sn_parent[s] != -1) S->ch_ptr[S->sn_parent[s] + 1]++;
    for (i64 s = 0; s < S->nsn; s++) S->ch_ptr[s + 1] += S->ch_ptr[s];
    {
        i64 *nx = (i64 *)xmalloc((size_t)(S->nsn + 1) * sizeof(i64));
        memcpy(nx, S->ch_ptr, (size_t)(S->nsn + 1) * sizeof(i64));
        for (i64 s = 0; s < S->nsn; s++)
            if (S->sn_parent[s] != -1) S->ch_idx[nx[S->sn_parent[s]]++] = s;
        free(nx);
    }
    /* rows of the strict upper triangle of the original matrix */
    S->Tp = (i64 *)xcalloc((size_t)n + 1, sizeof(i64));
    for (i64 c = 0; c < n; c++)
        for (i64 p = Ap[c]; p < Ap[c + 1]; p++)
            if (Ai[p] != c) S->Tp[Ai[p] + 1]++;
    for (i64 j = 0; j < n; j++) S->Tp[j + 1] += S->Tp[j];
    S->Ti = (i64 *)xmalloc((size_t)S->Tp[n] * sizeof(i64));
    S->Tsrc = (i64 *)xmalloc((size_t)S->Tp[n] * sizeof(i64));
    {
        i64 *nx = (i64 *)xmalloc((size_t)(n + 1) * sizeof(i64));
        memcpy(nx, S->Tp, (size_t)(n + 1) * sizeof(i64));
        for (i64 c = 0; c < n; c++)
            for (i64 p = Ap[c]; p < Ap[c + 1]; p++)
                if (Ai[p] != c) {
                    const i64 t = nx[Ai[p]]++;
                    S->Ti[t] = c;
                    S->Tsrc[t] = p;
                }
        free(nx);
    }
    S->nlev = nlev;
    S->lev_ptr = (i64 *)xcalloc((size_t)nlev + 1, sizeof(i64));
    S->lev_idx = (i64 *)xmalloc((size_t)S->nsn * sizeof(i64));
    for (i64 s = 0; s < S->nsn; s++) S->lev_ptr[height[s] + 1]++;
    for (i64 l = 0; l < nlev; l++) S->lev_ptr[l + 1] += S->lev_ptr[l];
    {
        i64 *nx = (i64 *)xmalloc((size_t)(nlev + 1) * sizeof(i64));
        memcpy(nx, S->lev_ptr, (size_t)(nlev + 1) * sizeof(i64));
        for (i64 s = 0; s < S->nsn; s++) S->lev_idx[nx[height[s]]++] = s;
        free(nx);
    }
    free(height); free(parent); free(cc); free(first);
    S->Lx = (double *)xmalloc((size_t)off * sizeof(double));
    S->D = (double *)xmalloc((size_t)n * sizeof(double));
    S->Dinv = (double *)xmalloc((size_t)n * sizeof(double));
    S->upd = (double **)xcalloc((size_t)S->nsn, sizeof(double *));
    TICK("tree, offsets, allocation");
#undef TICK
    return S;
}

i64 sn_nsn(const sn_t *S) { return S->nsn; }
i64 sn_nnzL(const sn_t *S) { return S->nnzL; }
i64 sn_panel_entries(const sn_t *S) { return S->xoff[S->nsn]; }
i64 sn_levels(const sn_t *S) { return S->nlev; }
double sn_flops(const sn_t *S) { return S->flops; }
const i64 *sn_perm(const sn_t *S) { return S->perm; }
const double *sn_D(const sn_t *S) { return S->D; }

/* C[i, j] -= sum_k A[i, k] * W[k, j]  for i in [i0, i1), j in [0, nj), k in [0, nk); C, A column-major (ldc, lda), W is
 * nk x nj row-major (W[k * nj + j]); only entries with (row offset) i >= jrow0 + j are needed when `lower` (the rest
 * is computed too in whole tiles: cheaper than masking) */
typedef double v8d __attribute__((vector_size(64), aligned(8)));
/* C[i, j] -= sum_k A[i, k] * W[k, j]  for i in [i0, i1), j in [0, nj), k in [0, nk <= 192); C, A column-major (ldc, lda),
 * W is nk x nj row-major.  Rows go in blocks of 128 whose slice of A is packed into 16-row micro-panels first (the
 * columns of A are lda apart: unpacked, every k is another page); a 16 x 4 tile of C lives in eight 512-bit
 * accumulators. */
static void gemm_sub(double *C, i64 ldc, const double *A, i64 lda, const double *W, i64 nj, i64 nk, i64 i0, i64 i1) {
    enum { MR = 16, NR = 4, IB = 128, KMAX = 192 };
    double pack[IB * KMAX] __attribute__((aligned(64)));
    for (i64 ib = i0; ib < i1; ib += IB) {
        const i64 ie = ib + IB < i1 ? ib + IB : i1, nfull = (ie - ib) / MR;
        for (i64 m = 0; m < nfull; m++) {
            double *pm = pack + m * MR * nk;
            const double *a = A + ib + m * MR;
            for (i64 k = 0; k < nk; k++, a += lda, pm += MR) {
                *(v8d *)pm = *(const v8d *)a;
                *(v8d *)(pm + 8) = *(const v8d *)(a + 8);
            }
        }
        for (i64 j0 = 0; j0 < nj; j0 += NR) {
            const i64 jn = nj - j0 < NR ? nj - j0 : NR;
            i64 i = ib;
            if (jn == NR) {
                for (i64 m = 0; m < nfull; m++, i += MR) {
                    v8d c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0}, c20 = {0}, c21 = {0}, c30 = {0}, c31 = {0};
                    const double *a = pack + m * MR * nk;
                    const double *w = W + j0;
                    for (i64 k = 0; k < nk; k++, a += MR, w += nj) {
                        const v8d a0 = *(const v8d *)a, a1 = *(const v8d *)(a + 8);
                        const double w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
                        c00 += a0 * w0; c01 += a1 * w0;
                        c10 += a0 * w1; c11 += a1 * w1;
                        c20 += a0 * w2; c21 += a1 * w2;
                        c30 += a0 * w3; c31 += a1 * w3;
                    }
                    double *c = C + j0 * ldc + i;
                    *(v8d *)c -= c00; *(v8d *)(c + 8) -= c01; c += ldc;
                    *(v8d *)c -= c10; *(v8d *)(c + 8) -= c11; c += ldc;
                    *(v8d *)c -= c20; *(v8d *)(c + 8) -= c21; c += ldc;
                    *(v8d *)c -= c30; *(v8d *)(c + 8) -= c31;
                }
            }
            for (; i < ie; i++) /* ragged rows / columns */
                for (i64 jj = 0; jj < jn; jj++) {
                    double s = 0.0;
                    for (i64 k = 0; k < nk; k++) s += A[k * lda + i] * W[k * nj + j0 + jj];
                    C[(j0 + jj) * ldc + i] -= s;
                }
        }
    }
}

/* blocked right-looking LDL' of the first w columns of the f x f front F (column-major, lower triangle), then the Schur
 * complement on the trailing (f - w) x (f - w) block.  Two levels of blocking: panels of NP columns are factored by
 * inner blocks of NB (their updates stay inside the panel), then ONE update of everything to the right with all NP
 * columns -- 2 NP flops per 16 bytes of the trailing matrix; with NB alone a large front is bound by memory.
 * par: thread the trailing updates.  Returns regularised pivots. */
static i64 front_factor(double *F, i64 f, i64 w, const int8_t *sg, double eps, double delta, double *d, double *dinv, int par,
                        int *bad) {
    enum { NB = 32, NP = 192, CH = 64 };
    i64 nreg = 0;
    for (i64 K0 = 0; K0 < w; K0 += NP) {
        const i64 kw = w - K0 < NP ? w - K0 : NP, Kend = K0 + kw;
        for (i64 k0 = K0; k0 < Kend; k0 += NB) {
            const i64 kb = Kend - k0 < NB ? Kend - k0 : NB;
            /* the block's columns, unblocked: column k scaled, rank-1 update of the block's later columns */
            for (i64 k = k0; k < k0 + kb; k++) {
                double piv = F[k * f + k];
                if (piv * sg[k] < eps) {
                    piv = delta * sg[k];
                    nreg++;
                }
                if (piv == 0.0 || !isfinite(piv)) *bad = 1;
                d[k] = piv;
                dinv[k] = 1.0 / piv;
                double *ck = F + k * f;
                for (i64 j = k + 1; j < k0 + kb; j++) {
                    const double ljk = ck[j] * dinv[k]; /* l(j,k); ck[j] still unscaled */
                    double *cj = F + j * f;
                    for (i64 i = j; i < f; i++) cj[i] -= ck[i] * ljk;
                }
                for (i64 i = k + 1; i < f; i++) ck[i] *= dinv[k];
            }
            /* the panel's columns right of the block */
            const i64 jt = k0 + kb, nt = Kend - jt;
            if (nt > 0) {
                double Wl[NB * NP];
                for (i64 k = 0; k < kb; k++)
                    for (i64 j = 0; j < nt; j++) Wl[k * nt + j] = F[(k0 + k) * f + jt + j] * d[k0 + k];
                gemm_sub(F + jt * f, f, F + k0 * f, f, Wl, nt, kb, jt, f);
            }
        }
        /* everything right of the panel: C[i, j] -= sum_k L[i,k] d_k L[j,k], k over the panel;  W[k][j] = d_k L[j,k],
         * in column chunks so that W (kw x CH) stays in cache and threads get independent column ranges */
        const i64 nt = f - Kend;
        if (nt <= 0) continue;
        const i64 nch = (nt + CH - 1) / CH;
#pragma omp parallel for schedule(dynamic, 1) if (par && nch >= 4)
        for (i64 c = 0; c < nch; c++) {
            const i64 j0 = Kend + c * CH, jn = (j0 + CH <= f ? CH : f - j0);
            double Wl[NP * CH];
            for (i64 k = 0; k < kw; k++)
                for (i64 j = 0; j < jn; j++) Wl[k * jn + j] = F[(K0 + k) * f + j0 + j] * d[K0 + k];
            gemm_sub(F + j0 * f, f, F + K0 * f, f, Wl, jn, kw, j0, f);
        }
    }
    return nreg;
}

/* numeric factorisation.  Ax: values of the caller's upper CSC (original order, regularised as the caller wants);
 * signs: per ORIGINAL index.  Returns 0 on success (1: a zero / non-finite pivot); *nreg = regularised pivots. */
int sn_factor(sn_t *S, const double *Ax, const int8_t *signs, double eps, double delta, i64 *nreg_out) {
    int bad = 0;
    i64 nreg = 0;
    omp_set_num_threads(S->nthreads);
    int8_t *sgp = (int8_t *)xmalloc((size_t)S->n);
    for (i64 k = 0; k < S->n; k++) sgp[k] = signs[S->perm[k]];
    for (i64 l = 0; l < S->nlev; l++) {
        const i64 cnt = S->lev_ptr[l + 1] - S->lev_ptr[l];
        const int outer = cnt >= S->nthreads; /* enough supernodes at this height: one per thread, serial kernels */
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : nreg) reduction(| : bad) if (outer)
        for (i64 u = S->lev_ptr[l]; u < S->lev_ptr[l + 1]; u++) {
            const i64 s = S->lev_idx[u], c0 = S->sn_first[s], w = S->sn_first[s + 1] - c0;
            const i64 r = S->sn_rptr[s + 1] - S->sn_rptr[s], f = w + r;
            const i64 *rows = S->sn_rows + S->sn_rptr[s];
            double *F = (double *)xcalloc((size_t)f * (size_t)f, sizeof(double));
            /* relative position of a global row in this front (columns first, then rows): rows ascending -> binary search */
            /* K's entries of the supernode's columns */
            for (i64 j = c0; j < c0 + w; j++)
                for (i64 q = S->Lp_[j]; q < S->Lp_[j + 1]; q++) {
                    const i64 i = S->Li_[q];
                    i64 ri;
                    if (i < c0 + w) ri = i - c0;
                    else {
                        i64 lo = 0, hi = r - 1;
                        while (lo < hi) {
                            const i64 mid = (lo + hi) >> 1;
                            if (rows[mid] < i) lo = mid + 1;
                            else hi = mid;
                        }
                        ri = w + lo;
                    }
                    F[(j - c0) * f + ri] += Ax[S->Lsrc[q]];
                }
            /* children: extend-add their update matrices (they sit at lower heights and are finished) */
            for (i64 cq = S->ch_ptr[s]; cq < S->ch_ptr[s + 1]; cq++) {
                const i64 c = S->ch_idx[cq];
                const i64 rc = S->sn_rptr[c + 1] - S->sn_rptr[c];
                const i64 *crows = S->sn_rows + S->sn_rptr[c];
                double *U = S->upd[c];
                if (!U) continue;
                i64 *rel = (i64 *)xmalloc((size_t)(rc ? rc : 1) * sizeof(i64));
                i64 p = 0;
                for (i64 t = 0; t < rc; t++) {
                    const i64 i = crows[t];
                    if (i < c0 + w) rel[t] = i - c0;
                    else {
                        while (rows[p] < i) p++;
                        rel[t] = w + p;
                    }
                }
                for (i64 b = 0; b < rc; b++) {
                    double *dst = F + rel[b] * f;
                    const double *src = U + b * rc;
                    for (i64 a = b; a < rc; a++) dst[rel[a]] += src[a];
                }
                free(rel);
                free(U);
                S->upd[c] = NULL;
            }
            int mybad = 0;
            nreg += front_factor(F, f, w, sgp + c0, eps, delta, S->D + c0, S->Dinv + c0, !outer, &mybad);
            bad |= mybad;
            /* keep the panel (f x w) and hand the Schur complement (r x r, lower) to the parent */
            double *P = S->Lx + S->xoff[s];
            for (i64 k = 0; k < w; k++) memcpy(P + k * f, F + k * f, (size_t)f * sizeof(double));
            if (r > 0 && S->sn_parent[s] != -1) {
                double *U = (double *)xmalloc((size_t)r * (size_t)r * sizeof(double));
                for (i64 b = 0; b < r; b++) memcpy(U + b * r + b, F + (w + b) * f + w + b, (size_t)(r - b) * sizeof(double));
                S->upd[s] = U;
            }
            free(F);
        }
    }
    free(sgp);
    if (nreg_out) *nreg_out = nreg;
    return bad;
}

/* x <- K^-1 x in the ORIGINAL numbering (permutes in and out) */
void sn_solve(sn_t *S, double *x_orig) {
    const i64 n = S->n;
    omp_set_num_threads(S->nthreads);
    double *x = (double *)xmalloc((size_t)n * sizeof(double));
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < n; k++) x[k] = x_orig[S->perm[k]];
    /* forward: leaves first; a supernode's pushes to its rows may collide with its siblings' */
    for (i64 l = 0; l < S->nlev; l++) {
#pragma omp parallel for schedule(dynamic, 1)
        for (i64 u = S->lev_ptr[l]; u < S->lev_ptr[l + 1]; u++) {
            const i64 s = S->lev_idx[u], c0 = S->sn_first[s], w = S->sn_first[s + 1] - c0;
            const i64 r = S->sn_rptr[s + 1] - S->sn_rptr[s], f = w + r;
            const i64 *rows = S->sn_rows + S->sn_rptr[s];
            const double *P = S->Lx + S->xoff[s];
            double *xs = x + c0;
            for (i64 k = 0; k < w; k++) {
                const double xk = xs[k];
                const double *ck = P + k * f;
                for (i64 i = k + 1; i < w; i++) xs[i] -= ck[i] * xk;
            }
            if (r > 0) {
                double *acc = (double *)xcalloc((size_t)r, sizeof(double));
                for (i64 k = 0; k < w; k++) {
                    const double xk = xs[k];
                    const double *ck = P + k * f + w;
                    for (i64 i = 0; i < r; i++) acc[i] += ck[i] * xk;
                }
                for (i64 i = 0; i < r; i++) {
#pragma omp atomic
                    x[rows[i]] -= acc[i];
                }
                free(acc);
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < n; k++) x[k] *= S->Dinv[k];
    for (i64 l = S->nlev - 1; l >= 0; l--) {
#pragma omp parallel for schedule(dynamic, 1)
        for (i64 u = S->lev_ptr[l]; u < S->lev_ptr[l + 1]; u++) {
            const i64 s = S->lev_idx[u], c0 = S->sn_first[s], w = S->sn_first[s + 1] - c0;
            const i64 r = S->sn_rptr[s + 1] - S->sn_rptr[s], f = w + r;
            const i64 *rows = S->sn_rows + S->sn_rptr[s];
            const double *P = S->Lx + S->xoff[s];
            double *xs = x + c0;
            double *xr = (double *)xmalloc((size_t)(r ? r : 1) * sizeof(double));
            for (i64 i = 0; i < r; i++) xr[i] = x[rows[i]];
            for (i64 k = w - 1; k >= 0; k--) {
                const double *ck = P + k * f;
                double sum = 0.0;
                for (i64 i = 0; i < r; i++) sum += ck[w + i] * xr[i];
                for (i64 i = k + 1; i < w; i++) sum += ck[i] * xs[i];
                xs[k] -= sum;
            }
            free(xr);
        }
    }
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < n; k++) x_orig[S->perm[k]] = x[k];
    free(x);
}

/* y = b - K x with K given by its upper CSC (original numbering) and the values Ax: gather-only, the upper triangle by
 * columns (entries (i, c), i <= c, serve y_c) and by rows (entries (a, c), c > a, serve y_a) */
void sn_residual(sn_t *S, const i64 *Ap, const i64 *Ai, const double *Ax, const double *x, const double *b, double *y) {
    const i64 n = S->n;
    omp_set_num_threads(S->nthreads);
#pragma omp parallel for schedule(dynamic, 256)
    for (i64 c = 0; c < n; c++) {
        double s = 0.0;
        for (i64 p = Ap[c]; p < Ap[c + 1]; p++) s += Ax[p] * x[Ai[p]];
        for (i64 t = S->Tp[c]; t < S->Tp[c + 1]; t++) s += Ax[S->Tsrc[t]] * x[S->Ti[t]];
        y[c] = b[c] - s;
    }
}
