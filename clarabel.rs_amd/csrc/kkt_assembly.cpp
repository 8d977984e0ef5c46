// kkt_assembly.cpp -- builds the upper-triangular quasidefinite KKT matrix
//
//        [ P   A'   0  ]      n  primal columns
//    K = [ .  -Hs   E  ]      m  cone rows
//        [ .   .    De ]      p  sparse-cone expansion columns
//
// in EXACTLY the CSC layout the reference produces (kkt_assembly.rs:20-183,
// datamaps.rs:112-405, csc/utils.rs), together with the
// LDLDataMap index vectors, so that indices exchanged over the C ABI mean the
// same thing on both sides.  The construction here is column-direct (each
// column's length is known in closed form, then every column is written front
// to back) rather than the reference's count/fill/backshift passes.
//
// Layout contract reproduced:
//  * columns 0..n-1: triu(P) column, then a structural-zero diagonal if P has
//    none there (the test is "last entry of the column is not the diagonal",
//    utils.rs:45-57) => the diagonal is the last entry of every column;
//  * column n+r: entries of row r of A (ascending primal index), then the cone
//    block: one diagonal entry for diagonal-Hs cones, or the packed-triu column
//    segment rows start..=col for dense cones (utils.rs:183-200);
//  * per sparse-expandable cone (SOC with dim > 4, GenPow) extra columns:
//    SOC -> [v, u] (datamaps.rs:186-189), GenPow -> [q, r, p] (:306-309), each
//    holding the cone's rows followed by its own diagonal entry;
//  * map.Hsblocks enumerates the Hs entries cone by cone, diag_full/diagP the
//    last entry of each column; dsigns = +1 (n), -1 (m), then [-1,+1] per SOC
//    / [-1,-1,+1] per GenPow (directldlkktsolver.rs:392-405).
#include "host.hpp"

namespace chip {

static i64 tri(i64 k) { return k * (k + 1) / 2; }

int build_cone_specs(i64 ncones, const i32 *tags, const i64 *dims, const i64 *dims2,
                     std::vector<ConeSpec> &out, i64 &m, i64 &p, i64 &nHs) {
    out.clear();
    i64 start = 0, bstart = 0, pcol = 0, sidx = 0;
    for (i64 i = 0; i < ncones; i++) {
        ConeSpec c{};
        c.tag = tags[i];
        c.dim = dims[i];
        c.dim2 = dims2 ? dims2[i] : 0;
        c.pdim = 0;
        c.sparse = false;
        switch (c.tag) {
        case 0: // Zero
        case 1: // Nonnegative
            c.numel = c.dim;
            c.hs_diag = true;
            break;
        case 2: // SOC: sparse expansion above SOC_NO_EXPANSION_MAX_SIZE = 4 (socone.rs:46)
            if (c.dim < 2) {
                set_error("second-order cone needs dim >= 2");
                return -9;
            }
            c.numel = c.dim;
            c.sparse = c.dim > 4;
            c.hs_diag = c.sparse;
            c.pdim = c.sparse ? 2 : 0;
            break;
        case 3: // Exp
        case 4: // Pow
            c.numel = 3;
            c.hs_diag = false;
            break;
        case 5: // GenPow
            c.numel = c.dim + c.dim2;
            c.hs_diag = true;
            c.sparse = true;
            c.pdim = 3;
            break;
        case 6: // PSD triangle
            c.numel = tri(c.dim);
            c.hs_diag = false;
            break;
        default:
            set_error("unknown cone tag");
            return -9;
        }
        if (c.numel < 0) return -9;
        c.start = start;
        c.block_start = bstart;
        c.block_len = c.hs_diag ? c.numel : tri(c.numel);
        c.sparse_col = c.sparse ? pcol : -1;
        c.sparse_idx = c.sparse ? sidx : -1;
        start += c.numel;
        bstart += c.block_len;
        if (c.sparse) {
            pcol += c.pdim;
            sidx++;
        }
        out.push_back(c);
    }
    m = start;
    nHs = bstart;
    p = pcol;
    return 0;
}

int assemble_kkt_triu(i64 n, i64 m, const i64 *Pp, const i64 *Pi, const double *Px, const i64 *Ap,
                      const i64 *Ai, const double *Ax, KktLayout &K) {
    const i64 p = K.p, N = n + m + p;
    K.n = n;
    K.m = m;
    K.N = N;
    const i64 nnzP = Pp[n], nnzA = Ap[n];
    // ---- column lengths -----------------------------------------------------
    std::vector<i64> len((size_t)N + 1, 0);
    std::vector<char> need_diag((size_t)n, 0);
    for (i64 j = 0; j < n; j++) {
        const bool has = Pp[j] < Pp[j + 1] && Pi[Pp[j + 1] - 1] == j;
        need_diag[j] = !has;
        len[j] = (Pp[j + 1] - Pp[j]) + (has ? 0 : 1);
        for (i64 q = Pp[j]; q < Pp[j + 1]; q++)
            if (Pi[q] > j || Pi[q] < 0) {
                set_error("P must be upper triangular");
                return -3;
            }
    }
    for (i64 q = 0; q < nnzA; q++) {
        if (Ai[q] < 0 || Ai[q] >= m) {
            set_error("A row index out of range");
            return -1;
        }
        len[n + Ai[q]]++;
    }
    for (const ConeSpec &c : K.cones) {
        for (i64 k = 0; k < c.numel; k++) len[n + c.start + k] += c.hs_diag ? 1 : (k + 1);
        if (c.sparse) {
            const i64 col = n + m + c.sparse_col;
            if (c.tag == 2) {
                len[col] += c.numel + 1;
                len[col + 1] += c.numel + 1;
            } else {
                len[col] += c.dim + 1;
                len[col + 1] += c.dim2 + 1;
                len[col + 2] += c.numel + 1;
            }
        }
    }
    K.colptr.assign((size_t)N + 1, 0);
    for (i64 j = 0; j < N; j++) K.colptr[j + 1] = K.colptr[j] + len[j];
    K.nnz = K.colptr[N];
    K.rowval.resize((size_t)K.nnz + 1); // (no fill: every slot below nnz is written exactly once, checked at the end)
    K.nzval.resize((size_t)K.nnz + 1);
    K.rowval[(size_t)K.nnz] = 0;
    K.nzval[(size_t)K.nnz] = 0.0;
    K.mapP.assign((size_t)nnzP + 1, 0);
    K.mapA.assign((size_t)nnzA + 1, 0);
    K.mapHs.resize((size_t)K.nHs + 1);
    K.mapHs[(size_t)K.nHs] = 0;
    K.diagP.assign((size_t)n + 1, 0);
    K.diag_full.assign((size_t)N + 1, 0);
    std::vector<i64> wr(K.colptr.begin(), K.colptr.end() - 1); // write cursors
    auto put = [&](i64 col, i64 row, double v) {
        const i64 d = wr[col]++;
        K.rowval[d] = row;
        K.nzval[d] = v;
        return d;
    };
    // ---- P block + missing diagonals ---------------------------------------
    for (i64 j = 0; j < n; j++) {
        for (i64 q = Pp[j]; q < Pp[j + 1]; q++) K.mapP[q] = put(j, Pi[q], Px[q]);
        if (need_diag[j]) put(j, j, 0.0);
    }
    // ---- A' block: traversing A column-major appends ascending primal
    //      indices to each cone-row column ------------------------------------
    for (i64 j = 0; j < n; j++)
        for (i64 q = Ap[j]; q < Ap[j + 1]; q++) K.mapA[q] = put(n + Ai[q], j, Ax[q]);
    // ---- cone blocks, then expansion columns --------------------------------
    const i64 nsparse = [&] {
        i64 c = 0;
        for (const ConeSpec &s : K.cones) c += s.sparse;
        return c;
    }();
    K.sp_ptr.assign((size_t)nsparse + 1, 0);
    K.sp_q_ptr.assign((size_t)nsparse + 1, 0);
    K.sp_r_ptr.assign((size_t)nsparse + 1, 0);
    K.sp_D.assign((size_t)(3 * nsparse) + 1, 0);
    for (const ConeSpec &c : K.cones)
        if (c.sparse) {
            K.sp_ptr[c.sparse_idx + 1] = c.numel;
            K.sp_q_ptr[c.sparse_idx + 1] = c.tag == 5 ? c.dim : 0;
            K.sp_r_ptr[c.sparse_idx + 1] = c.tag == 5 ? c.dim2 : 0;
        }
    for (i64 s = 0; s < nsparse; s++) {
        K.sp_ptr[s + 1] += K.sp_ptr[s];
        K.sp_q_ptr[s + 1] += K.sp_q_ptr[s];
        K.sp_r_ptr[s + 1] += K.sp_r_ptr[s];
    }
    K.sp_u.assign((size_t)K.sp_ptr[nsparse] + 1, 0);
    K.sp_v.assign((size_t)K.sp_ptr[nsparse] + 1, 0);
    K.sp_q.assign((size_t)K.sp_q_ptr[nsparse] + 1, 0);
    K.sp_r.assign((size_t)K.sp_r_ptr[nsparse] + 1, 0);
    K.dsigns.assign((size_t)N + 1, 1);
    for (i64 i = n; i < n + m; i++) K.dsigns[i] = -1;

    // the cones' blocks: every cone writes columns of its own only (its rows of the (2, 2) block, its expansion columns),
    // so the cones are shared out among host threads -- config 5's 200 dense PSD blocks are 1.6e8 of K's entries
    const int TC = K.nHs >= (i64)1 << 22 ? host_threads() : 1;
    run_threads(TC, [&](int tc, int TT) {
    for (size_t ci = (size_t)tc; ci < K.cones.size(); ci += (size_t)TT) {
        const ConeSpec &c = K.cones[ci];
        const i64 row0 = n + c.start;
        i64 h = c.block_start;
        if (c.hs_diag) {
            for (i64 k = 0; k < c.numel; k++) K.mapHs[h++] = put(row0 + k, row0 + k, 0.0);
        } else {
            for (i64 k = 0; k < c.numel; k++)
                for (i64 r = 0; r <= k; r++) K.mapHs[h++] = put(row0 + k, row0 + r, 0.0);
        }
        if (!c.sparse) continue;
        const i64 col = n + m + c.sparse_col, s = c.sparse_idx;
        if (c.tag == 2) {
            // first extra column carries v, the second u; signs (-1, +1)
            for (i64 k = 0; k < c.numel; k++) K.sp_v[K.sp_ptr[s] + k] = put(col, row0 + k, 0.0);
            for (i64 k = 0; k < c.numel; k++) K.sp_u[K.sp_ptr[s] + k] = put(col + 1, row0 + k, 0.0);
            K.sp_D[3 * s + 0] = put(col, col, 0.0);
            K.sp_D[3 * s + 1] = put(col + 1, col + 1, 0.0);
            K.dsigns[col] = -1;
            K.dsigns[col + 1] = 1;
        } else {
            // GenPow: q over the first dim1 rows, r over the next dim2 rows, p over all
            for (i64 k = 0; k < c.dim; k++) K.sp_q[K.sp_q_ptr[s] + k] = put(col, row0 + k, 0.0);
            for (i64 k = 0; k < c.dim2; k++)
                K.sp_r[K.sp_r_ptr[s] + k] = put(col + 1, row0 + c.dim + k, 0.0);
            for (i64 k = 0; k < c.numel; k++) K.sp_u[K.sp_ptr[s] + k] = put(col + 2, row0 + k, 0.0);
            K.sp_D[3 * s + 0] = put(col, col, 0.0);
            K.sp_D[3 * s + 1] = put(col + 1, col + 1, 0.0);
            K.sp_D[3 * s + 2] = put(col + 2, col + 2, 0.0);
            K.dsigns[col] = -1;
            K.dsigns[col + 1] = -1;
            K.dsigns[col + 2] = 1;
        }
    }
    });
    for (i64 j = 0; j < N; j++) {
        if (wr[j] != K.colptr[j + 1]) {
            set_error("internal: KKT column fill mismatch");
            return -9;
        }
        K.diag_full[j] = K.colptr[j + 1] - 1;
        if (j < n) K.diagP[j] = K.colptr[j + 1] - 1;
    }
    return 0;
}

} // namespace chip
