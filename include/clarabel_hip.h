/*
 * clarabel_hip.h -- C ABI of the MI355X-native KKT linear-system backend for
 * Clarabel (reference: oxfordcontrol/Clarabel.rs v0.11.1; citations are
 * relative to /root/reference/src).
 *
 * Two levels, both mirrored 1:1 from the reference's own trait surface:
 *
 *  L1  chip_ldl_*   ==  trait DirectLDLSolver<f64>
 *                       (solver/core/kktsolvers/direct/quasidef/mod.rs:14-26)
 *                       construction signature ldlsolvers/config.rs:21-22
 *                       reference behaviour   ldlsolvers/qdldl.rs:18-107
 *                       This is the strict drop-in: a Rust `HipDirectLDLSolver`
 *                        forwards each trait method to one call.
 *
 *  L2  chip_kkt_*   ==  trait KKTSolver<f64> (solver/core/kktsolvers/mod.rs:7-18)
 *                       as implemented by DirectLDLKKTSolver
 *                       (quasidef/directldlkktsolver.rs:18-405): KKT assembly,
 *                       cone Hs blocks fused into the value update, static
 *                       regularisation, refactor, solve + iterative refinement,
 *                       all device resident.
 *
 * Conventions: plain pointers + sizes, no C++/torch types.  Index arrays are
 * uint64_t/int64_t because the reference's CscMatrix uses usize
 * (algebra/csc/core.rs:45-60); values are double.  Functions return
 * CHIP_OK (0) or a negative chip_status unless documented as returning the
 * reference's `bool` (1 = success, 0 = numerical failure).  Pointers named
 * *_dev are device (HBM) pointers on the engine's device; all others are host.
 * A handle is owned by the caller, single-threaded use, movable across threads
 * (the reference requires Send+Sync: directldlkktsolver.rs:13-16).
 */
#ifndef CLARABEL_HIP_H
#define CLARABEL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    CHIP_OK = 0,
    CHIP_ERR_DIM = -1,            /* QDLDLError::IncompatibleDimension  qdldl.rs:13 */
    CHIP_ERR_EMPTY_COLUMN = -2,   /* QDLDLError::EmptyColumn            qdldl.rs:16 */
    CHIP_ERR_NOT_TRIU = -3,       /* QDLDLError::NotUpperTriangular     qdldl.rs:19 */
    CHIP_ERR_ZERO_PIVOT = -4,     /* QDLDLError::ZeroPivot              qdldl.rs:22 */
    CHIP_ERR_BAD_PERM = -5,       /* QDLDLError::InvalidPermutation     qdldl.rs:25 */
    CHIP_ERR_NOT_FACTORED = -6,   /* solve() before the first refactor(): the reference asserts, qdldl.rs:118 */
    CHIP_ERR_NO_DEVICE = -7,      /* HIP runtime/device unavailable (the product has NO CPU fallback) */
    CHIP_ERR_HIP = -8,            /* a HIP call failed; see chip_last_error() */
    CHIP_ERR_ARG = -9,
    CHIP_ERR_UNSUPPORTED = -10
} chip_status;

/* SupportedConeT tags (solver/core/cones/supportedcone.rs) */
typedef enum {
    CHIP_CONE_ZERO = 0,
    CHIP_CONE_NONNEGATIVE = 1,
    CHIP_CONE_SECONDORDER = 2,
    CHIP_CONE_EXPONENTIAL = 3,
    CHIP_CONE_POWER = 4,
    CHIP_CONE_GENPOWER = 5,
    CHIP_CONE_PSDTRIANGLE = 6
} chip_cone_tag;

/* The CoreSettings fields the path consumes
 * (solver/implementations/default/settings.rs:126-181) + engine knobs. */
typedef struct {
    int32_t static_regularization_enable;        /* true   */
    double static_regularization_constant;       /* 1e-8   */
    double static_regularization_proportional;   /* eps^2  */
    int32_t dynamic_regularization_enable;       /* true; NB the qdldl adapter ignores it
                                                    (ldlsolvers/qdldl.rs:38), and so do we  */
    double dynamic_regularization_eps;           /* 1e-13  */
    double dynamic_regularization_delta;         /* 2e-7   */
    int32_t iterative_refinement_enable;         /* true   */
    double iterative_refinement_reltol;          /* 1e-13  */
    double iterative_refinement_abstol;          /* 1e-12  */
    int32_t iterative_refinement_max_iter;       /* 10     */
    double iterative_refinement_stop_ratio;      /* 5.0    */
    /* engine knobs (no reference counterpart) */
    int32_t device;            /* HIP device ordinal, -1 = current device,
                                  CHIP_DEVICE_HOST_ONLY = symbolic analysis only (no GPU touched;
                                  every numeric call then returns CHIP_ERR_NO_DEVICE) -- the
                                  analogue of the reference's logical factorisation, qdldl.rs:40-42 */
    double amd_dense_scale;    /* 1.5, ldlsolvers/qdldl.rs:41 */
    int32_t use_graph;         /* replay the launch sequence of an LDL' solve as a hipGraph (captured once per
                                  vector pair); pays off for tall elimination trees (hundreds of launches) */
    int32_t reserved0;
    /* line search of the nonsymmetric cones (settings.rs:96-104), used by chip_kkt_step_length_dev */
    double linesearch_backtrack_step;            /* 0.8    */
    double min_terminate_step_length;            /* 1e-4   */
    int32_t reserved[2];
} chip_settings;

/* LinearSolverInfo (solver/core/kktsolvers/mod.rs:27-38) + factor statistics
 * (qdldl.rs:104-112, 203-210). */
typedef struct {
    char name[16];             /* "hip" */
    int64_t threads;           /* #GPUs working on the problem: 1, or the world size after chip_kkt_attach_comm */
    int32_t direct;            /* 1 */
    int64_t nnzA;              /* nnz(triu K) */
    int64_t nnzL;              /* nnz(L), strictly lower */
    int64_t positive_inertia;  /* valid after refactor */
    int64_t regularize_count;  /* dynamic regularisation hits of the last refactor */
    int64_t n;                 /* KKT dimension N */
    int64_t n_levels;          /* depth of the elimination tree (number of dependent phases) */
    double amd_lnz, amd_ndiv, amd_nmultsubs_ldl; /* amd::Info, used by ldlsolvers/auto.rs:69-77 */
    int32_t last_ir_iterations;/* refinement rounds of the last chip_kkt_solve */
    double last_regularizer;   /* static eps of the last update (directldlkktsolver.rs:249) */
} chip_info;

#define CHIP_DEVICE_HOST_ONLY (-2)

typedef struct chip_ldl chip_ldl;
typedef struct chip_kkt chip_kkt;

void chip_settings_default(chip_settings *s);
const char *chip_last_error(void);
/* number of visible HIP devices (0 when there is no GPU / runtime) */
int32_t chip_device_count(void);
/* The switching rule of ldl_auto_select (ldlsolvers/auto.rs:62-87) on the statistics of an AMD
 * ordering (chip_amd_order's info3, or chip_info.amd_*): (n_div + n_mult_subs_ldl) / lnz < 40 selects
 * the simplicial engine (returns 0, "qdldl"), otherwise the supernodal one (returns 1, "faer").
 * This library has a single engine; the rule is exported so that a host-side "auto" setting can be
 * kept without the amd crate, and as the census behind DESIGN.md section 9. */
int32_t chip_auto_select(double lnz, double n_div, double n_mult_subs_ldl);

/* ---- host-side symbolic utilities ------------------------------------------
 * Approximate-minimum-degree ordering of the symmetric matrix whose upper
 * triangle is given in CSC form.  Replaces the un-vendored crate `amd 0.2.2`
 * at its three call sites (qdldl.rs:905-917, ldlsolvers/mod.rs:15-23,
 * auto.rs:69): dense threshold = 10*dense_scale*sqrt(n).  perm[k] = the
 * original index eliminated k-th; iperm its inverse.  info3 (may be NULL)
 * receives {lnz, ndiv, nmultsubs_ldl}. */
int32_t chip_amd_order(int64_t n, const uint64_t *colptr, const uint64_t *rowval,
                       double dense_scale, uint64_t *perm, uint64_t *iperm, double *info3);

/* ---- L1: DirectLDLSolver ----------------------------------------------------
 * ctor  == LDLConstructor (config.rs:21-22):
 *          fn(&CscMatrix<T> [triu KKT], &[i8] [Dsigns], &CoreSettings<T>, Option<Vec<usize>> [perm])
 * required_matrix_shape() == Triu (quasidef/mod.rs:14-16). */
int32_t chip_ldl_create(chip_ldl **out, int64_t n, const uint64_t *colptr,
                        const uint64_t *rowval, const double *nzval, const int8_t *dsigns,
                        const uint64_t *perm_or_null, const chip_settings *settings);
void chip_ldl_destroy(chip_ldl *h);
/* update_values(&mut self, index:&[usize], values:&[T])   mod.rs:20, qdldl.rs:142-149 */
int32_t chip_ldl_update_values(chip_ldl *h, const uint64_t *index, const double *values, int64_t k);
/* scale_values(&mut self, index:&[usize], scale:T)        mod.rs:21, qdldl.rs:153-160 */
int32_t chip_ldl_scale_values(chip_ldl *h, const uint64_t *index, double scale, int64_t k);
/* offset_values(&mut self, index, offset, signs:&[i8])    mod.rs:22-23, qdldl.rs:166-183 */
int32_t chip_ldl_offset_values(chip_ldl *h, const uint64_t *index, double offset,
                               const int8_t *signs, int64_t k);
/* Pardiso-style wholesale value hand-over (pardiso.rs:277-322): replaces the
 * engine's copy of K.nzval by the caller's full array, so the 5 tiny per-SOC
 * update calls (datamaps.rs:212-219) can be no-ops on the Rust side. */
int32_t chip_ldl_set_values(chip_ldl *h, const double *kkt_nzval);
/* refactor(&mut self, kkt) -> bool   mod.rs:25, ldlsolvers/qdldl.rs:98-106.
 * returns 1 = all Dinv finite, 0 = numerical failure, <0 = chip_status. */
int32_t chip_ldl_refactor(chip_ldl *h);
/* solve(&mut self, kkt, x:&mut[T], b:&mut[T])   mod.rs:24, ldlsolvers/qdldl.rs:91-96.
 * b is left untouched. */
int32_t chip_ldl_solve(chip_ldl *h, double *x, const double *b);
/* ---- fast path of the strict drop-in (round 5; opt-in, the calls above keep their meaning) ----------------------
 * What DirectLDLKKTSolver changes in K per interior-point iteration are the entries of a FEW FIXED index vectors of its
 * LDLDataMap -- Hsblocks (directldlkktsolver.rs:143), diag_full with Dsigns (:245, :255-261: the static regulariser on and
 * off), the sparse cones' u / v / D (datamaps.rs:213-219) -- and its iterative refinement (:266-321) needs nothing from
 * the host but b.  Through the calls above every index vector crosses PCIe with every update (or, with set_values, all of
 * K.nzval with every refactor), and b / x of every LDL' solve of every refinement round: 352 MB per iteration of
 * BASELINE config 3.  The fast path:
 *   chip_ldl_register_index   once per index vector: the set lives on the device (as positions in the engine's value
 *                             order, with the signs of offset_values if given); returns its id
 *   chip_ldl_update_values_id / _scale_values_id / _offset_values_id
 *                             = update_values / scale_values / offset_values on a registered set: 8 bytes per changed entry
 *                             (update) or nothing but a scalar (scale, offset) cross the boundary.  From the first such call
 *                             the DEVICE copy is the caller's K.nzval (refactor uploads nothing); plain update / scale /
 *                             offset calls keep working (they then ship their index vector each time);
 *                             chip_ldl_set_values makes the host side current again.
 *   chip_ldl_solve_refined    solve + the refinement loop of directldlkktsolver.rs:266-321, decisions as the reference's,
 *                             with the handle's CURRENT values as K (i.e. after the caller restored the diagonal):
 *                             one host-to-device copy of b, one device-to-host copy of x.  ir_settings: the
 *                             iterative_refinement_* fields are read (NULL: the handle's settings).  Returns 1 ok /
 *                             0 numerical failure (non-finite residual) / < 0 chip_status; *iterations = rounds taken.
 *   chip_ldl_pin_buffer       page-locks a caller buffer that is handed to the calls above again and again (the solver's
 *                             workspace vectors), so that its transfers are direct DMA; unpinned when the handle goes.
 *                             The buffer must outlive the handle. */
int32_t chip_ldl_register_index(chip_ldl *h, const uint64_t *index, int64_t k, const int8_t *signs_or_null, int32_t *id_out);
int32_t chip_ldl_update_values_id(chip_ldl *h, int32_t id, const double *values);
int32_t chip_ldl_scale_values_id(chip_ldl *h, int32_t id, double scale);
int32_t chip_ldl_offset_values_id(chip_ldl *h, int32_t id, double offset);
int32_t chip_ldl_solve_refined(chip_ldl *h, double *x, const double *b, const chip_settings *ir_settings, int32_t *iterations);
int32_t chip_ldl_pin_buffer(chip_ldl *h, void *ptr, uint64_t bytes);
/* device-resident variant: x_dev / b_dev are HBM pointers (may alias). */
int32_t chip_ldl_solve_dev(chip_ldl *h, double *x_dev, const double *b_dev);
/* linear_solver_info()  kktsolvers/mod.rs:20-24 */
int32_t chip_ldl_info(const chip_ldl *h, chip_info *info);
/* final elimination order actually used (a topological re-sort of the AMD /
 * user permutation by elimination-tree level; same fill, same etree) */
int32_t chip_ldl_get_perm(const chip_ldl *h, uint64_t *perm);
/* symbolic results in the engine's final numbering (host copies; any pointer may
 * be NULL): etree[N] (UINT64_MAX = root, cf. qdldl.rs:426), Lp[N+1], Li[nnzL]
 * (ascending rows per column), level[N] (elimination-tree level of each node,
 * leaves = 0).  Also valid on CHIP_DEVICE_HOST_ONLY handles. */
int32_t chip_ldl_get_symbolic(const chip_ldl *h, uint64_t *etree, uint64_t *Lp, uint64_t *Li,
                              uint64_t *level);
/* test/diagnostic access to the factors (host copies): L as CSC with sorted
 * rows in the engine's numbering, D, Dinv.  Any pointer may be NULL. */
int32_t chip_ldl_get_factors(chip_ldl *h, uint64_t *Lp, uint64_t *Li, double *Lx, double *D,
                             double *Dinv);

/* ---- L2: KKTSolver (DirectLDLKKTSolver) --------------------------------------
 * new(P, A, cones, m, n, settings)  directldlkktsolver.rs:60-118.
 * P: n x n triu CSC, A: m x n CSC, both canonically sorted (csc/core.rs:322-338).
 * cone i: tags[i], dims[i] (Zero/NN/SOC: numel; PSD: matrix side; GenPow: dim1),
 * dims2[i] (GenPow dim2, else 0), alphas[i] (PowerConeT exponent, else ignored; NULL when
 * there is no power cone).  Exp/Pow have numel 3. */
int32_t chip_kkt_create(chip_kkt **out, int64_t n, int64_t m, const uint64_t *Pcolptr,
                        const uint64_t *Prowval, const double *Pnzval, const uint64_t *Acolptr,
                        const uint64_t *Arowval, const double *Anzval, int64_t ncones,
                        const int32_t *cone_tags, const int64_t *cone_dims,
                        const int64_t *cone_dims2, const double *cone_alphas_or_null,
                        const chip_settings *settings, const uint64_t *perm_or_null);
void chip_kkt_destroy(chip_kkt *h);
/* dimensions: out[0]=n out[1]=m out[2]=p out[3]=N out[4]=nnzK out[5]=nHsblocks
 * out[6]=NF (nodes inside subtree bundles; N-NF = "top" nodes) out[7]=nnzU (K entries whose
 * smaller index is a bundle node: what the bundle residual kernel streams) */
int32_t chip_kkt_dims(const chip_kkt *h, int64_t out[8]);
/* the assembled (unpermuted, triu) KKT matrix and the LDLDataMap index
 * vectors (datamaps.rs:350-362) -- host copies for tests / Rust-side mirrors. */
int32_t chip_kkt_get_matrix(const chip_kkt *h, uint64_t *colptr, uint64_t *rowval, double *nzval);
int32_t chip_kkt_get_map(const chip_kkt *h, uint64_t *mapP, uint64_t *mapA, uint64_t *mapHs,
                         uint64_t *diagP, uint64_t *diag_full, int8_t *dsigns);
/* CompositeCone::update_scaling(s, z, mu, scaling_strategy) (compositecone.rs:226-243) for
 * the cones held on the device: Nonnegative (nonnegativecone.rs:77-90), SecondOrder
 * (socone.rs:134-211), Exponential (expcone.rs:106-124) and Power (powcone.rs:99-117) with the
 * primal-dual / dual scalings of nonsymmetric_common.rs:53-143, PSDTriangle of any matrix side
 * (psdtrianglecone.rs:144-204: two Cholesky factors, SVD, R R', skron; the n x n work matrices live in LDS up
 * to side 64 and in HBM scratch beyond); Zero is a no-op.
 * strategy: 0 = ScalingStrategy::PrimalDual, 1 = ::Dual (core/solver.rs:77-80).
 * Returns the reference's bool.  s, z: m doubles (host / device variants; the _dev variant
 * defers the SOC interior check to the next chip_kkt_update so that it stays asynchronous). */
int32_t chip_kkt_update_scaling(chip_kkt *h, const double *s, const double *z, double mu,
                                int32_t strategy);
int32_t chip_kkt_update_scaling_dev(chip_kkt *h, const double *s_dev, const double *z_dev, double mu,
                                    int32_t strategy);
/* KKTSolver::update (directldlkktsolver.rs:134-158): Hs blocks (get_Hs fused,
 * negated), sparse-cone u/v/D columns, static regularisation, numeric refactor.
 * hsblocks_or_null: full Hsblocks vector (host), consulted ONLY for cone types whose scaling is not held on
 * the device.  Every SupportedConeT is held on the device now, so it may always be NULL; the parameter
 * is kept for callers that compute Hs themselves.  Returns the reference's bool. */
int32_t chip_kkt_update(chip_kkt *h, const double *hsblocks_or_null);
/* setrhs(rhsx, rhsz)   directldlkktsolver.rs:160-166.  The host variant copies the vectors.  The _dev
 * variant may BORROW the two device buffers until the NEXT chip_kkt_setrhs* (or chip_kkt_solve_full / the
 * handle's destruction): when the whole solve runs as one fused launch (systems made of subtree bundles and
 * at most a few dense top rows: configs 3 and 4) that launch reads and permutes them itself, in every
 * refinement round (no permuted copy of b is written: 24 N bytes less traffic per solve), and a further
 * chip_kkt_solve* on the same right-hand side (the reference keeps self.b, :168-175) reads them again.  They
 * must not be overwritten from another stream or from the host in between, and they must not overlap the
 * lhs buffers of those solves. */
int32_t chip_kkt_setrhs(chip_kkt *h, const double *rhsx, const double *rhsz);
int32_t chip_kkt_setrhs_dev(chip_kkt *h, const double *rhsx_dev, const double *rhsz_dev);
/* solve(lhsx, lhsz, settings) -> bool   directldlkktsolver.rs:168-189,
 * incl. iterative_refinement :266-321.  Either output may be NULL. */
int32_t chip_kkt_solve(chip_kkt *h, double *lhsx_or_null, double *lhsz_or_null);
int32_t chip_kkt_solve_dev(chip_kkt *h, double *lhsx_dev_or_null, double *lhsz_dev_or_null);
/* ---- asynchronous variants: enqueue on the handle's stream and return at once ------------------------
 * The reference's update() / solve() return their bool immediately because they run on the host.  On the
 * device the verdicts (all pivots finite, refinement residual finite, cones interior) are produced by the
 * kernels themselves; a device-resident driver enqueues one whole interior-point iteration -- update, then
 * the solves, chip_kkt_setrhs_dev before each -- and asks for the verdicts ONCE:
 *   chip_kkt_update_enqueue   = chip_kkt_update without the final device-to-host copy
 *   chip_kkt_solve_dev_enqueue = chip_kkt_solve_dev: for systems the fused launch covers (subtree bundles and
 *                               at most a few dense top rows) the refinement decisions of
 *                               directldlkktsolver.rs:266-321 are taken on the device and nothing is waited
 *                               for; other systems run the synchronous solve and queue its verdict
 *   chip_kkt_collect          synchronises the stream once; *update_ok = verdict of the last enqueued update
 *                               (1 if none), *nsolves = solves enqueued since the last collect (at most 16
 *                               may be pending), solves_ok[i] their verdicts in order: 0 = failed (the
 *                               reference's false), 1 = ok, 2 = ok BUT REPEATED AT COLLECT TIME: the fused launch of
 *                               that solve timed out (its workgroups were not all resident), so its lhs held garbage
 *                               until collect repeated it on the one-kernel-per-phase path.  Device work that was
 *                               enqueued behind such a solve and consumed its lhs -- an all-gather
 *                               (chip_kkt_allgather_step), a right-hand side computed from it -- used the garbage and
 *                               must be re-issued by the caller; the repeat itself read the right-hand side buffers
 *                               as they were at collect time.  Returns CHIP_OK or a negative chip_status. */
int32_t chip_kkt_update_enqueue(chip_kkt *h, const double *hsblocks_or_null);
/* chip_kkt_solve2_dev_enqueue (round 5): TWO INDEPENDENT solves of one interior-point iteration as one call -- the constant
 * right-hand side [-q; b] inside kktsystem.update (default/kktsystem.rs:108-125) and the affine direction
 * (core/solver.rs:351-361) do not depend on each other; only the combined direction depends on the affine result and stays a
 * call of its own (per iteration: 1 update + (2 paired + 1) solves).  Equivalent to setrhs_dev + solve_dev_enqueue twice (two
 * verdicts for chip_kkt_collect, in this order).  On systems whose top is level-scheduled the two chains of launches are
 * enqueued on two streams before either is waited for and overlap on the device; fused handles run their two persistent
 * launches one after the other. */
int32_t chip_kkt_solve2_dev_enqueue(chip_kkt *h, const double *rhsx_a, const double *rhsz_a, double *lhsx_a, double *lhsz_a,
                                    const double *rhsx_b, const double *rhsz_b, double *lhsx_b, double *lhsz_b);
/* chip_kkt_update_scaling_dev + chip_kkt_update_enqueue as ONE enqueue (core/solver.rs:334-352 calls
 * cones.update_scaling and kktsystem.update back to back): with Zero / Nonnegative / SecondOrder cones the scaling
 * and the Hs / sparse-cone writes of a cone run in one launch and the refactor's preparation launches are folded
 * into the bundle factorisation; other cone kinds take the two calls as they are.  Verdicts (cones interior, pivots
 * finite) with chip_kkt_collect. */
int32_t chip_kkt_update_scaled_enqueue(chip_kkt *h, const double *s_dev, const double *z_dev, double mu, int32_t strategy,
                                       const double *hsblocks_or_null);
int32_t chip_kkt_solve_dev_enqueue(chip_kkt *h, double *lhsx_dev_or_null, double *lhsz_dev_or_null);
int32_t chip_kkt_collect(chip_kkt *h, int32_t *update_ok, int32_t *nsolves, int32_t solves_ok[16]);
/* KKTSolver::solve / ::update receive `settings: &CoreSettings` on EVERY call in the reference
 * (kktsolvers/mod.rs:7-18, directldlkktsolver.rs:134,168): this hands the current values of the
 * regularisation and refinement fields to the following calls.  The engine knobs (device, amd_dense_scale,
 * use_graph) keep their construction-time values. */
int32_t chip_kkt_set_settings(chip_kkt *h, const chip_settings *settings);
/* The verdict of the last chip_kkt_update_scaling_dev, which itself stays asynchronous and returns 1:
 * 1 = every SOC / PSD scaling succeeded, 0 = a cone left its interior (socone.rs:146-149,
 * psdtrianglecone.rs:165-169).  One 256-byte device-to-host copy + synchronisation; a device-resident
 * driver calls it where the reference tests the bool of cones.update_scaling (core/solver.rs:334-338),
 * or relies on chip_kkt_update, which reports the same verdict after the refactor. */
int32_t chip_kkt_scaling_ok(chip_kkt *h);
/* update_P / update_A   directldlkktsolver.rs:191-197 */
int32_t chip_kkt_update_P(chip_kkt *h, const double *Pnzval);
int32_t chip_kkt_update_A(chip_kkt *h, const double *Anzval);
/* CompositeCone::mul_Hs (compositecone.rs:259-264) for the device-held cones:
 * y = Hs x, m doubles, device pointers. */
int32_t chip_kkt_mul_Hs_dev(chip_kkt *h, double *y_dev, const double *x_dev);
/* ---- the cone operations either side of the KKT solve, for problems whose
 * cones are Zero / Nonnegative / SecondOrder / Exponential / Power / GenPower / PSDTriangle;
 * m-vectors in HBM.  PSDTriangle:
 * psdtrianglecone.rs:104-303 with symmetric_common.rs:53-95 -- mul_W / mul_Winv as two n x n
 * products with R / Rinv, circ_op, lambda \ ., step length and margins from the eigenvalues of a
 * parallel two-sided Jacobi iteration, barrier from a Cholesky log-determinant (in LDS up to side 64, in HBM scratch beyond).  Exponential / Power: affine_ds = s
 * (expcone.rs:129-131), combined_ds_shift = sigma*mu*grad - 3rd-order correction
 * (expcone.rs:133-142,254-308, powcone.rs:132-141,260-337; step_z / step_s are left unchanged),
 * ds_from_dz_offset = ds, step_length = backtracking line searches (nonsymmetric_common.rs:164-192)
 * started -- as in compositecone.rs:300-340 -- from the symmetric cones' step backed off to
 * 1 - sqrt(eps).
 *   affine_ds            compositecone.rs:266-272   (nonnegativecone.rs:110-115, socone.rs:258-260)
 *   combined_ds_shift    compositecone.rs:274-289   (symmetric_common.rs:53-84): step_z and step_s are
 *                        overwritten by W dz and W^-1 ds exactly as in the reference
 *   ds_from_dz_offset    compositecone.rs:291-299   (nonnegativecone.rs:122-126, socone.rs:266-287)
 *   step_length          compositecone.rs:300-340   (nonnegativecone.rs:128-153, socone.rs:289-302,421-495)
 *                        *alpha_out (host) = the common (alpha_z, alpha_s) of the composite cone
 *   margins              compositecone.rs:197-206   (nonnegativecone.rs:58-62, socone.rs:104-108)        */
int32_t chip_kkt_affine_ds_dev(chip_kkt *h, double *ds_dev, const double *s_dev);
int32_t chip_kkt_combined_ds_shift_dev(chip_kkt *h, double *shift_dev, double *step_z_dev,
                                       double *step_s_dev, double sigma_mu);
int32_t chip_kkt_ds_from_dz_offset_dev(chip_kkt *h, double *out_dev, const double *ds_dev,
                                       const double *z_dev);
int32_t chip_kkt_step_length_dev(chip_kkt *h, const double *dz_dev, const double *ds_dev,
                                 const double *z_dev, const double *s_dev, double alpha_max,
                                 double *alpha_out);
int32_t chip_kkt_margins_dev(chip_kkt *h, const double *z_dev, double *alpha_out, double *beta_out);
/* scaled_unit_shift  compositecone.rs:208-214 (nonnegativecone.rs:64-66, socone.rs:110-112,
 * zerocone.rs:63-69): z += alpha * e; primal_cone != 0 selects PrimalOrDualCone::PrimalCone
 * (Zero-cone rows are zeroed); with margins this is the initial-point fix-up
 * _shift_to_cone_interior, default/variables.rs:231-256 */
int32_t chip_kkt_scaled_unit_shift_dev(chip_kkt *h, double *z_dev, double alpha, int32_t primal_cone);
/* unit_initialization  compositecone.rs:208-214 (zerocone.rs:71-74, nonnegativecone.rs:68-71,
 * socone.rs:114-119, expcone.rs:87-93, powcone.rs:79-87): fills z[m], s[m] */
int32_t chip_kkt_unit_initialization_dev(chip_kkt *h, double *z_dev, double *s_dev);
/* GenPowerConeT(alpha, dim2) (cones/genpowcone.rs:49-63): chip_kkt_create takes dims[i] = len(alpha) and
 * dims2[i] = dim2; the powers themselves (positive, summing to one) are handed over here, before the
 * first chip_kkt_update_scaling (until then alpha = 1/dim1).  The cone is nonsymmetric and allows the
 * dual scaling only (genpowcone.rs:96-98): Hs = mu H(z) as a diagonal plus the rank-3 sparse
 * expansion [q, r, p] of datamaps.rs:227-343, all computed and written into K on the device. */
int32_t chip_kkt_set_genpow_alpha(chip_kkt *h, int64_t cone_index, const double *alpha);
/* compute_barrier  compositecone.rs:342-352 at (z, s) + alpha (dz, ds): nonnegativecone.rs:155-166,
 * socone.rs:304-314, expcone.rs:170-252, powcone.rs:169-258; *barrier_out on the host */
int32_t chip_kkt_compute_barrier_dev(chip_kkt *h, const double *z_dev, const double *s_dev,
                                     const double *dz_dev, const double *ds_dev, double alpha,
                                     double *barrier_out);
int32_t chip_kkt_info(const chip_kkt *h, chip_info *info);
int32_t chip_kkt_get_perm(const chip_kkt *h, uint64_t *perm);
int32_t chip_kkt_get_symbolic(const chip_kkt *h, uint64_t *etree, uint64_t *Lp, uint64_t *Li,
                              uint64_t *level);
/* diagnostics: chain supernodes of the top chosen by the symbolic analysis -- runs of columns, each
 * the elimination-tree parent of the previous one, whose structures are padded (explicit zeros) to one
 * dense trapezoid that a single workgroup factors with dense block operations.  *count supernodes;
 * ptr[count + 1] / cols[ptr[count]] (permuted numbering) may be NULL.  Call with NULLs first to size. */
int32_t chip_kkt_get_supernodes(const chip_kkt *h, int64_t *count, uint64_t *ptr, uint64_t *cols);
int32_t chip_ldl_get_supernodes(const chip_ldl *h, int64_t *count, uint64_t *ptr, uint64_t *cols);
/* current device copy of K.nzval (unregularised), for tests */
int32_t chip_kkt_get_values(chip_kkt *h, double *nzval);
/* full-N right-hand side / solution (incl. the p sparse-cone rows), tests only */
int32_t chip_kkt_solve_full(chip_kkt *h, double *x, const double *b);
/* blocks until all work queued on the handle's stream has finished */
int32_t chip_kkt_synchronize(chip_kkt *h);
/* HIP stream (hipStream_t) the handle launches on, for event timing */
void *chip_kkt_stream(chip_kkt *h);
/* hipEvent pairs (recorded on the launch stream) around every launch of ONE
 * kernel family: 0 = off, 1 = symv residual k_bundle_symv, 2 = backward
 * substitution k_gather_T<BWD>, 3 = forward k_gather_T<FWD>, 4 = k_factor_T (families 1-4 force the
 * one-kernel-per-phase solve path), 5 = k_bundle_ir (the fused solve + refinement launch), 6 = k_bundle_factor.
 * chip_kkt_profile(h, family) resets the counters; chip_kkt_profile_read
 * returns out[0] = launches, out[1] = total ms, out[2] = family. */
int32_t chip_kkt_profile(chip_kkt *h, int32_t family);
int32_t chip_kkt_profile_read(chip_kkt *h, double out[8]);
/* further families (systems whose top runs as chain supernodes, configs 2 / 5): 7 = k_snode_update (f64 MFMA tiles),
 * 8 = k_snode_diag, 9 = k_snode_rows, 10 = k_snode_extend, 11 = k_snode_tri (pipelined substitution through wide
 * supernodes, both sweeps), 12 = k_gather_merged launches of the supernode substitution path.
 * chip_kkt_work_model: the work those kernels do per refactor / per sweep, from the supernode geometry:
 * out[0] = flops of k_snode_update per refactor (2 per multiply-add, rows at or below the block only),
 * out[1] = entries of the dense trapezoids (streamed once per sweep), out[2] = flops of k_snode_extend,
 * out[3] = flops of k_snode_diag + k_snode_rows, out[4] = number of supernodes; and of the bundle part:
 * out[5] = groups of a grouped fold in use (a forest cut into several bundles per tree, each tree's top of at most 8
 * nodes folded into its bundles' kernels; 0 = none), out[6] = subtree bundles, out[7] = threads per workgroup of the
 * fused solve launch (0 = the handle's solve is not the fused launch). */
int32_t chip_kkt_work_model(const chip_kkt *h, double out[8]);
/* chip_kkt_sweep_model (round 5): how the substitutions run through the chain supernodes.  Supernodes of moderate width
 * keep a substitution matrix G = [I; L_B] T^-1 (built once per refactor) and a sweep is ONE pass over it per unit level,
 * without a chain of block hops (csrc/snode_g.hip); wider ones keep the pipelined block-by-block substitution.
 * out[0] = doubles of all G (what one sweep through those supernodes streams), out[1] = unit levels on that path,
 * out[2] = unit levels that hold supernodes, out[3] = launches of the build per refactor (0 or 1). */
int32_t chip_kkt_sweep_model(const chip_kkt *h, double out[4]);
/* which of the grouped-fold step kernels (csrc/bundle_gstep.hip) this handle uses: bit 0 = the fused solve launch is
 * k_gstep_solve (a bundle's entries of L and K in registers), bit 1 = the refactor's bundle part is k_gstep_factor
 * (bundle columns + Schur shares + the groups' tops in one launch); bit 2 = the fused solve launch is k_bundle_irs (one
 * bundle per workgroup, the iterates on chip, no permuted copy of b: csrc/bundle_ir.hip); 0 = none of them (diagnostics) */
int32_t chip_kkt_step_kernels(const chip_kkt *h);
/* number of fused solve launches (k_bundle_ir) of this handle whose grid barrier timed out -- their workgroups were not
 * all resident because another long-running kernel held the slots -- and that were repeated on the
 * one-kernel-per-phase path (refinement decisions on the host, same results up to rounding).  For enqueued solves the
 * repeat happens in chip_kkt_collect, for the timed-out solve and every solve enqueued behind it, from the right-hand
 * side buffers as they are THEN: the buffers of pending solves must stay untouched until collect. */
int32_t chip_kkt_fused_fallbacks(const chip_kkt *h);
/* (test hooks -- chip_debug_spin, chip_debug_set_switch -- are declared in clarabel_hip_testing.h and exist only in
 * libraries built with -DCHIP_TESTING, the in-tree default: `make TESTING=0` builds without them) */

/* ===========================================================================
 * L3 -- the caller either side of the KKT solve, device resident:
 *   DefaultKKTSystem  src/solver/implementations/default/kktsystem.rs:16-292
 *   DefaultResiduals  src/solver/implementations/default/residuals.rs:69-111
 * Built on the chip_kkt handle (borrowed: it must outlive the chip_kktsystem)
 * and launching on that handle's stream.  Vectors are DEVICE pointers: x[n],
 * z[m], s[m] as in DefaultVariables (default/variables.rs:12-36); tau/kappa
 * travel by value.  Dot products are deterministic two-stage reductions; the
 * scalar algebra (tau numerator / denominator, kktsystem.rs:170-186) runs on
 * the host in the reference's order after ONE device-to-host copy per call.
 * Return values: 1 / 0 like the reference's bool, or a negative chip_status.
 * ===========================================================================*/
typedef struct chip_kktsystem chip_kktsystem;
typedef struct {
    double *x, *z, *s; /* device, n / m / m */
    double tau, kappa;
} chip_vars;
enum { CHIP_STEP_AFFINE = 0, CHIP_STEP_COMBINED = 1 }; /* StepDirection, core/mod.rs */
/* DefaultKKTSystem::new (kktsystem.rs:38-88) given the already-built KKTSolver; P triu CSC
 * (n x n), A CSC (m x n), q[n], b[m] are copied to the device */
int32_t chip_kktsystem_create(chip_kktsystem **out, chip_kkt *kkt, const uint64_t *Pcolptr,
                              const uint64_t *Prowval, const double *Pnzval, const uint64_t *Acolptr,
                              const uint64_t *Arowval, const double *Anzval, const double *q,
                              const double *b);
void chip_kktsystem_destroy(chip_kktsystem *h);
/* KKTSystem::update (kktsystem.rs:108-125): kktsolver.update (Hs from the device-held cone
 * state, see chip_kkt_update_scaling*) + the constant-RHS solve (x2, z2) (:264-279) */
int32_t chip_kktsystem_update(chip_kktsystem *h);
/* KKTSystem::solve (kktsystem.rs:127-209): lhs <- step for rhs at `variables`; lhs->tau/kappa
 * are written.  Needs cones with mul_Hs / ds_from_dz_offset on the device, else
 * CHIP_ERR_UNSUPPORTED. */
int32_t chip_kktsystem_solve(chip_kktsystem *h, chip_vars *lhs, const chip_vars *rhs,
                             const chip_vars *variables, int32_t step_direction);
/* KKTSystem::solve_initial_point (kktsystem.rs:211-258): fills variables->x, s, z */
int32_t chip_kktsystem_solve_initial_point(chip_kktsystem *h, chip_vars *variables);
/* Residuals::update (residuals.rs:69-111): device outputs rx[n], rz[m], rx_inf[n], rz_inf[m],
 * Px[n]; host outputs out5 = {r_tau, dot_qx, dot_bz, dot_sz, dot_xPx} */
int32_t chip_residuals_update(chip_kktsystem *h, const chip_vars *variables, double *rx_dev,
                              double *rz_dev, double *rx_inf_dev, double *rz_inf_dev, double *Px_dev,
                              double out5[5]);
/* the same + the Euclidean norms DefaultInfo::update (default/info.rs:142-165) takes of the vectors
 * involved, norms5 = {||x||, ||z||, ||s||, ||rz||, ||rx||} (NULL = skip), reduced in the same launches
 * and returned by the same single device-to-host copy */
int32_t chip_residuals_update_norms(chip_kktsystem *h, const chip_vars *variables, double *rx_dev,
                                    double *rz_dev, double *rx_inf_dev, double *rz_inf_dev, double *Px_dev,
                                    double out5[5], double *norms5_or_null);
/* data_updating.rs:98-133: new values on the same patterns (NULL = unchanged); P and A are
 * forwarded to chip_kkt_update_P / chip_kkt_update_A */
int32_t chip_kktsystem_update_data(chip_kktsystem *h, const double *Pnzval_or_null,
                                   const double *Anzval_or_null, const double *q_or_null,
                                   const double *b_or_null);


/* ---------------------------------------------------------------------------
 * DefaultVariables (src/solver/implementations/default/variables.rs:58-261) on
 * device-resident vectors: the step algebra either side of the KKT solve, so that a
 * whole IPM iteration (core/solver.rs:282-434) exchanges only scalars with the host.
 * rx / rz are the device vectors written by chip_residuals_update.
 * ---------------------------------------------------------------------------*/
/* CompositeCone::degree (compositecone.rs:106-108) */
int32_t chip_kkt_degree(const chip_kkt *h, int64_t *degree);
/* calc_mu, variables.rs:63-66 (host arithmetic on dot_sz of chip_residuals_update) */
int32_t chip_variables_calc_mu(chip_kktsystem *h, const chip_vars *variables, double dot_sz, double *mu_out);
/* affine_step_rhs, variables.rs:68-79: d <- (rx, rz, affine_ds(s), r_tau, tau*kappa) */
int32_t chip_variables_affine_step_rhs(chip_kktsystem *h, chip_vars *d, const double *rx_dev,
                                       const double *rz_dev, double rtau, const chip_vars *variables);
/* combined_step_rhs, variables.rs:81-118: d.s must hold affine_ds (as in the reference); step->z is
 * scaled by m when m != 1 and step->z / step->s are overwritten by the cones' combined_ds_shift */
int32_t chip_variables_combined_step_rhs(chip_kktsystem *h, chip_vars *d, const double *rx_dev,
                                         const double *rz_dev, double rtau, const chip_vars *variables,
                                         chip_vars *step, double sigma, double mu, double m);
/* calc_step_length, variables.rs:120-160 (max_step_fraction: settings.core().max_step_fraction) */
int32_t chip_variables_calc_step_length(chip_kktsystem *h, const chip_vars *variables, const chip_vars *step,
                                        int32_t step_direction, double max_step_fraction, double *alpha_out);
/* add_step, variables.rs:162-168 (variables->tau / kappa updated in the struct) */
int32_t chip_variables_add_step(chip_kktsystem *h, chip_vars *variables, const chip_vars *step, double alpha);
/* symmetric_initialization, variables.rs:170-176 with _shift_to_cone_interior (:231-261) */
int32_t chip_variables_symmetric_initialization(chip_kktsystem *h, chip_vars *variables);
/* unit_initialization, variables.rs:178-184 */
int32_t chip_variables_unit_initialization(chip_kktsystem *h, chip_vars *variables);
/* barrier, variables.rs:205-227 (dot_shifted of vecmath.rs:87-99 + the cones' barriers) */
int32_t chip_variables_barrier(chip_kktsystem *h, const chip_vars *variables, const chip_vars *step,
                               double alpha, double *barrier_out);
/* rescale, variables.rs:229-239 */
int32_t chip_variables_rescale(chip_kktsystem *h, chip_vars *variables);
/* Euclidean norms of up to 8 device vectors with one host synchronisation (the norms
 * DefaultInfo::update reads, default/info.rs:142-165, with identity equilibration) */
int32_t chip_vec_norms(chip_kktsystem *h, int32_t count, const double *const *vecs_dev, const int64_t *lens,
                       double *out);

/* ===========================================================================
 * Sharded path: one process per GPU, whole connected components of the elimination
 * forest per rank (BASELINE config 4: 1024 independent SOCPs, 128 per GPU at 8 GPUs).  Factorisation,
 * substitutions and refinement of a rank's blocks need no exchange; the exchange step is ONE RCCL
 * all-gather of the step direction per KKT solve over xGMI, plus scalar reductions.  The reference has
 * no distributed mode: these entry points have no reference counterpart; the
 * reference-side caller is the loop of core/solver.rs:282-434, which consumes the full (dx, dz) and the
 * scalars tau, kappa, mu, alpha.
 * ===========================================================================*/
typedef struct chip_comm chip_comm;
#define CHIP_COMM_ID_BYTES 128
/* rank 0 creates the rendezvous token (ncclGetUniqueId) and hands it to the other ranks by any
 * out-of-band channel (file, socket, MPI): the library opens no sockets of its own */
int32_t chip_comm_get_unique_id(uint8_t id[CHIP_COMM_ID_BYTES]);
/* collective over all `world` ranks (ncclCommInitRank); device: HIP ordinal, -1 = current */
int32_t chip_comm_create(chip_comm **out, const uint8_t id[CHIP_COMM_ID_BYTES], int32_t world, int32_t rank,
                         int32_t device);
void chip_comm_destroy(chip_comm *c);
int32_t chip_comm_info(const chip_comm *c, int32_t *world, int32_t *rank);
/* LinearSolverInfo.threads of the handle then reports the number of GPUs working on the problem */
int32_t chip_kkt_attach_comm(chip_kkt *h, chip_comm *c);
/* all-gather of the step direction: recv_dev[sum(counts[<r]) ..] <- rank r's send_dev[0 .. counts[r])
 * (counts: `world` entries, host).  Enqueued on the communicator's stream behind the work queued so
 * far on the handle's stream (event, no host synchronisation); returns immediately. */
int32_t chip_kkt_allgather_step(chip_kkt *h, chip_comm *c, const double *send_dev, double *recv_dev,
                                const int64_t *counts);
/* the handle's stream waits (on the device) for the last chip_kkt_allgather_step -- call before
 * send_dev / recv_dev are written again */
int32_t chip_kkt_wait_comm(chip_kkt *h, chip_comm *c);
int32_t chip_comm_synchronize(chip_comm *c);
/* in-place all-reduce of up to 64 host scalars; op: 0 = sum, 1 = min, 2 = max.  Blocking. */
int32_t chip_comm_allreduce(chip_comm *c, double *vals, int32_t count, int32_t op);

#ifdef __cplusplus
}
#endif
#endif /* CLARABEL_HIP_H */
