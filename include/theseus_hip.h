/* theseus_hip.h -- C ABI of libtheseus_hip.so: the MI355X (gfx950) batched Gauss-Newton /
 * Levenberg-Marquardt inner loop for Theseus-style SE3 pose graphs.
 *
 * The reference (facebookresearch/theseus, /root/reference) has NO C ABI on this path: the
 * boundary is two Python ABCs, `Linearization` (theseus/optimizer/linearization.py:16-87) and
 * `LinearSolver` (theseus/optimizer/linear/linear_solver.py:15-37), whose dense implementations
 * are plain ATen calls (theseus/optimizer/dense_linearization.py:29-62,
 * theseus/optimizer/linear/dense_solver.py:38-64,159-161).  Each entry point below names the
 * reference interface it replaces.  Host code (the theseus_amd Python package, or a theseus plugin, see
 * INTEGRATION.md) binds these with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all memory;
 *   - `dtype`: 0 = float32, 1 = float64 (arithmetic and storage type of every floating buffer);
 *   - `stream`: a hipStream_t (pass torch.cuda.current_stream().cuda_stream); nothing synchronises;
 *   - return value: 0 on success, negative on a bad argument or launch error
 *     (thx_last_error() gives the text); no global state besides that thread-local string and, per device, the
 *     launch-side state of the dense solver (raised dynamic-LDS limits, the auxiliary stream + events of
 *     thx_chol_factor's two-stream schedule): the entry points act on the CURRENT device (hipGetDevice), may be called
 *     for several devices and from several threads (thx_chol_* enqueues are serialised by a mutex);
 *   - batch layouts are "entity major": poses (P, B, 3, 4), measurements (E, Bm, 3, 4) with Bm in
 *     {1, B}; `*_bstride` is the element stride between consecutive batch items (12 / 6, or 0 when
 *     the tensor is shared by the whole batch);
 *   - dense matrices are row major (B, ld, ld) with ld >= n, ld % 32 == 0; only the lower
 *     triangle of H / L is meaningful to the solver.
 */
#ifndef THESEUS_HIP_H_
#define THESEUS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define THX_F32 0
#define THX_F64 1
#define THX_TILE 128 /* Cholesky tile edge; Winv holds ceil(n/THX_TILE) solve panels (see thx_chol_factor) */

/* Taylor-switch thresholds of torchlie (torchlie/torchlie/global_params.py:44-58), read by the
 * host at launch time so that runtime changes made through torchlie.set_global_params apply. */
typedef struct {
  double near_zero;   /* so3_near_zero_eps   */
  double d_near_zero; /* so3_d_near_zero_eps */
  double near_pi;     /* so3_near_pi_eps     */
} thx_lie_eps;

/* Immutable structure of one pose-graph objective ("problem compiler" output; all device int32).
 * Column layout = pose index * 6 (Linearization.var_start_cols, linearization.py:31-41). */
typedef struct {
  int32_t num_poses;   /* P : SE3 optimisation variables, dof 6 each                    */
  int32_t num_edges;   /* E : Between costs (v0 = pose edge_i[e], v1 = pose edge_j[e])   */
  int32_t num_priors;  /* K : Difference/Local costs on pose prior_pose[k]               */
  const int32_t* edge_i;     /* (E)                                                     */
  const int32_t* edge_j;     /* (E)                                                     */
  const int32_t* inc_ptr;    /* (P+1) CSR over poses: incident edge entries             */
  const int32_t* inc_edge;   /* (2E) edge id, entries of a pose sorted by other endpoint */
  const int32_t* inc_side;   /* (2E) 0: this pose is v0 of the edge, 1: it is v1        */
  const int32_t* inc_other;  /* (2E) the other endpoint's pose index                    */
  const int32_t* prior_pose; /* (K)                                                     */
  const int32_t* pri_ptr;    /* (P+1) CSR over poses: prior ids                         */
  const int32_t* pri_id;     /* (K)                                                     */
} thx_pg_structure;

/* Per-call tensors of the objective. */
typedef struct {
  int32_t batch;             /* B */
  const void* poses;         /* (P, B, 3, 4)                                            */
  const void* meas;          /* (E, Bm, 3, 4) Between measurements (aux variables)      */
  int64_t meas_bstride;      /* 12 or 0                                                 */
  const void* w_between;     /* (E, Bw, 6) sqrt-information diagonal (DiagonalCostWeight;
                                a ScaleCostWeight is passed expanded)                   */
  int64_t w_between_bstride; /* 6 or 0                                                  */
  const void* prior_target;  /* (K, Bt, 3, 4)                                           */
  int64_t prior_target_bstride;
  const void* w_prior;       /* (K, Bw, 6)                                              */
  int64_t w_prior_bstride;
  /* RobustCostFunction wrappers (theseus/core/robust_cost_function.py:52-135): a LOSS CODE = THX_LOSS_{NONE,WELSCH,HUBER,HINGE,GEMAN_MCCLURE},
   * | THX_LOSS_FLATTEN for flatten_dims = True (every residual row its own robust term, :89-96,118-133), and
   * log_loss_radius per cost.  robust_<role> is the code of every cost of the role; when the costs of a role differ
   * (some plain, some Welsch, some Huber, some flattened) robust_<role> is any non-zero code and loss_<role> holds one code
   * per cost.  Code THX_LOSS_NONE: the cost's log_radius entry is not read (role code NONE and no table: the pointer is
   * not read). */
  int32_t robust_between;
  const void* log_radius_between;    /* (E, Br, 1), Br in {1, B} */
  int64_t log_radius_between_bstride; /* 1 or 0 */
  int32_t robust_prior;
  const void* log_radius_prior;      /* (K, Br, 1) */
  int64_t log_radius_prior_bstride;
  const int32_t* loss_between;       /* (E) loss code per Between cost, or NULL: robust_between for all */
  const int32_t* loss_prior;         /* (K) */
} thx_pg_data;

/* theseus/core/robust_loss.py:33-113; THX_LOSS_FLATTEN: RobustCostFunction(flatten_dims=True) */
#define THX_LOSS_NONE 0
#define THX_LOSS_WELSCH 1
#define THX_LOSS_HUBER 2
#define THX_LOSS_HINGE 3   /* robust_loss.py:55-62: rho = sqrt(x) - sqrt(r) beyond the radius, rho' = 0 inside it */
#define THX_LOSS_FLATTEN 4
#define THX_LOSS_GEMAN_MCCLURE 8   /* robust_loss.py:92-113 (GNCRobustCostFunction, robust_cost_function.py:173-222): the cost's
                                    * log_radius entry is log(mu * radius) -- the host adds log(gnc_control_val) */

const char* thx_last_error(void);
int thx_abi_version(void);

/* ---- SE3 elementwise ops: torchlie.functional.SE3.{exp,log,compose,inv,adj}
 *      (torchlie/torchlie/functional/se3_impl.py:178-216,225-310,354-457,531-538,578-581,703-708).
 *      N independent elements; jac may be NULL; group tensors (N,3,4), tangents (N,6), jac (N,6,6). */
int thx_se3_exp(const void* xi, void* X, void* jac, int64_t N, int dtype, const thx_lie_eps* eps, void* stream);
int thx_se3_log(const void* X, void* xi, void* jac, int64_t N, int dtype, const thx_lie_eps* eps, void* stream);
int thx_se3_compose(const void* X, const void* Y, void* Z, int64_t N, int dtype, void* stream);
int thx_se3_inverse(const void* X, void* Y, int64_t N, int dtype, void* stream);
int thx_se3_adjoint(const void* X, void* A, int64_t N, int dtype, void* stream);

/* ---- SE2 (2-D SLAM) twins of the pose-graph entry points: theseus/geometry/se2.py (:165-229 log + Jlog, :239-300
 *      exp + Jexp, :309-339 adjoint / compose / inverse), tensors [x, y, cos, sin] -> group records of 4, tangents /
 *      weights of 3, 3x3 blocks, column layout pose * 3.  thx_pg_structure / thx_pg_data are shared (batch strides
 *      4 / 3 / 0).  thx_se2_op: 0 exp (a = xi (N,3) -> out (N,4), jac (N,3,3) or NULL), 1 log (a = X -> out (N,3),
 *      jac), 2 compose (a, b -> out), 3 inverse, 4 adjoint (out (N,3,3)).  The solver entry points are group
 *      agnostic. */
typedef struct {
  double near_zero;   /* se2_near_zero_eps   (theseus/global_params.py:46-59) */
  double d_near_zero; /* se2_d_near_zero_eps */
} thx_se2_eps;
int thx_pg2_assemble(const thx_pg_structure* s, const thx_pg_data* d, void* H, int64_t ld, void* g, int dtype,
                     const thx_se2_eps* eps, void* stream);
int thx_pg2_error(const thx_pg_structure* s, const thx_pg_data* d, void* partials, void* err, int dtype,
                  const thx_se2_eps* eps, void* stream);
int thx_pg2_jacobians(const thx_pg_structure* s, const thx_pg_data* d, void* J0, void* J1, void* eb, void* Jp,
                      void* ep, int dtype, const thx_se2_eps* eps, void* stream);
int thx_se2_retract(const void* poses, const void* delta, int64_t ldd, double step, const uint8_t* ignore_mask,
                    void* out, int32_t P, int32_t B, int dtype, const thx_se2_eps* eps, void* stream);
int thx_se2_op(int op, const void* a, const void* b, void* out, void* jac, int64_t N, int dtype,
               const thx_se2_eps* eps, void* stream);
/* implicit backward on SE2 pose graphs (thx_se3_retract_vjp / thx_pg_vjp below, with 4-element records, 3-vectors): plain
 * derivatives of the se2.py closed forms (the reference has no custom backward for SE2). */
int thx_se2_retract_vjp(const void* poses, const void* delta, int64_t ldd, double step, const void* grad_out,
                        void* grad_delta, int64_t ldg, int32_t P, int32_t B, int dtype, const thx_se2_eps* eps,
                        void* stream);
int thx_pg2_vjp(const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw, void* grad_meas,
                void* grad_w_between, void* grad_prior_target, void* grad_w_prior, void* grad_log_radius_between,
                void* grad_log_radius_prior, int dtype, const thx_se2_eps* eps, void* stream);

/* ---- SO2 variables (planar rotation-only graphs): theseus/geometry/so2.py -- exp_map :167-186 (update_from_angle :96-100),
 *      _log_map_impl :206-223, _adjoint_impl :116-117, _compose_impl :225-231, _inverse_impl :233-235 -- under Between / Local
 *      (embodied/measurements/between.py:38-45, embodied/misc/local_cost_fn.py:42-61) and the retraction
 *      (geometry/lie_group.py:197-198).  Group records of 2 ([cos, sin]), tangents / weights of 1, 1x1 blocks, column layout
 *      pose * 1; thx_pg_structure / thx_pg_data are shared (batch strides 2 / 1 / 0).  The group has no Taylor switches: no eps
 *      argument.  Records are never re-normalised (the reference's constructors pass tensors through unchanged).
 *      thx_so2_op: 0 exp (a = theta (N,1) -> out (N,2), jac (N,1,1) = 1 or NULL), 1 log (a = X -> out (N,1), jac), 2 compose
 *      (a, b -> out), 3 inverse, 4 adjoint (out (N,1,1) = 1). */
int thx_pgso2_assemble(const thx_pg_structure* s, const thx_pg_data* d, void* H, int64_t ld, void* g, int dtype, void* stream);
int thx_pgso2_error(const thx_pg_structure* s, const thx_pg_data* d, void* partials, void* err, int dtype, void* stream);
int thx_pgso2_jacobians(const thx_pg_structure* s, const thx_pg_data* d, void* J0, void* J1, void* eb, void* Jp, void* ep,
                        int dtype, void* stream);
int thx_so2_retract(const void* poses, const void* delta, int64_t ldd, double step, const uint8_t* ignore_mask, void* out,
                    int32_t P, int32_t B, int dtype, void* stream);
int thx_so2_op(int op, const void* a, const void* b, void* out, void* jac, int64_t N, int dtype, void* stream);

/* ---- SO3 variables (rotation-only graphs): theseus/geometry/so3.py over torchlie's SO3 closed forms
 *      (torchlie/torchlie/functional/so3_impl.py:220-261 exp, :270-320 Jexp, :390-433 log, :442-479 Jlog; adjoint = R,
 *      inverse = R^T, compose = R0 R1), group records of 9 (3x3 row major), tangents / weights of 3, 3x3 blocks, column
 *      layout pose * 3; thx_pg_structure / thx_pg_data are shared (batch strides 9 / 3 / 0), thresholds = the so3_* ones of
 *      thx_lie_eps.  thx_so3_op: 0 exp (a = w (N,3) -> out (N,3,3), jac (N,3,3) or NULL), 1 log (a = R -> out (N,3), jac),
 *      2 compose (a, b -> out), 3 inverse, 4 adjoint (out (N,3,3)). */
int thx_pgso3_assemble(const thx_pg_structure* s, const thx_pg_data* d, void* H, int64_t ld, void* g, int dtype,
                       const thx_lie_eps* eps, void* stream);
int thx_pgso3_error(const thx_pg_structure* s, const thx_pg_data* d, void* partials, void* err, int dtype,
                    const thx_lie_eps* eps, void* stream);
int thx_pgso3_jacobians(const thx_pg_structure* s, const thx_pg_data* d, void* J0, void* J1, void* eb, void* Jp,
                        void* ep, int dtype, const thx_lie_eps* eps, void* stream);
int thx_so3_retract(const void* poses, const void* delta, int64_t ldd, double step, const uint8_t* ignore_mask,
                    void* out, int32_t P, int32_t B, int dtype, const thx_lie_eps* eps, void* stream);
int thx_so3_op(int op, const void* a, const void* b, void* out, void* jac, int64_t N, int dtype,
               const thx_lie_eps* eps, void* stream);
/* implicit backward on SO3 rotation graphs (thx_se3_retract_vjp / thx_pg_vjp below, with 9-element records, 3-vectors), with
 * torchlie's SO3 backward semantics: Exp.backward (so3_impl.py:336-353), Log's passthrough backward (:489-496: the tangent
 * projection R lift(Jlog^T g / 2)), plain matrix derivatives for Inverse / Compose (:576-577, :702-707), plain autograd through
 * the Jlog closed forms. */
int thx_so3_retract_vjp(const void* poses, const void* delta, int64_t ldd, double step, const void* grad_out,
                        void* grad_delta, int64_t ldg, int32_t P, int32_t B, int dtype, const thx_lie_eps* eps, void* stream);
int thx_pgso3_vjp(const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw, void* grad_meas,
                  void* grad_w_between, void* grad_prior_target, void* grad_w_prior, void* grad_log_radius_between,
                  void* grad_log_radius_prior, int dtype, const thx_lie_eps* eps, void* stream);

/* ---- Linearization.linearize(): replaces DenseLinearization._linearize_jacobian_impl +
 *      _linearize_hessian_impl (dense_linearization.py:29-62) fused with Between / Local
 *      Jacobians (embodied/measurements/between.py:38-45, embodied/misc/local_cost_fn.py:58-61)
 *      and cost weights (core/cost_weight.py:81-90,125-136).  Never materialises A.
 *      Writes the non-zero 6x6 blocks of the lower triangle of H = A^T A into the dense
 *      (B, ld, ld) buffer (zero elsewhere must be pre-set once by the caller: the sparsity
 *      pattern is fixed) and g = A^T b = -J^T e into (B, n). */
int thx_pg_assemble(const thx_pg_structure* s, const thx_pg_data* d, void* H, int64_t ld, void* g,
                    int dtype, const thx_lie_eps* eps, void* stream);

/* ---- BLOCK-COMPACT Hessian.  The same lower triangle of H = A^T A (dense_linearization.py:55-62), stored as what it is: the
 *      list of its non-zero bd x bd blocks (SE3 pose graph: 6 x 6; P diagonal + one per connected pose pair), (B, nblocks, bd*bd)
 *      with problem stride `bstride` elements -- 184 KB per problem at 256 poses / 1024 edges where the dense frame's lower
 *      triangle is 4.7 MB of mostly zeros.  Blocks are ordered by the 128 x 128 Cholesky tile of their top-left element, so the
 *      blocks of a tile are one contiguous run.  All tables are DEVICE int32 arrays built once by the host
 *      (theseus_amd/compiler.py:HessianBlocks):
 *        diag_blk[k]          block id of the diagonal block of variable k;
 *        inc_blk[e]           for entry e of thx_pg_structure's incidence lists (inc_edge / inc_side / inc_other): the id of the
 *                             off-diagonal block it accumulates into when inc_other < the pose (lower triangle), else -1;
 *        tile_ptr / piece_*   per lower tile t = i (i + 1) / 2 + j (i >= j): the pieces [tile_ptr[t], tile_ptr[t + 1]) that fall
 *                             into it -- piece_blk: block id, piece_rc: position of the block's top-left element relative to the
 *                             tile origin, (int16 row) << 16 | (uint16) (int16 col), negative when the block starts in the tile
 *                             above / to the left (128 is not a multiple of 6: a block may straddle up to four tiles).
 *      thx_pg_assemble_blocks: thx_pg_assemble writing the block list (every block is written in full, 16-byte pieces: no
 *        zero-fill, no partial-sector stores) and g.
 *      thx_hblocks_expand: the block list -> the dense (B, ld, ld) frame of thx_pg_assemble (caller zero-fills once) -- for
 *        consumers of Linearization.AtA; the optimiser never calls it.
 *      thx_hblocks_diag: diag(H) -> d (B, nvars * bd), row stride ldv  (Linearization.diagonal_scaling, LM's rho test).
 *      thx_chol_factor_hblocks: thx_chol_factor_forward / thx_chol_factor_sparse (pattern != NULL) reading H from the block
 *        list: every Cholesky tile gathers its pieces into LDS / adds them to its Schur update instead of streaming a dense
 *        tile of zeros.  rhs / y may both be NULL (no fused forward substitution).  ld = 0 (with a pattern): L is the
 *        TILE-PACKED factor (see thx_tile_pattern) -- neither H nor L is a dense frame then. */
typedef struct {
  int32_t nblocks, bd, nvars, ntiles;
  const int32_t* diag_blk;
  const int32_t* inc_blk;
  const int32_t* tile_ptr;
  const int32_t* piece_blk;
  const int32_t* piece_rc;
  int32_t max_tile_pieces; /* the largest number of pieces of one OFF-DIAGONAL tile (host: max of tile_ptr[t + 1] - tile_ptr[t] over
                            * i > j), 0 = unknown.  The factorisations pick how an off-diagonal tile takes its pieces of H by it: up
                            * to 64 (pose graphs: ~15 blocks per tile) they are ADDED to the tile's Schur update by the matrix cores
                            * (a rank-bd MFMA per piece, one barrier); above (a bundle adjustment's reduced camera system: up to
                            * 21 x 21 blocks per tile) or unknown they are gathered through LDS.  Same results either way. */
} thx_hblock_layout;
int thx_pg_assemble_blocks(const thx_pg_structure* s, const thx_pg_data* d, const thx_hblock_layout* layout, void* Hc,
                           int64_t bstride, void* g, int dtype, const thx_lie_eps* eps, void* stream);
int thx_hblocks_expand(const thx_hblock_layout* layout, const void* Hc, int64_t bstride, int32_t B, void* H, int64_t ld,
                       int dtype, void* stream);
int thx_hblocks_diag(const thx_hblock_layout* layout, const void* Hc, int64_t bstride, int32_t B, void* d, int64_t ldv,
                     int dtype, void* stream);

/* ---- Generic assembly for ANY cost function: the same H (lower triangle) and g as thx_pg_assemble, from
 *      per-cost weighted Jacobian blocks J (B, dim, dof) and weighted errors e (B, dim) supplied as tensors
 *      (what CostFunction.weighted_jacobians_error returns, core/cost_function.py:107-122); replaces
 *      DenseLinearization._linearize_jacobian_impl + _linearize_hessian_impl (dense_linearization.py:29-62)
 *      for objectives the fused pose-graph kernels do not cover.  All tables are DEVICE arrays built by the host:
 *        h_targets[t]: one non-zero block of tril(H) (row0 >= col0), with its CSR range of terms;
 *        h_terms[k]  : one J_a^T J_b contribution (device pointers, element batch strides; 0 = shared);
 *        h_elem2target[e]: owning target of output element e (elements of a target are contiguous from
 *                          elem_begin, row major dof_a x dof_b).
 *      Same for g with (J, e) terms.  Deterministic (no atomics), sums carried in fp64. */
typedef struct {
  int32_t row0, col0, dof_a, dof_b, term_begin, term_end, elem_begin, pad_;
} thx_block_target;
typedef struct {
  const void* Ja;
  const void* Jb;
  int64_t bstride_a, bstride_b;
  int32_t dim, dof_a, dof_b, pad_;
} thx_block_term;
typedef struct {
  int32_t col0, dof, term_begin, term_end, elem_begin, pad_;
} thx_grad_target;
typedef struct {
  const void* J;
  const void* e;
  int64_t bstride_j, bstride_e;
  int32_t dim, dof;
} thx_grad_term;
int thx_block_assemble(const thx_block_target* h_targets, const thx_block_term* h_terms, const int32_t* h_elem2target,
                       int32_t n_h_elems, const thx_grad_target* g_targets, const thx_grad_term* g_terms,
                       const int32_t* g_elem2target, int32_t n_g_elems, void* H, int64_t ld, void* g, int64_t ldg,
                       int32_t B, int dtype, void* stream);

/* ---- Objective.error_metric(): 0.5 * ||weighted error||^2 per problem (core/objective.py:37-38,
 *      562-641).  `partials` is a (B, THX_ERR_CHUNKS) scratch; the reduction order is fixed
 *      (deterministic).  err is (B). */
#define THX_ERR_CHUNKS 128
int thx_pg_error(const thx_pg_structure* s, const thx_pg_data* d, void* partials, void* err, int dtype,
                 const thx_lie_eps* eps, void* stream);

/* ---- Weighted residuals and Jacobian blocks (what weighted_jacobians_error() returns,
 *      core/cost_function.py:107-122), for `Linearization.A/.b/.Av` and tests:
 *      J0,J1 (E,B,6,6), eb (E,B,6), Jp (K,B,6,6), ep (K,B,6); any output may be NULL. */
int thx_pg_jacobians(const thx_pg_structure* s, const thx_pg_data* d, void* J0, void* J1, void* eb,
                     void* Jp, void* ep, int dtype, const thx_lie_eps* eps, void* stream);

/* ---- Objective.retract_vars_sequence (core/objective.py:873-914, core/vectorizer.py:410-469,
 *      core/variable.py:65-69): out[p,b] = ignore[b] ? poses[p,b] : poses[p,b] * exp(step * delta[b, 6p:6p+6]).
 *      delta is (B, n) with row stride ldd; ignore_mask is (B) uint8 or NULL. */
int thx_se3_retract(const void* poses, const void* delta, int64_t ldd, double step,
                    const uint8_t* ignore_mask, void* out, int32_t P, int32_t B, int dtype,
                    const thx_lie_eps* eps, void* stream);

/* ---- PER-CALL schedule of the factorisations (every thx_chol_factor* takes one as its last argument; NULL = the defaults).  The
 *      library keeps no mutable schedule state: two callers with different schedules in one process do not see each other.  The
 *      factor is the same either way (bit-identical: tests/test_gpu_kernels.py, tests/test_gpu_sparse.py).
 *        split_diag_min_batch: the diagonal phase of a block column runs either as ONE kernel (SYRK + the serial tile
 *          factorisation in the same workgroup) or SPLIT into an MFMA-only SYRK kernel and a one-wave-per-tile kernel that keeps
 *          the tile in registers (eight tiles per CU in their pivot chains instead of three).  The split pays from this many
 *          problems per launch on (the level schedule: problems x block columns of the level); 0 = always split, INT32_MAX =
 *          never; < 0: the default (2048, or the environment's THX_CHOL_SPLIT_DIAG_MIN read once at load time).
 *        column_pairs: fp32 factorisations on dense factor frames, two block columns at a time -- diag(j), tile (j+1, j),
 *          diag(j+1), then ONE workgroup per row tile i >= j+2 produces L_ij and L_i,j+1, streaming row panel L_i,0:j from HBM
 *          once for both; 1 on, 0 off, < 0: the default (on, or THX_CHOL_COLPAIR).
 *        right_looking_max_batch: factorisations (fp32 and fp64) on dense factor frames (no tile pattern, ld >= ntiles * THX_TILE) of at most
 *          this many problems take the RIGHT-LOOKING schedule -- per block column the tile factorisation, the substitutions and
 *          one workgroup per tile of the trailing matrix, each a single 128^3 product -- instead of the left-looking one whose
 *          serial K-loops leave the chip empty at 8 ... 64 problems (the reference's published batch range,
 *          evaluations/pose_graph_synthetic.sh:7).  Another summation order: the factor agrees with the left-looking one to
 *          rounding, not bit for bit.  0 = never; < 0: the default (THX_CHOL_RL_MAX_BATCH, else by dtype and size: fp32 64 problems and
 *          fp64 40 up to 12 block columns, shrinking to 32 from 24 block columns on -- min(64, max(32, 768 / ntiles)) resp.
 *          min(40, max(32, 480 / ntiles))).
 *        hb_scatter_max_pieces: block-compact H (thx_hblock_layout) whose off-diagonal tiles hold at most this many pieces
 *          (layout.max_tile_pieces) has them ADDED to the tile's Schur update by the matrix cores; above, they are gathered through
 *          LDS (see thx_hblock_layout.max_tile_pieces; the same bits either way).  0 = always gather; < 0: the default (64, or
 *          THX_HB_SCATTER_MAX_PIECES).
 *        f64_wide_max_ktiles: fp64 factorisations on the column-by-column schedule: the off-diagonal tiles of the first this many
 *          block columns (K-loops shorter than that many tiles) are produced by EIGHT-wave workgroups (16 rows of the tile per
 *          wave, four waves per SIMD) instead of four-wave ones -- the same arithmetic in the same order, bit-identical.  0 =
 *          never; < 0: the default (every column, or THX_F64_WIDE_MAX_KTILES).
 *        f64_half_max_ktiles: fp64, column-by-column schedule on a dense factor frame: the off-diagonal tiles of the first this many
 *          block columns are produced as two HALF tiles (64 rows each) by four-wave workgroups that need 37 KB of LDS and 128
 *          VGPRs -- four per CU instead of two; takes precedence over f64_wide_max_ktiles for those columns.  Bit-identical.
 *          0 = never; < 0: the default (8, or THX_F64_HALF_MAX_KTILES). */
typedef struct {
  int32_t split_diag_min_batch;
  int32_t column_pairs;
  int32_t right_looking_max_batch;
  int32_t hb_scatter_max_pieces;
  int32_t f64_wide_max_ktiles;
  int32_t f64_half_max_ktiles;
} thx_chol_schedule;

/* ---- tile-sparse Cholesky for LARGE pose graphs -- the functional analogue of BaspachoSparseSolver
 *      (theseus/optimizer/linear/baspacho_sparse_solver.py:23-148; symbolic analysis once, numeric factorisation per iteration).
 *      Same kernels and storage as thx_chol_factor[_forward] (dense row-major H / L frames), but only the 128x128 tiles of L
 *      that are STRUCTURALLY non-zero are computed, and every tile's K-loop visits only the block columns in which both of
 *      its row panels are non-zero: the work follows the fill of the (fill-reducing ordered) block pattern instead of n^3/3.
 *      Tiles outside the pattern are never written: L must be zero-initialised once.  The pattern is the host's symbolic
 *      factorisation at tile granularity (theseus_amd/sparse.py:tile_pattern); all tables int32, device pointers except
 *      col_count_host.  rhs / y may both be NULL (no fused forward substitution).  The solves are thx_chol_solve_sparse (or the dense-frame
 *      thx_chol_solve*). */
typedef struct {
  int32_t ntiles;                 /* ceil(n / THX_TILE) */
  const int32_t* col_ptr;         /* (ntiles + 1) off-diagonal non-zero tiles of block column j: entries [col_ptr[j], col_ptr[j+1]) */
  const int32_t* col_row;         /* (entries) row tile of every entry (any order within a column; the host sorts by K-list length) */
  const int32_t* tile_kptr;       /* (entries + 1) K-list of entry e: */
  const int32_t* tile_k;          /*   block columns k < j in which L_ik and L_jk are both non-zero */
  const int32_t* diag_kptr;       /* (ntiles + 1) K-list of diagonal tile j: */
  const int32_t* diag_k;          /*   block columns k < j in which L_jk is non-zero */
  const int32_t* col_count_host;  /* (ntiles) HOST copy of col_ptr[j+1] - col_ptr[j] (launch sizes) */
  const int32_t* row_ptr;         /* (ntiles + 1) the same pattern by ROWS, for the list-driven solves: the non-zero off-diagonal */
  const int32_t* row_tile;        /*   tiles of block row i are the block columns row_tile[row_ptr[i] .. row_ptr[i+1]) (< i) */
  /* TILE-PACKED factor (thx_chol_factor_hblocks / thx_chol_solve_sparse with ld = 0): L is (B, nslots, THX_TILE, THX_TILE), only
   * the tiles of the pattern exist -- slot j = diagonal tile j, slot ntiles + e = off-diagonal entry e (tile (col_row[e], j)) --
   * instead of a dense (B, ld, ld) frame that is mostly zeros (n = 12288: 604 MB per problem in fp32 against ~12 MB).  The
   * role of BaspachoSparseSolver's packed factor data (baspacho_sparse_solver.py:58-148).  Zero-initialise once. */
  const int32_t* tile_sa;         /* (per tile_k element) slot of L_jk, the K-loop's A operand tile */
  const int32_t* tile_sb;         /* (per tile_k element) slot of L_ik, its B operand tile */
  const int32_t* diag_s;          /* (per diag_k element) slot of L_jk */
  const int32_t* row_slot;        /* (per row_tile element) slot of that tile */
  int32_t nslots;                 /* ntiles + entries */
  /* LOOK-AHEAD (batches below the two-stream threshold): (ntiles) HOST flags, 1 = the FIRST entry of block column j is tile
   * (j + 1, j).  The factorisation then launches that tile on its own and starts the diagonal phase of column j + 1 under the rest
   * of column j (left-looking: diag(j + 1) needs only row j + 1 of the earlier columns).  NULL: no look-ahead. */
  const int32_t* col_head_host;
} thx_tile_pattern;
int thx_chol_factor_sparse(const void* H, int64_t ld, int32_t n, int32_t B, const void* damping, int ellipsoidal,
                           double damping_eps, void* L, void* Winv, int32_t* info, const void* rhs, void* y, int64_t ldv,
                           const thx_tile_pattern* pattern, int dtype, void* stream, const thx_chol_schedule* schedule);
/*      thx_chol_solve_sparse: x = (L L^T)^-1 rhs (backward_only != 0: x = L^-T rhs, the second half after the fused forward
 *      substitution of thx_chol_factor_sparse) streaming only the structurally non-zero tiles of L (row lists of the pattern)
 *      -- where the dense-frame solves read ntiles^2 / 2 tiles per problem, the dominant cost of a large sparse graph's
 *      iteration.  The working vector stays in global memory: no limit on n (the dense-frame thx_chol_solve* keep it in LDS).
 *      Bit-identical to thx_chol_solve* on the same factor.  x may alias rhs. */
int thx_chol_solve_sparse(const void* L, int64_t ld, int32_t n, int32_t B, const void* Winv, const void* rhs, void* x,
                          int64_t ldv, int backward_only, const thx_tile_pattern* pattern, int dtype, void* stream);

/* ---- LEVEL-SCHEDULED tile-sparse Cholesky: elimination-tree parallelism INSIDE a problem -- what a fill-reducing permutation
 *      gives the reference's BaSpaCho path (theseus/extlib/baspacho_solver.cpp:284-291 createSolver with a permutation, :332;
 *      theseus/optimizer/linear/baspacho_sparse_solver.py:58-148), in the regime it is used in (evaluations/
 *      pose_graph_synthetic.sh:5-12: batch 8-256, up to 4096 poses), where a banded ordering leaves a chain of ntiles dependent
 *      launch pairs with only the batch as parallelism.
 *      The host (theseus_amd/sparse.py:LevelPattern) orders the variables by NESTED DISSECTION at tile granularity, pads every tile
 *      to whole variables (tile j holds tile_valid[j] <= THX_TILE rows / columns of the matrix, the rest is identity padding: no
 *      variable straddles a tile boundary, so tiles in different subtrees are independent), numbers the block columns level by
 *      level of the tile elimination tree and sorts each level's off-diagonal entries longest K-list first.  Then
 *        thx_chol_factor_levels: per level ONE diagonal launch over (problems x block columns of the level) and ONE off-diagonal
 *          launch over (problems x entries of the level) -- same kernels as thx_chol_factor_hblocks.  H is the block list read
 *          through `layout`, whose tile_ptr / piece_* tables are built for the PADDED tiles; L is the tile-packed factor
 *          (B, nslots, THX_TILE, THX_TILE) of `pattern` (zero-initialised once), Winv (B, ntiles, THX_TILE, THX_TILE).  rhs / y
 *          (both NULL or both given, vectors of the PADDED order, y must not alias rhs): the forward substitution y = L^-1 rhs
 *          fused into the diagonal launches as in thx_chol_factor_forward -- a column keeps its K-list's blocks of y in LDS.
 *        thx_chol_solve_levels: which = 0: x = (L L^T)^-1 rhs, 1: x = L^-T rhs, 2: x = L^-1 rhs; one launch per level and direction, one
 *          workgroup per (problem, block row); rhs / x are vectors of the PADDED order (B, >= ntiles * THX_TILE), row stride ldv;
 *          x may alias rhs.
 *        thx_vec_gather: dst[b][k] = idx[k] >= 0 ? src[b][idx[k]] : 0 for k < n -- between the linearization's vectors (g, delta)
 *          and the padded ones (idx: device int32). */
typedef struct {
  int32_t nlevels;
  const int32_t* level_col_host;  /* (nlevels + 1) HOST: level l = block columns [level_col_host[l], level_col_host[l + 1]) */
  const int32_t* level_ent_host;  /* (nlevels + 1) HOST: ... and off-diagonal entries [level_ent_host[l], level_ent_host[l + 1]) */
  const int32_t* level_maxk_host; /* (nlevels) HOST: longest diagonal K-list (diag_kptr) among the level's block columns */
  const int32_t* ent_col;         /* DEVICE (entries): block column of entry e (col_ptr is not used by the level kernels) */
  const int32_t* tile_valid;      /* DEVICE (ntiles): rows / columns of tile j that are matrix (a multiple of the block size) */
  const int32_t* level_stream_host; /* (nlevels) HOST or NULL: launch stream of level l -- 0 the caller's, 1 the library's second
                                     * stream; + 4: both streams join in front of this level (the trunk of the elimination tree).
                                     * A "level" then is one SUBTREE's share of a tree level: subtrees of the tile elimination tree
                                     * do not see each other, so the two streams need no ordering against each other and one
                                     * chain's diagonal phases (one busy wave per workgroup) run beside the other's off-diagonal
                                     * tiles.  NULL: every level on the caller's stream.  Read by thx_chol_factor_levels only. */
} thx_level_schedule;
int thx_chol_factor_levels(const thx_hblock_layout* layout, const void* Hc, int64_t bstride, int32_t B, const void* damping,
                           int ellipsoidal, double damping_eps, void* L, void* Winv, int32_t* info, const void* rhs, void* y,
                           int64_t ldv, const thx_tile_pattern* pattern, const thx_level_schedule* schedule, int dtype,
                           void* stream, const thx_chol_schedule* chol_schedule);
int thx_chol_solve_levels(const void* L, int32_t B, const void* Winv, const void* rhs, void* x, int64_t ldv, int which,
                          const thx_tile_pattern* pattern, const thx_level_schedule* schedule, int dtype, void* stream);
int thx_vec_gather(const void* src, int64_t lds, void* dst, int64_t ldd, const int32_t* idx, int32_t n, int32_t B, int dtype,
                   void* stream);

/* ---- LinearSolver.solve(): replaces DenseSolver._apply_damping + CholeskyDenseSolver._solve_sytem
 *      (linear/dense_solver.py:38-64,159-161).
 *      thx_chol_factor: L L^T = H + damping (out of place: H stays undamped, as the reference
 *      requires for LM's rho test, levenberg_marquardt.py:183-190).
 *        damping: (B) per-problem lambda or NULL for plain Gauss-Newton;
 *        ellipsoidal != 0: H + diag(lambda * diag(H) + damping_eps), else H + lambda I;
 *        L: (B, ld, ld) lower factor;
 *        Winv: (B, ceil(n/THX_TILE), THX_TILE, THX_TILE) solve panels, one per diagonal tile of L:
 *          32x32 diagonal sub-blocks hold (L_ss)^-1, strictly-lower sub-blocks hold -L_st (kept for
 *          the solves, including the implicit-backward solve);
 *        info: (B) int32, 0 = ok, k>0 = leading minor k not positive definite (LAPACK potrf
 *        convention; the host turns any non-zero into the reference's RuntimeError).
 *      thx_chol_factor_forward: the same factorisation with the forward substitution fused in
 *        (torch.cholesky_solve's first half): y = L^-1 rhs, rhs/y (B, n) with row stride ldv, y must
 *        not alias rhs.  The panel rows the factorisation streams anyway are re-used, so y costs no
 *        extra pass over L.
 *      thx_chol_solve_backward: x = L^-T y (second half; x may alias y).
 *      thx_chol_solve: x = (L L^T)^-1 rhs with a cached factor, any rhs -- the backward pass solves
 *        with it (cf. optimizer/autograd/baspacho_sparse_autograd.py:117-168); x may alias rhs. */
int thx_chol_factor(const void* H, int64_t ld, int32_t n, int32_t B, const void* damping, int ellipsoidal,
                    double damping_eps, void* L, void* Winv, int32_t* info, int dtype, void* stream,
                    const thx_chol_schedule* schedule);
int thx_chol_factor_forward(const void* H, int64_t ld, int32_t n, int32_t B, const void* damping, int ellipsoidal,
                            double damping_eps, void* L, void* Winv, int32_t* info, const void* rhs, void* y,
                            int64_t ldv, int dtype, void* stream, const thx_chol_schedule* schedule);
int thx_chol_solve_backward(const void* L, int64_t ld, int32_t n, int32_t B, const void* Winv, const void* y,
                            void* x, int64_t ldv, int dtype, void* stream);
int thx_chol_solve(const void* L, int64_t ld, int32_t n, int32_t B, const void* Winv, const void* rhs,
                   void* x, int64_t ldv, int dtype, void* stream);
/*      (block-compact Hessian, see thx_hblock_layout) */
int thx_chol_factor_hblocks(const thx_hblock_layout* layout, const void* Hc, int64_t bstride, int32_t n, int32_t B,
                            const void* damping, int ellipsoidal, double damping_eps, void* L, int64_t ld, void* Winv,
                            int32_t* info, const void* rhs, void* y, int64_t ldv, const thx_tile_pattern* pattern, int dtype,
                            void* stream, const thx_chol_schedule* schedule);

/* ---- Implicit backward (BackwardMode.IMPLICIT, nonlinear/nonlinear_least_squares.py:121-135,265-292): the
 *      grad-enabled last step is X_new = X exp(step * delta), delta = H^-1 g(theta) with H detached
 *      (dense_linearization.py:61).  The backward pass is
 *        grad_delta = thx_se3_retract_vjp(grad_X_new)      (torchlie Exp/Compose backward, se3_impl.py:313-343,739-747)
 *        w          = thx_chol_solve(L, grad_delta)         (the backward linear solve, cached factor)
 *        grad_theta = thx_pg_vjp(w)                         (d(w^T A^T b)/d theta, H detached: what autograd does to
 *                                                            Between/Local jacobians + errors + cost weights)
 *      thx_pg_vjp outputs are per problem: grad_meas (E,B,3,4), grad_w_between (E,B,6), grad_prior_target
 *      (K,B,3,4), grad_w_prior (K,B,6), grad_log_radius_between (E,B) / grad_log_radius_prior (K,B) (robust costs
 *      only, may be NULL) -- the host reduces over broadcast dimensions; w is (B, n), row stride ldw.
 *      Gradients w.r.t. raw 3x4 entries follow torchlie's conventions (log: tangent-projected passthrough
 *      backward, se3_impl.py:487-493; inverse / compose / jlog: plain derivatives). */
int thx_se3_retract_vjp(const void* poses, const void* delta, int64_t ldd, double step, const void* grad_out,
                        void* grad_delta, int64_t ldg, int32_t P, int32_t B, int dtype, const thx_lie_eps* eps,
                        void* stream);
int thx_pg_vjp(const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw, void* grad_meas,
               void* grad_w_between, void* grad_prior_target, void* grad_w_prior, void* grad_log_radius_between,
               void* grad_log_radius_prior, int dtype, const thx_lie_eps* eps, void* stream);

/* ---- Differentiating THROUGH the iterations (BackwardMode.UNROLL / TRUNCATED, nonlinear_least_squares.py:223-292; the Hessian is
 *      part of the graph: dense_linearization.py:58-62 without the detach) on SE3 pose graphs.  One differentiated iteration is
 *      delta = (H + D)^-1 g,  X_new = X exp(step * delta).  Its backward, given grad_X_new (raw 3x4 entries):
 *        grad_delta = thx_se3_retract_vjp(grad_X_new);  grad_X += Compose.backward (plain: [G_R E_R^T + G_t E_t^T | G_t])
 *        w          = (H + D)^-1 grad_delta                 (thx_chol_solve with a COPY of that iteration's factor)
 *        thx_pg_unroll_vjp(w, delta): per cost, the gradient of  phi = -(J w) . (r + J delta)  -- i.e. of w^T g - w^T H delta with
 *        w, delta constant -- w.r.t. the raw entries of BOTH poses (grad_pose_i / grad_pose_j (E,B,3,4): the host adds them to
 *        the poses' gradients in a fixed order), of the measurement (E,B,3,4) and the weights (E,B,6); priors: grad_pose_prior /
 *        grad_prior_target (K,B,3,4), grad_w_prior (K,B,6).  Same autograd conventions as thx_pg_vjp (log: passthrough backward;
 *        inverse / compose / adjoint / Jlog: plain).  ``ellipsoidal_damping``: NULL for D = lambda I (constant), or the (B) vector
 *        lambda of D = lambda diag(H) + eps (dense_solver.py:38-64) -- phi then has the third term -lambda sum_i w_i delta_i
 *        H_ii, H_ii = the cost's sum of squared (weighted) Jacobian entries of column i.  RobustCostFunction (not detached,
 *        robust_cost_function.py:115-135): Phi = sum_r m_r phi_r with m_r = rho'(x_r) + eps, every loss code of thx_pg_data;
 *        grad_log_radius_between (E,B) / grad_log_radius_prior (K,B) as in thx_pg_vjp (may be NULL).
 *        w, delta: (B, n), row strides ldw / ldd. */
int thx_pg_unroll_vjp(const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw, const void* delta, int64_t ldd,
                      const void* ellipsoidal_damping, void* grad_pose_i, void* grad_pose_j, void* grad_meas, void* grad_w_between, void* grad_pose_prior,
                      void* grad_prior_target, void* grad_w_prior, void* grad_log_radius_between, void* grad_log_radius_prior, int dtype,
                      const thx_lie_eps* eps, void* stream);

/* The 3-dof twins (SE2: 4-element records [x, y, cos, sin], plain autograd everywhere, theseus/geometry/se2.py; SO3: 9-element
 * records, torchlie's passthrough backward for log, so3_impl.py:489-496): same arguments, 3-vectors for the weights. */
int thx_pg2_unroll_vjp(const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw, const void* delta, int64_t ldd,
                       const void* ellipsoidal_damping, void* grad_pose_i, void* grad_pose_j, void* grad_meas, void* grad_w_between,
                       void* grad_pose_prior, void* grad_prior_target, void* grad_w_prior, void* grad_log_radius_between,
                       void* grad_log_radius_prior, int dtype, const thx_se2_eps* eps, void* stream);
int thx_pgso3_unroll_vjp(const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw, const void* delta, int64_t ldd,
                         const void* ellipsoidal_damping, void* grad_pose_i, void* grad_pose_j, void* grad_meas,
                         void* grad_w_between, void* grad_pose_prior, void* grad_prior_target, void* grad_w_prior,
                         void* grad_log_radius_between, void* grad_log_radius_prior, int dtype, const thx_lie_eps* eps, void* stream);

/* ---- Bundle adjustment (BASELINE.json configs[3]; examples/bundle_adjustment.py:103-160): camera poses SE3 + Point3
 *      world points, costs = Reprojection (theseus/embodied/measurements/reprojection.py:54-94, dim 2; SE3.transform_from
 *      + Jacobians: torchlie/functional/se3_impl.py:757-777) optionally wrapped in RobustCostFunction, Difference priors on
 *      cameras (local_cost_fn.py:39-61) and on points (vector.py:150-178: e = x - target, J = I).
 *      The damped normal equations (H + D) delta = g of DenseLinearization + CholeskyDenseSolver are solved EXACTLY by
 *      block elimination of the points:  H = [[Hcc, Hcp],[Hpc, Hpp]] (Hcc, Hpp block diagonal 6x6 / 3x3),
 *        S = Hcc' - Hcp Hpp'^-1 Hpc,  rhs = gc - Hcp Hpp'^-1 gp,  S delta_c = rhs  (thx_chol_factor_forward, no damping),
 *        delta_p = Hpp'^-1 (gp - Hpc delta_c);   ' = damping applied (dense_solver.py:38-64).
 *      Internal column order: cameras (6 each) then points (3 each); g / delta / diag are (B, n), n = 6C + 3Np.
 *      Layouts: variables and auxiliary data entity major, batch next: cams (C,B,3,4), points (Np,B,3); gd (B,n).  The fp64
 *      block workspaces are PLANAR -- (entity, component, B), the batch index INNERMOST: Hcc (C,36,B) row-major 6x6 blocks,
 *      Hpp / Hinv (Np,6,B) = [xx,xy,xz,yy,yz,zz], W (O,18,B) = Jc^T Jp (6x3 row major) per observation, tvec (Np,3,B).  A
 *      lane of these kernels is one (entity, problem) and a wave is 64 consecutive problems of one entity: component i of the
 *      wave is 512 contiguous bytes, every load / store instruction moves full lines (with a record per lane -- (entity, B,
 *      component) -- a 16-byte load of the 144-byte W records touched 72 lines, each of them nine times).
 *      PRECISION: the block quantities Hcc, Hpp, W, gd, Hinv, tvec are ALWAYS fp64 buffers, whatever `dtype` is: the
 *      Schur complement subtracts quantities of the size of Hcc from Hcc, so fp32 blocks would put fp32 rounding of
 *      |Hcc| -- not of |S| -- into S.  S, rhs, g, diag, delta are `dtype` (S feeds the fp32 / fp64 MFMA Cholesky). */
typedef struct {
  int32_t num_cams, num_points, num_obs, num_cam_priors, num_pt_priors, num_pairs, num_blocks;
  const int32_t* obs_cam;       /* (O) */
  const int32_t* obs_pt;        /* (O) */
  const int32_t* pt_ptr;        /* (Np+1) CSR: observations of a point */
  const int32_t* pt_obs;        /* (O)    */
  const int32_t* cam_ptr;       /* (C+1)  CSR: observations of a camera */
  const int32_t* cam_obs;       /* (O)    */
  const int32_t* cam_prior_cam; /* (Kc) camera of prior k */
  const int32_t* cam_prior_ptr; /* (C+1)  CSR: priors of a camera */
  const int32_t* cam_prior_id;  /* (Kc)   */
  const int32_t* pt_prior_pt;   /* (Kp) */
  const int32_t* pt_prior_ptr;  /* (Np+1) */
  const int32_t* pt_prior_id;   /* (Kp)   */
  const int32_t* pair_ptr;      /* (C+1)  Schur pairs of camera c1: (o1, o2) share a point, cam(o1) = c1, cam(o2) <= c1, */
  const int32_t* pair_o1;       /* (Npairs)  sorted by cam(o2)                                                           */
  const int32_t* pair_o2;
  const int32_t* pair_c2;
  const int32_t* pair_dptr;     /* (C)  first pair of camera c1 with cam(o2) = c1 (its diagonal pairs run to pair_ptr[c1+1]) */
  const int32_t* blk_ptr;       /* (Nblocks, 2) [begin, end) pair range of every off-diagonal block (c1, c2 < c1) of S   */
  const int32_t* blk_c1;        /* (Nblocks) */
  const int32_t* blk_c2;        /* (Nblocks) */
} thx_ba_structure;

typedef struct {
  int32_t batch;
  const void* cams;    /* (C, B, 3, 4) */
  const void* points;  /* (Np, B, 3)   */
  const void* feat;    /* (O, Bf, 2) image_feature_point */
  int64_t feat_bstride;   /* 2 or 0 */
  const void* w_obs;   /* (O, Bw, 2) sqrt-information diagonal of the Reprojection cost weight */
  int64_t w_obs_bstride;
  const void* focal;   /* (C, Bc, 1) focal_length, calib_k1, calib_k2 of the camera */
  const void* k1;
  const void* k2;
  int64_t calib_bstride;  /* 1 or 0 (shared by the three) */
  int32_t robust_obs;  /* loss code (THX_LOSS_* [| THX_LOSS_FLATTEN]) of the Reprojection costs */
  const void* log_radius_obs;  /* (O, Br, 1) */
  int64_t log_radius_obs_bstride;
  const void* cam_prior_target;  /* (Kc, Bt, 3, 4) */
  int64_t cam_prior_target_bstride;
  const void* w_cam_prior;       /* (Kc, Bw, 6) */
  int64_t w_cam_prior_bstride;
  const void* pt_prior_target;   /* (Kp, Bt, 3) */
  int64_t pt_prior_target_bstride;
  const void* w_pt_prior;        /* (Kp, Bw, 3) */
  int64_t w_pt_prior_bstride;
} thx_ba_data;

/* implicit backward of a bundle-adjustment objective (nonlinear_least_squares.py:121-135,265-292; examples/bundle_adjustment.py:
 * 184-215 learns log_loss_radius through it): gradients of phi = w^T g(theta), w (B, ldw) = H^-1 grad_delta in the internal
 * column order [cameras | points], w.r.t. the image features (O,B,2), the Reprojection cost weights (O,B,2), the calibration
 * (PER OBSERVATION (O,B): the host sums the observations of a camera), log_loss_radius (O,B), the SE3 Difference priors' targets
 * (Kc,B,3,4) / weights (Kc,B,6) and the Point3 Difference priors' targets / weights (Kp,B,3).  Any output may be NULL
 * (grad_focal / grad_k1 / grad_k2 together). */
int thx_ba_vjp(const thx_ba_structure* s, const thx_ba_data* d, const void* w, int64_t ldw, void* grad_feat, void* grad_w_obs,
               void* grad_focal, void* grad_k1, void* grad_k2, void* grad_log_radius_obs, void* grad_cam_prior_target,
               void* grad_w_cam_prior, void* grad_pt_prior_target, void* grad_w_pt_prior, int dtype, const thx_lie_eps* eps,
               void* stream);

/* BackwardMode.UNROLL / TRUNCATED on a bundle-adjustment objective (nonlinear_least_squares.py:223-292: the Hessian is part of
 * the graph; thx_pg_unroll_vjp above has the derivation).  Per iteration, with w = (H + D)^-1 grad_delta (one Schur solve with that
 * iteration's factor) and delta that iteration's step, both (B, n) in the internal column order [cameras | points]: per cost, the
 * gradient of  phi = -(J w) . (r + J delta)  [- lambda sum_k w_k delta_k H_kk with ``ellipsoidal_damping`` = the (B) vector lambda
 * of D = lambda diag(H) + eps; NULL for D = lambda I / none]  w.r.t. the raw entries of the camera (O,B,3,4) and the point (O,B,3) of
 * every Reprojection cost, its feature (O,B,2), weights (O,B,2), calibration (PER OBSERVATION (O,B) each) and log_loss_radius (O,B;
 * may be NULL); the camera (Kc,B,3,4), target (Kc,B,3,4) and weights (Kc,B,6) of every SE3 Difference prior; the point, target
 * and weights (Kp,B,3 each) of every Point3 Difference prior.  The host sums cam_obs / cam_prior_cam over the costs of a camera
 * and pt_obs / pt_prior_pt over the costs of a point.  RobustCostFunction is part of the graph (not detached,
 * robust_cost_function.py:115-135).  Reference graph: reprojection.py:54-94 with SE3.transform_from's plain backward
 * (torchlie/functional/se3_impl.py:757-800). */
typedef struct {
  void *cam_obs, *pt_obs, *feat, *w_obs, *focal, *k1, *k2, *log_radius_obs;
  void *cam_prior_cam, *cam_prior_target, *w_cam_prior;
  void *pt_prior_pt, *pt_prior_target, *w_pt_prior;
} thx_ba_unroll_grads;
int thx_ba_unroll_vjp(const thx_ba_structure* s, const thx_ba_data* d, const void* w, int64_t ldw, const void* delta, int64_t ldd,
                      const void* ellipsoidal_damping, const thx_ba_unroll_grads* out, int dtype, const thx_lie_eps* eps,
                      void* stream);

/* linearize: Hcc, Hpp, W, gd (fp64) and g = [gc | gp], diag = diag(H) (dtype); row stride ldv for gd, g, diag */
int thx_ba_assemble(const thx_ba_structure* s, const thx_ba_data* d, void* Hcc, void* Hpp, void* W, void* gd, void* g,
                    void* diag, int64_t ldv, int dtype, const thx_lie_eps* eps, void* stream);
/* Schur complement with the damping of DenseSolver._apply_damping: writes the lower blocks of S (B, ld, ld) (zero-filled
 * once by the caller: fixed pattern), rhs (B, 6C) (row stride ldr), Hinv, tvec; info[b] != 0 if a damped point block is
 * not positive definite. */
int thx_ba_schur(const thx_ba_structure* s, int32_t B, const void* Hcc, const void* Hpp, const void* W, const void* gd,
                 int64_t ldv, const void* damping, int ellipsoidal, double damping_eps, void* S, int64_t ld, void* rhs,
                 int64_t ldr, void* Hinv, void* tvec, int32_t* info, int dtype, void* stream);
/* thx_ba_schur with S written as a BLOCK LIST -- the values of a thx_hblock_layout with bd = 6, (B, bstride), 36 contiguous
 * elements per block and problem (bstride a multiple of 4, >= 36 (C + num_blocks)) -- instead of a dense frame: what
 * thx_chol_factor_levels reads, so that the reduced camera system is factorised along ITS elimination tree (the analogue of
 * BaSpaCho's fill-reducing permutation of the camera block, theseus/extlib/baspacho_solver.cpp:284-291,332; the reference's
 * Schur complement: theseus/optimizer/linear/baspacho_sparse_solver.py:58-148 eliminates the points the same way).  Block
 * diag_blk[c] receives S_cc (damped), block (blk_dst[k] & 0x3fffffff) the block of the structure's k-th camera pair
 * (blk_c1[k], blk_c2[k]) -- TRANSPOSED where bit 30 of blk_dst[k] is set: the list holds tril(S) in the SOLVER's camera order,
 * which may put c2 behind c1.  A lane writes its block as 144 (fp32) / 288 (fp64) contiguous bytes; rhs, Hinv, tvec, info as
 * thx_ba_schur (rhs in the structure's camera order: the solver gathers it into its padded order with thx_vec_gather). */
int thx_ba_schur_blocks(const thx_ba_structure* s, int32_t B, const void* Hcc, const void* Hpp, const void* W, const void* gd,
                        int64_t ldv, const void* damping, int ellipsoidal, double damping_eps, void* Sc, int64_t bstride,
                        const int32_t* diag_blk, const int32_t* blk_dst, void* rhs, int64_t ldr, void* Hinv, void* tvec,
                        int32_t* info, int dtype, void* stream);
/* delta_p = tvec - Hinv Hpc delta_c, written to delta[:, 6C:] (delta_c = delta[:, :6C] is read) */
int thx_ba_backsub(const thx_ba_structure* s, int32_t B, const void* W, const void* Hinv, const void* tvec, void* delta,
                   int64_t ldv, int dtype, void* stream);
/* partials: scratch (THX_BA_ERR_CHUNKS, B) */
#define THX_BA_ERR_CHUNKS 256
int thx_ba_error(const thx_ba_structure* s, const thx_ba_data* d, void* partials, void* err, int dtype,
                 const thx_lie_eps* eps, void* stream);
/* Linearization.Av (theseus/optimizer/dense_linearization.py:73-74; read by Dogleg / TrustRegion, nonlinear/dogleg.py:66,
 * trust_region.py:97) for the block linearization: out_t (m, B) = (A v)^T with the weighted (and robust-rescaled) Jacobian
 * blocks recomputed per cost, never the dense A.  v (B, ldv) in the linearization's column order (cameras, then points); rows
 * in cost ADD order: obs_row / cam_prior_row / pt_prior_row (device int32) give the first row of every cost. */
int thx_ba_av(const thx_ba_structure* s, const thx_ba_data* d, const void* v, int64_t ldv, const int32_t* obs_row,
              const int32_t* cam_prior_row, const int32_t* pt_prior_row, void* out_t, int dtype, const thx_lie_eps* eps,
              void* stream);
/* Vector retraction x <- x + step * delta (theseus/geometry/vector.py:177-178), masked like thx_se3_retract; x (N,B,dof),
 * delta (B, ldd) read at columns [col0, col0 + N*dof) */
int thx_vec_retract(const void* x, const void* delta, int64_t ldd, int64_t col0, double step, const uint8_t* ignore_mask,
                    void* out, int32_t N, int32_t dof, int32_t B, int dtype, void* stream);
/* thx_lm_accept with diag(H) given as a (B, n) vector (row stride ldv) instead of the dense H */
int thx_lm_accept_diag(const void* delta, const void* g, const void* diag, int64_t ldv, int32_t n, int32_t B, void* damping,
                       const void* prev_err, const void* new_err, int ellipsoidal, double accept, double down_ratio,
                       double up_ratio, uint8_t* reject, int dtype, void* stream);

/* ---- Per-problem selection on entity-major state buffers (N, B, record): dst[k, b, :] = src[k, b, :] where mask[b] != 0.
 *      The `torch.where(mask, new, old)` of Variable.update(batch_ignore_mask) (theseus/core/variable.py:65-69) and of the
 *      rejected-step / best-solution bookkeeping (nonlinear_least_squares.py:320, nonlinear_optimizer.py:160-172) applied to
 *      the packed state in place: no temporary, no allocation.  record_bytes must be a multiple of 4. */
int thx_copy_where(const uint8_t* mask, const void* src, void* dst, int64_t N, int32_t B, int32_t record_bytes, void* stream);

/* ---- Linearization.diagonal_scaling support: d[b, i] = H[b, i, i] (linearization.py:85-87). */
int thx_diag(const void* H, int64_t ld, int32_t n, int32_t B, void* d, int64_t ldv, int dtype, void* stream);

/* ---- LevenbergMarquardt._check_accept (nonlinear/levenberg_marquardt.py:173-201), fused:
 *      den = sum_j delta_j (lambda D_j delta_j + g_j) / 2, rho = (prev_err - new_err) / den,
 *      reject = rho <= accept; lambda <- clamp(reject ? lambda*up : lambda/down, 1e-7, 1e7).
 *      D = diag(H) if ellipsoidal else 1.  delta is the *scaled* step (delta * step_size).
 *      Outputs reject (B) uint8 and updates damping (B) in place. */
int thx_lm_accept(const void* delta, const void* g, int64_t ldv, const void* H, int64_t ld, int32_t n,
                  int32_t B, void* damping, const void* prev_err, const void* new_err, int ellipsoidal,
                  double accept, double down_ratio, double up_ratio, uint8_t* reject, int dtype,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* THESEUS_HIP_H_ */
