/* omc.h -- C ABI of the MI355X node-relaxation engine (libomc_hip.so).
 *
 * Drop-in boundary for the per-B&B-node hot path of OptimalMatrixCompletion.jl.  The reference has no FFI
 * today; the seam is four plain-Julia call sites in the driver (OMC.jl = /root/reference/src/
 * OptimalMatrixCompletion.jl):
 *     relaxation            OMC.jl:747-754   -> matrix_completion_SDP_relaxation   OMC.jl:1431-1943
 *     master feasibility    OMC.jl:814       -> matrix_completion_master_feasible  OMC.jl:1261-1277
 *     altmin                OMC.jl:540-546, 875-881 -> alternating_minimization    OMC.jl:1979-2279
 *     separation            OMC.jl:971-983   -> create_matrix_cut_child_nodes      OMC.jl:2466-2477
 *     objective             OMC.jl:565, 925  -> evaluate_objective                 OMC.jl:2330-2359
 * Each entry point below names the reference interface it replaces.  INTEGRATION.md shows the Julia
 * `ccall` shim a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every matrix is COLUMN-MAJOR fp64 (Julia `Matrix{Float64}` layout);
 *   - the caller owns every input and output array; the library keeps no host pointer after a call returns;
 *   - device copies of A / mask live in the handle (they are constant for a whole B&B run, OMC.jl:470-471);
 *   - return value: 0 = ok; < 0 = invalid argument (the reference's `error(...)` checks, OMC.jl:1455-1477,
 *     2337-2348, 2433-2462); > 0 = HIP runtime error code.  Message via omc_last_error().  Nothing unwinds
 *     across the boundary.
 *   - safe to call from one host thread per handle; the batch entry point is the source of parallelism.
 */
#ifndef OMC_H
#define OMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMC_VERSION 100

/* termination status, mapped on the MOI statuses the driver branches on (OMC.jl:780-785, 809-812, 841) */
#define OMC_OPTIMAL 0        /* MOI.OPTIMAL / LOCALLY_SOLVED: gap and feasibility tolerances met        */
#define OMC_SLOW_PROGRESS 1  /* MOI.SLOW_PROGRESS: iteration cap reached, values available              */
#define OMC_TIME_LIMIT 2     /* MOI.TIME_LIMIT: time limit reached, values available                    */
#define OMC_INFEASIBLE 3     /* MOI.INFEASIBLE family: "feasible" = false (OMC.jl:1921-1935)            */

/* disjunctive_cuts_type (OMC.jl:150, 1456) */
#define OMC_CUT_LINEAR 0
#define OMC_CUT_LINEAR2 1
#define OMC_CUT_LINEAR3 2
/* direction strings of a cut (OMC.jl:34, 2481-2491) */
#define OMC_DIR_LEFT 0
#define OMC_DIR_MIDDLE 1
#define OMC_DIR_RIGHT 2
#define OMC_DIR_INNER_LEFT 3
#define OMC_DIR_INNER_RIGHT 4
/* disjunctive_cuts_breakpoints (OMC.jl:151, 2466-2477) */
#define OMC_SMALLEST_1_EIGVEC 1
#define OMC_SMALLEST_2_EIGVEC 2

/* error classes (negative return values) */
#define OMC_ERR_INVALID_ENUM (-1)   /* OMC.jl:1456-1462, 2433-2446 */
#define OMC_ERR_DIMENSION (-2)      /* OMC.jl:240-254, 1465-1477, 2337-2348 */
#define OMC_ERR_ARGUMENT (-3)
#define OMC_ERR_UNSUPPORTED (-4)    /* valid in the reference, not built yet (see DESIGN.md scope table) */
#define OMC_ERR_NO_DEVICE (-5)
#define OMC_ERR_COMM (-6)           /* RCCL not loadable / communicator not initialised / a collective failed */

typedef struct omc_instance omc_instance; /* opaque handle: device copies of A, mask, index lists, workspaces */

/* solver parameters of the relaxation (all have defaults via omc_relax_params_default) */
typedef struct omc_relax_params {
  double eps_gap;      /* stop when objective - dual_bound <= eps_gap * max(1,|objective|)   (1e-6)  */
  double eps_feas;     /* and cone residual <= eps_feas * sqrt(n+k)                          (1e-7)  */
  int max_iters;       /* iteration cap -> OMC_SLOW_PROGRESS                                 (3000)  */
  int check_every;     /* certificate evaluated every this many iterations                   (25)    */
  double rho_scale;    /* penalty = rho_scale * gamma/2 ||A_Omega||^2 / (m (1+gamma k/n)^2)  (1.0)   */
  double rho_f_ratio;  /* penalty of the per-column blocks relative to the cone blocks       (0.1)   */
  double relax;        /* over-relaxation                                                    (1.6)   */
  double time_limit;   /* seconds (OMC.jl:752, 1489)                                         (3600)  */
  int reference_quirk_q1; /* 1: linear3/right piece exactly as OMC.jl:1675; 0: secant        (1)     */
  int breakpoints;     /* OMC_SMALLEST_1_EIGVEC / _2_ : which separation vector to return    (1)     */
  int stall_checks;    /* stop with OMC_SLOW_PROGRESS after this many stationary checks      (8)     */
  int bump_max;        /* penalty bumps per node (0 = off): rho *= bump_factor when the primal   (2)     */
  double bump_ratio;   /*   residual exceeds bump_ratio x the dual residual at a check ...      (4.0)   */
  double bump_factor;  /*                                                                       (4.0)   */
  int bump_after;      /*   ... from this iteration on, at least bump_window checks apart       (100)   */
  int bump_window;     /*                                                                       (4)     */
  int slots;           /* nodes relaxed concurrently (continuous batching); 0 = min(B, 256)       (0)     */
  int accel;           /* 1: Anderson acceleration of the ADMM fixed-point map (type II, safeguarded) (0)  */
  int aa_mem;          /*   differences kept (<= 10)                                                 (10)    */
  int aa_every;        /*   an extrapolated point every this many iterations                         (10)    */
  int aa_start;        /*   first iteration that records history                                     (50)    */
  double aa_reg;       /*   Tikhonov weight of the least squares, relative to mean diag              (1e-10) */
  double aa_safeguard; /*   the point is kept when the next fixed-point residual <= this x the last  (1.0)   */
  int first_wins;      /* 1: the batch ends at the first check at which some node is OPTIMAL; the others are
                          returned as they stand (SLOW_PROGRESS, values available).  Penalty autotune.  (0)     */
  int early_stop_after;     /* a node whose gap, at the geometric rate it closed over the last checks, needs more than       */
  double early_stop_factor; /*   early_stop_factor x the checks left before max_iters is returned as OMC_SLOW_PROGRESS at once
                                 (eight consecutive predictions, from iteration early_stop_after on); 0 = off   (400, 1.5) */
} omc_relax_params;

void omc_relax_params_default(omc_relax_params* p);

const char* omc_last_error(void);
int omc_version(void);
int omc_device_count(void);

/* Upload (A, indices, gamma) once.  Replaces nothing in the reference (it passes A/indices to every call,
 * OMC.jl:747-754); sizes are checked as OMC.jl:240-254 does.  `mask` is n*m bytes (0/1), column-major.   */
int omc_instance_create(int n, int m, int k, const double* A, const uint8_t* mask, double gamma, int device,
                        omc_instance** out);
/* Same, with Julia's BitMatrix storage: packed UInt64 chunks, column-major bit order, LSB first. */
int omc_instance_create_bits(int n, int m, int k, const double* A, const uint64_t* chunks, double gamma,
                             int device, omc_instance** out);
void omc_instance_destroy(omc_instance* h);

/* ---- relaxation: matrix_completion_SDP_relaxation (OMC.jl:1431-1943), disjunctive mode -----------------
 * B nodes at once.  Node b carries L[b] cuts; cuts of all nodes are concatenated:
 *   cut_x    n      doubles per cut   (breakpoint vector,            OMC.jl:34 tuple element 1)
 *   cut_Uhat n*k    doubles per cut   (parent's U, column-major,     tuple element 2)
 *   cut_dir  k      int8 per cut      (OMC_DIR_* codes,              tuple element 3)
 * U_lower / U_upper: n*k doubles per node, or NULL for the reference defaults (OMC.jl:1442-1449).
 * Outputs (any matrix pointer may be NULL to skip the copy):
 *   objective[b]   recomputed from primal values as OMC.jl:1880-1896 does
 *   dual_bound[b]  certified lower bound on the relaxation optimum (reference quirk Q2: the reference uses
 *                  the primal value as node bound; this engine returns both)
 *   status[b], iters[b]
 *   Y n*n, U n*k, X n*m, Theta m*m per node  (OMC.jl:1897-1900)
 *   lambda_min[2*b..]: two smallest eigenvalues of U U' - Y      (OMC.jl:1274, 2467-2470)
 *   breakpoint_x n per node: separation vector per params->breakpoints (OMC.jl:2466-2477), sign fixed so that
 *                  its largest-magnitude entry is positive
 *   solve_time[b]: seconds of device time attributed to the batch (same value for every node of the batch)
 */
int omc_relax_batch(omc_instance* h, int B, const omc_relax_params* params, int cut_type, const int* L,
                    const double* cut_x, const double* cut_Uhat, const int8_t* cut_dir, const double* U_lower,
                    const double* U_upper, double* objective, double* dual_bound, int* status, int* iters,
                    double* Y, double* U, double* X, double* Theta, double* lambda_min, double* breakpoint_x,
                    double* solve_time);

/* The same call split in three so that a caller (bench.py) can time the device-resident part alone:
 * stage = host->device of the node descriptors, solve = kernels only, fetch = device->host of results.   */
int omc_relax_stage(omc_instance* h, int B, const omc_relax_params* params, int cut_type, const int* L,
                    const double* cut_x, const double* cut_Uhat, const int8_t* cut_dir, const double* U_lower,
                    const double* U_upper);
/* Optional: per-node penalty scales for the NEXT omc_relax_stage / omc_relax_batch with the same B (consumed by it).
 * Lets one batch try several penalties on the root (autotune) or give children their parent's value. */
int omc_set_node_rho_scales(omc_instance* h, int B, const double* rho_scale);
int omc_relax_solve(omc_instance* h);
/* ---- warm start from the parent's state (an addition of the engine; the reference cold-starts every model, OMC.jl:1482) ----------------------
 * A child is its parent plus one cut.  omc_state_pool_create reserves `capacity` final states on the device (about 24 n^2 + 8 (n k + |Omega| + m
 * + 16 n) bytes each: 0.27 MB at 100 x 100).  omc_relax_set_warm applies to the NEXT omc_relax_stage / omc_relax_batch with the same B and is
 * consumed by it: node b starts from pool entry load_from[b] (-1 or NULL: cold start) and its final state is stored in entry save_to[b]
 * (-1 or NULL: not stored).  The caller owns the numbering (e.g. a ring).  A warm node keeps its own base penalty (the parent's scaled
 * duals are rescaled), inherits Y, the duals of the two full cones, the column multipliers and the tracked block of the cone; the result is the
 * same convex program's optimum (same certificate), reached in fewer iterations.  Not available in Shor mode. */
int omc_state_pool_create(omc_instance* h, int capacity);
int omc_relax_set_warm(omc_instance* h, int B, const int* load_from, const int* save_to);
/* Asynchronous form of omc_relax_solve: submit returns at once (the solve runs on a worker thread of the library), poll reports
 * progress (running flag, nodes harvested so far, nodes staged), wait joins and returns the solve's return code (message via
 * omc_last_error on the waiting thread).  The reference's loop is serial (OMC.jl:700-719); with this the host can prepare the next
 * batch -- pop, prune (OMC.jl:1220-1244), build children -- while the device relaxes the current one.  One solve in flight per handle;
 * between submit and wait only omc_relax_poll may be called on the handle. */
int omc_relax_submit(omc_instance* h);
/* Appending nodes to a staged or RUNNING batch: the reference's loop pops from a queue that the children of relaxed nodes keep filling
 * (OMC.jl:700-719, 2520-2542); a staged batch was closed until round 3.  omc_relax_reserve(h, extra_nodes, max_cuts) makes the NEXT
 * omc_relax_stage size its per-node arrays for extra_nodes more nodes with at most max_cuts cuts each (default U bounds, same cut type and
 * parameters as the staged batch).  omc_relax_append(h, B, L, cut_x, cut_Uhat, cut_dir, load_from, save_to) adds B nodes in the wire format of
 * omc_relax_stage (load_from / save_to: warm-start pool entries as in omc_relax_set_warm, or NULL): before the solve starts, or while a
 * submitted solve is running -- the loop hands them to free slots at its next check -- and is refused once that solve has ended (the end is
 * decided under the same lock, so a node is either relaxed or refused, never lost).  Results: omc_relax_fetch returns every node, appended
 * ones behind the staged ones in the order they were appended.  Not available in Shor mode. */
/* omc_relax_fetch_done: results of the nodes finished since the last call, in the order they finished -- callable while the submitted solve is
 * running (the other half of a queue-driven host loop: OMC.jl:700-719 pops, relaxes and pushes the children of one node at a time).
 * node_ids[i] indexes the staged + appended nodes; per node: U (n*k), lambda_min (2), breakpoint_x (n), Y (n*n) as in omc_relax_fetch (NULL: skipped). */
int omc_relax_fetch_done(omc_instance* h, int max_nodes, int* node_ids, double* objective, double* dual_bound, int* status, int* iters,
                         double* U, double* lambda_min, double* breakpoint_x, double* Y, int* n_out);
/* omc_relax_hold(h, 1) after staging: the submitted solve waits for omc_relax_append when it runs dry instead of ending (until omc_relax_hold(h, 0)
 * or the time limit of the parameters). */
int omc_relax_hold(omc_instance* h, int on);
int omc_relax_reserve(omc_instance* h, int extra_nodes, int max_cuts);
int omc_relax_append(omc_instance* h, int B, const int* L, const double* cut_x, const double* cut_Uhat, const int8_t* cut_dir,
                     const int* load_from, const int* save_to);
int omc_relax_poll(omc_instance* h, int* running, int* nodes_done, int* nodes_total);
int omc_relax_wait(omc_instance* h);
int omc_relax_fetch(omc_instance* h, double* objective, double* dual_bound, int* status, int* iters, double* Y,
                    double* U, double* X, double* Theta, double* lambda_min, double* breakpoint_x,
                    double* solve_time);

/* ---- relaxation with add_Shor_valid_inequalities = true (rank k = 1): OMC.jl:1431-1453 (keyword), 1503-1525 (W, V1, V2, V3),
 * 1755-1779 (rotated cones, Theta_jj = sum_i W_ij, one order-5 PSD block per minor), 1838-1846 (objective), called at OMC.jl:747-754 with
 * node.Shor_info = BBNodeShorInfo(constraints_indexes, SOC_constraints_indexes) (OMC.jl:37-40).
 *   n_shor[b]   number of minors of node b; shor_idx: their (i1, i2, j1, j2) tuples, 1-based Int64, 4 per minor, all nodes concatenated
 *               (node.Shor_info.constraints_indexes as Julia stores a Vector{NTuple{4,Int}});
 *   n_soc[b]    number of SOC coordinates of node b; soc_idx: (i, j) 1-based Int64 pairs concatenated (SOC_constraints_indexes);
 *               n_soc[b] = -1 (then the node contributes nothing to soc_idx) means "every coordinate that occurs in none of the node's
 *               minors" -- what the reference's driver always passes (OMC.jl:656-673, 2508-2517) -- without shipping n*m pairs per node.
 * Nodes with identical lists share one index structure on the device (the static mode hands every node the same list).
 * Everything else as omc_relax_stage; then omc_relax_solve / omc_relax_submit / _wait and omc_relax_fetch as usual (X and Theta are
 * the explicit variables of the Shor program), plus omc_relax_fetch_shor for W (OMC.jl:1908) and omc_relax_fetch_shor_V for V1, V2, V3.  Status, objective (recomputed as OMC.jl:1960-1967 does) and the certified dual bound as in the base mode.
 * Rank k > 1 (Xt, Wt, H, per-layer blocks, one order-(k+1) block per coordinate: OMC.jl:1491-1494, 1526-1551, 1780-1827): reference quirk Q5 -- the slack
 * that H cancels in W = sum Wt + 2 sum H makes every per-layer order-5 block satisfiable, so in that form the minors do not constrain (X, W); the
 * program has the value of the same program without its order-5 blocks and with W >= X^2 kept on their coordinates.  That program is what is solved
 * (same outputs); the lifted variables Xt, Wt, H, V1..V3 of the reference's result are an explicit extension of (X, W) (closed form, INTEGRATION.md;
 * the host mirror api.py builds it), checked against the reference's full constraint set in the tests. */
int omc_relax_stage_shor(omc_instance* h, int B, const omc_relax_params* params, int cut_type, const int* L,
                         const double* cut_x, const double* cut_Uhat, const int8_t* cut_dir, const double* U_lower,
                         const double* U_upper, const int64_t* n_shor, const int64_t* shor_idx, const int64_t* n_soc,
                         const int64_t* soc_idx);
int omc_relax_fetch_shor(omc_instance* h, double* W /* n*m per node, may be NULL */);
/* The lifted products the order-5 blocks use (results["V1"], ["V2"], ["V3"], OMC.jl:1909-1911, restricted to the entries that occur in a
 * block): 5 doubles per minor, in the node's minor order -- V1[i1,(j1,j2)], V1[i2,(j1,j2)], V2[(i1,i2),j1], V2[(i1,i2),j2],
 * V3[(i1,i2),(j1,j2)] -- nodes concatenated with stride 5 * max_b n_shor[b].  Kept only when requested BEFORE staging
 * (omc_set_shor_keep_V(h, 1): the copy costs 40 bytes per minor and node). */
int omc_set_shor_keep_V(omc_instance* h, int keep);
int omc_relax_fetch_shor_V(omc_instance* h, double* V);
int omc_relax_batch_shor(omc_instance* h, int B, const omc_relax_params* params, int cut_type, const int* L,
                         const double* cut_x, const double* cut_Uhat, const int8_t* cut_dir, const double* U_lower,
                         const double* U_upper, const int64_t* n_shor, const int64_t* shor_idx, const int64_t* n_soc,
                         const int64_t* soc_idx, double* objective, double* dual_bound, int* status, int* iters, double* Y,
                         double* U, double* X, double* Theta, double* W, double* lambda_min, double* breakpoint_x,
                         double* solve_time);
/* penalties of the Shor-mode splitting, in the scaled variables (defaults 0.05, 0 = automatic per list: 75 n m / (4 n_shor) clamped to
 * [0.25, 40], 2; params->rho_scale multiplies rho) */
int omc_set_shor_penalties(omc_instance* h, double rho, double r4, double r5);

/* ---- alternating_minimization (OMC.jl:1979-2279), disjunctive mode, B problems at once, rank k <= 4 --------
 * U_initial n*k per problem; cuts as above (only the per-cut bounds on v = U'x are imposed, OMC.jl:2047-2093);
 * k > 1 adds the pair cones ||U_j1 +- U_j2|| <= sqrt 2 of OMC.jl:2029-2045.
 * Outputs: U n*k, V k*m, converged, n_iters, objectives (max_iters doubles per problem, NaN padded).     */
int omc_altmin_batch(omc_instance* h, int B, int cut_type, int reference_quirk_q1, const int* L,
                     const double* cut_x, const double* cut_Uhat, const int8_t* cut_dir, const double* U_initial,
                     double eps, int max_iters, double time_limit, double* U, double* V, int* converged,
                     int* n_iters, double* objectives, double* solve_time);

/* evaluate_objective(U * V) of the LAST omc_altmin_batch call (OMC.jl:920 `X_local = U * V`, 925-927), computed on the device from the
 * factors inside the altmin kernel: the driver compares it with the incumbent without forming X; X is built for the winner only. */
int omc_altmin_master_objectives(omc_instance* h, int B, double* objective);

/* ---- evaluate_objective (OMC.jl:2330-2359) for B matrices X (n*m each) --------------------------------- */
int omc_evaluate_objective(omc_instance* h, int B, const double* X, double* objective);

/* ---- separation / feasibility on caller-supplied (Y, U): OMC.jl:1272-1277 and 2466-2477 -----------------
 * eigvals[2*b..] two smallest eigenvalues of U U' - Y; x n per problem; feasible[b] = eigvals[0] >= -1e-6 */
int omc_separation_batch(omc_instance* h, int B, int breakpoints, const double* Y, const double* U,
                         double* eigvals, double* x, int* feasible);

/* ---- rounding glue: svd(M).U[:,1:k] of the symmetric PSD Y (OMC.jl:873) -------------------------------- */
int omc_round_Y_batch(omc_instance* h, int B, const double* Y, double* U_rounded);
/* svd(X).U[:, 1:k] for B matrices X (n x m, column-major): the rank-k rounding of an incumbent or of the zero-filled A
 * (OMC.jl:524, 564, 921).  Gram product X X' on the matrix cores, then the k dominant eigenvectors; sign as omc_round_Y_batch. */
int omc_left_singular_batch(omc_instance* h, int B, const double* X, double* U_out);

/* ---- Shor minors ------------------------------------------------------------------------------------------
 * generate_rank1_matrix_completion_Shor_constraints_indexes (OMC.jl:2545-2612): the 2 x 2 minors (i1 < i2, j1 < j2) whose
 * four cells hold exactly p observed entries, for every p of `num_entries_present` in turn (values outside 0..4
 * contribute nothing, as in the reference), in the reference's push order; tuples are 1-based Int64, 4 per minor.
 *   omc_shor_count   : number of minors per list element.
 *   omc_shor_indexes : *count = total; the tuples are written only when out != NULL and capacity >= total.           */
int omc_shor_count(omc_instance* h, int n_classes, const int* num_entries_present, int64_t* count_per_class);
int omc_shor_indexes(omc_instance* h, int n_classes, const int* num_entries_present, int64_t capacity, int64_t* out,
                     int64_t* count);
/* generate_violated_Shor_minors (OMC.jl:2614-2640).  X is the reference's Array{Float64,3} of size (k, n, m)
 * (X[t,i,j] at t + k*(i + n*j)); `existing` = 4*n_existing Int64 (1-based) already imposed minors (setdiff!, OMC.jl:2626).
 * Output: the min(n_minors, #candidates) minors with the largest score sum_t |X[t,i1,j1] X[t,i2,j2] - X[t,i1,j2] X[t,i2,j1]|,
 * ordered as Julia orders (score, tuple) pairs with rev = true (ties: larger tuple first).                           */
int omc_violated_shor_minors(omc_instance* h, const double* X, int n_classes, const int* num_entries_present,
                             int64_t n_existing, const int64_t* existing, int n_minors, double* scores,
                             int64_t* minors, int* n_out);
/* device milliseconds and candidate count of the last omc_shor_indexes / omc_violated_shor_minors call */
int omc_shor_last_stats(omc_instance* h, double* ms, int64_t* candidates);

/* ---- multi-GPU: node-parallel B&B, one process per GPU ------------------------------------------------------
 * Nodes are independent given (A, indices, gamma): every rank holds its own omc_instance on its own device and relaxes its shard
 * of the popped nodes.  The only exchange of the loop (OMC.jl:700-1073: tree.best_upper_bound at 725 / 797 / 1225, the global
 * lower bound at 1207-1218) is a 16-byte MIN all-reduce of {incumbent upper bound, smallest open lower bound} per round, and --
 * only when the incumbent improved -- a broadcast of the new X from the rank that found it.  RCCL (librccl, loaded on first use)
 * over xGMI; the communicator lives in the handle.
 *   omc_comm_unique_id   : rank 0 creates the 128-byte id; the host language distributes it to the other ranks (any channel).
 *   omc_comm_init        : collective over all ranks, after every rank has created its instance on its device.
 *   omc_allreduce_bounds : in place; *owner (may be NULL) = smallest rank whose local ub equals the global minimum.
 *   omc_bcast_incumbent  : X (n*m, column-major) from rank `root` to every rank.
 *   omc_allgather_records: the small per-node records (status, objective, bound, eigenvalues, breakpoint vector, U) of every rank.       */
#define OMC_COMM_ID_BYTES 128
int omc_comm_unique_id(void* id_out);
int omc_comm_init(omc_instance* h, int rank, int world_size, const void* id);
int omc_allreduce_bounds(omc_instance* h, double* ub, double* lb, int* owner);
int omc_bcast_incumbent(omc_instance* h, int root, double* X);
/* all-gather of the per-node records a host driver needs on every rank to grow the same tree (OMC.jl:700-719 sees every relaxed node): `cnt` rows of
 * `width` doubles from this rank (counts may differ between ranks); out = rows of rank 0, rank 1, ... ; counts[r] = rows of rank r.  With this the
 * Julia host needs no second transport (MPI) beside the library's communicator. */
int omc_allgather_records(omc_instance* h, const double* rows, int cnt, int width, double* out, int out_capacity_rows, int* counts);
int omc_comm_destroy(omc_instance* h);

/* per-kernel accounting of the last omc_relax_solve: launches and HIP-event milliseconds per kernel class */
#define OMC_KERNEL_COLPROX 0
#define OMC_KERNEL_CONE 1
#define OMC_KERNEL_GLOBAL 2
#define OMC_KERNEL_CHECK 3
#define OMC_KERNEL_SETUP 4
#define OMC_KERNEL_SMALL 5
#define OMC_KERNEL_ACCEL 6
#define OMC_KERNEL_CONESUB 7   /* k_cone_sub: the cone block by tracking the dominant 16-dimensional subspace */
#define OMC_KERNEL_CHECK_COL 8     /* certificate: exact f(Y) (one factorization per column, k_colprox mode 1) */
#define OMC_KERNEL_CHECK_BUILD 9   /* certificate: Lagrangian matrix and constants (k_check_build) */
#define OMC_KERNEL_HARVEST 10      /* finished slots: feasible U, separation eigenvector (OMC.jl:2466-2477), copy to the per-node outputs */
#define OMC_KERNEL_SHOR_BIGCONE 11 /* Shor mode: PSD projection of the order-(n+m) matrix [Y X; X' Theta] */
#define OMC_KERNEL_SHOR_MINORS 12  /* Shor mode: order-5 blocks (projection in registers, duals) and the shared V1, V2, V3 */
#define OMC_KERNEL_SHOR_COLS 13    /* Shor mode: per-column step (paraboloid, X, W, Theta, duals of the big cone) */
#define OMC_KERNEL_NCLASS 14      /* OMC_KERNEL_CHECK = eigenvalues of the Lagrangian matrix + decisions */
/* info[8]: solve seconds, total Jacobi sweeps of k_cone, rho, r_max, LDS flags (cone, global, small), R_max */
int omc_last_solver_info(omc_instance* h, double* info);
/* out[8] of the last omc_relax_solve: calls of k_cone_sub, its power steps, calls that fell back to the full eigendecomposition,
 * seedings of the tracked subspace by the full kernel, fall-backs by cause (more than 12 positive Ritz values, step cap, Cholesky
 * breakdown), Rayleigh-Ritz passes */
int omc_last_subspace_stats(omc_instance* h, int64_t* out);
/* the same eight counters for the order-(n+m) cone of the last Shor-mode solve */
int omc_last_shor_subspace_stats(omc_instance* h, int64_t* out);
/* Tuning / diagnostic knobs (OMC_STREAMS, OMC_GRAPH_MAX, OMC_NO_COLPROX_PAIR, ...: the list is OMC_TUNING_KEYS in omc_api.cpp, each documented
 * where it is used).  The library reads the environment at omc_instance_create and nowhere else; omc_tuning_set overrides one knob of an
 * instance (value NULL removes it), omc_tuning_reload_env reads the environment again.  No counterpart in the reference (its knobs are the
 * Mosek parameters of OMC.jl:1482-1500). */
int omc_tuning_set(omc_instance* h, const char* name, const char* value);
int omc_tuning_reload_env(omc_instance* h);
/* diagnostic builds (-DOMC_STAMPS) only: accumulated s_memtime ticks per kernel phase of node 0; zeros otherwise */
int omc_debug_stamps(omc_instance* h, double* out32);
/* diagnostic builds only: per-slot counters, out[c * slots + b]: c = 0 colprox wave cycles, 1 factorizations, 2 cone cycles,
 * 3 cone calls, 4 global cycles, 5 small cycles (zeros otherwise) */
int omc_debug_diag(omc_instance* h, double* out);
/* accepted / rejected extrapolated points of the node last relaxed in every slot (diagnostics; needs accel = 1) */
int omc_debug_aa(omc_instance* h, int* accepted, int* rejected);
/* last primal / dual ADMM residuals of every node of the staged batch (diagnostics) */
int omc_debug_residuals(omc_instance* h, double* rp, double* rd);
int omc_last_kernel_stats(omc_instance* h, int64_t* launches /*NCLASS*/, double* ms /*NCLASS*/,
                          int64_t* units /*NCLASS*/);

#ifdef __cplusplus
}
#endif
#endif /* OMC_H */
