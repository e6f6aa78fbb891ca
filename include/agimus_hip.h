/*
 * agimus_hip.h -- C ABI of the MI355X-native OCP solve path (libagimus_hip.so).
 *
 * This is the drop-in boundary for the one hot path of agimus_controller:
 * OCPCrocoGeneric.solve() + warm-start shift + reference update.  Every entry
 * point cites the reference interface it replaces (paths are relative to the
 * upstream agimus_controller repository).  Plain pointers and sizes only: no
 * C++ types, no torch types.  All floating point is IEEE fp64, row-major.
 *
 * The same structures are consumed by oracle/ (the CPU checker), which is test
 * infrastructure and never part of the product path.
 *
 * Conventions
 *   nv = nq = nu        number of 1-DoF revolute joints (full actuation)
 *   nx = ndx = 2*nv     state x = [q; v]
 *   T                   number of controls (running nodes); T+1 states
 *   B                   batch: independent MPC instances resident on one GPU
 *   spatial vectors     [linear(3); angular(3)] (Pinocchio ordering)
 *   SE3 as 12 doubles   R row-major (9) then p (3)
 *
 * Status codes: 0 = ok, negative = error (message via agx_last_error()).
 * Handles are thread-compatible, not thread-safe.
 */
#ifndef AGIMUS_HIP_H
#define AGIMUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGX_MAX_ROWS 8   /* cost rows per node type (running / terminal)        */
#define AGX_MAX_NV 32    /* joints a model table may have; kernels are compiled for nv in {1,2,3,4,6,7} (register-resident
                            path) and 30 (LDS path for large models); other sizes are refused by agx_ocp_create */

/* Residual kinds: class names of the YAML schema,
 * agimus_controller/agimus_controller/ocp/ocp_croco_generic.py:147-550.        */
enum agx_residual_kind {
  AGX_RES_STATE = 0,             /* ResidualModelState            :154 */
  AGX_RES_CONTROL = 1,           /* ResidualModelControl          :170 */
  AGX_RES_CONTROL_GRAV = 2,      /* ResidualModelControlGrav      :186 */
  AGX_RES_FRAME_PLACEMENT = 3,   /* ResidualModelFramePlacement   :198 (+Static :223, VisualServoing :436) */
  AGX_RES_FRAME_TRANSLATION = 4, /* ResidualModelFrameTranslation :252 (+Static :277) */
  AGX_RES_FRAME_ROTATION = 5,    /* ResidualModelFrameRotation    :306 (+Static :331) */
  AGX_RES_FRAME_VELOCITY = 6,    /* ResidualModelFrameVelocity    :360 (+Static :396) */
  AGX_RES_COLLISION = 7          /* ResidualDistanceCollision     :524 */
};

/* Activation kinds, ocp_croco_generic.py:93-143.                               */
enum agx_activation_kind {
  AGX_ACT_WEIGHTED_QUAD = 0, /* a = 1/2 sum w_j r_j^2 (also the default quad, w = 1) */
  AGX_ACT_EXP = 1,           /* colmpc ActivationModelExp(nr, alpha):     a = exp(-|r| / alpha)   */
  AGX_ACT_QUAD_EXP = 2       /* colmpc ActivationModelQuadExp(nr, alpha): a = exp(-|r|^2 / alpha) */
};
/* Exp / QuadExp: any residual but ControlGrav / FrameVelocity; |r| over the nr components of the row, diagonal second
 * derivative; the activation weights of the row's tile are ignored.  Forms recalled, parity against colmpc unpinned. */

/* One CostModelSumItem (ocp_croco_generic.py:578-585) lowered to a table row.
 * Per-node data of the row lives in the reference tile (agx_ocp_set_refs):
 *   [ item weight (1) | reference (agx_row_nref) | activation weights (agx_row_nr) ]
 * rows are laid out back to back in table order.                               */
typedef struct agx_cost_row {
  int32_t kind;       /* agx_residual_kind                                      */
  int32_t activation; /* agx_activation_kind                                    */
  int32_t active;     /* CostModelSumItem.active                                */
  int32_t frame;      /* default frame id (may be overridden per node); collision: first geometry frame */
  int32_t frame_b;    /* collision: second geometry frame of the pair (:499-533);
                         FrameVelocity: reference frame 0 WORLD / 1 LOCAL / 2 LOCAL_WORLD_ALIGNED (:360-432) */
  int32_t pad_;
  double alpha;       /* Exp / QuadExp parameter                                */
  double weight;      /* CostModelSumItem.weight of the YAML: the item weight the device-resident
                         reference generators write (host tiles carry their own per node) */
} agx_cost_row;

/* One ConstraintListItem (ocp_croco_generic.py:554-647) lowered to a row:
 *   lower <= r(x, u) <= upper   with r one of the residual kinds above.
 * ConstraintModelControlLimit (:624-640) is the AGX_RES_CONTROL row with zero
 * reference and -/+ effort limit.  Constraints are not updated per node
 * (:720-721), so reference and bounds are static.  Every residual kind is
 * implemented (references laid out like the cost rows'; FrameVelocity: frame_b =
 * reference frame 0 WORLD / 1 LOCAL / 2 LOCAL_WORLD_ALIGNED); at most 4 rows and 8
 * components with a dense Jacobian (collision 1, translation / rotation 3,
 * placement / velocity 6, ControlGrav nv) per node type; residuals on u are
 * dropped at the terminal node.  Models above 7 joints (after padding to the compiled
 * capacity): State, Control, collision and frame translation / rotation / placement
 * rows, up to 104 components per node; FrameVelocity / ControlGrav rows are refused. */
typedef struct agx_constraint_row {
  int32_t kind;        /* agx_residual_kind                                     */
  int32_t active;      /* ConstraintListItem.active                             */
  int32_t frame;
  int32_t frame_b;     /* as in agx_cost_row                                    */
  const double *ref;   /* [agx_row_nref(kind)] or NULL = zeros                  */
  const double *lower; /* [agx_row_nr(kind)], -inf allowed                      */
  const double *upper; /* [agx_row_nr(kind)], +inf allowed                      */
} agx_constraint_row;

/* Robot description: what factory/robot_model.py:88-351 extracts from the URDF
 * (reduced model, armature :346-351), as a flat table.                         */
typedef struct agx_model_desc {
  int32_t nv;
  int32_t nframes;
  const int32_t *parent;         /* [nv]   parent joint, -1 = world            */
  const double *placement;       /* [nv][12] joint frame in parent joint frame  */
  const double *axis;            /* [nv][3]  unit revolute axis, joint frame    */
  const double *mass;            /* [nv]                                        */
  const double *com;             /* [nv][3]  joint frame                        */
  const double *inertia;         /* [nv][9]  about com, joint frame             */
  const double *armature;        /* [nv]                                        */
  const double *effort_limit;    /* [nv]                                        */
  const double *gravity;         /* [3]      linear gravity, world              */
  const int32_t *frame_parent;   /* [nframes] parent joint, -1 = world          */
  const double *frame_placement; /* [nframes][12] in parent joint frame         */
  /* Collision geometry (factory/robot_model.py:261-302: cylinders become coal.Capsule(radius,
   * halfLength)): a geometry object is a frame (parent joint + placement) whose local z axis
   * carries the capsule segment; halflen = 0 is a sphere.  NULL = no geometry.  */
  const double *frame_radius;    /* [nframes]                                   */
  const double *frame_halflen;   /* [nframes]                                   */
  /* Box geometry (coal.Box is kept as is by factory/robot_model.py:296-302): half extents along the
   * frame's axes, all 0 = not a box.  Pairs box / capsule and box / sphere are supported, box / box
   * is refused by agx_ocp_create.  NULL = no boxes.                              */
  const double *frame_box;       /* [nframes][3]                                */
} agx_model_desc;

/* Shooting problem + solver knobs:
 * OCPCrocoGeneric.create_running_model_list/create_terminal_model
 * (ocp_croco_generic.py:798-812) and OCPBaseCroco.__init__
 * (ocp_base_croco.py:55-80), OCPParamsBaseCroco (ocp_param_base.py:31-85).     */
typedef struct agx_ocp_desc {
  int32_t horizon;                  /* T = n_controls                           */
  const double *dt;                 /* [T] timesteps; terminal node has dt = 0  */
  int32_t n_running_rows;
  const agx_cost_row *running_rows; /* [n_running_rows]                         */
  int32_t n_terminal_rows;
  const agx_cost_row *terminal_rows;
  double termination_tolerance;     /* KKT tolerance, ocp_param_base.py:54      */
  int32_t max_qp_iters;             /* ocp_param_base.py:53                     */
  double eps_abs, eps_rel;          /* ocp_param_base.py:60-61                  */
  double mu_dynamic, mu_constraint; /* merit penalties (mim_solvers defaults)   */
  int32_t use_filter_line_search;   /* ocp_param_base.py:64                     */
  int32_t n_running_constraints;
  const agx_constraint_row *running_constraints;
  int32_t n_terminal_constraints;
  const agx_constraint_row *terminal_constraints;
} agx_ocp_desc;

/* Per-instance solver report: OCPDebugData fields filled by
 * OCPBaseCroco.fill_debug_data (ocp_base_croco.py:134-140).                    */
typedef struct agx_status {
  double kkt;       /* solver.KKT                                               */
  double cost;      /* total cost at the returned point's last evaluation       */
  double merit;     /* merit at that evaluation                                 */
  double gap_norm;  /* l1 norm of the dynamics gaps at that evaluation          */
  int32_t iter;     /* solver.iter                                              */
  int32_t qp_iters; /* solver.qp_iters                                          */
  int32_t solved;   /* return value of solver.solve()                           */
  int32_t flags;    /* bit0: non-finite result / discarded direction, bit1: a line search failed (all ten step lengths), bit2: a step length was rejected (the search backtracked) */
} agx_status;

/* Doubles per node in the derivative tile written by the node-parallel pass:
 * Fx ndx^2 | Fu ndx*nu | f ndx | Lx ndx | Lu nu | Lxx ndx^2 | Lxu ndx*nu | Luu nu^2 | cost 1 */
#define AGX_TILE_DOUBLES(nv) (4 * (nv) * (nv) + 2 * (nv) * (nv) + 2 * (nv) + 2 * (nv) + (nv) + 4 * (nv) * (nv) + 2 * (nv) * (nv) + (nv) * (nv) + 1)

typedef struct agx_model agx_model;
typedef struct agx_ocp agx_ocp;

const char *agx_last_error(void);
/* Number of visible HIP devices (0 when none); never throws.                   */
int agx_device_count(void);

/* Layout helpers shared by host code and kernels.                              */
int agx_row_nref(int kind, int nv); /* reference doubles of a row               */
int agx_row_nr(int kind, int nv);   /* residual dimension of a row              */
/* doubles per node of the reference tile for this problem (max of running and
 * terminal tables)                                                             */
int agx_ref_stride(const agx_ocp_desc *desc, int nv);

/* ---- model ------------------------------------------------------------- */
int agx_model_create(const agx_model_desc *desc, agx_model **out);
void agx_model_destroy(agx_model *m);

/* ---- problem ----------------------------------------------------------- */
/* Replaces OCPBaseCroco.__init__ (ocp_base_croco.py:17-80): allocates the
 * device-resident horizon buffers for `batch` instances on HIP device `device`. */
int agx_ocp_create(const agx_model *m, const agx_ocp_desc *desc, int batch, int device, agx_ocp **out);
void agx_ocp_destroy(agx_ocp *ocp);
/* Run the kernels on a caller-provided hipStream_t (0 = library-owned stream). */
int agx_ocp_set_stream(agx_ocp *ocp, void *hip_stream);
int agx_ocp_sync(agx_ocp *ocp);
/* Replaces OCPBaseCroco.update_geometry_placement (ocp_base_croco.py:110-132): new placement
 * (R row major 9 | p 3, in the parent joint frame; world for parent -1) of a geometry frame,
 * e.g. a moving obstacle.  Applies to every instance of the handle.              */
int agx_ocp_set_geom_placement(agx_ocp *ocp, int frame, const double *se3);
/* Batch policy -- no counterpart upstream, where every controller is its own process (mpc.py:14-19) and a slow solve
 * delays only itself.  In a batch the SQP loop of one step runs until EVERY instance has finished; with a quorum
 * below 1 it ends as soon as that fraction has, and the ADMM loop of an SQP iteration as soon as that fraction of
 * the QPs has converged.  The remaining instances keep their current iterate and report solved = 0 with the SQP /
 * ADMM iterations they really ran (iter, qp_iters): what a lone controller returns when it runs into max_iter /
 * max_solve_time (ocp_base_croco.py:160-171); the next MPC step continues from that iterate through the warm-start shift.
 * Defaults 1.0 / 1.0: wait for everyone.                                            */
int agx_ocp_set_quorum(agx_ocp *ocp, double sqp_fraction, double qp_fraction);
/* Constrained problems keep the ADMM multipliers y and the penalty rho between solves, as the
 * reference's solver object does (SolverCSQP reset_y = reset_rho = false).  This forgets them:
 * the state of a freshly constructed solver.                                      */
int agx_ocp_reset_duals(agx_ocp *ocp);

/* Replaces OCPCrocoGeneric.set_reference_weighted_trajectory
 * (ocp_croco_generic.py:855-892): ref_tile [B][T+1][stride] host doubles,
 * frame_ids [B][T+1][AGX_MAX_ROWS] host int32 (NULL = row defaults).            */
int agx_ocp_set_refs(agx_ocp *ocp, const double *ref_tile, const int32_t *frame_ids);
/* The same update without stalling the caller or the solver (the reference pays ~ T x #costs binding calls per step for it,
 * ocp_croco_generic.py:883-888), in two halves so that the tile of step k+1 can travel while step k is being solved:
 *   agx_ocp_set_refs_async  STAGES a tile: a second stream copies it into the handle's second device tile (the solver
 *                           keeps reading the first one); hand in page-locked memory (agx_host_alloc) and the copy runs
 *                           at the rate of the link, concurrently with kernels;
 *   agx_ocp_refs_activate   makes the staged tile the current one: the solver's stream waits for the copy (the host does
 *                           not) and the two device tiles swap roles.
 * Per step: activate (tile k, staged during step k-1), stage tile k+1, solve step k.  The host buffer of a staged tile must
 * stay untouched until agx_ocp_refs_wait returns or the solve after its activation has returned.                          */
int agx_ocp_set_refs_async(agx_ocp *ocp, const double *ref_tile, const int32_t *frame_ids);
int agx_ocp_refs_activate(agx_ocp *ocp);
int agx_ocp_refs_wait(agx_ocp *ocp);
/* Page-locked host memory for the tiles and results handed to this library (a plain allocation works everywhere, but its
 * transfers are staged by the runtime at a fraction of the link's rate).                                                */
int agx_host_alloc(size_t bytes, void **out);
int agx_host_free(void *p);
/* Same with device-resident tiles (no copy of the doubles is made when
 * `adopt` != 0: the solver then reads the caller's buffer directly).            */
int agx_ocp_set_refs_device(agx_ocp *ocp, const double *d_ref_tile, const int32_t *d_frame_ids, int adopt);

/* Replaces OCPBaseCroco.solve (ocp_base_croco.py:142-182) for B instances.
 * Host buffers: x0 [B][nx], xs_ws [B][T+1][nx], us_ws [B][T][nu];
 * outputs xs [B][T+1][nx], us [B][T][nu], K [B][T][nu][ndx], st [B].
 * max_iter <= 0 means 1000 (use_iteration_limits_and_timeout=False);
 * max_time <= 0 means no wall-clock cap.                                       */
int agx_ocp_solve(agx_ocp *ocp, const double *x0, const double *xs_ws, const double *us_ws,
                  int max_iter, double max_time, double *xs, double *us, double *K, agx_status *st);

/* Device-resident variant: the warm start is whatever currently sits in the
 * resident xs/us buffers (after agx_ocp_upload_warmstart or
 * agx_ocp_shift_warmstart); results stay on the device.                        */
int agx_ocp_upload_x0(agx_ocp *ocp, const double *x0);
int agx_ocp_upload_warmstart(agx_ocp *ocp, const double *xs_ws, const double *us_ws);
int agx_ocp_solve_resident(agx_ocp *ocp, int max_iter, double max_time);
int agx_ocp_download(agx_ocp *ocp, double *xs, double *us, double *K, agx_status *st);
/* The full OCPResults (ocp_base_croco.py:173-177: xs, K, us of every node) without holding up the next step: a device-side
 * snapshot is taken in the solver's stream and drained to the host by a second stream; the destination buffers (any may be
 * NULL; page-locked memory from agx_host_alloc for the rate of the link) are complete when agx_ocp_download_wait returns.  */
int agx_ocp_download_async(agx_ocp *ocp, double *xs, double *us, double *K);
int agx_ocp_download_wait(agx_ocp *ocp);
/* Only what the ROS node publishes (agimus_controller_ros/agimus_controller.py:418-426):
 * us0 [B][nu], K0 [B][nu][ndx], x1 [B][nx] (may be NULL).                      */
int agx_ocp_download_first(agx_ocp *ocp, double *us0, double *K0, double *x1, agx_status *st);
/* Same data without the host-side scatter: one device-side pack, one transfer into a pinned buffer
 * owned by the handle.  *host -> [B][*stride] doubles, per instance
 *   us0 (nu) | K0 (nu*ndx, row major) | x1 (nx) | kkt cost merit gap_norm iter qp_iters solved flags;
 * valid until the next call on this handle.                                      */
int agx_ocp_first_packed(agx_ocp *ocp, const double **host, int *stride);

/* Replaces WarmStartShiftPreviousSolution.shift
 * (warm_start_shift_previous_solution.py:85-109) on the resident solution.     */
int agx_ocp_shift_warmstart(agx_ocp *ocp);
/* x0 <- xs[1] of the resident solution (closed loop on the own prediction, as
 * agimus_controller_examples/scripts/dummy_mpc_test.py:127-129).               */
int agx_ocp_x0_from_prediction(agx_ocp *ocp);

/* Replaces OCPBaseCroco.integrate (ocp_base_croco.py:184-189): one Euler step
 * of the node-0 model for n states. Host buffers x [n][nx], u [n][nu].         */
int agx_ocp_integrate(agx_ocp *ocp, int n, const double *x, const double *u, double *xnext);

/* Replaces pin.rnea at warm_start_reference.py:78 and
 * trajectories/sine_wave_configuration_space.py:56 for n samples (host).       */
int agx_model_rnea(agx_ocp *ocp, int n, const double *q, const double *v, const double *a, double *tau);
/* Frame placement (pin.framesForwardKinematics,
 * trajectories/trajectory_base.py:38-41): out [n][12].                         */
int agx_model_frame_placement(agx_ocp *ocp, int n, int frame, const double *q, double *out);
/* pinocchio.getFrameJacobian: J [n][6][nv], rows linear | angular; local = 0 LOCAL_WORLD_ALIGNED,
 * 1 LOCAL (the inverse kinematics of trajectories/sine_wave_cartesian_space.py:62-111 uses both). */
int agx_model_frame_jacobian(agx_ocp *ocp, int n, int frame, int local, const double *q, double *J);

/* Replaces the per-node residual copies of OCPCrocoGeneric.fill_debug_data
 * (ocp_croco_generic.py:840-853): residual of running row `row` at the resident
 * solution, out [B][T][nr].                                                    */
int agx_ocp_get_residuals(agx_ocp *ocp, int row, double *out);

/* ---- kernel-level entry points (parity tests and bench instrumentation) - */
/* Node-parallel derivative pass at the resident (xs, us): writes the tiles to
 * the workspace and optionally copies them out, tiles [B][T+1][AGX_TILE_DOUBLES]. */
int agx_ocp_calc_diff(agx_ocp *ocp, double *tiles);
/* One QP direction at the resident (xs, us) exactly as the solver computes it (derivative pass,
 * Riccati backward + linear forward, KKT, reported gains): K [B][T][nu][ndx], k = acceleration-space
 * feed-forward [B][T][nu], dx [B][T+1][ndx], du [B][T][nu], kkt [B].                                */
int agx_ocp_direction(agx_ocp *ocp, double *K, double *k, double *dx, double *du, double *kkt);
/* The QP tiles of the node-parallel derivative pass at the resident (xs, us) exactly as the solver's sweep reads
 * them (acceleration-input form): qt [B][T+1][*qt_size], aux [B][T+1][*aux_size] (either may be
 * NULL; the sizes are always returned).  Layout per node, blocks nv x LD row major with LD = 8 (nv <= 8) or 32:
 *   qt  = Hqq | Hqv | Hvv | Hqw | Hvw | Hww | gx (2 nv of 2 LD) | gw (nv of LD) | f (2 nv of 2 LD) | cost (1 of 8)
 *   aux = M | dtau/dq | dtau/dqdot | Lqq | Lvv (LD) | Luu (LD) | Lu (LD)                                   */
int agx_ocp_qp_tiles(agx_ocp *ocp, double *qt, double *aux, int *qt_size, int *aux_size);
/* Average device time in milliseconds of `reps` launches of one kernel,
 * measured with hipEvents on the problem's stream.
 * which: 0 = derivative pass (running + terminal launches), 1 = Riccati backward + forward,
 * 2 = step kernel (du, KKT, line search; nothing committed), 3 = derivative pass over the running
 * nodes only (one launch), 4 = canonical-tile derivative pass (running nodes), 5 = Riccati backward
 * only, 6 = exit (gains) sweep alone, 7 = direction + speculative gains sweep in one launch.        */
int agx_ocp_time_kernel(agx_ocp *ocp, int which, int reps, double *avg_ms);

/* In-situ kernel timing: while enabled, every solve brackets the launches of its SQP loop with
 * hipEvents on the problem's stream; ms_sum / count [3] return the accumulated device time and
 * launch count of {derivative pass over the running nodes, Riccati backward + forward, step
 * (per-node KKT + convergence test + line search)} since profiling was switched on.           */
int agx_ocp_profile(agx_ocp *ocp, int enable, double *ms_sum, long long *count);

/* ---- device-resident reference trajectory --------------- */
/* Sine wave in configuration space, trajectories/sine_wave_configuration_space.py:41-72,
 * for B instances and n_points time samples t_k = t0[b] + k*dt, written as
 * reference tiles [B][n_points][stride] directly in HBM for the goal-reaching
 * row table (control | state | frame placement).  Host parameter arrays:
 * q0 [B][nv], amp [B][nv], pulsation [B][nv], scale_duration [B][nv], t0 [B];
 * weights w_q, w_qdot, w_effort [nv], w_pose [6].                              */
int agx_traj_sine_create(agx_ocp *ocp, int n_points, double dt, const double *q0, const double *amp,
                         const double *pulsation, const double *scale_duration, const double *t0,
                         const double *w_q, const double *w_qdot, const double *w_effort,
                         const double *w_pose, int frame);
/* Same resident trajectory from caller-given samples q, dq, ddq [B][n_points][nv]
 * (GenericTrajectory.build_trajectory_from_q_dq_ddq_arrays, trajectories/generic_trajectory.py:37-70):
 * feed-forward effort by RNEA and end-effector pose by FK on the device.          */
int agx_traj_generic_create(agx_ocp *ocp, int n_points, const double *q, const double *dq, const double *ddq,
                            const double *w_q, const double *w_qdot, const double *w_effort, const double *w_pose, int frame);
/* Same resident trajectory from the Cartesian sine generator (SinusWaveCartesianSpace,
 * trajectories/sine_wave_cartesian_space.py:62-111): the end effector `frame` of instance b follows
 * p0_b + amp_b s(t) sin(pulsation_b t) with its initial orientation (s: the quintic of scale_duration);
 * q by the iterative inverse kinematics on the device (warm-started point to point, all six error components,
 * stop at |log6| < precision, error once more than it_max steps were taken -- upstream's `if i > it_max`, default 10000),
 * dq from the LOCAL_WORLD_ALIGNED Jacobian, ddq = 0.  The end-effector reference of a point is the DESIRED pose
 * (initial orientation, p0 + amp s sin), as upstream stores ee_des_pos.  On an inverse-kinematics failure the call
 * returns an error and the handle is left WITHOUT a resident trajectory.  q0 [B][nv], amp / pulsation [B][3].  nv <= 7. */
int agx_traj_cartesian_sine_create(agx_ocp *ocp, int n_points, double dt, const double *q0, const double *amp,
                                   const double *pulsation, double scale_duration, double precision, int it_max,
                                   const double *w_q, const double *w_qdot, const double *w_effort, const double *w_pose,
                                   int frame);
/* Point the solver at the horizon window starting at sample `k0` of the
 * resident trajectory (TrajectoryBuffer.horizon, trajectory.py:218-222, with
 * uniform horizon indexes).                                                    */
int agx_traj_set_window(agx_ocp *ocp, int k0);
/* Non-uniform horizon (TrajectoryBuffer.compute_horizon_indexes, trajectory.py:195-216: with
 * dt factors the node t looks at sample k0 + idx[t], pinned by tests/test_buffer.py:82-93 to
 * [0,1,2,4,6,9,12,16,20,25,30] for factors 1..4 x 2..3 steps).  idx [T+1], NULL = uniform.  */
int agx_traj_set_horizon_indexes(agx_ocp *ocp, const int32_t *idx);
/* Copy trajectory sample k of every instance to the host: q,v,a,u [B][nv], pose [B][12]. */
int agx_traj_get_point(agx_ocp *ocp, int k, double *q, double *v, double *a, double *u, double *pose);
/* Warm start from the reference (WarmStartReference.generate,
 * warm_start_reference.py:33-96) on the device: xs <- [x0, ref[1:]], us <- ref efforts. */
int agx_traj_warmstart_from_reference(agx_ocp *ocp);

/* One receding-horizon step, MPC.run (mpc.py:32-66), fully device resident:
 * window k0, then by `first`: 1 = warm start and x0 from the reference (first step);
 * 0 = x0 <- previous xs[1] (closed loop on the own prediction) + warm-start shift;
 * 2 = x0 as it is (set by agx_ocp_upload_x0 / agx_ocp_feedback_rollout) + warm-start shift; then solve. */
int agx_ocp_mpc_step(agx_ocp *ocp, int k0, int max_iter, int first);
/* What consumes an MPC step: the linear feedback controller fed by
 * AgimusController.send_control_msg (agimus_controller_ros/agimus_controller.py:418-426) applies
 *   u = us[0] + K[0] (x0 - x_measured)
 * at the control rate.  Here the plant is the model: n_substeps semi-implicit Euler steps of dt_sub
 * from the resident x0 under that law (+ an optional constant torque disturbance [B][nu], host);
 * the end state replaces x0, ready for agx_ocp_mpc_step(..., first = 2).           */
int agx_ocp_feedback_rollout(agx_ocp *ocp, int n_substeps, double dt_sub, const double *disturbance);
/* The resident initial state x0 [B][nx] (measured state of the next step).          */
int agx_ocp_download_x0(agx_ocp *ocp, double *x0);

#ifdef __cplusplus
}
#endif
#endif /* AGIMUS_HIP_H */
