/*
 * ilqr_hip.h -- C ABI of the MI355X-native batched iLQR hot path (libilqr_hip.so).
 *
 * The reference (idiap/ilqr_planner) has no FFI layer: its solvers call `sys::System` virtuals in-process
 * and solve ONE problem per call.  This library is what a maintainer binds under the reference's solver
 * classes to solve B independent instances of one System at once on an MI355X.  Each entry point cites the
 * reference interface it replaces (paths relative to ilqr_planner/ilqr_planner in the reference tree).
 *
 * Conventions
 *  - plain C, POD structs, raw pointers + sizes; no C++/torch types.
 *  - every function returning int returns 0 on success, non-zero on error; the message is
 *    ilqr_last_error(ctx).  The C++ host layer turns non-zero into std::runtime_error (the reference's only
 *    error convention, e.g. src/sim/KDLRobot.cpp:49,95; src/system/System.cpp:366).
 *  - host arrays are row-major, batch OUTERMOST ("natural" layout: instance b is what the reference would have
 *    been given / returned for that instance): X[B][T][n_x], U[B][T-1][n_u], K[B][T-1][n_u][n_x] ...
 *    The callee never keeps a host pointer after return.  Device layout is private.
 *  - `*_dev` variants take/return DEVICE pointers in the same natural layout (for callers whose data already
 *    lives in HBM, e.g. torch tensors); all work is enqueued on the context's stream.
 *  - one context per host thread; calls on one context are serialised by the caller.
 *  - all arithmetic is IEEE double, as in the reference (Eigen::MatrixXd everywhere).
 */
#ifndef ILQR_HIP_H
#define ILQR_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILQR_MAX_SEG 24
#define ILQR_MAX_KP 8
#define ILQR_MAX_NX 15
#define ILQR_MAX_NU 8
#define ILQR_MAX_NF 15
#define ILQR_MAX_NQ 13

/* system kinds: which sys::System subclass is being lowered */
#define ILQR_SYS_POS_ORN 0      /* sys::PosOrnPlannerSys      (src/system/PosOrnPlannerSys.cpp) */
#define ILQR_SYS_POS_ORN_TIME 1 /* sys::PosOrnTimePlannerSys  (src/system/PosOrnTimePlannerSys.cpp) */
#define ILQR_SYS_JOINT 2        /* sys::JointSpacePlannerSys  (src/system/JointSpacePlannerSys.cpp), nb_deriv = 1: target space = joint
                                   space, J = I, AngularKeypoint targets (n_f = n_Q = dof); no chain needed (n_seg may be 0) */
#define ILQR_SYS_JOINT_TIME 3   /* sys::JointSpaceTimePlannerSys (src/system/JointSpaceTimePlannerSys.cpp), nb_deriv = 1: joint space +
                                   time state, dt = u_last^2, AngularTimeKeypoint targets [q*, t*] */

/* per-instance status word */
#define ILQR_STATUS_OK 0
#define ILQR_STATUS_NONFINITE 1   /* accepted cost is NaN/Inf (the reference prints -nan and carries on) */
#define ILQR_STATUS_ALPHA_FLOOR 2 /* last line search bottomed out at alpha <= alpha_floor (accept-anyway rule) */

typedef struct ilqr_ctx ilqr_ctx;
typedef struct ilqr_problem ilqr_problem;

/*
 * Flat description of one sys::System + its sim::KDLRobot, shared by all B instances.
 * Replaces what the solvers read through System virtuals:
 *   chain            <- KDLRobot's KDL::Chain after TinyURDFParser + the "robot_custom_tip" segment
 *                       (src/sim/KDLRobot.cpp:45-66): segment s is Trans(xyz) * R * Rot(axis, q[joint]) .
 *   kind,nb_deriv,dt <- System subclass + localInit (src/system/PosOrnPlannerSys.cpp:54-78,
 *                       src/system/PosOrnTimePlannerSys.cpp:50-83)
 *   R_diag           <- System::R (src/system/System.cpp:41,73)
 *   limits_*         <- state_max_/state_min_/joint_limits_weight_/penalty_ (src/system/System.cpp:38-61)
 *   kp_*             <- Keypoint::getTimestep()/getPrecision() (include/ilqr_planner/system/Keypoint.h:23-35);
 *                       timesteps must be unique and ascending (System.cpp:77-86 sorts them; a std::map keeps the
 *                       last of duplicates).  Keypoint TARGETS are per instance: ilqr_problem_set_keypoint_targets.
 *   reg,alpha_floor,stop_tol <- hard-coded constants of ILQRRecursive.cpp:89,155,174
 */
typedef struct {
    int kind;      /* ILQR_SYS_* */
    int nb_deriv;  /* 1 or 2 */
    int dof;       /* moving joints of the chain (device path: 7) */
    int horizon;   /* T */
    double dt;     /* PosOrn only; time systems take dt = u_last^2 */
    double R_diag[ILQR_MAX_NU];
    int limits_set;
    double penalty;
    double state_max[ILQR_MAX_NX + 1], state_min[ILQR_MAX_NX + 1];
    int limit_weight[ILQR_MAX_NX + 1];
    int n_seg;
    int seg_joint[ILQR_MAX_SEG];  /* -1 fixed, else joint index */
    double seg_xyz[ILQR_MAX_SEG][3];
    double seg_R[ILQR_MAX_SEG][9]; /* row-major */
    double seg_axis[ILQR_MAX_SEG][3];
    int n_kp;
    int kp_timestep[ILQR_MAX_KP];
    double kp_Q[ILQR_MAX_KP][ILQR_MAX_NQ * ILQR_MAX_NQ]; /* row-major n_Q x n_Q, leading dimension n_Q */
    /* PosOrnKeypointDistFunct (src/system/PosOrnKeypointDistFunct.cpp:13-35): dead zones on the residual of keypoint k --
     * position part shrunk by kp_pos_radius towards 0 (0 inside the ball), each orientation component by kp_orn_thresh.
     * kp_dist[k] = 0: plain PosOrnKeypoint (the default; NOT the same as radius 0, which renormalises the residual). */
    int kp_dist[ILQR_MAX_KP];
    double kp_pos_radius[ILQR_MAX_KP];
    double kp_orn_thresh[ILQR_MAX_KP][3];
    /* Object frames and sequential systems.  A keypoint whose sub-system drives the robot through a
     * sim::TransformedSimulationInterface (src/sim/TransformedSimulationInterface.cpp:53-103) sees the pose and the Jacobian in
     * the frame T = [kp_frame_R | kp_frame_p]: p' = R'(p - t), R_ee' = R' R_ee (Eigen quaternion), J' = blkdiag(R,R)' J.
     * kp_has_frame[k] = 0: base frame.  sys::SequentialSystem (src/system/SequentialSystem.cpp:78-168) sums the costs of its
     * sub-systems: every keypoint carries the control penalty of its own sub-system (kp_has_Ru / kp_Ru; the sequential
     * system's own Rt stays in R_diag for l_u, l_uu) and the limit terms are added once per sub-system (limit_multiplicity;
     * 0 = 1).  Sub-systems whose keypoints share a timestep are not lowered.  With limit_multiplicity > 1 the batch solvers
     * (ilqr_solve_batch_cp, ilqr_solve_batch) apply NO limit terms: the reference's SequentialSystem does not override fpBatch, which
     * then runs on the sequence object itself, constructed without limits (SequentialSystem.cpp:12-18). */
    int kp_has_frame[ILQR_MAX_KP];
    double kp_frame_R[ILQR_MAX_KP][9]; /* row-major */
    double kp_frame_p[ILQR_MAX_KP][3];
    int kp_has_Ru[ILQR_MAX_KP];
    double kp_Ru[ILQR_MAX_KP][ILQR_MAX_NU];
    /* Hybrid sequences (HYBRID_SYS*.ipynb): a SequentialSystem may mix a JointSpace(Time)PlannerSys with PosOrn(Time)PlannerSys
     * sub-systems (same state and controls, nb_deriv = 1).  kp_joint[k] = 1 marks keypoint k as the Angular(Time)Keypoint of the
     * joint-space sub-system: target = joint vector (+ continuous time) in the keypoint's n_f slots, residual target - x, J = I
     * (src/system/JointSpacePlannerSys.cpp:77-81), precision n_x x n_x in kp_Q[k] with leading dimension n_x. */
    int kp_joint[ILQR_MAX_KP];
    int limit_multiplicity;
    /* is_sequence: the problem is a sys::SequentialSystem (set by its lowering; limit_multiplicity > 1 implies it): the batch solvers
     * then apply no limit terms.  limits2_set: a second group of sub-systems whose bounds differ from the first group's
     * (HYBRID_SYS_TIME.ipynb gives its two sub-systems (qMax, qMin) and (qMax, -qMax)); every group adds its own limit terms,
     * limit_multiplicity2 times.  Problems with a second group run on the generic one-lane-per-instance kernels. */
    int is_sequence;
    int limits2_set;
    double penalty2;
    double state_max2[ILQR_MAX_NX + 1];
    double state_min2[ILQR_MAX_NX + 1];
    int limit_weight2[ILQR_MAX_NX + 1];
    int limit_multiplicity2;
    double reg;          /* 1e-6 */
    double alpha_floor;  /* 1e-3 */
    double stop_tol;     /* 1e-3 */
} ilqr_problem_desc;

/* dimensions derived from (kind, nb_deriv, dof): n_x, n_u, n_f (target space), n_Q (residual space) */
typedef struct { int n_x, n_u, n_f, n_Q; } ilqr_dims;
int ilqr_dims_of(const ilqr_problem_desc* desc, ilqr_dims* out);
void ilqr_desc_defaults(ilqr_problem_desc* desc); /* zero + reg/alpha_floor/stop_tol defaults */

/* ---- URDF -> chain (host only, no GPU needed) ------------------------------------------------------------- */
/* What sim::KDLRobot's constructor gets from TinyURDFParser + KDL (src/sim/KDLRobot.cpp:45-66): fills
 * desc->{dof, n_seg, seg_joint, seg_xyz, seg_R, seg_axis} with the joints from base_frame to tip_frame plus the user
 * tool frame Frame(EulerZYX(tool_rpy[0], tool_rpy[1], tool_rpy[2]), tool_xyz) (NULL = identity) as a last fixed
 * segment.  lower/upper (may be NULL) receive the URDF joint limits [dof].  Error text: ilqr_urdf_last_error()
 * ("[KDLRobot] Unable to build kinematic chain from <base> to <tip>" as KDLRobot.cpp:49,56 throws). */
int ilqr_chain_from_urdf(const char* urdf_text, const char* base_frame, const char* tip_frame, const double* tool_rpy,
                         const double* tool_xyz, ilqr_problem_desc* desc, double* lower, double* upper);
const char* ilqr_urdf_last_error(void);

/* ---- context: device + stream + error text -------------------------------------------------------------- */
int ilqr_ctx_create(int device_id, ilqr_ctx** out);
void ilqr_ctx_destroy(ilqr_ctx* ctx);
const char* ilqr_last_error(const ilqr_ctx* ctx);
/* run everything on the caller's hipStream_t (e.g. torch's current stream); NULL = the context's own stream */
int ilqr_ctx_set_stream(ilqr_ctx* ctx, void* hip_stream);
int ilqr_ctx_synchronize(ilqr_ctx* ctx);
/* Large batches of the systems that use the wave-per-instance MFMA sweep are solved as two halves on two internal streams, joined to the
 * context's stream by events (instances are independent: results do not depend on it).  on = 0 keeps every launch on the context's
 * stream, one kernel at a time -- what a profiler run wants.  Default: on (1).  on = 2 splits every cooperative path (experiments only: measured
 * slower on the single-integrator systems).  (No reference counterpart: the reference has no batch.) */
int ilqr_ctx_set_split(ilqr_ctx* ctx, int on);
/* Cross-check kernel variants for parity tests (no reference counterpart; the library reads no environment variable -- these are context state,
 * in force for every later solve on the context): generic_kernels = 1 runs ILQRRecursive / AL_ILQR on the generic one-lane-per-instance kernel
 * set instead of the cooperative one; cp_lane_solve = 1 solves the Batch-CP normal equations with one lane per instance instead of one wave;
 * cp_general = 1 sends Batch-CP on the constant-dt systems through the general path of the time systems; mfma_sweep selects the backward sweep of
 * the 2nd-order / time systems: 0 = by batch size (one instance per wave on the f64 matrix cores up to two waves per SIMD, 16 lanes per instance
 * with rows in registers beyond), 1 = always the former, 2 = always the latter (the two agree to rounding, not bit for bit); its bits 2-3 select
 * the forward pass of the single-integrator systems the same way (0 = by batch size, 4 = the bandwidth-built k_forward_wg, 8 = the
 * latency-built k_forward_dpp); bits 4-5 the re-roll of the line-search winner on the time systems (0 = by batch size, 16 = k_apply_rows_tm,
 * 32 = k_apply_dpp_tm).  All 0 = the product path. */
int ilqr_ctx_set_crosscheck(ilqr_ctx* ctx, int generic_kernels, int cp_lane_solve, int cp_general, int mfma_sweep);
const char* ilqr_version(void);

/* ---- a batch of B instances of one System ---------------------------------------------------------------- */
int ilqr_problem_create(ilqr_ctx* ctx, const ilqr_problem_desc* desc, int batch, ilqr_problem** out);
void ilqr_problem_destroy(ilqr_problem* p);

/* q0_/dq0_ captured by localInit (PosOrnPlannerSys.cpp:57-58): q0[B][dof], dq0[B][dof] (NULL = zeros) */
int ilqr_problem_set_init_state(ilqr_problem* p, const double* q0, const double* dq0);
/* Keypoint target in f(x) layout [p(3), quat wxyz(4) (, dp(3), dquat(4)) (, t)] : target[B][n_f]
 * (PosOrnKeypoint ctor args, include/ilqr_planner/system/PosOrnKeypoint.h:18-33; SpacetimeKeypoint.h:17-33) */
int ilqr_problem_set_keypoint_targets(ilqr_problem* p, int kp_index, const double* target);
/* U0 of ILQRRecursive::solve / AL_ILQR::solve (include/ilqr_planner/solver/ILQRRecursive.h:36): U0[B][T-1][n_u].
 * Kept on the device so a solve can be repeated from the same start (ilqr_solve_* always restart from it). */
int ilqr_problem_set_controls(ilqr_problem* p, const double* U0);
/* solver::Constraint list + initLambda of AL_ILQR's ctor (include/ilqr_planner/solver/AL-ILQR.h:20-36):
 * A is m x (n_x+n_u), b is m; per_step=0: one (A,b) for every k, else A[T-1][m][n_x+n_u], b[T-1][m];
 * shared by all instances.  lambda0[B][T-1][m] (NULL = zeros). */
int ilqr_problem_set_constraints(ilqr_problem* p, int m, int per_step, const double* A, const double* b, const double* lambda0);
/* put the multipliers back to the lambda0 given to ilqr_problem_set_constraints (= constructing a fresh AL_ILQR) */
int ilqr_problem_reset_multipliers(ilqr_problem* p);
/* device-pointer variants (same layouts, memory already in HBM) */
int ilqr_problem_set_init_state_dev(ilqr_problem* p, const double* q0, const double* dq0);
int ilqr_problem_set_keypoint_targets_dev(ilqr_problem* p, int kp_index, const double* target);
int ilqr_problem_set_controls_dev(ilqr_problem* p, const double* U0);

/* ---- solvers ----------------------------------------------------------------------------------------------- */
/* ILQRRecursive::solve(U0, nb_iter, line_search, early_stop, cb)  (src/solver/ILQRRecursive.cpp:21-181),
 * for all B instances; asynchronous on the context's stream. */
int ilqr_solve_recursive(ilqr_problem* p, int nb_iter, int line_search, int early_stop);
/* AL_ILQR::solve(U0, nb_iter, lag_update_step, penalty, scaling_factor, line_search, early_stop, cb)
 * (src/solver/AL-ILQR.cpp:50-232); multipliers persist in the problem across calls like the reference's
 * `multipliers` member unless re-set with ilqr_problem_set_constraints. */
int ilqr_solve_al(ilqr_problem* p, int nb_iter, int lag_update_step, double penalty, double scaling_factor,
                  int line_search, int early_stop);
/* BatchILQRCP::solve(nb_iter, u0, early_stop, cb) (src/solver/BatchILQRCP.cpp:109-175) with the shared basis
 * PSI ((T-1) n_u x Kw, row-major); u0 = the controls set with ilqr_problem_set_controls.  Kw: any on the constant-dt systems
 * (Kw > 16 through the low-rank form), up to 32 on the time systems. */
int ilqr_solve_batch_cp(ilqr_problem* p, const double* psi, int Kw, int nb_iter, int early_stop);
/* BatchILQR::solve(nb_iter, u0, early_stop, cb) (src/solver/BatchILQR.cpp:110-173): Gauss-Newton on the whole control
 * sequence, i.e. BatchILQRCP with the identity basis (Kw = (T-1) n_u); the identity is never materialised. */
int ilqr_solve_batch(ilqr_problem* p, int nb_iter, int early_stop);

/* ---- results (host, natural layout); each synchronises the stream ------------------------------------------ */
int ilqr_problem_get_X(ilqr_problem* p, double* X);       /* [B][T][n_x]     ILQRRecursive tuple<0> */
int ilqr_problem_get_fX(ilqr_problem* p, double* fX);     /* [B][T][n_f]     tuple<1> (one batched FK pass) */
int ilqr_problem_get_U(ilqr_problem* p, double* U);       /* [B][T-1][n_u]   tuple<2> */
int ilqr_problem_get_K(ilqr_problem* p, double* K);       /* [B][T-1][n_u][n_x] tuple<3> */
int ilqr_problem_get_d(ilqr_problem* p, double* d);       /* [B][T-1][n_u]   tuple<4> (scaled by accepted alpha) */
int ilqr_problem_get_cost(ilqr_problem* p, double* cost); /* [B]             tuple<5> */
int ilqr_problem_get_alpha(ilqr_problem* p, double* alpha); /* [B] last accepted alpha */
int ilqr_problem_get_iters(ilqr_problem* p, int* iters);  /* [B] iterations run (early stop) */
int ilqr_problem_get_status(ilqr_problem* p, int* status);/* [B] ILQR_STATUS_* */
int ilqr_problem_get_lambda(ilqr_problem* p, double* lambda); /* [B][T-1][m] */
/* per-iteration stream the reference prints through CallBackMessage ("Iteration i, Cost: c, alpha= a",
 * ILQRRecursive.cpp:167-172): cost_trace/alpha_trace[B][nb_iter] of the last solve (NaN after an early stop) */
int ilqr_problem_get_trace(ilqr_problem* p, double* cost_trace, double* alpha_trace, int nb_iter);
/* device-pointer variants */
int ilqr_problem_get_X_dev(ilqr_problem* p, double* X);
int ilqr_problem_get_U_dev(ilqr_problem* p, double* U);
int ilqr_problem_get_cost_dev(ilqr_problem* p, double* cost);

/* ---- receding horizon and tracking (the uses of the gains the tutorials mention, POS_ORN_SYS.ipynb cell 7; no reference
 * function: ILQRRecursive::solve returns K, k and the caller replays them) -------------------------------------- */
/* The next solve starts from the accepted plan shifted by `shift` timesteps: U0[k] = U[min(k+shift, T-2)] and, for
 * shift > 0, q0 (dq0) = joint part of x_shift -- MPC-style re-planning of the whole batch without leaving HBM. */
int ilqr_problem_warm_start(ilqr_problem* p, int shift);
/* u = ubar_k + K_k (x_meas - xbar_k) [+ alpha d_k if with_feedforward] for every instance:
 * x_meas[B][n_x] -> u_out[B][n_u]; K_k, d_k = the gains ilqr_problem_get_K / get_d return. */
int ilqr_problem_track(ilqr_problem* p, int k, const double* x_meas, int with_feedforward, double* u_out);
int ilqr_problem_track_dev(ilqr_problem* p, int k, const double* x_meas, int with_feedforward, double* u_out);

/* ---- stand-alone batched kinematics: KDLRobot::updateKinematics for n configurations ---------------------- */
/* (src/sim/KDLRobot.cpp:83-115): q[n][dof] (dq[n][dof] or NULL) -> pos[n][3], quat[n][4] (w,x,y,z), jac[n][6][dof];
 * any output may be NULL.  Host pointers. */
int ilqr_fk_batch(ilqr_ctx* ctx, const ilqr_problem_desc* desc, int n, const double* q, double* pos, double* quat, double* jac);

/* ---- instrumentation ----------------------------------------------------------------------------------------- */
/* When enabled, every kernel launch is bracketed by hipEvents on the launch stream; totals are read back with
 * ilqr_profile_get (index: ILQR_PROF_*).  Used by bench.py for the roofline figures. */
#define ILQR_PROF_ROLLOUT 0
#define ILQR_PROF_BACKWARD 1
#define ILQR_PROF_FORWARD 2
#define ILQR_PROF_OTHER 3
#define ILQR_PROF_APPLY 4 /* second forward pass: re-roll of the winning step size */
#define ILQR_PROF_COUNT 5
int ilqr_profile_enable(ilqr_ctx* ctx, int on);
int ilqr_profile_reset(ilqr_ctx* ctx);
int ilqr_profile_get(ilqr_ctx* ctx, int which, double* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif
