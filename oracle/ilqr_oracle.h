/*
 * ilqr_oracle.h -- CPU restatement ("oracle") of the idiap/ilqr_planner hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ilqr_planner_amd/ (the product) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker / the timed CPU baseline.
 *
 * Plain C, IEEE double, single instance, single thread -- exactly the shape of the reference
 * (one problem per solve() call).  Every function cites the reference file:line it restates;
 * paths are relative to /root/reference/ilqr_planner/ilqr_planner ($L in SURVEY.md).
 *
 * Parity pin: the reference has no tests and cannot be built here (Eigen3 / orocos_kdl /
 * tinyxml2 absent).  This restatement is pinned against the per-iteration cost
 * traces stored in the reference's tutorial notebooks (tests/golden/traces.json), 6 significant
 * digits of cost per iteration and the exact alpha sequence; orocos_kdl internals (un-vendored,
 * unpinned version) are restated from their published algorithm and are pinned only through
 * those traces plus the FK literals in the notebooks.
 */
#ifndef ILQR_ORACLE_H
#define ILQR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SEG 24
#define ORC_MAX_DOF 7
#define ORC_MAX_NX 16
#define ORC_MAX_NU 8
#define ORC_MAX_NF 16
#define ORC_MAX_NQ 14
#define ORC_MAX_KP 8
#define ORC_MAX_M 32

enum { ORC_SYS_POS_ORN = 0, ORC_SYS_POS_ORN_TIME = 1, ORC_SYS_JOINT = 2 /* JointSpacePlannerSys, nb_deriv = 1 (App. D-10: 2 is broken upstream) */,
       ORC_SYS_JOINT_TIME = 3 /* JointSpaceTimePlannerSys, nb_deriv = 1 */ };
#define ORC_IS_TM(kind) ((kind) == ORC_SYS_POS_ORN_TIME || (kind) == ORC_SYS_JOINT_TIME)
#define ORC_IS_JOINT(kind) ((kind) == ORC_SYS_JOINT || (kind) == ORC_SYS_JOINT_TIME)

/* URDF chain base->tip, as orocos_kdl sees it after TinyURDFParser: one segment per URDF joint,
 * T_seg(q) = Trans(xyz) * R_fixed * Rot(axis, q)  (fixed joints: no Rot), then the user tool frame
 * Frame(EulerZYX(rpy0,rpy1,rpy2), xyz) appended as a fixed segment (src/sim/KDLRobot.cpp:61-66). */
typedef struct {
    int n_seg;
    int dof;
    int seg_joint[ORC_MAX_SEG];     /* -1 = fixed segment, else moving-joint index 0..dof-1 */
    double seg_xyz[ORC_MAX_SEG][3];
    double seg_R[ORC_MAX_SEG][9];   /* row-major fixed rotation of the joint origin */
    double seg_axis[ORC_MAX_SEG][3];
} orc_chain;

typedef struct {
    int timestep;
    double pos[3], orn[4];      /* orn = (w,x,y,z) */
    double dpos[3], dorn[4];    /* 2nd order only */
    double ctime;               /* time systems only */
    double Q[ORC_MAX_NQ * ORC_MAX_NQ]; /* row-major n_Q x n_Q precision */
    int dist;                   /* 1 = PosOrnKeypointDistFunct (PosOrnKeypointDistFunct.h:15-44) */
    double pos_radius, orn_thresh[3];
    /* the keypoint's sub-system sees the robot through a TransformedSimulationInterface (sim/TransformedSimulationInterface.cpp:
     * 53-103): pose and Jacobian expressed in the frame T = [fR | fp] (fR row-major) */
    int has_frame;
    double fR[9], fp[3];
    /* control penalty of the OWNING sub-system of a SequentialSystem (its cost adds u'R_sub u at its keypoint steps,
     * SequentialSystem.cpp:144-150 -> System.cpp:221); 0 = the system's R_diag */
    int has_Ru;
    double Ru[ORC_MAX_NU];
    double jt[ORC_MAX_NX];      /* AngularKeypoint target (joint space, AngularKeypoint.cpp:15-27) */
    /* 1 = an Angular(Time)Keypoint inside a PosOrn(Time) system: the keypoint of a JointSpace(Time)PlannerSys sub-system of a
     * SequentialSystem (HYBRID_SYS*.ipynb; nb_deriv = 1).  Its sub-system has f(x) = x, J = I (JointSpacePlannerSys.cpp:77-81), so its
     * residual is jt - x with an n_x x n_x precision (stored in Q with leading dimension n_x). */
    int joint;
} orc_keypoint;

typedef struct {
    orc_chain chain;
    int kind;       /* ORC_SYS_* */
    int nb_deriv;   /* 1 or 2 */
    int T;          /* horizon */
    double dt;      /* PosOrn only */
    double R_diag[ORC_MAX_NU];
    int limits_set;
    int lim_mult;                         /* SequentialSystem: every sub-system adds the limit terms once (SequentialSystem.cpp:144-168); 0 = 1 */
    double penalty;                       /* System.cpp:40,72 */
    double state_max[ORC_MAX_NX], state_min[ORC_MAX_NX];
    int limit_weight[ORC_MAX_NX];
    double q0[ORC_MAX_DOF], dq0[ORC_MAX_DOF];
    int n_kp;
    orc_keypoint kp[ORC_MAX_KP];
    /* derived by orc_system_finalize() (localInit, PosOrnPlannerSys.cpp:54-78 / PosOrnTimePlannerSys.cpp:50-83) */
    int dof, n_x, n_u, n_f, n_Q;
    /* SequentialSystem whose sub-systems have different bounds (HYBRID_SYS_TIME.ipynb: (qMax, qMin) and (qMax, -qMax)): the second
     * group's limits; cost, cost_x, cost_xx add them lim2_mult times (SequentialSystem.cpp:143-165 sums the sub-systems) */
    int limits2_set, lim2_mult, sequence;  /* sequence: the batch solvers see no limit terms even when lim_mult is 1 */
    double state_max2[ORC_MAX_NX], state_min2[ORC_MAX_NX];
    int limit_weight2[ORC_MAX_NX];
} orc_system;

void orc_system_finalize(orc_system* s);

/* KDLRobot::updateKinematics (src/sim/KDLRobot.cpp:83-115) */
void orc_fk(const orc_chain* c, const double* q, const double* dq,
            double p[3], double quat[4], double J[6 * ORC_MAX_DOF], double dx[3], double w[3]);

/* include/ilqr_planner/utils/sd.h */
void orc_sd_H(const double q[4], double H[12]);                              /* :23-27 */
double orc_sd_distance(const double x[4], const double y[4]);                /* :48-62 */
void orc_sd_logmap(const double base[4], const double y[4], double out[4]);  /* :67-82 */
void orc_sd_expmap(const double base[4], const double u[4], double out[4]);  /* :32-43 */
void orc_sd_transport(const double v[4], const double b1[4], const double b2[4], double out[4]); /* :87-99 */

/* System virtuals */
void orc_get_fx_jac(const orc_system* s, const double* x, double* fx, double* J /* n_Q x n_x, may be NULL */);
void orc_kp_diff(const orc_system* s, const orc_keypoint* kp, const double* fx, double* e);
double orc_cost(const orc_system* s, const double* x, const double* u, int k);
void orc_cost_x(const orc_system* s, const double* x, int k, double* lx);
void orc_cost_xx(const orc_system* s, const double* x, int k, double* lxx);
/* forwardPass: x_next, fx_next, A (n_x^2), B (n_x*n_u), J (n_Q*n_x); any output may be NULL */
void orc_step(const orc_system* s, const double* x, const double* u,
              double* x_next, double* fx_next, double* A, double* B, double* J);

/* Solvers.  All trajectories row-major [t][dim].  trace_* have nb_iter entries; returns iterations run. */
int orc_solve_recursive(const orc_system* s, const double* U0, int nb_iter, int line_search, int early_stop,
                        double* X, double* fX, double* U, double* K, double* d, double* cost,
                        double* trace_cost, double* trace_alpha);

typedef struct {
    int m;              /* rows of each A_k */
    int per_step;       /* 0: one (A,b) for all k; 1: A,b are [T-1][...] */
    const double* A;    /* m x (n_x+n_u) row-major */
    const double* b;    /* m */
} orc_constraints;

int orc_solve_al(const orc_system* s, const orc_constraints* c, double* lambda /* [T-1][m], in/out */,
                 const double* U0, int nb_iter, int lag_update_step, double penalty, double scaling,
                 int line_search, int early_stop,
                 double* X, double* fX, double* U, double* cost, double* trace_cost, double* trace_alpha);

/* BatchILQRCP::solve; Qbig = block-diag keypoint precisions built from the system (getQMatrix(true)).
 * psi: ((T-1) n_u) x Kw row-major.  u: in = u0, out = solution ((T-1) n_u). */
int orc_solve_batch_cp(const orc_system* s, const double* psi, int Kw, double* u, int nb_iter, int early_stop,
                       double* trace_cost, double* trace_alpha);

/* primitives.cpp:19-96; out is dim x K (x 2K for linear) row-major */
void orc_psi_rbf(int dim, int K, double* out);
void orc_psi_bernstein(int dim, int K, double* out);
void orc_psi_unitstep(int dim, int K, double* out);
void orc_psi_sawtooth(int dim, int K, double* out);
void orc_psi_linear(int dim, int K, double* out);

/* general inverse by LU with partial pivoting (Eigen MatrixXd::inverse() = PartialPivLU). returns 0 ok */
int orc_inverse(int n, const double* A, double* Ainv);
void orc_set_variant(int v);  /* test aid, algebraically neutral variants: bit 0 = Qxu := Qux^T in the backward sweep; bit 1 = inverses by the
                                 pivot-free symmetric sweep operator; bit 2 = products accumulated with fused multiply-adds */

/* ---- test aids for the per-instance parity proof (tests/parity_proof.py).  Not part of the restated algorithm: they only RECORD the
 * quantities the reference's discontinuous decisions are taken on, and let a solve resume from a state handed in from outside.
 * Thread-local: set and solve on the same thread. */
#define ORC_MAX_TRIALS 16
typedef struct {
    double cost0;                        /* cost of the trajectory the iteration starts from (ILQRRecursive.cpp:155 compares against it) */
    int n_trials;                        /* line-search trials recorded (the last one is the accepted one unless orc_set_probe_all is on) */
    double trial_alpha[ORC_MAX_TRIALS];
    double trial_cost[ORC_MAX_TRIALS];   /* newCost of every trial */
    double dun;                          /* sum_k ||du_k|| of the accepted trial (early-stop test) */
    double mask_margin_in;               /* AL: the same margin on the trajectory the iteration STARTS from (its mask feeds this iteration's sweep) */
    double mask_margin;                  /* AL: min |g| over rows with lambda == 0 on the accepted rollout (AL-ILQR.cpp:38-42); inf if none */
    double clamp_margin;                 /* AL, update iterations: min |lambda + penalty g| over all rows (AL-ILQR.cpp:205 cwiseMax(0)); inf else */
    double limit_margin_in;              /* the same on the trajectory the iteration starts from (l_xx of this iteration's sweep) */
    double limit_margin;                 /* min distance of a weighted state coordinate of the accepted rollout to its bound (l_xx jumps there) */
} orc_probe_rec;
void orc_set_probe(orc_probe_rec* buf, int cap); /* buf[it] is filled for it < cap; NULL switches the probe off */
/* Resume an AL / recursive solve at iteration index it0 (the multiplier-update phase (it+1) % lag counts from it0); for AL the active-set
 * weights of the trajectory handed in are formed with init_penalty and the multipliers lambda_mask ([T-1][m], the multipliers in force when
 * that trajectory was rolled out) -- AL-ILQR.cpp:190 stores penalty * I_k at rollout time, before the update of :202-208.  it0 = 0 and
 * lambda_mask = NULL restore the plain behaviour. */
void orc_set_resume(int it0, double init_penalty, const double* lambda_mask);
/* Test aid: with on != 0 a probed line search (recursive / AL / Batch-CP) goes on evaluating the step sizes BELOW the accepted one,
 * recording their costs in the probe; the solve's result is unchanged (the accepted rollout is put aside and restored). */
void orc_set_probe_all(int on);
/* Test aid for resumed solves: X[T][n_x] = the caller's own rollout of the controls handed over.  The oracle rolls the controls out
 * itself, records the largest relative deviation from X (orc_get_resume_x_dev) and then takes X as the incoming trajectory, so that
 * its active-set / limit tests are taken on the very numbers the caller's were.  NULL switches it off. */
void orc_set_resume_x(const double* X);
double orc_get_resume_x_dev(void);

#ifdef __cplusplus
}
#endif
#endif
