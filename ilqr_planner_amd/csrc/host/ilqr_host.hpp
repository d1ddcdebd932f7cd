// ilqr_host.hpp -- C++ host mirror of the reference's user-facing classes for the batched iLQR hot path.
//
// Same names, argument meaning and error behaviour (std::runtime_error) as idiap/ilqr_planner:
//   sim::SimulationInterface / sim::KDLRobot          include/ilqr_planner/sim/{SimulationInterface,KDLRobot}.h
//   sys::Keypoint / PosOrnKeypoint / SpacetimeKeypoint include/ilqr_planner/system/*Keypoint.h
//   sys::System / PosOrnPlannerSys / PosOrnTimePlannerSys  include/ilqr_planner/system/{System,PosOrn*PlannerSys}.h
//   solver::ILQRRecursive / AL_ILQR / BatchILQRCP      include/ilqr_planner/solver/*.h
//   primitives, Sd, CallBackMessage                   include/ilqr_planner/utils/*.h
// No Eigen (absent from this image): Vec/Mat are plain row-major containers.  These classes hold state and LOWER it to
// the POD descriptor of include/ilqr_hip.h; every solve and every kinematics evaluation runs on the GPU through that C
// ABI -- there is no host solver for them (user-defined System / Keypoint subclasses, which have no descriptor, run
// solver::ILQRRecursive over their own virtuals: ilqr_host_loop.cpp).  The single-point System API (forwardPass, cost*, getFxJac, fpBatch ...) is host glue over
// the simulator for users who call it directly; the solvers never use it.  NOT mirrored: BatchILQR (no PSI), LQT -- see DESIGN.md.
#pragma once

#include <memory>
#include <typeinfo>
#include <string>
#include <tuple>
#include <vector>

#include "../../../include/ilqr_hip.h"

namespace ilqr_planner {

using Vec = std::vector<double>;
struct Mat {
    int rows = 0, cols = 0;
    std::vector<double> d;  // row-major
    Mat() {}
    Mat(int r, int c) : rows(r), cols(c), d((size_t)r * c, 0.0) {}
    double& operator()(int r, int c) { return d[(size_t)r * cols + c]; }
    double operator()(int r, int c) const { return d[(size_t)r * cols + c]; }
    static Mat Identity(int n) { Mat m(n, n); for (int i = 0; i < n; i++) m(i, i) = 1; return m; }
};

class CallBackMessage {  // include/ilqr_planner/utils/CallbackMessage.h:12-16
public:
    virtual ~CallBackMessage() {}
    virtual void notify(const std::string& msg) = 0;
};

// process-wide device context (one MI355X per process, as one-rank-per-GPU launches expect)
ilqr_ctx* device_context();
void check(int rc);  // non-zero -> std::runtime_error(ilqr_last_error)

namespace Sd {  // include/ilqr_planner/utils/sd.h
Mat dQuatToDxJac(const Vec& q);
Vec expMap(Vec base, const Vec& u);
double distance(const Vec& x, const Vec& y);
Vec logMap(Vec base, Vec y);
Vec transport(const Vec& v, const Vec& base1, const Vec& base2);
}  // namespace Sd

Mat buildPsiRBF(int dim, int K);  // src/utils/primitives.cpp:19-96
Mat buildPsiBernstein(int dim, int K);
Mat buildPsiUnitstep(int dim, int K);
Mat buildPsiSawtooth(int dim, int K);
Mat buildPsiLinear(int dim, int K);

namespace sim {
class SimulationInterface {  // SimulationInterface.h:13-124
public:
    virtual ~SimulationInterface() {}
    virtual void updateKinematics() = 0;
    Mat Jt();
    Mat Jr();
    virtual Mat J() { return Jac; }
    Mat Jtp();  // rows of Jp(), SimulationInterface.cpp:41-47
    Mat Jrp();
    virtual Mat Jp() { return dJac; }  // dJ/dt (:53-55)
    Mat dQuatToDxJac(const Vec& quat) { return Sd::dQuatToDxJac(quat); }
    virtual void sendAcc(double dt, const Vec& ddq, bool updateKin = true);
    virtual void sendVel(double dt, const Vec& dq, bool updateKin = true);
    virtual Vec getEEPosition() { return x; }
    virtual Vec getEEVelocity() { return dx; }
    virtual Vec getEEAngVel() { return w; }
    Vec getEEAngVelQuat();
    virtual Vec getEEOrnQuat() { return ornQuat; }
    Vec getJointsPos() { return q; }
    Vec getJointsVel() { return dq; }
    int getDOF() { return dof; }
    int getNbCarDim() { return nbCarDim; }
    double getTime() { return t; }
    virtual void setTime(double time) { t = time; }
    virtual void setConfiguration(const Vec& q, const Vec& dq, bool reset_time = true);
    // lowering hook: fill the chain part of the descriptor; false if this simulator cannot run on the device
    virtual bool lowerChain(ilqr_problem_desc*) const { return false; }
    // Is this object EXACTLY one of the simulators of this header (and, for a wrapper, is the wrapped one)?  A user subclass -- also one of
    // sim::KDLRobot that overrides the kinematics or the integrator -- inherits lowerChain() but not this: the device would solve it with
    // the base class's chain and dynamics, so the solvers take such a simulator over its virtuals instead (System::builtin()).
    virtual bool builtinSim() const { return false; }
    // object frame this simulator reports poses in (TransformedSimulationInterface): row-major R[9], p[3]; false = base frame
    virtual bool frame(double*, double*) const { return false; }

protected:
    Vec q, dq, ddq, x, dx, ornQuat, w;
    Mat Jac, dJac;
    int dof = 0, nbCarDim = 3;
    double t = 0;
};

class KDLRobot : public SimulationInterface {  // KDLRobot.h:22-57
public:
    KDLRobot(const std::string& urdf, const std::string& baseFrame, const std::string& tipFrame, const Vec& q, const Vec& dq,
             const Vec& transform_rpy, const Vec& transform_xyz, const bool& is_path);
    KDLRobot(const std::string& urdf, const std::string& baseFrame, const std::string& tipFrame, const Vec& q, const Vec& dq)
        : KDLRobot(urdf, baseFrame, tipFrame, q, dq, Vec(3, 0.0), Vec(3, 0.0), true) {}
    KDLRobot(const std::string& urdf, const std::string& baseFrame, const std::string& tipFrame, const Vec& q, const Vec& dq,
             const Vec& transform_rpy, const Vec& transform_xyz)
        : KDLRobot(urdf, baseFrame, tipFrame, q, dq, transform_rpy, transform_xyz, true) {}
    void updateKinematics() override;  // one-configuration call of ilqr_fk_batch (the FK kernel), then dx = Jt dq, w = Jr dq
    bool lowerChain(ilqr_problem_desc* d) const override;
    bool builtinSim() const override { return typeid(*this) == typeid(KDLRobot); }
    Vec jointLowerLimits() const { return lower_; }
    Vec jointUpperLimits() const { return upper_; }

protected:
    ilqr_problem_desc chain_;  // only the chain fields are meaningful
    Vec lower_, upper_;
};
// Planar arm of 2DRobot.h / 2DRobot.cpp:13-74 (host only: x = sum_i l_i [cos q_i, sin q_i] with ABSOLUTE joint angles, Jacobian by
// forward differences of step pi*1e-3, J = [Jt; 0] (4 x dof), identity quaternion).  Only joint-space systems work with it.
class Robot2D : public SimulationInterface {
public:
    Robot2D(const Vec& lengths, const Vec& default_q);
    Vec fkine(const Vec& q);
    Vec fkine() { return fkine(q); }
    void updateKinematics() override;
    bool builtinSim() const override { return typeid(*this) == typeid(Robot2D); }

protected:
    Vec lengths_;
};

// The wrapped robot seen from an object frame T (4x4 pose): p' = R'(p - t), R_ee' = R' R_ee, J' = blkdiag(R,R)' J
// (TransformedSimulationInterface.h / .cpp:20-103).  The Python constructor takes (robot, T) as the reference's binding does.
class TransformedSimulationInterface : public SimulationInterface {
public:
    TransformedSimulationInterface(const std::shared_ptr<SimulationInterface>& r, const Mat& T);
    void updateKinematics() override;
    Mat J() override;
    Mat Jp() override;
    Vec getEEPosition() override;
    Vec getEEVelocity() override;
    Vec getEEAngVel() override;
    Vec getEEOrnQuat() override;
    void sendAcc(double dt, const Vec& ddq, bool updateKin = true) override;
    void sendVel(double dt, const Vec& dq, bool updateKin = true) override;
    void setConfiguration(const Vec& q, const Vec& dq, bool reset_time = true) override;
    void setTime(double time) override;
    bool lowerChain(ilqr_problem_desc* d) const override { return r_->lowerChain(d); }
    bool builtinSim() const override { return typeid(*this) == typeid(TransformedSimulationInterface) && r_ && r_->builtinSim(); }
    bool frame(double* R, double* p) const override;

protected:
    std::shared_ptr<SimulationInterface> r_;
    Mat T_;
};
}  // namespace sim

namespace sys {
class Keypoint {  // Keypoint.h:15-41
public:
    enum KpType { FIRST_ORDER = 1, SECOND_ORDER = 2 };
    Keypoint(int timestep, KpType type, const std::string& TAG) : TAG_(TAG), type_(type), timestep_(timestep) {}
    virtual ~Keypoint() {}
    virtual Vec diff(const Vec& state) const = 0;
    virtual Vec getState() const = 0;
    virtual Mat getPrecision() const = 0;
    virtual Vec targetFx() const = 0;  // target in f(x) order [p, quat, (dp, dquat), (t)] -- what the device consumes
    int getTimestep() const { return timestep_; }
    std::string getTAG() const { return TAG_; }
    KpType getType() const { return type_; }

protected:
    std::string TAG_;
    KpType type_;
    int timestep_;
};

class PosOrnKeypoint : public Keypoint {  // PosOrnKeypoint.h:15-60
public:
    PosOrnKeypoint(const Vec& position, const Vec& orientation, const Mat& precision, const int& timestep);
    PosOrnKeypoint(const Vec& position, const Vec& dposition, const Vec& orientation, const Vec& dorientation, const Mat& precision, const int& timestep);
    Vec getPosition() const { return position_; }
    Vec getOrientation() const { return orientation_; }
    Mat getPrecision() const override { return precision_; }
    Vec diff(const Vec& state) const override;
    Vec getState() const override;
    Vec targetFx() const override;

protected:
    Vec position_, orientation_, dposition_, dorientation_;
    Mat precision_;
    int state_size_;
};

class PosOrnKeypointDistFunct : public PosOrnKeypoint {  // PosOrnKeypointDistFunct.h:15-44
public:
    PosOrnKeypointDistFunct(const Vec& position, const Vec& orientation, const Mat& precision, const double& pos_radius, const Vec& orn_thresh, const int& timestep);
    PosOrnKeypointDistFunct(const Vec& position, const Vec& dposition, const Vec& orientation, const Vec& dorientation, const Mat& precision,
                            const double& pos_radius, const Vec& orn_thresh, const int& timestep);
    Vec diff(const Vec& state) const override;  // PosOrnKeypointDistFunct.cpp:13-35
    double getPosRadius() const { return pos_radius_; }
    Vec getOrnThresh() const { return orn_thresh_; }

protected:
    double pos_radius_;
    Vec orn_thresh_;
};

class AngularKeypoint : public Keypoint {  // AngularKeypoint.h:15-45: joint-space target, residual = target - state
public:
    AngularKeypoint(const Vec& position, const Mat& precision, const int& timestep)
        : Keypoint(timestep, FIRST_ORDER, "JNT"), position_(position), precision_(precision) {}
    AngularKeypoint(const Vec& position, const Vec& dposition, const Mat& precision, const int& timestep)
        : Keypoint(timestep, SECOND_ORDER, "JNT"), position_(position), dposition_(dposition), precision_(precision) {}
    Vec getPosition() const { return position_; }
    Mat getPrecision() const override { return precision_; }
    Vec diff(const Vec& state) const override;
    Vec getState() const override;
    Vec targetFx() const override { return getState(); }

protected:
    Vec position_, dposition_;
    Mat precision_;
};

class AngularTimeKeypoint : public AngularKeypoint {  // AngularTimeKeypoint.h: joint target + continuous time
public:
    AngularTimeKeypoint(const Vec& position, const Mat& precision, const double& continuous_time, const int& timestep)
        : AngularKeypoint(position, precision, timestep), continuous_time_(continuous_time) { TAG_ = "JNT_TIME"; }
    AngularTimeKeypoint(const Vec& position, const Vec& dposition, const Mat& precision, const double& continuous_time, const int& timestep)
        : AngularKeypoint(position, dposition, precision, timestep), continuous_time_(continuous_time) { TAG_ = "JNT_TIME"; }
    double getContinuousTime() { return continuous_time_; }
    Vec diff(const Vec& state) const override;
    Vec getState() const override;

protected:
    double continuous_time_;
};

class SpacetimeKeypoint : public PosOrnKeypoint {  // SpacetimeKeypoint.h:15-45
public:
    SpacetimeKeypoint(const Vec& position, const Vec& orientation, const Mat& precision, const double& continuous_time, const int& timestep);
    SpacetimeKeypoint(const Vec& position, const Vec& dposition, const Vec& orientation, const Vec& dorientation, const Mat& precision,
                      const double& continuous_time, const int& timestep);
    double getContinuousTime() { return continuous_time_; }
    Vec diff(const Vec& state) const override;
    Vec getState() const override;
    Vec targetFx() const override;

protected:
    double continuous_time_;
};

class System {  // System.h:28-194
public:
    System(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
           const Vec& qMax, const Vec& qMin, const Vec& dqMax, const Vec& dqMin, int horizon, int nb_deriv, const std::vector<std::string>& tags);
    System(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
           int horizon, int nb_deriv, const std::vector<std::string>& tags);
    virtual ~System() {}
    virtual Vec getMuVector(bool sparse);
    virtual Mat getQMatrix(bool sparse);
    Mat getRt();
    int getNbStateVar() { return nb_state_var_; }
    int getNbCtrlVar() { return nb_ctrl_var_; }
    int getNbTargetVar() { return nb_target_var_; }
    int getNbQVar() { return nb_Q_var_; }
    int getHorizon() { return horizon_; }
    std::vector<int> getKpIndexes();
    Vec getInitState() { return x0_; }
    Vec getInitFoXState() { return f_x0_; }
    virtual Vec getState() = 0;
    virtual void reset() = 0;
    // ---- single-point evaluation API of System.h:66-194 (host glue over the simulator, whose kinematics run on the device).
    // The solvers never call these: they lower the system and run the batched kernels.
    typedef std::tuple<Vec, Vec, Mat, Mat, Mat> StepOut;                          // x, f(x), A, B, J
    typedef std::tuple<Vec, Vec, Vec, Vec, Mat, Mat, Mat, Mat> StepLimitsOut;     // x, f(x), q, u, A, B, J, L
    virtual StepOut forwardPass(const Vec& xk, const Vec& uk, int k) = 0;         // advances the simulator (xk is ignored as upstream)
    virtual std::tuple<Vec, Mat> getFxJac() = 0;
    virtual std::tuple<Vec, Mat> getFxJac(const Vec& xk);                         // System.cpp:163-179
    virtual Vec diff(const Vec& actual_state, int k);                             // :103-109
    Vec diffBatch(const Vec& x);                                                  // :111-119
    std::pair<Mat, Vec> inspectJointLimit(const Vec& xk);                         // :121-142
    StepLimitsOut forwardPassWithLimits(const Vec& xk, const Vec& uk, int k);     // :144-161
    std::tuple<Vec, Vec, std::vector<std::tuple<Mat, Mat, Mat, Mat>>> fpBatch(const Vec& u);  // :181-211
    virtual Vec cost(const Vec& xk, const Vec& uk, int k);                        // :213-234
    virtual Vec cost_x(const Vec& xk, const Vec& uk, int k);                      // :248-272
    virtual Mat cost_xx(const Vec& xk, const Vec& uk, int k);                     // :286-308
    Vec cost_u(const Vec&, const Vec& uk, int) { Vec o(uk.size()); for (size_t i = 0; i < uk.size(); i++) o[i] = Rdiag[i] * uk[i]; return o; }
    Mat cost_uu(const Vec&, const Vec&, int) { return getRt(); }
    Mat cost_ux(const Vec&, const Vec& uk, int) { return Mat((int)uk.size(), nb_state_var_); }
    Mat cost_xu(const Vec&, const Vec& uk, int) { return Mat(nb_state_var_, (int)uk.size()); }
    virtual Vec cost_F(const Vec& xk) { return cost(xk, Vec(nb_ctrl_var_, 0.0), horizon_ - 1); }
    virtual Vec cost_F_x(const Vec& xk) { return cost_x(xk, Vec(nb_ctrl_var_, 0.0), horizon_ - 1); }
    virtual Mat cost_F_xx(const Vec& xk) { return cost_xx(xk, Vec(nb_ctrl_var_, 0.0), horizon_ - 1); }
    std::shared_ptr<Keypoint> getKeypoint(int k) const;                           // :96-101 (the last of duplicates wins, as in the map)
    std::shared_ptr<sim::SimulationInterface> robot() { return r; }
    const std::vector<std::shared_ptr<Keypoint>>& getKeypoints() const { return keypoints; }
    // Lowering to the C ABI's POD descriptor.  Throws for systems the device cannot run.
    virtual void lower(ilqr_problem_desc* d) const;
    virtual bool builtinType() const { return false; }  // overridden by the classes of this header: typeid(*this) == typeid(<that class>)
    // per-instance pieces for B = 1: q0, dq0 captured by localInit
    Vec q0() const { return q0_; }
    Vec dq0() const { return dq0_; }
    int getNbDeriv() const { return nb_deriv_; }
    // joint-space systems of robots with fewer than 7 joints are padded to the device's 7 (zero precision, zero limit weight, u = 0):
    // number of joints the user sees, 0 = no padding
    virtual int paddedFromDof() const { return 0; }
    // true for the system classes of this header exactly as declared here (with built-in keypoints): the shapes the device descriptor
    // describes.  A user-defined subclass -- also one derived from a built-in class -- answers false and is solved over its virtuals by
    // ILQRRecursive::solve (ilqr_host_loop.cpp); the choice is made on the type, never on whether a device call succeeded.
    bool builtin() const;
    const Vec& Rt() const { return Rdiag; }
    int kind() const { return kind_; }

protected:
    void init();
    void checkKeypoints();
    std::shared_ptr<sim::SimulationInterface> r;
    std::vector<std::shared_ptr<Keypoint>> keypoints;
    Vec Rdiag;
    Vec state_max_, state_min_;
    std::vector<int> joint_limits_weight_;
    Vec q0_, dq0_, x0_, f_x0_;
    int horizon_, nb_deriv_;
    int nb_state_var_ = 0, nb_ctrl_var_ = 0, nb_target_var_ = 0, nb_Q_var_ = 0;
    double penalty_ = 0, dt_ = 0;
    bool limits_set_ = false;
    int kind_ = ILQR_SYS_POS_ORN;
    std::vector<std::string> EXPECTED_KP_TAGS_;
};

class PosOrnPlannerSys : public System {  // PosOrnPlannerSys.h
public:
    bool builtinType() const override { return typeid(*this) == typeid(PosOrnPlannerSys); }
    PosOrnPlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
                     const Vec& qMax, const Vec& qMin, const Vec& dqMax, const Vec& dqMin, int horizon, int nb_deriv, double dt);
    PosOrnPlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
                     const Vec& qMax, const Vec& qMin, int horizon, int nb_deriv, double dt);
    PosOrnPlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
                     int horizon, int nb_deriv, double dt);
    Vec getState() override;
    void reset() override;
    StepOut forwardPass(const Vec& xk, const Vec& uk, int k) override;
    std::tuple<Vec, Mat> getFxJac() override;
    using System::getFxJac;

protected:
    void localInit(double dt);
};

// Target space = joint space, J = I (JointSpacePlannerSys.h / .cpp:50-122).  Device path: nb_deriv = 1 and the 7 joints the
// kernels are built for (the reference's 2nd-order variant is dimensionally inconsistent).
class JointSpacePlannerSys : public System {
public:
    bool builtinType() const override { return typeid(*this) == typeid(JointSpacePlannerSys); }
    JointSpacePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
                         const Vec& qMax, const Vec& qMin, const Vec& dqMax, const Vec& dqMin, int horizon, int nb_deriv, double dt);
    JointSpacePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
                         const Vec& qMax, const Vec& qMin, int horizon, int nb_deriv, double dt);
    JointSpacePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
                         int horizon, int nb_deriv, double dt);
    Vec getState() override;
    void reset() override;
    StepOut forwardPass(const Vec& xk, const Vec& uk, int k) override;
    std::tuple<Vec, Mat> getFxJac() override;
    using System::getFxJac;
    void lower(ilqr_problem_desc* d) const override;
    int paddedFromDof() const override { return r->getDOF() < 7 ? r->getDOF() : 0; }

protected:
    void localInit(double dt);
};

// JointSpacePlannerSys with a time state and dt = u_last^2 (JointSpaceTimePlannerSys.h / .cpp:50-160); nb_deriv = 1, 7 joints on the device
class JointSpaceTimePlannerSys : public System {
public:
    bool builtinType() const override { return typeid(*this) == typeid(JointSpaceTimePlannerSys); }
    JointSpaceTimePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
                             const Vec& qMax, const Vec& qMin, const Vec& dqMax, const Vec& dqMin, int horizon, int nb_deriv);
    JointSpaceTimePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
                             const Vec& qMax, const Vec& qMin, int horizon, int nb_deriv);
    JointSpaceTimePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
                             int horizon, int nb_deriv);
    Vec getState() override;
    void reset() override;
    StepOut forwardPass(const Vec& xk, const Vec& uk, int k) override;
    std::tuple<Vec, Mat> getFxJac() override;
    std::tuple<Vec, Mat> getFxJac(const Vec& xk) override;
    void lower(ilqr_problem_desc* d) const override;
    int paddedFromDof() const override { return r->getDOF() < 7 ? r->getDOF() : 0; }

protected:
    void localInit();
};

class PosOrnTimePlannerSys : public System {  // PosOrnTimePlannerSys.h
public:
    bool builtinType() const override { return typeid(*this) == typeid(PosOrnTimePlannerSys); }
    PosOrnTimePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
                         const Vec& qMax, const Vec& qMin, const Vec& dqMax, const Vec& dqMin, int horizon, int nb_deriv);
    PosOrnTimePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
                         const Vec& qMax, const Vec& qMin, int horizon, int nb_deriv);
    PosOrnTimePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& keypoints, const Vec& RtDiag,
                         int horizon, int nb_deriv);
    Vec getState() override;
    void reset() override;
    StepOut forwardPass(const Vec& xk, const Vec& uk, int k) override;
    std::tuple<Vec, Mat> getFxJac() override;
    std::tuple<Vec, Mat> getFxJac(const Vec& xk) override;

protected:
    void localInit();
};
// Sum of the costs of several Systems sharing one robot (SequentialSystem.h / .cpp:20-168): dynamics of the first, keypoints of
// all (each in its own sub-system's frame, with that sub-system's control penalty), limit terms once per sub-system.
class SequentialSystem : public System {
public:
    bool builtinType() const override;  // ... and every sub-system is
    SequentialSystem(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<System>>& systems, const Vec& RtDiag,
                     int horizon, int nb_deriv);
    Vec getState() override { return systems_.at(0)->getState(); }
    void reset() override;
    StepOut forwardPass(const Vec& xk, const Vec& uk, int k) override;
    std::tuple<Vec, Mat> getFxJac() override;
    using System::getFxJac;
    Vec diff(const Vec& state, int k) override;                       // SequentialSystem.cpp:167-183
    Vec cost(const Vec& xk, const Vec& uk, int k) override;           // :143-149 (sums of the sub-systems, as are the five below)
    Vec cost_x(const Vec& xk, const Vec& uk, int k) override;
    Mat cost_xx(const Vec& xk, const Vec& uk, int k) override;
    Vec cost_F(const Vec& xk) override;
    Vec cost_F_x(const Vec& xk) override;
    Mat cost_F_xx(const Vec& xk) override;
    Vec getMuVector(bool sparse) override;                            // :185-226
    Mat getQMatrix(bool sparse) override;                             // :228-274 (block diagonals of the sub-systems)
    void lower(ilqr_problem_desc* d) const override;
    const std::vector<std::shared_ptr<System>>& systems() const { return systems_; }

protected:
    std::vector<std::shared_ptr<System>> systems_;
};
}  // namespace sys

namespace solver {

// Result of a batched solve: instance b is what the reference's solve() would have returned for that instance.
struct BatchResult {
    int B = 0, T = 0, n_x = 0, n_u = 0, n_f = 0;
    std::vector<double> X, fX, U, K, d;  // [B][T][n_x], [B][T][n_f], [B][T-1][n_u], [B][T-1][n_u][n_x], [B][T-1][n_u]
    std::vector<double> cost, alpha;     // [B]
    std::vector<int> iters, status;      // [B]
    std::vector<double> cost_trace, alpha_trace;  // [B][nb_iter]
    int nb_iter = 0;
    double seconds = 0;
};

// Per-instance inputs of a batched solve; anything left empty takes the System's own value for every instance.
struct BatchInputs {
    int B = 1;
    std::vector<double> q0, dq0;                    // [B][dof]
    std::vector<std::vector<double>> kp_targets;    // per keypoint: [B][n_f] in f(x) order
    std::vector<double> U0;                         // [B][T-1][n_u]  (or [T-1][n_u], broadcast)
    bool want_fX = true, want_gains = true;
};

struct Constraint {  // AL-ILQR.h:20-23
    Mat A;
    Vec b;
};

// ILQRRecursive over the virtual interface of a user-defined system (ilqr_host_loop.cpp)
std::tuple<std::vector<Vec>, std::vector<Vec>, std::vector<Vec>, std::vector<Mat>, std::vector<Vec>, double> solve_over_virtuals(
    sys::System& s, const std::vector<Vec>& U0, int nb_iter, bool line_search, bool early_stop, CallBackMessage* cb);

// AL_ILQR over the same interface (multipliers updated in place, as the reference's member is)
std::tuple<std::vector<Vec>, std::vector<Vec>, std::vector<Vec>> solve_al_over_virtuals(sys::System& s, const std::vector<Constraint>& inequality, std::vector<Vec>& multipliers,
                                                                                        const std::vector<Vec>& U0, int nb_iter, int lag_update_step, double penalty,
                                                                                        double scaling_factor, bool line_search, bool early_stop, CallBackMessage* cb);

// BatchILQR (psi = nullptr) / BatchILQRCP over the same interface: one Gauss-Newton step on the whole control sequence per iteration
Vec solve_batch_over_virtuals(sys::System& s, const Mat* psi, const Mat& Q, int nb_iter, const Vec& u0, bool early_stop, CallBackMessage* cb);

class ILQRRecursive {  // ILQRRecursive.h:21-42
public:
    explicit ILQRRecursive(const std::shared_ptr<sys::System>& s) : s(s) {}
    std::tuple<std::vector<Vec>, std::vector<Vec>, std::vector<Vec>, std::vector<Mat>, std::vector<Vec>, double> solve(
        const std::vector<Vec>& U0, int nb_iter, bool line_search = true, bool early_stop = true, CallBackMessage* cb = nullptr);
    BatchResult solveBatch(const BatchInputs& in, int nb_iter, bool line_search = true, bool early_stop = true);

private:
    std::shared_ptr<sys::System> s;
};

class AL_ILQR {  // AL-ILQR.h:25-72
public:
    AL_ILQR(const std::shared_ptr<sys::System>& s, const std::vector<Constraint>& inequality, const std::vector<Vec>& initLambda);
    std::tuple<std::vector<Vec>, std::vector<Vec>, std::vector<Vec>> solve(const std::vector<Vec>& U0, int nb_iter, int lag_update_step, double penalty,
                                                                          double scaling_factor, bool line_search = true, bool early_stop = false,
                                                                          CallBackMessage* cb = nullptr);
    BatchResult solveBatch(const BatchInputs& in, int nb_iter, int lag_update_step, double penalty, double scaling_factor, bool line_search = true,
                           bool early_stop = false);

private:
    std::shared_ptr<sys::System> s;
    std::vector<Constraint> inequality;
    std::vector<Vec> multipliers;  // persists across solve() calls like the reference's member
};

class BatchILQR {  // BatchILQR.h:21-53: the batch solver on the full control sequence (= BatchILQRCP with the identity basis)
public:
    BatchILQR(const std::shared_ptr<sys::System>& s, const Mat& Q);
    explicit BatchILQR(const std::shared_ptr<sys::System>& s);
    Vec solve(int nb_iter, const Vec& u0, bool early_stop = true, CallBackMessage* cb = nullptr);
    BatchResult solveBatch(const BatchInputs& in, int nb_iter, bool early_stop = true);

private:
    std::shared_ptr<sys::System> s;
    Mat Q;  // only the block-diagonal of the keypoints' precisions is supported on the device
    bool custom_Q = false;
};

class BatchILQRCP {  // BatchILQRCP.h:21-52
public:
    BatchILQRCP(const std::shared_ptr<sys::System>& s, const Mat& Q, const Mat& psi);
    BatchILQRCP(const std::shared_ptr<sys::System>& s, const Mat& psi);
    Vec solve(int nb_iter, const Vec& u0, bool early_stop = true, CallBackMessage* cb = nullptr);
    BatchResult solveBatch(const BatchInputs& in, int nb_iter, bool early_stop = true);

private:
    std::shared_ptr<sys::System> s;
    Mat PSI;
    Mat Q;  // only the block-diagonal of the keypoints' precisions is supported on the device
    bool custom_Q = false;
};
}  // namespace solver
}  // namespace ilqr_planner
