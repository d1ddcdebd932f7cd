// ilqr_host.cpp -- implementation of the host mirror (see ilqr_host.hpp).  Plumbing only: state, lowering, C-ABI calls.
#include "ilqr_host.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace ilqr_planner {

// ------------------------------------------------------------------------------------------------ device context

static ilqr_ctx* g_ctx = nullptr;

// One context per process, created when the first object needs the device.  Device choice, resolved ONCE here:
// ILQR_DEVICE if set, else LOCAL_RANK (one rank per GPU under torch.distributed.run), else 0.
ilqr_ctx* device_context() {
    if (!g_ctx) {
        int dev = 0;
        if (const char* e = std::getenv("LOCAL_RANK")) dev = std::atoi(e);
        if (const char* e = std::getenv("ILQR_DEVICE")) dev = std::atoi(e);
        const int rc = ilqr_ctx_create(dev, &g_ctx);
        if (rc) {
            g_ctx = nullptr;
            throw std::runtime_error("[ilqr_hip] no usable MI355X/HIP device " + std::to_string(dev) + " (ilqr_ctx_create code " + std::to_string(rc) +
                                     "); the solvers and the kinematics only run on the GPU");
        }
    }
    return g_ctx;
}

void check(int rc) {
    if (rc) throw std::runtime_error(ilqr_last_error(g_ctx));
}

// ------------------------------------------------------------------------------------------------ Sd (utils/sd.h)

static double vdot(const Vec& a, const Vec& b) {
    double s = 0;
    for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i];
    return s;
}
static double vnorm(const Vec& a) { return std::sqrt(vdot(a, a)); }
static bool is_zero(const Vec& a) {  // Eigen isZero(1e-12)
    for (double v : a)
        if (!(std::fabs(v) <= 1e-12)) return false;
    return true;
}

namespace Sd {
Mat dQuatToDxJac(const Vec& q) {  // sd.h:23-27
    Mat J(3, 4);
    const double r[12] = {-q[1], q[0], -q[3], q[2], -q[2], q[3], q[0], -q[1], -q[3], -q[2], q[1], q[0]};
    J.d.assign(r, r + 12);
    return J;
}
Vec expMap(Vec base, const Vec& u) {  // sd.h:32-43
    const double nb = vnorm(base);
    for (auto& v : base) v /= nb;
    const double nu = vnorm(u);
    if (nu == 0) return base;
    Vec r(base.size());
    for (size_t i = 0; i < r.size(); i++) r[i] = base[i] * std::cos(nu) + u[i] / nu * std::sin(nu);
    const double nr = vnorm(r);
    for (auto& v : r) v /= nr;
    return r;
}
double distance(const Vec& x, const Vec& y) {  // sd.h:48-62
    double d = vdot(x, y);
    if (d > 1) d = 1;
    else if (d < -1) d = -1;
    double ac = std::acos(d);
    if (d < 0) ac -= M_PI;
    return ac;
}
Vec logMap(Vec base, Vec y) {  // sd.h:67-82
    if (is_zero(base) || is_zero(y)) return Vec(base.size(), 0.0);
    const double nb = vnorm(base), ny = vnorm(y);
    for (auto& v : base) v /= nb;
    for (auto& v : y) v /= ny;
    const double by = vdot(base, y);
    Vec t(base.size());
    for (size_t i = 0; i < t.size(); i++) t[i] = y[i] - by * base[i];
    const double nt = vnorm(t);
    if (nt == 0) return Vec(base.size(), 0.0);
    const double d = distance(base, y);
    for (auto& v : t) v = d * v / nt;
    return t;
}
Vec transport(const Vec& v, const Vec& base1, const Vec& base2) {  // sd.h:87-99
    if (is_zero(base1) || is_zero(base2)) return v;
    const double dsq = std::pow(distance(base1, base2), 2);
    if (dsq == 0) return v;
    const Vec l12 = logMap(base1, base2), l21 = logMap(base2, base1);
    const double f = vdot(l12, v) / dsq;
    Vec o(v.size());
    for (size_t i = 0; i < o.size(); i++) o[i] = v[i] - f * (l12[i] + l21[i]);
    return o;
}
}  // namespace Sd

// ------------------------------------------------------------------------------------------------ primitives (utils/primitives.cpp)

static int binomial(int n, int k) { return (k == 0 || k == n) ? 1 : binomial(n - 1, k - 1) + binomial(n - 1, k); }

Mat buildPsiRBF(int dim, int K) {
    Mat psi(dim, K);
    const double bw = ((double)dim) / K, sig = bw;
    double avg = bw / 2;
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < dim; j++) psi(j, i) = 1 / (2 * M_PI * sig) * std::exp(-1 * (j - avg) * (j - avg) / (2 * sig * sig));
        avg += bw;
    }
    return psi;
}
Mat buildPsiBernstein(int dim, int K) {
    Mat psi(dim, K);
    const int order = K - 1;
    for (int i = 0; i < K; i++) {
        const int b = binomial(order, i);
        for (int j = 0; j < dim; j++) {
            const double t = ((double)j) / (dim - 1);
            psi(j, i) = b * std::pow(t, i) * std::pow(1 - t, order - i);
        }
    }
    return psi;
}
Mat buildPsiUnitstep(int dim, int K) {
    Mat psi(dim, K);
    const int bw = (int)std::round(((double)dim) / K);
    int lo = 0, hi = bw;
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < dim; j++) psi(j, i) = (j >= lo && j < hi) ? 1.0 / bw : 0;
        lo += bw;
        hi += bw;
    }
    return psi;
}
Mat buildPsiSawtooth(int dim, int K) {
    Mat psi(dim, K);
    const int bw = (int)std::ceil(((double)dim) / K);
    double lo = 0, hi = bw;
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < dim; j++) psi(j, i) = (j >= lo && j < hi) ? ((j - lo) / (bw - 1) - 0.5) : 0;
        lo += bw;
        hi += bw;
    }
    return psi;
}
Mat buildPsiLinear(int dim, int K) {
    const Mat a = buildPsiUnitstep(dim, K), b = buildPsiSawtooth(dim, K);
    Mat psi(dim, 2 * K);
    for (int j = 0; j < dim; j++)
        for (int i = 0; i < K; i++) {
            psi(j, i) = a(j, i);
            psi(j, K + i) = b(j, i);
        }
    return psi;
}

// ------------------------------------------------------------------------------------------------ sim

namespace sim {
Mat SimulationInterface::Jt() {
    Mat j = J(), o(nbCarDim, j.cols);
    for (int r = 0; r < nbCarDim; r++)
        for (int c = 0; c < j.cols; c++) o(r, c) = j(r, c);
    return o;
}
Mat SimulationInterface::Jr() {
    Mat j = J(), o(nbCarDim, j.cols);
    for (int r = 0; r < nbCarDim; r++)
        for (int c = 0; c < j.cols; c++) o(r, c) = j(j.rows - nbCarDim + r, c);
    return o;
}
static Mat rows_of_mat(const Mat& j, int r0, int n) {
    Mat o(n, j.cols);
    for (int r = 0; r < n && r0 + r < j.rows; r++)
        for (int c = 0; c < j.cols; c++) o(r, c) = j(r0 + r, c);
    return o;
}
Mat SimulationInterface::Jtp() { return rows_of_mat(Jp(), 0, nbCarDim); }
Mat SimulationInterface::Jrp() { const Mat j = Jp(); return rows_of_mat(j, j.rows - nbCarDim, nbCarDim); }
void SimulationInterface::sendAcc(double dt, const Vec& a, bool updateKin) {  // SimulationInterface.cpp:19-26
    for (int i = 0; i < dof; i++) {
        q[i] += dt * dq[i] + dt * dt / 2 * a[i];
        dq[i] += dt * a[i];
    }
    t += dt;
    if (updateKin) updateKinematics();
    ddq = a;
}
void SimulationInterface::sendVel(double dt, const Vec& v, bool updateKin) {  // :28-31
    dq = v;
    sendAcc(dt, Vec(v.size(), 0.0), updateKin);
}
Vec SimulationInterface::getEEAngVelQuat() {  // :69-73
    const Mat H = Sd::dQuatToDxJac(getEEOrnQuat());
    const Vec ww = getEEAngVel();
    Vec o(4, 0.0);
    for (int i = 0; i < 4; i++) o[i] = .5 * (H(0, i) * ww[0] + H(1, i) * ww[1] + H(2, i) * ww[2]);
    return o;
}
void SimulationInterface::setConfiguration(const Vec& q_, const Vec& dq_, bool reset_time) {  // :91-98
    q = q_;
    dq = dq_;
    updateKinematics();
    if (reset_time) t = 0;
}

KDLRobot::KDLRobot(const std::string& urdf, const std::string& baseFrame, const std::string& tipFrame, const Vec& q_, const Vec& dq_,
                   const Vec& transform_rpy, const Vec& transform_xyz, const bool& is_path) {
    dof = (int)q_.size();
    nbCarDim = 3;
    q = q_;
    dq = dq_;
    ddq = Vec(dof, 0.0);
    x = dx = w = Vec(3, 0.0);
    ornQuat = Vec(4, 0.0);
    Jac = Mat(6, dof);
    std::string text = urdf;
    if (is_path) {
        std::ifstream f(urdf);
        if (!f) throw std::runtime_error("[KDLRobot] Unable to open URDF file " + urdf);
        std::stringstream ss;
        ss << f.rdbuf();
        text = ss.str();
    }
    if (transform_rpy.size() != 3 || transform_xyz.size() != 3) throw std::runtime_error("[KDLRobot] transform_rpy / transform_xyz must have 3 entries");
    ilqr_desc_defaults(&chain_);
    lower_.assign(ILQR_MAX_SEG, 0.0);
    upper_.assign(ILQR_MAX_SEG, 0.0);
    if (ilqr_chain_from_urdf(text.c_str(), baseFrame.c_str(), tipFrame.c_str(), transform_rpy.data(), transform_xyz.data(), &chain_, lower_.data(), upper_.data()))
        throw std::runtime_error(ilqr_urdf_last_error());  // "[KDLRobot] Unable to build kinematic chain from <base> to <tip>"
    if (chain_.dof != dof)
        throw std::runtime_error("[KDLRobot] chain has " + std::to_string(chain_.dof) + " moving joints but q has " + std::to_string(dof) + " entries");
    lower_.resize(dof);
    upper_.resize(dof);
    updateKinematics();
}

void KDLRobot::updateKinematics() {  // KDLRobot.cpp:83-115
    ilqr_ctx* c = device_context();
    std::vector<double> jac(6 * dof);
    if (ilqr_fk_batch(c, &chain_, 1, q.data(), x.data(), ornQuat.data(), jac.data()))
        throw std::runtime_error(std::string("[KinModel] Error while computing Jacobian and FK! ") + ilqr_last_error(c));
    Jac = Mat(6, dof);
    Jac.d = jac;
    for (int i = 0; i < 3; i++) {
        dx[i] = 0;
        w[i] = 0;
        for (int j = 0; j < dof; j++) {
            dx[i] += Jac(i, j) * dq[j];
            w[i] += Jac(3 + i, j) * dq[j];
        }
    }
    // dJ/dt = sum_j dq_j dJ/dq_j of a geometric Jacobian with columns (v_i; w_i) (utils.h:70-112, KDLRobot.cpp:112):
    //   dJ_i/dq_j = (w_j x v_i; w_j x w_i) for j < i,  (w_i x v_j; 0) for j >= i.  Host-only; no solver reads it.
    dJac = Mat(6, dof);
    auto cross_into = [&](int row0, int col, double s, const double* a, const double* b) {
        dJac(row0 + 0, col) += s * (a[1] * b[2] - a[2] * b[1]);
        dJac(row0 + 1, col) += s * (a[2] * b[0] - a[0] * b[2]);
        dJac(row0 + 2, col) += s * (a[0] * b[1] - a[1] * b[0]);
    };
    for (int i = 0; i < dof; i++) {
        const double vi[3] = {Jac(0, i), Jac(1, i), Jac(2, i)}, wi[3] = {Jac(3, i), Jac(4, i), Jac(5, i)};
        for (int j = 0; j < dof; j++) {
            const double vj[3] = {Jac(0, j), Jac(1, j), Jac(2, j)}, wj[3] = {Jac(3, j), Jac(4, j), Jac(5, j)};
            if (j < i) {
                cross_into(0, i, dq[j], wj, vi);
                cross_into(3, i, dq[j], wj, wi);
            } else {
                cross_into(0, i, dq[j], wi, vj);
            }
        }
    }
}

bool KDLRobot::lowerChain(ilqr_problem_desc* d) const {
    d->dof = chain_.dof;
    d->n_seg = chain_.n_seg;
    std::copy(std::begin(chain_.seg_joint), std::end(chain_.seg_joint), std::begin(d->seg_joint));
    std::memcpy(d->seg_xyz, chain_.seg_xyz, sizeof(chain_.seg_xyz));
    std::memcpy(d->seg_R, chain_.seg_R, sizeof(chain_.seg_R));
    std::memcpy(d->seg_axis, chain_.seg_axis, sizeof(chain_.seg_axis));
    return true;
}

// ---- Robot2D (2DRobot.cpp:13-74)
Robot2D::Robot2D(const Vec& lengths, const Vec& default_q) : lengths_(lengths) {
    if (lengths.size() != default_q.size()) throw std::runtime_error("[Robot2D] lengths and default_q must have the same size");
    dof = (int)default_q.size();
    nbCarDim = 2;
    q = default_q;
    dq.assign(dof, 0.0);
    ddq.assign(dof, 0.0);
    x.assign(2, 0.0);
    dx.assign(2, 0.0);
    w.assign(3, 0.0);
    ornQuat = {1, 0, 0, 0};
    updateKinematics();
}
Vec Robot2D::fkine(const Vec& q_) {
    Vec xi(2, 0.0);
    for (int i = 0; i < dof; i++) {
        xi[0] += lengths_[i] * std::cos(q_[i]);
        xi[1] += lengths_[i] * std::sin(q_[i]);
    }
    return xi;
}
void Robot2D::updateKinematics() {
    x = fkine();
    const double h = M_PI * 1e-3;
    Jac = Mat(4, dof);
    const Vec old_pos = fkine();
    for (int i = 0; i < dof; i++) {
        Vec qi = q;
        qi[i] += h;
        const Vec new_pos = fkine(qi);
        Jac(0, i) = (new_pos[0] - old_pos[0]) / h;
        Jac(1, i) = (new_pos[1] - old_pos[1]) / h;
    }
    dx.assign(2, 0.0);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < dof; j++) dx[i] += Jac(i, j) * dq[j];
}

// ---- TransformedSimulationInterface (TransformedSimulationInterface.cpp:20-103)
TransformedSimulationInterface::TransformedSimulationInterface(const std::shared_ptr<SimulationInterface>& r, const Mat& T) : r_(r), T_(T) {
    if (!r) throw std::runtime_error("[TransformedSimulationInterface] Object is not initialized");  // :33
    if (T.rows != 4 || T.cols != 4) throw std::runtime_error("[TransformedSimulationInterface] the frame must be a 4x4 pose matrix");
    nbCarDim = r->getNbCarDim();
    dof = r->getDOF();
    updateKinematics();
}
void TransformedSimulationInterface::updateKinematics() {  // :31-47
    r_->updateKinematics();
    q = r_->getJointsPos();
    dq = r_->getJointsVel();
    Jac = r_->J();
    dJac = r_->Jp();
    x = r_->getEEPosition();
    ornQuat = r_->getEEOrnQuat();
    dx = r_->getEEVelocity();
    w = r_->getEEAngVel();
    t = r_->getTime();
}
void TransformedSimulationInterface::setTime(double time) { t = time; r_->setTime(time); }
static Vec rot_t(const Mat& T, const Vec& v) {  // R^T v
    Vec o(3, 0.0);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o[i] += T(j, i) * v[j];
    return o;
}
static Mat rot_t_blocks(const Mat& T, const Mat& M) {  // blkdiag(R,R)^T M
    Mat o(6, M.cols);
    if (M.rows != 6) return o;
    for (int blk = 0; blk < 2; blk++)
        for (int i = 0; i < 3; i++)
            for (int c = 0; c < M.cols; c++) {
                double a = 0;
                for (int l = 0; l < 3; l++) a += T(l, i) * M(3 * blk + l, c);
                o(3 * blk + i, c) = a;
            }
    return o;
}
Mat TransformedSimulationInterface::J() { return rot_t_blocks(T_, Jac); }    // :53-58
Mat TransformedSimulationInterface::Jp() { return rot_t_blocks(T_, dJac); }  // :60-65
Vec TransformedSimulationInterface::getEEPosition() {  // :67-69
    return rot_t(T_, Vec{x[0] - T_(0, 3), x[1] - T_(1, 3), x[2] - T_(2, 3)});
}
Vec TransformedSimulationInterface::getEEVelocity() { return rot_t(T_, dx); }
Vec TransformedSimulationInterface::getEEAngVel() { return rot_t(T_, w); }
Vec TransformedSimulationInterface::getEEOrnQuat() {  // :94-103, Eigen's quaternion <-> matrix conversions
    const double qw = ornQuat[0], qx = ornQuat[1], qy = ornQuat[2], qz = ornQuat[3];
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
    double m[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double a = 0;
            for (int l = 0; l < 3; l++) a += T_(l, i) * R[l * 3 + j];
            m[i * 3 + j] = a;
        }
    double tq = m[0] + m[4] + m[8], c[3], qo;
    if (tq > 0) {
        tq = std::sqrt(tq + 1.0);
        qo = 0.5 * tq;
        tq = 0.5 / tq;
        c[0] = (m[7] - m[5]) * tq; c[1] = (m[2] - m[6]) * tq; c[2] = (m[3] - m[1]) * tq;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        tq = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        c[i] = 0.5 * tq;
        tq = 0.5 / tq;
        qo = (m[k * 3 + j] - m[j * 3 + k]) * tq;
        c[j] = (m[j * 3 + i] + m[i * 3 + j]) * tq;
        c[k] = (m[k * 3 + i] + m[i * 3 + k]) * tq;
    }
    return Vec{qo, c[0], c[1], c[2]};
}
void TransformedSimulationInterface::sendAcc(double dt, const Vec& ddq, bool updateKin) { r_->sendAcc(dt, ddq, updateKin); updateKinematics(); }
void TransformedSimulationInterface::sendVel(double dt, const Vec& dq_, bool updateKin) { r_->sendVel(dt, dq_, updateKin); updateKinematics(); }
void TransformedSimulationInterface::setConfiguration(const Vec& q_, const Vec& dq_, bool reset_time) { r_->setConfiguration(q_, dq_, reset_time); updateKinematics(); }
bool TransformedSimulationInterface::frame(double* R, double* p) const {
    for (int i = 0; i < 3; i++) {
        p[i] = T_(i, 3);
        for (int j = 0; j < 3; j++) R[i * 3 + j] = T_(i, j);
    }
    return true;
}
}  // namespace sim

// ------------------------------------------------------------------------------------------------ keypoints

namespace sys {
PosOrnKeypoint::PosOrnKeypoint(const Vec& position, const Vec& orientation, const Mat& precision, const int& timestep)
    : Keypoint(timestep, KpType::FIRST_ORDER, "POS_ORN"), position_(position), orientation_(orientation), precision_(precision), state_size_(7) {}
PosOrnKeypoint::PosOrnKeypoint(const Vec& position, const Vec& dposition, const Vec& orientation, const Vec& dorientation, const Mat& precision,
                               const int& timestep)
    : Keypoint(timestep, KpType::SECOND_ORDER, "POS_ORN"), position_(position), orientation_(orientation), dposition_(dposition),
      dorientation_(dorientation), precision_(precision), state_size_(14) {}

static void append(Vec& o, const Vec& a) { o.insert(o.end(), a.begin(), a.end()); }

Vec PosOrnKeypoint::getState() const {  // PosOrnKeypoint.cpp:12-22 (2nd order: [p, dp, quat, dquat] -- the reference's own layout)
    Vec s;
    append(s, position_);
    if (type_ == KpType::SECOND_ORDER) append(s, dposition_);
    append(s, orientation_);
    if (type_ == KpType::SECOND_ORDER) append(s, dorientation_);
    return s;
}
Vec PosOrnKeypoint::targetFx() const {
    Vec s;
    append(s, position_);
    append(s, orientation_);
    if (type_ == KpType::SECOND_ORDER) { append(s, dposition_); append(s, dorientation_); }
    return s;
}
Vec PosOrnKeypoint::diff(const Vec& state) const {  // PosOrnKeypoint.cpp:24-45
    const int rs = state_size_ - type_;
    Vec res(rs, 0.0);
    if (!is_zero(state)) {
        const Mat H = Sd::dQuatToDxJac(orientation_);
        for (int i = 0; i < 3; i++) res[i] = position_[i] - state[i];
        const Vec lm = Sd::logMap(orientation_, Vec(state.begin() + 3, state.begin() + 7));
        for (int i = 0; i < 3; i++) res[3 + i] = -2 * (H(i, 0) * lm[0] + H(i, 1) * lm[1] + H(i, 2) * lm[2] + H(i, 3) * lm[3]);
        if (type_ == KpType::SECOND_ORDER) {
            for (int i = 0; i < 3; i++) res[6 + i] = dposition_[i] - state[7 + i];
            const Vec tr = Sd::transport(Vec(state.begin() + 10, state.begin() + 14), Vec(state.begin() + 3, state.begin() + 7), orientation_);
            Vec dv(4);
            for (int i = 0; i < 4; i++) dv[i] = dorientation_[i] - tr[i];
            for (int i = 0; i < 3; i++) res[9 + i] = -2 * (H(i, 0) * dv[0] + H(i, 1) * dv[1] + H(i, 2) * dv[2] + H(i, 3) * dv[3]);
        }
    }
    return res;
}

static void check_thresh(const Vec& t) {
    if (t.size() != 3) throw std::runtime_error("[PosOrnKeypointDistFunct] orn_thresh must have 3 entries");
}
PosOrnKeypointDistFunct::PosOrnKeypointDistFunct(const Vec& position, const Vec& orientation, const Mat& precision, const double& pos_radius,
                                                 const Vec& orn_thresh, const int& timestep)
    : PosOrnKeypoint(position, orientation, precision, timestep), pos_radius_(pos_radius), orn_thresh_(orn_thresh) { check_thresh(orn_thresh); }
PosOrnKeypointDistFunct::PosOrnKeypointDistFunct(const Vec& position, const Vec& dposition, const Vec& orientation, const Vec& dorientation,
                                                 const Mat& precision, const double& pos_radius, const Vec& orn_thresh, const int& timestep)
    : PosOrnKeypoint(position, dposition, orientation, dorientation, precision, timestep), pos_radius_(pos_radius), orn_thresh_(orn_thresh) { check_thresh(orn_thresh); }
Vec PosOrnKeypointDistFunct::diff(const Vec& state) const {  // PosOrnKeypointDistFunct.cpp:13-35
    Vec r = PosOrnKeypoint::diff(state);
    const double n = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (n <= pos_radius_) {
        r[0] = r[1] = r[2] = 0;
    } else {
        const double f = n - pos_radius_;
        for (int i = 0; i < 3; i++) r[i] = r[i] / n * f;
    }
    for (int i = 0; i < 3; i++) {
        const double v = r[3 + i];
        if (std::fabs(v) <= orn_thresh_[i]) r[3 + i] = 0;
        else r[3 + i] = v - (v < 0 ? -1 : 1) * orn_thresh_[i];
    }
    return r;
}

SpacetimeKeypoint::SpacetimeKeypoint(const Vec& position, const Vec& orientation, const Mat& precision, const double& continuous_time, const int& timestep)
    : PosOrnKeypoint(position, orientation, precision, timestep), continuous_time_(continuous_time) { TAG_ = "POS_ORN_TIME"; }
SpacetimeKeypoint::SpacetimeKeypoint(const Vec& position, const Vec& dposition, const Vec& orientation, const Vec& dorientation, const Mat& precision,
                                     const double& continuous_time, const int& timestep)
    : PosOrnKeypoint(position, dposition, orientation, dorientation, precision, timestep), continuous_time_(continuous_time) { TAG_ = "POS_ORN_TIME"; }
Vec SpacetimeKeypoint::getState() const {  // SpacetimeKeypoint.cpp:12-17
    Vec s = PosOrnKeypoint::getState();
    s.push_back(continuous_time_);
    return s;
}
Vec SpacetimeKeypoint::targetFx() const {
    Vec s = PosOrnKeypoint::targetFx();
    s.push_back(continuous_time_);
    return s;
}
Vec SpacetimeKeypoint::diff(const Vec& state) const {  // :19-25
    Vec r = PosOrnKeypoint::diff(Vec(state.begin(), state.end() - 1));
    r.push_back(continuous_time_ - state.back());
    return r;
}

// ------------------------------------------------------------------------------------------------ systems

System::System(const std::shared_ptr<sim::SimulationInterface>& r_, const std::vector<std::shared_ptr<Keypoint>>& kps, const Vec& RtDiag, const Vec& qMax,
               const Vec& qMin, const Vec& dqMax, const Vec& dqMin, int horizon, int nb_deriv, const std::vector<std::string>& tags)
    : r(r_), keypoints(kps), Rdiag(RtDiag), horizon_(horizon), nb_deriv_(nb_deriv), EXPECTED_KP_TAGS_(tags) {  // System.cpp:28-61
    limits_set_ = true;
    penalty_ = 1;
    init();
    const int dof = r->getDOF();
    const int n = nb_deriv_ * dof;
    state_max_.assign(n, 0.0);
    state_min_.assign(n, 0.0);
    joint_limits_weight_.assign(n, 1);
    if ((int)qMax.size() != dof || (int)qMin.size() != dof) throw std::runtime_error("[System] qMax/qMin must have one entry per joint");
    if (nb_deriv_ == 1) {
        state_max_ = qMax;
        state_min_ = qMin;
    } else if (nb_deriv_ == 2) {
        Vec dM = dqMax.empty() ? Vec(dof, 0.0) : dqMax, dm = dqMin.empty() ? Vec(dof, 0.0) : dqMin;
        for (int i = 0; i < dof; i++) {
            state_max_[i] = qMax[i]; state_min_[i] = qMin[i];
            state_max_[dof + i] = dM[i]; state_min_[dof + i] = dm[i];
        }
        double diff2 = 0, n1 = 0, n2 = 0;  // Eigen isApprox (System.cpp:58-60)
        for (int i = 0; i < dof; i++) { diff2 += (dM[i] - dm[i]) * (dM[i] - dm[i]); n1 += dM[i] * dM[i]; n2 += dm[i] * dm[i]; }
        if (diff2 <= 1e-24 * std::min(n1, n2))
            for (int i = 0; i < dof; i++) joint_limits_weight_[dof + i] = 0;
    }
}

System::System(const std::shared_ptr<sim::SimulationInterface>& r_, const std::vector<std::shared_ptr<Keypoint>>& kps, const Vec& RtDiag, int horizon,
               int nb_deriv, const std::vector<std::string>& tags)
    : r(r_), keypoints(kps), Rdiag(RtDiag), horizon_(horizon), nb_deriv_(nb_deriv), EXPECTED_KP_TAGS_(tags) {  // System.cpp:63-75
    limits_set_ = false;
    penalty_ = 0;
    init();
}

void System::init() {  // System.cpp:77-86
    std::stable_sort(keypoints.begin(), keypoints.end(),
                     [](const std::shared_ptr<Keypoint>& a, const std::shared_ptr<Keypoint>& b) { return a->getTimestep() < b->getTimestep(); });
    if (!EXPECTED_KP_TAGS_.empty()) checkKeypoints();
}

void System::checkKeypoints() {  // System.cpp:363-372
    for (auto& kp : keypoints) {
        if (std::find(EXPECTED_KP_TAGS_.begin(), EXPECTED_KP_TAGS_.end(), kp->getTAG()) == EXPECTED_KP_TAGS_.end())
            throw std::runtime_error("[PosOrnPlannerSys] Wrong keypoint type: got " + kp->getTAG());
        if (kp->getType() != nb_deriv_)
            throw std::runtime_error("[PosOrnPlannerSys] Wrong keypoint order (nb_deriv_): Expecting " + std::to_string(nb_deriv_) + " got " +
                                     std::to_string(kp->getType()));
    }
}

std::vector<int> System::getKpIndexes() {
    std::vector<int> v;
    for (auto& kp : keypoints) v.push_back(kp->getTimestep());
    return v;
}

Mat System::getRt() {
    Mat R((int)Rdiag.size(), (int)Rdiag.size());
    for (size_t i = 0; i < Rdiag.size(); i++) R((int)i, (int)i) = Rdiag[i];
    return R;
}

Vec System::getMuVector(bool sparse) {  // System.cpp:321-339
    const int nt = nb_target_var_;
    if (sparse) {
        Vec mu(nt * keypoints.size(), 0.0);
        for (size_t i = 0; i < keypoints.size(); i++) {
            const Vec s = keypoints[i]->getState();
            std::copy(s.begin(), s.end(), mu.begin() + i * nt);
        }
        return mu;
    }
    Vec mu((size_t)horizon_ * nt, 0.0);
    for (auto& kp : keypoints) {
        const Vec s = kp->getState();
        std::copy(s.begin(), s.end(), mu.begin() + (size_t)kp->getTimestep() * nt);
    }
    return mu;
}

Mat System::getQMatrix(bool sparse) {  // System.cpp:341-361
    const int nq = nb_Q_var_;
    const int n = sparse ? (int)keypoints.size() * nq : horizon_ * nq;
    Mat Q(n, n);
    for (size_t i = 0; i < keypoints.size(); i++) {
        const Mat P = keypoints[i]->getPrecision();
        const int o = sparse ? (int)i * nq : keypoints[i]->getTimestep() * nq;
        for (int a = 0; a < nq; a++)
            for (int b = 0; b < nq; b++) Q(o + a, o + b) = P(a, b);
    }
    return Q;
}

// ---- single-point evaluation API (System.cpp:96-312).  Host glue: the kinematics behind getFxJac() are the simulator's
// (KDLRobot::updateKinematics = one-configuration call of the FK kernel); the solvers do not come through here.
std::shared_ptr<Keypoint> System::getKeypoint(int k) const {  // :96-101; the map of System::init keeps the last keypoint given for a step
    std::shared_ptr<Keypoint> hit;
    for (auto& kp : keypoints)
        if (kp->getTimestep() == k) hit = kp;
    return hit;
}

Vec System::diff(const Vec& actual_state, int k) {  // :103-109
    auto kp = getKeypoint(k);
    if (!kp) return Vec(nb_Q_var_, 0.0);
    return kp->diff(actual_state);
}

Vec System::diffBatch(const Vec& x) {  // :111-119: one f(x) block per keypoint, in keypoint order
    const int n = (int)keypoints.size();
    if ((int)x.size() != n * nb_target_var_) throw std::runtime_error("[System] diffBatch expects nb_keypoints * nb_target_var entries");
    Vec res((size_t)n * nb_Q_var_, 0.0);
    for (int i = 0; i < n; i++) {
        const Vec xt(x.begin() + (size_t)i * nb_target_var_, x.begin() + (size_t)(i + 1) * nb_target_var_);
        const Vec d = diff(xt, keypoints[i]->getTimestep());
        std::copy(d.begin(), d.end(), res.begin() + (size_t)i * nb_Q_var_);
    }
    return res;
}

std::pair<Mat, Vec> System::inspectJointLimit(const Vec& xk) {  // :121-142
    Mat L(nb_state_var_, nb_state_var_);
    Vec q(nb_state_var_, 0.0);
    if (limits_set_) {
        if ((int)xk.size() != nb_state_var_) throw std::runtime_error("[System] state has the wrong size");
        for (int i = 0; i < nb_state_var_; i++) {
            if (joint_limits_weight_[i] == 0) continue;
            if (xk[i] > state_max_[i]) { q[i] = state_max_[i] - xk[i]; L(i, i) = penalty_; }
            else if (xk[i] < state_min_[i]) { q[i] = state_min_[i] - xk[i]; L(i, i) = penalty_; }
        }
    }
    return std::make_pair(L, q);
}

System::StepLimitsOut System::forwardPassWithLimits(const Vec& xk, const Vec& uk, int k) {  // :144-161: the limits look at the state given, not the new one
    auto s = forwardPass(xk, uk, k);
    auto Lq = inspectJointLimit(xk);
    return std::make_tuple(std::get<0>(s), std::get<1>(s), Lq.second, Vec(nb_ctrl_var_, 0.0), std::get<2>(s), std::get<3>(s), std::get<4>(s), Lq.first);
}

std::tuple<Vec, Mat> System::getFxJac(const Vec& xk) {  // :163-179: evaluate at xk, then put the simulator back
    const int dof = r->getDOF();
    if ((int)xk.size() < nb_deriv_ * dof) throw std::runtime_error("[System] state has the wrong size");
    const Vec qk(xk.begin(), xk.begin() + dof);
    const Vec dqk = nb_deriv_ == 2 ? Vec(xk.begin() + dof, xk.begin() + 2 * dof) : Vec(dof, 0.0);
    const Vec old_q = r->getJointsPos(), old_dq = r->getJointsVel();
    r->setConfiguration(qk, dqk);
    auto out = getFxJac();
    r->setConfiguration(old_q, old_dq);
    return out;
}

// getFxJac(xk) of the two time systems (PosOrnTimePlannerSys.cpp:114-136, JointSpaceTimePlannerSys.cpp:88-110): the last state is the clock
static std::tuple<Vec, Mat> fx_jac_at_time_state(System& s, const Vec& xk) {
    auto r = s.robot();
    const int dof = r->getDOF();
    if ((int)xk.size() != s.getNbStateVar()) throw std::runtime_error("[System] state has the wrong size");
    const Vec qk(xk.begin(), xk.begin() + dof);
    const Vec dqk = s.getNbDeriv() == 2 ? Vec(xk.begin() + dof, xk.begin() + 2 * dof) : Vec(dof, 0.0);
    const Vec old_q = r->getJointsPos(), old_dq = r->getJointsVel();
    const double old_t = r->getTime();
    r->setConfiguration(qk, dqk);
    r->setTime(xk.back());
    auto out = s.getFxJac();
    r->setConfiguration(old_q, old_dq);
    r->setTime(old_t);
    return out;
}

std::tuple<Vec, Vec, std::vector<std::tuple<Mat, Mat, Mat, Mat>>> System::fpBatch(const Vec& u) {  // :181-211
    reset();
    if (u.size() % nb_ctrl_var_) throw std::runtime_error("[System] fpBatch expects (horizon - 1) * nb_ctrl_var controls");
    const int T = (int)u.size() / nb_ctrl_var_ + 1;
    Vec fX((size_t)T * nb_target_var_, 0.0), qL((size_t)T * nb_state_var_, 0.0);
    std::vector<std::tuple<Mat, Mat, Mat, Mat>> ABJL;
    auto fJ0 = getFxJac();
    std::copy(std::get<0>(fJ0).begin(), std::get<0>(fJ0).end(), fX.begin());
    ABJL.push_back(std::make_tuple(Mat::Identity(nb_state_var_), Mat(nb_state_var_, nb_ctrl_var_), std::get<1>(fJ0), Mat(nb_state_var_, nb_state_var_)));
    for (int i = 0; i < T - 1; i++) {
        const Vec ut(u.begin() + (size_t)i * nb_ctrl_var_, u.begin() + (size_t)(i + 1) * nb_ctrl_var_);
        auto o = forwardPassWithLimits(getState(), ut, i + 1);
        std::copy(std::get<1>(o).begin(), std::get<1>(o).end(), fX.begin() + (size_t)(i + 1) * nb_target_var_);
        std::copy(std::get<2>(o).begin(), std::get<2>(o).end(), qL.begin() + (size_t)(i + 1) * nb_state_var_);
        ABJL.push_back(std::make_tuple(std::get<4>(o), std::get<5>(o), std::get<6>(o), std::get<7>(o)));
    }
    return std::make_tuple(fX, qL, ABJL);
}

Vec System::cost(const Vec& xk, const Vec& uk, int k) {  // :213-234: the control term only counts at keypoint steps
    double c = 0;
    if (auto kp = getKeypoint(k)) {
        const Vec fx = std::get<0>(getFxJac(xk));
        const Vec e = kp->diff(fx);
        const Mat Q = kp->getPrecision();
        for (int i = 0; i < Q.rows; i++)
            for (int j = 0; j < Q.cols; j++) c += e[i] * Q(i, j) * e[j];
        for (size_t i = 0; i < uk.size() && i < Rdiag.size(); i++) c += uk[i] * Rdiag[i] * uk[i];
    }
    if (limits_set_) {
        auto Lq = inspectJointLimit(xk);
        for (int i = 0; i < nb_state_var_; i++) c += Lq.second[i] * Lq.first(i, i) * Lq.second[i];
    }
    return Vec(1, c);
}

Vec System::cost_x(const Vec& xk, const Vec&, int k) {  // :248-272
    Vec g(nb_state_var_, 0.0);
    if (auto kp = getKeypoint(k)) {
        auto fJ = getFxJac(xk);
        const Mat& J = std::get<1>(fJ);
        const Vec e = kp->diff(std::get<0>(fJ));
        const Mat Q = kp->getPrecision();
        Vec Qe(Q.rows, 0.0);
        for (int i = 0; i < Q.rows; i++)
            for (int j = 0; j < Q.cols; j++) Qe[i] += Q(i, j) * e[j];
        for (int a = 0; a < J.cols && a < nb_state_var_; a++)
            for (int i = 0; i < J.rows; i++) g[a] -= J(i, a) * Qe[i];
    }
    if (limits_set_) {
        auto Lq = inspectJointLimit(xk);
        for (int i = 0; i < nb_state_var_; i++) g[i] -= Lq.first(i, i) * Lq.second[i];
    }
    return g;
}

Mat System::cost_xx(const Vec& xk, const Vec&, int k) {  // :286-308
    Mat H(nb_state_var_, nb_state_var_);
    if (auto kp = getKeypoint(k)) {
        const Mat J = std::get<1>(getFxJac(xk));
        const Mat Q = kp->getPrecision();
        Mat QJ(Q.rows, J.cols);
        for (int i = 0; i < Q.rows; i++)
            for (int j = 0; j < Q.cols; j++)
                if (Q(i, j) != 0.0)
                    for (int a = 0; a < J.cols; a++) QJ(i, a) += Q(i, j) * J(j, a);
        for (int a = 0; a < J.cols && a < nb_state_var_; a++)
            for (int b = 0; b < J.cols && b < nb_state_var_; b++) {
                double v = 0;
                for (int i = 0; i < J.rows; i++) v += J(i, a) * QJ(i, b);
                H(a, b) += v;
            }
    }
    if (limits_set_) {
        auto Lq = inspectJointLimit(xk);
        for (int i = 0; i < nb_state_var_; i++) H(i, i) += Lq.first(i, i) * Lq.first(i, i);
    }
    return H;
}

bool System::builtin() const {
    if (!builtinType()) return false;
    // the simulator is the reference's main extension point (SimulationInterface.h:20): only the simulators of this mirror, exactly, are
    // lowered; a Cartesian system additionally needs a chain the device can hold.  Everything else runs over the virtuals.
    if (r) {
        if (!r->builtinSim()) return false;
        ilqr_problem_desc tmp;
        if (kind_ != ILQR_SYS_JOINT && kind_ != ILQR_SYS_JOINT_TIME && !r->lowerChain(&tmp)) return false;
    }
    for (auto& k : keypoints) {
        const std::type_info& t = typeid(*k);
        if (!(t == typeid(PosOrnKeypoint) || t == typeid(PosOrnKeypointDistFunct) || t == typeid(AngularKeypoint) || t == typeid(AngularTimeKeypoint) ||
              t == typeid(SpacetimeKeypoint)))
            return false;
    }
    return true;
}

void System::lower(ilqr_problem_desc* d) const {
    ilqr_desc_defaults(d);
    if (!r->lowerChain(d)) {
        if (kind_ != ILQR_SYS_JOINT && kind_ != ILQR_SYS_JOINT_TIME) throw std::runtime_error("[ilqr_hip] this SimulationInterface cannot be lowered to the device (only sim::KDLRobot chains can)");
        d->dof = r->getDOF();  // joint-space systems need no kinematic chain
        d->n_seg = 0;
    }
    d->kind = kind_;
    d->nb_deriv = nb_deriv_;
    d->horizon = horizon_;
    d->dt = dt_;
    if ((int)Rdiag.size() != nb_ctrl_var_) throw std::runtime_error("[System] RtDiag must have nb_ctrl_var entries");
    for (int i = 0; i < nb_ctrl_var_; i++) d->R_diag[i] = Rdiag[i];
    d->limits_set = limits_set_ ? 1 : 0;
    d->penalty = penalty_;
    for (size_t i = 0; i < state_max_.size(); i++) {
        d->state_max[i] = state_max_[i];
        d->state_min[i] = state_min_[i];
        d->limit_weight[i] = joint_limits_weight_[i];
    }
    if (keypoints.size() > ILQR_MAX_KP) throw std::runtime_error("[ilqr_hip] too many keypoints for the device descriptor");
    d->n_kp = (int)keypoints.size();
    for (size_t k = 0; k < keypoints.size(); k++) {
        if (k > 0 && keypoints[k]->getTimestep() == keypoints[k - 1]->getTimestep())
            throw std::runtime_error("[ilqr_hip] two keypoints share a timestep: not supported on the device");
        d->kp_timestep[k] = keypoints[k]->getTimestep();
        const Mat P = keypoints[k]->getPrecision();
        if (P.rows != nb_Q_var_ || P.cols != nb_Q_var_) throw std::runtime_error("[System] keypoint precision must be nb_Q_var x nb_Q_var");
        for (int a = 0; a < nb_Q_var_; a++)
            for (int b = 0; b < nb_Q_var_; b++) d->kp_Q[k][a * nb_Q_var_ + b] = P(a, b);
        if (auto* df = dynamic_cast<const PosOrnKeypointDistFunct*>(keypoints[k].get())) {  // dead zones (PosOrnKeypointDistFunct.cpp:13-35)
            d->kp_dist[k] = 1;
            d->kp_pos_radius[k] = df->getPosRadius();
            for (int i = 0; i < 3; i++) d->kp_orn_thresh[k][i] = df->getOrnThresh()[i];
        }
        if (r->frame(d->kp_frame_R[k], d->kp_frame_p[k])) d->kp_has_frame[k] = 1;  // TransformedSimulationInterface
    }
}

// ---- SequentialSystem (SequentialSystem.cpp:20-76)
SequentialSystem::SequentialSystem(const std::shared_ptr<sim::SimulationInterface>& r_, const std::vector<std::shared_ptr<System>>& systems, const Vec& RtDiag,
                                   int horizon, int nb_deriv)
    : System(r_, {}, RtDiag, horizon, nb_deriv, {}), systems_(systems) {
    if (systems_.empty()) throw std::runtime_error("[SequentialSystem] needs at least one system");
    auto& s0 = systems_[0];
    nb_target_var_ = 0;
    nb_Q_var_ = 0;
    for (auto& sy : systems_) {
        nb_target_var_ += sy->getNbTargetVar();
        nb_Q_var_ += sy->getNbQVar();
        if (s0->getNbStateVar() != sy->getNbStateVar()) throw std::runtime_error(" All the systems does not have the same number of state variable ");
        if (s0->getNbCtrlVar() != sy->getNbCtrlVar()) throw std::runtime_error(" All the systems does not have the same number of control variable ");
        if (s0->getHorizon() != sy->getHorizon()) throw std::runtime_error(" All the systems does not have the same horizon ");
        if (s0->getNbDeriv() != sy->getNbDeriv()) throw std::runtime_error(" All the systems does not have the same number of derivatives ");
        if (s0->getInitState() != sy->getInitState()) throw std::runtime_error(" All the systems does not have the same initState ");
    }
    nb_state_var_ = s0->getNbStateVar();
    nb_ctrl_var_ = s0->getNbCtrlVar();
    horizon_ = s0->getHorizon();
    nb_deriv_ = s0->getNbDeriv();
    kind_ = s0->kind();
    for (auto& sy : systems_)  // a hybrid sequence runs as its PosOrn(Time) sub-system's kind (see lower())
        if (sy->kind() != ILQR_SYS_JOINT && sy->kind() != ILQR_SYS_JOINT_TIME) { kind_ = sy->kind(); break; }
    x0_ = s0->getInitState();
    q0_ = r->getJointsPos();
    dq0_ = r->getJointsVel();
    f_x0_.clear();
    for (auto& sy : systems_) {
        const Vec f = sy->getInitFoXState();
        f_x0_.insert(f_x0_.end(), f.begin(), f.end());
        keypoints.insert(keypoints.end(), sy->getKeypoints().begin(), sy->getKeypoints().end());
    }
    init();
}
void SequentialSystem::reset() {
    for (auto& sy : systems_) sy->reset();
}
// ---- SequentialSystem evaluation (SequentialSystem.cpp:78-274)
System::StepOut SequentialSystem::forwardPass(const Vec& xk, const Vec& uk, int k) {  // :78-91
    auto o = systems_[0]->forwardPass(xk, uk, k);
    for (size_t i = 1; i < systems_.size(); i++) systems_[i]->robot()->updateKinematics();
    auto fJ = getFxJac();
    return std::make_tuple(std::get<0>(o), std::get<0>(fJ), std::get<2>(o), std::get<3>(o), std::get<1>(fJ));
}
std::tuple<Vec, Mat> SequentialSystem::getFxJac() {  // :93-113: f(x) and J of the sub-systems stacked
    Vec fx;
    Mat J(nb_Q_var_, nb_state_var_);
    int row = 0;
    for (auto& sy : systems_) {
        auto fJ = sy->getFxJac();
        const Mat& Jk = std::get<1>(fJ);
        fx.insert(fx.end(), std::get<0>(fJ).begin(), std::get<0>(fJ).end());
        for (int i = 0; i < Jk.rows; i++)
            for (int j = 0; j < Jk.cols; j++) J(row + i, j) = Jk(i, j);
        row += Jk.rows;
    }
    return std::make_tuple(fx, J);
}
Vec SequentialSystem::diff(const Vec& state, int k) {
    Vec out;
    size_t o = 0;
    for (auto& sy : systems_) {
        const Vec part(state.begin() + o, state.begin() + o + sy->getNbTargetVar());
        const Vec d = sy->diff(part, k);
        out.insert(out.end(), d.begin(), d.end());
        o += sy->getNbTargetVar();
    }
    return out;
}
static void add_to(Vec& a, const Vec& b) { for (size_t i = 0; i < a.size(); i++) a[i] += b[i]; }
static void add_to(Mat& a, const Mat& b) { for (size_t i = 0; i < a.d.size(); i++) a.d[i] += b.d[i]; }
Vec SequentialSystem::cost(const Vec& xk, const Vec& uk, int k) { Vec c(1, 0.0); for (auto& sy : systems_) add_to(c, sy->cost(xk, uk, k)); return c; }
Vec SequentialSystem::cost_x(const Vec& xk, const Vec& uk, int k) { Vec c(nb_state_var_, 0.0); for (auto& sy : systems_) add_to(c, sy->cost_x(xk, uk, k)); return c; }
Mat SequentialSystem::cost_xx(const Vec& xk, const Vec& uk, int k) { Mat c(nb_state_var_, nb_state_var_); for (auto& sy : systems_) add_to(c, sy->cost_xx(xk, uk, k)); return c; }
Vec SequentialSystem::cost_F(const Vec& xk) { Vec c(1, 0.0); for (auto& sy : systems_) add_to(c, sy->cost_F(xk)); return c; }
Vec SequentialSystem::cost_F_x(const Vec& xk) { Vec c(nb_state_var_, 0.0); for (auto& sy : systems_) add_to(c, sy->cost_F_x(xk)); return c; }
Mat SequentialSystem::cost_F_xx(const Vec& xk) { Mat c(nb_state_var_, nb_state_var_); for (auto& sy : systems_) add_to(c, sy->cost_F_xx(xk)); return c; }

Vec SequentialSystem::getMuVector(bool sparse) {
    const int nt = nb_target_var_;
    if (!sparse) {
        Vec mu((size_t)horizon_ * nt, 0.0);
        int idx = 0;
        for (auto& sy : systems_) {
            const Vec mk = sy->getMuVector(false);
            const int ntk = sy->getNbTargetVar();
            for (int j = 0; j < horizon_; j++) std::copy(mk.begin() + (size_t)j * ntk, mk.begin() + (size_t)(j + 1) * ntk, mu.begin() + (size_t)j * nt + idx);
            idx += ntk;
        }
        return mu;
    }
    Vec mu((size_t)nt * keypoints.size(), 0.0);
    for (size_t i = 0; i < keypoints.size(); i++) {
        int idx = 0;
        for (auto& sy : systems_) {
            if (auto kp = sy->getKeypoint(keypoints[i]->getTimestep())) {
                const Vec st = kp->getState();
                std::copy(st.begin(), st.end(), mu.begin() + i * nt + idx);
            }
            idx += sy->getNbTargetVar();
        }
    }
    return mu;
}
Mat SequentialSystem::getQMatrix(bool sparse) {
    const int nq = nb_Q_var_;
    if (!sparse) {
        Mat Q(horizon_ * nq, horizon_ * nq);
        int idx = 0;
        for (auto& sy : systems_) {
            const Mat Qk = sy->getQMatrix(false);
            const int nk = sy->getNbQVar();
            for (int j = 0; j < horizon_; j++)
                for (int a = 0; a < nk; a++)
                    for (int b = 0; b < nk; b++) Q(j * nq + idx + a, j * nq + idx + b) = Qk(j * nk + a, j * nk + b);
            idx += nk;
        }
        return Q;
    }
    const int n = (int)keypoints.size();
    Mat Q(n * nq, n * nq);
    for (int i = 0; i < n; i++) {
        int idx = 0;
        for (auto& sy : systems_) {
            const int nk = sy->getNbQVar();
            if (auto kp = sy->getKeypoint(keypoints[i]->getTimestep())) {
                const Mat P = kp->getPrecision();
                for (int a = 0; a < nk; a++)
                    for (int b = 0; b < nk; b++) Q(i * nq + idx + a, i * nq + idx + b) = P(a, b);
            }
            idx += nk;
        }
    }
    return Q;
}
// Device form: dynamics, chain and limits of the first sub-system; every keypoint keeps the frame and the control penalty of
// its own sub-system; the limit terms count once per sub-system (SequentialSystem.cpp:144-168 sums cost, cost_x, cost_xx).
bool SequentialSystem::builtinType() const {
    if (typeid(*this) != typeid(SequentialSystem)) return false;
    for (auto& sy : systems_)
        if (!sy->builtin()) return false;
    return true;
}

void SequentialSystem::lower(ilqr_problem_desc* d) const {
    std::vector<ilqr_problem_desc> subs(systems_.size());
    for (size_t i = 0; i < systems_.size(); i++) systems_[i]->lower(&subs[i]);
    // Hybrid sequences (HYBRID_SYS*.ipynb): JointSpace(Time)PlannerSys sub-systems next to PosOrn(Time)PlannerSys ones.  The dynamics
    // are the same (JointSpacePlannerSys.cpp:93-116 = PosOrnPlannerSys.cpp:114-138); the device problem is the PosOrn one and the
    // keypoints of the joint-space sub-systems are flagged kp_joint (residual target - x, J = I).
    auto is_joint = [](int k) { return k == ILQR_SYS_JOINT || k == ILQR_SYS_JOINT_TIME; };
    auto is_time = [](int k) { return k == ILQR_SYS_POS_ORN_TIME || k == ILQR_SYS_JOINT_TIME; };
    size_t base = 0;
    for (size_t i = 0; i < subs.size(); i++)
        if (!is_joint(subs[i].kind)) { base = i; break; }
    *d = subs[base];
    const bool hybrid = !is_joint(d->kind);
    if ((int)Rdiag.size() != nb_ctrl_var_) throw std::runtime_error("[System] RtDiag must have nb_ctrl_var entries");
    for (int i = 0; i < nb_ctrl_var_; i++) d->R_diag[i] = Rdiag[i];  // l_u = R u, l_uu = R use the sequential system's own Rt
    // limit terms: every sub-system adds its own (SequentialSystem.cpp:143-165).  Sub-systems with the same bounds form a group that
    // counts with its multiplicity; a second group with other bounds goes to the descriptor's second limit set (generic kernels).
    int n_lim = 0, n_lim2 = 0;
    bool have2 = false;
    auto same_limits = [&](const ilqr_problem_desc& a, const double* smax, const double* smin, const int* lw, double pen) {
        for (int j = 0; j < nb_state_var_; j++)
            if (a.state_max[j] != smax[j] || a.state_min[j] != smin[j] || a.limit_weight[j] != lw[j]) return false;
        return a.penalty == pen;
    };
    bool first_set = false;
    for (size_t i = 0; i < subs.size(); i++) {
        const auto& a = subs[i];
        const bool mixed = a.kind != d->kind;
        if (mixed && !(hybrid && is_joint(a.kind) && is_time(a.kind) == is_time(d->kind) && d->nb_deriv == 1 && a.dof == d->dof))
            throw std::runtime_error("[ilqr_hip] these sub-system kinds cannot be lowered together (joint-space next to PosOrn needs nbDeriv = 1 and 7 joints)");
        if (a.dt != d->dt) throw std::runtime_error("[ilqr_hip] sub-systems of different kinds / dt cannot be lowered");
        if (!a.limits_set) continue;
        if (!first_set) {  // the first limited sub-system defines the first set (the base descriptor may be another sub-system's)
            first_set = true;
            d->limits_set = 1;
            d->penalty = a.penalty;
            for (int j = 0; j < nb_state_var_; j++) { d->state_max[j] = a.state_max[j]; d->state_min[j] = a.state_min[j]; d->limit_weight[j] = a.limit_weight[j]; }
            n_lim = 1;
        } else if (same_limits(a, d->state_max, d->state_min, d->limit_weight, d->penalty)) {
            n_lim++;
        } else if (!have2) {
            have2 = true;
            d->limits2_set = 1;
            d->penalty2 = a.penalty;
            for (int j = 0; j < nb_state_var_; j++) { d->state_max2[j] = a.state_max[j]; d->state_min2[j] = a.state_min[j]; d->limit_weight2[j] = a.limit_weight[j]; }
            n_lim2 = 1;
        } else if (same_limits(a, d->state_max2, d->state_min2, d->limit_weight2, d->penalty2)) {
            n_lim2++;
        } else {
            throw std::runtime_error("[ilqr_hip] sub-systems with more than two different sets of limits cannot be lowered");
        }
    }
    if (!first_set) d->limits_set = 0;
    d->limit_multiplicity = n_lim > 1 ? n_lim : 1;
    d->limit_multiplicity2 = n_lim2 > 1 ? n_lim2 : 1;
    d->is_sequence = 1;
    // merge the keypoints by timestep
    struct Src { int sys, k; };
    std::vector<Src> order;
    for (size_t i = 0; i < subs.size(); i++)
        for (int k = 0; k < subs[i].n_kp; k++) order.push_back({(int)i, k});
    std::stable_sort(order.begin(), order.end(), [&](const Src& a, const Src& b) { return subs[a.sys].kp_timestep[a.k] < subs[b.sys].kp_timestep[b.k]; });
    if (order.size() > ILQR_MAX_KP) throw std::runtime_error("[ilqr_hip] too many keypoints for the device descriptor");
    d->n_kp = (int)order.size();
    for (size_t o = 0; o < order.size(); o++) {
        const auto& a = subs[order[o].sys];
        const int k = order[o].k;
        if (o > 0 && a.kp_timestep[k] == d->kp_timestep[o - 1])
            throw std::runtime_error("[ilqr_hip] two keypoints share a timestep: not supported on the device");
        d->kp_timestep[o] = a.kp_timestep[k];
        d->kp_joint[o] = (hybrid && is_joint(a.kind)) ? 1 : 0;  // its kp_Q is n_x x n_x with leading dimension n_x already
        std::memcpy(d->kp_Q[o], a.kp_Q[k], sizeof(a.kp_Q[k]));
        d->kp_dist[o] = a.kp_dist[k];
        d->kp_pos_radius[o] = a.kp_pos_radius[k];
        std::memcpy(d->kp_orn_thresh[o], a.kp_orn_thresh[k], sizeof(a.kp_orn_thresh[k]));
        d->kp_has_frame[o] = a.kp_has_frame[k];
        std::memcpy(d->kp_frame_R[o], a.kp_frame_R[k], sizeof(a.kp_frame_R[k]));
        std::memcpy(d->kp_frame_p[o], a.kp_frame_p[k], sizeof(a.kp_frame_p[k]));
        d->kp_has_Ru[o] = 1;
        for (int i = 0; i < nb_ctrl_var_; i++) d->kp_Ru[o][i] = a.R_diag[i];
    }
}

PosOrnPlannerSys::PosOrnPlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& kps, const Vec& Rt,
                                   const Vec& qMax, const Vec& qMin, const Vec& dqMax, const Vec& dqMin, int horizon, int nb_deriv, double dt)
    : System(r, kps, Rt, qMax, qMin, dqMax, dqMin, horizon, nb_deriv, {"POS_ORN"}) { localInit(dt); }
PosOrnPlannerSys::PosOrnPlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& kps, const Vec& Rt,
                                   const Vec& qMax, const Vec& qMin, int horizon, int nb_deriv, double dt)
    : System(r, kps, Rt, qMax, qMin, Vec(), Vec(), horizon, nb_deriv, {"POS_ORN"}) { localInit(dt); }
PosOrnPlannerSys::PosOrnPlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& kps, const Vec& Rt,
                                   int horizon, int nb_deriv, double dt)
    : System(r, kps, Rt, horizon, nb_deriv, {"POS_ORN"}) { localInit(dt); }

void PosOrnPlannerSys::localInit(double dt) {  // PosOrnPlannerSys.cpp:54-78
    kind_ = ILQR_SYS_POS_ORN;
    dt_ = dt;
    q0_ = r->getJointsPos();
    dq0_ = r->getJointsVel();
    f_x0_.clear();
    x0_ = q0_;
    append(f_x0_, r->getEEPosition());
    append(f_x0_, r->getEEOrnQuat());
    if (nb_deriv_ != 1) {
        append(f_x0_, r->getEEVelocity());
        append(f_x0_, r->getEEAngVelQuat());
        append(x0_, dq0_);
    }
    nb_state_var_ = (int)x0_.size();
    nb_ctrl_var_ = r->getDOF();
    nb_target_var_ = (int)f_x0_.size();
    nb_Q_var_ = nb_target_var_ - nb_deriv_;
}
Vec PosOrnPlannerSys::getState() {
    Vec xk = r->getJointsPos();
    if (nb_deriv_ != 1) append(xk, r->getJointsVel());
    return xk;
}
void PosOrnPlannerSys::reset() { r->setConfiguration(q0_, dq0_); }

static void set_block(Mat& M, int r0, int c0, const Mat& S) {
    for (int i = 0; i < S.rows; i++)
        for (int j = 0; j < S.cols; j++) M(r0 + i, c0 + j) = S(i, j);
}
static void append_v(Vec& o, const Vec& a) { o.insert(o.end(), a.begin(), a.end()); }

std::tuple<Vec, Mat> PosOrnPlannerSys::getFxJac() {  // PosOrnPlannerSys.cpp:80-102
    Vec fx = r->getEEPosition();
    append_v(fx, r->getEEOrnQuat());
    const Mat Jk = r->J();
    if (nb_deriv_ == 1) return std::make_tuple(fx, Jk);
    append_v(fx, r->getEEVelocity());
    append_v(fx, r->getEEAngVelQuat());
    Mat J(2 * Jk.rows, 2 * Jk.cols);
    set_block(J, 0, 0, Jk);
    set_block(J, Jk.rows, Jk.cols, Jk);
    return std::make_tuple(fx, J);
}
// constant-dt dynamics shared by PosOrnPlannerSys and JointSpacePlannerSys (PosOrnPlannerSys.cpp:114-138, JointSpacePlannerSys.cpp:93-116)
static void step_fixed_dt(sim::SimulationInterface& r, int nb_deriv, double dt, const Vec& uk, Mat& A, Mat& B) {
    const int dof = r.getDOF();
    if ((int)uk.size() != dof) throw std::runtime_error("[System] control has the wrong size");
    const int nx = nb_deriv * dof;
    A = Mat::Identity(nx);
    B = Mat(nx, dof);
    if (nb_deriv == 1) {
        for (int i = 0; i < dof; i++) B(i, i) = dt;
        r.sendVel(dt, uk);
    } else {
        for (int i = 0; i < dof; i++) { A(i, dof + i) = dt; B(i, i) = dt * dt / 2; B(dof + i, i) = dt; }
        r.sendAcc(dt, uk);
    }
}
// dt = u_last^2 dynamics shared by the two time systems (PosOrnTimePlannerSys.cpp:149-185, JointSpaceTimePlannerSys.cpp:123-155);
// the 2nd-order time column uses the velocity AFTER the step, as upstream does
static void step_time(sim::SimulationInterface& r, int nb_deriv, const Vec& uk, Mat& A, Mat& B) {
    const int dof = r.getDOF();
    if ((int)uk.size() != dof + 1) throw std::runtime_error("[System] control has the wrong size");
    const int nx = nb_deriv * dof + 1, nu = dof + 1;
    const double ds = uk[nu - 1], dt = ds * ds;
    const Vec v(uk.begin(), uk.begin() + dof);
    A = Mat::Identity(nx);
    B = Mat(nx, nu);
    if (nb_deriv == 1) {
        r.sendVel(dt, v);
        for (int i = 0; i < dof; i++) { B(i, i) = dt; B(i, nu - 1) = 2 * ds * v[i]; }
        B(nx - 1, nu - 1) = 2 * ds;
    } else {
        r.sendAcc(dt, v);
        const Vec dq = r.getJointsVel();
        for (int i = 0; i < dof; i++) {
            A(i, dof + i) = dt;
            B(i, i) = dt * dt / 2;
            B(dof + i, i) = dt;
            B(i, nu - 1) = 2 * ds * dq[i] + 2 * ds * ds * ds * v[i];
            B(dof + i, nu - 1) = 2 * ds * v[i];
        }
        B(nx - 1, nu - 1) = 2 * ds;
    }
}
System::StepOut PosOrnPlannerSys::forwardPass(const Vec&, const Vec& uk, int) {
    Mat A, B;
    step_fixed_dt(*r, nb_deriv_, dt_, uk, A, B);
    auto fJ = getFxJac();
    return std::make_tuple(getState(), std::get<0>(fJ), A, B, std::get<1>(fJ));
}

// A joint-space descriptor of n < 7 joints widened to the 7 the device kernels are built for: the extra joints get zero precision,
// zero limit weight and the first joint's control penalty; with u = 0 they stay at rest, so the n-joint problem is unchanged.
static void pad_joint_desc(ilqr_problem_desc* d, int n, int tm) {
    if (n >= 7) return;
    const int D = 7, nu_old = n + tm, nx_old = n + tm, nn = D + tm;
    double R[ILQR_MAX_NU];
    for (int i = 0; i < D; i++) R[i] = d->R_diag[i < n ? i : 0];
    if (tm) R[D] = d->R_diag[nu_old - 1];
    for (int i = 0; i < nn; i++) d->R_diag[i] = R[i];
    double smax[ILQR_MAX_NX + 1] = {0}, smin[ILQR_MAX_NX + 1] = {0};
    int lw[ILQR_MAX_NX + 1] = {0};
    for (int i = 0; i < n; i++) { smax[i] = d->state_max[i]; smin[i] = d->state_min[i]; lw[i] = d->limit_weight[i]; }
    for (int i = 0; i < nn; i++) { d->state_max[i] = smax[i]; d->state_min[i] = smin[i]; d->limit_weight[i] = lw[i]; }
    for (int k = 0; k < d->n_kp; k++) {
        std::vector<double> Q((size_t)nn * nn, 0.0);
        auto map = [&](int i) { return i < n ? i : D; };  // user index -> device index (the time entry moves to the end)
        for (int a = 0; a < nx_old; a++)
            for (int b = 0; b < nx_old; b++) Q[(size_t)map(a) * nn + map(b)] = d->kp_Q[k][a * nx_old + b];
        for (int i = 0; i < nn * nn; i++) d->kp_Q[k][i] = Q[i];
    }
    d->dof = D;
    d->n_seg = 0;
}

// ---- AngularKeypoint (AngularKeypoint.cpp:15-27), JointSpacePlannerSys (JointSpacePlannerSys.cpp:50-122)
Vec AngularKeypoint::getState() const {
    Vec st = position_;
    if (type_ == SECOND_ORDER) st.insert(st.end(), dposition_.begin(), dposition_.end());
    return st;
}
Vec AngularKeypoint::diff(const Vec& state) const {
    const Vec tg = AngularKeypoint::getState();  // qualified as in AngularKeypoint.cpp:25 (AngularTimeKeypoint appends the time itself)
    if (state.size() != tg.size()) throw std::runtime_error("[AngularKeypoint] state size mismatch");
    Vec r(tg.size());
    for (size_t i = 0; i < tg.size(); i++) r[i] = tg[i] - state[i];
    return r;
}
JointSpacePlannerSys::JointSpacePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& kps, const Vec& Rt,
                                           const Vec& qMax, const Vec& qMin, const Vec& dqMax, const Vec& dqMin, int horizon, int nb_deriv, double dt)
    : System(r, kps, Rt, qMax, qMin, dqMax, dqMin, horizon, nb_deriv, {"JNT"}) { localInit(dt); }
JointSpacePlannerSys::JointSpacePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& kps, const Vec& Rt,
                                           const Vec& qMax, const Vec& qMin, int horizon, int nb_deriv, double dt)
    : System(r, kps, Rt, qMax, qMin, Vec(), Vec(), horizon, nb_deriv, {"JNT"}) { localInit(dt); }
JointSpacePlannerSys::JointSpacePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& kps, const Vec& Rt,
                                           int horizon, int nb_deriv, double dt)
    : System(r, kps, Rt, horizon, nb_deriv, {"JNT"}) { localInit(dt); }
void JointSpacePlannerSys::localInit(double dt) {  // :54-75
    kind_ = ILQR_SYS_JOINT;
    dt_ = dt;
    q0_ = r->getJointsPos();
    dq0_ = r->getJointsVel();
    x0_ = q0_;
    if (nb_deriv_ != 1) append(x0_, dq0_);
    f_x0_ = x0_;
    nb_state_var_ = (int)x0_.size();
    nb_ctrl_var_ = r->getDOF();
    nb_target_var_ = (int)f_x0_.size();
    nb_Q_var_ = nb_target_var_;
}
Vec JointSpacePlannerSys::getState() {
    Vec xk = r->getJointsPos();
    if (nb_deriv_ != 1) append(xk, r->getJointsVel());
    return xk;
}
void JointSpacePlannerSys::reset() { r->setConfiguration(q0_, dq0_); }

std::tuple<Vec, Mat> JointSpacePlannerSys::getFxJac() {  // JointSpacePlannerSys.cpp:77-81 (nb_Q_var x nb_ctrl_var identity, as upstream)
    Mat J(nb_Q_var_, nb_ctrl_var_);
    for (int i = 0; i < std::min(nb_Q_var_, nb_ctrl_var_); i++) J(i, i) = 1;
    return std::make_tuple(getState(), J);
}
System::StepOut JointSpacePlannerSys::forwardPass(const Vec&, const Vec& uk, int) {
    Mat A, B;
    step_fixed_dt(*r, nb_deriv_, dt_, uk, A, B);
    auto fJ = getFxJac();
    return std::make_tuple(getState(), std::get<0>(fJ), A, B, std::get<1>(fJ));
}
void JointSpacePlannerSys::lower(ilqr_problem_desc* d) const {
    if (nb_deriv_ != 1) throw std::runtime_error("[ilqr_hip] JointSpacePlannerSys is lowered for nbDeriv = 1 only (the 2nd-order variant is inconsistent upstream)");
    if (r->getDOF() > 7) throw std::runtime_error("[ilqr_hip] joint-space systems are lowered for at most 7 joints");
    System::lower(d);  // f(x) = x needs no chain, but a KDLRobot's chain is kept for ilqr_fk_batch users
    pad_joint_desc(d, r->getDOF(), 0);
}

// ---- AngularTimeKeypoint (AngularTimeKeypoint.cpp:15-30), JointSpaceTimePlannerSys (JointSpaceTimePlannerSys.cpp:50-160)
Vec AngularTimeKeypoint::getState() const {
    Vec st = AngularKeypoint::getState();
    st.push_back(continuous_time_);
    return st;
}
Vec AngularTimeKeypoint::diff(const Vec& state) const {
    if (state.empty()) throw std::runtime_error("[AngularTimeKeypoint] empty state");
    Vec r = AngularKeypoint::diff(Vec(state.begin(), state.end() - 1));
    r.push_back(continuous_time_ - state.back());
    return r;
}
JointSpaceTimePlannerSys::JointSpaceTimePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& kps,
                                                   const Vec& Rt, const Vec& qMax, const Vec& qMin, const Vec& dqMax, const Vec& dqMin, int horizon, int nb_deriv)
    : System(r, kps, Rt, qMax, qMin, dqMax, dqMin, horizon, nb_deriv, {"JNT_TIME"}) { localInit(); }
JointSpaceTimePlannerSys::JointSpaceTimePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& kps,
                                                   const Vec& Rt, const Vec& qMax, const Vec& qMin, int horizon, int nb_deriv)
    : System(r, kps, Rt, qMax, qMin, Vec(), Vec(), horizon, nb_deriv, {"JNT_TIME"}) { localInit(); }
JointSpaceTimePlannerSys::JointSpaceTimePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& kps,
                                                   const Vec& Rt, int horizon, int nb_deriv)
    : System(r, kps, Rt, horizon, nb_deriv, {"JNT_TIME"}) { localInit(); }
void JointSpaceTimePlannerSys::localInit() {  // :50-78
    kind_ = ILQR_SYS_JOINT_TIME;
    dt_ = 0;
    q0_ = r->getJointsPos();
    dq0_ = r->getJointsVel();
    x0_ = q0_;
    if (nb_deriv_ != 1) append(x0_, dq0_);
    x0_.push_back(0);
    f_x0_ = x0_;
    nb_state_var_ = (int)x0_.size();
    nb_ctrl_var_ = r->getDOF() + 1;
    nb_target_var_ = (int)f_x0_.size();
    nb_Q_var_ = nb_target_var_;
    state_max_.push_back(0);  // the time state is never limited (weight 0)
    state_min_.push_back(0);
    joint_limits_weight_.push_back(0);
}
Vec JointSpaceTimePlannerSys::getState() {
    Vec xk = r->getJointsPos();
    if (nb_deriv_ != 1) append(xk, r->getJointsVel());
    xk.push_back(r->getTime());
    return xk;
}
void JointSpaceTimePlannerSys::reset() { r->setConfiguration(q0_, dq0_); }

std::tuple<Vec, Mat> JointSpaceTimePlannerSys::getFxJac() {  // JointSpaceTimePlannerSys.cpp:82-86
    Mat J(nb_target_var_, nb_ctrl_var_);
    for (int i = 0; i < std::min(nb_target_var_, nb_ctrl_var_); i++) J(i, i) = 1;
    return std::make_tuple(getState(), J);
}
std::tuple<Vec, Mat> JointSpaceTimePlannerSys::getFxJac(const Vec& xk) { return fx_jac_at_time_state(*this, xk); }
System::StepOut JointSpaceTimePlannerSys::forwardPass(const Vec&, const Vec& uk, int) {
    Mat A, B;
    step_time(*r, nb_deriv_, uk, A, B);
    auto fJ = getFxJac();
    return std::make_tuple(getState(), std::get<0>(fJ), A, B, std::get<1>(fJ));
}
void JointSpaceTimePlannerSys::lower(ilqr_problem_desc* d) const {
    if (nb_deriv_ != 1) throw std::runtime_error("[ilqr_hip] JointSpaceTimePlannerSys is lowered for nbDeriv = 1 only");
    if (r->getDOF() > 7) throw std::runtime_error("[ilqr_hip] joint-space systems are lowered for at most 7 joints");
    System::lower(d);
    pad_joint_desc(d, r->getDOF(), 1);
}

PosOrnTimePlannerSys::PosOrnTimePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& kps, const Vec& Rt,
                                           const Vec& qMax, const Vec& qMin, const Vec& dqMax, const Vec& dqMin, int horizon, int nb_deriv)
    : System(r, kps, Rt, qMax, qMin, dqMax, dqMin, horizon, nb_deriv, {"POS_ORN_TIME"}) { localInit(); }
PosOrnTimePlannerSys::PosOrnTimePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& kps, const Vec& Rt,
                                           const Vec& qMax, const Vec& qMin, int horizon, int nb_deriv)
    : System(r, kps, Rt, qMax, qMin, Vec(), Vec(), horizon, nb_deriv, {"POS_ORN_TIME"}) { localInit(); }
PosOrnTimePlannerSys::PosOrnTimePlannerSys(const std::shared_ptr<sim::SimulationInterface>& r, const std::vector<std::shared_ptr<Keypoint>>& kps, const Vec& Rt,
                                           int horizon, int nb_deriv)
    : System(r, kps, Rt, horizon, nb_deriv, {"POS_ORN_TIME"}) { localInit(); }

void PosOrnTimePlannerSys::localInit() {  // PosOrnTimePlannerSys.cpp:50-83
    kind_ = ILQR_SYS_POS_ORN_TIME;
    dt_ = 0;
    q0_ = r->getJointsPos();
    dq0_ = r->getJointsVel();
    f_x0_.clear();
    x0_ = q0_;
    append(f_x0_, r->getEEPosition());
    append(f_x0_, r->getEEOrnQuat());
    if (nb_deriv_ != 1) {
        append(f_x0_, r->getEEVelocity());
        append(f_x0_, r->getEEAngVelQuat());
        append(x0_, dq0_);
    }
    f_x0_.push_back(0);
    x0_.push_back(0);
    nb_state_var_ = (int)x0_.size();
    nb_ctrl_var_ = r->getDOF() + 1;
    nb_target_var_ = (int)f_x0_.size();
    nb_Q_var_ = nb_target_var_ - nb_deriv_;
    state_max_.push_back(0);  // the time state is never limited (weight 0)
    state_min_.push_back(0);
    joint_limits_weight_.push_back(0);
}
Vec PosOrnTimePlannerSys::getState() {
    Vec xk = r->getJointsPos();
    if (nb_deriv_ != 1) append(xk, r->getJointsVel());
    xk.push_back(r->getTime());
    return xk;
}
void PosOrnTimePlannerSys::reset() { r->setConfiguration(q0_, dq0_); }

std::tuple<Vec, Mat> PosOrnTimePlannerSys::getFxJac() {  // PosOrnTimePlannerSys.cpp:85-112
    Vec fx = r->getEEPosition();
    append_v(fx, r->getEEOrnQuat());
    const Mat J = r->J();
    if (nb_deriv_ == 1) {
        fx.push_back(r->getTime());
        Mat Jk(J.rows + 1, J.cols + 1);
        set_block(Jk, 0, 0, J);
        Jk(J.rows, J.cols) = 1;
        return std::make_tuple(fx, Jk);
    }
    append_v(fx, r->getEEVelocity());
    append_v(fx, r->getEEAngVelQuat());
    fx.push_back(r->getTime());
    Mat Js(2 * J.rows + 1, 2 * J.cols + 1);
    set_block(Js, 0, 0, J);
    set_block(Js, J.rows, J.cols, J);
    Js(2 * J.rows, 2 * J.cols) = 1;
    return std::make_tuple(fx, Js);
}
std::tuple<Vec, Mat> PosOrnTimePlannerSys::getFxJac(const Vec& xk) { return fx_jac_at_time_state(*this, xk); }
System::StepOut PosOrnTimePlannerSys::forwardPass(const Vec&, const Vec& uk, int) {
    Mat A, B;
    step_time(*r, nb_deriv_, uk, A, B);
    auto fJ = getFxJac();
    return std::make_tuple(getState(), std::get<0>(fJ), A, B, std::get<1>(fJ));
}
}  // namespace sys

// ------------------------------------------------------------------------------------------------ solvers

namespace solver {

struct ProblemGuard {
    ilqr_problem* p = nullptr;
    ~ProblemGuard() { if (p) ilqr_problem_destroy(p); }
};

static std::string fmt(double v) {
    std::ostringstream o;
    o << v;
    return o.str();
}

// Lower `s`, upload the per-instance inputs, run `solve`, read everything back.
static BatchResult run_batch(sys::System& s, const BatchInputs& in, int nb_iter, bool gains, const ilqr_problem_desc* override_desc,
                             const std::function<void(ilqr_problem*)>& pre_solve, const std::function<void(ilqr_problem*)>& solve,
                             const std::function<void(ilqr_problem*)>& post_solve) {
    if (!s.builtin())
        throw std::runtime_error("[ilqr_hip] a user-defined System / Keypoint subclass has no device lowering: only ILQRRecursive::solve runs it (over its virtuals)");
    ilqr_problem_desc d;
    if (override_desc) d = *override_desc;
    else s.lower(&d);
    ilqr_dims dm;
    if (ilqr_dims_of(&d, &dm)) throw std::runtime_error("[ilqr_hip] unsupported system for the device");
    ilqr_ctx* ctx = device_context();
    const int B = in.B, T = d.horizon, dof = d.dof;
    if (B <= 0) throw std::runtime_error("[ilqr_hip] batch must be positive");
    static thread_local bool widened = false;  // set while the widened problem of a padded joint-space system runs
    const int padn = widened ? 0 : s.paddedFromDof();  // joint-space system of padn < 7 joints: inputs widened, outputs narrowed again
    if (padn > 0) {
        const int tm = (d.kind == ILQR_SYS_JOINT_TIME) ? 1 : 0, D = 7, wo = padn + tm, wn = D + tm;
        auto widen = [&](const std::vector<double>& v, size_t rows) {  // [rows][wo] -> [rows][wn], the time entry moves to the end
            std::vector<double> o(rows * wn, 0.0);
            for (size_t r_ = 0; r_ < rows; r_++) {
                for (int i = 0; i < padn; i++) o[r_ * wn + i] = v[r_ * wo + i];
                if (tm) o[r_ * wn + D] = v[r_ * wo + padn];
            }
            return o;
        };
        auto narrow = [&](const std::vector<double>& v, size_t rows) {
            std::vector<double> o(rows * wo, 0.0);
            for (size_t r_ = 0; r_ < rows; r_++) {
                for (int i = 0; i < padn; i++) o[r_ * wo + i] = v[r_ * wn + i];
                if (tm) o[r_ * wo + padn] = v[r_ * wn + D];
            }
            return o;
        };
        BatchInputs in2 = in;
        auto pad_q = [&](const std::vector<double>& v, const Vec& dflt) {
            std::vector<double> src = v;
            if (src.empty()) for (int b_ = 0; b_ < B; b_++) src.insert(src.end(), dflt.begin(), dflt.end());
            std::vector<double> o((size_t)B * D, 0.0);
            for (int b_ = 0; b_ < B; b_++) for (int i = 0; i < padn; i++) o[(size_t)b_ * D + i] = src[(size_t)b_ * padn + i];
            return o;
        };
        in2.q0 = pad_q(in.q0, s.q0());
        in2.dq0 = pad_q(in.dq0, s.dq0());
        const auto& kps_ = s.getKeypoints();
        in2.kp_targets.assign(kps_.size(), {});
        for (size_t k = 0; k < kps_.size(); k++) {
            std::vector<double> tg = (k < in.kp_targets.size() && !in.kp_targets[k].empty()) ? in.kp_targets[k] : std::vector<double>();
            if (tg.empty()) for (int b_ = 0; b_ < B; b_++) { const Vec t_ = kps_[k]->targetFx(); tg.insert(tg.end(), t_.begin(), t_.end()); }
            in2.kp_targets[k] = widen(tg, B);
        }
        const size_t nUo = (size_t)(T - 1) * wo;
        std::vector<double> U0o = in.U0;
        if (U0o.size() == nUo) { std::vector<double> t_; for (int b_ = 0; b_ < B; b_++) t_.insert(t_.end(), in.U0.begin(), in.U0.end()); U0o = t_; }
        if (U0o.size() != nUo * B) throw std::runtime_error("[ilqr_hip] U0 must be (T-1) x nb_ctrl_var per instance");
        in2.U0 = widen(U0o, (size_t)B * (T - 1));
        // run on the widened problem (the descriptor is already widened by lower()), then narrow the outputs
        widened = true;
        BatchResult r;
        try { r = run_batch(s, in2, nb_iter, gains, &d, pre_solve, solve, post_solve); } catch (...) { widened = false; throw; }
        widened = false;
        r.X = narrow(r.X, (size_t)B * T);
        r.U = narrow(r.U, (size_t)B * (T - 1));
        if (!r.fX.empty()) r.fX = narrow(r.fX, (size_t)B * T);
        if (!r.d.empty()) r.d = narrow(r.d, (size_t)B * (T - 1));
        if (!r.K.empty()) {  // [B][T-1][wn][wn] -> [B][T-1][wo][wo]
            std::vector<double> Ko((size_t)B * (T - 1) * wo * wo, 0.0);
            auto map = [&](int i) { return i < padn ? i : D; };
            for (size_t m_ = 0; m_ < (size_t)B * (T - 1); m_++)
                for (int a_ = 0; a_ < wo; a_++)
                    for (int b_ = 0; b_ < wo; b_++) Ko[(m_ * wo + a_) * wo + b_] = r.K[(m_ * wn + map(a_)) * wn + map(b_)];
            r.K = Ko;
        }
        r.n_x = r.n_u = r.n_f = wo;
        return r;
    }
    ProblemGuard g;
    check(ilqr_problem_create(ctx, &d, B, &g.p));
    auto tile = [&](const Vec& v, size_t per) {
        std::vector<double> o((size_t)B * per);
        for (int b = 0; b < B; b++) std::copy(v.begin(), v.begin() + per, o.begin() + (size_t)b * per);
        return o;
    };
    std::vector<double> q0 = in.q0.empty() ? tile(s.q0(), dof) : in.q0, dq0 = in.dq0.empty() ? tile(s.dq0(), dof) : in.dq0;
    if (q0.size() != (size_t)B * dof || dq0.size() != (size_t)B * dof) throw std::runtime_error("[ilqr_hip] q0/dq0 must be B x dof");
    check(ilqr_problem_set_init_state(g.p, q0.data(), dq0.data()));
    const auto& kps = s.getKeypoints();
    for (size_t k = 0; k < kps.size(); k++) {
        std::vector<double> tg = (k < in.kp_targets.size() && !in.kp_targets[k].empty()) ? in.kp_targets[k] : tile(kps[k]->targetFx(), dm.n_f);
        if (tg.size() != (size_t)B * dm.n_f) throw std::runtime_error("[ilqr_hip] keypoint targets must be B x nb_target_var");
        check(ilqr_problem_set_keypoint_targets(g.p, (int)k, tg.data()));
    }
    const size_t nU = (size_t)(T - 1) * dm.n_u;
    std::vector<double> U0;
    if (in.U0.size() == nU * B) U0 = in.U0;
    else if (in.U0.size() == nU) U0 = tile(in.U0, nU);
    else throw std::runtime_error("[ilqr_hip] U0 must be (T-1) x nb_ctrl_var per instance");
    check(ilqr_problem_set_controls(g.p, U0.data()));
    if (pre_solve) pre_solve(g.p);
    const auto t0 = std::chrono::steady_clock::now();
    solve(g.p);
    check(ilqr_ctx_synchronize(ctx));
    const std::chrono::duration<double> el = std::chrono::steady_clock::now() - t0;
    BatchResult r;
    r.B = B; r.T = T; r.n_x = dm.n_x; r.n_u = dm.n_u; r.n_f = dm.n_f; r.nb_iter = nb_iter; r.seconds = el.count();
    r.X.resize((size_t)B * T * dm.n_x);
    r.U.resize((size_t)B * nU);
    r.cost.resize(B); r.alpha.resize(B); r.iters.resize(B); r.status.resize(B);
    check(ilqr_problem_get_X(g.p, r.X.data()));
    check(ilqr_problem_get_U(g.p, r.U.data()));
    check(ilqr_problem_get_cost(g.p, r.cost.data()));
    check(ilqr_problem_get_alpha(g.p, r.alpha.data()));
    check(ilqr_problem_get_iters(g.p, r.iters.data()));
    check(ilqr_problem_get_status(g.p, r.status.data()));
    if (in.want_fX) { r.fX.resize((size_t)B * T * dm.n_f); check(ilqr_problem_get_fX(g.p, r.fX.data())); }
    if (gains && in.want_gains && nb_iter > 0) {
        r.K.resize((size_t)B * nU * dm.n_x);
        r.d.resize((size_t)B * nU);
        check(ilqr_problem_get_K(g.p, r.K.data()));
        check(ilqr_problem_get_d(g.p, r.d.data()));
    }
    if (nb_iter > 0) {
        r.cost_trace.resize((size_t)B * nb_iter);
        r.alpha_trace.resize((size_t)B * nb_iter);
        check(ilqr_problem_get_trace(g.p, r.cost_trace.data(), r.alpha_trace.data(), nb_iter));
    }
    if (post_solve) post_solve(g.p);
    return r;
}

// the per-iteration stream of the reference ("Iteration i, Cost: c, alpha= a[, time= t]") for instance 0
static void emit_trace(const BatchResult& r, CallBackMessage* cb, bool with_time) {
    const int n = r.iters.empty() ? 0 : r.iters[0];
    for (int i = 0; i < n; i++) {
        std::string msg = "Iteration " + std::to_string(i + 1) + ", Cost: " + fmt(r.cost_trace[i]) + ", alpha= " + fmt(r.alpha_trace[i]);
        if (with_time) msg += ", time= " + fmt(r.seconds / std::max(1, n));
        if (cb == nullptr) std::cout << msg << std::endl;
        else cb->notify(msg);
    }
}

static std::vector<double> flatten(const std::vector<Vec>& U0, int T, int nu) {
    if ((int)U0.size() != T - 1) throw std::runtime_error("[solver] U0 must hold horizon-1 control vectors");
    std::vector<double> o;
    for (auto& u : U0) {
        if ((int)u.size() != nu) throw std::runtime_error("[solver] each U0 entry must have nb_ctrl_var entries");
        o.insert(o.end(), u.begin(), u.end());
    }
    return o;
}
static std::vector<Vec> rows_of(const std::vector<double>& v, int n, int w) {
    std::vector<Vec> o(n);
    for (int i = 0; i < n; i++) o[i] = Vec(v.begin() + (size_t)i * w, v.begin() + (size_t)(i + 1) * w);
    return o;
}

BatchResult ILQRRecursive::solveBatch(const BatchInputs& in, int nb_iter, bool line_search, bool early_stop) {
    return run_batch(*s, in, nb_iter, true, nullptr, nullptr, [&](ilqr_problem* p) { check(ilqr_solve_recursive(p, nb_iter, line_search, early_stop)); }, nullptr);
}

std::tuple<std::vector<Vec>, std::vector<Vec>, std::vector<Vec>, std::vector<Mat>, std::vector<Vec>, double> ILQRRecursive::solve(
    const std::vector<Vec>& U0, int nb_iter, bool line_search, bool early_stop, CallBackMessage* cb) {  // ILQRRecursive.cpp:21-181
    if (!s->builtin()) return solve_over_virtuals(*s, U0, nb_iter, line_search, early_stop, cb);  // user-defined System / Keypoint
    BatchInputs in;
    in.B = 1;
    in.U0 = flatten(U0, s->getHorizon(), s->getNbCtrlVar());
    const BatchResult r = solveBatch(in, nb_iter, line_search, early_stop);
    emit_trace(r, cb, true);
    s->reset();
    const int T = r.T;
    std::vector<Mat> Ks;
    std::vector<Vec> ds;
    if (!r.K.empty()) {
        for (int k = 0; k < T - 1; k++) {
            Mat K(r.n_u, r.n_x);
            std::copy(r.K.begin() + (size_t)k * r.n_u * r.n_x, r.K.begin() + (size_t)(k + 1) * r.n_u * r.n_x, K.d.begin());
            Ks.push_back(K);
        }
        ds = rows_of(r.d, T - 1, r.n_u);
    }
    return std::make_tuple(rows_of(r.X, T, r.n_x), rows_of(r.fX, T, r.n_f), rows_of(r.U, T - 1, r.n_u), Ks, ds, r.cost[0]);
}

AL_ILQR::AL_ILQR(const std::shared_ptr<sys::System>& s_, const std::vector<Constraint>& ineq, const std::vector<Vec>& initLambda)
    : s(s_), inequality(ineq), multipliers(initLambda) {}

BatchResult AL_ILQR::solveBatch(const BatchInputs& in, int nb_iter, int lag, double penalty, double scaling, bool line_search, bool early_stop) {
    const int T = s->getHorizon(), ns = s->getNbStateVar() + s->getNbCtrlVar();
    if ((int)inequality.size() != T - 1 || (int)multipliers.size() != T - 1)
        throw std::runtime_error("[AL_ILQR] need one Constraint and one multiplier vector per timestep (horizon-1)");
    const int m_full = inequality[0].A.rows;
    for (auto& c : inequality)
        if (c.A.rows != m_full || c.A.cols != ns || (int)c.b.size() != m_full)
            throw std::runtime_error("[AL_ILQR] every Constraint must be m x (nb_state_var+nb_ctrl_var) with an m-vector b");
    // rows that are identically zero for every k contribute nothing to the backward pass: they are dropped on the device
    // and their multipliers follow lambda <- max(0, lambda + penalty * (-b)) on the host (AL-ILQR.cpp:202-208)
    std::vector<int> keep;
    for (int r_ = 0; r_ < m_full; r_++) {
        bool nz = false;
        for (auto& c : inequality)
            for (int j = 0; j < ns && !nz; j++) nz = c.A(r_, j) != 0.0;
        if (nz) keep.push_back(r_);
    }
    const int m = (int)keep.size();
    if (m == 0) throw std::runtime_error("[AL_ILQR] all constraint rows are zero");
    bool per_step = false;
    for (int k = 1; k < T - 1 && !per_step; k++)
        for (int r_ : keep) {
            for (int j = 0; j < ns; j++) per_step = per_step || inequality[k].A(r_, j) != inequality[0].A(r_, j);
            per_step = per_step || inequality[k].b[r_] != inequality[0].b[r_];
        }
    const int nk = per_step ? T - 1 : 1;
    std::vector<double> A((size_t)nk * m * ns), b((size_t)nk * m), lam((size_t)in.B * (T - 1) * m);
    for (int k = 0; k < nk; k++)
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < ns; j++) A[((size_t)k * m + i) * ns + j] = inequality[k].A(keep[i], j);
            b[(size_t)k * m + i] = inequality[k].b[keep[i]];
        }
    for (int bi = 0; bi < in.B; bi++)
        for (int k = 0; k < T - 1; k++)
            for (int i = 0; i < m; i++) lam[((size_t)bi * (T - 1) + k) * m + i] = multipliers[k][keep[i]];
    std::vector<double> lam_out;
    BatchResult r = run_batch(
        *s, in, nb_iter, false, nullptr, [&](ilqr_problem* p) { check(ilqr_problem_set_constraints(p, m, per_step ? 1 : 0, A.data(), b.data(), lam.data())); },
        [&](ilqr_problem* p) { check(ilqr_solve_al(p, nb_iter, lag, penalty, scaling, line_search, early_stop)); },
        [&](ilqr_problem* p) {
            lam_out.resize(lam.size());
            check(ilqr_problem_get_lambda(p, lam_out.data()));
        });
    // persist instance 0's multipliers (the reference's `multipliers` member)
    double pen = penalty;
    const int it_run = r.iters.empty() ? 0 : r.iters[0];
    for (int it = 0; it < it_run; it++) {
        if ((it + 1) % lag == 0) {
            pen *= scaling;
            for (int k = 0; k < T - 1; k++)
                for (int r_ = 0; r_ < m_full; r_++)
                    if (std::find(keep.begin(), keep.end(), r_) == keep.end())
                        multipliers[k][r_] = std::max(0.0, multipliers[k][r_] + pen * (0.0 - inequality[k].b[r_]));
        }
    }
    for (int k = 0; k < T - 1; k++)
        for (int i = 0; i < m; i++) multipliers[k][keep[i]] = lam_out[(size_t)k * m + i];
    return r;
}

std::tuple<std::vector<Vec>, std::vector<Vec>, std::vector<Vec>> AL_ILQR::solve(const std::vector<Vec>& U0, int nb_iter, int lag, double penalty, double scaling,
                                                                              bool line_search, bool early_stop, CallBackMessage* cb) {  // AL-ILQR.cpp:50-232
    if (!s->builtin())  // user-defined System / Keypoint / SimulationInterface: the same algorithm over its virtuals
        return solve_al_over_virtuals(*s, inequality, multipliers, U0, nb_iter, lag, penalty, scaling, line_search, early_stop, cb);
    BatchInputs in;
    in.B = 1;
    in.U0 = flatten(U0, s->getHorizon(), s->getNbCtrlVar());
    const BatchResult r = solveBatch(in, nb_iter, lag, penalty, scaling, line_search, early_stop);
    emit_trace(r, cb, true);
    s->reset();
    return std::make_tuple(rows_of(r.X, r.T, r.n_x), rows_of(r.fX, r.T, r.n_f), rows_of(r.U, r.T - 1, r.n_u));
}

BatchILQRCP::BatchILQRCP(const std::shared_ptr<sys::System>& s_, const Mat& Q_, const Mat& psi) : s(s_), PSI(psi), Q(Q_), custom_Q(true) {}
BatchILQRCP::BatchILQRCP(const std::shared_ptr<sys::System>& s_, const Mat& psi) : s(s_), PSI(psi), custom_Q(false) {}

// psi == nullptr: BatchILQR (identity basis, never materialised)
static BatchResult batch_gauss_newton(sys::System& s, const Mat* psi, const Mat* Q, const BatchInputs& in, int nb_iter, bool early_stop) {
    const int T = s.getHorizon(), nu = s.getNbCtrlVar();
    if (psi && psi->rows != (T - 1) * nu) throw std::runtime_error("[BatchILQRCP] psi must have (horizon-1)*nb_ctrl_var rows");
    ilqr_problem_desc d;
    s.lower(&d);
    if (Q) {  // BatchILQRCP.cpp:21-26 / BatchILQR.cpp:22-26: a user Q replaces the keypoints' precisions; the device takes its diagonal blocks
        const int nq = s.getNbQVar(), nkp = d.n_kp;
        if (Q->rows != nkp * nq || Q->cols != nkp * nq) throw std::runtime_error("[BatchILQRCP] Q must be (n_keypoints*nb_Q_var) square (sparse form)");
        for (int a = 0; a < Q->rows; a++)
            for (int b = 0; b < Q->cols; b++) {
                if (a / nq == b / nq) d.kp_Q[a / nq][(a % nq) * nq + (b % nq)] = (*Q)(a, b);
                else if ((*Q)(a, b) != 0.0) throw std::runtime_error("[ilqr_hip] a Q coupling different keypoints is not supported on the device");
            }
    }
    BatchInputs in2 = in;
    in2.want_gains = false;
    return run_batch(s, in2, nb_iter, false, &d, nullptr,
                     [&](ilqr_problem* p) {
                         if (psi) check(ilqr_solve_batch_cp(p, psi->d.data(), psi->cols, nb_iter, early_stop));
                         else check(ilqr_solve_batch(p, nb_iter, early_stop));
                     },
                     nullptr);
}

BatchResult BatchILQRCP::solveBatch(const BatchInputs& in, int nb_iter, bool early_stop) {
    return batch_gauss_newton(*s, &PSI, custom_Q ? &Q : nullptr, in, nb_iter, early_stop);
}

BatchILQR::BatchILQR(const std::shared_ptr<sys::System>& s_, const Mat& Q_) : s(s_), Q(Q_), custom_Q(true) {}
BatchILQR::BatchILQR(const std::shared_ptr<sys::System>& s_) : s(s_), custom_Q(false) {}
BatchResult BatchILQR::solveBatch(const BatchInputs& in, int nb_iter, bool early_stop) {
    return batch_gauss_newton(*s, nullptr, custom_Q ? &Q : nullptr, in, nb_iter, early_stop);
}
Vec BatchILQR::solve(int nb_iter, const Vec& u0, bool early_stop, CallBackMessage* cb) {  // BatchILQR.cpp:110-173
    if (!s->builtin()) return solve_batch_over_virtuals(*s, nullptr, custom_Q ? Q : s->getQMatrix(true), nb_iter, u0, early_stop, cb);  // user-defined System
    BatchInputs in;
    in.B = 1;
    in.U0 = u0;
    in.want_fX = false;
    const BatchResult r = solveBatch(in, nb_iter, early_stop);
    emit_trace(r, cb, false);
    s->reset();
    return r.U;
}

Vec BatchILQRCP::solve(int nb_iter, const Vec& u0, bool early_stop, CallBackMessage* cb) {  // BatchILQRCP.cpp:109-175
    if (!s->builtin()) return solve_batch_over_virtuals(*s, &PSI, custom_Q ? Q : s->getQMatrix(true), nb_iter, u0, early_stop, cb);  // user-defined System
    BatchInputs in;
    in.B = 1;
    in.U0 = u0;
    in.want_fX = false;
    const BatchResult r = solveBatch(in, nb_iter, early_stop);
    emit_trace(r, cb, false);
    s->reset();
    return r.U;
}

}  // namespace solver
}  // namespace ilqr_planner
