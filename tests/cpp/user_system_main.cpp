// A user-defined sys::System (a planar point mass: double integrator, quadratic terminal cost) solved by solver::ILQRRecursive over its
// virtuals -- SURVEY section 8(b): "user-defined System / Keypoint subclasses run the CPU loop".  Checked without any reference to the
// solver's internals: the returned trajectory is the rollout of the returned controls, the cost is the cost of that trajectory, and the
// controls are a stationary point of the total cost (finite differences over every control entry).  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include <stdexcept>

#include "../../ilqr_planner_amd/csrc/host/ilqr_host.hpp"

using namespace ilqr_planner;

// a user-defined keypoint for the batch solvers (they look at the system through its keypoints: getKpIndexes, diffBatch, getQMatrix)
struct PointKp : sys::Keypoint {
    Vec target, qd;
    PointKp(int t, const Vec& tg, const Vec& q) : sys::Keypoint(t, sys::Keypoint::SECOND_ORDER, "POINT_KP"), target(tg), qd(q) {}
    Vec diff(const Vec& st) const override { Vec e(4); for (int i = 0; i < 4; i++) e[i] = target[i] - st[i]; return e; }
    Vec getState() const override { return target; }
    Mat getPrecision() const override { Mat P(4, 4); for (int i = 0; i < 4; i++) P(i, i) = qd[i]; return P; }
    Vec targetFx() const override { return target; }
};

struct PointSys : sys::System {
    double dt;
    Vec x, xt, qdiag;
    // call log of the line search: the reference steps the system BEFORE it asks for the cost of the step (ILQRRecursive.cpp:133-150) and
    // resets it at the top of every iteration before cost_F_xx (:66)
    int n_fp = 0, n_cost_before_fp = 0, n_reset = 0, n_Fxx_without_reset = 0;
    bool stepped_since_cost = true, reset_since_Fxx = true, in_total = false;
    PointSys(int T, double dt_, const std::vector<std::shared_ptr<sys::Keypoint>>& kps = {}) : sys::System(nullptr, kps, Vec{1e-2, 1e-2}, T, 2, {}), dt(dt_) {
        nb_state_var_ = 4; nb_ctrl_var_ = 2; nb_target_var_ = 4; nb_Q_var_ = 4;
        x0_ = Vec{0.2, -0.1, 0.0, 0.3};
        xt = Vec{1.0, 0.5, 0.0, 0.0};
        qdiag = Vec{10, 10, 1, 1};
        x = x0_;
    }
    Vec getState() override { return x; }
    void reset() override { x = x0_; n_reset++; reset_since_Fxx = true; }
    std::tuple<Vec, Mat> getFxJac() override { return std::make_tuple(x, Mat::Identity(4)); }
    StepOut forwardPass(const Vec&, const Vec& u, int) override {  // advances the "simulator" (this object), as the reference's systems do
        Mat A = Mat::Identity(4), B(4, 2);
        A(0, 2) = dt; A(1, 3) = dt;
        B(0, 0) = dt * dt / 2; B(1, 1) = dt * dt / 2; B(2, 0) = dt; B(3, 1) = dt;
        Vec xn(4);
        xn[0] = x[0] + dt * x[2] + dt * dt / 2 * u[0];
        xn[1] = x[1] + dt * x[3] + dt * dt / 2 * u[1];
        xn[2] = x[2] + dt * u[0];
        xn[3] = x[3] + dt * u[1];
        x = xn;
        n_fp++; stepped_since_cost = true;
        return std::make_tuple(x, x, A, B, Mat::Identity(4));
    }
    // the reference's convention: cost = e'Qe + u'Ru, derivatives WITHOUT the factor 2 (cost_x = -J'Qe, cost_xx = J'QJ, cost_u = Ru)
    Mat cost_F_xx(const Vec& xk) override { if (!reset_since_Fxx) n_Fxx_without_reset++; reset_since_Fxx = false; return sys::System::cost_F_xx(xk); }
    Vec cost(const Vec& xk, const Vec& uk, int k) override {
        if (!in_total && k > 0 && k < horizon_ - 1 && !stepped_since_cost) n_cost_before_fp++;  // two costs in a row without a step between
        if (k < horizon_ - 1) stepped_since_cost = false;
        double c = 0;
        if (k == horizon_ - 1)
            for (int i = 0; i < 4; i++) c += qdiag[i] * (xt[i] - xk[i]) * (xt[i] - xk[i]);
        for (int i = 0; i < 2; i++) c += Rdiag[i] * uk[i] * uk[i];
        return Vec(1, c);
    }
    Vec cost_x(const Vec& xk, const Vec&, int k) override {
        Vec g(4, 0.0);
        if (k == horizon_ - 1)
            for (int i = 0; i < 4; i++) g[i] = -qdiag[i] * (xt[i] - xk[i]);
        return g;
    }
    Mat cost_xx(const Vec&, const Vec&, int k) override {
        Mat H(4, 4);
        if (k == horizon_ - 1)
            for (int i = 0; i < 4; i++) H(i, i) = qdiag[i];
        return H;
    }
    double total(const std::vector<Vec>& U) {  // independent of the solver: plain rollout
        in_total = true;
        reset();
        double c = 0;
        for (int k = 0; k < horizon_ - 1; k++) {
            c += cost(x, U[k], k)[0];
            forwardPass(x, U[k], k);
        }
        c += cost(x, Vec(2, 0.0), horizon_ - 1)[0];
        in_total = false; stepped_since_cost = true;
        return c;
    }
};

struct Collect : CallBackMessage {
    int n = 0, with_time = 0;
    void notify(const std::string& m) override { n++; if (m.find(", time= ") != std::string::npos) with_time++; }
};

// A user subclass of sim::KDLRobot that changes the kinematics: it inherits lowerChain(), but the device would solve the BASE chain --
// it must be taken over its virtuals (System::builtin() looks at the exact type of the simulator)
struct OffsetRobot : sim::KDLRobot {
    using sim::KDLRobot::KDLRobot;
    void updateKinematics() override { sim::KDLRobot::updateKinematics(); x[2] += 0.05; }
};

#define CHECK(cond)                                                    \
    do {                                                               \
        if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); return 1; } \
    } while (0)

int main() {
    const int T = 30;
    auto s = std::make_shared<PointSys>(T, 0.1);
    CHECK(!s->builtin());
    std::vector<Vec> U0(T - 1, Vec{0.1, -0.2});
    const double c_init = s->total(U0);
    solver::ILQRRecursive solver_(s);
    Collect cb;
    auto out = solver_.solve(U0, 3, true, false, &cb);
    const auto& X = std::get<0>(out);
    const auto& U = std::get<2>(out);
    const auto& K = std::get<3>(out);
    const auto& d = std::get<4>(out);
    const double cost = std::get<5>(out);
    CHECK(cb.n == 3);
    CHECK((int)X.size() == T && (int)U.size() == T - 1 && (int)K.size() == T - 1 && (int)d.size() == T - 1);
    CHECK(K[0].rows == 2 && K[0].cols == 4);
    CHECK(cost < c_init * 1e-2);
    // the cost is the cost of the returned controls, the states are their rollout
    CHECK(std::fabs(s->total(U) - cost) <= 1e-12 * std::fabs(cost));
    s->reset();
    for (int k = 0; k < T - 1; k++) {
        for (int i = 0; i < 4; i++) CHECK(std::fabs(s->getState()[i] - X[k][i]) <= 1e-13);
        s->forwardPass(X[k], U[k], k);
    }
    // stationary point of the (linear-quadratic) problem: central differences over every control entry
    for (int k = 0; k < T - 1; k++)
        for (int i = 0; i < 2; i++) {
            auto Up = U, Um = U;
            Up[k][i] += 1e-4; Um[k][i] -= 1e-4;
            const double g = (s->total(Up) - s->total(Um)) / 2e-4;
            CHECK(std::fabs(g) <= 1e-6);
        }
    // early stop needs cost < 1e-3: never here, all iterations run; without line search alpha stays 1
    auto out2 = solver_.solve(U0, 2, false, true, &cb);
    CHECK(std::get<5>(out2) <= cost * (1 + 1e-6) + 1e-12);
    // everything but ILQRRecursive::solve refuses a system without device lowering
    bool threw = false;
    try {
        solver::BatchInputs in;
        in.B = 1;
        for (auto& u : U0) in.U0.insert(in.U0.end(), u.begin(), u.end());
        solver_.solveBatch(in, 1, true, false);
    } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
    CHECK(cb.with_time == cb.n);  // "Iteration i, Cost: c, alpha= a, time= t" (ILQRRecursive.cpp:168)
    CHECK(s->n_cost_before_fp == 0);  // forwardPass before cost in the line search
    CHECK(s->n_Fxx_without_reset == 0);  // reset() before every backward pass
    // AL_ILQR over the virtuals (AL-ILQR.cpp:50-232): one row u_0 <= half the largest value the unconstrained optimum takes, multipliers start at 0
    {
        double umax = -1e9;
        for (int k = 0; k < T - 1; k++) umax = std::fmax(umax, U[k][0]);
        CHECK(umax > 0.05);
        const double bound = 0.5 * umax;
        auto s2 = std::make_shared<PointSys>(T, 0.1);
        solver::Constraint c;
        c.A = Mat(1, 6);
        c.A(0, 4) = 1.0;
        c.b = Vec{bound};
        std::vector<solver::Constraint> cons(T - 1, c);
        std::vector<Vec> lam0(T - 1, Vec{0.0});
        solver::AL_ILQR al(s2, cons, lam0);
        Collect cb2;
        auto o = al.solve(U0, 40, 2, 1.0, 2.0, false, false, &cb2);  // full steps: the reference line search judges by the plain cost (AL-ILQR.cpp:193-199), which a step towards the bound raises
        CHECK(cb2.n == 40 && cb2.with_time == 40);
        const auto& Ua = std::get<2>(o);
        double worst = -1e9;
        for (int k = 0; k < T - 1; k++) worst = std::fmax(worst, Ua[k][0] - bound);
        std::printf("AL: unconstrained max u0 %.4g, bound %.4g, worst violation %.3g\n", umax, bound, worst);
        CHECK(worst < 0.05 * umax);   // the unconstrained optimum violates the bound by umax / 2; the augmented-Lagrangian solve respects it
        CHECK(s2->n_cost_before_fp == 0 && s2->n_Fxx_without_reset == 0);
    }
    // BatchILQR / BatchILQRCP over the virtuals (BatchILQR.cpp:111-173, BatchILQRCP.cpp:109-175): the same point mass with its goal as a user keypoint
    {
        auto kp = std::make_shared<PointKp>(T - 1, Vec{1.0, 0.5, 0.0, 0.0}, Vec{10, 10, 1, 1});
        auto sb = std::make_shared<PointSys>(T, 0.1, std::vector<std::shared_ptr<sys::Keypoint>>{kp});
        CHECK(!sb->builtin());
        Vec u0;
        for (auto& u : U0) u0.insert(u0.end(), u.begin(), u.end());
        const int resets0 = sb->n_reset;
        solver::BatchILQR bs(sb);
        Collect cbb;
        const Vec ub = bs.solve(6, u0, false, &cbb);
        CHECK(cbb.n == 6 && cbb.with_time == 0);          // "Iteration i, Cost: c, alpha= a" -- no time field in the batch solvers
        CHECK(sb->n_reset >= resets0 + 2);                 // reset() at the start and at the end of solve (:111,171)
        CHECK((int)ub.size() == 2 * (T - 1));
        std::vector<Vec> Ub(T - 1, Vec(2));
        for (int k = 0; k < T - 1; k++) { Ub[k][0] = ub[2 * k]; Ub[k][1] = ub[2 * k + 1]; }
        const double cb_final = sb->total(Ub);
        std::printf("BatchILQR over the virtuals: cost %.6g -> %.6g (Riccati optimum %.6g)\n", c_init, cb_final, cost);
        CHECK(cb_final < c_init * 2e-2);                   // descends (the reference's sensitivities are shifted by one control: not the exact Newton step)
        CHECK(cb_final >= cost * (1 - 1e-9));              // ... and cannot beat the optimum of the same linear-quadratic problem
        // the identity basis in BatchILQRCP is the same computation
        Mat I(2 * (T - 1), 2 * (T - 1));
        for (int i = 0; i < I.rows; i++) I(i, i) = 1.0;
        solver::BatchILQRCP bcp(sb, I);
        Collect cbc;
        const Vec uc = bcp.solve(6, u0, false, &cbc);
        CHECK(cbc.n == 6);
        for (size_t i = 0; i < ub.size(); i++) CHECK(std::fabs(uc[i] - ub[i]) <= 1e-9 * (1 + std::fabs(ub[i])));
        // a two-column basis per control (constant + ramp over the horizon): the solve stays in its span
        Mat P(2 * (T - 1), 4);
        for (int k = 0; k < T - 1; k++)
            for (int i = 0; i < 2; i++) { P(2 * k + i, i) = 1.0; P(2 * k + i, 2 + i) = (double)k / (T - 2); }
        solver::BatchILQRCP bcp2(sb, P);
        const Vec u00(2 * (T - 1), 0.0);
        const Vec u2 = bcp2.solve(5, u00, false, &cbc);
        for (int k = 1; k + 1 < T - 1; k++)               // affine in k: second differences vanish
            for (int i = 0; i < 2; i++) CHECK(std::fabs(u2[2 * (k + 1) + i] - 2 * u2[2 * k + i] + u2[2 * (k - 1) + i]) <= 1e-9);
        std::vector<Vec> U2(T - 1, Vec(2));
        for (int k = 0; k < T - 1; k++) { U2[k][0] = u2[2 * k]; U2[k][1] = u2[2 * k + 1]; }
        std::vector<Vec> Uz(T - 1, Vec(2, 0.0));
        CHECK(sb->total(U2) < sb->total(Uz));
    }
    std::printf("ok: cost %.6g -> %.6g in 3 iterations\n", c_init, cost);
    return 0;
}
