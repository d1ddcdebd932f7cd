// The simulator is the reference's main extension point (SimulationInterface.h:20).  A user subclass of sim::KDLRobot that changes the
// kinematics inherits lowerChain(), but the device would solve the BASE chain: System::builtin() must look at the exact type of the
// simulator and send such a system over its virtuals (csrc/host/ilqr_host_loop.cpp), while the plain KDLRobot goes to the GPU.
// argv[1] = URDF path.  Needs a GPU (KDLRobot::updateKinematics is a call of the FK kernel).  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "../../ilqr_planner_amd/csrc/host/ilqr_host.hpp"

using namespace ilqr_planner;

struct OffsetRobot : sim::KDLRobot {  // a tool 5 cm further along the base z axis
    using sim::KDLRobot::KDLRobot;
    void updateKinematics() override { sim::KDLRobot::updateKinematics(); x[2] += 0.05; }
};

struct SameRobot : sim::KDLRobot {  // a subclass that changes nothing: still taken over its virtuals (the type decides), so host and device can be compared
    using sim::KDLRobot::KDLRobot;
    void updateKinematics() override { sim::KDLRobot::updateKinematics(); }
};

struct Quiet : CallBackMessage {
    int n = 0;
    void notify(const std::string&) override { n++; }
};

struct Costs : CallBackMessage {  // "Iteration i, Cost: c, alpha= a"
    std::vector<double> cost, alpha;
    void notify(const std::string& m) override {
        const size_t a = m.find("Cost: "), b = m.find(", alpha= ");
        cost.push_back(std::stod(m.substr(a + 6, b - a - 6)));
        alpha.push_back(std::stod(m.substr(b + 9)));
    }
};

#define CHECK(cond)                                                    \
    do {                                                               \
        if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const Vec q0{0.62991112, -0.2329776, -0.01423721, -1.70254115, 0.06251303, 1.50592777, 0.71771416}, dq0(7, 0.0);
    const int T = 40;
    Mat Q(6, 6);
    for (int i = 0; i < 3; i++) { Q(i, i) = 1.0; Q(i + 3, i + 3) = 0.0; }  // position only
    const Vec target{0.45, 0.10, 0.45}, quat{0.0, 1.0, 0.0, 0.0};
    auto make = [&](std::shared_ptr<sim::SimulationInterface> r) {
        std::vector<std::shared_ptr<sys::Keypoint>> kps{std::make_shared<sys::PosOrnKeypoint>(target, quat, Q, T - 1)};
        return std::make_shared<sys::PosOrnPlannerSys>(r, kps, Vec(7, 1e-5), T, 1, 0.1);
    };
    auto plain = make(std::make_shared<sim::KDLRobot>(argv[1], "panda_link0", "panda_tip", q0, dq0));
    auto offset = make(std::make_shared<OffsetRobot>(argv[1], "panda_link0", "panda_tip", q0, dq0));
    CHECK(plain->builtin());
    CHECK(!offset->builtin());
    std::vector<Vec> U0(T - 1, Vec(7, 0.0));
    Quiet cb;
    auto a = solver::ILQRRecursive(plain).solve(U0, 6, true, false, &cb);
    auto b = solver::ILQRRecursive(offset).solve(U0, 6, true, false, &cb);
    CHECK(cb.n == 12);
    CHECK(std::get<5>(a) < 1e-4 && std::get<5>(b) < 1e-4);
    // each solve reaches the target with ITS OWN kinematics: the reported end-effector heights agree, the joint solutions differ, and the
    // base chain evaluated at the offset robot's solution ends 5 cm lower
    const Vec& fa = std::get<1>(a).back();
    const Vec& fb = std::get<1>(b).back();
    CHECK(std::fabs(fa[2] - target[2]) < 5e-3 && std::fabs(fb[2] - target[2]) < 5e-3);
    sim::KDLRobot probe(argv[1], "panda_link0", "panda_tip", std::get<0>(b).back(), dq0);
    CHECK(std::fabs(probe.getEEPosition()[2] - (target[2] - 0.05)) < 5e-3);
    // BatchILQRCP / BatchILQR: the device solve of the plain robot against the host loop over the virtuals of a subclass that changes nothing
    {
        auto same = make(std::make_shared<SameRobot>(argv[1], "panda_link0", "panda_tip", q0, dq0));
        CHECK(!same->builtin());
        const int N = 7 * (T - 1);
        Mat P(N, 14);  // per joint: a constant and a step at half the horizon
        for (int k = 0; k < T - 1; k++)
            for (int i = 0; i < 7; i++) { P(7 * k + i, i) = 1.0; P(7 * k + i, 7 + i) = (k >= (T - 1) / 2) ? 1.0 : 0.0; }
        const Vec u0(N, 0.0);
        Costs cd, ch;
        const Vec ud = solver::BatchILQRCP(plain, P).solve(5, u0, false, &cd);
        const Vec uh = solver::BatchILQRCP(same, P).solve(5, u0, false, &ch);
        CHECK(cd.cost.size() == 5 && ch.cost.size() == 5);
        for (int i = 0; i < 5; i++) {
            CHECK(cd.alpha[i] == ch.alpha[i]);
            CHECK(std::fabs(cd.cost[i] - ch.cost[i]) <= 1e-5 * std::fabs(cd.cost[i]) + 1e-12);  // (printed with six significant digits)
        }
        double worst = 0;
        for (int i = 0; i < N; i++) worst = std::fmax(worst, std::fabs(ud[i] - uh[i]));
        CHECK(worst <= 1e-6);
        Costs bd, bh;
        const Vec vd = solver::BatchILQR(plain).solve(3, u0, false, &bd);
        const Vec vh = solver::BatchILQR(same).solve(3, u0, false, &bh);
        CHECK(bd.cost.size() == 3 && bh.cost.size() == 3);
        for (int i = 0; i < 3; i++) CHECK(bd.alpha[i] == bh.alpha[i] && std::fabs(bd.cost[i] - bh.cost[i]) <= 1e-5 * std::fabs(bd.cost[i]) + 1e-12);
        worst = 0;
        for (int i = 0; i < N; i++) worst = std::fmax(worst, std::fabs(vd[i] - vh[i]));
        CHECK(worst <= 1e-6);
        std::printf("batch solvers: device and host loop agree (BatchILQRCP cost %.6g -> %.6g, BatchILQR %.6g -> %.6g)\n", cd.cost.front(), cd.cost.back(), bd.cost.front(), bd.cost.back());
    }
    std::printf("ok: plain on the device (z %.4f), subclass over its virtuals (z %.4f, base chain %.4f)\n", fa[2], fb[2], probe.getEEPosition()[2]);
    return 0;
}
