// ilqr_host_loop.cpp -- ILQRRecursive, AL_ILQR, BatchILQR and BatchILQRCP over the virtual System interface, for USER-DEFINED System / Keypoint / SimulationInterface
// subclasses only.
//
// SURVEY.md section 8(b), last bullet: the reference's solver is written against the virtuals of sys::System (forwardPass, cost, cost_x,
// cost_xx, cost_F*; ILQRRecursive.cpp:21-181), so a C++ user may hand it a System of their own.  Such a system has no lowering to the
// device descriptor (include/ilqr_hip.h describes the built-in system shapes), so the mirror runs it with this loop: the same algorithm
// stated over the mirror's virtuals, on the mirror's own Vec / Mat.
//
// What this is NOT: a fallback for the device path.  Every built-in system shape (PosOrn / PosOrnTime / JointSpace / JointSpaceTime
// planner systems, sequences of them, the built-in keypoints) is lowered and solved on the GPU, and fails loudly when that is not possible;
// ILQRRecursive::solve chooses by the TYPE of the system (System::builtin()), never by whether a device call succeeded.  Nothing under
// oracle/ is used here or anywhere else in the product.
#include <chrono>
#include <cmath>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "ilqr_host.hpp"

namespace ilqr_planner {
namespace solver {
namespace {

Mat mul(const Mat& a, const Mat& b) {
    Mat o(a.rows, b.cols);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < b.cols; j++) {
            double s = 0;
            for (int l = 0; l < a.cols; l++) s += a(i, l) * b(l, j);
            o(i, j) = s;
        }
    return o;
}
Mat tmul(const Mat& a, const Mat& b) {  // a^T b
    Mat o(a.cols, b.cols);
    for (int i = 0; i < a.cols; i++)
        for (int j = 0; j < b.cols; j++) {
            double s = 0;
            for (int l = 0; l < a.rows; l++) s += a(l, i) * b(l, j);
            o(i, j) = s;
        }
    return o;
}
Vec mulv(const Mat& a, const Vec& v) {
    Vec o(a.rows, 0.0);
    for (int i = 0; i < a.rows; i++) {
        double s = 0;
        for (int l = 0; l < a.cols; l++) s += a(i, l) * v[l];
        o[i] = s;
    }
    return o;
}
Vec tmulv(const Mat& a, const Vec& v) {  // a^T v
    Vec o(a.cols, 0.0);
    for (int i = 0; i < a.cols; i++) {
        double s = 0;
        for (int l = 0; l < a.rows; l++) s += a(l, i) * v[l];
        o[i] = s;
    }
    return o;
}
Mat add(const Mat& a, const Mat& b) {
    Mat o = a;
    for (size_t i = 0; i < o.d.size(); i++) o.d[i] += b.d[i];
    return o;
}
Vec addv(const Vec& a, const Vec& b) {
    Vec o = a;
    for (size_t i = 0; i < o.size(); i++) o[i] += b[i];
    return o;
}
// inverse by LU with partial pivoting, solved against the identity (what MatrixXd::inverse() does for these sizes)
Mat inverse(Mat m) {
    const int n = m.rows;
    if (m.cols != n) throw std::runtime_error("[ILQRRecursive] Quu is not square");
    std::vector<int> piv(n);
    for (int i = 0; i < n; i++) piv[i] = i;
    for (int k = 0; k < n; k++) {
        int r = k;
        double best = std::fabs(m(k, k));
        for (int i = k + 1; i < n; i++)
            if (std::fabs(m(i, k)) > best) { best = std::fabs(m(i, k)); r = i; }
        if (r != k) {
            for (int j = 0; j < n; j++) std::swap(m(k, j), m(r, j));
            std::swap(piv[k], piv[r]);
        }
        const double pv = m(k, k);
        for (int i = k + 1; i < n; i++) {
            m(i, k) /= pv;
            const double f = m(i, k);
            for (int j = k + 1; j < n; j++) m(i, j) -= f * m(k, j);
        }
    }
    Mat inv(n, n);
    for (int c = 0; c < n; c++) {
        for (int i = 0; i < n; i++) {
            double s = (piv[i] == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) s -= m(i, j) * inv(j, c);
            inv(i, c) = s;
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = inv(i, c);
            for (int j = i + 1; j < n; j++) s -= m(i, j) * inv(j, c);
            inv(i, c) = s / m(i, i);
        }
    }
    return inv;
}
std::string num(double v) {
    std::ostringstream o;
    o << v;
    return o.str();
}

}  // namespace

// ILQRRecursive::solve (ILQRRecursive.cpp:21-181) and AL_ILQR::solve (AL-ILQR.cpp:50-232) over the virtual interface of `s`, with the
// reference's ORDER of calls into the system (a user's System may be stateful -- a simulator advanced by forwardPass): reset() at the top
// of the recursive solve (:29) but not of the AL one, reset() at the top of every iteration before cost_F_xx (:66 / AL :92), and in the
// line search reset(), getFxJac(), then per step forwardPass BEFORE cost (:131-147 / AL :165-185).  con == nullptr: plain recursive.
namespace {
struct ALState {
    const std::vector<Constraint>* con;
    std::vector<Vec>* lambda;
    int lag;
    double penalty, scaling;
};

// AL_ILQR::constraints (AL-ILQR.cpp:21-44): g = A [x; u] - b, I = diag(g < 0 && lambda == 0 ? 0 : 1); an empty constraint gives zeros
void constraints_of(const ALState& al, const Vec& x, const Vec& u, int k, double pen, Vec& I_out, Vec& g_out) {
    const Constraint& c = al.con->at(k);
    const int m = (int)c.b.size();
    I_out.assign(m, 0.0);
    g_out.assign(m, 0.0);
    if (m == 0) return;
    const int nx = (int)x.size(), ns = nx + (int)u.size();
    if (c.A.rows != m || c.A.cols != ns) throw std::runtime_error("[AL_ILQR] constraint A must be m x (nb_state_var + nb_ctrl_var)");
    for (int r = 0; r < m; r++) {
        double g = 0;
        for (int j = 0; j < ns; j++) g += c.A(r, j) * (j < nx ? x[j] : u[j - nx]);
        g -= c.b[r];
        g_out[r] = g;
        I_out[r] = pen * ((g < 0 && al.lambda->at(k)[r] == 0) ? 0.0 : 1.0);
    }
}

std::tuple<std::vector<Vec>, std::vector<Vec>, std::vector<Vec>, std::vector<Mat>, std::vector<Vec>, double> riccati_over_virtuals(
    sys::System& s, const std::vector<Vec>& U0, int nb_iter, bool line_search, bool early_stop, CallBackMessage* cb, ALState* al) {
    const int T = s.getHorizon(), nu = s.getNbCtrlVar();
    if ((int)U0.size() != T - 1) throw std::runtime_error("[solver] U0 must hold horizon-1 control vectors");
    for (auto& u : U0)
        if ((int)u.size() != nu) throw std::runtime_error("[solver] each U0 entry must have nb_ctrl_var entries");
    if (al && ((int)al->con->size() < T - 1 || (int)al->lambda->size() < T - 1)) throw std::runtime_error("[AL_ILQR] one constraint and one multiplier vector per control step");
    std::vector<Vec> X(T), fX(T), U = U0, nX(T), nfX(T), nU(T - 1), ds(T - 1), Is(T - 1), Cs(T - 1);
    std::vector<Mat> As(T - 1), Bs(T - 1), Ks(T - 1);
    double penalty = al ? al->penalty : 0.0;

    // initial rollout (:41-56 / AL :68-85): cost, then the step
    if (!al) s.reset();
    X[0] = s.getInitState();
    fX[0] = std::get<0>(s.getFxJac());
    if (al) (void)s.getFxJac();  // (AL :66 evaluates it a second time for J)
    double cost0 = 0;
    for (int k = 0; k < T - 1; k++) {
        if (al) constraints_of(*al, X[k], U[k], k, penalty, Is[k], Cs[k]);
        cost0 += s.cost(X[k], U[k], k)[0];
        auto st = s.forwardPass(X[k], U[k], k);
        X[k + 1] = std::get<0>(st);
        fX[k + 1] = std::get<1>(st);
        As[k] = std::get<2>(st);
        Bs[k] = std::get<3>(st);
    }
    cost0 += s.cost_F(X[T - 1])[0];

    double alpha = 1;
    int it_done = 0;
    for (int it = 0; it < nb_iter; it++) {
        const auto t_start = std::chrono::steady_clock::now();
        s.reset();
        // backward pass (:68-97 / AL :94-145)
        Mat P = s.cost_F_xx(X[T - 1]);
        Vec p = s.cost_F_x(X[T - 1]);
        for (int k = T - 2; k >= 0; k--) {
            const Mat &A = As[k], &B = Bs[k];
            const int nx = A.rows;
            const Mat BtP = tmul(B, P), AtP = tmul(A, P);
            Mat Qux = add(s.cost_ux(X[k], U[k], k), mul(BtP, A));
            Mat Quu = add(s.cost_uu(X[k], U[k], k), mul(BtP, B));
            Mat Qxx = add(s.cost_xx(X[k], U[k], k), mul(AtP, A));
            Mat Qxu = add(s.cost_xu(X[k], U[k], k), mul(AtP, B));
            Vec Qu = addv(s.cost_u(X[k], U[k], k), tmulv(B, p));
            Vec Qx = addv(s.cost_x(X[k], U[k], k), tmulv(A, p));
            if (al && !Is[k].empty()) {  // c_u' I c_x etc. and c' (lambda + I g)  (AL :124-129)
                const Mat& Ac = al->con->at(k).A;
                const Vec& lam = al->lambda->at(k);
                for (size_t r = 0; r < Is[k].size(); r++) {
                    const double w = lam[r] + Is[k][r] * Cs[k][r];
                    for (int i = 0; i < nu; i++) {
                        const double au = Ac((int)r, nx + i);
                        for (int j = 0; j < nx; j++) Qux(i, j) += au * Is[k][r] * Ac((int)r, j);
                        for (int j = 0; j < nu; j++) Quu(i, j) += au * Is[k][r] * Ac((int)r, nx + j);
                        Qu[i] += au * w;
                    }
                    for (int i = 0; i < nx; i++) {
                        const double ax = Ac((int)r, i);
                        for (int j = 0; j < nx; j++) Qxx(i, j) += ax * Is[k][r] * Ac((int)r, j);
                        for (int j = 0; j < nu; j++) Qxu(i, j) += ax * Is[k][r] * Ac((int)r, nx + j);
                        Qx[i] += ax * w;
                    }
                }
            }
            Mat Qr = Quu;
            for (int i = 0; i < nu; i++) Qr(i, i) += 1e-6;  // the regularisation enters the inverse only (:89)
            Mat Qi = inverse(Qr);
            for (auto& v : Qi.d) v = -1 * v;
            Ks[k] = mul(Qi, Qux);
            ds[k] = mulv(Qi, Qu);
            const Mat KtQuu = tmul(Ks[k], Quu);
            const Mat t1 = mul(KtQuu, Ks[k]), t2 = tmul(Ks[k], Qux), t3 = mul(Qxu, Ks[k]);
            for (size_t i = 0; i < P.d.size(); i++) P.d[i] = ((Qxx.d[i] + t1.d[i]) + t2.d[i]) + t3.d[i];
            const Vec v1 = mulv(KtQuu, ds[k]), v2 = tmulv(Ks[k], Qu), v3 = mulv(Qxu, ds[k]);
            for (size_t i = 0; i < p.size(); i++) p[i] = ((Qx[i] + v1[i]) + v2[i]) + v3[i];
        }
        // forward pass with the step-halving line search (:101-155): the last trial is accepted whatever its cost
        alpha = 2;
        double newCost = 0, dun = 0;
        do {
            s.reset();
            alpha /= 2.0;
            nX[0] = X[0];
            nfX[0] = std::get<0>(s.getFxJac());
            if (al) (void)s.getFxJac();
            dun = 0;
            newCost = 0;
            for (int k = 0; k < T - 1; k++) {
                Vec dx(nX[k].size());
                for (size_t i = 0; i < dx.size(); i++) dx[i] = nX[k][i] - X[k][i];
                Vec du = mulv(Ks[k], dx);
                double n2 = 0;
                for (int i = 0; i < nu; i++) { du[i] += alpha * ds[k][i]; n2 += du[i] * du[i]; }
                dun += std::sqrt(n2);
                nU[k] = addv(U[k], du);
                auto st = s.forwardPass(nX[k], nU[k], k);  // the step first, then the cost (:133-150)
                nX[k + 1] = std::get<0>(st);
                nfX[k + 1] = std::get<1>(st);
                As[k] = std::get<2>(st);
                Bs[k] = std::get<3>(st);
                if (al) constraints_of(*al, nX[k], nU[k], k, penalty, Is[k], Cs[k]);
                newCost += s.cost(nX[k], nU[k], k)[0];
            }
            newCost += s.cost_F(nX[T - 1])[0];
        } while (((newCost >= cost0) || std::isnan(newCost)) && alpha > 1e-3 && line_search);
        if (al && (it + 1) % al->lag == 0) {  // multiplier update with the UPDATED penalty (AL :202-208)
            penalty *= al->scaling;
            for (int k = 0; k < T - 1; k++)
                for (size_t r = 0; r < al->lambda->at(k).size() && r < Cs[k].size(); r++) {
                    const double v = al->lambda->at(k)[r] + penalty * Cs[k][r];
                    al->lambda->at(k)[r] = v > 0 ? v : 0;
                }
        }
        cost0 = newCost;
        X = nX; fX = nfX; U = nU;
        it_done = it + 1;
        const std::chrono::duration<double> dt = std::chrono::steady_clock::now() - t_start;
        const std::string msg = "Iteration " + std::to_string(it + 1) + ", Cost: " + num(cost0) + ", alpha= " + num(alpha) + ", time= " + num(dt.count());
        if (cb) cb->notify(msg);
        else std::cout << msg << std::endl;
        if (early_stop && alpha * std::sqrt(dun) < 1e-3 && (al || cost0 < 1e-3)) break;  // (:174; AL :225 has no cost test)
    }
    s.reset();
    std::vector<Vec> dso = ds;
    for (auto& v : dso)
        for (auto& e : v) e *= (it_done > 0 ? alpha : 1.0);  // the returned feed-forward terms are scaled by the accepted alpha (:128,144,162)
    if (it_done == 0) { Ks.clear(); dso.clear(); }
    return std::make_tuple(X, fX, U, Ks, dso, cost0);
}
// ---- BatchILQR / BatchILQRCP over the virtuals (BatchILQRCP.cpp:109-175, BatchILQR.cpp:111-173; psi == nullptr: the identity basis).
// One Gauss-Newton step on the whole control sequence per iteration: fpBatch (rollout with A, B, J, L per step), the sensitivities of the keypoint
// states assembled exactly as buildSuJL does (:61-97 -- including its order of statements: the block of a keypoint step is taken BEFORE that step's
// [A M, B] update, which is why the reference's W = Su PSI is the true sensitivity shifted by one control), the normal equations
// solved with inverse(), backtracking on the true cost until it improves or alpha < 1e-3.
Vec batch_over_virtuals(sys::System& s, const Mat* psi, const Mat& Q, int nb_iter, const Vec& u0, bool early_stop, CallBackMessage* cb) {
    const int n = s.getNbStateVar(), m = s.getNbCtrlVar(), T = s.getHorizon(), nf = s.getNbTargetVar(), N = m * (T - 1);
    const std::vector<int> kp = s.getKpIndexes();
    const int nk = (int)kp.size();
    if ((int)u0.size() != N) throw std::runtime_error("[BatchILQR] u0 must have (horizon - 1) * nb_ctrl_var entries");
    if (psi && psi->rows != N) throw std::runtime_error("[BatchILQRCP] psi must have (horizon - 1) * nb_ctrl_var rows");
    const Mat Rt = s.getRt();
    auto R_of = [&](int row) { return Rt(row % m, row % m); };  // R = diag(R_t) repeated along the horizon
    auto is_kp = [&](int i) { for (int t : kp) if (t == i) return true; return false; };
    auto pick = [&](const Vec& v, int sz) {  // truncateStates: the blocks of the keypoint steps, in keypoint order
        Vec o((size_t)nk * sz, 0.0);
        for (int t = 0; t < nk; t++) std::copy(v.begin() + (size_t)kp[t] * sz, v.begin() + (size_t)(kp[t] + 1) * sz, o.begin() + (size_t)t * sz);
        return o;
    };
    using Steps = std::vector<std::tuple<Mat, Mat, Mat, Mat>>;
    auto limits_of = [&](const Steps& st) {  // buildL: block-diagonal of the keypoint steps' L, in keypoint order
        Mat L(nk * n, nk * n);
        for (int t = 0; t < nk; t++) {
            const Mat& Lt = std::get<3>(st.at(kp[t]));
            for (int a = 0; a < n; a++) for (int b = 0; b < n; b++) L(t * n + a, t * n + b) = Lt(a, b);
        }
        return L;
    };
    auto quad = [&](const Vec& v, const Mat& M) { const Vec w = mulv(M, v); double c = 0; for (size_t i = 0; i < v.size(); i++) c += v[i] * w[i]; return c; };
    auto total = [&](const Vec& e, const Vec& u, const Vec& ql, const Mat& L) {
        double cu = 0;
        for (int i = 0; i < N; i++) cu += u[i] * R_of(i) * u[i];
        return quad(e, Q) + cu + quad(ql, L);
    };

    s.reset();
    Vec u = u0;
    for (int it = 0; it < nb_iter; it++) {
        auto fp = s.fpBatch(u);
        const Steps& st = std::get<2>(fp);
        if ((int)st.size() != T) throw std::runtime_error("[BatchILQR] fpBatch returned the wrong number of steps");
        const int jr = std::get<2>(st.at(0)).rows, jc = std::get<2>(st.at(0)).cols;
        Mat Su(nk * n, N), J(nk * jr, nk * jc), L(nk * n, nk * n);
        {   // buildSuJL, statement for statement
            Mat M = std::get<1>(st.at(0));
            int t = 0;
            for (int i = 0; i < T; i++) {
                const Mat& At = std::get<0>(st.at(i));
                const Mat& Bt = std::get<1>(st.at(i));
                if (is_kp(i)) {
                    const Mat& Lt = std::get<3>(st.at(i));
                    for (int a = 0; a < n; a++) for (int b = 0; b < n; b++) L(t * n + a, t * n + b) = Lt(a, b);
                    if (i > 0)
                        for (int a = 0; a < M.rows; a++) for (int b = 0; b < M.cols; b++) Su(t * n + a, b) = M(a, b);
                    const Mat& Jt = std::get<2>(st.at(i));
                    for (int a = 0; a < jr; a++) for (int b = 0; b < jc; b++) J(t * jr + a, t * jc + b) = Jt(a, b);
                    t++;
                }
                if (i > 0) {
                    const Mat AM = mul(At, M);
                    Mat Mn(M.rows, M.cols + Bt.cols);
                    for (int a = 0; a < M.rows; a++) {
                        for (int b = 0; b < M.cols; b++) Mn(a, b) = AM(a, b);
                        for (int b = 0; b < Bt.cols; b++) Mn(a, M.cols + b) = Bt(a, b);
                    }
                    M = Mn;
                }
            }
        }
        const Vec e = s.diffBatch(pick(std::get<0>(fp), nf));
        const Vec ql = pick(std::get<1>(fp), n);
        const Mat W = psi ? mul(Su, *psi) : Su;                               // Su PSI
        const Mat C = add(tmul(J, mul(Q, J)), L);                             // J'QJ + L
        Mat A = tmul(W, mul(C, W));                                           // + PSI'R PSI
        const int K = A.rows;
        for (int a = 0; a < K; a++)
            for (int b = 0; b < K; b++) {
                if (psi) { double r = 0; for (int i = 0; i < N; i++) r += (*psi)(i, a) * R_of(i) * (*psi)(i, b); A(a, b) += r; }
                else if (a == b) A(a, b) += R_of(a);
            }
        Vec rhs = tmulv(W, addv(tmulv(J, mulv(Q, e)), mulv(L, ql)));         // PSI'Su'(J'Q e + L ql) - PSI'R u
        for (int a = 0; a < K; a++) {
            if (psi) { double r = 0; for (int i = 0; i < N; i++) r += (*psi)(i, a) * R_of(i) * u[i]; rhs[a] -= r; }
            else rhs[a] -= R_of(a) * u[a];
        }
        const Vec dw = mulv(inverse(A), rhs);
        const Vec du = psi ? mulv(*psi, dw) : dw;
        const double cost0 = total(e, u, ql, L);
        double alpha = 1.0;
        while (true) {
            Vec ut = u;
            for (int i = 0; i < N; i++) ut[i] += alpha * du[i];
            auto ft = s.fpBatch(ut);
            const double c = total(s.diffBatch(pick(std::get<0>(ft), nf)), ut, pick(std::get<1>(ft), n), limits_of(std::get<2>(ft)));
            if ((c < cost0) || (alpha < 1e-3)) { u = ut; break; }
            alpha /= 2;
        }
        std::stringstream msg;
        msg << "Iteration " << it + 1 << ", Cost: " << cost0 << ", alpha= " << alpha;
        if (cb == nullptr) std::cout << msg.str() << std::endl;
        else cb->notify(msg.str());
        double dn = 0;
        for (double v : du) dn += v * v;
        if (early_stop && alpha * std::sqrt(dn) < 1e-3) break;
    }
    s.reset();
    return u;
}
}  // namespace

Vec solve_batch_over_virtuals(sys::System& s, const Mat* psi, const Mat& Q, int nb_iter, const Vec& u0, bool early_stop, CallBackMessage* cb) {
    return batch_over_virtuals(s, psi, Q, nb_iter, u0, early_stop, cb);
}

std::tuple<std::vector<Vec>, std::vector<Vec>, std::vector<Vec>, std::vector<Mat>, std::vector<Vec>, double> solve_over_virtuals(
    sys::System& s, const std::vector<Vec>& U0, int nb_iter, bool line_search, bool early_stop, CallBackMessage* cb) {
    return riccati_over_virtuals(s, U0, nb_iter, line_search, early_stop, cb, nullptr);
}

std::tuple<std::vector<Vec>, std::vector<Vec>, std::vector<Vec>> solve_al_over_virtuals(sys::System& s, const std::vector<Constraint>& inequality, std::vector<Vec>& multipliers,
                                                                                        const std::vector<Vec>& U0, int nb_iter, int lag_update_step, double penalty,
                                                                                        double scaling_factor, bool line_search, bool early_stop, CallBackMessage* cb) {
    if (lag_update_step <= 0) throw std::runtime_error("[AL_ILQR] lag_update_step must be positive");
    ALState al{&inequality, &multipliers, lag_update_step, penalty, scaling_factor};
    auto r = riccati_over_virtuals(s, U0, nb_iter, line_search, early_stop, cb, &al);
    return std::make_tuple(std::get<0>(r), std::get<1>(r), std::get<2>(r));
}

}  // namespace solver
}  // namespace ilqr_planner
