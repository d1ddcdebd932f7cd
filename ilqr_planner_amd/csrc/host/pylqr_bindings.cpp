// pylqr_bindings.cpp -- pybind11 module `PyLQR` with the reference's Python surface (pylqr_planner/src/bindings.cpp:48-908)
// for the classes the device hot path covers, plus the new `solve_batch` entry points.
// Same module / submodule / class / method / argument names; numpy in, numpy out (the reference's Eigen casters);
// no argument has a default, as in the reference.  Errors are std::runtime_error -> RuntimeError.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include "ilqr_host.hpp"

namespace py = pybind11;
using namespace ilqr_planner;
using arr_t = py::array_t<double, py::array::c_style | py::array::forcecast>;

// Vec <-> 1-D float64 ndarray, Mat <-> 2-D float64 ndarray (declared before <pybind11/stl.h> so they win over its list caster)
namespace pybind11 {
namespace detail {
template <>
struct type_caster<Vec> {
    PYBIND11_TYPE_CASTER(Vec, const_name("numpy.ndarray[float64[n]]"));
    bool load(handle src, bool) {
        if (!src || src.is_none()) return false;
        arr_t a = arr_t::ensure(src);
        if (!a) return false;
        value.assign(a.data(), a.data() + a.size());
        return true;
    }
    static handle cast(const Vec& v, return_value_policy, handle) {
        py::array_t<double> a((py::ssize_t)v.size());
        std::copy(v.begin(), v.end(), a.mutable_data());
        return a.release();
    }
};
template <>
struct type_caster<Mat> {
    PYBIND11_TYPE_CASTER(Mat, const_name("numpy.ndarray[float64[m, n]]"));
    bool load(handle src, bool) {
        if (!src || src.is_none()) return false;
        arr_t a = arr_t::ensure(src);
        if (!a || a.ndim() > 2) return false;
        if (a.ndim() == 2) value = Mat((int)a.shape(0), (int)a.shape(1));
        else value = Mat((int)a.size(), 1);
        std::copy(a.data(), a.data() + a.size(), value.d.begin());
        return true;
    }
    static handle cast(const Mat& m, return_value_policy, handle) {
        py::array_t<double> a({(py::ssize_t)m.rows, (py::ssize_t)m.cols});
        std::copy(m.d.begin(), m.d.end(), a.mutable_data());
        return a.release();
    }
};
}  // namespace detail
}  // namespace pybind11

#include <pybind11/stl.h>

class PythonCallbackMessage : public CallBackMessage {  // pylqr_planner/src/PythonCallbackMessage.cpp:14-17
public:
    void notify(const std::string& msg) override { py::print(msg); }
};

static py::array_t<double> shaped(const std::vector<double>& v, std::vector<py::ssize_t> shape) {
    py::array_t<double> a(shape);
    std::copy(v.begin(), v.end(), a.mutable_data());
    return a;
}

// numpy -> BatchInputs: U0 [B][T-1][n_u] or [T-1][n_u]; q0/dq0 [B][dof] or None; kp_targets list of [B][n_f] or None
static solver::BatchInputs batch_inputs(const py::object& U0, const py::object& q0, const py::object& dq0, const py::object& kp_targets, bool flat_u = false) {
    solver::BatchInputs in;
    in.B = 1;
    arr_t u = arr_t::ensure(U0);
    if (!u) throw std::runtime_error("[solve_batch] U0 must be a float array");
    in.U0.assign(u.data(), u.data() + u.size());
    if (u.ndim() == 3 || (flat_u && u.ndim() == 2)) in.B = (int)u.shape(0);  // flat_u: [B][(T-1) n_u] (Batch-CP's vectorised u)
    if (!q0.is_none()) {
        arr_t a = arr_t::ensure(q0);
        if (!a || a.ndim() != 2) throw std::runtime_error("[solve_batch] q0 must be B x dof");
        in.q0.assign(a.data(), a.data() + a.size());
        in.B = (int)a.shape(0);
    }
    if (!dq0.is_none()) {
        arr_t a = arr_t::ensure(dq0);
        if (!a || a.ndim() != 2) throw std::runtime_error("[solve_batch] dq0 must be B x dof");
        in.dq0.assign(a.data(), a.data() + a.size());
    }
    if (!kp_targets.is_none()) {
        for (auto h : kp_targets.cast<py::list>()) {
            arr_t a = arr_t::ensure(py::reinterpret_borrow<py::object>(h));
            if (!a || a.ndim() != 2) throw std::runtime_error("[solve_batch] every kp_targets entry must be B x nb_target_var");
            in.kp_targets.emplace_back(a.data(), a.data() + a.size());
            in.B = (int)a.shape(0);
        }
    }
    return in;
}

PYBIND11_MODULE(PyLQR, m) {
    m.doc() = "PyLQR: the reference's Python surface over the MI355X-native batched iLQR hot path (libilqr_hip.so)";

    // ------------------------------------------------------------------ sim
    py::module m_sim = m.def_submodule("sim");
    py::class_<sim::SimulationInterface, std::shared_ptr<sim::SimulationInterface>>(m_sim, "SimulationInterface")
        .def("update_kinematics", &sim::SimulationInterface::updateKinematics)
        .def("Jt", &sim::SimulationInterface::Jt)
        .def("Jr", &sim::SimulationInterface::Jr)
        .def("J", &sim::SimulationInterface::J)
        .def("Jtp", &sim::SimulationInterface::Jtp)
        .def("Jrp", &sim::SimulationInterface::Jrp)
        .def("Jp", &sim::SimulationInterface::Jp)
        .def("get_ee_pos", &sim::SimulationInterface::getEEPosition)
        .def("get_ee_orn", &sim::SimulationInterface::getEEOrnQuat)
        .def("get_ee_vel", &sim::SimulationInterface::getEEVelocity)
        .def("get_ee_ang_vel", &sim::SimulationInterface::getEEAngVel)
        .def("get_ee_ang_vel_quat", &sim::SimulationInterface::getEEAngVelQuat)
        .def("get_q", &sim::SimulationInterface::getJointsPos)
        .def("get_dq", &sim::SimulationInterface::getJointsVel)
        .def("get_time", &sim::SimulationInterface::getTime)
        .def("set_time", &sim::SimulationInterface::setTime, py::arg("time"))
        .def("dquat_to_w_jac", &sim::SimulationInterface::dQuatToDxJac, py::arg("quat"))
        .def("set_conf", &sim::SimulationInterface::setConfiguration, py::arg("q"), py::arg("dq"), py::arg("reset_time"))
        .def("send_acc", &sim::SimulationInterface::sendAcc, py::arg("dt"), py::arg("ddq"), py::arg("updateKin"))
        .def("send_vel", &sim::SimulationInterface::sendVel, py::arg("dt"), py::arg("dq"), py::arg("updateKin"));
    py::class_<sim::KDLRobot, sim::SimulationInterface, std::shared_ptr<sim::KDLRobot>>(m_sim, "KDLRobot")
        .def(py::init<const std::string&, const std::string&, const std::string&, const Vec&, const Vec&, const Vec&, const Vec&, const bool&>(),
             py::arg("urdf"), py::arg("base_frame"), py::arg("tip_frame"), py::arg("q"), py::arg("dq"), py::arg("transform_rpy"), py::arg("transform_xyz"),
             py::arg("is_path"))
        .def(py::init<const std::string&, const std::string&, const std::string&, const Vec&, const Vec&, const Vec&, const Vec&>(), py::arg("urdf"),
             py::arg("base_frame"), py::arg("tip_frame"), py::arg("q"), py::arg("dq"), py::arg("transform_rpy"), py::arg("transform_xyz"))
        .def(py::init<const std::string&, const std::string&, const std::string&, const Vec&, const Vec&>(), py::arg("urdf"), py::arg("base_frame"),
             py::arg("tip_frame"), py::arg("q"), py::arg("dq"))
        .def("joint_lower_limits", &sim::KDLRobot::jointLowerLimits)
        .def("joint_upper_limits", &sim::KDLRobot::jointUpperLimits);
    // bindings.cpp:192-195
    py::class_<sim::Robot2D, sim::SimulationInterface, std::shared_ptr<sim::Robot2D>>(m_sim, "Robot2D")
        .def(py::init<const Vec&, const Vec&>(), py::arg("lengths"), py::arg("default_q"))
        .def("fkine", static_cast<Vec (sim::Robot2D::*)()>(&sim::Robot2D::fkine))
        .def("fkine", static_cast<Vec (sim::Robot2D::*)(const Vec&)>(&sim::Robot2D::fkine), py::arg("q"));
    // bindings.cpp:206-207: TransformedSimulationInterface(r, T)
    py::class_<sim::TransformedSimulationInterface, sim::SimulationInterface, std::shared_ptr<sim::TransformedSimulationInterface>>(m_sim, "TransformedSimulationInterface")
        .def(py::init<const std::shared_ptr<sim::SimulationInterface>&, const Mat&>(), py::arg("r"), py::arg("T"));

    // ------------------------------------------------------------------ system
    py::module m_sys = m.def_submodule("system");
    py::class_<sys::Keypoint, std::shared_ptr<sys::Keypoint>>(m_sys, "Keypoint")
        .def("diff", &sys::Keypoint::diff, py::arg("state"))
        .def("get_state", &sys::Keypoint::getState)
        .def("get_precision", &sys::Keypoint::getPrecision)
        .def("get_timestep", &sys::Keypoint::getTimestep);
    py::class_<sys::PosOrnKeypoint, sys::Keypoint, std::shared_ptr<sys::PosOrnKeypoint>>(m_sys, "PosOrnKeypoint")
        .def(py::init<const Vec&, const Vec&, const Mat&, const int&>(), py::arg("position"), py::arg("orientation"), py::arg("precision"), py::arg("timestep"))
        .def(py::init<const Vec&, const Vec&, const Vec&, const Vec&, const Mat&, const int&>(), py::arg("position"), py::arg("dposition"), py::arg("orientation"),
             py::arg("dorientation"), py::arg("precision"), py::arg("timestep"))
        .def("get_position", &sys::PosOrnKeypoint::getPosition)
        .def("get_orientation", &sys::PosOrnKeypoint::getOrientation);
    // bindings.cpp:302-306 (argument names pos_thresh / orn_thresh as there)
    py::class_<sys::PosOrnKeypointDistFunct, sys::PosOrnKeypoint, std::shared_ptr<sys::PosOrnKeypointDistFunct>>(m_sys, "PosOrnKeypointDistFunct")
        .def(py::init<const Vec&, const Vec&, const Mat&, const double&, const Vec&, const int&>(), py::arg("position"), py::arg("orientation"), py::arg("precision"),
             py::arg("pos_thresh"), py::arg("orn_thresh"), py::arg("timestep"))
        .def(py::init<const Vec&, const Vec&, const Vec&, const Vec&, const Mat&, const double&, const Vec&, const int&>(), py::arg("position"), py::arg("dposition"),
             py::arg("orientation"), py::arg("dorientation"), py::arg("precision"), py::arg("pos_thresh"), py::arg("orn_thresh"), py::arg("timestep"));
    // bindings.cpp:368-371
    py::class_<sys::AngularKeypoint, sys::Keypoint, std::shared_ptr<sys::AngularKeypoint>>(m_sys, "AngularKeypoint")
        .def(py::init<const Vec&, const Mat&, const int&>(), py::arg("position"), py::arg("precision"), py::arg("timestep"))
        .def(py::init<const Vec&, const Vec&, const Mat&, const int&>(), py::arg("position"), py::arg("dposition"), py::arg("precision"), py::arg("timestep"))
        .def("get_position", &sys::AngularKeypoint::getPosition);
    // bindings.cpp:388-392
    py::class_<sys::AngularTimeKeypoint, sys::AngularKeypoint, std::shared_ptr<sys::AngularTimeKeypoint>>(m_sys, "AngularTimeKeypoint")
        .def(py::init<const Vec&, const Mat&, const double&, const int&>(), py::arg("position"), py::arg("precision"), py::arg("continuous_time"), py::arg("timestep"))
        .def(py::init<const Vec&, const Vec&, const Mat&, const double&, const int&>(), py::arg("position"), py::arg("dposition"), py::arg("precision"),
             py::arg("continuous_time"), py::arg("timestep"))
        .def("get_continuous_time", &sys::AngularTimeKeypoint::getContinuousTime);
    py::class_<sys::SpacetimeKeypoint, sys::PosOrnKeypoint, std::shared_ptr<sys::SpacetimeKeypoint>>(m_sys, "SpacetimeKeypoint")
        .def(py::init<const Vec&, const Vec&, const Mat&, const double&, const int&>(), py::arg("position"), py::arg("orientation"), py::arg("precision"),
             py::arg("continuous_time"), py::arg("timestep"))
        .def(py::init<const Vec&, const Vec&, const Vec&, const Vec&, const Mat&, const double&, const int&>(), py::arg("position"), py::arg("dposition"),
             py::arg("orientation"), py::arg("dorientation"), py::arg("precision"), py::arg("continuous_time"), py::arg("timestep"))
        .def("get_continuous_time", &sys::SpacetimeKeypoint::getContinuousTime);
    py::class_<sys::System, std::shared_ptr<sys::System>>(m_sys, "System")
        .def("get_mu_vector", &sys::System::getMuVector, py::arg("sparse") = false)
        .def("get_Q_matrix", &sys::System::getQMatrix, py::arg("sparse") = false)
        // single-point evaluation API (bindings.cpp:414-497); host glue, the solvers run the batched device path instead
        .def("forward_pass", &sys::System::forwardPass, py::arg("xk"), py::arg("uk"), py::arg("k"))
        .def("diff", &sys::System::diff, py::arg("actual_state"), py::arg("k"))
        .def("diff_batch", &sys::System::diffBatch, py::arg("x"))
        .def("forward_pass_with_limits", &sys::System::forwardPassWithLimits, py::arg("xk"), py::arg("uk"), py::arg("k"))
        .def("forward_pass_batch", &sys::System::fpBatch, py::arg("u"))
        .def("cost_F", &sys::System::cost_F, py::arg("xk"))
        .def("cost_F_x", &sys::System::cost_F_x, py::arg("xk"))
        .def("cost_F_xx", &sys::System::cost_F_xx, py::arg("xk"))
        .def("cost", &sys::System::cost, py::arg("xk"), py::arg("uk"), py::arg("k"))
        .def("cost_x", &sys::System::cost_x, py::arg("xk"), py::arg("uk"), py::arg("k"))
        .def("cost_xx", &sys::System::cost_xx, py::arg("xk"), py::arg("uk"), py::arg("k"))
        .def("cost_u", &sys::System::cost_u, py::arg("xk"), py::arg("uk"), py::arg("k"))
        .def("cost_uu", &sys::System::cost_uu, py::arg("xk"), py::arg("uk"), py::arg("k"))
        .def("cost_xu", &sys::System::cost_xu, py::arg("xk"), py::arg("uk"), py::arg("k"))
        .def("cost_ux", &sys::System::cost_ux, py::arg("xk"), py::arg("uk"), py::arg("k"))
        .def("get_fx_jac", static_cast<std::tuple<Vec, Mat> (sys::System::*)()>(&sys::System::getFxJac))
        .def("get_fx_jac", static_cast<std::tuple<Vec, Mat> (sys::System::*)(const Vec&)>(&sys::System::getFxJac), py::arg("xk"))
        .def("get_nb_state_var", &sys::System::getNbStateVar)
        .def("get_nb_ctrl_var", &sys::System::getNbCtrlVar)
        .def("get_nb_target_var", &sys::System::getNbTargetVar)
        .def("get_horizon", &sys::System::getHorizon)
        .def("get_state", &sys::System::getState)
        .def("get_init_state", &sys::System::getInitState)
        .def("get_init_fx_state", &sys::System::getInitFoXState)
        .def("reset", &sys::System::reset);
    using KPs = std::vector<std::shared_ptr<sys::Keypoint>>;
    using SimP = std::shared_ptr<sim::SimulationInterface>;
    py::class_<sys::PosOrnPlannerSys, sys::System, std::shared_ptr<sys::PosOrnPlannerSys>>(m_sys, "PosOrnPlannerSys")
        .def(py::init<const SimP&, const KPs&, const Vec&, const Vec&, const Vec&, const Vec&, const Vec&, int, int, double>(), py::arg("r"), py::arg("keypoints"),
             py::arg("RtDiag"), py::arg("qMax"), py::arg("qMin"), py::arg("dqMax"), py::arg("dqMin"), py::arg("horizon"), py::arg("nbDeriv"), py::arg("dt"))
        .def(py::init<const SimP&, const KPs&, const Vec&, const Vec&, const Vec&, int, int, double>(), py::arg("r"), py::arg("keypoints"), py::arg("RtDiag"),
             py::arg("qMax"), py::arg("qMin"), py::arg("horizon"), py::arg("nbDeriv"), py::arg("dt"))
        .def(py::init<const SimP&, const KPs&, const Vec&, int, int, double>(), py::arg("r"), py::arg("keypoints"), py::arg("RtDiag"), py::arg("horizon"),
             py::arg("nbDeriv"), py::arg("dt"));
    // bindings.cpp:528-536
    py::class_<sys::JointSpacePlannerSys, sys::System, std::shared_ptr<sys::JointSpacePlannerSys>>(m_sys, "JointSpacePlannerSys")
        .def(py::init<const SimP&, const KPs&, const Vec&, const Vec&, const Vec&, const Vec&, const Vec&, int, int, double>(), py::arg("r"), py::arg("keypoints"),
             py::arg("RtDiag"), py::arg("qMax"), py::arg("qMin"), py::arg("dqMax"), py::arg("dqMin"), py::arg("horizon"), py::arg("nbDeriv"), py::arg("dt"))
        .def(py::init<const SimP&, const KPs&, const Vec&, const Vec&, const Vec&, int, int, double>(), py::arg("r"), py::arg("keypoints"), py::arg("RtDiag"),
             py::arg("qMax"), py::arg("qMin"), py::arg("horizon"), py::arg("nbDeriv"), py::arg("dt"))
        .def(py::init<const SimP&, const KPs&, const Vec&, int, int, double>(), py::arg("r"), py::arg("keypoints"), py::arg("RtDiag"), py::arg("horizon"), py::arg("nbDeriv"),
             py::arg("dt"));
    // bindings.cpp:571-579
    py::class_<sys::JointSpaceTimePlannerSys, sys::System, std::shared_ptr<sys::JointSpaceTimePlannerSys>>(m_sys, "JointSpaceTimePlannerSys")
        .def(py::init<const SimP&, const KPs&, const Vec&, const Vec&, const Vec&, const Vec&, const Vec&, int, int>(), py::arg("r"), py::arg("keypoints"),
             py::arg("RtDiag"), py::arg("qMax"), py::arg("qMin"), py::arg("dqMax"), py::arg("dqMin"), py::arg("horizon"), py::arg("nbDeriv"))
        .def(py::init<const SimP&, const KPs&, const Vec&, const Vec&, const Vec&, int, int>(), py::arg("r"), py::arg("keypoints"), py::arg("RtDiag"),
             py::arg("qMax"), py::arg("qMin"), py::arg("horizon"), py::arg("nbDeriv"))
        .def(py::init<const SimP&, const KPs&, const Vec&, int, int>(), py::arg("r"), py::arg("keypoints"), py::arg("RtDiag"), py::arg("horizon"), py::arg("nbDeriv"));
    py::class_<sys::PosOrnTimePlannerSys, sys::System, std::shared_ptr<sys::PosOrnTimePlannerSys>>(m_sys, "PosOrnTimePlannerSys")
        .def(py::init<const SimP&, const KPs&, const Vec&, const Vec&, const Vec&, const Vec&, const Vec&, int, int>(), py::arg("r"), py::arg("keypoints"),
             py::arg("RtDiag"), py::arg("qMax"), py::arg("qMin"), py::arg("dqMax"), py::arg("dqMin"), py::arg("horizon"), py::arg("nbDeriv"))
        .def(py::init<const SimP&, const KPs&, const Vec&, const Vec&, const Vec&, int, int>(), py::arg("r"), py::arg("keypoints"), py::arg("RtDiag"),
             py::arg("qMax"), py::arg("qMin"), py::arg("horizon"), py::arg("nbDeriv"))
        .def(py::init<const SimP&, const KPs&, const Vec&, int, int>(), py::arg("r"), py::arg("keypoints"), py::arg("RtDiag"), py::arg("horizon"), py::arg("nbDeriv"));
    // bindings.cpp:505-507: SequentialSystem(r, systems, RtDiag, horizon, nbDeriv)
    py::class_<sys::SequentialSystem, sys::System, std::shared_ptr<sys::SequentialSystem>>(m_sys, "SequentialSystem")
        .def(py::init<const SimP&, const std::vector<std::shared_ptr<sys::System>>&, const Vec&, int, int>(), py::arg("r"), py::arg("systems"), py::arg("RtDiag"),
             py::arg("horizon"), py::arg("nbDeriv"));

    // ------------------------------------------------------------------ utils (before solver: CallBackMessage is an argument type)
    py::module m_ut = m.def_submodule("utils");
    py::class_<CallBackMessage>(m_ut, "CallBackMessage");
    py::class_<PythonCallbackMessage, CallBackMessage>(m_ut, "PythonCallbackMessage").def(py::init<>());
    py::module m_sd = m_ut.def_submodule("Sd");
    m_sd.def("logMap", &Sd::logMap, py::arg("base"), py::arg("y"));
    m_sd.def("expMap", &Sd::expMap, py::arg("base"), py::arg("u"));
    m_sd.def("distance", &Sd::distance, py::arg("x"), py::arg("y"));
    m_sd.def("transport", &Sd::transport, py::arg("v"), py::arg("base1"), py::arg("base2"));
    m_sd.def("dquat_to_w_jac", &Sd::dQuatToDxJac, py::arg("q"));
    py::module m_prim = m_ut.def_submodule("primitives");
    m_prim.def("build_psi_RBF", &buildPsiRBF, py::arg("dim"), py::arg("K"));
    m_prim.def("build_psi_bernstein", &buildPsiBernstein, py::arg("dim"), py::arg("K"));
    m_prim.def("build_psi_unitstep", &buildPsiUnitstep, py::arg("dim"), py::arg("K"));
    m_prim.def("build_psi_sawtooth", &buildPsiSawtooth, py::arg("dim"), py::arg("K"));
    m_prim.def("build_psi_linear", &buildPsiLinear, py::arg("dim"), py::arg("K"));

    // ------------------------------------------------------------------ solver
    py::module m_sol = m.def_submodule("solver");
    py::class_<solver::BatchResult>(m_sol, "BatchResult")
        .def_property_readonly("X", [](const solver::BatchResult& r) { return shaped(r.X, {r.B, r.T, r.n_x}); })
        .def_property_readonly("fX", [](const solver::BatchResult& r) { return shaped(r.fX, {r.fX.empty() ? 0 : r.B, r.T, r.n_f}); })
        .def_property_readonly("U", [](const solver::BatchResult& r) { return shaped(r.U, {r.B, r.T - 1, r.n_u}); })
        .def_property_readonly("K", [](const solver::BatchResult& r) { return shaped(r.K, {r.K.empty() ? 0 : r.B, r.T - 1, r.n_u, r.n_x}); })
        .def_property_readonly("d", [](const solver::BatchResult& r) { return shaped(r.d, {r.d.empty() ? 0 : r.B, r.T - 1, r.n_u}); })
        .def_property_readonly("cost", [](const solver::BatchResult& r) { return shaped(r.cost, {r.B}); })
        .def_property_readonly("alpha", [](const solver::BatchResult& r) { return shaped(r.alpha, {r.B}); })
        .def_property_readonly("iters", [](const solver::BatchResult& r) { return py::array_t<int>(r.iters.size(), r.iters.data()); })
        .def_property_readonly("status", [](const solver::BatchResult& r) { return py::array_t<int>(r.status.size(), r.status.data()); })
        .def_property_readonly("cost_trace", [](const solver::BatchResult& r) { return shaped(r.cost_trace, {r.cost_trace.empty() ? 0 : r.B, r.nb_iter}); })
        .def_property_readonly("alpha_trace", [](const solver::BatchResult& r) { return shaped(r.alpha_trace, {r.alpha_trace.empty() ? 0 : r.B, r.nb_iter}); })
        .def_readonly("seconds", &solver::BatchResult::seconds);
    py::class_<solver::Constraint>(m_sol, "Constraint").def(py::init<>()).def_readwrite("A", &solver::Constraint::A).def_readwrite("b", &solver::Constraint::b);
    py::class_<solver::ILQRRecursive>(m_sol, "ILQRRecursive")
        .def(py::init<const std::shared_ptr<sys::System>&>(), py::arg("s"))
        .def("solve", &solver::ILQRRecursive::solve, py::arg("U0"), py::arg("nb_iter"), py::arg("line_search"), py::arg("early_stop"), py::arg("cb"))
        .def("solve_batch",
             [](solver::ILQRRecursive& self, const py::object& U0, int nb_iter, bool ls, bool es, const py::object& q0, const py::object& dq0, const py::object& kp) {
                 return self.solveBatch(batch_inputs(U0, q0, dq0, kp), nb_iter, ls, es);
             },
             py::arg("U0"), py::arg("nb_iter"), py::arg("line_search"), py::arg("early_stop"), py::arg("q0") = py::none(), py::arg("dq0") = py::none(),
             py::arg("kp_targets") = py::none());
    py::class_<solver::AL_ILQR>(m_sol, "AL_ILQR")
        .def(py::init<const std::shared_ptr<sys::System>&, const std::vector<solver::Constraint>&, const std::vector<Vec>&>(), py::arg("s"), py::arg("inequality"),
             py::arg("initLambda"))
        .def("solve", &solver::AL_ILQR::solve, py::arg("U0"), py::arg("nb_iter"), py::arg("lag_update_step"), py::arg("penalty"), py::arg("scaling_factor"),
             py::arg("line_search"), py::arg("early_stop"), py::arg("cb"))
        .def("solve_batch",
             [](solver::AL_ILQR& self, const py::object& U0, int nb_iter, int lag, double pen, double sc, bool ls, bool es, const py::object& q0, const py::object& dq0,
                const py::object& kp) { return self.solveBatch(batch_inputs(U0, q0, dq0, kp), nb_iter, lag, pen, sc, ls, es); },
             py::arg("U0"), py::arg("nb_iter"), py::arg("lag_update_step"), py::arg("penalty"), py::arg("scaling_factor"), py::arg("line_search"),
             py::arg("early_stop"), py::arg("q0") = py::none(), py::arg("dq0") = py::none(), py::arg("kp_targets") = py::none());
    // bindings.cpp:778-782
    py::class_<solver::BatchILQR>(m_sol, "BatchILQR")
        .def(py::init<const std::shared_ptr<sys::System>&, const Mat&>(), py::arg("s"), py::arg("Q"))
        .def(py::init<const std::shared_ptr<sys::System>&>(), py::arg("s"))
        .def("solve", &solver::BatchILQR::solve, py::arg("nb_iter"), py::arg("u0"), py::arg("early_stop"), py::arg("cb"))
        .def("solve_batch",
             [](solver::BatchILQR& self, int nb_iter, const py::object& u0, bool es, const py::object& q0, const py::object& dq0, const py::object& kp) {
                 return self.solveBatch(batch_inputs(u0, q0, dq0, kp, true), nb_iter, es);
             },
             py::arg("nb_iter"), py::arg("u0"), py::arg("early_stop"), py::arg("q0") = py::none(), py::arg("dq0") = py::none(), py::arg("kp_targets") = py::none());
    py::class_<solver::BatchILQRCP>(m_sol, "BatchILQRCP")
        .def(py::init<const std::shared_ptr<sys::System>&, const Mat&, const Mat&>(), py::arg("s"), py::arg("Q"), py::arg("psi"))
        .def(py::init<const std::shared_ptr<sys::System>&, const Mat&>(), py::arg("s"), py::arg("psi"))
        .def("solve", &solver::BatchILQRCP::solve, py::arg("nb_iter"), py::arg("u0"), py::arg("early_stop"), py::arg("cb"))
        .def("solve_batch",
             [](solver::BatchILQRCP& self, int nb_iter, const py::object& u0, bool es, const py::object& q0, const py::object& dq0, const py::object& kp) {
                 return self.solveBatch(batch_inputs(u0, q0, dq0, kp, true), nb_iter, es);
             },
             py::arg("nb_iter"), py::arg("u0"), py::arg("early_stop"), py::arg("q0") = py::none(), py::arg("dq0") = py::none(), py::arg("kp_targets") = py::none());
}
