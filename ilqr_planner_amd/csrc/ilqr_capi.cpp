// ilqr_capi.cpp -- implementation of the C ABI declared in include/ilqr_hip.h (compiled with hipcc).
//
// Host-side orchestration only: lowering of the POD problem description to the device descriptor, buffer
// ownership, layout conversion at the boundary and the per-iteration launch sequence
//   init rollout -> nb_iter x { backward sweep, forward line search }
// which mirrors ILQRRecursive::solve / AL_ILQR::solve (reference src/solver/ILQRRecursive.cpp:21-181,
// src/solver/AL-ILQR.cpp:50-232).  No CPU fallback exists: every entry point fails loudly if HIP does.
#include "../../include/ilqr_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ilqr_batchcp.hpp"
#include "ilqr_kernels.hpp"

using namespace ilqr;

struct ilqr_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    bool profile = false;
    int split = 1;  // ilqr_ctx_set_split: 0 off, 1 where it was measured to pay, 2 every cooperative path (experiments)
    double prof_ms[ILQR_PROF_COUNT] = {0, 0, 0, 0, 0};
    int prof_n[ILQR_PROF_COUNT] = {0, 0, 0, 0, 0};
    struct Pending { hipEvent_t a, b; int which; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    std::vector<ilqr_problem*> problems;  // live problems of this context (destroyed with it)
    // split solves (solve_riccati): the two halves of a batch run on their own streams, joined to `stream` by events
    int n_simd = 1024;  // SIMDs of the device (4 per CU)
    bool xc_generic = false, xc_cp_lane = false, xc_cp_general = false;  // cross-check kernel variants (ilqr_ctx_set_crosscheck)
    int xc_sweep = 0;  // sweep of the 2nd-order / time systems: 0 = by batch size, 1 = matrix-core sweep, 2 = row sweep
    int xc_apply = 0;  // re-roll of the winner on the time systems: 0 = by batch size, 1 = k_apply_rows_tm, 2 = k_apply_dpp_tm
    int xc_fwd = 0;    // forward pass of the single-integrator systems: 0 = by batch size, 1 = k_forward_wg (bandwidth), 2 = k_forward_dpp (latency)
    hipStream_t half_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_begin = nullptr, ev_half_done[2] = {nullptr, nullptr}, ev_stagger = nullptr;
};

struct ilqr_problem {
    ilqr_ctx* ctx = nullptr;
    ilqr_problem_desc desc;
    ilqr_dims dims;
    int B = 0, Bp = 0, T = 0;
    DevDesc hdesc;
    DevDesc* ddesc = nullptr;
    DevDesc* ddesc_half[2] = {nullptr, nullptr};  // the descriptor with B = the half's instance count (split solves)
    int half_b0[2] = {0, 0}, half_B[2] = {0, 0};
    Bufs bufs;
    std::vector<void*> allocs;
    double *conA = nullptr, *conb = nullptr;  // device copies of the shared constraint rows
    double* lambda0 = nullptr;                // initial multipliers, kept for ilqr_problem_reset_multipliers
    bool con_state_only = false;              // no constraint row touches the controls (enables the closed-form sweep)
    int trace_iters = 0;
    double* staging = nullptr;  // device staging for host<->device natural-layout transfers
    size_t staging_elems = 0;
    int last_nb_iter = 0;
    bool has_controls = false, has_state = false;
    bool u0_zero = false;  // the initial controls given from the host are all zero (lets the wide-basis batch solver skip their projection)
    BatchCPState cp;
    BatchWideState cpw;
};

// Batches of at least this many instances may be solved as two halves on two streams (solve_riccati): every kernel of an iteration is a
// chain of T dependent steps that leaves much of the machine idle at these batch sizes, so one half's sweep can run under the other
// half's forward pass and decision.
constexpr int SPLIT_MIN_BATCH = 2048;

static int fail(ilqr_ctx* c, const std::string& m) {
    if (c) c->err = m;
    return 1;
}
#define HIPCHK(ctx, call)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return fail((ctx), std::string(#call) + ": " + hipGetErrorString(e_));       \
    } while (0)

// ------------------------------------------------------------------------------------------------ misc

extern "C" const char* ilqr_version(void) { return "ilqr_hip 0.1 (gfx950, fp64)"; }

extern "C" void ilqr_desc_defaults(ilqr_problem_desc* d) {
    std::memset(d, 0, sizeof(*d));
    d->reg = 1e-6;
    d->alpha_floor = 1e-3;
    d->stop_tol = 1e-3;
    d->nb_deriv = 1;
    d->dof = 7;
}

extern "C" int ilqr_dims_of(const ilqr_problem_desc* d, ilqr_dims* o) {
    if (!d || !o) return 1;
    if (d->kind != ILQR_SYS_POS_ORN && d->kind != ILQR_SYS_POS_ORN_TIME && d->kind != ILQR_SYS_JOINT && d->kind != ILQR_SYS_JOINT_TIME) return 1;
    if (d->nb_deriv != 1 && d->nb_deriv != 2) return 1;
    if (d->kind == ILQR_SYS_JOINT || d->kind == ILQR_SYS_JOINT_TIME) {  // JointSpace(Time)PlannerSys localInit; 2nd order inconsistent upstream
        if (d->nb_deriv != 1) return 1;
        o->n_x = o->n_u = o->n_f = o->n_Q = d->dof + (d->kind == ILQR_SYS_JOINT_TIME ? 1 : 0);
        return 0;
    }
    const int tm = d->kind == ILQR_SYS_POS_ORN_TIME ? 1 : 0;
    o->n_x = d->nb_deriv * d->dof + tm;  // PosOrnPlannerSys.cpp:74 / PosOrnTimePlannerSys.cpp:67
    o->n_u = d->dof + tm;
    o->n_f = 7 * d->nb_deriv + tm;
    o->n_Q = o->n_f - d->nb_deriv;
    return 0;
}

extern "C" int ilqr_ctx_create(int device_id, ilqr_ctx** out) {
    if (!out) return 1;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 2;  // no HIP device: fail loudly, there is no CPU path
    if (device_id < 0 || device_id >= n) return 3;
    if (hipSetDevice(device_id) != hipSuccess) return 4;
    auto* c = new ilqr_ctx();
    c->device = device_id;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return 5;
    }
    c->stream = c->own_stream;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) c->n_simd = 4 * cus; }
    *out = c;
    return 0;
}

extern "C" void ilqr_ctx_destroy(ilqr_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    while (!c->problems.empty()) ilqr_problem_destroy(c->problems.back());  // handles held by the caller become invalid
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->pending) (void)hipEventDestroy(p.a);
    for (auto e : c->pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; i++) {
        if (c->half_stream[i]) (void)hipStreamSynchronize(c->half_stream[i]);
        if (c->half_stream[i]) (void)hipStreamDestroy(c->half_stream[i]);
        if (c->ev_half_done[i]) (void)hipEventDestroy(c->ev_half_done[i]);
    }
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_stagger) (void)hipEventDestroy(c->ev_stagger);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" const char* ilqr_last_error(const ilqr_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" int ilqr_ctx_set_stream(ilqr_ctx* c, void* s) {
    if (!c) return 1;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return 0;
}

extern "C" int ilqr_ctx_set_split(ilqr_ctx* c, int on) {
    if (!c) return 1;
    c->split = on;
    return 0;
}

extern "C" int ilqr_ctx_set_crosscheck(ilqr_ctx* c, int generic_kernels, int cp_lane_solve, int cp_general, int mfma_sweep) {
    if (!c) return 1;
    c->xc_sweep = ((mfma_sweep & 3) == 1 || (mfma_sweep & 3) == 2) ? (mfma_sweep & 3) : 0;
    c->xc_fwd = (((mfma_sweep >> 2) & 3) == 1 || ((mfma_sweep >> 2) & 3) == 2) ? ((mfma_sweep >> 2) & 3) : 0;
    c->xc_apply = (mfma_sweep >> 4) & 3;
    c->xc_generic = generic_kernels != 0;
    c->xc_cp_lane = cp_lane_solve != 0;
    c->xc_cp_general = cp_general != 0;
    return 0;
}

extern "C" int ilqr_ctx_synchronize(ilqr_ctx* c) {
    if (!c) return 1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------ profiling

static hipEvent_t ev_get(ilqr_ctx* c) {
    if (!c->pool.empty()) {
        hipEvent_t e = c->pool.back();
        c->pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
// One event is recorded in FRONT of every kernel launch (and one behind the last launch of a solve): the interval between two
// consecutive marks is charged to the kernel that the first one precedes.  Half the events of a start/stop pair per kernel.
static void prof_collect(ilqr_ctx* c) {
    if (c->pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (size_t i = 0; i + 1 < c->pending.size(); i++) {
        const auto& p = c->pending[i];
        float ms = 0;
        if (p.which >= 0 && hipEventElapsedTime(&ms, p.a, c->pending[i + 1].a) == hipSuccess) {
            c->prof_ms[p.which] += ms;
            c->prof_n[p.which] += 1;
        }
    }
    for (auto& p : c->pending) c->pool.push_back(p.a);
    c->pending.clear();
}
static void prof_mark(ilqr_ctx* c, int which) {  // which < 0: end mark (closes the previous interval, charges nothing itself)
    if (!c->profile) return;
    hipEvent_t e = ev_get(c);
    (void)hipEventRecord(e, c->stream);
    c->pending.push_back({e, nullptr, which});
    if (which < 0 && c->pending.size() > 4096) prof_collect(c);  // synchronises; only with profiling on and only every few hundred solves
}
struct ProfScope {  // marks the launch that follows; the interval is closed by the next mark
    ProfScope(ilqr_ctx* c, int w) { prof_mark(c, w); }
};

static void prof_hook_fn(void* c, int which) { prof_mark((ilqr_ctx*)c, which); }
static ilqr::ProfHook prof_hook(ilqr_ctx* c) {
    ilqr::ProfHook h;
    if (c->profile) { h.mark = prof_hook_fn; h.ctx = c; }
    return h;
}

extern "C" int ilqr_profile_enable(ilqr_ctx* c, int on) {
    if (!c) return 1;
    prof_collect(c);
    c->profile = on != 0;
    return 0;
}
extern "C" int ilqr_profile_reset(ilqr_ctx* c) {
    if (!c) return 1;
    prof_collect(c);
    for (int i = 0; i < ILQR_PROF_COUNT; i++) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
    return 0;
}
extern "C" int ilqr_profile_get(ilqr_ctx* c, int which, double* ms, int* n) {
    if (!c || which < 0 || which >= ILQR_PROF_COUNT) return 1;
    prof_collect(c);
    if (ms) *ms = c->prof_ms[which];
    if (n) *n = c->prof_n[which];
    return 0;
}

// ------------------------------------------------------------------------------------------------ lowering

static void mat3(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// Fold runs of fixed segments into the following joint's pre-transform; the rest becomes the tail.
static int lower_chain(ilqr_ctx* c, const ilqr_problem_desc& d, DevChain& ch) {
    if (d.dof != DOF) return fail(c, "device path supports chains with exactly 7 moving joints (got " + std::to_string(d.dof) + ")");
    if (d.n_seg < 1 || d.n_seg > ILQR_MAX_SEG) return fail(c, "bad n_seg");
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
    int nj = 0;
    for (int s = 0; s < d.n_seg; s++) {
        double Rn[9];
        for (int i = 0; i < 3; i++) p[i] += R[3 * i] * d.seg_xyz[s][0] + R[3 * i + 1] * d.seg_xyz[s][1] + R[3 * i + 2] * d.seg_xyz[s][2];
        mat3(R, d.seg_R[s], Rn);
        std::memcpy(R, Rn, sizeof(R));
        const int j = d.seg_joint[s];
        if (j >= 0) {
            if (j != nj || nj >= DOF) return fail(c, "moving joints must be numbered 0..dof-1 in chain order");
            std::memcpy(ch.Rpre[nj], R, sizeof(R));
            std::memcpy(ch.ppre[nj], p, sizeof(p));
            std::memcpy(ch.axis[nj], d.seg_axis[s], 3 * sizeof(double));
            nj++;
            const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            std::memcpy(R, I, sizeof(R));
            p[0] = p[1] = p[2] = 0;
        }
    }
    if (nj != DOF) return fail(c, "chain has " + std::to_string(nj) + " moving joints, descriptor says " + std::to_string(d.dof));
    std::memcpy(ch.Rtail, R, sizeof(R));
    std::memcpy(ch.ptail, p, sizeof(p));
    return 0;
}

static int lower_desc(ilqr_ctx* c, const ilqr_problem_desc& d, int B, int Bp, DevDesc& h) {
    ilqr_dims dm;
    if (ilqr_dims_of(&d, &dm)) return fail(c, "unsupported system kind / nb_deriv");
    std::memset(&h, 0, sizeof(h));
    if (!((d.kind == ILQR_SYS_JOINT || d.kind == ILQR_SYS_JOINT_TIME) && d.n_seg == 0) && lower_chain(c, d, h.chain)) return 1;  // joint-space systems need no chain
    if (d.horizon < 2) return fail(c, "horizon must be >= 2");
    h.kind = d.kind; h.nd = d.nb_deriv; h.T = d.horizon; h.B = B; h.Bp = Bp; h.dt = d.dt;
    for (int i = 0; i < dm.n_u; i++) h.R_diag[i] = d.R_diag[i];
    {   // SequentialSystem: every sub-system adds the limit terms once -- q'Lq and L'q scale with the multiplicity, L'L too
        const double mult = d.limit_multiplicity > 1 ? (double)d.limit_multiplicity : 1.0;
        h.limits_set = d.limits_set;
        h.penalty = d.penalty * mult;
        h.pen_xx = d.penalty * d.penalty * mult;
        h.batch_limits = (d.limit_multiplicity > 1 || d.is_sequence) ? 0 : 1;
        h.lim2 = d.limits2_set ? 1 : 0;
        if (d.limits2_set) {
            const double mult2 = d.limit_multiplicity2 > 1 ? (double)d.limit_multiplicity2 : 1.0;
            h.penalty2 = d.penalty2 * mult2;
            h.pen_xx2 = d.penalty2 * d.penalty2 * mult2;
            for (int i = 0; i < dm.n_x; i++) { h.smax2[i] = d.state_max2[i]; h.smin2[i] = d.state_min2[i]; h.lw2[i] = d.limit_weight2[i]; }
        }
    }
    for (int i = 0; i < dm.n_x; i++) { h.smax[i] = d.state_max[i]; h.smin[i] = d.state_min[i]; h.lw[i] = d.limit_weight[i]; }
    if (d.n_kp < 0 || d.n_kp > ILQR_MAX_KP) return fail(c, "bad n_kp");
    h.n_kp = d.n_kp;
    for (int k = 0; k < d.n_kp; k++) {
        if (d.kp_timestep[k] < 0 || d.kp_timestep[k] >= d.horizon) return fail(c, "keypoint timestep outside the horizon");
        if (k > 0 && d.kp_timestep[k] <= d.kp_timestep[k - 1]) return fail(c, "keypoint timesteps must be unique and ascending");
        h.kp_t[k] = d.kp_timestep[k];
        if (d.kp_joint[k] && !((d.kind == ILQR_SYS_POS_ORN || d.kind == ILQR_SYS_POS_ORN_TIME) && d.nb_deriv == 1))
            return fail(c, "kp_joint is for PosOrn / PosOrnTime systems with nb_deriv = 1 (a joint-space system needs no flag)");
        if (d.kp_joint[k] && (d.kp_dist[k] || d.kp_has_frame[k])) return fail(c, "a joint-space keypoint has no dead zone and no object frame");
        h.kp_joint[k] = d.kp_joint[k];
        const int nqk = d.kp_joint[k] ? dm.n_x : dm.n_Q;
        for (int i = 0; i < nqk * nqk; i++) h.kp_Q[k][i] = d.kp_Q[k][i];
        h.kp_dist[k] = d.kp_dist[k];
        h.kp_frame[k] = d.kp_has_frame[k];
        for (int i = 0; i < 9; i++) h.kp_fR[k][i] = d.kp_frame_R[k][i];
        for (int i = 0; i < 3; i++) h.kp_fp[k][i] = d.kp_frame_p[k][i];
        h.kp_has_Ru[k] = d.kp_has_Ru[k];
        for (int i = 0; i < dm.n_u; i++) h.kp_Ru[k][i] = d.kp_Ru[k][i];
        h.kp_pos_radius[k] = d.kp_pos_radius[k];
        for (int i = 0; i < 3; i++) h.kp_orn_thresh[k][i] = d.kp_orn_thresh[k][i];
    }
    h.reg = d.reg; h.alpha_floor = d.alpha_floor; h.stop_tol = d.stop_tol;
    return 0;
}

template <class T>
static int dalloc(ilqr_problem* p, T** ptr, size_t n, bool zero = true) {
    void* q = nullptr;
    if (n == 0) n = 1;
    HIPCHK(p->ctx, hipMalloc(&q, n * sizeof(T)));
    p->allocs.push_back(q);
    if (zero) HIPCHK(p->ctx, hipMemsetAsync(q, 0, n * sizeof(T), p->ctx->stream));
    *ptr = (T*)q;
    return 0;
}

static int ensure_staging(ilqr_problem* p, size_t elems) {
    if (p->staging_elems >= elems) return 0;
    if (p->staging) { HIPCHK(p->ctx, hipStreamSynchronize(p->ctx->stream)); HIPCHK(p->ctx, hipFree(p->staging)); p->staging = nullptr; }
    HIPCHK(p->ctx, hipMalloc((void**)&p->staging, elems * sizeof(double)));
    p->staging_elems = elems;
    return 0;
}

extern "C" int ilqr_problem_create(ilqr_ctx* c, const ilqr_problem_desc* d, int batch, ilqr_problem** out) {
    if (!c) return 1;
    if (!d || !out || batch <= 0) return fail(c, "bad arguments");
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    auto* p = new ilqr_problem();
    p->ctx = c;
    p->desc = *d;
    p->B = batch;
    p->Bp = (batch + 63) / 64 * 64;
    // Row stride of every [row][Bp] buffer is Bp*8 bytes.  A power-of-two stride (Bp = 4096 -> 32 KiB) maps the 49 rows
    // of a K block onto the same L2 sets / HBM channel; one extra 64-instance pad column breaks the alignment.
    if (p->Bp % 512 == 0) p->Bp += 64;
    p->T = d->horizon;
    if (ilqr_dims_of(d, &p->dims) || lower_desc(c, *d, p->B, p->Bp, p->hdesc)) { delete p; return 1; }
    const int T = p->T, NX = p->dims.n_x, NU = p->dims.n_u, NF = p->dims.n_f, Bp = p->Bp;
    std::memset(&p->bufs, 0, sizeof(p->bufs));
    Bufs& b = p->bufs;
    int rc = 0;
    rc |= dalloc(p, &p->ddesc, 1);
    double *q0, *dq0, *U0, *tg;
    // the two buffers of X (of U) are the halves of ONE allocation, U padded to T rows: for n_x = n_u the two pairs have the same stride,
    // which lets the register-resident sweep address all four with one 32-bit offset per buffer (ilqr_kernels_dpp.hip)
    rc |= dalloc(p, &b.X[0], (size_t)2 * T * NX * Bp); b.X[1] = b.X[0] + (size_t)T * NX * Bp;
    rc |= dalloc(p, &b.U[0], (size_t)2 * T * NU * Bp); b.U[1] = b.U[0] + (size_t)T * NU * Bp;
    rc |= dalloc(p, &U0, (size_t)(T - 1) * NU * Bp);
    rc |= dalloc(p, &b.KD, (size_t)(T - 1) * Bp * NU * kd_rowp(NX));
    rc |= dalloc(p, &q0, (size_t)DOF * Bp);
    rc |= dalloc(p, &dq0, (size_t)DOF * Bp);
    rc |= dalloc(p, &tg, (size_t)(d->n_kp > 0 ? d->n_kp : 1) * NF * Bp);
    rc |= dalloc(p, &b.cost, Bp);
    rc |= dalloc(p, &b.alpha, Bp);
    rc |= dalloc(p, &b.cur, Bp);
    rc |= dalloc(p, &b.active, Bp);
    rc |= dalloc(p, &b.iters, Bp);
    rc |= dalloc(p, &b.status, Bp);
    rc |= dalloc(p, &b.pend, Bp);
    rc |= dalloc(p, &b.pred, Bp);
    rc |= dalloc(p, &b.lsc, (size_t)16 * Bp);
    rc |= dalloc(p, &b.dun, Bp);
    rc |= dalloc(p, &b.kpdev, (size_t)(d->n_kp > 0 ? d->n_kp : 1) * (NX + NU) * Bp);
    rc |= dalloc(p, &b.kpx, (size_t)(d->n_kp > 0 ? d->n_kp : 1) * 16 * (NX + NU) * Bp);
    rc |= dalloc(p, &b.dunA, (size_t)16 * Bp);
    rc |= dalloc(p, &b.kpd, (size_t)(d->n_kp > 0 ? d->n_kp : 1) * (NX + NX * NX) * Bp);
    if (rc) { ilqr_problem_destroy(p); return 1; }
    b.U0 = U0; b.q0 = q0; b.dq0 = dq0; b.kp_tg = tg; b.desc = p->ddesc;
    bool up_ok = hipMemcpyAsync(p->ddesc, &p->hdesc, sizeof(DevDesc), hipMemcpyHostToDevice, c->stream) == hipSuccess;
    if (p->B >= SPLIT_MIN_BATCH) {  // halves for the two-stream solve: cut at a multiple of 64 instances (whole waves, whole 128-byte lines)
        p->half_b0[0] = 0;
        p->half_B[0] = ((p->B / 2) + 63) / 64 * 64;
        p->half_b0[1] = p->half_B[0];
        p->half_B[1] = p->B - p->half_B[0];
        for (int i = 0; i < 2 && up_ok; i++) {
            DevDesc hd = p->hdesc;
            hd.B = p->half_B[i];
            up_ok = dalloc(p, &p->ddesc_half[i], 1) == 0 &&
                    hipMemcpyAsync(p->ddesc_half[i], &hd, sizeof(DevDesc), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
                    hipStreamSynchronize(c->stream) == hipSuccess;  // hd is a stack copy
        }
    }
    if (!up_ok || hipStreamSynchronize(c->stream) != hipSuccess) {
        ilqr_problem_destroy(p);
        return fail(c, "descriptor upload failed");
    }
    c->problems.push_back(p);
    *out = p;
    return 0;
}

extern "C" void ilqr_problem_destroy(ilqr_problem* p) {
    if (!p) return;
    for (auto it = p->ctx->problems.begin(); it != p->ctx->problems.end(); ++it)
        if (*it == p) { p->ctx->problems.erase(it); break; }
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    for (void* q : p->allocs) (void)hipFree(q);
    if (p->staging) (void)hipFree(p->staging);
    batchcp_free(p->cp);
    batchwide_free(p->cpw);
    delete p;
}

// natural [B][rows] host or device array -> SoA [rows][Bp] device buffer
static int upload(ilqr_problem* p, const double* src, bool src_is_dev, double* dst, int rows) {
    ilqr_ctx* c = p->ctx;
    const size_t n = (size_t)p->B * rows;
    const double* dsrc = src;
    if (!src_is_dev) {
        if (ensure_staging(p, n)) return 1;
        HIPCHK(c, hipMemcpyAsync(p->staging, src, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        dsrc = p->staging;
    }
    launch_to_soa(dsrc, dst, p->B, p->Bp, rows, c->stream);
    HIPCHK(c, hipGetLastError());
    if (!src_is_dev) HIPCHK(c, hipStreamSynchronize(c->stream));  // staging is reused; the host buffer may go away
    return 0;
}

static int set_init_state(ilqr_problem* p, const double* q0, const double* dq0, bool dev) {
    if (!p) return 1;
    if (!q0) return fail(p->ctx, "q0 is required");
    if (upload(p, q0, dev, (double*)p->bufs.q0, DOF)) return 1;
    if (dq0) {
        if (upload(p, dq0, dev, (double*)p->bufs.dq0, DOF)) return 1;
    } else {
        HIPCHK(p->ctx, hipMemsetAsync((void*)p->bufs.dq0, 0, sizeof(double) * DOF * p->Bp, p->ctx->stream));
    }
    p->has_state = true;
    return 0;
}
extern "C" int ilqr_problem_set_init_state(ilqr_problem* p, const double* q0, const double* dq0) { return set_init_state(p, q0, dq0, false); }
extern "C" int ilqr_problem_set_init_state_dev(ilqr_problem* p, const double* q0, const double* dq0) { return set_init_state(p, q0, dq0, true); }

static int set_kp(ilqr_problem* p, int k, const double* tg, bool dev) {
    if (!p) return 1;
    if (k < 0 || k >= p->desc.n_kp || !tg) return fail(p->ctx, "bad keypoint index / null target");
    return upload(p, tg, dev, (double*)p->bufs.kp_tg + (size_t)k * p->dims.n_f * p->Bp, p->dims.n_f);
}
extern "C" int ilqr_problem_set_keypoint_targets(ilqr_problem* p, int k, const double* tg) { return set_kp(p, k, tg, false); }
extern "C" int ilqr_problem_set_keypoint_targets_dev(ilqr_problem* p, int k, const double* tg) { return set_kp(p, k, tg, true); }

static int set_controls(ilqr_problem* p, const double* U0, bool dev) {
    if (!p) return 1;
    if (!U0) return fail(p->ctx, "U0 is required");
    if (upload(p, U0, dev, (double*)p->bufs.U0, (p->T - 1) * p->dims.n_u)) return 1;
    p->has_controls = true;
    p->u0_zero = false;
    if (!dev) {
        const size_t n = (size_t)p->B * (p->T - 1) * p->dims.n_u;
        size_t i = 0;
        while (i < n && U0[i] == 0.0) i++;
        p->u0_zero = (i == n);
    }
    return 0;
}
extern "C" int ilqr_problem_set_controls(ilqr_problem* p, const double* U0) { return set_controls(p, U0, false); }
extern "C" int ilqr_problem_set_controls_dev(ilqr_problem* p, const double* U0) { return set_controls(p, U0, true); }

extern "C" int ilqr_problem_set_constraints(ilqr_problem* p, int m, int per_step, const double* A, const double* b, const double* lambda0) {
    if (!p) return 1;
    ilqr_ctx* c = p->ctx;
    if (m <= 0 || !A || !b) return fail(c, "bad constraint arguments");
    const int ns = p->dims.n_x + p->dims.n_u, T = p->T;
    const size_t nk = per_step ? (size_t)(T - 1) : 1;
    if (p->bufs.m != m || p->bufs.per_step != per_step || !p->conA) {
        if (p->conA) {  // another shape: release the previous constraint buffers (nothing in flight may still read them)
            HIPCHK(c, hipStreamSynchronize(c->stream));
            void* old[5] = {p->conA, p->conb, p->bufs.lambda, p->bufs.Is, p->lambda0};
            for (void* q : old) {
                for (auto it = p->allocs.begin(); it != p->allocs.end(); ++it)
                    if (*it == q) { p->allocs.erase(it); break; }
                (void)hipFree(q);
            }
            p->conA = p->conb = p->lambda0 = nullptr;
            p->bufs.lambda = p->bufs.Is = nullptr;
            p->bufs.m = 0;
        }
        if (dalloc(p, &p->conA, nk * m * ns) || dalloc(p, &p->conb, nk * m)) return 1;
        double *lam, *Is;
        if (dalloc(p, &lam, (size_t)(T - 1) * m * p->Bp) || dalloc(p, &Is, (size_t)(T - 1) * m * p->Bp)) return 1;
        p->bufs.lambda = lam;
        p->bufs.Is = Is;
        if (dalloc(p, &p->lambda0, (size_t)(T - 1) * m * p->Bp)) return 1;
    }
    HIPCHK(c, hipMemcpyAsync(p->conA, A, nk * m * ns * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(p->conb, b, nk * m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    p->bufs.m = m; p->bufs.per_step = per_step; p->bufs.conA = p->conA; p->bufs.conb = p->conb;
    p->con_state_only = true;
    for (size_t k = 0; k < nk * m; k++)
        for (int j = p->dims.n_x; j < ns; j++)
            if (A[k * ns + j] != 0.0) p->con_state_only = false;
    if (lambda0) {
        if (upload(p, lambda0, false, p->bufs.lambda, (T - 1) * m)) return 1;
    } else {
        HIPCHK(c, hipMemsetAsync(p->bufs.lambda, 0, sizeof(double) * (size_t)(T - 1) * m * p->Bp, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(p->lambda0, p->bufs.lambda, sizeof(double) * (size_t)(T - 1) * m * p->Bp, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

extern "C" int ilqr_problem_reset_multipliers(ilqr_problem* p) {
    if (!p) return 1;
    if (p->bufs.m <= 0) return fail(p->ctx, "no constraints set");
    HIPCHK(p->ctx, hipMemcpyAsync(p->bufs.lambda, p->lambda0, sizeof(double) * (size_t)(p->T - 1) * p->bufs.m * p->Bp,
                                  hipMemcpyDeviceToDevice, p->ctx->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------ solvers

static int ensure_trace(ilqr_problem* p, int nb_iter) {
    if (nb_iter > p->trace_iters) {
        double *ct, *at;
        if (dalloc(p, &ct, (size_t)nb_iter * p->Bp, false) || dalloc(p, &at, (size_t)nb_iter * p->Bp, false)) return 1;
        p->bufs.cost_trace = ct;
        p->bufs.alpha_trace = at;
        p->trace_iters = nb_iter;
    }
    if (nb_iter > 0) {  // NaN-fill: entries after an instance's early stop stay NaN
        HIPCHK(p->ctx, hipMemsetAsync(p->bufs.cost_trace, 0xFF, sizeof(double) * (size_t)nb_iter * p->Bp, p->ctx->stream));
        HIPCHK(p->ctx, hipMemsetAsync(p->bufs.alpha_trace, 0xFF, sizeof(double) * (size_t)nb_iter * p->Bp, p->ctx->stream));
    }
    return 0;
}

// Which kernels run an iteration.  Default ("v2"): cooperative kernels -- closed-form single-integrator sweep or the f64-MFMA sweep, all
// step sizes of the line search in one pass.  ilqr_ctx_set_crosscheck(ctx, 1, ..) forces the generic lane-per-instance kernels, the cross-check
// set of the parity tests (also the product path where the cooperative kernels do not apply: joint-space AL rows on the controls, more
// than 16 AL rows, a second limit set).  The library reads NO environment variable: the switch is context state.
static int path_choice(const ilqr_ctx* c) { return c->xc_generic ? 1 : 2; }

// The buffer table of one half of a split problem: every per-instance array is [..][Bp] with the instance innermost, so a half is the
// same table with the base pointers moved by its first instance (gain records: by whole records) and its own descriptor (B = its size).
static Bufs half_bufs(const ilqr_problem* p, int half) {
    Bufs v = p->bufs;
    const size_t o = (size_t)p->half_b0[half];
    v.desc = p->ddesc_half[half];
    for (int i = 0; i < 2; i++) { v.X[i] += o; v.U[i] += o; }
    v.U0 += o; v.q0 += o; v.dq0 += o; v.kp_tg += o;
    v.KD += o * (size_t)kd_rs(p->bufs.kd_sym, p->dims.n_u, kd_rowp(p->dims.n_x));
    v.cost += o; v.alpha += o; v.cur += o; v.active += o; v.iters += o; v.status += o; v.kpd += o; v.pend += o; v.pred += o;
    v.lsc += o; v.dun += o; v.kpdev += o; v.kpx += o; v.dunA += o;
    if (v.cost_trace) { v.cost_trace += o; v.alpha_trace += o; }
    if (v.lambda) { v.lambda += o; v.Is += o; }
    return v;
}

static int solve_riccati(ilqr_problem* p, bool al, int nb_iter, int lag, double penalty0, double scaling, int line_search, int early_stop) {
    if (!p) return 1;
    ilqr_ctx* c = p->ctx;
    if (!p->has_state || !p->has_controls) return fail(c, "set_init_state and set_controls must be called before a solve");
    if (nb_iter < 0) return fail(c, "nb_iter < 0");
    if (al && (p->bufs.m <= 0 || lag <= 0)) return fail(c, "AL solve needs constraints (ilqr_problem_set_constraints) and lag_update_step > 0");
    HIPCHK(c, hipSetDevice(c->device));
    if (ensure_trace(p, nb_iter)) return 1;
    p->last_nb_iter = nb_iter;
    const int kind = p->desc.kind, nd = p->desc.nb_deriv;
    // number of step sizes the do/while of ILQRRecursive.cpp:101-155 can reach: 1, 1/2, ... until alpha <= alpha_floor
    int n_alpha = 1;
    if (line_search) { double al_ = 1.0; while (al_ > p->desc.alpha_floor && n_alpha < 64) { al_ *= 0.5; n_alpha++; } }
    const int path = p->desc.limits2_set ? 1 : path_choice(c);  // a second limit set exists in the generic kernels only
    const bool coop = (path != 1) && n_alpha <= 16;            // all step sizes at once (16 lanes / rows per instance)
    const bool fwd_wave = coop && forward_wave_supported(kind, nd, n_alpha);  // PosOrn-1 / JointSpace-1: linear line search, 32 lanes per instance
    const bool fwd_lin = coop && !fwd_wave && forward_lin_supported(kind, nd, n_alpha);  // PosOrn-2: linear line search, 8 lanes per instance
    // the register-resident sweep addresses x, u and the multipliers with 32-bit byte offsets (ilqr_kernels_dpp.hip): batches whose arrays pass
    // 4 GiB (T * Bp beyond ~38 M) take the other sweeps
    const bool off32 = (size_t)2 * p->T * p->dims.n_x * p->Bp * 8 < ((size_t)1 << 32) && (size_t)p->T * (p->bufs.m > 0 ? p->bufs.m : 1) * p->Bp * 8 < ((size_t)1 << 32);
    const bool bwd_si = (path != 1) && off32 && backward_si_supported(kind, nd, al, p->bufs.m, p->bufs.per_step, p->con_state_only);
    const bool bwd_mfma = (path != 1) && !bwd_si && backward_mfma_supported(kind, nd, al, p->bufs.m);  // wave per instance, f64 matrix cores
    // 16 lanes per instance, rows in registers (round 3): a lone wave's chain is longer than the matrix-core sweep's (553 against 316 us at the C4 shape), but four
    // instances share a wave: it wins as soon as the wave-per-instance sweep needs a second round of waves (measured: B = 2048 600 against 548 us, B = 4096 597 against 1040)
    const bool bwd_rows = bwd_mfma && c->xc_sweep != 1 && (c->xc_sweep == 2 || p->B > 2 * c->n_simd) && backward_rows_supported(kind, nd, al, p->bufs.m);
    bool uniform_R = true;
    for (int i = 1; i < p->dims.n_u; i++) uniform_R = uniform_R && (p->desc.R_diag[i] == p->desc.R_diag[0]);
    const bool fused = bwd_si && fwd_wave;  // the sweep applies the previous line search's winner itself (ilqr_kernels_dpp.hip)
    // uniform control weights: the sweep's closed form for N = M D - I with D a multiple of I, writing the packed symmetric gain record (ilqr_kernels.hpp:
    // KD_SYM_RS) that the two forward passes of this path and the getters read; any other forward pass reads plain records, so then the general form runs
    const bool unif_sym = bwd_si && uniform_R && fwd_wave;
    p->bufs.kd_sym = unif_sym ? 1 : 0;

    // ---- one or two independent halves ("lanes" of the launch schedule).  Instances never interact, so the halves of a large batch are
    // two complete solves on two streams; the second one starts one sweep later, so that its latency-bound sweep runs under the other
    // half's bandwidth-bound forward pass.  Results do not depend on the split (bit for bit: tests/test_gpu_fullsize.py).  With
    // per-launch profiling on the problem runs unsplit on the context's stream: the event marks time one kernel at a time.
    struct Half { Bufs bufs; int B; hipStream_t st; };
    Half hv[2];
    int nh = 1;
    // Measured (rocprofv3 kernel trace, B = 4096): it pays for the wave-per-instance MFMA sweep (C4: 60.3 -> 51.7 ms per solve; 4096 one-wave
    // workgroups on 3072 wave slots otherwise leave a one-third-full second round), not for the single-integrator pipeline (C3: the forward
    // pass slows from 0.125 to 0.24 ms and k_kp_derivs from 0.017 to 0.08-0.14 ms when they share the SIMDs with the other half's sweep:
    // 0.53 ms per iteration against 0.49 unsplit).
    const bool split = coop && ((bwd_mfma && !bwd_rows && c->split) || c->split == 2) && !c->profile && p->ddesc_half[0] && nb_iter > 0;
    if (!bwd_si && !bwd_mfma && nb_iter > 0 && !p->bufs.ws)  // the generic sweep keeps the matrices of a step in an explicit workspace
        if (dalloc(p, &p->bufs.ws, (size_t)backward_ws_entries(kind, nd) * p->Bp, false)) return 1;
    if (split) {
        for (int i = 0; i < 2; i++) {
            if (!c->half_stream[i]) HIPCHK(c, hipStreamCreateWithFlags(&c->half_stream[i], hipStreamNonBlocking));
            if (!c->ev_half_done[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_half_done[i], hipEventDisableTiming));
        }
        if (!c->ev_begin) HIPCHK(c, hipEventCreateWithFlags(&c->ev_begin, hipEventDisableTiming));
        if (!c->ev_stagger) HIPCHK(c, hipEventCreateWithFlags(&c->ev_stagger, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_begin, c->stream));  // inputs (and the trace fill) are ordered on the caller's stream
        nh = 2;
        for (int i = 0; i < 2; i++) {
            HIPCHK(c, hipStreamWaitEvent(c->half_stream[i], c->ev_begin, 0));
            hv[i].bufs = half_bufs(p, i); hv[i].B = p->half_B[i]; hv[i].st = c->half_stream[i];
        }
    } else {
        hv[0].bufs = p->bufs; hv[0].B = p->B; hv[0].st = c->stream;
    }

    // Join on EVERY exit path: an early return between here and the end of the solve (a failed launch or event call on one half) must not
    // leave work queued on a half stream that the context's stream never waits for -- the caller's next call would race with it.
    struct SplitJoin {
        ilqr_ctx* c; bool on;
        ~SplitJoin() {
            if (!on) return;
            for (int i = 0; i < 2; i++)
                if (hipEventRecord(c->ev_half_done[i], c->half_stream[i]) != hipSuccess || hipStreamWaitEvent(c->stream, c->ev_half_done[i], 0) != hipSuccess)
                    (void)hipStreamSynchronize(c->half_stream[i]);  // last resort: drain it here
        }
    } split_join{c, split};
    FwdArgs f;
    std::memset(&f, 0, sizeof(f));
    f.line_search = line_search; f.early_stop = early_stop; f.nb_iter = nb_iter; f.penalty_roll = penalty0; f.n_alpha = n_alpha; f.al = al ? 1 : 0; f.n_kp = p->desc.n_kp;
    f.fused = fused ? 1 : 0;
    f.limits = p->desc.limits_set ? 1 : 0;
    // time systems, re-roll of the winner: 16 lanes per instance on registers up to a quarter wave of 4 instances per SIMD (B = 256: 64 against 92 us), the
    // 8-lanes-per-instance kernel beyond (its waves cover 64 contiguous bytes of every [row][b] line, the other's 32: B = 2048 117 against 126 us, 4096 190 / 240)
    f.apply_dpp = (c->xc_apply == 2 || (c->xc_apply == 0 && (p->B + 3) / 4 <= c->n_simd / 4)) ? 1 : 0;
    // small batches (up to three quarters of a wave of 4 instances per SIMD): the rollout is a chain, not a stream -- k_forward_dpp.  Measured crossover
    // with k_forward_wg on C3 (forward + decision, us): B = 2048 92 / 122, 3072 114 / 124, 4096 139 / 126
    f.small = (fwd_wave && c->xc_fwd != 1 && (c->xc_fwd == 2 || (p->B + 3) / 4 <= 3 * c->n_simd / 4)) ? 1 : 0;
    for (int k = 0; k < p->desc.n_kp; k++) f.kp_ext |= p->desc.kp_dist[k] | p->desc.kp_has_frame[k] | p->desc.kp_has_Ru[k] | p->desc.kp_joint[k];
    for (int h = 0; h < nh; h++) {
        const Bufs& bf = hv[h].bufs;
        const int B = hv[h].B;
        hipStream_t st = hv[h].st;
        ProfScope ps(c, ILQR_PROF_ROLLOUT);
        if (path != 1 && init_lti_supported(kind, nd)) {
            launch_init_lti(kind, nd, bf, B, st);
            if (al && !fused) {  // active-set weights of the initial trajectory: I_k = penalty * (g<0 && lambda==0 ? 0 : 1)
                f.it = -1; f.do_update = 0;
                launch_solver_v2(kind, nd, KER_AL_UPDATE, al, bf, B, p->T, st, f);
            }
        } else {
            launch_solver(kind, nd, KER_INIT, al, bf, B, st, f);
        }
    }
    HIPCHK(c, hipGetLastError());
    double penalty = penalty0;
    SweepArgs sw;
    sw.pen_in = penalty; sw.pen_update_prev = penalty; sw.do_update_prev = 0;
    for (int it = 0; it < nb_iter; it++) {
        FwdArgs fi = f;
        fi.it = it;
        fi.penalty_roll = penalty;  // I_k is stored pre-multiplied by the penalty current at rollout time (AL-ILQR.cpp:190)
        fi.do_update = al && ((it + 1) % lag == 0);
        if (fi.do_update) penalty *= scaling;  // multipliers use the UPDATED penalty (AL-ILQR.cpp:203-205)
        fi.penalty_update = penalty;
        for (int h = 0; h < nh; h++) {
            const Bufs& bf = hv[h].bufs;
            const int B = hv[h].B;
            hipStream_t st = hv[h].st;
            if (split && it == 0 && h == 1) HIPCHK(c, hipStreamWaitEvent(st, c->ev_stagger, 0));  // one sweep behind the first half
            {
                {   // l_x, l_xx at the keypoint steps (FK, log map, J'QJ) for every sweep: none of them holds keypoint code
                    ProfScope ps(c, ILQR_PROF_OTHER);
                    launch_solver(kind, nd, KER_KP_DERIVS, al, bf, B, st, fi);
                }
                ProfScope ps(c, ILQR_PROF_BACKWARD);
                // rows in registers, DPP broadcasts (ilqr_kernels_dpp.hip): 16 lanes per instance while that gives every SIMD at most one wave
                // (the launch is then bound by one wave's instruction stream, which is shorter with 4 instances per wave), 8 lanes per
                // instance beyond (half the instructions per instance).  Measured crossover between 4096 and 8192 instances on 1024 SIMDs.
                if (bwd_si) launch_backward_si_dpp(al, fused, unif_sym, (B + 3) / 4 <= c->n_simd ? 16 : 8, bf, B, st, sw);
                else if (bwd_rows) launch_backward_rows(kind, nd, al, bf, B, st);
                else if (bwd_mfma) launch_backward_mfma(kind, nd, al, bf, B, st);
                else launch_solver(kind, nd, KER_BACKWARD, al, bf, B, st, fi);
            }
            if (split && it == 0 && h == 0) HIPCHK(c, hipEventRecord(c->ev_stagger, st));
            if (coop) {
                {
                    ProfScope ps(c, ILQR_PROF_FORWARD);
                    if (fwd_wave) launch_forward_wave(kind, bf, B, st, fi);
                    else if (fwd_lin) launch_forward_lin(nd, KER_FWD_SPEC, bf, B, p->T, st, fi);
                    else launch_solver_v2(kind, nd, KER_FWD_SPEC, al, bf, B, p->T, st, fi);
                }
                if (fwd_wave) {  // winner applied, AL bookkeeping and buffer flip in one pass over the trajectory -- or by the next sweep (fused)
                    if (!fused || it == nb_iter - 1) {
                        ProfScope ps(c, ILQR_PROF_APPLY);
                        launch_apply_wave(kind, bf, B, p->T, st, fi);
                    }
                } else if (fwd_lin) {  // the cost pass writes no trajectory: the winner is always re-rolled
                    ProfScope ps(c, ILQR_PROF_APPLY);
                    launch_forward_lin(nd, KER_FWD_APPLY, bf, B, p->T, st, fi);
                } else if (line_search) {  // time systems: re-roll of the winner where the speculated step size lost, 8 lanes per instance
                    ProfScope ps(c, ILQR_PROF_APPLY);
                    launch_apply_rows_tm(kind, nd, bf, B, st, fi);
                }
                if (al && !fwd_wave) {  // active-set weights of the accepted trajectory (+ multiplier update every `lag` iterations)
                    ProfScope ps(c, ILQR_PROF_OTHER);
                    launch_solver_v2(kind, nd, KER_AL_UPDATE, al, bf, B, p->T, st, fi);
                }
            } else {
                ProfScope ps(c, ILQR_PROF_FORWARD);
                launch_solver(kind, nd, KER_FORWARD, al, bf, B, st, fi);
            }
        }
        sw.pen_in = fi.penalty_roll; sw.pen_update_prev = fi.penalty_update; sw.do_update_prev = fi.do_update;  // for the next sweep
        HIPCHK(c, hipGetLastError());
    }
    // (the caller's stream continues when both halves are done: SplitJoin below, on every exit path)
    prof_mark(c, -1);
    return 0;
}

extern "C" int ilqr_solve_recursive(ilqr_problem* p, int nb_iter, int line_search, int early_stop) {
    return solve_riccati(p, false, nb_iter, 1, 0.0, 1.0, line_search, early_stop);
}
extern "C" int ilqr_solve_al(ilqr_problem* p, int nb_iter, int lag, double penalty, double scaling, int line_search, int early_stop) {
    return solve_riccati(p, true, nb_iter, lag, penalty, scaling, line_search, early_stop);
}

extern "C" int ilqr_solve_batch_cp(ilqr_problem* p, const double* psi, int Kw, int nb_iter, int early_stop) {
    if (!p) return 1;
    ilqr_ctx* c = p->ctx;
    if (!p->has_state || !p->has_controls) return fail(c, "set_init_state and set_controls must be called before a solve");
    HIPCHK(c, hipSetDevice(c->device));
    std::string err;
    if (ensure_trace(p, nb_iter)) return 1;
    p->last_nb_iter = nb_iter;
    const bool cp_time = p->desc.kind == ILQR_SYS_POS_ORN_TIME || p->desc.kind == ILQR_SYS_JOINT_TIME;
    if (psi && Kw > 16 && !(cp_time && Kw <= 32)) {  // wide basis: low-rank form of the normal equations (ilqr_batchwide.hip)
        if (batchwide_solve(p->cpw, p->hdesc, p->bufs, p->dims.n_x, p->dims.n_u, psi, Kw, nb_iter, early_stop, p->u0_zero, c->stream, err, prof_hook(c))) return fail(c, err);
        prof_mark(c, -1);
        return 0;
    }
    p->cp.xc_lane_solve = c->xc_cp_lane; p->cp.xc_general = c->xc_cp_general;
    if (batchcp_solve(p->cp, p->hdesc, p->bufs, p->dims.n_x, p->dims.n_u, p->dims.n_f, p->dims.n_Q, psi, Kw, nb_iter, early_stop, c->stream, err, prof_hook(c)))
        return fail(c, err);
    prof_mark(c, -1);
    return 0;
}

extern "C" int ilqr_solve_batch(ilqr_problem* p, int nb_iter, int early_stop) {
    if (!p) return 1;
    ilqr_ctx* c = p->ctx;
    if (!p->has_state || !p->has_controls) return fail(c, "set_init_state and set_controls must be called before a solve");
    HIPCHK(c, hipSetDevice(c->device));
    std::string err;
    if (ensure_trace(p, nb_iter)) return 1;
    p->last_nb_iter = nb_iter;
    if (batchwide_solve(p->cpw, p->hdesc, p->bufs, p->dims.n_x, p->dims.n_u, nullptr, 0, nb_iter, early_stop, p->u0_zero, c->stream, err, prof_hook(c))) return fail(c, err);
    prof_mark(c, -1);
    return 0;
}

// ------------------------------------------------------------------------------------------------ results

enum { GET_PLAIN, GET_CUR, GET_SCALED };
static int download(ilqr_problem* p, int mode, const double* s0, const double* s1, double* dst, bool dst_is_dev, int rows) {
    ilqr_ctx* c = p->ctx;
    if (!dst) return fail(c, "null output pointer");
    const size_t n = (size_t)p->B * rows;
    double* ddst = dst;
    if (!dst_is_dev) {
        if (ensure_staging(p, n)) return 1;
        ddst = p->staging;
    }
    if (mode == GET_CUR) launch_from_soa_cur(s0, s1, p->bufs.cur, ddst, p->B, p->Bp, rows, c->stream);
    else if (mode == GET_SCALED) launch_from_soa_scaled(s0, p->bufs.alpha, p->bufs.iters, ddst, p->B, p->Bp, rows, c->stream);
    else launch_from_soa(s0, ddst, p->B, p->Bp, rows, c->stream);
    HIPCHK(c, hipGetLastError());
    if (!dst_is_dev) {
        HIPCHK(c, hipMemcpyAsync(dst, p->staging, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

extern "C" int ilqr_problem_get_X(ilqr_problem* p, double* X) { return p ? download(p, GET_CUR, p->bufs.X[0], p->bufs.X[1], X, false, p->T * p->dims.n_x) : 1; }
extern "C" int ilqr_problem_get_U(ilqr_problem* p, double* U) { return p ? download(p, GET_CUR, p->bufs.U[0], p->bufs.U[1], U, false, (p->T - 1) * p->dims.n_u) : 1; }
extern "C" int ilqr_problem_get_X_dev(ilqr_problem* p, double* X) { return p ? download(p, GET_CUR, p->bufs.X[0], p->bufs.X[1], X, true, p->T * p->dims.n_x) : 1; }
extern "C" int ilqr_problem_get_U_dev(ilqr_problem* p, double* U) { return p ? download(p, GET_CUR, p->bufs.U[0], p->bufs.U[1], U, true, (p->T - 1) * p->dims.n_u) : 1; }
static int get_gains(ilqr_problem* p, double* K, double* d) {
    if (!p) return 1;
    ilqr_ctx* c = p->ctx;
    double* dst = K ? K : d;
    if (!dst) return fail(c, "null output pointer");
    const int T1 = p->T - 1, nu = p->dims.n_u, nx = p->dims.n_x;
    const size_t n = (size_t)p->B * T1 * nu * (K ? nx : 1);
    if (ensure_staging(p, n)) return 1;
    launch_get_gains(p->bufs.KD, p->bufs.kd_sym, p->bufs.alpha, p->bufs.iters, K ? p->staging : nullptr, K ? nullptr : p->staging, p->B, p->Bp, T1, nu, nx, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(dst, p->staging, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
// ---- receding horizon / tracking
extern "C" int ilqr_problem_warm_start(ilqr_problem* p, int shift) {
    if (!p) return 1;
    ilqr_ctx* c = p->ctx;
    if (!p->has_state || !p->has_controls) return fail(c, "warm start needs a previous solve (set_init_state, set_controls, solve)");
    if (shift < 0 || shift >= p->T) return fail(c, "shift must be in [0, T)");
    HIPCHK(c, hipSetDevice(c->device));
    launch_warm_start(p->bufs, const_cast<double*>(p->bufs.U0), const_cast<double*>(p->bufs.q0), const_cast<double*>(p->bufs.dq0), shift, p->B, p->T,
                      p->dims.n_x, p->dims.n_u, p->desc.nb_deriv, c->stream);
    HIPCHK(c, hipGetLastError());
    return 0;
}
static int track(ilqr_problem* p, int k, const double* x_meas, int with_ff, double* u_out, bool dev) {
    if (!p) return 1;
    ilqr_ctx* c = p->ctx;
    if (!x_meas || !u_out) return fail(c, "null pointer");
    if (k < 0 || k >= p->T - 1) return fail(c, "timestep outside [0, T-2]");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nxb = (size_t)p->B * p->dims.n_x, nub = (size_t)p->B * p->dims.n_u;
    const double* xs = x_meas;
    double* us = u_out;
    if (!dev) {
        if (ensure_staging(p, nxb + nub)) return 1;
        HIPCHK(c, hipMemcpyAsync(p->staging, x_meas, nxb * sizeof(double), hipMemcpyHostToDevice, c->stream));
        xs = p->staging;
        us = p->staging + nxb;
    }
    launch_track(p->bufs, xs, k, with_ff, us, p->B, p->dims.n_x, p->dims.n_u, c->stream);
    HIPCHK(c, hipGetLastError());
    if (!dev) {
        HIPCHK(c, hipMemcpyAsync(u_out, us, nub * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}
extern "C" int ilqr_problem_track(ilqr_problem* p, int k, const double* x_meas, int with_feedforward, double* u_out) {
    return track(p, k, x_meas, with_feedforward, u_out, false);
}
extern "C" int ilqr_problem_track_dev(ilqr_problem* p, int k, const double* x_meas, int with_feedforward, double* u_out) {
    return track(p, k, x_meas, with_feedforward, u_out, true);
}

extern "C" int ilqr_problem_get_K(ilqr_problem* p, double* K) { return get_gains(p, K, nullptr); }
extern "C" int ilqr_problem_get_d(ilqr_problem* p, double* d) { return get_gains(p, nullptr, d); }
extern "C" int ilqr_problem_get_cost(ilqr_problem* p, double* cost) { return p ? download(p, GET_PLAIN, p->bufs.cost, nullptr, cost, false, 1) : 1; }
extern "C" int ilqr_problem_get_cost_dev(ilqr_problem* p, double* cost) { return p ? download(p, GET_PLAIN, p->bufs.cost, nullptr, cost, true, 1) : 1; }
extern "C" int ilqr_problem_get_alpha(ilqr_problem* p, double* alpha) { return p ? download(p, GET_PLAIN, p->bufs.alpha, nullptr, alpha, false, 1) : 1; }
extern "C" int ilqr_problem_get_lambda(ilqr_problem* p, double* lam) {
    if (!p) return 1;
    if (p->bufs.m <= 0) return fail(p->ctx, "no constraints set");
    return download(p, GET_PLAIN, p->bufs.lambda, nullptr, lam, false, (p->T - 1) * p->bufs.m);
}

static int get_ints(ilqr_problem* p, const int* src, int* dst) {
    if (!p) return 1;
    if (!dst) return fail(p->ctx, "null output pointer");
    HIPCHK(p->ctx, hipMemcpyAsync(dst, src, sizeof(int) * p->B, hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(p->ctx, hipStreamSynchronize(p->ctx->stream));
    return 0;
}
extern "C" int ilqr_problem_get_iters(ilqr_problem* p, int* iters) { return get_ints(p, p ? p->bufs.iters : nullptr, iters); }
extern "C" int ilqr_problem_get_status(ilqr_problem* p, int* status) { return get_ints(p, p ? p->bufs.status : nullptr, status); }

extern "C" int ilqr_problem_get_trace(ilqr_problem* p, double* ct, double* at, int nb_iter) {
    if (!p) return 1;
    if (nb_iter <= 0 || nb_iter > p->last_nb_iter) return fail(p->ctx, "nb_iter exceeds the last solve's iteration count");
    if (ct && download(p, GET_PLAIN, p->bufs.cost_trace, nullptr, ct, false, nb_iter)) return 1;
    if (at && download(p, GET_PLAIN, p->bufs.alpha_trace, nullptr, at, false, nb_iter)) return 1;
    return 0;
}

extern "C" int ilqr_problem_get_fX(ilqr_problem* p, double* fX) {
    if (!p) return 1;
    ilqr_ctx* c = p->ctx;
    if (!fX) return fail(c, "null output pointer");
    const size_t n = (size_t)p->B * p->T * p->dims.n_f;
    if (ensure_staging(p, n)) return 1;
    {
        ProfScope ps(c, ILQR_PROF_OTHER);
        launch_fx_all(p->desc.kind, p->desc.nb_deriv, p->bufs, p->B, p->T, p->staging, c->stream);
    }
    prof_mark(c, -1);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(fX, p->staging, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------ stand-alone FK

extern "C" int ilqr_fk_batch(ilqr_ctx* c, const ilqr_problem_desc* d, int n, const double* q, double* pos, double* quat, double* jac) {
    if (!c) return 1;
    if (!d || n <= 0 || !q) return fail(c, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    DevDesc h;
    std::memset(&h, 0, sizeof(h));
    if (lower_chain(c, *d, h.chain)) return 1;
    DevDesc* dd = nullptr;
    double *dq = nullptr, *dp = nullptr, *dqt = nullptr, *dj = nullptr;
    int rc = 0;
    auto cleanup = [&]() { (void)hipFree(dd); (void)hipFree(dq); (void)hipFree(dp); (void)hipFree(dqt); (void)hipFree(dj); };
#define FKCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return fail(c, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)
    FKCHK(hipMalloc((void**)&dd, sizeof(DevDesc)));
    FKCHK(hipMalloc((void**)&dq, sizeof(double) * n * DOF));
    if (pos) FKCHK(hipMalloc((void**)&dp, sizeof(double) * n * 3));
    if (quat) FKCHK(hipMalloc((void**)&dqt, sizeof(double) * n * 4));
    if (jac) FKCHK(hipMalloc((void**)&dj, sizeof(double) * n * 6 * DOF));
    FKCHK(hipMemcpyAsync(dd, &h, sizeof(DevDesc), hipMemcpyHostToDevice, c->stream));
    FKCHK(hipMemcpyAsync(dq, q, sizeof(double) * n * DOF, hipMemcpyHostToDevice, c->stream));
    launch_fk_batch(dd, n, dq, dp, dqt, dj, c->stream);
    FKCHK(hipGetLastError());
    if (pos) FKCHK(hipMemcpyAsync(pos, dp, sizeof(double) * n * 3, hipMemcpyDeviceToHost, c->stream));
    if (quat) FKCHK(hipMemcpyAsync(quat, dqt, sizeof(double) * n * 4, hipMemcpyDeviceToHost, c->stream));
    if (jac) FKCHK(hipMemcpyAsync(jac, dj, sizeof(double) * n * 6 * DOF, hipMemcpyDeviceToHost, c->stream));
    FKCHK(hipStreamSynchronize(c->stream));
#undef FKCHK
    cleanup();
    return rc;
}
