// ilqr_kernels_wave.hip -- the single-integrator fast path around the backward sweep (PosOrn / JointSpace, nb_deriv = 1):
//   k_forward_wg      forward pass of the linear line search, one 8-wave workgroup per 16 instances, 32 lanes per instance
//   k_select          line-search decision: task cost of every step size from the exported keypoint deviations (one lane per alpha)
//   k_apply, k_flip   accepted trajectory x(alpha) = xbar + alpha (x(1) - xbar) and the AL bookkeeping in one pass; buffer flip
//   k_init_roll_lti, k_init_finish   initial rollout with one lane per (instance, coordinate)
//
// The forward pass.  k_forward_lin (ilqr_kernels_rows.hip) gives each instance 8 lanes: B = 4096 is then 512 single-wave
// workgroups, half of the 1024 SIMDs stay empty and the other half run one wave whose per-step chain (LDS gather -> dot product
// -> sqrt -> dynamics) is fully exposed.  Here the 7 x 8 gain record {K | d} of a step is spread over 32 lanes -- lane (r, jl)
// owns the two adjacent entries (r, 2jl), (r, 2jl+1) and reads them with ONE 16-byte load, so a wave reads two whole records as
// 1 KiB of contiguous memory -- and B = 4096 becomes 2048 waves, two per SIMD.  Per step and lane:
//     partial = K[r][c0] dx[c0] + K[r][c1] dx[c1]          (c1 = 7 is the feed-forward column: dx[7] := 1)
//     du[r]   = sum over the row's 4 lanes                 (two DPP quad permutes, no LDS)
//     du[c0], du[c1] <- ds_bpermute from the rows c0, c1   (the transpose the product needs; LDS crossbar, no LDS memory)
//     dx[c]  += dt du[c]                                    every lane keeps its two columns of the state deviation
// xbar, ubar come in and x(1), u(1) go out through LDS in blocks of 8 timesteps, loaded / stored by the whole workgroup as full
// 128-byte lines (see k_forward_wg).  The keypoint cost (FK) is not evaluated here: at keypoint steps the deviation (dx, du) is
// written to `kpdev`, and k_select evaluates the task cost of all step sizes with one lane per (instance, alpha), picks the
// winner (ILQRRecursive.cpp:101-155) and does the bookkeeping.  That keeps the FK call -- 256 VGPRs + scratch -- out of the rollout.
#include <cstdlib>
#include <cstring>

#include "ilqr_step.hpp"

namespace ilqr {

#define LDS_ORDER() asm volatile("" ::: "memory")

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);  // every lane is written: no "old" operand, no copy
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the 4 lanes of a quad, result in all 4 (quad_perm [1,0,3,2] then [2,3,0,1])
__device__ __forceinline__ double quad_sum(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    return v;
}
__device__ __forceinline__ double bperm_f64(int byte_addr, double v) {
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Workgroup barrier that only drains LDS traffic (__syncthreads() also emits s_waitcnt vmcnt(0): it would wait for the gain
// prefetches issued 7 steps ahead).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One workgroup = 8 waves = 16 consecutive instances; a wave owns two of them (32 lanes each, see the file header).
//   gains        : read by the owning wave straight from HBM, one 16-byte load per lane and step (a wave's two records are
//                  1 KiB of contiguous memory), prefetched 7 steps ahead in a register ring
//   xbar, ubar   : [row][b] layout -- 16 instances of one row are exactly one 128-byte line.  The workgroup loads a block of
//                  S = 8 timesteps (112 lines) cooperatively, one block ahead, and stages it in LDS; the waves read their
//                  values from there.  A wave loading its own two instances would touch 16 bytes of each line.
//   x(1), u(1)   : written to an LDS block by the owning waves, stored by the whole workgroup as full lines once per block.
//                  (Measured with 16-byte pieces stored by each wave: 0.09 ms of a 0.24 ms launch went into the stores.)
// Two workgroup barriers per block of 8 steps; nothing else couples the waves.
// instances per workgroup of k_forward_wg: 16 = eight waves per workgroup and full 128-byte lines per row segment (one workgroup per
// CU at B = 4096); 8 (64-byte segments, two workgroups per CU) measured 116 us against 110
#ifndef FW_IW
#define FW_IW 16
#endif
template <int NA>
__global__ __launch_bounds__(FW_IW * 32) void k_forward_wg(Bufs a, FwdArgs f) {
    constexpr int NX = 7, NU = 7, ROWP = kd_rowp(NX), RS = NU * ROWP, NR = NX + NU;
    static_assert(ROWP == 8, "record row = 7 gains + feed-forward");
    constexpr int S = 8, PF = S - 1;     // block length = ring length: slot (step mod 8) is static in the unrolled block
    constexpr int IW = FW_IW, NT = IW * 32;  // instances per workgroup (32 lanes each) and its threads
    constexpr int NSEG = S * NR;         // 112 row segments (lines) per block
    constexpr int NRND = (NSEG * IW + NT - 1) / NT;  // loader rounds: 4 (the last one half used)
    __shared__ double sIn[2][NSEG][IW];
    __shared__ double sOut[NSEG][IW];

    const DevDesc& d = *a.desc;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 5, r8 = (lane >> 2) & 7, jl = lane & 3;
    const int b0 = xcd_tile() * IW;
    const int Bp = d.Bp, T = d.T, B = d.B;
    {   // workgroup-uniform exit: every wave looks at all 16 instances
        const int bi = b0 + (lane & (IW - 1));
        const bool any = (bi < B) && (a.active[bi < B ? bi : 0] != 0);
        if (__ballot(any ? 1 : 0) == 0ull) return;
    }
    const int li = wv * 2 + g;           // instance slot in the workgroup
    const int b = b0 + li;
    const bool inst_ok = (b < B) && (a.active[b < B ? b : 0] != 0);
    const int bb = (b < B) ? b : 0;
    const int r = r8 < NU ? r8 : NU - 1;  // row 7 shadows row 6 (never stores, never read by a bpermute)
    const int c0 = 2 * jl, c1 = 2 * jl + 1;
    const bool c1ok = c1 < NX;            // jl == 3: c1 is the feed-forward column
    const int c1c = c1ok ? c1 : NX - 1;
    const bool colst = inst_ok && (r8 == 0);               // row 0 publishes x (its 4 lanes hold all 7 columns)
    const bool rowst = inst_ok && (jl == 0) && (r8 < NU);  // the first lane of a row publishes u
    const int src0 = (g * 32 + c0 * 4) * 4, src1 = (g * 32 + c1c * 4) * 4;  // byte addresses for ds_bpermute

    const double dt = d.dt, dt1 = c1ok ? dt : 0.0;
    const int lim_on = d.limits_set;
    const double pen = d.penalty;
    const int lw0 = d.lw[c0], lw1 = c1ok ? d.lw[c1c] : 0;
    const double mx0 = lw0 ? d.smax[c0] : INFINITY, mn0 = lw0 ? d.smin[c0] : -INFINITY;
    const double mx1 = lw1 ? d.smax[c1c] : INFINITY, mn1 = lw1 ? d.smin[c1c] : -INFINITY;
    const int n_kp = d.n_kp;
    int kpi = 0, kp_next = (n_kp > 0) ? d.kp_t[0] : -1;

    // ---- cooperative loader / writer: round q of thread tid handles (segment, instance) = ((q*512 + tid) / 16, tid % 16)
    const int l_inst = tid % IW;
    const int l_b = (b0 + l_inst < B) ? b0 + l_inst : 0;
    const bool l_ok = (b0 + l_inst < B) && (a.active[l_b] != 0);
    const int l_cur = a.cur[l_b];
    const double* l_src[NRND];   // row base in the accepted trajectory (timestep 0)
    double* l_dst[NRND];         // row base in the other buffer
    int l_s[NRND], l_kmax[NRND];
    size_t l_stride[NRND];
    bool l_valid[NRND];
    UNR for (int q = 0; q < NRND; q++) {
        const int seg = (q * NT + tid) / IW;
        l_valid[q] = seg < NSEG;
        const int sg = l_valid[q] ? seg : 0;
        const int s = sg / NR, row = sg % NR;
        l_s[q] = s;
        if (row < NX) {
            l_src[q] = a.X[l_cur] + (size_t)row * Bp + l_b;
            l_dst[q] = a.X[1 - l_cur] + (size_t)row * Bp + l_b;
            l_stride[q] = (size_t)NX * Bp;
            l_kmax[q] = T - 1;
        } else {
            l_src[q] = a.U[l_cur] + (size_t)(row - NX) * Bp + l_b;
            l_dst[q] = a.U[1 - l_cur] + (size_t)(row - NX) * Bp + l_b;
            l_stride[q] = (size_t)NU * Bp;
            l_kmax[q] = T - 2;
        }
    }
    double pre[NRND];
    auto block_load = [&](int k0) {  // unconditional; timesteps beyond the end re-read the last one
        UNR for (int q = 0; q < NRND; q++) {
            const int k = k0 + l_s[q];
            pre[q] = l_src[q][(size_t)(k < l_kmax[q] ? k : l_kmax[q]) * l_stride[q]];
        }
    };
    auto block_stage = [&](int buf) {
        UNR for (int q = 0; q < NRND; q++)
            if (l_valid[q]) (&sIn[buf][0][0])[q * NT + tid] = pre[q];
    };
    auto block_store = [&](int k0) {
        UNR for (int q = 0; q < NRND; q++) {
            const int k = k0 + l_s[q];
            if (l_valid[q] && l_ok && k <= l_kmax[q]) l_dst[q][(size_t)k * l_stride[q]] = (&sOut[0][0])[q * NT + tid];
        }
    };

    // ---- gain ring
    // plain record: the lane's two entries are adjacent (one 16-byte load); packed symmetric record (a.kd_sym, ilqr_kernels.hpp): entry (r, c) lies at
    // (min, max) of the upper triangle, d_r behind it -- two 8-byte loads, the lanes of an instance still cover one contiguous record
    const int sym = a.kd_sym, rs_ = kd_rs(sym, NU, ROWP);
    const double* pK = a.KD + (size_t)bb * rs_ + kd_off(sym, ROWP, r, c0);  // plain: 16-byte aligned
    const double* pK1 = a.KD + (size_t)bb * rs_ + kd_off(sym, ROWP, r, c1);
    const size_t sK_ = (size_t)Bp * rs_;
    double rk0[S], rk1[S];
    auto fetch = [&](int slot, int k) {  // unconditional; the pointers stop at the last timestep
        if (sym) {  // uniform
            rk0[slot] = *pK;
            rk1[slot] = *pK1;
        } else {
            const double2 v2 = *reinterpret_cast<const double2*>(pK);
            rk0[slot] = v2.x;
            rk1[slot] = v2.y;
        }
        if (k < T - 2) { pK += sK_; pK1 += sK_; }  // uniform; no load inside the branch
    };
    block_load(0);
    __builtin_amdgcn_sched_barrier(0);
    UNR for (int q = 0; q < PF; q++) { fetch(q, q); __builtin_amdgcn_sched_barrier(0); }  // issue order matters (vmcnt)
    block_stage(0);
    block_load(S);
    lds_barrier();

    double dx0 = 0, dx1 = c1ok ? 0.0 : 1.0, dun = 0, pcA = 0, pcB = 0;

    // a coordinate's segment [xbar, x(1)] leaves [mn, mx] iff its larger end exceeds mx or its smaller end is below mn
    // (+-inf stand for weight 0); branch-free on purpose
    auto seg_bad = [&](double xa, double xb) -> bool {
        const double ea = xa + dx0, eb = xb + dx1;
        return (fmax(xa, ea) > mx0) | (fmin(xa, ea) < mn0) | (fmax(xb, eb) > mx1) | (fmin(xb, eb) < mn1);
    };
    // inspectJointLimit on one coordinate (System.cpp:163-179): q = limit - v on the violated side, cost q * penalty * q.
    // Branch-free: the distance beyond the limit is max(v - mx, 0) + max(mn - v, 0), and (-q) pen (-q) == q pen q bit for bit.
    auto limit_cost_of = [&](double v, double mx, double mn) -> double {
        const double q = fmax(v - mx, 0.0) + fmax(mn - v, 0.0);
        return q * pen * q;
    };
    // Limit cost of the stage for every alpha: all 8 rows of an instance hold the full state deviation, so row r8 evaluates
    // alpha_{r8} = 2^-r8 and alpha_{r8+8} for its two columns -- two evaluations instead of a loop over n_alpha in row 0.
    const double myA = ldexp(1.0, -r8), myB = ldexp(1.0, -(r8 + 8));
    auto limits_all = [&](double xb0, double xb1) {
        pcA += limit_cost_of(fma(myA, dx0, xb0), mx0, mn0) + limit_cost_of(fma(myA, dx1, xb1), mx1, mn1);
        if (NA > 8) pcB += limit_cost_of(fma(myB, dx0, xb0), mx0, mn0) + limit_cost_of(fma(myB, dx1, xb1), mx1, mn1);
    };
    double* kpdev = a.kpdev;

    // steps 0 .. T-2 are control steps, step T-1 is the terminal state, later steps of the last block are dummies
    const int nblocks = (T + S - 1) / S;
    for (int j = 0; j < nblocks; j++) {
        const int k0 = j * S, buf = j & 1;
        UNR for (int s = 0; s < S; s++) {
            const int k = k0 + s;
            fetch((s + PF) % S, k + PF);  // into the slot freed by the previous step
            if (k > T - 1) continue;      // uniform; dummy step (the fetch is issued, the work skipped)
            const double xb0 = sIn[buf][s * NR + c0][li], xb1 = sIn[buf][s * NR + c1c][li];
            if (k < T - 1) {
                const double ub = sIn[buf][s * NR + NX + r][li];
                const double du = quad_sum(fma(rk0[s], dx0, rk1[s] * dx1));
                const double du0 = bperm_f64(src0, du), du1 = bperm_f64(src1, du);
                if (colst) {
                    sOut[s * NR + c0][li] = xb0 + dx0;
                    if (c1ok) sOut[s * NR + c1][li] = xb1 + dx1;
                }
                if (rowst) sOut[s * NR + NX + r][li] = ub + du;
                if (f.early_stop) {  // ||du_k(1)||
                    const double n2 = quad_sum(fma(du0, du0, c1ok ? du1 * du1 : 0.0));
                    dun += __builtin_amdgcn_sqrt(n2);  // v_sqrt_f64: the sum only feeds the stop threshold
                }
                if (lim_on && __ballot((seg_bad(xb0, xb1) & inst_ok) ? 1 : 0) != 0ull) limits_all(xb0, xb1);
                if (k == kp_next) {  // uniform, rare: hand the deviation of this step to k_select
                    double* o = kpdev + (size_t)kpi * NR * Bp;
                    if (colst) {
                        AT(o, c0, bb) = dx0;
                        if (c1ok) AT(o, c1, bb) = dx1;
                    }
                    if (rowst) AT(o, NX + r, bb) = du;
                    kpi++;
                    kp_next = (kpi < n_kp) ? __builtin_amdgcn_readfirstlane(d.kp_t[kpi]) : -1;  // (scalar register: see k_forward_mfma)
                }
                // ---- deviation dynamics: dx' = dx + dt du
                dx0 = dx0 + dt * du0;
                dx1 = dx1 + dt1 * du1;
            } else {  // terminal state
                if (colst) {
                    sOut[s * NR + c0][li] = xb0 + dx0;
                    if (c1ok) sOut[s * NR + c1][li] = xb1 + dx1;
                }
                if (lim_on && __ballot((seg_bad(xb0, xb1) & inst_ok) ? 1 : 0) != 0ull) limits_all(xb0, xb1);
                if (kp_next == T - 1 && colst) {
                    double* o = kpdev + (size_t)kpi * NR * Bp;
                    AT(o, c0, bb) = dx0;
                    if (c1ok) AT(o, c1, bb) = dx1;
                }
            }
        }
        lds_barrier();            // block j computed by every wave: sOut complete, sIn[buf] free
        block_store(k0);
        block_stage(buf ^ 1);     // block j+1 (loaded while block j was computed)
        block_load(k0 + 2 * S);
        lds_barrier();            // sIn[buf^1] visible, sOut free
    }
    // ---- limit cost per alpha (sum over the 7 columns = the 4 lanes of a row; row r8 holds alpha_{r8}, alpha_{r8+8})
    {
        const double sA = quad_sum(pcA), sB = quad_sum(pcB);
        if (inst_ok && jl == 0) {
            if (r8 < NA) AT(a.lsc, r8, bb) = sA;
            if (NA > 8 && r8 + 8 < NA) AT(a.lsc, r8 + 8, bb) = sB;
        }
    }
    if (colst && jl == 0) a.dun[bb] = dun;
}


// ------------------------------------------------------------------------------------------------ small batches
// k_forward_wg is built for bandwidth: 32 lanes and an LDS round trip per instance-step, two workgroup barriers per 8 steps -- 0.47 us per
// step whatever the batch, 47 us of a 147 us iteration at B = 256 (BASELINE configs[1]).  At one wave per SIMD or less the rollout is a bare
// chain, so this version makes the chain short: 16 lanes per instance as in the sweep (ilqr_kernels_dpp.hip), no LDS, no workgroup.
//   lane (h, r), h = half of the 16-lane row, r = gain row / coordinate:  half 0 holds {K_r0 .. K_r3} and forms K_r0 dx_0 + .. + K_r3 dx_3, half 1 holds
//   {K_r4, K_r5, K_r6, d_r} and forms K_r4 dx_4 + .. + d_r; dx_c comes straight out of lane c's register as the `row_newbcast` operand of
//   `v_fmac_f64_dpp`; the halves meet by one `row_ror:8` move, so both hold du_r and both advance dx_r += dt du_r.
//   Memory: a wave may have 64 vector-memory instructions in flight, so what bounds the chain is instructions per step x latency / 64 (measured with
//   8 per step: 0.23 us per step at any ring depth).  Hence 4 per step: the gain row as two 16-byte loads per lane (the halves take the two halves of
//   the row), xbar_r | ubar_r as ONE load (half 0 / half 1), x(1)_r | u(1)_r as ONE store.  Ring of PF = 14 steps; a slot is refilled after its last
//   use, so the loop-carried value and the load share a register (see ring_take in ilqr_kernels_dpp.hip for what happens otherwise).
// Outputs as k_forward_wg: x(1), u(1) into the other buffer, (dx, du) at the keypoint steps, the limit cost of every step size, sum ||du||.
// (Measured and dropped: this kernel, the decision and the next sweep's keypoint derivatives as ONE launch, a wave carrying its four instances through the
// three phases -- bit-identical, and no faster at any batch size (C2, B = 256: 2.472 against 2.463 ms per solve): what the two launches cost is made up
// by k_select's two waves per instance group, which the one-wave form cannot have.)
#define FD_LO " row_mask:0xf bank_mask:0x3"
#define FD_HI " row_mask:0xf bank_mask:0xc"
// this half's part of du_r = d_r + sum_c K_rc dx_c (see above); three accumulators in rotation: a DPP instruction reads its accumulator early, no
// register is touched again within two instructions
__device__ __forceinline__ double fd_half_dot(const double (&K)[4], double x, double one) {
    double s0, s1, s2;
#define D_(A, J, C, M, X) "v_fmac_f64_dpp %[" A "], %[" X "], %[k" #J "] row_newbcast:" #C M "\n\t"
    asm volatile("v_mov_b64 %[s0], 0\n\tv_mov_b64 %[s1], 0\n\tv_mov_b64 %[s2], 0\n\ts_nop 1\n\t"
                 D_("s0", 0, 0, FD_LO, "x") D_("s1", 1, 1, FD_LO, "x") D_("s2", 2, 2, FD_LO, "x") D_("s0", 0, 4, FD_HI, "x") D_("s1", 1, 5, FD_HI, "x") D_("s2", 2, 6, FD_HI, "x")
                 D_("s0", 3, 3, FD_LO, "x") D_("s1", 3, 0, FD_HI, "one") "s_nop 0"
                 : [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2)
                 : [x] "v"(x), [one] "v"(one), [k0] "v"(K[0]), [k1] "v"(K[1]), [k2] "v"(K[2]), [k3] "v"(K[3]));
#undef D_
    return (s0 + s1) + s2;
}
// sum over lanes 8m .. 8m+7, result in all eight (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror)
__device__ __forceinline__ double fd_oct_sum(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    return v;
}

// LIM / ES: limits set / early stop asked for -- template parameters because a branch instruction costs a lone wave 20-30 clocks whether taken or not
// (probe build/exp/lat.hip: the 22-instruction chain of a step takes 89 clocks, the first version of this loop 530 with its six branches per step).
// For the same reason a group of PF steps that holds no keypoint step and does not reach the end of the horizon runs a copy of the body without
// those two tests.
template <int NA, bool LIM, bool ES>
__device__ __forceinline__ bool forward_dpp_body(const Bufs& a, const FwdArgs& f) {  // false: no running instance in this wave
    constexpr int NX = 7, NU = 7, ROWP = kd_rowp(NX), RS = NU * ROWP, PF = 8;
    static_assert(ROWP == 8, "record row = 7 gains + feed-forward");
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15, h = l >> 3, r8 = l & 7;
    const int b = xcd_tile() * 4 + g;
    const int Bp = d.Bp, T = d.T, B = d.B;
    const bool inst_ok = (b < B) && (a.active[b < B ? b : 0] != 0);
    if (__ballot(inst_ok ? 1 : 0) == 0ull) return false;  // wave-uniform
    const int bb = (b < B) ? b : 0;
    const bool row = r8 < NX;              // lanes (h, 7) shadow row 6: never store, never the source of a broadcast
    const int r = row ? r8 : NX - 1;
    const bool st = inst_ok && row;        // this lane stores (x in half 0, u in half 1)
    const bool sx = st && h == 0;          // ... and holds a state coordinate
    const int cur = a.cur[bb];
    const double dt = d.dt, pen = d.penalty, one = 1.0;
    const int lw = LIM ? d.lw[r] : 0;
    const double mx = lw ? d.smax[r] : INFINITY, mn = lw ? d.smin[r] : -INFINITY;
    const int n_kp = d.n_kp;
    int kpi = 0, kp_next = (n_kp > 0) ? d.kp_t[0] : -1;

    const size_t Vstep = (size_t)NX * Bp;
    const double* pV = (h ? a.U[cur] : a.X[cur]) + (size_t)r * Bp + bb;      // xbar_r (half 0) | ubar_r (half 1): same stride
    // gains: a wave-uniform base that walks the horizon (scalar registers) + the lane's 32-bit byte offsets into the step's records -- no vector address
    // arithmetic per load.  Plain record: this half's 32 bytes of the row (two 16-byte loads); packed symmetric record (a.kd_sym, ilqr_kernels.hpp): four 8-byte loads
    const int sym = a.kd_sym, rs_ = kd_rs(sym, NU, ROWP);
    const double* kb = a.KD;
    unsigned ko[4];
    UNR for (int j = 0; j < 4; j++) ko[j] = ((unsigned)bb * rs_ + (sym ? kd_sym_off(r, 4 * h + j) : r * ROWP + 4 * h + j)) * 8u;
    const size_t Kstep = (size_t)Bp * rs_;
    double* qV = (h ? a.U[1 - cur] : a.X[1 - cur]) + (size_t)r * Bp + bb;
    const double xT = a.X[cur][((size_t)(T - 1) * NX + r) * Bp + bb];         // terminal xbar

    double Kr[PF][4], vr[PF];
    auto fetch = [&](int slot, int kk) {  // unconditional; the pointers stop at the last control step
        const char* kbb = reinterpret_cast<const char*>(kb);
        if (sym) {  // uniform
            UNR for (int j = 0; j < 4; j++) Kr[slot][j] = *reinterpret_cast<const double*>(kbb + ko[j]);
        } else {
            const double2 v0 = *reinterpret_cast<const double2*>(kbb + ko[0]), v1 = *reinterpret_cast<const double2*>(kbb + ko[2]);
            Kr[slot][0] = v0.x; Kr[slot][1] = v0.y; Kr[slot][2] = v1.x; Kr[slot][3] = v1.y;
        }
        vr[slot] = *pV;
        const bool more = kk < T - 2;  // uniform: a scalar select, not a branch
        kb += more ? Kstep : 0; pV += more ? Vstep : 0;
    };
    UNR for (int q = 0; q < PF; q++) { fetch(q, q); __builtin_amdgcn_sched_barrier(0); }

    double dx = 0, dun = 0, pc[NA];
    UNR for (int i = 0; i < NA; i++) pc[i] = 0;
    // inspectJointLimit on one coordinate (System.cpp:163-179), branch-free as in k_forward_wg
    auto limit_cost_of = [&](double v) -> double {
        const double q = fmax(v - mx, 0.0) + fmax(mn - v, 0.0);
        return q * pen * q;
    };
    auto limits_all = [&](double xb) {  // the stage's limit cost of this coordinate for every step size
        UNR for (int i = 0; i < NA; i++) pc[i] += limit_cost_of(fma(ldexp(1.0, -i), dx, xb));
    };
    auto seg_bad = [&](double xb) -> bool {  // the segment [xbar, x(1)] of this coordinate leaves [mn, mx]
        const double e = xb + dx;
        return (fmax(xb, e) > mx) | (fmin(xb, e) < mn);
    };
    double* kpdev = a.kpdev;

#define FD_STEP_(JJ, CHK)                                                                                          \
    {                                                                                                              \
        const int k = k0 + JJ;                                                                                     \
        if (!CHK || k < T - 1) {                                                                                   \
            const double vb = vr[JJ];                        /* xbar_r | ubar_r */                                 \
            const double part = fd_half_dot(Kr[JJ], dx, one);                                                      \
            const double du = part + dpp_f64<0x128>(part);   /* row_ror:8: the other half's part */                \
            if (st) *qV = vb + (h ? du : dx);                                                                      \
            qV += Vstep;                                                                                           \
            if (ES) dun += __builtin_amdgcn_sqrt(fd_oct_sum(row ? du * du : 0.0));   /* ||du_k(1)|| */             \
            if (LIM && __builtin_expect(__ballot((seg_bad(vb) & sx) ? 1 : 0) != 0ull, 0)) limits_all(vb);          \
            if (CHK && k == kp_next) {   /* hand the deviation of this step to k_select */                         \
                double* o = kpdev + (size_t)kpi * (NX + NU) * Bp;                                                  \
                if (st) AT(o, h * NX + r, bb) = h ? du : dx;                                                       \
                kpi++;                                                                                             \
                kp_next = (kpi < n_kp) ? __builtin_amdgcn_readfirstlane(d.kp_t[kpi]) : -1;                         \
            }                                                                                                      \
            dx = dx + dt * du;   /* deviation dynamics (both halves) */                                            \
        }                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);   /* the slot is free: only now its next load */                        \
        fetch(JJ, k + PF);                                                                                         \
    }
    for (int k0 = 0; k0 < T - 1; k0 += PF) {
        const bool plain = (k0 + PF <= T - 1) && (kp_next < 0 || kp_next >= k0 + PF);  // uniform
        if (plain) {
            FD_STEP_(0, false) FD_STEP_(1, false) FD_STEP_(2, false) FD_STEP_(3, false) FD_STEP_(4, false) FD_STEP_(5, false) FD_STEP_(6, false) FD_STEP_(7, false)
        } else {
            FD_STEP_(0, true) FD_STEP_(1, true) FD_STEP_(2, true) FD_STEP_(3, true) FD_STEP_(4, true) FD_STEP_(5, true) FD_STEP_(6, true) FD_STEP_(7, true)
        }
    }
#undef FD_STEP_
    static_assert(PF == 8, "eight step copies above");
    {   // terminal state
        if (sx) *qV = xT + dx;
        if (LIM && __ballot((seg_bad(xT) & sx) ? 1 : 0) != 0ull) limits_all(xT);
        if (kp_next == T - 1 && sx) {
            double* o = kpdev + (size_t)kpi * (NX + NU) * Bp;
            AT(o, r, bb) = dx;
        }
    }
    // limit cost per step size: the seven coordinates of an instance (half 0), lane 0 writes
    UNR for (int i = 0; i < NA; i++) {
        const double sA = LIM ? fd_oct_sum(row ? pc[i] : 0.0) : 0.0;
        if (inst_ok && l == 0) AT(a.lsc, i, bb) = sA;
    }
    if (inst_ok && l == 0) a.dun[bb] = dun;
    return true;
}
template <int NA, bool LIM, bool ES>
__global__ __launch_bounds__(64) void k_forward_dpp(Bufs a, FwdArgs f) { (void)forward_dpp_body<NA, LIM, ES>(a, f); }
#undef FD_LO
#undef FD_HI

// The decision of k_select once every lane holds the task cost of its step size: limit cost added, the first step size (descending) below the current
// cost wins, else the last one tried (ILQRRecursive.cpp:101-155); bookkeeping by the instance's first lane.  Returns the winner's index (uniform over
// the instance's 16 lanes); *stopped = the instance left the iteration (early stop).
template <int NA>
__device__ __forceinline__ int select_decide(const Bufs& a, const FwdArgs& f, const DevDesc& d, int bb, bool inst_ok, bool mine, int gi, int al, int n_alpha, double c,
                                             bool* stopped = nullptr) {
    const int Bp = d.Bp;
    if (mine) c += AT(a.lsc, al, bb);
    const double cost0 = a.cost[bb];
    const bool okc = mine && !((c >= cost0) || isnan(c));
    const unsigned m16 = (unsigned)((__ballot(okc ? 1 : 0) >> (gi * 16)) & 0xffffull);
    const int w = m16 ? (__ffs(m16) - 1) : (n_alpha - 1);
    const double wcost = __shfl(c, gi * 16 + w);
    bool stop = false;
    if (inst_ok && al == 0) {
        const double walpha = ldexp(1.0, -w);
        a.cost[bb] = wcost;
        a.alpha[bb] = walpha;
        a.iters[bb] = f.it + 1;
        a.status[bb] = (isfinite(wcost) ? 0 : 1) | ((walpha <= d.alpha_floor) ? 2 : 0);
        if (a.cost_trace) {
            a.cost_trace[(size_t)f.it * Bp + bb] = wcost;
            a.alpha_trace[(size_t)f.it * Bp + bb] = walpha;
        }
        a.pend[bb] = w + 1;  // k_blend / k_flip finish the acceptance (w == 0: the buffer already holds x(1), u(1))
        a.pred[bb] = w;
        stop = f.early_stop && (walpha * sqrt(walpha * a.dun[bb]) < d.stop_tol);  // sum ||du(alpha)|| = alpha sum ||du(1)||
        if (!f.al) stop = stop && (wcost < 1e-3);  // ILQRRecursive.cpp:174 vs AL-ILQR.cpp:225
        if (stop) a.active[bb] = 0;
    }
    if (stopped) *stopped = __shfl(stop ? 1 : 0, gi * 16) != 0;
    return w;
}

// Line-search decision of iteration f.it, one lane per (instance, alpha): task cost of x(alpha) = xbar + alpha dx at the keypoint
// steps + the limit cost k_forward_w32 accumulated; the first alpha (descending) whose cost is below the current one wins, else
// the last one tried (ILQRRecursive.cpp:101-155).  Writes cost/alpha/iters/status/traces, `pend` for k_blend/k_flip, and the
// early-stop flag.
// KW = 2: the keypoints are dealt to the two waves of the workgroup (keypoint kpi to wave kpi mod 2) -- the kernel is one chain of dependent
// instructions per keypoint (FK: seven sincos and rotation products) on 1024 waves, one per SIMD, so a second wave per SIMD hides half of
// it.  The per-keypoint costs meet in LDS and wave 0 adds them in keypoint order: the bits of the one-wave sum.
template <class S, int NA, bool EXT, int KW>
__global__ __launch_bounds__(64 * KW) void k_select(Bufs a, FwdArgs f) {
    constexpr int NX = S::NX, NU = S::NU;
    static_assert(NA <= 16, "16 lanes per instance");
    static_assert(KW == 1 || KW == 2, "one or two waves per four instances");
    __shared__ double sc[KW == 2 ? MAX_KP : 1][64];
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, gi = lane >> 4, al = lane & 15;
    const int b = blockIdx.x * 4 + gi;
    const int Bp = d.Bp, T = d.T, B = d.B;
    const bool inst_ok = (b < B) && (a.active[b < B ? b : 0] != 0);
    const int bb = (b < B) ? b : 0;
    const int n_alpha = f.n_alpha < NA ? f.n_alpha : NA;
    const bool mine = inst_ok && al < n_alpha;
    const double aa = ldexp(1.0, -al);
    double c = 0;
    if (mine) {
        const double* Xb = a.X[a.cur[bb]];
        const double* Ub = a.U[a.cur[bb]];
        for (int kpi = wv; kpi < d.n_kp; kpi += KW) {
            const int k = d.kp_t[kpi];
            const double* dv = a.kpdev + (size_t)kpi * (NX + NU) * Bp;
            double xt[NX], ut[NU], tg[S::NF];
            UNR for (int i = 0; i < NX; i++) xt[i] = fma(aa, AT(dv, i, bb), AT(Xb, k * NX + i, bb));
            UNR for (int i = 0; i < NU; i++) ut[i] = (k < T - 1) ? fma(aa, AT(dv, NX + i, bb), AT(Ub, k * NU + i, bb)) : 0.0;
            UNR for (int i = 0; i < S::NF; i++) tg[i] = AT(a.kp_tg, kpi * S::NF + i, bb);
            const double ck = kp_cost<S, EXT>(d, kpi, tg, xt, ut);
            if (KW == 1) c += ck;
            else sc[kpi][lane] = ck;
        }
    }
    if (KW == 2) {
        __syncthreads();
        if (wv != 0) return;
        if (mine) { for (int kpi = 0; kpi < d.n_kp; kpi++) c += sc[kpi][lane]; }
    }
    (void)select_decide<NA>(a, f, d, bb, inst_ok, mine, gi, al, n_alpha, c);
}

// Accepted trajectory + AL bookkeeping in one pass (k_blend and k_al_post each re-read the trajectory): one thread per
// (timestep, instance), a block = 16 timesteps x 16 instances (16 instances of a row = one 128-byte line).
//   x(alpha) = xbar + alpha (x(1) - xbar) written over x(1) unless alpha = 1; on that accepted (x, u) the AL bookkeeping of
//   AL-ILQR.cpp:190,202-208: I_k = penalty (g<0 && lambda==0 ? 0 : 1) with the multipliers BEFORE the update,
//   lambda_k = max(0, lambda_k + penalty' g) on update iterations.  k_flip then swaps the buffers of the instances that ran.
template <class S>
__global__ __launch_bounds__(256) void k_apply(Bufs a, FwdArgs f) {
    constexpr int NX = S::NX, NU = S::NU;
    const DevDesc& d = *a.desc;
    const int tid = threadIdx.x, inst = tid & 15, hi = tid >> 4;
    const int b = blockIdx.x * 16 + inst, k = blockIdx.y * 16 + hi;
    const int Bp = d.Bp, T = d.T;
    if (b >= d.B || k >= T) return;
    const int w = a.pend[b] - 1;  // winner index of this iteration (k_select), -1: the instance did not run
    if (w < 0) return;
    const int cur = a.cur[b];
    const double aa = ldexp(1.0, -w);
    const double* Xb = a.X[cur];
    const double* Ub = a.U[cur];
    double* Xn = a.X[1 - cur];
    double* Un = a.U[1 - cur];
    const int m = f.al ? a.m : 0;
    if (w == 0 && (m == 0 || k == T - 1)) return;  // nothing to blend, no constraint rows at this step
    double x[NX], u[NU];
    UNR for (int i = 0; i < NX; i++) x[i] = AT(Xn, k * NX + i, b);
    if (k < T - 1) { UNR for (int i = 0; i < NU; i++) u[i] = AT(Un, k * NU + i, b); }
    if (w > 0) {
        double xb[NX], ub[NU];
        UNR for (int i = 0; i < NX; i++) xb[i] = AT(Xb, k * NX + i, b);
        if (k < T - 1) { UNR for (int i = 0; i < NU; i++) ub[i] = AT(Ub, k * NU + i, b); }
        UNR for (int i = 0; i < NX; i++) {
            x[i] = fma(aa, x[i] - xb[i], xb[i]);
            AT(Xn, k * NX + i, b) = x[i];
        }
        if (k < T - 1) {
            UNR for (int i = 0; i < NU; i++) {
                u[i] = fma(aa, u[i] - ub[i], ub[i]);
                AT(Un, k * NU + i, b) = u[i];
            }
        }
    }
    if (k < T - 1) {
        for (int r = 0; r < m; r++) {
            const double g = con_g<S>(a, k, r, x, u);
            const double lam = AT(a.lambda, k * m + r, b);
            AT(a.Is, k * m + r, b) = f.penalty_roll * ((g < 0 && lam == 0) ? 0.0 : 1.0);
            if (f.do_update) {
                const double v = lam + f.penalty_update * g;
                AT(a.lambda, k * m + r, b) = v > 0 ? v : 0;
            }
        }
    }
}
__global__ void k_flip_ran(Bufs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.desc->B) return;
    if (a.pend[b] > 0) {
        a.cur[b] = 1 - a.cur[b];
        a.pend[b] = 0;
    }
}

// ------------------------------------------------------------------------------------------------ initial rollout (PosOrn)
// k_init_rollout (ilqr_kernels.hip) walks the horizon with one lane per instance: 64 waves for B = 4096, 0.85 ms -- as much as
// 1.3 iterations.  For PosOrn systems the coordinates integrate independently, so the rollout is one lane per (instance,
// coordinate): q' = q + dt dq (+ dt^2/2 ddq), the limit cost of that coordinate summed along the way; k_init_finish adds the
// task cost at the keypoints (one lane per instance) and sets the bookkeeping; the AL weights come from k_al_post.
template <class S>
__global__ __launch_bounds__(256) void k_init_roll_lti(Bufs a) {
    // One lane per (instance, coordinate): joint i (position and, 2nd order, velocity); for the time systems one more lane per instance
    // carries the time state, and every lane reads the step's time control (dt = u_last^2, PosOrnTimePlannerSys.cpp:154-155).
    constexpr int NX = S::NX, NU = S::NU, ND = S::ND, TM = S::TM, CH = 8;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (b >= d.B) return;
    const int Bp = d.Bp, T = d.T;
    const bool isT = TM && i == DOF;   // the time-state lane
    const int ij = isT ? 0 : i;        // joint index (clamped for the time lane)
    const int xrow = isT ? NX - 1 : i, urow = isT ? NU - 1 : i;
    const double pen = d.penalty;
    const bool lim = d.limits_set != 0 && !isT;
    const double mxq = (lim && d.lw[ij]) ? d.smax[ij] : INFINITY, mnq = (lim && d.lw[ij]) ? d.smin[ij] : -INFINITY;
    const double mxv = (lim && ND == 2 && d.lw[DOF + ij]) ? d.smax[DOF + ij] : INFINITY, mnv = (lim && ND == 2 && d.lw[DOF + ij]) ? d.smin[DOF + ij] : -INFINITY;
    auto lcost = [&](double v, double mx, double mn) { const double q = fmax(v - mx, 0.0) + fmax(mn - v, 0.0); return q * pen * q; };
    double q = isT ? 0.0 : AT(a.q0, ij, b), v = (ND == 2 && !isT) ? AT(a.dq0, ij, b) : 0.0, cost = 0;
    const double* __restrict__ U0 = a.U0;
    double* __restrict__ X = a.X[0];
    double* __restrict__ U = a.U[0];
    for (int k0 = 0; k0 < T - 1; k0 += CH) {
        double u[CH], us[CH];
        UNR for (int j = 0; j < CH; j++) {
            const int kk = (k0 + j < T - 1 ? k0 + j : T - 2);
            u[j] = AT(U0, kk * NU + urow, b);
            us[j] = TM ? AT(U0, kk * NU + NU - 1, b) : 0.0;
        }
        UNR for (int j = 0; j < CH; j++) {
            const int k = k0 + j;
            if (k >= T - 1) break;
            const double dt = TM ? us[j] * us[j] : d.dt;
            AT(X, k * NX + xrow, b) = q;
            if (ND == 2 && !isT) AT(X, k * NX + DOF + ij, b) = v;
            AT(U, k * NU + urow, b) = u[j];
            cost += lcost(q, mxq, mnq);
            if (ND == 2) cost += lcost(v, mxv, mnv);
            if (isT) {
                q = q + dt;                                 // dyn_step, same expressions
            } else if (ND == 1) {
                q = q + (dt * u[j] + dt * dt / 2 * 0.0);
            } else {
                q = q + (dt * v + dt * dt / 2 * u[j]);
                v = v + dt * u[j];
            }
        }
    }
    AT(X, (T - 1) * NX + xrow, b) = q;
    if (ND == 2 && !isT) AT(X, (T - 1) * NX + DOF + ij, b) = v;
    cost += lcost(q, mxq, mnq);
    if (ND == 2) cost += lcost(v, mxv, mnv);
    if (!isT) AT(a.lsc, ij, b) = cost;  // scratch: limit cost of coordinate i over the horizon
}

template <class S>
__global__ __launch_bounds__(64) void k_init_finish(Bufs a) {
    constexpr int NX = S::NX, NU = S::NU;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= d.B) return;
    const int Bp = d.Bp, T = d.T;
    double cost = 0;
    for (int kpi = 0; kpi < d.n_kp; kpi++) {
        const int k = d.kp_t[kpi];
        double x[NX], u[NU], tg[S::NF];
        UNR for (int i = 0; i < NX; i++) x[i] = AT(a.X[0], k * NX + i, b);
        UNR for (int i = 0; i < NU; i++) u[i] = (k < T - 1) ? AT(a.U[0], k * NU + i, b) : 0.0;
        UNR for (int i = 0; i < S::NF; i++) tg[i] = AT(a.kp_tg, kpi * S::NF + i, b);
        cost += kp_cost<S>(d, kpi, tg, x, u);
    }
    if (d.limits_set) { UNR for (int i = 0; i < DOF; i++) cost += AT(a.lsc, i, b); }
    a.cost[b] = cost;
    a.alpha[b] = 1.0;
    a.cur[b] = 0;
    a.active[b] = 1;
    a.iters[b] = 0;
    a.pend[b] = 0;
    a.pred[b] = 0;
    a.status[b] = isfinite(cost) ? 0 : 1;
}

bool init_lti_supported(int kind, int nd) {  // every system: the coordinates integrate independently given the step's dt
    return ((kind == 0 || kind == 1) && (nd == 1 || nd == 2)) || ((kind == 2 || kind == 3) && nd == 1);
}

template <class S>
static void launch_init_lti_sys(const Bufs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL((k_init_roll_lti<S>), dim3((B + 255) / 256, DOF + S::TM), dim3(256), 0, st, a);
    hipLaunchKernelGGL((k_init_finish<S>), dim3((B + 63) / 64), dim3(64), 0, st, a);
}
void launch_init_lti(int kind, int nd, const Bufs& a, int B, hipStream_t st) {
    if (kind == 2) launch_init_lti_sys<Sys<2, 1>>(a, B, st);
    else if (kind == 3) launch_init_lti_sys<Sys<3, 1>>(a, B, st);
    else if (kind == 1 && nd == 1) launch_init_lti_sys<Sys<1, 1>>(a, B, st);
    else if (kind == 1) launch_init_lti_sys<Sys<1, 2>>(a, B, st);
    else if (nd == 1) launch_init_lti_sys<Sys<0, 1>>(a, B, st);
    else launch_init_lti_sys<Sys<0, 2>>(a, B, st);
}

bool forward_wave_supported(int kind, int nd, int n_alpha) {
    return (kind == 0 || kind == 2) && nd == 1 && n_alpha <= 16;
}

template <class S>
static void launch_apply_wave_sys(const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f) {
    hipLaunchKernelGGL((k_apply<S>), dim3((B + 15) / 16, (T + 15) / 16), dim3(256), 0, st, a, f);
    hipLaunchKernelGGL(k_flip_ran, dim3((B + 255) / 256), dim3(256), 0, st, a);
}
void launch_apply_wave(int kind, const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f) {
    if (kind == 2) launch_apply_wave_sys<Sys<2, 1>>(a, B, T, st, f);
    else launch_apply_wave_sys<Sys<0, 1>>(a, B, T, st, f);
}

template <class S, int NA>
static void launch_select(const Bufs& a, int B, hipStream_t st, const FwdArgs& f) {
    const dim3 sgrid((B + 3) / 4);
    if (f.n_kp >= 2) {  // two waves share the keypoints of four instances
        if (f.kp_ext) hipLaunchKernelGGL((k_select<S, NA, true, 2>), sgrid, dim3(128), 0, st, a, f);
        else hipLaunchKernelGGL((k_select<S, NA, false, 2>), sgrid, dim3(128), 0, st, a, f);
    } else {
        if (f.kp_ext) hipLaunchKernelGGL((k_select<S, NA, true, 1>), sgrid, dim3(64), 0, st, a, f);
        else hipLaunchKernelGGL((k_select<S, NA, false, 1>), sgrid, dim3(64), 0, st, a, f);
    }
}
template <int NA>
static void launch_forward_dpp(const Bufs& a, dim3 grid, hipStream_t st, const FwdArgs& f) {
    const dim3 block(64);
    if (f.limits) {
        if (f.early_stop) hipLaunchKernelGGL((k_forward_dpp<NA, true, true>), grid, block, 0, st, a, f);
        else hipLaunchKernelGGL((k_forward_dpp<NA, true, false>), grid, block, 0, st, a, f);
    } else {
        if (f.early_stop) hipLaunchKernelGGL((k_forward_dpp<NA, false, true>), grid, block, 0, st, a, f);
        else hipLaunchKernelGGL((k_forward_dpp<NA, false, false>), grid, block, 0, st, a, f);
    }
}
// the rollout itself knows no keypoint function (single-integrator dynamics); the decision kernel is per system kind
template <class S>
static void launch_forward_wave_sys(const Bufs& a, int B, hipStream_t st, const FwdArgs& f) {
    const dim3 grid(grid_x8((B + FW_IW - 1) / FW_IW)), block(FW_IW * 32);
    const dim3 sgrid(grid_x8((B + 3) / 4));  // small batches: 16 lanes per instance, one wave per workgroup (k_forward_dpp)
    if (f.n_alpha <= 1) {
        if (f.small) launch_forward_dpp<1>(a, sgrid, st, f);
        else hipLaunchKernelGGL((k_forward_wg<1>), grid, block, 0, st, a, f);
        launch_select<S, 1>(a, B, st, f);
    } else if (f.n_alpha <= 11) {
        if (f.small) launch_forward_dpp<11>(a, sgrid, st, f);
        else hipLaunchKernelGGL((k_forward_wg<11>), grid, block, 0, st, a, f);
        launch_select<S, 11>(a, B, st, f);
    } else {
        if (f.small) launch_forward_dpp<16>(a, sgrid, st, f);
        else hipLaunchKernelGGL((k_forward_wg<16>), grid, block, 0, st, a, f);
        launch_select<S, 16>(a, B, st, f);
    }
}
void launch_forward_wave(int kind, const Bufs& a, int B, hipStream_t st, const FwdArgs& f) {
    if (kind == 2) launch_forward_wave_sys<Sys<2, 1>>(a, B, st, f);
    else launch_forward_wave_sys<Sys<0, 1>>(a, B, st, f);
}

}  // namespace ilqr
