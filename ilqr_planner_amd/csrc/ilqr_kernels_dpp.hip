// ilqr_kernels_dpp.hip -- closed-form backward Riccati sweep with the matrices in REGISTERS, rows broadcast by DPP (gfx950, fp64).
//
// Same mathematics as the cooperative sweep it replaces (reference: ILQRRecursive.cpp:68-97, AL-ILQR.cpp:110-145) for
// single-integrator dynamics (A = I, B = dt I):  S = Quu + reg I = D + dt^2 P,  M = S^-1,  N = M D - I,
//   K = N / dt,   d = -M Qu,   P' = l_xx - (D N + reg N'N) / dt^2,   p' = l_x + p - (Qu + D d)/dt - reg (D M d - d)/dt
// (exact consequences of (Quu + reg I) K = -Qux with the UN-regularised Quu in the value update, quirk D-3:
//  D - D M D = -D N and D M^2 D - D M - M D + I = N'N).
//
// Mapping.  The cooperative kernel gave every entry of the symmetric matrices a lane and moved the pivot column / row through LDS:
// one FMA per lane and pivot behind an LDS round trip.  Here LPI = 8 lanes own an instance (8 instances per wave), lane r of the
// group holds ROW r of every matrix (7 doubles; row 7 is padding and stays zero).  A pivot is
//     v_mov_b64_dpp  acc, s[c]  row_newbcast:c        the pivot element to all lanes of the DPP row
//     rc = 1/acc                                       (rcp + one cubic correction)
//     v_fmac_f64_dpp s[j], s[j], -t  row_newbcast:c    s[j] -= t * (pivot row)[j], t = s[c] rc, for the six j != c
// i.e. the broadcast of the pivot row is the DPP operand of the FMA itself: no LDS, no separate move, no wait.  A DPP row is 16
// lanes = two instances, so every broadcast instruction is issued twice, once per half with a bank mask (lanes 0-7 take lane c,
// lanes 8-15 take lane 8 + c); everything that is not a broadcast serves all eight instances with one instruction.  (LPI = 16 --
// one instance per DPP row, the second half a redundant copy -- needs one broadcast instruction but twice the waves; measured, see
// DESIGN.md.)  The products with M (N'N, M Qu, M d) are rows times broadcast rows in the same way.  No LDS at all; no barrier.
//
// Deferred row scaling.  The sweep operator turns the pivot row into a_cj / a_cc.  Scaling that one row would cost six multiplications
// executed for a single lane; instead row c keeps its values and remembers the factor (myrc): later pivots act on a row by
// row_i -= (a_ic rc') row_c', which commutes with a row scaling, so the stored row evolves exactly as the scaled one would, divided by
// its factor; the pivot element is stored as -1 and the multiplier of the pivot lane is 0 (the lane constants nz / nm1).  After the
// seven pivots the true row is myrc * stored = -(S^-1)_r.
//
// Hazards.  The compiler's hazard recogniser does not look inside inline assembly, so each pivot is ONE hand-ordered block: a DPP
// read of a VGPR needs two wait states after the VALU write (the block ends with two plain instructions after its last broadcast
// FMA, and inside it no instruction reads as DPP source what one of the two before it wrote); a transcendental result (v_rcp_f64)
// needs one (an independent multiplication follows the rcp).  Blocks whose DPP sources come straight from compiler-scheduled code
// start with s_nop 1.
//
// Addressing.  x / u live in the two halves of ONE allocation each (the problem allocates the double buffers back to back, with
// equal strides for X and U), so a 32-bit byte offset per lane and buffer serves every load and store of the loop with the
// uniform base in scalar registers; the host refuses batches whose arrays pass 4 GiB (they take the generic sweep).
//
// FUSED / AL bookkeeping exactly as before: the acceptance of the previous line search rides in the load path,
// x = xbar + alpha (x(1) - xbar) with k_apply's expression, I_k from the multipliers before the update, lambda update on update
// iterations, buffers flipped at the end (AL-ILQR.cpp:190, 202-208).
#include "ilqr_kernels.hpp"
#include "ilqr_step.hpp"

namespace ilqr {

namespace {

#define DPPM_ALL " row_mask:0xf bank_mask:0xf"
#define DPPM_LO " row_mask:0xf bank_mask:0x3"
#define DPPM_HI " row_mask:0xf bank_mask:0xc"

// A DPP instruction reads its VGPR operands -- the broadcast source AND, for the lanes a bank mask leaves out, the old destination --
// ahead of the normal operand fetch: two wait states after any VALU write of one of them (found the hard way: a broadcast FMA
// right behind the `v_mov` that zeroed its accumulator gave wrong sums).  So every multi-instruction DPP sequence below is ONE asm
// statement that starts with `s_nop 1` (the compiler schedules freely around asm statements, never inside one), and inside a
// statement no instruction touches a register written by one of the two before it.

// acc[j] += sum_k bcast_k(src[j]) * mul[k]  for the seven j: row r of (lane rows of mul) x (matrix whose rows are the lanes' src)
template <int LPI>
__device__ __forceinline__ void fmac_rows_all(double (&acc)[7], const double (&src)[7], const double (&mul)[7]) {
#define R_(K, LL, MASK)                                                                                                                   \
    "v_fmac_f64_dpp %[a0], %[s0], %[m" #K "] row_newbcast:" LL MASK "\n\tv_fmac_f64_dpp %[a1], %[s1], %[m" #K "] row_newbcast:" LL MASK "\n\t" \
    "v_fmac_f64_dpp %[a2], %[s2], %[m" #K "] row_newbcast:" LL MASK "\n\tv_fmac_f64_dpp %[a3], %[s3], %[m" #K "] row_newbcast:" LL MASK "\n\t" \
    "v_fmac_f64_dpp %[a4], %[s4], %[m" #K "] row_newbcast:" LL MASK "\n\tv_fmac_f64_dpp %[a5], %[s5], %[m" #K "] row_newbcast:" LL MASK "\n\t" \
    "v_fmac_f64_dpp %[a6], %[s6], %[m" #K "] row_newbcast:" LL MASK "\n\t"
#define OPS_                                                                                                                              \
    : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]), [a5] "+v"(acc[5]), [a6] "+v"(acc[6])    \
    : [s0] "v"(src[0]), [s1] "v"(src[1]), [s2] "v"(src[2]), [s3] "v"(src[3]), [s4] "v"(src[4]), [s5] "v"(src[5]), [s6] "v"(src[6]),          \
      [m0] "v"(mul[0]), [m1] "v"(mul[1]), [m2] "v"(mul[2]), [m3] "v"(mul[3]), [m4] "v"(mul[4]), [m5] "v"(mul[5]), [m6] "v"(mul[6])
    if (LPI == 16)
        asm volatile("s_nop 1\n\t" R_(0, "0", DPPM_ALL) R_(1, "1", DPPM_ALL) R_(2, "2", DPPM_ALL) R_(3, "3", DPPM_ALL) R_(4, "4", DPPM_ALL) R_(5, "5", DPPM_ALL)
                         R_(6, "6", DPPM_ALL) "s_nop 0" OPS_);
    else
        asm volatile("s_nop 1\n\t" R_(0, "0", DPPM_LO) R_(0, "8", DPPM_HI) R_(1, "1", DPPM_LO) R_(1, "9", DPPM_HI) R_(2, "2", DPPM_LO) R_(2, "10", DPPM_HI)
                         R_(3, "3", DPPM_LO) R_(3, "11", DPPM_HI) R_(4, "4", DPPM_LO) R_(4, "12", DPPM_HI) R_(5, "5", DPPM_LO) R_(5, "13", DPPM_HI)
                             R_(6, "6", DPPM_LO) R_(6, "14", DPPM_HI) "s_nop 0" OPS_);
#undef R_
#undef OPS_
}
// sum_k bcast_k(src) * M[k]: three accumulators in rotation (no register is touched again within two instructions), zeroed inside
template <int LPI>
__device__ __forceinline__ double row_dot_bc(const double (&M)[7], double src) {
    double s0, s1, s2;
#define D_(A, K, LL, MASK) "v_fmac_f64_dpp %[" A "], %[x], %[m" #K "] row_newbcast:" LL MASK "\n\t"
#define OPS_                                                                                                                       \
    : [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2)                                                                                \
    : [x] "v"(src), [m0] "v"(M[0]), [m1] "v"(M[1]), [m2] "v"(M[2]), [m3] "v"(M[3]), [m4] "v"(M[4]), [m5] "v"(M[5]), [m6] "v"(M[6])
#define ZERO_ "v_mov_b64 %[s0], 0\n\tv_mov_b64 %[s1], 0\n\tv_mov_b64 %[s2], 0\n\ts_nop 1\n\t"
    if (LPI == 16)
        asm volatile(ZERO_ D_("s0", 0, "0", DPPM_ALL) D_("s1", 1, "1", DPPM_ALL) D_("s2", 2, "2", DPPM_ALL) D_("s0", 3, "3", DPPM_ALL) D_("s1", 4, "4", DPPM_ALL)
                         D_("s2", 5, "5", DPPM_ALL) D_("s0", 6, "6", DPPM_ALL) "s_nop 0" OPS_);
    else
        // the same accumulator for a term in both halves (the sum's grouping must not depend on which half an instance sits in)
        asm volatile(ZERO_ D_("s0", 0, "0", DPPM_LO) D_("s1", 1, "1", DPPM_LO) D_("s2", 2, "2", DPPM_LO) D_("s0", 0, "8", DPPM_HI) D_("s1", 1, "9", DPPM_HI)
                         D_("s2", 2, "10", DPPM_HI) D_("s0", 3, "3", DPPM_LO) D_("s1", 4, "4", DPPM_LO) D_("s2", 5, "5", DPPM_LO) D_("s0", 3, "11", DPPM_HI)
                             D_("s1", 4, "12", DPPM_HI) D_("s2", 5, "13", DPPM_HI) D_("s0", 6, "6", DPPM_LO) "s_nop 1\n\t" D_("s0", 6, "14", DPPM_HI) "s_nop 0" OPS_);
#undef D_
#undef OPS_
#undef ZERO_
    return (s0 + s1) + s2;
}

// The seven pivots of the symmetric sweep with deferred row scaling as ONE hand-ordered block (see the header).  Pivot C:
//   acc = bcast_C(s[C]);  rc = 1/acc (v_rcp_f64 + cubic step);  t = (s[C] + s[C] nm1) rc  (nm1 = -1 in the pivot lane: t = 0 there);
//   s[j] -= t bcast_C(s[j]) (j != C);  s[C] = t + nm1 (-1 in the pivot lane);  myrc -= rc nm1 (the pivot lane remembers its factor)
// A DPP read needs two wait states after the write of its source: s[C+1] is last written by one of pivot C's FMAs, at least the two
// instructions of its tail ahead of pivot C+1's move; the rcp result is used one instruction later (an independent FMA between).
#define PV_HEAD16_(C) "v_mov_b64_dpp %[acc], %[s" #C "] row_newbcast:" #C DPPM_ALL "\n\t"
#define PV_HEAD8_(C, C8) "v_mov_b64_dpp %[acc], %[s" #C "] row_newbcast:" #C DPPM_LO "\n\tv_mov_b64_dpp %[acc], %[s" #C "] row_newbcast:" #C8 DPPM_HI "\n\t"
#define PV_RCP_(C)                                                                                                                \
    "v_rcp_f64 %[rc], %[acc]\n\t"                                                                                                 \
    "v_fma_f64 %[t], %[s" #C "], %[n" #C "], %[s" #C "]\n\t"                                                                      \
    "v_fma_f64 %[e], -%[acc], %[rc], 1.0\n\t"                                                                                     \
    "v_fma_f64 %[e], %[e], %[e], %[e]\n\t"                                                                                        \
    "v_fma_f64 %[rc], %[e], %[rc], %[rc]\n\t"                                                                                     \
    "v_mul_f64 %[t], %[t], %[rc]\n\t"
#define PV_F_(J, LL, MASK) "v_fmac_f64_dpp %[s" #J "], %[s" #J "], -%[t] row_newbcast:" LL MASK "\n\t"
#define PV_TAIL_(C) "v_add_f64 %[s" #C "], %[t], %[n" #C "]\n\tv_fma_f64 %[myrc], -%[rc], %[n" #C "], %[myrc]\n\t"
#define PV16_(C, A, B, D, E, F, G) PV_HEAD16_(C) PV_RCP_(C) PV_F_(A, #C, DPPM_ALL) PV_F_(B, #C, DPPM_ALL) PV_F_(D, #C, DPPM_ALL) PV_F_(E, #C, DPPM_ALL) PV_F_(F, #C, DPPM_ALL) PV_F_(G, #C, DPPM_ALL) PV_TAIL_(C)
#define PV8_(C, C8, A, B, D, E, F, G)                                                                                              \
    PV_HEAD8_(C, C8) PV_RCP_(C) PV_F_(A, #C, DPPM_LO) PV_F_(B, #C, DPPM_LO) PV_F_(D, #C, DPPM_LO) PV_F_(E, #C, DPPM_LO) PV_F_(F, #C, DPPM_LO) PV_F_(G, #C, DPPM_LO) \
        PV_F_(A, #C8, DPPM_HI) PV_F_(B, #C8, DPPM_HI) PV_F_(D, #C8, DPPM_HI) PV_F_(E, #C8, DPPM_HI) PV_F_(F, #C8, DPPM_HI) PV_F_(G, #C8, DPPM_HI) PV_TAIL_(C)
template <int LPI>
__device__ __forceinline__ void pivots(double (&s)[7], const double (&nm1)[7], double& myrc) {
    double acc, rc, e, t;
#define OPS_                                                                                                                              \
    : [acc] "=&v"(acc), [rc] "=&v"(rc), [e] "=&v"(e), [t] "=&v"(t), [s0] "+v"(s[0]), [s1] "+v"(s[1]), [s2] "+v"(s[2]), [s3] "+v"(s[3]),     \
      [s4] "+v"(s[4]), [s5] "+v"(s[5]), [s6] "+v"(s[6]), [myrc] "+v"(myrc)                                                                  \
    : [n0] "v"(nm1[0]), [n1] "v"(nm1[1]), [n2] "v"(nm1[2]), [n3] "v"(nm1[3]), [n4] "v"(nm1[4]), [n5] "v"(nm1[5]), [n6] "v"(nm1[6])
    if (LPI == 16)
        asm volatile("s_nop 1\n\t" PV16_(0, 1, 2, 3, 4, 5, 6) PV16_(1, 0, 2, 3, 4, 5, 6) PV16_(2, 0, 1, 3, 4, 5, 6) PV16_(3, 0, 1, 2, 4, 5, 6) PV16_(4, 0, 1, 2, 3, 5, 6)
                         PV16_(5, 0, 1, 2, 3, 4, 6) PV16_(6, 0, 1, 2, 3, 4, 5) "s_nop 0" OPS_);
    else
        asm volatile("s_nop 1\n\t" PV8_(0, 8, 1, 2, 3, 4, 5, 6) PV8_(1, 9, 0, 2, 3, 4, 5, 6) PV8_(2, 10, 0, 1, 3, 4, 5, 6) PV8_(3, 11, 0, 1, 2, 4, 5, 6)
                         PV8_(4, 12, 0, 1, 2, 3, 5, 6) PV8_(5, 13, 0, 1, 2, 3, 4, 6) PV8_(6, 14, 0, 1, 2, 3, 4, 5) "s_nop 0" OPS_);
#undef OPS_
}

// DPP move of a double by 32-bit halves (quad_perm / row_half_mirror: 8-lane butterfly)
template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double oct_sum(double v) {  // sum over lanes 8m .. 8m+7, result in all eight
    v += dpp64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp64<0x141>(v);  // row_half_mirror
    return v;
}

__device__ __forceinline__ double ldg(const double* base, unsigned byte_off) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off); }
// A ring value moved out of its slot by an instruction the compiler cannot fold away: the slot register is then free BEFORE the slot's next
// load is issued, the loop-carried value and the load destination share one register, and no copy is left on the back edge.  (Without it
// the old value stayed in place for the whole step, the new load went to a second register, and the copies that rotate the ring at the
// end of the unrolled group waited for the loads issued ONE step earlier: s_waitcnt vmcnt(4) .. vmcnt(0) once per group.)
__device__ __forceinline__ double ring_take(double v) {
    double r;
    asm volatile("v_mov_b64_e32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ void stg(double* base, unsigned byte_off, double v) { *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + byte_off) = v; }

}  // namespace

// UNIF: every control weight is the same (R = r I, the common case): then N' = D M - I equals N = M D - I entry by entry
template <int LPI, int MR, bool FUSED, bool UNIF>
__global__ __launch_bounds__(64) void k_backward_si_dpp(Bufs a, SweepArgs sw) {
    constexpr int N = 7, IPW = 64 / LPI;
    constexpr int MRR = MR > 0 ? MR : 1;
    constexpr int ROWP = kd_rowp(N), RS = UNIF ? KD_SYM_RS : N * ROWP;  // uniform R: K is symmetric, the record holds its upper triangle (ilqr_kernels.hpp)
    static_assert(ROWP == 8, "a gain row is the lane's eight doubles {K_r0..K_r6, d_r}");
    static_assert(LPI == 8 || LPI == 16, "");
    // two images of the wave's gain records (written at the end of a step, sent out during the next) + one dump slot per lane behind each
    constexpr int NPQ_ = (IPW * (RS / 2) + 63) / 64, IMG = (IPW * RS > 128 * NPQ_) ? IPW * RS : 128 * NPQ_;
    __shared__ __attribute__((aligned(16))) double sK[2][IMG + 64];
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, g = lane / LPI, l = lane % LPI;
    const int r = l & 7;                 // row of the matrices / component of the vectors this lane owns (7 = padding: all zeros)
    const bool low = l < 8;              // LPI = 16: the half that stores (the other half holds a second copy)
    const int b = xcd_tile() * IPW + g;
    const int Bp = d.Bp, T = d.T;
    const bool ok = (b < d.B) && a.active[b < d.B ? b : 0];
    const int bb = (b < d.B) ? b : 0;    // clamp so that every address stays valid; stores are guarded
    const int pendw = (FUSED && b < d.B) ? a.pend[bb] : 0;
    const bool acc = pendw > 0;
    if (__ballot((ok || acc) ? 1 : 0) == 0ull) return;  // wave-uniform

    const bool isV = r < N;
    const int v = isV ? r : 0;
    const int cur = a.cur[bb];
    // the two buffers of X (of U) are the halves of one allocation with the same stride for X and U
    double* const Xb = a.X[0];
    double* const Ub = a.U[0];
    const unsigned bufo = (unsigned)((a.X[1] - a.X[0]) * (ptrdiff_t)sizeof(double));
    const double aacc = acc ? ldexp(1.0, -(pendw - 1)) : 1.0;
    const bool blend = acc && pendw > 1;  // alpha = 1: the other buffer already holds the accepted trajectory
    // xbar / ubar are read from buffer `rd`, x(1) / u(1) from buffer `r1`, and the accepted x, u are written over x(1), u(1).  Without a
    // pending acceptance both read xbar (x(1) - xbar = 0 reproduces xbar exactly) and nothing is stored.
    int rd = cur, r1 = blend ? 1 - cur : cur;
    if (FUSED && acc && !blend) rd = r1 = 1 - cur;
    const bool upd = FUSED && acc && sw.do_update_prev;
    const bool wr = blend && isV && low;  // this lane stores the accepted x, u
    const double dt = d.dt, idt = 1.0 / dt, idt2 = idt * idt, reg = d.reg, dt2 = dt * dt;
    const double Rv = isV ? d.R_diag[v] : 0.0, Dv = isV ? Rv + reg : 0.0;
    const int lim_on = d.limits_set;
    const double pen = d.penalty, pen_xx = d.pen_xx;
    const int lw_v = isV ? d.lw[v] : 0;
    // bounds with the weight folded in: an unweighted coordinate gets (+inf, -inf), which no x violates (a NaN neither)
    const double smax_v = lw_v != 0 ? d.smax[v] : __builtin_inf(), smin_v = lw_v != 0 ? d.smin[v] : -__builtin_inf();

    // lane constants: nm1[c] = -1 in the lane whose row is the pivot row of pivot c, 0 elsewhere
    double nm1[N], Dd[N], Dj[N];
    UNR for (int c = 0; c < N; c++) {
        nm1[c] = (r == c) ? -1.0 : 0.0;
        Dj[c] = d.R_diag[c] + reg;       // wave-uniform
        Dd[c] = (r == c) ? Dj[c] : 0.0;  // row r of D
    }
    const double cDN = Dv * idt2, cRN = reg * idt2;
    // constraint rows (state part only: checked on the host)
    const int m = (MR == 1) ? 1 : a.m;
    double Av[MRR], bbr[MRR], Arow[MRR][N];
    UNR for (int rr = 0; rr < MRR; rr++) {
        Av[rr] = bbr[rr] = 0;
        UNR for (int q = 0; q < N; q++) Arow[rr][q] = 0;
        if (MR > 0 && rr < m) {
            const double* Ar = a.conA + (size_t)rr * 2 * N;
            Av[rr] = isV ? Ar[v] : 0.0;
            bbr[rr] = a.conb[rr];
            UNR for (int q = 0; q < N; q++) Arow[rr][q] = Ar[q];
        }
    }
    // byte offsets (32 bit) of the lane's element at the step being loaded, one step = Vstep bytes back
    const unsigned Vstep = (unsigned)N * Bp * 8u, Lstep = (unsigned)m * Bp * 8u;
    unsigned oA = (unsigned)rd * bufo + ((unsigned)((T - 2) * N + v) * Bp + bb) * 8u;  // xbar, ubar
    unsigned oB = (unsigned)r1 * bufo + ((unsigned)((T - 2) * N + v) * Bp + bb) * 8u;  // x(1), u(1)
    unsigned oL = ((unsigned)(T - 2) * m * Bp + bb) * 8u;                              // multipliers / I_k of row 0
    // gain records of the wave's instances (adjacent in memory): 16-byte pieces, piece c of the image belongs to instance c / (RS / 2)
    constexpr int PCS = RS / 2, NPQ = (IPW * PCS + 63) / 64;
    const unsigned long long okm = __ballot(ok ? 1 : 0);
    bool pst[NPQ];
    const bool kfull = okm == ~0ull;  // every instance of the wave stores its gains: only the ragged tail of the last round is masked
    UNR for (int q = 0; q < NPQ; q++) {
        const int c = lane + 64 * q, gi = (c / PCS < IPW) ? c / PCS : 0;
        pst[q] = c < IPW * PCS && ((okm >> (gi * LPI)) & 1ull);
    }
    // where this lane's entries go in the image (doubles): uniform R -- the upper triangle of its row and d_r, the rest to the lane's dump slot
    int wo[N], wod = IMG + lane;
    UNR for (int j = 0; j < N; j++) wo[j] = IMG + lane;
    if (UNIF) {
        if (isV && low) {
            UNR for (int j = 0; j < N; j++) wo[j] = (j >= v) ? g * RS + kd_sym_tri(v, j) : IMG + lane;
            wod = g * RS + KD_SYM_D + v;
        }
        if (l == 0) { sK[0][g * RS + KD_SYM_RS - 1] = 0; sK[1][g * RS + KD_SYM_RS - 1] = 0; }  // the pad entry of the record
    }
    double* Kout = KD_REC(a.KD, Bp, RS, T - 2, xcd_tile() * IPW);  // (the gains of a big batch pass 4 GiB: 64-bit pointer)
    const ptrdiff_t Kstep = (ptrdiff_t)Bp * RS;

    int kpi = d.n_kp - 1;
    int kp_next = (kpi >= 0) ? d.kp_t[kpi] : -1;
    const size_t kpd_stride = (size_t)(N + N * N) * Bp;

    // terminal values: P = l_xx(x_{T-1}), p = l_x(x_{T-1})
    double P[N], p = 0;
    UNR for (int j = 0; j < N; j++) P[j] = 0;
    {
        double xv = ldg(Xb, oA + Vstep);
        if (FUSED) {
            const double x1 = ldg(Xb, oB + Vstep);
            xv = fma(aacc, x1 - xv, xv);
            if (wr) stg(Xb, oB + Vstep, xv);
        }
        if (kp_next == T - 1) {
            const double* src = a.kpd + (size_t)kpi * kpd_stride;
            if (isV) {
                UNR for (int j = 0; j < N; j++) P[j] = AT(src, N + v * N + j, bb);
                p = AT(src, v, bb);
            }
            kpi--;
            kp_next = (kpi >= 0) ? d.kp_t[kpi] : -1;
        } else if (lim_on) {
            const double beyond = fmax(xv - smax_v, 0.0) + fmax(smin_v - xv, 0.0);
            const double lim = (beyond > 0.0) ? pen_xx : 0.0;
            UNR for (int j = 0; j < N; j++) P[j] = -nm1[j] * lim;
            if (lw_v != 0) {
                if (xv > smax_v) p = -pen * (smax_v - xv);
                else if (xv < smin_v) p = -pen * (smin_v - xv);
            }
        }
    }
    // PF-steps-ahead prefetch ring (vmcnt retires in issue order: a load issued one step ahead would wait for the previous step's stores)
    constexpr int PF = 4;
    double xr[PF], ur[PF], x1r[PF], u1r[PF], lr[PF][MRR], ir[PF][MRR];
    unsigned rofs[MRR];
    UNR for (int rr = 0; rr < MRR; rr++) rofs[rr] = (unsigned)((MR > 0 && rr < m) ? rr : 0) * Bp * 8u;
    auto fetch = [&](int slot, int kk) {  // loads of timestep max(kk, 0) into ring slot; every load unconditional (a CFG path that skips
        xr[slot] = ldg(Xb, oA);           // one makes the waitcnt pass fall back to vmcnt(0))
        ur[slot] = ldg(Ub, oA);
        x1r[slot] = u1r[slot] = 0;
        if (FUSED) { x1r[slot] = ldg(Xb, oB); u1r[slot] = ldg(Ub, oB); }
        UNR for (int rr = 0; rr < MRR; rr++) {
            lr[slot][rr] = ir[slot][rr] = 0;
            if (MR > 0) { lr[slot][rr] = ldg(a.lambda, oL + rofs[rr]); if (!FUSED) ir[slot][rr] = ldg(a.Is, oL + rofs[rr]); }
        }
        if (kk > 0) { oA -= Vstep; oB -= Vstep; oL -= Lstep; }  // uniform; no load inside the branch
    };
    UNR for (int q = 0; q < PF; q++) { fetch(q, T - 2 - q); __builtin_amdgcn_sched_barrier(0); }
    // store side of the fused acceptance: the accepted x, u of step k go where x(1), u(1) were read (PF steps behind the loads)
    unsigned oW = (unsigned)r1 * bufo + ((unsigned)((T - 2) * N + v) * Bp + bb) * 8u;
    unsigned oLw = ((unsigned)(T - 2) * m * Bp + bb) * 8u;

    for (int k0 = T - 2; k0 >= 0; k0 -= PF) {
      UNR for (int jj = 0; jj < PF; jj++) {
        const int k = k0 - jj;
        double xv = ring_take(xr[jj]), uv = ring_take(ur[jj]);
        if (FUSED) {  // accepted trajectory (k_apply's expression)
            const double x1v = ring_take(x1r[jj]), u1v = ring_take(u1r[jj]);
            xv = fma(aacc, x1v - xv, xv);
            uv = fma(aacc, u1v - uv, uv);
        }
        double lam[MRR], Isk[MRR];
        UNR for (int rr = 0; rr < MRR; rr++) {
            lam[rr] = (MR > 0) ? ring_take(lr[jj][rr]) : 0.0;
            Isk[rr] = (MR > 0 && !FUSED) ? ring_take(ir[jj][rr]) : ir[jj][rr];
        }
        __builtin_amdgcn_sched_barrier(0);  // the slots are free: only now their next loads
        fetch(jj, k - PF);
        // the image of the step before this one (in time) leaves now: LDS -> registers here, registers -> memory after the pivots
        const bool kprev = k < T - 2;
        double kqa[NPQ], kqb[NPQ];  // (two arrays of doubles: an array of double2 -- or anything a lambda captures by reference -- goes to scratch)
        UNR for (int q = 0; q < NPQ; q++) {  // unconditional: stays inside the (padded) array
            const double2 t2 = reinterpret_cast<const double2*>(sK[(jj + 1) & 1])[lane + 64 * q];
            kqa[q] = t2.x; kqb[q] = t2.y;
        }
#define SEND_()                                                                                                         \
        {                                                                                                               \
            double2* dst = reinterpret_cast<double2*>(Kout + Kstep) + lane;                                             \
            if (kfull) {                                                                                                \
                UNR for (int q = 0; q + 1 < NPQ; q++) dst[64 * q] = make_double2(kqa[q], kqb[q]);                       \
                if (lane + 64 * (NPQ - 1) < IPW * PCS) dst[64 * (NPQ - 1)] = make_double2(kqa[NPQ - 1], kqb[NPQ - 1]);  \
            } else {                                                                                                    \
                UNR for (int q = 0; q < NPQ; q++) if (pst[q]) dst[64 * q] = make_double2(kqa[q], kqb[q]);               \
            }                                                                                                           \
        }
        if (k < 0) {  // uniform: dummy step of the last group; the one right behind the last real step sends its image out
            if (k == -1) SEND_()
            continue;
        }
        if (FUSED) {
            if (wr) { stg(Xb, oW, xv); stg(Ub, oW, uv); }
            oW -= Vstep;
        }
        const double Qu = Rv * uv + dt * p;  // Qu = R u + B'p
        // ---- S = D + dt^2 P (row r), then the seven pivots: afterwards myrc * s = row r of -S^-1
        double s[N];
        UNR for (int j = 0; j < N; j++) s[j] = fma(dt2, P[j], Dd[j]);
        double myrc = 0.0;
        pivots<LPI>(s, nm1, myrc);
        if (kprev) SEND_()
        const double nrc = -myrc;
        // row r of S^-1 is M = nrc s.  With uniform control weights M is never formed: N = M D - I = (nrc D) s - I, and the two
        // products with M run on s and are scaled afterwards.
        double M[N], Nn[N], ntc[N], Kr[N];
        if (UNIF) {
            const double nrcD = nrc * Dj[0];
            UNR for (int j = 0; j < N; j++) {
                Nn[j] = fma(s[j], nrcD, nm1[j]);  // row r of N = M D - I (= row r of N' = D M - I)
                ntc[j] = Nn[j] * -cRN;
                Kr[j] = Nn[j] * idt;              // K = N / dt
            }
        } else {
            UNR for (int j = 0; j < N; j++) {
                M[j] = nrc * s[j];
                Nn[j] = fma(M[j], Dj[j], nm1[j]);
                ntc[j] = fma(Dv, M[j], nm1[j]) * -cRN;  // -reg/dt^2 x row r of N' = D M - I (M symmetric)
                Kr[j] = Nn[j] * idt;
            }
        }
        // ---- d = -M Qu
        const double dv = UNIF ? myrc * row_dot_bc<LPI>(s, Qu) : -row_dot_bc<LPI>(M, Qu);
        // ---- gains out.  The lane's row {K_r0 .. K_r6, d_r} is 64 contiguous bytes of the record, but stored from here a store
        // instruction would write 16-byte pieces 64 bytes apart: 56 partial-sector writes per instruction, which at four waves per CU
        // doubled the launch (measured: 381 -> 194 us at B = 8192 with these stores removed).  The rows go into a wave-local LDS image of
        // the wave's records as they lie in memory; the image leaves during the NEXT step as 16-byte pieces of whole lines (1 KiB
        // contiguous per instruction), so neither the LDS round trip nor the stores sit on this step's chain.
        if (UNIF) {  // packed symmetric record: row r contributes K_rr .. K_r6 and d_r; everything else (and every lane without a row) writes its dump slot -- no mask
            double* img = sK[jj & 1];
            UNR for (int j = 0; j < N; j++) img[wo[j]] = Kr[j];
            img[wod] = dv;
        } else if (isV && low) {
            double2* w = reinterpret_cast<double2*>(&sK[jj & 1][g * RS + v * ROWP]);
            w[0] = make_double2(Kr[0], Kr[1]);
            w[1] = make_double2(Kr[2], Kr[3]);
            w[2] = make_double2(Kr[4], Kr[5]);
            w[3] = make_double2(Kr[6], dv);
        }
        Kout -= Kstep;
        // ---- stage derivatives l_xx (row r), l_x (component r), accumulated onto -D N / dt^2
        double Pn[N], lx = 0;
        UNR for (int j = 0; j < N; j++) Pn[j] = -cDN * Nn[j];
        if (k == kp_next) {  // uniform: keypoint step, precomputed by k_kp_derivs (incl. limits)
            const double* src = a.kpd + (size_t)kpi * kpd_stride;
            double lxx[N];
            UNR for (int j = 0; j < N; j++) lxx[j] = AT(src, N + v * N + j, bb);
            lx = AT(src, v, bb);
            if (!isV) { UNR for (int j = 0; j < N; j++) lxx[j] = 0; lx = 0; }
            UNR for (int j = 0; j < N; j++) Pn[j] += lxx[j];
            // consume the loads inside the branch: otherwise their wait lands after the join and every step drains vmcnt to 0
            UNR for (int j = 0; j < N; j++) asm volatile("" : "+v"(Pn[j]));
            asm volatile("" : "+v"(lx));
            kpi--;
            kp_next = (kpi >= 0) ? __builtin_amdgcn_readfirstlane(d.kp_t[kpi]) : -1;
        } else if (lim_on) {  // uniform.  inspectJointLimit (System.cpp:121-142) branch-free; l_xx of the limits is diagonal
            const double beyond = fmax(xv - smax_v, 0.0) + fmax(smin_v - xv, 0.0);
            const double lim = (beyond > 0.0) ? pen_xx : 0.0;
            if (__ballot(beyond > 0.0 ? 1 : 0) != 0ull) { UNR for (int j = 0; j < N; j++) Pn[j] = fma(-nm1[j], lim, Pn[j]); }  // uniform; adds exact zeros otherwise
            // l_x = -L q, q = limit - x on the violated side: the bits of -pen (max - x), -pen (min - x)
            lx = pen * fmax(xv - smax_v, 0.0) - pen * fmax(smin_v - xv, 0.0);
        }
        if (MR > 0) {
            UNR for (int rr = 0; rr < MRR; rr++) {
                if (MR == 1 || rr < m) {
                    // g = A_r . x - b: every lane multiplies its own component, 8-lane butterfly (lane 7 adds 0); all lanes of the instance hold it
                    const double gr = -bbr[rr] + oct_sum(Av[rr] * xv);
                    if (FUSED) {  // AL-ILQR.cpp:190 (mask with the multipliers before the update), :202-208 (update)
                        Isk[rr] = sw.pen_in * ((gr < 0 && lam[rr] == 0) ? 0.0 : 1.0);
                        const double nv = lam[rr] + sw.pen_update_prev * gr;
                        const double nl = nv > 0 ? nv : 0;  // cwiseMax(0); a NaN becomes 0 as before
                        lam[rr] = upd ? nl : lam[rr];
                        if (upd && l == 0) stg(a.lambda, oLw + rofs[rr], lam[rr]);
                    }
                    const double wI = Av[rr] * Isk[rr];
                    UNR for (int j = 0; j < N; j++) Pn[j] = fma(wI, Arow[rr][j], Pn[j]);
                    lx += Av[rr] * (lam[rr] + Isk[rr] * gr);
                }
            }
        }
        // ---- P' = l_xx - (D N + reg N'N)/dt^2: row r of N'N = sum_k (N')_rk (row k of N), accumulated onto the rest
        fmac_rows_all<LPI>(Pn, Nn, ntc);
        UNR for (int j = 0; j < N; j++) P[j] = Pn[j];
        const double Md = UNIF ? nrc * row_dot_bc<LPI>(s, dv) : row_dot_bc<LPI>(M, dv);
        p = lx + p - (Qu + Dv * dv) * idt - reg * (Dv * Md - dv) * idt;
        if (FUSED && MR > 0) oLw -= Lstep;
      }
    }
    if ((T - 2) % PF == PF - 1) {  // the last real step closed its group: no dummy step sent its image out
        UNR for (int q = 0; q < NPQ; q++) if (pst[q]) reinterpret_cast<double2*>(Kout + Kstep)[lane + 64 * q] = reinterpret_cast<const double2*>(sK[(PF - 1) & 1])[lane + 64 * q];
    }
    if (FUSED && acc && l == 0) {  // the acceptance is complete: the other buffer is the instance's trajectory now
        a.cur[bb] = 1 - cur;
        a.pend[bb] = 0;
    }
}

template <int LPI, bool FUSED, bool UNIF>
static void launch_dpp2(bool al, const Bufs& a, int B, hipStream_t st, const SweepArgs& sw) {
    const dim3 grid(grid_x8((B + 64 / LPI - 1) / (64 / LPI))), block(64);
    if (!al || a.m == 0) hipLaunchKernelGGL((k_backward_si_dpp<LPI, 0, FUSED, UNIF>), grid, block, 0, st, a, sw);
    else if (a.m <= 1) hipLaunchKernelGGL((k_backward_si_dpp<LPI, 1, FUSED, UNIF>), grid, block, 0, st, a, sw);
    else hipLaunchKernelGGL((k_backward_si_dpp<LPI, 4, FUSED, UNIF>), grid, block, 0, st, a, sw);
}

void launch_backward_si_dpp(bool al, bool fused, bool uniform_R, int lpi, const Bufs& a, int B, hipStream_t st, const SweepArgs& sw) {
    if (lpi == 16) {
        if (fused) { if (uniform_R) launch_dpp2<16, true, true>(al, a, B, st, sw); else launch_dpp2<16, true, false>(al, a, B, st, sw); }
        else { if (uniform_R) launch_dpp2<16, false, true>(al, a, B, st, sw); else launch_dpp2<16, false, false>(al, a, B, st, sw); }
    } else {
        if (fused) { if (uniform_R) launch_dpp2<8, true, true>(al, a, B, st, sw); else launch_dpp2<8, true, false>(al, a, B, st, sw); }
        else { if (uniform_R) launch_dpp2<8, false, true>(al, a, B, st, sw); else launch_dpp2<8, false, false>(al, a, B, st, sw); }
    }
}

}  // namespace ilqr
