// lane_math.hpp -- per-lane FP64 building blocks of the batched DLS kernels.
//
// One IK problem lives in one wavefront lane; every function here is straight-line,
// branch-free (selects, not divergent branches) scalar double code that hipcc keeps in VGPRs.
// Uniform model constants arrive through the kernel-argument segment (SGPRs).
//
// The file also compiles as plain host C++ (g++) -- ONLY so that tests/ can run the exact
// lane program on the CPU ("lane emulation") and compare it with the oracle before a GPU
// is involved.  libikgpu.so never contains or calls a host build of this code.
#pragma once

#if defined(__HIPCC_RTC__)
// Compiled at run time through hipRTC (rtc.cpp): no standard headers there -- hipRTC pre-includes the HIP built-ins; the few
// standard names the lane programs use are declared here.
typedef signed long long int64_t;
typedef unsigned long long uint64_t;
typedef signed int int32_t;
typedef unsigned int uint32_t;
typedef unsigned char uint8_t;
namespace std {
template <class A, class B> struct is_same { static constexpr bool value = false; };
template <class A> struct is_same<A, A> { static constexpr bool value = true; };
template <bool C, class A, class B> struct conditional { typedef A type; };
template <class A, class B> struct conditional<false, A, B> { typedef B type; };
}  // namespace std
#else
#include <cmath>
#include <cstdint>
#endif

#if defined(__HIPCC_RTC__)
#define IKD_FN __host__ __device__ inline __attribute__((always_inline))
#elif defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define IKD_FN __host__ __device__ inline __attribute__((always_inline))
#else
#define IKD_FN inline __attribute__((always_inline))
#endif

// HIP language mode (both passes of hipcc, and hipRTC): __device__-only code (wave intrinsics, atomics) may be declared
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define IKD_HIP_LANG 1
#else
#define IKD_HIP_LANG 0
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define IKD_ON_DEVICE 1
#else
#define IKD_ON_DEVICE 0
#endif

// Scheduling fence between the phases of a long unrolled lane program: keeps the machine scheduler from
// hoisting a later phase's LDS / global loads (and their destination registers) across an earlier phase,
// which is what pushes the big kernels over the 512-register budget.  No instruction is emitted.
#if IKD_ON_DEVICE
#define IKD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define IKD_ANY(pred) (__any(pred) != 0)  // true in every lane when the predicate holds in some lane of the wave
// Pins a value in a VGPR at this point: keeps hipcc from sinking its computation into one arm of a later select and
// turning that select into a divergent branch (a branch splits the iteration's basic block and strands the LDS /
// global loads behind it: with one wave per SIMD their latency is then fully exposed).
#define IKD_PIN(x) asm volatile("" : "+v"(x))
// A wave-uniform constant table read through the constant address space comes in by scalar loads (s_load_dwordx16 into
// SGPRs, which the FP64 instructions take as operands directly): no LDS round trip, no VGPRs.  IKD_LAUNDER makes the
// table's address opaque once per iteration, otherwise the loads are hoisted out of the loop and the ~100 doubles spill.
#define IKD_CONST_AS __attribute__((address_space(4)))
#define IKD_LAUNDER(ptr) asm volatile("" : "+s"(ptr))
#else
#define IKD_CONST_AS
#define IKD_LAUNDER(ptr) ((void)0)
#define IKD_PIN(x) ((void)0)
#define IKD_SCHED_FENCE() ((void)0)
#define IKD_ANY(pred) (pred)
#endif

// The theta -> pi formula of log3 is either gated by a wave-uniform branch (skipped unless some lane needs it)
// or always evaluated and selected per lane (no branch: the iteration stays one basic block).
#ifdef IKD_NEAR_PI_BRANCHLESS
#define IKD_NEAR_PI_GATE(pred) true
#else
#define IKD_NEAR_PI_GATE(pred) IKD_ANY(pred)
#endif

namespace ikdev {

// TaylorSeriesExpansion<double>::precision<3>() = eps^(1/4)
constexpr double kTaylorPrec3 = 1.220703125e-4;  // 2^-13 exactly == (2^-52)^(1/4)
constexpr double kPi = 3.141592653589793238462643383279502884;

IKD_FN double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// a * b + c as ONE three-address v_fma_f64 whatever the register allocation: for a Horner step whose addend is a constant that
// lives in a VGPR across the loop hipcc emits v_mov_b64 (8 cycles of issue) + v_fmac_f64 (the two-address form needs the addend in
// its destination); the explicit form has no copy.  Only where that pattern was seen in the listing (the acos polynomials).
IKD_FN double dfma3(double a, double b, double c) {
#if IKD_ON_DEVICE
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return __builtin_fma(a, b, c);
#endif
}
IKD_FN double dsel(bool c, double a, double b) { return c ? a : b; }
// A per-lane choice between two ALREADY EVALUATED values.  In the run-time specialised programs (IKD_STATIC_TABLES: thousands of
// instructions per iteration, up to 512 registers and some spill code) a per-lane `c ? a : b` must not become a divergent branch:
// the register allocator was seen to put a spill store of a loop-carried value into the flow block of such a branch, where the EXEC
// mask is partial -- masked-off lanes never wrote their slot and got a stale value back (tools/spill_exec_check.py).
// There every choice goes through dsel (both arms evaluated, one v_cndmask pair); elsewhere it is the plain conditional operator,
// which hipcc may turn into a branch around an expensive arm (measured faster in the chain kernels, which do not spill).
#ifdef IKD_STATIC_TABLES
#define IKD_CHOOSE(c, a, b) dsel((c), (a), (b))
#else
#define IKD_CHOOSE(c, a, b) ((c) ? (a) : (b))
#endif
IKD_FN double dmin(double a, double b) { return __builtin_fmin(a, b); }
IKD_FN double dmax(double a, double b) { return __builtin_fmax(a, b); }

// 1/x for normal-range x: v_rcp_f64 seed (measured 24.4 bits on gfx950, tools/seed_probe.hip) + one
// second-order Newton step: r (1 + e + e^2), e = 1 - x r  ->  relative error e^3 ~ 1e-22, i.e. rounding-limited
// (no div_scale / div_fmas / div_fixup sequence; denormal and huge inputs are outside the path's range).
IKD_FN double drcp(double x) {
#if IKD_ON_DEVICE
    const double r = __builtin_amdgcn_rcp(x);
    const double e = dfma(-x, r, 1.0);
    return dfma(dfma(e, e, e), r, r);
#else
    return 1.0 / x;
#endif
}

// 1/sqrt(x) for normal-range x > 0: v_rsq_f64 seed (24.2 bits) + one third-order (Halley) step:
// y (1 + e/2 + 3 e^2/8), e = 1 - x y^2  ->  relative error ~e^3, rounding-limited.
IKD_FN double drsqrt(double x) {
#if IKD_ON_DEVICE
    const double y = __builtin_amdgcn_rsq(x);
    const double e = dfma(-x * y, y, 1.0);
    return dfma(y * e, dfma(e, 0.375, 0.5), y);
#else
    return 1.0 / std::sqrt(x);
#endif
}

// sqrt(x) for x >= 0 in the normal range (0 maps to 0): x * rsqrt(x) with one correction step.
IKD_FN double dsqrt(double x) {
#if IKD_ON_DEVICE
    const double y = drsqrt(dmax(x, 1e-300));
    const double s = x * y;
    return dfma(dfma(-s, s, x), 0.5 * y, s);
#else
    return __builtin_sqrt(x);
#endif
}

// sin and cos of one argument.  Cody-Waite reduction by pi/2 with an FMA'd two-constant split
// (exact enough for |x| <~ 1e5; joint angles are clamped to their limits every iteration),
// fdlibm kernel polynomials on [-pi/4, pi/4], quadrant fix-up by selects.  < 1.5 ulp.
IKD_FN void dsincos(double x, double &s_out, double &c_out) {
    constexpr double kTwoOverPi = 0.63661977236758134308;
    constexpr double kPio2Hi = 1.57079632679489655800e+00;
    constexpr double kPio2Lo = 6.12323399573676603587e-17;
    const double k = __builtin_rint(x * kTwoOverPi);
    double r = dfma(-k, kPio2Hi, x);
    r = dfma(-k, kPio2Lo, r);
    const int n = static_cast<int>(k);
    const double z = r * r;
    // sin(r) = r + r^3 (S1 + z (S2 + ...))
    double ps = dfma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = dfma(z, ps, 2.75573137070700676789e-06);
    ps = dfma(z, ps, -1.98412698298579493134e-04);
    ps = dfma(z, ps, 8.33333333332248946124e-03);
    ps = dfma(z, ps, -1.66666666666666324348e-01);
    const double sr = dfma(z * r, ps, r);
    // cos(r) = 1 - z/2 + z^2 (C1 + z (C2 + ...))
    double pc = dfma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = dfma(z, pc, -2.75573143513906633035e-07);
    pc = dfma(z, pc, 2.48015872894767294178e-05);
    pc = dfma(z, pc, -1.38888888888741095749e-03);
    pc = dfma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + z * z * pc);
    const bool swap = (n & 1) != 0;
    const double s0 = swap ? cr : sr;
    const double c0 = swap ? sr : cr;
    s_out = (n & 2) ? -s0 : s0;
    c_out = ((n + 1) & 2) ? -c0 : c0;
}

// sin and cos of a bounded argument, |x| <= 2^DBL * pi/4 (DBL = 2: pi, DBL = 3: 2 pi).  No quadrant selects: the same
// fdlibm kernels at x / 2^DBL, then DBL angle doublings s' = 2 s c, c' = 1 - 2 s^2.  22 (DBL = 2) / 25 (DBL = 3) FP64
// instructions and no integer ones.  Absolute error < 2e-15 on the whole range (tests/test_lane_emulation.py); the caller
// guarantees the range (dsincos_fast reduces by 2 pi first).
template <int DBL>
IKD_FN void dsincos_bounded(double x, double &s_out, double &c_out) {
    const double t = x * (1.0 / static_cast<double>(1 << DBL));
    const double z = t * t;
    double ps = dfma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = dfma(z, ps, 2.75573137070700676789e-06);
    ps = dfma(z, ps, -1.98412698298579493134e-04);
    ps = dfma(z, ps, 8.33333333332248946124e-03);
    ps = dfma(z, ps, -1.66666666666666324348e-01);
    double s = dfma(z * t, ps, t);
    double pc = dfma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = dfma(z, pc, -2.75573143513906633035e-07);
    pc = dfma(z, pc, 2.48015872894767294178e-05);
    pc = dfma(z, pc, -1.38888888888741095749e-03);
    pc = dfma(z, pc, 4.16666666666666019037e-02);
    double c = dfma(z, dfma(z, pc, -0.5), 1.0);
#pragma unroll
    for (int k = 0; k < DBL; ++k) {
        const double a = s + s;
        const double s2 = a * c;
        c = dfma(-a, s, 1.0);
        s = s2;
    }
    s_out = s;
    c_out = c;
}

// sin and cos of any argument for the iteration loops: Cody-Waite reduction by 2 pi into [-pi, pi] (two FMAs, exact enough
// for |x| <~ 1e5 like dsincos), then dsincos_bounded<2>.  26 FP64 instructions, no integer ones, no selects, against
// 32 + 14 for dsincos; absolute error < 2e-15.
IKD_FN void dsincos_fast(double x, double &s_out, double &c_out) {
    constexpr double kOneOverTwoPi = 0.15915494309189533577;
    constexpr double kTwoPiHi = 6.28318530717958623200e+00;  // 2 * kPio2Hi * 2, exactly representable splits of 2 pi
    constexpr double kTwoPiLo = 2.44929359829470641435e-16;
    const double k = __builtin_rint(x * kOneOverTwoPi);
    double r = dfma(-k, kTwoPiHi, x);
    r = dfma(-k, kTwoPiLo, r);
    dsincos_bounded<2>(r, s_out, c_out);
}

// sin and cos for the headline loop (device/chain_hot.hpp): dsincos_fast with two instructions less per call.
//  * the reduction by 2 pi drops the low word of the constant: the angle then carries an error of |x| * 3.9e-17 -- below half an
//    ulp of x itself (a joint angle inside its limits: < 2.5e-16 rad), invisible next to the 2e-15 of the kernels;
//  * no division of the reduced angle by 4: the fdlibm kernels are evaluated at t = r / 4 through coefficients scaled by exact
//    powers of two -- sin t = r (1/4 + z' Q(z')), cos t = 1 + z' (-1/32 + z' C(z')), z' = r^2 -- so every Horner intermediate is
//    2^m times the one of dsincos_bounded<2> (same mantissa bits), then the two angle doublings.
IKD_FN void dsincos_hot(double x, double &s_out, double &c_out) {
    constexpr double kOneOverTwoPi = 0.15915494309189533577;
    constexpr double kTwoPiHi = 6.28318530717958623200e+00;
    const double k = __builtin_rint(x * kOneOverTwoPi);
    const double r = dfma(-k, kTwoPiHi, x);
    const double z = r * r;   // = 16 t^2
    // S_k / (64 * 16^(k-1)),  k = 1..6
    double ps = dfma(z, 1.58969099521155010221e-10 / 67108864.0, -2.50507602534068634195e-08 / 4194304.0);
    ps = dfma(z, ps, 2.75573137070700676789e-06 / 262144.0);
    ps = dfma(z, ps, -1.98412698298579493134e-04 / 16384.0);
    ps = dfma(z, ps, 8.33333333332248946124e-03 / 1024.0);
    ps = dfma(z, ps, -1.66666666666666324348e-01 / 64.0);
    double s = r * dfma(z, ps, 0.25);
    // C_k / (256 * 16^(k-1)),  k = 1..6
    double pc = dfma(z, -1.13596475577881948265e-11 / 268435456.0, 2.08757232129817482790e-09 / 16777216.0);
    pc = dfma(z, pc, -2.75573143513906633035e-07 / 1048576.0);
    pc = dfma(z, pc, 2.48015872894767294178e-05 / 65536.0);
    pc = dfma(z, pc, -1.38888888888741095749e-03 / 4096.0);
    pc = dfma(z, pc, 4.16666666666666019037e-02 / 256.0);
    double c = dfma(z, dfma(z, pc, -0.03125), 1.0);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const double a = s + s;
        const double s2 = a * c;
        c = dfma(-a, s, 1.0);
        s = s2;
    }
    s_out = s;
    c_out = c;
}

// acos on [-1, 1] (fdlibm e_acos rational approximation, divisions by drcp).
IKD_FN double dacos(double x) {
    constexpr double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    constexpr double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                     pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                     pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
    constexpr double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                     qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const double ax = __builtin_fabs(x);
    const bool small = ax < 0.5;
    // small: z = x^2 ; large: z = (1-|x|)/2
    const double z = IKD_CHOOSE(small, x * x, (1.0 - ax) * 0.5);
    double p = dfma(z, pS5, pS4);
    p = dfma(z, p, pS3);
    p = dfma(z, p, pS2);
    p = dfma(z, p, pS1);
    p = dfma(z, p, pS0);
    p = p * z;
    double q = dfma(z, qS4, qS3);
    q = dfma(z, q, qS2);
    q = dfma(z, q, qS1);
    q = dfma(z, q, 1.0);
    const double r = p * drcp(q);
    // |x| < 0.5: pi/2 - (x - (pio2_lo - x*r))
    const double res_small = pio2_hi - (x - (pio2_lo - x * r));
    // |x| >= 0.5: s = sqrt(z)   (hipcc turns the final selects into a branch around this arm: measured faster than
    // pinning both arms, because near convergence every lane of a wave takes the same one)
    const double s = dsqrt(z);
    // x >= 0.5: 2*(df + w) with df = s truncated, c = (z - df*df)/(s + df), w = r*s + c
#if IKD_ON_DEVICE
    const double df = __longlong_as_double(__double_as_longlong(s) & 0xFFFFFFFF00000000ll);
#else
    double df;
    {
        std::uint64_t u;
        __builtin_memcpy(&u, &s, 8);
        u &= 0xFFFFFFFF00000000ull;
        __builtin_memcpy(&df, &u, 8);
    }
#endif
    const double c = dfma(-df, df, z) * drcp(s + df);
    const double res_pos = 2.0 * (df + dfma(r, s, c));
    // x <= -0.5: pi - 2*(s + (r*s - pio2_lo))
    const double res_neg = kPi - 2.0 * (s + dfma(r, s, -pio2_lo));
    const double res_large = IKD_CHOOSE(x > 0.0, res_pos, res_neg);
    double res = IKD_CHOOSE(small, res_small, res_large);
    res = IKD_CHOOSE(ax >= 1.0, IKD_CHOOSE(x > 0.0, 0.0, kPi), res);
    return res;
}

// ---------------------------------------------------------------------------------------------
// SE(3) pieces.  A rotation is 9 doubles row-major; everything is fully unrolled.
// ---------------------------------------------------------------------------------------------

// (R, p) <- (R, p) * (Rc, pc)   with (Rc, pc) = c[0..11], uniform constants.
// rot_identity is wave-uniform (a bit of the problem's placement mask, computed on the host): when Rc is
// exactly the identity the 27 rotation FMAs are skipped -- multiplying by exact ones and zeros changes no bit.
template <class ConstPtr>  // const double *, in LDS or (constant address space) in HBM
IKD_FN void se3_compose_const(double (&R)[9], double (&p)[3], ConstPtr c, bool rot_identity = false) {
    if (rot_identity) {
#pragma unroll
        for (int i = 0; i < 3; ++i) p[i] = dfma(R[3 * i], c[9], dfma(R[3 * i + 1], c[10], dfma(R[3 * i + 2], c[11], p[i])));
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a = R[3 * i], b = R[3 * i + 1], d = R[3 * i + 2];
        p[i] = dfma(a, c[9], dfma(b, c[10], dfma(d, c[11], p[i])));
        R[3 * i] = dfma(a, c[0], dfma(b, c[3], d * c[6]));
        R[3 * i + 1] = dfma(a, c[1], dfma(b, c[4], d * c[7]));
        R[3 * i + 2] = dfma(a, c[2], dfma(b, c[5], d * c[8]));
    }
}

// R <- R * Rz(angle) given (sin, cos)
IKD_FN void rot_z_right(double (&R)[9], double s, double c) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a = R[3 * i], b = R[3 * i + 1];
        R[3 * i] = dfma(a, c, b * s);
        R[3 * i + 1] = dfma(b, c, -(a * s));
    }
}

// out = R^T v
IKD_FN void rotT_vec(const double (&R)[9], const double (&v)[3], double (&out)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = dfma(R[i], v[0], dfma(R[3 + i], v[1], R[6 + i] * v[2]));
}

IKD_FN void cross(const double (&a)[3], const double (&b)[3], double (&c)[3]) {
    c[0] = dfma(a[1], b[2], -(a[2] * b[1]));
    c[1] = dfma(a[2], b[0], -(a[0] * b[2]));
    c[2] = dfma(a[0], b[1], -(a[1] * b[0]));
}

IKD_FN double dot(const double (&a)[3], const double (&b)[3]) { return dfma(a[0], b[0], dfma(a[1], b[1], a[2] * b[2])); }

// Result of log6(fMt) and of -W * Jlog6(tMf) for tMf = fMt^-1, sharing one log3.
// (reference ik/ik/frame.hpp:50-61 and :162-166; formulas SURVEY.md App. A.3)
struct LogAndJlog {
    double e[6];    // log6(fMt) = [v; w]
    double A[9];    // Jlog3(theta, -w)
    double Bm[9];   // C * A  (top-right block of Jlog6(tMf))
};

// Re, pe : rotation / translation of fMt.  any_near_pi: wave-uniform hint that some lane is in
// the theta >= pi - 1e-2 regime (the alternative log3 formula is evaluated only then).
IKD_FN void log6_and_jlog6_inv(const double (&Re)[9], const double (&pe)[3], LogAndJlog &o) {
    const double tr = Re[0] + Re[4] + Re[8];
    double x = (tr - 1.0) * 0.5;
    x = dmin(1.0, dmax(-1.0, x));  // tr >= 3 -> theta = 0 ; tr <= -1 -> theta = pi
    const double theta = dacos(x);
    const double omc = 1.0 - x;                                 // 1 - cos(theta)
    const double st = dsqrt(dmax(0.0, omc * (1.0 + x)));        // sin(theta), theta in [0, pi]
    const bool small = theta < kTaylorPrec3;
    const double inv_t = drcp(theta);
    const double inv_omc = drcp(omc);
    const double t2 = theta * theta;

    // log3, regular branch
    const double fac = 0.5 * dsel(theta > kTaylorPrec3, theta * drcp(st), 1.0);
    double w[3] = {fac * (Re[7] - Re[5]), fac * (Re[2] - Re[6]), fac * (Re[3] - Re[1])};
    // log3, theta >= pi - 1e-2: evaluated only when some lane of the wave is there (wave-uniform branch)
    const bool near_pi = theta >= kPi - 1e-2;
    if (IKD_NEAR_PI_GATE(near_pi)) {
        const double cphi = -x;
        const double beta_pi = t2 * drcp(1.0 + cphi);
        const double t0 = (Re[0] + cphi) * beta_pi, t1 = (Re[4] + cphi) * beta_pi, t2v = (Re[8] + cphi) * beta_pi;
        const double a0 = dsel(Re[7] > Re[5], 1.0, -1.0) * dsel(t0 > 0.0, dsqrt(dmax(t0, 0.0)), 0.0);
        const double a1 = dsel(Re[2] > Re[6], 1.0, -1.0) * dsel(t1 > 0.0, dsqrt(dmax(t1, 0.0)), 0.0);
        const double a2 = dsel(Re[3] > Re[1], 1.0, -1.0) * dsel(t2v > 0.0, dsqrt(dmax(t2v, 0.0)), 0.0);
        w[0] = dsel(near_pi, a0, w[0]);
        w[1] = dsel(near_pi, a1, w[1]);
        w[2] = dsel(near_pi, a2, w[2]);
    }

    const double st_omc = st * inv_omc;  // sin / (1 - cos)
    const double alpha = dsel(small, 1.0 - t2 * (1.0 / 12.0) - t2 * t2 * (1.0 / 720.0), theta * st_omc * 0.5);
    const double beta = dsel(small, 1.0 / 12.0 + t2 * (1.0 / 720.0), dfma(-0.5 * inv_t, st_omc, inv_t * inv_t));
    double wxp[3];
    cross(w, pe, wxp);
    const double bwp = beta * dot(w, pe);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o.e[i] = dfma(alpha, pe[i], dfma(-0.5, wxp[i], bwp * w[i]));
        o.e[3 + i] = w[i];
    }

    // Jlog6(tMf), tMf = (Re^T, -Re^T pe): log3 = -w, same theta.
    const double u[3] = {-w[0], -w[1], -w[2]};
    double pp[3];
    rotT_vec(Re, pe, pp);
    pp[0] = -pp[0]; pp[1] = -pp[1]; pp[2] = -pp[2];
    const double a3 = beta;  // 1/t^2 - (st/(1-ct))/(2t) ; same Taylor expansion
    const double diag = dsel(small, 0.5 * (2.0 - t2 * (1.0 / 6.0)), 0.5 * (theta * st_omc));
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o.A[3 * i + j] = a3 * u[i] * u[j];
    o.A[0] += diag; o.A[4] += diag; o.A[8] += diag;
    // + skew(0.5 u)
    o.A[1] -= 0.5 * u[2]; o.A[2] += 0.5 * u[1];
    o.A[3] += 0.5 * u[2]; o.A[5] -= 0.5 * u[0];
    o.A[6] -= 0.5 * u[1]; o.A[7] += 0.5 * u[0];

    const double inv_t2 = inv_t * inv_t;
    const double bdot = dsel(small, 1.0 / 360.0,
                             dfma(-2.0 * inv_t2, inv_t2, (1.0 + st * inv_t) * inv_t2 * (0.5 * inv_omc)));
    const double up = dot(u, pp);
    const double k1 = bdot * up, k2 = dfma(t2, bdot, 2.0 * beta);
    const double v3[3] = {dfma(k1, u[0], -(k2 * pp[0])), dfma(k1, u[1], -(k2 * pp[1])), dfma(k1, u[2], -(k2 * pp[2]))};
    double C[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = dfma(v3[i], u[j], beta * u[i] * pp[j]);
    const double ub = up * beta;
    C[0] += ub; C[4] += ub; C[8] += ub;
    C[1] -= 0.5 * pp[2]; C[2] += 0.5 * pp[1];
    C[3] += 0.5 * pp[2]; C[5] -= 0.5 * pp[0];
    C[6] -= 0.5 * pp[1]; C[7] += 0.5 * pp[0];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o.Bm[3 * i + j] = dfma(C[3 * i], o.A[j], dfma(C[3 * i + 1], o.A[3 + j], C[3 * i + 2] * o.A[6 + j]));
}

// log6(fMt) and Jlog6(tMf) sharing one log3 -- log6_and_jlog6_inv (above; formulas SURVEY.md App. A.3; reference
// ik/ik/frame.hpp:50-61,162-166) with a front end that costs fewer issue slots:
//  * acos without a branch: one rational R(z) and ONE square root serve the three ranges of fdlibm's e_acos
//    (|x| < 1/2: pi/2 - (x + x R(x^2));  x >= 1/2: 2 (s + s R(z));  x <= -1/2: pi - 2 (s + s R(z)),  z = (1 - |x|)/2, s = sqrt z);
//    the compensated square root (s refined with fma(-s, s, z)) replaces e_acos's split of s into two halves and its division;
//  * ONE reciprocal instead of three: with h = sin(theta)/2 = sqrt(z (1 - z)), z = (1 - cos theta)/2, r = 1 / (theta h) gives
//    1/theta = r h and 1/h = r theta, and sin/(1 - cos) = cot(theta/2) = (1 - z)/h, 1/(1 - cos) = (1 - z)/(2 h^2).
// Four FP64 transcendentals (16 cycles of issue each) instead of seven.  Results differ from the other front end by rounding only.
// WITH_BM = false: o.Bm is left unset and C is returned instead -- a caller that goes on to multiply by the frame rotation forms
// (C A) Rf^T as C (A Rf^T) and saves the 27 multiply-adds of C A (device/chain_hot.hpp).
template <bool WITH_BM = true>
IKD_FN void log6_and_jlog6_hot(const double (&Re)[9], const double (&pe)[3], LogAndJlog &o, double (*Cout)[9] = nullptr) {
    constexpr double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    constexpr double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                     pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                     pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
    constexpr double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                     qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const double tr = Re[0] + Re[4] + Re[8];
    double x = dfma(tr, 0.5, -0.5);
    x = dmin(1.0, dmax(-1.0, x));  // tr >= 3 -> theta = 0 ; tr <= -1 -> theta = pi
    // ---- theta = acos(x)
    const double ax = __builtin_fabs(x);
    const bool mid = ax < 0.5;
    const double z = dfma(-0.5, x, 0.5);       // (1 - cos theta)/2 = sin^2(theta/2)
    // (1 + cos theta)/2 = cos^2(theta/2) = 1 - z, kept off zero: at theta = pi to rounding (trace <= -1: x clamps to -1) an exact 0
    // would make h = 0 below, and the ONE reciprocal r = 1 / (theta h) would return 1/theta = r h = 0 -- beta, Jlog3's diagonal and the
    // translation of log6 all wrong, a step off by a radian (found by the step-synchronised parity test on its tenth seed: one lane-step
    // in 3e7).  With the floor h = 2^-52, r = 1.4e15, r h = 1/pi, and cot = zc / h = 2^-52 stands in for 0.
    const double zc = dmax(dfma(0.5, x, 0.5), 0x1p-104);
    const double za = mid ? x * x : dfma(-0.5, ax, 0.5);
    double pp_ = dfma3(za, pS5, pS4);
    pp_ = dfma3(za, pp_, pS3);
    pp_ = dfma3(za, pp_, pS2);
    pp_ = dfma3(za, pp_, pS1);
    pp_ = dfma3(za, pp_, pS0);
    pp_ = pp_ * za;
    double qq = dfma3(za, qS4, qS3);
    qq = dfma3(za, qq, qS2);
    qq = dfma3(za, qq, qS1);
    qq = dfma(za, qq, 1.0);
    const double rr = pp_ * drcp(qq);
    const double sa = dsqrt(za);
    const double tl = dfma(sa, rr, sa);        // asin(sqrt(za)) for za <= 1/4
    const double res_mid = pio2_hi - (x - dfma(-x, rr, pio2_lo));
    const bool pos = x > 0.0;
    const double res_out = dfma(pos ? 2.0 : -2.0, tl, pos ? 0.0 : kPi);
    const double theta = mid ? res_mid : res_out;
    // ---- the scalars of log3 / log6 / Jlog6
    const double h = dsqrt(z * zc);            // sin(theta)/2
    const double r = drcp(dmax(theta * h, 1e-300));
    const double inv_t = r * h, inv_h = r * theta;
    const double cot = zc * inv_h;             // sin / (1 - cos)
    const bool small = theta < kTaylorPrec3;
    const double t2 = theta * theta;
    const double inv_t2 = inv_t * inv_t;

    // log3, regular branch:  w = theta / (2 sin theta) * (R - R^T)^vee
    const double fac = dsel(theta > kTaylorPrec3, 0.25 * (theta * inv_h), 0.5);
    double w[3] = {fac * (Re[7] - Re[5]), fac * (Re[2] - Re[6]), fac * (Re[3] - Re[1])};
    // log3, theta >= pi - 1e-2: evaluated only when some lane of the wave is there (wave-uniform, out of line)
    const bool near_pi = theta >= kPi - 1e-2;
    if (__builtin_expect(IKD_ANY(near_pi), 0)) {
        const double cphi = -x;
        const double beta_pi = t2 * drcp(2.0 * z);
        const double t0 = (Re[0] + cphi) * beta_pi, t1 = (Re[4] + cphi) * beta_pi, t2v = (Re[8] + cphi) * beta_pi;
        const double a0 = dsel(Re[7] > Re[5], 1.0, -1.0) * dsel(t0 > 0.0, dsqrt(dmax(t0, 0.0)), 0.0);
        const double a1 = dsel(Re[2] > Re[6], 1.0, -1.0) * dsel(t1 > 0.0, dsqrt(dmax(t1, 0.0)), 0.0);
        const double a2 = dsel(Re[3] > Re[1], 1.0, -1.0) * dsel(t2v > 0.0, dsqrt(dmax(t2v, 0.0)), 0.0);
        w[0] = dsel(near_pi, a0, w[0]);
        w[1] = dsel(near_pi, a1, w[1]);
        w[2] = dsel(near_pi, a2, w[2]);
    }

    // alpha = theta sin / (2 (1 - cos)); the same expression is Jlog3's diagonal term (their Taylor forms differ by t^4/720,
    // below half an ulp of 1 for theta < 2^-13)
    const double alpha = dsel(small, 1.0 - t2 * (1.0 / 12.0) - t2 * t2 * (1.0 / 720.0), 0.5 * (theta * cot));
    const double beta = dsel(small, 1.0 / 12.0 + t2 * (1.0 / 720.0), dfma(-0.5 * inv_t, cot, inv_t2));
    double wxp[3];
    cross(w, pe, wxp);
    const double bwp = beta * dot(w, pe);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o.e[i] = dfma(alpha, pe[i], dfma(-0.5, wxp[i], bwp * w[i]));
        o.e[3 + i] = w[i];
    }

    // Jlog6(tMf), tMf = (Re^T, -Re^T pe): log3 = -w, same theta.
    const double u[3] = {-w[0], -w[1], -w[2]};
    double pp[3];
    rotT_vec(Re, pe, pp);
    pp[0] = -pp[0]; pp[1] = -pp[1]; pp[2] = -pp[2];
    const double a3 = beta;
    const double diag = dsel(small, 0.5 * (2.0 - t2 * (1.0 / 6.0)), 0.5 * (theta * cot));
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o.A[3 * i + j] = a3 * u[i] * u[j];
    o.A[0] += diag; o.A[4] += diag; o.A[8] += diag;
    o.A[1] -= 0.5 * u[2]; o.A[2] += 0.5 * u[1];
    o.A[3] += 0.5 * u[2]; o.A[5] -= 0.5 * u[0];
    o.A[6] -= 0.5 * u[1]; o.A[7] += 0.5 * u[0];

    // beta_dot_over_theta = -2/t^4 + (1 + sin/t) / (2 t^2 (1 - cos)),   1 / (2 (1 - cos)) = cot / (4 h) = cot inv_h / 4
    const double bdot = dsel(small, 1.0 / 360.0,
                             dfma(-2.0 * inv_t2, inv_t2, dfma(2.0 * h, inv_t, 1.0) * inv_t2 * (0.25 * (cot * inv_h))));
    const double up = dot(u, pp);
    const double k1 = bdot * up, k2 = dfma(t2, bdot, 2.0 * beta);
    const double v3[3] = {dfma(k1, u[0], -(k2 * pp[0])), dfma(k1, u[1], -(k2 * pp[1])), dfma(k1, u[2], -(k2 * pp[2]))};
    double C[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = dfma(v3[i], u[j], beta * u[i] * pp[j]);
    const double ub = up * beta;
    C[0] += ub; C[4] += ub; C[8] += ub;
    C[1] -= 0.5 * pp[2]; C[2] += 0.5 * pp[1];
    C[3] += 0.5 * pp[2]; C[5] -= 0.5 * pp[0];
    C[6] -= 0.5 * pp[1]; C[7] += 0.5 * pp[0];
    if (WITH_BM) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                o.Bm[3 * i + j] = dfma(C[3 * i], o.A[j], dfma(C[3 * i + 1], o.A[3 + j], C[3 * i + 2] * o.A[6 + j]));
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) (*Cout)[k] = C[k];
    }
}

// In-place Cholesky of the SPD matrix held in the lower triangle of G (row-major M x M, only
// i >= j used), then solve G x = b.  Diagonal entries are replaced by 1/L_ii.
// Stands in for `JJ.ldlt().solve(et)` (reference ik/ik/dls.cpp:53): JJ = J J^T + lambda^2 I is
// SPD for lambda > 0, so an unpivoted factorisation solves the same system.
template <int M>
IKD_FN void chol_solve(double (&G)[M * M], const double (&b)[M], double (&x)[M]) {
    // Right-looking (outer-product) form with the right-hand side carried as an extra row: after column k is
    // scaled, every trailing update is an independent FMA, so the 1/sqrt of the next pivot overlaps them instead
    // of waiting behind a dependent dot-product chain (the kernels run one wave per SIMD: latency is exposed).
    double y[M];
#pragma unroll
    for (int k = 0; k < M; ++k) y[k] = b[k];
#pragma unroll
    for (int k = 0; k < M; ++k) {
        const double inv = drsqrt(G[k * M + k]);
        G[k * M + k] = inv;
        y[k] = y[k] * inv;
#pragma unroll
        for (int i = k + 1; i < M; ++i) G[i * M + k] = G[i * M + k] * inv;
#pragma unroll
        for (int i = k + 1; i < M; ++i) {
#pragma unroll
            for (int j = k + 1; j <= i; ++j) G[i * M + j] = dfma(-G[i * M + k], G[j * M + k], G[i * M + j]);
            y[i] = dfma(-G[i * M + k], y[k], y[i]);
        }
    }
    // back substitution, column-oriented: once x_k is known every remaining y_m is updated independently
#pragma unroll
    for (int k = M - 1; k >= 0; --k) {
        x[k] = y[k] * G[k * M + k];
#pragma unroll
        for (int m = 0; m < k; ++m) y[m] = dfma(-G[k * M + m], x[k], y[m]);
    }
}

}  // namespace ikdev
