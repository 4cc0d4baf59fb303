// Branch-free fp64 sine and cosine for the rollouts of the batched iLQR kernels (the angle of
// systems.py:36-37, :46-49 and its likes), in a header of its own so that a host program can check
// the very code the kernels run (tests/test_fast_trig_cpu.py compiles it with g++).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ILQR_TRIG_FN __host__ __device__ __forceinline__
#else
#define ILQR_TRIG_FN static inline
#endif

namespace ilqr {

// hi ^ (bits & 0x80000000): the sign of a double (its high word) flipped if bit 31 of `bits` is set
ILQR_TRIG_FN uint32_t flip_sign_if(uint32_t hi, uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(hi, bits, 0x80000000u, 0x78); // one instruction on gfx950 (0x78: a ^ (b & c))
#else
    return hi ^ (bits & 0x80000000u);
#endif
}

// s = sin x, c = cos x to about 1.5 ulp (measured against long-double libm on 2 * 10^7 points up to
// |x| = 2^20).  Cody-Waite reduction by pi/2 split into three doubles (the products are exact inside the
// fused multiply-adds), then the fdlibm minimax kernels on [-pi/4, pi/4] (S1..S6, C1..C6 of
// __kernel_sin / __kernel_cos, Sun Microsystems, freely distributable) and the quadrant.  One basic
// block: a time step that calls it stays one block the scheduler can interleave.  Returns false for
// an argument outside its range (|x| >= 2^20, infinities, NaN); s and c mean nothing then.
ILQR_TRIG_FN bool fast_sincos(double x, double &s, double &c) {
    const bool in_range = __builtin_fabs(x) < 1048576.0;
    // q = the integer nearest to x 2/pi, by adding and subtracting 1.5 * 2^52: the sum's low mantissa
    // bits are that integer in two's complement, so the quadrant needs no conversion
    const double qm = __builtin_fma(x, 6.36619772367581382433e-01, 6755399441055744.0);
    const double q = qm - 6755399441055744.0;
    double r = __builtin_fma(q, -1.57079632679489655800e+00, x);
    r = __builtin_fma(q, -6.12323399573676603587e-17, r);
    r = __builtin_fma(q, 1.49738490485916983528e-33, r);
    const uint32_t n = (uint32_t)__builtin_bit_cast(uint64_t, qm);
    const double z = r * r;
    // sin(r) = r + r^3 (S1 + z (S2 + ...))
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(z * r, ps, r);
    // cos(r) = 1 - z/2 + z^2 (C1 + z (C2 + ...)), with the rounding error of 1 - z/2 recovered
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cr = w + __builtin_fma(z, z * pc, (1.0 - w) - hz);
    // quadrant: n mod 4 = 0: (s, c) ; 1: (c, -s) ; 2: (-s, -c) ; 3: (-c, s).  The sine polynomial's
    // value is negated in the quadrants 1 and 2 and the cosine polynomial's in 2 and 3, wherever they
    // end up: a shift and one bit operation on the high word each, then one exchange
    const uint64_t su = __builtin_bit_cast(uint64_t, sr), cu = __builtin_bit_cast(uint64_t, cr);
    const uint32_t sh = flip_sign_if((uint32_t)(su >> 32), (n + 1u) << 30);
    const uint32_t ch = flip_sign_if((uint32_t)(cu >> 32), n << 30);
    const double sq = __builtin_bit_cast(double, ((uint64_t)sh << 32) | (uint32_t)su);
    const double cq = __builtin_bit_cast(double, ((uint64_t)ch << 32) | (uint32_t)cu);
    s = (n & 1) ? cq : sq;
    c = (n & 1) ? sq : cq;
    return in_range;
}

} // namespace ilqr
