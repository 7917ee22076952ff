// gfa_goldilocks.h -- lazy arithmetic modulo p = 2^64 - 2^32 + 1 for the register NTT networks.
//
// Inside a network every value is a 96-bit two's-complement integer (G3): any representative of its residue class with
// |value| < 2^70.  Then
//   * add / sub are three carry-chained 32-bit instructions and never reduce (a radix-32 network grows a value by 5 bits);
//   * a twiddle that is a power of two is a bit shift, a word rotation and one fold (2^64 == 2^32 - 1, 2^96 == -1):
//     9-12 instructions (mul_pow2_small);
//   * a product with a 64-bit twiddle first folds the operand to 64 bits, forms the 128-bit product with four
//     v_mad_u64_u32 and folds it back with 32-bit carry chains: no compare / select chains anywhere;
//   * only values that leave the registers (LDS exchange, global store) are brought to 64 bits / to the canonical [0, p).
// The value type is a native 96-bit integer (_BitInt(96); __int128 where the host compiler has no _BitInt in C++), so the
// SAME expressions compile for the device and for the host: tests/csrc/goldilocks_host_test.cpp checks them against 128-bit
// integer arithmetic, and the compiler -- not inline assembly with tied operands -- allocates the limbs (the round-3 form of
// this header spent a fifth of the kernel's vector instructions on register copies around its asm statements).
#pragma once
#include <cstdint>

#ifndef GFA_HD
#ifdef __HIPCC__
#define GFA_HD __host__ __device__ __forceinline__
#else
#define GFA_HD inline
#endif
#endif

namespace gfa {
namespace gl {

typedef uint32_t gu32;
typedef uint64_t gu64;
#if defined(__clang__)
typedef signed _BitInt(96) gint;
typedef unsigned _BitInt(96) guint;
#else
typedef __int128 gint; // host g++: the checked ranges never leave 96 bits
typedef unsigned __int128 guint;
#endif

struct G3 {
    gint v;
};

constexpr gu64 P = 0xFFFFFFFF00000001ull;

GFA_HD gu32 limb0(G3 x) { return (gu32)x.v; }
GFA_HD gu32 limb1(G3 x) { return (gu32)(x.v >> 32); }
GFA_HD int32_t limb2(G3 x) { return (int32_t)(gu32)(x.v >> 64); }
// lo + 2^32 * mid + 2^64 * hi (hi signed)
GFA_HD G3 from_limbs(gu32 lo, gu32 mid, int32_t hi)
{
    return G3{(gint)(guint)(((gu64)mid << 32) | lo) + (gint)hi * ((gint)1 << 64)};
}
GFA_HD G3 from_u64(gu64 x) { return G3{(gint)(guint)x}; }
GFA_HD G3 from_i64(int64_t x) { return G3{(gint)x}; }

GFA_HD G3 add(G3 a, G3 b) { return G3{a.v + b.v}; }
GFA_HD G3 sub(G3 a, G3 b) { return G3{a.v - b.v}; }

// acc + a * (2^32 - 1) as (64-bit result, carry out)
GFA_HD gu64 mad_eps_carry(gu64 acc, gu32 a, gu32 *carry)
{
#ifdef __HIP_DEVICE_COMPILE__
    gu32 c;
    // s_nop 1: the two wait states the gfx940 family wants between a VALU write of VCC and a VALU read of it as carry-in (the
    // compiler pads its own chains the same way; nothing inside an asm string is padded for us)
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %0\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc" : "+v"(acc), "=v"(c) : "v"(a) : "vcc");
    *carry = c;
    return acc;
#else
    const gu64 w = acc + (gu64)a * 0xFFFFFFFFu;
    *carry = w < acc ? 1u : 0u;
    return w;
#endif
}

// Any representative below 2^64 of a value >= -64 p (so -63 <= hi <= 63 is always safe).  64 p = [64, 2^32 - 64, 63] is
// added first so that the high limb h is in [0, 127]; h * 2^64 == h * (2^32 - 1) then goes into the low 64 bits with one
// multiply-add.  The sum is below 2^64 + 2^39, so a carry out leaves a small low part and the second fold cannot carry.
GFA_HD gu64 to_u64(G3 x)
{
    const gint bias = (gint)(guint)0xFFFFFFC000000040ull + (gint)63 * ((gint)1 << 64);
    const G3 y{x.v + bias};
    gu32 c;
    const gu64 w = mad_eps_carry((gu64)(guint)y.v, (gu32)limb2(y), &c);
    return w + (gu64)c * 0xFFFFFFFFu;
}

// canonical residue in [0, p) of any 64-bit representative
GFA_HD gu64 canon_u64(gu64 y)
{
    // y + (2^32 - 1) == y - p (mod 2^64) carries exactly when y >= p
    const gu64 t = y + 0xFFFFFFFFull;
    return t < y ? t : y;
}

GFA_HD gu64 canon(G3 x) { return canon_u64(to_u64(x)); }

// y * w for 64-bit y, w (w need not be canonical): 4 multiply-adds for the 128-bit product [r0 r1 r2 r3], then
// (r0 - r2 - r3) + 2^32 (r1 + r2): a G3 with -2 <= hi <= 1
GFA_HD G3 mul_u64(gu64 y, gu64 w)
{
    const gu32 y0 = (gu32)y, y1 = (gu32)(y >> 32), w0 = (gu32)w, w1 = (gu32)(w >> 32);
    const gu64 t0 = (gu64)y0 * w0;
    const gu64 t1 = (gu64)y0 * w1 + (t0 >> 32);
    const gu64 t2 = (gu64)y1 * w0 + (gu32)t1;
    const gu64 t3 = (gu64)y1 * w1 + (t1 >> 32) + (t2 >> 32); // < 2^64
    const gu32 r0 = (gu32)t0, r1 = (gu32)t2, r2 = (gu32)t3, r3 = (gu32)(t3 >> 32);
    const gint hi = ((gint)(guint)((gu64)r1 + r2)) << 32;
    return G3{hi + (gint)(guint)r0 - (gint)(guint)((gu64)r2 + r3)};
}

// y * w as a 64-bit representative (not canonical): the 128-bit product [r0 r1 r2 r3] is lo64 - r3 + (2^32 - 1) r2 modulo p; a
// borrow of the subtraction wraps by 2^64 == 2^32 - 1 and is taken back at once (the wrapped value is >= 2^64 - 2^32), a carry
// of the multiply-add likewise (the wrapped value is < (2^32 - 1)^2)
GFA_HD gu64 mul_red(gu64 y, gu64 w)
{
    const gu32 y0 = (gu32)y, y1 = (gu32)(y >> 32), w0 = (gu32)w, w1 = (gu32)(w >> 32);
    const gu64 t0 = (gu64)y0 * w0;
    const gu64 t1 = (gu64)y0 * w1 + (t0 >> 32);
    const gu64 t2 = (gu64)y1 * w0 + (gu32)t1;
    const gu64 t3 = (gu64)y1 * w1 + (t1 >> 32) + (t2 >> 32); // < 2^64
    const gu64 lo64 = ((gu64)(gu32)t2 << 32) | (gu32)t0;
    const gu32 r2 = (gu32)t3, r3 = (gu32)(t3 >> 32);
    gu64 t = lo64 - r3;
    t -= lo64 < (gu64)r3 ? 0xFFFFFFFFull : 0ull;
    gu32 c;
    const gu64 u = mad_eps_carry(t, r2, &c);
    return u + (gu64)c * 0xFFFFFFFFu;
}

GFA_HD G3 mul(G3 x, gu64 w) { return mul_u64(to_u64(x), w); }

// x * 2^S (0 < S < 96, compile-time) WITHOUT a multiplication: 2 has order 192 modulo p (2^96 == -1), so every twiddle inside a
// radix-R <= 64 network of the canonical root 2^(192/R) is a power of two.  S = 32 q + r: the bit shift by r gives four limbs
// t0 + 2^32 t1 + 2^64 t2 + 2^96 t3 (t3 signed: the bits shifted out of the signed high limb), the word shift by q moves them up,
// and 2^64 == 2^32 - 1, 2^96 == -1, 2^128 == -2^32, 2^160 == 1 - 2^32 fold the positions back:
//   q = 0:  (t0 - t2 - t3) + 2^32 (t1 + t2)
//   q = 1:  (-t1 - t2)     + 2^32 (t0 + t1 - t3)
//   q = 2:  (-t0 - t1 + t3) + 2^32 (t0 - t2 - t3)
// Any |x| < 2^94 (|hi| < 2^30); the result has |value| < 2^66.
template <int S>
GFA_HD G3 mul_pow2(G3 x)
{
    static_assert(S > 0 && S < 96, "shift out of range");
    constexpr int q = S / 32, r = S % 32;
    const gu32 a = limb0(x), b = limb1(x);
    const int32_t c = limb2(x);
    gu32 t0, t1, t2;
    int32_t t3;
    if (r == 0) {
        t0 = a; t1 = b; t2 = (gu32)c; t3 = c >> 31;
    } else {
        t0 = a << r;
        t1 = (b << r) | (a >> ((32 - r) & 31));
        t2 = ((gu32)c << r) | (b >> ((32 - r) & 31));
        t3 = c >> ((32 - r) & 31);
    }
    const gint T0 = (gint)(guint)t0, T1 = (gint)(guint)t1, T2 = (gint)(guint)t2, T3 = (gint)t3;
    if (q == 0) return G3{T0 - T2 - T3 + ((T1 + T2) << 32)};
    if (q == 1) return G3{-T1 - T2 + ((T0 + T1 - T3) << 32)};
    return G3{-T0 - T1 + T3 + ((T0 - T2 - T3) << 32)};
}

// The same product for |x| < 2^(95 - r), r = S mod 32 > 0: the shifted high limb t2 = (hi << r) | (mid >> (32 - r)) then holds
// its value as a SIGNED 32-bit number and no fourth limb exists:
//   q = 0:  (t0 - t2) + 2^32 (t1 + t2)
//   q = 1:  (-t1 - t2) + 2^32 (t0 + t1)
//   q = 2:  (-t0 - t1) + 2^32 (t0 - t2)
// 9 / 12 / 11 instructions.  Inside the radix-32 / radix-16 networks of the NTT kernel the operand of a shift by 32 q + r at
// butterfly level s has |x| < 2^(69 - s), which is within the precondition for every shift used: dif_shift_bounds_ok() below
// replays the networks' schedule on bounds, and tests/csrc/goldilocks_host_test.cpp runs both.  |result| < 2^66.
template <int S>
GFA_HD G3 mul_pow2_small(G3 x)
{
    static_assert(S > 0 && S < 96 && S % 32 != 0, "shift out of range");
    constexpr int q = S / 32, r = S % 32;
    const gu32 a = limb0(x), b = limb1(x);
    const int32_t c = limb2(x);
    const gu32 t0 = a << r;
    const gu32 t1 = (b << r) | (a >> (32 - r));
    const int32_t t2 = (int32_t)(((gu32)c << r) | (b >> (32 - r)));
    const gint T0 = (gint)(guint)t0, T1 = (gint)(guint)t1, T2 = (gint)t2;
    if (q == 0) return G3{T0 - T2 + ((T1 + T2) << 32)};
    if (q == 1) return G3{-T1 - T2 + ((T0 + T1) << 32)};
    return G3{-T0 - T1 + ((T0 - T2) << 32)};
}

// x * 2^(6k), k = 1 .. 15: after unrolling k is a constant and the switch folds to one instance
GFA_HD G3 mul_pow2_6k(G3 x, int k)
{
    switch (k) {
    case 1: return mul_pow2_small<6>(x);
    case 2: return mul_pow2_small<12>(x);
    case 3: return mul_pow2_small<18>(x);
    case 4: return mul_pow2_small<24>(x);
    case 5: return mul_pow2_small<30>(x);
    case 6: return mul_pow2_small<36>(x);
    case 7: return mul_pow2_small<42>(x);
    case 8: return mul_pow2_small<48>(x);
    case 9: return mul_pow2_small<54>(x);
    case 10: return mul_pow2_small<60>(x);
    case 11: return mul_pow2_small<66>(x);
    case 12: return mul_pow2_small<72>(x);
    case 13: return mul_pow2_small<78>(x);
    case 14: return mul_pow2_small<84>(x);
    default: return mul_pow2_small<90>(x);
    }
}

// Radix-R decimation-in-frequency network on the CANONICAL root w_R = 2^(192/R) (2 has order 192 modulo p), R = 2^LOGR <= 32:
// v[bitrev(k)] <- sum_a v[a] * w_R^(a k).  Every twiddle is a power of two (mul_pow2_small).  A transform whose root is
// w_R = canonical^u (u odd) is served by feeding the inputs in the order a' = u * a mod R -- the caller's job.
// Inputs: any |v[a]| <= 2^64 (the kernel feeds [0, 2^64)).  Bounds, with B(x) = bits of |x|: every butterfly adds one bit and a
// shifted value restarts at 66, so the operand of a shift at level s (half = 2^s, the first level being s = LOGR - 1) is
// below 2^(65 + LOGR - 1 - s) <= 2^69 -- the largest shift remainders r = 30, 28 only occur at the first two levels (odd and
// singly-even multiples of 6), where 65 <= 95 - 30 and 67 <= 95 - 28 hold: dif_shift_bounds_ok() replays exactly this.
template <int LOGR>
GFA_HD void dif_shift(G3 (&v)[1 << LOGR])
{
    constexpr int R = 1 << LOGR;
    constexpr int UNIT6 = 32 >> LOGR; // 2^(192/R) = 2^(6 * UNIT6)
#pragma unroll
    for (int s = LOGR - 1; s >= 0; s--) {
        const int half = 1 << s;
#pragma unroll
        for (int b = 0; b < R; b += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                const G3 u = v[b + j], x = v[b + j + half];
                v[b + j] = add(u, x);
                const int tj = j << (LOGR - 1 - s);
                if (tj != 0) v[b + j + half] = mul_pow2_6k(sub(u, x), tj * UNIT6);
                else v[b + j + half] = sub(u, x);
            }
        }
    }
}

// the schedule of dif_shift<LOGR> on magnitude bounds alone (bits[i]: |v[i]| < 2^bits[i], 64 on entry): true when every
// mul_pow2_small meets its precondition and every result stays below 2^70 (to_u64's range)
inline bool dif_shift_bounds_ok(int logr)
{
    const int R = 1 << logr, unit6 = 32 >> logr;
    int bits[32];
    for (int i = 0; i < R; i++) bits[i] = 64;
    for (int s = logr - 1; s >= 0; s--) {
        const int half = 1 << s;
        for (int b = 0; b < R; b += 2 * half)
            for (int j = 0; j < half; j++) {
                const int m = (bits[b + j] > bits[b + j + half] ? bits[b + j] : bits[b + j + half]) + 1;
                bits[b + j] = m;
                const int tj = j << (logr - 1 - s);
                if (tj != 0) {
                    const int r = (6 * tj * unit6) % 32;
                    if (r == 0 || m > 95 - r) return false;
                    bits[b + j + half] = 66;
                } else bits[b + j + half] = m;
            }
    }
    for (int i = 0; i < R; i++)
        if (bits[i] > 70) return false;
    return true;
}

} // namespace gl
} // namespace gfa
