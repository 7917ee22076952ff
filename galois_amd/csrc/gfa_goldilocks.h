// gfa_goldilocks.h -- lazy arithmetic modulo p = 2^64 - 2^32 + 1 for the register NTT networks.
//
// Inside a network every value is a 96-bit two's-complement integer in three 32-bit limbs (G3): any representative of
// its residue class with |value| < 2^70.  Then
//   * add / sub are three carry-chained 32-bit instructions and never reduce (a radix-32 network grows a value by 5 bits);
//   * a product with a 64-bit twiddle first folds the operand to 64 bits (2^64 == 2^32 - 1: one 32x32+64 multiply-add per
//     round, two rounds), forms the 128-bit product with four v_mad_u64_u32 and folds it back with one more
//     (2^64 == 2^32 - 1, 2^96 == -1): no compare / select chains anywhere;
//   * only values that leave the registers (LDS exchange, global store) are brought to 64 bits / to the canonical [0, p).
// On the device the carry chains are inline assembly (clang's own lowering of the same expressions goes through 64-bit
// compares and v_cndmask); the portable expressions below them are what the host compiles, and
// tests/test_host_logic.py checks those against Python integers (the device path is pinned by the NTT parity tests).
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

struct G3 {
    gu32 lo, mid;
    int32_t hi; // value = lo + 2^32 * mid + 2^64 * hi
};

constexpr gu64 P = 0xFFFFFFFF00000001ull;

GFA_HD G3 from_u64(gu64 x) { return G3{(gu32)x, (gu32)(x >> 32), 0}; }

GFA_HD G3 add(G3 a, G3 b)
{
#ifdef __HIP_DEVICE_COMPILE__
    asm("v_add_co_u32 %0, vcc, %0, %3\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, %2, %5, vcc"
        : "+v"(a.lo), "+v"(a.mid), "+v"(a.hi)
        : "v"(b.lo), "v"(b.mid), "v"(b.hi)
        : "vcc");
    return a;
#else
    const gu64 s0 = (gu64)a.lo + b.lo;
    const gu64 s1 = (gu64)a.mid + b.mid + (s0 >> 32);
    return G3{(gu32)s0, (gu32)s1, (int32_t)((gu32)a.hi + (gu32)b.hi + (gu32)(s1 >> 32))};
#endif
}

GFA_HD G3 sub(G3 a, G3 b)
{
#ifdef __HIP_DEVICE_COMPILE__
    asm("v_sub_co_u32 %0, vcc, %0, %3\n\tv_subb_co_u32 %1, vcc, %1, %4, vcc\n\tv_subb_co_u32 %2, vcc, %2, %5, vcc"
        : "+v"(a.lo), "+v"(a.mid), "+v"(a.hi)
        : "v"(b.lo), "v"(b.mid), "v"(b.hi)
        : "vcc");
    return a;
#else
    const gu64 d0 = (gu64)a.lo - b.lo; // bit 63 set <=> borrow
    const gu64 d1 = (gu64)a.mid - b.mid - (d0 >> 63);
    return G3{(gu32)d0, (gu32)d1, (int32_t)((gu32)a.hi - (gu32)b.hi - (gu32)(d1 >> 63))};
#endif
}

// acc + a * (2^32 - 1) as (64-bit result, carry out)
GFA_HD gu64 mad_eps_carry(gu64 acc, gu32 a, gu32 *carry)
{
#ifdef __HIP_DEVICE_COMPILE__
    gu32 c;
    const gu32 zero = 0;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %0\n\tv_addc_co_u32 %1, vcc, 0, %3, vcc" : "+v"(acc), "=v"(c) : "v"(a), "v"(zero) : "vcc");
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
#ifdef __HIP_DEVICE_COMPILE__
    const gu32 k64 = 64u, km64 = 0xFFFFFFC0u, k63 = 63u;
    asm("v_add_co_u32 %0, vcc, %0, %3\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, %2, %5, vcc"
        : "+v"(x.lo), "+v"(x.mid), "+v"(x.hi)
        : "v"(k64), "v"(km64), "v"(k63)
        : "vcc");
    gu32 c;
    const gu64 w = mad_eps_carry(((gu64)x.mid << 32) | x.lo, (gu32)x.hi, &c);
    return w + (gu64)c * 0xFFFFFFFFu;
#else
    const gu64 s0 = (gu64)x.lo + 64u;
    const gu64 s1 = (gu64)x.mid + 0xFFFFFFC0u + (s0 >> 32);
    const gu32 h = (gu32)x.hi + 63u + (gu32)(s1 >> 32);
    gu32 c;
    const gu64 w = mad_eps_carry(((gu64)(gu32)s1 << 32) | (gu32)s0, h, &c);
    return w + (gu64)c * 0xFFFFFFFFu;
#endif
}

// canonical residue in [0, p) of any 64-bit representative
GFA_HD gu64 canon_u64(gu64 y)
{
#ifdef __HIP_DEVICE_COMPILE__
    // y + (2^32 - 1) == y - p (mod 2^64) carries exactly when y >= p; the carry then selects it through an arithmetic mask
    gu32 c;
    const gu64 t = mad_eps_carry(y, 1u, &c);
    const gu64 m = (gu64)0 - (gu64)c;
    return (t & m) | (y & ~m);
#else
    const gu64 t = y + 0xFFFFFFFFull;
    return t < y ? t : y;
#endif
}

GFA_HD gu64 canon(G3 x) { return canon_u64(to_u64(x)); }

// y * w for 64-bit y, w (w need not be canonical): 4 multiply-adds for the 128-bit product [r0 r1 r2 r3], then
// r0 + 2^32 r1 + (2^32 - 1) r2 - r3 as a G3 with -1 <= hi <= 1
GFA_HD G3 mul_u64(gu64 y, gu64 w)
{
    const gu32 y0 = (gu32)y, y1 = (gu32)(y >> 32), w0 = (gu32)w, w1 = (gu32)(w >> 32);
    const gu64 t0 = (gu64)y0 * w0;
    const gu64 t1 = (gu64)y0 * w1 + (t0 >> 32);
    const gu64 t2 = (gu64)y1 * w0 + (gu32)t1;
    const gu64 t3 = (gu64)y1 * w1 + (t1 >> 32) + (t2 >> 32); // < 2^64
    const gu64 lo64 = ((gu64)(gu32)t2 << 32) | (gu32)t0;    // r0 + 2^32 r1
    const gu32 r2 = (gu32)t3, r3 = (gu32)(t3 >> 32);
    gu32 c;
    const gu64 u = mad_eps_carry(lo64, r2, &c);
    return sub(G3{(gu32)u, (gu32)(u >> 32), (int32_t)c}, G3{r3, 0u, 0});
}

GFA_HD G3 mul(G3 x, gu64 w) { return mul_u64(to_u64(x), w); }

// x * 2^S (0 < S < 96, compile-time) WITHOUT a multiplication: 2 has order 192 modulo p (2^96 == -1), so every twiddle inside a
// radix-R <= 64 network of the canonical root 2^(192/R) is a power of two.  S = 32 q + r: the bit shift by r gives four limbs
// t0 + 2^32 t1 + 2^64 t2 + 2^96 t3 (t3 signed: the bits shifted out of the signed high limb), the word shift by q moves them up,
// and 2^64 == 2^32 - 1, 2^96 == -1, 2^128 == -2^32, 2^160 == 1 - 2^32 fold the positions back:
//   q = 0:  (t0 - t2 - t3) + 2^32 (t1 + t2)
//   q = 1:  (-t1 - t2)     + 2^32 (t0 + t1 - t3)
//   q = 2:  (-t0 - t1 + t3) + 2^32 (t0 - t2 - t3)
// as three to five carry-chained 96-bit add / sub (3 instructions each) after four shifts: 14-20 instructions against ~32 for
// a general product.  Any |x| < 2^94 (|hi| < 2^30); the result has |value| < 2^66.
template <int S>
GFA_HD G3 mul_pow2(G3 x)
{
    static_assert(S > 0 && S < 96, "shift out of range");
    constexpr int q = S / 32, r = S % 32;
    gu32 t0, t1, t2;
    int32_t t3;
    if (r == 0) {
        t0 = x.lo; t1 = x.mid; t2 = (gu32)x.hi; t3 = x.hi >> 31;
    } else {
        t0 = x.lo << r;
        t1 = (x.mid << r) | (x.lo >> (32 - r));
        t2 = ((gu32)x.hi << r) | (x.mid >> (32 - r));
        t3 = x.hi >> (32 - r);
    }
    const int32_t sx = t3 >> 31; // sign extension of t3
    if (q == 0) {
        G3 v = sub(G3{t0, t1, 0}, G3{t2, 0u, 0});
        v = add(v, G3{0u, t2, 0});
        return sub(v, G3{(gu32)t3, (gu32)sx, sx});
    } else if (q == 1) {
        G3 v = add(G3{0u, t0, 0}, G3{0u, t1, 0});
        v = sub(v, G3{t1, 0u, 0});
        v = sub(v, G3{t2, 0u, 0});
        return sub(v, G3{0u, (gu32)t3, sx});
    } else {
        G3 v = sub(G3{0u, t0, 0}, G3{t0, 0u, 0});
        v = sub(v, G3{t1, 0u, 0});
        v = sub(v, G3{0u, t2, 0});
        v = add(v, G3{(gu32)t3, (gu32)sx, sx});
        return sub(v, G3{0u, (gu32)t3, sx});
    }
}

} // namespace gl
} // namespace gfa
