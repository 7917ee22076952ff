// gfa_arith.h -- scalar finite-field arithmetic for gfx950 kernels (and, compiled with g++, for the
// host-side formula tests in tests/csrc/).  Everything here is exact integer arithmetic: any correct
// formula yields the reference's bits, so the device uses reductions that suit the CDNA4 VALU
// (Barrett / Montgomery / Goldilocks folding, clz-driven binary EGCD) instead of the reference's
// `%`, extended Euclid and shift-xor loops (reference: src/galois/_domains/_calculate.py:133-592,
// _lookup.py:31-270).  Zero-handling and error semantics follow the reference exactly.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GFA_HD __host__ __device__ __forceinline__
#else
#define GFA_HD inline
#endif
#include "gfa_goldilocks.h"

namespace gfa {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

enum Kind : u32 {
    KIND_PRIME32 = 0,    // GF(p), p < 2^32: Barrett reduction of the 64-bit product
    KIND_PRIME64 = 1,    // GF(p), odd p < 2^64: Montgomery
    KIND_GOLDILOCKS = 2, // GF(2^64 - 2^32 + 1): folding reduction
    KIND_BIN = 3,        // GF(2^m), m <= 63: carry-less shift/xor, clz-driven polynomial EGCD inverse
    KIND_LUT = 4,        // any GF(p^m) with q <= 2^20: EXP/LOG/Zech tables (the reference's lookup mode)
    KIND_EXT = 5,        // GF(p^m), p odd < 2^32, m <= 16, q < 2^64: base-p digit vectors
};

#define GFA_MAX_EXT_DEGREE 16

// Passed by value as a kernel argument (fits comfortably in kernarg space).
struct FieldDev {
    u64 p;      // characteristic
    u64 q;      // order p^m (0 for 2^64, which is not supported)
    u32 m;      // degree
    u32 kind;   // Kind used for explicit calculation
    u64 irr;    // GF(2^m): full irreducible polynomial (bit m set)
    u64 mu;     // PRIME32/EXT: floor(2^64 / p)
    u64 nprime; // PRIME64: -p^-1 mod 2^64
    u64 r2;     // PRIME64: 2^128 mod p.  EXT (r05): 1 when mul_m_small may skip its intermediate reductions (bound replayed at creation)
    u32 qm1;    // LUT: q - 1
    u32 zech_e; // LUT: (q-1)/2 for odd characteristic, 0 for characteristic 2
    const u32 *exp_tab;  // LUT: 2q entries (second half = EXP[1..q], as _lookup.py:371)
    const u32 *log_tab;  // LUT: q entries, LOG[0] = 0 placeholder
    const u32 *zech_tab; // LUT: q entries
    u32 ext_irr[GFA_MAX_EXT_DEGREE]; // EXT: irreducible polynomial minus x^m, digits of degree m-1..0
};

GFA_HD u64 mulhi64(u64 a, u64 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

GFA_HD int clz64(u64 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return x ? __builtin_clzll(x) : 64;
#endif
}

GFA_HD int clz32(u32 x) { return clz64((u64)x) - 32; }
GFA_HD int ctz32(u32 x) { return __builtin_ctz(x); } // x != 0

// ------------------------------------------------------------------------------------------------
// GF(p), p < 2^32
// ------------------------------------------------------------------------------------------------
struct Prime32 {
    typedef u32 elem;
    static GFA_HD u32 reduce64(const FieldDev &f, u64 t)
    {
        // Barrett: qh = floor(t * mu / 2^64) underestimates floor(t / p) by at most 2
        u64 qh = mulhi64(t, f.mu);
        u64 r = t - qh * f.p;
        if (r >= f.p) r -= f.p;
        if (r >= f.p) r -= f.p;
        return (u32)r;
    }
    static GFA_HD u32 add(const FieldDev &f, u32 a, u32 b)
    {
        u64 c = (u64)a + b;
        if (c >= f.p) c -= f.p;
        return (u32)c;
    }
    static GFA_HD u32 sub(const FieldDev &f, u32 a, u32 b) { return a >= b ? a - b : (u32)(f.p + a - b); }
    static GFA_HD u32 neg(const FieldDev &f, u32 a) { return a == 0 ? 0 : (u32)(f.p - a); }
    static GFA_HD u32 mul(const FieldDev &f, u32 a, u32 b) { return reduce64(f, (u64)a * b); }
    static GFA_HD u32 one(const FieldDev &) { return 1; }
    static GFA_HD u32 pow_barrett(const FieldDev &f, u32 a, u64 e)
    {
        u32 r = 1;
        while (e) {
            if (e & 1) r = mul(f, r, a);
            a = mul(f, a, a);
            e >>= 1;
        }
        return r;
    }
    // Montgomery reduction of x < p * 2^32 for odd p < 2^31: (x + (x * ninv mod 2^32) * p) / 2^32, ninv = -p^-1 mod 2^32
    static GFA_HD u32 redc32(u64 x, u32 p, u32 ninv)
    {
        const u32 m = (u32)x * ninv;
        const u32 t = (u32)((x + (u64)m * p) >> 32); // < 2p < 2^32; the sum stays below 2^64 because x, m * p < p * 2^32 <= 2^63
        const u32 d = t - p;
        return d < t ? d : t; // min(t, t - p): t - p wraps above t exactly when t < p (no compare-and-select on the condition mask)
    }
    // a^e.  For odd p < 2^31 the whole exponentiation runs in Montgomery form -- 6 instructions per product instead of the
    // 14 of a 64-bit Barrett reduction; the constants come from the descriptor: 2^64 mod p = -mu * p (mod 2^64), -p^-1 by
    // Newton's iteration.  (2^31 < p < 2^32: the 64-bit sum inside redc32 could overflow; p = 2 has nothing to reduce.)
    static GFA_HD u32 pow_u(const FieldDev &f, u32 a, u64 e)
    {
        const u32 p = (u32)f.p;
        if (!(p & 1) || (p >> 31) || e < 4) return pow_barrett(f, a, e);
        u32 pinv = p; // p * pinv == 1 (mod 2^32): 3 correct bits to start with, doubled by every step
        for (int i = 0; i < 4; i++) pinv *= 2u - p * pinv;
        const u32 ninv = 0u - pinv;
        const u32 r2 = (u32)((u64)0 - f.mu * f.p); // 2^64 mod p  (mu = floor(2^64 / p))
        const u32 am = redc32((u64)a * r2, p, ninv);
        u32 r = am; // the leading one of e
        for (int i = 62 - clz64(e); i >= 0; i--) {
            r = redc32((u64)r * r, p, ninv);
            if ((e >> i) & 1) r = redc32((u64)r * am, p, ninv);
        }
        return redc32((u64)r, p, ninv);
    }
    // a != 0.  Fermat: a^(p-2) (same value as the reference's extended Euclid, _calculate.py:395-417).
    static GFA_HD u32 inv(const FieldDev &f, u32 a) { return f.p == 2 ? a : pow_u(f, a, f.p - 2); }
    static GFA_HD u32 from_int(const FieldDev &f, i64 k)
    { // integer -> prime subfield (np.mod(int, characteristic), _ufunc.py:399)
        i64 r = k % (i64)f.p;
        if (r < 0) r += (i64)f.p;
        return (u32)r;
    }
};

// ------------------------------------------------------------------------------------------------
// GF(p), odd p < 2^64 (Montgomery; operands and results are plain residues)
// ------------------------------------------------------------------------------------------------
struct Prime64 {
    typedef u64 elem;
    static GFA_HD u64 redc(const FieldDev &f, u64 lo, u64 hi)
    { // (hi*2^64 + lo) * 2^-64 mod p, for hi*2^64+lo < p*2^64
        u64 mq = lo * f.nprime;
        u64 mph = mulhi64(mq, f.p);
        // lo + (mq*p mod 2^64) == 0 mod 2^64, carrying 1 unless lo == 0
        u64 carry = lo != 0;
        u64 r = hi + mph;
        bool ov = r < hi;
        u64 r2 = r + carry;
        ov |= r2 < r;
        if (ov || r2 >= f.p) r2 -= f.p;
        return r2;
    }
    static GFA_HD u64 montmul(const FieldDev &f, u64 a, u64 b) { return redc(f, a * b, mulhi64(a, b)); }
    static GFA_HD u64 add(const FieldDev &f, u64 a, u64 b)
    {
        u64 c = a + b;
        if (c < a || c >= f.p) c -= f.p;
        return c;
    }
    static GFA_HD u64 sub(const FieldDev &f, u64 a, u64 b) { return a >= b ? a - b : a + (f.p - b); }
    static GFA_HD u64 neg(const FieldDev &f, u64 a) { return a == 0 ? 0 : f.p - a; }
    static GFA_HD u64 mul(const FieldDev &f, u64 a, u64 b) { return montmul(f, montmul(f, a, b), f.r2); }
    static GFA_HD u64 one(const FieldDev &) { return 1; }
    static GFA_HD u64 pow_u(const FieldDev &f, u64 a, u64 e)
    {
        u64 r = 1;
        while (e) {
            if (e & 1) r = mul(f, r, a);
            a = mul(f, a, a);
            e >>= 1;
        }
        return r;
    }
    static GFA_HD u64 inv(const FieldDev &f, u64 a) { return pow_u(f, a, f.p - 2); }
    static GFA_HD u64 from_int(const FieldDev &f, i64 k)
    {
        if (k >= 0) return (u64)k % f.p;
        u64 r = ((u64)0 - (u64)k) % f.p; // |k| mod p
        return r == 0 ? 0 : f.p - r;
    }
};

// ------------------------------------------------------------------------------------------------
// GF(2^64 - 2^32 + 1)
// ------------------------------------------------------------------------------------------------
struct Goldilocks {
    typedef u64 elem;
    static constexpr u64 P = 0xFFFFFFFF00000001ull;
    static constexpr u64 EPS = 0xFFFFFFFFull; // 2^32 - 1 == 2^64 mod p
    static GFA_HD u64 reduce128(u64 lo, u64 hi)
    {
        // 2^64 == 2^32 - 1 and 2^96 == -1 (mod p)
        u64 hi_hi = hi >> 32, hi_lo = hi & EPS;
        u64 t0 = lo - hi_hi;
        if (lo < hi_hi) t0 -= EPS; // borrowed 2^64 == EPS (mod p)
        u64 t1 = hi_lo * EPS;
        u64 t2 = t0 + t1;
        if (t2 < t0) t2 += EPS;
        if (t2 >= P) t2 -= P;
        return t2;
    }
    static GFA_HD u64 add(const FieldDev &, u64 a, u64 b)
    {
        u64 c = a + b;
        if (c < a || c >= P) c -= P;
        return c;
    }
    static GFA_HD u64 sub(const FieldDev &, u64 a, u64 b) { return a >= b ? a - b : a + (P - b); }
    static GFA_HD u64 neg(const FieldDev &, u64 a) { return a == 0 ? 0 : P - a; }
    // products through gl::mul_red (gfa_goldilocks.h): 4 multiply-adds and a fold on carries, any 64-bit representatives in
    // and out; chains (powers, the inversion ladder) stay on such representatives and only their result is brought to [0, p)
    static GFA_HD u64 mul_lazy(u64 a, u64 b) { return gl::mul_red(a, b); }
    static GFA_HD u64 mul(const FieldDev &, u64 a, u64 b) { return gl::canon_u64(gl::mul_red(a, b)); }
    static GFA_HD u64 one(const FieldDev &) { return 1; }
    static GFA_HD u64 pow_u(const FieldDev &, u64 a, u64 e)
    {
        u64 r = 1;
        while (e) {
            if (e & 1) r = mul_lazy(r, a);
            a = mul_lazy(a, a);
            e >>= 1;
        }
        return gl::canon_u64(r);
    }
    static GFA_HD u64 sqn_lazy(u64 a, int k)
    {
#pragma unroll 1
        for (int i = 0; i < k; i++) a = mul_lazy(a, a);
        return a;
    }
    // a^(p-2), p - 2 = 2^64 - 2^32 - 1 = (2^31 - 1) * 2^33 + (2^32 - 1): 64 squarings and 9 products along the chain
    // 2^k - 1 for k = 2, 3, 6, 12, 24, 30, 31, 32 (binary square-and-multiply takes 64 + 63)
    static GFA_HD u64 inv(const FieldDev &f, u64 a)
    {
        const u64 x2 = mul_lazy(sqn_lazy(a, 1), a);
        const u64 x3 = mul_lazy(sqn_lazy(x2, 1), a);
        const u64 x6 = mul_lazy(sqn_lazy(x3, 3), x3);
        const u64 x12 = mul_lazy(sqn_lazy(x6, 6), x6);
        const u64 x24 = mul_lazy(sqn_lazy(x12, 12), x12);
        const u64 x30 = mul_lazy(sqn_lazy(x24, 6), x6);
        const u64 x31 = mul_lazy(sqn_lazy(x30, 1), a);
        const u64 x32 = mul_lazy(sqn_lazy(x31, 1), a);
        return gl::canon_u64(mul_lazy(sqn_lazy(x31, 33), x32));
    }
    static GFA_HD u64 from_int(const FieldDev &, i64 k)
    {
        if (k >= 0) return (u64)k % P;
        u64 r = ((u64)0 - (u64)k) % P;
        return r == 0 ? 0 : P - r;
    }
};

// ------------------------------------------------------------------------------------------------
// GF(2^m), m <= 63
// ------------------------------------------------------------------------------------------------
struct Bin {
    typedef u64 elem;
    static GFA_HD u64 add(const FieldDev &, u64 a, u64 b) { return a ^ b; }
    static GFA_HD u64 sub(const FieldDev &, u64 a, u64 b) { return a ^ b; }
    static GFA_HD u64 neg(const FieldDev &, u64 a) { return a; }
    // m <= 32 on the device: the same shift-and-xor product with a fixed m steps, no data-dependent branch and 32-bit
    // registers -- per step one sign-extended bit extract of b, and / xor into the result, and a masked reduction
    // (7 vector instructions; the 64-bit data-dependent loop below diverges inside a wavefront)
    static GFA_HD u32 mul32(const FieldDev &f, u32 a, u32 b)
    {
        const int m = (int)f.m;
        const u32 red = (u32)f.irr; // m < 32: includes bit m, which cancels the bit shifted out of the field; m = 32: low word
        u32 c = 0;
        for (int i = 0; i < m; i++) {
            const u32 bm = (u32)(((int32_t)(b << (31 - i))) >> 31); // all-ones iff bit i of b
            c ^= a & bm;
            const u32 hm = (u32)(((int32_t)(a << (32 - m))) >> 31); // all-ones iff bit m-1 of a
            a = (a << 1) ^ (red & hm);
        }
        return c;
    }
    // Carry-less 32 x 32 -> 64-bit product out of INTEGER multiplies ("multiplication with holes"): split each operand into
    // the four classes of bit positions mod 4; within a class the set bits are four apart, so in an integer product of two
    // classes every position receives at most 8 partial ones -- a sum that fits the four bits up to the next position of its
    // class, and whose lowest bit is the parity, i.e. the carry-less sum.  16 products (v_mad_u64_u32), xor by result class, mask.
    static GFA_HD u64 clmul32(u32 x, u32 y)
    {
        const u32 x0 = x & 0x11111111u, x1 = x & 0x22222222u, x2 = x & 0x44444444u, x3 = x & 0x88888888u;
        const u32 y0 = y & 0x11111111u, y1 = y & 0x22222222u, y2 = y & 0x44444444u, y3 = y & 0x88888888u;
        const u64 z0 = ((u64)x0 * y0) ^ ((u64)x1 * y3) ^ ((u64)x2 * y2) ^ ((u64)x3 * y1);
        const u64 z1 = ((u64)x0 * y1) ^ ((u64)x1 * y0) ^ ((u64)x2 * y3) ^ ((u64)x3 * y2);
        const u64 z2 = ((u64)x0 * y2) ^ ((u64)x1 * y1) ^ ((u64)x2 * y0) ^ ((u64)x3 * y3);
        const u64 z3 = ((u64)x0 * y3) ^ ((u64)x1 * y2) ^ ((u64)x2 * y1) ^ ((u64)x3 * y0);
        return (z0 & 0x1111111111111111ull) | (z1 & 0x2222222222222222ull) | (z2 & 0x4444444444444444ull) |
               (z3 & 0x8888888888888888ull);
    }
    // operands below 2^21: three classes of positions mod 3 are enough (at most 7 set bits per class, sums fit three bits) --
    // 9 products instead of 16
    static GFA_HD u64 clmul21(u32 x, u32 y)
    {
        const u32 x0 = x & 0x49249249u, x1 = x & 0x92492492u, x2 = x & 0x24924924u;
        const u32 y0 = y & 0x49249249u, y1 = y & 0x92492492u, y2 = y & 0x24924924u;
        const u64 z0 = ((u64)x0 * y0) ^ ((u64)x1 * y2) ^ ((u64)x2 * y1);
        const u64 z1 = ((u64)x0 * y1) ^ ((u64)x1 * y0) ^ ((u64)x2 * y2);
        const u64 z2 = ((u64)x0 * y2) ^ ((u64)x1 * y1) ^ ((u64)x2 * y0);
        return (z0 & 0x9249249249249249ull) | (z1 & 0x2492492492492492ull) | (z2 & 0x4924924924924924ull);
    }
    // 16 x 16 -> 31-bit carry-less products of the LOW halves / of the HIGH halves of two registers, nine integer multiplies each
    // (three classes of bit positions mod 3, at most six set bits per class: a position of an integer product of two classes
    // receives at most six partial ones, which fits the three bits up to the next position of its class; the lowest bit of the
    // sum is the carry-less sum).  The high halves are multiplied in place: (x << 16)(y << 16) = xy << 32, i.e. the high word.
    static GFA_HD u32 holes16_pick(u32 z0, u32 z1, u32 z2) { return (z0 & 0x49249249u) | (z1 & 0x92492492u) | (z2 & 0x24924924u); }
    static GFA_HD u32 clmul16_lo(u32 x, u32 y)
    {
        const u32 x0 = x & 0x9249u, x1 = x & 0x2492u, x2 = x & 0x4924u, y0 = y & 0x9249u, y1 = y & 0x2492u, y2 = y & 0x4924u;
        return holes16_pick((x0 * y0) ^ (x1 * y2) ^ (x2 * y1), (x0 * y1) ^ (x1 * y0) ^ (x2 * y2), (x0 * y2) ^ (x1 * y1) ^ (x2 * y0));
    }
    static GFA_HD u32 mulhi32(u32 a, u32 b) { return (u32)(((u64)a * b) >> 32); }
    static GFA_HD u32 clmul16_hi(u32 x, u32 y)
    {
        const u32 x0 = x & 0x92490000u, x1 = x & 0x24920000u, x2 = x & 0x49240000u;
        const u32 y0 = y & 0x92490000u, y1 = y & 0x24920000u, y2 = y & 0x49240000u;
        return holes16_pick(mulhi32(x0, y0) ^ mulhi32(x1, y2) ^ mulhi32(x2, y1), mulhi32(x0, y1) ^ mulhi32(x1, y0) ^ mulhi32(x2, y2),
                            mulhi32(x0, y2) ^ mulhi32(x1, y1) ^ mulhi32(x2, y0));
    }
    // v(x) mod f for v below 2^(m + extra): the entries of the byte-indexed reduction tables (h(x) x^(m + 8k) mod f)
    static GFA_HD u64 reduce_bits(u64 v, int m, int extra, u64 irr)
    {
        for (int bit = m + extra - 1; bit >= m; bit--)
            if ((v >> bit) & 1u) v ^= irr << (bit - m);
        return v;
    }
    // rounds of "fold the part above x^m back through g = f - x^m" that clear a (2m-1)-bit product, 0 when that costs more than
    // the bit-serial product (g with many terms or of high degree).  Stored in FieldDev::mu of a binary field (gfa_field.hip).
    static GFA_HD u32 fold_rounds(u64 irr, u32 m)
    {
        if (m > 32 || m < 2) return 0;
        const u64 g = irr ^ ((u64)1 << m);
        int dg = -1, w = 0;
        for (int i = 0; i < 64; i++)
            if ((g >> i) & 1) { dg = i; w++; }
        int deg = 2 * (int)m - 2 - (int)m; // degree of the part above x^m
        u32 r = 0;
        while (deg >= 0) {
            r++;
            deg = deg + dg - (int)m;
            if (r > 8) return 0;
        }
        // measured on MI355X (tools/ew_bench.py --bin): bit-serial ~7 issue slots per bit of m; this form ~60 (16 products) or
        // ~35 (9 products, m <= 21) plus ~4 per term of g and round
        const u32 fold = (m <= 21 ? 35u : 60u) + r * (4u * (u32)w + 4u);
        return (10u * fold < 9u * 7u * m) ? r : 0;
    }
    // a * b mod f for m <= 32: clmul32, then `rounds` folds; the loop over the terms of g is uniform (scalar) on the device
    static GFA_HD u32 mul_fold(const FieldDev &f, u32 a, u32 b)
    {
        const u32 m = f.m;
        const u64 low = (((u64)1 << m) - 1);
        const u32 g = (u32)(f.irr ^ ((u64)1 << m));
        u64 P = m <= 21 ? clmul21(a, b) : clmul32(a, b);
        for (u32 r = 0; r < (u32)f.mu; r++) {
            const u64 H = P >> m;
            P &= low;
            for (u32 gg = g; gg; gg &= gg - 1) P ^= H << ctz32(gg);
        }
        return (u32)P;
    }
    static GFA_HD u64 mul(const FieldDev &f, u64 a, u64 b)
    { // shift-and-xor with reduction by the irreducible polynomial (value-identical to _calculate.py:308-324)
        if (f.mu) return mul_fold(f, (u32)a, (u32)b);
#if defined(__HIP_DEVICE_COMPILE__)
        if (f.m <= 32) return mul32(f, (u32)a, (u32)b);
#endif
        u64 c = 0;
        const u64 top = (u64)1 << (f.m - 1);
        const u64 red = f.irr ^ ((u64)1 << f.m); // low m bits of the irreducible polynomial
        while (b) {
            if (b & 1) c ^= a;
            b >>= 1;
            u64 carry = a & top;
            a = (a ^ carry) << 1;
            if (carry) a ^= red;
        }
        return c;
    }
    static GFA_HD u64 one(const FieldDev &) { return 1; }
    static GFA_HD u64 pow_u(const FieldDev &f, u64 a, u64 e)
    {
        u64 r = 1;
        while (e) {
            if (e & 1) r = mul(f, r, a);
            a = mul(f, a, a);
            e >>= 1;
        }
        return r;
    }
    // a != 0: polynomial extended Euclid over GF(2), one clz per step (replaces Itoh-Tsujii, _calculate.py:469-489)
    static GFA_HD u64 inv(const FieldDev &f, u64 a)
    {
        u64 u = a, v = f.irr, g1 = 1, g2 = 0;
        while (u != 1) {
            int j = clz64(v) - clz64(u); // deg(u) - deg(v)
            if (j < 0) {
                u64 t = u; u = v; v = t;
                t = g1; g1 = g2; g2 = t;
                j = -j;
            }
            u ^= v << j;
            g1 ^= g2 << j;
        }
        return g1;
    }
    static GFA_HD u64 from_int(const FieldDev &, i64 k) { return (u64)(k & 1); }
};

// ------------------------------------------------------------------------------------------------
// Lookup tables (EXP / LOG / Zech), q <= 2^20.  Formulas are the reference's, _lookup.py:31-270.
// ------------------------------------------------------------------------------------------------
struct Lut {
    typedef u32 elem;
    static GFA_HD u32 add(const FieldDev &f, u32 a, u32 b)
    {
        if (f.p == 2) return a ^ b;
        if (f.m == 1) { u64 c = (u64)a + b; if (c >= f.p) c -= f.p; return (u32)c; }
        if (a == 0) return b;
        if (b == 0) return a;
        u32 mm = f.log_tab[a], nn = f.log_tab[b];
        if (mm > nn) { u32 t = mm; mm = nn; nn = t; }
        if (nn - mm == f.zech_e) return 0;
        return f.exp_tab[mm + f.zech_tab[nn - mm]];
    }
    static GFA_HD u32 neg(const FieldDev &f, u32 a)
    {
        if (f.p == 2) return a;
        if (f.m == 1) return a == 0 ? 0 : (u32)(f.p - a);
        if (a == 0) return 0;
        return f.exp_tab[f.zech_e + f.log_tab[a]];
    }
    static GFA_HD u32 sub(const FieldDev &f, u32 a, u32 b)
    {
        if (f.p == 2) return a ^ b;
        if (f.m == 1) return a >= b ? a - b : (u32)(f.p + a - b);
        if (b == 0) return a;
        u32 nn = f.log_tab[b] + f.zech_e;
        if (a == 0) return f.exp_tab[nn];
        u32 mm = f.log_tab[a];
        if (mm > nn) { u32 t = mm; mm = nn; nn = t; }
        u32 z = nn - mm;
        if (z == f.zech_e) return 0;
        if (z >= f.qm1) z -= f.qm1;
        return f.exp_tab[mm + f.zech_tab[z]];
    }
    static GFA_HD u32 mul(const FieldDev &f, u32 a, u32 b)
    {
        if (a == 0 || b == 0) return 0;
        return f.exp_tab[f.log_tab[a] + f.log_tab[b]];
    }
    static GFA_HD u32 one(const FieldDev &) { return 1; }
    static GFA_HD u32 inv(const FieldDev &f, u32 a) { return f.exp_tab[f.qm1 - f.log_tab[a]]; }
    static GFA_HD u32 div_nz(const FieldDev &f, u32 a, u32 b)
    { // b != 0
        if (a == 0) return 0;
        return f.exp_tab[f.qm1 + f.log_tab[a] - f.log_tab[b]];
    }
    // x mod m for x < 2^52, 0 < m < 2^21, through one double-precision quotient estimate (exact after one correction either
    // way): a 64-bit `%` is a ~100-instruction software division on the device, and pow_nz needs two per element
    static GFA_HD u64 mod52(u64 x, u64 m)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const double q = __builtin_floor((double)x * (1.0 / (double)m));
        i64 r = (i64)x - (i64)q * (i64)m;
        if (r < 0) r += (i64)m;
        if (r >= (i64)m) r -= (i64)m;
        return (u64)r;
#else
        return x % m;
#endif
    }
    // a != 0, any signed exponent: EXP[(LOG[a] * b) mod (q-1)] with a floor modulo (power_ufunc.lookup, _lookup.py:247-270)
    static GFA_HD u32 pow_nz(const FieldDev &f, u32 a, i64 e)
    {
        const u64 m = f.qm1; // < 2^20
        const u64 ae = e < 0 ? (u64)0 - (u64)e : (u64)e;
        u64 em = ae < ((u64)1 << 52) ? mod52(ae, m) : ae % m; // reduce first: mathematically identical, cannot overflow (the reference's TODO)
        if (e < 0 && em != 0) em = m - em;
        const u64 idx = mod52((u64)f.log_tab[a] * em, m); // < 2^40
        return f.exp_tab[idx];
    }
    static GFA_HD u32 pow_u(const FieldDev &f, u32 a, u64 e)
    {
        if (e == 0) return 1;
        if (a == 0) return 0;
        const u64 m = f.qm1;
        const u64 em = e < ((u64)1 << 52) ? mod52(e, m) : e % m;
        const u64 idx = mod52((u64)f.log_tab[a] * em, m);
        return f.exp_tab[idx];
    }
    static GFA_HD u32 from_int(const FieldDev &f, i64 k)
    {
        i64 r = k % (i64)f.p;
        if (r < 0) r += (i64)f.p;
        return (u32)r;
    }
};

// ------------------------------------------------------------------------------------------------
// GF(p^m), p odd, explicit digit-vector arithmetic (reference: *_vector.calculate, _calculate.py:150-383)
// ------------------------------------------------------------------------------------------------
struct Ext {
    typedef u64 elem;
    static GFA_HD void to_vec(const FieldDev &f, u64 a, u32 *v)
    { // most significant digit first, like int_to_vector (_calculate.py:22-33).  Division by p is a multiplication by
      // mu = floor(2^64 / p) with at most two corrections (hardware 64-bit div/mod costs ~100 instructions each and
      // used to dominate every GF(p^m) operation); 32-bit arithmetic when the whole element fits 32 bits.
        if (f.q <= 0xffffffffull) {
            u32 x = (u32)a;
            const u32 p32 = (u32)f.p, mu32 = (u32)(f.mu >> 32); // floor(2^32 / p)
            for (int i = (int)f.m - 1; i >= 0; i--) {
#if defined(__HIP_DEVICE_COMPILE__)
                u32 qd = __umulhi(x, mu32);
#else
                u32 qd = (u32)(((u64)x * mu32) >> 32);
#endif
                u32 r = x - qd * p32;
                if (r >= p32) { r -= p32; qd++; }
                if (r >= p32) { r -= p32; qd++; }
                v[i] = r;
                x = qd;
            }
        } else {
            for (int i = (int)f.m - 1; i >= 0; i--) {
                u64 qd = mulhi64(a, f.mu);
                u64 r = a - qd * f.p;
                if (r >= f.p) { r -= f.p; qd++; }
                if (r >= f.p) { r -= f.p; qd++; }
                v[i] = (u32)r;
                a = qd;
            }
        }
    }
    static GFA_HD u64 from_vec(const FieldDev &f, const u32 *v)
    {
        u64 a = 0;
        for (u32 i = 0; i < f.m; i++) a = a * f.p + v[i];
        return a;
    }
    // ---- degree known at compile time (2 <= M <= 8): every digit array lives in registers and the loops unroll; the
    // run-time-m versions below index their arrays dynamically, which puts them in scratch memory ----
    template <int M>
    static GFA_HD void to_vec_m(const FieldDev &f, u64 a, u32 (&v)[M])
    {
        if (f.q <= 0xffffffffull) {
            u32 x = (u32)a;
            const u32 p32 = (u32)f.p, mu32 = (u32)(f.mu >> 32);
#pragma unroll
            for (int i = M - 1; i >= 0; i--) {
#if defined(__HIP_DEVICE_COMPILE__)
                u32 qd = __umulhi(x, mu32);
#else
                u32 qd = (u32)(((u64)x * mu32) >> 32);
#endif
                u32 r = x - qd * p32;
                if (r >= p32) { r -= p32; qd++; }
                if (r >= p32) { r -= p32; qd++; }
                v[i] = r;
                x = qd;
            }
        } else {
#pragma unroll
            for (int i = M - 1; i >= 0; i--) {
                u64 qd = mulhi64(a, f.mu);
                u64 r = a - qd * f.p;
                if (r >= f.p) { r -= f.p; qd++; }
                if (r >= f.p) { r -= f.p; qd++; }
                v[i] = (u32)r;
                a = qd;
            }
        }
    }
    template <int M>
    static GFA_HD u64 from_vec_m(const FieldDev &f, const u32 (&v)[M])
    {
        u64 a = 0;
#pragma unroll
        for (int i = 0; i < M; i++) a = a * f.p + v[i];
        return a;
    }
    static GFA_HD u32 red64(const FieldDev &f, u64 x)
    { // x mod p, any 64-bit x (mu = floor(2^64 / p): the estimate is short by at most 2)
        u64 r = x - mulhi64(x, f.mu) * f.p;
        if (r >= f.p) r -= f.p;
        if (r >= f.p) r -= f.p;
        return (u32)r;
    }
    // OP: 0 add, 1 sub, 2 neg (b unused)
    template <int M, int OP>
    static GFA_HD u64 lin_m(const FieldDev &f, u64 a, u64 b)
    {
        u32 av[M], bv[M];
        to_vec_m<M>(f, a, av);
        if (OP != 2) to_vec_m<M>(f, b, bv);
#pragma unroll
        for (int i = 0; i < M; i++) av[i] = OP == 0 ? Prime32::add(f, av[i], bv[i]) : OP == 1 ? Prime32::sub(f, av[i], bv[i]) : Prime32::neg(f, av[i]);
        return from_vec_m<M>(f, av);
    }
    // schoolbook product with the coefficients left unreduced in 64 bits ((2M - 1) p^2 < 2^64: always for M >= 3, and for
    // M = 2 when p < 2^31), the top M - 1 coefficients folded back through x^M = -(irr), one reduction per coefficient
    // x mod p for x < 2^32 (mu32 = floor(2^32 / p): the estimate is short by at most 2)
    static GFA_HD u32 red32(u32 x, u32 p32, u32 mu32)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const u32 qd = __umulhi(x, mu32);
#else
        const u32 qd = (u32)(((u64)x * mu32) >> 32);
#endif
        u32 r = x - qd * p32;
        u32 d = r - p32;
        r = d < r ? d : r; // min(r, r - p): r - p wraps above r exactly when r < p
        d = r - p32;
        return d < r ? d : r;
    }
    // The same product for p < 2^13 (and M <= 8) entirely in 32-bit registers: every coefficient stays below (2M - 1) p^2 < 2^30,
    // so the sums need no 64-bit accumulators and each reduction is one v_mul_hi_u32 + a multiply-subtract + two v_min instead of
    // a 64-bit Barrett step (four v_mad_u64_u32 for the high product alone).  GF(251^3): 0.27 -> 0.5 of the roofline.
    template <int M>
    static GFA_HD u64 mul_m_small(const FieldDev &f, u64 a, u64 b)
    {
        const u32 p32 = (u32)f.p, mu32 = (u32)(f.mu >> 32);
        u32 av[M], bv[M];
        to_vec_m<M>(f, a, av);
        to_vec_m<M>(f, b, bv);
        u32 c[2 * M - 1];
#pragma unroll
        for (int k = 0; k < 2 * M - 1; k++) c[k] = 0;
#pragma unroll
        for (int i = 0; i < M; i++)
#pragma unroll
            for (int j = 0; j < M; j++) c[i + j] += av[i] * bv[j];
        u32 nir[M];
#pragma unroll
        for (int j = 0; j < M; j++) nir[j] = f.ext_irr[j] ? p32 - f.ext_irr[j] : 0u;
        if (f.r2) {
            // r05: no reduction before the end -- every partial sum of the product and of the folds stays below 2^32 for this field's
            // own polynomial (ext_lazy_ok replays the worst case when the field is created): M - 1 reductions less per product
#pragma unroll
            for (int k = 0; k + 1 < M; k++) {
                const u32 t = c[k];
#pragma unroll
                for (int j = 0; j < M; j++) c[k + 1 + j] += t * nir[j];
            }
        } else {
#pragma unroll
            for (int k = 0; k + 1 < M; k++) {
                const u32 t = red32(c[k], p32, mu32);
#pragma unroll
                for (int j = 0; j < M; j++) c[k + 1 + j] += t * nir[j];
            }
        }
        u32 out[M];
#pragma unroll
        for (int i = 0; i < M; i++) out[i] = red32(c[M - 1 + i], p32, mu32);
        return from_vec_m<M>(f, out);
    }
    // worst case of every partial sum of mul_m_small WITHOUT its intermediate reductions (host, at field creation)
    static bool ext_lazy_ok(u64 p, u32 m, const u32 *ext_irr)
    {
        if (p >= (1u << 13) || m < 2 || m > 8) return false;
        u64 B[15] = {};
        for (u32 k = 0; k + 1 < 2 * m; k++) B[k] = (u64)(k < m ? k + 1 : 2 * m - 1 - k) * (p - 1) * (p - 1);
        for (u32 k = 0; k + 1 < m; k++)
            for (u32 j = 0; j < m; j++) {
                B[k + 1 + j] += B[k] * (ext_irr[j] ? p - ext_irr[j] : 0);
                if (B[k + 1 + j] >> 32) return false;
            }
        return true;
    }
    template <int M>
    static GFA_HD u64 mul_m(const FieldDev &f, u64 a, u64 b)
    {
        return f.p < (1u << 13) ? mul_m_small<M>(f, a, b) : mul_m_wide<M>(f, a, b); // uniform over the launch
    }
    template <int M>
    static GFA_HD u64 mul_m_wide(const FieldDev &f, u64 a, u64 b)
    {
        u32 av[M], bv[M];
        to_vec_m<M>(f, a, av);
        to_vec_m<M>(f, b, bv);
        u64 c[2 * M - 1];
#pragma unroll
        for (int k = 0; k < 2 * M - 1; k++) c[k] = 0;
#pragma unroll
        for (int i = 0; i < M; i++)
#pragma unroll
            for (int j = 0; j < M; j++) c[i + j] += (u64)av[i] * bv[j]; // index k <-> degree 2M - 2 - k
        u32 nir[M]; // x^M == sum_j nir[j] x^(M-1-j)
#pragma unroll
        for (int j = 0; j < M; j++) nir[j] = f.ext_irr[j] ? (u32)f.p - f.ext_irr[j] : 0u;
#pragma unroll
        for (int k = 0; k + 1 < M; k++) {
            const u32 t = red64(f, c[k]);
#pragma unroll
            for (int j = 0; j < M; j++) c[k + 1 + j] += (u64)t * nir[j];
        }
        u32 out[M];
#pragma unroll
        for (int i = 0; i < M; i++) out[i] = red64(f, c[M - 1 + i]);
        return from_vec_m<M>(f, out);
    }
    // degrees the element-wise kernels are instantiated for (ExtM below)
    static GFA_HD bool fixed_degree(const FieldDev &f) { return f.m >= 2 && f.m <= 8 && f.p < (1ull << 31); }
    static GFA_HD u64 add(const FieldDev &f, u64 a, u64 b)
    {
        u32 av[GFA_MAX_EXT_DEGREE], bv[GFA_MAX_EXT_DEGREE];
        to_vec(f, a, av); to_vec(f, b, bv);
        for (u32 i = 0; i < f.m; i++) av[i] = Prime32::add(f, av[i], bv[i]);
        return from_vec(f, av);
    }
    static GFA_HD u64 sub(const FieldDev &f, u64 a, u64 b)
    {
        u32 av[GFA_MAX_EXT_DEGREE], bv[GFA_MAX_EXT_DEGREE];
        to_vec(f, a, av); to_vec(f, b, bv);
        for (u32 i = 0; i < f.m; i++) av[i] = Prime32::sub(f, av[i], bv[i]);
        return from_vec(f, av);
    }
    static GFA_HD u64 neg(const FieldDev &f, u64 a)
    {
        u32 av[GFA_MAX_EXT_DEGREE];
        to_vec(f, a, av);
        for (u32 i = 0; i < f.m; i++) av[i] = Prime32::neg(f, av[i]);
        return from_vec(f, av);
    }
    static GFA_HD u64 mul(const FieldDev &f, u64 a, u64 b)
    {
        u32 av[GFA_MAX_EXT_DEGREE], bv[GFA_MAX_EXT_DEGREE], cv[GFA_MAX_EXT_DEGREE];
        const u32 m = f.m;
        to_vec(f, a, av); to_vec(f, b, bv);
        for (u32 i = 0; i < m; i++) cv[i] = 0;
        // consume b from its lowest digit upward; keep a(x) * x^it reduced modulo the irreducible polynomial
        for (u32 it = 0; it < m; it++) {
            u32 bl = bv[m - 1 - it];
            if (bl)
                for (u32 i = 0; i < m; i++) cv[i] = Prime32::add(f, cv[i], Prime32::mul(f, bl, av[i]));
            u32 qd = av[0];
            for (u32 i = 0; i + 1 < m; i++) av[i] = av[i + 1];
            av[m - 1] = 0;
            if (qd)
                for (u32 i = 0; i < m; i++) av[i] = Prime32::sub(f, av[i], Prime32::mul(f, qd, f.ext_irr[i]));
        }
        return from_vec(f, cv);
    }
    static GFA_HD u64 one(const FieldDev &) { return 1; }
    static GFA_HD u64 pow_u(const FieldDev &f, u64 a, u64 e)
    {
        u64 r = 1;
        while (e) {
            if (e & 1) r = mul(f, r, a);
            a = mul(f, a, a);
            e >>= 1;
        }
        return r;
    }
    // a != 0.  Itoh-Tsujii as the reference (_calculate.py:469-489): a^-1 = (a^r)^-1 * a^(r-1), r = (q-1)/(p-1), a^r in GF(p)
    static GFA_HD u64 inv(const FieldDev &f, u64 a)
    {
        u64 r = (f.q - 1) / (f.p - 1);
        u64 a_r1 = pow_u(f, a, r - 1);
        u64 a_r = mul(f, a_r1, a);
        u32 norm_inv = Prime32::inv(f, (u32)a_r);
        return mul(f, (u64)norm_inv, a_r1);
    }
    static GFA_HD u64 from_int(const FieldDev &f, i64 k)
    {
        i64 r = k % (i64)f.p;
        if (r < 0) r += (i64)f.p;
        return (u64)r;
    }
};

// GF(p^M) with the degree fixed at compile time: the element-wise kernels are instantiated per degree (the host picks the
// instance), so every digit array stays in registers.  Same values as Ext for every operation.
template <int M>
struct ExtM {
    typedef u64 elem;
    static GFA_HD u64 add(const FieldDev &f, u64 a, u64 b) { return Ext::lin_m<M, 0>(f, a, b); }
    static GFA_HD u64 sub(const FieldDev &f, u64 a, u64 b) { return Ext::lin_m<M, 1>(f, a, b); }
    static GFA_HD u64 neg(const FieldDev &f, u64 a) { return Ext::lin_m<M, 2>(f, a, a); }
    static GFA_HD u64 mul(const FieldDev &f, u64 a, u64 b) { return Ext::mul_m<M>(f, a, b); }
    static GFA_HD u64 one(const FieldDev &) { return 1; }
    static GFA_HD u64 pow_u(const FieldDev &f, u64 a, u64 e)
    {
        u64 r = 1;
        while (e) {
            if (e & 1) r = mul(f, r, a);
            a = mul(f, a, a);
            e >>= 1;
        }
        return r;
    }
    static GFA_HD u64 inv(const FieldDev &f, u64 a)
    { // Itoh-Tsujii as Ext::inv
        const u64 r = (f.q - 1) / (f.p - 1);
        const u64 a_r1 = pow_u(f, a, r - 1);
        const u64 a_r = mul(f, a_r1, a);
        const u32 norm_inv = Prime32::inv(f, (u32)a_r);
        return mul(f, (u64)norm_inv, a_r1);
    }
    static GFA_HD u64 from_int(const FieldDev &f, i64 k) { return Ext::from_int(f, k); }
};

// a^e for signed e on the explicit-calculation kinds (power_square_and_multiply.calculate, _calculate.py:579-592).
// Returns false when the reference would raise ZeroDivisionError (0 ** negative).
template <class F>
GFA_HD bool pow_signed(const FieldDev &f, typename F::elem a, i64 e, typename F::elem *out)
{
    if (e == 0) { *out = F::one(f); return true; }
    if (a == 0) { *out = 0; return e > 0; }
    u64 ue;
    if (e < 0) {
        a = F::inv(f, a);
        ue = (u64)0 - (u64)e;
    } else {
        ue = (u64)e;
    }
    // a != 0: reduce the exponent modulo the multiplicative group order when it is known to fit
    if (f.q != 0) ue %= (f.q - 1);
    *out = F::pow_u(f, a, ue);
    return true;
}
template <>
GFA_HD bool pow_signed<Lut>(const FieldDev &f, u32 a, i64 e, u32 *out)
{
    if (e == 0) { *out = 1; return true; }
    if (a == 0) { *out = 0; return e > 0; }
    *out = Lut::pow_nz(f, a, e);
    return true;
}

} // namespace gfa
