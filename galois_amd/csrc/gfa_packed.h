// gfa_packed.h -- sums, differences and negatives in GF(p^m), p odd, as PACKED-DIGIT arithmetic (r05).
//
// The reference adds two elements of GF(p^m) digit by digit (add_vector / subtract_vector / negative_vector,
// src/galois/_domains/_calculate.py:150-181, 254-285, 202-232) or, in lookup mode, through the Zech logarithm
// (_lookup.py:31-60, 89-150).  For fields whose tables leave LDS (8192 < q <= 2^20 in odd characteristic) the table route is a
// chain of gathers from L2 (GF(7^7): 0.04 of the roofline) or a three-table staged kernel (GF(3^10): 0.26).  Sums do not need
// the multiplicative structure at all: with the m base-p digits of an element packed into one 32-bit word, W bits per digit,
// an addition is ONE integer add plus a per-field conditional subtraction done for all digits at once:
//     s = A + B                      every field <= 2p - 2 < 2^W
//     g = (s + BIAS) & GUARD         BIAS adds 2^(W-1) - p to every field: its top bit comes out set exactly where s_i >= p
//     r = s - (g >> (W-1)) * p       subtract p in the flagged fields (one multiply: fields cannot carry into each other)
// Integer <-> packed form goes through small tables that fit LDS: x = hi * P + lo with P = p^hl <= 4096 (an EXACT quotient by one
// v_mul_hi with ceil(2^32 / P): x < 2^20), PK_LO[lo] | PK_HI[hi] gives the packed word; the way back sums nch table entries
// indexed by chunks of k fields (k W <= 11 bits).  Six LDS reads + ~25 vector instructions per element, no division, no gather
// from L2.  Value-identical to the reference: the result digits are (a_i +- b_i) mod p by construction.
//
// Everything here is plain integer code shared by the kernel (gfa_elementwise_packed.hip) and the host check
// (tests/csrc/packed_host_test.cpp: every element pair of small fields, random pairs of every field shape up to 2^20).
#pragma once
#include <cstdint>
#include <vector>

#if defined(__HIPCC__)
#define GFP_HD __host__ __device__ __forceinline__
#else
#define GFP_HD inline
#endif

namespace gfa_packed {

typedef uint32_t pu32;

struct Plan {
    pu32 p, m, q;
    pu32 W;        // bits per digit field: 2^(W-1) >= p
    pu32 hl, hh;   // digits in the low / high part of the integer
    pu32 P;        // p^hl (<= 4096)
    pu32 magic;    // ceil(2^32 / P): floor(x / P) == mulhi(x, magic) for x < 2^20
    pu32 QH;       // p^hh: entries of the high table
    pu32 bias, guard, pfull; // per-field constants replicated over the m fields
    pu32 k, nch, chunk_bits; // unpacking: nch chunks of k fields (chunk_bits = k * W <= 11)
    pu32 off_hi, off_un, words; // table offsets (32-bit words): PK_LO at 0, PK_HI at off_hi, chunk c of the unpack tables at off_un + (c << chunk_bits)
};

inline pu32 ipow(pu32 b, pu32 e) { pu32 r = 1; while (e--) r *= b; return r; }

// false: the field does not fit the scheme (even characteristic, prime field, more than 32 packed bits, q > 2^20)
inline bool make_plan(uint64_t p64, uint32_t m, Plan *pl)
{
    if (p64 < 3 || (p64 & 1) == 0 || m < 2 || p64 > 1021) return false;
    uint64_t q = 1;
    for (uint32_t i = 0; i < m; i++) { q *= p64; if (q > (1u << 20)) return false; }
    const pu32 p = (pu32)p64;
    pu32 W = 1;
    while ((1u << (W - 1)) < p) W++;
    if (m * W > 32) return false;
    pu32 hl = (m + 1) / 2;
    while (hl > 1 && ipow(p, hl) > 4096) hl--;
    if (ipow(p, hl) > 4096) return false;
    const pu32 hh = m - hl;
    if (ipow(p, hh) > 16384) return false;
    pu32 k = 11 / W;
    if (k < 1) return false;
    if (k > m) k = m;
    Plan r{};
    r.p = p; r.m = m; r.q = (pu32)q; r.W = W; r.hl = hl; r.hh = hh; r.P = ipow(p, hl); r.QH = ipow(p, hh);
    r.magic = (pu32)((((uint64_t)1 << 32) + r.P - 1) / r.P);
    for (pu32 i = 0; i < m; i++) {
        r.bias |= ((1u << (W - 1)) - p) << (W * i);
        r.guard |= (1u << (W - 1)) << (W * i);
        r.pfull |= p << (W * i);
    }
    r.k = k; r.nch = (m + k - 1) / k; r.chunk_bits = k * W;
    if (r.nch > 4) return false; // (cannot happen for m W <= 32; from_packed is written for at most four chunks)
    r.off_hi = r.P; r.off_un = r.P + r.QH; r.words = r.off_un + (r.nch << r.chunk_bits);
    *pl = r;
    return true;
}

// digit i (i = 0: least significant) of the integer lives in field i
inline void build_tables(const Plan &pl, std::vector<pu32> &t)
{
    t.assign(pl.words, 0);
    auto pack = [&](pu32 v, pu32 ndig, pu32 first_field) {
        pu32 w = 0;
        for (pu32 i = 0; i < ndig; i++) { w |= (v % pl.p) << (pl.W * (first_field + i)); v /= pl.p; }
        return w;
    };
    for (pu32 v = 0; v < pl.P; v++) t[v] = pack(v, pl.hl, 0);
    for (pu32 v = 0; v < pl.QH; v++) t[pl.off_hi + v] = pack(v, pl.hh, pl.hl);
    for (pu32 c = 0; c < pl.nch; c++) {
        const pu32 nf = (c + 1) * pl.k <= pl.m ? pl.k : pl.m - c * pl.k;
        for (pu32 idx = 0; idx < (1u << pl.chunk_bits); idx++) {
            pu32 val = 0, scale = ipow(pl.p, c * pl.k);
            bool ok = true;
            for (pu32 i = 0; i < nf; i++) {
                const pu32 d = (idx >> (pl.W * i)) & ((1u << pl.W) - 1);
                if (d >= pl.p) ok = false;
                val += d * scale;
                scale *= pl.p;
            }
            if (idx >> (pl.W * nf)) ok = false;
            t[pl.off_un + (c << pl.chunk_bits) + idx] = ok ? val : 0;
        }
    }
}

GFP_HD pu32 mulhi32(pu32 a, pu32 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (pu32)(((uint64_t)a * b) >> 32);
#endif
}

// integer -> packed digits (tab: the table image, LDS on the device)
GFP_HD pu32 to_packed(const Plan &pl, const pu32 *tab, pu32 x)
{
    const pu32 hi = mulhi32(x, pl.magic);
    const pu32 lo = x - hi * pl.P;
    return tab[lo] | tab[pl.off_hi + hi];
}
// one conditional subtraction of p in every field (fields <= 2p - 1 in, < p out)
GFP_HD pu32 reduce_fields(const Plan &pl, pu32 s)
{
    const pu32 g = (s + pl.bias) & pl.guard;
    return s - (g >> (pl.W - 1)) * pl.p;
}
template <int OP> // 0 add, 1 sub, 2 neg (b unused)
GFP_HD pu32 lin_packed(const Plan &pl, pu32 a, pu32 b)
{
    return reduce_fields(pl, OP == 0 ? a + b : OP == 1 ? a + pl.pfull - b : pl.pfull - a);
}
GFP_HD pu32 from_packed(const Plan &pl, const pu32 *tab, pu32 w)
{
    // nch <= 4 (m W <= 32 bits in chunks of k W, k = floor(11 / W) fields: ceil(m / k) <= 4 for every accepted field); written out
    // with uniform branches: a run-time loop costs ~30 scalar instructions per element on the device (profiles/r05_pmc_packed.txt)
    const pu32 cb = pl.chunk_bits, mask = (1u << cb) - 1u;
    const pu32 *t = tab + pl.off_un;
    pu32 v = t[w & mask];
    if (pl.nch > 1) v += t[(1u << cb) + ((w >> cb) & mask)];
    if (pl.nch > 2) v += t[(2u << cb) + ((w >> (2 * cb)) & mask)];
    if (pl.nch > 3) v += t[(3u << cb) + ((w >> (3 * cb)) & mask)];
    return v;
}

// ---- two packed words (r06): fields whose m digits need more than 32 packed bits -- GF(3^11) (33 bits) and GF(3^12) (36), the only ones
// with q <= 2^20.  Sums never mix digits, so the element splits into x = xh * p^ml + xl and each part is a digit vector of its own
// Plan (ml = ceil(m / 2) and m - ml digits): two independent packed words, the same six instructions on each.  The quotient by
// PL = p^ml <= 4096 is exact through one v_mul_hi (x < 2^20), as inside to_packed.
struct Plan2 {
    Plan lo, hi;
    pu32 PL, magicL; // p^ml and ceil(2^32 / PL)
    pu32 off2;       // 32-bit words of the low plan's tables: the high plan's tables follow
    pu32 words;
};
inline bool make_plan2(uint64_t p64, uint32_t m, Plan2 *out)
{
    Plan single;
    if (make_plan(p64, m, &single)) return false; // one word is enough
    if (p64 < 3 || (p64 & 1) == 0 || m < 4 || p64 > 1021) return false;
    uint64_t q = 1;
    for (uint32_t i = 0; i < m; i++) { q *= p64; if (q > (1u << 20)) return false; }
    const uint32_t ml = (m + 1) / 2;
    Plan2 r{};
    if (!make_plan(p64, ml, &r.lo) || !make_plan(p64, m - ml, &r.hi)) return false;
    r.PL = ipow((pu32)p64, ml);
    if (r.PL > 4096) return false;
    r.magicL = (pu32)((((uint64_t)1 << 32) + r.PL - 1) / r.PL);
    r.off2 = r.lo.words;
    r.words = r.lo.words + r.hi.words;
    *out = r;
    return true;
}
inline void build_tables2(const Plan2 &pl, std::vector<pu32> &t)
{
    std::vector<pu32> a, b;
    build_tables(pl.lo, a);
    build_tables(pl.hi, b);
    t = a;
    t.insert(t.end(), b.begin(), b.end());
}
struct Pk2 { pu32 lo, hi; };
GFP_HD Pk2 to_packed2(const Plan2 &pl, const pu32 *tab, pu32 x)
{
    const pu32 xh = mulhi32(x, pl.magicL);
    const pu32 xl = x - xh * pl.PL;
    return Pk2{to_packed(pl.lo, tab, xl), to_packed(pl.hi, tab + pl.off2, xh)};
}
template <int OP>
GFP_HD Pk2 lin_packed2(const Plan2 &pl, Pk2 a, Pk2 b)
{
    return Pk2{lin_packed<OP>(pl.lo, a.lo, b.lo), lin_packed<OP>(pl.hi, a.hi, b.hi)};
}
GFP_HD pu32 from_packed2(const Plan2 &pl, const pu32 *tab, Pk2 w)
{
    return from_packed(pl.lo, tab, w.lo) + pl.PL * from_packed(pl.hi, tab + pl.off2, w.hi);
}

// ---- products on the same digit tables (r05): schoolbook product of the two digit vectors with NO reduction until the end.
// multiply_vector (_calculate.py:343-383) reduces modulo p after every step; for these small characteristics every partial sum of
// the product AND of the folds through x^m = -(irr) stays below 2^32 (bound_ok replays the worst case), so the only reductions
// are the m final ones.  nir[j]: coefficient of x^j of  x^m mod irr, i.e. p - irr_j.
struct MulAux {
    pu32 nir[8];  // degree 0 .. m-1
    pu32 mu32;    // floor(2^32 / p)
};

inline bool mul_bound_ok(const Plan &pl, const MulAux &ax)
{
    const pu32 m = pl.m;
    if (m > 8) return false;
    uint64_t B[15] = {};
    for (pu32 k = 0; k + 1 < 2 * m; k++) B[k] = (uint64_t)(k < m ? k + 1 : 2 * m - 1 - k) * (pl.p - 1) * (pl.p - 1);
    for (pu32 k = 2 * m - 2; k >= m; k--)
        for (pu32 j = 0; j < m; j++) {
            B[k - m + j] += B[k] * ax.nir[j];
            if (B[k - m + j] >> 32) return false;
        }
    return true;
}

GFP_HD pu32 red32(pu32 x, pu32 p, pu32 mu32)
{ // x mod p, any 32-bit x (the estimate is short by at most 2)
    const pu32 qd = mulhi32(x, mu32);
    pu32 r = x - qd * p;
    pu32 d = r - p;
    r = d < r ? d : r;
    d = r - p;
    return d < r ? d : r;
}

GFP_HD pu32 mul24(pu32 a, pu32 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (a & 0xffffffu) * (b & 0xffffffu);
#endif
}

template <int M>
GFP_HD pu32 mul_digits(const Plan &pl, const MulAux &ax, pu32 pa, pu32 pb)
{ // pa / pb: packed digits (field i = coefficient of x^i); returns the product as an integer
    const pu32 fm = (1u << pl.W) - 1u;
    pu32 da[M], db[M], c[2 * M - 1];
#pragma unroll
    for (int i = 0; i < M; i++) { da[i] = (pa >> (pl.W * i)) & fm; db[i] = (pb >> (pl.W * i)) & fm; }
#pragma unroll
    for (int k = 0; k < 2 * M - 1; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < M; i++)
#pragma unroll
        for (int j = 0; j < M; j++) c[i + j] += mul24(da[i], db[j]); // digits are below 2^11: one v_mad_u32_u24 per term (58 lane-ops/clk/CU, as every multiply: profiles/r03_valu_issue_rates.txt)
#pragma unroll
    for (int k = 2 * M - 2; k >= M; k--)
#pragma unroll
        for (int j = 0; j < M; j++) c[k - M + j] += c[k] * ax.nir[j];
    pu32 v = red32(c[M - 1], pl.p, ax.mu32);
#pragma unroll
    for (int i = M - 2; i >= 0; i--) v = v * pl.p + red32(c[i], pl.p, ax.mu32);
    return v;
}

// floor(x / d) for EVERY 32-bit x (Granlund-Montgomery round-up form): l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1,
// q = (t + ((x - t) >> 1)) >> (l - 1) with t = mulhi(m, x).  d odd, 3 <= d < 2^31.  Five instructions; the one-mulhi form (Plan::magic,
// Div2Aux::magic) is exact only for x < 2^20.
struct ExactDiv {
    pu32 m, sh;
};
inline ExactDiv make_exact_div(pu32 d)
{
    pu32 l = 0;
    while (((uint64_t)1 << l) < d) l++;
    ExactDiv e;
    e.m = (pu32)(((((uint64_t)1 << l) - d) << 32) / d + 1);
    e.sh = l - 1;
    return e;
}
GFP_HD pu32 exact_div(const ExactDiv &e, pu32 x)
{
    const pu32 t = mulhi32(x, e.m);
    return (t + ((x - t) >> 1)) >> e.sh;
}

// ---- quotients and reciprocals of GF(p^2), 32768 < q <= 2^20 (r06): by the norm.  With X^2 = s X + t (s, t = nir[1], nir[0]) the
// conjugate of b = b0 + b1 X is b^p = (b0 + s b1) - b1 X and N(b) = b b^p = b0^2 + s b0 b1 - t b1^2 lies in GF(p), non-zero for
// b != 0 (the polynomial is irreducible); 1 / b = b^p / N with 1 / N from a p-entry table (LDS).  Replaces, for these fields, the
// reference's divide / reciprocal through LOG / EXP tables that do not fit LDS here (_lookup.py:176-235) or its digit-vector
// loop with a Fermat power (_calculate.py:447-513) -- same field, same values.  Every partial sum stays below 2^32 (p <= 1021).
struct Div2Aux {
    pu32 p, s, t, mu32; // X^2 = s X + t; mu32 = floor(2^32 / p)
    pu32 magic;         // ceil(2^32 / p): floor(x / p) == mulhi(x, magic) for x < 2^20
    ExactDiv xd;        // WIDE (p > 1021, elements up to 2^32): the digit split
};
// WIDE = false: 182 <= p <= 1021 (32768 < p^2 <= 2^20).  WIDE = true (r06): 1021 < p <= 37813, elements below 2^32 on uint32 arrays -- fields
// that have no tables at all (q > 2^20); the same formulas (every partial sum below 3 p^2 < 2^32), an exact digit split, and the inverse
// table as 16-bit entries (at most 74 KiB of LDS).
constexpr pu32 DIV2_WIDE_MAX_P = 37813; // the largest prime with 3 p^2 < 2^32
inline bool make_div2(uint64_t p, uint32_t m, const pu32 *nir, Div2Aux *ax, bool wide = false)
{
    if (m != 2 || (p & 1) == 0) return false;
    if (wide ? (p <= 1021 || p > DIV2_WIDE_MAX_P) : (p < 182 || p > 1021)) return false;
    ax->p = (pu32)p; ax->s = nir[1]; ax->t = nir[0];
    ax->mu32 = (pu32)(((uint64_t)1 << 32) / p);
    ax->magic = (pu32)((((uint64_t)1 << 32) + p - 1) / p);
    ax->xd = make_exact_div((pu32)p);
    return true;
}
inline void build_inverse_table(pu32 p, std::vector<pu32> &t)
{ // t[v] = v^-1 mod p, t[0] = 0
    t.assign(p, 0);
    for (pu32 v = 1; v < p; v++) {
        uint64_t r = 1, b = v;
        for (pu32 e = p - 2; e; e >>= 1) { if (e & 1) r = r * b % p; b = b * b % p; }
        t[v] = (pu32)r;
    }
}
// a / b (RECIP: 1 / b) as integers d1 * p + d0; *zero is set when b == 0 (the result is then 0)
template <bool RECIP, bool WIDE = false, typename IT = pu32>
GFP_HD pu32 div2(const Div2Aux &ax, const IT *inv, pu32 a, pu32 b, bool *zero)
{
    const pu32 p = ax.p;
    const pu32 b1 = WIDE ? exact_div(ax.xd, b) : mulhi32(b, ax.magic), b0 = b - b1 * p;
    const pu32 u = red32(b0 + ax.s * b1, p, ax.mu32);                                         // conjugate: u - b1 X
    const pu32 nrm = red32(b0 * b0 + ax.s * red32(b0 * b1, p, ax.mu32) + (p - ax.t) * red32(b1 * b1, p, ax.mu32), p, ax.mu32);
    *zero = b == 0;
    const pu32 ni = (pu32)inv[nrm];
    const pu32 i0 = red32(u * ni, p, ax.mu32), i1 = red32((p - b1) * ni, p, ax.mu32);     // 1 / b = i0 + i1 X  ((p - 0) * ni = 0 mod p)
    if (RECIP) return i1 * p + i0;
    const pu32 a1 = WIDE ? exact_div(ax.xd, a) : mulhi32(a, ax.magic), a0 = a - a1 * p;
    const pu32 w = red32(a1 * i1, p, ax.mu32);                                                 // a1 i1 X^2 = w (s X + t)
    const pu32 q0 = red32(a0 * i0 + ax.t * w, p, ax.mu32);
    const pu32 q1 = red32(a0 * i1 + a1 * i0 + ax.s * w, p, ax.mu32);
    return q1 * p + q0;
}

// ---- quotients and reciprocals of GF(p^3), 65536 < q <= 2^20 (41 <= p <= 101; r06): by Cramer's rule on the multiplication matrix.  With
// X^3 = n2 X^2 + n1 X + n0, b * c = M c for M = [b | X b | X^2 b] (columns in the basis 1, X, X^2), so 1 / b = M^-1 e0 = (cofactors of M's first
// row) / det M, det M = N(b) in GF(p), non-zero for b != 0.  Six products for the two shifted columns, six for the cofactors, three for the
// determinant, one table inverse, three to scale; every partial sum below 2^23.  Same field, same values as the reference's table division.
struct Div3Aux {
    pu32 p, n0, n1, n2, mu32, magic; // magic = ceil(2^32 / p): exact quotients for x < 2^20
    ExactDiv xd;                      // WIDE (p > 101, elements up to 2^32): the digit split
};
// WIDE = false: 41 <= p <= 101 (65536 < p^3 <= 2^20).  WIDE = true (r06): 101 < p <= 1621 (p^3 < 2^32), uint32 arrays of fields without
// tables: the cofactors are reduced BEFORE the determinant (its three products then stay below 3 p^2), the split is exact for 32-bit elements.
constexpr pu32 DIV3_WIDE_MAX_P = 1621; // the largest prime with p^3 < 2^32
inline bool make_div3(uint64_t p, uint32_t m, const pu32 *nir, Div3Aux *ax, bool wide = false)
{
    if (m != 3 || (p & 1) == 0) return false;
    if (wide ? (p <= 101 || p > DIV3_WIDE_MAX_P) : (p < 41 || p > 101)) return false;
    ax->p = (pu32)p; ax->n0 = nir[0]; ax->n1 = nir[1]; ax->n2 = nir[2];
    ax->mu32 = (pu32)(((uint64_t)1 << 32) / p);
    ax->magic = (pu32)((((uint64_t)1 << 32) + p - 1) / p);
    ax->xd = make_exact_div((pu32)p);
    return true;
}
template <bool RECIP, bool WIDE = false>
GFP_HD pu32 div3(const Div3Aux &ax, const pu32 *inv, pu32 a, pu32 b, bool *zero)
{
    const pu32 p = ax.p, pp = p * p;
    auto split = [&](pu32 x, pu32 &d0, pu32 &d1, pu32 &d2) { // x = d2 p^2 + d1 p + d0
        const pu32 q1 = WIDE ? exact_div(ax.xd, x) : mulhi32(x, ax.magic);
        d0 = x - q1 * p;
        d2 = WIDE ? exact_div(ax.xd, q1) : mulhi32(q1, ax.magic);
        d1 = q1 - d2 * p;
    };
    auto R = [&](pu32 x) { return red32(x, p, ax.mu32); };
    pu32 b0, b1, b2;
    split(b, b0, b1, b2);
    *zero = b == 0;
    // columns X b and X^2 b
    const pu32 d0 = R(b2 * ax.n0), d1 = R(b0 + b2 * ax.n1), d2 = R(b1 + b2 * ax.n2);
    const pu32 e0 = R(d2 * ax.n0), e1 = R(d0 + d2 * ax.n1), e2 = R(d1 + d2 * ax.n2);
    // M = [[b0 d0 e0] [b1 d1 e1] [b2 d2 e2]]; cofactors of the first row (each in (0, 2 p^2))
    const pu32 c0 = pp + d1 * e2 - e1 * d2;
    const pu32 c1 = pp + e1 * b2 - b1 * e2;
    const pu32 c2 = pp + b1 * d2 - d1 * b2;
    const pu32 r0 = R(c0), r1 = R(c1), r2 = R(c2);
    const pu32 det = WIDE ? R(b0 * r0 + d0 * r1 + e0 * r2) : R(b0 * c0 + d0 * c1 + e0 * c2);
    const pu32 ni = inv[det];
    const pu32 i0 = R(r0 * ni), i1 = R(r1 * ni), i2 = R(r2 * ni); // 1 / b = i0 + i1 X + i2 X^2
    if (RECIP) return (i2 * p + i1) * p + i0;
    pu32 a0, a1, a2;
    split(a, a0, a1, a2);
    // (a0 + a1 X + a2 X^2)(i0 + i1 X + i2 X^2), folded through X^3 = n2 X^2 + n1 X + n0 and X^4 = X * X^3
    const pu32 t0 = a0 * i0, t1 = a0 * i1 + a1 * i0, t2 = a0 * i2 + a1 * i1 + a2 * i0;
    const pu32 t3 = R(a1 * i2 + a2 * i1), t4 = R(a2 * i2);
    // X^4 = n2 X^3 + n1 X^2 + n0 X: fold t4 first, then the X^3 term
    const pu32 u3 = R(t3 + t4 * ax.n2);
    const pu32 q0 = R(t0 + u3 * ax.n0);
    const pu32 q1 = R(t1 + t4 * ax.n0 + u3 * ax.n1);
    const pu32 q2 = R(t2 + t4 * ax.n1 + u3 * ax.n2);
    return (q2 * p + q1) * p + q0;
}

} // namespace gfa_packed
