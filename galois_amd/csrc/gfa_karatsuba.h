// gfa_karatsuba.h -- Karatsuba over the bit / digit positions of an extension-field element (r06): what turns a product of GF(2^m) or
// GF(p^m) elements into products of single bits / digit sums, shared by the matrix-core matmul (gfa_matmul_mfma.hip) and the NTT
// convolution (gfa_conv_crt.hip).  An element a = sum_i a_i x^i (a_i its bits or base-p digits); levels of
//     a b = P_lo (1 -+ x^h) + P_hi (x^2h -+ x^h) + P_mid x^h,   P_mid = (a_lo + a_hi)(b_lo + b_hi)
// over the positions (padded to a power of two; empty halves dropped) end in LEAVES t: a set of positions E_t -- the operand of the leaf is
// the sum of the digits in E_t (characteristic 2: the parity of the bits under a mask) -- and a weight polynomial r_t(x); the product is
// sum_t P_t r_t(x), and since reduction mod f is linear, sum_t P_t (r_t mod f).
#pragma once
#include <algorithm>
#include <vector>

#include "gfa_arith.h"

namespace gfa {

struct PlaneMasks {
    u32 m[243];
};
struct BinFold {
    u32 red[243]; // r_t(x) mod f
    int nt;
};
// the masks and weights of the Karatsuba leaves over `n` (a power of two) bit positions of which the first m are real
inline void karatsuba_leaves(const u32 *pos, int n, u64 w, u32 *masks, u64 *weights, int *count)
{
    if (n == 1) {
        if (pos[0]) { masks[*count] = pos[0]; weights[*count] = w; (*count)++; } // a zero mask is a zero operand: no product
        return;
    }
    const int h = n / 2;
    u32 mid[16];
    bool hi_any = false;
    for (int i = 0; i < h; i++) hi_any |= pos[h + i] != 0;
    if (!hi_any) { karatsuba_leaves(pos, h, w, masks, weights, count); return; } // both high halves zero (positions padded to a power of two)
    for (int i = 0; i < h; i++) mid[i] = pos[i] ^ pos[h + i];
    karatsuba_leaves(pos, h, w ^ (w << h), masks, weights, count);
    karatsuba_leaves(pos + h, h, (w << h) ^ (w << (2 * h)), masks, weights, count);
    karatsuba_leaves(mid, h, w << h, masks, weights, count);
}

struct DigitFold {
    uint8_t R[81][16]; // R[t][k]: coefficient of x^k of (weight polynomial of leaf t) mod f, in [0, p)
    uint16_t set[81];  // E_t as a mask over the digit positions
    int nt, m;
    u32 p;
};

template <typename T>
__device__ __forceinline__ void digits_of(T v, u32 p, int m, u32 (&d)[16])
{
    u64 x = (u64)v; // (static indices throughout: the digits stay in registers)
#pragma unroll
    for (int i = 0; i < 16; i++) {
        d[i] = 0;
        if (i < m) {
            if (sizeof(T) <= 4) { const u32 x32 = (u32)x, qd = x32 / p; d[i] = x32 - qd * p; x = qd; }
            else { const u64 qd = x / p; d[i] = (u32)(x - qd * p); x = qd; }
        }
    }
}
// Karatsuba leaves over n (a power of two) positions with integer weights mod p; pos[i] = set of digit positions summed at place i
inline void karatsuba_leaves_p(const u32 *pos, int n, const std::vector<i64> &w, u32 p, std::vector<u32> &sets, std::vector<std::vector<i64>> &weights)
{
    if (n == 1) {
        if (pos[0]) { sets.push_back(pos[0]); weights.push_back(w); }
        return;
    }
    const int h = n / 2;
    u32 mid[8];
    bool lo_any = false, hi_any = false;
    for (int i = 0; i < h; i++) { lo_any |= pos[i] != 0; hi_any |= pos[h + i] != 0; }
    auto shifted = [&](int by, i64 sign) { std::vector<i64> r(w.size() + by, 0); for (size_t i = 0; i < w.size(); i++) r[i + by] = sign * w[i]; return r; };
    auto add = [&](std::vector<i64> a, const std::vector<i64> &b) { if (a.size() < b.size()) a.resize(b.size(), 0); for (size_t i = 0; i < b.size(); i++) a[i] += b[i]; return a; };
    if (!hi_any) { karatsuba_leaves_p(pos, h, w, p, sets, weights); return; } // a(x) b(x) with both high halves zero: the low product alone
    // (positions hold SETS whose digits are summed: lo + hi at place i is the union -- the two are disjoint by construction)
    for (int i = 0; i < h; i++) mid[i] = pos[i] | pos[h + i];
    karatsuba_leaves_p(pos, h, add(w, shifted(h, -1)), p, sets, weights);                 // P_lo (1 - x^h)
    karatsuba_leaves_p(pos + h, h, add(shifted(2 * h, 1), shifted(h, -1)), p, sets, weights); // P_hi (x^2h - x^h)
    karatsuba_leaves_p(mid, h, shifted(h, 1), p, sets, weights);                          // P_mid x^h
    (void)lo_any;
}


// GF(2^m), 2 <= m <= 32: masks and x-weights reduced mod f
inline void make_bin_fold(const FieldDev &fd, PlaneMasks *pm, BinFold *bf)
{
    const int m = (int)fd.m;
    int n2 = 1;
    while (n2 < m) n2 *= 2;
    u32 pos[32], masks[243];
    u64 weights[243];
    for (int i = 0; i < n2; i++) pos[i] = i < m ? 1u << i : 0u;
    int nt = 0;
    karatsuba_leaves(pos, n2, 1, masks, weights, &nt);
    bf->nt = nt;
    for (int t = 0; t < nt; t++) pm->m[t] = masks[t];
    for (int t = 0; t < nt; t++) bf->red[t] = (u32)Bin::reduce_bits(weights[t], m, 2 * n2 - m, fd.irr); // deg r_t <= 2 n2 - 2
}

// GF(p^m), odd p, 2 <= m <= 16: sets and weights reduced mod p and mod f; false if there are more than 81 leaves
inline bool make_digit_fold(const FieldDev &fd, DigitFold *df)
{
    const int m = (int)fd.m;
    const u32 p = (u32)fd.p;
    int n2 = 1;
    while (n2 < m) n2 *= 2;
    u32 pos[16];
    for (int i = 0; i < n2; i++) pos[i] = i < m ? 1u << i : 0u;
    std::vector<u32> sets;
    std::vector<std::vector<i64>> weights;
    karatsuba_leaves_p(pos, n2, std::vector<i64>{1}, p, sets, weights);
    const int nt = (int)sets.size();
    if (nt > 81) return false;
    df->nt = nt; df->m = m; df->p = p;
    for (int t = 0; t < nt; t++) {
        df->set[t] = (uint16_t)sets[t];
        // weight polynomial mod p, then mod f: x^m = sum_k nir[k] x^k with nir[k] = -irr_k
        std::vector<i64> c(weights[t]);
        c.resize(std::max<size_t>(c.size(), (size_t)m), 0);
        for (auto &v : c) v = ((v % (i64)p) + (i64)p) % (i64)p;
        for (int sdeg = (int)c.size() - 1; sdeg >= m; sdeg--) {
            const i64 top = c[sdeg];
            if (!top) continue;
            for (int k = 0; k < m; k++) { // coefficient of x^k of f below the leading term: ext_irr[m - 1 - k]
                const i64 nir = fd.ext_irr[m - 1 - k] ? (i64)p - (i64)fd.ext_irr[m - 1 - k] : 0;
                c[sdeg - m + k] = (c[sdeg - m + k] + top * nir) % (i64)p;
            }
            c[sdeg] = 0;
        }
        for (int k = 0; k < m; k++) df->R[t][k] = (uint8_t)c[k];
    }
    return true;
}

} // namespace gfa
