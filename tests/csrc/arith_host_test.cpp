// Host build of galois_amd/csrc/gfa_arith.h: closed-form pieces that the GPU tests only see through whole kernels.
//   * Goldilocks::inv (addition chain for p - 2) against binary square-and-multiply and against a * a^-1 = 1;
//   * Prime32 Montgomery power against the Barrett power for odd p < 2^31;
//   * GF(2^m), 17 <= m <= 32: carry-less product out of integer multiplies + folds, against shift-and-xor;
//   * GF(2^m), 9 <= m <= 32: the nine-multiply 16-bit product (low and high halves) and the byte-indexed reduction tables of
//     bin16_holes_mul_kernel / bin32_tab_mul_kernel, against shift-and-xor.
#include "gfa_arith.h"
#include <cstdio>
#include <initializer_list>
using namespace gfa;

// Ext::mul_m_small (32-bit accumulators, p < 2^13) against Ext::mul_m_wide (64-bit) on random and extreme digit vectors
static long g_lazy_checked = 0;
template <int M>
static int check_ext_small(u32 p)
{
    int fails = 0;
    FieldDev f{};
    f.p = p; f.m = M; f.kind = KIND_EXT;
    f.q = 1;
    for (int i = 0; i < M; i++) {
        if (f.q > (((u64)1 << 63) / p)) return 0; // p^M beyond 63 bits: not a u64-element field
        f.q *= p;
    }
    f.mu = ~(u64)0 / p;
    if ((p & (p - 1)) == 0) f.mu += 1; // floor(2^64 / p) for a power of two (p = 2 never reaches these kernels; kept exact anyway)
    u64 x = 0x9E3779B97F4A7C15ull ^ p ^ (M * 1315423911u);
    for (int trial = 0; trial < 6; trial++) {
        for (int j = 0; j < M; j++) { x = x * 6364136223846793005ull + 1442695040888963407ull; f.ext_irr[j] = trial == 0 ? p - 1 : (u32)((x >> 33) % p); }
        for (int i = 0; i < 3000; i++) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            u64 a = (x >> 11) % f.q;
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            u64 b = (x >> 11) % f.q;
            if (i % 50 == 0) a = f.q - 1; // every digit p - 1
            if (i % 75 == 0) b = f.q - 1;
            f.r2 = 0;
            const u64 want = Ext::mul_m_wide<M>(f, a, b);
            if (Ext::mul_m_small<M>(f, a, b) != want) fails++;
            // r05: the same product without intermediate reductions wherever the replayed bound allows it for THIS polynomial
            if (Ext::ext_lazy_ok(p, M, f.ext_irr)) {
                f.r2 = 1;
                g_lazy_checked++;
                if (Ext::mul_m_small<M>(f, a, b) != want) fails++;
            }
        }
    }
    return fails;
}

int main()
{
    int fails = 0;
    for (u32 p : {3u, 5u, 7u, 251u, 257u, 4099u, 8191u})
        fails += check_ext_small<2>(p) + check_ext_small<3>(p) + check_ext_small<4>(p) + check_ext_small<5>(p) +
                 (p <= 251 ? check_ext_small<7>(p) + check_ext_small<8>(p) : 0);
    if (fails) printf("Ext::mul_m_small mismatches: %d\n", fails);
    if (g_lazy_checked < 100000) { printf("lazy-fold path hardly exercised: %ld\n", g_lazy_checked); fails++; }
    {
        FieldDev f{};
        f.p = Goldilocks::P;
        f.q = Goldilocks::P;
        u64 x = 0x123456789abcdef1ull;
        for (int i = 0; i < 20000; i++) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const u64 a = x % Goldilocks::P;
            if (!a) continue;
            const u64 v = Goldilocks::inv(f, a);
            if (Goldilocks::mul(f, a, v) != 1 || v != Goldilocks::pow_u(f, a, Goldilocks::P - 2)) fails++;
        }
        const u64 edge[] = {1, 2, Goldilocks::P - 1, Goldilocks::P - 2, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull};
        for (u64 a : edge)
            if (Goldilocks::mul(f, a, Goldilocks::inv(f, a)) != 1) fails++;
    }
    {
        const u32 primes[] = {3, 5, 251, 65537, 7340033, 2147483647u, 2147483629u, 1073741827u};
        for (u32 p : primes) {
            FieldDev f{};
            f.p = p;
            f.q = p;
            f.mu = ~(u64)0 / p; // floor((2^64 - 1) / p) = floor(2^64 / p) for odd p
            u64 x = p;
            for (int i = 0; i < 4000; i++) {
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                const u32 a = (u32)((x >> 20) % p);
                const u64 e = (x >> 7) % ((u64)p * 3);
                if (Prime32::pow_u(f, a, e) != Prime32::pow_barrett(f, a, e)) fails++;
                if (a && Prime32::mul(f, a, Prime32::inv(f, a)) != 1) fails++;
            }
        }
    }
    { // GF(2^m) products: integer-multiply carry-less product + folds through f, against the bit-serial definition
        struct { u64 irr; int m; } cases[] = {{0x100000000ull | 0x8299, 32}, {(1ull << 32) | 0x8D, 32}, {(1ull << 20) | 0x9, 20},
                                             {(1ull << 17) | 0x9, 17},      {(1ull << 24) | 0x1B, 24},   {(1ull << 31) | 0x9, 31},
                                             {(1ull << 32) | 0xC0000401ull, 32}, {(1ull << 8) | 0x1D, 8}, {(1ull << 2) | 3, 2},
                                             {(1ull << 21) | 5, 21}, {(1ull << 18) | 0x27, 18}, {(1ull << 20) | 0x6F3, 20}, {(1ull << 22) | 3, 22},
                                             {(1ull << 16) | 0x2D, 16}};
        for (auto c : cases) {
            FieldDev f{};
            f.p = 2; f.m = c.m; f.q = (u64)1 << c.m; f.kind = KIND_BIN; f.irr = c.irr;
            f.mu = Bin::fold_rounds(c.irr, c.m);
            const u64 mask = ((u64)1 << c.m) - 1, top = (u64)1 << (c.m - 1), red = c.irr ^ ((u64)1 << c.m);
            u64 x = 12345;
            for (int i = 0; i < 100000; i++) {
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                u64 a = (x >> 11) & mask;
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                u64 b = (x >> 13) & mask;
                if (i < 4) { a = mask; b = mask - i; }
                u64 want = 0, aa = a, bb = b;
                while (bb) { if (bb & 1) want ^= aa; bb >>= 1; const u64 carry = aa & top; aa = (aa ^ carry) << 1; if (carry) aa ^= red; }
                if (Bin::mul(f, a, b) != want) fails++;
                if (f.mu && Bin::mul_fold(f, (u32)a, (u32)b) != want) fails++;
            }
        }
    }
    { // r03: the products of bin16_holes_mul_kernel / bin32_tab_mul_kernel, exactly as the kernels compute them -- the nine-multiply
      // carry-less product of the low / high halves, then the part above x^m through byte-indexed reduction tables
        struct C { u64 irr; u32 m; };
        const C cases[] = {{(1ull << 9) | 0x11, 9}, {(1ull << 10) | 0x9, 10}, {(1ull << 11) | 0x5, 11}, {(1ull << 12) | 0x53, 12}, {(1ull << 13) | 0x1B, 13},
                           {(1ull << 14) | 0x2B, 14}, {(1ull << 15) | 0x3, 15}, {(1ull << 16) | 0x2D, 16}, {(1ull << 16) | 0x100B, 16},
                           {(1ull << 16) | 0xF00F, 16}, // dense polynomials: the tables do not care how many terms there are
                           {(1ull << 17) | 0x9, 17}, {(1ull << 20) | 0x6F3, 20}, {(1ull << 21) | 5, 21}, {(1ull << 22) | 3, 22}, {(1ull << 24) | 0x1B, 24},
                           {(1ull << 24) | 0xAB6D5F, 24}, {(1ull << 25) | 0x9, 25}, {(1ull << 31) | 0x9, 31}, {(1ull << 32) | 0xC0000401ull, 32},
                           {(1ull << 32) | 0x8299ull, 32}};
        for (auto c : cases) {
            const int m = (int)c.m;
            const u64 mask = ((u64)1 << m) - 1, top = (u64)1 << (m - 1), red = c.irr ^ ((u64)1 << m);
            u32 R[4][256];
            for (u32 h = 0; h < 256; h++) {
                u64 v = Bin::reduce_bits((u64)h << m, m, 8, c.irr);
                R[0][h] = (u32)v;
                for (int k = 1; k < 4; k++) { v = Bin::reduce_bits(v << 8, m, 8, c.irr); R[k][h] = (u32)v; }
                if (m <= 16 && R[1][h] != (u32)Bin::reduce_bits((u64)h << (m + 8), m, 16, c.irr)) fails++; // the 16-bit kernel's second table
            }
            u64 x = 777 + c.m;
            for (int i = 0; i < 200000; i++) {
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                u64 a = (x >> 11) & mask;
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                u64 b = (x >> 13) & mask;
                if (i < 4) { a = mask; b = mask - i; }
                u64 want = 0, aa = a, bb = b;
                while (bb) { if (bb & 1) want ^= aa; bb >>= 1; const u64 carry = aa & top; aa = (aa ^ carry) << 1; if (carry) aa ^= red; }
                if (m <= 16) {
                    const u32 other = (u32)(x >> 40) & 0xffffu; // the neighbouring element of the register must not leak in
                    const u32 xr = (u32)a | (other << 16), yr = (u32)b | ((other ^ 0x5a5au) << 16);
                    const u32 P = Bin::clmul16_lo(xr, yr);
                    if (((P & (u32)mask) ^ R[0][(P >> m) & 0xffu] ^ R[1][P >> (m + 8)]) != want) fails++;
                    const u32 Ph = Bin::clmul16_hi(other | ((u32)a << 16), (other ^ 0x1234u) | ((u32)b << 16));
                    if (Ph != P) fails++;
                } else {
                    const u64 P = m <= 21 ? Bin::clmul21((u32)a, (u32)b) : Bin::clmul32((u32)a, (u32)b);
                    const u32 H = (u32)(P >> m);
                    u32 r = (u32)(P & mask) ^ R[0][H & 0xffu] ^ R[1][(H >> 8) & 0xffu] ^ R[2][(H >> 16) & 0xffu];
                    if (m > 24) r ^= R[3][H >> 24];
                    if (r != want) fails++;
                }
            }
        }
        // the nine-multiply product against the bit-serial carry-less product on every pair of a small exhaustive range and on edge words
        for (u32 a = 0; a < 65536; a += 257)
            for (u32 b = 0; b < 65536; b += 263) {
                u32 want = 0;
                for (int i = 0; i < 16; i++)
                    if ((b >> i) & 1) want ^= a << i;
                if (Bin::clmul16_lo(a, b) != want || Bin::clmul16_hi(a << 16, b << 16) != want) fails++;
            }
        if (Bin::clmul16_lo(0xffffu, 0xffffu) != 0x55555555u || Bin::clmul16_hi(0xffff0000u, 0xffff0000u) != 0x55555555u) fails++;
    }
    printf("fails %d\n", fails);
    return fails != 0;
}
