// Host build of galois_amd/csrc/gfa_arith.h: closed-form pieces that the GPU tests only see through whole kernels.
//   * Goldilocks::inv (addition chain for p - 2) against binary square-and-multiply and against a * a^-1 = 1;
//   * Prime32 Montgomery power against the Barrett power for odd p < 2^31;
//   * GF(2^m), 17 <= m <= 32: carry-less product out of integer multiplies + folds, against shift-and-xor.
#include "gfa_arith.h"
#include <cstdio>
using namespace gfa;

int main()
{
    int fails = 0;
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
    printf("fails %d\n", fails);
    return fails != 0;
}
