// Host build of galois_amd/csrc/gfa_arith.h: closed-form pieces that the GPU tests only see through whole kernels.
//   * Goldilocks::inv (addition chain for p - 2) against binary square-and-multiply and against a * a^-1 = 1;
//   * Prime32 Montgomery power against the Barrett power for odd p < 2^31.
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
    printf("fails %d\n", fails);
    return fails != 0;
}
