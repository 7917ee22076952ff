// Host check of galois_amd/csrc/gfa_goldilocks.h (the portable expressions of the lazy 96-bit Goldilocks arithmetic) against
// 128-bit integer arithmetic: random and edge-case operands; built and run by tests/test_host_logic.py.
#include "gfa_goldilocks.h"
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace gfa::gl;
typedef unsigned __int128 u128;
typedef __int128 i128;
static i128 val(G3 x) { return (i128)x.lo + ((i128)x.mid << 32) + ((i128)x.hi * ((i128)1 << 64)); }
static gu64 modp(i128 v) { i128 r = v % (i128)P; if (r < 0) r += P; return (gu64)r; }
template <int S>
static long check_pow2(std::mt19937_64 &rng, const gu64 *edge)
{
    long fails = 0;
    for (int it = 0; it < 20000; it++) {
        G3 x = it % 3 == 0 ? G3{(gu32)rng(), (gu32)rng(), (int32_t)(rng() % 257) - 128}
                           : (it % 3 == 1 ? from_u64(edge[rng() % 11]) : G3{(gu32)rng(), (gu32)rng(), (int32_t)(rng() % 5) - 2});
        if (it == 7) x = G3{0xFFFFFFFFu, 0xFFFFFFFFu, 127};
        if (it == 8) x = G3{0u, 0u, -128};
        const G3 y = mul_pow2<S>(x);
        const i128 want = (i128)(((u128)modp(val(x)) * (u128)((((u128)1) << S) % P)) % P);
        if (modp(val(y)) != (gu64)want) { fails++; if (fails < 3) printf("mul_pow2<%d> mismatch\n", S); }
        if (y.hi < -8 || y.hi > 8) { fails++; if (fails < 3) printf("mul_pow2<%d> hi range %d\n", S, y.hi); }
    }
    return fails;
}

int main() {
    std::mt19937_64 rng(1);
    const gu64 edge[] = {0, 1, P - 1, P, P + 1, ~0ull, 0xFFFFFFFFull, 0xFFFFFFFF00000000ull, 1ull << 32, (1ull << 32) - 1, 0x8000000000000000ull};
    auto pick = [&]() -> gu64 { unsigned k = rng() % 4; return k == 0 ? edge[rng() % 11] : rng(); };
    long fails = 0;
    for (long it = 0; it < 1000000; it++) {
        G3 a = from_u64(pick()), b = from_u64(pick());
        // random lazy values: a few adds/subs deep
        G3 x = a; i128 vx = val(a);
        int depth = rng() % 6;
        for (int d = 0; d < depth; d++) { G3 c = from_u64(pick()); if (rng() & 1) { x = add(x, c); vx += val(c); } else { x = sub(x, c); vx -= val(c); } }
        // extreme: +-(2^69)
        if (it % 7 == 0) { x = G3{(gu32)rng(), (gu32)rng(), (int32_t)(rng() % 127) - 63}; vx = val(x); }
        if (val(x) != vx) { fails++; if (fails < 5) printf("add/sub mismatch\n"); }
        gu64 y = to_u64(x);
        if (modp((i128)y) != modp(vx)) { fails++; if (fails < 5) printf("to_u64 mismatch hi=%d\n", x.hi); }
        gu64 cn = canon(x);
        if (cn >= P || cn != modp(vx)) { fails++; if (fails < 5) printf("canon mismatch\n"); }
        gu64 w = pick();
        G3 m = mul(x, w);
        if (m.hi < -1 || m.hi > 1) { fails++; if (fails < 5) printf("mul hi range %d\n", m.hi); }
        u128 pr = (u128)modp(vx) * (u128)(w % P) % P;
        if (modp(val(m)) != (gu64)pr) { fails++; if (fails < 5) printf("mul mismatch\n"); }
        G3 m2 = mul_u64(y, w);
        if (modp(val(m2)) != (gu64)pr) { fails++; if (fails < 5) printf("mul_u64 mismatch\n"); }
    }
    // shift twiddles: every multiple of 6 below 96 (the twiddles of radix-4 ... radix-32 networks), and the word-aligned cases
    fails += check_pow2<6>(rng, edge) + check_pow2<12>(rng, edge) + check_pow2<18>(rng, edge) + check_pow2<24>(rng, edge) +
             check_pow2<30>(rng, edge) + check_pow2<36>(rng, edge) + check_pow2<42>(rng, edge) + check_pow2<48>(rng, edge) +
             check_pow2<54>(rng, edge) + check_pow2<60>(rng, edge) + check_pow2<66>(rng, edge) + check_pow2<72>(rng, edge) +
             check_pow2<78>(rng, edge) + check_pow2<84>(rng, edge) + check_pow2<90>(rng, edge) + check_pow2<32>(rng, edge) +
             check_pow2<64>(rng, edge) + check_pow2<1>(rng, edge) + check_pow2<95>(rng, edge);
    printf("fails %ld\n", fails);
    return fails != 0;
}
