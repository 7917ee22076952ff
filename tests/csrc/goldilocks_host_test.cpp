// Host check of galois_amd/csrc/gfa_goldilocks.h (the portable expressions of the lazy 96-bit Goldilocks arithmetic) against
// 128-bit integer arithmetic: random and edge-case operands; built and run by tests/test_host_logic.py.
#include "gfa_goldilocks.h"
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace gfa::gl;
typedef unsigned __int128 u128;
typedef __int128 i128;
static i128 val(G3 x) { return (i128)limb0(x) + ((i128)limb1(x) << 32) + ((i128)limb2(x) * ((i128)1 << 64)); }
static gu64 modp(i128 v) { i128 r = v % (i128)P; if (r < 0) r += P; return (gu64)r; }
template <int S>
static long check_pow2(std::mt19937_64 &rng, const gu64 *edge)
{
    long fails = 0;
    for (int it = 0; it < 20000; it++) {
        G3 x = it % 3 == 0 ? from_limbs((gu32)rng(), (gu32)rng(), (int32_t)(rng() % 257) - 128)
                           : (it % 3 == 1 ? from_u64(edge[rng() % 11]) : from_limbs((gu32)rng(), (gu32)rng(), (int32_t)(rng() % 5) - 2));
        if (it == 7) x = from_limbs(0xFFFFFFFFu, 0xFFFFFFFFu, 127);
        if (it == 8) x = from_limbs(0u, 0u, -128);
        const G3 y = mul_pow2<S>(x);
        const i128 want = (i128)(((u128)modp(val(x)) * (u128)((((u128)1) << S) % P)) % P);
        if (modp(val(y)) != (gu64)want) { fails++; if (fails < 3) printf("mul_pow2<%d> mismatch\n", S); }
        if (limb2(y) < -8 || limb2(y) > 8) { fails++; if (fails < 3) printf("mul_pow2<%d> hi range %d\n", S, limb2(y)); }
    }
    return fails;
}

// mul_pow2_small<S>: operands up to its precondition |x| < 2^(95 - S % 32), against the same 128-bit product
template <int S>
static long check_pow2_small(std::mt19937_64 &rng)
{
    long fails = 0;
    constexpr int r = S % 32;
    const int hbits = 31 - r; // |hi| < 2^hbits
    for (int it = 0; it < 20000; it++) {
        int64_t span = (int64_t)1 << hbits;
        int32_t hi = (int32_t)((int64_t)(rng() % (2 * span)) - span);
        if (it % 5 == 0) hi = (int32_t)(span - 1);
        if (it % 5 == 1) hi = (int32_t)(-span);
        if (it % 5 == 2) hi = (int32_t)(rng() % 3) - 1;
        if (hi > span - 1) hi = (int32_t)(span - 1);
        if (hi < -span) hi = (int32_t)(-span);
        gu32 lo = (gu32)rng(), mid = (gu32)rng();
        if (it % 11 == 0) { lo = 0xFFFFFFFFu; mid = 0xFFFFFFFFu; }
        if (it % 11 == 1) { lo = 0; mid = 0; }
        const G3 x = from_limbs(lo, mid, hi);
        const G3 y = mul_pow2_small<S>(x);
        const i128 want = (i128)(((u128)modp(val(x)) * (u128)((((u128)1) << S) % P)) % P);
        if (modp(val(y)) != (gu64)want) { fails++; if (fails < 3) printf("mul_pow2_small<%d> mismatch hi=%d\n", S, hi); }
        const i128 lim = (i128)1 << 66;
        if (val(y) >= lim || val(y) <= -lim) { fails++; if (fails < 3) printf("mul_pow2_small<%d> range\n", S); }
    }
    return fails;
}

// dif_shift<LOGR> (the network the kernel runs) against the defining sums with root 2^(192/R), on random and extreme inputs
template <int LOGR>
static long check_network(std::mt19937_64 &rng)
{
    constexpr int R = 1 << LOGR;
    long fails = 0;
    if (!dif_shift_bounds_ok(LOGR)) { fails++; printf("dif_shift<%d>: bounds\n", LOGR); }
    gu64 wr = 1; // 2^(192/R) mod p
    for (int i = 0; i < 192 / R; i++) wr = (gu64)(((u128)wr * 2) % P);
    for (int it = 0; it < 3000; it++) {
        gu64 in[R]; G3 v[R];
        for (int a = 0; a < R; a++) {
            in[a] = it % 4 == 0 ? ~0ull : (it % 4 == 1 ? ((rng() & 1) ? ~0ull : 0ull) : rng());
            v[a] = from_u64(in[a]);
        }
        dif_shift<LOGR>(v);
        for (int k = 0; k < R; k++) {
            u128 acc = 0; gu64 wk = 1;
            for (int i = 0; i < k; i++) wk = (gu64)(((u128)wk * wr) % P);
            gu64 t = 1;
            for (int a = 0; a < R; a++) { acc = (acc + (u128)(in[a] % P) * t) % P; t = (gu64)(((u128)t * wk) % P); }
            int br = 0;
            for (int i = 0; i < LOGR; i++) br |= ((k >> i) & 1) << (LOGR - 1 - i);
            if (canon(v[br]) != (gu64)acc) { fails++; if (fails < 3) printf("dif_shift<%d> mismatch k=%d\n", LOGR, k); }
            const i128 lim = (i128)1 << 70;
            if (val(v[br]) >= lim || val(v[br]) <= -lim) { fails++; if (fails < 3) printf("dif_shift<%d> range\n", LOGR); }
        }
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
        if (it % 7 == 0) { x = from_limbs((gu32)rng(), (gu32)rng(), (int32_t)(rng() % 127) - 63); vx = val(x); }
        if (val(x) != vx) { fails++; if (fails < 5) printf("add/sub mismatch\n"); }
        gu64 y = to_u64(x);
        if (modp((i128)y) != modp(vx)) { fails++; if (fails < 5) printf("to_u64 mismatch hi=%d\n", limb2(x)); }
        gu64 cn = canon(x);
        if (cn >= P || cn != modp(vx)) { fails++; if (fails < 5) printf("canon mismatch\n"); }
        gu64 w = pick();
        G3 m = mul(x, w);
        if (limb2(m) < -2 || limb2(m) > 1) { fails++; if (fails < 5) printf("mul hi range %d\n", limb2(m)); }
        u128 pr = (u128)modp(vx) * (u128)(w % P) % P;
        if (modp(val(m)) != (gu64)pr) { fails++; if (fails < 5) printf("mul mismatch\n"); }
        gu64 mr = mul_red(y, w);
        if (modp((i128)mr) != (gu64)pr) { fails++; if (fails < 5) printf("mul_red mismatch\n"); }
        G3 m2 = mul_u64(y, w);
        if (modp(val(m2)) != (gu64)pr) { fails++; if (fails < 5) printf("mul_u64 mismatch\n"); }
    }
    // shift twiddles: every multiple of 6 below 96 (the twiddles of radix-4 ... radix-32 networks), and the word-aligned cases
    fails += check_pow2<6>(rng, edge) + check_pow2<12>(rng, edge) + check_pow2<18>(rng, edge) + check_pow2<24>(rng, edge) +
             check_pow2<30>(rng, edge) + check_pow2<36>(rng, edge) + check_pow2<42>(rng, edge) + check_pow2<48>(rng, edge) +
             check_pow2<54>(rng, edge) + check_pow2<60>(rng, edge) + check_pow2<66>(rng, edge) + check_pow2<72>(rng, edge) +
             check_pow2<78>(rng, edge) + check_pow2<84>(rng, edge) + check_pow2<90>(rng, edge) + check_pow2<32>(rng, edge) +
             check_pow2<64>(rng, edge) + check_pow2<1>(rng, edge) + check_pow2<95>(rng, edge);
    fails += check_pow2_small<6>(rng) + check_pow2_small<12>(rng) + check_pow2_small<18>(rng) + check_pow2_small<24>(rng) +
             check_pow2_small<30>(rng) + check_pow2_small<36>(rng) + check_pow2_small<42>(rng) + check_pow2_small<48>(rng) +
             check_pow2_small<54>(rng) + check_pow2_small<60>(rng) + check_pow2_small<66>(rng) + check_pow2_small<72>(rng) +
             check_pow2_small<78>(rng) + check_pow2_small<84>(rng) + check_pow2_small<90>(rng) + check_pow2_small<1>(rng) +
             check_pow2_small<31>(rng) + check_pow2_small<95>(rng);
    fails += check_network<1>(rng) + check_network<2>(rng) + check_network<3>(rng) + check_network<4>(rng) + check_network<5>(rng);
    printf("fails %ld\n", fails);
    return fails != 0;
}
