// Host check of galois_amd/csrc/gfa_packed.h (r05): packed-digit sums / differences / negatives of GF(p^m), p odd, against digit-wise
// arithmetic -- the definition of add_vector / subtract_vector / negative_vector (src/galois/_domains/_calculate.py:150-285).
// Every element pair of the small fields, random and edge pairs of every field shape with 8192 < q <= 2^20 the scheme accepts; the
// fields it must refuse.  Built and run by tests/test_host_logic.py.
#include "gfa_packed.h"
#include <cstdio>
#include <random>
using namespace gfa_packed;

static pu32 digitwise(pu32 p, pu32 m, pu32 x, pu32 y, int op)
{
    pu32 r = 0, scale = 1;
    for (pu32 i = 0; i < m; i++) {
        const pu32 a = x % p, b = y % p;
        x /= p; y /= p;
        const pu32 d = op == 0 ? (a + b) % p : op == 1 ? (a + p - b) % p : (p - a) % p;
        r += d * scale;
        scale *= p;
    }
    return r;
}

static long check(pu32 p, pu32 m, std::mt19937 &rng)
{
    Plan pl;
    if (!make_plan(p, m, &pl)) { printf("GF(%u^%u): refused\n", p, m); return 1; }
    std::vector<pu32> t;
    build_tables(pl, t);
    long fails = 0;
    auto one = [&](pu32 x, pu32 y) {
        const pu32 a = to_packed(pl, t.data(), x), b = to_packed(pl, t.data(), y);
        if (from_packed(pl, t.data(), a) != x) fails++;
        if (from_packed(pl, t.data(), lin_packed<0>(pl, a, b)) != digitwise(p, m, x, y, 0)) fails++;
        if (from_packed(pl, t.data(), lin_packed<1>(pl, a, b)) != digitwise(p, m, x, y, 1)) fails++;
        if (from_packed(pl, t.data(), lin_packed<2>(pl, a, b)) != digitwise(p, m, x, y, 2)) fails++;
    };
    if (pl.q <= 729) {
        for (pu32 x = 0; x < pl.q; x++)
            for (pu32 y = 0; y < pl.q; y++) one(x, y);
    } else {
        for (pu32 x = 0; x < pl.q; x += 1 + pl.q / 50000) one(x, pl.q - 1 - x); // the exact-quotient claim over the whole range
        const pu32 edge[6] = {0, 1, pl.q - 1, pl.q - 2, pl.P, pl.P - 1};
        for (pu32 a : edge)
            for (pu32 b : edge) one(a, b);
        for (int it = 0; it < 200000; it++) one(rng() % pl.q, rng() % pl.q);
    }
    printf("GF(%u^%u): W %u, split %u + %u digits, %u table words, %s\n", p, m, pl.W, pl.hl, pl.hh, pl.words, fails ? "FAIL" : "ok");
    return fails;
}

// two packed words (r06): GF(3^11), GF(3^12)
static long check2(pu32 p, pu32 m, std::mt19937 &rng)
{
    Plan2 pl;
    if (!make_plan2(p, m, &pl)) { printf("GF(%u^%u): two-word plan refused\n", p, m); return 1; }
    std::vector<pu32> t;
    build_tables2(pl, t);
    const pu32 q = ipow(p, m);
    long fails = 0;
    auto one = [&](pu32 x, pu32 y) {
        const Pk2 a = to_packed2(pl, t.data(), x), b = to_packed2(pl, t.data(), y);
        if (from_packed2(pl, t.data(), a) != x) fails++;
        if (from_packed2(pl, t.data(), lin_packed2<0>(pl, a, b)) != digitwise(p, m, x, y, 0)) fails++;
        if (from_packed2(pl, t.data(), lin_packed2<1>(pl, a, b)) != digitwise(p, m, x, y, 1)) fails++;
        if (from_packed2(pl, t.data(), lin_packed2<2>(pl, a, b)) != digitwise(p, m, x, y, 2)) fails++;
    };
    for (pu32 x = 0; x < q; x += 1 + q / 200000) one(x, q - 1 - x); // the exact-quotient claim over the whole range
    const pu32 edge[8] = {0, 1, q - 1, q - 2, pl.PL, pl.PL - 1, pl.PL + 1, q - pl.PL};
    for (pu32 a : edge)
        for (pu32 b : edge) one(a, b);
    for (int it = 0; it < 300000; it++) one(rng() % q, rng() % q);
    printf("GF(%u^%u): two words of %u + %u digits, %u table words, %s\n", p, m, pl.lo.m, pl.hi.m, pl.words, fails ? "FAIL" : "ok");
    return fails;
}

// products: mul_digits against the textbook product of the digit polynomials reduced by x^m + irr, for a (random) monic irr -- the
// arithmetic does not need irreducibility -- incl. all-(p-1) operands (the bound replay's worst case)
template <int M>
static long check_mul(pu32 p, std::mt19937 &rng)
{
    Plan pl;
    if (!make_plan(p, M, &pl)) return 0; // (fields the scheme refuses are covered above)
    std::vector<pu32> t;
    build_tables(pl, t);
    long fails = 0, fields = 0;
    for (int poly = 0; poly < 6; poly++) {
        pu32 irr[M]; // coefficient of x^j of the monic polynomial, j < M
        for (int j = 0; j < M; j++) irr[j] = poly == 0 ? p - 1 - (j & 1) : rng() % p; // poly 0: nir small but non-zero; others random
        if (poly == 1) for (int j = 0; j < M; j++) irr[j] = 1;                         // nir = p - 1 everywhere: the largest folds
        MulAux ax{};
        for (int j = 0; j < M; j++) ax.nir[j] = irr[j] ? p - irr[j] : 0;
        ax.mu32 = (pu32)(((uint64_t)1 << 32) / p);
        if (!mul_bound_ok(pl, ax)) continue;
        fields++;
        for (int it = 0; it < 20000; it++) {
            pu32 x = rng() % pl.q, y = rng() % pl.q;
            if (it == 0) x = y = pl.q - 1;
            if (it == 1) { x = pl.q - 1; y = 1; }
            // reference: digit polynomials, product, fold from the top, everything reduced mod p at once
            uint64_t c[2 * M - 1] = {};
            pu32 xd[M], yd[M], xx = x, yy = y;
            for (int i = 0; i < M; i++) { xd[i] = xx % p; xx /= p; yd[i] = yy % p; yy /= p; }
            for (int i = 0; i < M; i++)
                for (int j = 0; j < M; j++) c[i + j] = (c[i + j] + (uint64_t)xd[i] * yd[j]) % p;
            for (int k = 2 * M - 2; k >= M; k--)
                for (int j = 0; j < M; j++) c[k - M + j] = (c[k - M + j] + c[k] * ax.nir[j]) % p;
            pu32 want = 0;
            for (int i = M - 1; i >= 0; i--) want = want * p + (pu32)c[i];
            const pu32 got = mul_digits<M>(pl, ax, to_packed(pl, t.data(), x), to_packed(pl, t.data(), y));
            if (got != want) fails++;
        }
    }
    printf("GF(%u^%d) products: %ld polynomial(s) inside the 32-bit bound, %s\n", p, M, fields, fails ? "FAIL" : "ok");
    return fails;
}

// quotients of GF(p^2) by the norm (r06): (a / b) * b == a and (1 / b) * b == 1 through the textbook digit product, for random
// irreducible x^2 + c1 x + c0 (no root mod p), every b of a small slice and random pairs; b == 0 is flagged
template <bool WIDE = false>
static long check_div2(pu32 p, std::mt19937 &rng)
{
    long fails = 0;
    for (int poly = 0; poly < 4; poly++) {
        pu32 c1, c0;
        for (;;) {
            c1 = rng() % p; c0 = rng() % p;
            bool root = false;
            for (pu32 x = 0; x < p && !root; x++) root = ((uint64_t)x * x + (uint64_t)c1 * x + c0) % p == 0;
            if (!root) break;
        }
        pu32 nir[8] = {c0 ? p - c0 : 0, c1 ? p - c1 : 0};
        Div2Aux ax;
        if (!make_div2(p, 2, nir, &ax, WIDE)) { printf("GF(%u^2): refused\n", p); return 1; }
        std::vector<pu32> inv;
        build_inverse_table(p, inv);
        auto mul = [&](pu32 x, pu32 y) { // textbook: (x0 + x1 X)(y0 + y1 X), X^2 = s X + t
            const uint64_t x0 = x % p, x1 = x / p, y0 = y % p, y1 = y / p;
            const uint64_t hi = x1 * y1 % p;
            const uint64_t d0 = (x0 * y0 + ax.t * hi) % p, d1 = (x0 * y1 + x1 * y0 + ax.s * hi) % p;
            return (pu32)(d1 * p + d0);
        };
        const pu32 q = p * p;
        auto one = [&](pu32 a, pu32 b) {
            bool z = false, z2 = false;
            const pu32 qt = div2<false, WIDE>(ax, inv.data(), a, b, &z), rc = div2<true, WIDE>(ax, inv.data(), a, b, &z2);
            if (z != (b == 0) || z2 != (b == 0)) fails++;
            if (b == 0) return;
            if (qt >= q || rc >= q || mul(qt, b) != a || mul(rc, b) != 1) fails++;
        };
        for (pu32 b = 0; b < 3 * p; b++) one(rng() % q, b);
        const pu32 edge[6] = {0, 1, q - 1, p, p - 1, q - p};
        for (pu32 a : edge)
            for (pu32 b : edge) one(a, b);
        for (int it = 0; it < 100000; it++) one(rng() % q, rng() % q);
    }
    printf("GF(%u^2) quotients by the norm: %s\n", p, fails ? "FAIL" : "ok");
    return fails;
}

// quotients of GF(p^3) by Cramer's rule (r06): (a / b) * b == a and (1 / b) * b == 1 through the textbook digit product, for random
// irreducible cubics (no root mod p: a cubic without roots is irreducible)
template <bool WIDE = false>
static long check_div3(pu32 p, std::mt19937 &rng)
{
    long fails = 0;
    for (int poly = 0; poly < 4; poly++) {
        pu32 c2, c1, c0;
        for (;;) {
            c2 = rng() % p; c1 = rng() % p; c0 = rng() % p;
            bool root = false;
            for (pu32 x = 0; x < p && !root; x++) root = (((uint64_t)x * x % p) * x + (uint64_t)c2 * x % p * x + (uint64_t)c1 * x + c0) % p == 0;
            if (!root) break;
        }
        pu32 nir[8] = {c0 ? p - c0 : 0, c1 ? p - c1 : 0, c2 ? p - c2 : 0};
        Div3Aux ax;
        if (!make_div3(p, 3, nir, &ax, WIDE)) { printf("GF(%u^3): refused\n", p); return 1; }
        std::vector<pu32> inv;
        build_inverse_table(p, inv);
        auto mul = [&](pu32 x, pu32 y) {
            uint64_t xd[3] = {x % p, x / p % p, x / p / p}, yd[3] = {y % p, y / p % p, y / p / p}, c[5] = {0, 0, 0, 0, 0};
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) c[i + j] = (c[i + j] + xd[i] * yd[j]) % p;
            for (int k = 4; k >= 3; k--)
                for (int j = 0; j < 3; j++) c[k - 3 + j] = (c[k - 3 + j] + c[k] * nir[j]) % p;
            return (pu32)((c[2] * p + c[1]) * p + c[0]);
        };
        const pu32 q = p * p * p;
        auto one = [&](pu32 a, pu32 b) {
            bool z = false, z2 = false;
            const pu32 qt = div3<false, WIDE>(ax, inv.data(), a, b, &z), rc = div3<true, WIDE>(ax, inv.data(), a, b, &z2);
            if (z != (b == 0) || z2 != (b == 0)) fails++;
            if (b == 0) return;
            if (qt >= q || rc >= q || mul(qt, b) != a || mul(rc, b) != 1) fails++;
        };
        for (pu32 b = 0; b < 4 * p; b++) one(rng() % q, b);
        const pu32 edge[7] = {0, 1, q - 1, p, p - 1, p * p, q - p};
        for (pu32 a : edge)
            for (pu32 b : edge) one(a, b);
        for (int it = 0; it < 100000; it++) one(rng() % q, rng() % q);
    }
    printf("GF(%u^3) quotients by Cramer's rule: %s\n", p, fails ? "FAIL" : "ok");
    return fails;
}

int main()
{
    std::mt19937 rng(5);
    long fails = 0;
    fails += check_div3(41, rng) + check_div3(97, rng) + check_div3(101, rng) + check_div3(67, rng);
    // r06, fields without tables (q > 2^20): the exact digit split over the whole 32-bit range, the largest primes of each form
    fails += check_div3<true>(103, rng) + check_div3<true>(251, rng) + check_div3<true>(1021, rng) + check_div3<true>(1621, rng);
    fails += check_div2<true>(1031, rng) + check_div2<true>(8191, rng) + check_div2<true>(32771, rng) + check_div2<true>(37813, rng);
    for (pu32 d : {3u, 103u, 1621u, 37813u, 65521u, 2147483647u}) {
        const ExactDiv e = make_exact_div(d);
        for (unsigned long long x : {0ull, 1ull, (unsigned long long)d - 1, (unsigned long long)d, (unsigned long long)d + 1, 0xffffffffull, 0xfffffffeull, 0x80000000ull, (unsigned long long)d * d % 0x100000000ull})
            if (exact_div(e, (pu32)x) != (pu32)x / d) fails++;
        for (int i = 0; i < 200000; i++) { const pu32 x = rng(); if (exact_div(e, x) != x / d) fails++; }
    }
    Div2Aux w2; Div3Aux w3; pu32 nz[8] = {1, 1, 1};
    if (make_div2(37831, 2, nz, &w2, true) || make_div2(1021, 2, nz, &w2, true) || make_div3(1627, 3, nz, &w3, true) || make_div3(101, 3, nz, &w3, true)) { printf("a prime outside the wide forms was accepted\n"); fails++; }
    fails += check_div2(997, rng) + check_div2(257, rng) + check_div2(1021, rng) + check_div2(509, rng) + check_div2(251, rng) + check_div2(191, rng);
    const pu32 fields[][2] = {{3, 2}, {3, 5}, {5, 3}, {7, 2}, {3, 9}, {3, 10}, {5, 6}, {5, 7}, {5, 8}, {7, 5}, {7, 6}, {7, 7}, {11, 4}, {11, 5}, {13, 5},
                              {17, 4}, {31, 4}, {41, 3}, {97, 3}, {101, 2}, {257, 2}, {1021, 2}};
    for (auto &f : fields) fails += check(f[0], f[1], rng);
    fails += check_mul<2>(997, rng) + check_mul<2>(257, rng) + check_mul<3>(97, rng) + check_mul<3>(41, rng) + check_mul<4>(31, rng) + check_mul<4>(17, rng);
    fails += check_mul<5>(13, rng) + check_mul<5>(11, rng) + check_mul<6>(7, rng) + check_mul<7>(7, rng) + check_mul<7>(5, rng) + check_mul<8>(5, rng) + check_mul<8>(3, rng);
    fails += check2(3, 11, rng) + check2(3, 12, rng);
    Plan2 pl2;
    if (make_plan2(7, 7, &pl2) || make_plan2(3, 13, &pl2) || make_plan2(2, 12, &pl2)) { // one word suffices / order above 2^20 / even characteristic
        printf("a field outside the two-word scheme was accepted\n");
        fails++;
    }
    Plan pl;
    // refused: even characteristic, prime fields, more than 32 packed bits (3^11: 33), orders above 2^20, digits above 1021
    if (make_plan(2, 8, &pl) || make_plan(7, 1, &pl) || make_plan(3, 11, &pl) || make_plan(3, 13, &pl) || make_plan(1031, 2, &pl) || make_plan(101, 4, &pl)) {
        printf("a field outside the scheme was accepted\n");
        fails++;
    }
    if (fails) { printf("FAILED: %ld\n", fails); return 1; }
    printf("packed digits ok\n");
    return 0;
}
