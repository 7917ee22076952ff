// Host check of galois_amd/csrc/gfa_m32_net.h: the in-register DIF networks of the signed-Montgomery NTT kernels with the
// compile-time reduction schedule (DifSched) that admits primes up to 2^29.  The networks are run here on a range-CHECKING
// integer: every sum, difference and product operand is tested against the int32 range, for every network shape the kernels
// instantiate (radix 4 .. 64) and every prime class (BMAX = 64, 32, 8, 4) with the largest prime of the class, on inputs at the
// magnitude limit (all +-(p - 1), alternating signs, random).  Outputs are compared with a direct DFT (the reference's
// fft_jit computes exactly that: src/galois/_domains/_function.py:246-384).  Built and run by tests/test_host_logic.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

static long g_range_fail = 0;
struct m32v { // an int32 register whose every write is range-checked
    int64_t x;
    m32v() : x(0) {}
    m32v(int64_t v) : x(v)
    {
        if (v < INT32_MIN || v > INT32_MAX) g_range_fail++;
    }
};
static inline m32v m32_add(m32v u, m32v v) { return m32v(u.x + v.x); }
static inline m32v m32_sub(m32v u, m32v v) { return m32v(u.x - v.x); }
// the device expression: m = lo32(x * wp) (signed), result = hi32(x * wm) - hi32(m * p)
static inline m32v m32_mulm(m32v x, int wm, int wp, int p)
{
    const int32_t xi = (int32_t)x.x;
    const int32_t m = (int32_t)((uint32_t)xi * (uint32_t)wp);
    const int64_t hi1 = ((int64_t)xi * wm) >> 32, hi2 = ((int64_t)m * p) >> 32;
    return m32v(hi1 - hi2);
}
#define GFA_M32_HOST_CHECK
#include "gfa_m32_net.h"

typedef unsigned long long u64;
static u64 powmod(u64 b, u64 e, u64 p)
{
    u64 r = 1;
    b %= p;
    while (e) {
        if (e & 1) r = (unsigned __int128)r * b % p;
        b = (unsigned __int128)b * b % p;
        e >>= 1;
    }
    return r;
}
static bool is_prime(u64 n)
{
    if (n < 2) return false;
    for (u64 d = 2; d * d <= n; d++)
        if (n % d == 0) return false;
    return true;
}
static uint32_t inv_2_32(uint32_t p)
{
    uint32_t x = p;
    for (int i = 0; i < 5; i++) x *= 2u - p * x;
    return x;
}
static int mont_centred(u64 w, u64 p)
{
    const u64 m = (w << 32) % p;
    return m > p / 2 ? (int)((int64_t)m - (int64_t)p) : (int)m;
}
static int brev(int x, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

template <int LOGR, int BMAX>
static long check(std::mt19937_64 &rng)
{
    constexpr int R = 1 << LOGR;
    // the largest prime of the class with a 64-th root of unity
    u64 p = (((u64)1 << 31) / BMAX - 1) / 64 * 64 + 1;
    while (!is_prime(p)) p -= 64;
    u64 w = 0;
    for (u64 a = 2; !w; a++) {
        const u64 c = powmod(a, (p - 1) / R, p);
        if (powmod(c, R / 2, p) == p - 1) w = c;
    }
    const uint32_t pinv = inv_2_32((uint32_t)p);
    std::vector<int> net(2 * (R / 2 > 0 ? R / 2 : 1));
    for (int j = 0; j < R / 2; j++) {
        net[2 * j] = mont_centred(powmod(w, j, p), p);
        net[2 * j + 1] = (int)((uint32_t)net[2 * j] * pinv);
    }
    const int one = mont_centred(1, p), onep = (int)((uint32_t)one * pinv);
    long fails = 0;
    const long range_before = g_range_fail;
    for (int it = 0; it < 400; it++) {
        int64_t in[R];
        for (int i = 0; i < R; i++) {
            switch (it % 8) {
            case 0: in[i] = (int64_t)p - 1; break;
            case 1: in[i] = -((int64_t)p - 1); break;
            case 2: in[i] = (i & 1) ? (int64_t)p - 1 : -((int64_t)p - 1); break;
            case 3: in[i] = (brev(i, LOGR) & 1) ? (int64_t)p - 1 : -((int64_t)p - 1); break;
            case 4: in[i] = (rng() & 1) ? (int64_t)p - 1 : -((int64_t)p - 1); break;
            case 5: in[i] = i == (int)(rng() % R) ? (int64_t)p - 1 : 0; break;
            default: in[i] = (int64_t)(rng() % (2 * p - 1)) - ((int64_t)p - 1);
            }
        }
        m32v v[R];
        for (int i = 0; i < R; i++) v[i] = m32v(in[i]);
        dif<LOGR, BMAX>(v, net.data(), (int)p, one, onep);
        for (int k = 0; k < R; k++) {
            unsigned __int128 acc = 0;
            for (int a = 0; a < R; a++) {
                const u64 x = (u64)((in[a] % (int64_t)p + (int64_t)p) % (int64_t)p);
                acc += (unsigned __int128)x * powmod(w, (u64)a * k % R, p);
            }
            const u64 want = (u64)(acc % p);
            const int64_t got = v[brev(k, LOGR)].x;
            if ((u64)((got % (int64_t)p + (int64_t)p) % (int64_t)p) != want) { fails++; if (fails < 4) printf("dif<%d,%d> p=%llu output %d differs\n", LOGR, BMAX, p, k); }
            if (got >= (int64_t)BMAX * (int64_t)p || got <= -(int64_t)BMAX * (int64_t)p) { fails++; if (fails < 4) printf("dif<%d,%d> output %d beyond BMAX*p\n", LOGR, BMAX, k); }
        }
    }
    if (g_range_fail != range_before) { fails++; printf("dif<%d,%d> p=%llu: %ld values left the int32 range\n", LOGR, BMAX, p, g_range_fail - range_before); }
    printf("dif<%d,%d>: p = %llu, %d reductions, %s\n", LOGR, BMAX, p, DifSched<LOGR, BMAX>::tab.reductions, fails ? "FAIL" : "ok");
    return fails;
}

// the product itself: any int32 operand, |wm| < p  ->  result in (-p, p) and congruent to x * w
static long check_mulm(std::mt19937_64 &rng)
{
    long fails = 0;
    for (u64 p : {(u64)469762049, (u64)536608769, (u64)268369921, (u64)67043329, (u64)7340033}) {
        const uint32_t pinv = inv_2_32((uint32_t)p);
        for (int it = 0; it < 200000; it++) {
            int32_t x = (int32_t)rng();
            if (it % 7 == 0) x = INT32_MAX;
            if (it % 7 == 1) x = INT32_MIN;
            const int64_t wv = (int64_t)(rng() % (2 * p - 1)) - ((int64_t)p - 1); // Montgomery-form twiddle anywhere in (-p, p)
            const int wm = (int)wv, wp = (int)((uint32_t)wm * pinv);
            const int64_t r = m32_mulm(m32v(x), wm, wp, (int)p).x;
            if (r <= -(int64_t)p || r >= (int64_t)p) { fails++; if (fails < 4) printf("mulm range p=%llu\n", p); }
            // r * 2^32 == x * wm (mod p)
            const __int128 lhs = (__int128)r * ((__int128)1 << 32) - (__int128)x * wm;
            if (lhs % (__int128)p != 0) { fails++; if (fails < 4) printf("mulm value p=%llu\n", p); }
        }
    }
    printf("mulm: %s\n", fails ? "FAIL" : "ok");
    return fails;
}

int main()
{
    std::mt19937_64 rng(2026);
    long fails = check_mulm(rng);
    fails += check<2, 4>(rng) + check<3, 4>(rng) + check<4, 4>(rng) + check<5, 4>(rng) + check<6, 4>(rng);
    fails += check<2, 8>(rng) + check<3, 8>(rng) + check<4, 8>(rng) + check<5, 8>(rng) + check<6, 8>(rng);
    fails += check<2, 32>(rng) + check<3, 32>(rng) + check<4, 32>(rng) + check<5, 32>(rng);
    fails += check<5, 64>(rng) + check<6, 64>(rng);
    // the schedule's claims: nothing to do where rounds 3-4 ran without it
    static_assert(DifSched<5, 32>::tab.reductions == 0 && DifSched<6, 64>::tab.reductions == 0, "p < 2^26 / 2^25 networks are unchanged");
    static_assert(DifSched<5, 4>::tab.reductions == 28 && DifSched<5, 8>::tab.reductions == 9, "counts quoted in gfa_m32_net.h");
    if (fails || g_range_fail) { printf("FAILED: %ld (range %ld)\n", fails, g_range_fail); return 1; }
    printf("m32 networks ok\n");
    return 0;
}
