// Host check of galois_amd/csrc/gfa_fermat_tw.h (r06): the first inter-network twiddles w^(m k0) of the one-pass GF(65537)
// kernel, formed in registers from two seeds instead of streamed from a table.  The SAME header is instantiated here over a
// range-checking integer: every product, fold and balanced fold is tested against the int32 range, for every column m and every
// output k0 of several primitive 2^16-th roots of unity, and the applied product is compared with q * w^(m k0) mod 65537 for
// network outputs q at the magnitude limit the kernel's contract allows (|q| < 2^29), the generated networks' own bound, and
// random values.  What the twiddles stand in for: the factors w^(m k0) between the stages of the reference's fft_jit
// (src/galois/_domains/_function.py:246-384).  Built and run by tests/test_host_logic.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

static long g_range_fail = 0;
struct rv { // an int32 register whose every write is range-checked
    int64_t x;
    rv() : x(0) {}
    rv(int64_t v) : x(v)
    {
        if (v < INT32_MIN || v > INT32_MAX) g_range_fail++;
    }
};
static inline rv fm_mulc(rv a, rv b) { return rv(a.x * b.x); }
// lo16(x) - (x >> 16): v_sub_u32_sdwa dst, x.WORD_0, sext(x.WORD_1)
static inline rv fm_fold(rv t)
{
    const int32_t v = (int32_t)t.x;
    return rv((int64_t)(uint16_t)v - (int64_t)(int16_t)(v >> 16));
}
// sext16(x) - ((x + 2^15) >> 16): v_sub_u32_sdwa dst, sext(x.WORD_0), sext((x + 0x8000).WORD_1)
static inline rv fm_bfold(rv t)
{
    const int32_t v = (int32_t)t.x;
    const int32_t t2 = (int32_t)((uint32_t)v + 0x8000u); // the device adds in unsigned arithmetic: a wrap is not an overflow...
    if (t.x + 0x8000 > INT32_MAX) g_range_fail++;         // ...but the formula needs the true sum
    return rv((int64_t)(int16_t)v - (int64_t)(int16_t)(t2 >> 16));
}
#define FM_TW_FN static inline
#include "gfa_fermat_tw.h"

static const int64_t P = 65537;
static int64_t modp(int64_t x) { return ((x % P) + P) % P; }
static int64_t powmod(int64_t b, int64_t e)
{
    int64_t r = 1;
    b = modp(b);
    while (e) {
        if (e & 1) r = r * b % P;
        b = b * b % P;
        e >>= 1;
    }
    return r;
}
static int balanced(int64_t c) { return c > 32768 ? (int)(c - P) : (int)c; }

int main()
{
    std::mt19937_64 rng(7);
    long bad = 0, checked = 0;
    int64_t max_t = 0, max_tight = 0;
    const int64_t qs_fixed[] = {(1ll << 29) - 1, -((1ll << 29) - 1), 374808382, -374808382, 0, 1, -1, 65536, -65536, 98303, -32767};
    for (int64_t e : {1, 3, 12345, 65535, 32769}) { // odd exponents of the generator 3: primitive 2^16-th roots of unity
        const int64_t w = powmod(3, e);
        for (int m = 0; m < 1024; m++) {
            const rv seed1(balanced(powmod(w, m))), seed8(balanced(powmod(w, 8 * m)));
            rv A[8], B[8];
            fm_tw_progressions(seed1, seed8, A, B);
            for (int i = 1; i < 8; i++) {
                if (modp(A[i].x) != powmod(w, 8ll * m * i) || modp(B[i].x) != powmod(w, (int64_t)m * i)) bad++;
                max_tight = std::max<int64_t>(max_tight, std::max(std::llabs(A[i].x), std::llabs(B[i].x)));
            }
            for (int k0 = 1; k0 < 64; k0++) {
                const rv T = fm_tw_of(A, B, k0);
                const int64_t want = powmod(w, (int64_t)m * k0);
                if (modp(T.x) != want) bad++;
                max_t = std::max<int64_t>(max_t, std::llabs(T.x));
                for (int j = 0; j < 14; j++) {
                    const int64_t q = j < 11 ? qs_fixed[j] : (int64_t)(rng() % ((1ull << 30) - 1)) - ((1ll << 29) - 1);
                    const rv out = fm_tw_apply(rv(q), T);
                    if (modp(out.x) != modp(q) * want % P) bad++;
                    if (out.x < -32767 || out.x > 98303) bad++;
                    checked++;
                }
            }
        }
    }
    printf("fermat_tw: %ld applications, max |tight| %lld, max |T| %lld, range failures %ld, wrong %ld\n", checked, (long long)max_tight,
           (long long)max_t, g_range_fail, bad);
    if (max_tight > 32770 || max_t > 49153) bad++;
    return (bad || g_range_fail) ? 1 : 0;
}
