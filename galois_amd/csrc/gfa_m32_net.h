// gfa_m32_net.h -- the in-register decimation-in-frequency networks of gfa_ntt_m32.hip on signed Montgomery representatives,
// with the compile-time magnitude bookkeeping that lets primes up to 2^29 through them.
//
// Kept in a header so that tests/csrc/m32_net_host_test.cpp can run the SAME networks on the host over a range-checking
// integer type (GFA_M32_HOST_CHECK: the test supplies m32v, m32_add, m32_sub, m32_mulm): every intermediate value of every
// network shape is then checked against the int32 range at the magnitude limit of its prime class, and the outputs against
// a direct DFT.  Reference behaviour replaced: fft_jit.implementation, src/galois/_domains/_function.py:246-384 (exact field
// arithmetic: any correct DFT algorithm reproduces its bits).
#pragma once

#ifndef GFA_M32_HOST_CHECK
typedef int m32v;
// v_mul_hi_i32 through inline assembly: the compiler matches the signed high product only while the sign extensions sit in
// the same basic block; once it hoists sext(p) out of a block it expands every product into four unsigned multiplies.
// "s": wave-uniform operand in a scalar register (kernel argument or scalar load) -- one constant-bus read per VOP3.
__device__ __forceinline__ int m32_mulhi_vs(int x, int s)
{
    int r;
    asm("v_mul_hi_i32 %0, %1, %2" : "=v"(r) : "v"(x), "s"(s));
    return r;
}
__device__ __forceinline__ m32v m32_add(m32v u, m32v x) { return (int)((unsigned)u + (unsigned)x); }
__device__ __forceinline__ m32v m32_sub(m32v u, m32v x) { return (int)((unsigned)u - (unsigned)x); }
// x * w * 2^-32 (mod p) as a representative in (-p, p); any int32 x, |wm| < p, wp = wm * p^-1 mod 2^32
// (uniform twiddle: wm, wp, p in scalar registers)
__device__ __forceinline__ m32v m32_mulm(m32v x, int wm, int wp, int p)
{
    const int m = (int)((unsigned)x * (unsigned)wp);
    return m32_mulhi_vs(x, wm) - m32_mulhi_vs(m, p);
}
#define M32_FN __device__ __forceinline__
#else
#define M32_FN inline
#endif

// Magnitudes inside a network, in units of p (|v| < b * p), followed at compile time.  A value that meets a product comes back to
// (-p, p) (b = 1); sums and untwiddled differences add their operands' bounds.  BMAX = floor(2^31 / p) is what an int32 holds:
// p < 2^26 -> 32, i.e. a radix-32 network never needs help (rounds 3-4); for 2^26 <= p < 2^29 (BMAX = 4) an operand is brought
// back with one product by the Montgomery form of 1 exactly where the NEXT butterfly would overflow -- 28 of the 160 operands of
// a radix-32 network (+3.5 vector instructions per point and network; BMAX = 8, p < 2^28: 9 operands).  [2^29, 2^30) would need
// 79 (no gain over the lazy Shoup kernels of gfa_ntt.hip), and from 2^30 a sum of two representatives leaves int32 altogether:
// those primes stay there.
template <int LOGR, int BMAX>
struct DifSched {
    static constexpr int R = 1 << LOGR;
    struct Tab {
        bool red[LOGR > 0 ? LOGR : 1][R];
        int reductions;
    };
    static constexpr Tab make()
    {
        Tab t{};
        int b[R] = {};
        for (int i = 0; i < R; i++) b[i] = 1;
        for (int s = LOGR - 1, lv = 0; s >= 0; s--, lv++) {
            const int half = 1 << s;
            for (int blk = 0; blk < R; blk += 2 * half)
                for (int j = 0; j < half; j++) {
                    const int i0 = blk + j, i1 = i0 + half;
                    for (int rep = 0; rep < 2 && b[i0] + b[i1] > BMAX; rep++) {
                        const int big = b[i0] >= b[i1] ? i0 : i1;
                        t.red[lv][big] = true;
                        t.reductions++;
                        b[big] = 1;
                    }
                    const int sum = b[i0] + b[i1];
                    b[i0] = sum;
                    b[i1] = (j << (LOGR - 1 - s)) != 0 ? 1 : sum;
                }
        }
        return t;
    }
    static constexpr Tab tab = make();
};

// v[bitrev(k)] <- sum_a v[a] * w_R^(a*k);  net[2j], net[2j+1] = Montgomery form of w_R^j and its p^-1 companion (uniform).
// Inputs in (-p, p); outputs below BMAX * p in magnitude (every one of them meets a product next).
template <int LOGR, int BMAX>
M32_FN void dif(m32v (&v)[1 << LOGR], const int *__restrict__ net, int p, int one, int onep)
{
    constexpr int R = 1 << LOGR;
    static_assert(BMAX >= 2, "a butterfly adds two representatives");
    typedef DifSched<LOGR, BMAX> S;
#pragma unroll
    for (int s = LOGR - 1; s >= 0; s--) {
        const int half = 1 << s;
#pragma unroll
        for (int b = 0; b < R; b += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                m32v u = v[b + j], x = v[b + j + half];
                if (S::tab.red[LOGR - 1 - s][b + j]) u = m32_mulm(u, one, onep, p);
                if (S::tab.red[LOGR - 1 - s][b + j + half]) x = m32_mulm(x, one, onep, p);
                v[b + j] = m32_add(u, x);
                const m32v d = m32_sub(u, x);
                const int tj = j << (LOGR - 1 - s);
                if (tj != 0) v[b + j + half] = m32_mulm(d, net[2 * tj], net[2 * tj + 1], p);
                else v[b + j + half] = d;
            }
        }
    }
}

