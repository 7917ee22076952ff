// gfa_big.hip -- element-wise arithmetic in finite fields of order above 2^128 (r05).
//
// The reference has no upper bound on the order: every field beyond int64 runs as dtype=object arrays of Python integers
// through its pure-Python ufuncs (src/galois/_domains/_meta.py:38-41, _fields/_ufunc.py:36-48; scalar formulas
// _domains/_calculate.py:133-592).  gfa_wide.hip serves 2^64 <= q <= 2^128 on two limbs; this file is the k-limb path for
// everything larger, up to 1024 bits: NL = 4, 8 or 16 little-endian 64-bit limbs per element (interleaved: element i at words
// NL*i .. NL*i + NL - 1), the same three arithmetic kinds as gfa_wide.hip written as loops over the limbs:
//   BPRIME  GF(p): Montgomery products (CIOS over NL limbs), conditional-subtract add / sub, a^(p-2) inverse
//   BBIN    GF(2^m): xor add; shift-and-xor product with a masked reduction (m steps); m = 64 NL has its top bit implicit
//   BEXT    GF(p^m), p < 2^32, m <= 32: base-p digit vectors (long division by p), schoolbook product reduced by the irreducible
//           polynomial (multiply_vector), Itoh-Tsujii inverse
// One element per lane.  A coverage path: exact and simple, a few thousand instructions per product.
#include <algorithm>
#include <map>
#include <mutex>

#include "gfa_internal.h"

using namespace gfa;

namespace {

typedef unsigned __int128 u128;
enum { BKIND_PRIME = 1, BKIND_BIN = 2, BKIND_EXT = 3 };
constexpr int BMAXL = 16;  // limbs
constexpr int BMAXD = 32;  // digits of an extension-field element

struct BField {
    int kind, m, nl;
    u64 p[BMAXL];     // BPRIME: the modulus.  BEXT: characteristic in p[0]
    u64 nprime;       // BPRIME: -p^-1 mod 2^64
    u64 r2[BMAXL];    // BPRIME: 2^(128 NL) mod p
    u64 einv[BMAXL];  // BPRIME: p - 2.  BBIN: 2^m - 2
    u64 red[BMAXL];   // BBIN: irreducible polynomial without its x^m term
    u64 itr[BMAXL];   // BEXT: (q - 1) / (p - 1) - 1
    u32 irr[BMAXD];   // BEXT: irreducible polynomial minus x^m, digits of degree m-1 .. 0
};

template <int NL>
struct WN {
    u64 l[NL];
};

template <int NL> __device__ __forceinline__ WN<NL> w_zero() { WN<NL> r; for (int i = 0; i < NL; i++) r.l[i] = 0; return r; }
template <int NL> __device__ __forceinline__ WN<NL> w_one() { WN<NL> r = w_zero<NL>(); r.l[0] = 1; return r; }
template <int NL> __device__ __forceinline__ WN<NL> w_from(const u64 *s) { WN<NL> r; for (int i = 0; i < NL; i++) r.l[i] = s[i]; return r; }
template <int NL> __device__ __forceinline__ bool w_is_zero(const WN<NL> &a) { u64 o = 0; for (int i = 0; i < NL; i++) o |= a.l[i]; return o == 0; }
template <int NL> __device__ __forceinline__ bool w_ge(const WN<NL> &a, const WN<NL> &b)
{
    for (int i = NL - 1; i >= 0; i--) {
        if (a.l[i] > b.l[i]) return true;
        if (a.l[i] < b.l[i]) return false;
    }
    return true;
}
template <int NL> __device__ __forceinline__ WN<NL> w_sub(const WN<NL> &a, const WN<NL> &b)
{
    WN<NL> r;
    u64 borrow = 0;
    for (int i = 0; i < NL; i++) {
        const u64 t = a.l[i] - b.l[i], t2 = t - borrow;
        borrow = (a.l[i] < b.l[i]) | (t < borrow);
        r.l[i] = t2;
    }
    return r;
}
template <int NL> __device__ __forceinline__ WN<NL> w_add(const WN<NL> &a, const WN<NL> &b, u64 *carry)
{
    WN<NL> r;
    u64 c = 0;
    for (int i = 0; i < NL; i++) {
        const u64 t = a.l[i] + b.l[i], t2 = t + c;
        c = (t < a.l[i]) | (t2 < t);
        r.l[i] = t2;
    }
    *carry = c;
    return r;
}
template <int NL> __device__ __forceinline__ u64 w_bit(const WN<NL> &a, int i) { return (a.l[i >> 6] >> (i & 63)) & 1; }
template <int NL> __device__ __forceinline__ int w_top_bit(const WN<NL> &a)
{ // index of the highest set bit, -1 for zero
    for (int i = NL - 1; i >= 0; i--)
        if (a.l[i]) return 64 * i + 63 - __clzll((long long)a.l[i]);
    return -1;
}

// ---------------------------------------------------------------- BPRIME
template <int NL> __device__ WN<NL> bp_add(const BField &f, const WN<NL> &a, const WN<NL> &b)
{
    u64 c;
    WN<NL> s = w_add(a, b, &c);
    const WN<NL> p = w_from<NL>(f.p);
    if (c || w_ge(s, p)) s = w_sub(s, p);
    return s;
}
template <int NL> __device__ WN<NL> bp_sub(const BField &f, const WN<NL> &a, const WN<NL> &b)
{
    if (w_ge(a, b)) return w_sub(a, b);
    u64 c;
    return w_sub(w_add(a, w_from<NL>(f.p), &c), b); // a + p - b (the carry cancels against the borrow)
}
// Montgomery product a * b * 2^(-64 NL) mod p (CIOS)
template <int NL> __device__ WN<NL> bp_mont(const BField &f, const WN<NL> &a, const WN<NL> &b)
{
    u64 t[NL + 2];
    for (int i = 0; i < NL + 2; i++) t[i] = 0;
    for (int i = 0; i < NL; i++) {
        u64 c = 0;
        for (int j = 0; j < NL; j++) {
            const u128 x = (u128)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (u64)x;
            c = (u64)(x >> 64);
        }
        u128 y = (u128)t[NL] + c;
        t[NL] = (u64)y;
        t[NL + 1] = (u64)(y >> 64);
        const u64 mq = t[0] * f.nprime;
        u128 x = (u128)mq * f.p[0] + t[0];
        c = (u64)(x >> 64);
        for (int j = 1; j < NL; j++) {
            x = (u128)mq * f.p[j] + t[j] + c;
            t[j - 1] = (u64)x;
            c = (u64)(x >> 64);
        }
        y = (u128)t[NL] + c;
        t[NL - 1] = (u64)y;
        t[NL] = t[NL + 1] + (u64)(y >> 64);
    }
    WN<NL> r;
    for (int i = 0; i < NL; i++) r.l[i] = t[i];
    const WN<NL> p = w_from<NL>(f.p);
    if (t[NL] || w_ge(r, p)) r = w_sub(r, p);
    return r;
}
template <int NL> __device__ WN<NL> bp_mul(const BField &f, const WN<NL> &a, const WN<NL> &b) { return bp_mont(f, bp_mont(f, a, w_from<NL>(f.r2)), b); }
template <int NL> __device__ WN<NL> bp_pow(const BField &f, const WN<NL> &a, const WN<NL> &e)
{
    const WN<NL> one_m = bp_mont(f, w_one<NL>(), w_from<NL>(f.r2));
    WN<NL> am = bp_mont(f, a, w_from<NL>(f.r2)), r = one_m;
    const int top = w_top_bit(e);
    for (int i = 0; i <= top; i++) {
        if (w_bit(e, i)) r = bp_mont(f, r, am);
        if (i < top) am = bp_mont(f, am, am);
    }
    return bp_mont(f, r, w_one<NL>());
}

// ---------------------------------------------------------------- BBIN
template <int NL> __device__ WN<NL> bb_mul(const BField &f, WN<NL> a, const WN<NL> &b)
{
    const int m = f.m;
    WN<NL> c = w_zero<NL>();
    const int tb = w_top_bit(b);
    for (int i = 0; i <= tb; i++) {
        const u64 bm = (u64)0 - w_bit(b, i);
        for (int k = 0; k < NL; k++) c.l[k] ^= a.l[k] & bm;
        const u64 tm = (u64)0 - w_bit(a, m - 1);
        for (int k = NL - 1; k > 0; k--) a.l[k] = (a.l[k] << 1) | (a.l[k - 1] >> 63);
        a.l[0] <<= 1;
        if (m < 64 * NL) { // clear bit m and above (m = 64 NL: the bit has left the top limb already)
            const int wl = m >> 6, wb = m & 63;
            a.l[wl] &= wb ? (((u64)1 << wb) - 1) : 0;
            for (int k = wl + 1; k < NL; k++) a.l[k] = 0;
        }
        for (int k = 0; k < NL; k++) a.l[k] ^= f.red[k] & tm;
    }
    return c;
}
template <int NL> __device__ WN<NL> bb_pow(const BField &f, WN<NL> a, const WN<NL> &e)
{
    WN<NL> r = w_one<NL>();
    const int top = w_top_bit(e);
    for (int i = 0; i <= top; i++) {
        if (w_bit(e, i)) r = bb_mul(f, r, a);
        if (i < top) a = bb_mul(f, a, a);
    }
    return r;
}

// ---------------------------------------------------------------- BEXT
struct BDigits {
    u32 d[BMAXD]; // most significant digit first, m entries used
};
template <int NL> __device__ void be_to_vec(const BField &f, const WN<NL> &a, BDigits &v)
{
    const u64 p = f.p[0];
    u32 limb[2 * NL];
    for (int i = 0; i < NL; i++) { limb[2 * i] = (u32)a.l[i]; limb[2 * i + 1] = (u32)(a.l[i] >> 32); }
    for (int i = f.m - 1; i >= 0; i--) {
        u64 rem = 0;
        for (int k = 2 * NL - 1; k >= 0; k--) {
            const u64 cur = (rem << 32) | limb[k];
            limb[k] = (u32)(cur / p);
            rem = cur % p;
        }
        v.d[i] = (u32)rem;
    }
}
template <int NL> __device__ WN<NL> be_from_vec(const BField &f, const BDigits &v)
{
    const u64 p = f.p[0];
    WN<NL> a = w_zero<NL>();
    for (int i = 0; i < f.m; i++) {
        u64 c = v.d[i];
        for (int k = 0; k < NL; k++) {
            const u128 x = (u128)a.l[k] * p + c;
            a.l[k] = (u64)x;
            c = (u64)(x >> 64);
        }
    }
    return a;
}
__device__ void be_mul_vec(const BField &f, const BDigits &a, const BDigits &b, BDigits &c)
{ // multiply_vector (_domains/_calculate.py:343-383): consume b from its lowest digit, keep a * x^it reduced
    const u64 p = f.p[0];
    const int m = f.m;
    BDigits av = a;
    for (int i = 0; i < m; i++) c.d[i] = 0;
    for (int it = 0; it < m; it++) {
        const u64 bl = b.d[m - 1 - it];
        if (bl)
            for (int i = 0; i < m; i++) c.d[i] = (u32)((c.d[i] + bl * av.d[i]) % p);
        const u64 qd = av.d[0];
        for (int i = 0; i + 1 < m; i++) av.d[i] = av.d[i + 1];
        av.d[m - 1] = 0;
        if (qd)
            for (int i = 0; i < m; i++) av.d[i] = (u32)((av.d[i] + (p - (qd * f.irr[i]) % p)) % p);
    }
}
template <int NL> __device__ WN<NL> be_mul(const BField &f, const WN<NL> &a, const WN<NL> &b)
{
    BDigits av, bv, cv;
    be_to_vec(f, a, av);
    be_to_vec(f, b, bv);
    be_mul_vec(f, av, bv, cv);
    return be_from_vec<NL>(f, cv);
}
template <int NL, int OP> // 0 add, 1 sub, 2 neg
__device__ WN<NL> be_lin(const BField &f, const WN<NL> &a, const WN<NL> &b)
{
    const u64 p = f.p[0];
    BDigits av, bv;
    be_to_vec(f, a, av);
    if (OP != 2) be_to_vec(f, b, bv);
    for (int i = 0; i < f.m; i++) {
        const u64 x = av.d[i], y = OP != 2 ? bv.d[i] : 0;
        av.d[i] = (u32)(OP == 0 ? (x + y) % p : OP == 1 ? (x + p - y) % p : (p - x) % p);
    }
    return be_from_vec<NL>(f, av);
}
template <int NL> __device__ WN<NL> be_pow(const BField &f, const WN<NL> &a, const WN<NL> &e)
{
    BDigits r, x, t;
    for (int i = 0; i < f.m; i++) r.d[i] = 0;
    r.d[f.m - 1] = 1;
    be_to_vec(f, a, x);
    const int top = w_top_bit(e);
    for (int i = 0; i <= top; i++) {
        if (w_bit(e, i)) { be_mul_vec(f, r, x, t); r = t; }
        if (i < top) { be_mul_vec(f, x, x, t); x = t; }
    }
    return be_from_vec<NL>(f, r);
}
__device__ u64 b_powmod64(u64 a, u64 e, u64 p)
{ // p < 2^32
    u64 r = 1;
    a %= p;
    while (e) {
        if (e & 1) r = r * a % p;
        a = a * a % p;
        e >>= 1;
    }
    return r;
}
// a != 0.  Itoh-Tsujii (reciprocal_itoh_tsujii, _domains/_calculate.py:447-489): a^-1 = (a^r)^-1 * a^(r-1), a^r in GF(p)
template <int NL> __device__ WN<NL> be_inv(const BField &f, const WN<NL> &a)
{
    const WN<NL> a_r1 = be_pow(f, a, w_from<NL>(f.itr));
    const WN<NL> a_r = be_mul(f, a_r1, a); // < p
    WN<NL> ninv = w_zero<NL>();
    ninv.l[0] = b_powmod64(a_r.l[0], f.p[0] - 2, f.p[0]);
    return be_mul(f, ninv, a_r1);
}

// ---------------------------------------------------------------- dispatch on the kind (wave-uniform)
template <int NL> __device__ WN<NL> bf_add(const BField &f, const WN<NL> &a, const WN<NL> &b)
{
    if (f.kind == BKIND_PRIME) return bp_add(f, a, b);
    if (f.kind == BKIND_BIN) { WN<NL> r; for (int i = 0; i < NL; i++) r.l[i] = a.l[i] ^ b.l[i]; return r; }
    return be_lin<NL, 0>(f, a, b);
}
template <int NL> __device__ WN<NL> bf_sub(const BField &f, const WN<NL> &a, const WN<NL> &b)
{
    if (f.kind == BKIND_PRIME) return bp_sub(f, a, b);
    if (f.kind == BKIND_BIN) { WN<NL> r; for (int i = 0; i < NL; i++) r.l[i] = a.l[i] ^ b.l[i]; return r; }
    return be_lin<NL, 1>(f, a, b);
}
template <int NL> __device__ WN<NL> bf_neg(const BField &f, const WN<NL> &a)
{
    if (f.kind == BKIND_BIN) return a;
    if (f.kind == BKIND_PRIME) return w_is_zero(a) ? a : w_sub(w_from<NL>(f.p), a);
    return be_lin<NL, 2>(f, a, a);
}
template <int NL> __device__ WN<NL> bf_mul(const BField &f, const WN<NL> &a, const WN<NL> &b)
{
    return f.kind == BKIND_PRIME ? bp_mul(f, a, b) : f.kind == BKIND_BIN ? bb_mul(f, a, b) : be_mul(f, a, b);
}
template <int NL> __device__ WN<NL> bf_pow(const BField &f, const WN<NL> &a, const WN<NL> &e)
{
    return f.kind == BKIND_PRIME ? bp_pow(f, a, e) : f.kind == BKIND_BIN ? bb_pow(f, a, e) : be_pow(f, a, e);
}
template <int NL> __device__ WN<NL> bf_inv(const BField &f, const WN<NL> &a)
{ // a != 0
    return f.kind == BKIND_PRIME ? bp_pow(f, a, w_from<NL>(f.einv)) : f.kind == BKIND_BIN ? bb_pow(f, a, w_from<NL>(f.einv)) : be_inv(f, a);
}

template <int NL> __device__ __forceinline__ WN<NL> bload(const u64 *p, i64 i) { WN<NL> r; for (int k = 0; k < NL; k++) r.l[k] = p[NL * i + k]; return r; }
template <int NL> __device__ __forceinline__ void bstore(u64 *p, i64 i, const WN<NL> &v) { for (int k = 0; k < NL; k++) p[NL * i + k] = v.l[k]; }

// the field descriptor lives in global memory (it is ~0.9 KiB: too large for the kernel-argument segment to be worth it)
template <int NL>
__global__ __launch_bounds__(64) void big_binary_kernel(const BField *__restrict__ fp, int op, const u64 *__restrict__ a, int sa, const u64 *__restrict__ b, int sb,
                                                        u64 *__restrict__ out, i64 n, int32_t *err)
{
    const BField &f = *fp;
    bool bad = false;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const WN<NL> x = bload<NL>(a, sa ? i : 0), y = bload<NL>(b, sb ? i : 0);
        WN<NL> r;
        switch (op) {
        case GFA_OP_ADD: r = bf_add(f, x, y); break;
        case GFA_OP_SUB: r = bf_sub(f, x, y); break;
        case GFA_OP_MUL: r = bf_mul(f, x, y); break;
        default:
            if (w_is_zero(y)) { bad = true; r = w_zero<NL>(); }
            else r = bf_mul(f, x, bf_inv(f, y));
        }
        bstore<NL>(out, i, r);
    }
    if (bad && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
}

template <int NL>
__global__ __launch_bounds__(64) void big_unary_kernel(const BField *__restrict__ fp, int op, const u64 *__restrict__ a, u64 *__restrict__ out, i64 n, int32_t *err)
{
    const BField &f = *fp;
    bool bad = false;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const WN<NL> x = bload<NL>(a, i);
        WN<NL> r;
        if (op == GFA_OP_NEG) r = bf_neg(f, x);
        else if (w_is_zero(x)) { bad = true; r = w_zero<NL>(); }
        else r = bf_inv(f, x);
        bstore<NL>(out, i, r);
    }
    if (bad && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
}

// exps: exponents the host has reduced into [0, q - 1) (NL limbs); sign: the sign of the ORIGINAL exponent, which decides the
// zero-base cases exactly as power_square_and_multiply does (_domains/_calculate.py:558-592)
template <int NL>
__global__ __launch_bounds__(64) void big_power_kernel(const BField *__restrict__ fp, const u64 *__restrict__ a, int sa, const u64 *__restrict__ exps, int se,
                                                       const int8_t *__restrict__ sign, u64 *__restrict__ out, i64 n, int32_t *err)
{
    const BField &f = *fp;
    bool bad = false;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const WN<NL> x = bload<NL>(a, sa ? i : 0), e = bload<NL>(exps, se ? i : 0);
        const int sg = sign[se ? i : 0];
        WN<NL> r;
        if (sg == 0) r = w_one<NL>();
        else if (w_is_zero(x)) { r = w_zero<NL>(); bad |= sg < 0; }
        else r = bf_pow(f, x, e);
        bstore<NL>(out, i, r);
    }
    if (bad && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
}

} // namespace

struct gfa_bfield {
    BField host;
    std::mutex mu;
    std::map<int, BField *> dev; // device id -> the descriptor's device copy
    const BField *on_device()
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return nullptr;
        std::lock_guard<std::mutex> lock(mu);
        auto it = dev.find(d);
        if (it != dev.end()) return it->second;
        BField *p = nullptr;
        if (hipMalloc((void **)&p, sizeof(BField)) != hipSuccess) return nullptr;
        if (hipMemcpy(p, &host, sizeof(BField), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p); return nullptr; }
        dev[d] = p;
        return p;
    }
};

#define GFA_BIG_DISPATCH(NLV, CALL)                  \
    switch (NLV) {                                   \
    case 4: { constexpr int NL = 4; CALL; break; }   \
    case 8: { constexpr int NL = 8; CALL; break; }   \
    default: { constexpr int NL = 16; CALL; break; } \
    }

extern "C" {

int gfa_bfield_create(int kind, uint32_t m, uint32_t nl, const uint64_t *params, gfa_bfield_t **out)
{
    // params (uint64 words): [0:16] p  [16] nprime  [17:33] r2  [33:49] p-2 | 2^m-2  [49:65] red  [65:81] (q-1)/(p-1)-1  [81:113] irr digits
    if (!params || !out || kind < BKIND_PRIME || kind > BKIND_EXT || m < 1 || (nl != 4 && nl != 8 && nl != 16) ||
        (kind == BKIND_BIN && m > 64 * nl) || (kind == BKIND_EXT && m > (uint32_t)BMAXD)) {
        set_error("gfa_bfield_create: bad arguments");
        return GFA_ERR_INVALID;
    }
    gfa_bfield *w = new gfa_bfield();
    BField &f = w->host;
    f.kind = kind; f.m = (int)m; f.nl = (int)nl;
    for (int i = 0; i < BMAXL; i++) {
        f.p[i] = params[i];
        f.r2[i] = params[17 + i];
        f.einv[i] = params[33 + i];
        f.red[i] = params[49 + i];
        f.itr[i] = params[65 + i];
    }
    f.nprime = params[16];
    for (int i = 0; i < BMAXD; i++) f.irr[i] = (u32)params[81 + i];
    *out = w;
    return GFA_OK;
}

void gfa_bfield_destroy(gfa_bfield_t *w)
{
    if (!w) return;
    for (auto &kv : w->dev) (void)hipFree(kv.second);
    delete w;
}

int gfa_big_binary(gfa_bfield_t *w, int op, const void *a, int64_t sa, const void *b, int64_t sb, void *out, int64_t n, gfa_stream_t stream,
                   int32_t *dev_err)
{
    if (!w || !a || !b || !out || n < 0 || op < GFA_OP_ADD || op > GFA_OP_DIV) { set_error("gfa_big_binary: bad arguments"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    const BField *fd = w->on_device();
    if (!fd) { set_error("gfa_big_binary: no device copy of the field"); return GFA_ERR_HIP; }
    const int grid = (int)std::min<i64>((n + 63) / 64, 256 * 16);
    GFA_BIG_DISPATCH(w->host.nl, hipLaunchKernelGGL((big_binary_kernel<NL>), dim3(grid), dim3(64), 0, (hipStream_t)stream, fd, op, (const u64 *)a, (int)sa,
                                                    (const u64 *)b, (int)sb, (u64 *)out, n, dev_err));
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int gfa_big_unary(gfa_bfield_t *w, int op, const void *a, void *out, int64_t n, gfa_stream_t stream, int32_t *dev_err)
{
    if (!w || !a || !out || n < 0 || (op != GFA_OP_NEG && op != GFA_OP_RECIP)) { set_error("gfa_big_unary: bad arguments"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    const BField *fd = w->on_device();
    if (!fd) { set_error("gfa_big_unary: no device copy of the field"); return GFA_ERR_HIP; }
    const int grid = (int)std::min<i64>((n + 63) / 64, 256 * 16);
    GFA_BIG_DISPATCH(w->host.nl, hipLaunchKernelGGL((big_unary_kernel<NL>), dim3(grid), dim3(64), 0, (hipStream_t)stream, fd, op, (const u64 *)a, (u64 *)out, n, dev_err));
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int gfa_big_power(gfa_bfield_t *w, const void *a, int64_t sa, const void *exps, int64_t se, const int8_t *sign, void *out, int64_t n,
                  gfa_stream_t stream, int32_t *dev_err)
{
    if (!w || !a || !exps || !sign || !out || n < 0) { set_error("gfa_big_power: bad arguments"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    const BField *fd = w->on_device();
    if (!fd) { set_error("gfa_big_power: no device copy of the field"); return GFA_ERR_HIP; }
    const int grid = (int)std::min<i64>((n + 63) / 64, 256 * 16);
    GFA_BIG_DISPATCH(w->host.nl, hipLaunchKernelGGL((big_power_kernel<NL>), dim3(grid), dim3(64), 0, (hipStream_t)stream, fd, (const u64 *)a, (int)sa,
                                                    (const u64 *)exps, (int)se, sign, (u64 *)out, n, dev_err));
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

} // extern "C"
