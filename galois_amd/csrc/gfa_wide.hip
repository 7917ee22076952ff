// gfa_wide.hip -- element-wise arithmetic in finite fields of order 2^64 <= q < 2^128.
//
// The reference runs these fields as dtype=object arrays of Python integers through its pure-Python ufuncs
// (src/galois/_fields/_ufunc.py:36-48 picks [np.object_], _domains/_meta.py:39-41 the "python-calculate" mode; the scalar
// kernels are the same _domains/_calculate.py:133-592 as for small fields).  Here an element is two 64-bit limbs
// (little endian, interleaved: element i at words 2i, 2i+1) and three arithmetic kinds cover what the reference covers:
//   WPRIME  GF(p), 2^64 <= p < 2^128: two-limb Montgomery products (CIOS), conditional-subtract add / sub, a^(p-2) inverse
//   WBIN    GF(2^m), 64 < m <= 127:   xor add; shift-and-xor product over 128 bits with a masked reduction (m fixed steps)
//   WEXT    GF(p^m), p < 2^32, q >= 2^64: base-p digit vectors (128-by-32 long division), schoolbook product with reduction
//           by the irreducible polynomial, Itoh-Tsujii inverse -- the reference's *_vector / itoh_tsujii formulas
// One element per lane, grid-stride.  These kernels exist for coverage and exactness (the three Sage folders GF(2^100),
// GF(36893488147419103183), GF(109987^4) of the reference's test-suite), not for the roofline: they are VALU-bound at a few
// hundred instructions per element, which is still orders of magnitude above the reference's ~1 us per element.
#include "gfa_internal.h"

using namespace gfa;

namespace {

struct W128 {
    u64 lo, hi;
};
typedef unsigned __int128 u128;

enum { WKIND_PRIME = 1, WKIND_BIN = 2, WKIND_EXT = 3 };

struct WField {
    int kind;
    int m;
    W128 p;       // WPRIME: the modulus.  WEXT: characteristic in p.lo
    u64 nprime;   // WPRIME: -p^-1 mod 2^64
    W128 r2;      // WPRIME: 2^256 mod p
    W128 pm2;     // WPRIME: p - 2
    W128 red;     // WBIN: irreducible polynomial without its x^m term
    W128 qm2;     // WBIN: 2^m - 2
    W128 itr;     // WEXT: (q - 1) / (p - 1) - 1 (the Itoh-Tsujii exponent r - 1)
    u32 irr[16];  // WEXT: irreducible polynomial minus x^m, digits of degree m-1..0
};

__device__ __forceinline__ bool w_is_zero(W128 a) { return (a.lo | a.hi) == 0; }
__device__ __forceinline__ bool w_ge(W128 a, W128 b) { return a.hi > b.hi || (a.hi == b.hi && a.lo >= b.lo); }
__device__ __forceinline__ W128 w_sub(W128 a, W128 b) { return W128{a.lo - b.lo, a.hi - b.hi - (a.lo < b.lo ? 1u : 0u)}; }
__device__ __forceinline__ W128 w_add_carry(W128 a, W128 b, u32 *c)
{
    W128 r;
    r.lo = a.lo + b.lo;
    const u64 c0 = r.lo < a.lo ? 1u : 0u;
    const u64 t = a.hi + b.hi;
    const u64 c1 = t < a.hi ? 1u : 0u;
    r.hi = t + c0;
    *c = (u32)(c1 | (r.hi < t ? 1u : 0u));
    return r;
}

// ---------------------------------------------------------------- WPRIME
__device__ __forceinline__ W128 wp_add(const WField &f, W128 a, W128 b)
{
    u32 c;
    W128 s = w_add_carry(a, b, &c);
    if (c || w_ge(s, f.p)) s = w_sub(s, f.p);
    return s;
}
__device__ __forceinline__ W128 wp_sub(const WField &f, W128 a, W128 b)
{
    if (w_ge(a, b)) return w_sub(a, b);
    u32 c;
    return w_sub(w_add_carry(a, f.p, &c), b); // a + p - b (the 2^128 carry cancels against the borrow)
}
// Montgomery product a * b * 2^-128 mod p (CIOS, two 64-bit limbs)
__device__ __forceinline__ W128 wp_mont(const WField &f, W128 a, W128 b)
{
    const u64 av[2] = {a.lo, a.hi}, bv[2] = {b.lo, b.hi}, pv[2] = {f.p.lo, f.p.hi};
    u64 t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        u128 x = (u128)av[0] * bv[i] + t0;
        t0 = (u64)x;
        x = (u128)av[1] * bv[i] + t1 + (u64)(x >> 64);
        t1 = (u64)x;
        u128 y = (u128)t2 + (u64)(x >> 64);
        t2 = (u64)y;
        const u64 t3 = (u64)(y >> 64);
        const u64 mq = t0 * f.nprime;
        x = (u128)mq * pv[0] + t0; // low word becomes zero
        x = (u128)mq * pv[1] + t1 + (u64)(x >> 64);
        t0 = (u64)x;
        y = (u128)t2 + (u64)(x >> 64);
        t1 = (u64)y;
        t2 = t3 + (u64)(y >> 64);
    }
    W128 r{t0, t1};
    if (t2 || w_ge(r, f.p)) r = w_sub(r, f.p);
    return r;
}
__device__ __forceinline__ W128 wp_mul(const WField &f, W128 a, W128 b) { return wp_mont(f, wp_mont(f, a, f.r2), b); }
__device__ W128 wp_pow(const WField &f, W128 a, W128 e)
{
    const W128 one_m = wp_mont(f, W128{1, 0}, f.r2); // 1 in Montgomery form
    W128 am = wp_mont(f, a, f.r2), r = one_m;
    for (int i = 0; i < 128; i++) {
        const u64 bit = i < 64 ? (e.lo >> i) & 1 : (e.hi >> (i - 64)) & 1;
        if (bit) r = wp_mont(f, r, am);
        am = wp_mont(f, am, am);
    }
    return wp_mont(f, r, W128{1, 0});
}

// ---------------------------------------------------------------- WBIN
__device__ W128 wb_mul(const WField &f, W128 a, W128 b)
{
    const int m = f.m;
    W128 c{0, 0};
    for (int i = 0; i < m; i++) {
        const u64 bit = i < 64 ? (b.lo >> i) & 1 : (b.hi >> (i - 64)) & 1;
        const u64 bm = (u64)0 - bit;
        c.lo ^= a.lo & bm;
        c.hi ^= a.hi & bm;
        const u64 top = m > 64 ? (a.hi >> (m - 65)) & 1 : (a.lo >> (m - 1)) & 1; // bit m-1
        const u64 tm = (u64)0 - top;
        a.hi = (a.hi << 1) | (a.lo >> 63);
        a.lo <<= 1;
        // the bit shifted into position m is cleared by masking; the reduction adds the low part of the polynomial
        if (m < 128) {
            if (m > 64) a.hi &= ((u64)1 << (m - 64)) - 1;
            else { a.hi = 0; if (m < 64) a.lo &= ((u64)1 << m) - 1; }
        }
        a.lo ^= f.red.lo & tm;
        a.hi ^= f.red.hi & tm;
    }
    return c;
}
__device__ W128 wb_pow(const WField &f, W128 a, W128 e)
{
    W128 r{1, 0};
    for (int i = 0; i < 128; i++) {
        const u64 bit = i < 64 ? (e.lo >> i) & 1 : (e.hi >> (i - 64)) & 1;
        if (bit) r = wb_mul(f, r, a);
        if ((i < 64 ? (e.lo >> i) >> 1 : (e.hi >> (i - 64)) >> 1) == 0 && (i >= 64 || e.hi == 0)) break; // no higher bits left
        a = wb_mul(f, a, a);
    }
    return r;
}

// ---------------------------------------------------------------- WEXT
struct Digits {
    u32 d[16]; // most significant digit first, m entries used
};
__device__ void we_to_vec(const WField &f, W128 a, Digits &v)
{
    const u64 p = f.p.lo;
    u32 limb[4] = {(u32)a.lo, (u32)(a.lo >> 32), (u32)a.hi, (u32)(a.hi >> 32)};
    for (int i = f.m - 1; i >= 0; i--) {
        u64 rem = 0;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const u64 cur = (rem << 32) | limb[k];
            limb[k] = (u32)(cur / p);
            rem = cur % p;
        }
        v.d[i] = (u32)rem;
    }
}
__device__ W128 we_from_vec(const WField &f, const Digits &v)
{
    const u64 p = f.p.lo;
    W128 a{0, 0};
    for (int i = 0; i < f.m; i++) {
        const u128 lo = (u128)a.lo * p + v.d[i];
        a.hi = a.hi * p + (u64)(lo >> 64);
        a.lo = (u64)lo;
    }
    return a;
}
__device__ void we_mul_vec(const WField &f, const Digits &a, const Digits &b, Digits &c)
{ // multiply_vector (_domains/_calculate.py:343-383): consume b from its lowest digit, keep a * x^it reduced
    const u64 p = f.p.lo;
    const int m = f.m;
    Digits av = a;
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
__device__ W128 we_mul(const WField &f, W128 a, W128 b)
{
    Digits av, bv, cv;
    we_to_vec(f, a, av);
    we_to_vec(f, b, bv);
    we_mul_vec(f, av, bv, cv);
    return we_from_vec(f, cv);
}
template <int OP> // 0 add, 1 sub, 2 neg
__device__ W128 we_lin(const WField &f, W128 a, W128 b)
{
    const u64 p = f.p.lo;
    Digits av, bv;
    we_to_vec(f, a, av);
    if (OP != 2) we_to_vec(f, b, bv);
    for (int i = 0; i < f.m; i++) {
        const u64 x = av.d[i], y = OP != 2 ? bv.d[i] : 0;
        av.d[i] = (u32)(OP == 0 ? (x + y) % p : OP == 1 ? (x + p - y) % p : (p - x) % p);
    }
    return we_from_vec(f, av);
}
__device__ W128 we_pow(const WField &f, W128 a, W128 e)
{
    Digits r, x, t;
    for (int i = 0; i < f.m; i++) r.d[i] = 0;
    r.d[f.m - 1] = 1;
    we_to_vec(f, a, x);
    for (int i = 0; i < 128; i++) {
        const u64 bit = i < 64 ? (e.lo >> i) & 1 : (e.hi >> (i - 64)) & 1;
        if (bit) { we_mul_vec(f, r, x, t); r = t; }
        if ((i < 64 ? (e.lo >> i) >> 1 : (e.hi >> (i - 64)) >> 1) == 0 && (i >= 64 || e.hi == 0)) break;
        we_mul_vec(f, x, x, t);
        x = t;
    }
    return we_from_vec(f, r);
}
__device__ u64 powmod64(u64 a, u64 e, u64 p)
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
__device__ W128 we_inv(const WField &f, W128 a)
{
    const W128 a_r1 = we_pow(f, a, f.itr);
    const W128 a_r = we_mul(f, a_r1, a); // < p
    const u64 ninv = powmod64(a_r.lo, f.p.lo - 2, f.p.lo);
    return we_mul(f, W128{ninv, 0}, a_r1);
}

// ---------------------------------------------------------------- dispatch on the kind (wave-uniform)
__device__ W128 wf_add(const WField &f, W128 a, W128 b)
{
    return f.kind == WKIND_PRIME ? wp_add(f, a, b) : f.kind == WKIND_BIN ? W128{a.lo ^ b.lo, a.hi ^ b.hi} : we_lin<0>(f, a, b);
}
__device__ W128 wf_sub(const WField &f, W128 a, W128 b)
{
    return f.kind == WKIND_PRIME ? wp_sub(f, a, b) : f.kind == WKIND_BIN ? W128{a.lo ^ b.lo, a.hi ^ b.hi} : we_lin<1>(f, a, b);
}
__device__ W128 wf_neg(const WField &f, W128 a)
{
    if (f.kind == WKIND_BIN) return a;
    if (f.kind == WKIND_PRIME) return w_is_zero(a) ? a : w_sub(f.p, a);
    return we_lin<2>(f, a, a);
}
__device__ W128 wf_mul(const WField &f, W128 a, W128 b)
{
    return f.kind == WKIND_PRIME ? wp_mul(f, a, b) : f.kind == WKIND_BIN ? wb_mul(f, a, b) : we_mul(f, a, b);
}
__device__ W128 wf_inv(const WField &f, W128 a)
{ // a != 0
    return f.kind == WKIND_PRIME ? wp_pow(f, a, f.pm2) : f.kind == WKIND_BIN ? wb_pow(f, a, f.qm2) : we_inv(f, a);
}
__device__ W128 wf_pow(const WField &f, W128 a, W128 e)
{
    return f.kind == WKIND_PRIME ? wp_pow(f, a, e) : f.kind == WKIND_BIN ? wb_pow(f, a, e) : we_pow(f, a, e);
}

__device__ __forceinline__ W128 wload(const u64 *p, i64 i) { return W128{p[2 * i], p[2 * i + 1]}; }
__device__ __forceinline__ void wstore(u64 *p, i64 i, W128 v) { p[2 * i] = v.lo; p[2 * i + 1] = v.hi; }

__global__ __launch_bounds__(256) void wide_binary_kernel(WField f, int op, const u64 *__restrict__ a, int sa, const u64 *__restrict__ b,
                                                          int sb, u64 *__restrict__ out, i64 n, int32_t *err)
{
    bool bad = false;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const W128 x = wload(a, sa ? i : 0), y = wload(b, sb ? i : 0);
        W128 r;
        switch (op) {
        case GFA_OP_ADD: r = wf_add(f, x, y); break;
        case GFA_OP_SUB: r = wf_sub(f, x, y); break;
        case GFA_OP_MUL: r = wf_mul(f, x, y); break;
        default:
            if (w_is_zero(y)) { bad = true; r = W128{0, 0}; }
            else r = wf_mul(f, x, wf_inv(f, y));
        }
        wstore(out, i, r);
    }
    if (bad && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
}

__global__ __launch_bounds__(256) void wide_unary_kernel(WField f, int op, const u64 *__restrict__ a, u64 *__restrict__ out, i64 n, int32_t *err)
{
    bool bad = false;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const W128 x = wload(a, i);
        W128 r;
        if (op == GFA_OP_NEG) r = wf_neg(f, x);
        else if (w_is_zero(x)) { bad = true; r = W128{0, 0}; }
        else r = wf_inv(f, x);
        wstore(out, i, r);
    }
    if (bad && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
}

// exps: exponent already reduced into [0, q - 1) by the host (two limbs); sign: the sign of the ORIGINAL exponent, which
// decides the zero-base cases exactly as power_square_and_multiply does (_domains/_calculate.py:558-592)
__global__ __launch_bounds__(256) void wide_power_kernel(WField f, const u64 *__restrict__ a, int sa, const u64 *__restrict__ exps, int se,
                                                         const int8_t *__restrict__ sign, u64 *__restrict__ out, i64 n, int32_t *err)
{
    bool bad = false;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const W128 x = wload(a, sa ? i : 0), e = wload(exps, se ? i : 0);
        const int sg = sign[se ? i : 0];
        W128 r;
        if (sg == 0) r = W128{1, 0};
        else if (w_is_zero(x)) { r = W128{0, 0}; bad |= sg < 0; }
        else r = wf_pow(f, x, e);
        wstore(out, i, r);
    }
    if (bad && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
}

// ---------------------------------------------------------------- reductions, convolution, matrix product
// The reference runs ufunc.reduce / accumulate, np.convolve and @ on these fields as object-dtype loops over the same scalar
// kernels (_fields/_ufunc.py:36-48, _domains/_function.py:141-167, _domains/_linalg.py:286-308).  Coverage paths: exact, simple.
__device__ __forceinline__ W128 wf_op(const WField &f, int op, W128 x, W128 y, bool &bad)
{
    switch (op) {
    case GFA_OP_ADD: return wf_add(f, x, y);
    case GFA_OP_SUB: return wf_sub(f, x, y);
    case GFA_OP_MUL: return wf_mul(f, x, y);
    default:
        if (w_is_zero(y)) { bad = true; return W128{0, 0}; }
        return wf_mul(f, x, wf_inv(f, y));
    }
}

// One workgroup per row.  reduce: a[0] op (a[1] dual a[2] dual ...) with dual = + for -, * for / -- the left fold the
// reference computes, regrouped so that the tail is an associative, commutative fold the 256 threads can share.
// accumulate (every prefix) is inherently sequential per row: thread 0 walks the row.
__global__ __launch_bounds__(256) void wide_reduce_kernel(WField f, int op, const u64 *__restrict__ a, u64 *__restrict__ out, i64 n_inner,
                                                          int accumulate, int32_t *err)
{
    __shared__ W128 part[256];
    const i64 row = blockIdx.x;
    const u64 *ar = a + 2 * row * n_inner;
    bool bad = false;
    if (accumulate) {
        if (threadIdx.x == 0) {
            W128 acc = wload(ar, 0);
            wstore(out + 2 * row * n_inner, 0, acc);
            for (i64 i = 1; i < n_inner; i++) {
                acc = wf_op(f, op, acc, wload(ar, i), bad);
                wstore(out + 2 * row * n_inner, i, acc);
            }
        }
    } else {
        const int dual = (op == GFA_OP_ADD || op == GFA_OP_SUB) ? GFA_OP_ADD : GFA_OP_MUL;
        W128 acc = dual == GFA_OP_ADD ? W128{0, 0} : W128{1, 0};
        for (i64 i = 1 + threadIdx.x; i < n_inner; i += 256) acc = wf_op(f, dual, acc, wload(ar, i), bad);
        part[threadIdx.x] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) part[threadIdx.x] = wf_op(f, dual, part[threadIdx.x], part[threadIdx.x + s], bad);
            __syncthreads();
        }
        if (threadIdx.x == 0) wstore(out, row, n_inner > 1 ? wf_op(f, op, wload(ar, 0), part[0], bad) : wload(ar, 0));
    }
    if (bad && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
}

__global__ __launch_bounds__(256) void wide_convolve_kernel(WField f, const u64 *__restrict__ a, i64 na, const u64 *__restrict__ b, i64 nb,
                                                            u64 *__restrict__ out)
{ // out[k] = sum_i a[i] * b[k - i]
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= na + nb - 1) return;
    const i64 lo = k - (nb - 1) > 0 ? k - (nb - 1) : 0, hi = k < na - 1 ? k : na - 1;
    W128 acc{0, 0};
    for (i64 i = lo; i <= hi; i++) acc = wf_add(f, acc, wf_mul(f, wload(a, i), wload(b, k - i)));
    wstore(out, k, acc);
}

__global__ __launch_bounds__(256) void wide_matmul_kernel(WField f, const u64 *__restrict__ a, const u64 *__restrict__ b, u64 *__restrict__ out,
                                                          i64 batch, i64 M, i64 K, i64 N, i64 a_bstride, i64 b_bstride)
{ // out[t][i][j] = sum_k a[t][i][k] * b[t][k][j]; batch strides in elements (0: one matrix for every batch item)
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * M * N) return;
    const i64 t = idx / (M * N), i = (idx / N) % M, j = idx % N;
    const u64 *ap = a + 2 * (t * a_bstride + i * K), *bp = b + 2 * (t * b_bstride + j);
    W128 acc{0, 0};
    for (i64 k = 0; k < K; k++) acc = wf_add(f, acc, wf_mul(f, wload(ap, k), wload(bp, k * N)));
    wstore(out, idx, acc);
}

// ---------------------------------------------------------------- elimination and Horner (r05)
// The reference runs row_reduce / lu / plu / det / inv / solve and polynomial evaluation on these fields as object-dtype loops over
// the same scalar kernels (_domains/_linalg.py:315-424, 447-548; _polys/_dense.py:404-440).  One workgroup per matrix, the pivot
// rule, the swap order and the L / P conventions of gfa_linalg.hip's kernels (which follow the reference line by line); matrices are
// small (the Sage fixtures: up to 6 x 6) and an inversion is an exponentiation, so nothing here is tuned.
constexpr int WGJ_THREADS = 256;

__device__ __forceinline__ bool w_nonzero(const u64 *A, i64 i) { return (A[2 * i] | A[2 * i + 1]) != 0; }

// row_reduce_jit (_linalg.py:315-351).  A: (batch, m, n) in place; rank_out[b] = number of pivots found in the first ncols columns.
__global__ __launch_bounds__(WGJ_THREADS) void wide_row_reduce_kernel(WField f, u64 *__restrict__ Aall, int m, int n, int ncols, i64 *__restrict__ rank_out)
{
    __shared__ int piv_row;
    __shared__ W128 piv_inv;
    u64 *A = Aall + 2 * (i64)blockIdx.x * m * n;
    int p = 0;
    for (int j = 0; j < ncols && p < m; j++) {
        if (threadIdx.x == 0) piv_row = m;
        __syncthreads();
        for (int i = p + threadIdx.x; i < m; i += WGJ_THREADS)
            if (w_nonzero(A, (i64)i * n + j)) { atomicMin(&piv_row, i); break; }
        __syncthreads();
        const int pr = piv_row;
        if (pr == m) { __syncthreads(); continue; }
        if (threadIdx.x == 0) piv_inv = wf_inv(f, wload(A, (i64)pr * n + j));
        __syncthreads();
        const W128 inv = piv_inv;
        for (int c = threadIdx.x; c < n; c += WGJ_THREADS) { // swap rows p and pr, the pivot row scaled to a leading 1
            const W128 top = wload(A, (i64)p * n + c);
            const W128 piv = wf_mul(f, wload(A, (i64)pr * n + c), inv);
            if (pr != p) wstore(A, (i64)pr * n + c, top);
            wstore(A, (i64)p * n + c, piv);
        }
        __syncthreads();
        // A[i, :] -= A[i, j] * A[p, :] for every other row; column j itself last (it holds the factors)
        for (i64 e = threadIdx.x; e < (i64)m * n; e += WGJ_THREADS) {
            const int i = (int)(e / n), c = (int)(e % n);
            if (i == p || c == j) continue;
            const W128 fct = wload(A, (i64)i * n + j);
            if (w_is_zero(fct)) continue;
            wstore(A, e, wf_sub(f, wload(A, e), wf_mul(f, fct, wload(A, (i64)p * n + c))));
        }
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += WGJ_THREADS)
            if (i != p) wstore(A, (i64)i * n + j, W128{0, 0});
        __syncthreads();
        p++;
    }
    if (threadIdx.x == 0) rank_out[blockIdx.x] = p;
}

// lu_decompose_jit / plu_decompose_jit (_linalg.py:354-424), det_jit (:447-477): conventions of plu_kernel in gfa_linalg.hip
template <bool PIVOT>
__global__ __launch_bounds__(WGJ_THREADS) void wide_plu_kernel(WField f, u64 *__restrict__ Aall, u64 *__restrict__ Lall, u64 *__restrict__ Pall, int m, int n,
                                                               i64 *__restrict__ nperm_out, u64 *__restrict__ det_out, int32_t *__restrict__ err)
{
    __shared__ int piv_row;
    __shared__ W128 piv_inv;
    u64 *A = Aall + 2 * (i64)blockIdx.x * m * n;
    u64 *Lm = Lall ? Lall + 2 * (i64)blockIdx.x * m * m : nullptr;
    u64 *Pm = Pall ? Pall + 2 * (i64)blockIdx.x * m * m : nullptr;
    for (i64 e = threadIdx.x; e < (i64)m * m; e += WGJ_THREADS) { // L = 0 (PLU) or I (LU); P = I
        const int r = (int)(e / m), c = (int)(e % m);
        if (Lm) wstore(Lm, e, W128{(u64)((!PIVOT && r == c) ? 1 : 0), 0});
        if (Pm) wstore(Pm, e, W128{(u64)(r == c ? 1 : 0), 0});
    }
    __syncthreads();
    int nperm = 0;
    bool failed = false;
    const int steps = PIVOT ? (m < n ? m : n) : m - 1;
    for (int i = 0; i < steps; i++) {
        if (threadIdx.x == 0) piv_row = m;
        __syncthreads();
        const bool diag_zero = !w_nonzero(A, (i64)i * n + i);
        if (diag_zero) {
            for (int r = i + threadIdx.x; r < m; r += WGJ_THREADS)
                if (w_nonzero(A, (i64)r * n + i)) { atomicMin(&piv_row, r); break; }
        }
        __syncthreads();
        if (diag_zero) {
            const int pr = piv_row;
            if (pr == m) {
                if (Lm && threadIdx.x == 0) wstore(Lm, (i64)i * m + i, W128{1, 0});
                __syncthreads();
                continue;
            }
            if (!PIVOT) { failed = true; break; }
            for (int c = threadIdx.x; c < n; c += WGJ_THREADS) {
                const W128 t = wload(A, (i64)i * n + c);
                wstore(A, (i64)i * n + c, wload(A, (i64)pr * n + c));
                wstore(A, (i64)pr * n + c, t);
            }
            for (int c = threadIdx.x; c < m; c += WGJ_THREADS) {
                if (Pm) { const W128 t = wload(Pm, (i64)i * m + c); wstore(Pm, (i64)i * m + c, wload(Pm, (i64)pr * m + c)); wstore(Pm, (i64)pr * m + c, t); }
                if (Lm) { const W128 t = wload(Lm, (i64)i * m + c); wstore(Lm, (i64)i * m + c, wload(Lm, (i64)pr * m + c)); wstore(Lm, (i64)pr * m + c, t); }
            }
            nperm++;
            __syncthreads();
        }
        if (threadIdx.x == 0) piv_inv = wf_inv(f, wload(A, (i64)i * n + i));
        __syncthreads();
        const W128 inv = piv_inv;
        // l_r = A[r, i] / A[i, i]; rows below the pivot: A[r, c] -= l_r * A[i, c] for c > i, then A[r, i] = 0
        for (i64 e = threadIdx.x; e < (i64)(m - i - 1) * n; e += WGJ_THREADS) {
            const int r = i + 1 + (int)(e / n), c = (int)(e % n);
            if (c <= i) continue;
            const W128 lead = wload(A, (i64)r * n + i);
            if (w_is_zero(lead)) continue;
            const W128 l = wf_mul(f, lead, inv);
            wstore(A, (i64)r * n + c, wf_sub(f, wload(A, (i64)r * n + c), wf_mul(f, l, wload(A, (i64)i * n + c))));
        }
        __syncthreads();
        for (int r = i + 1 + threadIdx.x; r < m; r += WGJ_THREADS) {
            const W128 lead = wload(A, (i64)r * n + i);
            if (Lm) wstore(Lm, (i64)r * m + i, w_is_zero(lead) ? lead : wf_mul(f, lead, inv));
            wstore(A, (i64)r * n + i, W128{0, 0});
        }
        if (Lm && threadIdx.x == 0) wstore(Lm, (i64)i * m + i, W128{1, 0});
        __syncthreads();
    }
    if (failed) {
        if (threadIdx.x == 0 && err) atomicOr((int *)err, GFA_DEVERR_NO_LU);
        return;
    }
    if (threadIdx.x == 0) {
        if (Lm && PIVOT) wstore(Lm, (i64)(m - 1) * m + (m - 1), W128{1, 0}); // "set the final diagonal to 1" (_linalg.py:419)
        if (nperm_out) nperm_out[blockIdx.x] = nperm;
        if (det_out) {
            W128 d{1, 0};
            const int k = m < n ? m : n;
            for (int i = 0; i < k; i++) d = wf_mul(f, d, wload(A, (i64)i * n + i));
            if (nperm & 1) d = wf_neg(f, d);
            wstore(det_out, blockIdx.x, d);
        }
    }
}

// evaluate_elementwise_jit (_polys/_dense.py:404-423): Horner, coefficients in descending degree, one x per lane
__global__ __launch_bounds__(256) void wide_poly_eval_kernel(WField f, const u64 *__restrict__ coeffs, i64 ncoef, const u64 *__restrict__ x, u64 *__restrict__ out, i64 n)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const W128 xv = wload(x, i);
        W128 acc = wload(coeffs, 0);
        for (i64 k = 1; k < ncoef; k++) acc = wf_add(f, wload(coeffs, k), wf_mul(f, acc, xv));
        wstore(out, i, acc);
    }
}

} // namespace

struct gfa_wfield {
    WField f;
};

extern "C" {

int gfa_wfield_create(int kind, uint32_t m, const uint64_t *params, gfa_wfield_t **out)
{
    // params (uint64 words): [0:2] p  [2] nprime  [3:5] r2  [5:7] p-2 | 2^m-2  [7:9] red  [9:11] (q-1)/(p-1)-1  [11:27] irr digits
    if (!params || !out || kind < WKIND_PRIME || kind > WKIND_EXT || m < 1 || m > 128 || (m == 128 && kind != WKIND_BIN) || (kind == WKIND_EXT && m > 16)) {
        set_error("gfa_wfield_create: bad arguments");
        return GFA_ERR_INVALID;
    }
    gfa_wfield *w = new gfa_wfield();
    WField &f = w->f;
    f.kind = kind; f.m = (int)m;
    f.p = W128{params[0], params[1]};
    f.nprime = params[2];
    f.r2 = W128{params[3], params[4]};
    f.pm2 = W128{params[5], params[6]};
    f.qm2 = f.pm2;
    f.red = W128{params[7], params[8]};
    f.itr = W128{params[9], params[10]};
    for (int i = 0; i < 16; i++) f.irr[i] = (u32)params[11 + i];
    *out = w;
    return GFA_OK;
}

void gfa_wfield_destroy(gfa_wfield_t *w) { delete w; }

int gfa_wide_binary(gfa_wfield_t *w, int op, const void *a, int64_t sa, const void *b, int64_t sb, void *out, int64_t n, gfa_stream_t stream,
                    int32_t *dev_err)
{
    if (!w || !a || !b || !out || n < 0 || op < GFA_OP_ADD || op > GFA_OP_DIV) { set_error("gfa_wide_binary: bad arguments"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    const int grid = (int)std::min<i64>((n + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(wide_binary_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w->f, op, (const u64 *)a, (int)sa, (const u64 *)b, (int)sb,
                       (u64 *)out, n, dev_err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int gfa_wide_unary(gfa_wfield_t *w, int op, const void *a, void *out, int64_t n, gfa_stream_t stream, int32_t *dev_err)
{
    if (!w || !a || !out || n < 0 || (op != GFA_OP_NEG && op != GFA_OP_RECIP)) { set_error("gfa_wide_unary: bad arguments"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    const int grid = (int)std::min<i64>((n + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(wide_unary_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w->f, op, (const u64 *)a, (u64 *)out, n, dev_err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int gfa_wide_power(gfa_wfield_t *w, const void *a, int64_t sa, const void *exps, int64_t se, const int8_t *sign, void *out, int64_t n,
                   gfa_stream_t stream, int32_t *dev_err)
{
    if (!w || !a || !exps || !sign || !out || n < 0) { set_error("gfa_wide_power: bad arguments"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    const int grid = (int)std::min<i64>((n + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(wide_power_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w->f, (const u64 *)a, (int)sa, (const u64 *)exps, (int)se, sign,
                       (u64 *)out, n, dev_err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int gfa_wide_reduce(gfa_wfield_t *w, int op, const void *a, void *out, int64_t n_outer, int64_t n_inner, int accumulate, gfa_stream_t stream,
                    int32_t *dev_err)
{
    if (!w || !a || !out || n_outer < 0 || n_inner < 1 || op < GFA_OP_ADD || op > GFA_OP_DIV || n_outer > 0x7fffffff) {
        set_error("gfa_wide_reduce: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (n_outer == 0) return GFA_OK;
    hipLaunchKernelGGL(wide_reduce_kernel, dim3((unsigned)n_outer), dim3(256), 0, (hipStream_t)stream, w->f, op, (const u64 *)a, (u64 *)out, n_inner,
                       accumulate, dev_err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int gfa_wide_convolve(gfa_wfield_t *w, const void *a, int64_t na, const void *b, int64_t nb, void *out, gfa_stream_t stream)
{
    if (!w || !a || !b || !out || na < 1 || nb < 1) { set_error("gfa_wide_convolve: bad arguments"); return GFA_ERR_INVALID; }
    const i64 n = na + nb - 1;
    hipLaunchKernelGGL(wide_convolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w->f, (const u64 *)a, na,
                       (const u64 *)b, nb, (u64 *)out);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int gfa_wide_matmul(gfa_wfield_t *w, const void *a, const void *b, void *out, int64_t batch, int64_t M, int64_t K, int64_t N, int64_t a_bstride,
                    int64_t b_bstride, gfa_stream_t stream)
{
    if (!w || !a || !b || !out || batch < 0 || M < 0 || K < 0 || N < 0) { set_error("gfa_wide_matmul: bad arguments"); return GFA_ERR_INVALID; }
    const i64 n = batch * M * N;
    if (n == 0) return GFA_OK;
    hipLaunchKernelGGL(wide_matmul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w->f, (const u64 *)a, (const u64 *)b,
                       (u64 *)out, batch, M, K, N, a_bstride, b_bstride);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int gfa_wide_row_reduce(gfa_wfield_t *w, void *a, int64_t batch, int64_t m, int64_t n, int64_t ncols, int64_t *rank_out, gfa_stream_t stream)
{
    if (!w || !a || !rank_out || batch < 0 || m < 1 || n < 1 || ncols < 0 || ncols > n || m > 0x7fffffff / 2 || n > 0x7fffffff / 2 || batch > 0x7fffffff) {
        set_error("gfa_wide_row_reduce: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (batch == 0) return GFA_OK;
    hipLaunchKernelGGL(wide_row_reduce_kernel, dim3((unsigned)batch), dim3(WGJ_THREADS), 0, (hipStream_t)stream, w->f, (u64 *)a, (int)m, (int)n, (int)ncols,
                       (i64 *)rank_out);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int gfa_wide_plu_decompose(gfa_wfield_t *w, void *a, void *l_out, void *p_out, int64_t batch, int64_t m, int64_t n, int pivoting, int64_t *nperm_out,
                           void *det_out, gfa_stream_t stream, int32_t *dev_err)
{
    if (!w || !a || batch < 0 || m < 1 || n < 1 || m > 0x7fffffff / 2 || n > 0x7fffffff / 2 || batch > 0x7fffffff) {
        set_error("gfa_wide_plu_decompose: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (batch == 0) return GFA_OK;
    if (pivoting)
        hipLaunchKernelGGL((wide_plu_kernel<true>), dim3((unsigned)batch), dim3(WGJ_THREADS), 0, (hipStream_t)stream, w->f, (u64 *)a, (u64 *)l_out, (u64 *)p_out,
                           (int)m, (int)n, (i64 *)nperm_out, (u64 *)det_out, dev_err);
    else
        hipLaunchKernelGGL((wide_plu_kernel<false>), dim3((unsigned)batch), dim3(WGJ_THREADS), 0, (hipStream_t)stream, w->f, (u64 *)a, (u64 *)l_out, (u64 *)p_out,
                           (int)m, (int)n, (i64 *)nperm_out, (u64 *)det_out, dev_err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int gfa_wide_poly_evaluate(gfa_wfield_t *w, const void *coeffs, int64_t ncoef, const void *x, void *out, int64_t n, gfa_stream_t stream)
{
    if (!w || !coeffs || !x || !out || ncoef < 1 || n < 0) { set_error("gfa_wide_poly_evaluate: bad arguments"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    const int grid = (int)std::min<i64>((n + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(wide_poly_eval_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w->f, (const u64 *)coeffs, ncoef, (const u64 *)x, (u64 *)out, n);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

} // extern "C"
