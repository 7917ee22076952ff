// gfa_elementwise.hip -- element-wise field ufuncs as HBM-streaming integer kernels for gfx950.
//
// Replaces the numba.vectorize'd scalar loops behind UFunc.ufunc (reference: src/galois/_domains/_ufunc.py:97-144
// with the scalar bodies of _lookup.py:31-270 and _calculate.py:133-592).
//
// Two kernel families:
//   * tab8_*  : fields with order <= 256 stored as uint8 (GF(2^8) is the headline case).  A full 64 KiB
//               binary-operation table (index (a<<8)|b) is staged into LDS once per 1024-thread workgroup, so each
//               element costs ONE LDS byte gather instead of the three dependent gathers of EXP[LOG a + LOG b].
//               Unary tables (reciprocal / negative) are replicated 32x so that lane l only ever touches LDS
//               bank l%32: conflict-free by construction.  Loads/stores are 16 B per lane, fully coalesced.
//   * ew_*    : every other field / dtype, templated on the arithmetic (gfa_arith.h) and the storage width.
#include <algorithm>
#include <cstdlib>

#include "gfa_internal.h"

using namespace gfa;

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct alignas(16) Vec16 {
    static constexpr int N = 16 / sizeof(T);
    T v[N];
};

__device__ __forceinline__ void flag_error(int32_t *err, bool bad)
{
    // one atomic per wave at most
    if (__any(bad)) {
        if ((threadIdx.x & 63) == 0 && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
    }
}

// ------------------------------------------------------------------------------------------------
// Generic element-wise kernels
// ------------------------------------------------------------------------------------------------
template <class F, int OP>
__device__ __forceinline__ typename F::elem apply_binary(const FieldDev &fd, typename F::elem x, typename F::elem y, bool &bad)
{
    if constexpr (OP == GFA_OP_ADD) return F::add(fd, x, y);
    else if constexpr (OP == GFA_OP_SUB) return F::sub(fd, x, y);
    else if constexpr (OP == GFA_OP_MUL) return F::mul(fd, x, y);
    else { // DIV: reciprocal of the divisor then multiply (divide_ufunc.__call__, _ufunc.py:433-437)
        if (y == 0) { bad = true; return 0; }
        if (x == 0) return 0;
        if constexpr (std::is_same<F, Lut>::value) return Lut::div_nz(fd, x, y);
        else return F::mul(fd, x, F::inv(fd, y));
    }
}

template <class F, int OP>
__device__ __forceinline__ typename F::elem apply_unary(const FieldDev &fd, typename F::elem x, bool &bad)
{
    if constexpr (OP == GFA_OP_NEG) return F::neg(fd, x);
    else { // RECIP
        if (x == 0) { bad = true; return 0; }
        return F::inv(fd, x);
    }
}


// Simultaneous inversion (Montgomery's trick) of the V elements one lane holds: one field inversion + 3(V-1)
// multiplications instead of V inversions.  Used where an inversion is an exponentiation (prime fields: a^(p-2)).
// Zero entries are flagged and yield 0; they are replaced by 1 inside the product chain.
template <class F>
struct BatchInv {
    static constexpr bool value = std::is_same<F, Prime32>::value || std::is_same<F, Prime64>::value ||
                                  std::is_same<F, Goldilocks>::value;
};
// GF(p^M) with the degree fixed: an inversion is ~27 field products (Itoh-Tsujii down to GF(p), then a prime-field power),
// so sharing one among 16 elements pays as well
template <int M>
struct BatchInv<ExtM<M>> {
    static constexpr bool value = true;
};
// GF(2^m): the Euclidean inversion is a data-dependent loop of up to 2m steps that diverges inside a wavefront (~900 issue slots
// at m = 32); three products per element (~120 each with the integer-multiply carry-less product) plus one inversion per 16 is less
template <>
struct BatchInv<Bin> {
    static constexpr bool value = true;
};

// elements one lane inverts together.  32 for 32-bit primes was measured (profiles/r04_ew_prime_recip.txt): the inversion's share halves
// but the kernel drops to three waves per SIMD -- GF(65537) reciprocal 0.57 against 0.64 with 16
template <class F, typename T>
struct BatchN {
    static constexpr int value = 16;
};

// 1 for a non-zero element, 0 for zero -- by arithmetic: a select on a compare is a v_cmp + a v_cndmask that reads the condition
// mask, the slowest vector instruction there is on this part (profiles/r03_valu_issue_rates.txt: 11 lane-ops/clk/CU against 57-110)
__device__ __forceinline__ u32 nz_flag(u32 x) { return x < 1u ? x : 1u; } // v_min_u32

template <class F, int V>
__device__ __forceinline__ void batch_inverse(const FieldDev &fd, typename F::elem (&x)[V], bool &bad)
{
    typedef typename F::elem E;
    // (64-bit elements keep compare-and-select for the zeros: measured, the arithmetic flags of batch_inverse_m31 cost more there)
    E pre[V]; // pre[j] = x'[0] * ... * x'[j]
    E acc = F::one(fd);
#pragma unroll
    for (int j = 0; j < V; j++) {
        const E xj = x[j] == 0 ? F::one(fd) : x[j];
        bad |= x[j] == 0;
        acc = j == 0 ? xj : F::mul(fd, acc, xj);
        pre[j] = acc;
    }
    E inv = F::inv(fd, acc);
#pragma unroll
    for (int j = V - 1; j >= 0; j--) {
        const E xj = x[j] == 0 ? F::one(fd) : x[j];
        const E r = j == 0 ? inv : F::mul(fd, inv, pre[j - 1]);
        inv = F::mul(fd, inv, xj);
        x[j] = x[j] == 0 ? (E)0 : r;
    }
}

// The same for a prime field with odd p < 2^31 in SKEWED Montgomery form: every product of the trick is a Montgomery reduction
// (5 instructions: v_mad_u64_u32, v_mul_lo, v_mad_u64_u32, v_sub, v_min) instead of a 64-bit Barrett reduction (14), and no
// conversion is needed at either end.  pre[j] = redc(pre[j-1] * x_j) = x_0 ... x_j * R^-j, so the plain inverse of the total,
// T^-1 = (x_0 ... x_{V-1})^-1 * R^(V-1), is exactly the start value the way back needs: with I_j = (x_0 ... x_j)^-1 * R^j,
// redc(I_j * pre[j-1]) = x_j^-1 (plain) and redc(I_j * x_j) = I_{j-1}.
template <int V>
__device__ __forceinline__ void batch_inverse_m31(const FieldDev &fd, u32 (&x)[V], bool &bad)
{
    const u32 p = (u32)fd.p;
    u32 pinv = p; // p * pinv == 1 (mod 2^32)
    for (int i = 0; i < 4; i++) pinv *= 2u - p * pinv;
    const u32 ninv = 0u - pinv;
    auto mm = [&](u32 a, u32 b) -> u32 {
        const u64 t = (u64)a * b;
        const u32 m = (u32)t * ninv;
        const u32 r = (u32)((t + (u64)m * p) >> 32); // < 2p < 2^32
        const u32 d = r - p;
        return d < r ? d : r; // min(r, r - p) as unsigned: r - p wraps above r exactly when r < p
    };
    u32 pre[V], nz[V];
    u32 acc = 1, any_zero = 0;
#pragma unroll
    for (int j = 0; j < V; j++) {
        nz[j] = nz_flag(x[j]);
        const u32 xj = x[j] | (nz[j] ^ 1u);
        any_zero |= nz[j] ^ 1u;
        acc = j == 0 ? xj : mm(acc, xj);
        pre[j] = acc;
    }
    bad |= any_zero != 0;
    u32 inv = Prime32::inv(fd, acc);
#pragma unroll
    for (int j = V - 1; j >= 0; j--) {
        const u32 xj = x[j] | (nz[j] ^ 1u);
        const u32 r = j == 0 ? inv : mm(inv, pre[j - 1]);
        inv = mm(inv, xj);
        x[j] = r & (0u - nz[j]);
    }
}
template <class F, int V>
__device__ __forceinline__ void batch_inverse_fast(const FieldDev &fd, typename F::elem (&x)[V], bool &bad)
{
    if constexpr (std::is_same<F, Prime32>::value) {
        if ((fd.p & 1) && !(fd.p >> 31)) { // wave-uniform
            batch_inverse_m31<V>(fd, x, bad);
            return;
        }
    }
    batch_inverse<F, V>(fd, x, bad);
}

template <class F, typename T, int OP, bool VEC>
__global__ __launch_bounds__(256) void ew_binary_kernel(FieldDev fd, const T *__restrict__ a, int sa,
                                                        const T *__restrict__ b, int sb, T *__restrict__ out, i64 n,
                                                        int32_t *err)
{
    typedef typename F::elem E;
    bool bad = false;
    const i64 tid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 nth = (i64)gridDim.x * blockDim.x;
    if constexpr (VEC) {
        constexpr int V = Vec16<T>::N;
        const i64 nvec = n / V;
        const E a0 = sa ? 0 : (E)a[0];
        const E b0 = sb ? 0 : (E)b[0];
        i64 i0 = tid;
        constexpr int NB = BatchN<F, T>::value;
        if constexpr (OP == GFA_OP_DIV && BatchInv<F>::value && (V < 16)) {
            if (sb) { // divisors are an array: invert 16 of them per lane with one exponentiation (see ew_unary_kernel)
                constexpr int NV = NB / V;
                for (; i0 + (i64)(NV - 1) * nth < nvec; i0 += (i64)NV * nth) {
                    E yv[NB];
#pragma unroll
                    for (int k = 0; k < NV; k++) {
                        const Vec16<T> bv = reinterpret_cast<const Vec16<T> *>(b)[i0 + (i64)k * nth];
#pragma unroll
                        for (int j = 0; j < V; j++) yv[k * V + j] = (E)bv.v[j];
                    }
                    batch_inverse_fast<F, NB>(fd, yv, bad);
#pragma unroll
                    for (int k = 0; k < NV; k++) {
                        Vec16<T> av, ov;
                        if (sa) av = reinterpret_cast<const Vec16<T> *>(a)[i0 + (i64)k * nth];
#pragma unroll
                        for (int j = 0; j < V; j++) ov.v[j] = (T)F::mul(fd, sa ? (E)av.v[j] : a0, yv[k * V + j]);
                        reinterpret_cast<Vec16<T> *>(out)[i0 + (i64)k * nth] = ov;
                    }
                }
            }
        }
        for (i64 i = i0; i < nvec; i += nth) {
            Vec16<T> av, bv, ov;
            if (sa) av = reinterpret_cast<const Vec16<T> *>(a)[i];
            if (sb) bv = reinterpret_cast<const Vec16<T> *>(b)[i];
            if constexpr (OP == GFA_OP_DIV && BatchInv<F>::value) {
                E yv[V];
#pragma unroll
                for (int j = 0; j < V; j++) yv[j] = sb ? (E)bv.v[j] : b0;
                batch_inverse_fast<F, V>(fd, yv, bad);
#pragma unroll
                for (int j = 0; j < V; j++) ov.v[j] = (T)F::mul(fd, sa ? (E)av.v[j] : a0, yv[j]);
            } else {
#pragma unroll
                for (int j = 0; j < V; j++) {
                    E x = sa ? (E)av.v[j] : a0;
                    E y = sb ? (E)bv.v[j] : b0;
                    ov.v[j] = (T)apply_binary<F, OP>(fd, x, y, bad);
                }
            }
            reinterpret_cast<Vec16<T> *>(out)[i] = ov;
        }
        for (i64 i = nvec * V + tid; i < n; i += nth)
            out[i] = (T)apply_binary<F, OP>(fd, (E)a[sa ? i : 0], (E)b[sb ? i : 0], bad);
    } else {
        for (i64 i = tid; i < n; i += nth)
            out[i] = (T)apply_binary<F, OP>(fd, (E)a[sa ? i : 0], (E)b[sb ? i : 0], bad);
    }
    if constexpr (OP == GFA_OP_DIV) flag_error(err, bad);
}

template <class F, typename T, int OP, bool VEC>
__global__ __launch_bounds__(256) void ew_unary_kernel(FieldDev fd, const T *__restrict__ a, T *__restrict__ out, i64 n,
                                                       int32_t *err)
{
    typedef typename F::elem E;
    bool bad = false;
    const i64 tid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 nth = (i64)gridDim.x * blockDim.x;
    if constexpr (VEC) {
        constexpr int V = Vec16<T>::N;
        const i64 nvec = n / V;
        i64 i0 = tid;
        constexpr int NB = BatchN<F, T>::value;
        if constexpr (OP == GFA_OP_RECIP && BatchInv<F>::value && (V < 16)) {
            // Montgomery's trick over 16 elements per lane (16 / V vectors, nth apart so that every load stays coalesced):
            // one exponentiation a^(p-2) per 16 elements instead of one per vector
            constexpr int NV = NB / V;
            for (; i0 + (i64)(NV - 1) * nth < nvec; i0 += (i64)NV * nth) {
                E xv[NB];
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    const Vec16<T> av = reinterpret_cast<const Vec16<T> *>(a)[i0 + (i64)k * nth];
#pragma unroll
                    for (int j = 0; j < V; j++) xv[k * V + j] = (E)av.v[j];
                }
                batch_inverse_fast<F, NB>(fd, xv, bad);
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    Vec16<T> ov;
#pragma unroll
                    for (int j = 0; j < V; j++) ov.v[j] = (T)xv[k * V + j];
                    reinterpret_cast<Vec16<T> *>(out)[i0 + (i64)k * nth] = ov;
                }
            }
        }
        for (i64 i = i0; i < nvec; i += nth) {
            Vec16<T> av = reinterpret_cast<const Vec16<T> *>(a)[i], ov;
            if constexpr (OP == GFA_OP_RECIP && BatchInv<F>::value) {
                E xv[V];
#pragma unroll
                for (int j = 0; j < V; j++) xv[j] = (E)av.v[j];
                batch_inverse_fast<F, V>(fd, xv, bad);
#pragma unroll
                for (int j = 0; j < V; j++) ov.v[j] = (T)xv[j];
            } else {
#pragma unroll
                for (int j = 0; j < V; j++) ov.v[j] = (T)apply_unary<F, OP>(fd, (E)av.v[j], bad);
            }
            reinterpret_cast<Vec16<T> *>(out)[i] = ov;
        }
        for (i64 i = nvec * V + tid; i < n; i += nth) out[i] = (T)apply_unary<F, OP>(fd, (E)a[i], bad);
    } else {
        for (i64 i = tid; i < n; i += nth) out[i] = (T)apply_unary<F, OP>(fd, (E)a[i], bad);
    }
    if constexpr (OP == GFA_OP_RECIP) flag_error(err, bad);
}

// np.power(x, e) with an int64 exponent array, and field * integer (both operands array or broadcast scalar)
template <class F, typename T, bool IS_POW>
__global__ __launch_bounds__(256) void ew_intarg_kernel(FieldDev fd, const T *__restrict__ a, int sa,
                                                        const i64 *__restrict__ e, int se, T *__restrict__ out, i64 n,
                                                        int32_t *err)
{
    typedef typename F::elem E;
    bool bad = false;
    const i64 tid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 nth = (i64)gridDim.x * blockDim.x;
    i64 done = 0;
    if (sa && !se && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        // broadcast exponent (a scalar register): 16-byte vectors of the field operand and of the result.  With one exponent
        // per element the 8-byte exponent stream dominates and the element-per-lane loop below keeps it coalesced.
        constexpr int V = Vec16<T>::N;
        const i64 nvec = n / V;
        const i64 k0 = se ? 0 : e[0];
        for (i64 i = tid; i < nvec; i += nth) {
            const Vec16<T> av = reinterpret_cast<const Vec16<T> *>(a)[i];
            Vec16<T> ov;
            if constexpr (IS_POW && BatchInv<F>::value && (V > 1)) {
                if (k0 < 0) {
                    // x^-k = (x^k)^-1: the V powers of this vector share one inversion (Montgomery's trick) instead of V
                    E y[V];
                    // |k0| as an unsigned value (k0 == INT64_MIN has no signed negation), reduced modulo q - 1 like the
                    // exponent-array loop below
                    u64 ue = (u64)0 - (u64)k0;
                    if (fd.q != 0 && ue >= fd.q - 1) ue %= (fd.q - 1);
#pragma unroll
                    for (int j = 0; j < V; j++) {
                        const E x = (E)av.v[j];
                        y[j] = x == 0 ? (E)0 : F::pow_u(fd, x, ue); // x = 0 gives 0, which batch_inverse flags
                    }
                    batch_inverse_fast<F, V>(fd, y, bad);
#pragma unroll
                    for (int j = 0; j < V; j++) ov.v[j] = (T)y[j];
                    reinterpret_cast<Vec16<T> *>(out)[i] = ov;
                    continue;
                }
            }
#pragma unroll
            for (int j = 0; j < V; j++) {
                const i64 k = se ? e[i * V + j] : k0;
                E r;
                if constexpr (IS_POW) {
                    if (!pow_signed<F>(fd, (E)av.v[j], k, &r)) bad = true;
                } else {
                    r = F::mul(fd, (E)av.v[j], (E)F::from_int(fd, k));
                }
                ov.v[j] = (T)r;
            }
            reinterpret_cast<Vec16<T> *>(out)[i] = ov;
        }
        done = nvec * V;
    }
    i64 i1 = done + tid;
    if constexpr (IS_POW && BatchInv<F>::value) {
        if (se) {
            // One exponent per element.  Eight elements per lane and round (each a coalesced access of the wavefront): their
            // powers x^|k| first, then ONE inversion for the negative exponents among them -- an inversion is an exponentiation
            // by p - 2 (or ~27 products in GF(p^m)), and with a few negative exponents anywhere in a wavefront every lane used to
            // wait for one per element.  |k| is reduced modulo q - 1 only where it is not below it already (a 64-bit division).
            constexpr int PB = 8;
            for (; i1 + (i64)(PB - 1) * nth < n; i1 += (i64)PB * nth) {
                E y[PB];
                bool ng[PB];
                bool anyneg = false;
#pragma unroll
                for (int k = 0; k < PB; k++) {
                    const i64 idx = i1 + (i64)k * nth;
                    const E x = (E)a[sa ? idx : 0];
                    const i64 kk = e[idx];
                    u64 ue = kk < 0 ? (u64)0 - (u64)kk : (u64)kk;
                    if (fd.q != 0 && ue >= fd.q - 1) ue %= (fd.q - 1);
                    E r;
                    if (kk == 0) r = F::one(fd);
                    else if (x == 0) r = 0;
                    else r = F::pow_u(fd, x, ue);
                    bad |= (x == 0 && kk < 0);
                    ng[k] = kk < 0 && x != 0;
                    anyneg |= ng[k];
                    y[k] = r;
                }
                if (anyneg) {
                    E z[PB];
                    bool unused = false;
#pragma unroll
                    for (int k = 0; k < PB; k++) z[k] = ng[k] ? y[k] : F::one(fd);
                    batch_inverse_fast<F, PB>(fd, z, unused);
#pragma unroll
                    for (int k = 0; k < PB; k++) y[k] = ng[k] ? z[k] : y[k];
                }
#pragma unroll
                for (int k = 0; k < PB; k++) out[i1 + (i64)k * nth] = (T)y[k];
            }
        }
    }
    for (i64 i = i1; i < n; i += nth) {
        E x = (E)a[sa ? i : 0];
        i64 k = e[se ? i : 0];
        E r;
        if constexpr (IS_POW) {
            if (!pow_signed<F>(fd, x, k, &r)) bad = true;
        } else {
            r = F::mul(fd, x, (E)F::from_int(fd, k));
        }
        out[i] = (T)r;
    }
    if constexpr (IS_POW) flag_error(err, bad);
}

// ------------------------------------------------------------------------------------------------
// Order <= 256, uint8: full-table kernels
// ------------------------------------------------------------------------------------------------
constexpr int TAB8_THREADS = 1024;

__device__ __forceinline__ u32 lookup4(const uint8_t *lds, u32 aw, u32 bw)
{
    // index = (a_k << 8) | b_k assembled with one v_perm_b32 per element
    u32 i0 = __builtin_amdgcn_perm(aw, bw, 0x0c0c0400u);
    u32 i1 = __builtin_amdgcn_perm(aw, bw, 0x0c0c0501u);
    u32 i2 = __builtin_amdgcn_perm(aw, bw, 0x0c0c0602u);
    u32 i3 = __builtin_amdgcn_perm(aw, bw, 0x0c0c0703u);
    u32 r0 = lds[i0], r1 = lds[i1], r2 = lds[i2], r3 = lds[i3];
    return r0 | (r1 << 8) | (r2 << 16) | (r3 << 24);
}

// out = TABLE[a][b].  CHECK_ZERO_B: division flags b == 0.
// Software-pipelined: the first operand vectors are requested BEFORE the table is staged (the 64 KiB L2->LDS copy then
// overlaps the first HBM round trip), and each lane re-arms its load for the next vector before it does the 16 LDS
// lookups of the current one.  Measured (tools/ubench/stream3.hip, 1e8 elements): 46.9 us against 49.6 us for the
// load-all / lookup-all / store-all loop; the plain a^b stream in the same launch shape takes 46.5 us.
template <bool CHECK_ZERO_B>
__global__ __launch_bounds__(TAB8_THREADS) void tab8_binary_kernel(const uint8_t *__restrict__ table,
                                                                    const uint8_t *__restrict__ a,
                                                                    const uint8_t *__restrict__ b,
                                                                    uint8_t *__restrict__ out, i64 n, int32_t *err)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const i64 nvec = n >> 4;
    const u32x4 *av = reinterpret_cast<const u32x4 *>(a);
    const u32x4 *bv = reinterpret_cast<const u32x4 *>(b);
    u32x4 *ov = reinterpret_cast<u32x4 *>(out);
    const i64 stride = (i64)gridDim.x * TAB8_THREADS;
    i64 i = (i64)blockIdx.x * TAB8_THREADS + threadIdx.x;
    u32x4 x = {0, 0, 0, 0}, y = {0, 0, 0, 0};
    if (i < nvec) { x = av[i]; y = bv[i]; }
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(table);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        for (int t = threadIdx.x; t < 65536 / 16; t += TAB8_THREADS) dst[t] = src[t];
    }
    __syncthreads();
    bool bad = false;
    for (; i < nvec; i += stride) {
        const u32x4 cx = x, cy = y;
        const i64 nxt = i + stride;
        if (nxt < nvec) { x = av[nxt]; y = bv[nxt]; }
        u32x4 r;
        r.x = lookup4(lds, cx.x, cy.x);
        r.y = lookup4(lds, cx.y, cy.y);
        r.z = lookup4(lds, cx.z, cy.z);
        r.w = lookup4(lds, cx.w, cy.w);
        if constexpr (CHECK_ZERO_B) {
            // a byte of y is zero  <=>  (v - 0x01010101) & ~v & 0x80808080 != 0
            u32 z = ((cy.x - 0x01010101u) & ~cy.x) | ((cy.y - 0x01010101u) & ~cy.y) | ((cy.z - 0x01010101u) & ~cy.z) |
                    ((cy.w - 0x01010101u) & ~cy.w);
            bad |= (z & 0x80808080u) != 0;
        }
        ov[i] = r;
    }
    // tail (< 16 elements)
    for (i64 j = (nvec << 4) + (i64)blockIdx.x * TAB8_THREADS + threadIdx.x; j < n; j += stride) {
        uint8_t yb = b[j];
        if (CHECK_ZERO_B && yb == 0) bad = true;
        out[j] = lds[((u32)a[j] << 8) | yb];
    }
    if constexpr (CHECK_ZERO_B) flag_error(err, bad);
}

// Large arrays (beyond the Infinity Cache): same table kernel, but the persistent workgroups CLAIM their next 32 KiB block
// from a global counter instead of striding statically.  Statically assigned workgroups drift apart, the set of DRAM pages
// in use spreads, and the stream falls to ~5.0 TB/s; claiming in completion order keeps the active window compact, like a
// flat launch (measured on 1e9-byte operands, tools/ubench/stream3.hip: static 5.0, claimed 5.97, flat 6.0 TB/s).  The claim
// for the block after next is taken while the current one is looked up, and its loads are issued before the current
// results are stored.  For small arrays the atomic costs more than the drift (51 vs 47 us at 1e8 elements).
template <bool CHECK_ZERO_B>
__global__ __launch_bounds__(TAB8_THREADS) void tab8_binary_claim_kernel(const uint8_t *__restrict__ table,
                                                                          const uint8_t *__restrict__ a,
                                                                          const uint8_t *__restrict__ b,
                                                                          uint8_t *__restrict__ out, i64 n, int32_t *err,
                                                                          unsigned int *__restrict__ counter)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ unsigned int s_next;
    constexpr int U = 2; // vectors per thread per block: 2 * 1024 * 16 B = 32 KiB per operand
    const i64 nvec = n >> 4;
    const i64 nblk = (nvec + U * TAB8_THREADS - 1) / (U * TAB8_THREADS);
    const u32x4 *av = reinterpret_cast<const u32x4 *>(a);
    const u32x4 *bv = reinterpret_cast<const u32x4 *>(b);
    u32x4 *ov = reinterpret_cast<u32x4 *>(out);
    i64 blk = blockIdx.x; // the first block of every workgroup is static
    u32x4 x[U], y[U];
    auto issue = [&](i64 bk) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const i64 i = (bk * U + u) * TAB8_THREADS + threadIdx.x;
            if (i < nvec) { x[u] = __builtin_nontemporal_load(av + i); y[u] = __builtin_nontemporal_load(bv + i); }
        }
    };
    if (blk < nblk) issue(blk);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(table);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        for (int t = threadIdx.x; t < 65536 / 16; t += TAB8_THREADS) dst[t] = src[t];
    }
    __syncthreads();
    bool bad = false;
    while (blk < nblk) {
        if (threadIdx.x == 0) s_next = atomicAdd(counter, 1u) + gridDim.x;
        u32x4 r[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            r[u].x = lookup4(lds, x[u].x, y[u].x);
            r[u].y = lookup4(lds, x[u].y, y[u].y);
            r[u].z = lookup4(lds, x[u].z, y[u].z);
            r[u].w = lookup4(lds, x[u].w, y[u].w);
            if constexpr (CHECK_ZERO_B) {
                u32 z = ((y[u].x - 0x01010101u) & ~y[u].x) | ((y[u].y - 0x01010101u) & ~y[u].y) | ((y[u].z - 0x01010101u) & ~y[u].z) |
                        ((y[u].w - 0x01010101u) & ~y[u].w);
                bad |= (z & 0x80808080u) != 0;
            }
        }
        __syncthreads();
        const i64 cur = blk;
        blk = s_next;
        __syncthreads();
        if (blk < nblk) issue(blk);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const i64 i = (cur * U + u) * TAB8_THREADS + threadIdx.x;
            if (i < nvec) __builtin_nontemporal_store(r[u], ov + i);
        }
    }
    // tail (< 16 elements): first workgroup
    if (blockIdx.x == 0)
        for (i64 j = (nvec << 4) + threadIdx.x; j < n; j += TAB8_THREADS) {
            uint8_t yb = b[j];
            if (CHECK_ZERO_B && yb == 0) bad = true;
            out[j] = lds[((u32)a[j] << 8) | yb];
        }
    if constexpr (CHECK_ZERO_B) flag_error(err, bad);
}

// Unary / scalar-operand form: out = TABLE256[a].  The 256-entry table is replicated 32x in LDS as dwords,
// entry v of copy c at dword v*32 + c, and lane l reads copy l%32 => every lane of a 32-lane LDS group hits its
// own bank, no conflicts for any data.
// POW: the table is not read but built by the first 256 lanes of every workgroup as v -> v ** e[0] from the field's EXP / LOG
// (x ** k with one exponent for the whole array is a unary map of at most 256 values); a zero in the data raises only for
// e[0] < 0, as the reference's 0 ** negative does.
template <bool CHECK_ZERO, bool POW = false>
__global__ __launch_bounds__(TAB8_THREADS) void tab8_unary_kernel(const uint8_t *__restrict__ table256,
                                                                   const uint8_t *__restrict__ a,
                                                                   uint8_t *__restrict__ out, i64 n, int32_t *err,
                                                                   FieldDev fd = FieldDev{}, const i64 *__restrict__ e = nullptr)
{
    __shared__ u32 rep[256 * 32];
    __shared__ uint8_t powtab[POW ? 256 : 1];
    if constexpr (POW) {
        if (threadIdx.x < 256) {
            u32 r = 0;
            if (threadIdx.x < fd.q) (void)pow_signed<Lut>(fd, (u32)threadIdx.x, e[0], &r);
            powtab[threadIdx.x] = (uint8_t)r;
        }
        __syncthreads();
        table256 = powtab;
    }
    const i64 nvec = n >> 4;
    const u32x4 *av = reinterpret_cast<const u32x4 *>(a);
    u32x4 *ov = reinterpret_cast<u32x4 *>(out);
    const i64 stride = (i64)gridDim.x * TAB8_THREADS;
    i64 i = (i64)blockIdx.x * TAB8_THREADS + threadIdx.x;
    u32x4 x = {0, 0, 0, 0};
    if (i < nvec) x = av[i];   // in flight while the table is replicated (same pipelining as the binary kernel)
    for (int t = threadIdx.x; t < 256 * 32; t += TAB8_THREADS) rep[t] = table256[t >> 5];
    __syncthreads();
    const u32 *my = rep + (threadIdx.x & 31);
    bool bad = false;
    auto map4 = [&](u32 w) -> u32 {
        u32 r0 = my[(w & 0xff) << 5], r1 = my[((w >> 8) & 0xff) << 5], r2 = my[((w >> 16) & 0xff) << 5],
            r3 = my[(w >> 24) << 5];
        return r0 | (r1 << 8) | (r2 << 16) | (r3 << 24);
    };
    auto haszero = [](u32 v) -> u32 { return (v - 0x01010101u) & ~v & 0x80808080u; };
    for (; i < nvec; i += stride) {
        const u32x4 cx = x;
        const i64 nxt = i + stride;
        if (nxt < nvec) x = av[nxt];
        u32x4 r;
        r.x = map4(cx.x); r.y = map4(cx.y); r.z = map4(cx.z); r.w = map4(cx.w);
        if constexpr (CHECK_ZERO) bad |= (haszero(cx.x) | haszero(cx.y) | haszero(cx.z) | haszero(cx.w)) != 0;
        ov[i] = r;
    }
    for (i64 j = (nvec << 4) + (i64)blockIdx.x * TAB8_THREADS + threadIdx.x; j < n; j += stride) {
        uint8_t xb = a[j];
        if (CHECK_ZERO && xb == 0) bad = true;
        out[j] = (uint8_t)my[(u32)xb << 5];
    }
    if constexpr (POW) bad = bad && e[0] < 0;
    if constexpr (CHECK_ZERO) flag_error(err, bad);
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
int num_cus()
{
    static int cached[64] = {0};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return 256;
    if (!cached[d]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) return 256;
        cached[d] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cached[d];
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int grid_for(i64 work_items, int threads, int blocks_per_cu)
{
    i64 blocks = (work_items + threads - 1) / threads;
    i64 cap = (i64)num_cus() * blocks_per_cu;
    if (blocks < 1) blocks = 1;
    return (int)(blocks < cap ? blocks : cap);
}

// Streaming kernels without per-workgroup set-up are launched FLAT (one 16-byte vector per thread).  With a persistent
// grid-stride launch the workgroups drift apart, the active address window spreads and HBM efficiency drops once the
// arrays exceed the Infinity Cache: 5.0 vs 6.0-6.6 TB/s on 1e9-byte operands (tools/ubench/stream_big.hip).
inline int grid_flat(i64 work_items, int threads)
{
    i64 blocks = (work_items + threads - 1) / threads;
    if (blocks < 1) blocks = 1;
    return (int)(blocks < 0x7fffffff ? blocks : 0x7fffffff);
}

template <class F, typename T>
int launch_binary_ft(const FieldDev &fd, int op, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n,
                     hipStream_t st, int32_t *err)
{
    const T *pa = (const T *)a, *pb = (const T *)b;
    T *po = (T *)out;
    const bool vec = aligned16(out) && (sa == 0 || aligned16(a)) && (sb == 0 || aligned16(b));
    constexpr int V = Vec16<T>::N;
    // array / array division in a prime field inverts 16 divisors per lane at a time: 16 / V vectors per thread
    const int per_thread = (op == GFA_OP_DIV && sb != 0 && BatchInv<F>::value && V < 16) ? BatchN<F, T>::value / V : 1;
    const int grid = grid_flat(vec ? ((n + V - 1) / V + per_thread - 1) / per_thread : n, 256);
#define GFA_LAUNCH_B(OPC)                                                                                              \
    if (vec) hipLaunchKernelGGL((ew_binary_kernel<F, T, OPC, true>), dim3(grid), dim3(256), 0, st, fd, pa, (int)sa, pb, \
                                (int)sb, po, n, err);                                                                  \
    else hipLaunchKernelGGL((ew_binary_kernel<F, T, OPC, false>), dim3(grid), dim3(256), 0, st, fd, pa, (int)sa, pb,    \
                            (int)sb, po, n, err);
    switch (op) {
    case GFA_OP_ADD: GFA_LAUNCH_B(GFA_OP_ADD) break;
    case GFA_OP_SUB: GFA_LAUNCH_B(GFA_OP_SUB) break;
    case GFA_OP_MUL: GFA_LAUNCH_B(GFA_OP_MUL) break;
    case GFA_OP_DIV: GFA_LAUNCH_B(GFA_OP_DIV) break;
    default: set_error("gfa_binary: bad op"); return GFA_ERR_INVALID;
    }
#undef GFA_LAUNCH_B
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

template <class F, typename T>
int launch_unary_ft(const FieldDev &fd, int op, const void *a, void *out, i64 n, hipStream_t st, int32_t *err)
{
    const T *pa = (const T *)a;
    T *po = (T *)out;
    const bool vec = aligned16(out) && aligned16(a);
    constexpr int V = Vec16<T>::N;
    const int per_thread = (op == GFA_OP_RECIP && BatchInv<F>::value && V < 16) ? BatchN<F, T>::value / V : 1; // see ew_unary_kernel
    const int grid = grid_flat(vec ? ((n + V - 1) / V + per_thread - 1) / per_thread : n, 256);
#define GFA_LAUNCH_U(OPC)                                                                                             \
    if (vec) hipLaunchKernelGGL((ew_unary_kernel<F, T, OPC, true>), dim3(grid), dim3(256), 0, st, fd, pa, po, n, err); \
    else hipLaunchKernelGGL((ew_unary_kernel<F, T, OPC, false>), dim3(grid), dim3(256), 0, st, fd, pa, po, n, err);
    switch (op) {
    case GFA_OP_NEG: GFA_LAUNCH_U(GFA_OP_NEG) break;
    case GFA_OP_RECIP: GFA_LAUNCH_U(GFA_OP_RECIP) break;
    default: set_error("gfa_unary: bad op"); return GFA_ERR_INVALID;
    }
#undef GFA_LAUNCH_U
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

template <class F, typename T>
int launch_intarg_ft(const FieldDev &fd, bool is_pow, const void *a, i64 sa, const i64 *e, i64 se, void *out, i64 n,
                     hipStream_t st, int32_t *err)
{
    const int grid = se ? grid_for(n, 256, 8) : grid_for((n + Vec16<T>::N - 1) / Vec16<T>::N, 256, 16);
    if (is_pow)
        hipLaunchKernelGGL((ew_intarg_kernel<F, T, true>), dim3(grid), dim3(256), 0, st, fd, (const T *)a, (int)sa, e,
                           (int)se, (T *)out, n, err);
    else
        hipLaunchKernelGGL((ew_intarg_kernel<F, T, false>), dim3(grid), dim3(256), 0, st, fd, (const T *)a, (int)sa, e,
                           (int)se, (T *)out, n, err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}


// ------------------------------------------------------------------------------------------------
// GF(2^m) multiply in calculate mode on packed storage: four uint8 (m <= 8) or two uint16 (m <= 16) elements per 32-bit
// register go through the shift-and-xor product together (reference: multiply_binary, _domains/_calculate.py:288-324,
// same steps, m of them, for every element at once).  Per step and register: the multiplier's bit i of every element is
// spread to an element-wide mask by one multiplication, the partial product is and / xor-ed in, and the multiplicand is
// doubled with its top bits folded back through a second multiplication by the low part of the irreducible polynomial
// (no carries between elements: the top bit is cleared before the shift).  12 vector instructions per step and register,
// i.e. 24 per uint8 element at m = 8 -- against one LDS gather per element for the table kernel.
// ------------------------------------------------------------------------------------------------
template <int W> // element width in bits: 8 or 16
__device__ __forceinline__ u32 swar_gf2m_mul(u32 a, u32 b, int m, u32 lsb, u32 top, u32 red)
{
    constexpr u32 ONES = (1u << W) - 1u;
    u32 c = 0;
    for (int i = 0; i < m; i++) {
        const u32 mask = ((b >> i) & lsb) * ONES; // 0x00 / 0xff per element
        c ^= a & mask;
        const u32 hi = a & top;                   // elements whose bit m-1 is set
        a = ((a ^ hi) << 1) ^ ((hi >> (m - 1)) * red);
    }
    return c;
}

template <typename T>
__global__ __launch_bounds__(256) void bin_swar_mul_kernel(const T *__restrict__ a, int sa, const T *__restrict__ b, int sb,
                                                           T *__restrict__ out, i64 n, int m, u32 irr_low)
{
    constexpr int W = 8 * (int)sizeof(T);
    constexpr int PER = 32 / W;              // elements per register
    constexpr int V = 16 / (int)sizeof(T);   // elements per 16-byte vector
    constexpr u32 lsb = W == 8 ? 0x01010101u : 0x00010001u;
    const u32 top = lsb << (m - 1);
    const i64 tid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 nth = (i64)gridDim.x * blockDim.x;
    const i64 nvec = n / V;
    const u32 a0 = sa ? 0u : (u32)a[0] * lsb, b0 = sb ? 0u : (u32)b[0] * lsb; // broadcast scalars, replicated per element
    for (i64 i = tid; i < nvec; i += nth) {
        uint4 av = sa ? reinterpret_cast<const uint4 *>(a)[i] : make_uint4(a0, a0, a0, a0);
        uint4 bv = sb ? reinterpret_cast<const uint4 *>(b)[i] : make_uint4(b0, b0, b0, b0);
        uint4 ov;
        ov.x = swar_gf2m_mul<W>(av.x, bv.x, m, lsb, top, irr_low);
        ov.y = swar_gf2m_mul<W>(av.y, bv.y, m, lsb, top, irr_low);
        ov.z = swar_gf2m_mul<W>(av.z, bv.z, m, lsb, top, irr_low);
        ov.w = swar_gf2m_mul<W>(av.w, bv.w, m, lsb, top, irr_low);
        reinterpret_cast<uint4 *>(out)[i] = ov;
    }
    for (i64 i = nvec * V + tid; i < n; i += nth) // tail: one element in the low lanes of a register
        out[i] = (T)swar_gf2m_mul<W>((u32)a[sa ? i : 0], (u32)b[sb ? i : 0], m, lsb, top, irr_low);
    (void)PER;
}

// ------------------------------------------------------------------------------------------------
// GF(2^m), 9 <= m <= 16 on uint16 storage: the carry-less product out of INTEGER multiplies ("multiplication with holes",
// Bin::clmul21's idea at 16 bits): the bit positions of an operand are split into three classes mod 3 (at most six set bits
// each, three apart), so an integer product of two classes piles at most six partial ones on a position -- a sum that fits
// the three bits below the next position of its class, and whose lowest bit is the carry-less sum.  Nine 32-bit products
// per element (the element in the high half of a register is multiplied in place through v_mul_hi_u32: (x << 16)(y << 16)
// = xy << 32), then the 2m - 1 bit product is reduced through two 256-entry tables in LDS (h(x) x^m mod f and
// h(x) x^(m+8) mod f, built at kernel entry for whatever irreducible polynomial the field has).  About 46 fast-instruction
// equivalents per element against 6 m for the packed shift-and-xor product above (96 at m = 16).
// ------------------------------------------------------------------------------------------------
constexpr int BIN16_VECS = 8;
__global__ __launch_bounds__(256) void bin16_holes_mul_kernel(const uint16_t *__restrict__ a, int sa, const uint16_t *__restrict__ b, int sb,
                                                              uint16_t *__restrict__ out, i64 n, int m, u32 irr)
{
    __shared__ uint16_t R[512];
    R[threadIdx.x] = (uint16_t)Bin::reduce_bits((u64)threadIdx.x << m, m, 8, irr);
    R[256 + threadIdx.x] = (uint16_t)Bin::reduce_bits((u64)threadIdx.x << (m + 8), m, 16, irr);
    __syncthreads();
    const u32 low = (1u << m) - 1u;
    auto finish = [&](u32 P) -> u32 { return (P & low) ^ (u32)R[(P >> m) & 0xffu] ^ (u32)R[256 + (P >> (m + 8))]; };
    auto mul2 = [&](u32 x, u32 y) -> u32 { // two elements per register (Bin::clmul16_lo / _hi, gfa_arith.h)
        return finish(Bin::clmul16_lo(x, y)) | (finish(Bin::clmul16_hi(x, y)) << 16);
    };
    // a workgroup takes BIN16_VECS consecutive blocks of 256 vectors (the table set-up is a quarter of one block's work)
    const i64 nvec = n / 8;
    const u32 a0 = sa ? 0u : (u32)a[0] * 0x10001u, b0 = sb ? 0u : (u32)b[0] * 0x10001u; // broadcast scalars, replicated per element
    for (i64 blk = (i64)blockIdx.x * BIN16_VECS; blk * 256 < nvec; blk += (i64)gridDim.x * BIN16_VECS) {
#pragma unroll 1
        for (int k = 0; k < BIN16_VECS; k++) {
            const i64 i = (blk + k) * 256 + threadIdx.x;
            if (i >= nvec) break;
            const uint4 av = sa ? reinterpret_cast<const uint4 *>(a)[i] : make_uint4(a0, a0, a0, a0);
            const uint4 bv = sb ? reinterpret_cast<const uint4 *>(b)[i] : make_uint4(b0, b0, b0, b0);
            uint4 ov;
            ov.x = mul2(av.x, bv.x); ov.y = mul2(av.y, bv.y); ov.z = mul2(av.z, bv.z); ov.w = mul2(av.w, bv.w);
            reinterpret_cast<uint4 *>(out)[i] = ov;
        }
    }
    const i64 tid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (i64 i = nvec * 8 + tid; i < n; i += (i64)gridDim.x * blockDim.x) // tail: one element in the low half of a register
        out[i] = (uint16_t)mul2((u32)a[sa ? i : 0], (u32)b[sb ? i : 0]);
}

// GF(2^m), 17 <= m <= 32 on uint32 storage: Bin::clmul21 / clmul32 (carry-less product out of 9 / 16 integer multiplies), then the
// part above x^m is reduced through four 256-entry tables in LDS, one per byte of it (h(x) x^(m+8k) mod f, k = 0..3; each table is
// the previous one times x^8, so building them costs eight reduction steps per entry) instead of `rounds` folds of one 64-bit
// shift-and-xor per term of the irreducible polynomial (GF(2^32): 2 x 6 terms).
//
// r06, DIV (17 <= m <= 20, AUTO): the divisor is replaced by its inverse through ONE gather from the field's 3-byte inverse table
// (gfa_field::inverse_table) before the same product -- Itoh-Tsujii chains of these products ran at 0.12 of the roofline.
typedef u32 __attribute__((aligned(1))) u32_unaligned;
__device__ __forceinline__ u32 inv24_at(const uint8_t *__restrict__ t, u32 x) { return *reinterpret_cast<const u32_unaligned *>(t + 3u * x) & 0xffffffu; }

template <bool DIV>
__global__ __launch_bounds__(256) void bin32_tab_mul_kernel(const u32 *__restrict__ a, int sa, const u32 *__restrict__ b, int sb,
                                                            u32 *__restrict__ out, i64 n, int m, u64 irr, const uint8_t *__restrict__ inv24, int *err)
{
    __shared__ u32 R[4 * 256];
    {
        u64 v = Bin::reduce_bits((u64)threadIdx.x << m, m, 8, irr);
        R[threadIdx.x] = (u32)v;
#pragma unroll
        for (int k = 1; k < 4; k++) {
            v = Bin::reduce_bits(v << 8, m, 8, irr);
            R[k * 256 + threadIdx.x] = (u32)v;
        }
    }
    __syncthreads();
    const u64 low = m == 32 ? 0xffffffffull : (((u64)1 << m) - 1);
    const bool small = m <= 21, three = m <= 24; // operands below 2^21: nine products; the part above x^m below 2^24: three tables
    auto mul1 = [&](u32 x, u32 y) -> u32 {
        const u64 P = small ? Bin::clmul21(x, y) : Bin::clmul32(x, y);
        const u32 H = (u32)(P >> m);
        u32 r = (u32)(P & low) ^ R[H & 0xffu] ^ R[256 + ((H >> 8) & 0xffu)] ^ R[512 + ((H >> 16) & 0xffu)];
        if (!three) r ^= R[768 + (H >> 24)];
        return r;
    };
    const i64 nvec = n / 4;
    const u32 a0 = sa ? 0u : a[0], b0 = sb ? 0u : b[0];
    bool bad = false;
    for (i64 blk = (i64)blockIdx.x * BIN16_VECS; blk * 256 < nvec; blk += (i64)gridDim.x * BIN16_VECS) {
#pragma unroll 1
        for (int k = 0; k < BIN16_VECS; k++) {
            const i64 i = (blk + k) * 256 + threadIdx.x;
            if (i >= nvec) break;
            const uint4 av = sa ? reinterpret_cast<const uint4 *>(a)[i] : make_uint4(a0, a0, a0, a0);
            uint4 bv = sb ? reinterpret_cast<const uint4 *>(b)[i] : make_uint4(b0, b0, b0, b0);
            if (DIV) {
                bad |= bv.x == 0u || bv.y == 0u || bv.z == 0u || bv.w == 0u;
                bv = make_uint4(inv24_at(inv24, bv.x), inv24_at(inv24, bv.y), inv24_at(inv24, bv.z), inv24_at(inv24, bv.w));
            }
            uint4 ov;
            ov.x = mul1(av.x, bv.x); ov.y = mul1(av.y, bv.y); ov.z = mul1(av.z, bv.z); ov.w = mul1(av.w, bv.w);
            reinterpret_cast<uint4 *>(out)[i] = ov;
        }
    }
    const i64 tid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (i64 i = nvec * 4 + tid; i < n; i += (i64)gridDim.x * blockDim.x) {
        u32 y = b[sb ? i : 0];
        if (DIV) { bad |= y == 0u; y = inv24_at(inv24, y); }
        out[i] = mul1(a[sa ? i : 0], y);
    }
    if (DIV && bad && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
}

// quotients of GF(2^m), 17 <= m <= 20, uint32 arrays, through the inverse table (see bin32_tab_mul_kernel); GFA_ERR_UNSUPPORTED: not this shape
int bin32_div_by_table(const FieldDev &fd, const uint8_t *inv24, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st, int32_t *err)
{
    if (fd.kind != KIND_BIN || fd.m < 17 || fd.m > 20 || !inv24 || n < 1024 || !aligned16(out) || (sa && !aligned16(a)) || (sb && !aligned16(b)))
        return GFA_ERR_UNSUPPORTED;
    const int hgrid = grid_flat((n + 4 * BIN16_VECS - 1) / (4 * BIN16_VECS), 256);
    hipLaunchKernelGGL(bin32_tab_mul_kernel<true>, dim3(hgrid), dim3(256), 0, st, (const u32 *)a, (int)sa, (const u32 *)b, (int)sb, (u32 *)out, n,
                       (int)fd.m, (u64)fd.irr, inv24, err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// GF(p^m), 2 <= m <= 8, calculate mode: the kernels are instantiated per degree (ExtM<M>: digit arrays in registers)
#define GFA_EXT_FIXED_T(FUNC, M, dtype, ...)                                      \
    switch (dtype) {                                                              \
    case GFA_U8: return FUNC<ExtM<M>, uint8_t>(__VA_ARGS__);                      \
    case GFA_U16: return FUNC<ExtM<M>, uint16_t>(__VA_ARGS__);                    \
    case GFA_U32: return FUNC<ExtM<M>, uint32_t>(__VA_ARGS__);                    \
    default: return FUNC<ExtM<M>, uint64_t>(__VA_ARGS__);                         \
    }
#define GFA_EXT_FIXED(FUNC, fd, dtype, ...)                                       \
    if ((fd).kind == KIND_EXT && Ext::fixed_degree(fd)) {                         \
        switch ((fd).m) {                                                         \
        case 2: GFA_EXT_FIXED_T(FUNC, 2, dtype, __VA_ARGS__)                      \
        case 3: GFA_EXT_FIXED_T(FUNC, 3, dtype, __VA_ARGS__)                      \
        case 4: GFA_EXT_FIXED_T(FUNC, 4, dtype, __VA_ARGS__)                      \
        case 5: GFA_EXT_FIXED_T(FUNC, 5, dtype, __VA_ARGS__)                      \
        case 6: GFA_EXT_FIXED_T(FUNC, 6, dtype, __VA_ARGS__)                      \
        case 7: GFA_EXT_FIXED_T(FUNC, 7, dtype, __VA_ARGS__)                      \
        default: GFA_EXT_FIXED_T(FUNC, 8, dtype, __VA_ARGS__)                     \
        }                                                                         \
    }

int dispatch_binary(const FieldDev &fd, int dtype, int op, const void *a, i64 sa, const void *b, i64 sb, void *out,
                    i64 n, hipStream_t st, int32_t *err)
{
    if (fd.kind == KIND_BIN && op == GFA_OP_MUL && ((dtype == GFA_U8 && fd.m <= 8) || (dtype == GFA_U16 && fd.m <= 16)) &&
        aligned16(out) && (sa == 0 || aligned16(a)) && (sb == 0 || aligned16(b))) {
        const u32 irr_low = (u32)(fd.irr ^ ((u64)1 << fd.m));
        const int V = dtype == GFA_U8 ? 16 : 8;
        const int grid = grid_flat((n + V - 1) / V, 256);
        if (dtype == GFA_U16 && fd.m >= 9) {
            const int hgrid = grid_flat((n + 8 * BIN16_VECS - 1) / (8 * BIN16_VECS), 256);
            hipLaunchKernelGGL(bin16_holes_mul_kernel, dim3(hgrid), dim3(256), 0, st, (const uint16_t *)a, (int)sa, (const uint16_t *)b, (int)sb,
                               (uint16_t *)out, n, (int)fd.m, (u32)fd.irr);
            GFA_HIP(hipGetLastError());
            return GFA_OK;
        }
        if (dtype == GFA_U8)
            hipLaunchKernelGGL((bin_swar_mul_kernel<uint8_t>), dim3(grid), dim3(256), 0, st, (const uint8_t *)a, (int)sa, (const uint8_t *)b,
                               (int)sb, (uint8_t *)out, n, (int)fd.m, irr_low);
        else
            hipLaunchKernelGGL((bin_swar_mul_kernel<uint16_t>), dim3(grid), dim3(256), 0, st, (const uint16_t *)a, (int)sa,
                               (const uint16_t *)b, (int)sb, (uint16_t *)out, n, (int)fd.m, irr_low);
        GFA_HIP(hipGetLastError());
        return GFA_OK;
    }
    if (fd.kind == KIND_BIN && op == GFA_OP_MUL && dtype == GFA_U32 && fd.m >= 17 && fd.m <= 32 && aligned16(out) && (sa == 0 || aligned16(a)) &&
        (sb == 0 || aligned16(b))) {
        const int hgrid = grid_flat((n + 4 * BIN16_VECS - 1) / (4 * BIN16_VECS), 256);
        hipLaunchKernelGGL(bin32_tab_mul_kernel<false>, dim3(hgrid), dim3(256), 0, st, (const u32 *)a, (int)sa, (const u32 *)b, (int)sb, (u32 *)out, n,
                           (int)fd.m, (u64)fd.irr, (const uint8_t *)nullptr, (int *)nullptr);
        GFA_HIP(hipGetLastError());
        return GFA_OK;
    }
    GFA_EXT_FIXED(launch_binary_ft, fd, dtype, fd, op, a, sa, b, sb, out, n, st, err);
    GFA_DISPATCH_FT(launch_binary_ft, fd, dtype, fd, op, a, sa, b, sb, out, n, st, err);
}
int dispatch_unary(const FieldDev &fd, int dtype, int op, const void *a, void *out, i64 n, hipStream_t st,
                   int32_t *err)
{
    GFA_EXT_FIXED(launch_unary_ft, fd, dtype, fd, op, a, out, n, st, err);
    GFA_DISPATCH_FT(launch_unary_ft, fd, dtype, fd, op, a, out, n, st, err);
}
int dispatch_intarg(const FieldDev &fd, int dtype, bool is_pow, const void *a, i64 sa, const i64 *e, i64 se,
                    void *out, i64 n, hipStream_t st, int32_t *err)
{
    GFA_EXT_FIXED(launch_intarg_ft, fd, dtype, fd, is_pow, a, sa, e, se, out, n, st, err);
    GFA_DISPATCH_FT(launch_intarg_ft, fd, dtype, fd, is_pow, a, sa, e, se, out, n, st, err);
}


// ------------------------------------------------------------------------------------------------
// ufunc.reduce over the last axis (add / multiply are commutative monoids -> tree reduction; subtract / divide are
// the reference's left folds a0 - a1 - ... = a0 - sum(rest), a0 / a1 / ... = a0 / prod(rest))
// ------------------------------------------------------------------------------------------------
template <class F, typename T, bool IS_MUL>
__global__ __launch_bounds__(256) void reduce_segments_kernel(FieldDev fd, const T *__restrict__ in, i64 n_inner,
                                                              i64 col_begin, i64 seg_len, i64 nseg,
                                                              u64 *__restrict__ partial)
{
    typedef typename F::elem E;
    __shared__ u64 sh[256];
    const i64 row = blockIdx.x / nseg, seg = blockIdx.x % nseg;
    const i64 lo = col_begin + seg * seg_len;
    i64 hi = lo + seg_len;
    if (hi > n_inner) hi = n_inner;
    const T *x = in + row * n_inner;
    E acc = IS_MUL ? F::one(fd) : (E)0;
    for (i64 i = lo + threadIdx.x; i < hi; i += 256) {
        E v = (E)x[i];
        acc = IS_MUL ? F::mul(fd, acc, v) : F::add(fd, acc, v);
    }
    sh[threadIdx.x] = (u64)acc;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            E a = (E)sh[threadIdx.x], b = (E)sh[threadIdx.x + off];
            sh[threadIdx.x] = (u64)(IS_MUL ? F::mul(fd, a, b) : F::add(fd, a, b));
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// mode 0: out = reduce(partials); 1: out = a0 - sum(partials); 2: out = a0 / prod(partials)
template <class F, typename T, bool IS_MUL>
__global__ void reduce_finalize_kernel(FieldDev fd, const T *__restrict__ in, i64 n_inner, const u64 *__restrict__ partial,
                                       i64 nseg, T *__restrict__ out, i64 n_outer, int mode, int32_t *err)
{
    typedef typename F::elem E;
    const i64 row = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (row < n_outer) {
        E acc = IS_MUL ? F::one(fd) : (E)0;
        for (i64 s = 0; s < nseg; s++) {
            E v = (E)partial[row * nseg + s];
            acc = IS_MUL ? F::mul(fd, acc, v) : F::add(fd, acc, v);
        }
        if (mode == 1) acc = F::sub(fd, (E)in[row * n_inner], acc);
        if (mode == 2) {
            E a0 = (E)in[row * n_inner];
            if (acc == 0) { bad = true; acc = 0; }
            else if (a0 == 0) acc = 0;
            else {
                if constexpr (std::is_same<F, Lut>::value) acc = Lut::div_nz(fd, a0, acc);
                else acc = F::mul(fd, a0, F::inv(fd, acc));
            }
        }
        out[row] = (T)acc;
    }
    flag_error(err, bad);
}

// ufunc.reduceat: out[s] = fold(a[starts[s] : ends[s]]) with NumPy's convention that an empty or reversed slice yields
// a[starts[s]].  One 64-lane workgroup per segment; subtract / divide are left folds a0 - sum(rest), a0 / prod(rest).
template <class F, typename T, bool IS_MUL>
__global__ __launch_bounds__(64) void reduce_ragged_kernel(FieldDev fd, const T *__restrict__ in, const i64 *__restrict__ starts,
                                                           const i64 *__restrict__ ends, T *__restrict__ out, int mode, int32_t *err)
{
    typedef typename F::elem E;
    __shared__ u64 sh[64];
    const i64 lo = starts[blockIdx.x];
    i64 hi = ends[blockIdx.x];
    if (hi <= lo) hi = lo + 1;
    const i64 first = mode ? lo + 1 : lo;
    E acc = IS_MUL ? F::one(fd) : (E)0;
    for (i64 i = first + threadIdx.x; i < hi; i += 64) {
        const E v = (E)in[i];
        acc = IS_MUL ? F::mul(fd, acc, v) : F::add(fd, acc, v);
    }
    sh[threadIdx.x] = (u64)acc;
    __syncthreads();
    for (int off = 32; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const E x = (E)sh[threadIdx.x], y = (E)sh[threadIdx.x + off];
            sh[threadIdx.x] = (u64)(IS_MUL ? F::mul(fd, x, y) : F::add(fd, x, y));
        }
        __syncthreads();
    }
    bool bad = false;
    if (threadIdx.x == 0) {
        E r = (E)sh[0];
        if (mode == 1) r = F::sub(fd, (E)in[lo], r);
        if (mode == 2) {
            const E a0 = (E)in[lo];
            if (r == 0) { bad = true; r = 0; }
            else if (a0 == 0) r = 0;
            else {
                if constexpr (std::is_same<F, Lut>::value) r = Lut::div_nz(fd, a0, r);
                else r = F::mul(fd, a0, F::inv(fd, r));
            }
        }
        out[blockIdx.x] = (T)r;
    }
    flag_error(err, bad);
}

template <class F, typename T>
int launch_reduceat_ft(const FieldDev &fd, int op, const void *a, const i64 *starts, const i64 *ends, i64 nseg, void *out,
                       hipStream_t st, int32_t *err)
{
    const bool is_mul = op == GFA_OP_MUL || op == GFA_OP_DIV;
    const int mode = op == GFA_OP_SUB ? 1 : op == GFA_OP_DIV ? 2 : 0;
    if (is_mul)
        hipLaunchKernelGGL((reduce_ragged_kernel<F, T, true>), dim3((unsigned)nseg), dim3(64), 0, st, fd, (const T *)a, starts, ends,
                           (T *)out, mode, err);
    else
        hipLaunchKernelGGL((reduce_ragged_kernel<F, T, false>), dim3((unsigned)nseg), dim3(64), 0, st, fd, (const T *)a, starts, ends,
                           (T *)out, mode, err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}
int dispatch_reduceat(const FieldDev &fd, int dtype, int op, const void *a, const i64 *starts, const i64 *ends, i64 nseg, void *out,
                      hipStream_t st, int32_t *err)
{
    GFA_DISPATCH_FT(launch_reduceat_ft, fd, dtype, fd, op, a, starts, ends, nseg, out, st, err);
}

// ---- r06: streaming forms of the folds a 1-D array of 1e8 elements asks for (the generic kernel above reads one element per lane and load:
// 0.04 of the roofline for np.add.reduce over GF(2^8)) ----
// MODE 0: xor of the words (every field of characteristic 2: the fold of the elements is the fold of the packed words, folded once more
//         across the word at the end).  MODE 1: plain integer sums in 64 bits, reduced mod p once per block (prime fields, elements of at
//         most 32 bits: a block adds at most 2^32 of them).  MODE 2: np.multiply.reduce of a table field of at most 256 elements: sum of
//         LOG[x] from a 256-entry LDS table, EXP once per block, zero if any element is zero.
// Each block covers [lo, hi) of one row: 16-byte loads over the aligned middle, the unaligned head and tail element by element.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void reduce_stream_kernel(const T *__restrict__ in, i64 n_inner, i64 col_begin, i64 seg_len, i64 nseg, u64 *__restrict__ partial,
                                                            u64 p, const uint8_t *__restrict__ log8, const uint8_t *__restrict__ exp8, u32 qm1)
{
    __shared__ u64 sh[256];
    __shared__ uint8_t lg[256];
    __shared__ int any_zero;
    extern __shared__ __attribute__((aligned(16))) uint8_t rs_add8[]; // MODE 3: the field's 64 KiB sum table (log8 points at it)
    if (MODE == 3) {
        const uint4 *s0 = reinterpret_cast<const uint4 *>(log8);
        uint4 *d0 = reinterpret_cast<uint4 *>(rs_add8);
        for (int i = threadIdx.x; i < 4096; i += 256) d0[i] = s0[i];
        __syncthreads();
    }
    if (MODE == 2) {
        lg[threadIdx.x] = log8[threadIdx.x];
        if (threadIdx.x == 0) any_zero = 0;
        __syncthreads();
    }
    constexpr int V = 16 / (int)sizeof(T);
    const i64 row = blockIdx.x / nseg, seg = blockIdx.x % nseg;
    const i64 lo = col_begin + seg * seg_len;
    i64 hi = lo + seg_len;
    if (hi > n_inner) hi = n_inner;
    const T *x = in + row * n_inner;
    u64 acc = 0;
    u32 a4[4] = {0, 0, 0, 0}; // MODE 3: four chains of table additions per lane (independent gathers in flight)
    bool zero = false;
    auto one = [&](T v) {
        if (MODE == 0) acc ^= (u64)v;
        else if (MODE == 1) acc += (u64)v;
        else if (MODE == 2) { zero |= v == 0; acc += (u64)lg[(uint8_t)v]; }
        else a4[0] = rs_add8[(a4[0] << 8) | (u32)(uint8_t)v];
    };
    if (hi > lo) {
        const uintptr_t addr = reinterpret_cast<uintptr_t>(x + lo);
        i64 head = (i64)(((16 - (addr & 15)) & 15) / sizeof(T));
        if (head > hi - lo) head = hi - lo;
        const i64 a_lo = lo + head, nvec = (hi - a_lo) / V, a_hi = a_lo + nvec * V;
        if ((i64)threadIdx.x < head) one(x[lo + threadIdx.x]);
        if (a_hi + (i64)threadIdx.x < hi) one(x[a_hi + threadIdx.x]);
        const uint4 *xv = reinterpret_cast<const uint4 *>(x + a_lo);
        if (MODE == 0) {
            u32 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
            for (i64 v = threadIdx.x; v < nvec; v += 256) { const uint4 w = xv[v]; w0 ^= w.x; w1 ^= w.y; w2 ^= w.z; w3 ^= w.w; }
            u32 w = w0 ^ w1 ^ w2 ^ w3; // the elements of the four words sit at the same offsets inside a word (sizeof(T) divides 4), or T is 8 bytes
            if (sizeof(T) == 8) acc ^= ((u64)(w1 ^ w3) << 32) | (u64)(w0 ^ w2);
            else {
                if (sizeof(T) <= 2) w ^= w >> 16;
                if (sizeof(T) == 1) w ^= w >> 8;
                acc ^= (u64)(w & (sizeof(T) == 1 ? 0xffu : sizeof(T) == 2 ? 0xffffu : 0xffffffffu));
            }
        } else {
            for (i64 v = threadIdx.x; v < nvec; v += 256) {
                const uint4 w = xv[v];
                const u32 ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (sizeof(T) == 4) one((T)ww[j]);
                    else if (sizeof(T) == 2) { one((T)(ww[j] & 0xffffu)); one((T)(ww[j] >> 16)); }
                    else if (sizeof(T) == 1) {
                        if (MODE == 3) { // static chain per byte lane of the word
#pragma unroll
                            for (int b = 0; b < 4; b++) a4[b] = rs_add8[(a4[b] << 8) | ((ww[j] >> (8 * b)) & 0xffu)];
                        } else { one((T)(ww[j] & 0xffu)); one((T)((ww[j] >> 8) & 0xffu)); one((T)((ww[j] >> 16) & 0xffu)); one((T)(ww[j] >> 24)); }
                    }
                }
                if (sizeof(T) == 8) { one((T)(((u64)w.y << 32) | w.x)); one((T)(((u64)w.w << 32) | w.z)); }
            }
        }
    }
    if (MODE == 2 && zero) any_zero = 1; // (benign race: every writer stores 1)
    if (MODE == 3) {
        u32 r = rs_add8[(a4[0] << 8) | a4[1]];
        r = rs_add8[(r << 8) | a4[2]];
        acc = rs_add8[(r << 8) | a4[3]];
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            if (MODE == 3) sh[threadIdx.x] = rs_add8[(sh[threadIdx.x] << 8) | sh[threadIdx.x + off]];
            else sh[threadIdx.x] = MODE == 0 ? sh[threadIdx.x] ^ sh[threadIdx.x + off] : sh[threadIdx.x] + sh[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        u64 r = sh[0];
        if (MODE == 1) r %= p;
        if (MODE == 2) r = any_zero ? 0 : (u64)exp8[r % qm1];
        partial[blockIdx.x] = r;
    }
}

// the same finalisation with ONE WORKGROUP per row: a 1-D array is cut into up to 4096 segments, and one thread walking their partial results
// was 180 of the 196 us np.add.reduce took over 1e8 bytes (r06)
template <class F, typename T, bool IS_MUL>
__global__ __launch_bounds__(256) void reduce_finalize_block_kernel(FieldDev fd, const T *__restrict__ in, i64 n_inner, const u64 *__restrict__ partial,
                                                                    i64 nseg, T *__restrict__ out, int mode, int32_t *err)
{
    typedef typename F::elem E;
    __shared__ u64 sh[256];
    const i64 row = blockIdx.x;
    E acc = IS_MUL ? F::one(fd) : (E)0;
    for (i64 sg = threadIdx.x; sg < nseg; sg += 256) {
        const E v = (E)partial[row * nseg + sg];
        acc = IS_MUL ? F::mul(fd, acc, v) : F::add(fd, acc, v);
    }
    sh[threadIdx.x] = (u64)acc;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const E a = (E)sh[threadIdx.x], b = (E)sh[threadIdx.x + off];
            sh[threadIdx.x] = (u64)(IS_MUL ? F::mul(fd, a, b) : F::add(fd, a, b));
        }
        __syncthreads();
    }
    bool bad = false;
    if (threadIdx.x == 0) {
        acc = (E)sh[0];
        if (mode == 1) acc = F::sub(fd, (E)in[row * n_inner], acc);
        if (mode == 2) {
            const E a0 = (E)in[row * n_inner];
            if (acc == 0) { bad = true; acc = 0; }
            else if (a0 == 0) acc = 0;
            else {
                if constexpr (std::is_same<F, Lut>::value) acc = Lut::div_nz(fd, a0, acc);
                else acc = F::mul(fd, a0, F::inv(fd, acc));
            }
        }
        out[row] = (T)acc;
    }
    flag_error(err, bad);
}

struct ReduceScratch {
    u64 *p = nullptr;
    size_t n = 0;
};
ReduceScratch g_reduce_scratch[64];
// byte LOG / EXP / sum table of the field a gfa_reduce / gfa_accumulate call is for (q <= 256; null otherwise)
struct ByteTables {
    const uint8_t *log8 = nullptr, *exp8 = nullptr, *add8 = nullptr;
};

// the fold of each of nseg segments of every row into partial[row * nseg + seg]: the streaming kernels where the fold is an xor of words / an
// integer sum / a sum of byte logarithms (r06), else the generic kernel
template <class F, typename T>
void reduce_phase1(const FieldDev &fd, const ByteTables &bt, bool is_mul, const void *a, i64 n_inner, i64 col_begin, i64 seg_len, i64 nseg, i64 n_outer, u64 *partial,
                   hipStream_t st)
{
    const unsigned grid = (unsigned)(n_outer * nseg);
    int stream_mode = -1;
    if (!is_mul && fd.p == 2) stream_mode = 0;
    else if (!is_mul && fd.m == 1 && sizeof(T) <= 4 && seg_len < ((i64)1 << 32)) stream_mode = 1;
    else if (is_mul && std::is_same<F, Lut>::value && sizeof(T) == 1 && fd.q <= 256 && bt.log8 && bt.exp8 && seg_len < ((i64)1 << 40)) stream_mode = 2;
    else if (!is_mul && std::is_same<F, Lut>::value && sizeof(T) == 1 && fd.q <= 256 && fd.m > 1 && bt.add8) stream_mode = 3; // odd-characteristic table fields: the sum table in LDS
    if (stream_mode == 0)
        hipLaunchKernelGGL((reduce_stream_kernel<T, 0>), dim3(grid), dim3(256), 0, st, (const T *)a, n_inner, col_begin, seg_len, nseg, partial, (u64)fd.p, nullptr, nullptr, 0u);
    else if (stream_mode == 1)
        hipLaunchKernelGGL((reduce_stream_kernel<T, 1>), dim3(grid), dim3(256), 0, st, (const T *)a, n_inner, col_begin, seg_len, nseg, partial, (u64)fd.p, nullptr, nullptr, 0u);
    else if (stream_mode == 2)
        hipLaunchKernelGGL((reduce_stream_kernel<T, 2>), dim3(grid), dim3(256), 0, st, (const T *)a, n_inner, col_begin, seg_len, nseg, partial, (u64)fd.p, bt.log8,
                           bt.exp8, (u32)(fd.q - 1));
    else if (stream_mode == 3) {
        static bool attr = false;
        auto k3 = reduce_stream_kernel<T, 3>;
        if (!attr) { (void)hipFuncSetAttribute((const void *)k3, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); attr = true; }
        hipLaunchKernelGGL(k3, dim3(grid), dim3(256), 65536, st, (const T *)a, n_inner, col_begin, seg_len, nseg, partial, (u64)fd.p, bt.add8, nullptr, 0u);
    } else if (is_mul)
        hipLaunchKernelGGL((reduce_segments_kernel<F, T, true>), dim3(grid), dim3(256), 0, st, fd, (const T *)a, n_inner, col_begin, seg_len, nseg, partial);
    else
        hipLaunchKernelGGL((reduce_segments_kernel<F, T, false>), dim3(grid), dim3(256), 0, st, fd, (const T *)a, n_inner, col_begin, seg_len, nseg, partial);
}

template <class F, typename T>
int launch_reduce_ft(const FieldDev &fd, const ByteTables &bt, int op, const void *a, void *out, i64 n_outer, i64 n_inner, hipStream_t st,
                     int32_t *err)
{
    const bool is_mul = op == GFA_OP_MUL || op == GFA_OP_DIV;
    const int mode = op == GFA_OP_SUB ? 1 : op == GFA_OP_DIV ? 2 : 0;
    const i64 col_begin = mode ? 1 : 0;
    const i64 len = n_inner - col_begin;
    // enough segments to fill the chip when there are few rows, at least 4096 elements each
    i64 nseg = 1;
    // (the sum-table fold of reduce_phase1 stages 64 KiB per workgroup: two workgroups per CU, long segments)
    const bool tab_add = !is_mul && std::is_same<F, Lut>::value && sizeof(T) == 1 && fd.q <= 256 && fd.m > 1 && bt.add8;
    const i64 want_blocks = (i64)num_cus() * (tab_add ? 2 : 8);
    if (n_outer < want_blocks && len > 8192) {
        nseg = std::min<i64>((want_blocks + n_outer - 1) / n_outer, (len + 4095) / 4096);
        if (nseg > 4096) nseg = 4096;
    }
    if (nseg < 1) nseg = 1;
    const i64 seg_len = len > 0 ? (len + nseg - 1) / nseg : 1;
    int d = 0;
    GFA_HIP(hipGetDevice(&d));
    ReduceScratch &rs = g_reduce_scratch[d & 63];
    const size_t need = (size_t)(n_outer * nseg);
    if (rs.n < need) {
        if (rs.p) (void)hipFree(rs.p);
        rs.p = nullptr; rs.n = 0;
        GFA_HIP(hipMalloc((void **)&rs.p, need * sizeof(u64)));
        rs.n = need;
    }
    reduce_phase1<F, T>(fd, bt, is_mul, a, n_inner, col_begin, seg_len, nseg, n_outer, rs.p, st);
    if (nseg > 8) { // few rows, many segments: a workgroup per row
        if (is_mul)
            hipLaunchKernelGGL((reduce_finalize_block_kernel<F, T, true>), dim3((unsigned)n_outer), dim3(256), 0, st, fd, (const T *)a, n_inner, rs.p, nseg, (T *)out, mode, err);
        else
            hipLaunchKernelGGL((reduce_finalize_block_kernel<F, T, false>), dim3((unsigned)n_outer), dim3(256), 0, st, fd, (const T *)a, n_inner, rs.p, nseg, (T *)out, mode, err);
    } else if (is_mul)
        hipLaunchKernelGGL((reduce_finalize_kernel<F, T, true>), dim3((unsigned)((n_outer + 255) / 256)), dim3(256), 0, st, fd, (const T *)a, n_inner, rs.p, nseg,
                           (T *)out, n_outer, mode, err);
    else
        hipLaunchKernelGGL((reduce_finalize_kernel<F, T, false>), dim3((unsigned)((n_outer + 255) / 256)), dim3(256), 0, st, fd, (const T *)a, n_inner, rs.p, nseg,
                           (T *)out, n_outer, mode, err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int dispatch_reduce(const FieldDev &fd, const ByteTables &bt, int dtype, int op, const void *a, void *out, i64 n_outer, i64 n_inner,
                    hipStream_t st, int32_t *err)
{
    GFA_DISPATCH_FT(launch_reduce_ft, fd, dtype, fd, bt, op, a, out, n_outer, n_inner, st, err);
}


// np.convolve(a, b) = polynomial product, direct form (convolve_jit.implementation, _domains/_function.py:141-167):
// out[k] = sum_i a[i] * b[k - i].  One output coefficient per thread; large prime-field products go through the NTT
// on the host side (galois_amd/_ntt.py) instead.
template <class F, typename T>
__global__ __launch_bounds__(256) void convolve_kernel(FieldDev fd, const T *__restrict__ a, i64 na, const T *__restrict__ b,
                                                       i64 nb, T *__restrict__ out)
{
    typedef typename F::elem E;
    const i64 n = na + nb - 1;
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (i64)gridDim.x * blockDim.x) {
        const i64 lo = k - (nb - 1) > 0 ? k - (nb - 1) : 0;
        const i64 hi = k < na - 1 ? k : na - 1;
        E acc = 0;
        for (i64 i = lo; i <= hi; i++) acc = F::add(fd, acc, F::mul(fd, (E)a[i], (E)b[k - i]));
        out[k] = (T)acc;
    }
}

template <class F, typename T>
int launch_convolve_ft(const FieldDev &fd, const void *a, i64 na, const void *b, i64 nb, void *out, hipStream_t st)
{
    const int grid = grid_for(na + nb - 1, 256, 8);
    hipLaunchKernelGGL((convolve_kernel<F, T>), dim3(grid), dim3(256), 0, st, fd, (const T *)a, na, (const T *)b, nb, (T *)out);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int dispatch_convolve(const FieldDev &fd, int dtype, const void *a, i64 na, const void *b, i64 nb, void *out, hipStream_t st)
{
    GFA_DISPATCH_FT(launch_convolve_ft, fd, dtype, fd, a, na, b, nb, out, st);
}


// evaluate_elementwise_jit (_polys/_dense.py:432-440): y[i] = Horner(coeffs, x[i]), coefficients highest degree first.
// The coefficient index is uniform across the wave, so the compiler keeps the coefficient stream in scalar loads.
template <class F, typename T>
__global__ __launch_bounds__(256) void poly_eval_kernel(FieldDev fd, const T *__restrict__ coeffs, i64 ncoef,
                                                        const T *__restrict__ x, T *__restrict__ out, i64 n)
{
    typedef typename F::elem E;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const E xv = (E)x[i];
        E acc = (E)coeffs[0];
        for (i64 j = 1; j < ncoef; j++) acc = F::add(fd, F::mul(fd, acc, xv), (E)coeffs[j]);
        out[i] = (T)acc;
    }
}

template <class F, typename T>
int launch_poly_eval_ft(const FieldDev &fd, const void *coeffs, i64 ncoef, const void *x, void *out, i64 n, hipStream_t st)
{
    const int grid = grid_for(n, 256, 8);
    hipLaunchKernelGGL((poly_eval_kernel<F, T>), dim3(grid), dim3(256), 0, st, fd, (const T *)coeffs, ncoef, (const T *)x, (T *)out, n);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// r06: Horner's rule for the fields of at most 256 elements on uint8 arrays with the full 64 KiB PRODUCT table in LDS (row = the point x, fixed
// per lane; column = the running value: random banks) -- one LDS gather per coefficient where the generic kernel does two or three gathers
// from L2 (and two more through Zech logarithms per addition in odd characteristic; here the 64 KiB SUM table, row = the coefficient).
// Four points per lane: four independent chains cover the gather latency.  One persistent 1024-thread workgroup per CU.
template <bool ODD>
__global__ __launch_bounds__(1024) void poly_eval_tab8_kernel(const uint8_t *__restrict__ mul8, const uint8_t *__restrict__ add8, const uint8_t *__restrict__ coeffs,
                                                              i64 ncoef, const uint8_t *__restrict__ x, uint8_t *__restrict__ out, i64 n)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t pe_lds[];
    {
        const uint4 *s0 = reinterpret_cast<const uint4 *>(mul8);
        uint4 *d0 = reinterpret_cast<uint4 *>(pe_lds);
        for (int i = threadIdx.x; i < 4096; i += 1024) d0[i] = s0[i];
        if (ODD) {
            const uint4 *s1 = reinterpret_cast<const uint4 *>(add8);
            for (int i = threadIdx.x; i < 4096; i += 1024) d0[4096 + i] = s1[i];
        }
    }
    __syncthreads();
    const uint8_t *mt = pe_lds, *at = pe_lds + 65536;
    const i64 stride = (i64)gridDim.x * 1024;
    for (i64 i0 = (i64)blockIdx.x * 1024 + threadIdx.x; i0 < n; i0 += 4 * stride) {
        u32 row[4], acc[4];
        const u32 c0 = coeffs[0];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const i64 i = i0 + k * stride;
            row[k] = (i < n ? (u32)x[i] : 0u) << 8;
            acc[k] = c0;
        }
        for (i64 j = 1; j < ncoef; j++) {
            const u32 c = coeffs[j]; // uniform: a scalar load
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 prod = mt[row[k] | acc[k]];
                acc[k] = ODD ? (u32)at[(c << 8) | prod] : (prod ^ c);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const i64 i = i0 + k * stride;
            if (i < n) out[i] = (uint8_t)acc[k];
        }
    }
}

int launch_poly_eval_tab8(const uint8_t *mul8, const uint8_t *add8, bool odd, const void *coeffs, i64 ncoef, const void *x, void *out, i64 n, hipStream_t st)
{
    static bool attr[2] = {false, false};
    const size_t lds = odd ? 131072 : 65536;
    const void *k = odd ? (const void *)poly_eval_tab8_kernel<true> : (const void *)poly_eval_tab8_kernel<false>;
    if (!attr[odd]) { GFA_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr[odd] = true; }
    const i64 blocks = (n + 4095) / 4096;
    const int grid = (int)std::min<i64>(blocks, (i64)num_cus());
    if (odd)
        hipLaunchKernelGGL(poly_eval_tab8_kernel<true>, dim3(grid), dim3(1024), lds, st, mul8, add8, (const uint8_t *)coeffs, ncoef, (const uint8_t *)x, (uint8_t *)out, n);
    else
        hipLaunchKernelGGL(poly_eval_tab8_kernel<false>, dim3(grid), dim3(1024), lds, st, mul8, add8, (const uint8_t *)coeffs, ncoef, (const uint8_t *)x, (uint8_t *)out, n);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int dispatch_poly_eval(const FieldDev &fd, int dtype, const void *coeffs, i64 ncoef, const void *x, void *out, i64 n, hipStream_t st)
{
    GFA_DISPATCH_FT(launch_poly_eval_ft, fd, dtype, fd, coeffs, ncoef, x, out, n, st);
}

// log_ufunc.lookup (_domains/_lookup.py:273-294): out[i] = LOG[a[i]] for base alpha.  For another primitive element
// beta (stride-0 scalar or an array): log_beta(a) = LOG[a] * LOG[beta]^-1 mod (q - 1); a base that is not primitive has
// no inverse exponent and is flagged (the reference's search raises ArithmeticError for it, _calculate.py:617).
template <typename T>
__global__ __launch_bounds__(256) void log_lut_kernel(FieldDev fd, const T *__restrict__ a, int sa, const T *__restrict__ base,
                                                      int sb, i64 *__restrict__ out, i64 n, int32_t *err)
{
    int bad = 0;
    const i64 qm1 = (i64)fd.qm1;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const u32 av = (u32)a[sa ? i : 0];
        if (av == 0) { bad |= GFA_DEVERR_LOG_ZERO; out[i] = 0; continue; }
        i64 la = fd.log_tab[av];
        if (base) {
            const u32 bv = (u32)base[sb ? i : 0];
            if (bv == 0) { bad |= GFA_DEVERR_LOG_BASE; out[i] = 0; continue; }
            const i64 lb = fd.log_tab[bv];
            // inverse of lb modulo q - 1 by the extended Euclidean algorithm
            i64 r0 = qm1, r1 = lb % qm1, t0 = 0, t1 = 1;
            while (r1 != 0) {
                const i64 qq = r0 / r1;
                i64 tmp = r0 - qq * r1; r0 = r1; r1 = tmp;
                tmp = t0 - qq * t1; t0 = t1; t1 = tmp;
            }
            if (r0 != 1 && qm1 != 1) { bad |= GFA_DEVERR_LOG_BASE; out[i] = 0; continue; }
            if (t0 < 0) t0 += qm1;
            la = qm1 == 1 ? 0 : (i64)(((unsigned __int128)(u64)la * (u64)t0) % (u64)qm1);
        }
        out[i] = la;
    }
    if (bad && err) atomicOr((int *)err, bad);
}

// FieldArray.vector / FieldArray.Vector (_fields/_array.py:383-491): base-p digits of the integer representation, most
// significant digit (degree m-1) first.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void to_digits_kernel(u64 p, int m, const TI *__restrict__ in, TO *__restrict__ out, i64 n)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        u64 x = (u64)in[i];
        for (int j = m - 1; j >= 0; j--) {
            const u64 q = x / p;
            out[i * m + j] = (TO)(x - q * p);
            x = q;
        }
    }
}
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void from_digits_kernel(u64 p, int m, const TI *__restrict__ in, TO *__restrict__ out, i64 n)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        u64 x = 0;
        for (int j = 0; j < m; j++) x = x * p + (u64)in[i * m + j];
        out[i] = (TO)x;
    }
}

template <bool TO_DIGITS>
int launch_digits(u64 p, int m, const void *in, int dtype_in, void *out, int dtype_out, i64 n, hipStream_t st)
{
    const int grid = grid_for(n, 256, 8);
#define GFA_DG(TI, TO)                                                                                                    \
    do {                                                                                                                  \
        if (TO_DIGITS) hipLaunchKernelGGL((to_digits_kernel<TI, TO>), dim3(grid), dim3(256), 0, st, p, m, (const TI *)in, (TO *)out, n);   \
        else hipLaunchKernelGGL((from_digits_kernel<TI, TO>), dim3(grid), dim3(256), 0, st, p, m, (const TI *)in, (TO *)out, n);           \
    } while (0)
#define GFA_DG_OUT(TI)                                                                                                    \
    switch (dtype_out) {                                                                                                  \
    case GFA_U8: GFA_DG(TI, uint8_t); break;                                                                              \
    case GFA_U16: GFA_DG(TI, uint16_t); break;                                                                            \
    case GFA_U32: GFA_DG(TI, uint32_t); break;                                                                            \
    default: GFA_DG(TI, uint64_t); break;                                                                                 \
    }
    switch (dtype_in) {
    case GFA_U8: GFA_DG_OUT(uint8_t) break;
    case GFA_U16: GFA_DG_OUT(uint16_t) break;
    case GFA_U32: GFA_DG_OUT(uint32_t) break;
    default: GFA_DG_OUT(uint64_t) break;
    }
#undef GFA_DG_OUT
#undef GFA_DG
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// berlekamp_massey_jit.implementation (_lfsr.py:1647-1702): shortest LFSR (connection polynomial C, ascending) of each of
// `batch` sequences of length n.  One 64-lane workgroup per sequence, C / B / T in LDS; the discrepancy is a strided
// partial sum folded in LDS.  out_c: (batch, n) ascending coefficients, zero padded; out_len: trimmed length (>= 1).
template <class F, typename T>
__global__ __launch_bounds__(64) void berlekamp_massey_kernel(FieldDev fd, const T *__restrict__ seq, i64 n, T *__restrict__ out_c,
                                                              i64 *__restrict__ out_len)
{
    typedef typename F::elem E;
    extern __shared__ __attribute__((aligned(16))) unsigned char bm_raw[];
    E *C = reinterpret_cast<E *>(bm_raw), *B = C + n, *Tm = B + n;
    __shared__ u64 part[64];
    const T *S = seq + (i64)blockIdx.x * n;
    const int tid = threadIdx.x;
    for (i64 i = tid; i < n; i += 64) { C[i] = i == 0 ? F::one(fd) : (E)0; B[i] = C[i]; }
    __syncthreads();
    i64 L = 0, m = 1;
    E b = F::one(fd);
    for (i64 k = 0; k < n; k++) {
        E acc = 0;
        for (i64 i = tid; i <= L; i += 64) acc = F::add(fd, acc, F::mul(fd, (E)S[k - i], C[i]));
        part[tid] = (u64)acc;
        __syncthreads();
        for (int off = 32; off >= 1; off >>= 1) {
            if (tid < off) part[tid] = (u64)F::add(fd, (E)part[tid], (E)part[tid + off]);
            __syncthreads();
        }
        const E d = (E)part[0];
        __syncthreads();
        if (d == 0) { m++; continue; }
        E coef;
        if constexpr (std::is_same<F, Lut>::value) coef = Lut::div_nz(fd, d, b);
        else coef = F::mul(fd, d, F::inv(fd, b));
        const bool grow = !(2 * L > k);
        if (grow) for (i64 i = tid; i < n; i += 64) Tm[i] = C[i];
        __syncthreads();
        for (i64 i = m + tid; i < n; i += 64) C[i] = F::sub(fd, C[i], F::mul(fd, coef, B[i - m]));
        __syncthreads();
        if (grow) {
            for (i64 i = tid; i < n; i += 64) B[i] = Tm[i];
            L = k + 1 - L; b = d; m = 1;
        } else {
            m++;
        }
        __syncthreads();
    }
    // C[: L + 1], trailing zeros trimmed (at least one coefficient)
    const i64 clen = L + 1 < n ? L + 1 : n;
    if (tid == 0) {
        i64 last = 0;
        for (i64 i = 0; i < clen; i++) if (C[i] != 0) last = i;
        out_len[blockIdx.x] = last + 1;
    }
    for (i64 i = tid; i < n; i += 64) out_c[(i64)blockIdx.x * n + i] = i < clen ? (T)C[i] : (T)0;
}

template <class F, typename T>
int launch_bm_ft(const FieldDev &fd, const void *seq, i64 n, i64 batch, void *out_c, i64 *out_len, hipStream_t st)
{
    typedef typename F::elem E;
    const size_t lds = 3 * (size_t)n * sizeof(E);
    auto k = berlekamp_massey_kernel<F, T>;
    static bool attr = false;
    if (!attr) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr = true; }
    hipLaunchKernelGGL(k, dim3((unsigned)batch), dim3(64), lds, st, fd, (const T *)seq, n, (T *)out_c, out_len);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}
int dispatch_bm(const FieldDev &fd, int dtype, const void *seq, i64 n, i64 batch, void *out_c, i64 *out_len, hipStream_t st)
{
    GFA_DISPATCH_FT(launch_bm_ft, fd, dtype, fd, seq, n, batch, out_c, out_len, st);
}

// ufunc.accumulate over the last axis: one workgroup per row, 256-element chunks scanned in LDS with a running carry.
// mode 0: inclusive scan with the op; 1: out[i] = a0 - (a1 + ... + ai); 2: out[i] = a0 / (a1 * ... * ai)
// r06: a row may be cut into nseg segments of seg_len elements, one workgroup each, that start from carry_in[row * nseg + seg] -- the fold of
// everything before the segment (accumulate_carries_kernel) -- so that ONE long row fills the chip; nseg = 1, carry_in = nullptr: the whole row.
template <class F, typename T, bool IS_MUL>
__global__ __launch_bounds__(256) void accumulate_kernel(FieldDev fd, const T *__restrict__ in, T *__restrict__ out, i64 n_inner,
                                                         int mode, int32_t *err, i64 nseg, i64 seg_len, const u64 *__restrict__ carry_in)
{
    typedef typename F::elem E;
    __shared__ u64 sh[256];
    const i64 row = (i64)blockIdx.x / nseg, seg = (i64)blockIdx.x % nseg;
    const T *x = in + row * n_inner;
    T *y = out + row * n_inner;
    const E ident = IS_MUL ? F::one(fd) : (E)0;
    const E a0 = (E)x[0];
    E carry = carry_in ? (E)carry_in[blockIdx.x] : ident;
    bool bad = false;
    const i64 start = (mode ? 1 : 0) + seg * seg_len;
    i64 stop = nseg == 1 ? n_inner : start + seg_len;
    if (stop > n_inner) stop = n_inner;
    if (mode && seg == 0 && threadIdx.x == 0) y[0] = (T)a0;
    constexpr int PER = 8; // consecutive elements per thread and iteration: one LDS scan (16 barriers) per 2048 elements instead of per 256
    for (i64 base = start; base < stop; base += 256 * PER) {
        const i64 i0 = base + (i64)threadIdx.x * PER;
        E w[PER];
        E v = ident; // running fold of this thread's elements
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const E e = i0 + j < stop ? (E)x[i0 + j] : ident;
            v = IS_MUL ? F::mul(fd, v, e) : F::add(fd, v, e);
            w[j] = v; // inclusive scan inside the thread
        }
        sh[threadIdx.x] = (u64)v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            E o = ident;
            if ((int)threadIdx.x >= off) o = (E)sh[threadIdx.x - off];
            __syncthreads();
            if ((int)threadIdx.x >= off) {
                v = IS_MUL ? F::mul(fd, o, v) : F::add(fd, o, v);
                sh[threadIdx.x] = (u64)v;
            }
            __syncthreads();
        }
        const E before = threadIdx.x ? (E)sh[threadIdx.x - 1] : ident;               // the threads before this one, in this chunk
        const E lead = IS_MUL ? F::mul(fd, carry, before) : F::add(fd, carry, before); // everything before this thread's elements
        const E total = IS_MUL ? F::mul(fd, carry, (E)sh[255]) : F::add(fd, carry, (E)sh[255]);
#pragma unroll
        for (int j = 0; j < PER; j++) {
            if (i0 + j < stop) {
                const E r = IS_MUL ? F::mul(fd, lead, w[j]) : F::add(fd, lead, w[j]);
                E o = r;
                if (mode == 1) o = F::sub(fd, a0, r);
                if (mode == 2) {
                    if (r == 0) { bad = true; o = 0; }
                    else if (a0 == 0) o = 0;
                    else {
                        if constexpr (std::is_same<F, Lut>::value) o = Lut::div_nz(fd, a0, r);
                        else o = F::mul(fd, a0, F::inv(fd, r));
                    }
                }
                y[i0 + j] = (T)o;
            }
        }
        carry = total;
        __syncthreads();
    }
    flag_error(err, bad);
}

// partial[row * nseg + seg] (the fold of segment seg) -> the fold of the segments before it.  One workgroup per row: every thread folds
// its run of ceil(nseg / 256) partials, the 256 run totals are scanned in LDS, the runs are rewritten from their prefix
template <class F, bool IS_MUL>
__global__ __launch_bounds__(256) void accumulate_carries_kernel(FieldDev fd, u64 *__restrict__ partial, i64 nseg, i64 n_outer)
{
    typedef typename F::elem E;
    __shared__ u64 sh[256];
    const i64 row = blockIdx.x;
    const E ident = IS_MUL ? F::one(fd) : (E)0;
    const i64 c = (nseg + 255) / 256, lo = (i64)threadIdx.x * c;
    i64 hi = lo + c;
    if (hi > nseg) hi = nseg;
    u64 *pr = partial + row * nseg;
    E loc = ident;
    for (i64 sg = lo; sg < hi; sg++) { const E v = (E)pr[sg]; loc = IS_MUL ? F::mul(fd, loc, v) : F::add(fd, loc, v); }
    E v = loc;
    sh[threadIdx.x] = (u64)v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        E o = ident;
        if ((int)threadIdx.x >= off) o = (E)sh[threadIdx.x - off];
        __syncthreads();
        if ((int)threadIdx.x >= off) { v = IS_MUL ? F::mul(fd, o, v) : F::add(fd, o, v); sh[threadIdx.x] = (u64)v; }
        __syncthreads();
    }
    E run = threadIdx.x ? (E)sh[threadIdx.x - 1] : ident; // the fold of every run before this one
    for (i64 sg = lo; sg < hi; sg++) {
        const E x = (E)pr[sg];
        pr[sg] = (u64)run;
        run = IS_MUL ? F::mul(fd, run, x) : F::add(fd, run, x);
    }
    (void)n_outer;
}

template <class F, typename T>
int launch_accumulate_ft(const FieldDev &fd, const ByteTables &bt, int op, const void *a, void *out, i64 n_outer, i64 n_inner, hipStream_t st,
                         int32_t *err)
{
    const int mode = op == GFA_OP_SUB ? 1 : op == GFA_OP_DIV ? 2 : 0;
    const bool is_mul = op == GFA_OP_MUL || op == GFA_OP_DIV;
    // r06: few long rows (np.cumsum of a 1-D array): segment folds -> carries -> segment scans, instead of one workgroup for the whole row
    const i64 col_begin = mode ? 1 : 0, len = n_inner - col_begin;
    const i64 want_blocks = (i64)num_cus() * 8;
    if (n_outer < want_blocks / 4 && len >= ((i64)1 << 16)) {
        i64 nseg = std::min<i64>((want_blocks + n_outer - 1) / n_outer, len / 8192);
        if (nseg > 4096) nseg = 4096;
        if (nseg >= 2) {
            const i64 seg_len = ((len + nseg - 1) / nseg + 255) / 256 * 256;
            nseg = (len + seg_len - 1) / seg_len;
            int d = 0;
            GFA_HIP(hipGetDevice(&d));
            ReduceScratch &rs = g_reduce_scratch[d & 63];
            const size_t need = (size_t)(n_outer * nseg);
            if (rs.n < need) {
                if (rs.p) (void)hipFree(rs.p);
                rs.p = nullptr; rs.n = 0;
                GFA_HIP(hipMalloc((void **)&rs.p, need * sizeof(u64)));
                rs.n = need;
            }
            reduce_phase1<F, T>(fd, bt, is_mul, a, n_inner, col_begin, seg_len, nseg, n_outer, rs.p, st);
            if (is_mul) {
                hipLaunchKernelGGL((accumulate_carries_kernel<F, true>), dim3((unsigned)n_outer), dim3(256), 0, st, fd, rs.p, nseg, n_outer);
                hipLaunchKernelGGL((accumulate_kernel<F, T, true>), dim3((unsigned)(n_outer * nseg)), dim3(256), 0, st, fd, (const T *)a, (T *)out, n_inner, mode, err, nseg,
                                   seg_len, (const u64 *)rs.p);
            } else {
                hipLaunchKernelGGL((accumulate_carries_kernel<F, false>), dim3((unsigned)n_outer), dim3(256), 0, st, fd, rs.p, nseg, n_outer);
                hipLaunchKernelGGL((accumulate_kernel<F, T, false>), dim3((unsigned)(n_outer * nseg)), dim3(256), 0, st, fd, (const T *)a, (T *)out, n_inner, mode, err, nseg,
                                   seg_len, (const u64 *)rs.p);
            }
            GFA_HIP(hipGetLastError());
            return GFA_OK;
        }
    }
    if (is_mul)
        hipLaunchKernelGGL((accumulate_kernel<F, T, true>), dim3((unsigned)n_outer), dim3(256), 0, st, fd, (const T *)a, (T *)out,
                           n_inner, mode, err, (i64)1, (i64)0, (const u64 *)nullptr);
    else
        hipLaunchKernelGGL((accumulate_kernel<F, T, false>), dim3((unsigned)n_outer), dim3(256), 0, st, fd, (const T *)a, (T *)out,
                           n_inner, mode, err, (i64)1, (i64)0, (const u64 *)nullptr);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int dispatch_accumulate(const FieldDev &fd, const ByteTables &bt, int dtype, int op, const void *a, void *out, i64 n_outer, i64 n_inner,
                        hipStream_t st, int32_t *err)
{
    GFA_DISPATCH_FT(launch_accumulate_ft, fd, dtype, fd, bt, op, a, out, n_outer, n_inner, st, err);
}



int tab8_grid(i64 n)
{
    i64 blocks = ((n >> 4) + TAB8_THREADS - 1) / TAB8_THREADS;
    i64 cap = (i64)num_cus() * 2; // 64 KiB of LDS per workgroup -> two resident workgroups per CU
    if (blocks < 1) blocks = 1;
    return (int)(blocks < cap ? blocks : cap);
}

int launch_tab8_binary(const uint8_t *table, bool check_zero_b, const void *a, const void *b, void *out, i64 n,
                       hipStream_t st, int32_t *err)
{
    const int grid = tab8_grid(n);
    static bool attr[4] = {false, false, false, false};
    if (n >= ((i64)1 << 28)) { // operands beyond the Infinity Cache: claimed blocks (see tab8_binary_claim_kernel)
        unsigned int *counter = nullptr;
        GFA_HIP(gfa::scratch_alloc((void **)&counter, sizeof(unsigned int), st));
        GFA_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned int), st));
        if (check_zero_b) {
            auto k = tab8_binary_claim_kernel<true>;
            if (!attr[3]) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); attr[3] = true; }
            hipLaunchKernelGGL(k, dim3(grid), dim3(TAB8_THREADS), 65536, st, table, (const uint8_t *)a, (const uint8_t *)b,
                               (uint8_t *)out, n, err, counter);
        } else {
            auto k = tab8_binary_claim_kernel<false>;
            if (!attr[2]) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); attr[2] = true; }
            hipLaunchKernelGGL(k, dim3(grid), dim3(TAB8_THREADS), 65536, st, table, (const uint8_t *)a, (const uint8_t *)b,
                               (uint8_t *)out, n, err, counter);
        }
        GFA_HIP(hipGetLastError());
        GFA_HIP(gfa::scratch_free(counter, st));
        return GFA_OK;
    }
    if (check_zero_b) {
        auto k = tab8_binary_kernel<true>;
        if (!attr[1]) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); attr[1] = true; }
        hipLaunchKernelGGL(k, dim3(grid), dim3(TAB8_THREADS), 65536, st, table, (const uint8_t *)a, (const uint8_t *)b,
                           (uint8_t *)out, n, err);
    } else {
        auto k = tab8_binary_kernel<false>;
        if (!attr[0]) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); attr[0] = true; }
        hipLaunchKernelGGL(k, dim3(grid), dim3(TAB8_THREADS), 65536, st, table, (const uint8_t *)a, (const uint8_t *)b,
                           (uint8_t *)out, n, err);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int launch_tab8_unary(const uint8_t *table256, bool check_zero, const void *a, void *out, i64 n, hipStream_t st,
                      int32_t *err)
{
    // Arrays far beyond the Infinity Cache: statically strided persistent workgroups drift apart over a long launch and the
    // DRAM page set spreads (section 4 of DESIGN.md); consecutive launches over 2^26-element slices keep them in step
    // (reciprocal of 1e9 elements: 5.07 -> 5.27 TB/s, tools/unary_big.py).
    constexpr i64 slice = (i64)1 << 26;
    if (slice > 0 && n >= ((i64)1 << 28)) {
        for (i64 o = 0; o < n; o += slice) {
            const int rc = launch_tab8_unary(table256, check_zero, (const uint8_t *)a + o, (uint8_t *)out + o, std::min(slice, n - o), st, err);
            if (rc) return rc;
        }
        return GFA_OK;
    }
    i64 blocks = ((n >> 4) + TAB8_THREADS - 1) / TAB8_THREADS;
    i64 cap = (i64)num_cus() * 2;
    if (blocks < 1) blocks = 1;
    const int grid = (int)(blocks < cap ? blocks : cap);
    if (check_zero)
        hipLaunchKernelGGL((tab8_unary_kernel<true>), dim3(grid), dim3(TAB8_THREADS), 0, st, table256,
                           (const uint8_t *)a, (uint8_t *)out, n, err);
    else
        hipLaunchKernelGGL((tab8_unary_kernel<false>), dim3(grid), dim3(TAB8_THREADS), 0, st, table256,
                           (const uint8_t *)a, (uint8_t *)out, n, err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// x ** k on uint8 storage, one exponent for the whole array (device memory): tab8_unary_kernel<true, true>, one launch
int launch_pow8(const FieldDev &lut, const void *a, const i64 *e, void *out, i64 n, hipStream_t st, int32_t *err)
{
    i64 blocks = ((n >> 4) + TAB8_THREADS - 1) / TAB8_THREADS;
    const i64 cap = (i64)num_cus() * 2;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((tab8_unary_kernel<true, true>), dim3((int)(blocks < cap ? blocks : cap)), dim3(TAB8_THREADS), 0, st,
                       (const uint8_t *)nullptr, (const uint8_t *)a, (uint8_t *)out, n, err, lut, e);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// 8192 < q <= 65536: do products go through the two-phase LOG / EXP kernel?  (division / reciprocal / power always do)
inline bool big16_products(const gfa_field *f) { return f->use_lookup(); }

} // namespace

namespace gfa { // gfa_elementwise_mid.hip, gfa_elementwise_packed.hip
int big16_power_each(const FieldDev &lut, const void *image, const void *a, const i64 *e, void *out, i64 n, hipStream_t st, int32_t *err);
int pow24_run(const u32 *exp_tab, const u32 *log_tab, u64 q, const void *a, const i64 *e_ptr, void *out, i64 n, hipStream_t st, int *dev_err);
}

extern "C" {

int gfa_binary(gfa_field_t *f, int op, const void *a, int64_t sa, const void *b, int64_t sb, void *out, int64_t n,
               int dtype, gfa_stream_t stream, int32_t *dev_err)
{
    if (f && n == 0) return GFA_OK; // empty arrays: nothing to launch (pointers may be NULL)
    if (!f || !a || !b || !out || n < 0 || (sa != 0 && sa != 1) || (sb != 0 && sb != 1)) {
        set_error("gfa_binary: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (op < GFA_OP_ADD || op > GFA_OP_DIV) { set_error("gfa_binary: bad op"); return GFA_ERR_INVALID; }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    hipStream_t st = (hipStream_t)stream;
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    // r05: sums and differences of odd-characteristic extension fields whose Zech tables leave LDS (8192 < q <= 2^20), in either mode:
    // packed base-p digits (gfa_elementwise_packed.hip) -- same values as the table and the digit-vector routes
    const bool pinned = f->mode == GFA_MODE_CALCULATE;
    if ((op == GFA_OP_ADD || op == GFA_OP_SUB) && packed_eligible(f->calc, dtype, n, pinned)) {
        rc = packed_run(f->calc, dtype, op, a, sa, b, sb, out, n, st);
        if (rc != GFA_ERR_UNSUPPORTED) return rc;
    }
    // a field pinned to explicit calculation: its products on the digit tables too, whatever its order (GF(3^5), GF(3^7), ...)
    if (pinned && op == GFA_OP_MUL && f->calc.kind == KIND_EXT && packed_mul_eligible(f->calc, dtype, n, true)) {
        rc = packed_mul_run(f->calc, dtype, a, sa, b, sb, out, n, st);
        if (rc != GFA_ERR_UNSUPPORTED) return rc;
    }
    // r06, AUTO only: GF(2^17) .. GF(2^20) quotients: one gather from the 3-byte inverse table, then the carry-less product (0.12 -> see
    // profiles/r06_ew_bin_inverse_table.txt); a field pinned to either mode keeps what it asked for
    if (f->mode == GFA_MODE_AUTO && op == GFA_OP_DIV && f->calc.kind == KIND_BIN && dtype == GFA_U32 && f->has_lut && f->calc.m >= 17 && n >= 1024) {
        const uint8_t *inv24 = nullptr;
        if ((rc = f->inverse_table(*ds, &inv24))) return rc;
        rc = bin32_div_by_table(f->calc, inv24, a, sa, b, sb, out, n, st, dev_err);
        if (rc != GFA_ERR_UNSUPPORTED) return rc;
    }
    // r06, AUTO only: GF(p^m), p odd, 32768 < q <= 65536 (GF(181^2) .. GF(251^2), GF(37^3)): LOG / EXP no longer fit LDS together, the staged
    // kernels below run products and quotients at 0.40 (uint16) / 0.27 (uint32); the digit tables of gfa_packed.h stream
    // (measured, profiles/r06_ew_band16.txt: uint32 arrays -- products 0.27 -> 0.68, GF(p^2) quotients 0.27 -> 0.54; uint16 arrays -- GF(p^2)
    // products 0.40 -> 0.54, but cubic products 0.37 and quotients by the norm 0.27 LOSE to the staged tables' 0.40: those stay there)
    if (f->mode == GFA_MODE_AUTO && f->calc.kind == KIND_EXT && (f->calc.p & 1) && f->calc.q > 32768 && f->calc.q <= 65536) {
        if (op == GFA_OP_MUL && (dtype == GFA_U32 || f->calc.m == 2) && packed_mul_eligible(f->calc, dtype, n, false)) {
            rc = packed_mul_run(f->calc, dtype, a, sa, b, sb, out, n, st);
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
        if (op == GFA_OP_DIV && dtype == GFA_U32 && packed_divn_eligible(f->calc, dtype, n)) {
            rc = packed_divn_run(f->calc, dtype, a, sa, b, sb, out, n, st, dev_err);
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
    }
    if (f->mode != GFA_MODE_CALCULATE) {
        // uint16 storage, tables in LDS (gfa_elementwise_mid.hip), unless the field was pinned to explicit calculation:
        //  * 256 < q <= 32768 (LOG and EXP both resident): every operation that is not a plain xor / modular add (measured 0.7-0.8
        //    of the HBM roofline, above the packed shift-and-xor product and the Montgomery-trick division);
        //  * 32768 < q <= 65536, LOG / (ZECH /) EXP staged in turn: division always (it is an exponentiation otherwise); products,
        //    and sums in odd characteristic, when the field is in lookup mode.
        const FieldDev &c = f->calc;
        const bool trivial_addsub = (op == GFA_OP_ADD || op == GFA_OP_SUB) && (c.p == 2 || c.m == 1);
        const bool zech_op = op == GFA_OP_ADD || op == GFA_OP_SUB;
        // uint32 / int64 storage: products of prime and binary fields stay on the calculated kernels unless the field is in lookup
        // mode (measured at 50 M elements, profiles/r03_ew_widestore.txt: GF(7919) uint32 0.74 vs 0.66, GF(2^12) int64 0.76 vs 0.68
        // of the roofline); everything else -- division, reciprocal, power, GF(p^m) sums and products -- gains 1.4-3.5x from LDS
        const bool wide_calc_mul = dtype != GFA_U16 && op == GFA_OP_MUL && !f->use_lookup() && (c.m == 1 || c.p == 2);
        if (!trivial_addsub && !wide_calc_mul && mid_eligible(c, ds->mid16, dtype, n) && (!zech_op || mid_has_zech_room(c))) {
            rc = mid_binary(f->lut_desc(*ds), ds->mid16, dtype, op, a, sa, b, sb, out, n, st, dev_err);
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
        // GF(2^16) products: the integer-multiply carry-less kernel (bin16_holes_mul_kernel, 0.69 of the roofline) beats the staged
        // LOG / EXP tables (0.35) whatever mode the field is in -- same values
        if (op == GFA_OP_MUL && c.p == 2 && c.m == 16 && dtype == GFA_U16 && f->calc.kind == KIND_BIN)
            return dispatch_binary(f->calc, dtype, op, a, sa, b, sb, out, n, st, dev_err);
        // (reached for 8192 < q <= 32768 only by the sums / differences whose three tables do not fit in LDS together)
        if ((op == GFA_OP_DIV || (!trivial_addsub && big16_products(f))) && big16_eligible(c, ds->mid16, dtype, n)) {
            rc = big16_run(f->lut_desc(*ds), ds->mid16, op, a, sa, b, sb, nullptr, out, n, st, dev_err);
            if (rc == GFA_OK && (n & 7)) { // the last n & 7 elements
                const i64 o = n & ~(i64)7;
                const FieldDev td = f->use_lookup() ? f->lut_desc(*ds) : f->calc;
                return dispatch_binary(td, dtype, op, (const uint16_t *)a + (sa ? o : 0), sa, (const uint16_t *)b + (sb ? o : 0), sb,
                                       (uint16_t *)out + o, n & 7, st, dev_err);
            }
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
        // the same fields held as uint32 / int64 arrays, lookup mode (calculate mode keeps the arithmetic kernels, which stream)
        if (f->use_lookup() && !trivial_addsub && !(op == GFA_OP_MUL && c.p == 2 && c.m == 16) && big16_wide_eligible(c, ds->mid16, dtype, n)) {
            rc = big16_run_wide(f->lut_desc(*ds), ds->mid16, dtype, op, a, sa, b, sb, nullptr, out, n, st, dev_err);
            if (rc == GFA_OK && (n & 7)) {
                const i64 o = n & ~(i64)7, es = dtype == GFA_U32 ? 4 : 8;
                return dispatch_binary(f->lut_desc(*ds), dtype, op, (const char *)a + (sa ? o * es : 0), sa, (const char *)b + (sb ? o * es : 0), sb,
                                       (char *)out + o * es, n & 7, st, dev_err);
            }
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
    }
    // r05, AUTO only: odd-characteristic extension fields above 65536 elements keep their EXP / LOG tables in L2, and a product through
    // them is three dependent gathers (0.06-0.10 of the roofline); the digit-vector kernels of degree <= 8 beat that for products --
    // GF(997^2) 0.67, GF(97^3) 0.47, GF(31^4) 0.33, GF(13^5) 0.25, GF(7^7) 0.15, GF(5^8) 0.13 (profiles/r05_ew_extcalc.txt) -- and, in
    // degree 2, for quotients (0.15 vs 0.06).  Same values either way; a field pinned to jit-lookup keeps its tables.
    if (f->mode == GFA_MODE_AUTO && f->calc.m > 1 && (f->calc.p & 1) && f->calc.q > 65536 && f->calc.kind == KIND_EXT && Ext::fixed_degree(f->calc) &&
        (op == GFA_OP_MUL || (op == GFA_OP_DIV && (f->calc.m == 2 || (f->calc.m == 3 && packed_divn_eligible(f->calc, dtype, n)) ||
                                                   (f->has_lut && packed_divt_eligible(f->calc, dtype, n)))))) {
        // products: digits through LDS tables and no reduction before the end (gfa_packed.h::mul_digits) where the 32-bit bound holds
        if (op == GFA_OP_MUL && packed_mul_eligible(f->calc, dtype, n, false)) {
            rc = packed_mul_run(f->calc, dtype, a, sa, b, sb, out, n, st);
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
        // r06: degree-2 quotients by the norm (conjugate / N(b), 1 / N from a p-entry LDS table): 0.15 -> see profiles/r06_ew_div2.txt
        if (op == GFA_OP_DIV && packed_divn_eligible(f->calc, dtype, n)) {
            rc = packed_divn_run(f->calc, dtype, a, sa, b, sb, out, n, st, dev_err);
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
        // r06: degrees 4 .. 8: 1 / b by one gather from the 3-byte inverse table, then the digit-table product (profiles/r06_ew_divt.txt)
        if (op == GFA_OP_DIV && f->calc.m >= 4 && f->has_lut && packed_divt_eligible(f->calc, dtype, n)) {
            const uint8_t *inv24 = nullptr;
            if ((rc = f->inverse_table(*ds, &inv24))) return rc;
            rc = packed_divt_run(f->calc, inv24, a, sa, b, sb, out, n, st, dev_err);
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
            return dispatch_binary(f->lut_desc(*ds), dtype, op, a, sa, b, sb, out, n, st, dev_err);
        }
        return dispatch_binary(f->calc, dtype, op, a, sa, b, sb, out, n, st, dev_err);
    }
    if (f->use_lookup()) {
        const FieldDev &c = f->calc;
        const bool trivial_addsub = (op == GFA_OP_ADD || op == GFA_OP_SUB) && (c.p == 2 || c.m == 1);
        if (f->has_tab8 && dtype == GFA_U8 && sa == 1 && sb == 1 && !trivial_addsub && aligned16(a) && aligned16(b) &&
            aligned16(out)) {
            const uint8_t *tab = op == GFA_OP_MUL ? ds->mul8 : op == GFA_OP_DIV ? ds->div8 : op == GFA_OP_ADD ? ds->add8 : ds->sub8;
            return launch_tab8_binary(tab, op == GFA_OP_DIV, a, b, out, n, st, dev_err);
        }
        return dispatch_binary(f->lut_desc(*ds), dtype, op, a, sa, b, sb, out, n, st, dev_err);
    }
    return dispatch_binary(f->calc, dtype, op, a, sa, b, sb, out, n, st, dev_err);
}

int gfa_unary(gfa_field_t *f, int op, const void *a, void *out, int64_t n, int dtype, gfa_stream_t stream,
              int32_t *dev_err)
{
    if (f && n == 0) return GFA_OK;
    if (!f || !a || !out || n < 0) { set_error("gfa_unary: bad arguments"); return GFA_ERR_INVALID; }
    if (op != GFA_OP_NEG && op != GFA_OP_RECIP) { set_error("gfa_unary: bad op"); return GFA_ERR_INVALID; }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    hipStream_t st = (hipStream_t)stream;
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    if (op == GFA_OP_NEG && packed_eligible(f->calc, dtype, n, f->mode == GFA_MODE_CALCULATE)) { // r05: packed base-p digits, as in gfa_binary
        rc = packed_run(f->calc, dtype, GFA_OP_NEG, a, 1, nullptr, 0, out, n, st);
        if (rc != GFA_ERR_UNSUPPORTED) return rc;
    }
    if (f->mode == GFA_MODE_AUTO && op == GFA_OP_RECIP && f->calc.kind == KIND_EXT && f->calc.q > 32768 && f->calc.q <= 65536 && dtype == GFA_U32 &&
        packed_divn_eligible(f->calc, dtype, n)) { // r06: GF(p^2), 181 <= p <= 251: by the norm, as in gfa_binary
        rc = packed_divn_run(f->calc, dtype, nullptr, 0, a, 1, out, n, st, dev_err);
        if (rc != GFA_ERR_UNSUPPORTED) return rc;
    }
    if (f->mode != GFA_MODE_CALCULATE) { // tables in LDS, as in gfa_binary
        const FieldDev &c = f->calc;
        const bool trivial_neg = op == GFA_OP_NEG && (c.p == 2 || c.m == 1);
        if (!trivial_neg && mid_eligible(c, ds->mid16, dtype, n) && (op != GFA_OP_NEG || mid_has_zech_room(c))) {
            rc = mid_unary(f->lut_desc(*ds), ds->mid16, dtype, op, a, out, n, st, dev_err);
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
        if ((op == GFA_OP_RECIP || (!trivial_neg && big16_products(f))) && big16_eligible(c, ds->mid16, dtype, n)) {
            rc = big16_run(f->lut_desc(*ds), ds->mid16, op, a, 1, nullptr, 0, nullptr, out, n, st, dev_err);
            if (rc == GFA_OK && (n & 7)) {
                const i64 o = n & ~(i64)7;
                const FieldDev td = f->use_lookup() ? f->lut_desc(*ds) : f->calc;
                return dispatch_unary(td, dtype, op, (const uint16_t *)a + o, (uint16_t *)out + o, n & 7, st, dev_err);
            }
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
        if (f->use_lookup() && !trivial_neg && big16_wide_eligible(c, ds->mid16, dtype, n)) {
            rc = big16_run_wide(f->lut_desc(*ds), ds->mid16, dtype, op, a, 1, nullptr, 0, nullptr, out, n, st, dev_err);
            if (rc == GFA_OK && (n & 7)) {
                const i64 o = n & ~(i64)7, es = dtype == GFA_U32 ? 4 : 8;
                return dispatch_unary(f->lut_desc(*ds), dtype, op, (const char *)a + o * es, (char *)out + o * es, n & 7, st, dev_err);
            }
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
    }
    // r05, AUTO only: reciprocals of degree-2 extension fields above 65536 elements on the digit-vector kernel (0.12 vs 0.06, see gfa_binary)
    if (f->mode == GFA_MODE_AUTO && op == GFA_OP_RECIP && (f->calc.m == 2 || (f->calc.m == 3 && packed_divn_eligible(f->calc, dtype, n))) && (f->calc.p & 1) &&
        f->calc.q > 65536 && f->calc.kind == KIND_EXT && Ext::fixed_degree(f->calc)) {
        if (packed_divn_eligible(f->calc, dtype, n)) { // r06: by the norm
            rc = packed_divn_run(f->calc, dtype, nullptr, 0, a, 1, out, n, st, dev_err);
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
        return dispatch_unary(f->calc, dtype, op, a, out, n, st, dev_err);
    }
    if (f->mode == GFA_MODE_AUTO && op == GFA_OP_RECIP && ((f->calc.kind == KIND_EXT && (f->calc.p & 1) && f->calc.m >= 4) || f->calc.kind == KIND_BIN) && f->has_lut && f->calc.q > 65536 &&
        dtype == GFA_U32 && n >= 1024) { // r06: one gather from the 3-byte inverse table where LOG + EXP are two
        const uint8_t *inv24 = nullptr;
        if ((rc = f->inverse_table(*ds, &inv24))) return rc;
        rc = packed_divt_run(f->calc, inv24, nullptr, 0, a, 1, out, n, st, dev_err);
        if (rc != GFA_ERR_UNSUPPORTED) return rc;
    }
    if (f->use_lookup()) {
        const FieldDev &c = f->calc;
        const bool trivial_neg = op == GFA_OP_NEG && (c.p == 2 || c.m == 1);
        if (f->has_tab8 && dtype == GFA_U8 && !trivial_neg && aligned16(a) && aligned16(out))
            return launch_tab8_unary(op == GFA_OP_RECIP ? ds->inv8 : ds->neg8, op == GFA_OP_RECIP, a, out, n, st, dev_err);
        return dispatch_unary(f->lut_desc(*ds), dtype, op, a, out, n, st, dev_err);
    }
    return dispatch_unary(f->calc, dtype, op, a, out, n, st, dev_err);
}

int gfa_power(gfa_field_t *f, const void *a, int64_t sa, const int64_t *exps, int64_t se, void *out, int64_t n, int dtype,
              gfa_stream_t stream, int32_t *dev_err)
{
    if (f && n == 0) return GFA_OK;
    if (!f || !a || !exps || !out || n < 0 || (sa != 0 && sa != 1) || (se != 0 && se != 1)) {
        set_error("gfa_power: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    if (sa == 1 && se == 0 && dtype == GFA_U8 && f->calc.q <= 256 && n >= 4096 && aligned16(a) && aligned16(out) && f->use_lookup()) {
        // one exponent, at most 256 field values: a 256-entry map applied at streaming speed (a field pinned to explicit
        // calculation keeps computing every element)
        return launch_pow8(f->lut_desc(*ds), a, exps, out, n, (hipStream_t)stream, dev_err);
    }
    if (sa == 1 && f->mode != GFA_MODE_CALCULATE) { // tables in LDS, as in gfa_binary
        if (mid_eligible(f->calc, ds->mid16, dtype, n)) {
            rc = se == 0 ? mid_power(f->lut_desc(*ds), ds->mid16, dtype, a, exps, out, n, (hipStream_t)stream, dev_err)
                         : (dtype == GFA_U16 ? mid_power_each(f->lut_desc(*ds), ds->mid16, a, exps, out, n, (hipStream_t)stream, dev_err)
                                             : GFA_ERR_UNSUPPORTED);
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
        if (se == 1 && big16_eligible(f->calc, ds->mid16, dtype, n) && f->calc.q > 32768) { // r06: an exponent per element: LOG pass + EXP pass
            rc = big16_power_each(f->lut_desc(*ds), ds->mid16, a, exps, out, n, (hipStream_t)stream, dev_err);
            if (rc == GFA_OK && (n & 7)) {
                const i64 o = n & ~(i64)7;
                const FieldDev td = f->use_lookup() ? f->lut_desc(*ds) : f->calc;
                return dispatch_intarg(td, dtype, true, (const uint16_t *)a + o, 1, exps + o, 1, (uint16_t *)out + o, n & 7, (hipStream_t)stream, dev_err);
            }
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
        if (se == 0 && big16_eligible(f->calc, ds->mid16, dtype, n)) {
            rc = big16_run(f->lut_desc(*ds), ds->mid16, GFA_OP_POW, a, 1, nullptr, 0, exps, out, n, (hipStream_t)stream, dev_err);
            if (rc == GFA_OK && (n & 7)) {
                const i64 o = n & ~(i64)7;
                const FieldDev td = f->use_lookup() ? f->lut_desc(*ds) : f->calc;
                return dispatch_intarg(td, dtype, true, (const uint16_t *)a + o, 1, exps, 0, (uint16_t *)out + o, n & 7, (hipStream_t)stream, dev_err);
            }
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
    }
    // GF(2^m), m <= 16, in AUTO: square-and-multiply over shift-and-xor products is far behind two table gathers (14 vs 86 Gop/s)
    const bool bin_tables_ = f->mode == GFA_MODE_AUTO && f->has_lut && f->calc.kind == KIND_BIN;
    if ((f->use_lookup() || bin_tables_) && sa == 1 && se == 0 && dtype == GFA_U32 && f->calc.q > 65536) { // r06: one gather from a per-call table of x ** e
        rc = pow24_run(ds->exp_tab, ds->log_tab, f->calc.q, a, exps, out, n, (hipStream_t)stream, dev_err);
        if (rc != GFA_ERR_UNSUPPORTED) return rc;
    }
    const bool bin_tables = f->mode == GFA_MODE_AUTO && f->has_lut && f->calc.kind == KIND_BIN; // r06: also GF(2^17) .. GF(2^20) (17 -> see profiles/r06_ew_bin_inverse_table.txt)
    if (f->use_lookup() || bin_tables)
        return dispatch_intarg(f->lut_desc(*ds), dtype, true, a, sa, exps, se, out, n, (hipStream_t)stream, dev_err);
    return dispatch_intarg(f->calc, dtype, true, a, sa, exps, se, out, n, (hipStream_t)stream, dev_err);
}

int gfa_scalar_multiply(gfa_field_t *f, const void *a, int64_t sa, const int64_t *ks, int64_t sk, void *out, int64_t n,
                        int dtype, gfa_stream_t stream)
{
    if (f && n == 0) return GFA_OK;
    if (!f || !a || !ks || !out || n < 0 || (sa != 0 && sa != 1) || (sk != 0 && sk != 1)) {
        set_error("gfa_scalar_multiply: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    if (f->use_lookup())
        return dispatch_intarg(f->lut_desc(*ds), dtype, false, a, sa, ks, sk, out, n, (hipStream_t)stream, nullptr);
    return dispatch_intarg(f->calc, dtype, false, a, sa, ks, sk, out, n, (hipStream_t)stream, nullptr);
}

int gfa_reduce(gfa_field_t *f, int op, const void *a, void *out, int64_t n_outer, int64_t n_inner, int dtype,
               gfa_stream_t stream, int32_t *dev_err)
{
    if (!f || !a || !out || n_outer < 0 || n_inner < 1 || op < GFA_OP_ADD || op > GFA_OP_DIV) {
        set_error("gfa_reduce: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    if (n_outer == 0) return GFA_OK;
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    ByteTables bt;
    bt.log8 = ds->log8; bt.exp8 = ds->exp8; bt.add8 = ds->add8;
    if (f->use_lookup()) return dispatch_reduce(f->lut_desc(*ds), bt, dtype, op, a, out, n_outer, n_inner, (hipStream_t)stream, dev_err);
    return dispatch_reduce(f->calc, bt, dtype, op, a, out, n_outer, n_inner, (hipStream_t)stream, dev_err);
}

int gfa_reduceat(gfa_field_t *f, int op, const void *a, const int64_t *starts, const int64_t *ends, int64_t nseg, void *out, int dtype,
                 gfa_stream_t stream, int32_t *dev_err)
{
    if (!f || nseg < 0 || op < GFA_OP_ADD || op > GFA_OP_DIV) { set_error("gfa_reduceat: bad arguments"); return GFA_ERR_INVALID; }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    if (nseg == 0) return GFA_OK;
    if (!a || !starts || !ends || !out || nseg > 0x7fffffff) { set_error("gfa_reduceat: bad arguments"); return GFA_ERR_INVALID; }
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    if (f->use_lookup())
        return dispatch_reduceat(f->lut_desc(*ds), dtype, op, a, (const i64 *)starts, (const i64 *)ends, nseg, out, (hipStream_t)stream, dev_err);
    return dispatch_reduceat(f->calc, dtype, op, a, (const i64 *)starts, (const i64 *)ends, nseg, out, (hipStream_t)stream, dev_err);
}

int gfa_accumulate(gfa_field_t *f, int op, const void *a, void *out, int64_t n_outer, int64_t n_inner, int dtype,
                   gfa_stream_t stream, int32_t *dev_err)
{
    if (!f || n_outer < 0 || n_inner < 0 || op < GFA_OP_ADD || op > GFA_OP_DIV) {
        set_error("gfa_accumulate: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (n_outer == 0 || n_inner == 0) return GFA_OK;
    if (!a || !out) { set_error("gfa_accumulate: bad arguments"); return GFA_ERR_INVALID; }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    ByteTables bt;
    bt.log8 = ds->log8; bt.exp8 = ds->exp8; bt.add8 = ds->add8;
    if (f->use_lookup()) return dispatch_accumulate(f->lut_desc(*ds), bt, dtype, op, a, out, n_outer, n_inner, (hipStream_t)stream, dev_err);
    return dispatch_accumulate(f->calc, bt, dtype, op, a, out, n_outer, n_inner, (hipStream_t)stream, dev_err);
}

int gfa_convolve(gfa_field_t *f, const void *a, int64_t na, const void *b, int64_t nb, void *out, int dtype,
                 gfa_stream_t stream)
{
    if (!f || !a || !b || !out || na < 1 || nb < 1) { set_error("gfa_convolve: bad arguments"); return GFA_ERR_INVALID; }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    if (convolve_crt_eligible(f->calc, na, nb)) return convolve_crt(f, dtype, a, na, b, nb, out, (hipStream_t)stream);
    if (f->use_lookup()) return dispatch_convolve(f->lut_desc(*ds), dtype, a, na, b, nb, out, (hipStream_t)stream);
    return dispatch_convolve(f->calc, dtype, a, na, b, nb, out, (hipStream_t)stream);
}

int gfa_berlekamp_massey(gfa_field_t *f, const void *seq, int64_t n, int64_t batch, void *out_coeffs, int64_t *out_len, int dtype,
                          gfa_stream_t stream)
{
    if (!f || n < 1 || batch < 0) { set_error("gfa_berlekamp_massey: bad arguments"); return GFA_ERR_INVALID; }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    if (batch == 0) return GFA_OK;
    if (!seq || !out_coeffs || !out_len) { set_error("gfa_berlekamp_massey: bad arguments"); return GFA_ERR_INVALID; }
    if (n > 6000 || batch > 0x7fffffff) { set_error("gfa_berlekamp_massey: sequences are limited to 6000 terms"); return GFA_ERR_UNSUPPORTED; }
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    if (f->use_lookup()) return dispatch_bm(f->lut_desc(*ds), dtype, seq, n, batch, out_coeffs, (i64 *)out_len, (hipStream_t)stream);
    return dispatch_bm(f->calc, dtype, seq, n, batch, out_coeffs, (i64 *)out_len, (hipStream_t)stream);
}

int gfa_vector(gfa_field_t *f, int to_digits, const void *in, int dtype_in, void *out, int dtype_out, int64_t n, gfa_stream_t stream)
{
    if (!f || n < 0 || dtype_in < GFA_U8 || dtype_in > GFA_U64 || dtype_out < GFA_U8 || dtype_out > GFA_U64) {
        set_error("gfa_vector: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (!dtype_holds(to_digits ? dtype_in : dtype_out, f->calc.q) || !dtype_holds(to_digits ? dtype_out : dtype_in, f->calc.p)) {
        set_error("gfa_vector: dtype cannot hold the elements");
        return GFA_ERR_INVALID;
    }
    if (n == 0) return GFA_OK;
    if (!in || !out) { set_error("gfa_vector: bad arguments"); return GFA_ERR_INVALID; }
    if (to_digits) return launch_digits<true>(f->calc.p, (int)f->calc.m, in, dtype_in, out, dtype_out, n, (hipStream_t)stream);
    return launch_digits<false>(f->calc.p, (int)f->calc.m, in, dtype_in, out, dtype_out, n, (hipStream_t)stream);
}

int gfa_poly_evaluate(gfa_field_t *f, const void *coeffs, int64_t ncoef, const void *x, void *out, int64_t n, int dtype,
                      gfa_stream_t stream)
{
    if (!f || !coeffs || ncoef < 1 || n < 0) { set_error("gfa_poly_evaluate: bad arguments"); return GFA_ERR_INVALID; }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    if (!x || !out) { set_error("gfa_poly_evaluate: bad arguments"); return GFA_ERR_INVALID; }
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    if (f->has_tab8 && f->use_lookup() && dtype == GFA_U8 && ds->mul8 && (f->calc.p == 2 || ds->add8) && n >= 65536 && ncoef >= 4) // r06: product (and sum) table in LDS
        return launch_poly_eval_tab8(ds->mul8, ds->add8, f->calc.p != 2, coeffs, ncoef, x, out, n, (hipStream_t)stream);
    if (f->use_lookup()) return dispatch_poly_eval(f->lut_desc(*ds), dtype, coeffs, ncoef, x, out, n, (hipStream_t)stream);
    return dispatch_poly_eval(f->calc, dtype, coeffs, ncoef, x, out, n, (hipStream_t)stream);
}

int gfa_log(gfa_field_t *f, const void *a, int64_t a_stride, const void *base, int64_t base_stride, int64_t *out, int64_t n,
            int dtype, gfa_stream_t stream, int32_t *dev_err)
{
    if (!f || n < 0 || (a_stride != 0 && a_stride != 1) || (base_stride != 0 && base_stride != 1)) {
        set_error("gfa_log: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    if (n == 0) return GFA_OK;
    if (!a || !out) { set_error("gfa_log: bad arguments"); return GFA_ERR_INVALID; }
    if (!f->has_lut) // no LOG table: Pohlig-Hellman with baby-step / giant-step digits (gfa_dlog.hip)
        return dlog_run(f, a, a_stride, base, base_stride, out, n, dtype, (hipStream_t)stream, dev_err);
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    const FieldDev fd = f->lut_desc(*ds);
    const int grid = grid_for(n, 256, 8);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case GFA_U8: hipLaunchKernelGGL(log_lut_kernel<uint8_t>, dim3(grid), dim3(256), 0, st, fd, (const uint8_t *)a, (int)a_stride, (const uint8_t *)base, (int)base_stride, (i64 *)out, n, dev_err); break;
    case GFA_U16: hipLaunchKernelGGL(log_lut_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, fd, (const uint16_t *)a, (int)a_stride, (const uint16_t *)base, (int)base_stride, (i64 *)out, n, dev_err); break;
    case GFA_U32: hipLaunchKernelGGL(log_lut_kernel<uint32_t>, dim3(grid), dim3(256), 0, st, fd, (const uint32_t *)a, (int)a_stride, (const uint32_t *)base, (int)base_stride, (i64 *)out, n, dev_err); break;
    case GFA_U64: hipLaunchKernelGGL(log_lut_kernel<uint64_t>, dim3(grid), dim3(256), 0, st, fd, (const uint64_t *)a, (int)a_stride, (const uint64_t *)base, (int)base_stride, (i64 *)out, n, dev_err); break;
    default: set_error("gfa_log: bad dtype"); return GFA_ERR_INVALID;
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int gfa_time_binary(gfa_field_t *f, int op, const void *a, const void *b, void *out, int64_t n, int dtype,
                    gfa_stream_t stream, int iters, float *ms_out)
{
    return gfa::time_loop((hipStream_t)stream, iters, ms_out,
                     [&]() { return gfa_binary(f, op, a, 1, b, 1, out, n, dtype, stream, nullptr); });
}

int gfa_time_unary(gfa_field_t *f, int op, const void *a, void *out, int64_t n, int dtype, gfa_stream_t stream, int iters,
                   float *ms_out)
{
    return gfa::time_loop((hipStream_t)stream, iters, ms_out,
                     [&]() { return gfa_unary(f, op, a, out, n, dtype, stream, nullptr); });
}

} // extern "C"
