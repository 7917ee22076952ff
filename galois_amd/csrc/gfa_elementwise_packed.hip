// gfa_elementwise_packed.hip -- np.add / np.subtract / np.negative over GF(p^m), p odd, 8192 < q <= 2^20, as packed-digit arithmetic
// (r06: GF(3^11) and GF(3^12), whose digits need 33 / 36 packed bits, through two words: packed_lin2_kernel).
//
// Replaces, for these fields, the reference's add / subtract / negative ufuncs in BOTH of its modes -- the Zech-logarithm lookups
// (src/galois/_domains/_lookup.py:31-60, 89-150) and the digit-vector loops (_calculate.py:150-285) -- with the scheme of
// gfa_packed.h: two small LDS tables turn an integer into its base-p digits packed W bits apart, the m digit sums and their
// conditional subtractions are six integer instructions on that word, a few chunk tables turn it back.  The tables of these
// fields do not fit LDS together (32768 < q <= 65536: one staged at a time, GF(3^10) sums 0.26 of the roofline) or at all
// (q > 65536: gathers from L2, GF(7^7) sums 0.04); this kernel needs 3-41 KiB of LDS whatever the order and streams.
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

#include "gfa_internal.h"
#include "gfa_packed.h"

using namespace gfa;
using namespace gfa_packed;

namespace {

constexpr int PK_THREADS = 512;

template <typename T> struct PkVec;
template <> struct PkVec<uint8_t> { static constexpr int N = 16; };
template <> struct PkVec<uint16_t> { static constexpr int N = 8; };
template <> struct PkVec<uint32_t> { static constexpr int N = 4; };

template <typename T>
__device__ __forceinline__ void unpack_vec(const uint4 &v, pu32 (&e)[PkVec<T>::N])
{
    const pu32 w[4] = {v.x, v.y, v.z, v.w};
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < 4; j++) e[j] = w[j];
    } else if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < 4; j++) { e[2 * j] = w[j] & 0xffffu; e[2 * j + 1] = w[j] >> 16; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 4; k++) e[4 * j + k] = (w[j] >> (8 * k)) & 0xffu;
    }
}
template <typename T>
__device__ __forceinline__ uint4 pack_vec(const pu32 (&e)[PkVec<T>::N])
{
    if constexpr (sizeof(T) == 4) return make_uint4(e[0], e[1], e[2], e[3]);
    else if constexpr (sizeof(T) == 2) return make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    else {
        pu32 w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = e[4 * j] | (e[4 * j + 1] << 8) | (e[4 * j + 2] << 16) | (e[4 * j + 3] << 24);
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// OP: 0 add, 1 sub, 2 neg (b unused).  sa / sb: 1 = one element per output, 0 = one element for the whole array.
template <typename T, int OP>
__global__ __launch_bounds__(PK_THREADS) void packed_lin_kernel(Plan pl, const pu32 *__restrict__ gtab, const T *__restrict__ a, int sa,
                                                                 const T *__restrict__ b, int sb, T *__restrict__ out, i64 n)
{
    extern __shared__ pu32 pk_tab[];
    for (pu32 i = threadIdx.x; i < pl.words; i += PK_THREADS) pk_tab[i] = gtab[i];
    __syncthreads();
    constexpr int V = PkVec<T>::N;
    const i64 nvec = n / V;
    const pu32 pa0 = sa ? 0u : to_packed(pl, pk_tab, (pu32)a[0]);
    const pu32 pb0 = (OP == 2 || sb) ? 0u : to_packed(pl, pk_tab, (pu32)b[0]);
    const uint4 *av = reinterpret_cast<const uint4 *>(a), *bv = reinterpret_cast<const uint4 *>(b);
    uint4 *ov = reinterpret_cast<uint4 *>(out);
    for (i64 i = (i64)blockIdx.x * PK_THREADS + threadIdx.x; i < nvec; i += (i64)gridDim.x * PK_THREADS) {
        pu32 xa[V], xb[V], r[V];
        if (sa) unpack_vec<T>(av[i], xa);
        if (OP != 2 && sb) unpack_vec<T>(bv[i], xb);
#pragma unroll
        for (int j = 0; j < V; j++) {
            const pu32 pa = sa ? to_packed(pl, pk_tab, xa[j]) : pa0;
            const pu32 pb = OP == 2 ? 0u : (sb ? to_packed(pl, pk_tab, xb[j]) : pb0);
            r[j] = from_packed(pl, pk_tab, lin_packed<OP>(pl, pa, pb));
        }
        ov[i] = pack_vec<T>(r);
    }
    // the last n % V elements
    const i64 t0 = nvec * V + (i64)blockIdx.x * PK_THREADS + threadIdx.x;
    if (t0 < n) {
        const pu32 pa = sa ? to_packed(pl, pk_tab, (pu32)a[t0]) : pa0;
        const pu32 pb = OP == 2 ? 0u : (sb ? to_packed(pl, pk_tab, (pu32)b[t0]) : pb0);
        out[t0] = (T)from_packed(pl, pk_tab, lin_packed<OP>(pl, pa, pb));
    }
}

// the same for the two fields that need TWO packed words (r06: GF(3^11), GF(3^12); gfa_packed.h::Plan2), uint32 arrays
template <int OP>
__global__ __launch_bounds__(PK_THREADS) void packed_lin2_kernel(Plan2 pl, const pu32 *__restrict__ gtab, const uint32_t *__restrict__ a, int sa,
                                                                  const uint32_t *__restrict__ b, int sb, uint32_t *__restrict__ out, i64 n)
{
    extern __shared__ pu32 pk_tab[];
    for (pu32 i = threadIdx.x; i < pl.words; i += PK_THREADS) pk_tab[i] = gtab[i];
    __syncthreads();
    constexpr int V = 4;
    const i64 nvec = n / V;
    const Pk2 pa0 = sa ? Pk2{0u, 0u} : to_packed2(pl, pk_tab, a[0]);
    const Pk2 pb0 = (OP == 2 || sb) ? Pk2{0u, 0u} : to_packed2(pl, pk_tab, b[0]);
    const uint4 *av = reinterpret_cast<const uint4 *>(a), *bv = reinterpret_cast<const uint4 *>(b);
    uint4 *ov = reinterpret_cast<uint4 *>(out);
    for (i64 i = (i64)blockIdx.x * PK_THREADS + threadIdx.x; i < nvec; i += (i64)gridDim.x * PK_THREADS) {
        pu32 xa[V], xb[V], r[V];
        if (sa) unpack_vec<uint32_t>(av[i], xa);
        if (OP != 2 && sb) unpack_vec<uint32_t>(bv[i], xb);
#pragma unroll
        for (int j = 0; j < V; j++) {
            const Pk2 pa = sa ? to_packed2(pl, pk_tab, xa[j]) : pa0;
            const Pk2 pb = OP == 2 ? Pk2{0u, 0u} : (sb ? to_packed2(pl, pk_tab, xb[j]) : pb0);
            r[j] = from_packed2(pl, pk_tab, lin_packed2<OP>(pl, pa, pb));
        }
        ov[i] = pack_vec<uint32_t>(r);
    }
    const i64 t0 = nvec * V + (i64)blockIdx.x * PK_THREADS + threadIdx.x;
    if (t0 < n) {
        const Pk2 pa = sa ? to_packed2(pl, pk_tab, a[t0]) : pa0;
        const Pk2 pb = OP == 2 ? Pk2{0u, 0u} : (sb ? to_packed2(pl, pk_tab, b[t0]) : pb0);
        out[t0] = from_packed2(pl, pk_tab, lin_packed2<OP>(pl, pa, pb));
    }
}

// products of uint32 arrays (65536 < q <= 2^20): digits through the same LDS tables, gfa_packed.h::mul_digits (no reduction before the end)
template <typename T, int M>
__global__ __launch_bounds__(PK_THREADS) void packed_mul_kernel(Plan pl, MulAux ax, const pu32 *__restrict__ gtab, const T *__restrict__ a, int sa,
                                                                 const T *__restrict__ b, int sb, T *__restrict__ out, i64 n)
{
    constexpr int V = PkVec<T>::N;
    extern __shared__ pu32 pk_tab[];
    for (pu32 i = threadIdx.x; i < pl.off_un; i += PK_THREADS) pk_tab[i] = gtab[i]; // PK_LO and PK_HI only: the way back is Horner's rule
    __syncthreads();
    const i64 nvec = n / V;
    const pu32 pa0 = sa ? 0u : to_packed(pl, pk_tab, (pu32)a[0]);
    const pu32 pb0 = sb ? 0u : to_packed(pl, pk_tab, (pu32)b[0]);
    const uint4 *av = reinterpret_cast<const uint4 *>(a), *bv = reinterpret_cast<const uint4 *>(b);
    uint4 *ov = reinterpret_cast<uint4 *>(out);
    for (i64 i = (i64)blockIdx.x * PK_THREADS + threadIdx.x; i < nvec; i += (i64)gridDim.x * PK_THREADS) {
        pu32 xa[V], xb[V], r[V];
        if (sa) unpack_vec<T>(av[i], xa);
        if (sb) unpack_vec<T>(bv[i], xb);
#pragma unroll
        for (int j = 0; j < V; j++)
            r[j] = mul_digits<M>(pl, ax, sa ? to_packed(pl, pk_tab, xa[j]) : pa0, sb ? to_packed(pl, pk_tab, xb[j]) : pb0);
        ov[i] = pack_vec<T>(r);
    }
    const i64 t0 = nvec * V + (i64)blockIdx.x * PK_THREADS + threadIdx.x;
    if (t0 < n) out[t0] = (T)mul_digits<M>(pl, ax, sa ? to_packed(pl, pk_tab, (pu32)a[t0]) : pa0, sb ? to_packed(pl, pk_tab, (pu32)b[t0]) : pb0);
}

// quotients / reciprocals of GF(p^2), 32768 < q <= 2^20, uint16 / uint32 arrays: gfa_packed.h::div2 (norm + a p-entry inverse table in LDS)
// WIDE (r06): 1021 < p <= 37813, the field has no tables at all (q > 2^20): exact digit split, inverse table as 16-bit entries (ginv: (p + 1) / 2 words)
template <typename T, bool RECIP, bool WIDE = false>
__global__ __launch_bounds__(PK_THREADS) void packed_div2_kernel(Div2Aux ax, const pu32 *__restrict__ ginv, const T *__restrict__ a, int sa,
                                                                  const T *__restrict__ b, int sb, T *__restrict__ out, i64 n, int *err)
{
    extern __shared__ pu32 pk_tab[];
    for (pu32 i = threadIdx.x; i < (WIDE ? (ax.p + 1) / 2 : ax.p); i += PK_THREADS) pk_tab[i] = ginv[i];
    __syncthreads();
    using IT = typename std::conditional<WIDE, uint16_t, pu32>::type;
    const IT *inv_tab = reinterpret_cast<const IT *>(pk_tab);
    constexpr int V = PkVec<T>::N;
    const i64 nvec = n / V;
    const uint4 *av = reinterpret_cast<const uint4 *>(a), *bv = reinterpret_cast<const uint4 *>(b);
    uint4 *ov = reinterpret_cast<uint4 *>(out);
    const pu32 a0 = (RECIP || sa) ? 0u : (pu32)a[0], b0 = sb ? 0u : (pu32)b[0];
    bool bad = false;
    for (i64 i = (i64)blockIdx.x * PK_THREADS + threadIdx.x; i < nvec; i += (i64)gridDim.x * PK_THREADS) {
        pu32 xa[V], xb[V], r[V];
        if (!RECIP && sa) unpack_vec<T>(av[i], xa);
        if (sb) unpack_vec<T>(bv[i], xb);
#pragma unroll
        for (int j = 0; j < V; j++) {
            bool z;
            r[j] = div2<RECIP, WIDE, IT>(ax, inv_tab, (!RECIP && sa) ? xa[j] : a0, sb ? xb[j] : b0, &z);
            bad |= z;
        }
        ov[i] = pack_vec<T>(r);
    }
    const i64 t0 = nvec * V + (i64)blockIdx.x * PK_THREADS + threadIdx.x;
    if (t0 < n) {
        bool z;
        out[t0] = (T)div2<RECIP, WIDE, IT>(ax, inv_tab, (!RECIP && sa) ? (pu32)a[t0] : a0, sb ? (pu32)b[t0] : b0, &z);
        bad |= z;
    }
    if (bad && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
}

// quotients / reciprocals of GF(p^3), 65536 < q <= 2^20: gfa_packed.h::div3 (Cramer's rule on the multiplication matrix + the same inverse table)
template <typename T, bool RECIP, bool WIDE = false>
__global__ __launch_bounds__(PK_THREADS) void packed_div3_kernel(Div3Aux ax, const pu32 *__restrict__ ginv, const T *__restrict__ a, int sa,
                                                                  const T *__restrict__ b, int sb, T *__restrict__ out, i64 n, int *err)
{
    extern __shared__ pu32 pk_tab[];
    for (pu32 i = threadIdx.x; i < ax.p; i += PK_THREADS) pk_tab[i] = ginv[i];
    __syncthreads();
    constexpr int V = PkVec<T>::N;
    const i64 nvec = n / V;
    const uint4 *av = reinterpret_cast<const uint4 *>(a), *bv = reinterpret_cast<const uint4 *>(b);
    uint4 *ov = reinterpret_cast<uint4 *>(out);
    const pu32 a0 = (RECIP || sa) ? 0u : (pu32)a[0], b0 = sb ? 0u : (pu32)b[0];
    bool bad = false;
    for (i64 i = (i64)blockIdx.x * PK_THREADS + threadIdx.x; i < nvec; i += (i64)gridDim.x * PK_THREADS) {
        pu32 xa[V], xb[V], r[V];
        if (!RECIP && sa) unpack_vec<T>(av[i], xa);
        if (sb) unpack_vec<T>(bv[i], xb);
#pragma unroll
        for (int j = 0; j < V; j++) {
            bool z;
            r[j] = div3<RECIP, WIDE>(ax, pk_tab, (!RECIP && sa) ? xa[j] : a0, sb ? xb[j] : b0, &z);
            bad |= z;
        }
        ov[i] = pack_vec<T>(r);
    }
    const i64 t0 = nvec * V + (i64)blockIdx.x * PK_THREADS + threadIdx.x;
    if (t0 < n) {
        bool z;
        out[t0] = (T)div3<RECIP, WIDE>(ax, pk_tab, (!RECIP && sa) ? (pu32)a[t0] : a0, sb ? (pu32)b[t0] : b0, &z);
        bad |= z;
    }
    if (bad && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
}

// quotients / reciprocals of the degrees Cramer's rule does not reach (4 .. 8), uint32 arrays, 65536 < q <= 2^20 (r06): 1 / b is ONE gather from
// the field's 3-byte inverse table (gfa_field::inverse_table: 2.4 MB for GF(7^7), inside one XCD's L2, where LOG + EXP were two gathers out of
// 6.6 MB), the quotient a * (1 / b) the digit-table product of packed_mul_kernel.  Gathers run one iteration ahead of the products, operand loads two.
typedef pu32 __attribute__((aligned(1))) pu32_unaligned;
#ifndef GFA_INV24_NT
#define GFA_INV24_NT 0 // 1: non-temporal gathers (measured: see DESIGN 4.2 item 6)
#endif
__device__ __forceinline__ pu32 inv24_load(const uint8_t *__restrict__ t, pu32 x)
{
    const pu32_unaligned *q = reinterpret_cast<const pu32_unaligned *>(t + 3u * x);
#if GFA_INV24_NT == 1
    return __builtin_nontemporal_load(q) & 0xffffffu;
#else
    return *q & 0xffffffu;
#endif
}

template <int M, bool RECIP>
__global__ __launch_bounds__(PK_THREADS) void packed_divt_kernel(Plan pl, MulAux ax, const pu32 *__restrict__ gtab, const uint8_t *__restrict__ inv24,
                                                                  const uint32_t *__restrict__ a, int sa, const uint32_t *__restrict__ b, int sb,
                                                                  uint32_t *__restrict__ out, i64 n, int *err, const i64 *__restrict__ e_ptr = nullptr)
{
    constexpr int V = 4;
    extern __shared__ pu32 pk_tab[];
    if (!RECIP) {
        for (pu32 i = threadIdx.x; i < pl.off_un; i += PK_THREADS) pk_tab[i] = gtab[i];
        __syncthreads();
    }
    const i64 nvec = n / V, S = (i64)gridDim.x * PK_THREADS;
    const pu32 a0 = (RECIP || sa) ? 0u : a[0], b0 = sb ? 1u : b[0];
    bool bad = b0 == 0u;
    const uint4 *av = reinterpret_cast<const uint4 *>(a), *bv = reinterpret_cast<const uint4 *>(b);
    uint4 *ov = reinterpret_cast<uint4 *>(out);
    // No branch around a gather (each would get its own wait): a scalar operand is a vector of copies, loads past the end re-read the last
    // vector (always there: n >= 1024) -- their zero flags are masked, their results not stored.
    const i64 last = nvec - 1;
    auto load_b = [&](i64 k) { uint4 x = make_uint4(b0, b0, b0, b0); if (sb) x = bv[k < nvec ? k : last]; return x; };
    auto load_a = [&](i64 k) { uint4 x = make_uint4(a0, a0, a0, a0); if (!RECIP && sa) x = av[k < nvec ? k : last]; return x; };
    i64 i = (i64)blockIdx.x * PK_THREADS + threadIdx.x;
    uint4 xb2 = load_b(i + S), xa0, xa1 = load_a(i);
    pu32 g0[V], g1[V];
    {
        pu32 e[V];
        unpack_vec<uint32_t>(load_b(i), e);
#pragma unroll
        for (int j = 0; j < V; j++) { g1[j] = inv24_load(inv24, e[j]); bad |= e[j] == 0u && i < nvec; }
    }
    for (; i < nvec; i += S) {
        xa0 = xa1;
#pragma unroll
        for (int j = 0; j < V; j++) g0[j] = g1[j];
        { // gathers of the next iteration, operand loads of the one after
            pu32 e[V];
            unpack_vec<uint32_t>(xb2, e);
#pragma unroll
            for (int j = 0; j < V; j++) { g1[j] = inv24_load(inv24, e[j]); bad |= e[j] == 0u && i + S < nvec; }
            xb2 = load_b(i + 2 * S);
            xa1 = load_a(i + S);
        }
        pu32 r[V];
        if (RECIP) {
#pragma unroll
            for (int j = 0; j < V; j++) r[j] = g0[j];
        } else {
            pu32 e[V];
            unpack_vec<uint32_t>(xa0, e);
#pragma unroll
            for (int j = 0; j < V; j++) r[j] = mul_digits<M>(pl, ax, to_packed(pl, pk_tab, e[j]), to_packed(pl, pk_tab, g0[j]));
        }
        ov[i] = pack_vec<uint32_t>(r);
    }
    const i64 t0 = nvec * V + (i64)blockIdx.x * PK_THREADS + threadIdx.x;
    if (t0 < n) {
        const pu32 xb = sb ? b[t0] : b0;
        bad |= xb == 0u;
        const pu32 ib = inv24_load(inv24, xb);
        out[t0] = RECIP ? ib : mul_digits<M>(pl, ax, to_packed(pl, pk_tab, sa ? a[t0] : a0), to_packed(pl, pk_tab, ib));
    }
    if (e_ptr) bad = bad && e_ptr[0] < 0; // the table holds x ** e (pow24_run): only 0 ** negative is an error
    if (bad && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
}

// P[x] = x ** e for every element of a table field (65536 < q <= 2^20), 3-byte entries as the inverse table: EXP[(LOG[x] e) mod (q - 1)],
// 0 ** e = 0, x ** 0 = 1 (power_ufunc.lookup, _lookup.py:247-270); the exponent is in device memory
__global__ __launch_bounds__(256) void pow24_table_kernel(const u32 *__restrict__ exp_tab, const u32 *__restrict__ log_tab, u32 q, const i64 *__restrict__ e_ptr,
                                                           uint8_t *__restrict__ tab)
{
    const i64 e = e_ptr[0], qm1 = (i64)q - 1;
    i64 em = e % qm1;
    em = em < 0 ? em + qm1 : em;
    for (u32 x = blockIdx.x * 256 + threadIdx.x; x < q; x += gridDim.x * 256) {
        u32 r;
        if (e == 0) r = 1u;
        else if (x == 0) r = 0u;
        else r = exp_tab[(u32)(((u64)log_tab[x] * (u64)em) % (u64)qm1)];
        tab[3 * (size_t)x] = (uint8_t)r; tab[3 * (size_t)x + 1] = (uint8_t)(r >> 8); tab[3 * (size_t)x + 2] = (uint8_t)(r >> 16);
    }
}

struct PackedDev {
    Plan pl;
    pu32 *tab = nullptr;
};
std::mutex g_pk_mu;
std::map<std::tuple<u64, u32, int>, PackedDev> g_pk; // (p, m, device)

int get_dev(const FieldDev &c, PackedDev *out)
{
    int dev = 0;
    GFA_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_pk_mu);
    auto key = std::make_tuple(c.p, c.m, dev);
    auto it = g_pk.find(key);
    if (it == g_pk.end()) {
        PackedDev d;
        if (!make_plan(c.p, c.m, &d.pl)) return GFA_ERR_UNSUPPORTED;
        std::vector<pu32> t;
        build_tables(d.pl, t);
        GFA_HIP(hipMalloc((void **)&d.tab, sizeof(pu32) * t.size()));
        GFA_HIP(hipMemcpy(d.tab, t.data(), sizeof(pu32) * t.size(), hipMemcpyHostToDevice)); // synchronous: usable from any stream afterwards
        it = g_pk.emplace(key, d).first;
    }
    *out = it->second;
    return GFA_OK;
}

struct Packed2Dev {
    Plan2 pl;
    pu32 *tab = nullptr;
};
std::map<std::tuple<u64, u32, int>, Packed2Dev> g_pk2; // (p, m, device)

int get_dev2(const FieldDev &c, Packed2Dev *out)
{
    int dev = 0;
    GFA_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_pk_mu);
    auto key = std::make_tuple(c.p, c.m, dev);
    auto it = g_pk2.find(key);
    if (it == g_pk2.end()) {
        Packed2Dev d;
        if (!make_plan2(c.p, c.m, &d.pl)) return GFA_ERR_UNSUPPORTED;
        std::vector<pu32> t;
        build_tables2(d.pl, t);
        GFA_HIP(hipMalloc((void **)&d.tab, sizeof(pu32) * t.size()));
        GFA_HIP(hipMemcpy(d.tab, t.data(), sizeof(pu32) * t.size(), hipMemcpyHostToDevice));
        it = g_pk2.emplace(key, d).first;
    }
    *out = it->second;
    return GFA_OK;
}

int num_cus()
{
    static const int cus = [] { int dev = 0, n = 0; return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }();
    return cus;
}

template <typename T, int OP>
int launch(const PackedDev &d, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st)
{
    constexpr int V = PkVec<T>::N;
    const i64 blocks = std::max<i64>(1, (n / V + PK_THREADS - 1) / PK_THREADS);
    const int grid = (int)std::min<i64>(blocks, (i64)num_cus() * 4);
    hipLaunchKernelGGL((packed_lin_kernel<T, OP>), dim3(grid), dim3(PK_THREADS), sizeof(pu32) * d.pl.words, st, d.pl, (const pu32 *)d.tab, (const T *)a, (int)sa,
                       (const T *)b, (int)sb, (T *)out, n);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

template <int OP>
int launch2(const Packed2Dev &d, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st)
{
    const i64 blocks = std::max<i64>(1, (n / 4 + PK_THREADS - 1) / PK_THREADS);
    const int grid = (int)std::min<i64>(blocks, (i64)num_cus() * 4);
    hipLaunchKernelGGL((packed_lin2_kernel<OP>), dim3(grid), dim3(PK_THREADS), sizeof(pu32) * d.pl.words, st, d.pl, (const pu32 *)d.tab, (const uint32_t *)a,
                       (int)sa, (const uint32_t *)b, (int)sb, (uint32_t *)out, n);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

inline bool al16p(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

} // namespace

namespace gfa {

// sums / differences / negatives of odd-characteristic extension fields whose Zech tables leave LDS; uint16 (q <= 65536) or uint32 arrays
// `pinned`: the field is pinned to explicit calculation -- then the scheme also replaces the digit-vector kernels of the small fields
// (q <= 8192), whose tables AUTO would keep in LDS (gfa_elementwise_mid.hip, faster); uint8 arrays included
bool packed_eligible(const FieldDev &c, int dtype, i64 n, bool pinned)
{
    if (c.m < 2 || (c.p & 1) == 0 || (c.q <= 8192 && !pinned) || c.q > ((u64)1 << 20) || n < 1024) return false;
    if (!(dtype == GFA_U32 || (dtype == GFA_U16 && c.q <= 65536) || (dtype == GFA_U8 && c.q <= 256))) return false;
    Plan pl;
    if (make_plan(c.p, c.m, &pl)) return true;
    Plan2 pl2; // r06: more than 32 packed bits (GF(3^11), GF(3^12)): two words, uint32 arrays
    return dtype == GFA_U32 && make_plan2(c.p, c.m, &pl2);
}

// op: GFA_OP_ADD / GFA_OP_SUB / GFA_OP_NEG.  GFA_ERR_UNSUPPORTED (nothing launched): misaligned operands.
int packed_run(const FieldDev &c, int dtype, int op, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st)
{
    if (!al16p(out) || (sa && !al16p(a)) || (op != GFA_OP_NEG && sb && !al16p(b))) return GFA_ERR_UNSUPPORTED;
    {
        Plan one;
        if (!make_plan(c.p, c.m, &one)) { // two packed words
            Packed2Dev d2;
            const int rc2 = get_dev2(c, &d2);
            if (rc2) return rc2;
            if (dtype != GFA_U32) return GFA_ERR_UNSUPPORTED;
            if (op == GFA_OP_NEG) { b = a; sb = 0; }
            switch (op) {
            case GFA_OP_ADD: return launch2<0>(d2, a, sa, b, sb, out, n, st);
            case GFA_OP_SUB: return launch2<1>(d2, a, sa, b, sb, out, n, st);
            default: return launch2<2>(d2, a, sa, b, sb, out, n, st);
            }
        }
    }
    PackedDev d;
    const int rc = get_dev(c, &d);
    if (rc) return rc;
    if (op == GFA_OP_NEG) { b = a; sb = 0; }
#define GFA_PK(T)                                                                 \
    switch (op) {                                                                 \
    case GFA_OP_ADD: return launch<T, 0>(d, a, sa, b, sb, out, n, st);            \
    case GFA_OP_SUB: return launch<T, 1>(d, a, sa, b, sb, out, n, st);            \
    default: return launch<T, 2>(d, a, sa, b, sb, out, n, st);                    \
    }
    if (dtype == GFA_U8) { GFA_PK(uint8_t) }
    if (dtype == GFA_U16) { GFA_PK(uint16_t) }
    GFA_PK(uint32_t)
#undef GFA_PK
}

// products: uint32 arrays of odd-characteristic extension fields with 65536 < q <= 2^20, degree <= 8, whose unreduced schoolbook
// product stays below 2^32 (mul_bound_ok); ext_irr: FieldDev's digits of (irr - x^m), degree m-1 .. 0
static bool packed_mul_aux(const FieldDev &c, Plan *pl, MulAux *ax, bool pinned)
{
    if (c.m < 2 || c.m > 8 || (c.p & 1) == 0 || (c.q <= 32768 && !pinned) || c.q > ((u64)1 << 20) || !make_plan(c.p, c.m, pl)) return false; // (r06: from 32768, where the LDS tables stop fitting together)
    for (u32 j = 0; j < 8; j++) ax->nir[j] = 0;
    for (u32 j = 0; j < c.m; j++) ax->nir[j] = c.ext_irr[c.m - 1 - j] ? (pu32)c.p - c.ext_irr[c.m - 1 - j] : 0u;
    ax->mu32 = (pu32)(((u64)1 << 32) / c.p);
    return mul_bound_ok(*pl, *ax);
}

bool packed_mul_eligible(const FieldDev &c, int dtype, i64 n, bool pinned)
{
    Plan pl;
    MulAux ax;
    if (!(dtype == GFA_U32 || (dtype == GFA_U16 && c.q <= 65536 && (pinned || c.q > 32768)) || (pinned && dtype == GFA_U8 && c.q <= 256))) return false;
    return n >= 1024 && packed_mul_aux(c, &pl, &ax, pinned);
}

int packed_mul_run(const FieldDev &c, int dtype, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st)
{
    if (!al16p(out) || (sa && !al16p(a)) || (sb && !al16p(b))) return GFA_ERR_UNSUPPORTED;
    PackedDev d;
    MulAux ax;
    Plan pl;
    if (!packed_mul_aux(c, &pl, &ax, true)) return GFA_ERR_UNSUPPORTED;
    const int rc = get_dev(c, &d);
    if (rc) return rc;
    const int vec = dtype == GFA_U32 ? 4 : dtype == GFA_U16 ? 8 : 16;
    const i64 blocks = std::max<i64>(1, (n / vec + PK_THREADS - 1) / PK_THREADS);
    const int grid = (int)std::min<i64>(blocks, (i64)num_cus() * 4);
    const size_t lds = sizeof(pu32) * d.pl.off_un;
#define GFA_PKM(T, MV)                                                                                                                 \
    case MV:                                                                                                                           \
        hipLaunchKernelGGL((packed_mul_kernel<T, MV>), dim3(grid), dim3(PK_THREADS), lds, st, d.pl, ax, (const pu32 *)d.tab, (const T *)a, (int)sa, \
                           (const T *)b, (int)sb, (T *)out, n);                                                                        \
        break;
#define GFA_PKM_ALL(T)                                                                                                      \
    switch (c.m) {                                                                                                          \
        GFA_PKM(T, 2) GFA_PKM(T, 3) GFA_PKM(T, 4) GFA_PKM(T, 5) GFA_PKM(T, 6) GFA_PKM(T, 7) GFA_PKM(T, 8)                   \
    default: return GFA_ERR_UNSUPPORTED;                                                                                    \
    }
    if (dtype == GFA_U32) { GFA_PKM_ALL(uint32_t) }
    else if (dtype == GFA_U16) { GFA_PKM_ALL(uint16_t) }
    else { GFA_PKM_ALL(uint8_t) }
#undef GFA_PKM_ALL
#undef GFA_PKM
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// quotients (reciprocals: a == nullptr) of GF(p^2) / GF(p^3), odd p (r06): 32768 < q <= 2^20 on the tight forms; q > 2^20 -- fields without tables,
// uint32 arrays, p <= 37813 (degree 2) / p <= 1621 (degree 3) -- on the WIDE forms (exact digit split; degree 2: 16-bit inverse table)
static std::map<std::pair<u64, int>, pu32 *> g_inv_tab; // (p [+ 2^40: 16-bit entries], device) -> p-entry inverse table

static void ext_nir(const FieldDev &c, pu32 *nir)
{
    for (u32 j = 0; j < 8; j++) nir[j] = 0;
    for (u32 j = 0; j < c.m && j < 8; j++) nir[j] = c.ext_irr[c.m - 1 - j] ? (pu32)c.p - c.ext_irr[c.m - 1 - j] : 0u;
}

static bool divn_wide(const FieldDev &c) { return c.q > ((u64)1 << 20); }

bool packed_divn_eligible(const FieldDev &c, int dtype, i64 n)
{
    if (c.kind != KIND_EXT || (c.p & 1) == 0 || n < 1024 || c.q <= 32768 || c.q > 0xffffffffull || (c.m != 2 && c.m != 3)) return false;
    pu32 nir[8];
    ext_nir(c, nir);
    Div2Aux a2;
    Div3Aux a3;
    if (divn_wide(c)) return dtype == GFA_U32 && (c.m == 2 ? make_div2(c.p, c.m, nir, &a2, true) : make_div3(c.p, c.m, nir, &a3, true));
    if (!(dtype == GFA_U32 || (dtype == GFA_U16 && c.q <= 65536))) return false;
    return (c.m == 2 && make_div2(c.p, c.m, nir, &a2)) || (c.m == 3 && dtype == GFA_U32 && make_div3(c.p, c.m, nir, &a3));
}

template <typename T, bool WIDE>
static int launch_div3(bool recip, int grid, const Div3Aux &ax, const pu32 *inv, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st, int *dev_err)
{
    if (recip)
        hipLaunchKernelGGL((packed_div3_kernel<T, true, WIDE>), dim3(grid), dim3(PK_THREADS), sizeof(pu32) * ax.p, st, ax, inv, (const T *)nullptr, 0, (const T *)b, (int)sb,
                           (T *)out, n, dev_err);
    else
        hipLaunchKernelGGL((packed_div3_kernel<T, false, WIDE>), dim3(grid), dim3(PK_THREADS), sizeof(pu32) * ax.p, st, ax, inv, (const T *)a, (int)sa, (const T *)b, (int)sb,
                           (T *)out, n, dev_err);
    return GFA_OK;
}

template <typename T, bool WIDE>
static int launch_div2(bool recip, int grid, const Div2Aux &ax, const pu32 *inv, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st, int *dev_err)
{
    const size_t lds = WIDE ? sizeof(pu32) * ((ax.p + 1) / 2) : sizeof(pu32) * ax.p;
    if (WIDE) { // up to 74 KiB of LDS
        static bool attr[2] = {false, false};
        const void *k = recip ? (const void *)packed_div2_kernel<T, true, WIDE> : (const void *)packed_div2_kernel<T, false, WIDE>;
        if (!attr[recip]) { GFA_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)); attr[recip] = true; }
    }
    if (recip)
        hipLaunchKernelGGL((packed_div2_kernel<T, true, WIDE>), dim3(grid), dim3(PK_THREADS), lds, st, ax, inv, (const T *)nullptr, 0, (const T *)b, (int)sb,
                           (T *)out, n, dev_err);
    else
        hipLaunchKernelGGL((packed_div2_kernel<T, false, WIDE>), dim3(grid), dim3(PK_THREADS), lds, st, ax, inv, (const T *)a, (int)sa, (const T *)b, (int)sb,
                           (T *)out, n, dev_err);
    return GFA_OK;
}

int packed_divn_run(const FieldDev &c, int dtype, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st, int *dev_err)
{
    const bool recip = a == nullptr;
    if (!al16p(out) || (!recip && sa && !al16p(a)) || (sb && !al16p(b))) return GFA_ERR_UNSUPPORTED;
    if (!packed_divn_eligible(c, dtype, n)) return GFA_ERR_UNSUPPORTED;
    const bool wide = divn_wide(c), half = wide && c.m == 2; // half: 16-bit table entries
    pu32 nir[8];
    ext_nir(c, nir);
    Div2Aux a2;
    Div3Aux a3;
    if (c.m == 2 ? !make_div2(c.p, c.m, nir, &a2, wide) : !make_div3(c.p, c.m, nir, &a3, wide)) return GFA_ERR_UNSUPPORTED;
    int dev = 0;
    GFA_HIP(hipGetDevice(&dev));
    pu32 *inv = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pk_mu);
        auto key = std::make_pair((u64)c.p + (half ? (u64)1 << 40 : 0), dev);
        auto it = g_inv_tab.find(key);
        if (it == g_inv_tab.end()) {
            std::vector<pu32> t;
            build_inverse_table((pu32)c.p, t);
            if (half) { // two 16-bit entries per word
                std::vector<pu32> h((t.size() + 1) / 2, 0);
                for (size_t v = 0; v < t.size(); v++) h[v / 2] |= t[v] << (16 * (v & 1));
                t.swap(h);
            }
            pu32 *d = nullptr;
            GFA_HIP(hipMalloc((void **)&d, sizeof(pu32) * t.size()));
            GFA_HIP(hipMemcpy(d, t.data(), sizeof(pu32) * t.size(), hipMemcpyHostToDevice));
            it = g_inv_tab.emplace(key, d).first;
        }
        inv = it->second;
    }
    const int vec = dtype == GFA_U32 ? 4 : 8;
    const i64 blocks = std::max<i64>(1, (n / vec + PK_THREADS - 1) / PK_THREADS);
    const int grid = (int)std::min<i64>(blocks, (i64)num_cus() * 4);
    int rc;
    if (c.m == 3) rc = wide ? launch_div3<uint32_t, true>(recip, grid, a3, inv, a, sa, b, sb, out, n, st, dev_err) : launch_div3<uint32_t, false>(recip, grid, a3, inv, a, sa, b, sb, out, n, st, dev_err);
    else if (wide) rc = launch_div2<uint32_t, true>(recip, grid, a2, inv, a, sa, b, sb, out, n, st, dev_err);
    else if (dtype == GFA_U32) rc = launch_div2<uint32_t, false>(recip, grid, a2, inv, a, sa, b, sb, out, n, st, dev_err);
    else rc = launch_div2<uint16_t, false>(recip, grid, a2, inv, a, sa, b, sb, out, n, st, dev_err);
    if (rc) return rc;
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// x ** e, one exponent (device memory), uint32 arrays of a table field with 65536 < q <= 2^20 (r06): a per-call table of x ** e in stream-ordered
// scratch (q look-ups), then ONE gather per element where LOG + EXP were two
int pow24_run(const u32 *exp_tab, const u32 *log_tab, u64 q, const void *a, const i64 *e_ptr, void *out, i64 n, hipStream_t st, int *dev_err)
{
    if (!exp_tab || !log_tab || q <= 65536 || q > ((u64)1 << 20) || n < 8 * (i64)q || !al16p(out) || !al16p(a)) return GFA_ERR_UNSUPPORTED;
    uint8_t *tab = nullptr;
    if (scratch_alloc((void **)&tab, 3 * (size_t)q + 4, st) != hipSuccess) { (void)hipGetLastError(); return GFA_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(pow24_table_kernel, dim3((int)((q + 255) / 256)), dim3(256), 0, st, exp_tab, log_tab, (u32)q, e_ptr, tab);
    const i64 blocks = std::max<i64>(1, (n / 4 + PK_THREADS - 1) / PK_THREADS);
    hipLaunchKernelGGL((packed_divt_kernel<4, true>), dim3((int)std::min<i64>(blocks, (i64)num_cus() * 4)), dim3(PK_THREADS), 0, st, Plan{}, MulAux{},
                       (const pu32 *)nullptr, (const uint8_t *)tab, (const uint32_t *)nullptr, 0, (const uint32_t *)a, 1, (uint32_t *)out, n, dev_err, e_ptr);
    const hipError_t le = hipGetLastError();
    GFA_HIP(scratch_free(tab, st));
    GFA_HIP(le);
    return GFA_OK;
}

// quotients / reciprocals (a == nullptr) of GF(p^m), m = 4 .. 8, odd p, 65536 < q <= 2^20, uint32 arrays (r06)
bool packed_divt_eligible(const FieldDev &c, int dtype, i64 n)
{
    Plan pl;
    MulAux mx;
    return dtype == GFA_U32 && n >= 1024 && c.m >= 4 && c.q > 65536 && packed_mul_aux(c, &pl, &mx, true);
}

int packed_divt_run(const FieldDev &c, const uint8_t *inv24, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st, int *dev_err)
{
    const bool recip = a == nullptr;
    if (!inv24 || !al16p(out) || (!recip && sa && !al16p(a)) || (sb && !al16p(b))) return GFA_ERR_UNSUPPORTED;
    if (recip && n >= 1024 && c.q > 65536) { // the gather alone: whatever the degree (GF(3^11), GF(3^12) have no one-word digit plan)
        const i64 blocks = std::max<i64>(1, (n / 4 + PK_THREADS - 1) / PK_THREADS);
        hipLaunchKernelGGL((packed_divt_kernel<4, true>), dim3((int)std::min<i64>(blocks, (i64)num_cus() * 4)), dim3(PK_THREADS), 0, st, Plan{}, MulAux{},
                           (const pu32 *)nullptr, inv24, (const uint32_t *)nullptr, 0, (const uint32_t *)b, (int)sb, (uint32_t *)out, n, dev_err);
        GFA_HIP(hipGetLastError());
        return GFA_OK;
    }
    if (!packed_divt_eligible(c, GFA_U32, n)) return GFA_ERR_UNSUPPORTED;
    PackedDev d;
    MulAux ax;
    Plan pl;
    if (!packed_mul_aux(c, &pl, &ax, true)) return GFA_ERR_UNSUPPORTED;
    const int rc = get_dev(c, &d);
    if (rc) return rc;
    const i64 blocks = std::max<i64>(1, (n / 4 + PK_THREADS - 1) / PK_THREADS);
    const int grid = (int)std::min<i64>(blocks, (i64)num_cus() * 4);
    const size_t lds = sizeof(pu32) * d.pl.off_un;
#define GFA_PKD(MV)                                                                                                                      \
    case MV:                                                                                                                             \
        hipLaunchKernelGGL((packed_divt_kernel<MV, false>), dim3(grid), dim3(PK_THREADS), lds, st, d.pl, ax, (const pu32 *)d.tab, inv24, \
                           (const uint32_t *)a, (int)sa, (const uint32_t *)b, (int)sb, (uint32_t *)out, n, dev_err);                     \
        break;
    switch (c.m) {
        GFA_PKD(4) GFA_PKD(5) GFA_PKD(6) GFA_PKD(7) GFA_PKD(8)
    default: return GFA_ERR_UNSUPPORTED;
    }
#undef GFA_PKD
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

} // namespace gfa
