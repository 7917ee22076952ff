// gfa_elementwise_mid.hip -- element-wise arithmetic over fields of 257 .. 8192 elements on uint16 storage with the
// EXP / LOG / Zech-log tables resident in LDS as 16-bit entries.
//
// Reference seam: the lookup ufuncs of galois/_domains/_lookup.py:153-270 (add / negative / subtract / multiply / reciprocal /
// divide / power through EXP, LOG, ZECH_LOG) -- same tables (gfa_field.hip builds them entry for entry), same index
// arithmetic, so results are the same integers as the generic Lut kernels of gfa_elementwise.hip, which gather from
// global memory (L1 / L2).  Here one workgroup stages LOG (q), EXP (2q) and ZECH (q) once -- 8q bytes, at most 64 KiB -- and
// then streams 16-byte vectors of the operands through them; every table access is a ds_read_u16.
//
// What bounds it: the LDS random-gather rate (2 gathers for a reciprocal, 3 for a product or quotient, 4 for a sum in odd
// characteristic), not HBM -- see DESIGN.md section 4.2 (7).
#include "gfa_internal.h"

using namespace gfa;

namespace {

constexpr int MID_THREADS = 512;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16;

struct MidDesc {
    const u16 *image; // LOG[qa] | EXP[2*qa] | ZECH[qa], qa = q rounded up to a multiple of 8 (gfa_field::ensure_device)
    u32 q, qa, qm1, zech_e;
    const i64 *e_ptr; // power: the one exponent of the call (device memory)
    u32 mu;           // power: floor(2^32 / (q-1))
};

struct MidPow {
    u32 em; // exponent reduced to [0, q-1) with a floor modulo, as Lut::pow_nz
    bool e_zero, e_neg;
};

enum { MID_NEG = 16, MID_RECIP = 17, MID_POW = 18 };

struct MidTabs {
    const u16 *lg, *ex, *ze;
};

template <int OP>
__device__ __forceinline__ u32 mid_op(const MidTabs &t, const MidDesc &d, const MidPow &pw, u32 a, u32 b, bool &bad)
{
    if constexpr (OP == GFA_OP_MUL) { // multiply_ufunc.lookup: EXP[LOG[a] + LOG[b]], 0 if either is 0
        const u32 r = t.ex[(u32)t.lg[a] + (u32)t.lg[b]];
        return (a == 0 || b == 0) ? 0u : r;
    } else if constexpr (OP == GFA_OP_DIV) { // divide_ufunc.lookup: EXP[(q-1) + LOG[a] - LOG[b]]
        bad |= b == 0;
        const u32 r = t.ex[d.qm1 + (u32)t.lg[a] - (u32)t.lg[b]];
        return (a == 0 || b == 0) ? 0u : r;
    } else if constexpr (OP == GFA_OP_ADD) { // add_ufunc.lookup (odd characteristic): EXP[m + ZECH[n - m]], m <= n the two logs
        const u32 la = t.lg[a], lb = t.lg[b];
        const u32 mm = min(la, lb), nn = max(la, lb);
        const u32 z = nn - mm;
        const u32 r = t.ex[mm + (u32)t.ze[z]];
        u32 res = z == d.zech_e ? 0u : r;
        res = b == 0 ? a : res;
        res = a == 0 ? b : res;
        return res;
    } else if constexpr (OP == GFA_OP_SUB) { // subtract_ufunc.lookup: a + (-b), -b = EXP[LOG[b] + ZECH_E]
        const u32 nn0 = (u32)t.lg[b] + d.zech_e;
        const u32 la = t.lg[a];
        const u32 mm = min(la, nn0), nn = max(la, nn0);
        u32 z = nn - mm;
        const bool cancel = z == d.zech_e;
        z = z >= d.qm1 ? z - d.qm1 : z;
        const u32 r = t.ex[mm + (u32)t.ze[z]];
        u32 res = cancel ? 0u : r;
        if (a == 0) res = t.ex[nn0]; // rare: skipped by the whole wave almost always
        res = b == 0 ? a : res;
        return res;
    } else if constexpr (OP == MID_NEG) { // negative_ufunc.lookup: EXP[LOG[a] + ZECH_E]
        const u32 r = t.ex[(u32)t.lg[a] + d.zech_e];
        return a == 0 ? 0u : r;
    } else if constexpr (OP == MID_RECIP) { // reciprocal_ufunc.lookup: EXP[(q-1) - LOG[a]]
        bad |= a == 0;
        const u32 r = t.ex[d.qm1 - (u32)t.lg[a]];
        return a == 0 ? 0u : r;
    } else { // MID_POW, one exponent for the whole array: EXP[(LOG[a] * e) mod (q-1)] (power_ufunc.lookup, _lookup.py:247-270)
        const u32 x = (u32)t.lg[a] * pw.em; // < 2^26
        u32 idx = x - __umulhi(x, d.mu) * d.qm1; // in [0, 2(q-1))
        idx = idx >= d.qm1 ? idx - d.qm1 : idx;
        const u32 r = t.ex[idx];
        bad |= (a == 0) && pw.e_neg;
        return pw.e_zero ? 1u : (a == 0 ? 0u : r);
    }
}

template <int OP>
__global__ __launch_bounds__(MID_THREADS) void mid_kernel(MidDesc d, const u16 *__restrict__ a, int sa, const u16 *__restrict__ b,
                                                          int sb, u16 *__restrict__ out, i64 n, int32_t *err)
{
    extern __shared__ __attribute__((aligned(16))) u16 mid_lds[];
    constexpr bool BINARY = OP <= GFA_OP_DIV;
    constexpr bool NEED_ZECH = OP == GFA_OP_ADD || OP == GFA_OP_SUB;
    const i64 nvec = n >> 3;
    const u32x4 *av = reinterpret_cast<const u32x4 *>(a);
    const u32x4 *bv = reinterpret_cast<const u32x4 *>(b);
    u32x4 *ov = reinterpret_cast<u32x4 *>(out);
    const i64 stride = (i64)gridDim.x * MID_THREADS;
    i64 i = (i64)blockIdx.x * MID_THREADS + threadIdx.x;
    u32x4 x = {0, 0, 0, 0}, y = {0, 0, 0, 0};
    if (!sa) { const u32 s = a[0]; x = u32x4{s, s, s, s} * 0x10001u; }
    if (BINARY && !sb) { const u32 s = b[0]; y = u32x4{s, s, s, s} * 0x10001u; }
    // the first operand vectors are requested before the tables are staged (as in tab8_binary_kernel)
    if (i < nvec) {
        if (sa) x = av[i];
        if (BINARY && sb) y = bv[i];
    }
    {
        const int words = (int)((NEED_ZECH ? 4u : 3u) * d.qa / 8u); // 16-byte units
        const uint4 *src = reinterpret_cast<const uint4 *>(d.image);
        uint4 *dst = reinterpret_cast<uint4 *>(mid_lds);
        for (int t = threadIdx.x; t < words; t += MID_THREADS) dst[t] = src[t];
    }
    __syncthreads();
    MidTabs t;
    t.lg = mid_lds;
    t.ex = mid_lds + d.qa;
    t.ze = mid_lds + 3 * d.qa;
    MidPow pw{0, false, false};
    if constexpr (OP == MID_POW) {
        const i64 e = d.e_ptr[0];
        i64 em = e % (i64)d.qm1;
        if (em < 0) em += d.qm1;
        pw = MidPow{(u32)em, e == 0, e < 0};
    }
    bool bad = false;
    for (; i < nvec; i += stride) {
        const u32x4 cx = x, cy = y;
        const i64 nxt = i + stride;
        if (nxt < nvec) {
            if (sa) x = av[nxt];
            if (BINARY && sb) y = bv[nxt];
        }
        u32x4 r;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const u32 lo = mid_op<OP>(t, d, pw, cx[w] & 0xffffu, cy[w] & 0xffffu, bad);
            const u32 hi = mid_op<OP>(t, d, pw, cx[w] >> 16, cy[w] >> 16, bad);
            r[w] = lo | (hi << 16);
        }
        ov[i] = r;
    }
    for (i64 j = (nvec << 3) + (i64)blockIdx.x * MID_THREADS + threadIdx.x; j < n; j += stride)
        out[j] = (u16)mid_op<OP>(t, d, pw, (u32)a[sa ? j : 0], BINARY ? (u32)b[sb ? j : 0] : 0u, bad);
    if constexpr (OP == GFA_OP_DIV || OP == MID_RECIP || OP == MID_POW) {
        if (__any(bad)) {
            if ((threadIdx.x & 63) == 0 && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
        }
    }
}

int mid_num_cus()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int OP>
int mid_launch(const MidDesc &d, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st, int32_t *err)
{
    constexpr bool NEED_ZECH = OP == GFA_OP_ADD || OP == GFA_OP_SUB;
    const size_t lds = (size_t)(NEED_ZECH ? 4 : 3) * d.qa * sizeof(u16);
    static bool attr = false;
    auto k = mid_kernel<OP>;
    if (!attr) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); attr = true; }
    // 160 KiB of LDS and 32 waves per CU: up to four 8-wave workgroups when the tables are small
    i64 per_cu = (i64)(160 * 1024) / (i64)(lds + 1024);
    per_cu = per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu;
    i64 blocks = ((n >> 3) + MID_THREADS - 1) / MID_THREADS;
    const i64 cap = (i64)mid_num_cus() * per_cu;
    if (blocks < 1) blocks = 1;
    const int grid = (int)(blocks < cap ? blocks : cap);
    hipLaunchKernelGGL(k, dim3(grid), dim3(MID_THREADS), lds, st, d, (const u16 *)a, (int)sa, (const u16 *)b, (int)sb, (u16 *)out, n, err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

MidDesc make_desc(const FieldDev &lut, const u16 *image)
{
    MidDesc d{};
    d.image = image;
    d.q = (u32)lut.q;
    d.qa = (d.q + 7u) & ~7u;
    d.qm1 = lut.qm1;
    d.zech_e = lut.zech_e;
    return d;
}

} // namespace

namespace gfa {

// arrays below this many elements stay on the generic kernels (staging 8q bytes per workgroup would dominate)
static const i64 MID_MIN_N = [] { const char *e = getenv("GFA_MID_MIN_N"); return e ? (i64)atoll(e) : (i64)1 << 17; }();

bool mid_eligible(const FieldDev &calc, const void *image, int dtype, i64 n)
{
    static const bool enabled = [] { const char *e = getenv("GFA_MID_LDS"); return !(e && e[0] == '0'); }();
    return enabled && image != nullptr && dtype == GFA_U16 && calc.q > 256 && calc.q <= 8192 && n >= MID_MIN_N;
}

int mid_binary(const FieldDev &lut, const void *image, int op, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n,
               hipStream_t st, int32_t *err)
{
    if (!al16(out) || (sa && !al16(a)) || (sb && !al16(b))) return GFA_ERR_UNSUPPORTED;
    const MidDesc d = make_desc(lut, (const u16 *)image);
    switch (op) {
    case GFA_OP_ADD: return mid_launch<GFA_OP_ADD>(d, a, sa, b, sb, out, n, st, err);
    case GFA_OP_SUB: return mid_launch<GFA_OP_SUB>(d, a, sa, b, sb, out, n, st, err);
    case GFA_OP_MUL: return mid_launch<GFA_OP_MUL>(d, a, sa, b, sb, out, n, st, err);
    case GFA_OP_DIV: return mid_launch<GFA_OP_DIV>(d, a, sa, b, sb, out, n, st, err);
    default: return GFA_ERR_UNSUPPORTED;
    }
}

int mid_unary(const FieldDev &lut, const void *image, int op, const void *a, void *out, i64 n, hipStream_t st, int32_t *err)
{
    if (!al16(out) || !al16(a)) return GFA_ERR_UNSUPPORTED;
    const MidDesc d = make_desc(lut, (const u16 *)image);
    if (op == GFA_OP_NEG) return mid_launch<MID_NEG>(d, a, 1, a, 0, out, n, st, err);
    if (op == GFA_OP_RECIP) return mid_launch<MID_RECIP>(d, a, 1, a, 0, out, n, st, err);
    return GFA_ERR_UNSUPPORTED;
}

int mid_power(const FieldDev &lut, const void *image, const void *a, const i64 *e, void *out, i64 n, hipStream_t st, int32_t *err)
{
    if (!al16(out) || !al16(a)) return GFA_ERR_UNSUPPORTED;
    MidDesc d = make_desc(lut, (const u16 *)image);
    d.e_ptr = e;
    d.mu = (u32)(0x100000000ull / d.qm1);
    return mid_launch<MID_POW>(d, a, 1, a, 0, out, n, st, err);
}

} // namespace gfa
