// gfa_elementwise_mid.hip -- element-wise arithmetic over fields of 257 .. 65536 elements on uint16 storage with the
// EXP / LOG / Zech-log tables in LDS as 16-bit entries.
//
// Reference seam: the lookup ufuncs of galois/_domains/_lookup.py:153-270 (add / negative / subtract / multiply / reciprocal /
// divide / power through EXP, LOG, ZECH_LOG) -- same tables (gfa_field.hip builds them entry for entry), same index
// arithmetic, so results are the same integers as the generic Lut kernels of gfa_elementwise.hip, which gather from
// global memory (L1 / L2).  Three regimes, by table size against the CU's 160 KiB:
//   * q <= 8192   mid_kernel: LOG (q), EXP (2q), ZECH (q) resident, 8q <= 64 KiB, up to four 512-thread workgroups per CU;
//   * q <= 32768  mid_kernel<.., REDUCED>: EXP shortened to q entries, every index brought below q - 1 first; LOG + EXP (+ ZECH
//                 while 6q bytes fit) resident, one 1024-thread workgroup per CU;
//   * q <= 65536  big16_kernel / big16_addsub_kernel: one table at a time -- LOG, (ZECH,) EXP staged in turn per tile, the
//                 indices in between kept in registers.
// 16-byte operand vectors throughout; every table access is a ds_read_u16.
//
// What bounds them (measured): HBM up to 32768 elements (0.7-0.8 of the 6 B/element roofline: the
// LDS random-gather rate of 2-4 gathers per element is just sufficient); LDS -- gathers plus table re-staging -- above.
#include "gfa_internal.h"

using namespace gfa;

namespace {

constexpr int MID_THREADS = 512;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16;

struct MidDesc {
    const u16 *image; // LOG[qa] | EXP[2*qa] | ZECH[qa], qa = q rounded up to a multiple of 8 (gfa_field::ensure_device); q > 32768: LOG | EXP[qa] | ZECH | INV[qa]
    u32 q, qa, qm1, zech_e;
    u32 exp_len;      // entries of EXP in the image: 2 qa (q <= 8192) or qa
    const i64 *e_ptr; // power: the one exponent of the call (device memory)
    u32 mu, c32;      // power: floor(2^32 / (q-1)), 2^32 mod (q-1)
};

struct MidPow {
    u32 em; // exponent reduced to [0, q-1) with a floor modulo, as Lut::pow_nz
    bool e_zero, e_neg;
};

enum { MID_NEG = 16, MID_RECIP = 17, MID_POW = 18 };

struct MidTabs {
    const u16 *lg, *ex, *ze;
};

// REDUCED: EXP holds q (not 2q) entries and every index is brought below q - 1 first (fields above 8192 elements, where the
// doubled table would not fit next to LOG)
template <int OP, bool REDUCED>
__device__ __forceinline__ u32 mid_op(const MidTabs &t, const MidDesc &d, const MidPow &pw, u32 a, u32 b, bool &bad)
{
    auto red = [&](u32 s) -> u32 { return REDUCED ? (s >= d.qm1 ? s - d.qm1 : s) : s; };
    if constexpr (OP == GFA_OP_MUL) { // multiply_ufunc.lookup: EXP[LOG[a] + LOG[b]], 0 if either is 0
        const u32 r = t.ex[red((u32)t.lg[a] + (u32)t.lg[b])];
        return (a == 0 || b == 0) ? 0u : r;
    } else if constexpr (OP == GFA_OP_DIV) { // divide_ufunc.lookup: EXP[(q-1) + LOG[a] - LOG[b]]
        bad |= b == 0;
        const u32 r = t.ex[red(d.qm1 + (u32)t.lg[a] - (u32)t.lg[b])];
        return (a == 0 || b == 0) ? 0u : r;
    } else if constexpr (OP == GFA_OP_ADD) { // add_ufunc.lookup (odd characteristic): EXP[m + ZECH[n - m]], m <= n the two logs
        const u32 la = t.lg[a], lb = t.lg[b];
        const u32 mm = min(la, lb), nn = max(la, lb);
        const u32 z = nn - mm;
        const u32 r = t.ex[red(mm + (u32)t.ze[z])];
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
        const u32 r = t.ex[red(mm + (u32)t.ze[z])];
        u32 res = cancel ? 0u : r;
        if (a == 0) res = t.ex[red(nn0)]; // rare: skipped by the whole wave almost always
        res = b == 0 ? a : res;
        return res;
    } else if constexpr (OP == MID_NEG) { // negative_ufunc.lookup: EXP[LOG[a] + ZECH_E]
        const u32 r = t.ex[red((u32)t.lg[a] + d.zech_e)];
        return a == 0 ? 0u : r;
    } else if constexpr (OP == MID_RECIP) { // reciprocal_ufunc.lookup: EXP[(q-1) - LOG[a]]
        bad |= a == 0;
        const u32 r = t.ex[red(d.qm1 - (u32)t.lg[a])];
        return a == 0 ? 0u : r;
    } else { // MID_POW, one exponent for the whole array: EXP[(LOG[a] * e) mod (q-1)] (power_ufunc.lookup, _lookup.py:247-270)
        const u32 x = (u32)t.lg[a] * pw.em;        // < 2^26
        u32 idx = x - __umulhi(x, d.mu) * d.qm1;   // in [0, 2(q-1)): the quotient estimate is short by at most one
        idx = idx >= d.qm1 ? idx - d.qm1 : idx;
        const u32 r = t.ex[idx];
        bad |= (a == 0) && pw.e_neg;
        return pw.e_zero ? 1u : (a == 0 ? 0u : r);
    }
}

// ESZ = bytes per stored element: 2 (uint16), 4 (uint32 / int32), 8 (int64 -- the `dtype=int` arrays of the reference's
// documentation, _domains/_array.py:445-451).  Wider storage only changes how many elements a 16-byte vector carries (8 / 4 / 2)
// and how they are unpacked; the tables, the index arithmetic and therefore the results are the same.
template <int ESZ> struct MidElem;
template <> struct MidElem<2> { typedef u16 T; };
template <> struct MidElem<4> { typedef u32 T; };
template <> struct MidElem<8> { typedef u64 T; };

template <int OP, int THREADS, bool REDUCED, int ESZ = 2>
__global__ __launch_bounds__(THREADS) void mid_kernel(MidDesc d, const typename MidElem<ESZ>::T *__restrict__ a, int sa,
                                                          const typename MidElem<ESZ>::T *__restrict__ b, int sb,
                                                          typename MidElem<ESZ>::T *__restrict__ out, i64 n, int32_t *err)
{
    typedef typename MidElem<ESZ>::T T;
    extern __shared__ __attribute__((aligned(16))) u16 mid_lds[];
    constexpr bool BINARY = OP <= GFA_OP_DIV;
    constexpr bool NEED_ZECH = OP == GFA_OP_ADD || OP == GFA_OP_SUB;
    constexpr int LOGV = ESZ == 2 ? 3 : ESZ == 4 ? 2 : 1; // log2(elements per 16-byte vector)
    const i64 nvec = n >> LOGV;
    const u32x4 *av = reinterpret_cast<const u32x4 *>(a);
    const u32x4 *bv = reinterpret_cast<const u32x4 *>(b);
    u32x4 *ov = reinterpret_cast<u32x4 *>(out);
    const i64 stride = (i64)gridDim.x * THREADS;
    i64 i = (i64)blockIdx.x * THREADS + threadIdx.x;
    u32x4 x = {0, 0, 0, 0}, y = {0, 0, 0, 0};
    if (!sa) { const u32 s = (u32)a[0]; x = ESZ == 2 ? u32x4{s, s, s, s} * 0x10001u : ESZ == 4 ? u32x4{s, s, s, s} : u32x4{s, 0, s, 0}; }
    if (BINARY && !sb) { const u32 s = (u32)b[0]; y = ESZ == 2 ? u32x4{s, s, s, s} * 0x10001u : ESZ == 4 ? u32x4{s, s, s, s} : u32x4{s, 0, s, 0}; }
    // the first operand vectors are requested before the tables are staged (as in tab8_binary_kernel)
    if (i < nvec) {
        if (sa) x = av[i];
        if (BINARY && sb) y = bv[i];
    }
    {
        const int words = (int)((d.qa + d.exp_len + (NEED_ZECH ? d.qa : 0u)) / 8u); // 16-byte units
        const uint4 *src = reinterpret_cast<const uint4 *>(d.image);
        uint4 *dst = reinterpret_cast<uint4 *>(mid_lds);
        for (int t = threadIdx.x; t < words; t += THREADS) dst[t] = src[t];
    }
    __syncthreads();
    MidTabs t;
    t.lg = mid_lds;
    t.ex = mid_lds + d.qa;
    t.ze = mid_lds + d.qa + d.exp_len;
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
        if constexpr (ESZ == 2) {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const u32 lo = mid_op<OP, REDUCED>(t, d, pw, cx[w] & 0xffffu, cy[w] & 0xffffu, bad);
                const u32 hi = mid_op<OP, REDUCED>(t, d, pw, cx[w] >> 16, cy[w] >> 16, bad);
                r[w] = lo | (hi << 16);
            }
        } else if constexpr (ESZ == 4) {
#pragma unroll
            for (int w = 0; w < 4; w++) r[w] = mid_op<OP, REDUCED>(t, d, pw, cx[w] & 0xffffu, cy[w] & 0xffffu, bad);
        } else { // field elements are below 2^16: the upper word of an int64 element is zero on the way in and on the way out
            r[0] = mid_op<OP, REDUCED>(t, d, pw, cx[0] & 0xffffu, cy[0] & 0xffffu, bad);
            r[2] = mid_op<OP, REDUCED>(t, d, pw, cx[2] & 0xffffu, cy[2] & 0xffffu, bad);
            r[1] = 0; r[3] = 0;
        }
        ov[i] = r;
    }
    for (i64 j = (nvec << LOGV) + (i64)blockIdx.x * THREADS + threadIdx.x; j < n; j += stride)
        out[j] = (T)mid_op<OP, REDUCED>(t, d, pw, (u32)a[sa ? j : 0] & 0xffffu, BINARY ? (u32)b[sb ? j : 0] & 0xffffu : 0u, bad);
    if constexpr (OP == GFA_OP_DIV || OP == MID_RECIP || OP == MID_POW) {
        if (__any(bad)) {
            if ((threadIdx.x & 63) == 0 && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
        }
    }
}

// x mod m for any 32-bit x, mu = floor(2^32 / m): the quotient estimate is short by at most one
__device__ __forceinline__ u32 mod_barrett(u32 x, u32 m, u32 mu)
{
    const u32 r = x - __umulhi(x, mu) * m;
    return r >= m ? r - m : r;
}

// e mod m with the sign of a floor modulo (Python's %), m < 2^16, c32 = 2^32 mod m
__device__ __forceinline__ u32 exponent_mod(i64 e, u32 m, u32 mu, u32 c32)
{
    const bool neg = e < 0;
    const u64 ue = neg ? (u64)0 - (u64)e : (u64)e;
    const u32 hi = mod_barrett((u32)(ue >> 32), m, mu), lo = mod_barrett((u32)ue, m, mu);
    const u32 r = mod_barrett(hi * c32 + lo, m, mu); // < 2^32: hi, c32, lo < 2^16
    return (neg && r != 0) ? m - r : r;
}

// np.power with one exponent per element (int64 array): EXP[(LOG[a] * (e mod (q-1))) mod (q-1)], tables in LDS as in mid_kernel
template <int THREADS>
__global__ __launch_bounds__(THREADS) void mid_powv_kernel(MidDesc d, const u16 *__restrict__ a, const i64 *__restrict__ e,
                                                           u16 *__restrict__ out, i64 n, int32_t *err)
{
    extern __shared__ __attribute__((aligned(16))) u16 mid_lds[];
    const i64 nvec = n >> 3;
    const u32x4 *av = reinterpret_cast<const u32x4 *>(a);
    const u32x4 *ev = reinterpret_cast<const u32x4 *>(e); // 4 vectors (8 exponents) per vector of a
    u32x4 *ov = reinterpret_cast<u32x4 *>(out);
    const i64 stride = (i64)gridDim.x * THREADS;
    i64 i = (i64)blockIdx.x * THREADS + threadIdx.x;
    {
        const int words = (int)((d.qa + d.exp_len) / 8u); // the index is below q - 1 whatever the length of EXP
        const uint4 *src = reinterpret_cast<const uint4 *>(d.image);
        uint4 *dst = reinterpret_cast<uint4 *>(mid_lds);
        for (int t = threadIdx.x; t < words; t += THREADS) dst[t] = src[t];
    }
    __syncthreads();
    const u16 *lg = mid_lds, *ex = mid_lds + d.qa;
    bool bad = false;
    auto one = [&](u32 x, i64 k) -> u32 {
        const u32 em = exponent_mod(k, d.qm1, d.mu, d.c32);
        const u32 r = ex[mod_barrett((u32)lg[x] * em, d.qm1, d.mu)];
        bad |= x == 0 && k < 0;
        return k == 0 ? 1u : (x == 0 ? 0u : r);
    };
    for (; i < nvec; i += stride) {
        const u32x4 cx = av[i];
        u32x4 r;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const u32x4 k2 = ev[4 * i + w]; // exponents of elements 2w, 2w + 1
            const u32 lo = one(cx[w] & 0xffffu, (i64)(((u64)k2[1] << 32) | k2[0]));
            const u32 hi = one(cx[w] >> 16, (i64)(((u64)k2[3] << 32) | k2[2]));
            r[w] = lo | (hi << 16);
        }
        ov[i] = r;
    }
    for (i64 j = (nvec << 3) + (i64)blockIdx.x * THREADS + threadIdx.x; j < n; j += stride) out[j] = (u16)one((u32)a[j], e[j]);
    if (__any(bad)) {
        if ((threadIdx.x & 63) == 0 && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
    }
}

// ------------------------------------------------------------------------------------------------
// 32768 < q <= 65536 on uint16 storage: LOG and EXP are 2q bytes each (up to 128 KiB) -- one of them fits in LDS, not both.
// One 1024-thread workgroup per CU works on tiles of 1024 * 8 * J elements in two phases: with LOG staged, every lane turns
// its J operand vectors into exponent indices ((LOG[a] +- LOG[b]) mod (q-1), or 0xFFFF for "the result is 0") that stay in
// registers; the workgroup then re-stages LDS with EXP[0 .. q-1) and every lane looks its indices up and stores.  The array
// is read once and written once; the 2 x 2q bytes of table re-staging per tile come out of L2.
// ------------------------------------------------------------------------------------------------
#ifndef GFA_B16_THREADS
#define GFA_B16_THREADS 512
#endif
constexpr int B16_THREADS = GFA_B16_THREADS;

template <int OP>
__device__ __forceinline__ u32 big16_index(const u16 *lg, const MidDesc &d, const MidPow &pw, u32 a, u32 b, bool &bad)
{
    u32 s;
    bool zero;
    if constexpr (OP == GFA_OP_MUL) {
        s = (u32)lg[a] + (u32)lg[b];
        zero = a == 0 || b == 0;
    } else if constexpr (OP == GFA_OP_DIV) {
        bad |= b == 0;
        s = d.qm1 + (u32)lg[a] - (u32)lg[b];
        zero = a == 0 || b == 0;
    } else if constexpr (OP == MID_RECIP) {
        bad |= a == 0;
        s = d.qm1 - (u32)lg[a];
        zero = a == 0;
    } else if constexpr (OP == MID_NEG) {
        s = (u32)lg[a] + d.zech_e;
        zero = a == 0;
    } else { // MID_POW
        s = mod_barrett((u32)lg[a] * pw.em, d.qm1, d.mu);
        bad |= a == 0 && pw.e_neg;
        zero = a == 0 && !pw.e_zero;
        s = pw.e_zero ? 0u : s;
    }
    s = s >= d.qm1 ? s - d.qm1 : s;
    return zero ? 0xffffu : s;
}

template <int OP, int J>
__global__ __launch_bounds__(B16_THREADS) void big16_kernel(MidDesc d, const u16 *__restrict__ a, int sa, const u16 *__restrict__ b,
                                                            int sb, u16 *__restrict__ out, i64 nvec, int32_t *err)
{
    extern __shared__ __attribute__((aligned(16))) u16 mid_lds[];
    constexpr bool BINARY = OP == GFA_OP_MUL || OP == GFA_OP_DIV;
    constexpr int SK = 131072 / 16 / B16_THREADS; // staging vectors per lane for the largest table
    const u32x4 *av = reinterpret_cast<const u32x4 *>(a);
    const u32x4 *bv = reinterpret_cast<const u32x4 *>(b);
    u32x4 *ov = reinterpret_cast<u32x4 *>(out);
    const int tid = threadIdx.x;
    const i64 tile_vecs = (i64)B16_THREADS * J;
    const i64 ntiles = (nvec + tile_vecs - 1) / tile_vecs;
    const int words = (int)(d.qa / 8u); // 16-byte units of one table
    const int K = (words + B16_THREADS - 1) / B16_THREADS;
    u32x4 xs = {0, 0, 0, 0}, ys = {0, 0, 0, 0};
    if (!sa) { const u32 s = a[0]; xs = u32x4{s, s, s, s} * 0x10001u; }
    if (BINARY && !sb) { const u32 s = b[0]; ys = u32x4{s, s, s, s} * 0x10001u; }
    MidPow pw{0, false, false};
    if constexpr (OP == MID_POW) {
        const i64 e = d.e_ptr[0];
        pw = MidPow{exponent_mod(e, d.qm1, d.mu, d.c32), e == 0, e < 0};
    }
    bool bad = false;
    const u32 voff = (u32)tid * 16u;
    // Buffer addressing throughout: one descriptor per operand and tile (base = first vector of the tile, size = what is left of
    // the array, at most the tile), the lane offset in one VGPR, the vector index in the scalar offset -- no per-vector 64-bit
    // addresses in VGPRs, and the hardware bounds check stands in for `v < nvec` (loads past the end return 0, stores are dropped).
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)d.image, 0, 4u * d.qa, 0x00020000); // LOG | EXP
    u32x4 S[SK], x[J], y[BINARY ? J : 1];
    auto fetch_table = [&](int which) { // into registers; written to LDS once the previous table is no longer needed
#pragma unroll
        for (int k = 0; k < SK; k++)
            if (k < K) S[k] = __builtin_amdgcn_raw_buffer_load_b128(rt, voff, (int)(which * 2u * d.qa) + k * B16_THREADS * 16, 0);
    };
    auto put_table = [&]() {
        uint4 *dst = reinterpret_cast<uint4 *>(mid_lds);
#pragma unroll
        for (int k = 0; k < SK; k++) {
            const int t = tid + k * B16_THREADS;
            if (k < K && t < words) dst[t] = uint4{S[k][0], S[k][1], S[k][2], S[k][3]};
        }
    };
    auto tile_bytes = [&](i64 tile) -> u32 {
        const i64 left = (nvec - tile * tile_vecs) * 16;
        return (u32)(left < tile_vecs * 16 ? left : tile_vecs * 16);
    };
    auto fetch_operands = [&](i64 tile) {
        const i64 vbase = tile * tile_vecs;
        const u32 nrec = tile_bytes(tile);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(av + (sa ? vbase : 0)), 0, sa ? nrec : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(bv + (sb ? vbase : 0)), 0, (BINARY && sb) ? nrec : 0u, 0x00020000);
#pragma unroll
        for (int j = 0; j < J; j++) {
            x[j] = xs;
            if (BINARY) y[j] = ys;
            if (sa) x[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, voff, j * B16_THREADS * 16, 0);
            if (BINARY && sb) y[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, voff, j * B16_THREADS * 16, 0);
        }
        if (!BINARY) y[0] = ys;
    };
    i64 tile = blockIdx.x;
    if (tile < ntiles) {
        fetch_table(0);
        fetch_operands(tile);
        for (;;) {
            put_table(); // LOG
            __syncthreads();
            fetch_table(1); // EXP travels while the logarithms are gathered
            const u32 nrec = tile_bytes(tile);
            u32x4 idx[J];
#pragma unroll
            for (int j = 0; j < J; j++) {
                bool bj = false;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const u32 lo = big16_index<OP>(mid_lds, d, pw, x[j][w] & 0xffffu, y[BINARY ? j : 0][w] & 0xffffu, bj);
                    const u32 hi = big16_index<OP>(mid_lds, d, pw, x[j][w] >> 16, y[BINARY ? j : 0][w] >> 16, bj);
                    idx[j][w] = lo | (hi << 16);
                }
                bad |= bj && (voff + (u32)(j * B16_THREADS * 16) < nrec); // vectors past the end of the array read zeros
            }
            __syncthreads(); // every wave is done with LOG
            put_table();     // EXP
            __syncthreads();
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(ov + tile * tile_vecs), 0, nrec, 0x00020000);
            const i64 next = tile + gridDim.x;
            if (next < ntiles) { // the next tile's LOG and operands travel while this tile's results are gathered and stored
                fetch_table(0);
                fetch_operands(next);
            }
#pragma unroll
            for (int j = 0; j < J; j++) {
                u32x4 r;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const u32 il = idx[j][w] & 0xffffu, ih = idx[j][w] >> 16;
                    const u32 rl = mid_lds[il], rh = mid_lds[ih];
                    r[w] = (il == 0xffffu ? 0u : rl) | ((ih == 0xffffu ? 0u : rh) << 16);
                }
                // The vector index goes into the VGPR offset, NOT the scalar offset: a buffer_store_dwordx4 with an SGPR soffset
                // followed at once by a VALU write of its data registers stored corrupted words on gfx950 (intermittently, words 0
                // of the vectors whose registers the next vector's index arithmetic reused).  LLVM inserts the wait states this
                // hazard needs only for the immediate-soffset form (its rule exempts SGPR soffsets); DESIGN.md section 4.2 (7).
                __builtin_amdgcn_raw_buffer_store_b128(r, ro, voff + (u32)(j * B16_THREADS * 16), 0, 0);
            }
            __syncthreads(); // every wave is done with EXP
            if (next >= ntiles) break;
            tile = next;
        }
    }
    if constexpr (OP != GFA_OP_MUL && OP != MID_NEG) {
        if (__any(bad)) {
            if ((tid & 63) == 0 && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
        }
    }
}

// Sums and differences in odd characteristic above 32768 elements: three tables, three phases per tile.
//   A (LOG):  every element becomes (m, z) = (smaller logarithm, difference of the logarithms); a - b is a + (-b) with
//             LOG[-b] = (LOG[b] + ZECH_E) mod (q-1).  An operand that is zero makes the element a pass-through (z = 0xFFFF, m = the
//             logarithm of the result, or 0xFFFF for "result 0").
//   B (ZECH): index = (m + ZECH[z]) mod (q-1); z = ZECH_E (the operands cancel) gives 0xFFFF.
//   C (EXP):  result = EXP[index], 0 for 0xFFFF.
// (m, z) take the registers the operands came in, the index then replaces m.
template <int OP, int J, int THREADS>
__global__ __launch_bounds__(THREADS) void big16_addsub_kernel(MidDesc d, const u16 *__restrict__ a, int sa, const u16 *__restrict__ b,
                                                               int sb, u16 *__restrict__ out, i64 nvec)
{
    extern __shared__ __attribute__((aligned(16))) u16 mid_lds[];
    const u32x4 *av = reinterpret_cast<const u32x4 *>(a);
    const u32x4 *bv = reinterpret_cast<const u32x4 *>(b);
    u32x4 *ov = reinterpret_cast<u32x4 *>(out);
    const int tid = threadIdx.x;
    const i64 tile_vecs = (i64)THREADS * J;
    const i64 ntiles = (nvec + tile_vecs - 1) / tile_vecs;
    const int words = (int)(d.qa / 8u);
    u32x4 xs = {0, 0, 0, 0}, ys = {0, 0, 0, 0};
    if (!sa) { const u32 s = a[0]; xs = u32x4{s, s, s, s} * 0x10001u; }
    if (!sb) { const u32 s = b[0]; ys = u32x4{s, s, s, s} * 0x10001u; }
    const u32 voff = (u32)tid * 16u;
    auto stage = [&](int which) { // table `which` of the image (0 LOG, 1 EXP, 2 ZECH) into LDS
        const uint4 *src = reinterpret_cast<const uint4 *>(d.image + (size_t)which * d.qa);
        uint4 *dst = reinterpret_cast<uint4 *>(mid_lds);
#pragma unroll 4
        for (int t = tid; t < words; t += THREADS) dst[t] = src[t];
    };
    // phase A on one element: (m, z)
    auto logs = [&](u32 av_, u32 bv_, u32 &m, u32 &z) {
        const u32 la = mid_lds[av_];
        u32 lb = mid_lds[bv_];
        if constexpr (OP == GFA_OP_SUB) {
            lb += d.zech_e;
            lb = lb >= d.qm1 ? lb - d.qm1 : lb;
        }
        const u32 mm = min(la, lb), nn = max(la, lb);
        m = mm;
        z = nn - mm;
        if (bv_ == 0) { m = av_ == 0 ? 0xffffu : la; z = 0xffffu; }
        else if (av_ == 0) { m = lb; z = 0xffffu; }
    };
    for (i64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const i64 vbase = tile * tile_vecs;
        const i64 left = (nvec - vbase) * 16;
        const u32 nrec = (u32)(left < tile_vecs * 16 ? left : tile_vecs * 16);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(av + (sa ? vbase : 0)), 0, sa ? nrec : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(bv + (sb ? vbase : 0)), 0, sb ? nrec : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(ov + vbase), 0, nrec, 0x00020000);
        u32x4 x[J], y[J];
#pragma unroll
        for (int j = 0; j < J; j++) {
            x[j] = xs;
            y[j] = ys;
            if (sa) x[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, voff, j * THREADS * 16, 0);
            if (sb) y[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, voff, j * THREADS * 16, 0);
        }
        stage(0); // LOG
        __syncthreads();
#pragma unroll
        for (int j = 0; j < J; j++) {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                u32 m0, z0, m1, z1;
                logs(x[j][w] & 0xffffu, y[j][w] & 0xffffu, m0, z0);
                logs(x[j][w] >> 16, y[j][w] >> 16, m1, z1);
                x[j][w] = m0 | (m1 << 16);
                y[j][w] = z0 | (z1 << 16);
            }
        }
        __syncthreads();
        stage(2); // ZECH
        __syncthreads();
#pragma unroll
        for (int j = 0; j < J; j++) {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                u32 r = 0;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const u32 m = (x[j][w] >> (16 * h)) & 0xffffu, z = (y[j][w] >> (16 * h)) & 0xffffu;
                    u32 s = m + (u32)mid_lds[z == 0xffffu ? 0u : z];
                    s = s >= d.qm1 ? s - d.qm1 : s;
                    s = z == d.zech_e ? 0xffffu : s;
                    s = z == 0xffffu ? m : s;
                    r |= s << (16 * h);
                }
                x[j][w] = r;
            }
        }
        __syncthreads();
        stage(1); // EXP
        __syncthreads();
#pragma unroll
        for (int j = 0; j < J; j++) {
            u32x4 r;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const u32 il = x[j][w] & 0xffffu, ih = x[j][w] >> 16;
                const u32 rl = mid_lds[il], rh = mid_lds[ih];
                r[w] = (il == 0xffffu ? 0u : rl) | ((ih == 0xffffu ? 0u : rh) << 16);
            }
            __builtin_amdgcn_raw_buffer_store_b128(r, ro, voff + (u32)(j * THREADS * 16), 0, 0); // VGPR offset: see big16_kernel
        }
        __syncthreads();
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

template <int OP, int ESZ>
int mid_launch_e(const MidDesc &d, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st, int32_t *err)
{
    typedef typename MidElem<ESZ>::T T;
    constexpr bool NEED_ZECH = OP == GFA_OP_ADD || OP == GFA_OP_SUB;
    constexpr int LOGV = ESZ == 2 ? 3 : ESZ == 4 ? 2 : 1;
    const size_t lds = ((size_t)d.qa + d.exp_len + (NEED_ZECH ? d.qa : 0)) * sizeof(u16);
    const bool reduced = d.exp_len == d.qa;
    static bool attr[2] = {false, false};
    const int cus = mid_num_cus();
    const i64 vec_blocks = (n >> LOGV);
    if (reduced) { // 8192 < q <= 32768: up to 160 KiB of tables, one 16-wave workgroup per CU
        auto k = mid_kernel<OP, 1024, true, ESZ>;
        if (!attr[1]) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)); attr[1] = true; }
        i64 blocks = (vec_blocks + 1023) / 1024;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k, dim3((int)(blocks < cus ? blocks : cus)), dim3(1024), lds, st, d, (const T *)a, (int)sa, (const T *)b, (int)sb,
                           (T *)out, n, err);
    } else {
        auto k = mid_kernel<OP, MID_THREADS, false, ESZ>;
        if (!attr[0]) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); attr[0] = true; }
        // 160 KiB of LDS and 32 waves per CU: up to four 8-wave workgroups when the tables are small
        i64 per_cu = (i64)(160 * 1024) / (i64)(lds + 1024);
        per_cu = per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu;
        i64 blocks = (vec_blocks + MID_THREADS - 1) / MID_THREADS;
        const i64 cap = (i64)cus * per_cu;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k, dim3((int)(blocks < cap ? blocks : cap)), dim3(MID_THREADS), lds, st, d, (const T *)a, (int)sa, (const T *)b,
                           (int)sb, (T *)out, n, err);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

template <int OP>
int mid_launch(const MidDesc &d, int dtype, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st, int32_t *err)
{
    switch (dtype) {
    case GFA_U16: return mid_launch_e<OP, 2>(d, a, sa, b, sb, out, n, st, err);
    case GFA_U32: return mid_launch_e<OP, 4>(d, a, sa, b, sb, out, n, st, err);
    case GFA_U64: return mid_launch_e<OP, 8>(d, a, sa, b, sb, out, n, st, err);
    default: return GFA_ERR_UNSUPPORTED;
    }
}

// ---- the same two phases as TWO streaming kernels (r04), for arrays large enough to hide a second launch ----
// big16_kernel re-stages 2 x 2q bytes of table per tile behind three workgroup barriers (PMC: vector ALU 27 %, LDS 40 % busy:
// barrier-bound, 0.30-0.38 of the roofline).  Here each table is staged ONCE per persistent workgroup and the array streams through it:
// pass A turns the operands into exponent indices (two gathers per element, 2 B written per element), pass B looks the indices up
// (one gather).  10 instead of 6 bytes per element cross the memory system, but the 2-byte index array of up to ~1e8 elements is
// written and re-read through the 256 MiB Infinity Cache, and neither pass has a barrier in its loop.  Measured at 5e7 elements
// (profiles/r04_ew_big16_split.txt): GF(3^10) mul 494 -> 540 Gop/s, GF(65521) lookup-mode mul 469 -> 533, div 459 -> 519.
template <int OP, int T>
__global__ __launch_bounds__(T) void big16_index_kernel(MidDesc d, const u16 *__restrict__ a, int sa, const u16 *__restrict__ b, int sb,
                                                        u16 *__restrict__ idx_out, i64 nvec, int32_t *err)
{
    extern __shared__ __attribute__((aligned(16))) u16 mid_lds[];
    constexpr bool BINARY = OP == GFA_OP_MUL || OP == GFA_OP_DIV;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(d.image); // LOG: qa entries
        uint4 *dst = reinterpret_cast<uint4 *>(mid_lds);
        for (int i = threadIdx.x; i < (int)(d.qa / 8u); i += T) dst[i] = src[i];
    }
    MidPow pw{0, false, false};
    if constexpr (OP == MID_POW) {
        const i64 e = d.e_ptr[0];
        pw = MidPow{exponent_mod(e, d.qm1, d.mu, d.c32), e == 0, e < 0};
    }
    u32x4 xs = {0, 0, 0, 0}, ys = {0, 0, 0, 0};
    if (!sa) { const u32 s = a[0]; xs = u32x4{s, s, s, s} * 0x10001u; }
    if (BINARY && !sb) { const u32 s = b[0]; ys = u32x4{s, s, s, s} * 0x10001u; }
    const u32x4 *av = reinterpret_cast<const u32x4 *>(a), *bv = reinterpret_cast<const u32x4 *>(b);
    u32x4 *ov = reinterpret_cast<u32x4 *>(idx_out);
    bool bad = false;
    __syncthreads();
    const i64 stride = (i64)gridDim.x * T;
    i64 i = (i64)blockIdx.x * T + threadIdx.x;
    u32x4 x = xs, y = ys;
    if (i < nvec) { if (sa) x = av[i]; if (BINARY && sb) y = bv[i]; }
    while (i < nvec) {
        const i64 nx = i + stride;
        u32x4 xn = xs, yn = ys;
        if (nx < nvec) { if (sa) xn = av[nx]; if (BINARY && sb) yn = bv[nx]; } // the next vectors travel while this one is gathered
        u32x4 r;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const u32 lo = big16_index<OP>(mid_lds, d, pw, x[w] & 0xffffu, y[w] & 0xffffu, bad);
            const u32 hi = big16_index<OP>(mid_lds, d, pw, x[w] >> 16, y[w] >> 16, bad);
            r[w] = lo | (hi << 16);
        }
        ov[i] = r;
        x = xn; y = yn; i = nx;
    }
    if constexpr (OP != GFA_OP_MUL && OP != MID_NEG) {
        if (__any(bad)) {
            if ((threadIdx.x & 63) == 0 && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
        }
    }
}

template <int T>
__global__ __launch_bounds__(T) void big16_exp_kernel(MidDesc d, const u16 *__restrict__ idx, u16 *__restrict__ out, i64 nvec)
{
    extern __shared__ __attribute__((aligned(16))) u16 mid_lds[];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(d.image + d.qa); // EXP: the second table of the image
        uint4 *dst = reinterpret_cast<uint4 *>(mid_lds);
        for (int i = threadIdx.x; i < (int)(d.qa / 8u); i += T) dst[i] = src[i];
    }
    const u32x4 *iv = reinterpret_cast<const u32x4 *>(idx);
    u32x4 *ov = reinterpret_cast<u32x4 *>(out);
    __syncthreads();
    const i64 stride = (i64)gridDim.x * T;
    i64 i = (i64)blockIdx.x * T + threadIdx.x;
    u32x4 x = {0, 0, 0, 0};
    if (i < nvec) x = iv[i];
    while (i < nvec) {
        const i64 nx = i + stride;
        u32x4 xn = {0, 0, 0, 0};
        if (nx < nvec) xn = iv[nx];
        u32x4 r;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const u32 il = x[w] & 0xffffu, ih = x[w] >> 16;
            const u32 rl = mid_lds[il == 0xffffu ? 0u : il], rh = mid_lds[ih == 0xffffu ? 0u : ih];
            r[w] = (il == 0xffffu ? 0u : rl) | ((ih == 0xffffu ? 0u : rh) << 16);
        }
        ov[i] = r;
        x = xn; i = nx;
    }
}

// ---- 32768 < q <= 65536, reciprocals and (where the product is explicit) quotients through ONE table (r06) ----
// LOG + EXP of these fields do not fit LDS together, which is why big16_kernel stages them in turn (0.38-0.40); the 2q-byte table
// INV[x] = 1 / x (the fourth table of the image, gfa_field::ensure_device) does, once per persistent workgroup, and the array streams
// through it as through big16_exp_kernel.  MODE 0: 1 / b.  MODE 1: a / b = a * INV[b] mod p in a PRIME field (one 32-bit product, Barrett).
// MODE 2: the same in GF(2^16): Bin::clmul16_lo / _hi (nine integer multiplies per element, two elements per register) and two
// 256-entry reduction tables behind INV.  Same values as divide_ufunc / reciprocal_ufunc (_lookup.py:176-235, _calculate.py:447-513).
// MODE 3: x ** e for ONE exponent: the table is P[x] = x^e, filled per call by big16_pow_table_kernel (q table look-ups) -- the array then needs one
// gather per element where LOG and EXP staged in turn ran at 0.34; 0 ** negative is flagged (power_ufunc.lookup, _lookup.py:247-270).
template <int MODE, int T>
__global__ __launch_bounds__(T) void big16_inv_kernel(MidDesc d, const u16 *__restrict__ table, u32 p, u32 mu_p, u64 irr, const u16 *__restrict__ a, int sa,
                                                      const u16 *__restrict__ b, int sb, u16 *__restrict__ out, i64 nvec, int32_t *err)
{
    extern __shared__ __attribute__((aligned(16))) u16 mid_lds[];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(table); // INV (the image's fourth table) or P: qa entries
        uint4 *dst = reinterpret_cast<uint4 *>(mid_lds);
        for (int i = threadIdx.x; i < (int)(d.qa / 8u); i += T) dst[i] = src[i];
    }
    const u16 *R = mid_lds + d.qa; // MODE 2: h(x) x^16 mod f | h(x) x^24 mod f
    if constexpr (MODE == 2) {
        u16 *Rw = mid_lds + d.qa;
        for (int t = threadIdx.x; t < 256; t += T) {
            Rw[t] = (u16)Bin::reduce_bits((u64)t << 16, 16, 8, irr);
            Rw[256 + t] = (u16)Bin::reduce_bits((u64)t << 24, 16, 16, irr);
        }
    }
    u32x4 xs = {0, 0, 0, 0}, ys = {0, 0, 0, 0};
    if (!sb) { const u32 s = b[0]; xs = u32x4{s, s, s, s} * 0x10001u; }
    if ((MODE == 1 || MODE == 2) && !sa) { const u32 s = a[0]; ys = u32x4{s, s, s, s} * 0x10001u; }
    const u32x4 *av = reinterpret_cast<const u32x4 *>(a), *bv = reinterpret_cast<const u32x4 *>(b);
    u32x4 *ov = reinterpret_cast<u32x4 *>(out);
    bool bad = false;
    __syncthreads();
    auto prime_mul = [&](u32 x, u32 y) -> u32 { // x, y < p < 2^16
        const u32 t = x * y, r = t - Bin::mulhi32(t, mu_p) * p; // the estimate is short by at most one
        const u32 r2 = r - p;
        return r2 < r ? r2 : r;
    };
    auto fold16 = [&](u32 P) -> u32 { return (P & 0xffffu) ^ R[(P >> 16) & 0xffu] ^ R[256 + (P >> 24)]; }; // P below 2^31
    const i64 stride = (i64)gridDim.x * T;
    i64 i = (i64)blockIdx.x * T + threadIdx.x;
    u32x4 x = xs, y = ys;
    if (i < nvec) { if (sb) x = bv[i]; if ((MODE == 1 || MODE == 2) && sa) y = av[i]; }
    while (i < nvec) {
        const i64 nx = i + stride;
        u32x4 xn = xs, yn = ys;
        if (nx < nvec) { if (sb) xn = bv[nx]; if ((MODE == 1 || MODE == 2) && sa) yn = av[nx]; } // the next vectors travel while this one is gathered
        u32x4 r;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const u32 bl = x[w] & 0xffffu, bh = x[w] >> 16;
            bad |= bl == 0u || bh == 0u;
            const u32 il = mid_lds[bl], ih = mid_lds[bh];
            if constexpr (MODE == 0 || MODE == 3) r[w] = il | (ih << 16);
            else if constexpr (MODE == 1) r[w] = prime_mul(y[w] & 0xffffu, il) | (prime_mul(y[w] >> 16, ih) << 16);
            else {
                const u32 iw = il | (ih << 16);
                r[w] = fold16(Bin::clmul16_lo(y[w], iw)) | (fold16(Bin::clmul16_hi(y[w], iw)) << 16);
            }
        }
        ov[i] = r;
        x = xn; y = yn; i = nx;
    }
    if (MODE == 3) bad = bad && d.e_ptr[0] < 0;
    if (__any(bad)) {
        if ((threadIdx.x & 63) == 0 && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
    }
}

// P[x] = x ** e for every element x of the field (the exponent in device memory, d.e_ptr): EXP[(LOG[x] * e) mod (q - 1)], 0 ** e = 0, x ** 0 = 1
__global__ __launch_bounds__(256) void big16_pow_table_kernel(MidDesc d, u16 *__restrict__ tab)
{
    const i64 e = d.e_ptr[0];
    const MidPow pw{exponent_mod(e, d.qm1, d.mu, d.c32), e == 0, e < 0};
    const u16 *lg = d.image, *ex = d.image + d.qa;
    for (u32 x = blockIdx.x * 256 + threadIdx.x; x < d.qa; x += gridDim.x * 256) {
        u32 r = 0;
        if (x < d.q) {
            u32 s = mod_barrett((u32)lg[x] * pw.em, d.qm1, d.mu);
            s = s >= d.qm1 ? s - d.qm1 : s;
            r = pw.e_zero ? 1u : (x == 0 ? 0u : (u32)ex[s]);
        }
        tab[x] = (u16)r;
    }
}

template <int MODE>
int big16_inv_launch(const MidDesc &d, const FieldDev &lut, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st, int32_t *err)
{
    constexpr int T = 1024;
    auto k = big16_inv_kernel<MODE, T>;
    static bool attr = false;
    if (!attr) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 1024)); attr = true; }
    const size_t lds = (size_t)d.qa * sizeof(u16) + (MODE == 2 ? 1024 : 0);
    const i64 nvec = n >> 3;
    const i64 blocks = (nvec + T - 1) / T;
    const int cus = mid_num_cus();
    if (MODE == 3) { // the per-call power table: qa entries of stream-ordered scratch
        u16 *tab = nullptr;
        if (gfa::scratch_alloc((void **)&tab, (size_t)d.qa * sizeof(u16), st) != hipSuccess) { (void)hipGetLastError(); return GFA_ERR_UNSUPPORTED; }
        hipLaunchKernelGGL(big16_pow_table_kernel, dim3((int)(d.qa / 256u)), dim3(256), 0, st, d, tab);
        hipLaunchKernelGGL(k, dim3((int)(blocks < cus ? blocks : cus)), dim3(T), lds, st, d, (const u16 *)tab, 0u, 0u, (u64)0, (const u16 *)nullptr, 0, (const u16 *)b, 1,
                           (u16 *)out, nvec, err);
        const hipError_t le = hipGetLastError();
        GFA_HIP(gfa::scratch_free(tab, st));
        GFA_HIP(le);
        return GFA_OK;
    }
    hipLaunchKernelGGL(k, dim3((int)(blocks < cus ? blocks : cus)), dim3(T), lds, st, d, d.image + 3 * (size_t)d.qa, (u32)lut.p, (u32)(0x100000000ull / lut.p), (u64)lut.irr,
                       (const u16 *)a, (int)sa, (const u16 *)b, (int)sb, (u16 *)out, nvec, err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// np.power with one exponent PER ELEMENT (int64 array) for 32768 < q <= 65536 (r06): the first of the two streaming passes with LOG in LDS and
// the exponent array read beside the operands -- index = (LOG[x] (e mod (q - 1))) mod (q - 1), 0xFFFF where the result is 0, 0 where e == 0 --
// then big16_exp_kernel.  The generic table kernel gathered both tables from L2 (0.13 of the roofline).
template <int T>
__global__ __launch_bounds__(T) void big16_powv_index_kernel(MidDesc d, const u16 *__restrict__ a, const i64 *__restrict__ e, u16 *__restrict__ idx_out, i64 nvec,
                                                             int32_t *err)
{
    extern __shared__ __attribute__((aligned(16))) u16 mid_lds[];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(d.image); // LOG: qa entries
        uint4 *dst = reinterpret_cast<uint4 *>(mid_lds);
        for (int i = threadIdx.x; i < (int)(d.qa / 8u); i += T) dst[i] = src[i];
    }
    const u32x4 *av = reinterpret_cast<const u32x4 *>(a);
    const u32x4 *ev = reinterpret_cast<const u32x4 *>(e); // 4 vectors (8 exponents) per vector of a
    u32x4 *ov = reinterpret_cast<u32x4 *>(idx_out);
    bool bad = false;
    __syncthreads();
    auto one = [&](u32 x, i64 k) -> u32 {
        const u32 em = exponent_mod(k, d.qm1, d.mu, d.c32);
        u32 s = mod_barrett((u32)mid_lds[x] * em, d.qm1, d.mu);
        s = s >= d.qm1 ? s - d.qm1 : s;
        bad |= x == 0 && k < 0;
        return k == 0 ? 0u : (x == 0 ? 0xffffu : s);
    };
    const i64 stride = (i64)gridDim.x * T;
    for (i64 i = (i64)blockIdx.x * T + threadIdx.x; i < nvec; i += stride) {
        const u32x4 cx = av[i];
        u32x4 k2[4];
#pragma unroll
        for (int w = 0; w < 4; w++) k2[w] = ev[4 * i + w];
        u32x4 r;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const u32 lo = one(cx[w] & 0xffffu, (i64)(((u64)k2[w][1] << 32) | k2[w][0]));
            const u32 hi = one(cx[w] >> 16, (i64)(((u64)k2[w][3] << 32) | k2[w][2]));
            r[w] = lo | (hi << 16);
        }
        ov[i] = r;
    }
    if (__any(bad)) {
        if ((threadIdx.x & 63) == 0 && err) atomicOr(err, GFA_DEVERR_ZERO_DIVISION);
    }
}

template <int OP>
int big16_launch(const MidDesc &d, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st, int32_t *err)
{
    const size_t lds = (size_t)d.qa * sizeof(u16);
    const i64 nvec = n >> 3;
    const int cus = mid_num_cus();
    // JB vectors per lane (tiles of 4096 * JB elements) when that
    // still gives every CU two tiles, else two vectors per lane
    constexpr int JB = ((OP == GFA_OP_MUL || OP == GFA_OP_DIV) ? 4 : 8) * 512 / B16_THREADS; // two operand streams: half the vectors per lane (registers)
    const bool big = (nvec + (i64)B16_THREADS * JB - 1) / ((i64)B16_THREADS * JB) >= 2 * (i64)cus;
    static bool attr[2] = {false, false};
    // products and quotients of >= 2^22 elements: two streaming passes through an index array (see big16_index_kernel): 0.36-0.37 -> 0.40;
    // reciprocal / power / negative LOSE that way (4 instead of 8 B/element to begin with: 0.38 -> 0.33) and keep the fused kernel
    if ((OP == GFA_OP_MUL || OP == GFA_OP_DIV) && nvec >= ((i64)1 << 19)) {
        constexpr int T = 1024;
        auto ka = big16_index_kernel<OP, T>;
        auto kb = big16_exp_kernel<T>;
        static bool sattr = false;
        if (!sattr) {
            GFA_HIP(hipFuncSetAttribute((const void *)ka, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            GFA_HIP(hipFuncSetAttribute((const void *)kb, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            sattr = true;
        }
        // The gain rests on the 2-byte index array staying in the 256 MiB Infinity Cache between the two kernels, so the
        // array is walked in slices of 2^26 elements (a 128 MiB index slice) that share ONE work buffer: a multi-GB operand
        // costs no more scratch than a small one.  Without a work buffer the fused kernel below does the job.
        const i64 slice = std::min<i64>(nvec, (i64)1 << 23);
        u16 *idx = nullptr;
        if (gfa::scratch_alloc((void **)&idx, (size_t)slice * 16, st) == hipSuccess) {
            for (i64 v0 = 0; v0 < nvec; v0 += slice) {
                const i64 cnt = std::min<i64>(slice, nvec - v0);
                hipLaunchKernelGGL(ka, dim3(cus), dim3(T), lds, st, d, (const u16 *)a + (sa ? v0 * 8 : 0), (int)sa, (const u16 *)b + (sb ? v0 * 8 : 0),
                                   (int)sb, idx, cnt, err);
                hipLaunchKernelGGL(kb, dim3(cus), dim3(T), lds, st, d, (const u16 *)idx, (u16 *)out + v0 * 8, cnt);
            }
            const hipError_t le = hipGetLastError();
            GFA_HIP(gfa::scratch_free(idx, st));
            GFA_HIP(le);
            return GFA_OK;
        }
        (void)hipGetLastError(); // allocation failed: clear the sticky error and take the fused kernel, which needs no scratch
    }
    if (big) {
        auto k = big16_kernel<OP, JB>;
        if (!attr[1]) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); attr[1] = true; }
        hipLaunchKernelGGL(k, dim3(cus), dim3(B16_THREADS), lds, st, d, (const u16 *)a, (int)sa, (const u16 *)b, (int)sb, (u16 *)out, nvec, err);
    } else {
        auto k = big16_kernel<OP, 2>;
        if (!attr[0]) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); attr[0] = true; }
        const i64 tiles = (nvec + (i64)B16_THREADS * 2 - 1) / ((i64)B16_THREADS * 2);
        hipLaunchKernelGGL(k, dim3((int)(tiles < cus ? tiles : cus)), dim3(B16_THREADS), lds, st, d, (const u16 *)a, (int)sa, (const u16 *)b,
                           (int)sb, (u16 *)out, nvec, err);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

template <int OP>
int big16_addsub_launch(const MidDesc &d, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st)
{
    constexpr int T = 1024; // 16 waves: the three phases are latency chains (gather, barrier, stage, barrier)
    const size_t lds = (size_t)d.qa * sizeof(u16);
    const i64 nvec = n >> 3;
    const int cus = mid_num_cus();
    static bool attr = false;
    auto k = big16_addsub_kernel<OP, 2, T>;
    if (!attr) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); attr = true; }
    const i64 tiles = (nvec + (i64)T * 2 - 1) / ((i64)T * 2);
    hipLaunchKernelGGL(k, dim3((int)(tiles < cus ? tiles : cus)), dim3(T), lds, st, d, (const u16 *)a, (int)sa, (const u16 *)b, (int)sb, (u16 *)out,
                       nvec);
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
    d.exp_len = d.q <= 8192 ? 2 * d.qa : d.qa;
    d.qm1 = lut.qm1;
    d.zech_e = lut.zech_e;
    d.mu = (u32)(0x100000000ull / d.qm1);
    d.c32 = (u32)(0x100000000ull % d.qm1);
    return d;
}

} // namespace

namespace gfa {

// arrays below this many elements stay on the generic kernels (staging 8q bytes per workgroup would dominate)
constexpr i64 MID_MIN_N = (i64)1 << 17;

bool mid_eligible(const FieldDev &calc, const void *image, int dtype, i64 n)
{
    return image != nullptr && (dtype == GFA_U16 || dtype == GFA_U32 || dtype == GFA_U64) && calc.q > 256 && calc.q <= 32768 &&
           n >= MID_MIN_N;
}

// sums / differences / negation in odd characteristic need ZECH next to LOG and EXP: 6q bytes of LDS above 8192 elements
bool mid_has_zech_room(const FieldDev &calc) { return calc.q <= 8192 || 6 * ((calc.q + 7) & ~7ull) <= 160 * 1024; }

int mid_binary(const FieldDev &lut, const void *image, int dtype, int op, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n,
               hipStream_t st, int32_t *err)
{
    if (!al16(out) || (sa && !al16(a)) || (sb && !al16(b))) return GFA_ERR_UNSUPPORTED;
    const MidDesc d = make_desc(lut, (const u16 *)image);
    switch (op) {
    case GFA_OP_ADD: return mid_launch<GFA_OP_ADD>(d, dtype, a, sa, b, sb, out, n, st, err);
    case GFA_OP_SUB: return mid_launch<GFA_OP_SUB>(d, dtype, a, sa, b, sb, out, n, st, err);
    case GFA_OP_MUL: return mid_launch<GFA_OP_MUL>(d, dtype, a, sa, b, sb, out, n, st, err);
    case GFA_OP_DIV: return mid_launch<GFA_OP_DIV>(d, dtype, a, sa, b, sb, out, n, st, err);
    default: return GFA_ERR_UNSUPPORTED;
    }
}

int mid_unary(const FieldDev &lut, const void *image, int dtype, int op, const void *a, void *out, i64 n, hipStream_t st, int32_t *err)
{
    if (!al16(out) || !al16(a)) return GFA_ERR_UNSUPPORTED;
    const MidDesc d = make_desc(lut, (const u16 *)image);
    if (op == GFA_OP_NEG) return mid_launch<MID_NEG>(d, dtype, a, 1, a, 0, out, n, st, err);
    if (op == GFA_OP_RECIP) return mid_launch<MID_RECIP>(d, dtype, a, 1, a, 0, out, n, st, err);
    return GFA_ERR_UNSUPPORTED;
}

int mid_power(const FieldDev &lut, const void *image, int dtype, const void *a, const i64 *e, void *out, i64 n, hipStream_t st, int32_t *err)
{
    if (!al16(out) || !al16(a)) return GFA_ERR_UNSUPPORTED;
    MidDesc d = make_desc(lut, (const u16 *)image);
    d.e_ptr = e;
    return mid_launch<MID_POW>(d, dtype, a, 1, a, 0, out, n, st, err);
}

int mid_power_each(const FieldDev &lut, const void *image, const void *a, const i64 *e, void *out, i64 n, hipStream_t st, int32_t *err)
{
    if (!al16(out) || !al16(a) || !al16(e)) return GFA_ERR_UNSUPPORTED;
    const MidDesc d = make_desc(lut, (const u16 *)image);
    const size_t lds = ((size_t)d.qa + d.exp_len) * sizeof(u16);
    static bool attr = false;
    if (d.exp_len == d.qa) { // above 8192 elements: one 16-wave workgroup per CU
        auto k = mid_powv_kernel<1024>;
        if (!attr) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)); attr = true; }
        i64 blocks = ((n >> 3) + 1023) / 1024;
        const i64 cap = mid_num_cus();
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k, dim3((int)(blocks < cap ? blocks : cap)), dim3(1024), lds, st, d, (const u16 *)a, e, (u16 *)out, n, err);
    } else {
        i64 per_cu = (i64)(160 * 1024) / (i64)(lds + 1024);
        per_cu = per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu;
        i64 blocks = ((n >> 3) + MID_THREADS - 1) / MID_THREADS;
        const i64 cap = (i64)mid_num_cus() * per_cu;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(mid_powv_kernel<MID_THREADS>, dim3((int)(blocks < cap ? blocks : cap)), dim3(MID_THREADS), lds, st, d, (const u16 *)a, e,
                           (u16 *)out, n, err);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// ---- above 32768 elements (and sums without room for ZECH above 8192): staged-table kernels; they cover the first n & ~7 elements, the caller runs the generic kernels on the rest
constexpr i64 BIG16_MIN_N = (i64)1 << 19;

bool big16_eligible(const FieldDev &calc, const void *image, int dtype, i64 n)
{
    return image != nullptr && dtype == GFA_U16 && calc.q > 8192 && calc.q <= 65536 && n >= BIG16_MIN_N;
}

int big16_run(const FieldDev &lut, const void *image, int op, const void *a, i64 sa, const void *b, i64 sb, const i64 *e, void *out, i64 n,
              hipStream_t st, int32_t *err)
{
    if (!al16(out) || (sa && !al16(a)) || (b && sb && !al16(b))) return GFA_ERR_UNSUPPORTED;
    MidDesc d = make_desc(lut, (const u16 *)image);
    d.e_ptr = e;
    switch (op) {
    case GFA_OP_ADD: return big16_addsub_launch<GFA_OP_ADD>(d, a, sa, b, sb, out, n, st);
    case GFA_OP_SUB: return big16_addsub_launch<GFA_OP_SUB>(d, a, sa, b, sb, out, n, st);
    case GFA_OP_MUL: return big16_launch<GFA_OP_MUL>(d, a, sa, b, sb, out, n, st, err);
    case GFA_OP_DIV:
        if (lut.q > 32768 && lut.m == 1) return big16_inv_launch<1>(d, lut, a, sa, b, sb, out, n, st, err);           // r06: a * INV[b], explicit product
        if (lut.q > 32768 && lut.p == 2 && lut.m == 16) return big16_inv_launch<2>(d, lut, a, sa, b, sb, out, n, st, err);
        return big16_launch<GFA_OP_DIV>(d, a, sa, b, sb, out, n, st, err);
    case GFA_OP_NEG: return big16_launch<MID_NEG>(d, a, 1, a, 0, out, n, st, err);
    case GFA_OP_RECIP:
        if (lut.q > 32768) return big16_inv_launch<0>(d, lut, nullptr, 0, a, 1, out, n, st, err); // r06: one table instead of two staged in turn
        return big16_launch<MID_RECIP>(d, a, 1, a, 0, out, n, st, err);
    case GFA_OP_POW:
        if (lut.q > 32768) { // r06: a per-call table of x ** e, then one gather per element
            const int rc = big16_inv_launch<3>(d, lut, nullptr, 0, a, 1, out, n, st, err);
            if (rc != GFA_ERR_UNSUPPORTED) return rc;
        }
        return big16_launch<MID_POW>(d, a, 1, a, 0, out, n, st, err);
    default: return GFA_ERR_UNSUPPORTED;
    }
}

// x ** e with an exponent per element, 32768 < q <= 65536, uint16 arrays (r06): covers the first n & ~7 elements (as big16_run)
int big16_power_each(const FieldDev &lut, const void *image, const void *a, const i64 *e, void *out, i64 n, hipStream_t st, int32_t *err)
{
    if (!image || lut.q <= 32768 || lut.q > 65536 || n < BIG16_MIN_N || !al16(out) || !al16(a) || !al16(e)) return GFA_ERR_UNSUPPORTED;
    const MidDesc d = make_desc(lut, (const u16 *)image);
    constexpr int T = 1024;
    auto ka = big16_powv_index_kernel<T>;
    auto kb = big16_exp_kernel<T>;
    static bool sattr = false;
    if (!sattr) {
        GFA_HIP(hipFuncSetAttribute((const void *)ka, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        GFA_HIP(hipFuncSetAttribute((const void *)kb, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        sattr = true;
    }
    const size_t lds = (size_t)d.qa * sizeof(u16);
    const i64 nvec = n >> 3;
    const int cus = mid_num_cus();
    const i64 slice = std::min<i64>(nvec, (i64)1 << 23); // index slices of at most 128 MiB: they stay in the Infinity Cache between the two kernels
    u16 *idx = nullptr;
    if (gfa::scratch_alloc((void **)&idx, (size_t)slice * 16, st) != hipSuccess) { (void)hipGetLastError(); return GFA_ERR_UNSUPPORTED; }
    for (i64 v0 = 0; v0 < nvec; v0 += slice) {
        const i64 cnt = std::min<i64>(slice, nvec - v0);
        hipLaunchKernelGGL(ka, dim3(cus), dim3(T), lds, st, d, (const u16 *)a + v0 * 8, e + v0 * 8, idx, cnt, err);
        hipLaunchKernelGGL(kb, dim3(cus), dim3(T), lds, st, d, (const u16 *)idx, (u16 *)out + v0 * 8, cnt);
    }
    const hipError_t le = hipGetLastError();
    GFA_HIP(gfa::scratch_free(idx, st));
    GFA_HIP(le);
    return GFA_OK;
}

// ---- uint32 / int64 STORAGE of fields with 32768 < q <= 65536 (r05; the reference's dtype list: _fields/_ufunc.py:97-111) ----
// The staged-table kernels above work on 16-bit vectors.  A wider array is narrowed into a 16-bit work buffer (one streaming pass per
// operand), run through them, and the result widened again: 24 instead of 12 B/element cross the memory system for uint32 (40 instead
// of 24 for int64), against table gathers from L2 on the generic kernels.
template <typename T>
__global__ __launch_bounds__(256) void narrow16_kernel(const T *__restrict__ in, u16 *__restrict__ out, i64 n)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) out[i] = (u16)in[i];
}
template <typename T>
__global__ __launch_bounds__(256) void widen16_kernel(const u16 *__restrict__ in, T *__restrict__ out, i64 n)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) out[i] = (T)in[i];
}

bool big16_wide_eligible(const FieldDev &calc, const void *image, int dtype, i64 n)
{
    return image != nullptr && (dtype == GFA_U32 || dtype == GFA_U64) && calc.q > 32768 && calc.q <= 65536 && n >= BIG16_MIN_N;
}

// covers the first n & ~7 elements (as big16_run); the caller runs the generic kernels on the rest
template <typename T>
int big16_run_wide_t(const FieldDev &lut, const void *image, int op, const T *a, i64 sa, const T *b, i64 sb, const i64 *e, T *out, i64 n, hipStream_t st,
                     int32_t *err)
{
    const i64 n8 = n & ~(i64)7;
    if (n8 == 0) return GFA_OK;
    const i64 na = sa ? n8 : 8, nb = b ? (sb ? n8 : 8) : 0;
    u16 *wa = nullptr, *wb = nullptr, *wo = nullptr;
    if (gfa::scratch_alloc((void **)&wa, sizeof(u16) * (size_t)(na + nb + n8) + 64, st) != hipSuccess) {
        (void)hipGetLastError();
        return GFA_ERR_UNSUPPORTED; // no work buffer: the generic kernels take the call
    }
    wb = wa + ((na + 7) & ~(i64)7);
    wo = wb + ((nb + 7) & ~(i64)7);
    const int grid = (int)std::min<i64>((n8 + 255) / 256, 256 * 16);
    if (sa) hipLaunchKernelGGL((narrow16_kernel<T>), dim3(grid), dim3(256), 0, st, a, wa, n8);
    else hipLaunchKernelGGL((narrow16_kernel<T>), dim3(1), dim3(64), 0, st, a, wa, (i64)1);
    if (b) {
        if (sb) hipLaunchKernelGGL((narrow16_kernel<T>), dim3(grid), dim3(256), 0, st, b, wb, n8);
        else hipLaunchKernelGGL((narrow16_kernel<T>), dim3(1), dim3(64), 0, st, b, wb, (i64)1);
    }
    int rc = big16_run(lut, image, op, wa, sa, b ? wb : nullptr, sb, e, wo, n8, st, err);
    if (rc == GFA_OK) {
        hipLaunchKernelGGL((widen16_kernel<T>), dim3(grid), dim3(256), 0, st, (const u16 *)wo, out, n8);
        if (hipGetLastError() != hipSuccess) rc = GFA_ERR_HIP;
    }
    (void)gfa::scratch_free(wa, st);
    return rc;
}

int big16_run_wide(const FieldDev &lut, const void *image, int dtype, int op, const void *a, i64 sa, const void *b, i64 sb, const i64 *e, void *out, i64 n,
                   hipStream_t st, int32_t *err)
{
    if (dtype == GFA_U32) return big16_run_wide_t<u32>(lut, image, op, (const u32 *)a, sa, (const u32 *)b, sb, e, (u32 *)out, n, st, err);
    return big16_run_wide_t<u64>(lut, image, op, (const u64 *)a, sa, (const u64 *)b, sb, e, (u64 *)out, n, st, err);
}

} // namespace gfa
