// gfa_ntt.hip -- number-theoretic / finite-field Fourier transforms for gfx950.
//
// Replaces fft_jit / ifft_jit (reference: src/galois/_domains/_function.py:177-392): X[k] = sum_j x[j] w^(jk),
// natural order in and out, any n | q-1 over any field.  Field arithmetic is exact, so any correct DFT algorithm
// reproduces the reference's bits; the device does NOT follow the reference's stage order:
//
//   * power-of-two n over GF(p) (p < 2^32 as uint32, Goldilocks / 64-bit primes as uint64):
//       n <= TILE_MAX : one kernel; each 256-thread workgroup keeps a tile of whole transforms in LDS, radix-2
//                       decimation-in-time with the bit reversal folded into the LDS staging, twiddles (plus their
//                       Shoup quotients for p < 2^31) in LDS.
//       n  > TILE_MAX : four-step (Bailey) = two such passes.  n = n1*n2; pass 1 transforms the n2 columns (length
//                       n1, stride n2) in tiles of adjacent columns so that every global access is a contiguous
//                       segment, and multiplies by w^(j2*k1) from a two-level power table; pass 2 transforms the n1
//                       rows (contiguous) in tiles of adjacent rows and stores transposed, giving natural order.
//                       => 2 reads + 2 writes of the array in total (the intermediate is normally L2/MALL resident).
//   * anything else (mixed radix, extension fields): one global-memory Stockham stage per prime factor, same index
//     maps as the reference's stages (_function.py:315-384), twiddles from a w^t table.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <tuple>
#include <type_traits>

#include "gfa_internal.h"
#include "gfa_goldilocks.h"

using namespace gfa;

namespace {

// ------------------------------------------------------------------------------------------------
// twiddle multiplication
// ------------------------------------------------------------------------------------------------
template <class F>
struct Tw { // generic: plain field multiply
    typedef typename F::elem E;
    static constexpr bool HAS_SHOUP = false;
    struct W { E w; };
    static __device__ __forceinline__ W load(const E *tab, const E *, u32 i) { return W{tab[i]}; }
    static __device__ __forceinline__ E mul(const FieldDev &fd, E x, W t) { return F::mul(fd, x, t.w); }
    static __device__ __forceinline__ E add(const FieldDev &fd, E a, E b) { return F::add(fd, a, b); }
    static __device__ __forceinline__ E sub(const FieldDev &fd, E a, E b) { return F::sub(fd, a, b); }
    // (a - b) * w
    static __device__ __forceinline__ E submul(const FieldDev &fd, E a, E b, W t) { return F::mul(fd, F::sub(fd, a, b), t.w); }
    typedef const FieldDev &Ctx; // the register kernel's per-thread arithmetic context
    static __device__ __forceinline__ Ctx make_ctx(const FieldDev &fd) { return fd; }
    static __device__ __forceinline__ W make_w(Ctx, E w, E) { return W{w}; }
};

// GF(p), p < 2^31: Shoup multiplication by a constant w with wq = floor(w * 2^32 / p):
//   q = hi32(wq * x);  r = w*x - q*p  in [0, 2p)  (all in 32-bit wrap-around arithmetic)
struct TwShoup32 {
    typedef u32 E;
    static constexpr bool HAS_SHOUP = true;
    static constexpr int QBITS = 32;
    struct W { u32 w, wq; };
    static __device__ __forceinline__ W load(const u32 *tab, const u32 *tabq, u32 i) { return W{tab[i], tabq[i]}; }
    static __device__ __forceinline__ u32 mul(const FieldDev &fd, u32 x, W t)
    {
        u32 q = __umulhi(t.wq, x);
        u32 r = t.w * x - q * (u32)fd.p;
        u32 p = (u32)fd.p;
        return r >= p ? r - p : r;
    }
    // p < 2^31: all sums stay below 2^32; unsigned min() selects the reduced value
    static __device__ __forceinline__ u32 add(const FieldDev &fd, u32 a, u32 b)
    {
        const u32 c = a + b;
        return min(c, c - (u32)fd.p);
    }
    static __device__ __forceinline__ u32 sub(const FieldDev &fd, u32 a, u32 b)
    {
        const u32 d = a - b;
        return min(d, d + (u32)fd.p);
    }
    // (a - b) * w without reducing the difference first: a + p - b lies in (0, 2p), a valid multiplier input
    static __device__ __forceinline__ u32 submul(const FieldDev &fd, u32 a, u32 b, W t) { return mul(fd, a + (u32)fd.p - b, t); }
    // Per-thread context for the register kernel: the modulus in a VGPR whose value the optimiser cannot trace back to
    // the (uniform) kernel argument, so that every use is a plain VALU operand.
    struct Ctx { u32 p; };
    static __device__ __forceinline__ Ctx make_ctx(const FieldDev &fd)
    {
        u32 pv = (u32)fd.p;
        asm volatile("" : "+v"(pv));
        return Ctx{pv};
    }
    static __device__ __forceinline__ W make_w(Ctx, u32 w, u32 wq) { return W{w, wq}; }
    static __device__ __forceinline__ u32 add(Ctx c, u32 a, u32 b) { const u32 s = a + b; return min(s, s - c.p); }
    static __device__ __forceinline__ u32 sub(Ctx c, u32 a, u32 b) { const u32 d = a - b; return min(d, d + c.p); }
    static __device__ __forceinline__ u32 submul(Ctx c, u32 a, u32 b, W t) { return mul(c, a + c.p - b, t); }
    static __device__ __forceinline__ u32 mul(Ctx c, u32 x, W t)
    {
        const u32 q = __umulhi(t.wq, x);
        const u32 r = t.w * x - q * c.p;
        return min(r, r - c.p);
    }
    static constexpr bool REDC = true;
    static __device__ __forceinline__ u32 canon(Ctx c, u32 a) { return min(a, a - c.p); } // [0, 2p) -> [0, p)
    // Montgomery product x * y * 2^-32 mod p for x*y < p * 2^32 (x < 2p, y < p, p < 2^31); result in (0, 2p)
    static __device__ __forceinline__ u32 redc(Ctx c, u32 x, u32 y, u32 pinv)
    {
        const u32 hi = __umulhi(x, y), lo = x * y;
        const u32 m = lo * pinv;
        return hi - __umulhi(m, c.p) + c.p;
    }
};

// GF(p), p < 2^30, register kernel only: Harvey-style lazy butterflies.  Values live in [0, 2p); a Shoup product
// w*x - floor(wq*x / 2^32)*p lands in [0, 2p) for ANY 32-bit x, so neither the difference feeding a multiplication
// nor its result needs a conditional correction.  (Measured on gfx950: v_mul_lo_u32 / v_mul_hi_u32 issue at the same
// rate as the 24-bit multiplies, so the 32-bit form -- 3 multiplies -- beats the 24-bit one -- 4 multiplies + alignbit.)
struct TwShoupLazy {
    typedef u32 E;
    static constexpr bool HAS_SHOUP = true;
    static constexpr int QBITS = 32;
    static constexpr bool LAZY = true;
    static constexpr bool REDC = true;
    struct W { u32 w, wq; };
    struct Ctx { u32 p, p2; };
    static __device__ __forceinline__ W load(const u32 *tab, const u32 *tabq, u32 i) { return W{tab[i], tabq[i]}; }
    static __device__ __forceinline__ Ctx make_ctx(const FieldDev &fd)
    {
        u32 pv = (u32)fd.p;
        asm volatile("" : "+v"(pv));
        return Ctx{pv, 2 * pv};
    }
    static __device__ __forceinline__ W make_w(Ctx, u32 w, u32 wq) { return W{w, wq}; }
    static __device__ __forceinline__ u32 mul(Ctx c, u32 x, W t)
    { // x: any u32; result in [0, 2p)
        const u32 q = __umulhi(t.wq, x);
        return t.w * x - q * c.p;
    }
    static __device__ __forceinline__ u32 add(Ctx c, u32 a, u32 b) { const u32 s = a + b; return min(s, s - c.p2); }
    static __device__ __forceinline__ u32 sub(Ctx c, u32 a, u32 b) { const u32 d = a + c.p2 - b; return min(d, d - c.p2); }
    static __device__ __forceinline__ u32 submul(Ctx c, u32 a, u32 b, W t) { return mul(c, a + c.p2 - b, t); }
    static __device__ __forceinline__ u32 canon(Ctx c, u32 a) { return min(a, a - c.p); } // [0, 2p) -> [0, p)
    // Montgomery product x * y * 2^-32 mod p for x*y < p * 2^32 (pinv = p^-1 mod 2^32); result in (0, 2p)
    static __device__ __forceinline__ u32 redc(Ctx c, u32 x, u32 y, u32 pinv)
    {
        const u32 hi = __umulhi(x, y), lo = x * y;
        const u32 m = lo * pinv;
        return hi - __umulhi(m, c.p) + c.p;
    }
};

// p < 2^24: the register DFTs run WITHOUT reducing sums and differences.  Inside one radix-32 network values grow to at
// most 2^5 * 2p < 2^30; the difference path adds a stage-dependent multiple of p (>= any operand of that stage) instead of
// correcting, and every Shoup product brings its result back to [0, 2p) whatever the size of its input.  A butterfly is then
// add | sub, add | mulhi, mul, mul, sub = 7 instructions instead of 9, a twiddle-free one 3 instead of 6.
// The values that meet no multiplication on their way out of a network (line 0 of the exchange, the untwiddled last
// pass) are normalised explicitly with a Shoup product by one.
struct TwShoupWide : TwShoupLazy {
    static constexpr bool WIDE = true;
    struct Ctx { u32 p, p2, one_q; };
    static __device__ __forceinline__ Ctx make_ctx(const FieldDev &fd)
    {
        u32 pv = (u32)fd.p;
        asm volatile("" : "+v"(pv));
        return Ctx{pv, 2 * pv, (u32)(0xffffffffull / fd.p)}; // floor((2^32 - 1) / p) = floor(2^32 / p) for odd p
    }
    static __device__ __forceinline__ W make_w(Ctx, u32 w, u32 wq) { return W{w, wq}; }
    static __device__ __forceinline__ u32 mul(Ctx c, u32 x, W t)
    {
        const u32 q = __umulhi(t.wq, x);
        return t.w * x - q * c.p;
    }
    static __device__ __forceinline__ u32 add(Ctx, u32 a, u32 b) { return a + b; }
    // stage t of a network whose inputs are below 2p: operands are below 2^t * 2p = p << (t + 1), so adding that constant
    // keeps the difference non-negative and below twice the bound -- sums and twiddle-free differences then grow by the
    // same factor 2 per stage, to 64p after five stages (v_lshl_add_u32 + v_sub_u32)
    static __device__ __forceinline__ u32 sub(Ctx c, u32 a, u32 b, int t) { return a + (c.p << (t + 1)) - b; }
    static __device__ __forceinline__ u32 submul(Ctx c, u32 a, u32 b, W w, int t) { return mul(c, a + (c.p << (t + 1)) - b, w); }
    static __device__ __forceinline__ u32 norm(Ctx c, u32 a) { return a - __umulhi(c.one_q, a) * c.p; } // any u32 -> [0, 2p)
    static __device__ __forceinline__ u32 canon(Ctx c, u32 a) { return min(a, a - c.p); }
    static __device__ __forceinline__ u32 redc(Ctx c, u32 x, u32 y, u32 pinv)
    { // x < 64p, y < 2p, p < 2^24  =>  x*y < p * 2^32
        const u32 hi = __umulhi(x, y), lo = x * y;
        const u32 m = lo * pinv;
        return hi - __umulhi(m, c.p) + c.p;
    }
};

template <class TW, class = void>
struct WideTrait { static constexpr bool value = false; };
template <class TW>
struct WideTrait<TW, std::enable_if_t<TW::WIDE>> { static constexpr bool value = true; };
template <class TW>
constexpr bool is_wide() { return WideTrait<TW>::value; }

__device__ __forceinline__ u32 bitrev(u32 x, int bits) { return __brev(x) >> (32 - bits); }

// ------------------------------------------------------------------------------------------------
// table builders
// ------------------------------------------------------------------------------------------------
template <class F>
__global__ void pow_table_kernel(FieldDev fd, typename F::elem base, u64 exp_stride, typename F::elem *out, i64 count)
{
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = F::pow_u(fd, base, (u64)i * exp_stride);
}

__global__ void shoup_table_kernel(u32 p, const u32 *w, u32 *wq, i64 count, int qbits)
{
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) wq[i] = (u32)((((u64)w[i]) << qbits) / p);
}

// Goldilocks inter-pass twiddles as a table: T[line + lines * k] = w^(line * k mod count), count = lines * L entries -- the
// layout of the column pass's output inside one transform, so a thread reads its twiddle at the offset it stores to (the same
// 64-byte segments), and every transform of the batch shares the table through L2 / the Infinity Cache.  This replaces the
// per-thread progression t <- t * ratio: one product per point instead of two in that pass.
__global__ void gl_post_table_kernel(const u64 *__restrict__ powA, const u64 *__restrict__ powB, int lo_bits, u64 *__restrict__ out,
                                     u32 lines, u32 count)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u32 line = i % lines, k = i / lines;
    const u32 e = (line * k) & (count - 1); // count is a power of two <= 2^20: the product fits 32 bits
    out[i] = gl::canon_u64(gl::mul_red(powA[e >> lo_bits], powB[e & ((1u << lo_bits) - 1)]));
}

// ------------------------------------------------------------------------------------------------
// LDS tile transform
// ------------------------------------------------------------------------------------------------
// One workgroup transforms `lines` lines of length L = 2^logL held in LDS as lds[line*L + pos].
// Global addressing of element (line c, position t): base + c*stride_c + t*stride_t   (in and out separately).
// LOAD_ALONG_LINE / STORE_ALONG_LINE select which index runs fastest across lanes (the contiguous one in memory).
struct TileArgs {
    i64 in_stride_c, in_stride_t;   // elements
    i64 out_stride_c, out_stride_t; // elements
    i64 in_batch_stride, out_batch_stride;
    int logL;
    int lines_per_tile;  // C
    int tiles_per_batch; // total lines per batch item / C
    // optional post-multiplication by w_N^((line0 + c) * k): two-level power table A[e >> lo_bits] * B[e & lo_mask]
    int post_twiddle;
    int lo_bits;
    u64 n_mask; // N - 1
    i64 line_offset; // added to the line index in the post-twiddle exponent (distributed column pass)
    // optional scalar scale
    int do_scale;
    u64 scale;
    int tw_in_lds; // stage the L/2 twiddles (and Shoup quotients) in LDS; long lines read them through L1/L2
};

template <class F, class TW, bool LOAD_ALONG_LINE, bool STORE_ALONG_LINE>
__global__ __launch_bounds__(256) void ntt_tile_kernel(FieldDev fd, const typename F::elem *__restrict__ in,
                                                       typename F::elem *__restrict__ out, TileArgs ta,
                                                       const typename F::elem *__restrict__ wtab,
                                                       const typename F::elem *__restrict__ wtabq,
                                                       const typename F::elem *__restrict__ powA,
                                                       const typename F::elem *__restrict__ powB)
{
    typedef typename F::elem E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int logL = ta.logL;
    const u32 L = 1u << logL;
    const u32 C = (u32)ta.lines_per_tile; // power of two
    const int logC = __ffs((int)C) - 1;
    const u32 LP = L + 1;                         // line pitch: +1 element so that a column of the tile spreads over banks
    E *data = reinterpret_cast<E *>(smem_raw);    // C * LP
    E *twl = data + (size_t)C * LP;               // L/2 twiddles
    E *twql = twl + (L >> 1);                     // L/2 Shoup quotients (only if TW::HAS_SHOUP)
    const E *tw = ta.tw_in_lds ? twl : wtab;
    const E *twq = ta.tw_in_lds ? twql : wtabq;

    const u32 tid = threadIdx.x;
    const i64 batch = blockIdx.x / ta.tiles_per_batch;
    const u32 tile = blockIdx.x % ta.tiles_per_batch;
    const i64 line0 = (i64)tile * C;
    const E *gin = in + batch * ta.in_batch_stride;
    E *gout = out + batch * ta.out_batch_stride;

    if (ta.tw_in_lds)
        for (u32 i = tid; i < (L >> 1); i += 256) {
            twl[i] = wtab[i];
            if constexpr (TW::HAS_SHOUP) twql[i] = wtabq[i];
        }
    // ---- load (bit-reversed position inside each line) ----
    const u32 total = C * L;
    if constexpr (LOAD_ALONG_LINE) {
        for (u32 e = tid; e < total; e += 256) {
            u32 c = e >> logL, t = e & (L - 1);
            data[c * LP + bitrev(t, logL)] = gin[(line0 + c) * ta.in_stride_c + (i64)t * ta.in_stride_t];
        }
    } else {
        for (u32 e = tid; e < total; e += 256) {
            u32 t = e >> logC, c = e & (C - 1);
            data[c * LP + bitrev(t, logL)] = gin[(line0 + c) * ta.in_stride_c + (i64)t * ta.in_stride_t];
        }
    }
    __syncthreads();
    // ---- log2(L) radix-2 decimation-in-time stages, in place ----
    const u32 nbf = total >> 1;
    for (int s = 0; s < logL; s++) {
        const u32 half = 1u << s;
        const int tshift = logL - 1 - s; // twiddle index = j << tshift
        for (u32 u = tid; u < nbf; u += 256) {
            u32 line = u >> (logL - 1);
            u32 w = u & ((L >> 1) - 1);
            u32 j = w & (half - 1);
            u32 i0 = line * LP + ((w >> s) << (s + 1)) + j;
            u32 i1 = i0 + half;
            E a = data[i0];
            E b = data[i1];
            if (s > 0) b = TW::mul(fd, b, TW::load(tw, twq, j << tshift));
            data[i0] = F::add(fd, a, b);
            data[i1] = F::sub(fd, a, b);
        }
        __syncthreads();
    }
    // ---- store ----
    const E scale = (E)ta.scale;
    auto finish = [&](u32 c, u32 k) -> E {
        E v = data[c * LP + k];
        if (ta.post_twiddle) {
            u64 e = ((u64)(ta.line_offset + line0 + c) * (u64)k) & ta.n_mask;
            E t = F::mul(fd, powA[e >> ta.lo_bits], powB[e & ((1ull << ta.lo_bits) - 1)]);
            v = F::mul(fd, v, t);
        }
        if (ta.do_scale) v = F::mul(fd, v, scale);
        return v;
    };
    if constexpr (STORE_ALONG_LINE) {
        for (u32 e = tid; e < total; e += 256) {
            u32 c = e >> logL, k = e & (L - 1);
            gout[(line0 + c) * ta.out_stride_c + (i64)k * ta.out_stride_t] = finish(c, k);
        }
    } else {
        for (u32 e = tid; e < total; e += 256) {
            u32 k = e >> logC, c = e & (C - 1);
            gout[(line0 + c) * ta.out_stride_c + (i64)k * ta.out_stride_t] = finish(c, k);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// register-blocked line transform: L = R1 * R2 points per line, R1, R2 <= 32
// ------------------------------------------------------------------------------------------------
// Each thread owns R1 (then R2) points in VGPRs and runs a fully unrolled decimation-in-frequency network on them,
// so a length-1024 line costs ONE exchange through LDS instead of ten LDS round trips:
//   step A: thread (line c, r) loads x[r + R2*a], a < R1, transforms over a, multiplies by w_L^(r*ka), writes LDS
//   step B: thread (line c, ka) reads the R2 values of its ka, transforms over r, stores X[ka + R1*kr]
// The constants of the in-register networks (w_R^j) are wave-uniform, so they are scalar loads.
struct RegArgs {
    i64 in_stride_c, in_stride_t;   // elements: element (line c, position t) at c*stride_c + t*stride_t
    i64 out_stride_c, out_stride_t;
    i64 in_batch_stride, out_batch_stride;
    i64 total_lines; // lines per batch item
    int tiles_per_batch;
    int load_along_line, store_along_line; // which index is contiguous in memory (runs fastest across lanes)
    int post_twiddle; // multiply output (line, k) by w_N^((line_offset + line) * k) = A[e >> lo_bits] * B[e & mask]; 2 (Goldilocks
                      // kernel only): the products come from a table passed in place of A, laid out like the output of one transform
    int pre_twiddle;  // multiply INPUT (line, t) by w_N^((line_offset + line) * t) before the transform (inverse four-step)
    int lo_bits;
    u64 n_mask;
    i64 line_offset;
    int do_scale;
    u64 scale;
    u64 scale_q; // Shoup quotient of `scale` for the twiddle class in use
    int xcd_remap;
    u64 pinv;    // p^-1 mod 2^32 (lazy 32-bit path)
    // Two-level position strides (the distributed transform's exchange buffers, which are laid out in per-peer chunks):
    // position t of a line lives at (t & (2^split - 1)) * stride_t + (t >> split) * chunk_stride.  split = 31: one level.
    // The kernel needs 2^in_split >= R2 and 2^out_split >= R1 (the per-lane part of a position never crosses a chunk).
    int in_split = 31, out_split = 31;
    i64 in_chunk_stride = 0, out_chunk_stride = 0; // elements
    int perm1 = 1, perm2 = 1; // Goldilocks shift-twiddle networks: input permutations (see ntt_reg_kernel_gl)
    int waves4 = 0;           // Goldilocks: hold the kernel to 128 VGPRs (four waves per SIMD); set by the three-pass driver
    int lazy_out = 0;         // Goldilocks, twiddled output that only feeds the next pass: any 64-bit representative, not the canonical one
};

// byte offset of position t0 (a multiple of the lane-owned low part; wave-uniform => scalar arithmetic)
__device__ __forceinline__ u32 pos_offset(u32 t0, int split, u32 stride_bytes, u32 chunk_bytes)
{
    return (t0 & ((1u << split) - 1u)) * stride_bytes + (t0 >> split) * chunk_bytes;
}

template <class F, class TW, int LOGR>
__device__ __forceinline__ void reg_dif(typename TW::Ctx fd, typename F::elem (&v)[1 << LOGR],
                                        const typename F::elem *__restrict__ w, const typename F::elem *__restrict__ wq,
                                        int wstride)
{ // v[bitrev(k)] <- sum_a v[a] * w_R^(a*k), with w[j * wstride] = w_R^j
    typedef typename F::elem E;
    constexpr int R = 1 << LOGR;
#pragma unroll
    for (int s = LOGR - 1; s >= 0; s--) {
        const int half = 1 << s;
#pragma unroll
        for (int b = 0; b < R; b += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                const E u = v[b + j], x = v[b + j + half];
                v[b + j] = TW::add(fd, u, x);
                const int tj = j << (LOGR - 1 - s);
                if constexpr (is_wide<TW>()) {
                    if (tj != 0) v[b + j + half] = TW::submul(fd, u, x, TW::load(w, wq, (u32)(tj * wstride)), LOGR - 1 - s);
                    else v[b + j + half] = TW::sub(fd, u, x, LOGR - 1 - s);
                } else {
                    if (tj != 0) v[b + j + half] = TW::submul(fd, u, x, TW::load(w, wq, (u32)(tj * wstride)));
                    else v[b + j + half] = TW::sub(fd, u, x);
                }
            }
        }
    }
}

template <class TW, class = void>
struct LazyTrait { static constexpr bool value = false; };
template <class TW>
struct LazyTrait<TW, std::enable_if_t<TW::LAZY>> { static constexpr bool value = true; };
template <class TW>
constexpr bool is_lazy() { return LazyTrait<TW>::value; }

template <class TW, class = void>
struct RedcTrait { static constexpr bool value = false; };
template <class TW>
struct RedcTrait<TW, std::enable_if_t<TW::REDC>> { static constexpr bool value = true; };
template <class TW>
constexpr bool has_redc() { return RedcTrait<TW>::value; } // 32-bit Montgomery products for the post-twiddle progression

constexpr int brev_c(int x, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// SPLIT: the LDS exchange runs in two rounds of R1/2 rows each (half the buffer: three 512-thread workgroups per CU
// instead of two for 1024-point lines of 32-bit elements; the kernel is occupancy sensitive -- 8 instead of 16 waves per
// CU cost 40 %).
template <class F, class TW, int LOGR1, int LOGR2, int THREADS, bool SPLIT = false>
__global__ __launch_bounds__(THREADS) void ntt_reg_kernel(FieldDev fdk, const typename F::elem *__restrict__ in,
                                                          typename F::elem *__restrict__ out, RegArgs ra,
                                                          const typename F::elem *__restrict__ wL,
                                                          const typename F::elem *__restrict__ wLq,
                                                          const typename F::elem *__restrict__ powA,
                                                          const typename F::elem *__restrict__ powAq,
                                                          const typename F::elem *__restrict__ powB,
                                                          const typename F::elem *__restrict__ powBq,
                                                          const typename F::elem *__restrict__ powAm)
{
    typedef typename F::elem E;
    constexpr int R1 = 1 << LOGR1, R2 = 1 << LOGR2, L = R1 * R2;
    constexpr int C = THREADS / R1;           // lines per tile
    constexpr int LOGC = __builtin_ctz(C);
    constexpr int ROW = R2 + 1;               // padded row of R2 values
    constexpr int RROWS = SPLIT ? R1 / 2 : R1; // rows of a line resident in LDS at a time
    constexpr int PC = RROWS * ROW + 1;       // padded line pitch
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E *data = reinterpret_cast<E *>(smem_raw);      // C * PC
    E *twl = data + C * PC;                          // L middle twiddles w_L^e
    E *twql = twl + L;                               // their Shoup quotients

    const int tid = threadIdx.x;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (each XCD has its own L2).  Neighbouring tiles of a strided
    // pass touch the two halves of the same 128-byte lines, so consecutive tiles are kept on ONE XCD.
    u32 vb = blockIdx.x;
    if (ra.xcd_remap) vb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const i64 batch = vb / ra.tiles_per_batch;
    const i64 line0 = (i64)(vb % ra.tiles_per_batch) * C;
    // tile base pointers are wave-uniform (scalar 64-bit arithmetic); per-thread offsets inside a tile fit in 32 bits
    const E *gin = in + batch * ra.in_batch_stride + line0 * ra.in_stride_c;
    E *gout = out + batch * ra.out_batch_stride + line0 * ra.out_stride_c;
    // byte strides (the host guarantees that every offset inside a tile stays below 2^31 bytes)
    const u32 isc = (u32)ra.in_stride_c * (u32)sizeof(E), ist = (u32)ra.in_stride_t * (u32)sizeof(E);
    const u32 osc = (u32)ra.out_stride_c * (u32)sizeof(E), ost = (u32)ra.out_stride_t * (u32)sizeof(E);
    const char *ginb = reinterpret_cast<const char *>(gin);
    char *goutb = reinterpret_cast<char *>(gout);
    const typename TW::Ctx fd = TW::make_ctx(fdk);

    for (int i = tid; i < L; i += THREADS) {
        twl[i] = wL[i];
        if constexpr (TW::HAS_SHOUP) twql[i] = wLq[i];
    }
    // ---- step A ----
    const bool active_a = tid < C * R2;
    int ca, ra_;
    if (ra.load_along_line) { ca = tid >> LOGR2; ra_ = tid & (R2 - 1); }
    else { ra_ = tid >> LOGC; ca = tid & (C - 1); }
    int c, ka; // step B: line and output row of this thread
    if (ra.store_along_line) { c = tid >> LOGR1; ka = tid & (R1 - 1); }
    else { ka = tid >> LOGC; c = tid & (C - 1); }
    E v[R2];
    {
        E va[R1];
        if (active_a) {
            // lines beyond the end of the batch are clamped to the last line (they compute garbage that is never stored)
            const i64 last = ra.total_lines - 1 - line0;
            const u32 cl = (u32)((i64)ca <= last ? ca : last);
            const u32 off = cl * isc + (u32)ra_ * ist;
            const u32 icb = (u32)ra.in_chunk_stride * (u32)sizeof(E);
#pragma unroll
            for (int a = 0; a < R1; a++) va[a] = *reinterpret_cast<const E *>(ginb + (off + pos_offset((u32)(a * R2), ra.in_split, ist, icb)));
            if (ra.pre_twiddle) {
                // * w_N^(line * (r + R2*a)), a = 0..R1-1: the same per-thread geometric progression as the post-twiddle below
                const u32 line = (u32)(ra.line_offset + line0 + cl);
                const u32 nmask = (u32)ra.n_mask, lo_mask = (1u << ra.lo_bits) - 1;
                const u32 e0 = (line * (u32)ra_) & nmask, es = (line * (u32)R2) & nmask;
                if constexpr (has_redc<TW>()) {
                    u32 tm = TW::mul(fd, powAm[e0 >> ra.lo_bits], TW::load(powB, powBq, e0 & lo_mask));
                    const u32 sm = TW::canon(fd, TW::mul(fd, powAm[es >> ra.lo_bits], TW::load(powB, powBq, es & lo_mask)));
                    const u32 pinv = (u32)ra.pinv;
#pragma unroll
                    for (int a = 0; a < R1; a++) {
                        u32 x = TW::redc(fd, va[a], tm, pinv); // canonical input * t_a, in (0, 2p)
                        if constexpr (!is_lazy<TW>()) x = TW::canon(fd, x);
                        va[a] = x;
                        if (a + 1 < R1) tm = TW::redc(fd, tm, sm, pinv);
                    }
                } else {
                    E t = F::mul(fdk, powA[e0 >> ra.lo_bits], powB[e0 & lo_mask]);
                    const E sr = F::mul(fdk, powA[es >> ra.lo_bits], powB[es & lo_mask]);
#pragma unroll
                    for (int a = 0; a < R1; a++) {
                        va[a] = F::mul(fdk, va[a], t);
                        if (a + 1 < R1) t = F::mul(fdk, t, sr);
                    }
                }
            }
            reg_dif<F, TW, LOGR1>(fd, va, wL, wLq, R2); // w_R1 = w_L^R2
        }
        __syncthreads(); // middle-twiddle table staged
#pragma unroll
        for (int h = 0; h < (SPLIT ? 2 : 1); h++) {
            if (active_a) {
                E *dst = data + ca * PC + ra_;
                u32 idx = (u32)ra_ * (u32)(h * RROWS); // r * ka
#pragma unroll
                for (int kl = 0; kl < RROWS; kl++) {
                    const int kaa = h * RROWS + kl;
                    if (kaa == 0) {
                        if constexpr (is_wide<TW>()) dst[0] = TW::norm(fd, va[0]);
                        else dst[0] = va[0];
                    } else {
                        dst[kl * ROW] = TW::mul(fd, va[brev_c(kaa, LOGR1)], TW::load(twl, twql, idx));
                    }
                    idx += (u32)ra_;
                }
            }
            __syncthreads();
            // ---- step B reads its row once the round that carries it has landed ----
            if (!SPLIT || (ka / RROWS) == h) {
                const E *srcl = data + c * PC + (ka - h * RROWS) * ROW;
#pragma unroll
                for (int r = 0; r < R2; r++) v[r] = srcl[r];
            }
            if (SPLIT && h == 0) __syncthreads();
        }
    }
    // ---- step B ----
    {
        reg_dif<F, TW, LOGR2>(fd, v, wL, wLq, R1); // w_R2 = w_L^R1
        if (ra.post_twiddle) {
            const u32 line = (u32)(ra.line_offset + line0 + c);
            const u32 nmask = (u32)ra.n_mask, lo_mask = (1u << ra.lo_bits) - 1;
            // * w_N^(line * (ka + R1*kr)) for kr = 0..R2-1 is a geometric progression per thread: t_0 = w^(line*ka),
            // ratio w^(line*R1).  Both are fetched ONCE per thread from the two-level power table and every later factor
            // comes from a product in registers.  (The first version gathered two table entries per element from global
            // memory; those fully divergent gathers cost more than the whole transform -- pass 1 ran 2x slower than pass 2.)
            const u32 e0 = (line * (u32)ka) & nmask, es = (line * (u32)R1) & nmask;
            if constexpr (has_redc<TW>()) {
                // 32-bit classes: A kept in Montgomery form (A * 2^32 mod p), Montgomery products from then on
                u32 tm = TW::mul(fd, powAm[e0 >> ra.lo_bits], TW::load(powB, powBq, e0 & lo_mask));               // t_0 * R, [0,2p)
                const u32 sm = TW::canon(fd, TW::mul(fd, powAm[es >> ra.lo_bits], TW::load(powB, powBq, es & lo_mask))); // ratio * R, [0,p)
                const u32 pinv = (u32)ra.pinv;
#pragma unroll
                for (int kr = 0; kr < R2; kr++) {
                    u32 x = TW::redc(fd, v[brev_c(kr, LOGR2)], tm, pinv); // x * t_kr, in (0, 2p)
                    if constexpr (!is_lazy<TW>()) x = TW::canon(fd, x);
                    v[brev_c(kr, LOGR2)] = x;
                    if (kr + 1 < R2) tm = TW::redc(fd, tm, sm, pinv);      // t_{kr+1} * R
                }
            } else {
                E t = F::mul(fdk, powA[e0 >> ra.lo_bits], powB[e0 & lo_mask]);
                const E sr = F::mul(fdk, powA[es >> ra.lo_bits], powB[es & lo_mask]);
#pragma unroll
                for (int kr = 0; kr < R2; kr++) {
                    v[brev_c(kr, LOGR2)] = F::mul(fdk, v[brev_c(kr, LOGR2)], t);
                    if (kr + 1 < R2) t = F::mul(fdk, t, sr);
                }
            }
        }
        if (ra.do_scale) {
            const typename TW::W sw = TW::make_w(fd, (E)ra.scale, (E)ra.scale_q);
#pragma unroll
            for (int kr = 0; kr < R2; kr++) v[kr] = TW::mul(fd, v[kr], sw);
        }
        if constexpr (is_wide<TW>()) {
            if (!ra.post_twiddle && !ra.do_scale) {
#pragma unroll
                for (int kr = 0; kr < R2; kr++) v[kr] = TW::norm(fd, v[kr]);
            }
        }
        if constexpr (is_lazy<TW>()) {
#pragma unroll
            for (int kr = 0; kr < R2; kr++) v[kr] = TW::canon(fd, v[kr]);
        }
        if (line0 + c < ra.total_lines) {
            const u32 off = (u32)c * osc + (u32)ka * ost;
            const u32 ocb = (u32)ra.out_chunk_stride * (u32)sizeof(E);
#pragma unroll
            for (int kr = 0; kr < R2; kr++)
                *reinterpret_cast<E *>(goutb + (off + pos_offset((u32)(kr * R1), ra.out_split, ost, ocb))) = v[brev_c(kr, LOGR2)];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the same line transform for GF(2^64 - 2^32 + 1) on lazy 96-bit registers (gfa_goldilocks.h)
// ------------------------------------------------------------------------------------------------
// Identical data movement to ntt_reg_kernel (same RegArgs, tiles, LDS exchange); the arithmetic keeps every value as a
// three-limb two's-complement integer: butterflies are 3 + 3 carry-chained instructions with no reduction, a twiddle
// product is 4 v_mad_u64_u32 plus folds, and only what goes to LDS / memory is brought back to 64 bits.  The compiler's
// lowering of the plain 64-bit modular add / sub / mul (Tw<Goldilocks>) spends 287 instructions per point and pass on
// compare / select chains; this formulation needs about 130.
struct TwGoldi : Tw<Goldilocks> {};
#ifndef GFA_GL_WAVES
#define GFA_GL_WAVES 3 // waves per SIMD the register allocation of the Goldilocks kernel is held to (168 VGPRs)
#endif

template <int LOGR>
__device__ __forceinline__ void reg_dif_gl(gl::G3 (&v)[1 << LOGR], const u64 *__restrict__ w, int wstride)
{ // v[bitrev(k)] <- sum_a v[a] * w_R^(a*k), w[j * wstride] = w_R^j (wave-uniform: scalar loads)
    constexpr int R = 1 << LOGR;
#pragma unroll
    for (int s = LOGR - 1; s >= 0; s--) {
        const int half = 1 << s;
#pragma unroll
        for (int b = 0; b < R; b += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                const gl::G3 u = v[b + j], x = v[b + j + half];
                v[b + j] = gl::add(u, x);
                const int tj = j << (LOGR - 1 - s);
                if (tj != 0) v[b + j + half] = gl::mul(gl::sub(u, x), w[tj * wstride]);
                else v[b + j + half] = gl::sub(u, x);
            }
        }
    }
}

// SHIFT: networks on the canonical roots with shift twiddles; the input permutations that make them compute the wanted
// transform are ra.perm1 (register a' of the first network loads position (perm1 * a') mod R1) and ra.perm2 (the thread that
// owns column r of the exchange writes it at position (perm2 * r) mod R2).  Not combined with pre_twiddle (whose progression
// runs over the registers in position order).
// WAVES: waves per SIMD the register allocation is held to (3: 168 VGPRs, 4: 128).  Measured: the two strided passes of the
// three-pass transform (2^24, 2^26 points) run 5-6 % faster with four (more loads in flight); the column pass of a two-pass
// 2^20-point transform 9 % slower and contiguous 1024-point lines 18 % slower (the tighter allocation) -- so only RegArgs::waves4.
template <int LOGR1, int LOGR2, int THREADS, bool SPLIT, bool SHIFT = false, int WAVES = GFA_GL_WAVES>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void ntt_reg_kernel_gl(FieldDev fdk, const u64 *__restrict__ in, u64 *__restrict__ out, RegArgs ra,
                                                             const u64 *__restrict__ wL, const u64 *__restrict__ powA,
                                                             const u64 *__restrict__ powB)
{
    typedef u64 E;
    using gl::G3;
    constexpr int R1 = 1 << LOGR1, R2 = 1 << LOGR2, L = R1 * R2;
    constexpr int C = THREADS / R1;
    constexpr int LOGC = __builtin_ctz(C);
    constexpr int ROW = R2 + 1;
    constexpr int RROWS = SPLIT ? R1 / 2 : R1;
    constexpr int PC = RROWS * ROW + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E *data = reinterpret_cast<E *>(smem_raw); // C * PC
    E *twl = data + C * PC;                     // L middle twiddles w_L^e
    (void)fdk;
    const int tid = threadIdx.x;
    u32 vb = blockIdx.x;
    if (ra.xcd_remap) vb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const i64 batch = vb / ra.tiles_per_batch;
    const i64 line0 = (i64)(vb % ra.tiles_per_batch) * C;
    const E *gin = in + batch * ra.in_batch_stride + line0 * ra.in_stride_c;
    E *gout = out + batch * ra.out_batch_stride + line0 * ra.out_stride_c;
    const u32 isc = (u32)ra.in_stride_c * 8u, ist = (u32)ra.in_stride_t * 8u;
    const u32 osc = (u32)ra.out_stride_c * 8u, ost = (u32)ra.out_stride_t * 8u;
    const char *ginb = reinterpret_cast<const char *>(gin);
    char *goutb = reinterpret_cast<char *>(gout);
    for (int i = tid; i < L; i += THREADS) twl[i] = wL[i];
    const bool active_a = tid < C * R2;
    int ca, ra_;
    if (ra.load_along_line) { ca = tid >> LOGR2; ra_ = tid & (R2 - 1); }
    else { ra_ = tid >> LOGC; ca = tid & (C - 1); }
    int c, ka;
    if (ra.store_along_line) { c = tid >> LOGR1; ka = tid & (R1 - 1); }
    else { ka = tid >> LOGC; c = tid & (C - 1); }
    const u32 nmask = (u32)ra.n_mask, lo_mask = (1u << ra.lo_bits) - 1;
    G3 v[R2];
    {
        G3 va[R1];
        if (active_a) {
            const i64 last = ra.total_lines - 1 - line0;
            const u32 cl = (u32)((i64)ca <= last ? ca : last);
            const u32 off = cl * isc + (u32)ra_ * ist;
            const u32 icb = (u32)ra.in_chunk_stride * 8u;
#pragma unroll
            for (int a = 0; a < R1; a++) {
                const u32 pa = SHIFT ? (((u32)ra.perm1 * (u32)a) & (u32)(R1 - 1)) : (u32)a; // uniform: scalar arithmetic
                va[a] = gl::from_u64(*reinterpret_cast<const E *>(ginb + (off + pos_offset(pa * (u32)R2, ra.in_split, ist, icb))));
            }
            if (!SHIFT && ra.pre_twiddle) {
                // * w_N^(line * (r + R2*a)): per-thread geometric progression, one table fetch for its start and its ratio
                const u32 line = (u32)(ra.line_offset + line0 + cl);
                const u32 e0 = (line * (u32)ra_) & nmask, es = (line * (u32)R2) & nmask;
                u64 t = gl::mul_red(powA[e0 >> ra.lo_bits], powB[e0 & lo_mask]);
                const u64 sr = gl::mul_red(powA[es >> ra.lo_bits], powB[es & lo_mask]);
#pragma unroll
                for (int a = 0; a < R1; a++) {
                    va[a] = gl::from_u64(gl::mul_red(gl::to_u64(va[a]), t)); // back to [0, 2^64): the network's input range
                    if (a + 1 < R1) t = gl::mul_red(t, sr);
                }
            }
            if constexpr (SHIFT) gl::dif_shift<LOGR1>(va);
            else reg_dif_gl<LOGR1>(va, wL, R2); // w_R1 = w_L^R2
        }
        __syncthreads(); // middle-twiddle table staged
#pragma unroll
        for (int h = 0; h < (SPLIT ? 2 : 1); h++) {
            if (active_a) {
                E *dst = data + ca * PC + (SHIFT ? (int)(((u32)ra.perm2 * (u32)ra_) & (u32)(R2 - 1)) : ra_);
                u32 idx = (u32)ra_ * (u32)(h * RROWS); // r * ka
#pragma unroll
                for (int kl = 0; kl < RROWS; kl++) {
                    const int kaa = h * RROWS + kl;
                    if (kaa == 0) dst[0] = gl::to_u64(va[0]);
                    else dst[kl * ROW] = gl::mul_red(gl::to_u64(va[brev_c(kaa, LOGR1)]), twl[idx]);
                    idx += (u32)ra_;
                }
            }
            __syncthreads();
            if (!SPLIT || (ka / RROWS) == h) {
                const E *srcl = data + c * PC + (ka - h * RROWS) * ROW;
#pragma unroll
                for (int r = 0; r < R2; r++) v[r] = gl::from_u64(srcl[r]);
            }
            if (SPLIT && h == 0) __syncthreads();
        }
    }
    if constexpr (SHIFT) gl::dif_shift<LOGR2>(v);
    else reg_dif_gl<LOGR2>(v, wL, R1); // w_R2 = w_L^R1
    // results are reduced to the canonical [0, p) and stored one by one, so that the registers of v[] free up as they go
    const bool live = line0 + c < ra.total_lines;
    char *const obase = goutb + ((u32)c * osc + (u32)ka * ost);
    const u32 ocb = (u32)ra.out_chunk_stride * 8u;
    if (ra.post_twiddle == 2) {
        // twiddles from the table (powA), laid out like this pass's output within one transform
        const char *const tbase = reinterpret_cast<const char *>(powA) + ((u32)(line0 + (live ? c : 0)) * osc + (u32)ka * ost);
        constexpr int G = R2 < 8 ? R2 : 8; // twiddles in flight (16 registers beside the values still to be stored)
#pragma unroll
        for (int k0 = 0; k0 < R2; k0 += G) {
            u64 tw[G];
#pragma unroll
            for (int g = 0; g < G; g++) tw[g] = *reinterpret_cast<const E *>(tbase + (u32)((k0 + g) * R1) * ost);
#pragma unroll
            for (int g = 0; g < G; g++) {
                u64 y = gl::mul_red(gl::to_u64(v[brev_c(k0 + g, LOGR2)]), tw[g]);
                if (!ra.lazy_out) y = gl::canon_u64(y);
                if (live) *reinterpret_cast<E *>(obase + (u32)((k0 + g) * R1) * ost) = y;
            }
        }
    } else if (ra.post_twiddle) {
        const u32 line = (u32)(ra.line_offset + line0 + c);
        const u32 e0 = (line * (u32)ka) & nmask, es = (line * (u32)R1) & nmask;
        u64 t = gl::mul_red(powA[e0 >> ra.lo_bits], powB[e0 & lo_mask]);
        const u64 sr = gl::mul_red(powA[es >> ra.lo_bits], powB[es & lo_mask]);
#pragma unroll
        for (int kr = 0; kr < R2; kr++) {
            u64 y = gl::mul_red(gl::to_u64(v[brev_c(kr, LOGR2)]), t);
            if (!ra.lazy_out) y = gl::canon_u64(y);
            if (live) *reinterpret_cast<E *>(obase + pos_offset((u32)(kr * R1), ra.out_split, ost, ocb)) = y;
            if (kr + 1 < R2) t = gl::mul_red(t, sr);
        }
    } else if (ra.do_scale) {
#pragma unroll
        for (int kr = 0; kr < R2; kr++) {
            const u64 y = gl::canon_u64(gl::mul_red(gl::to_u64(v[brev_c(kr, LOGR2)]), ra.scale));
            if (live) *reinterpret_cast<E *>(obase + pos_offset((u32)(kr * R1), ra.out_split, ost, ocb)) = y;
        }
    } else {
#pragma unroll
        for (int kr = 0; kr < R2; kr++) {
            const u64 y = gl::canon(v[brev_c(kr, LOGR2)]);
            if (live) *reinterpret_cast<E *>(obase + pos_offset((u32)(kr * R1), ra.out_split, ost, ocb)) = y;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// generic Stockham stage (any radix, any field)
// ------------------------------------------------------------------------------------------------
// in viewed as (r, q, m), out as (q, r, m); out[qi, f, b] = sum_k in[k, qi, b] * tw^k, tw = w^(q*(f*m + b))
template <class F>
__global__ __launch_bounds__(256) void ntt_stage_kernel(FieldDev fd, const typename F::elem *__restrict__ in,
                                                        typename F::elem *__restrict__ out, i64 N, i64 r, i64 m, i64 q,
                                                        const typename F::elem *__restrict__ wpow, i64 batch,
                                                        int do_scale, typename F::elem scale)
{
    typedef typename F::elem E;
    const i64 total = N * batch;
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (i64)gridDim.x * blockDim.x) {
        const i64 bi = g / N, o = g % N;
        const E *x = in + bi * N;
        const i64 b = o % m;
        const i64 f = (o / m) % r;
        const i64 qi = o / (m * r);
        const E tw = wpow[(q * (f * m + b)) % N];
        E acc = x[((r - 1) * q + qi) * m + b];
        for (i64 k = r - 2; k >= 0; k--) acc = F::add(fd, F::mul(fd, acc, tw), x[(k * q + qi) * m + b]);
        if (do_scale) acc = F::mul(fd, acc, scale);
        out[bi * N + o] = acc;
    }
}

// Any-radix transform of up to 4096 points entirely in LDS: one workgroup per transform, the same Stockham stages as
// ntt_stage_kernel (index maps of _function.py:315-384) ping-ponging between two LDS buffers -- ONE launch instead of one per
// prime factor (the reference's own FFT benchmark sizes are 256*K, K = 1..9: 768 = 2^8*3, 1280 = 2^8*5, 2304 = 2^8*9, ...).
struct SmallArgs {
    int nf;
    int r[24]; // radix of stage s (taken from the end of the ascending factor list, as the reference does)
};

template <class F>
__global__ __launch_bounds__(256) void ntt_small_kernel(FieldDev fd, const typename F::elem *__restrict__ in, typename F::elem *__restrict__ out, int n,
                                                        SmallArgs sa, const typename F::elem *__restrict__ wpow, int do_scale, typename F::elem scale)
{
    typedef typename F::elem E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E *buf0 = reinterpret_cast<E *>(smem_raw), *buf1 = buf0 + n;
    const E *x = in + (i64)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += 256) buf0[i] = x[i];
    __syncthreads();
    E *src = buf0, *dst = buf1;
    int m = 1;
    for (int s = 0; s < sa.nf; s++) {
        const int r = sa.r[s], q = n / (m * r);
        const bool last = s == sa.nf - 1;
        auto emit = [&](int o, E v) {
            if (last) {
                if (do_scale) v = F::mul(fd, v, scale);
                out[(i64)blockIdx.x * n + o] = v;
            } else {
                dst[o] = v;
            }
        };
        if (r <= 4) {
            // radix 2 / 3 / 4 as butterflies: one work item per group (b, qi) -- inputs x_k = src[(k q + qi) m + b] times the
            // twiddles w^(q b k) (none in the first stage, where b = 0), then the r-point transform with w_r = w^(n / r):
            // r - 1 + (r > 2) products per r outputs instead of r (r - 1) for the evaluation form below
            const E wr = r > 2 ? wpow[n / r] : (E)0; // w_3 or w_4
            for (int t = threadIdx.x; t < n / r; t += 256) {
                const int b = t % m, qi = t / m;
                E y[4];
#pragma unroll
                for (int k = 0; k < 4; k++) y[k] = k < r ? src[(k * q + qi) * m + b] : (E)0;
                if (b) {
#pragma unroll
                    for (int k = 1; k < 4; k++)
                        if (k < r) y[k] = F::mul(fd, y[k], wpow[q * b * k]);
                }
                const int o0 = qi * r * m + b;
                if (r == 2) {
                    emit(o0, F::add(fd, y[0], y[1]));
                    emit(o0 + m, F::sub(fd, y[0], y[1]));
                } else if (r == 3) { // w^2 = -1 - w:  X1 = (y0 - y2) + w (y1 - y2),  X2 = (y0 - y1) - w (y1 - y2)
                    const E u = F::mul(fd, wr, F::sub(fd, y[1], y[2]));
                    emit(o0, F::add(fd, y[0], F::add(fd, y[1], y[2])));
                    emit(o0 + m, F::add(fd, F::sub(fd, y[0], y[2]), u));
                    emit(o0 + 2 * m, F::sub(fd, F::sub(fd, y[0], y[1]), u));
                } else {
                    const E t0 = F::add(fd, y[0], y[2]), t1 = F::sub(fd, y[0], y[2]), t2 = F::add(fd, y[1], y[3]);
                    const E t3 = F::mul(fd, wr, F::sub(fd, y[1], y[3]));
                    emit(o0, F::add(fd, t0, t2));
                    emit(o0 + m, F::add(fd, t1, t3));
                    emit(o0 + 2 * m, F::sub(fd, t0, t2));
                    emit(o0 + 3 * m, F::sub(fd, t1, t3));
                }
            }
        } else {
            for (int o = threadIdx.x; o < n; o += 256) {
                const int b = o % m, f = (o / m) % r, qi = o / (m * r);
                const E tw = wpow[q * (f * m + b)]; // f * m + b < m * r, so the exponent stays below q * m * r = n
                E acc = src[((r - 1) * q + qi) * m + b];
                for (int k = r - 2; k >= 0; k--) acc = F::add(fd, F::mul(fd, acc, tw), src[(k * q + qi) * m + b]);
                emit(o, acc);
            }
        }
        __syncthreads();
        E *t = src; src = dst; dst = t;
        m *= r;
    }
}

// The same one-workgroup Stockham transform for table fields of at most 32768 elements, carried out on LOGARITHMS: values live
// in LDS as LOG[x] (0xFFFFFFFF for 0), a twiddle product is an addition mod q - 1 and a sum one Zech-logarithm gather,
// log(a + b) = m + ZECH[n - m] for the two logs m <= n (add_ufunc.lookup, _lookup.py:153-168) -- one LDS gather per
// multiply-add instead of the seven table gathers from global memory that EXP[LOG + LOG] followed by the Zech sum costs when
// every intermediate is converted back (the reference's own FFT benchmark has such a case: 2304 points over GF(127^2)).
// ZECH is staged in LDS as 16-bit entries; LOG / EXP are touched once per element on the way in and out.
__global__ __launch_bounds__(256) void ntt_small_log_kernel(FieldDev fd, const u32 *__restrict__ in, u32 *__restrict__ out, int n, SmallArgs sa,
                                                            u32 omega, int do_scale, u32 scale)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *buf0 = reinterpret_cast<u32 *>(smem_raw), *buf1 = buf0 + n;
    uint16_t *ze = reinterpret_cast<uint16_t *>(buf1 + n);
    constexpr u32 ZERO = 0xffffffffu;
    const u32 qm1 = fd.qm1, zech_e = fd.zech_e;
    for (u32 i = threadIdx.x; i < (u32)fd.q; i += 256) ze[i] = (uint16_t)fd.zech_tab[i];
    const u32 *x = in + (i64)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const u32 v = x[i];
        buf0[i] = v ? fd.log_tab[v] : ZERO;
    }
    const u32 lw = fd.log_tab[omega];
    __syncthreads();
    auto log_add = [&](u32 a, u32 b) -> u32 {
        const u32 mm = min(a, b), nn = max(a, b); // ZERO is the largest value: it ends up in nn
        if (nn == ZERO) return mm;                // one operand (or both) zero
        const u32 z = nn - mm;
        u32 r = mm + (u32)ze[z];
        r = r >= qm1 ? r - qm1 : r;
        return z == zech_e ? ZERO : r;
    };
    u32 *src = buf0, *dst = buf1;
    int m = 1;
    for (int s = 0; s < sa.nf; s++) {
        const int r = sa.r[s], q = n / (m * r);
        const bool last = s == sa.nf - 1;
        for (int o = threadIdx.x; o < n; o += 256) {
            const int b = o % m, f = (o / m) % r, qi = o / (m * r);
            const u32 tl = ((u32)(q * (f * m + b)) * lw) % qm1; // log of w^(q (f m + b)): exponent < n <= 4096, lw < 2^15
            u32 acc = src[((r - 1) * q + qi) * m + b];
            for (int k = r - 2; k >= 0; k--) {
                if (acc != ZERO) {
                    acc += tl;
                    acc = acc >= qm1 ? acc - qm1 : acc;
                }
                acc = log_add(acc, src[(k * q + qi) * m + b]);
            }
            if (last) {
                u32 v = acc == ZERO ? 0u : fd.exp_tab[acc];
                if (do_scale) v = Lut::mul(fd, v, scale);
                out[(i64)blockIdx.x * n + o] = v;
            } else {
                dst[o] = acc;
            }
        }
        __syncthreads();
        u32 *t = src; src = dst; dst = t;
        m *= r;
    }
}

template <typename E>
__global__ void convert_in_kernel(const void *src, int dtype, E *dst, i64 n)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        E v;
        switch (dtype) {
        case GFA_U8: v = (E)((const uint8_t *)src)[i]; break;
        case GFA_U16: v = (E)((const uint16_t *)src)[i]; break;
        case GFA_U32: v = (E)((const uint32_t *)src)[i]; break;
        default: v = (E)((const uint64_t *)src)[i]; break;
        }
        dst[i] = v;
    }
}
template <typename E>
__global__ void convert_out_kernel(const E *src, void *dst, int dtype, i64 n)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        E v = src[i];
        switch (dtype) {
        case GFA_U8: ((uint8_t *)dst)[i] = (uint8_t)v; break;
        case GFA_U16: ((uint16_t *)dst)[i] = (uint16_t)v; break;
        case GFA_U32: ((uint32_t *)dst)[i] = (uint32_t)v; break;
        default: ((uint64_t *)dst)[i] = (uint64_t)v; break;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// plans (device tables per (field, n, omega)), cached per process
// ------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return GFA_OK;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        GFA_HIP(hipMalloc(&p, need));
        bytes = need;
        return GFA_OK;
    }
};

struct Plan {
    // power-of-two tile path
    int logn = 0, log1 = 0, log2 = 0; // n = 2^logn = 2^log1 * 2^log2 ; log2 == 0 => single pass
    void *w1 = nullptr, *w1q = nullptr; // twiddles of the length-2^log1 transform (and Shoup quotients)
    void *w2 = nullptr, *w2q = nullptr; // twiddles of the length-2^log2 transform
    void *powA = nullptr, *powB = nullptr, *powAq = nullptr, *powBq = nullptr;
    void *powAm = nullptr; // A * 2^32 mod p (Montgomery form), lazy 32-bit path
    int lo_bits = 0;
    // register-blocked path: full w_L^e tables (L entries) per pass, with Shoup quotients
    void *wl1 = nullptr, *wl1q = nullptr, *wl2 = nullptr, *wl2q = nullptr;
    bool reg_ready = false;
    // three-pass register path (2^21 .. 2^29 points): outer lines of 2^log0, then the two-pass transform of 2^(log1+log2)
    int log0 = 0, lo_bits2 = 0;
    void *wl0 = nullptr, *wl0q = nullptr;
    void *powA2 = nullptr, *powB2 = nullptr, *powA2q = nullptr, *powB2q = nullptr, *powA2m = nullptr; // w_M^e, M = n >> log0
    bool reg3_ready = false;
    // Goldilocks: inter-pass twiddles w^(line * k) as a table laid out like one transform's intermediate (see gl_post_table_kernel)
    void *gl_ptab2 = nullptr;
    // generic path
    void *wpow = nullptr; // n entries
    std::vector<i64> factors;
    struct Scratch *sc = nullptr; // work buffers of the (device, stream) this call runs on; set at lookup
};

// Work buffers (inter-pass intermediates, dtype conversion) are shared by every plan used on one (device, stream): they are
// grow-only, so their size is the largest transform seen on that stream rather than the sum over all cached plans.
struct Scratch {
    DevBuf ws0, ws1, cvt;
};

struct PlanKey {
    const gfa_field *f; int device; i64 n; u64 omega; int lookup;
    i64 variant; // 0: full transform; n1 > 0: distributed column pass with lines of length n1
    bool operator<(const PlanKey &o) const
    {
        return std::tie(f, device, n, omega, lookup, variant) < std::tie(o.f, o.device, o.n, o.omega, o.lookup, o.variant);
    }
};

std::mutex g_plan_mu;
std::map<PlanKey, Plan *> g_plans;
std::map<std::pair<int, hipStream_t>, Scratch *> g_scratch;

// under g_plan_mu
Plan *lookup_plan(const PlanKey &key, hipStream_t st)
{
    auto it = g_plans.find(key);
    if (it == g_plans.end()) it = g_plans.emplace(key, new Plan()).first;
    auto sk = std::make_pair(key.device, st);
    auto is = g_scratch.find(sk);
    if (is == g_scratch.end()) is = g_scratch.emplace(sk, new Scratch()).first;
    it->second->sc = is->second;
    return it->second;
}

template <class F>
int build_pow_table(const FieldDev &fd, u64 base, u64 exp_stride, i64 count, void **out, hipStream_t st)
{
    typedef typename F::elem E;
    GFA_HIP(hipMalloc(out, sizeof(E) * (size_t)std::max<i64>(count, 1)));
    hipLaunchKernelGGL((pow_table_kernel<F>), dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, fd, (E)base,
                       exp_stride, (E *)*out, count);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

template <class TW>
constexpr int qbits_of()
{
    if constexpr (TW::HAS_SHOUP) return TW::QBITS;
    else return 0;
}

__global__ void mont_table_kernel(u32 p, const u32 *w, u32 *wm, i64 count)
{
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) wm[i] = (u32)((((u64)w[i]) << 32) % p);
}

inline u64 inverse_mod_2_32(u64 p)
{ // Newton iteration, p odd
    u32 x = (u32)p;
    for (int i = 0; i < 5; i++) x *= 2u - (u32)p * x;
    return x;
}

template <class TW>
u64 shoup_quotient(const FieldDev &fd, u64 w)
{
    if constexpr (TW::HAS_SHOUP) return (w << TW::QBITS) / fd.p;
    else return 0;
}

int build_shoup(const FieldDev &fd, const void *w, i64 count, void **out, hipStream_t st, int qbits)
{
    GFA_HIP(hipMalloc(out, sizeof(u32) * (size_t)std::max<i64>(count, 1)));
    hipLaunchKernelGGL(shoup_table_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, (u32)fd.p,
                       (const u32 *)w, (u32 *)*out, count, qbits);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

std::vector<i64> prime_factors(i64 n)
{ // ascending with multiplicity, as fft_jit._prime_factors (_function.py:214-229)
    std::vector<i64> out;
    for (i64 d = 2; d * d <= n; d++)
        while (n % d == 0) { out.push_back(d); n /= d; }
    if (n > 1) out.push_back(n);
    return out;
}

constexpr size_t TILE_LDS_BYTES = 64 * 1024; // data part of the tile; two workgroups per CU

template <class F>
int max_log_tile()
{ // longest line that fits the LDS budget with one line per tile
    int lg = 0;
    while ((sizeof(typename F::elem) << (lg + 1)) <= TILE_LDS_BYTES) lg++;
    return lg; // u32: 14, u64: 13
}

template <class F, class TW>
int launch_tile(const FieldDev &fd, bool load_along, bool store_along, const void *in, void *out, const TileArgs &ta,
                i64 batch, const void *w, const void *wq, const void *pa, const void *pb, hipStream_t st)
{
    typedef typename F::elem E;
    const size_t L = (size_t)1 << ta.logL;
    const size_t lds = sizeof(E) * ((size_t)ta.lines_per_tile * (L + 1) + (ta.tw_in_lds ? L : 0));
    const unsigned grid = (unsigned)(batch * ta.tiles_per_batch);
#define GFA_TILE(LA, SA)                                                                                          \
    do {                                                                                                          \
        auto kern = ntt_tile_kernel<F, TW, LA, SA>;                                                               \
        static bool attr = false;                                                                                 \
        if (!attr) {                                                                                              \
            GFA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr = true;                                                                                          \
        }                                                                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, fd, (const E *)in, (E *)out, ta, (const E *)w,   \
                           (const E *)wq, (const E *)pa, (const E *)pb);                                          \
    } while (0)
    if (load_along && store_along) GFA_TILE(true, true);
    else if (load_along && !store_along) GFA_TILE(true, false);
    else if (!load_along && store_along) GFA_TILE(false, true);
    else GFA_TILE(false, false);
#undef GFA_TILE
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int choose_lines(int logL, size_t esize, i64 total_lines)
{
    size_t c = TILE_LDS_BYTES / (esize << logL);
    if (c < 8) c = (2 * TILE_LDS_BYTES - 4096) / (esize << logL); // long lines: one 128 KiB workgroup per CU instead
    if (c < 1) c = 1;
    if (c > 32) c = 32;
    while (c > 1 && (total_lines % (i64)c) != 0) c >>= 1;
    // round down to a power of two
    size_t p2 = 1;
    while (p2 * 2 <= c) p2 *= 2;
    return (int)p2;
}

// Runs the transform on element-typed (F::elem) buffers.
template <class F, class TW>
int run_pow2(gfa_field *f, const FieldDev &fd, Plan *pl, const void *in, void *out, i64 n, i64 batch, u64 omega,
             int do_scale, u64 scale, hipStream_t st)
{
    typedef typename F::elem E;
    if (!pl->w1) {
        const int maxlg = std::min(max_log_tile<F>(), 12); // keep >= 4 lines per tile for the strided passes
        pl->logn = 0;
        while (((i64)1 << pl->logn) < n) pl->logn++;
        if (pl->logn <= max_log_tile<F>() - 1 && pl->logn <= 13) {
            pl->log1 = pl->logn; pl->log2 = 0;
        } else {
            pl->log1 = (pl->logn + 1) / 2; pl->log2 = pl->logn - pl->log1;
            if (pl->log1 > maxlg) { set_error("gfa_ntt: power-of-two length beyond the two-pass range"); return GFA_ERR_UNSUPPORTED; }
        }
        int rc;
        const i64 n1 = (i64)1 << pl->log1, n2 = (i64)1 << pl->log2;
        // w_{n1} = omega^(n2): table of w_{n1}^t, t < n1/2
        if ((rc = build_pow_table<F>(fd, omega, (u64)n2, std::max<i64>(n1 / 2, 1), &pl->w1, st))) return rc;
        if (TW::HAS_SHOUP && (rc = build_shoup(fd, pl->w1, std::max<i64>(n1 / 2, 1), &pl->w1q, st, qbits_of<TW>()))) return rc;
        if (pl->log2) {
            if ((rc = build_pow_table<F>(fd, omega, (u64)n1, n2 / 2, &pl->w2, st))) return rc;
            if (TW::HAS_SHOUP && (rc = build_shoup(fd, pl->w2, n2 / 2, &pl->w2q, st, qbits_of<TW>()))) return rc;
            pl->lo_bits = (pl->logn + 1) / 2;
            const i64 nb = (i64)1 << pl->lo_bits, na = n >> pl->lo_bits;
            if ((rc = build_pow_table<F>(fd, omega, (u64)nb, na, &pl->powA, st))) return rc;
            if ((rc = build_pow_table<F>(fd, omega, 1, nb, &pl->powB, st))) return rc;
        }
    }
    (void)f;
    if (pl->log2 == 0) {
        TileArgs ta{};
        ta.logL = pl->log1;
        ta.lines_per_tile = choose_lines(pl->log1, sizeof(E), batch);
        ta.tiles_per_batch = 1;
        // treat the whole batch as `batch` lines of one "batch item"
        ta.in_stride_c = n; ta.in_stride_t = 1; ta.out_stride_c = n; ta.out_stride_t = 1;
        ta.in_batch_stride = 0; ta.out_batch_stride = 0;
        ta.tiles_per_batch = (int)(batch / ta.lines_per_tile);
        ta.do_scale = do_scale; ta.scale = scale;
        ta.tw_in_lds = pl->log1 <= 12;
        return launch_tile<F, TW>(fd, true, true, in, out, ta, 1, pl->w1, pl->w1q, nullptr, nullptr, st);
    }
    const i64 n1 = (i64)1 << pl->log1, n2 = (i64)1 << pl->log2;
    int rc;
    if ((rc = pl->sc->ws0.ensure(sizeof(E) * (size_t)(n * batch)))) return rc;
    // pass 1: columns j2 (lines), position j1 with stride n2; output A[k1*n2 + j2] * w^(j2*k1)
    {
        TileArgs ta{};
        ta.logL = pl->log1;
        ta.lines_per_tile = choose_lines(pl->log1, sizeof(E), n2);
        ta.tiles_per_batch = (int)(n2 / ta.lines_per_tile);
        ta.in_stride_c = 1; ta.in_stride_t = n2; ta.out_stride_c = 1; ta.out_stride_t = n2;
        ta.in_batch_stride = n; ta.out_batch_stride = n;
        ta.post_twiddle = 1; ta.lo_bits = pl->lo_bits; ta.n_mask = (u64)n - 1;
        ta.tw_in_lds = pl->log1 <= 12;
        if ((rc = launch_tile<F, TW>(fd, false, false, in, pl->sc->ws0.p, ta, batch, pl->w1, pl->w1q, pl->powA, pl->powB, st)))
            return rc;
    }
    // pass 2: rows k1 (lines), contiguous along j2; output X[k1 + n1*k2]
    {
        TileArgs ta{};
        ta.logL = pl->log2;
        ta.lines_per_tile = choose_lines(pl->log2, sizeof(E), n1);
        ta.tiles_per_batch = (int)(n1 / ta.lines_per_tile);
        ta.in_stride_c = n2; ta.in_stride_t = 1; ta.out_stride_c = 1; ta.out_stride_t = n1;
        ta.in_batch_stride = n; ta.out_batch_stride = n;
        ta.do_scale = do_scale; ta.scale = scale;
        ta.tw_in_lds = pl->log2 <= 12;
        if ((rc = launch_tile<F, TW>(fd, true, false, pl->sc->ws0.p, out, ta, batch, pl->w2, pl->w2q, nullptr, nullptr, st)))
            return rc;
    }
    return GFA_OK;
}


// Goldilocks shift-twiddle networks: for the line table at `wl` (w_L = omega^(n_total / L), L = R1 * R2) the odd exponents u1, u2
// with w_L^R2 = (2^(192/R1))^u1 and w_L^R1 = (2^(192/R2))^u2, stored as the kernel wants them: perm1 = u1^-1 mod R1, perm2 = u2.
struct GlPerm { int perm1, perm2; };
std::mutex g_glperm_mu;
std::map<const void *, GlPerm> g_glperm;

inline void goldi_register_perms(const FieldDev &fd, u64 omega, i64 n_total, int logL, const void *wl)
{
    const int l1 = (logL + 1) / 2, l2 = logL / 2; // the (R1, R2) split launch_reg uses
    {   // the map is keyed by the table's device address: an entry left by a freed table whose address is reused must not
        // outlive this call, whichever way it ends (a stale pair would switch the kernel to shift twiddles with the wrong permutations)
        std::lock_guard<std::mutex> lock(g_glperm_mu);
        g_glperm.erase(wl);
    }
    u64 wL = 0;
    HostArith::pow(fd, omega, (i64)(n_total >> logL), &wL);
    auto find_u = [&](int lr, int lother) -> int { // odd u with (w_L^(2^lother)) == (2^(192 / 2^lr))^u, 0 if none
        if (lr == 0) return 1;
        u64 w = 0, canon = 0;
        HostArith::pow(fd, wL, (i64)1 << lother, &w);
        HostArith::pow(fd, 2, (i64)(192 >> lr), &canon);
        for (int u = 1; u < (1 << lr) || u == 1; u += 2) {
            u64 c = 0;
            HostArith::pow(fd, canon, u, &c);
            if (c == w) return u;
        }
        return 0;
    };
    const int u1 = find_u(l1, l2), u2 = find_u(l2, l1);
    if (!u1 || !u2) return; // (cannot happen for a primitive root of unity; the kernel then keeps the general products)
    int inv1 = 1;
    for (int t = 1; t < (1 << l1); t += 2)
        if (((t * u1) & ((1 << l1) - 1)) == 1) inv1 = t;
    std::lock_guard<std::mutex> lock(g_glperm_mu);
    g_glperm[wl] = GlPerm{inv1, u2};
}

template <class F, class TW, int LOGR1, int LOGR2, int THREADS, bool SPLIT = false>
int launch_reg_tt(const FieldDev &fd, const void *in, void *out, RegArgs ra, i64 batch, const void *wl, const void *wlq,
                  const void *pa, const void *paq, const void *pb, const void *pbq, const void *pam, hipStream_t st)
{
    typedef typename F::elem E;
    constexpr int R1 = 1 << LOGR1, R2 = 1 << LOGR2, L = R1 * R2, C = THREADS / R1;
    constexpr size_t lds = sizeof(E) * ((size_t)C * ((SPLIT ? R1 / 2 : R1) * (R2 + 1) + 1) + (TW::HAS_SHOUP ? 2 : 1) * L); // quotient table only with Shoup twiddles
    ra.tiles_per_batch = (int)((ra.total_lines + C - 1) / C);
    const unsigned grid = (unsigned)(batch * ra.tiles_per_batch);
    ra.xcd_remap = ((grid % 8) == 0 && grid >= 16) ? 1 : 0; // XCD-aware tile order (plain order: 0.31 instead of 0.25 ms at 2^20 x 64)
    {
        const i64 lim = ((i64)1 << 31) / (i64)sizeof(E);
        const i64 in_chunks = ra.in_split < 31 ? ((i64)L >> ra.in_split) : 0, out_chunks = ra.out_split < 31 ? ((i64)L >> ra.out_split) : 0;
        if ((C - 1) * ra.in_stride_c + (L - 1) * ra.in_stride_t + in_chunks * ra.in_chunk_stride >= lim ||
            (C - 1) * ra.out_stride_c + (L - 1) * ra.out_stride_t + out_chunks * ra.out_chunk_stride >= lim) {
            set_error("register NTT: tile extent exceeds the 32-bit offset range");
            return GFA_ERR_UNSUPPORTED;
        }
    }
    if constexpr (std::is_same<TW, TwGoldi>::value) {
        bool shift = false;
        if (!ra.pre_twiddle) {
            std::lock_guard<std::mutex> lock(g_glperm_mu);
            auto it = g_glperm.find(wl);
            if (it != g_glperm.end()) { ra.perm1 = it->second.perm1; ra.perm2 = it->second.perm2; shift = true; }
        }
        // four waves per SIMD for the strided passes of the three-pass driver (see WAVES above)
        const bool strided = ra.waves4 && !ra.load_along_line && !ra.store_along_line && LOGR1 == 5 && LOGR2 == 4; // (32 x 32: 128 registers spill)
#define GFA_GL_LAUNCH(SH, WV)                                                                                                                    \
    do {                                                                                                                                         \
        auto kern = ntt_reg_kernel_gl<LOGR1, LOGR2, THREADS, SPLIT, SH, WV>;                                                                     \
        static bool attr = false;                                                                                                                \
        if (!attr) {                                                                                                                             \
            GFA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                            \
            attr = true;                                                                                                                         \
        }                                                                                                                                        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, st, fd, (const u64 *)in, (u64 *)out, ra, (const u64 *)wl,     \
                           (const u64 *)pa, (const u64 *)pb);                                                                                    \
    } while (0)
        if (shift && strided) GFA_GL_LAUNCH(true, 4);
        else if (shift) GFA_GL_LAUNCH(true, GFA_GL_WAVES);
        else GFA_GL_LAUNCH(false, GFA_GL_WAVES);
#undef GFA_GL_LAUNCH
    } else {
        auto kern = ntt_reg_kernel<F, TW, LOGR1, LOGR2, THREADS, SPLIT>;
        static bool attr = false;
        if (!attr) {
            GFA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, st, fd, (const E *)in, (E *)out, ra, (const E *)wl,
                           (const E *)wlq, (const E *)pa, (const E *)paq, (const E *)pb, (const E *)pbq, (const E *)pam);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

template <class F, class TW, int LOGR1, int LOGR2>
int launch_reg_t(const FieldDev &fd, const void *in, void *out, const RegArgs &ra, i64 batch, const void *wl,
                 const void *wlq, const void *pa, const void *paq, const void *pb, const void *pbq, const void *pam, hipStream_t st)
{
    typedef typename F::elem E;
    if constexpr (sizeof(E) == 4 && LOGR1 == 5) {
        // 32 lines per tile (128-byte global segments, one 1024-thread workgroup per CU) or 16 lines (two 512-thread ones)
        // (measured and dropped: 1024-thread workgroups, and the whole line staged in one round)
        return launch_reg_tt<F, TW, LOGR1, LOGR2, 512, true>(fd, in, out, ra, batch, wl, wlq, pa, paq, pb, pbq, pam, st); // two-round LDS exchange: 3 workgroups per CU
    } else if constexpr (sizeof(E) == 8 && LOGR1 == 5 && !TW::HAS_SHOUP) {
        // 64-bit elements: 16 lines per tile (128-byte global segments) in one 512-thread workgroup per CU, or 8 lines in
        // 256-thread workgroups, two per CU now that the unused quotient table is no longer reserved (GFA_NTT_T64=256)
        // two-round exchange: half the LDS buffer, three workgroups per CU (Goldilocks 2^20 x 16: 0.283 -> 0.251 ms)
        // r04: with the cheaper Goldilocks arithmetic the strided passes of the three-pass transform are no longer bound by the
        // vector ALU alone, and 16-line tiles (128-byte segments, 512 threads) win there: 2^26 points 0.92 -> 0.86 ms; two-pass
        // column passes and contiguous lines still prefer 8 lines (2^20 x 16: 0.19 against 0.22 ms; 2^10 x 16384: 0.072 / 0.098)
        if (ra.waves4) return launch_reg_tt<F, TW, LOGR1, LOGR2, 512, true>(fd, in, out, ra, batch, wl, wlq, pa, paq, pb, pbq, pam, st);
        return launch_reg_tt<F, TW, LOGR1, LOGR2, 256, true>(fd, in, out, ra, batch, wl, wlq, pa, paq, pb, pbq, pam, st);
    } else {
        return launch_reg_tt<F, TW, LOGR1, LOGR2, 256>(fd, in, out, ra, batch, wl, wlq, pa, paq, pb, pbq, pam, st);
    }
}

template <class F, class TW>
int launch_reg(const FieldDev &fd, int logL, const void *in, void *out, const RegArgs &ra, i64 batch, const void *wl,
               const void *wlq, const void *pa, const void *paq, const void *pb, const void *pbq, const void *pam, hipStream_t st)
{
#define GFA_REG(A, B) return launch_reg_t<F, TW, A, B>(fd, in, out, ra, batch, wl, wlq, pa, paq, pb, pbq, pam, st)
    switch (logL) {
    case 2: GFA_REG(1, 1);
    case 3: GFA_REG(2, 1);
    case 4: GFA_REG(2, 2);
    case 5: GFA_REG(3, 2);
    case 6: GFA_REG(3, 3);
    case 7: GFA_REG(4, 3);
    case 8: GFA_REG(4, 4);
    case 9: GFA_REG(5, 4);
    case 10: GFA_REG(5, 5);
    default: set_error("register NTT: unsupported line length"); return GFA_ERR_UNSUPPORTED;
    }
#undef GFA_REG
}

constexpr int REG_MAX_LOG = 10; // longest line of the register-blocked kernel

template <class F, class TW>
int build_line_tables(const FieldDev &fd, u64 omega, i64 n_total, int logL, void **wl, void **wlq, hipStream_t st)
{ // w_L^e for e < L, w_L = omega^(n_total / L)
    const i64 L = (i64)1 << logL;
    int rc;
    if ((rc = build_pow_table<F>(fd, omega, (u64)(n_total / L), L, wl, st))) return rc;
    if (TW::HAS_SHOUP && (rc = build_shoup(fd, *wl, L, wlq, st, qbits_of<TW>()))) return rc;
    if constexpr (std::is_same<F, Goldilocks>::value) goldi_register_perms(fd, omega, n_total, logL, *wl);
    return GFA_OK;
}

// Two-level table of w^e, e < domain, w = omega^mult: w^e = A[e >> lo_bits] * B[e & mask]
template <class F, class TW>
int build_post_tables_at(const FieldDev &fd, u64 omega, i64 domain, u64 mult, void **A, void **B, void **Aq, void **Bq, void **Am,
                         int *lo_bits_out, hipStream_t st)
{
    int logn = 0;
    while (((i64)1 << logn) < domain) logn++;
    const int lo_bits = (logn + 1) / 2;
    *lo_bits_out = lo_bits;
    const i64 nb = (i64)1 << lo_bits, na = domain >> lo_bits;
    int rc;
    if ((rc = build_pow_table<F>(fd, omega, mult * (u64)nb, na, A, st))) return rc;
    if ((rc = build_pow_table<F>(fd, omega, mult, nb, B, st))) return rc;
    if (TW::HAS_SHOUP) {
        if ((rc = build_shoup(fd, *A, na, Aq, st, qbits_of<TW>()))) return rc;
        if ((rc = build_shoup(fd, *B, nb, Bq, st, qbits_of<TW>()))) return rc;
    }
    if constexpr (has_redc<TW>()) {
        // (w << 32) mod p == floor-free Montgomery form; reuse the quotient kernel's arithmetic: w*2^32 - floor(w*2^32/p)*p
        GFA_HIP(hipMalloc(Am, sizeof(u32) * (size_t)na));
        hipLaunchKernelGGL(mont_table_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, st, (u32)fd.p,
                           (const u32 *)*A, (u32 *)*Am, na);
        GFA_HIP(hipGetLastError());
    }
    return GFA_OK;
}

template <class F, class TW>
int build_post_tables(const FieldDev &fd, u64 omega, i64 n_total, Plan *pl, hipStream_t st)
{
    return build_post_tables_at<F, TW>(fd, omega, n_total, 1, &pl->powA, &pl->powB, &pl->powAq, &pl->powBq, &pl->powAm,
                                       &pl->lo_bits, st);
}

// Goldilocks: the inter-pass twiddle table of a column pass with `lines` lines of length count / lines (gl_post_table_kernel)
inline int build_gl_post_table(const void *powA, const void *powB, int lo_bits, i64 lines, i64 count, void **out, hipStream_t st)
{
    if (*out) return GFA_OK;
    void *tab = nullptr;
    GFA_HIP(hipMalloc(&tab, sizeof(u64) * (size_t)count));
    hipLaunchKernelGGL(gl_post_table_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, (const u64 *)powA, (const u64 *)powB,
                       lo_bits, (u64 *)tab, (u32)lines, (u32)count);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        (void)hipFree(tab); // *out stays null: a later call builds the table again
        return gfa::hip_fail(le, "gl_post_table_kernel");
    }
    *out = tab;
    return GFA_OK;
}

// Where the rows of a batched transform live.  Contiguous (chunk_len == 0): row b at b * n.  Chunked (the exchange buffers of
// the distributed transform): element j of row b at (j / chunk_len) * chunk_stride + b * row_stride + (j % chunk_len).
struct RowLayout {
    i64 chunk_len = 0, chunk_stride = 0, row_stride = 0;
};
thread_local RowLayout g_layout_in, g_layout_out; // set by gfa_ntt_chunked around its call into the plain transform

inline int log2_exact(i64 v)
{
    int l = 0;
    while (((i64)1 << l) < v) l++;
    return ((i64)1 << l) == v ? l : -1;
}

// power-of-two n, 4 <= n <= 2^20, on the register-blocked kernel: one pass up to 2^10, else four-step in two passes
template <class F, class TW>
int run_pow2_reg(const FieldDev &fd, Plan *pl, const void *in, void *out, i64 n, i64 batch, u64 omega, int do_scale,
                 u64 scale, hipStream_t st)
{
    typedef typename F::elem E;
    int rc;
    const RowLayout li = g_layout_in, lo = g_layout_out;
    if (!pl->reg_ready) {
        pl->logn = 0;
        while (((i64)1 << pl->logn) < n) pl->logn++;
        if (pl->logn <= REG_MAX_LOG) { pl->log1 = pl->logn; pl->log2 = 0; }
        else { pl->log1 = (pl->logn + 1) / 2; pl->log2 = pl->logn - pl->log1; }
        if ((rc = build_line_tables<F, TW>(fd, omega, n, pl->log1, &pl->wl1, &pl->wl1q, st))) return rc;
        if (pl->log2) {
            if ((rc = build_line_tables<F, TW>(fd, omega, n, pl->log2, &pl->wl2, &pl->wl2q, st))) return rc;
            if ((rc = build_post_tables<F, TW>(fd, omega, n, pl, st))) return rc;
        }
        GFA_HIP(hipStreamSynchronize(st)); // tables filled on this stream; the shared plan may next run on another
        pl->reg_ready = true;
    }
    // chunked rows: the lane-owned low part of a position (R2 values on the load side, R1 on the store side) must not
    // cross a chunk boundary, and in the two-pass form a chunk must hold whole sub-lines
    auto bad_layout = []() { set_error("gfa_ntt_chunked: chunk length not supported for this transform length"); return GFA_ERR_UNSUPPORTED; };
    if (pl->log2 == 0) {
        RegArgs ra{};
        ra.in_stride_c = n; ra.in_stride_t = 1; ra.out_stride_c = n; ra.out_stride_t = 1;
        if (li.chunk_len) {
            const int sp = log2_exact(li.chunk_len);
            if (sp < pl->log1 / 2) return bad_layout();
            ra.in_stride_c = li.row_stride; ra.in_split = sp; ra.in_chunk_stride = li.chunk_stride;
        }
        if (lo.chunk_len) {
            const int sp = log2_exact(lo.chunk_len);
            if (sp < (pl->log1 + 1) / 2) return bad_layout();
            ra.out_stride_c = lo.row_stride; ra.out_split = sp; ra.out_chunk_stride = lo.chunk_stride;
        }
        ra.total_lines = batch;
        ra.load_along_line = 1; ra.store_along_line = 1;
        ra.do_scale = do_scale; ra.scale = scale; ra.scale_q = shoup_quotient<TW>(fd, scale);
        return launch_reg<F, TW>(fd, pl->log1, in, out, ra, 1, pl->wl1, pl->wl1q, nullptr, nullptr, nullptr, nullptr, nullptr, st);
    }
    const i64 n1 = (i64)1 << pl->log1, n2 = (i64)1 << pl->log2;
    int in_split = 31, out_split = 31;
    if (li.chunk_len) { // pass 1 reads position t of column c at element t * n2 + c
        const int sp = log2_exact(li.chunk_len);
        if (sp < pl->log2 || sp - pl->log2 < pl->log1 / 2) return bad_layout();
        in_split = sp - pl->log2;
    }
    if (lo.chunk_len) { // pass 2 stores position t of column k1 at element k1 + n1 * t
        const int sp = log2_exact(lo.chunk_len);
        if (sp < pl->log1 || sp - pl->log1 < (pl->log2 + 1) / 2) return bad_layout();
        out_split = sp - pl->log1;
    }
    const i64 in_row = li.chunk_len ? li.row_stride : n, out_row = lo.chunk_len ? lo.row_stride : n;
    // (sub-batches that keep the pass-1 -> pass-2 intermediate inside the 256 MiB Infinity Cache were measured and lose to two plain
    // passes over the whole batch, profiles/r03_ntt_fused_skeleton.txt)
    const i64 sub = batch;
    if ((rc = pl->sc->ws0.ensure(sizeof(E) * (size_t)(n * sub)))) return rc;
    for (i64 b0 = 0; b0 < batch; b0 += sub) {
        const i64 nb = std::min(sub, batch - b0);
        const E *src = (const E *)in + b0 * in_row;
        E *dst = (E *)out + b0 * out_row;
        { // pass 1: the n2 columns (length n1, stride n2), then * w^(j2*k1); same layout out
            RegArgs ra{};
            ra.in_stride_c = 1; ra.in_stride_t = n2; ra.out_stride_c = 1; ra.out_stride_t = n2;
            ra.in_batch_stride = in_row; ra.out_batch_stride = n;
            ra.in_split = in_split; ra.in_chunk_stride = li.chunk_stride;
            ra.total_lines = n2;
            ra.post_twiddle = 1; ra.lo_bits = pl->lo_bits; ra.n_mask = (u64)n - 1; ra.pinv = inverse_mod_2_32(fd.p);
            // (Goldilocks: a table of the n twiddles in place of the per-thread progression was measured here too and loses,
            // 2^20 x 16 0.188 -> 0.199 ms: 8 MiB of table do not stay in L2 beside the data)
            if ((rc = launch_reg<F, TW>(fd, pl->log1, src, pl->sc->ws0.p, ra, nb, pl->wl1, pl->wl1q, pl->powA, pl->powAq, pl->powB,
                                        pl->powBq, pl->powAm, st)))
                return rc;
        }
        { // pass 2: the n1 rows (contiguous), stored transposed: X[k1 + n1*k2]
            RegArgs ra{};
            ra.in_stride_c = n2; ra.in_stride_t = 1; ra.out_stride_c = 1; ra.out_stride_t = n1;
            ra.in_batch_stride = n; ra.out_batch_stride = out_row;
            ra.out_split = out_split; ra.out_chunk_stride = lo.chunk_stride;
            ra.total_lines = n1;
            ra.load_along_line = 1; ra.store_along_line = 0;
            ra.do_scale = do_scale; ra.scale = scale; ra.scale_q = shoup_quotient<TW>(fd, scale);
            if ((rc = launch_reg<F, TW>(fd, pl->log2, pl->sc->ws0.p, dst, ra, nb, pl->wl2, pl->wl2q, nullptr, nullptr, nullptr,
                                        nullptr, nullptr, st)))
                return rc;
        }
    }
    return GFA_OK;
}

// power-of-two n, 2^21 <= n <= 2^29, in THREE passes of the register-blocked kernel (each pass reads and writes every
// point once): n = L0 * M, M = L1 * L2.
//   A : the M columns of length L0 (stride M), * w_n^(jr*k1)                   -> ws[k1*M + jr]
//   B1: per row k1, the L2 columns of length L1 (stride L2), * w_M^(j3*k2)     -> ws[k1*M + k2*L2 + j3]   (in place)
//   B2: per (k2, k1) the contiguous row of length L2                           -> X[k1 + L0*(k2 + L1*k3)]
// B2 takes k2 as the "batch" index and k1 as the line index, so that the C lines of a tile are adjacent in the output.
// Returns GFA_ERR_UNSUPPORTED (before touching anything) when a tile would leave the kernel's 32-bit offset range.
template <class F, class TW>
int run_pow2_reg3(const FieldDev &fd, Plan *pl, const void *in, void *out, i64 n, i64 batch, u64 omega, int do_scale,
                  u64 scale, hipStream_t st)
{
    typedef typename F::elem E;
    int rc;
    int logn = 0;
    while (((i64)1 << logn) < n) logn++;
    // measured on 2^22 .. 2^26 points (tools/ntt3_tune.py): longest lines in the widest-strided pass, shortest in the last
    int log0 = (logn + 2) / 3, log1 = (logn - log0 + 1) / 2;
    // r04 (Goldilocks, 2^28 points): 1024-point lines in the widest-strided pass lose to 256-point ones there, 4.34 -> 4.04 ms
    if (std::is_same<TW, TwGoldi>::value && logn == 28) { log0 = 8; log1 = 10; }
    if (pl->reg3_ready) { log0 = pl->log0; log1 = pl->log1; }
    const int log2 = logn - log0 - log1;
    const i64 L0 = (i64)1 << log0, L1 = (i64)1 << log1, L2 = (i64)1 << log2, M = L1 * L2;
    {
        const i64 lim = ((i64)1 << 31) / (i64)sizeof(E), cmax = 32;
        if (log2 > REG_MAX_LOG || log2 < 2 || log1 > REG_MAX_LOG || log0 < 2 || cmax + (L0 - 1) * M >= lim || cmax + (L2 - 1) * L0 * L1 >= lim ||
            cmax * M + L2 >= lim)
            return GFA_ERR_UNSUPPORTED;
    }
    if (!pl->reg3_ready) {
        pl->logn = logn; pl->log0 = log0; pl->log1 = log1; pl->log2 = log2;
        if ((rc = build_line_tables<F, TW>(fd, omega, n, log0, &pl->wl0, &pl->wl0q, st))) return rc;
        if ((rc = build_line_tables<F, TW>(fd, omega, n, log1, &pl->wl1, &pl->wl1q, st))) return rc;
        if ((rc = build_line_tables<F, TW>(fd, omega, n, log2, &pl->wl2, &pl->wl2q, st))) return rc;
        if ((rc = build_post_tables<F, TW>(fd, omega, n, pl, st))) return rc;
        if ((rc = build_post_tables_at<F, TW>(fd, omega, M, (u64)L0, &pl->powA2, &pl->powB2, &pl->powA2q, &pl->powB2q,
                                              &pl->powA2m, &pl->lo_bits2, st)))
            return rc;
        if constexpr (std::is_same<TW, TwGoldi>::value) {
            // Goldilocks: the L0 sub-transforms of M points share one table of M inter-pass twiddles (at most 8 MiB;
            // 2^26 points: 0.874 -> 0.847 ms).  Built here with the other tables of the plan.
            if ((rc = build_gl_post_table(pl->powA2, pl->powB2, pl->lo_bits2, L2, M, &pl->gl_ptab2, st))) return rc;
        }
        // the tables were filled by kernels on THIS stream; the plan is shared and may next be used from another one
        GFA_HIP(hipStreamSynchronize(st));
        pl->reg3_ready = true;
    }
    if ((rc = pl->sc->ws0.ensure(sizeof(E) * (size_t)n))) return rc;
    E *ws = (E *)pl->sc->ws0.p;
    for (i64 b = 0; b < batch; b++) {
        const E *src = (const E *)in + b * n;
        E *dst = (E *)out + b * n;
        {
            RegArgs ra{};
            ra.in_stride_c = 1; ra.in_stride_t = M; ra.out_stride_c = 1; ra.out_stride_t = M;
            ra.total_lines = M; ra.waves4 = 1; ra.lazy_out = 1;
            ra.post_twiddle = 1; ra.lo_bits = pl->lo_bits; ra.n_mask = (u64)n - 1; ra.pinv = inverse_mod_2_32(fd.p);
            if ((rc = launch_reg<F, TW>(fd, log0, src, ws, ra, 1, pl->wl0, pl->wl0q, pl->powA, pl->powAq, pl->powB, pl->powBq,
                                        pl->powAm, st)))
                return rc;
        }
        {
            RegArgs ra{};
            ra.in_stride_c = 1; ra.in_stride_t = L2; ra.out_stride_c = 1; ra.out_stride_t = L2;
            ra.in_batch_stride = M; ra.out_batch_stride = M;
            ra.total_lines = L2; ra.waves4 = 1; ra.lazy_out = 1;
            ra.post_twiddle = 1; ra.lo_bits = pl->lo_bits2; ra.n_mask = (u64)M - 1; ra.pinv = inverse_mod_2_32(fd.p);
            const void *pa = pl->powA2;
            if constexpr (std::is_same<TW, TwGoldi>::value) {
                ra.post_twiddle = 2; pa = pl->gl_ptab2; // built with the plan
            }
            if ((rc = launch_reg<F, TW>(fd, log1, ws, ws, ra, L0, pl->wl1, pl->wl1q, pa, pl->powA2q, pl->powB2,
                                        pl->powB2q, pl->powA2m, st)))
                return rc;
        }
        {
            RegArgs ra{};
            ra.in_batch_stride = L2; ra.out_batch_stride = L0;
            ra.in_stride_c = M; ra.in_stride_t = 1; ra.out_stride_c = 1; ra.out_stride_t = L0 * L1;
            ra.total_lines = L0;
            ra.load_along_line = 1; ra.store_along_line = 0;
            ra.do_scale = do_scale; ra.scale = scale; ra.scale_q = shoup_quotient<TW>(fd, scale);
            if ((rc = launch_reg<F, TW>(fd, log2, ws, dst, ra, L1, pl->wl2, pl->wl2q, nullptr, nullptr, nullptr, nullptr, nullptr,
                                        st)))
                return rc;
        }
    }
    return GFA_OK;
}

template <class F>
int run_generic(const FieldDev &fd, Plan *pl, const void *in, void *out, i64 n, i64 batch, u64 omega, int do_scale,
                u64 scale, hipStream_t st)
{
    typedef typename F::elem E;
    int rc;
    if (!pl->wpow) {
        pl->factors = prime_factors(n);
        if ((rc = build_pow_table<F>(fd, omega, 1, n, &pl->wpow, st))) return rc;
    }
    const size_t bytes = sizeof(E) * (size_t)(n * batch);
    const int S = (int)pl->factors.size();
    if (S == 0) { // n == 1
        if (in != out) GFA_HIP(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, st));
        return GFA_OK;
    }
    if (n <= 4096 && S <= 24 && batch <= 0x7fffffff) { // whole transform in one workgroup's LDS: one launch
        SmallArgs sa{};
        sa.nf = 0;
        for (int s = 0; s < S; s++) { // pairs of 2 become one radix-4 stage: the same products, half the trips through LDS
            const int r = (int)pl->factors[S - 1 - s];
            if (r == 2 && s + 1 < S && pl->factors[S - 2 - s] == 2) { sa.r[sa.nf++] = 4; s++; }
            else sa.r[sa.nf++] = r;
        }
        if constexpr (std::is_same<F, Lut>::value) {
            if (fd.q <= 32768) { // on logarithms, ZECH in LDS
                static bool lattr = false;
                if (!lattr) {
                    GFA_HIP(hipFuncSetAttribute((const void *)ntt_small_log_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    lattr = true;
                }
                const size_t lds = 2 * sizeof(u32) * (size_t)n + 2 * (((size_t)fd.q + 7) & ~(size_t)7);
                hipLaunchKernelGGL(ntt_small_log_kernel, dim3((unsigned)batch), dim3(256), lds, st, fd, (const u32 *)in, (u32 *)out, (int)n, sa,
                                   (u32)omega, do_scale, (u32)scale);
                GFA_HIP(hipGetLastError());
                return GFA_OK;
            }
        }
        auto kern = ntt_small_kernel<F>;
        static bool attr = false;
        if (!attr) {
            GFA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(256), 2 * sizeof(E) * (size_t)n, st, fd, (const E *)in, (E *)out, (int)n, sa,
                           (const E *)pl->wpow, do_scale, (E)scale);
        GFA_HIP(hipGetLastError());
        return GFA_OK;
    }
    if ((rc = pl->sc->ws0.ensure(bytes))) return rc;
    if ((rc = pl->sc->ws1.ensure(bytes))) return rc;
    const E *src = (const E *)in;
    i64 m = 1;
    const int grid = (int)std::min<i64>((n * batch + 255) / 256, 256 * 16);
    for (int s = 0; s < S; s++) {
        const i64 r = pl->factors[S - 1 - s];
        const i64 q = n / (m * r);
        const bool last = s == S - 1;
        E *dst = last ? ((S == 1 && in == out) ? (E *)pl->sc->ws0.p : (E *)out) : (E *)((s & 1) ? pl->sc->ws1.p : pl->sc->ws0.p);
        hipLaunchKernelGGL((ntt_stage_kernel<F>), dim3(grid), dim3(256), 0, st, fd, src, dst, n, r, m, q,
                           (const E *)pl->wpow, batch, (last && do_scale) ? 1 : 0, (E)scale);
        GFA_HIP(hipGetLastError());
        if (last && dst != (E *)out) GFA_HIP(hipMemcpyAsync(out, dst, bytes, hipMemcpyDeviceToDevice, st));
        src = dst;
        m *= r;
    }
    return GFA_OK;
}

bool is_pow2(i64 n) { return n > 0 && (n & (n - 1)) == 0; }

template <class F>
int run_typed(gfa_field *f, const FieldDev &fd, Plan *pl, const void *in, void *out, i64 n, i64 batch, u64 omega,
              int do_scale, u64 scale, int dtype, hipStream_t st)
{
    typedef typename F::elem E;
    const int native = sizeof(E) == 4 ? GFA_U32 : GFA_U64;
    const void *ein = in;
    void *eout = out;
    int rc;
    if (dtype != native) { // widen / narrow through a scratch buffer in the arithmetic's element type
        if ((rc = pl->sc->cvt.ensure(sizeof(E) * (size_t)(n * batch)))) return rc;
        const int grid = (int)std::min<i64>((n * batch + 255) / 256, 256 * 16);
        hipLaunchKernelGGL((convert_in_kernel<E>), dim3(grid), dim3(256), 0, st, in, dtype, (E *)pl->sc->cvt.p, n * batch);
        GFA_HIP(hipGetLastError());
        ein = pl->sc->cvt.p; eout = pl->sc->cvt.p;
    }
    constexpr bool prime_kind = std::is_same<F, Prime32>::value || std::is_same<F, Prime64>::value ||
                                std::is_same<F, Goldilocks>::value;
    bool done = false;
    if constexpr (prime_kind) if (is_pow2(n) && n >= 2) {
        int lg = 0;
        while (((i64)1 << lg) < n) lg++;
        if constexpr (std::is_same<F, Prime32>::value) {
            // GF(65537), 2^16 points: whole transform in one workgroup's registers, shift twiddles (gfa_ntt_fermat.hip)
            if (ntt_fermat16_eligible(fd, n, batch) && (!do_scale || scale == fd.p - (u64)(65536 / n)) && !g_layout_in.chunk_len && !g_layout_out.chunk_len) { // 1 / n = -2^16 / n: 2^16 = -1
                rc = ntt_fermat16(ein, eout, batch, omega, do_scale ? 1 : 0, st);
                if (rc && rc != GFA_ERR_UNSUPPORTED) return rc;
                done = rc == GFA_OK;
            }
        }
        if constexpr (std::is_same<F, Prime32>::value) {
            // p < 2^26: signed Montgomery butterflies, table twiddles (gfa_ntt_m32.hip)
            if (!done && ntt_m32_eligible(fd, n) && !g_layout_in.chunk_len && !g_layout_out.chunk_len) {
                if ((rc = pl->sc->ws0.ensure(ntt_m32_scratch_bytes(n, batch)))) return rc;
                if ((rc = ntt_m32(fd, ein, eout, pl->sc->ws0.p, n, batch, omega, do_scale, scale, st))) return rc;
                done = true;
            }
        }
        if (done) {
        } else if (lg >= 2 && lg <= 2 * REG_MAX_LOG) {
            if constexpr (std::is_same<F, Prime32>::value) {
                if (fd.p < (1ull << 24)) rc = run_pow2_reg<F, TwShoupWide>(fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
                else if (fd.p < (1ull << 30)) rc = run_pow2_reg<F, TwShoupLazy>(fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
                else if (fd.p < (1ull << 31)) rc = run_pow2_reg<F, TwShoup32>(fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
                else rc = run_pow2_reg<F, Tw<F>>(fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
            } else if constexpr (std::is_same<F, Goldilocks>::value) {
                rc = run_pow2_reg<F, TwGoldi>(fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
            } else {
                rc = run_pow2_reg<F, Tw<F>>(fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
            }
            if (rc) return rc;
            done = true;
        } else if (lg <= 29) {
            if constexpr (std::is_same<F, Prime32>::value) {
                // (no prime below 2^24 has 2^21 | p - 1, so the unreduced 24-bit butterflies never apply here)
                if (fd.p < (1ull << 30)) rc = run_pow2_reg3<F, TwShoupLazy>(fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
                else if (fd.p < (1ull << 31)) rc = run_pow2_reg3<F, TwShoup32>(fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
                else rc = run_pow2_reg3<F, Tw<F>>(fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
            } else if constexpr (std::is_same<F, Goldilocks>::value) {
                rc = run_pow2_reg3<F, TwGoldi>(fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
            } else {
                rc = run_pow2_reg3<F, Tw<F>>(fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
            }
            if (rc && rc != GFA_ERR_UNSUPPORTED) return rc;
            done = rc == GFA_OK;
        }
        if (!done && lg > 2 * REG_MAX_LOG && lg <= 24) {
            if constexpr (std::is_same<F, Prime32>::value) {
                if (fd.p < (1ull << 31)) rc = run_pow2<F, TwShoup32>(f, fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
                else rc = run_pow2<F, Tw<F>>(f, fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
            } else {
                rc = run_pow2<F, Tw<F>>(f, fd, pl, ein, eout, n, batch, omega, do_scale, scale, st);
            }
            if (rc) return rc;
            done = true;
        }
    }
    if (!done && (rc = run_generic<F>(fd, pl, ein, eout, n, batch, omega, do_scale, scale, st))) return rc;
    if (dtype != native) {
        const int grid = (int)std::min<i64>((n * batch + 255) / 256, 256 * 16);
        hipLaunchKernelGGL((convert_out_kernel<E>), dim3(grid), dim3(256), 0, st, (const E *)pl->sc->cvt.p, out, dtype, n * batch);
        GFA_HIP(hipGetLastError());
    }
    return GFA_OK;
}

} // namespace

namespace gfa {
void ntt_forget_field(const gfa_field *f)
{
    std::lock_guard<std::mutex> lock(g_plan_mu);
    for (auto it = g_plans.begin(); it != g_plans.end();) {
        if (it->first.f == f) {
            Plan *pl = it->second;
            {
                std::lock_guard<std::mutex> gl_lock(g_glperm_mu);
                for (const void *p : {(const void *)pl->wl0, (const void *)pl->wl1, (const void *)pl->wl2}) g_glperm.erase(p);
            }
            for (void *p : {pl->w1, pl->w1q, pl->w2, pl->w2q, pl->powA, pl->powB, pl->powAq, pl->powBq, pl->powAm, pl->wl1, pl->wl1q, pl->wl2,
                            pl->wl2q, pl->wpow, pl->wl0, pl->wl0q, pl->powA2, pl->powB2, pl->powA2q,
                            pl->powB2q, pl->powA2m, pl->gl_ptab2})
                if (p) (void)hipFree(p);
            delete pl;
            it = g_plans.erase(it);
        } else {
            ++it;
        }
    }
}
} // namespace gfa

extern "C" {

int gfa_ntt(gfa_field_t *f, const void *in, void *out, int64_t n, int64_t batch, uint64_t omega, int scale_by_n_inverse,
            int dtype, gfa_stream_t stream)
{
    if (!f || !in || !out || n < 1 || batch < 0 || dtype < GFA_U8 || dtype > GFA_U64) {
        set_error("gfa_ntt: bad arguments");
        return GFA_ERR_INVALID;
    }
    const FieldDev &c = f->calc;
    if ((c.q - 1) % (u64)n != 0) { set_error("gfa_ntt: n must divide q - 1"); return GFA_ERR_INVALID; }
    if (omega == 0 || omega >= c.q) { set_error("gfa_ntt: omega out of range"); return GFA_ERR_INVALID; }
    if (batch == 0) return GFA_OK;
    hipStream_t st = (hipStream_t)stream;
    FieldDeviceState *ds;
    int dev = 0;
    int rc = f->ensure_device(&dev, &ds);
    if (rc) return rc;
    u64 scale = 1;
    if (scale_by_n_inverse) {
        // y /= GF(n % characteristic) (_function.py:209-210)
        u64 nm = (u64)n % c.p;
        if (!HostArith::inv(c, nm, &scale)) { set_error("gfa_ntt: n is not invertible in the field"); return GFA_ERR_INVALID; }
    }
    const bool lookup = f->use_lookup();
    Plan *pl;
    {
        std::lock_guard<std::mutex> lock(g_plan_mu);
        pl = lookup_plan(PlanKey{f, dev, n, omega, lookup ? 1 : 0, 0}, st);
    }
    if (lookup) return run_typed<Lut>(f, f->lut_desc(*ds), pl, in, out, n, batch, omega, scale_by_n_inverse, scale, dtype, st);
    switch (c.kind) {
    case KIND_PRIME32: return run_typed<Prime32>(f, c, pl, in, out, n, batch, omega, scale_by_n_inverse, scale, dtype, st);
    case KIND_PRIME64: return run_typed<Prime64>(f, c, pl, in, out, n, batch, omega, scale_by_n_inverse, scale, dtype, st);
    case KIND_GOLDILOCKS: return run_typed<Goldilocks>(f, c, pl, in, out, n, batch, omega, scale_by_n_inverse, scale, dtype, st);
    case KIND_BIN: return run_typed<Bin>(f, c, pl, in, out, n, batch, omega, scale_by_n_inverse, scale, dtype, st);
    case KIND_EXT: return run_typed<Ext>(f, c, pl, in, out, n, batch, omega, scale_by_n_inverse, scale, dtype, st);
    default: set_error("gfa_ntt: unsupported field kind"); return GFA_ERR_UNSUPPORTED;
    }
}

static int ntt_columns_impl(gfa_field_t *f, const void *in, void *out, int64_t n1, int64_t cols, int64_t col0, int64_t n_total,
                            uint64_t omega, int dtype, gfa_stream_t stream, bool inverse_form, int scale_by_n_total_inverse,
                            int64_t in_pitch = 0, int64_t out_pitch = 0)
{
    // pitches: elements between consecutive rows of the local array (0 = cols: a contiguous (n1 x cols) array); a sub-block of
    // the columns of a wider array is transformed in place with in_pitch = the wider array's row length
    if (in_pitch == 0) in_pitch = cols;
    if (out_pitch == 0) out_pitch = cols;
    if (in_pitch < cols || out_pitch < cols) { set_error("gfa_ntt_columns: a pitch below the number of columns"); return GFA_ERR_INVALID; }
    if (!f || !in || !out || n1 < 2 || cols < 1 || col0 < 0 || n_total < n1 || (n_total % n1) != 0 || !is_pow2(n1) ||
        !is_pow2(n_total) || !is_pow2(cols)) {
        set_error("gfa_ntt_columns: bad arguments (power-of-two n1, cols, n_total required)");
        return GFA_ERR_INVALID;
    }
    const FieldDev &c = f->calc;
    if ((c.q - 1) % (u64)n_total != 0) { set_error("gfa_ntt_columns: n_total must divide q - 1"); return GFA_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    FieldDeviceState *ds;
    int dev = 0;
    int rc = f->ensure_device(&dev, &ds);
    if (rc) return rc;
    Plan *pl;
    {
        std::lock_guard<std::mutex> lock(g_plan_mu);
        pl = lookup_plan(PlanKey{f, dev, n_total, omega, 0, n1}, st);
    }
    auto run = [&](auto Ftag, auto TWtag) -> int {
        typedef decltype(Ftag) F;
        typedef decltype(TWtag) TW;
        typedef typename F::elem E;
        if ((sizeof(E) == 4 ? GFA_U32 : GFA_U64) != dtype) {
            set_error("gfa_ntt_columns: dtype must be the field's native device width (uint32 for p < 2^32, else uint64)");
            return GFA_ERR_UNSUPPORTED;
        }
        int lg1 = 0, lgn = 0;
        while (((i64)1 << lg1) < n1) lg1++;
        while (((i64)1 << lgn) < n_total) lgn++;
        if (lg1 > 13 || (sizeof(E) << lg1) > 2 * TILE_LDS_BYTES - 4096) { set_error("gfa_ntt_columns: n1 too long for one LDS tile"); return GFA_ERR_UNSUPPORTED; }
        int rc2;
        if (lg1 >= 2 && lg1 <= REG_MAX_LOG) {
            if (!pl->reg_ready) {
                pl->log1 = lg1; pl->logn = lgn;
                if ((rc2 = build_line_tables<F, TW>(c, omega, n_total, lg1, &pl->wl1, &pl->wl1q, st))) return rc2;
                if ((rc2 = build_post_tables<F, TW>(c, omega, n_total, pl, st))) return rc2;
                GFA_HIP(hipStreamSynchronize(st)); // as above
                pl->reg_ready = true;
            }
            RegArgs ra{};
            ra.in_stride_c = 1; ra.in_stride_t = in_pitch; ra.out_stride_c = 1; ra.out_stride_t = out_pitch;
            ra.total_lines = cols;
            ra.post_twiddle = inverse_form ? 0 : 1; ra.pre_twiddle = inverse_form ? 1 : 0;
            ra.lo_bits = pl->lo_bits; ra.n_mask = (u64)n_total - 1; ra.line_offset = col0;
            ra.pinv = inverse_mod_2_32(c.p);
            if (scale_by_n_total_inverse) {
                u64 sc = 1;
                if (!HostArith::inv(c, (u64)n_total % c.p, &sc)) { set_error("gfa_ntt_columns_inv: n_total is not invertible in the field"); return GFA_ERR_INVALID; }
                ra.do_scale = 1; ra.scale = sc; ra.scale_q = shoup_quotient<TW>(c, sc);
            }
            return launch_reg<F, TW>(c, lg1, in, out, ra, 1, pl->wl1, pl->wl1q, pl->powA, pl->powAq, pl->powB, pl->powBq, pl->powAm, st);
        }
        if (inverse_form) { set_error("gfa_ntt_columns_inv: n1 must be at most 2^10"); return GFA_ERR_UNSUPPORTED; }
        if (!pl->w1) {
            pl->log1 = lg1; pl->logn = lgn;
            if ((rc2 = build_pow_table<F>(c, omega, (u64)(n_total / n1), n1 / 2, &pl->w1, st))) return rc2;
            if (TW::HAS_SHOUP && (rc2 = build_shoup(c, pl->w1, n1 / 2, &pl->w1q, st, qbits_of<TW>()))) return rc2;
            pl->lo_bits = (lgn + 1) / 2;
            const i64 nb = (i64)1 << pl->lo_bits, na = n_total >> pl->lo_bits;
            if ((rc2 = build_pow_table<F>(c, omega, (u64)nb, na, &pl->powA, st))) return rc2;
            if ((rc2 = build_pow_table<F>(c, omega, 1, nb, &pl->powB, st))) return rc2;
        }
        TileArgs ta{};
        ta.logL = lg1;
        ta.lines_per_tile = choose_lines(lg1, sizeof(E), cols);
        ta.tiles_per_batch = (int)(cols / ta.lines_per_tile);
        ta.in_stride_c = 1; ta.in_stride_t = in_pitch; ta.out_stride_c = 1; ta.out_stride_t = out_pitch;
        ta.post_twiddle = 1; ta.lo_bits = pl->lo_bits; ta.n_mask = (u64)n_total - 1; ta.line_offset = col0;
        ta.tw_in_lds = lg1 <= 12;
        if constexpr (std::is_same<TW, TwShoupLazy>::value || std::is_same<TW, TwShoupWide>::value) {
            set_error("gfa_ntt_columns: internal: lazy twiddles are only used by the register kernel");
            return GFA_ERR_UNSUPPORTED;
        } else {
            return launch_tile<F, TW>(c, false, false, in, out, ta, 1, pl->w1, pl->w1q, pl->powA, pl->powB, st);
        }
    };
    switch (c.kind) {
    case KIND_PRIME32:
        if (c.p < (1ull << 24) && n1 <= ((i64)1 << REG_MAX_LOG) ) return run(Prime32{}, TwShoupWide{});
        if (c.p < (1ull << 30) && n1 <= ((i64)1 << REG_MAX_LOG)) return run(Prime32{}, TwShoupLazy{});
        if (c.p < (1ull << 31)) return run(Prime32{}, TwShoup32{});
        return run(Prime32{}, Tw<Prime32>{});
    case KIND_PRIME64: return run(Prime64{}, Tw<Prime64>{});
    case KIND_GOLDILOCKS:
        if (n1 <= ((i64)1 << REG_MAX_LOG)) return run(Goldilocks{}, TwGoldi{});
        return run(Goldilocks{}, Tw<Goldilocks>{});
    default: set_error("gfa_ntt_columns: prime fields only"); return GFA_ERR_UNSUPPORTED;
    }
}

int gfa_ntt_chunked(gfa_field_t *f, const void *in, void *out, int64_t n, int64_t batch, uint64_t omega, int scale_by_n_inverse,
                    int64_t in_chunk_len, int64_t in_chunk_stride, int64_t in_row_stride, int64_t out_chunk_len,
                    int64_t out_chunk_stride, int64_t out_row_stride, int dtype, gfa_stream_t stream)
{
    if (!f || n < 4 || (n & (n - 1)) || n > ((int64_t)1 << (2 * REG_MAX_LOG)) || in == out) {
        set_error("gfa_ntt_chunked: power-of-two 4 <= n <= 2^20 and distinct buffers required");
        return GFA_ERR_UNSUPPORTED;
    }
    const FieldDev &c = f->calc;
    const bool prime = c.kind == KIND_PRIME32 || c.kind == KIND_PRIME64 || c.kind == KIND_GOLDILOCKS;
    if (!prime || f->use_lookup() || dtype != (c.kind == KIND_PRIME32 ? GFA_U32 : GFA_U64)) {
        set_error("gfa_ntt_chunked: prime fields in calculate mode, native device width only");
        return GFA_ERR_UNSUPPORTED;
    }
    if ((in_chunk_len && (in_chunk_len & (in_chunk_len - 1))) || (out_chunk_len && (out_chunk_len & (out_chunk_len - 1)))) {
        set_error("gfa_ntt_chunked: chunk lengths must be powers of two");
        return GFA_ERR_INVALID;
    }
    g_layout_in = RowLayout{in_chunk_len, in_chunk_stride, in_row_stride};
    g_layout_out = RowLayout{out_chunk_len, out_chunk_stride, out_row_stride};
    const int rc = gfa_ntt(f, in, out, n, batch, omega, scale_by_n_inverse, dtype, stream);
    g_layout_in = RowLayout{};
    g_layout_out = RowLayout{};
    return rc;
}

int gfa_ntt_columns(gfa_field_t *f, const void *in, void *out, int64_t n1, int64_t cols, int64_t col0, int64_t n_total,
                    uint64_t omega, int dtype, gfa_stream_t stream)
{
    return ntt_columns_impl(f, in, out, n1, cols, col0, n_total, omega, dtype, stream, false, 0);
}

int gfa_ntt_columns_pitched(gfa_field_t *f, const void *in, int64_t in_pitch, void *out, int64_t out_pitch, int64_t n1, int64_t cols,
                            int64_t col0, int64_t n_total, uint64_t omega, int dtype, gfa_stream_t stream)
{
    return ntt_columns_impl(f, in, out, n1, cols, col0, n_total, omega, dtype, stream, false, 0, in_pitch, out_pitch);
}

int gfa_ntt_columns_inv(gfa_field_t *f, const void *in, void *out, int64_t n1, int64_t cols, int64_t col0, int64_t n_total,
                        uint64_t omega, int scale_by_n_total_inverse, int dtype, gfa_stream_t stream)
{
    return ntt_columns_impl(f, in, out, n1, cols, col0, n_total, omega, dtype, stream, true, scale_by_n_total_inverse);
}

int gfa_time_ntt(gfa_field_t *f, const void *in, void *out, int64_t n, int64_t batch, uint64_t omega, int dtype,
                 gfa_stream_t stream, int iters, float *ms_out)
{
    return gfa::time_loop((hipStream_t)stream, iters, ms_out,
                          [&]() { return gfa_ntt(f, in, out, n, batch, omega, 0, dtype, stream); });
}

} // extern "C"
