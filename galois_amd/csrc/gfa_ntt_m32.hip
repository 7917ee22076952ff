// gfa_ntt_m32.hip -- power-of-two transforms over GF(p), odd p < 2^26, on SIGNED Montgomery representatives.
//
// Replaces fft_jit / ifft_jit (reference: src/galois/_domains/_function.py:246-392) for 32 <= n <= 2^20 points.  Exact field
// arithmetic: any correct DFT algorithm reproduces the reference's bits.  Same tiling as ntt_reg_kernel (gfa_ntt.hip) --
// lines of L = R1 * R2 <= 1024 points, R1 points of one line per thread, two in-register decimation-in-frequency networks
// joined by one LDS exchange, n > 1024 as a four-step transform in two passes -- with the arithmetic and the addressing redone
// after the round-2 counters (56.5 vector instructions per point and pass, VALU-bound):
//
//   * values are int32 representatives of their residue class, never "reduced": a butterfly is v_add + v_sub with no bias
//     and no conditional correction.  A twiddle product is Montgomery's  (x*w - m*p) / 2^32  with m = x * (w * p^-1 mod
//     2^32): for ANY int32 x and |w| <= p/2 the result lies in (-p, p) -- v_mul_lo, v_mul_hi_i32, v_mul_hi_i32, v_sub.  A
//     radix-32 network grows its inputs by at most 2^5, and every value that leaves a network meets one product (the middle
//     twiddle, the inter-pass twiddle, or the final normalisation that carries the 1/n scale), so |v| < 32 p < 2^31 always;
//   * twiddles inside the networks are wave-uniform (scalar registers), the middle twiddle w_L^(r*k) comes from a
//     transposed LDS table addressed by immediates, the inter-pass twiddle w_n^(j2*k1) is a per-thread Montgomery
//     progression seeded from two small tables (an n-entry table was measured: slower, see ntt_m32_kernel);
//   * every global access is  buffer descriptor on the tile + one per-thread 32-bit offset + a scalar offset: no per-point
//     address arithmetic on the vector ALU.
// Result (profiles/r03_*): 33 vector instructions per point and pass (56.5), 2^20 x 64 over GF(7340033) 0.283 -> 0.236 ms.
// Both passes now sit on the memory system: the same access pattern with NO arithmetic (tools/ubench/ntt_access.hip)
// takes 0.20-0.22 ms, and neither sub-batches that keep the intermediate in the Infinity Cache nor an XCD-fused
// single-launch form (tools/ubench/ntt_fused_skel.hip) beat two plain passes -- DESIGN.md section 4.3.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "gfa_internal.h"

using namespace gfa;

namespace {

typedef int i32;

struct M32Args {
    i64 in_stride_c, in_stride_t;   // elements: element (line c, position t) at c * stride_c + t * stride_t
    i64 out_stride_c, out_stride_t;
    i64 in_batch_stride, out_batch_stride;
    i64 total_lines; // lines per batch item
    int tiles_per_batch;
    int load_along_line, store_along_line; // which index runs fastest across lanes (the contiguous one in memory)
    int tile_order;                        // 0: identity; 1: XCD x owns a contiguous range of tiles of every batch item
    i32 p;
    u32 pinv;     // p^-1 mod 2^32
    i32 fin, finp; // last pass: outputs * fin (Montgomery form of 1 or of 1/n), finp = fin * pinv
    i32 one, onep; // Montgomery form of 1 and its companion: the in-network reductions of primes >= 2^26 (DifSched)
    // MODE 1 with a SPLIT progression table (first pass of a three-pass transform: 2^11 .. 2^18 columns): line = lh * 2^tw_bits + ll,
    // t_0 = tw[ll * R1 + ka] * twh[lh * R1 + ka], ratio = tw2[ll] * tw2h[lh]; tw_bits = 0: tw / tw2 hold every line
    const i32 *twh, *tw2h;
    int tw_bits;
};

#include "gfa_m32_net.h" // m32_mulm / DifSched / dif: shared with the host check of the networks (tests/csrc/m32_net_host_test.cpp)

__device__ __forceinline__ i32 mulhi_vs(i32 x, i32 s) { return m32_mulhi_vs(x, s); }
__device__ __forceinline__ i32 mulhi_vv(i32 x, i32 y)
{
    i32 r;
    asm("v_mul_hi_i32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

__device__ __forceinline__ i32 mulm(i32 x, i32 wm, i32 wp, i32 p) { return m32_mulm(x, wm, wp, p); }
// per-lane twiddle (wm, wp in vector registers)
__device__ __forceinline__ i32 mulm_v(i32 x, i32 wm, i32 wp, i32 p)
{
    const i32 m = (i32)((u32)x * (u32)wp);
    return mulhi_vv(x, wm) - mulhi_vs(m, p);
}

// the same product when only wm is at hand (table twiddles, 4 bytes per entry): t = x * wm as one 64-bit multiply-add,
// m = lo(t) * p^-1, result = hi(t + m * (-p)).  (Inline assembly: clang expands the second signed 64-bit product into an
// unsigned multiply-add plus sign corrections.)
__device__ __forceinline__ i32 mulm1(i32 x, i32 wm, u32 pinv, i32 negp)
{
    long long t, r;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"(x), "v"(wm) : "vcc");
    const i32 m = (i32)((u32)t * pinv);
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(m), "s"(negp), "v"(t) : "vcc");
    return (i32)(r >> 32);
}

constexpr int brev_c(int x, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

template <int C>
constexpr int line_pitch(int rows, int row)
{ // words per staged line: rows * row rounded up so that the lanes of one LDS access fall into distinct banks
    // C >= 32: 32 lines per half wave -> odd pitch; C <= 16: pitch = 4 (mod 32) separates 8..16 lines x 4..2 positions
    const int base = rows * row;
    const int want = C >= 32 ? 1 : 4;
    return base + ((want - base) % 32 + 32) % 32;
}

// MODE 0: a whole transform, or the last pass of a four-step one (outputs * fin, canonical).
// MODE 1: first pass of a four-step transform: output k1 = ka + R1 * kr of column j2 times w_n^(j2 * k1), formed per thread as
//         the geometric progression t_0 = tw[j2 * R1 + ka] = w_n^(j2 * ka), ratio tw2[j2] = w_n^(j2 * R1).
// Measured alternatives for that twiddle (profiles/r03_m32_sweep.txt, 2^20 x 64): an n-entry table read with the store's own
// offsets 0.257-0.264 ms, the same table on the load side of pass 2 (contiguous rows) 0.243-0.252 ms, the progression
// 0.236-0.247 ms -- the table costs more in the memory system (which bounds both passes) than two products per point cost
// on the vector ALU (which has slack since the arithmetic went from 56 to 33 instructions per point and pass).
// NT (last pass of a two-pass transform only): non-temporal loads and stores.  Measured (profiles/r04_m32_aux.txt): when the batch's
// intermediate fits the 256 MiB Infinity Cache the last pass then finds it there -- 2^20 x 64: 0.230 -> 0.217 ms -- while on larger batches
// the hint costs 4 % (2^20 x 256), so the host sets it by size.
#ifndef GFA_M32_WAVES
#define GFA_M32_WAVES 4 // waves per SIMD the register allocation is held to (128 VGPRs): two 512-thread workgroups per CU
#endif
template <int LOGR1, int LOGR2, int THREADS, bool SPLIT, int MODE, bool NT, int BMAX>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(GFA_M32_WAVES))) void ntt_m32_kernel(const i32 *__restrict__ in, i32 *__restrict__ out, M32Args a,
                                                          const i32 *__restrict__ net1, const i32 *__restrict__ net2,
                                                          const i32 *__restrict__ mid, const i32 *__restrict__ tw,
                                                          const i32 *__restrict__ tw2)
{
    constexpr int R1 = 1 << LOGR1, R2 = 1 << LOGR2, L = R1 * R2;
    constexpr int C = THREADS / R1; // lines per tile
    constexpr int LOGC = __builtin_ctz(C);
    constexpr int ROW = R2 + 1;
    constexpr int RROWS = SPLIT ? R1 / 2 : R1; // rows of a line resident in LDS at a time
    constexpr int PC = line_pitch<C>(RROWS, ROW);
    static_assert(C * R2 <= THREADS, "step A needs C * R2 threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    i32 *data = reinterpret_cast<i32 *>(smem_raw);   // C * PC
    i32 *midl = data + ((C * PC + 1) & ~1);           // [ka][r] pairs (w_L^(r*ka) in Montgomery form, companion)

    const int tid = threadIdx.x;
    u32 vb = blockIdx.x;
    if (a.tile_order == 2) { // XCD x owns a contiguous eighth of ALL tiles (whole batch items when there are >= 8 of them)
        vb = (vb & 7u) * (gridDim.x >> 3) + (vb >> 3);
    } else if (a.tile_order == 1) {
        // workgroup b runs on XCD b % 8 (own L2).  XCD x gets tiles [x * per, (x + 1) * per) of every batch item: the two
        // halves of a 128-byte line of a strided pass meet in one L2 (identity order: 0.31 instead of 0.25 ms).
        const u32 x = vb & 7u, i = vb >> 3, per = (u32)a.tiles_per_batch >> 3;
        vb = (i / per) * (u32)a.tiles_per_batch + x * per + (i % per);
    }
    const u32 batch = vb / (u32)a.tiles_per_batch;
    const u32 tile = vb % (u32)a.tiles_per_batch;
    const i64 line0 = (i64)tile * C;
    const i32 *gin = in + (i64)batch * a.in_batch_stride + line0 * a.in_stride_c;
    i32 *gout = out + (i64)batch * a.out_batch_stride + line0 * a.out_stride_c;
    const i32 p = a.p;

    // middle twiddles, transposed: midl[2 * (ka * R2 + r)] -- staged while the first loads are in flight
    for (int i = tid; i < L; i += THREADS) {
        const int ka = i >> LOGR2, r = i & (R2 - 1);
        const int2 wv = reinterpret_cast<const int2 *>(mid)[r * ka];
        reinterpret_cast<int2 *>(midl)[i] = wv;
    }

    const bool active_a = (C * R2 == THREADS) || tid < C * R2;
    int ca, ra_;
    if (a.load_along_line) { ca = tid >> LOGR2; ra_ = tid & (R2 - 1); }
    else { ra_ = tid >> LOGC; ca = tid & (C - 1); }
    int c, ka;
    if (a.store_along_line) { c = tid >> LOGR1; ka = tid & (R1 - 1); }
    else { ka = tid >> LOGC; c = tid & (C - 1); }

    i32 v[R2];
    {
        i32 va[R1];
        if (C * R2 != THREADS) {
#pragma unroll
            for (int k = 0; k < R1; k++) asm volatile("" : "=v"(va[k]));
        }
        if (active_a) {
            const i64 last = a.total_lines - 1 - line0; // lines past the end are clamped (computed, never stored)
            const u32 cl = (u32)((i64)ca <= last ? ca : last);
            // buffer addressing: descriptor on the tile base, one per-thread byte offset in a VGPR, the position's byte
            // offset in the scalar operand -- no vector-ALU address arithmetic per load (the host checks the 32-bit range)
            const u32 off = (cl * (u32)a.in_stride_c + (u32)ra_ * (u32)a.in_stride_t) * 4u;
            const u32 step = (u32)R2 * (u32)a.in_stride_t * 4u; // uniform
            const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)gin, 0, 0xffffffffu, 0x00020000);
#pragma unroll
            for (int k = 0; k < R1; k++) va[k] = __builtin_amdgcn_raw_buffer_load_b32(rin, (int)off, (int)(k * step), NT ? 2 : 0);
            dif<LOGR1, BMAX>(va, net1, p, a.one, a.onep);
        }
        __syncthreads(); // middle-twiddle table staged
#pragma unroll
        for (int h = 0; h < (SPLIT ? 2 : 1); h++) {
            if (active_a) {
                i32 *dst = data + ca * PC + ra_;
                const int2 *mrow = reinterpret_cast<const int2 *>(midl) + ra_;
#pragma unroll
                for (int kl = 0; kl < RROWS; kl++) {
                    const int kaa = h * RROWS + kl;
                    const int2 wv = mrow[kaa * R2];
                    dst[kl * ROW] = mulm_v(va[brev_c(kaa, LOGR1)], wv.x, wv.y, p);
                }
            }
            __syncthreads();
            if (!SPLIT || (ka / RROWS) == h) {
                const i32 *srcl = data + c * PC + (ka - h * RROWS) * ROW;
#pragma unroll
                for (int r = 0; r < R2; r++) v[r] = srcl[r];
            }
            if (SPLIT && h == 0) __syncthreads();
        }
    }
    dif<LOGR2, BMAX>(v, net2, p, a.one, a.onep);
    const u32 ooff = ((u32)c * (u32)a.out_stride_c + (u32)ka * (u32)a.out_stride_t) * 4u;
    const u32 ostep = (u32)R1 * (u32)a.out_stride_t * 4u; // uniform
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)gout, 0, 0xffffffffu, 0x00020000);
    const bool live = line0 + c < a.total_lines;
    if (MODE == 1) {
        // output k1 = ka + R1 * kr of line j2 times w_n^(j2 * k1) = t_0 * ratio^kr, t_0 = w_n^(j2 * ka), ratio = w_n^(j2 * R1)
        const u32 pinv = a.pinv;
        const i32 negp = -p;
        if (live) {
            const u32 line = (u32)line0 + (u32)c;
            i32 t, sr;
            if (a.tw_bits) { // wave-uniform branch: one more product per thread, not per point
                const u32 ll = line & ((1u << a.tw_bits) - 1u), lh = line >> a.tw_bits;
                t = mulm1(tw[ll * (u32)R1 + (u32)ka], a.twh[lh * (u32)R1 + (u32)ka], pinv, negp);
                sr = mulm1(tw2[ll], a.tw2h[lh], pinv, negp);
            } else {
                t = tw[line * (u32)R1 + (u32)ka];
                sr = tw2[line];
            }
#pragma unroll
            for (int kr = 0; kr < R2; kr++) {
                __builtin_amdgcn_raw_buffer_store_b32(mulm1(v[brev_c(kr, LOGR2)], t, pinv, negp), rout, (int)ooff, (int)(kr * ostep), 0);
                if (kr + 1 < R2) t = mulm1(t, sr, pinv, negp);
            }
        }
    } else {
        const i32 fin = a.fin, finp = a.finp;
        if (live) {
#pragma unroll
            for (int kr = 0; kr < R2; kr++) {
                i32 x = mulm(v[brev_c(kr, LOGR2)], fin, finp, p); // (-p, p)
                x += p & (x >> 31);                                // [0, p)
                __builtin_amdgcn_raw_buffer_store_b32(x, rout, (int)ooff, (int)(kr * ostep), NT ? 2 : 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2^11 .. 2^15 points in ONE pass over HBM: one workgroup owns one whole transform
// ------------------------------------------------------------------------------------------------
// n = R0 * 1024, R0 = 2 .. 32, T = n / 32 = 32 * R0 threads, 32 points per thread throughout, three register networks and two
// exchanges through LDS -- the array is read once and written once (8 B/point; the two-pass form moves 16):
//   phase 1: thread t holds, for its P = 32 / R0 positions j = t + T*i of the 1024-point sub-lines, the R0 points x[j + 1024*a0];
//            radix-R0 network over a0, times w_n^(j*k0) (a per-position Montgomery progression), to LDS at [k0][j];
//   phase 2: thread (k0, r) reads sub-line k0 at positions r + 32*a: radix-32 network over a, times w_1024^(r*ka), to LDS at [k0][ka][r];
//   phase 3: thread (ka, k0) reads its row r = 0..31: radix-32 network over r; output kr goes to X[k0 + R0*(ka + 32*kr)] = out[t + T*kr].
// Loads and stores are fully coalesced (consecutive threads, consecutive words) in every phase, LDS accesses conflict-free
// (sub-line pitch = 32*33 + pad with pitch = 32/R0 mod 32).
struct M32OneArgs {
    i32 p;
    u32 pinv;
    i32 one, onep; // Montgomery form of 1 and its companion (normalises the untwiddled k0 = 0 outputs of phase 1)
    i32 fin, finp; // last product: Montgomery form of 1 or of 1/n
    const int *never; // always nullptr: `if (a.never) s_nop` is a never-taken branch that splits the loop body's basic block (M32_SPLIT)
};
// One straight-line loop body lets the scheduler stretch live ranges across phases until the allocator spills; a phase boundary the
// scheduler cannot cross keeps the early requests of the next transform within the 128-register budget (r06, gfa_ntt_fermat.hip)
#ifndef GFA_M32_SPLIT_MODE
#define GFA_M32_SPLIT_MODE 0 // 0: scheduling barrier only; 1: never-taken branch; 2: both.  Measured per kernel: the Montgomery 2^16 kernel
                             // stays spill-free with the barrier alone and spills 112-268 bytes with the branch; the GF(65537) kernel is the other way round
#endif
#define M32_SPLIT()                                                                      \
    do {                                                                                 \
        if (GFA_M32_SPLIT_MODE != 1) __builtin_amdgcn_sched_barrier(0);                  \
        if (GFA_M32_SPLIT_MODE != 0 && a.never != nullptr) asm volatile("s_nop 0");      \
    } while (0)

template <int LOGR0>
constexpr int one_pitch()
{
    constexpr int base = 32 * 33, want = 32 >> LOGR0;
    return base + ((want - base) % 32 + 32) % 32;
}

// G transforms per workgroup (G * T threads, one staged middle-twiddle table for all of them): 2^11- and 2^12-point transforms would
// otherwise run in 64- and 128-thread workgroups whose 8 KiB table limits the CU to 9-12 waves.
// LDS operations of a wave complete in order, so lgkmcnt(0) + s_barrier is all an exchange needs; __syncthreads() would also wait
// for the next transform's loads that are in flight by design.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int LOGR0, int G, int BMAX>
__global__ __launch_bounds__(G * (32 << LOGR0)) void ntt_m32_one_kernel(const i32 *__restrict__ in, i32 *__restrict__ out, M32OneArgs a,
                                                                       const i32 *__restrict__ net0, const i32 *__restrict__ net1,
                                                                       const i32 *__restrict__ mid, const i32 *__restrict__ wj, i64 batch)
{
    constexpr int R0 = 1 << LOGR0, T = 32 * R0, P = 32 / R0, PITCH = one_pitch<LOGR0>(), THREADS = G * T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    i32 *midl = reinterpret_cast<i32 *>(smem_raw);                              // [ka][r] pairs of w_1024^(r*ka), shared by the G transforms
    const int g = (int)threadIdx.x / T, tid = (int)threadIdx.x % T;
    i32 *data = midl + 2 * 1024 + g * (R0 * PITCH);                             // R0 * PITCH words per transform, both exchanges
    const i64 n = (i64)1024 * R0;
    const i64 nblk = (batch + G - 1) / G;
    const i32 p = a.p, negp = -a.p;
    const u32 pinv = a.pinv;
    // persistent workgroups: block index blk, blk + gridDim.x, ...; a partly filled last block repeats the last transform
    auto row_of = [&](i64 blk) -> i64 { const i64 w = blk * G + g; return w < batch ? w : batch - 1; };
    i32 va[P][R0];
    auto load_inputs = [&](i64 blk) {
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + row_of(blk) * n), 0, (u32)(n * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < P; i++)
#pragma unroll
            for (int a0 = 0; a0 < R0; a0++) va[i][a0] = __builtin_amdgcn_raw_buffer_load_b32(rin, tid * 4, (i * T + a0 * 1024) * 4, 0);
    };
    i64 blk = blockIdx.x;
    load_inputs(blk);
    for (int i = (int)threadIdx.x; i < 1024; i += THREADS) {
        const int ka = i >> 5, r = i & 31;
        reinterpret_cast<int2 *>(midl)[i] = reinterpret_cast<const int2 *>(mid)[r * ka];
    }
    for (; blk < nblk; blk += gridDim.x) {
        const bool live = blk * G + g < batch;
        // ---- phase 1 ----
#pragma unroll
        for (int i = 0; i < P; i++) {
            dif<LOGR0, BMAX>(va[i], net0, p, a.one, a.onep);
            const i32 ratio = wj[tid + i * T]; // w_n^j in Montgomery form
            i32 t = ratio;
            i32 *dst = data + tid + i * T;
            dst[0] = mulm(va[i][0], a.one, a.onep, p);
#pragma unroll
            for (int k0 = 1; k0 < R0; k0++) {
                dst[k0 * PITCH] = mulm1(va[i][brev_c(k0, LOGR0)], t, pinv, negp);
                if (k0 + 1 < R0) t = mulm1(t, ratio, pinv, negp);
            }
        }
        lds_barrier();
        // the registers of va are free: the next transform's loads travel while this one runs its last two phases
        if (blk + gridDim.x < nblk) load_inputs(blk + gridDim.x);
        // ---- phase 2 ----
        i32 v[32];
        {
            const int k0 = tid >> 5, r = tid & 31;
            const i32 *src = data + k0 * PITCH + r;
#pragma unroll
            for (int x = 0; x < 32; x++) v[x] = src[32 * x];
            dif<5, BMAX>(v, net1, p, a.one, a.onep);
            lds_barrier(); // every thread has read its sub-line: the buffer can take the second layout
            i32 *dst = data + k0 * PITCH + r;
            const int2 *mrow = reinterpret_cast<const int2 *>(midl) + r;
#pragma unroll
            for (int ka = 0; ka < 32; ka++) {
                const int2 wv = mrow[ka * 32];
                dst[ka * 33] = mulm_v(v[brev_c(ka, 5)], wv.x, wv.y, p);
            }
        }
        lds_barrier();
        // ---- phase 3 ----
        {
            const int k0 = tid & (R0 - 1), ka = tid >> LOGR0;
            const i32 *src = data + k0 * PITCH + ka * 33;
#pragma unroll
            for (int r = 0; r < 32; r++) v[r] = src[r];
            lds_barrier(); // the rows are in registers: the next iteration's phase 1 may overwrite the buffer
            dif<5, BMAX>(v, net1, p, a.one, a.onep);
            const i32 fin = a.fin, finp = a.finp;
            const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + row_of(blk) * n), 0, (u32)(n * 4), 0x00020000);
#pragma unroll
            for (int kr = 0; kr < 32; kr++) {
                i32 x = mulm(v[brev_c(kr, 5)], fin, finp, p);
                x += p & (x >> 31);
                if (live) __builtin_amdgcn_raw_buffer_store_b32(x, rout, tid * 4, kr * T * 4, 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2^16 points in ONE pass over HBM: one 1024-thread workgroup owns one whole transform, 64 points per thread
// ------------------------------------------------------------------------------------------------
// The shape of the GF(65537) kernel (gfa_ntt_fermat.hip) with Montgomery arithmetic: N = 64 * 32 * 32,
//   n = 1024 a + m, m = 32 b + r,  k = k0 + 64 (k1 + 32 k2):   n k = 1024 a k0 + m k0 + 2048 b k1 + 64 r k1 + 2048 r k2  (mod 2^16)
//   network 0: thread m = tid, radix 64 over a (stride 1024: every load a full 256-byte wave access), times w^(m k0) -- a per-thread
//              Montgomery progression seeded with w^m (as in the 2^11..2^15 kernel: a 256 KiB table costs more in the memory system);
//   exchange 1 (two rounds of 128 KiB): thread (g, r) collects k0 = g and g + 32: radix 32 over b, times w^(64 r k1) from LDS;
//   exchange 2 (two rounds): thread (k0 = lane, wv) collects k1 = wv and wv + 16: radix 32 over r; X[k0 + 64 (k1 + 32 k2)] is word
//              tid + 1024 h + 2048 k2 of the output row.
// The radix-64 network grows its inputs by 2^6 before the first product: p < 2^25.  Workgroups are persistent (one per CU); the next
// transform's loads are issued as registers come free.  8 B/point of HBM traffic (the two-pass form moves 16).
constexpr int B16_E2_PITCH = 33;
constexpr int B16_EX_WORDS = 16 * 64 * B16_E2_PITCH; // 33792 words >= exchange 1's 32 * 1024
constexpr size_t B16_LDS_BYTES = sizeof(i32) * (size_t)(B16_EX_WORDS + 2 * 1024);

// the next transform's rows requested ahead (r06, as in gfa_ntt_fermat.hip): [0, E1) once exchange 1 has taken the points, [E1, E2) before the
// second half of network 1, [E2, E3) after the first half's stores, the rest at the end
#ifndef GFA_M32_B16_E1
#define GFA_M32_B16_E1 16
#endif
#ifndef GFA_M32_B16_E2
#define GFA_M32_B16_E2 36 // 40 spills (profiles/r06_m32_2e16_early_loads.txt)
#endif
#ifndef GFA_M32_B16_E3
#define GFA_M32_B16_E3 (GFA_M32_B16_E2 > 32 ? GFA_M32_B16_E2 : 32)
#endif
template <int BMAX>
__global__ __launch_bounds__(1024) void ntt_m32_2e16_kernel(const i32 *in, i32 *out, M32OneArgs a, const i32 *__restrict__ net0,
                                                            const i32 *__restrict__ net1, const i32 *__restrict__ mid, const i32 *__restrict__ wj, i64 batch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    i32 *ex = reinterpret_cast<i32 *>(smem_raw);
    i32 *midl = ex + B16_EX_WORDS; // [k1][r] pairs of w^(64 r k1)
    const int tid = (int)threadIdx.x;
    const int voff = tid * 4;
    const int g = tid >> 5, r = tid & 31;  // exchange 1 / network 1 coordinates
    const int l = tid & 63, wv = tid >> 6; // exchange 2 / network 2 coordinates
    i32 *const e1w = ex + tid;
    const i32 *const e1r = ex + g * 1024 + r;
    i32 *const e2w = ex + g * B16_E2_PITCH + r;
    const i32 *const e2r = ex + wv * (64 * B16_E2_PITCH) + l * B16_E2_PITCH;
    const i32 p = a.p, negp = -a.p;
    const u32 pinv = a.pinv;
    {
        const int k1 = tid >> 5;
        reinterpret_cast<int2 *>(midl)[tid] = reinterpret_cast<const int2 *>(mid)[r * k1];
    }
    const i32 ratio = wj[tid]; // w^m in Montgomery form
    // `live` = false: a descriptor of zero records -- the last round's look-ahead requests return 0 and move nothing
    auto in_rsrc = [&](i64 t, bool live) { return __builtin_amdgcn_make_buffer_rsrc((void *)(in + t * 65536), 0, live ? 65536 * 4 : 0, 0x00020000); };
    i32 v[64];
    {
        const __amdgpu_buffer_rsrc_t xr = in_rsrc(blockIdx.x, true);
#pragma unroll
        for (int ap = 0; ap < 64; ap++) v[ap] = __builtin_amdgcn_raw_buffer_load_b32(xr, voff, ap * 4096, 0);
    }
    for (i64 tr_i = blockIdx.x; tr_i < batch; tr_i += gridDim.x) {
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)(out + tr_i * 65536), 0, 65536 * 4, 0x00020000);
        const bool has_next = tr_i + gridDim.x < batch;
        const __amdgpu_buffer_rsrc_t xn = in_rsrc(has_next ? tr_i + gridDim.x : tr_i, has_next);
        auto next_loads = [&](int lo, int hi) { // rows lo .. hi-1 of the next transform, into the point registers it starts from
#pragma unroll
            for (int ap = lo; ap < hi; ap++) v[ap] = __builtin_amdgcn_raw_buffer_load_b32(xn, voff, ap * 4096, 0);
        };
        // ---- network 0 and the first twiddle: Y[k0] * w^(m k0), t <- t * w^m ----
        dif<6, BMAX>(v, net0, p, a.one, a.onep);
        M32_SPLIT();
        v[0] = mulm(v[0], a.one, a.onep, p);
        {
            i32 t = ratio;
#pragma unroll
            for (int k0 = 1; k0 < 64; k0++) {
                // the application in 32-bit registers only (companion = one more multiply), the progression through the 64-bit
                // multiply-add: with BOTH as v_mad_i64_i32 the even-aligned register pairs beside 64 live points cost 52-108 bytes
                // of scratch per thread and 8-15 % of the kernel (profiles/r04_m32_2e16.txt)
                v[brev_c(k0, 6)] = mulm_v(v[brev_c(k0, 6)], t, (i32)((u32)t * pinv), p);
                if (k0 + 1 < 64) t = mulm1(t, ratio, pinv, negp);
            }
        }
        M32_SPLIT();
        // ---- exchange 1 + network 1 ----
        i32 w[2][32];
        auto net1f = [&](int h) {
            dif<5, BMAX>(w[h], net1, p, a.one, a.onep);
            const int2 *mrow = reinterpret_cast<const int2 *>(midl) + r;
            w[h][0] = mulm(w[h][0], a.one, a.onep, p);
#pragma unroll
            for (int k1 = 1; k1 < 32; k1++) {
                const int2 wt = mrow[k1 * 32];
                w[h][brev_c(k1, 5)] = mulm_v(w[h][brev_c(k1, 5)], wt.x, wt.y, p);
            }
        };
        lds_barrier(); // the previous transform's exchange-2 reads (first round: the staging of midl) are complete
#pragma unroll
        for (int kl = 0; kl < 32; kl++) e1w[kl * 1024] = v[brev_c(kl, 6)];
        lds_barrier();
#pragma unroll
        for (int bp = 0; bp < 32; bp++) w[0][bp] = e1r[bp * 32];
        lds_barrier();
#pragma unroll
        for (int kl = 0; kl < 32; kl++) e1w[kl * 1024] = v[brev_c(kl + 32, 6)];
        M32_SPLIT();
        next_loads(0, GFA_M32_B16_E1); // the point registers are free from here
        net1f(0);
        M32_SPLIT();
        lds_barrier();
#pragma unroll
        for (int bp = 0; bp < 32; bp++) w[1][bp] = e1r[bp * 32];
        M32_SPLIT();
        next_loads(GFA_M32_B16_E1, GFA_M32_B16_E2);
        net1f(1);
        M32_SPLIT();
        // ---- exchange 2 + network 2 ----
        i32 z[2][32];
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int kl = 0; kl < 16; kl++) e2w[kl * (64 * B16_E2_PITCH) + 32 * i * B16_E2_PITCH] = w[i][brev_c(kl, 5)];
        lds_barrier();
#pragma unroll
        for (int rp = 0; rp < 32; rp++) z[0][rp] = e2r[rp];
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int kl = 0; kl < 16; kl++) e2w[kl * (64 * B16_E2_PITCH) + 32 * i * B16_E2_PITCH] = w[i][brev_c(kl + 16, 5)];
        const i32 fin = a.fin, finp = a.finp;
        auto net2f = [&](int h) {
            dif<5, BMAX>(z[h], net1, p, a.one, a.onep);
#pragma unroll
            for (int k2 = 0; k2 < 32; k2++) {
                i32 x = mulm(z[h][brev_c(k2, 5)], fin, finp, p); // (-p, p)
                x += p & (x >> 31);                               // [0, p)
                __builtin_amdgcn_raw_buffer_store_b32(x, yr, voff, (2048 * k2 + 1024 * h) * 4, 0);
            }
        };
        M32_SPLIT();
        net2f(0);
        M32_SPLIT();
        next_loads(GFA_M32_B16_E2, GFA_M32_B16_E3);
        lds_barrier();
#pragma unroll
        for (int rp = 0; rp < 32; rp++) z[1][rp] = e2r[rp];
        M32_SPLIT();
        net2f(1);
        next_loads(GFA_M32_B16_E3, 64);
        M32_SPLIT();
    }
}

// ------------------------------------------------------------------------------------------------
// host side: tables and plans
// ------------------------------------------------------------------------------------------------
__global__ void m32_progression_table_kernel(u32 p, u32 omega, int logn, int log2, int logr1, i32 *t0, i32 *ratio)
{ // t0[j2 * R1 + ka] = omega^(j2 * ka), ratio[j2] = omega^(j2 * R1): both in Montgomery form, so that the per-thread progression
    // t <- t * ratio * 2^-32 stays in Montgomery form
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 n2 = (i64)1 << log2, R1 = (i64)1 << logr1;
    if (i >= n2 * (R1 + 1)) return;
    u64 e;
    if (i < n2 * R1) e = (u64)(i >> logr1) * (u64)(i & (R1 - 1));
    else e = (u64)(i - n2 * R1) * (u64)R1;
    e &= ((u64)1 << logn) - 1;
    u64 b = omega, r = 1;
    while (e) {
        if (e & 1) r = r * b % p;
        b = b * b % p;
        e >>= 1;
    }
    const u64 m = (r << 32) % p;
    const i32 c = m > p / 2 ? (i32)((i64)m - (i64)p) : (i32)m;
    if (i < n2 * R1) t0[i] = c;
    else ratio[i - n2 * R1] = c;
}

inline u64 powmod(u64 b, u64 e, u64 p)
{
    u64 r = 1;
    b %= p;
    while (e) {
        if (e & 1) r = r * b % p;
        b = b * b % p;
        e >>= 1;
    }
    return r;
}

inline u32 inv_2_32(u32 p)
{ // Newton iteration, p odd
    u32 x = p;
    for (int i = 0; i < 5; i++) x *= 2u - p * x;
    return x;
}

inline i32 mont_centred(u64 w, u64 p)
{
    const u64 m = (w << 32) % p;
    return m > p / 2 ? (i32)((i64)m - (i64)p) : (i32)m;
}

struct M32Plan {
    int log1 = 0, log2 = 0, log3 = 0;     // line lengths per pass (log3 != 0: three passes, n = 2^(log1 + log2 + log3))
    i32 *net1 = nullptr, *net2 = nullptr, *net3 = nullptr; // R/2 pairs each
    i32 *mid1 = nullptr, *mid2 = nullptr, *mid3 = nullptr; // L pairs: w_L^e, companion
    i32 *pt0 = nullptr, *pratio = nullptr; // progression form of the inter-pass twiddle: n2 * R1 and n2 entries (three passes: the SECOND pass)
    // three passes, first pass (2^(log2 + log3) columns): the progression seeds split in a low and a high table
    i32 *at0 = nullptr, *aratio = nullptr, *at0h = nullptr, *aratioh = nullptr;
    int a_bits = 0;
    // one-pass form (2^11 .. 2^15 points): radix-R0 network twiddles, radix-32 network twiddles, w_1024^e pairs, w_n^j (j < 1024)
    i32 *one_net0 = nullptr, *one_net1 = nullptr, *one_mid = nullptr, *one_wj = nullptr;
};

struct M32Key {
    u64 p; int device; i64 n; u64 omega; int split; // split: the tuning override of the three-pass line lengths (0: default)
    bool operator<(const M32Key &o) const { return std::tie(p, device, n, omega, split) < std::tie(o.p, o.device, o.n, o.omega, o.split); }
};

// tuning knobs of gfa_debug_m32_tune (tools/m32_tune3.py): [0] log1 and [1] log2 of the three-pass form (0: default),
// [2] non-temporal last pass: -1 by size (default), 0 never, 1 always; [3] last pass of 1024-point lines in 1024-thread workgroups
// (32 lines per tile: every transposed store covers a whole 128-byte line)
int g_m32_tune[4] = {0, 0, -1, 0};

std::mutex g_m32_mu;
std::map<M32Key, M32Plan *> g_m32_plans;

void free_plan(M32Plan *pl)
{
    for (void *q : {(void *)pl->net1, (void *)pl->net2, (void *)pl->net3, (void *)pl->mid1, (void *)pl->mid2, (void *)pl->mid3, (void *)pl->pt0,
                    (void *)pl->pratio, (void *)pl->at0, (void *)pl->aratio, (void *)pl->at0h, (void *)pl->aratioh, (void *)pl->one_net0,
                    (void *)pl->one_net1, (void *)pl->one_mid, (void *)pl->one_wj})
        if (q) (void)hipFree(q);
    delete pl;
}

int upload(const std::vector<i32> &h, i32 **d)
{
    GFA_HIP(hipMalloc((void **)d, sizeof(i32) * h.size()));
    GFA_HIP(hipMemcpy(*d, h.data(), sizeof(i32) * h.size(), hipMemcpyHostToDevice));
    return GFA_OK;
}

// pairs (Montgomery form of w^e, companion) for e < count, w = omega^mult
int pair_table(u64 p, u32 pinv, u64 omega, u64 mult, int count, i32 **d)
{
    std::vector<i32> h(2 * (size_t)count);
    const u64 w = powmod(omega, mult, p);
    u64 cur = 1;
    for (int e = 0; e < count; e++) {
        const i32 wm = mont_centred(cur, p);
        h[2 * e] = wm;
        h[2 * e + 1] = (i32)((u32)wm * pinv);
        cur = cur * w % p;
    }
    return upload(h, d);
}

constexpr int split_log1(int logL) { return (logL + 1) / 2; } // R1 >= R2

// progression tables of a MODE-1 pass with `lines` = 2^loglines lines of line length 2^logL inside a transform with root `w`
// of order 2^logw:  t0[j * R1 + ka] = w^(j * ka), ratio[j] = w^(j * R1)
int progression_tables(u64 p, u64 w, int logw, int loglines, int logL, i32 **t0, i32 **ratio, hipStream_t st)
{
    const int lr1 = split_log1(logL);
    const i64 lines = (i64)1 << loglines, cnt = lines * (((i64)1 << lr1) + 1);
    GFA_HIP(hipMalloc((void **)t0, sizeof(i32) * (size_t)(lines << lr1)));
    GFA_HIP(hipMalloc((void **)ratio, sizeof(i32) * (size_t)lines));
    hipLaunchKernelGGL(m32_progression_table_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, (u32)p, (u32)w, logw, loglines, lr1, *t0,
                       *ratio);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int build_plan(M32Plan *pl, u64 p, i64 n, u64 omega, int split, hipStream_t st)
{
    int logn = 0;
    while (((i64)1 << logn) < n) logn++;
    if (logn <= 10) { pl->log1 = logn; pl->log2 = 0; }
    else if (logn <= 20) { pl->log1 = (logn + 1) / 2; pl->log2 = logn - pl->log1; }
    else {
        // three passes (measured on the register kernels of gfa_ntt.hip, tools/ntt3_tune.py): longest lines in the widest-strided pass,
        // shortest in the last
        pl->log1 = (logn + 2) / 3;
        pl->log2 = (logn - pl->log1 + 1) / 2;
        // r05 sweep of every split (tools/m32_tune3.py, profiles/r05_m32_tune3.txt): 2^24 and 2^26 are best at these defaults; 2^22 points
        // run 23 % faster with 32-point lines in the first pass (tiles of 32 columns: whole 128-byte lines), 0.030 instead of 0.038 ms
        if (logn == 22) { pl->log1 = 5; pl->log2 = 9; }
        if (split) { pl->log1 = split >> 8; pl->log2 = split & 0xff; }
        pl->log3 = logn - pl->log1 - pl->log2;
        if (pl->log1 < 5 || pl->log1 > 10 || pl->log2 < 5 || pl->log2 > 10 || pl->log3 < 5 || pl->log3 > 10) {
            set_error("m32 NTT: three-pass line lengths out of range");
            return GFA_ERR_INVALID;
        }
    }
    const u32 pinv = inv_2_32((u32)p);
    int rc;
    auto line_tables = [&](int logL, i32 **net_a, i32 **mid) -> int {
        // networks of a line of L = R1 * R2 points: w_R1 = w_L^R2, w_R2 = w_L^R1; both tables in one allocation [R1/2 | R2/2]
        const int lr1 = split_log1(logL), lr2 = logL - lr1;
        const i64 Lh = (i64)1 << logL;
        const u64 wl_mult = (u64)(n / Lh);
        std::vector<i32> h;
        for (int part = 0; part < 2; part++) {
            const int lr = part ? lr2 : lr1;
            const int R = 1 << lr;
            const u64 w = powmod(omega, wl_mult * (u64)(Lh / R), p);
            u64 cur = 1;
            for (int e = 0; e < std::max(R / 2, 1); e++) {
                const i32 wm = mont_centred(cur, p);
                h.push_back(wm);
                h.push_back((i32)((u32)wm * pinv));
                cur = cur * w % p;
            }
        }
        if ((rc = upload(h, net_a))) return rc;
        return pair_table(p, pinv, omega, wl_mult, (int)Lh, mid);
    };
    if (logn >= 11 && logn <= 16) {
        const int R0 = (int)(n >> 10);
        auto pairs = [&](u64 w, int count, i32 **d) -> int { // Montgomery form of w^e and its companion, e < count
            std::vector<i32> h;
            u64 cur = 1;
            for (int e = 0; e < count; e++) {
                const i32 wm = mont_centred(cur, p);
                h.push_back(wm);
                h.push_back((i32)((u32)wm * pinv));
                cur = cur * w % p;
            }
            return upload(h, d);
        };
        if ((rc = pairs(powmod(omega, 1024, p), std::max(R0 / 2, 1), &pl->one_net0))) return rc;       // w_R0 = w_n^1024
        if ((rc = pairs(powmod(omega, (u64)R0 * 32, p), 16, &pl->one_net1))) return rc;                // w_32 = w_1024^32, w_1024 = w_n^R0
        if ((rc = pair_table(p, pinv, omega, (u64)R0, 1024, &pl->one_mid))) return rc;                 // w_1024^e
        std::vector<i32> hw(1024);
        u64 cur = 1;
        for (int j = 0; j < 1024; j++) { hw[j] = mont_centred(cur, p); cur = cur * omega % p; }          // w_n^j
        if ((rc = upload(hw, &pl->one_wj))) return rc;
    }
    if ((rc = line_tables(pl->log1, &pl->net1, &pl->mid1))) return rc;
    if (pl->log3) {
        if ((rc = line_tables(pl->log2, &pl->net2, &pl->mid2))) return rc;
        if ((rc = line_tables(pl->log3, &pl->net3, &pl->mid3))) return rc;
        // first pass: M = 2^(log2 + log3) columns j, twiddle w_n^(j * k1).  j = jh * 2^a_bits + jl: w^(j ka) = w^(jl ka) * (w^(2^a_bits))^(jh ka)
        const int logm = pl->log2 + pl->log3;
        pl->a_bits = logm / 2;
        if ((rc = progression_tables(p, omega, logn, pl->a_bits, pl->log1, &pl->at0, &pl->aratio, st))) return rc;
        if ((rc = progression_tables(p, powmod(omega, (u64)1 << pl->a_bits, p), logn, logm - pl->a_bits, pl->log1, &pl->at0h, &pl->aratioh, st)))
            return rc;
        // second pass, per row k1: the 2^log3 columns of an M-point transform with root w_M = w_n^(2^log1)
        if ((rc = progression_tables(p, powmod(omega, (u64)1 << pl->log1, p), logm, pl->log3, pl->log2, &pl->pt0, &pl->pratio, st))) return rc;
        GFA_HIP(hipStreamSynchronize(st)); // the plan may next be used from another stream
    } else if (pl->log2) {
        if ((rc = line_tables(pl->log2, &pl->net2, &pl->mid2))) return rc;
        if ((rc = progression_tables(p, omega, logn, pl->log2, pl->log1, &pl->pt0, &pl->pratio, st))) return rc;
        GFA_HIP(hipStreamSynchronize(st)); // the plan may next be used from another stream
    }
    return GFA_OK;
}

// BMAX = floor(2^31 / p) rounded down to the classes that are built: 32 (p < 2^26: no reduction inside a radix-32 network), 8, 4
template <int LOGR1, int LOGR2, int THREADS, bool SPLIT, int MODE, bool NT, int BMAX>
int launch_ttm(const i32 *in, i32 *out, M32Args a, i64 batch, const i32 *net, const i32 *mid, const i32 *tw, const i32 *tw2, hipStream_t st)
{
    constexpr int R1 = 1 << LOGR1, R2 = 1 << LOGR2, L = R1 * R2, C = THREADS / R1;
    constexpr int PC = line_pitch<C>(SPLIT ? R1 / 2 : R1, R2 + 1);
    constexpr size_t lds = sizeof(i32) * (size_t)(((C * PC + 1) & ~1) + 2 * L);
    a.tiles_per_batch = (int)((a.total_lines + C - 1) / C);
    const i64 grid = batch * a.tiles_per_batch;
    if (grid <= 0 || grid > 0x7fffffff) { set_error("m32 NTT: grid out of range"); return GFA_ERR_UNSUPPORTED; }
    {
        const i64 lim = (i64)1 << 30; // elements: byte offsets stay below 2^32
        if ((C - 1) * a.in_stride_c + (L - 1) * a.in_stride_t >= lim || (C - 1) * a.out_stride_c + (L - 1) * a.out_stride_t >= lim) {
            set_error("m32 NTT: tile extent exceeds the 32-bit offset range");
            return GFA_ERR_UNSUPPORTED;
        }
    }
    // tile order (measured, profiles/r03_m32_sweep.txt): column slices per XCD from 64 tiles per transform (2^20 x 64: identity order 0.31 ms,
    // slices 0.25), whole transforms per XCD below that (2^16 x 1024: 0.229 vs 0.265 ms)
    {
        const bool can1 = (a.tiles_per_batch % 8) == 0 && a.tiles_per_batch >= 64, can2 = (grid % 8) == 0 && grid >= 16;
        a.tile_order = can1 ? 1 : (can2 ? 2 : 0);
    }
    auto kern = ntt_m32_kernel<LOGR1, LOGR2, THREADS, SPLIT, MODE, NT, BMAX>;
    static bool attr = false;
    if (!attr) {
        GFA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), lds, st, in, out, a, net, net + R1, mid, tw, tw2);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

template <int LOGR1, int LOGR2, int THREADS, bool SPLIT, int BMAX>
int launch_tt(const i32 *in, i32 *out, const M32Args &a, i64 batch, const i32 *net, const i32 *mid, const i32 *tw, const i32 *tw2, hipStream_t st, int mode)
{
    if (mode == 0) return launch_ttm<LOGR1, LOGR2, THREADS, SPLIT, 0, false, BMAX>(in, out, a, batch, net, mid, tw, tw2, st);
    if (mode == 2) return launch_ttm<LOGR1, LOGR2, THREADS, SPLIT, 0, true, BMAX>(in, out, a, batch, net, mid, tw, tw2, st); // last pass, non-temporal
    return launch_ttm<LOGR1, LOGR2, THREADS, SPLIT, 1, false, BMAX>(in, out, a, batch, net, mid, tw, tw2, st);
}

template <int LOGR1, int LOGR2, int BMAX>
int launch_t(const i32 *in, i32 *out, const M32Args &a, i64 batch, const i32 *net, const i32 *mid, const i32 *tw, const i32 *tw2, hipStream_t st, int mode)
{
    // 1024-point lines: 512-thread workgroups, two per CU, the whole line staged (measured against 256 / 1024 threads and a two-round
    // exchange with three workgroups per CU: 0.236 vs 0.268 ms at 2^20 x 64, profiles/r03_m32_sweep.txt); shorter lines: 256 threads
    if constexpr (LOGR1 == 5 && LOGR2 == 5) {
        if (g_m32_tune[3] && mode != 1) { // tuning: 32 lines per tile in the last pass
            if (mode == 0) return launch_ttm<LOGR1, LOGR2, 1024, false, 0, false, BMAX>(in, out, a, batch, net, mid, tw, tw2, st);
            return launch_ttm<LOGR1, LOGR2, 1024, false, 0, true, BMAX>(in, out, a, batch, net, mid, tw, tw2, st);
        }
    }
    if constexpr (LOGR1 == 5) return launch_tt<LOGR1, LOGR2, 512, false, BMAX>(in, out, a, batch, net, mid, tw, tw2, st, mode);
    else return launch_tt<LOGR1, LOGR2, 256, false, BMAX>(in, out, a, batch, net, mid, tw, tw2, st, mode);
}

template <int BMAX>
int launch_b(int logL, const i32 *in, i32 *out, const M32Args &a, i64 batch, const i32 *net, const i32 *mid, const i32 *tw, const i32 *tw2, hipStream_t st, int mode)
{
    switch (logL) {
    case 5: return launch_t<3, 2, BMAX>(in, out, a, batch, net, mid, tw, tw2, st, mode);
    case 6: return launch_t<3, 3, BMAX>(in, out, a, batch, net, mid, tw, tw2, st, mode);
    case 7: return launch_t<4, 3, BMAX>(in, out, a, batch, net, mid, tw, tw2, st, mode);
    case 8: return launch_t<4, 4, BMAX>(in, out, a, batch, net, mid, tw, tw2, st, mode);
    case 9: return launch_t<5, 4, BMAX>(in, out, a, batch, net, mid, tw, tw2, st, mode);
    case 10: return launch_t<5, 5, BMAX>(in, out, a, batch, net, mid, tw, tw2, st, mode);
    default: set_error("m32 NTT: unsupported line length"); return GFA_ERR_UNSUPPORTED;
    }
}

// cls: 0 p < 2^26, 1 p < 2^28, 2 p < 2^29
int launch(int cls, int logL, const i32 *in, i32 *out, const M32Args &a, i64 batch, const i32 *net, const i32 *mid, const i32 *tw, const i32 *tw2, hipStream_t st, int mode)
{
    if (cls == 0) return launch_b<32>(logL, in, out, a, batch, net, mid, tw, tw2, st, mode);
    if (cls == 1) return launch_b<8>(logL, in, out, a, batch, net, mid, tw, tw2, st, mode);
    return launch_b<4>(logL, in, out, a, batch, net, mid, tw, tw2, st, mode);
}

template <int LOGR0, int BMAX>
int launch_one_t(const i32 *in, i32 *out, const M32OneArgs &oa, i64 batch, const M32Plan *pl, hipStream_t st)
{
    constexpr int R0 = 1 << LOGR0;
    // transforms per workgroup, measured (ms for 2^26 points; run-to-run spread about 5 %): 2^11 one per 64-thread workgroup 0.177, four per
    // 256 threads 0.162, eight per 512 threads 0.145; 2^12 one 0.155, two 0.161, four 0.148; 2^13 one per 256 threads 0.140-0.150, two 0.152
    constexpr int G = R0 <= 4 ? 16 / R0 : 1;
    constexpr size_t lds = sizeof(i32) * (size_t)(2 * 1024 + G * R0 * one_pitch<LOGR0>());
    auto kern = ntt_m32_one_kernel<LOGR0, G, BMAX>;
    static bool attr = false;
    if (!attr) {
        GFA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    // persistent: as many workgroups as fit the chip at once (LDS-limited), each looping over transforms with the next one's loads in flight
    static const int cus = [] { int dev = 0, n = 0; return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }();
    // measured: persistence + prefetch gains 14 % at 2^15 points (one workgroup per CU: nothing else overlaps its load phase) and
    // nothing or a loss below (2^14: 0.150 -> 0.163 ms), where two or three workgroups per CU overlap each other
    const bool persist = LOGR0 == 5;
    const i64 nblk = (batch + G - 1) / G;
    const i64 per_cu = std::max<i64>(1, (i64)(160 * 1024) / (i64)lds);
    const i64 grid = persist ? std::min<i64>(nblk, (i64)cus * per_cu) : nblk;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G * 32 * R0), lds, st, in, out, oa, pl->one_net0, pl->one_net1, pl->one_mid, pl->one_wj, batch);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

template <int BMAX>
int launch_2e16_b(const i32 *in, i32 *out, const M32OneArgs &oa, i64 batch, const M32Plan *pl, hipStream_t st)
{
    auto kern = ntt_m32_2e16_kernel<BMAX>;
    static bool attr = false;
    if (!attr) {
        GFA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    static const int cus = [] { int dev = 0, n = 0; return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }();
    const i64 grid = std::min<i64>(batch, cus); // persistent: one workgroup per CU (LDS-limited)
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(1024), B16_LDS_BYTES, st, in, out, oa, pl->one_net0, pl->one_net1, pl->one_mid, pl->one_wj, batch);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// the radix-64 network grows by 2^6: no reduction needed below 2^25 (BMAX 64); above, the schedule of the prime's class
int launch_2e16(u64 p, const i32 *in, i32 *out, const M32OneArgs &oa, i64 batch, const M32Plan *pl, hipStream_t st)
{
    if (p < (1ull << 25)) return launch_2e16_b<64>(in, out, oa, batch, pl, st);
    if (p < (1ull << 28)) return launch_2e16_b<8>(in, out, oa, batch, pl, st);
    return launch_2e16_b<4>(in, out, oa, batch, pl, st);
}

template <int BMAX>
int launch_one_b(int logr0, const i32 *in, i32 *out, const M32OneArgs &oa, i64 batch, const M32Plan *pl, hipStream_t st)
{
    switch (logr0) {
    case 1: return launch_one_t<1, BMAX>(in, out, oa, batch, pl, st);
    case 2: return launch_one_t<2, BMAX>(in, out, oa, batch, pl, st);
    case 3: return launch_one_t<3, BMAX>(in, out, oa, batch, pl, st);
    case 4: return launch_one_t<4, BMAX>(in, out, oa, batch, pl, st);
    case 5: return launch_one_t<5, BMAX>(in, out, oa, batch, pl, st);
    default: set_error("m32 NTT: unsupported one-pass length"); return GFA_ERR_UNSUPPORTED;
    }
}

int launch_one(int cls, int logr0, const i32 *in, i32 *out, const M32OneArgs &oa, i64 batch, const M32Plan *pl, hipStream_t st)
{
    if (cls == 0) return launch_one_b<32>(logr0, in, out, oa, batch, pl, st);
    if (cls == 1) return launch_one_b<8>(logr0, in, out, oa, batch, pl, st);
    return launch_one_b<4>(logr0, in, out, oa, batch, pl, st);
}

} // namespace

namespace gfa {

bool ntt_m32_eligible(const FieldDev &fd, i64 n)
{
    // odd p < 2^29 (signed int32 representatives: DifSched); 32 <= n <= 2^28 points (byte offsets of a tile stay below 2^32)
    if (fd.kind != KIND_PRIME32 || fd.p >= (1ull << 29) || (fd.p & 1) == 0) return false;
    if (n < 32 || n > ((i64)1 << 28) || (n & (n - 1))) return false;
    int logn = 0;
    while (((i64)1 << logn) < n) logn++;
    return logn <= 10 || logn >= 21 || (logn - (logn + 1) / 2) >= 5; // every pass of a multi-pass transform needs lines of >= 32 points
}

// Scratch for the multi-pass forms: n * batch elements for two passes, n for three (one transform at a time); the caller owns it.
// Contiguous rows only.
size_t ntt_m32_scratch_bytes(i64 n, i64 batch) { return n > ((i64)1 << 20) ? sizeof(i32) * (size_t)n : n > 1024 ? sizeof(i32) * (size_t)n * (size_t)batch : 0; }

int ntt_m32(const FieldDev &fd, const void *in, void *out, void *ws, i64 n, i64 batch, u64 omega, int do_scale, u64 scale,
            hipStream_t st)
{
    int dev = 0;
    GFA_HIP(hipGetDevice(&dev));
    M32Plan *pl;
    {
        std::lock_guard<std::mutex> lock(g_m32_mu);
        const int split = n > ((i64)1 << 20) && g_m32_tune[0] ? (g_m32_tune[0] << 8 | g_m32_tune[1]) : 0;
        const M32Key key{fd.p, dev, n, omega, split};
        auto it = g_m32_plans.find(key);
        if (it == g_m32_plans.end()) {
            // plans are kept for the life of the process (another host thread may be launching from one): a plan is ~140 KiB of
            // tables for a 2^20-point transform, one per (p, device, n, omega) ever used
            M32Plan *np = new M32Plan();
            const int rc = build_plan(np, fd.p, n, omega, split, st);
            if (rc) { free_plan(np); return rc; }
            it = g_m32_plans.emplace(key, np).first;
        }
        pl = it->second;
    }
    const u32 pinv = inv_2_32((u32)fd.p);
    const int cls = fd.p < (1ull << 26) ? 0 : fd.p < (1ull << 28) ? 1 : 2;
    M32Args base{};
    base.p = (i32)fd.p;
    base.pinv = pinv;
    base.fin = mont_centred(do_scale ? scale % fd.p : 1, fd.p);
    base.finp = (i32)((u32)base.fin * pinv);
    base.one = mont_centred(1, fd.p);
    base.onep = (i32)((u32)base.one * pinv);
    const i32 *src = (const i32 *)in;
    i32 *dst = (i32 *)out;
    if (pl->log2 == 0) {
        M32Args a = base;
        a.in_stride_c = n; a.in_stride_t = 1; a.out_stride_c = n; a.out_stride_t = 1;
        a.total_lines = batch;
        a.load_along_line = 1; a.store_along_line = 1;
        return launch(cls, pl->log1, src, dst, a, 1, pl->net1, pl->mid1, nullptr, nullptr, st, 0);
    }
    if (pl->log3) {
        // n = L0 * M, M = L1 * L2, three passes, each reading and writing every point once (input index j1 * M + j2 * L2 + j3,
        // output index k1 + L0 * (k2 + L1 * k3)):
        //   A : the M columns of length L0 (stride M), times w_n^(jr * k1)                       -> ws[k1 * M + jr]
        //   B1: per row k1, the L2 columns of length L1 (stride L2), times w_M^(j3 * k2)         -> ws[k1 * M + k2 * L2 + j3]  (in place)
        //   B2: per (k2, k1) the contiguous row of L2 points                                     -> X[k1 + L0 * (k2 + L1 * k3)]
        // B2 takes k2 as the batch index and k1 as the line index: the lines of a tile are then adjacent in the output.
        const i64 L0 = (i64)1 << pl->log1, L1 = (i64)1 << pl->log2, L2 = (i64)1 << pl->log3, M = L1 * L2;
        i32 *w = (i32 *)ws;
        for (i64 b = 0; b < batch; b++) {
            const i32 *sb = src + b * n;
            i32 *db = dst + b * n;
            int rc;
            {
                M32Args a = base;
                a.in_stride_c = 1; a.in_stride_t = M; a.out_stride_c = 1; a.out_stride_t = M;
                a.total_lines = M;
                a.twh = pl->at0h; a.tw2h = pl->aratioh; a.tw_bits = pl->a_bits;
                if ((rc = launch(cls, pl->log1, sb, w, a, 1, pl->net1, pl->mid1, pl->at0, pl->aratio, st, 1))) return rc;
            }
            {
                M32Args a = base;
                a.in_stride_c = 1; a.in_stride_t = L2; a.out_stride_c = 1; a.out_stride_t = L2;
                a.in_batch_stride = M; a.out_batch_stride = M;
                a.total_lines = L2;
                if ((rc = launch(cls, pl->log2, w, w, a, L0, pl->net2, pl->mid2, pl->pt0, pl->pratio, st, 1))) return rc;
            }
            {
                M32Args a = base;
                a.in_stride_c = M; a.in_stride_t = 1; a.out_stride_c = 1; a.out_stride_t = L0 * L1;
                a.in_batch_stride = L2; a.out_batch_stride = L0;
                a.total_lines = L0;
                a.load_along_line = 1; a.store_along_line = 0;
                // non-temporal last pass: measured per size (profiles/r05_m32_tune3.txt): 2^26 points 0.359 ms with, 0.406 without; 2^24 points
                // 0.082 with, 0.076 without (the whole working set of a 64 MiB transform stays in the Infinity Cache on its own)
                bool in_cache = (size_t)n * sizeof(i32) >= ((size_t)128 << 20);
                if (g_m32_tune[2] >= 0) in_cache = g_m32_tune[2] != 0;
                if ((rc = launch(cls, pl->log3, w, db, a, L1, pl->net3, pl->mid3, nullptr, nullptr, st, in_cache ? 2 : 0))) return rc;
            }
        }
        return GFA_OK;
    }
    // 2^11 .. 2^16 points: one workgroup per transform (one pass over HBM: 0.41-0.48 of the roofline against 0.29-0.35 in two passes,
    // profiles/r03_ntt_mid_sizes.txt, r04_m32_2e16.txt).  2^16 points need at least 64 transforms (one persistent workgroup per CU);
    // smaller batches keep the two-pass form.
    const bool is16 = pl->log1 + pl->log2 == 16;
    if (pl->one_wj && batch <= 0x7fffffff && (!is16 || batch >= 64)) {
        M32OneArgs oa{};
        oa.p = base.p; oa.pinv = pinv;
        oa.one = base.one; oa.onep = base.onep;
        oa.fin = base.fin; oa.finp = base.finp;
        if (is16) return launch_2e16(fd.p, src, dst, oa, batch, pl, st);
        return launch_one(cls, (int)(pl->log1 + pl->log2 - 10), src, dst, oa, batch, pl, st);
    }
    const i64 n1 = (i64)1 << pl->log1, n2 = (i64)1 << pl->log2;
    i32 *w = (i32 *)ws;
    { // pass 1: the n2 columns (length n1, stride n2), times w^(j2*k1); same layout out: ws[k1 * n2 + j2]
        M32Args a = base;
        a.in_stride_c = 1; a.in_stride_t = n2; a.out_stride_c = 1; a.out_stride_t = n2;
        a.in_batch_stride = n; a.out_batch_stride = n;
        a.total_lines = n2;
        const int rc = launch(cls, pl->log1, src, w, a, batch, pl->net1, pl->mid1, pl->pt0, pl->pratio, st, 1);
        if (rc) return rc;
    }
    { // pass 2: the n1 rows (contiguous), stored transposed: X[k1 + n1*k2]
        M32Args a = base;
        a.in_stride_c = n2; a.in_stride_t = 1; a.out_stride_c = 1; a.out_stride_t = n1;
        a.in_batch_stride = n; a.out_batch_stride = n;
        a.total_lines = n1;
        a.load_along_line = 1; a.store_along_line = 0;
        bool in_cache = (size_t)n * (size_t)batch * sizeof(i32) <= ((size_t)256 << 20); // the intermediate fits the Infinity Cache
        if (g_m32_tune[2] >= 0) in_cache = g_m32_tune[2] != 0;
        return launch(cls, pl->log2, w, dst, a, batch, pl->net2, pl->mid2, nullptr, nullptr, st, in_cache ? 2 : 0);
    }
}

} // namespace gfa

// tuning aid (tools/m32_tune3.py): see g_m32_tune
extern "C" void gfa_debug_m32_tune(int key, int value)
{
    if (key >= 0 && key < 4) g_m32_tune[key] = value;
}
