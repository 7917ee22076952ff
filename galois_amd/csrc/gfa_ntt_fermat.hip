// gfa_ntt_fermat.hip -- 2^16-point transforms over the Fermat prime field GF(65537) in ONE pass over HBM.
//
// Replaces fft_jit / ifft_jit (reference: src/galois/_domains/_function.py:246-392) for BASELINE config C3-i
// (batches of 2^16-point transforms over GF(65537)).  Exact integer arithmetic, so any correct DFT algorithm reproduces
// the reference's bits; the structure here is chosen for the machine:
//
//   * one 1024-thread workgroup owns one whole transform: 64 points per thread live in VGPRs (256 KiB of the CU's 512 KiB
//     register file), the array is read once and written once (8 B/point, the algorithmic minimum);
//   * N = 64 * 32 * 32: three fully unrolled in-register decimation-in-frequency networks (radix 64, 32, 32) joined by
//     two exchanges through LDS (each in two rounds of 128 KiB);
//   * 2 has order 32 and sqrt(2) = 2^12 - 2^4 order 64 modulo 2^16 + 1, so every twiddle INSIDE a network is a shift (or
//     one product with a 16-bit constant) followed by a fold  lo16(x) - (x >> 16)  -- one v_sub_u32_sdwa.  Values stay
//     loose signed 32-bit representatives; tools/gen_fermat_net.py places the folds with exact interval tracking
//     (gfa_fermat_nets.inc is its output);
//   * only the two twiddles BETWEEN networks are general products: balanced 16-bit factors, one v_mul_lo_u32 + two folds.
//     w^(m*k0) is formed in registers from two per-thread seeds (gfa_fermat_tw.h; r03-r05 streamed a 256 KiB table through the
//     same in-order memory queue as the data: 2-3.5 us of waiting per ~28 us round), w_1024^(r*k1) comes from a 4 KiB LDS table;
//   * the next transform's input is requested as early as registers allow: 16 rows once exchange 1 has taken the points, 24
//     more before the second half of network 1, the last 24 after the stores (r06: 0.57 -> 0.65 of the HBM roofline at 1024
//     transforms, 0.51 -> 0.64 at 4096; tools/ubench/fermat_r06.hip holds the variants, profiles/r06_fermat_*.txt the numbers);
//   * the networks use the canonical roots (sqrt(2), 2).  A transform with root of unity w has w^(N/64) = sqrt(2)^u for
//     one odd u; feeding network inputs in the order a' = u*a mod R turns the canonical network into the wanted one, and
//     that permutation is folded into the global load addresses and the LDS write positions (no instructions).
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "gfa_internal.h"

using namespace gfa;

namespace {

__device__ __forceinline__ int fm_add(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int fm_sub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int fm_shl(int a, int k) { return (int)((unsigned)a << k); }
__device__ __forceinline__ int fm_mulc(int a, int c) { return (int)((unsigned)a * (unsigned)c); }
// x == lo16(x) - (x >> 16)  (mod 2^16 + 1); any int32 -> [-32767, 98303]
__device__ __forceinline__ int fm_fold(int t)
{
    int r;
    asm("v_sub_u32_sdwa %0, %1, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(r) : "v"(t));
    return r;
}
// balanced fold: x == sext16(x) - ((x + 2^15) >> 16); |x| < 2^29 -> |result| <= 32768 + 2^13 + 1
__device__ __forceinline__ int fm_bfold(int t)
{
    const int t2 = fm_add(t, 0x8000);
    int r;
    asm("v_sub_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
        : "=v"(r)
        : "v"(t), "v"(t2));
    return r;
}

#include "gfa_fermat_nets.inc"
#include "gfa_fermat_tw.h"

constexpr int brev_c(int x, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// general product with a balanced table factor |w| <= 32768: |x| < 2^29 in, [-32767, 98303] out
__device__ __forceinline__ int fm_mul_tw(int x, int w) { return fm_fold(fm_mulc(fm_bfold(x), w)); }

constexpr int E2_PITCH = 33;                  // exchange 2: r' runs fastest, k0 pitch 33 words (conflict-free both ways)
constexpr int e2_kpitch(int r0) { return 64 * E2_PITCH + r0; } // ... and k1 pitch 64 * 33 + R0: a reading wave's lanes (k0, k1 low bits) fall into distinct banks
constexpr int EX_WORDS = 16 * e2_kpitch(64);  // 34816 words >= exchange 1's 32 * 1024
constexpr int FERMAT_LDS_BYTES = (EX_WORDS + 1024) * 4;

struct FermatArgs {
    const u32 *in;
    u32 *out;
    const int *tw1; // [2][1024]: balanced w^m and w^(8 m), the seeds of the first twiddles
    const int *tw2; // [32][32]:   balanced w^(64 * r * k1), index k1 * 32 + r
    int u, uinv;    // w^(N/64) == sqrt(2)^u, uinv = u^-1 mod 64 (grouped kernels: w^(n/32) == 2^u, uinv = u^-1 mod 32)
    int batch;      // blocks of 2^16 words: transforms (LOGG = 0) or groups of 2^LOGG transforms
    long long words; // batch_of_transforms * n: the descriptors of the last block end there (its missing transforms read 0, store nothing)
    int stagger;    // first-round start offset between the four workgroup groups, in units of 4096 clocks (0: none)
    unsigned long long *dbg; // optional phase timestamps (100 MHz), 8 per (workgroup, round); nullptr in production
};
// final reduction of a network output |c| < 2^29 to the canonical [0, 65536] (NEGATE: of -c, the 1/N of the inverse):
// adding a multiple of p first makes the value non-negative, the first fold then ends in [-2^14, 65535] and the second
// in [0, 65536] -- no conditional step
constexpr int FM_OFFSET = 65537 * 8192; // == 0 mod p, >= 2^29
template <bool NEGATE>
__device__ __forceinline__ u32 fm_canon(int c)
{
    c = NEGATE ? fm_sub(FM_OFFSET, c) : fm_add(c, FM_OFFSET);
    return (u32)fm_fold(fm_fold(c));
}

// Workgroup barrier for LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL access (vmcnt(0)), which
// would stall each exchange on the twiddle loads in flight, on the stores of the finished half and on the early loads of
// the next transform.  LDS operations of a wave complete in order, so lgkmcnt(0) + s_barrier is all the exchange needs.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Cache hints, measured (tools/fermat_variants.sh, two repetitions): non-temporal STORES (the output is never re-read by
// this kernel; keeps the shared twiddle table in L2) but default-policy LOADS -- with non-temporal loads a batch whose
// input is about the size of the 256 MiB Infinity Cache (1024 transforms = 256 MiB) loses the hits it otherwise gets from one
// launch to the next: 0.125 vs 0.139 ms per 1024 transforms; at 4096 transforms (1 GiB, no reuse possible) the four
// combinations are within noise of each other (0.553 - 0.575 ms).
#ifndef GFA_FERMAT_AUX_LD
#define GFA_FERMAT_AUX_LD 0
#endif
#ifndef GFA_FERMAT_AUX_ST
#define GFA_FERMAT_AUX_ST 2
#endif
// the next transform's rows requested ahead: [0, E1) after the second exchange-1 write burst (the point registers are free),
// [E1, E2) before the second half of network 1; 42 and more spill (tools/ubench/fermat_r06.hip, profiles/r06_fermat_varB_early_loads.txt)
#ifndef GFA_FERMAT_E1
#define GFA_FERMAT_E1 16
#endif
#ifndef GFA_FERMAT_E2
#define GFA_FERMAT_E2 40
#endif
constexpr int AUX_NT = GFA_FERMAT_AUX_LD; // streaming (non-temporal) hint on the data loads: keep the shared twiddle table in L2
constexpr int AUX_ST = GFA_FERMAT_AUX_ST; // ... and on the stores

// The body of the loop is ONE basic block for the compiler unless something splits it, and its scheduler then stretches live
// ranges across phases until the allocator spills (every r04 variant with early loads did, 40-250 bytes, and lost 5-25 %).  A
// never-taken branch at each phase boundary -- what the time stamps of the DBG build are -- keeps the allocation at 128 registers
// with no scratch (r06).
#define FM_PHASE(i)                                          \
    do {                                                     \
        if (DBG) ts[i] = __builtin_amdgcn_s_memrealtime();   \
        else if (a.dbg != nullptr) asm volatile("s_nop 0");  \
    } while (0)
#define FM_SPLIT()                                           \
    do {                                                     \
        if (a.dbg != nullptr) asm volatile("s_nop 0");       \
    } while (0)

// first network by radix (the generated canonical-root networks, inputs from memory)
template <int LOGR> struct FermatNet0;
template <> struct FermatNet0<6> { static __device__ __forceinline__ void run(int (&v)[64]) { fermat_net64_canon(v); } };
template <> struct FermatNet0<5> { static __device__ __forceinline__ void run(int (&v)[32]) { fermat_net32_canon(v); } };
template <> struct FermatNet0<4> { static __device__ __forceinline__ void run(int (&v)[16]) { fermat_net16_canon(v); } };
template <> struct FermatNet0<3> { static __device__ __forceinline__ void run(int (&v)[8]) { fermat_net8_canon(v); } };
template <> struct FermatNet0<2> { static __device__ __forceinline__ void run(int (&v)[4]) { fermat_net4_canon(v); } };
template <> struct FermatNet0<1> { static __device__ __forceinline__ void run(int (&v)[2]) { fermat_net2_canon(v); } };
template <> struct FermatNet0<0> { static __device__ __forceinline__ void run(int (&)[1]) {} }; // 1024-point transforms: no first network

// LOGG > 0 (r06): the same workgroup transforms G = 2^LOGG consecutive transforms of n = 2^16 / G points -- one 2^16-word block of the
// batch.  Only the first network changes: G radix-(64 / G) networks over the rows of each transform instead of one radix-64 network
// (row a of transform t is row t * R0 + a of the block, so every address stays what it was); the combined index c = t * R0 + k0 then
// plays the part k0 plays at G = 1 through both exchanges and the two radix-32 networks (the 1024-point sub-transform over m, root
// w^R0).  Output X_t[k0 + R0 (k1 + 32 k2)] is word t * n + k0 + R0 (k1 + 32 k2) of the block: lane l = (t, k0) writes runs of R0
// words.  First twiddles w^(m k0), k0 < R0: one progression per half of the transforms, each value applied to G / 2 points.
// 1 / n = -2^LOGG (2^16 = -1): the scaled inverse folds once more and shifts.
template <bool NEGATE, bool DBG, int LOGG>
__global__ __launch_bounds__(1024) void ntt_fermat16_kernel(FermatArgs a)
{
    constexpr int LOGR0 = 6 - LOGG, R0 = 1 << LOGR0, G = 1 << LOGG;
    // rows of the next transform requested before network 1's second half: 40, less where that spills (LOGG 1: 8 bytes at 40; LOGG 5, whose second
    // exchange keeps a whole first-network half alive: 20-68 bytes above 24)
    constexpr int E2 = LOGG == 1 ? (GFA_FERMAT_E2 < 38 ? GFA_FERMAT_E2 : 38) : LOGG >= 5 ? (GFA_FERMAT_E2 < 24 ? GFA_FERMAT_E2 : 24) : GFA_FERMAT_E2;
    // position of the combined output c = t * R0 + k0 in the point registers (each network leaves its outputs bit-reversed)
    auto pos = [](int c) constexpr { return (c >> LOGR0) * R0 + brev_c(c & (R0 - 1), LOGR0); };
    extern __shared__ int lds[];
    int *ex = lds;
    int *tw2l = lds + EX_WORDS;
    const unsigned tid = threadIdx.x;
    const int voff = (int)(tid * 4u);
    const int g = (int)(tid >> 5), r = (int)(tid & 31);   // exchange 1 / network 1 coordinates
    const int l = (int)(tid & 63), wv = (int)(tid >> 6);  // exchange 2 / network 2 coordinates
    const int wpos1 = (((a.u * g) & 31) << 5) + r;        // exchange 1 write slot b' = u * b mod 32
    const int wpos2 = ((a.u * r) & 31);                   // exchange 2 write slot r' = u * r mod 32
    int *const e1w = ex + wpos1;
    const int *const e1r = ex + g * 1024 + r;
    // exchange 2, [k1 (16 per round)][c (64)][r']: the reader of round h is (lane l = (k1 low LOGG bits, k0), wave wv = (k1 high bits, t)):
    // row k1 = (l >> LOGR0) + G * (wv >> LOGG) + 16 h of the combined column c = (wv mod G) * R0 + (l mod R0), so that a wave's 64 lanes
    // own 64 CONSECUTIVE output words k0 + R0 * k1_low of one transform (256-byte stores whatever G)
    // G = 32 (2^11 points): a wave's 64 output words are k0 + 2 k1 for ALL k1 of one transform t = wv + 16 h, so the rounds go by transform
    // half instead: round h holds [k1 (32)][c - 32 h (32)][r'] of the first-network half h, k1 pitch 32 * 33 + 2
    // G = 64 (2^10 points): lanes = (one bit of t, all 32 k1), t = 32 h + 2 wv + (l >> 5): two runs of 32 words per wave store; k1 pitch 32 * 33 + 1
    constexpr int KP = e2_kpitch(R0), KP5 = 32 * E2_PITCH + (LOGG == 6 ? 1 : 2);
    int *const e2w = ex + g * E2_PITCH + wpos2;
    const int *const e2r = LOGG < 5    ? ex + ((l >> LOGR0) + G * (wv >> LOGG)) * KP + (((wv & (G - 1)) << LOGR0) + (l & (R0 - 1))) * E2_PITCH
                           : LOGG == 5 ? ex + (l >> 1) * KP5 + (2 * wv + (l & 1)) * E2_PITCH
                                       : ex + (l & 31) * KP5 + (2 * wv + (l >> 5)) * E2_PITCH;
    tw2l[tid] = a.tw2[tid];
    const int seed1 = a.tw1[tid], seed8 = LOGG == 0 ? a.tw1[1024 + tid] : 0; // w^m, w^(8 m): the whole kernel
    // stores: X_t[k0 + R0 (k1 + 32 k2)] = word t * n + l + 64 * (wv >> LOGG) of the block, + (1024 / G) h + 32 R0 k2 as the scalar offset
    const int soff = LOGG < 6 ? (int)((((wv & (G - 1)) << (16 - LOGG)) + l + 64 * (wv >> LOGG)) * 4)
                              : (int)(((2 * wv + (l >> 5)) * 1024 + (l & 31)) * 4);
    // Workgroups are persistent (one per CU) and all run the same program, so without help every CU would read, compute
    // and write at the same moments and HBM would idle while the chip computes.  The first round is staggered in four
    // groups (each XCD holds all four): group j starts j * stagger later, and the offset persists from round to round.
    if (a.stagger > 0) {
        const int grp = (int)((blockIdx.x >> 3) & 3u);
        for (int i = 0; i < grp * a.stagger; i++) __builtin_amdgcn_s_sleep(64);
    }
    // every global access is `buffer_* v, v_off, s[rsrc], s_off offen`: lane offset tid*4 in one VGPR, row offset in an
    // SGPR, descriptors from kernel arguments and the (wave-uniform) transform index -- no vector address arithmetic.
    // `live` = false gives a descriptor of zero records: its loads return 0 and move nothing (the last round's look-ahead)
    auto block_bytes = [&](unsigned t) { // a whole block, or what is left of the batch in the last one
        const long long left = a.words - (long long)t * 65536;
        return (int)((left < 65536 ? left : 65536) * 4);
    };
    auto in_rsrc = [&](unsigned t, bool live) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + (size_t)t * 65536u), 0, live ? block_bytes(t) : 0, 0x00020000);
    };
    // row of the block behind network position ap = t * R0 + a': a = uinv * a' mod R0 (the permutation that makes the canonical network compute this root's)
    auto row_off = [&](unsigned uinv_, int ap) { return (int)((((uinv_ * (unsigned)(ap & (R0 - 1))) & (unsigned)(R0 - 1)) + (unsigned)(ap & ~(R0 - 1))) << 12); };
    int v[64];
    {
        const __amdgpu_buffer_rsrc_t xr = in_rsrc(blockIdx.x, true);
#pragma unroll
        for (int ap = 0; ap < 64; ap++) v[ap] = (int)__builtin_amdgcn_raw_buffer_load_b32(xr, voff, row_off((unsigned)a.uinv, ap), AUX_NT);
    }
    for (unsigned tr_i = blockIdx.x; tr_i < (unsigned)a.batch; tr_i += gridDim.x) {
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (size_t)tr_i * 65536u), 0, block_bytes(tr_i), 0x00020000);
        const unsigned tr_next = tr_i + gridDim.x;
        const bool has_next = tr_next < (unsigned)a.batch;
        const __amdgpu_buffer_rsrc_t xn = in_rsrc(has_next ? tr_next : tr_i, has_next);
        unsigned uinv = (unsigned)a.uinv;
        asm volatile("" : "+s"(uinv)); // keep the 64 row offsets out of loop-invariant SGPRs (they are 3 scalar ops each)
        unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // phase timestamps: scalar registers, written out once at the end
        auto request = [&](int lo, int hi) { // rows lo .. hi-1 of the next transform, into the point registers it will start from
#pragma unroll
            for (int ap = lo; ap < hi; ap++) v[ap] = (int)__builtin_amdgcn_raw_buffer_load_b32(xn, voff, row_off(uinv, ap), AUX_NT);
        };
        FM_PHASE(0);
        // ---- network 0: radix 64 over a (stride 1024); thread m = tid.  Twiddles w^(m * k0), k0 = 1..63: gfa_fermat_tw.h ----
#pragma unroll
        for (int t = 0; t < G; t++) FermatNet0<LOGR0>::run(reinterpret_cast<int (&)[R0]>(v[t * R0]));
        FM_PHASE(1);
        if (LOGG < 6) { // (2^10 points: the inputs themselves, canonical, feed network 1)
#pragma unroll
            for (int t = 0; t < G; t++) v[t * R0] = fm_fold(v[t * R0]);
        }
        int twa[8], twb[8];
        if (LOGG == 0) fm_tw_progressions(seed1, seed8, twa, twb);
        auto tw1_range = [&](int lo, int hi) { // combined outputs lo .. hi-1 (halves: 1..31 / 32..63 at G = 1, transforms t < G / 2 / the rest above)
            if (LOGG == 0) {
#pragma unroll
                for (int k0 = lo; k0 < hi; k0++) {
                    int &q = v[brev_c(k0, 6)];
                    q = fm_tw_apply(q, fm_tw_of(twa, twb, k0));
                }
            } else {
                int T = seed1; // w^(m k0), k0 = 1, 2, ...: tight products by the seed (|T| <= 32770)
#pragma unroll
                for (int k0 = 1; k0 < R0; k0++) {
#pragma unroll
                    for (int t = lo >> LOGR0; t < (hi + R0 - 1) >> LOGR0; t++) {
                        int &q = v[t * R0 + brev_c(k0, LOGR0)];
                        q = fm_tw_apply(q, T);
                    }
                    if (k0 + 1 < R0) T = fm_tw_tight(T, seed1);
                }
            }
        };
        tw1_range(1, 32);
        FM_PHASE(2);
        // ---- exchange 1 + network 1: thread (g, r) takes k0 = g and g + 32, radix 32 over b (m = 32 b + r).  Each LDS
        // write burst is followed by arithmetic that does not depend on it, so the LDS pipe and the VALU overlap ----
        int w[2][32];
        lds_barrier(); // the previous transform's exchange-2 reads (first round: the staging of tw2l) are complete
#pragma unroll
        for (int kl = 0; kl < 32; kl++) e1w[kl * 1024] = v[pos(kl)];
        tw1_range(32, 64);
        lds_barrier();
#pragma unroll
        for (int bp = 0; bp < 32; bp++) w[0][bp] = e1r[bp * 32];
        lds_barrier();
#pragma unroll
        for (int kl = 0; kl < 32; kl++) e1w[kl * 1024] = v[pos(kl + 32)];
        FM_PHASE(3);
        request(0, GFA_FERMAT_E1);
        auto net1 = [&](int h) {
            fermat_net32_fold(w[h]);
            w[h][0] = fm_fold(w[h][0]);
#pragma unroll
            for (int k1 = 1; k1 < 32; k1++) {
                int &q = w[h][brev_c(k1, 5)];
                q = fm_mul_tw(q, tw2l[k1 * 32 + r]);
            }
        };
        net1(0);
        lds_barrier();
#pragma unroll
        for (int bp = 0; bp < 32; bp++) w[1][bp] = e1r[bp * 32];
        FM_SPLIT();
        request(GFA_FERMAT_E1, E2);
        net1(1);
        FM_PHASE(4);
        // ---- exchange 2 + network 2: the reader (lane l, wave wv) of round h: see e2r; radix 32 over r ----
        int z[2][32];
        lds_barrier();
        if constexpr (LOGG < 5) { // rounds by k1 half: both first-network halves' k1 = 16 h .. 16 h + 15
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int kl = 0; kl < 16; kl++) e2w[kl * KP + 32 * i * E2_PITCH] = w[i][brev_c(kl, 5)];
        } else { // G = 32: rounds by TRANSFORM half (a reader's 64 lanes are all 32 k1 of one transform): w[h], every k1
#pragma unroll
            for (int k1 = 0; k1 < 32; k1++) e2w[k1 * KP5] = w[0][brev_c(k1, 5)];
        }
        lds_barrier();
#pragma unroll
        for (int rp = 0; rp < 32; rp++) z[0][rp] = e2r[rp];
        lds_barrier();
        if constexpr (LOGG < 5) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int kl = 0; kl < 16; kl++) e2w[kl * KP + 32 * i * E2_PITCH] = w[i][brev_c(kl + 16, 5)];
        } else {
#pragma unroll
            for (int k1 = 0; k1 < 32; k1++) e2w[k1 * KP5] = w[1][brev_c(k1, 5)];
        }
        FM_PHASE(5);
        auto net2 = [&](int h) {
            fermat_net32_fold(z[h]);
#pragma unroll
            for (int k2 = 0; k2 < 32; k2++) {
                int c = z[h][brev_c(k2, 5)];
                if (NEGATE && LOGG > 0) c = fm_shl(fm_fold(c), LOGG); // 1 / n = -2^LOGG
                const int so = LOGG < 5 ? (32 * R0 * k2 + (1024 >> LOGG) * h) * 4 : (32 * R0 * k2 + 32768 * h) * 4; // (G >= 32: h = transform half)
                __builtin_amdgcn_raw_buffer_store_b32(fm_canon<NEGATE>(c), yr, soff, so, AUX_ST);
            }
        };
        net2(0);
        FM_PHASE(6);
        lds_barrier();
#pragma unroll
        for (int rp = 0; rp < 32; rp++) z[1][rp] = e2r[rp];
        FM_SPLIT();
        net2(1);
        request(E2, 64);
        FM_PHASE(7);
        if (DBG && tid == 0) {
            unsigned long long *d = a.dbg + ((size_t)(tr_i / gridDim.x) * gridDim.x + blockIdx.x) * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) d[i] = ts[i];
        }
    }
}

unsigned long long *g_fermat_dbg = nullptr;

struct FermatPlan {
    int *tw1 = nullptr, *tw2 = nullptr;
    int u = 0, uinv = 0, cus = 256, logg = 0; // logg: 2^logg transforms of 2^(16 - logg) points per workgroup
    bool ok = false;
};
std::mutex g_mu;
std::map<std::pair<int, u64>, FermatPlan> g_plans; // (device, omega)

inline u32 mulmod(u32 a, u32 b) { return (u32)(((u64)a * b) % 65537u); }
inline int balanced(u32 c) { return c > 32768u ? (int)c - 65537 : (int)c; }

} // namespace

namespace gfa {

// GFA_FERMAT_MIN_LOGN: the shortest transform the grouped kernel takes (2^10: sixty-four per workgroup, no first network)
#ifndef GFA_FERMAT_MIN_LOGN
#define GFA_FERMAT_MIN_LOGN 10
#endif
bool ntt_fermat16_eligible(const FieldDev &fd, i64 n, i64 batch)
{
    if (fd.kind != KIND_PRIME32 || fd.p != 65537 || n > 65536 || n < ((i64)1 << GFA_FERMAT_MIN_LOGN) || (n & (n - 1))) return false;
    return batch * n >= (i64)64 * 65536; // one persistent workgroup per CU and block of 2^16 words: fewer leave the chip idle
}

// in / out: uint32, batch transforms of n = 2^13 .. 2^16 points (n = the order of omega).  negate: the scaled inverse (1 / n = -2^(16 - log n)).
// Returns GFA_ERR_UNSUPPORTED (nothing launched) when omega is not a primitive root of unity of such an order.
int ntt_fermat16(const void *in, void *out, i64 batch, u64 omega, int negate, hipStream_t st)
{
    int dev = 0;
    GFA_HIP(hipGetDevice(&dev));
    FermatPlan pl;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        FermatPlan &p = g_plans[std::make_pair(dev, omega)];
        if (!p.tw1 && !p.ok) {
            // the order of w: w^(2^k) == -1  <=>  order 2^(k + 1)
            int logn = 0;
            {
                u32 t = (u32)(omega % 65537u);
                for (int k = 0; k < 16 && !logn; k++) {
                    if (t == 65536u) logn = k + 1;
                    t = mulmod(t, t);
                }
            }
            if (logn < GFA_FERMAT_MIN_LOGN || logn > 16) return GFA_ERR_UNSUPPORTED;
            const int logg = 16 - logn, r0 = 64 >> logg;
            std::vector<u32> pw((size_t)1 << logn);
            pw[0] = 1;
            for (size_t e = 1; e < pw.size(); e++) pw[e] = mulmod(pw[e - 1], (u32)omega);
            const u32 nmask = (u32)pw.size() - 1;
            int u = 0;
            if (logg == 0) { // w^(n / 64) = sqrt(2)^u, u odd mod 64
                const u32 w64 = pw[1024];
                u32 z = 4080u, zz = mulmod(z, z), cur = z;
                for (int c = 1; c < 64; c += 2) {
                    if (cur == w64) { u = c; break; }
                    cur = mulmod(cur, zz);
                }
            } else { // w^(n / 32) = 2^u, u odd mod 32
                const u32 w32 = pw[pw.size() / 32];
                u32 cur = 2;
                for (int c = 1; c < 32; c += 2) {
                    if (cur == w32) { u = c; break; }
                    cur = mulmod(cur, 4);
                }
            }
            if (!u) return GFA_ERR_UNSUPPORTED;
            const int umod = logg == 0 ? 64 : 32;
            int uinv = 1;
            while ((u * uinv) % umod != 1) uinv += 2;
            std::vector<int> t1(2 * 1024), t2(32 * 32);
            for (u32 m = 0; m < 1024; m++) {
                t1[m] = balanced(pw[m & nmask]);                  // w^m
                t1[1024 + m] = balanced(pw[(8 * m) & nmask]);     // w^(8 m) (n = 2^16 only)
            }
            for (u32 k1 = 0; k1 < 32; k1++)
                for (u32 r = 0; r < 32; r++) t2[k1 * 32 + r] = balanced(pw[((u32)r0 * r * k1) & nmask]); // the 1024-point sub-transform's root is w^R0
            GFA_HIP(hipMalloc((void **)&p.tw1, t1.size() * sizeof(int)));
            GFA_HIP(hipMalloc((void **)&p.tw2, t2.size() * sizeof(int)));
            GFA_HIP(hipMemcpy(p.tw1, t1.data(), t1.size() * sizeof(int), hipMemcpyHostToDevice));
            GFA_HIP(hipMemcpy(p.tw2, t2.data(), t2.size() * sizeof(int), hipMemcpyHostToDevice));
            hipDeviceProp_t prop;
            GFA_HIP(hipGetDeviceProperties(&prop, dev));
            p.u = u; p.uinv = uinv; p.cus = prop.multiProcessorCount; p.logg = logg; p.ok = true;
        }
        pl = p;
    }
    constexpr int stagger_env = 2; // first-round stagger of the four workgroup groups, in units of 4096 clocks (measured against 0 and 4)
    const i64 n = (i64)65536 >> pl.logg;
    const i64 nblk = (batch * n + 65535) / 65536;
    const i64 grid = std::min<i64>(nblk, pl.cus);
    // the stagger only pays when every group has later rounds to keep busy
    FermatArgs a{(const u32 *)in, (u32 *)out, pl.tw1, pl.tw2, pl.u, pl.uinv, (int)nblk, (long long)(batch * n), nblk >= 2 * grid ? stagger_env : 0, g_fermat_dbg};
#define GFA_FERMAT_LAUNCH(NEG, DBGV, LG)                                                                                          \
    do {                                                                                                                          \
        static bool attr = false;                                                                                                 \
        if (!attr) {                                                                                                              \
            GFA_HIP(hipFuncSetAttribute((const void *)ntt_fermat16_kernel<NEG, DBGV, LG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr = true;                                                                                                          \
        }                                                                                                                         \
        hipLaunchKernelGGL((ntt_fermat16_kernel<NEG, DBGV, LG>), dim3((unsigned)grid), dim3(1024), FERMAT_LDS_BYTES, st, a);      \
    } while (0)
#define GFA_FERMAT_BY_NEG(LG)                      \
    do {                                           \
        if (negate) GFA_FERMAT_LAUNCH(true, false, LG);  \
        else GFA_FERMAT_LAUNCH(false, false, LG);  \
    } while (0)
    switch (pl.logg) {
    case 0:
        if (a.dbg && !negate) GFA_FERMAT_LAUNCH(false, true, 0);
        else GFA_FERMAT_BY_NEG(0);
        break;
    case 1: GFA_FERMAT_BY_NEG(1); break;
    case 2: GFA_FERMAT_BY_NEG(2); break;
    case 3: GFA_FERMAT_BY_NEG(3); break;
    case 4: GFA_FERMAT_BY_NEG(4); break;
    case 5: GFA_FERMAT_BY_NEG(5); break;
    case 6: GFA_FERMAT_BY_NEG(6); break;
    default: return GFA_ERR_UNSUPPORTED;
    }
#undef GFA_FERMAT_BY_NEG
#undef GFA_FERMAT_LAUNCH
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// tuning aid (tools/fermat_phases.py): device buffer of 8 timestamps per (round, workgroup), or nullptr to switch off
extern "C" void gfa_debug_fermat_stamps(unsigned long long *buf) { g_fermat_dbg = buf; }

} // namespace gfa
