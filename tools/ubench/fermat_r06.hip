// Round-6 experiments on the one-pass GF(65537) 2^16-point kernel: stand-alone, one executable per
// configuration (-D macros), checked against the product kernel through the C-ABI (libgalois_amd.so).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVAR=... tools/ubench/fermat_r06.hip -Iinclude -Lgalois_amd -lgalois_amd -o ...
//   ./fermat_r06_<cfg> [check_batch] [time_batch ...]
//
// VARIANT A (this file's kernel): ONE 512-thread workgroup per CU, 128 points per thread, 256-register budget.
//   * same decomposition N = 64 * 32 * 32 and the same generated networks as the product kernel;
//   * both exchanges in FOUR rounds of 64 KiB through TWO buffers: one workgroup barrier per round (a round's writes go to the
//     buffer whose last reads lie behind the previous barrier), arithmetic of round q-1 between a round's writes and its barrier;
//   * the next transform's input is requested into the registers the FIRST butterfly layer has just freed (PF of the 128
//     values: the loop-carried input array), i.e. a whole transform ahead of its use; the remaining 128 - PF as the last
//     network frees registers.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

#include "galois_amd.h"

typedef unsigned u32;
typedef unsigned long long u64;

#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)

#ifdef DFS
#define DFS_INC DFS
#else
#define DFS_INC 0
#endif
#ifndef PF
#define PF 96 // values of the next transform requested a round ahead (multiple of 32)
#endif
#ifndef TWW
#define TWW 32 // window of first twiddles in flight
#endif
#ifndef PFPOS
#define PFPOS 0 // 0: all early requests after the first-twiddle products; 1: set 0 straight after its network
#endif
#ifndef TSHIFT
#define TSHIFT 0 // tail requests one network-2 round earlier
#endif
#ifndef SKEL
#define SKEL 0 // 1: no HBM traffic; 2: no arithmetic
#endif
#ifndef STAGGER
#define STAGGER 2
#endif
#ifndef AUX_LD
#define AUX_LD 0
#endif
#ifndef AUX_ST
#define AUX_ST 2
#endif
#ifndef PRIO
#define PRIO 0
#endif
#ifndef VARB
#define VARB 0 // 1: time variant B (1024 threads) instead of variant A
#endif
#ifndef DEEP
#define DEEP 0 // 1: a round's reads and the NEXT round's writes are issued before the previous round's network (one round more in flight)
#endif
#ifndef TWMODE
#define TWMODE 0 // 1: first twiddles w^(m k0) formed in registers from the per-thread seeds w^m and w^(8m) (no table stream)
#endif
#ifndef TAILFAKE
#define TAILFAKE 0 // 1: the tail requests are not made (WRONG results: prices them)
#endif
#ifndef TWFAKE
#define TWFAKE 0 // 1: first twiddles never re-requested (WRONG results: prices the table stream)
#endif

namespace {

__device__ __forceinline__ int fm_add(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int fm_sub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int fm_shl(int a, int k) { return (int)((unsigned)a << k); }
__device__ __forceinline__ int fm_mulc(int a, int c) { return (int)((unsigned)a * (unsigned)c); }
__device__ __forceinline__ int fm_fold(int t)
{
    int r;
    asm("v_sub_u32_sdwa %0, %1, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(r) : "v"(t));
    return r;
}
__device__ __forceinline__ int fm_bfold(int t)
{
    const int t2 = fm_add(t, 0x8000);
    int r;
    asm("v_sub_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
        : "=v"(r)
        : "v"(t), "v"(t2));
    return r;
}

#if DFS_INC
#include "../../_variants/gfa_fermat_nets_split.inc" // python tools/gen_fermat_net.py --split -o _variants/gfa_fermat_nets_split.inc
#else
#include "../../galois_amd/csrc/gfa_fermat_nets.inc"
#endif

constexpr int brev_c(int x, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}
__device__ __forceinline__ int fm_mul_tw(int x, int w) { return fm_fold(fm_mulc(fm_bfold(x), w)); }

constexpr int E2_PITCH = 33;
constexpr int XW = 8 * 64 * E2_PITCH; // 16896 words per buffer >= exchange 1's 16 * 1024
constexpr int A_LDS_BYTES = (2 * XW + 1024) * 4;

struct FermatArgs {
    const u32 *in;
    u32 *out;
    const int *tw1; // [64][1024]: balanced w^(m * k0)
    const int *tw2; // [32][32]:   balanced w^(64 * r * k1), index k1 * 32 + r
    int u, uinv;
    int batch;
    int stagger;
    u64 *dbg;
};

constexpr int FM_OFFSET = 65537 * 8192;
__device__ __forceinline__ u32 fm_canon(int c)
{
    c = fm_add(c, FM_OFFSET);
    return (u32)fm_fold(fm_fold(c));
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifndef SCHEDB
#define SCHEDB 1 // scheduling barrier at every phase boundary (keeps the compiler from stretching live ranges across phases)
#endif
#define A_STAMP(i)                                                          \
    do {                                                                    \
        if (SCHEDB == 1) __builtin_amdgcn_sched_barrier(0);                 \
        if (SCHEDB == 2 && !DBG) {                                          \
            /* a never-taken branch: splits the basic block, as the stamps of the DBG build do (that build does not spill) */ \
            if (a.dbg != nullptr) asm volatile("s_nop 0");                  \
        }                                                                   \
        if (DBG) {                                                          \
            const u64 t_ = __builtin_amdgcn_s_memrealtime();                \
            if ((tid & 63u) == 0) __builtin_nontemporal_store(t_, dts + (i)); \
        }                                                                   \
    } while (0)

template <bool DBG>
__global__ __launch_bounds__(512) void fermat_a_kernel(FermatArgs a)
{
    extern __shared__ int lds[];
    int *const tw2l = lds + 2 * XW;
    const unsigned tid = threadIdx.x;
    const int voff = (int)(tid * 4u);
    const int gl = (int)(tid >> 5), r = (int)(tid & 31);  // exchange 1 / network 1 coordinates
    const int l = (int)(tid & 63), wv = (int)(tid >> 6);  // exchange 2 / network 2 coordinates
    // exchange 1, chunk of 16 k0: [k0 local][b' = u * b mod 32][r]; m = tid + 512 s has b = gl + 16 s
    int *const e1w0 = lds + (((a.u * gl) & 31) << 5) + r;
    int *const e1w1 = lds + (((a.u * (gl + 16)) & 31) << 5) + r;
    const int *const e1r = lds + gl * 1024 + r;
    // exchange 2, chunk of 8 k1: [k1 local][k0][r' = u * r mod 32], k0 pitch 33
    int *const e2w = lds + gl * E2_PITCH + ((a.u * r) & 31);
    const int *const e2r = lds + wv * (64 * E2_PITCH) + l * E2_PITCH;
    const __amdgpu_buffer_rsrc_t tr = __builtin_amdgcn_make_buffer_rsrc((void *)a.tw1, 0, 65536 * 4, 0x00020000);
    tw2l[tid] = a.tw2[tid];
    tw2l[tid + 512] = a.tw2[tid + 512];
    if (a.stagger > 0) {
        const int grp = (int)((blockIdx.x >> 3) & 3u);
        for (int i = 0; i < grp * a.stagger; i++) __builtin_amdgcn_s_sleep(64);
    }
    if (PRIO) {
        if (tid & 256u) __builtin_amdgcn_s_setprio(1);
    }
    auto in_rsrc = [&](unsigned t, bool live) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + (size_t)t * 65536u), 0, live ? 65536 * 4 : 0, 0x00020000);
    };
    auto tw_load = [&](int j) { // stream position j: s = j / 63, k0 = j % 63 + 1
        return (int)__builtin_amdgcn_raw_buffer_load_b32(tr, voff + 2048 * (j / 63), (j % 63 + 1) * 4096, 0);
    };
    int x[2][64]; // the loop-carried input: position ap of set s holds row a = uinv * ap mod 64 of column m = tid + 512 s
    int tw[TWW];
    {
        const __amdgpu_buffer_rsrc_t xr = in_rsrc(blockIdx.x, true);
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int ap = 0; ap < 64; ap++)
                x[s][ap] = (int)__builtin_amdgcn_raw_buffer_load_b32(xr, voff + 2048 * s, (int)((((unsigned)a.uinv * ap) & 63u) << 12), AUX_LD);
    }
    if (TWMODE == 0) {
#pragma unroll
        for (int i = 0; i < TWW; i++) tw[i] = tw_load(i);
    }
    // TWMODE 1: w^m and w^(8m) of both columns, balanced (|.| <= 32768), for the whole kernel
    int seed1[2], seed8[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
        seed1[s2] = TWMODE ? (int)__builtin_amdgcn_raw_buffer_load_b32(tr, voff + 2048 * s2, 1 * 4096, 0) : 0;
        seed8[s2] = TWMODE ? (int)__builtin_amdgcn_raw_buffer_load_b32(tr, voff + 2048 * s2, 8 * 4096, 0) : 0;
    }
    int acc = 0;
    lds_barrier();
    for (unsigned tr_i = blockIdx.x; tr_i < (unsigned)a.batch; tr_i += gridDim.x) {
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (size_t)tr_i * 65536u), 0, 65536 * 4, 0x00020000);
        const unsigned tr_next = tr_i + gridDim.x;
        const bool has_next = tr_next < (unsigned)a.batch;
        // a descriptor of zero records for the last round: its requests return 0 and move nothing
        const __amdgpu_buffer_rsrc_t xn = in_rsrc(has_next ? tr_next : tr_i, has_next);
        unsigned uinv = (unsigned)a.uinv;
        asm volatile("" : "+s"(uinv));
        u64 *const dts = DBG ? a.dbg + (((size_t)(tr_i / gridDim.x) * gridDim.x + blockIdx.x) * 8 + (tid >> 6)) * 16 : nullptr;
        A_STAMP(0);
        int v[2][64];
        int nx[2][64];
        auto request = [&](int lo, int hi) { // positions lo..hi-1 of the next transform (i = 64 s + ap)
#pragma unroll
            for (int i = lo; i < hi; i++) {
                if (SKEL == 1) nx[i / 64][i % 64] = acc + i;
                else nx[i / 64][i % 64] = (int)__builtin_amdgcn_raw_buffer_load_b32(xn, voff + 2048 * (i / 64), (int)(((uinv * (i % 64)) & 63u) << 12), AUX_LD);
            }
        };
        // tight product of two twiddles: |x|, |y| <= 32770 in, |result| <= 32770 out
        auto tw_tight = [&](int x, int y) { return fm_bfold(fm_fold(fm_mulc(x, y))); };
        auto tw1_range = [&](int s, int lo, int hi) {
            if (TWMODE == 1) {
                // T(k0 = 8 kh + kl) = A[kh] * B[kl], A[i] = w^(8 m i), B[j] = w^(m j): 12 tight products for the two progressions,
                // 49 loose ones (|T| <= 49154: the application  fold(bfold(x) * T)  stays below 2^31 for |bfold(x)| <= 40961)
                int A[8], B[8];
                A[1] = seed8[s];
                B[1] = seed1[s];
#pragma unroll
                for (int i = 2; i < 8; i++) {
                    A[i] = tw_tight(A[i - 1], seed8[s]);
                    B[i] = tw_tight(B[i - 1], seed1[s]);
                }
#pragma unroll
                for (int k0 = lo; k0 < hi; k0++) {
                    const int kh = k0 >> 3, kl = k0 & 7;
                    const int T = kh == 0 ? B[kl] : kl == 0 ? A[kh] : fm_bfold(fm_mulc(A[kh], B[kl]));
                    int &q = v[s][brev_c(k0, 6)];
                    q = fm_fold(fm_mulc(fm_bfold(q), T));
                }
                return;
            }
#pragma unroll
            for (int k0 = lo; k0 < hi; k0++) {
                const int j = 63 * s + k0 - 1;
                int &q = v[s][brev_c(k0, 6)];
                q = fm_mul_tw(q, tw[j % TWW]);
                if (!TWFAKE && j + TWW < 126) tw[j % TWW] = tw_load(j + TWW);
            }
        };
        // ---- network 0 (both columns) and the first twiddles ----
#pragma unroll
        for (int i = 0; i < 64; i++) v[0][i] = x[0][i];
        if (SKEL != 2) {
            fermat_net64_canon(v[0]);
            v[0][0] = fm_fold(v[0][0]);
        }
        if (PFPOS == 1 && PF >= 64) request(0, 64);
        if (SKEL != 2) tw1_range(0, 1, 64);
        A_STAMP(1);
#pragma unroll
        for (int i = 0; i < 64; i++) v[1][i] = x[1][i];
        if (SKEL != 2) {
            fermat_net64_canon(v[1]);
            v[1][0] = fm_fold(v[1][0]);
            tw1_range(1, 1, 64);
        }
        A_STAMP(2);
        if (PFPOS == 1 && PF >= 64) request(64, PF);
        else request(0, PF);
        // ---- exchange 1 + network 1 ----
        int w[4][32];
        auto net1 = [&](int h) {
            fermat_net32_fold(w[h]);
            w[h][0] = fm_fold(w[h][0]);
#pragma unroll
            for (int k1 = 1; k1 < 32; k1++) {
                int &q = w[h][brev_c(k1, 5)];
                q = fm_mul_tw(q, tw2l[k1 * 32 + r]);
            }
        };
        auto x1_write = [&](int q) {
            const int bo = (q & 1) * XW;
#pragma unroll
            for (int kl = 0; kl < 16; kl++) {
                e1w0[bo + kl * 1024] = v[0][brev_c(16 * q + kl, 6)];
                e1w1[bo + kl * 1024] = v[1][brev_c(16 * q + kl, 6)];
            }
        };
        auto x1_read = [&](int q) {
            const int bo = (q & 1) * XW;
#pragma unroll
            for (int bp = 0; bp < 32; bp++) w[q][bp] = e1r[bo + bp * 32];
        };
        if (SKEL != 2 && !DEEP) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                x1_write(q);
                if (q > 0) net1(q - 1);
                lds_barrier();
                x1_read(q);
                A_STAMP(3 + q);
            }
            net1(3);
        }
        if (SKEL != 2 && DEEP) {
            x1_write(0);
            lds_barrier();
#pragma unroll
            for (int q = 0; q < 4; q++) {
                x1_read(q);
                if (q < 3) x1_write(q + 1);
                if (q > 0) net1(q - 1);
                if (q < 3) lds_barrier();
                A_STAMP(3 + q);
            }
            net1(3);
        }
        A_STAMP(7);
        // ---- exchange 2 + network 2 + stores ----
        int z[4][32];
        auto net2 = [&](int h) {
            if (SKEL == 2) {
#pragma unroll
                for (int k2 = 0; k2 < 32; k2++) __builtin_amdgcn_raw_buffer_store_b32(v[h / 2][32 * (h % 2) + k2], yr, voff, (2048 * k2 + 512 * h) * 4, AUX_ST);
                return;
            }
            fermat_net32_fold(z[h]);
#pragma unroll
            for (int k2 = 0; k2 < 32; k2++) {
                const u32 o = fm_canon(z[h][brev_c(k2, 5)]);
                if (SKEL == 1) acc ^= (int)o;
                else __builtin_amdgcn_raw_buffer_store_b32(o, yr, voff, (2048 * k2 + 512 * h) * 4, AUX_ST);
            }
        };
        constexpr int NT = (128 - PF + 31) / 32; // tail chunks of (up to) 32 requests
        auto tail_after = [&](int h) {           // called after net2(h): chunk c follows net2(max(0, 4 - NT + c - TSHIFT))
#pragma unroll
            for (int c = 0; c < NT; c++) {
                const int hc = 4 - NT + c - TSHIFT < 0 ? 0 : 4 - NT + c - TSHIFT;
                if (hc == h) {
                    if (TAILFAKE) {
                        const __amdgpu_buffer_rsrc_t xz = in_rsrc(tr_i, false); // zero records: the request returns 0 and moves nothing
#pragma unroll
                        for (int i = PF + 32 * c; i < (PF + 32 * c + 32 < 128 ? PF + 32 * c + 32 : 128); i++)
                            nx[i / 64][i % 64] = (int)__builtin_amdgcn_raw_buffer_load_b32(xz, voff + 2048 * (i / 64), (int)(((uinv * (i % 64)) & 63u) << 12), AUX_LD);
                    } else request(PF + 32 * c, PF + 32 * c + 32 < 128 ? PF + 32 * c + 32 : 128);
                }
            }
        };
        if (SKEL == 2) {
#pragma unroll
            for (int q = 0; q < 4; q++) { net2(q); tail_after(q); }
        } else {
            auto x2_write = [&](int q) {
                const int bo = (q & 1) * XW;
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int kl = 0; kl < 8; kl++) e2w[bo + kl * (64 * E2_PITCH) + 16 * i * E2_PITCH] = w[i][brev_c(8 * q + kl, 5)];
            };
            auto x2_read = [&](int q) {
                const int bo = (q & 1) * XW;
#pragma unroll
                for (int rp = 0; rp < 32; rp++) z[q][rp] = e2r[bo + rp];
            };
            if (!DEEP) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    x2_write(q);
                    if (q > 0) {
                        net2(q - 1);
                        tail_after(q - 1);
                    }
                    lds_barrier();
                    x2_read(q);
                    A_STAMP(8 + q);
                }
                net2(3);
            } else {
                x2_write(0);
                lds_barrier();
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    x2_read(q);
                    if (q < 3) x2_write(q + 1);
                    if (q > 0) {
                        net2(q - 1);
                        tail_after(q - 1);
                    }
                    if (q < 3) lds_barrier();
                    A_STAMP(8 + q);
                }
                net2(3);
            }
        }
        A_STAMP(12);
        // the next round's first-twiddle window goes ahead of the tail requests in the memory queue (loads return in order)
        if (TWMODE == 0) {
#pragma unroll
            for (int i = 0; i < TWW; i++) tw[i] = tw_load(i);
        }
        tail_after(3);
        A_STAMP(13);
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int i = 0; i < 64; i++) x[s][i] = nx[s][i];
    }
    if (SKEL == 1 && acc == 0x7fffffff) a.out[tid] = (u32)acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// VARIANT B: the product's 1024-thread x 64-point kernel (gfa_ntt_fermat.hip) with the first twiddles formed in registers
// (TWMODE 1) instead of streamed from the 256 KiB table, phase boundaries as never-taken branches (SCHEDB 2), the last round's
// requests through a zero-record descriptor.  BEARLY: the next transform's first BEARLY loads are requested before exchange 2.
#ifndef BEARLY
#define BEARLY 0
#endif
#ifndef DFS
#define DFS 0 // 1: network 0 in two parts (layers 1-3 on the 40 positions = 0..4 mod 8 first); those positions are the ones requested ahead
#endif
#ifndef X2MODE
#define X2MODE 0 // 1: exchange 2 in rounds by first-network half (w[0] travels under net1(1), w[1] under net2(0)) instead of by k1 half
#endif
#ifndef BE1
#define BE1 0 // requests [0, BE1) after the second exchange-1 write burst (the 64 point registers are free from there)
#endif
#ifndef BE3
#define BE3 BEARLY // requests [BEARLY, BE3) after the second exchange-2 write burst
#endif
#ifndef BE4
#define BE4 (BE3 > 32 ? BE3 : 32) // requests [BE3, BE4) after the first half's stores; the rest after the second half's
#endif
// request order: the 40 positions = 0..4 (mod 8) first, then the 24 positions = 5..7 (mod 8)
constexpr int dfs_order(int i) { return DFS ? (i < 40 ? (i / 5) * 8 + i % 5 : ((i - 40) / 3) * 8 + 5 + (i - 40) % 3) : i; }
constexpr int B_EX_WORDS = 16 * 64 * E2_PITCH;
constexpr int B_LDS_BYTES = (B_EX_WORDS + 1024) * 4;
#define B_SPLIT()                                                   \
    do {                                                            \
        if (SCHEDB == 1) __builtin_amdgcn_sched_barrier(0);         \
        if (SCHEDB == 2) {                                          \
            if (a.dbg != nullptr) asm volatile("s_nop 0");          \
        }                                                           \
    } while (0)
__global__ __launch_bounds__(1024) void fermat_b_kernel(FermatArgs a)
{
    extern __shared__ int lds[];
    int *ex = lds;
    int *tw2l = lds + B_EX_WORDS;
    const unsigned tid = threadIdx.x;
    const int voff = (int)(tid * 4u);
    const int g = (int)(tid >> 5), r = (int)(tid & 31);
    const int l = (int)(tid & 63), wv = (int)(tid >> 6);
    const int wpos1 = (((a.u * g) & 31) << 5) + r;
    const int wpos2 = ((a.u * r) & 31);
    int *const e1w = ex + wpos1;
    const int *const e1r = ex + g * 1024 + r;
    int *const e2w = ex + g * E2_PITCH + wpos2;
    const int *const e2r = ex + wv * (64 * E2_PITCH) + l * E2_PITCH;
    const __amdgpu_buffer_rsrc_t tr = __builtin_amdgcn_make_buffer_rsrc((void *)a.tw1, 0, 65536 * 4, 0x00020000);
    tw2l[tid] = a.tw2[tid];
    if (a.stagger > 0) {
        const int grp = (int)((blockIdx.x >> 3) & 3u);
        for (int i = 0; i < grp * a.stagger; i++) __builtin_amdgcn_s_sleep(64);
    }
    auto in_rsrc = [&](unsigned t, bool live) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + (size_t)t * 65536u), 0, live ? 65536 * 4 : 0, 0x00020000);
    };
    int v[64];
    {
        const __amdgpu_buffer_rsrc_t xr = in_rsrc(blockIdx.x, true);
#pragma unroll
        for (int ap = 0; ap < 64; ap++) v[ap] = (int)__builtin_amdgcn_raw_buffer_load_b32(xr, voff, (int)((((unsigned)a.uinv * ap) & 63u) << 12), AUX_LD);
    }
    const int seed1 = TWMODE ? (int)__builtin_amdgcn_raw_buffer_load_b32(tr, voff, 1 * 4096, 0) : 0;
    const int seed8 = TWMODE ? (int)__builtin_amdgcn_raw_buffer_load_b32(tr, voff, 8 * 4096, 0) : 0;
    for (unsigned tr_i = blockIdx.x; tr_i < (unsigned)a.batch; tr_i += gridDim.x) {
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (size_t)tr_i * 65536u), 0, 65536 * 4, 0x00020000);
        const unsigned tr_next = tr_i + gridDim.x;
        const bool has_next = tr_next < (unsigned)a.batch;
        const __amdgpu_buffer_rsrc_t xn = in_rsrc(has_next ? tr_next : tr_i, has_next);
        unsigned uinv = (unsigned)a.uinv;
        asm volatile("" : "+s"(uinv));
        constexpr int TW = TWMODE ? 1 : TWW;
        int tw[TW];
        if (!TWMODE) {
#pragma unroll
            for (int i = 0; i < TW; i++) tw[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(tr, voff, (i + 1) * 4096, 0);
        }
        auto breq = [&](int lo, int hi) {
#pragma unroll
            for (int i = lo; i < hi; i++) {
                const int ap = dfs_order(i);
                v[ap] = (int)__builtin_amdgcn_raw_buffer_load_b32(xn, voff, (int)(((uinv * (unsigned)ap) & 63u) << 12), AUX_LD);
            }
        };
#if DFS
        fermat_net64_canon_head(v);
        B_SPLIT();
        fermat_net64_canon_tail(v);
#else
        fermat_net64_canon(v);
#endif
        B_SPLIT();
        v[0] = fm_fold(v[0]);
        int A[8], B[8];
        auto tw_tight = [&](int x, int y) { return fm_bfold(fm_fold(fm_mulc(x, y))); };
        if (TWMODE) {
            A[1] = seed8;
            B[1] = seed1;
#pragma unroll
            for (int i = 2; i < 8; i++) {
                A[i] = tw_tight(A[i - 1], seed8);
                B[i] = tw_tight(B[i - 1], seed1);
            }
        }
        auto tw1_range = [&](int lo, int hi) {
#pragma unroll
            for (int k0 = lo; k0 < hi; k0++) {
                int &q = v[brev_c(k0, 6)];
                if (TWMODE) {
                    const int kh = k0 >> 3, kl = k0 & 7;
                    const int T = kh == 0 ? B[kl] : kl == 0 ? A[kh] : fm_bfold(fm_mulc(A[kh], B[kl]));
                    q = fm_fold(fm_mulc(fm_bfold(q), T));
                } else {
                    q = fm_mul_tw(q, tw[(k0 - 1) % TW]);
                    if (k0 + TW < 64) tw[(k0 - 1) % TW] = (int)__builtin_amdgcn_raw_buffer_load_b32(tr, voff, (k0 + TW) * 4096, 0);
                }
            }
        };
        tw1_range(1, 32);
        B_SPLIT();
        int w[2][32];
        lds_barrier();
#pragma unroll
        for (int kl = 0; kl < 32; kl++) e1w[kl * 1024] = v[brev_c(kl, 6)];
        tw1_range(32, 64);
        lds_barrier();
#pragma unroll
        for (int bp = 0; bp < 32; bp++) w[0][bp] = e1r[bp * 32];
        lds_barrier();
#pragma unroll
        for (int kl = 0; kl < 32; kl++) e1w[kl * 1024] = v[brev_c(kl + 32, 6)];
        B_SPLIT();
        breq(0, BE1);
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
        B_SPLIT();
        int z[2][32];
        if (X2MODE == 1) {
            // round h: [k1][k0 local = g][r'] of w[h] (k0 = g + 32 h), 32 x 32 rows of pitch 33; reader (k0 local = l & 31, k1 = (l >> 5) + 2 wv)
            int *const xw = ex + g * E2_PITCH + wpos2;
            const int *const xr = ex + ((l >> 5) + 2 * wv) * (32 * E2_PITCH) + (l & 31) * E2_PITCH;
            const int soff = (int)((((unsigned)l & 31u) + 64u * (((unsigned)l >> 5) + 2u * (unsigned)wv)) * 4u);
            auto net2x = [&](int h) {
                fermat_net32_fold(z[h]);
#pragma unroll
                for (int k2 = 0; k2 < 32; k2++)
                    __builtin_amdgcn_raw_buffer_store_b32(fm_canon(z[h][brev_c(k2, 5)]), yr, soff, (2048 * k2 + 32 * h) * 4, AUX_ST);
            };
            lds_barrier(); // every thread has its exchange-1 rows
#pragma unroll
            for (int k1 = 0; k1 < 32; k1++) xw[k1 * (32 * E2_PITCH)] = w[0][brev_c(k1, 5)];
            breq(BE1, BEARLY);
            net1(1);
            B_SPLIT();
            lds_barrier();
#pragma unroll
            for (int rp = 0; rp < 32; rp++) z[0][rp] = xr[rp];
            lds_barrier();
#pragma unroll
            for (int k1 = 0; k1 < 32; k1++) xw[k1 * (32 * E2_PITCH)] = w[1][brev_c(k1, 5)];
            B_SPLIT();
            breq(BEARLY, BE3);
            net2x(0);
            B_SPLIT();
            breq(BE3, BE4);
            lds_barrier();
#pragma unroll
            for (int rp = 0; rp < 32; rp++) z[1][rp] = xr[rp];
            B_SPLIT();
            net2x(1);
        } else {
        breq(BE1, BEARLY);
        net1(1);
        B_SPLIT();
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int kl = 0; kl < 16; kl++) e2w[kl * (64 * E2_PITCH) + 32 * i * E2_PITCH] = w[i][brev_c(kl, 5)];
        lds_barrier();
#pragma unroll
        for (int rp = 0; rp < 32; rp++) z[0][rp] = e2r[rp];
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int kl = 0; kl < 16; kl++) e2w[kl * (64 * E2_PITCH) + 32 * i * E2_PITCH] = w[i][brev_c(kl + 16, 5)];
        B_SPLIT();
        breq(BEARLY, BE3);
        auto net2 = [&](int h) {
            fermat_net32_fold(z[h]);
#pragma unroll
            for (int k2 = 0; k2 < 32; k2++)
                __builtin_amdgcn_raw_buffer_store_b32(fm_canon(z[h][brev_c(k2, 5)]), yr, voff, (2048 * k2 + 1024 * h) * 4, AUX_ST);
        };
        net2(0);
        B_SPLIT();
        breq(BE3, BE4);
        lds_barrier();
#pragma unroll
        for (int rp = 0; rp < 32; rp++) z[1][rp] = e2r[rp];
        B_SPLIT();
        net2(1);
        }
        breq(BE4, 64);
        B_SPLIT();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
inline u32 mulmod(u32 a, u32 b) { return (u32)(((u64)a * b) % 65537u); }
inline int balanced(u32 c) { return c > 32768u ? (int)c - 65537 : (int)c; }

__global__ void fill_kernel(u32 *p, size_t n, u32 seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u64 h = (i + seed) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        u32 v = (u32)(h % 65537u);
        if ((h >> 40) % 1000 == 0) v = 65536u; // the value that needs 17 bits, often
        p[i] = v;
    }
}

} // namespace

int main(int argc, char **argv)
{
    const int check_batch = argc > 1 ? atoi(argv[1]) : 600;
    std::vector<int> time_batches;
    for (int i = 2; i < argc; i++) time_batches.push_back(atoi(argv[i]));
    if (time_batches.empty()) time_batches = {1024, 4096};
    int max_batch = check_batch;
    for (int b : time_batches) max_batch = b > max_batch ? b : max_batch;
    const u32 omega = 3; // 3 generates GF(65537)*
    HIPCHK(hipSetDevice(0));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;

    u32 w64 = omega;
    for (int i = 0; i < 10; i++) w64 = mulmod(w64, w64);
    u32 z = 4080u, zz = mulmod(z, z), cur = z;
    int u = 0;
    for (int c = 1; c < 64; c += 2) {
        if (cur == w64) { u = c; break; }
        cur = mulmod(cur, zz);
    }
    if (!u) { fprintf(stderr, "no u\n"); return 2; }
    int uinv = 1;
    while ((u * uinv) % 64 != 1) uinv += 2;
    std::vector<int> t1(64 * 1024), t2(32 * 32);
    std::vector<u32> pw(65536);
    pw[0] = 1;
    for (int e = 1; e < 65536; e++) pw[e] = mulmod(pw[e - 1], omega);
    for (int k0 = 0; k0 < 64; k0++)
        for (int m = 0; m < 1024; m++) t1[k0 * 1024 + m] = balanced(pw[(m * k0) & 65535]);
    for (int k1 = 0; k1 < 32; k1++)
        for (int rr = 0; rr < 32; rr++) t2[k1 * 32 + rr] = balanced(pw[(64 * rr * k1) & 65535]);
    int *d_t1, *d_t2;
    HIPCHK(hipMalloc((void **)&d_t1, t1.size() * 4));
    HIPCHK(hipMalloc((void **)&d_t2, t2.size() * 4));
    HIPCHK(hipMemcpy(d_t1, t1.data(), t1.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_t2, t2.data(), t2.size() * 4, hipMemcpyHostToDevice));

    const size_t words = (size_t)max_batch * 65536;
    u32 *d_in, *d_out, *d_ref;
    HIPCHK(hipMalloc((void **)&d_in, words * 4));
    HIPCHK(hipMalloc((void **)&d_out, words * 4));
    HIPCHK(hipMalloc((void **)&d_ref, (size_t)check_batch * 65536 * 4));
    fill_kernel<<<4096, 256>>>(d_in, words, 12345u);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipFuncSetAttribute((const void *)fermat_a_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)fermat_a_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)fermat_b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    gfa_field_t *f = nullptr;
    if (gfa_field_create(65537, 1, nullptr, 3, &f) != GFA_OK) { fprintf(stderr, "field\n"); return 2; }
    auto launch = [&](int batch, u64 *dbg) {
        const int grid = batch < cus ? batch : cus;
        FermatArgs a{d_in, d_out, d_t1, d_t2, u, uinv, batch, batch >= 2 * grid ? STAGGER : 0, dbg};
        if (VARB) hipLaunchKernelGGL(fermat_b_kernel, dim3(grid), dim3(1024), B_LDS_BYTES, 0, a);
        else if (dbg) hipLaunchKernelGGL((fermat_a_kernel<true>), dim3(grid), dim3(512), A_LDS_BYTES, 0, a);
        else hipLaunchKernelGGL((fermat_a_kernel<false>), dim3(grid), dim3(512), A_LDS_BYTES, 0, a);
    };
    printf("config: DFS=%d X2MODE=%d VARB=%d BE1=%d BEARLY=%d BE3=%d BE4=%d TWMODE=%d DEEP=%d PF=%d TWW=%d PFPOS=%d TSHIFT=%d SKEL=%d STAGGER=%d AUX_LD=%d AUX_ST=%d PRIO=%d cus=%d\n", DFS, X2MODE, VARB, BE1, BEARLY, BE3, BE4, TWMODE, DEEP, PF, TWW, PFPOS, TSHIFT, SKEL, STAGGER, AUX_LD, AUX_ST, PRIO, cus);
    // ---- check against the product kernel ----
    if (SKEL == 0 && check_batch > 0) {
        if (gfa_ntt(f, d_in, d_ref, 65536, check_batch, omega, 0, GFA_U32, nullptr) != GFA_OK) { fprintf(stderr, "gfa_ntt\n"); return 2; }
        HIPCHK(hipMemset(d_out, 0xff, (size_t)check_batch * 65536 * 4));
        launch(check_batch, nullptr);
        HIPCHK(hipDeviceSynchronize());
        std::vector<u32> ho((size_t)check_batch * 65536), hr((size_t)check_batch * 65536);
        HIPCHK(hipMemcpy(ho.data(), d_out, ho.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hr.data(), d_ref, hr.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < ho.size(); i++)
            if (ho[i] != hr[i]) { if (!bad) first = i; bad++; }
        printf("check batch %d: %s (%zu mismatches%s)\n", check_batch, bad ? "FAIL" : "ok", bad, bad ? "" : "");
        if (bad) printf("  first at transform %zu index %zu: got %u want %u\n", first >> 16, first & 65535, ho[first], hr[first]);
    }
    // ---- timing ----
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int batch : time_batches) {
        for (int rep = 0; rep < 2; rep++) {
            // clock pre-warm: ~60 ms of the same launch
            for (int i = 0; i < (batch >= 4096 ? 100 : 400); i++) launch(batch, nullptr);
            HIPCHK(hipDeviceSynchronize());
            const int iters = 20;
            HIPCHK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; i++) launch(batch, nullptr);
            HIPCHK(hipEventRecord(e1, 0));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters;
            const double pts = (double)batch * 65536;
            printf("batch %5d: %.4f ms  frac %.3f  %.1f us per round of %d\n", batch, ms, 8 * pts / ms / 1e6 / 8000, ms * 1e3 / ((batch + cus - 1) / cus), cus);
        }
    }
    // ---- stamps (lane 0 of every wave of every workgroup), batch 1024 ----
    if (SKEL != 2 && !VARB) {
        const int batch = 1024;
        u64 *d_dbg;
        HIPCHK(hipMalloc((void **)&d_dbg, (size_t)batch * 8 * 16 * 8));
        HIPCHK(hipMemset(d_dbg, 0, (size_t)batch * 8 * 16 * 8));
        for (int i = 0; i < 50; i++) launch(batch, nullptr);
        launch(batch, d_dbg);
        HIPCHK(hipDeviceSynchronize());
        std::vector<u64> h((size_t)batch * 8 * 16);
        HIPCHK(hipMemcpy(h.data(), d_dbg, h.size() * 8, hipMemcpyDeviceToHost));
        const int grid = cus;
        const char *names[14] = {"top", "net0a+tw", "net0b+tw", "X1r0", "X1r1", "X1r2", "X1r3", "net1(3)", "X2r0", "X2r1", "X2r2", "X2r3", "net2(3)", "tw+tail"};
        auto med = [](std::vector<double> &d) { std::sort(d.begin(), d.end()); return d[d.size() / 2]; };
        for (int rnd = 0; rnd < batch / grid; rnd++) {
            printf("round %d wave 0 (median us):", rnd);
            for (int i = 0; i < 13; i++) {
                std::vector<double> d;
                for (int b = 0; b < grid; b++) {
                    const u64 *p = &h[(((size_t)rnd * grid + b) * 8 + 0) * 16];
                    d.push_back((double)(p[i + 1] - p[i]) / 100.0);
                }
                printf("  %s %.2f", names[i + 1], med(d));
            }
            std::vector<double> tot;
            for (int b = 0; b < grid; b++) {
                const u64 *p = &h[(((size_t)rnd * grid + b) * 8 + 0) * 16];
                tot.push_back((double)(p[13] - p[0]) / 100.0);
            }
            printf("  | total %.2f\n", med(tot));
        }
        // per wave, round 2: stamp times relative to the workgroup's earliest 'top' of that round (median over workgroups)
        for (int rnd = 1; rnd < 3; rnd++) {
            printf("round %d, every wave: stamp - min over waves of top (median us over workgroups)\n", rnd);
            for (int wvi = 0; wvi < 8; wvi++) {
                printf("  wave %d:", wvi);
                for (int i = 0; i < 14; i++) {
                    std::vector<double> d;
                    for (int b = 0; b < grid; b++) {
                        u64 t0 = ~0ull;
                        for (int w2 = 0; w2 < 8; w2++) t0 = std::min(t0, h[(((size_t)rnd * grid + b) * 8 + w2) * 16]);
                        d.push_back((double)(h[(((size_t)rnd * grid + b) * 8 + wvi) * 16 + i] - t0) / 100.0);
                    }
                    printf(" %6.2f", med(d));
                }
                printf("\n");
            }
        }
    }
    return 0;
}
