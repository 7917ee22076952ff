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
constexpr int EX_WORDS = 16 * 64 * E2_PITCH;  // 33792 words >= exchange 1's 32 * 1024
constexpr int FERMAT_LDS_BYTES = (EX_WORDS + 1024) * 4;

struct FermatArgs {
    const u32 *in;
    u32 *out;
    const int *tw1; // [2][1024]: balanced w^m and w^(8 m), the seeds of the first twiddles
    const int *tw2; // [32][32]:   balanced w^(64 * r * k1), index k1 * 32 + r
    int u, uinv;    // w^(N/64) == sqrt(2)^u, uinv = u^-1 mod 64
    int batch;
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

template <bool NEGATE, bool DBG>
__global__ __launch_bounds__(1024) void ntt_fermat16_kernel(FermatArgs a)
{
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
    int *const e2w = ex + g * E2_PITCH + wpos2;
    const int *const e2r = ex + wv * (64 * E2_PITCH) + l * E2_PITCH;
    tw2l[tid] = a.tw2[tid];
    const int seed1 = a.tw1[tid], seed8 = a.tw1[1024 + tid]; // w^m, w^(8 m): the whole kernel
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
    auto in_rsrc = [&](unsigned t, bool live) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + (size_t)t * 65536u), 0, live ? 65536 * 4 : 0, 0x00020000);
    };
    int v[64];
    {
        const __amdgpu_buffer_rsrc_t xr = in_rsrc(blockIdx.x, true);
#pragma unroll
        for (int ap = 0; ap < 64; ap++) v[ap] = (int)__builtin_amdgcn_raw_buffer_load_b32(xr, voff, (int)((((unsigned)a.uinv * ap) & 63u) << 12), AUX_NT);
    }
    for (unsigned tr_i = blockIdx.x; tr_i < (unsigned)a.batch; tr_i += gridDim.x) {
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (size_t)tr_i * 65536u), 0, 65536 * 4, 0x00020000);
        const unsigned tr_next = tr_i + gridDim.x;
        const bool has_next = tr_next < (unsigned)a.batch;
        const __amdgpu_buffer_rsrc_t xn = in_rsrc(has_next ? tr_next : tr_i, has_next);
        unsigned uinv = (unsigned)a.uinv;
        asm volatile("" : "+s"(uinv)); // keep the 64 row offsets out of loop-invariant SGPRs (they are 3 scalar ops each)
        unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // phase timestamps: scalar registers, written out once at the end
        auto request = [&](int lo, int hi) { // rows lo .. hi-1 of the next transform, into the point registers it will start from
#pragma unroll
            for (int ap = lo; ap < hi; ap++) v[ap] = (int)__builtin_amdgcn_raw_buffer_load_b32(xn, voff, (int)(((uinv * ap) & 63u) << 12), AUX_NT);
        };
        FM_PHASE(0);
        // ---- network 0: radix 64 over a (stride 1024); thread m = tid.  Twiddles w^(m * k0), k0 = 1..63: gfa_fermat_tw.h ----
        fermat_net64_canon(v);
        FM_PHASE(1);
        v[0] = fm_fold(v[0]);
        int twa[8], twb[8];
        fm_tw_progressions(seed1, seed8, twa, twb);
        auto tw1_range = [&](int lo, int hi) {
#pragma unroll
            for (int k0 = lo; k0 < hi; k0++) {
                int &q = v[brev_c(k0, 6)];
                q = fm_tw_apply(q, fm_tw_of(twa, twb, k0));
            }
        };
        tw1_range(1, 32);
        FM_PHASE(2);
        // ---- exchange 1 + network 1: thread (g, r) takes k0 = g and g + 32, radix 32 over b (m = 32 b + r).  Each LDS
        // write burst is followed by arithmetic that does not depend on it, so the LDS pipe and the VALU overlap ----
        int w[2][32];
        lds_barrier(); // the previous transform's exchange-2 reads (first round: the staging of tw2l) are complete
#pragma unroll
        for (int kl = 0; kl < 32; kl++) e1w[kl * 1024] = v[brev_c(kl, 6)];
        tw1_range(32, 64);
        lds_barrier();
#pragma unroll
        for (int bp = 0; bp < 32; bp++) w[0][bp] = e1r[bp * 32];
        lds_barrier();
#pragma unroll
        for (int kl = 0; kl < 32; kl++) e1w[kl * 1024] = v[brev_c(kl + 32, 6)];
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
        request(GFA_FERMAT_E1, GFA_FERMAT_E2);
        net1(1);
        FM_PHASE(4);
        // ---- exchange 2 + network 2: thread (lane l = k0, wave wv) takes k1 = wv and wv + 16, radix 32 over r ----
        int z[2][32];
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
        FM_PHASE(5);
        auto net2 = [&](int h) {
            fermat_net32_fold(z[h]);
            // X[k0 + 64 * (k1 + 32 * k2)], k1 = wv + 16 h: lane offset tid
#pragma unroll
            for (int k2 = 0; k2 < 32; k2++)
                __builtin_amdgcn_raw_buffer_store_b32(fm_canon<NEGATE>(z[h][brev_c(k2, 5)]), yr, voff, (2048 * k2 + 1024 * h) * 4, AUX_ST);
        };
        net2(0);
        FM_PHASE(6);
        lds_barrier();
#pragma unroll
        for (int rp = 0; rp < 32; rp++) z[1][rp] = e2r[rp];
        FM_SPLIT();
        net2(1);
        request(GFA_FERMAT_E2, 64);
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
    int u = 0, uinv = 0, cus = 256;
    bool ok = false;
};
std::mutex g_mu;
std::map<std::pair<int, u64>, FermatPlan> g_plans; // (device, omega)

inline u32 mulmod(u32 a, u32 b) { return (u32)(((u64)a * b) % 65537u); }
inline int balanced(u32 c) { return c > 32768u ? (int)c - 65537 : (int)c; }

} // namespace

namespace gfa {

bool ntt_fermat16_eligible(const FieldDev &fd, i64 n, i64 batch)
{
    return fd.kind == KIND_PRIME32 && fd.p == 65537 && n == 65536 && batch >= 64; // one persistent workgroup per CU: fewer leave the chip idle
}

// in / out: uint32, batch transforms of 2^16 points.  Returns GFA_ERR_UNSUPPORTED (nothing launched) when omega is not a
// primitive 2^16-th root of unity.
int ntt_fermat16(const void *in, void *out, i64 batch, u64 omega, int negate, hipStream_t st)
{
    int dev = 0;
    GFA_HIP(hipGetDevice(&dev));
    FermatPlan pl;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        FermatPlan &p = g_plans[std::make_pair(dev, omega)];
        if (!p.tw1 && !p.ok) {
            // w must have order exactly 2^16: w^(2^15) == -1
            u32 t = (u32)omega;
            for (int i = 0; i < 15; i++) t = mulmod(t, t);
            if (t != 65536u) return GFA_ERR_UNSUPPORTED;
            u32 w64 = (u32)omega; // w^(N/64) = w^1024
            for (int i = 0; i < 10; i++) w64 = mulmod(w64, w64);
            u32 z = 4080u, zz = mulmod(z, z), cur = z; // sqrt(2)^u for odd u
            int u = 0;
            for (int c = 1; c < 64; c += 2) {
                if (cur == w64) { u = c; break; }
                cur = mulmod(cur, zz);
            }
            if (!u) return GFA_ERR_UNSUPPORTED;
            int uinv = 1;
            while ((u * uinv) % 64 != 1) uinv += 2;
            std::vector<int> t1(2 * 1024), t2(32 * 32);
            std::vector<u32> pw(65536);
            pw[0] = 1;
            for (int e = 1; e < 65536; e++) pw[e] = mulmod(pw[e - 1], (u32)omega);
            for (int m = 0; m < 1024; m++) {
                t1[m] = balanced(pw[m]);                      // w^m
                t1[1024 + m] = balanced(pw[(8 * m) & 65535]); // w^(8 m)
            }
            for (int k1 = 0; k1 < 32; k1++)
                for (int r = 0; r < 32; r++) t2[k1 * 32 + r] = balanced(pw[(64 * r * k1) & 65535]);
            GFA_HIP(hipMalloc((void **)&p.tw1, t1.size() * sizeof(int)));
            GFA_HIP(hipMalloc((void **)&p.tw2, t2.size() * sizeof(int)));
            GFA_HIP(hipMemcpy(p.tw1, t1.data(), t1.size() * sizeof(int), hipMemcpyHostToDevice));
            GFA_HIP(hipMemcpy(p.tw2, t2.data(), t2.size() * sizeof(int), hipMemcpyHostToDevice));
            hipDeviceProp_t prop;
            GFA_HIP(hipGetDeviceProperties(&prop, dev));
            p.u = u; p.uinv = uinv; p.cus = prop.multiProcessorCount; p.ok = true;
            GFA_HIP(hipFuncSetAttribute((const void *)ntt_fermat16_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            GFA_HIP(hipFuncSetAttribute((const void *)ntt_fermat16_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            GFA_HIP(hipFuncSetAttribute((const void *)ntt_fermat16_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        pl = p;
    }
    constexpr int stagger_env = 2; // first-round stagger of the four workgroup groups, in units of 4096 clocks (measured against 0 and 4)
    const i64 grid = std::min<i64>(batch, pl.cus);
    // the stagger only pays when every group has later rounds to keep busy
    FermatArgs a{(const u32 *)in, (u32 *)out, pl.tw1, pl.tw2, pl.u, pl.uinv, (int)batch, batch >= 2 * grid ? stagger_env : 0, g_fermat_dbg};
    if (a.dbg && !negate) hipLaunchKernelGGL((ntt_fermat16_kernel<false, true>), dim3((unsigned)grid), dim3(1024), FERMAT_LDS_BYTES, st, a);
    else if (negate) hipLaunchKernelGGL((ntt_fermat16_kernel<true, false>), dim3((unsigned)grid), dim3(1024), FERMAT_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((ntt_fermat16_kernel<false, false>), dim3((unsigned)grid), dim3(1024), FERMAT_LDS_BYTES, st, a);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// tuning aid (tools/fermat_phases.py): device buffer of 8 timestamps per (round, workgroup), or nullptr to switch off
extern "C" void gfa_debug_fermat_stamps(unsigned long long *buf) { g_fermat_dbg = buf; }

} // namespace gfa
