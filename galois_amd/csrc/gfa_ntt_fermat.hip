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
//     w^(m*k0) comes from a 256 KiB table that every workgroup shares (L2 resident), w_1024^(r*k1) from a 4 KiB LDS table;
//   * the networks use the canonical roots (sqrt(2), 2).  A transform with root of unity w has w^(N/64) = sqrt(2)^u for
//     one odd u; feeding network inputs in the order a' = u*a mod R turns the canonical network into the wanted one, and
//     that permutation is folded into the global load addresses and the LDS write positions (no instructions).
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
    const int *tw1; // [64][1024]: balanced w^(m * k0)
    const int *tw2; // [32][32]:   balanced w^(64 * r * k1), index k1 * 32 + r
    int u, uinv;    // w^(N/64) == sqrt(2)^u, uinv = u^-1 mod 64
    int negate;     // inverse transform of length 2^16: the scale 1/N == -1
};

__global__ __launch_bounds__(1024) void ntt_fermat16_kernel(FermatArgs a)
{
    extern __shared__ int lds[];
    int *ex = lds;
    int *tw2l = lds + EX_WORDS;
    const int tid = threadIdx.x;
    const u32 *x = a.in + (size_t)blockIdx.x * 65536u;
    u32 *y = a.out + (size_t)blockIdx.x * 65536u;
    tw2l[tid] = a.tw2[tid];

    // ---- network 0: radix 64 over a (stride 1024); thread m = tid ----
    int v[64];
#pragma unroll
    for (int ap = 0; ap < 64; ap++) v[ap] = (int)x[(((a.uinv * ap) & 63) << 10) + tid];
    fermat_net64_canon(v);
    {
        const int *t1 = a.tw1 + tid;
#pragma unroll
        for (int k0 = 0; k0 < 64; k0++) {
            int &r = v[brev_c(k0, 6)];
            r = (k0 == 0) ? fm_fold(r) : fm_mul_tw(r, t1[k0 * 1024]);
        }
    }
    // ---- exchange 1 + network 1: thread (g, r) takes k0 = g and g + 32, radix 32 over b (m = 32 b + r) ----
    const int g = tid >> 5, r = tid & 31;
    const int wpos1 = (((a.u * g) & 31) << 5) + r; // slot b' = u * b mod 32
    int w[2][32];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (h) __syncthreads();
#pragma unroll
        for (int kl = 0; kl < 32; kl++) ex[kl * 1024 + wpos1] = v[brev_c(kl + 32 * h, 6)];
        __syncthreads();
#pragma unroll
        for (int bp = 0; bp < 32; bp++) w[h][bp] = ex[g * 1024 + bp * 32 + r];
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        fermat_net32_fold(w[h]);
#pragma unroll
        for (int k1 = 0; k1 < 32; k1++) {
            int &q = w[h][brev_c(k1, 5)];
            q = (k1 == 0) ? fm_fold(q) : fm_mul_tw(q, tw2l[k1 * 32 + r]);
        }
    }
    // ---- exchange 2 + network 2: thread (lane l = k0, wave wv) takes k1 = wv and wv + 16, radix 32 over r ----
    const int l = tid & 63, wv = tid >> 6;
    const int wpos2 = ((a.u * r) & 31); // slot r' = u * r mod 32
    int z[2][32];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int kl = 0; kl < 16; kl++) ex[kl * (64 * E2_PITCH) + (g + 32 * i) * E2_PITCH + wpos2] = w[i][brev_c(kl + 16 * h, 5)];
        __syncthreads();
#pragma unroll
        for (int rp = 0; rp < 32; rp++) z[h][rp] = ex[wv * (64 * E2_PITCH) + l * E2_PITCH + rp];
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        fermat_net32_fold(z[h]);
        u32 *yo = y + l + 64 * (wv + 16 * h);
#pragma unroll
        for (int k2 = 0; k2 < 32; k2++) {
            int c = z[h][brev_c(k2, 5)];
            if (a.negate) c = fm_sub(0, c);
            c = fm_fold(fm_fold(c));                 // [-1, 65536]
            yo[2048 * k2] = min((u32)c, 65536u);     // -1 == 65536
        }
    }
}

struct FermatPlan {
    int *tw1 = nullptr, *tw2 = nullptr;
    int u = 0, uinv = 0;
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
    static const int min_batch = [] { const char *e = getenv("GFA_NTT_FERMAT_MIN_BATCH"); return e ? atoi(e) : 64; }();
    return fd.kind == KIND_PRIME32 && fd.p == 65537 && n == 65536 && batch >= min_batch;
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
            std::vector<int> t1(64 * 1024), t2(32 * 32);
            std::vector<u32> pw(65536);
            pw[0] = 1;
            for (int e = 1; e < 65536; e++) pw[e] = mulmod(pw[e - 1], (u32)omega);
            for (int k0 = 0; k0 < 64; k0++)
                for (int m = 0; m < 1024; m++) t1[k0 * 1024 + m] = balanced(pw[(m * k0) & 65535]);
            for (int k1 = 0; k1 < 32; k1++)
                for (int r = 0; r < 32; r++) t2[k1 * 32 + r] = balanced(pw[(64 * r * k1) & 65535]);
            GFA_HIP(hipMalloc((void **)&p.tw1, t1.size() * sizeof(int)));
            GFA_HIP(hipMalloc((void **)&p.tw2, t2.size() * sizeof(int)));
            GFA_HIP(hipMemcpy(p.tw1, t1.data(), t1.size() * sizeof(int), hipMemcpyHostToDevice));
            GFA_HIP(hipMemcpy(p.tw2, t2.data(), t2.size() * sizeof(int), hipMemcpyHostToDevice));
            p.u = u; p.uinv = uinv; p.ok = true;
            GFA_HIP(hipFuncSetAttribute((const void *)ntt_fermat16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        pl = p;
    }
    FermatArgs a{(const u32 *)in, (u32 *)out, pl.tw1, pl.tw2, pl.u, pl.uinv, negate};
    hipLaunchKernelGGL(ntt_fermat16_kernel, dim3((unsigned)batch), dim3(1024), FERMAT_LDS_BYTES, st, a);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

} // namespace gfa
