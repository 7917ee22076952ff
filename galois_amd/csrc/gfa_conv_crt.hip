// gfa_conv_crt.hip -- long polynomial products over ANY prime field GF(p), p < 2^32, through number-theoretic transforms.
//
// convolve_jit (_domains/_function.py:111-167) computes c_k = sum_{i+j=k} a_i b_j in GF(p); for prime fields the reference
// itself takes the integer route `np.convolve(a, b) % p` (_function.py:141-150).  The integer convolution of two sequences
// with entries below p has terms below min(na, nb) * (p-1)^2, so it is recovered EXACTLY from its residues modulo three
// NTT-friendly 31-bit primes (product ~ 2^90.6) by the Chinese remainder theorem, and then reduced modulo p.  Each residue
// convolution is two forward transforms, a pointwise product and one inverse transform on the library's own NTT kernels
// (gfa_ntt.hip).  The result is the same exact polynomial product the direct O(na*nb) kernel gives -- 1.1e12 multiply-adds
// for two 2^20-term inputs against nine 2^21-point transforms.
#include "gfa_internal.h"
#include "gfa_karatsuba.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>

using namespace gfa;

namespace {

// P_i = c_i * 2^k_i + 1 with primitive roots g_i.  Two sets:
//   set 0: three 31-bit primes (product 2^90.6), 2^26 divides every P_i - 1 -- any p < 2^32, products up to 2^26 terms;
//   set 1 (r05): three primes BELOW 2^29 (product 2^84.6), 2^23 divides every P_i - 1.  Their transforms run on the signed-Montgomery
//          kernels (gfa_ntt_m32.hip: 40 instead of 56 vector instructions per point and pass, memory-bound), so whenever the
//          coefficient bound  min(na, nb) * (p - 1)^2  fits below their product and the transform below 2^23 points, this is the set.
struct CrtSet {
    u64 P[3], G[3];
    int max_log;
};
constexpr CrtSet CRT_SETS[2] = {{{2013265921ull, 469762049ull, 1811939329ull}, {31ull, 3ull, 13ull}, 26},
                                {{469762049ull, 377487361ull, 167772161ull}, {3ull, 7ull, 3ull}, 23}};
constexpr int CRT_MAX_LOG = 26;
// extension fields take the plane route from na * nb >= 2^GFA_CONV_PLANES_MIN_LOG (environment, read once; default 20; 62 switches it off)
inline int planes_min_log()
{
    static const int v = [] { const char *e = getenv("GFA_CONV_PLANES_MIN_LOG"); const int x = e ? atoi(e) : 20; return x < 10 ? 10 : (x > 62 ? 62 : x); }();
    return v;
}

u64 host_powmod(u64 b, u64 e, u64 m)
{
    unsigned __int128 r = 1, x = b % m;
    while (e) {
        if (e & 1) r = r * x % m;
        x = x * x % m;
        e >>= 1;
    }
    return (u64)r;
}

std::mutex g_mu;
gfa_field *g_aux[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};

int aux_fields(int set, gfa_field **out)
{
    std::lock_guard<std::mutex> lock(g_mu);
    for (int i = 0; i < 3; i++) {
        if (!g_aux[set][i]) {
            int rc = gfa_field_create(CRT_SETS[set].P[i], 1, nullptr, CRT_SETS[set].G[i], &g_aux[set][i]);
            if (rc) return rc;
        }
        out[i] = g_aux[set][i];
    }
    return GFA_OK;
}

// every coefficient of the integer product must stay below P1*P2*P3
bool crt_set_fits(int set, u64 p, i64 lo, int lg)
{
    const long double bound = (long double)lo * (long double)(p - 1) * (long double)(p - 1);
    const long double M = (long double)CRT_SETS[set].P[0] * (long double)CRT_SETS[set].P[1] * (long double)CRT_SETS[set].P[2];
    return lg <= CRT_SETS[set].max_log && bound < M * 0.99L;
}

// buf[i][0][j] = a[j] mod P_i, buf[i][1][j] = b[j] mod P_i, zero padded to n_fft
template <typename T, int SET>
__global__ __launch_bounds__(256) void crt_spread_kernel(const T *__restrict__ a, i64 na, const T *__restrict__ b, i64 nb,
                                                         u32 *__restrict__ buf, i64 n_fft)
{
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n_fft; j += (i64)gridDim.x * blockDim.x) {
        const u64 av = j < na ? (u64)a[j] : 0, bv = j < nb ? (u64)b[j] : 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            buf[(2 * i) * n_fft + j] = (u32)(av % CRT_SETS[SET].P[i]);
            buf[(2 * i + 1) * n_fft + j] = (u32)(bv % CRT_SETS[SET].P[i]);
        }
    }
}

// buf[i][0][j] <- buf[i][0][j] * buf[i][1][j] mod P_i
template <int SET>
__global__ __launch_bounds__(256) void crt_pointwise_kernel(u32 *__restrict__ buf, i64 n_fft)
{
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n_fft; j += (i64)gridDim.x * blockDim.x) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const u64 x = buf[(2 * i) * n_fft + j], y = buf[(2 * i + 1) * n_fft + j];
            buf[(2 * i) * n_fft + j] = (u32)(x * y % CRT_SETS[SET].P[i]);
        }
    }
}

struct CrtConsts {
    u64 inv_p1_mod_p2;   // P1^-1 mod P2
    u64 inv_p12_mod_p3;  // (P1*P2)^-1 mod P3
    u64 p12_mod_p;       // P1*P2 mod p
};

// Garner: x = x1 + x2*P1 + x3*P1*P2 (0 <= x < P1*P2*P3), then x mod p
template <typename T, int SET>
__global__ __launch_bounds__(256) void crt_combine_kernel(FieldDev fd, const u32 *__restrict__ buf, i64 n_fft, T *__restrict__ out,
                                                          i64 n_out, CrtConsts cc)
{
    constexpr u64 P1 = CRT_SETS[SET].P[0], P2 = CRT_SETS[SET].P[1], P3 = CRT_SETS[SET].P[2];
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n_out; j += (i64)gridDim.x * blockDim.x) {
        const u64 r1 = buf[j], r2 = buf[2 * n_fft + j], r3 = buf[4 * n_fft + j];
        const u64 x2 = (r2 + P2 - r1 % P2) % P2 * cc.inv_p1_mod_p2 % P2;
        const u64 t = r1 + x2 * P1; // < P1 * P2 < 2^60
        const u64 x3 = (r3 + P3 - t % P3) % P3 * cc.inv_p12_mod_p3 % P3;
        const u64 lo = Prime32::reduce64(fd, t);
        const u64 hi = Prime32::reduce64(fd, (u64)Prime32::reduce64(fd, x3) * cc.p12_mod_p);
        out[j] = (T)Prime32::reduce64(fd, lo + hi);
    }
}

template <typename T, int SET>
int run_crt_set(gfa_field *f, const void *a, i64 na, const void *b, i64 nb, void *out, hipStream_t st)
{
    constexpr const u64 *CRT_P = CRT_SETS[SET].P, *CRT_G = CRT_SETS[SET].G;
    const FieldDev &fd = f->calc;
    const i64 n_out = na + nb - 1;
    int lg = 0;
    while (((i64)1 << lg) < n_out) lg++;
    const i64 n_fft = (i64)1 << lg;
    gfa_field *aux[3];
    int rc = aux_fields(SET, aux);
    if (rc) return rc;
    u32 *buf = nullptr;
    GFA_HIP(gfa::scratch_alloc((void **)&buf, sizeof(u32) * 6 * (size_t)n_fft, st));
    const int grid = (int)std::min<i64>((n_fft + 255) / 256, 256 * 16);
    hipLaunchKernelGGL((crt_spread_kernel<T, SET>), dim3(grid), dim3(256), 0, st, (const T *)a, na, (const T *)b, nb, buf, n_fft);
    rc = GFA_OK;
    u64 omega[3];
    for (int i = 0; i < 3 && !rc; i++) {
        omega[i] = host_powmod(CRT_G[i], (CRT_P[i] - 1) / (u64)n_fft, CRT_P[i]);
        rc = gfa_ntt(aux[i], buf + 2 * i * n_fft, buf + 2 * i * n_fft, n_fft, 2, omega[i], 0, GFA_U32, (gfa_stream_t)st);
    }
    if (!rc) {
        hipLaunchKernelGGL((crt_pointwise_kernel<SET>), dim3(grid), dim3(256), 0, st, buf, n_fft);
        for (int i = 0; i < 3 && !rc; i++) {
            const u64 winv = host_powmod(omega[i], CRT_P[i] - 2, CRT_P[i]);
            rc = gfa_ntt(aux[i], buf + 2 * i * n_fft, buf + 2 * i * n_fft, n_fft, 1, winv, 1, GFA_U32, (gfa_stream_t)st);
        }
    }
    if (!rc) {
        CrtConsts cc;
        cc.inv_p1_mod_p2 = host_powmod(CRT_P[0] % CRT_P[1], CRT_P[1] - 2, CRT_P[1]);
        const u64 p12_mod_p3 = (u64)((unsigned __int128)CRT_P[0] * CRT_P[1] % CRT_P[2]);
        cc.inv_p12_mod_p3 = host_powmod(p12_mod_p3, CRT_P[2] - 2, CRT_P[2]);
        cc.p12_mod_p = (u64)((unsigned __int128)CRT_P[0] * CRT_P[1] % fd.p);
        const int g2 = (int)std::min<i64>((n_out + 255) / 256, 256 * 16);
        hipLaunchKernelGGL((crt_combine_kernel<T, SET>), dim3(g2), dim3(256), 0, st, fd, (const u32 *)buf, n_fft, (T *)out, n_out, cc);
        if (hipGetLastError() != hipSuccess) rc = GFA_ERR_HIP;
    }
    (void)gfa::scratch_free(buf, st);
    return rc;
}

template <typename T>
int run_crt(gfa_field *f, const void *a, i64 na, const void *b, i64 nb, void *out, hipStream_t st)
{
    int lg = 0;
    while (((i64)1 << lg) < na + nb - 1) lg++;
    if (crt_set_fits(1, f->calc.p, std::min(na, nb), lg)) return run_crt_set<T, 1>(f, a, na, b, nb, out, st);
    return run_crt_set<T, 0>(f, a, na, b, nb, out, st);
}

// ---- r06: extension fields.  GF(2^m) and GF(p^m) have no transform of their own for these lengths, and the direct kernel is O(na nb)
// (2^18 x 2^18 terms over GF(2^8): 121 ms).  Karatsuba over the bit / digit positions (gfa_karatsuba.h) turns the product into nt products
// of INTEGER sequences -- parity(a_j & mask_t), or (sum of the digits of a_j in E_t) mod p -- whose coefficients stay below
// min(na, nb) (p - 1)^2: exact modulo ONE transform prime (469762049, the signed-Montgomery kernels) whenever that bound is below it.
// 2 nt forward transforms in one batched call, nt pointwise products, nt inverse transforms, and the fold of gfa_matmul_mfma.hip.
constexpr u64 PLANE_P = 469762049ull, PLANE_G = 3ull;

// buf[t][j] = plane t of a[j], buf[nt + t][j] = plane t of b[j], zero padded to n_fft
template <typename T, bool BITS>
__global__ __launch_bounds__(256) void planes_spread_kernel(const T *__restrict__ a, i64 na, const T *__restrict__ b, i64 nb, u32 *__restrict__ buf, i64 n_fft, int nt,
                                                            PlaneMasks pm, DigitFold df)
{
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n_fft; j += (i64)gridDim.x * blockDim.x) {
#pragma unroll 1
        for (int side = 0; side < 2; side++) {
            const bool in = side == 0 ? j < na : j < nb;
            const T v = in ? (side == 0 ? a[j] : b[j]) : (T)0;
            u32 *dst = buf + (i64)side * nt * n_fft + j;
            if (BITS) {
                for (int t = 0; t < nt; t++) dst[(i64)t * n_fft] = (u32)(__popc((u32)v & pm.m[t]) & 1);
            } else {
                u32 d[16];
                digits_of(v, df.p, df.m, d);
                for (int t = 0; t < nt; t++) {
                    u32 sum = 0;
#pragma unroll
                    for (int i = 0; i < 16; i++) sum += ((df.set[t] >> i) & 1u) ? d[i] : 0u;
                    dst[(i64)t * n_fft] = sum % df.p;
                }
            }
        }
    }
}
// buf[t][j] <- buf[t][j] * buf[nt + t][j] mod P
__global__ __launch_bounds__(256) void planes_pointwise_kernel(u32 *__restrict__ buf, i64 n_fft, int nt)
{
    const i64 total = (i64)nt * n_fft;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (i64)gridDim.x * blockDim.x)
        buf[e] = (u32)((u64)buf[e] * (u64)buf[total + e] % PLANE_P);
}
template <typename T, bool BITS>
__global__ __launch_bounds__(256) void planes_fold_kernel(const u32 *__restrict__ buf, i64 n_fft, T *__restrict__ out, i64 n_out, BinFold bf, DigitFold df)
{
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n_out; j += (i64)gridDim.x * blockDim.x) {
        if (BITS) {
            u32 v = 0;
            for (int t = 0; t < bf.nt; t++) v ^= (buf[(i64)t * n_fft + j] & 1u) ? bf.red[t] : 0u;
            out[j] = (T)v;
        } else {
            u32 acc[16];
            for (int k = 0; k < df.m; k++) acc[k] = 0;
            for (int t = 0; t < df.nt; t++) {
                const u32 c = buf[(i64)t * n_fft + j] % df.p;
                if (c)
                    for (int k = 0; k < df.m; k++) acc[k] += c * df.R[t][k];
            }
            u64 r = 0;
            for (int k = df.m - 1; k >= 0; k--) r = r * df.p + acc[k] % df.p;
            out[j] = (T)r;
        }
    }
}

template <typename T>
int run_planes(gfa_field *f, const void *a, i64 na, const void *b, i64 nb, void *out, hipStream_t st)
{
    const FieldDev &fd = f->calc;
    const bool bits = fd.p == 2;
    PlaneMasks pm{};
    BinFold bf{};
    DigitFold df{};
    if (bits) make_bin_fold(fd, &pm, &bf);
    else if (!make_digit_fold(fd, &df)) return GFA_ERR_UNSUPPORTED;
    const int nt = bits ? bf.nt : df.nt;
    const i64 n_out = na + nb - 1;
    int lg = 0;
    while (((i64)1 << lg) < n_out) lg++;
    const i64 n_fft = (i64)1 << lg;
    gfa_field *aux[3];
    int rc = aux_fields(1, aux); // set 1: P[0] = 469762049
    if (rc) return rc;
    u32 *buf = nullptr;
    GFA_HIP(gfa::scratch_alloc((void **)&buf, sizeof(u32) * 2 * (size_t)nt * (size_t)n_fft, st));
    const int grid = (int)std::min<i64>((n_fft + 255) / 256, 256 * 16);
    if (bits) hipLaunchKernelGGL((planes_spread_kernel<T, true>), dim3(grid), dim3(256), 0, st, (const T *)a, na, (const T *)b, nb, buf, n_fft, nt, pm, df);
    else hipLaunchKernelGGL((planes_spread_kernel<T, false>), dim3(grid), dim3(256), 0, st, (const T *)a, na, (const T *)b, nb, buf, n_fft, nt, pm, df);
    const u64 omega = host_powmod(PLANE_G, (PLANE_P - 1) / (u64)n_fft, PLANE_P);
    rc = gfa_ntt(aux[0], buf, buf, n_fft, 2 * nt, omega, 0, GFA_U32, (gfa_stream_t)st);
    if (!rc) {
        const int g2 = (int)std::min<i64>(((i64)nt * n_fft + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(planes_pointwise_kernel, dim3(g2), dim3(256), 0, st, buf, n_fft, nt);
        rc = gfa_ntt(aux[0], buf, buf, n_fft, nt, host_powmod(omega, PLANE_P - 2, PLANE_P), 1, GFA_U32, (gfa_stream_t)st);
    }
    if (!rc) {
        const int g3 = (int)std::min<i64>((n_out + 255) / 256, 256 * 16);
        if (bits) hipLaunchKernelGGL((planes_fold_kernel<T, true>), dim3(g3), dim3(256), 0, st, (const u32 *)buf, n_fft, (T *)out, n_out, bf, df);
        else hipLaunchKernelGGL((planes_fold_kernel<T, false>), dim3(g3), dim3(256), 0, st, (const u32 *)buf, n_fft, (T *)out, n_out, bf, df);
        if (hipGetLastError() != hipSuccess) rc = GFA_ERR_HIP;
    }
    (void)gfa::scratch_free(buf, st);
    return rc;
}

// extension fields the plane route serves: GF(2^m), m <= 32; GF(p^m), odd p <= 251, m <= 16 with at most 81 leaves; coefficients exact modulo PLANE_P
bool planes_eligible(const FieldDev &fd, i64 na, i64 nb)
{
    const bool bits = fd.kind == KIND_BIN && fd.m >= 2 && fd.m <= 32;
    const bool digs = fd.kind == KIND_EXT && (fd.p & 1) && fd.p <= 251 && fd.m >= 2 && fd.m <= 16;
    if (!bits && !digs) return false;
    const i64 lo = std::min(na, nb), n_out = na + nb - 1;
    if (lo < 32 || (double)na * (double)nb < (double)((i64)1 << planes_min_log())) return false;
    int lg = 0;
    while (((i64)1 << lg) < n_out) lg++;
    const int nt_max = bits ? (fd.m <= 8 ? 27 : fd.m <= 16 ? 81 : 243) : 81;
    if (lg > 26 || ((i64)2 * nt_max << lg) > ((i64)1 << 31)) return false; // at most 8 GiB of planes
    const long double bound = (long double)lo * (long double)(fd.p - 1) * (long double)(fd.p - 1);
    return bound < (long double)PLANE_P * 0.99L;
}

} // namespace

namespace gfa {

// GFA_CONVOLVE_CRT=0 keeps every product on the direct kernel (A/B measurements)
bool convolve_crt_eligible(const FieldDev &fd, i64 na, i64 nb)
{
    if (planes_eligible(fd, na, nb)) return true; // r06: extension fields through Karatsuba planes
    if (fd.kind != KIND_PRIME32 || fd.m != 1) return false;
    const i64 lo = std::min(na, nb), n_out = na + nb - 1;
    constexpr i64 min_work = (i64)1 << 22;
    if (lo < 64 || n_out > ((i64)1 << CRT_MAX_LOG)) return false;
    // the CRT route costs ~0.1 ms whatever the size (15 launches); the direct kernel is faster below ~2^22 multiply-adds
    // (tools/convolve_bench.py: 4096 x 4096 terms 0.55 ms direct, 0.094 ms here; 2^20 x 2^20 terms 0.38 ms here)
    if ((double)na * (double)nb < (double)min_work) return false;
    int lg = 0;
    while (((i64)1 << lg) < n_out) lg++;
    return crt_set_fits(0, fd.p, lo, lg) || crt_set_fits(1, fd.p, lo, lg);
}

int convolve_crt(gfa_field *f, int dtype, const void *a, i64 na, const void *b, i64 nb, void *out, hipStream_t st)
{
    if (f->calc.m > 1) {
        switch (dtype) {
        case GFA_U8: return run_planes<uint8_t>(f, a, na, b, nb, out, st);
        case GFA_U16: return run_planes<uint16_t>(f, a, na, b, nb, out, st);
        case GFA_U32: return run_planes<uint32_t>(f, a, na, b, nb, out, st);
        case GFA_U64: return run_planes<uint64_t>(f, a, na, b, nb, out, st);
        }
        set_error("gfa_convolve: bad dtype");
        return GFA_ERR_INVALID;
    }
    switch (dtype) {
    case GFA_U8: return run_crt<uint8_t>(f, a, na, b, nb, out, st);
    case GFA_U16: return run_crt<uint16_t>(f, a, na, b, nb, out, st);
    case GFA_U32: return run_crt<uint32_t>(f, a, na, b, nb, out, st);
    case GFA_U64: return run_crt<uint64_t>(f, a, na, b, nb, out, st);
    }
    set_error("gfa_convolve: bad dtype");
    return GFA_ERR_INVALID;
}

} // namespace gfa
