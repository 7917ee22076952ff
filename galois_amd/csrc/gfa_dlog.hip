// gfa_dlog.hip -- discrete logarithms in fields without LOG tables (order > 2^20).
//
// The reference switches between brute force, Pollard's rho and Pohlig-Hellman (_domains/_calculate.py:595-755); the
// logarithm itself is unique, so the device uses the one algorithm that is data parallel: Pohlig-Hellman over the
// factorisation N = q - 1 = prod r_i^e_i (supplied by the host, gfa_log_prepare), each base-r digit found by baby-step /
// giant-step in the subgroup of order r with a sorted baby table built once per field.  One element per thread.
#include <algorithm>
#include <map>

#include "gfa_internal.h"

using namespace gfa;

namespace {

constexpr int MAX_FACTORS = 16;

struct FactorDev {
    u64 r;          // prime
    u32 e;          // multiplicity
    u32 m;          // baby steps: ceil(sqrt(r))
    u64 M;          // r^e
    u64 cof;        // N / M
    u64 gM_inv;     // (alpha^(N/M))^-1, generator of the subgroup of order M, inverted
    u64 giant;      // (alpha^(N/r))^-m
    u64 crt;        // (N/M) * ((N/M)^-1 mod M) mod N
    u64 r_pow_top;  // r^(e-1)
    const u64 *baby_val; // sorted gamma_r^j
    const u32 *baby_idx; // the j that belongs to it
};

struct PlanDev {
    int count;
    u64 N; // q - 1
    FactorDev f[MAX_FACTORS];
};

struct Plan {
    std::vector<u64> primes;
    std::vector<u32> mults;
    struct PerDevice { bool ready = false; PlanDev pd; std::vector<void *> allocs; };
    std::map<int, PerDevice> dev;
};

std::mutex g_mu;
std::map<const gfa_field *, Plan> g_plans;

u64 mulmod64(u64 a, u64 b, u64 n) { return (u64)((unsigned __int128)a * b % n); }

u64 inv_mod(u64 a, u64 n)
{ // a^-1 mod n, gcd(a, n) = 1
    __int128 r0 = n, r1 = a % n, t0 = 0, t1 = 1;
    while (r1 != 0) {
        __int128 qq = r0 / r1;
        __int128 tmp = r0 - qq * r1; r0 = r1; r1 = tmp;
        tmp = t0 - qq * t1; t0 = t1; t1 = tmp;
    }
    if (t0 < 0) t0 += n;
    return (u64)t0;
}

// t -> d in [0, r) with gamma_r^d == t, or r if t is not in the subgroup (cannot happen for valid input)
template <class F>
__device__ u64 bsgs(const FieldDev &fd, const FactorDev &fc, typename F::elem t)
{
    typedef typename F::elem E;
    for (u32 i = 0; i <= fc.m; i++) {
        // binary search for t among the sorted baby values
        u32 lo = 0, hi = fc.m;
        const u64 key = (u64)t;
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (fc.baby_val[mid] < key) lo = mid + 1; else hi = mid;
        }
        if (lo < fc.m && fc.baby_val[lo] == key) {
            const u64 d = (u64)i * fc.m + fc.baby_idx[lo];
            return d < fc.r ? d : fc.r;
        }
        t = F::mul(fd, t, (E)fc.giant);
    }
    return fc.r;
}

template <class F, typename T>
__global__ __launch_bounds__(128) void dlog_kernel(FieldDev fd, PlanDev pl, const T *__restrict__ a, int sa, u64 *__restrict__ out,
                                                   i64 n, int32_t *err)
{
    typedef typename F::elem E;
    int bad = 0;
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (i64)gridDim.x * blockDim.x) {
        const E x = (E)a[sa ? idx : 0];
        if (x == 0) { bad |= GFA_DEVERR_LOG_ZERO; out[idx] = 0; continue; }
        u64 acc = 0;
        for (int fi = 0; fi < pl.count; fi++) {
            const FactorDev &fc = pl.f[fi];
            const E h = F::pow_u(fd, x, fc.cof); // in the subgroup of order M = r^e
            u64 d = 0, rk = 1, rtop = fc.r_pow_top;
            for (u32 k = 0; k < fc.e; k++) {
                // t = (h * gM^-d)^(r^(e-1-k)) has order dividing r and equals gamma_r^(d_k)
                E t = F::mul(fd, h, F::pow_u(fd, (E)fc.gM_inv, d));
                t = F::pow_u(fd, t, rtop);
                const u64 dk = bsgs<F>(fd, fc, t);
                if (dk >= fc.r) { bad |= GFA_DEVERR_LOG_BASE; break; }
                d += dk * rk;
                rk *= fc.r;
                rtop /= fc.r;
            }
            acc = (u64)(((unsigned __int128)d * fc.crt + acc) % pl.N);
        }
        out[idx] = acc;
    }
    if (bad && err) atomicOr((int *)err, bad);
}

// out = la * lb^-1 mod N (logarithm to another primitive base); a base whose logarithm shares a factor with N is not primitive
__global__ void dlog_rebase_kernel(u64 *__restrict__ la, const u64 *__restrict__ lb, int sb, u64 N, i64 n, int32_t *err)
{
    int bad = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        __int128 r0 = N, r1 = lb[sb ? i : 0] % N, t0 = 0, t1 = 1;
        while (r1 != 0) {
            const __int128 qq = r0 / r1;
            __int128 tmp = r0 - qq * r1; r0 = r1; r1 = tmp;
            tmp = t0 - qq * t1; t0 = t1; t1 = tmp;
        }
        if (r0 != 1 && N != 1) { bad |= GFA_DEVERR_LOG_BASE; la[i] = 0; continue; }
        if (t0 < 0) t0 += N;
        la[i] = N == 1 ? 0 : (u64)((unsigned __int128)la[i] * (u64)t0 % N);
    }
    if (bad && err) atomicOr((int *)err, bad);
}

int ensure_plan_device(const gfa_field *f, Plan &pl, PlanDev **out)
{
    int d = 0;
    GFA_HIP(hipGetDevice(&d));
    Plan::PerDevice &pd = pl.dev[d];
    if (!pd.ready) {
        const FieldDev &fd = f->calc;
        const u64 N = fd.q - 1;
        pd.pd.count = (int)pl.primes.size();
        pd.pd.N = N;
        for (size_t i = 0; i < pl.primes.size(); i++) {
            FactorDev &fc = pd.pd.f[i];
            const u64 r = pl.primes[i];
            const u32 e = pl.mults[i];
            u64 M = 1, top = 1;
            for (u32 k = 0; k < e; k++) { if (k + 1 < e) top *= r; M *= r; }
            fc.r = r; fc.e = e; fc.M = M; fc.cof = N / M; fc.r_pow_top = top;
            u64 gM, gr;
            // alpha^(N/M) and alpha^(N/r): exponents may exceed int64, so go through unsigned halves
            auto upow = [&](u64 base, u64 ex) {
                u64 result = 1, b = base;
                while (ex) {
                    if (ex & 1) result = HostArith::mul(fd, result, b);
                    b = HostArith::mul(fd, b, b);
                    ex >>= 1;
                }
                return result;
            };
            gM = upow(f->alpha, N / M);
            gr = upow(f->alpha, N / r);
            u64 gM_inv;
            HostArith::inv(fd, gM, &gM_inv);
            fc.gM_inv = gM_inv;
            u64 m = 1;
            while (m * m < r) m++;
            fc.m = (u32)m;
            u64 gr_inv;
            HostArith::inv(fd, gr, &gr_inv);
            fc.giant = upow(gr_inv, m);
            fc.crt = M == N ? 1 % N : mulmod64((N / M) % N, inv_mod((N / M) % M, M), N);
            std::vector<std::pair<u64, u32>> baby(m);
            u64 cur = 1;
            for (u64 j = 0; j < m; j++) { baby[j] = {cur, (u32)j}; cur = HostArith::mul(fd, cur, gr); }
            std::sort(baby.begin(), baby.end());
            std::vector<u64> vals(m);
            std::vector<u32> idxs(m);
            for (u64 j = 0; j < m; j++) { vals[j] = baby[j].first; idxs[j] = baby[j].second; }
            u64 *dv; u32 *di;
            GFA_HIP(hipMalloc((void **)&dv, sizeof(u64) * m));
            GFA_HIP(hipMalloc((void **)&di, sizeof(u32) * m));
            GFA_HIP(hipMemcpy(dv, vals.data(), sizeof(u64) * m, hipMemcpyHostToDevice));
            GFA_HIP(hipMemcpy(di, idxs.data(), sizeof(u32) * m, hipMemcpyHostToDevice));
            pd.allocs.push_back(dv); pd.allocs.push_back(di);
            fc.baby_val = dv; fc.baby_idx = di;
        }
        pd.ready = true;
    }
    *out = &pd.pd;
    return GFA_OK;
}

template <class F, typename T>
int launch_dlog_ft(const FieldDev &fd, const PlanDev &pl, const void *a, i64 sa, u64 *out, i64 n, hipStream_t st, int32_t *err)
{
    const i64 blocks = std::max<i64>(1, std::min<i64>((n + 127) / 128, 1 << 20));
    hipLaunchKernelGGL((dlog_kernel<F, T>), dim3((unsigned)blocks), dim3(128), 0, st, fd, pl, (const T *)a, (int)sa, out, n, err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}
int dispatch_dlog(const FieldDev &fd, int dtype, const PlanDev &pl, const void *a, i64 sa, u64 *out, i64 n, hipStream_t st, int32_t *err)
{
    GFA_DISPATCH_FT(launch_dlog_ft, fd, dtype, fd, pl, a, sa, out, n, st, err);
}

} // namespace

namespace gfa {

void dlog_forget_field(const gfa_field *f)
{
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_plans.find(f);
    if (it == g_plans.end()) return;
    for (auto &kv : it->second.dev)
        for (void *p : kv.second.allocs) (void)hipFree(p);
    g_plans.erase(it);
}

bool dlog_prepared(const gfa_field *f)
{
    std::lock_guard<std::mutex> lock(g_mu);
    return g_plans.count(f) != 0;
}

int dlog_run(gfa_field *f, const void *a, i64 sa, const void *base, i64 sb, int64_t *out, i64 n, int dtype, hipStream_t st, int32_t *err)
{
    PlanDev *pd;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_plans.find(f);
        if (it == g_plans.end()) { set_error("gfa_log: call gfa_log_prepare with the factorisation of q - 1 first"); return GFA_ERR_INVALID; }
        int rc = ensure_plan_device(f, it->second, &pd);
        if (rc) return rc;
    }
    int rc = dispatch_dlog(f->calc, dtype, *pd, a, sa, (u64 *)out, n, st, err);
    if (rc || !base) return rc;
    const i64 nb = sb ? n : 1;
    u64 *lb = nullptr;
    GFA_HIP(gfa::scratch_alloc((void **)&lb, sizeof(u64) * (size_t)nb, st));
    rc = dispatch_dlog(f->calc, dtype, *pd, base, sb, lb, nb, st, err);
    if (!rc) {
        const unsigned blocks = (unsigned)std::max<i64>(1, std::min<i64>((n + 255) / 256, 65535));
        hipLaunchKernelGGL(dlog_rebase_kernel, dim3(blocks), dim3(256), 0, st, (u64 *)out, lb, (int)sb, pd->N, n, err);
    }
    GFA_HIP(gfa::scratch_free(lb, st));
    return rc;
}

} // namespace gfa

extern "C" int gfa_log_prepare(gfa_field_t *f, const uint64_t *primes, const uint32_t *multiplicities, int count)
{
    if (!f || !primes || !multiplicities || count < 1 || count > MAX_FACTORS) { set_error("gfa_log_prepare: bad arguments"); return GFA_ERR_INVALID; }
    const u64 N = f->calc.q - 1;
    unsigned __int128 prod = 1;
    for (int i = 0; i < count; i++) {
        if (primes[i] < 2 || multiplicities[i] < 1) { set_error("gfa_log_prepare: bad factor"); return GFA_ERR_INVALID; }
        if (primes[i] > ((u64)1 << 40)) {
            set_error("gfa_log_prepare: a prime factor of q - 1 exceeds 2^40 (baby-step table too large)");
            return GFA_ERR_UNSUPPORTED;
        }
        for (u32 k = 0; k < multiplicities[i]; k++) prod *= primes[i];
    }
    if (prod != (unsigned __int128)N) { set_error("gfa_log_prepare: the factors do not multiply to q - 1"); return GFA_ERR_INVALID; }
    std::lock_guard<std::mutex> lock(g_mu);
    Plan &pl = g_plans[f];
    if (pl.primes.empty()) {
        pl.primes.assign(primes, primes + count);
        pl.mults.assign(multiplicities, multiplicities + count);
    }
    return GFA_OK;
}
