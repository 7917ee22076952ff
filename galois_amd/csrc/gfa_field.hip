// gfa_field.hip -- field handles: constants, lookup tables, host scalar arithmetic, lazy device upload.
// Replaces the arithmetic set-up done by the reference's class factory (src/galois/_fields/_factory.py:364-532)
// and UFuncMixin._build_lookup_tables (src/galois/_domains/_lookup.py:319-371).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "gfa_internal.h"

namespace gfa {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char *what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return GFA_ERR_HIP;
}

namespace {
std::mutex g_pool_mu;
std::vector<hipMemPool_t> g_pools; // by device ordinal; nullptr = not created (or creation refused: default pool)
std::vector<char> g_pool_tried;
} // namespace

hipError_t scratch_alloc(void **p, size_t bytes, hipStream_t st)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemPool_t pool = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        if ((int)g_pools.size() <= dev) { g_pools.resize(dev + 1, nullptr); g_pool_tried.resize(dev + 1, 0); }
        if (!g_pool_tried[dev]) {
            g_pool_tried[dev] = 1;
            hipMemPoolProps props = {};
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            hipMemPool_t np = nullptr;
            if (hipMemPoolCreate(&np, &props) == hipSuccess) {
                // Freed blocks up to this many bytes stay in the pool across synchronisations (the point of the private pool: no
                // driver round trip per Reed-Solomon decode); anything above goes back at the next synchronisation, so one huge
                // call cannot pin memory that the host framework's own allocator then fails to get.  GFA_SCRATCH_KEEP_MB overrides.
                const char *env = getenv("GFA_SCRATCH_KEEP_MB");
                uint64_t keep = (uint64_t)(env ? strtoull(env, nullptr, 10) : 256ull) << 20;
                (void)hipMemPoolSetAttribute(np, hipMemPoolAttrReleaseThreshold, &keep);
                g_pools[dev] = np;
            } else {
                (void)hipGetLastError();
            }
        }
        pool = g_pools[dev];
    }
    if (pool) return hipMallocFromPoolAsync(p, bytes, pool, st);
    return hipMallocAsync(p, bytes, st);
}

hipError_t scratch_free(void *p, hipStream_t st) { return hipFreeAsync(p, st); }

// returns the pool's unused blocks beyond `keep_bytes` to the driver (current device); GFA_OK when no pool exists yet
int scratch_trim(size_t keep_bytes)
{
    int dev = 0;
    GFA_HIP(hipGetDevice(&dev));
    hipMemPool_t pool = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        if (dev < (int)g_pools.size()) pool = g_pools[dev];
    }
    if (!pool) return GFA_OK;
    GFA_HIP(hipDeviceSynchronize()); // blocks freed on streams still running are not trimmable
    GFA_HIP(hipMemPoolTrimTo(pool, keep_bytes));
    return GFA_OK;
}

int time_loop(hipStream_t st, int iters, float *ms_out, const std::function<int()> &launch)
{ // three warm-up calls (table upload, attribute set-up, work-buffer pools), then about 50 ms more of the same call, then three
    // groups of `iters` back-to-back calls bracketed by HIP events on the launch stream; the MEDIAN of the three group averages is
    // reported.  The 50 ms are for the clocks: after a host-side pause (generating the next test input takes a second) the part
    // needs tens of milliseconds of load to come back up, and a 0.2 ms kernel timed straight away reads 10-15 % slow
    // (profiles/r04_bench_clock_ramp.txt: the same 2^20 x 64 transform 0.249 -> 0.230 -> 0.220 ms over three consecutive timings).
    int rc;
    hipEvent_t e[4];
    for (auto &ev : e) GFA_HIP(hipEventCreate(&ev));
    GFA_HIP(hipStreamSynchronize(st));
    {
        float total = 0.f;
        long launched = 0;
        for (int chunk = 3; total < 50.f && launched < 20000; chunk = chunk < 1024 ? chunk * 2 : chunk) {
            GFA_HIP(hipEventRecord(e[0], st));
            for (int i = 0; i < chunk; i++)
                if ((rc = launch())) return rc;
            GFA_HIP(hipEventRecord(e[1], st));
            GFA_HIP(hipEventSynchronize(e[1]));
            float t = 0.f;
            GFA_HIP(hipEventElapsedTime(&t, e[0], e[1]));
            if (launched > 0) total += t; // the first chunk carries table uploads and attribute set-up: not load
            launched += chunk;
        }
    }
    GFA_HIP(hipStreamSynchronize(st));
    const int n = iters > 0 ? iters : 1;
    GFA_HIP(hipEventRecord(e[0], st));
    for (int g = 0; g < 3; g++) {
        for (int i = 0; i < n; i++)
            if ((rc = launch())) return rc;
        GFA_HIP(hipEventRecord(e[g + 1], st));
    }
    GFA_HIP(hipEventSynchronize(e[3]));
    float ms[3];
    for (int g = 0; g < 3; g++) GFA_HIP(hipEventElapsedTime(&ms[g], e[g], e[g + 1]));
    if (ms[0] > ms[1]) std::swap(ms[0], ms[1]);
    if (ms[1] > ms[2]) std::swap(ms[1], ms[2]);
    if (ms[0] > ms[1]) std::swap(ms[0], ms[1]);
    *ms_out = ms[1] / n;
    for (auto &ev : e) (void)hipEventDestroy(ev);
    return GFA_OK;
}

u64 HostArith::add(const FieldDev &f, u64 a, u64 b)
{
    switch (f.kind) {
    case KIND_PRIME32: return Prime32::add(f, (u32)a, (u32)b);
    case KIND_PRIME64: return Prime64::add(f, a, b);
    case KIND_GOLDILOCKS: return Goldilocks::add(f, a, b);
    case KIND_BIN: return a ^ b;
    default: return Ext::add(f, a, b);
    }
}
u64 HostArith::sub(const FieldDev &f, u64 a, u64 b)
{
    switch (f.kind) {
    case KIND_PRIME32: return Prime32::sub(f, (u32)a, (u32)b);
    case KIND_PRIME64: return Prime64::sub(f, a, b);
    case KIND_GOLDILOCKS: return Goldilocks::sub(f, a, b);
    case KIND_BIN: return a ^ b;
    default: return Ext::sub(f, a, b);
    }
}
u64 HostArith::neg(const FieldDev &f, u64 a)
{
    switch (f.kind) {
    case KIND_PRIME32: return Prime32::neg(f, (u32)a);
    case KIND_PRIME64: return Prime64::neg(f, a);
    case KIND_GOLDILOCKS: return Goldilocks::neg(f, a);
    case KIND_BIN: return a;
    default: return Ext::neg(f, a);
    }
}
u64 HostArith::mul(const FieldDev &f, u64 a, u64 b)
{
    switch (f.kind) {
    case KIND_PRIME32: return Prime32::mul(f, (u32)a, (u32)b);
    case KIND_PRIME64: return Prime64::mul(f, a, b);
    case KIND_GOLDILOCKS: return Goldilocks::mul(f, a, b);
    case KIND_BIN: return Bin::mul(f, a, b);
    default: return Ext::mul(f, a, b);
    }
}
bool HostArith::inv(const FieldDev &f, u64 a, u64 *out)
{
    if (a == 0) { *out = 0; return false; }
    switch (f.kind) {
    case KIND_PRIME32: *out = Prime32::inv(f, (u32)a); break;
    case KIND_PRIME64: *out = Prime64::inv(f, a); break;
    case KIND_GOLDILOCKS: *out = Goldilocks::inv(f, a); break;
    case KIND_BIN: *out = Bin::inv(f, a); break;
    default: *out = Ext::inv(f, a); break;
    }
    return true;
}
bool HostArith::pow(const FieldDev &f, u64 a, i64 e, u64 *out)
{
    switch (f.kind) {
    case KIND_PRIME32: { u32 r = 0; bool ok = pow_signed<Prime32>(f, (u32)a, e, &r); *out = r; return ok; }
    case KIND_PRIME64: return pow_signed<Prime64>(f, a, e, out);
    case KIND_GOLDILOCKS: return pow_signed<Goldilocks>(f, a, e, out);
    case KIND_BIN: return pow_signed<Bin>(f, a, e, out);
    default: return pow_signed<Ext>(f, a, e, out);
    }
}

} // namespace gfa

using namespace gfa;

bool gfa_field::use_lookup() const
{
    if (mode == GFA_MODE_LOOKUP) return has_lut;
    if (mode == GFA_MODE_CALCULATE) return false;
    // AUTO, from tools/ew_bench.py on MI355X:
    //  * order <= 256 (any field, uint8): the LDS full-table kernels run at the streaming ceiling (mul/div/reciprocal
    //    66-76 % of HBM peak vs 48 % / 5 % for explicit GF(31) arithmetic) -> lookup;
    //  * odd-characteristic extension fields up to 2^20: digit-vector arithmetic is far slower than table gathers -> lookup;
    //  * GF(2^m), 8 < m <= 20 and all larger prime fields: explicit arithmetic beats EXP/LOG gathers from L2 -> calculate.
    if (!has_lut) return false;
    if (calc.q <= 256) return true;
    return calc.m > 1 && calc.p != 2;
}

gfa::FieldDev gfa_field::lut_desc(const gfa::FieldDeviceState &st) const
{
    FieldDev d = calc;
    d.kind = KIND_LUT;
    d.qm1 = (u32)(calc.q - 1);
    d.zech_e = zech_e;
    d.exp_tab = st.exp_tab;
    d.log_tab = st.log_tab;
    d.zech_tab = st.zech_tab;
    return d;
}

template <typename T>
static int upload(T **dst, const std::vector<T> &src)
{
    if (src.empty()) { *dst = nullptr; return GFA_OK; }
    GFA_HIP(hipMalloc((void **)dst, src.size() * sizeof(T)));
    GFA_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return GFA_OK;
}

int gfa_field::ensure_device(int *device_out, gfa::FieldDeviceState **st_out)
{
    int d = 0;
    GFA_HIP(hipGetDevice(&d));
    std::lock_guard<std::mutex> lock(mu);
    if ((size_t)d >= dev.size()) dev.resize(d + 1);
    FieldDeviceState &st = dev[d];
    if (!st.ready) {
        int rc;
        if ((rc = upload(&st.exp_tab, h_exp))) return rc;
        if ((rc = upload(&st.log_tab, h_log))) return rc;
        if ((rc = upload(&st.zech_tab, h_zech))) return rc;
        if ((rc = upload(&st.mul8, h_mul8))) return rc;
        if ((rc = upload(&st.add8, h_add8))) return rc;
        if ((rc = upload(&st.sub8, h_sub8))) return rc;
        if ((rc = upload(&st.div8, h_div8))) return rc;
        if ((rc = upload(&st.inv8, h_inv8))) return rc;
        if ((rc = upload(&st.neg8, h_neg8))) return rc;
        if ((rc = upload(&st.exp8, h_exp8))) return rc;
        if ((rc = upload(&st.log8, h_log8))) return rc;
        if (has_lut && calc.q > 256 && calc.q <= 8192) {
            const size_t q = (size_t)calc.q, qa = (q + 7) & ~(size_t)7;
            std::vector<uint16_t> image(4 * qa, 0);
            for (size_t i = 0; i < q; i++) image[i] = (uint16_t)h_log[i];
            for (size_t i = 0; i < 2 * q; i++) image[qa + i] = (uint16_t)h_exp[i];
            for (size_t i = 0; i < q; i++) image[3 * qa + i] = (uint16_t)h_zech[i];
            if ((rc = upload(&st.mid16, image))) return rc;
        } else if (has_lut && calc.q > 8192 && calc.q <= 65536) { // LOG[qa] | EXP[0 .. q) | ZECH[qa]: indices reduced below q - 1
            const size_t q = (size_t)calc.q, qa = (q + 7) & ~(size_t)7;
            std::vector<uint16_t> image((q > 32768 ? 4 : 3) * qa, 0);
            for (size_t i = 0; i < q; i++) image[i] = (uint16_t)h_log[i];
            for (size_t i = 0; i < q; i++) image[qa + i] = (uint16_t)h_exp[i];
            for (size_t i = 0; i < q; i++) image[2 * qa + i] = (uint16_t)h_zech[i];
            if (q > 32768) // r06: INV[x] = 1 / x (INV[0] = 0): the one table big16_inv_kernel keeps in LDS
                for (size_t x = 1; x < q; x++) image[3 * qa + x] = (uint16_t)h_exp[(q - 1) - h_log[x]];
            if ((rc = upload(&st.mid16, image))) return rc;
        }
        st.ready = true;
    }
    if (device_out) *device_out = d;
    *st_out = &st;
    return GFA_OK;
}

int gfa_field::inverse_table(gfa::FieldDeviceState &st, const uint8_t **out)
{
    std::lock_guard<std::mutex> lock(mu);
    if (!st.inv24) {
        if (!has_lut || calc.q > ((u64)1 << 24)) return GFA_ERR_UNSUPPORTED;
        const size_t q = (size_t)calc.q;
        std::vector<uint8_t> t(3 * q + 4, 0);
        for (size_t x = 1; x < q; x++) {
            const u32 v = h_exp[(q - 1) - h_log[x]]; // EXP[q - 1] = EXP[0] = 1
            t[3 * x] = (uint8_t)v; t[3 * x + 1] = (uint8_t)(v >> 8); t[3 * x + 2] = (uint8_t)(v >> 16);
        }
        int rc;
        if ((rc = upload(&st.inv24, t))) return rc;
    }
    *out = st.inv24;
    return GFA_OK;
}

static u64 neg_inverse_mod_2_64(u64 p)
{ // Newton iteration: x <- x * (2 - p*x), doubling the number of correct low bits each step
    u64 x = p; // correct to 3 bits for odd p
    for (int i = 0; i < 6; i++) x *= 2 - p * x;
    return (u64)0 - x;
}

extern "C" {

int gfa_abi_version(void) { return GFA_ABI_VERSION; }

const char *gfa_last_error(void) { return g_last_error.c_str(); }

int gfa_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gfa_trim_scratch(uint64_t keep_bytes) { return scratch_trim((size_t)keep_bytes); }

int gfa_field_create(uint64_t p, uint32_t m, const uint64_t *irr, uint64_t alpha, gfa_field_t **out)
{
    if (!out || p < 2 || m < 1) { set_error("gfa_field_create: bad arguments"); return GFA_ERR_INVALID; }
    unsigned __int128 q128 = 1;
    for (u32 i = 0; i < m; i++) {
        q128 *= p;
        if (q128 >= ((unsigned __int128)1 << 64)) {
            set_error("gfa_field_create: field order does not fit in 64 bits (the reference holds such fields as "
                      "dtype=object Python integers; no device representation)");
            return GFA_ERR_UNSUPPORTED;
        }
    }
    gfa_field *f = new gfa_field();
    FieldDev &d = f->calc;
    memset(&d, 0, sizeof(d));
    d.p = p; d.m = m; d.q = (u64)q128;
    f->alpha = alpha;
    if (alpha == 0 || alpha >= d.q) {
        delete f;
        set_error("gfa_field_create: primitive element out of range");
        return GFA_ERR_INVALID;
    }
    if (m == 1) {
        if (p < ((u64)1 << 32)) {
            d.kind = KIND_PRIME32;
            d.mu = (u64)((((unsigned __int128)1) << 64) / p);
        } else if (p == Goldilocks::P) {
            d.kind = KIND_GOLDILOCKS;
        } else {
            if ((p & 1) == 0) { delete f; set_error("even modulus"); return GFA_ERR_INVALID; }
            d.kind = KIND_PRIME64;
            d.nprime = neg_inverse_mod_2_64(p);
            u64 r1 = (u64)((((unsigned __int128)1) << 64) % p);
            d.r2 = (u64)(((unsigned __int128)r1 * r1) % p);
        }
        f->irr_coeffs = {1, (p - alpha) % p}; // f(x) = x - alpha, as the reference's prime fields (_factory.py:405)
    } else {
        if (!irr) { delete f; set_error("gfa_field_create: irreducible polynomial required"); return GFA_ERR_INVALID; }
        f->irr_coeffs.assign(irr, irr + m + 1);
        if (irr[0] != 1) { delete f; set_error("irreducible polynomial must be monic of degree m"); return GFA_ERR_INVALID; }
        for (u32 i = 0; i <= m; i++)
            if (irr[i] >= p) { delete f; set_error("irreducible polynomial coefficient out of range"); return GFA_ERR_INVALID; }
        if (p == 2) {
            if (m > 63) { delete f; set_error("GF(2^m) is supported for m <= 63"); return GFA_ERR_UNSUPPORTED; }
            d.kind = KIND_BIN;
            u64 v = 0;
            for (u32 i = 0; i <= m; i++) v = (v << 1) | irr[i];
            d.irr = v;
            d.mu = Bin::fold_rounds(v, m); // > 0: products as an integer-multiply carry-less product + that many folds through f
        } else {
            if (p >= ((u64)1 << 32) || m > GFA_MAX_EXT_DEGREE) {
                delete f;
                set_error("GF(p^m) with odd p is supported for p < 2^32 and m <= 16");
                return GFA_ERR_UNSUPPORTED;
            }
            d.kind = KIND_EXT;
            d.mu = (u64)((((unsigned __int128)1) << 64) / p);
            for (u32 i = 0; i < m; i++) d.ext_irr[i] = (u32)irr[i + 1];
            d.r2 = Ext::ext_lazy_ok(p, m, d.ext_irr) ? 1 : 0;
        }
    }

    // Lookup tables, semantics of _build_lookup_tables (_lookup.py:319-371) incl. the doubled EXP.
    if (d.q <= ((u64)1 << 20)) {
        const u32 q = (u32)d.q;
        f->h_exp.assign(2 * (size_t)q, 0);
        f->h_log.assign(q, 0);
        f->h_zech.assign(q, 0);
        f->zech_e = (p == 2) ? 0 : (q - 1) / 2;
        std::vector<uint8_t> seen(q, 0);
        u64 e = 1;
        f->h_exp[0] = 1;
        bool ok = true;
        for (u32 i = 1; i < q; i++) {
            e = HostArith::mul(d, e, alpha);
            f->h_exp[i] = (u32)e;
            if (i < q - 1) {
                if (e == 0 || e >= q || seen[e] || e == 1) { ok = false; break; }
                seen[e] = 1;
                f->h_log[e] = i;
            }
        }
        if (!ok || f->h_exp[q - 1] != 1) {
            delete f;
            set_error("gfa_field_create: the primitive element is not a multiplicative generator of the field");
            return GFA_ERR_INVALID;
        }
        for (u32 i = 0; i < q; i++) f->h_zech[i] = f->h_log[HostArith::add(d, 1, f->h_exp[i])];
        for (u32 i = 0; i + 1 < q; i++) f->h_exp[q + i] = f->h_exp[1 + i];
        f->h_exp[2 * (size_t)q - 1] = 0;
        f->has_lut = true;

        if (q <= 256) {
            f->h_mul8.assign(65536, 0); f->h_add8.assign(65536, 0); f->h_sub8.assign(65536, 0);
            f->h_div8.assign(65536, 0); f->h_inv8.assign(256, 0); f->h_neg8.assign(256, 0);
            f->h_exp8.assign(512, 0); f->h_log8.assign(256, 0);
            for (u32 a = 0; a < q; a++) {
                f->h_neg8[a] = (uint8_t)HostArith::neg(d, a);
                u64 iv = 0;
                if (a) HostArith::inv(d, a, &iv);
                f->h_inv8[a] = (uint8_t)iv;
                f->h_log8[a] = (uint8_t)f->h_log[a];
            }
            for (u32 i = 0; i < 2 * q; i++) f->h_exp8[i] = (uint8_t)f->h_exp[i];
            for (u32 a = 0; a < q; a++)
                for (u32 b = 0; b < q; b++) {
                    f->h_mul8[(a << 8) | b] = (uint8_t)HostArith::mul(d, a, b);
                    f->h_add8[(a << 8) | b] = (uint8_t)HostArith::add(d, a, b);
                    f->h_sub8[(a << 8) | b] = (uint8_t)HostArith::sub(d, a, b);
                    f->h_div8[(a << 8) | b] = b ? (uint8_t)HostArith::mul(d, a, f->h_inv8[b]) : 0;
                }
            f->has_tab8 = true;
        }
    }
    *out = f;
    return GFA_OK;
}

void gfa_field_destroy(gfa_field_t *f)
{
    if (!f) return;
    gfa::ntt_forget_field(f);
    gfa::dlog_forget_field(f);
    for (auto &st : f->dev) {
        if (!st.ready) continue;
        (void)hipFree(st.exp_tab); (void)hipFree(st.log_tab); (void)hipFree(st.zech_tab);
        (void)hipFree(st.mul8); (void)hipFree(st.add8); (void)hipFree(st.sub8); (void)hipFree(st.div8);
        (void)hipFree(st.inv8); (void)hipFree(st.neg8); (void)hipFree(st.exp8); (void)hipFree(st.log8);
        (void)hipFree(st.mid16); (void)hipFree(st.inv24);
    }
    delete f;
}

int gfa_field_set_mode(gfa_field_t *f, int mode)
{
    if (!f || mode < GFA_MODE_AUTO || mode > GFA_MODE_CALCULATE) { set_error("bad mode"); return GFA_ERR_INVALID; }
    if (mode == GFA_MODE_LOOKUP && !f->has_lut) {
        set_error("lookup mode needs order <= 2^20");
        return GFA_ERR_UNSUPPORTED;
    }
    f->mode = mode;
    return GFA_OK;
}

int gfa_field_get_mode(const gfa_field_t *f) { return f->use_lookup() ? GFA_MODE_LOOKUP : GFA_MODE_CALCULATE; }

uint64_t gfa_field_order(const gfa_field_t *f) { return f->calc.q; }

int gfa_field_tables(gfa_field_t *f, int64_t *exp_out, int64_t *log_out, int64_t *zech_out, int64_t *zech_e_out)
{
    if (!f->has_lut) { set_error("field has no lookup tables (order > 2^20)"); return GFA_ERR_UNSUPPORTED; }
    if (exp_out) for (size_t i = 0; i < f->h_exp.size(); i++) exp_out[i] = f->h_exp[i];
    if (log_out) for (size_t i = 0; i < f->h_log.size(); i++) log_out[i] = f->h_log[i];
    if (zech_out) for (size_t i = 0; i < f->h_zech.size(); i++) zech_out[i] = f->h_zech[i];
    if (zech_e_out) *zech_e_out = f->zech_e;
    return GFA_OK;
}

int gfa_scalar(const gfa_field_t *f, int op, uint64_t a, uint64_t b, uint64_t *out)
{
    const FieldDev &d = f->calc;
    if (a >= d.q || (op != GFA_OP_POW && op != GFA_OP_NEG && op != GFA_OP_RECIP && b >= d.q)) {
        set_error("gfa_scalar: operand out of range");
        return GFA_ERR_INVALID;
    }
    switch (op) {
    case GFA_OP_ADD: *out = HostArith::add(d, a, b); return GFA_OK;
    case GFA_OP_SUB: *out = HostArith::sub(d, a, b); return GFA_OK;
    case GFA_OP_MUL: *out = HostArith::mul(d, a, b); return GFA_OK;
    case GFA_OP_NEG: *out = HostArith::neg(d, a); return GFA_OK;
    case GFA_OP_RECIP:
        if (!HostArith::inv(d, a, out)) { set_error("division by zero"); return GFA_ERR_INVALID; }
        return GFA_OK;
    case GFA_OP_DIV: {
        u64 bi;
        if (!HostArith::inv(d, b, &bi)) { set_error("division by zero"); return GFA_ERR_INVALID; }
        *out = HostArith::mul(d, a, bi);
        return GFA_OK;
    }
    case GFA_OP_POW:
        if (!HostArith::pow(d, a, (i64)b, out)) { set_error("division by zero"); return GFA_ERR_INVALID; }
        return GFA_OK;
    default: set_error("gfa_scalar: bad op"); return GFA_ERR_INVALID;
    }
}

} // extern "C"
