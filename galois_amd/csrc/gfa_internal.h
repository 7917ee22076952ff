// gfa_internal.h -- host-side objects behind the opaque C-ABI handles.
#pragma once
#include <hip/hip_runtime.h>

#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/galois_amd.h"
#include "gfa_arith.h"

struct gfa_field;

namespace gfa {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what);
// average milliseconds per call of `launch` over `iters` calls, HIP events recorded on `st` itself
int time_loop(hipStream_t st, int iters, float *ms_out, const std::function<int()> &launch);

#define GFA_HIP(call)                                        \
    do {                                                     \
        hipError_t _e = (call);                              \
        if (_e != hipSuccess) return gfa::hip_fail(_e, #call); \
    } while (0)

// Stream-ordered work buffers of one call (hipMallocFromPoolAsync / hipFreeAsync) from a pool the library owns, one per
// device, with a release threshold of 256 MiB (GFA_SCRATCH_KEEP_MB): freed blocks up to that total stay in the pool across
// synchronisations instead of going back to the driver (the device's default pool releases at every synchronisation --
// 0.05 ms per Reed-Solomon decode); gfa_trim_scratch() returns the rest on demand.
hipError_t scratch_alloc(void **p, size_t bytes, hipStream_t st);
hipError_t scratch_free(void *p, hipStream_t st);
int scratch_trim(size_t keep_bytes); // synchronises the device, returns unused pool memory beyond keep_bytes to the driver

void ntt_forget_field(const struct ::gfa_field *f); // drops cached NTT plans of a field being destroyed

// 2^16-point transforms over GF(65537) in one pass over HBM (gfa_ntt_fermat.hip)
bool ntt_fermat16_eligible(const FieldDev &fd, i64 n, i64 batch);
int ntt_fermat16(const void *in, void *out, i64 batch, u64 omega, int negate, hipStream_t st);

// power-of-two transforms over GF(p), p < 2^26, on signed Montgomery representatives (gfa_ntt_m32.hip); contiguous rows.
// `ws` holds ntt_m32_scratch_bytes(n, batch) bytes (the two-pass intermediate; 0 for n <= 1024).
bool ntt_m32_eligible(const FieldDev &fd, i64 n);
size_t ntt_m32_scratch_bytes(i64 n, i64 batch);
int ntt_m32(const FieldDev &fd, const void *in, void *out, void *ws, i64 n, i64 batch, u64 omega, int do_scale, u64 scale,
            hipStream_t st);

// discrete logarithms without tables (gfa_dlog.hip)
void dlog_forget_field(const struct ::gfa_field *f);
int dlog_run(struct ::gfa_field *f, const void *a, i64 sa, const void *base, i64 sb, int64_t *out, i64 n, int dtype, hipStream_t st,
             int32_t *err);

// matrix-core path of gfa_matmul for prime fields with p <= 256 (gfa_matmul_mfma.hip)
bool matmul_mfma_eligible(const FieldDev &fd, i64 M, i64 K, i64 N);
int matmul_mfma(const FieldDev &fd, int dtype, const void *a, const void *b, void *out, i64 batch, i64 M, i64 K, i64 N,
                i64 a_bstride, i64 b_bstride, hipStream_t st);

// long prime-field products through three NTT primes + CRT (gfa_conv_crt.hip)
bool convolve_crt_eligible(const FieldDev &fd, i64 na, i64 nb);
int convolve_crt(struct ::gfa_field *f, int dtype, const void *a, i64 na, const void *b, i64 nb, void *out, hipStream_t st);

// Reed-Solomon / BCH codes whose (syndrome) field has 256 < q <= 2^20 elements (gfa_rs_wide.hip)
bool rs_wide_code(const struct ::gfa_rs *code);
int rs_wide_check(const struct ::gfa_rs *code, int dtype, const char *what);
int rs_wide_encode(struct ::gfa_rs *code, const void *msg, i64 ks, void *out, i64 batch, int parity_only, int dtype, hipStream_t st);
int rs_wide_decode(struct ::gfa_rs *code, const void *recv, const uint8_t *eras, i64 ns, void *out, i64 *nerr, uint8_t *detected,
                   i64 batch, bool detect_only, int dtype, hipStream_t st);
int rs_wide_polydiv(struct ::gfa_rs *code, const void *cw, i64 ns, void *out, i64 batch, int dtype, hipStream_t st);

// element-wise kernels with 16-bit EXP / LOG / Zech tables in LDS, 256 < q <= 32768 on uint16 / uint32 / int64 storage
// (gfa_elementwise_mid.hip; mid_power_each: uint16 only).
// `lut` is gfa_field::lut_desc(), `image` FieldDeviceState::mid16.  GFA_ERR_UNSUPPORTED = operands not 16-byte aligned.
bool mid_eligible(const FieldDev &calc, const void *image, int dtype, i64 n);
bool mid_has_zech_room(const FieldDev &calc);
int mid_binary(const FieldDev &lut, const void *image, int dtype, int op, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n,
               hipStream_t st, int32_t *err);
int mid_unary(const FieldDev &lut, const void *image, int dtype, int op, const void *a, void *out, i64 n, hipStream_t st, int32_t *err);
int mid_power(const FieldDev &lut, const void *image, int dtype, const void *a, const i64 *e, void *out, i64 n, hipStream_t st, int32_t *err);
int mid_power_each(const FieldDev &lut, const void *image, const void *a, const i64 *e, void *out, i64 n, hipStream_t st, int32_t *err);
// 32768 < q <= 65536 on uint16 storage: LOG, then EXP, staged in LDS in two phases per tile; op in {MUL, DIV, RECIP, POW (one
// exponent at e[0])}.  Covers the first n & ~7 elements -- the caller runs the generic kernels on the last n & 7.
// sums / differences / negatives of GF(p^m), p odd, 8192 < q <= 2^20, as packed-digit arithmetic (gfa_elementwise_packed.hip, gfa_packed.h)
bool packed_eligible(const FieldDev &calc, int dtype, i64 n, bool pinned_to_calculate);
int packed_run(const FieldDev &calc, int dtype, int op, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st);
bool packed_mul_eligible(const FieldDev &calc, int dtype, i64 n, bool pinned_to_calculate); // products on the same digit tables
// r06: quotients (a == nullptr: reciprocals of b) of GF(p^2), odd p, 32768 < q <= 2^20 (norm; gfa_packed.h::div2) and of GF(p^3), 65536 < q <= 2^20
// (Cramer's rule; div3), uint16 / uint32 arrays
bool packed_divn_eligible(const FieldDev &calc, int dtype, i64 n);
int packed_divn_run(const FieldDev &calc, int dtype, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st, int *dev_err);
// r06: quotients / reciprocals of degrees 4 .. 8 (uint32 arrays, 65536 < q <= 2^20): ONE gather from the field's 3-byte inverse table, then the
// digit-table product; inv24 from gfa_field::inverse_table
bool packed_divt_eligible(const FieldDev &calc, int dtype, i64 n);
int packed_divt_run(const FieldDev &calc, const uint8_t *inv24, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st, int *dev_err);
int packed_mul_run(const FieldDev &calc, int dtype, const void *a, i64 sa, const void *b, i64 sb, void *out, i64 n, hipStream_t st);
bool big16_eligible(const FieldDev &calc, const void *image, int dtype, i64 n);
// uint32 / int64 storage of the same fields: narrowed into a 16-bit work buffer, run, widened (first n & ~7 elements)
bool big16_wide_eligible(const FieldDev &calc, const void *image, int dtype, i64 n);
int big16_run_wide(const FieldDev &lut, const void *image, int dtype, int op, const void *a, i64 sa, const void *b, i64 sb, const i64 *e, void *out, i64 n,
                   hipStream_t st, int32_t *err);
int big16_run(const FieldDev &lut, const void *image, int op, const void *a, i64 sa, const void *b, i64 sb, const i64 *e, void *out, i64 n,
              hipStream_t st, int32_t *err);

// Host scalar arithmetic on a field, dispatched on FieldDev::kind with the same formulas the kernels use.
struct HostArith {
    static u64 add(const FieldDev &f, u64 a, u64 b);
    static u64 sub(const FieldDev &f, u64 a, u64 b);
    static u64 neg(const FieldDev &f, u64 a);
    static u64 mul(const FieldDev &f, u64 a, u64 b);
    static bool inv(const FieldDev &f, u64 a, u64 *out);
    static bool pow(const FieldDev &f, u64 a, i64 e, u64 *out);
};

// Per-device copies of a field's tables.
struct FieldDeviceState {
    bool ready = false;
    u32 *exp_tab = nullptr, *log_tab = nullptr, *zech_tab = nullptr; // LUT kind (q <= 2^20)
    // q <= 256: full binary-operation tables, row stride 256, index (a << 8) | b
    uint8_t *mul8 = nullptr, *add8 = nullptr, *sub8 = nullptr, *div8 = nullptr;
    uint8_t *inv8 = nullptr, *neg8 = nullptr; // 256-entry unary tables (inv8[0] = 0)
    uint8_t *exp8 = nullptr, *log8 = nullptr; // byte EXP (512) / LOG (256) for the RS kernels
    // 256 < q <= 8192: LOG[qa] | EXP[2 qa] | ZECH[qa] as 16-bit entries, qa = q rounded up to a multiple of 8 -- the image the
    // LDS-table kernels of gfa_elementwise_mid.hip stage with 16-byte copies; 8192 < q <= 65536: LOG[qa] | EXP[qa] | ZECH[qa]
    uint16_t *mid16 = nullptr;
    // r06, 65536 < q <= 2^20, built on first use (gfa_field::inverse_table): INV[x] = 1 / x as 3-byte entries at byte 3 x
    // (INV[0] = 0; 4 bytes of padding behind the last entry) -- one gather where LOG + EXP are two, from a table that fits one XCD's L2
    uint8_t *inv24 = nullptr;
};

// can the storage dtype hold every element of a field of order q?
inline bool dtype_holds(int dtype, u64 q)
{
    switch (dtype) {
    case GFA_U8: return q - 1 <= 0xffull;
    case GFA_U16: return q - 1 <= 0xffffull;
    case GFA_U32: return q - 1 <= 0xffffffffull;
    case GFA_U64: return true;
    default: return false;
    }
}

} // namespace gfa

// dispatch on (arithmetic kind, storage dtype)
#define GFA_DISPATCH_FT(FUNC, fd, dtype, ...)                                                              \
    do {                                                                                                   \
        switch ((fd).kind) {                                                                               \
        case KIND_PRIME32:                                                                                 \
            switch (dtype) {                                                                               \
            case GFA_U8: return FUNC<Prime32, uint8_t>(__VA_ARGS__);                                       \
            case GFA_U16: return FUNC<Prime32, uint16_t>(__VA_ARGS__);                                     \
            case GFA_U32: return FUNC<Prime32, uint32_t>(__VA_ARGS__);                                     \
            case GFA_U64: return FUNC<Prime32, uint64_t>(__VA_ARGS__);                                     \
            }                                                                                              \
            break;                                                                                         \
        case KIND_LUT:                                                                                     \
            switch (dtype) {                                                                               \
            case GFA_U8: return FUNC<Lut, uint8_t>(__VA_ARGS__);                                           \
            case GFA_U16: return FUNC<Lut, uint16_t>(__VA_ARGS__);                                         \
            case GFA_U32: return FUNC<Lut, uint32_t>(__VA_ARGS__);                                         \
            case GFA_U64: return FUNC<Lut, uint64_t>(__VA_ARGS__);                                         \
            }                                                                                              \
            break;                                                                                         \
        case KIND_BIN:                                                                                     \
            switch (dtype) {                                                                               \
            case GFA_U8: return FUNC<Bin, uint8_t>(__VA_ARGS__);                                           \
            case GFA_U16: return FUNC<Bin, uint16_t>(__VA_ARGS__);                                         \
            case GFA_U32: return FUNC<Bin, uint32_t>(__VA_ARGS__);                                         \
            case GFA_U64: return FUNC<Bin, uint64_t>(__VA_ARGS__);                                         \
            }                                                                                              \
            break;                                                                                         \
        case KIND_EXT:                                                                                     \
            switch (dtype) {                                                                               \
            case GFA_U8: return FUNC<Ext, uint8_t>(__VA_ARGS__);                                           \
            case GFA_U16: return FUNC<Ext, uint16_t>(__VA_ARGS__);                                         \
            case GFA_U32: return FUNC<Ext, uint32_t>(__VA_ARGS__);                                         \
            case GFA_U64: return FUNC<Ext, uint64_t>(__VA_ARGS__);                                         \
            }                                                                                              \
            break;                                                                                         \
        case KIND_PRIME64:                                                                                 \
            if (dtype == GFA_U64) return FUNC<Prime64, uint64_t>(__VA_ARGS__);                             \
            break;                                                                                         \
        case KIND_GOLDILOCKS:                                                                              \
            if (dtype == GFA_U64) return FUNC<Goldilocks, uint64_t>(__VA_ARGS__);                          \
            break;                                                                                         \
        }                                                                                                  \
        set_error("unsupported (field kind, dtype) combination");                                          \
        return GFA_ERR_UNSUPPORTED;                                                                        \
    } while (0)

struct gfa_field {
    gfa::FieldDev calc;   // descriptor for explicit calculation (kind = PRIME32/PRIME64/GOLDILOCKS/BIN/EXT)
    bool has_lut = false; // q <= 2^20: host tables exist
    bool has_tab8 = false; // q <= 256
    int mode = GFA_MODE_AUTO;
    uint64_t alpha = 0;
    std::vector<uint64_t> irr_coeffs; // m+1, highest degree first
    // host tables in the device layout
    std::vector<uint32_t> h_exp, h_log, h_zech;
    uint32_t zech_e = 0;
    std::vector<uint8_t> h_mul8, h_add8, h_sub8, h_div8, h_inv8, h_neg8, h_exp8, h_log8;
    std::mutex mu;
    std::deque<gfa::FieldDeviceState> dev; // indexed by HIP device ordinal; a deque: growing it keeps handed-out pointers valid

    // true if the lookup path is the one to launch for this field in its current mode
    bool use_lookup() const;
    int ensure_device(int *device_out, gfa::FieldDeviceState **st_out); // lazy upload to the current device
    gfa::FieldDev lut_desc(const gfa::FieldDeviceState &st) const;      // descriptor with kind = KIND_LUT
    int inverse_table(gfa::FieldDeviceState &st, const uint8_t **out);  // st.inv24, uploaded on first use (has_lut, q <= 2^24)
};

struct gfa_rs {
    gfa_field *field = nullptr;
    int64_t n = 0, k = 0, c = 1;
    uint64_t alpha = 0;
    bool systematic = true;
    int64_t base_p = 0; // 0: symbols live in `field` (Reed-Solomon).  p: BCH code over the prime subfield GF(p); the roots,
                        // syndromes and locators are in `field` = GF(p^m) while corrections use GF(p) subtraction
    std::vector<uint64_t> roots, gpoly, P; // roots: d-1 entries; gpoly: n-k+1, highest degree first; P: k x (n-k) row-major
    struct Dev {
        bool ready = false;
        uint8_t *P8 = nullptr;     // k x (n-k)
        uint8_t *roots8 = nullptr; // n-k
        uint8_t *g8 = nullptr;     // generator polynomial, highest degree first, n-k+1 coefficients
        uint32_t *lfsr = nullptr;  // 256 x (n-k)/4 words: rows f * (g_{nk-1} .. g_0) of the byte-wide LFSR (binary fields)
        uint8_t *aux8 = nullptr;   // rs_decode_bin_kernel: 64 syndrome points by lane, 64 source lanes, 256 positions by field element
        // codes over fields above 256 elements (gfa_rs_wide.hip): 32-bit copies instead of the byte arrays
        uint32_t *Pw = nullptr, *rootsw = nullptr, *gw = nullptr;
    };
    std::mutex mu;
    std::deque<Dev> dev; // a deque: growing it keeps handed-out pointers valid
    int ensure_device(int *device_out, Dev **out);
};
