/*
 * gf_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT part of the product.
 *
 * A plain-C, single-threaded CPU restatement of the reference (mhostetter/galois) algorithms on
 * the hot path named by BASELINE.json:north_star.  It exists to (1) check the HIP kernels bit for
 * bit at sizes the pure-Python reference cannot reach and (2) serve as the timed "port" CPU
 * baseline in bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product package (galois_amd/) never imports, links or executes anything in oracle/.
 *
 * Parity pinning: this file is validated against (a) the reference itself, imported in place in
 * the build container through oracle/ref_shim (tests/test_oracle_vs_reference.py, skipped where
 * /root/reference is absent) and (b) the committed golden fixtures in tests/golden/ that were
 * generated from the reference (tests/golden/generate_golden.py) and from the reference's own
 * Sage/SymPy pickles (tests/test_oracle_golden.py).
 *
 * Every function cites the reference file:line (relative to /root/reference/src/galois) it follows.
 * Element type is uint64_t (the reference's JIT kernels are int64(int64,int64); for fields whose
 * order exceeds int64 the reference holds Python ints -- here p up to 2^64-1 is carried through
 * unsigned __int128 intermediates).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef int64_t i64;

#define GFO_MAX_DEGREE 64

enum { GFO_OK = 0, GFO_ZERO_DIVISION = 1, GFO_BAD_ARG = 2 };
enum { GFO_ADD = 0, GFO_SUB = 1, GFO_MUL = 2, GFO_DIV = 3, GFO_NEG = 4, GFO_RECIP = 5, GFO_POW = 6 };

typedef struct gfo_field {
    u64 p;      /* characteristic */
    u64 m;      /* degree */
    u64 q;      /* order p^m (must fit in 64 bits; q==0 means 2^64, unsupported) */
    u64 irr;    /* irreducible polynomial as an integer in base p (degree m) -- prime fields: unused */
    u64 alpha;  /* primitive element as an integer */
    int lookup; /* 1: use EXP/LOG/ZECH tables (jit-lookup semantics); 0: explicit calculation */
    i64 *EXP, *LOG, *ZECH;
    i64 ZECH_E;
    u64 irr_vec[GFO_MAX_DEGREE]; /* irreducible poly minus x^m, base-p digits, MSB first (degree m-1 .. 0) */
} gfo_field;

/* ------------------------------------------------------------------------------------------------
 * Explicit ("calculate") scalar arithmetic.  _domains/_calculate.py
 * ---------------------------------------------------------------------------------------------- */

/* int_to_vector / vector_to_int: _domains/_calculate.py:22-47 (base-p digits, most significant first) */
static void int_to_vector(u64 a, u64 p, u64 m, u64 *vec)
{
    for (i64 i = (i64)m - 1; i >= 0; i--) {
        vec[i] = a % p;
        a /= p;
    }
}

static u64 vector_to_int(const u64 *vec, u64 p, u64 m)
{
    u64 a = 0;
    for (u64 i = 0; i < m; i++) a = a * p + vec[i];
    return a;
}

static inline u64 mulmod(u64 a, u64 b, u64 p) { return (u64)(((u128)a * b) % p); }

/* add_modular.calculate _calculate.py:142-147 ; GF(2^m) override np.bitwise_xor _fields/_ufunc.py:59 ;
 * add_vector.calculate _calculate.py:174-181 */
static u64 calc_add(const gfo_field *f, u64 a, u64 b)
{
    if (f->m == 1) {
        u128 c = (u128)a + b;
        if (c >= f->p) c -= f->p;
        return (u64)c;
    }
    if (f->p == 2) return a ^ b;
    u64 av[GFO_MAX_DEGREE], bv[GFO_MAX_DEGREE];
    int_to_vector(a, f->p, f->m, av);
    int_to_vector(b, f->p, f->m, bv);
    for (u64 i = 0; i < f->m; i++) av[i] = (av[i] + bv[i]) % f->p;
    return vector_to_int(av, f->p, f->m);
}

/* negative_modular.calculate _calculate.py:193-199 ; np.positive override for GF(2^m) ;
 * negative_vector.calculate _calculate.py:226-232 */
static u64 calc_neg(const gfo_field *f, u64 a)
{
    if (f->m == 1) return a == 0 ? 0 : f->p - a;
    if (f->p == 2) return a;
    u64 av[GFO_MAX_DEGREE];
    int_to_vector(a, f->p, f->m, av);
    for (u64 i = 0; i < f->m; i++) av[i] = (f->p - av[i]) % f->p;
    return vector_to_int(av, f->p, f->m);
}

/* subtract_modular.calculate _calculate.py:244-251 ; subtract_vector.calculate _calculate.py:278-285 */
static u64 calc_sub(const gfo_field *f, u64 a, u64 b)
{
    if (f->m == 1) return a >= b ? a - b : (u64)((u128)f->p + a - b);
    if (f->p == 2) return a ^ b;
    u64 av[GFO_MAX_DEGREE], bv[GFO_MAX_DEGREE];
    int_to_vector(a, f->p, f->m, av);
    int_to_vector(b, f->p, f->m, bv);
    for (u64 i = 0; i < f->m; i++) av[i] = (av[i] + f->p - bv[i]) % f->p;
    return vector_to_int(av, f->p, f->m);
}

/* multiply_modular.calculate _calculate.py:336-340 ; multiply_binary.calculate _calculate.py:308-324 ;
 * multiply_vector.calculate _calculate.py:355-383 */
static u64 calc_mul(const gfo_field *f, u64 a, u64 b)
{
    if (f->m == 1) return mulmod(a, b, f->p);
    if (f->p == 2) {
        if (b > a) { u64 t = a; a = b; b = t; }
        u64 c = 0;
        while (b > 0) {
            if (b & 1) c ^= a;
            b >>= 1;
            /* a <<= 1; if (a >= ORDER) a ^= IRREDUCIBLE_POLY -- done on the top bit so m = 64 also works */
            u64 top = (f->m == 64) ? (a >> 63) : ((a >> (f->m - 1)) & 1);
            a <<= 1;
            if (top) a ^= f->irr; /* for m == 64 irr holds the low 64 bits (x^64 term dropped by the shift) */
            if (f->m < 64) a &= (f->q - 1);
        }
        return c;
    }
    u64 av[GFO_MAX_DEGREE], bv[GFO_MAX_DEGREE], cv[GFO_MAX_DEGREE];
    u64 m = f->m, p = f->p;
    int_to_vector(a, p, m, av);
    int_to_vector(b, p, m, bv);
    memset(cv, 0, sizeof(u64) * m);
    for (u64 it = 0; it < m; it++) {
        u64 bl = bv[m - 1];
        if (bl > 0)
            for (u64 i = 0; i < m; i++) cv[i] = (cv[i] + mulmod(bl, av[i], p)) % p;
        /* multiply a(x) by x */
        u64 qd = av[0];
        for (u64 i = 0; i + 1 < m; i++) av[i] = av[i + 1];
        av[m - 1] = 0;
        if (qd > 0)
            for (u64 i = 0; i < m; i++) av[i] = (av[i] + p - mulmod(qd, f->irr_vec[i], p)) % p;
        /* divide b(x) by x */
        for (i64 i = (i64)m - 1; i >= 1; i--) bv[i] = bv[i - 1];
        bv[0] = 0;
    }
    return vector_to_int(cv, p, m);
}

/* positive_power_square_and_multiply.calculate _calculate.py:534-555 (b >= 0) */
static u64 calc_pos_pow(const gfo_field *f, u64 a, u64 b)
{
    if (b == 0) return 1;
    u64 c_square = a, c_mult = 1;
    while (b > 1) {
        if ((b & 1) == 0) {
            c_square = calc_mul(f, c_square, c_square);
            b >>= 1;
        } else {
            c_mult = calc_mul(f, c_mult, c_square);
            b -= 1;
        }
    }
    return calc_mul(f, c_mult, c_square);
}

/* reciprocal_modular_egcd.calculate _calculate.py:395-417 ; reciprocal_itoh_tsujii.calculate :469-489 */
static int calc_recip(const gfo_field *f, u64 a, u64 *out)
{
    if (a == 0) return GFO_ZERO_DIVISION;
    if (f->m == 1) {
        __int128 r2 = f->p, r1 = a, t2 = 0, t1 = 1;
        while (r1 != 0) {
            __int128 qq = r2 / r1, t;
            t = r2 - qq * r1; r2 = r1; r1 = t;
            t = t2 - qq * t1; t2 = t1; t1 = t;
        }
        if (t2 < 0) t2 += f->p;
        *out = (u64)t2;
        return GFO_OK;
    }
    /* r = (q-1)/(p-1); a^(r-1); norm a^r in GF(p); invert in GF(p); multiply back */
    u64 r = (f->q - 1) / (f->p - 1);
    u64 a_r1 = calc_pos_pow(f, a, r - 1);
    u64 a_r = calc_mul(f, a_r1, a);
    gfo_field sub = *f;
    sub.m = 1; sub.q = f->p; sub.lookup = 0;
    u64 a_r_inv;
    int rc = calc_recip(&sub, a_r, &a_r_inv);
    if (rc) return rc;
    *out = calc_mul(f, a_r_inv, a_r1);
    return GFO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Lookup-table arithmetic.  _domains/_lookup.py
 * ---------------------------------------------------------------------------------------------- */

/* _build_lookup_tables _lookup.py:319-371 */
static int build_tables(gfo_field *f)
{
    u64 q = f->q;
    f->EXP = (i64 *)calloc(2 * q, sizeof(i64));
    f->LOG = (i64 *)calloc(q, sizeof(i64));
    f->ZECH = (i64 *)calloc(q, sizeof(i64));
    if (!f->EXP || !f->LOG || !f->ZECH) return GFO_BAD_ARG;
    f->ZECH_E = (f->p == 2) ? 0 : (i64)((q - 1) / 2);
    u64 element = 1;
    f->EXP[0] = 1;
    f->LOG[0] = 0;
    for (u64 i = 1; i < q; i++) {
        element = calc_mul(f, element, f->alpha);
        f->EXP[i] = (i64)element;
        if (i < q - 1) f->LOG[element] = (i64)i;
    }
    for (u64 i = 0; i < q; i++) {
        u64 one_plus = calc_add(f, 1, (u64)f->EXP[i]);
        f->ZECH[i] = f->LOG[one_plus];
    }
    if (f->EXP[q - 1] != 1) return GFO_BAD_ARG; /* alpha is not a generator */
    /* cls._EXP[order : 2*order] = cls._EXP[1 : 1+order]: NumPy copies from the pre-assignment values, so the last
     * entry receives the old EXP[order] == 0 (_lookup.py:371) */
    for (u64 i = 0; i + 1 < q; i++) f->EXP[q + i] = f->EXP[1 + i];
    f->EXP[2 * q - 1] = 0;
    return GFO_OK;
}

/* add_ufunc.lookup _lookup.py:31-60 */
static u64 lut_add(const gfo_field *f, u64 a, u64 b)
{
    if (a == 0) return b;
    if (b == 0) return a;
    i64 m = f->LOG[a], n = f->LOG[b];
    if (m > n) { i64 t = m; m = n; n = t; }
    if (n - m == f->ZECH_E) return 0;
    return (u64)f->EXP[m + f->ZECH[n - m]];
}

/* negative_ufunc.lookup _lookup.py:74-89 */
static u64 lut_neg(const gfo_field *f, u64 a)
{
    if (a == 0) return 0;
    return (u64)f->EXP[f->ZECH_E + f->LOG[a]];
}

/* subtract_ufunc.lookup _lookup.py:105-140 */
static u64 lut_sub(const gfo_field *f, u64 a, u64 b)
{
    i64 m = f->LOG[a], n = f->LOG[b] + f->ZECH_E;
    if (b == 0) return a;
    if (a == 0) return (u64)f->EXP[n];
    if (m > n) { i64 t = m; m = n; n = t; }
    i64 z = n - m;
    if (z == f->ZECH_E) return 0;
    if (z >= (i64)f->q - 1) z -= (i64)f->q - 1;
    return (u64)f->EXP[m + f->ZECH[z]];
}

/* multiply_ufunc.lookup _lookup.py:153-168 */
static u64 lut_mul(const gfo_field *f, u64 a, u64 b)
{
    if (a == 0 || b == 0) return 0;
    return (u64)f->EXP[f->LOG[a] + f->LOG[b]];
}

/* reciprocal_ufunc.lookup _lookup.py:182-198 */
static int lut_recip(const gfo_field *f, u64 a, u64 *out)
{
    if (a == 0) return GFO_ZERO_DIVISION;
    *out = (u64)f->EXP[((i64)f->q - 1) - f->LOG[a]];
    return GFO_OK;
}

/* divide_ufunc.lookup _lookup.py:212-233 */
static int lut_div(const gfo_field *f, u64 a, u64 b, u64 *out)
{
    if (b == 0) return GFO_ZERO_DIVISION;
    if (a == 0) { *out = 0; return GFO_OK; }
    *out = (u64)f->EXP[((i64)f->q - 1) + f->LOG[a] - f->LOG[b]];
    return GFO_OK;
}

/* power_ufunc.lookup _lookup.py:247-270.  The reference computes (m*b) % (ORDER-1) in int64 with
 * Python (floor) modulo semantics; exponents are kept small enough in tests that m*b cannot wrap. */
static int lut_pow(const gfo_field *f, u64 a, i64 b, u64 *out)
{
    if (a == 0 && b < 0) return GFO_ZERO_DIVISION;
    if (b == 0) { *out = 1; return GFO_OK; }
    if (a == 0) { *out = 0; return GFO_OK; }
    __int128 prod = (__int128)f->LOG[a] * b;
    __int128 mod = (__int128)f->q - 1;
    __int128 r = prod % mod;
    if (r < 0) r += mod;
    *out = (u64)f->EXP[(i64)r];
    return GFO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Mode-dispatched scalar ops (what `self.ufunc` resolves to, _domains/_ufunc.py:69-78).
 * GF(p) add/sub/neg are always_calculate=True (_fields/_ufunc.py:23-25); GF(2^m) add/sub/neg are
 * NumPy XOR/positive overrides (_fields/_ufunc.py:59-61).
 * ---------------------------------------------------------------------------------------------- */
static u64 f_add(const gfo_field *f, u64 a, u64 b)
{
    if (f->lookup && f->m > 1 && f->p > 2) return lut_add(f, a, b);
    return calc_add(f, a, b);
}
static u64 f_sub(const gfo_field *f, u64 a, u64 b)
{
    if (f->lookup && f->m > 1 && f->p > 2) return lut_sub(f, a, b);
    return calc_sub(f, a, b);
}
static u64 f_neg(const gfo_field *f, u64 a)
{
    if (f->lookup && f->m > 1 && f->p > 2) return lut_neg(f, a);
    return calc_neg(f, a);
}
static u64 f_mul(const gfo_field *f, u64 a, u64 b) { return f->lookup ? lut_mul(f, a, b) : calc_mul(f, a, b); }
static int f_recip(const gfo_field *f, u64 a, u64 *out)
{
    return f->lookup ? lut_recip(f, a, out) : calc_recip(f, a, out);
}
/* divide.calculate _calculate.py:502-513 */
static int f_div(const gfo_field *f, u64 a, u64 b, u64 *out)
{
    if (f->lookup) return lut_div(f, a, b, out);
    if (b == 0) return GFO_ZERO_DIVISION;
    if (a == 0) { *out = 0; return GFO_OK; }
    u64 binv;
    int rc = calc_recip(f, b, &binv);
    if (rc) return rc;
    *out = calc_mul(f, a, binv);
    return GFO_OK;
}
/* power_square_and_multiply.calculate _calculate.py:579-592 */
static int f_pow(const gfo_field *f, u64 a, i64 b, u64 *out)
{
    if (f->lookup) return lut_pow(f, a, b, out);
    if (a == 0 && b < 0) return GFO_ZERO_DIVISION;
    if (b == 0) { *out = 1; return GFO_OK; }
    if (b > 0) { *out = calc_pos_pow(f, a, (u64)b); return GFO_OK; }
    u64 ainv;
    int rc = calc_recip(f, a, &ainv);
    if (rc) return rc;
    u64 nb = (b == INT64_MIN) ? ((u64)1 << 63) : (u64)(-b);
    *out = calc_pos_pow(f, ainv, nb);
    return GFO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Public API
 * ---------------------------------------------------------------------------------------------- */

/* irr_digits: coefficients of the irreducible polynomial, highest degree first, m+1 entries (ignored
 * for m == 1).  lookup != 0 builds the EXP/LOG/ZECH tables (q must then be small enough). */
gfo_field *gfo_field_new(u64 p, u64 m, const u64 *irr_digits, u64 alpha, int lookup)
{
    if (m < 1 || m > GFO_MAX_DEGREE) return NULL;
    gfo_field *f = (gfo_field *)calloc(1, sizeof(gfo_field));
    f->p = p; f->m = m; f->alpha = alpha; f->lookup = 0;
    u128 q = 1;
    for (u64 i = 0; i < m; i++) q *= p;
    if (q > ((u128)1 << 64)) { free(f); return NULL; }
    f->q = (u64)q; /* 2^64 wraps to 0: only GF(2^64), handled by the m == 64 branches */
    if (m > 1) {
        u64 irr = 0;
        for (u64 i = 1; i <= m; i++) {
            f->irr_vec[i - 1] = irr_digits[i];
            irr = irr * p + irr_digits[i];
        }
        /* irr now holds the polynomial without its leading x^m term */
        if (p == 2 && m < 64) irr |= ((u64)1 << m);
        f->irr = irr;
    }
    if (lookup) {
        if (build_tables(f) != GFO_OK) {
            free(f->EXP); free(f->LOG); free(f->ZECH); free(f);
            return NULL;
        }
        f->lookup = 1;
    }
    return f;
}

void gfo_field_free(gfo_field *f)
{
    if (!f) return;
    free(f->EXP); free(f->LOG); free(f->ZECH); free(f);
}

void gfo_field_set_lookup(gfo_field *f, int lookup) { f->lookup = (lookup && f->EXP) ? 1 : 0; }

/* copies of the tables for validation against GF._EXP/_LOG/_ZECH_LOG */
void gfo_field_tables(const gfo_field *f, i64 *exp_out, i64 *log_out, i64 *zech_out, i64 *zech_e)
{
    memcpy(exp_out, f->EXP, sizeof(i64) * 2 * f->q);
    memcpy(log_out, f->LOG, sizeof(i64) * f->q);
    memcpy(zech_out, f->ZECH, sizeof(i64) * f->q);
    *zech_e = f->ZECH_E;
}

/* Element-wise ufunc loop (what numba.vectorize compiles to; _domains/_ufunc.py:97-144).
 * a_stride/b_stride in elements: 0 broadcasts a scalar.  For GFO_POW `b` holds int64 exponents.
 * Returns GFO_ZERO_DIVISION at the first offending element (the reference raises from inside the
 * scalar kernel). */
int gfo_ufunc(const gfo_field *f, int op, const u64 *a, i64 a_stride, const u64 *b, i64 b_stride, u64 *out, i64 n)
{
    int rc = GFO_OK;
    for (i64 i = 0; i < n; i++) {
        u64 x = a[i * a_stride];
        u64 y = b ? b[i * b_stride] : 0;
        switch (op) {
        case GFO_ADD: out[i] = f_add(f, x, y); break;
        case GFO_SUB: out[i] = f_sub(f, x, y); break;
        case GFO_MUL: out[i] = f_mul(f, x, y); break;
        case GFO_DIV: rc = f_div(f, x, y, &out[i]); break;
        case GFO_NEG: out[i] = f_neg(f, x); break;
        case GFO_RECIP: rc = f_recip(f, x, &out[i]); break;
        case GFO_POW: rc = f_pow(f, x, (i64)y, &out[i]); break;
        default: return GFO_BAD_ARG;
        }
        if (rc) return rc;
    }
    return GFO_OK;
}

/* Byte-typed GF(2^8)-class fast loop for the timed CPU baseline (same arithmetic as gfo_ufunc with
 * lookup tables, minus the u64 widening) -- mirrors what the JIT-lookup ufunc does on uint8 data. */
int gfo_ufunc_u8(const gfo_field *f, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, i64 n)
{
    if (!f->lookup) return GFO_BAD_ARG;
    for (i64 i = 0; i < n; i++) {
        u64 r = 0;
        int rc = GFO_OK;
        switch (op) {
        case GFO_ADD: r = f_add(f, a[i], b[i]); break;
        case GFO_SUB: r = f_sub(f, a[i], b[i]); break;
        case GFO_MUL: r = lut_mul(f, a[i], b[i]); break;
        case GFO_DIV: rc = lut_div(f, a[i], b[i], &r); break;
        case GFO_RECIP: rc = lut_recip(f, a[i], &r); break;
        default: return GFO_BAD_ARG;
        }
        if (rc) return rc;
        out[i] = (uint8_t)r;
    }
    return GFO_OK;
}

/* fft_jit.implementation _domains/_function.py:246-384.  `factors` are the prime factors of n in
 * ascending order with multiplicity (_function.py:214-229); stages consume them from the end. */
int gfo_ntt(const gfo_field *f, const u64 *x, i64 n, u64 omega, const i64 *factors, i64 n_factors, u64 *out)
{
    u64 *buf0 = (u64 *)malloc(sizeof(u64) * (size_t)(n > 0 ? n : 1));
    u64 *buf1 = (u64 *)malloc(sizeof(u64) * (size_t)(n > 0 ? n : 1));
    if (!buf0 || !buf1) { free(buf0); free(buf1); return GFO_BAD_ARG; }
    memcpy(buf0, x, sizeof(u64) * (size_t)n);
    u64 *in = buf0, *ob = buf1;
    i64 m = 1;
    for (i64 index = 0; index < n_factors; index++) {
        i64 r = factors[n_factors - 1 - index];
        i64 q = n / (m * r);
        u64 twiddle = 1, twiddle_step;
        if (f_pow(f, omega, q, &twiddle_step)) { free(buf0); free(buf1); return GFO_ZERO_DIVISION; }
        /* in_view[k, qi, b] = in[(k*q + qi)*m + b] ; out_view[qi, f, b] = ob[(qi*r + f)*m + b] */
        if (r == 2) {
            for (i64 b = 0; b < m; b++) {
                for (i64 qi = 0; qi < q; qi++) {
                    u64 x0 = in[(0 * q + qi) * m + b];
                    u64 x1 = f_mul(f, in[(1 * q + qi) * m + b], twiddle);
                    ob[(qi * 2 + 0) * m + b] = f_add(f, x0, x1);
                    ob[(qi * 2 + 1) * m + b] = f_sub(f, x0, x1);
                }
                twiddle = f_mul(f, twiddle, twiddle_step);
            }
        } else {
            for (i64 ff = 0; ff < r; ff++) {
                for (i64 b = 0; b < m; b++) {
                    for (i64 qi = 0; qi < q; qi++) {
                        u64 acc = in[((r - 1) * q + qi) * m + b];
                        for (i64 k = r - 2; k >= 0; k--) acc = f_add(f, f_mul(f, acc, twiddle), in[(k * q + qi) * m + b]);
                        ob[(qi * r + ff) * m + b] = acc;
                    }
                    twiddle = f_mul(f, twiddle, twiddle_step);
                }
            }
        }
        m *= r;
        u64 *t = in; in = ob; ob = t;
    }
    memcpy(out, in, sizeof(u64) * (size_t)n);
    free(buf0); free(buf1);
    return GFO_OK;
}

/* uint32-typed radix-2 NTT over a prime field for the timed CPU baseline: identical stage order and
 * arithmetic to gfo_ntt (r == 2 branch) without the u64 widening.  n must be a power of two. */
int gfo_ntt_u32_pow2(u64 p, const uint32_t *x, i64 n, u64 omega, uint32_t *out)
{
    uint32_t *buf0 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    uint32_t *buf1 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    if (!buf0 || !buf1) { free(buf0); free(buf1); return GFO_BAD_ARG; }
    memcpy(buf0, x, sizeof(uint32_t) * (size_t)n);
    uint32_t *in = buf0, *ob = buf1;
    i64 m = 1;
    while (m < n) {
        i64 q = n / (m * 2);
        u64 step = 1, base = omega;
        for (i64 e = q; e > 0; e >>= 1) { /* omega^q by square-and-multiply (value-identical to POWER) */
            if (e & 1) step = step * base % p;
            base = base * base % p;
        }
        u64 tw = 1;
        for (i64 b = 0; b < m; b++) {
            for (i64 qi = 0; qi < q; qi++) {
                u64 x0 = in[qi * m + b];
                u64 x1 = (u64)in[(q + qi) * m + b] * tw % p;
                u64 s = x0 + x1; if (s >= p) s -= p;
                u64 d = x0 >= x1 ? x0 - x1 : p + x0 - x1;
                ob[(qi * 2) * m + b] = (uint32_t)s;
                ob[(qi * 2 + 1) * m + b] = (uint32_t)d;
            }
            tw = tw * step % p;
        }
        m *= 2;
        uint32_t *t = in; in = ob; ob = t;
    }
    memcpy(out, in, sizeof(uint32_t) * (size_t)n);
    free(buf0); free(buf1);
    return GFO_OK;
}

/* matmul_jit.implementation _domains/_linalg.py:286-308 : C[i,j] = sum_k A[i,k]*B[k,j] (batch of 1) */
void gfo_matmul(const gfo_field *f, const u64 *A, const u64 *B, u64 *C, i64 M, i64 K, i64 N)
{
    for (i64 i = 0; i < M; i++)
        for (i64 j = 0; j < N; j++) {
            u64 acc = 0;
            for (i64 k = 0; k < K; k++) acc = f_add(f, acc, f_mul(f, A[i * K + k], B[k * N + j]));
            C[i * N + j] = acc;
        }
}

/* evaluate_elementwise_jit.implementation _polys/_dense.py:432-440 (coeffs descending) */
static void poly_eval(const gfo_field *f, const u64 *coeffs, i64 nc, const u64 *values, i64 nv, u64 *y)
{
    for (i64 i = 0; i < nv; i++) {
        u64 acc = coeffs[0];
        for (i64 j = 1; j < nc; j++) acc = f_add(f, coeffs[j], f_mul(f, acc, values[i]));
        y[i] = acc;
    }
}
void gfo_poly_eval(const gfo_field *f, const u64 *coeffs, i64 nc, const u64 *values, i64 nv, u64 *y)
{
    poly_eval(f, coeffs, nc, values, nv, y);
}

/* convolve_jit.implementation _domains/_function.py:141-167 (exact integer result either branch) */
static void convolve(const gfo_field *f, const u64 *a, i64 na, const u64 *b, i64 nb, u64 *c)
{
    for (i64 i = 0; i < na + nb - 1; i++) c[i] = 0;
    for (i64 i = 0; i < na; i++)
        for (i64 j = nb - 1; j >= 0; j--) c[i + j] = f_add(f, c[i + j], f_mul(f, a[i], b[j]));
}
void gfo_convolve(const gfo_field *f, const u64 *a, i64 na, const u64 *b, i64 nb, u64 *c)
{
    convolve(f, a, na, b, nb, c);
}

/* berlekamp_massey_jit.implementation _lfsr.py:1647-1702.  S ascending; returns C(x) ascending in
 * `C` (caller-provided, length n) and its trimmed length in *len (the reference returns it reversed). */
static void berlekamp_massey(const gfo_field *f, const u64 *S, i64 n, u64 *C, i64 *len)
{
    u64 B[512], T[512];
    for (i64 i = 0; i < n; i++) { C[i] = 0; B[i] = 0; }
    C[0] = 1; B[0] = 1;
    i64 L = 0, m = 1;
    u64 b = 1;
    for (i64 k = 0; k < n; k++) {
        u64 d = 0;
        for (i64 i = 0; i <= L; i++) {
            /* S[k - i]: NumPy negative indices wrap; with i <= L <= k they never go negative here */
            i64 idx = k - i;
            if (idx < 0) idx += n;
            d = f_add(f, d, f_mul(f, S[idx], C[i]));
        }
        if (d == 0) {
            m += 1;
        } else {
            u64 binv, d_over_b;
            f_recip(f, b, &binv);
            d_over_b = f_mul(f, d, binv);
            if (2 * L > k) {
                for (i64 i = m; i < n; i++) C[i] = f_sub(f, C[i], f_mul(f, d_over_b, B[i - m]));
                m += 1;
            } else {
                memcpy(T, C, sizeof(u64) * (size_t)n);
                for (i64 i = m; i < n; i++) C[i] = f_sub(f, C[i], f_mul(f, d_over_b, B[i - m]));
                L = k + 1 - L;
                memcpy(B, T, sizeof(u64) * (size_t)n);
                b = d;
                m = 1;
            }
        }
    }
    i64 clen = L + 1;
    if (clen > n) clen = n; /* C[: L + 1] on a length-n array */
    i64 last = -1;
    for (i64 i = 0; i < clen; i++)
        if (C[i] != 0) last = i;
    *len = last >= 0 ? last + 1 : 1;
}
void gfo_berlekamp_massey(const gfo_field *f, const u64 *S, i64 n, u64 *C, i64 *len)
{
    berlekamp_massey(f, S, n, C, len);
}

/* bch_decode_jit.implementation _codes/_bch.py:1337-1578 with base field == extension field
 * (reed_solomon_decode_jit, _codes/_reed_solomon.py:1105-1113).
 * codewords: (N, n) row-major, index 0 = highest degree.  erasures: (N, n) bytes (0/1) or NULL.
 * dec_codewords: (N, n) out.  n_errors: (N) out. */
static int decode_impl(const gfo_field *f, u64 base_p, const u64 *codewords, const uint8_t *erasures, i64 N, i64 n,
                       i64 design_n, u64 alpha, i64 c, const u64 *roots, i64 n_roots, u64 *dec_codewords, i64 *n_errors)
{
    i64 d = n_roots + 1;
    if (n < 1 || d > 256) return GFO_BAD_ARG;
    /* per-codeword buffers sized by the codeword length (codes over fields above 256 elements: n up to 2^20 - 1) */
    const size_t cap = (size_t)(n > 1024 ? n : 1024);
    u64 *received = malloc(sizeof(u64) * cap), *tmp = malloc(sizeof(u64) * cap), *codeword = malloc(sizeof(u64) * cap);
    i64 *epos = malloc(sizeof(i64) * cap);
    if (!received || !tmp || !codeword || !epos) { free(received); free(tmp); free(codeword); free(epos); return GFO_BAD_ARG; }
    u64 syndrome[256], gamma[512], gtmp[512], sprime[1024], lambda_[512], ltotal[1024];
    u64 omega_p[1024], ltp[1024];
    i64 error_positions[512];
    u64 error_locators_inv[512], error_values[512];

    memcpy(dec_codewords, codewords, sizeof(u64) * (size_t)(N * n));
    for (i64 ni = 0; ni < N; ni++) {
        n_errors[ni] = 0;
        const u64 *cw = codewords + ni * n;
        i64 u = 0;
        for (i64 i = 0; i < n; i++) received[i] = cw[n - 1 - i]; /* ascending degrees */
        if (erasures)
            for (i64 i = 0; i < n; i++)
                if (erasures[ni * n + (n - 1 - i)]) epos[u++] = i;
        for (i64 k = 0; k < u; k++) received[epos[k]] = 0;
        if (u > d - 1) { n_errors[ni] = -1; continue; }

        /* 1. syndromes */
        for (i64 i = 0; i < n; i++) tmp[i] = received[n - 1 - i];
        poly_eval(f, tmp, n, roots, n_roots, syndrome);
        int all_zero = 1;
        for (i64 i = 0; i < n_roots; i++) if (syndrome[i]) { all_zero = 0; break; }
        if (all_zero && u == 0) continue;

        /* 2. erasure locator */
        i64 glen = 1;
        gamma[0] = 1;
        for (i64 k = 0; k < u; k++) {
            u64 Yk, fac[2];
            f_pow(f, alpha, epos[k], &Yk);
            fac[0] = 1; fac[1] = f_sub(f, 0, Yk);
            convolve(f, gamma, glen, fac, 2, gtmp);
            glen += 1;
            memcpy(gamma, gtmp, sizeof(u64) * (size_t)glen);
        }

        /* 3. modified syndromes mod x^(d-1) */
        convolve(f, gamma, glen, syndrome, n_roots, sprime);
        i64 splen = glen + n_roots - 1;
        if (splen > d - 1) splen = d - 1;

        /* 4. Berlekamp-Massey on S'[u:] */
        i64 llen;
        if (u < d - 1) {
            berlekamp_massey(f, sprime + u, splen - u, lambda_, &llen);
        } else {
            lambda_[0] = 1; llen = 1;
        }
        i64 v = llen - 1;
        if (2 * v + u > d - 1) { n_errors[ni] = -1; continue; }

        /* 5. total locator */
        convolve(f, gamma, glen, lambda_, llen, ltotal);
        i64 ltlen = glen + llen - 1;
        i64 L_total = ltlen - 1;

        /* 6. Chien search */
        i64 v_total = 0;
        for (i64 i = 0; i < ltlen; i++) tmp[i] = ltotal[ltlen - 1 - i];
        for (i64 i = 0; i < design_n; i++) {
            u64 Xi_inv, val;
            f_pow(f, alpha, -i, &Xi_inv);
            poly_eval(f, tmp, ltlen, &Xi_inv, 1, &val);
            if (val == 0) {
                if (i >= n) { n_errors[ni] = -1; continue; }
                error_positions[v_total] = i;
                error_locators_inv[v_total] = Xi_inv;
                v_total++;
            }
        }
        if (v_total != v + u) { n_errors[ni] = -1; continue; }

        /* 7. evaluator */
        convolve(f, lambda_, llen, sprime, splen, omega_p);
        i64 oplen = llen + splen - 1;
        if (oplen > d - 1) oplen = d - 1;

        /* 8. derivative */
        for (i64 j = 1; j <= L_total; j++) ltp[j - 1] = f_mul(f, (u64)j % f->p, ltotal[j]);

        /* 9. Forney */
        int forney_fail = 0;
        for (i64 k = 0; k < v_total; k++) {
            u64 num, den, den_inv, Ej, pw;
            for (i64 i = 0; i < oplen; i++) tmp[i] = omega_p[oplen - 1 - i];
            poly_eval(f, tmp, oplen, &error_locators_inv[k], 1, &num);
            if (L_total > 0) {
                for (i64 i = 0; i < L_total; i++) tmp[i] = ltp[L_total - 1 - i];
                poly_eval(f, tmp, L_total, &error_locators_inv[k], 1, &den);
            } else {
                den = 0;
            }
            if (den == 0) { forney_fail = 1; break; } /* unreachable when v_total == deg: roots are simple */
            f_recip(f, den, &den_inv);
            Ej = f_mul(f, num, den_inv);
            f_pow(f, error_locators_inv[k], c - 1, &pw);
            Ej = f_mul(f, Ej, pw);
            Ej = f_sub(f, 0, Ej);
            error_values[k] = Ej;
        }
        if (forney_fail) { n_errors[ni] = -1; continue; }

        /* 10. correct */
        memcpy(codeword, received, sizeof(u64) * (size_t)n);
        for (i64 k = 0; k < v_total; k++) {
            i64 pos = error_positions[k];
            if (base_p == 0) {
                codeword[pos] = f_sub(f, codeword[pos], error_values[k]);
            } else if (base_p == 2) {
                codeword[pos] ^= error_values[k]; /* GF(2) subtract = np.bitwise_xor (_fields/_gf2.py) */
            } else {
                /* SUBTRACT_BASE = subtract_modular of the prime base field on int64 (_calculate.py:235-251) */
                i64 a = (i64)codeword[pos], b = (i64)error_values[k];
                codeword[pos] = (u64)(a >= b ? a - b : (i64)base_p + a - b);
            }
        }
        for (i64 i = 0; i < n; i++) dec_codewords[ni * n + i] = codeword[n - 1 - i];
        n_errors[ni] = v;
    }
    free(received); free(tmp); free(codeword); free(epos);
    return GFO_OK;
}

int gfo_rs_decode(const gfo_field *f, const u64 *codewords, const uint8_t *erasures, i64 N, i64 n, i64 design_n,
                  u64 alpha, i64 c, const u64 *roots, i64 n_roots, u64 *dec_codewords, i64 *n_errors)
{
    return decode_impl(f, 0, codewords, erasures, N, n, design_n, alpha, c, roots, n_roots, dec_codewords, n_errors);
}

/* bch_decode_jit with base field GF(p) != extension field `f` = GF(p^m) (_codes/_bch.py:1255-1578): identical steps,
 * corrections through SUBTRACT_BASE (:1310, :1573). */
int gfo_bch_decode(const gfo_field *f, u64 base_p, const u64 *codewords, const uint8_t *erasures, i64 N, i64 n,
                   i64 design_n, u64 alpha, i64 c, const u64 *roots, i64 n_roots, u64 *dec_codewords, i64 *n_errors)
{
    return decode_impl(f, base_p, codewords, erasures, N, n, design_n, alpha, c, roots, n_roots, dec_codewords, n_errors);
}

/* Poly.Roots + _poly_to_generator_matrix (systematic) _codes/_cyclic.py:198-226 and
 * ReedSolomon.__init__ _codes/_reed_solomon.py:206-207.
 * Outputs: roots[d-1] = alpha^(c+i); gpoly[d] descending (monic); P (k x (n-k)) row-major. */
void gfo_rs_construct(const gfo_field *f, i64 n, i64 k, u64 alpha, i64 c, u64 *roots, u64 *gpoly, u64 *P)
{
    i64 nk = n - k;
    u64 g[512], t[512];
    i64 glen = 1;
    g[0] = 1; /* ascending */
    for (i64 i = 0; i < nk; i++) {
        f_pow(f, alpha, c + i, &roots[i]);
        u64 fac[2];
        fac[0] = f_sub(f, 0, roots[i]); fac[1] = 1; /* (x - root), ascending */
        convolve(f, g, glen, fac, 2, t);
        glen += 1;
        memcpy(g, t, sizeof(u64) * (size_t)glen);
    }
    for (i64 i = 0; i < glen; i++) gpoly[i] = g[glen - 1 - i]; /* descending */
    if (nk == 0) return;
    /* _cyclic.py:211-218: P[0,:] = coeffs[0:-1] / coeffs[-1] (coeffs descending), then each row is the
     * previous one shifted right by one, minus P[i-1,-1] * P[0,:]. */
    u64 g0_inv;
    f_recip(f, gpoly[nk], &g0_inv);
    for (i64 j = 0; j < nk; j++) P[j] = f_mul(f, gpoly[j], g0_inv);
    for (i64 i = 1; i < k; i++) {
        u64 *row = P + i * nk;
        const u64 *prev = P + (i - 1) * nk;
        row[0] = 0;
        for (i64 j = 1; j < nk; j++) row[j] = prev[j - 1];
        if (prev[nk - 1] > 0)
            for (i64 j = 0; j < nk; j++) row[j] = f_sub(f, row[j], f_mul(f, prev[nk - 1], P[j]));
    }
}

/* _LinearCode._encode_message (systematic) _codes/_linear.py:270-284: parity = message @ G[pad:, k:],
 * codeword = hstack(message, parity).  message: (N, ks) with ks <= k; P: (k, n-k).  Byte-typed for the
 * timed CPU baseline; arithmetic via the field's lookup tables exactly as matmul_jit does. */
int gfo_rs_encode_u8(const gfo_field *f, const uint8_t *msg, i64 N, i64 ks, i64 k, i64 nk, const uint8_t *P,
                     uint8_t *codewords)
{
    i64 pad = k - ks, ns = ks + nk;
    for (i64 i = 0; i < N; i++) {
        const uint8_t *mrow = msg + i * ks;
        uint8_t *crow = codewords + i * ns;
        memcpy(crow, mrow, (size_t)ks);
        for (i64 j = 0; j < nk; j++) {
            u64 acc = 0;
            for (i64 t = 0; t < ks; t++) acc = f_add(f, acc, f_mul(f, mrow[t], P[(pad + t) * nk + j]));
            crow[ks + j] = (uint8_t)acc;
        }
    }
    return GFO_OK;
}

/* byte-typed wrapper around gfo_rs_decode for the timed CPU baseline (widen, decode, narrow) */
int gfo_rs_decode_u8(const gfo_field *f, const uint8_t *codewords, const uint8_t *erasures, i64 N, i64 n,
                     i64 design_n, u64 alpha, i64 c, const u64 *roots, i64 n_roots, uint8_t *dec, i64 *n_errors)
{
    u64 *in = (u64 *)malloc(sizeof(u64) * (size_t)(N * n));
    u64 *out = (u64 *)malloc(sizeof(u64) * (size_t)(N * n));
    if (!in || !out) { free(in); free(out); return GFO_BAD_ARG; }
    for (i64 i = 0; i < N * n; i++) in[i] = codewords[i];
    int rc = gfo_rs_decode(f, in, erasures, N, n, design_n, alpha, c, roots, n_roots, out, n_errors);
    for (i64 i = 0; i < N * n; i++) dec[i] = (uint8_t)out[i];
    free(in); free(out);
    return rc;
}
