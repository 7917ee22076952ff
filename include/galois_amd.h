/*
 * galois_amd.h -- C-ABI of the MI355X (gfx950) finite-field engine.
 *
 * The reference (mhostetter/galois) is pure Python with no FFI; its internal operator seam is the set
 * of per-field dispatcher objects that __array_ufunc__/__array_function__ call with int64 views of the
 * data.  Each entry point below names the reference interface it replaces
 * (paths relative to /root/reference/src/galois).  INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add at those seams.
 *
 * Conventions
 *  - All array pointers are DEVICE pointers owned by the caller (e.g. torch tensors); nothing is copied
 *    to or from the host by the data-path calls.  `stream` is a hipStream_t (NULL = default stream).
 *  - Elements are non-negative integers < order stored in `dtype` (an unsigned storage width; the
 *    reference's signed dtypes of the same width hold the same bit patterns).
 *  - Every function returns a gfa_status.  No C++ exceptions cross the boundary.
 *  - Arithmetic errors raised by the reference from inside its scalar kernels (ZeroDivisionError) are
 *    reported through `dev_err`, a caller-owned device int32 that kernels OR GFA_DEVERR_* bits into;
 *    the host checks it after synchronising and raises the reference's exception type.
 *  - A gfa_field_t is immutable after creation (tables, constants) => safe to share between threads and
 *    streams.  Unlike the reference, no module-level globals are involved (_domains/_ufunc.py:110,137).
 */
#ifndef GALOIS_AMD_H
#define GALOIS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFA_ABI_VERSION 1

typedef struct gfa_field gfa_field_t; /* one finite field GF(p^m) */
typedef struct gfa_rs gfa_rs_t;       /* one Reed-Solomon code over a field */
typedef void *gfa_stream_t;           /* hipStream_t */

typedef enum {
    GFA_OK = 0,
    GFA_ERR_INVALID = 1,     /* bad argument (ValueError/TypeError on the Python side) */
    GFA_ERR_UNSUPPORTED = 2, /* field / dtype / size combination has no device path */
    GFA_ERR_HIP = 3,         /* a HIP runtime call failed; see gfa_last_error() */
    GFA_ERR_NOMEM = 4
} gfa_status;

typedef enum { GFA_U8 = 0, GFA_U16 = 1, GFA_U32 = 2, GFA_U64 = 3 } gfa_dtype;

/* ufuncs overridden by the reference: _domains/_ufunc.py:616-631 */
typedef enum {
    GFA_OP_ADD = 0,   /* np.add        -> cls._add        (_fields/_ufunc.py:23,59,85) */
    GFA_OP_SUB = 1,   /* np.subtract   -> cls._subtract */
    GFA_OP_MUL = 2,   /* np.multiply   -> cls._multiply */
    GFA_OP_DIV = 3,   /* np.true_divide/floor_divide -> cls._divide (reciprocal+multiply, _ufunc.py:433-437) */
    GFA_OP_NEG = 4,   /* np.negative   -> cls._negative */
    GFA_OP_RECIP = 5, /* np.reciprocal -> cls._reciprocal */
    GFA_OP_POW = 6    /* np.power      -> cls._power */
} gfa_op;

/* FieldArray.compile(mode) (_domains/_array.py:322-362).  LOOKUP = EXP/LOG/Zech (or full product)
 * tables, the reference's "jit-lookup"; CALCULATE = explicit arithmetic, the reference's "jit-calculate". */
typedef enum { GFA_MODE_AUTO = 0, GFA_MODE_LOOKUP = 1, GFA_MODE_CALCULATE = 2 } gfa_mode;

#define GFA_DEVERR_ZERO_DIVISION 1 /* reciprocal(0), x/0, 0**negative (_lookup.py:194-195, _calculate.py:403-404,536-537) */
#define GFA_DEVERR_LOG_ZERO 4      /* log(0): ArithmeticError (_lookup.py:291-292, _calculate.py:613-614) */
#define GFA_DEVERR_LOG_BASE 8      /* log base that is not a primitive element (_calculate.py:621) */
#define GFA_DEVERR_NO_LU 2         /* lu_decompose needs a row exchange ("The LU decomposition of 'A' does not exist", _linalg.py:374) */

/* The library is built with -fvisibility=hidden: exactly the entry points declared between this push and the pop at the
 * end of the file are exported. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* ---- library -------------------------------------------------------------------------------------- */
int gfa_abi_version(void);
const char *gfa_last_error(void); /* thread-local description of the last non-OK status */
int gfa_device_count(void);       /* number of visible HIP devices (0 if none / no driver) */
/* Work buffers of the data-path calls (Reed-Solomon remainders, NTT intermediates, panel copies ...) come from a
 * library-owned stream-ordered pool per device, so that the reference's per-call scratch arrays (e.g. the int64 copies of
 * _codes/_bch.py:1281-1299, _domains/_function.py:201-202) cost no driver round trip.  The pool keeps at most
 * GFA_SCRATCH_KEEP_MB (default 256) MiB of freed blocks across synchronisations; this call synchronises the current
 * device and returns everything unused beyond keep_bytes to the driver (call it when the host framework's allocator
 * runs out of memory, e.g. next to torch.cuda.empty_cache()). */
int gfa_trim_scratch(uint64_t keep_bytes);

/* ---- fields: replaces the class factory's arithmetic set-up --------------------------------------- *
 * galois.GF(...) -> _GF_prime/_GF_extension (_fields/_factory.py:364-532) and
 * UFuncMixin._build_lookup_tables (_domains/_lookup.py:319-371).  The field is defined by exactly what
 * defines a reference field class: characteristic p, degree m, irreducible polynomial (m+1 coefficients
 * in GF(p), highest degree first; ignored when m == 1) and primitive element (integer representation).
 * Host-side only; device tables are uploaded lazily on the first data-path call. */
int gfa_field_create(uint64_t p, uint32_t m, const uint64_t *irreducible_poly_coeffs, uint64_t primitive_element,
                     gfa_field_t **out);
void gfa_field_destroy(gfa_field_t *f);
int gfa_field_set_mode(gfa_field_t *f, int mode); /* gfa_mode; GFA_ERR_UNSUPPORTED if the mode is illegal for the field */
int gfa_field_get_mode(const gfa_field_t *f);     /* resolved mode: GFA_MODE_LOOKUP or GFA_MODE_CALCULATE */
uint64_t gfa_field_order(const gfa_field_t *f);   /* p^m (0 if it does not fit in 64 bits) */
/* Host copies of the lookup tables in the reference's layout (int64: EXP 2q, LOG q, ZECH_LOG q entries;
 * cls._EXP/_LOG/_ZECH_LOG/_ZECH_E, _domains/_meta.py:60-63).  Any pointer may be NULL. */
int gfa_field_tables(gfa_field_t *f, int64_t *exp_out, int64_t *log_out, int64_t *zech_out, int64_t *zech_e_out);
/* One scalar operation on the host (class-construction-time maths: roots of unity, generator polynomials).
 * For GFA_OP_POW `b` is an int64 exponent passed as its two's-complement bits. */
int gfa_scalar(const gfa_field_t *f, int op, uint64_t a, uint64_t b, uint64_t *out);

/* ---- element-wise ufuncs: replaces `getattr(self.ufunc, "__call__")(*inputs)` (_domains/_ufunc.py:403) *
 * i.e. the numba.vectorize'd int64(int64,int64) loops of _lookup.py:31-270 and _calculate.py:133-592.
 * Strides are in elements and must be 0 (broadcast one scalar) or 1 (contiguous). */
int gfa_binary(gfa_field_t *f, int op, const void *a, int64_t a_stride, const void *b, int64_t b_stride, void *out,
               int64_t n, int dtype, gfa_stream_t stream, int32_t *dev_err);
int gfa_unary(gfa_field_t *f, int op, const void *a, void *out, int64_t n, int dtype, gfa_stream_t stream,
              int32_t *dev_err);
/* np.power(x, e): `exps` is a device int64 array (stride 0 or 1) -- power_ufunc, _ufunc.py:477-490 */
int gfa_power(gfa_field_t *f, const void *a, int64_t a_stride, const int64_t *exps, int64_t e_stride, void *out,
              int64_t n, int dtype, gfa_stream_t stream, int32_t *dev_err);
/* field * integer = repeated addition (multiply_ufunc.__call__, _ufunc.py:392-401): out = a * (k mod p) with the
 * integer taken in the prime subfield.  `ks` is a device int64 array (stride 0 or 1). */
int gfa_scalar_multiply(gfa_field_t *f, const void *a, int64_t a_stride, const int64_t *ks, int64_t k_stride, void *out,
                        int64_t n, int dtype, gfa_stream_t stream);
/* ufunc.reduce over the last axis of an (n_outer, n_inner) array for op in {ADD, SUB, MUL, DIV}
 * (_domains/_ufunc.py:180-198 allows reduce/accumulate only for the binary ops).  SUB/DIV are left folds. */
int gfa_reduce(gfa_field_t *f, int op, const void *a, void *out, int64_t n_outer, int64_t n_inner, int dtype,
               gfa_stream_t stream, int32_t *dev_err);

/* ufunc.accumulate over the last axis of an (n_outer, n_inner) array, same ops and fold conventions as gfa_reduce. */
/* ufunc.reduceat (dispatched like reduce, _domains/_ufunc.py:689): out[s] = fold of a[starts[s] : ends[s]] with the op; an
 * empty or reversed slice yields a[starts[s]] (NumPy's convention).  starts / ends: device int64 arrays of nseg entries. */
int gfa_reduceat(gfa_field_t *f, int op, const void *a, const int64_t *starts, const int64_t *ends, int64_t nseg, void *out, int dtype,
                 gfa_stream_t stream, int32_t *dev_err);
int gfa_accumulate(gfa_field_t *f, int op, const void *a, void *out, int64_t n_outer, int64_t n_inner, int dtype,
                   gfa_stream_t stream, int32_t *dev_err);
/* np.convolve(a, b), mode "full": out[k] = sum_i a[i] * b[k-i], na + nb - 1 outputs -- convolve_jit
 * (_domains/_function.py:111-167).  Direct O(na*nb) kernel for any field.  Prime fields below 2^32 with na*nb >= 2^22
 * (and at most 2^26 outputs) instead take the integer route the reference itself uses for prime fields
 * (`np.convolve(a, b) % p`, _function.py:141-150): residues modulo three 31-bit NTT primes, nine gfa_ntt transforms, CRT,
 * then mod p -- the same exact product.  (The host additionally routes fields that have the needed root of unity
 * themselves through three gfa_ntt calls + one gfa_binary multiply.) */
int gfa_convolve(gfa_field_t *f, const void *a, int64_t na, const void *b, int64_t nb, void *out, int dtype,
                 gfa_stream_t stream);

/* berlekamp_massey_jit `int64[:](int64[:])` (_lfsr.py:1627-1702): connection polynomial of the shortest LFSR generating each of
 * `batch` contiguous sequences of n terms.  out_coeffs: (batch, n) coefficients C_0 = 1, C_1, ... in ASCENDING degree, zero
 * padded; out_len[b]: number of coefficients after trimming (the reference returns them degree-descending). */
int gfa_berlekamp_massey(gfa_field_t *f, const void *seq, int64_t n, int64_t batch, void *out_coeffs, int64_t *out_len, int dtype,
                          gfa_stream_t stream);
/* FieldArray.vector (to_digits != 0: n field elements -> n*m digits of GF(p), degree m-1 first) and FieldArray.Vector
 * (to_digits == 0: n*m digits -> n elements) of _fields/_array.py:383-491; element and digit arrays may use different
 * storage widths. */
int gfa_vector(gfa_field_t *f, int to_digits, const void *in, int dtype_in, void *out, int dtype_out, int64_t n, gfa_stream_t stream);
/* evaluate_elementwise_jit `int64[:](int64[:] coeffs_desc, int64[:] x)` (_polys/_dense.py:404-440): out[i] =
 * poly(x[i]) by Horner's rule; `coeffs` holds ncoef coefficients, highest degree first, in device memory. */
int gfa_poly_evaluate(gfa_field_t *f, const void *coeffs, int64_t ncoef, const void *x, void *out, int64_t n, int dtype,
                      gfa_stream_t stream);
/* log_ufunc (_domains/_lookup.py:273-294; FieldArray.log _fields/_array.py:2127-2200): out[i] = log_base(a[i]) as
 * int64.  base == NULL: the field's primitive element.  Strides in {0, 1} as for gfa_binary.  a[i] == 0 ORs
 * GFA_DEVERR_LOG_ZERO, a non-primitive base GFA_DEVERR_LOG_BASE into *dev_err (both ArithmeticError in the reference).
 * Fields of order <= 2^20 read their LOG table.  Larger fields (the reference: log_pollard_rho / log_pohlig_hellman,
 * _calculate.py:630-755) run Pohlig-Hellman on the device and need gfa_log_prepare once: the prime factorisation of q - 1
 * (primes ascending, multiplicities; every prime at most 2^40).  Results of fields with q > 2^63 are uint64 bit patterns. */
int gfa_log_prepare(gfa_field_t *f, const uint64_t *primes, const uint32_t *multiplicities, int count);
int gfa_log(gfa_field_t *f, const void *a, int64_t a_stride, const void *base, int64_t base_stride, int64_t *out, int64_t n,
            int dtype, gfa_stream_t stream, int32_t *dev_err);

/* ---- Fields of order 2^64 <= q <= 2^128 (the reference's dtype=object fields: _fields/_ufunc.py:36-48 selects [np.object_],
 * _domains/_meta.py:39-41 the python-calculate mode; same scalar formulas _domains/_calculate.py:133-592) ----------------- *
 * Elements are two little-endian uint64 limbs, interleaved (element i at words 2i, 2i+1).  kind: 1 = GF(p), p >= 2^64
 * (Montgomery), 2 = GF(2^m), 64 < m <= 128 (m = 128: the modulus' x^128 term is implicit), 3 = GF(p^m) with p < 2^32.  `params` (27 uint64 words, computed by the host):
 * [0:2] p, [2] -p^-1 mod 2^64, [3:5] 2^256 mod p, [5:7] p - 2 (kind 1) or 2^m - 2 (kind 2), [7:9] the irreducible polynomial
 * without x^m (kind 2), [9:11] (q-1)/(p-1) - 1 (kind 3), [11:27] digits of the irreducible polynomial minus x^m, degree
 * m-1..0 (kind 3).  Strides are 0 (broadcast scalar) or 1.  gfa_wide_power takes exponents the host has reduced into
 * [0, q-1) (two limbs each) plus the SIGN of the original exponents (int8: -1, 0, 1), which decides the zero-base cases as
 * power_square_and_multiply does (_calculate.py:558-592).  dev_err: sticky GFA_DEVERR_ZERO_DIVISION word, may be NULL. */
typedef struct gfa_wfield gfa_wfield_t;
int gfa_wfield_create(int kind, uint32_t m, const uint64_t *params, gfa_wfield_t **out);
void gfa_wfield_destroy(gfa_wfield_t *w);
int gfa_wide_binary(gfa_wfield_t *w, int op, const void *a, int64_t sa, const void *b, int64_t sb, void *out, int64_t n,
                    gfa_stream_t stream, int32_t *dev_err);
int gfa_wide_unary(gfa_wfield_t *w, int op, const void *a, void *out, int64_t n, gfa_stream_t stream, int32_t *dev_err);
int gfa_wide_power(gfa_wfield_t *w, const void *a, int64_t sa, const void *exps, int64_t se, const int8_t *sign, void *out,
                   int64_t n, gfa_stream_t stream, int32_t *dev_err);
/* ufunc.reduce (accumulate = 0: n_outer results) / ufunc.accumulate (accumulate = 1: n_outer x n_inner prefixes) over the last axis
 * of an (n_outer, n_inner) array, op in {ADD, SUB, MUL, DIV}, left folds as the reference's object-dtype loops compute them
 * (_domains/_ufunc.py:686-689 with _fields/_ufunc.py:36-48); np.convolve(a, b) "full" (_domains/_function.py:141-167: the
 * object-dtype branch of convolve_jit); batched C = A B (_domains/_linalg.py:286-308), batch strides in elements, 0 = broadcast. */
int gfa_wide_reduce(gfa_wfield_t *w, int op, const void *a, void *out, int64_t n_outer, int64_t n_inner, int accumulate,
                    gfa_stream_t stream, int32_t *dev_err);
int gfa_wide_convolve(gfa_wfield_t *w, const void *a, int64_t na, const void *b, int64_t nb, void *out, gfa_stream_t stream);
int gfa_wide_matmul(gfa_wfield_t *w, const void *a, const void *b, void *out, int64_t batch, int64_t M, int64_t K, int64_t N,
                    int64_t a_bstride, int64_t b_bstride, gfa_stream_t stream);
/* Elimination on these fields (r05): row_reduce_jit (_domains/_linalg.py:315-351), lu_decompose_jit / plu_decompose_jit (:354-424),
 * det_jit (:447-477) -- arguments, pivot rule, L / P conventions and the GFA_DEVERR_NO_LU flag exactly as gfa_row_reduce /
 * gfa_plu_decompose below, on (batch, m, n) stacks of two-limb elements.  gfa_wide_poly_evaluate: Horner evaluation of one polynomial
 * (coefficients in descending degree) at n points, evaluate_elementwise_jit (_polys/_dense.py:404-423). */
int gfa_wide_row_reduce(gfa_wfield_t *w, void *a, int64_t batch, int64_t m, int64_t n, int64_t ncols, int64_t *rank_out, gfa_stream_t stream);
int gfa_wide_plu_decompose(gfa_wfield_t *w, void *a, void *l_out, void *p_out, int64_t batch, int64_t m, int64_t n, int pivoting,
                           int64_t *nperm_out, void *det_out, gfa_stream_t stream, int32_t *dev_err);
int gfa_wide_poly_evaluate(gfa_wfield_t *w, const void *coeffs, int64_t ncoef, const void *x, void *out, int64_t n, gfa_stream_t stream);

/* ---- Fields of order above 2^128 (r05; the reference has no upper bound: _domains/_meta.py:38-41, same scalar formulas) --------- *
 * Elements are nl = 4, 8 or 16 little-endian uint64 limbs, interleaved.  Kinds as gfa_wfield_create (1 GF(p), 2 GF(2^m) with
 * m <= 64 nl, 3 GF(p^m) with p < 2^32 and m <= 32).  `params` (113 uint64 words, computed by the host): [0:16] p, [16] -p^-1 mod 2^64,
 * [17:33] 2^(128 nl) mod p, [33:49] p - 2 (kind 1) or 2^m - 2 (kind 2), [49:65] the irreducible polynomial without x^m (kind 2),
 * [65:81] (q-1)/(p-1) - 1 (kind 3), [81:113] digits of the irreducible polynomial minus x^m, degree m-1..0 (kind 3).
 * gfa_big_binary / unary / power: the element-wise ufuncs with the argument meaning of gfa_wide_binary / unary / power. */
typedef struct gfa_bfield gfa_bfield_t;
int gfa_bfield_create(int kind, uint32_t m, uint32_t nl, const uint64_t *params, gfa_bfield_t **out);
void gfa_bfield_destroy(gfa_bfield_t *w);
int gfa_big_binary(gfa_bfield_t *w, int op, const void *a, int64_t sa, const void *b, int64_t sb, void *out, int64_t n,
                   gfa_stream_t stream, int32_t *dev_err);
int gfa_big_unary(gfa_bfield_t *w, int op, const void *a, void *out, int64_t n, gfa_stream_t stream, int32_t *dev_err);
int gfa_big_power(gfa_bfield_t *w, const void *a, int64_t sa, const void *exps, int64_t se, const int8_t *sign, void *out,
                  int64_t n, gfa_stream_t stream, int32_t *dev_err);

/* ---- NTT: replaces fft_jit/ifft_jit `self.jit(x.astype(int64), int64(omega), factors)` (_domains/_function.py:201) *
 * Computes out[k] = sum_j in[j] * omega^(j*k) for each of `batch` contiguous length-n rows, natural order in and
 * out.  `omega` must be a primitive n-th root of unity (for the inverse pass omega^-1, as fft_jit.__call__ does at
 * _function.py:194-195).  If scale_by_n_inverse != 0 the result is multiplied by (n mod p)^-1 (_function.py:209-210).
 * in == out is allowed.  Any n with n | q-1 is accepted; powers of two take the LDS radix path. */
int gfa_ntt(gfa_field_t *f, const void *in, void *out, int64_t n, int64_t batch, uint64_t omega, int scale_by_n_inverse,
            int dtype, gfa_stream_t stream);
/* gfa_ntt on rows that live in per-peer chunks -- the send / receive buffers of the distributed transform's all-to-all, so
 * that no re-layout pass is needed on either side of the exchange (new design).  Element j of row b
 * is at (j / chunk_len) * chunk_stride + b * row_stride + (j % chunk_len) (elements); chunk_len == 0 selects the plain
 * contiguous layout (row b at b * n) for that side.  Power-of-two 4 <= n <= 2^20 over a prime field, native device width,
 * in != out.  Returns GFA_ERR_UNSUPPORTED (nothing launched) when a chunk is shorter than the kernel's access granule. */
int gfa_ntt_chunked(gfa_field_t *f, const void *in, void *out, int64_t n, int64_t batch, uint64_t omega, int scale_by_n_inverse,
                    int64_t in_chunk_len, int64_t in_chunk_stride, int64_t in_row_stride, int64_t out_chunk_len,
                    int64_t out_chunk_stride, int64_t out_row_stride, int dtype, gfa_stream_t stream);
/* Per-rank kernel of the distributed four-step transform of ONE length-n_total sequence over G GPUs (n_total =
 * n1 * n2, all powers of two; new design -- the reference has no distributed path).
 * The rank holds `cols` adjacent columns [col0, col0+cols) of the (n1 x n2) row-major view x[j1*n2 + j2] as a local
 * (n1 x cols) row-major array.  Each local column is transformed (length n1, root omega^(n_total/n1)) and multiplied
 * by omega^((col0 + c) * k1).  The host then performs the single RCCL all-to-all (torch.distributed) that turns the
 * column-block layout into a row-block layout, and finishes with a plain batched gfa_ntt of length n2 on its rows
 * (root omega^n1).  Result layout: rank g holds X[k1 + n1*k2] for its n1/G values of k1, all k2.
 * dtype must be the field's native device width (GFA_U32 for p < 2^32, GFA_U64 otherwise). */
int gfa_ntt_columns(gfa_field_t *f, const void *in, void *out, int64_t n1, int64_t cols, int64_t col0, int64_t n_total,
                    uint64_t omega, int dtype, gfa_stream_t stream);
/* The same column pass on a SUB-BLOCK of the rank's columns: `in` points at column col0 - (rank's first column) of the rank's
 * (n1 x in_pitch) array, `cols` columns are transformed and written as an (n1 x out_pitch) array (pitches in elements, 0 = cols).
 * Splitting the column pass lets the exchange of one sub-block run while the next is being transformed (gfa_ntt_dist does). */
int gfa_ntt_columns_pitched(gfa_field_t *f, const void *in, int64_t in_pitch, void *out, int64_t out_pitch, int64_t n1, int64_t cols,
                            int64_t col0, int64_t n_total, uint64_t omega, int dtype, gfa_stream_t stream);
/* Per-rank LAST kernel of the distributed INVERSE transform (ifft_jit semantics, _domains/_function.py:387-392, spread
 * over G GPUs).  The inverse runs the forward steps backwards: a plain batched gfa_ntt of length n2 (root omega^n1) on the
 * rank's row block, the one all-to-all back to column blocks, then THIS call on the local (n1 x cols) array: element
 * (k1, c) is first multiplied by omega^(k1 * (col0 + c)), then every column is transformed (length n1, root
 * omega^(n_total/n1)), optionally scaled by (n_total mod p)^-1.  `omega` is the root of the inverse transform (w^-1 of
 * the forward one).  Output: x[j1*n2 + col0 + c] at (j1, c) -- the column-block layout gfa_ntt_columns consumes.
 * n1 <= 2^10; dtype as for gfa_ntt_columns. */
int gfa_ntt_columns_inv(gfa_field_t *f, const void *in, void *out, int64_t n1, int64_t cols, int64_t col0, int64_t n_total,
                        uint64_t omega, int scale_by_n_total_inverse, int dtype, gfa_stream_t stream);

/* The whole distributed transform, collective included (SURVEY.md section 8(b) `gf_ntt_dist(comm, ...)`; semantics of
 * fft_jit / ifft_jit, _domains/_function.py:246-392, for ONE sequence of n1 * n2 points spread over `world` GPUs, one
 * process per GPU).  `nccl_comm` is the caller's ncclComm_t (RCCL; bound with dlsym at first use, so the library has no
 * link-time dependency on it).  Forward: local_cols = the rank's (n1 x n2/world) column block of the row-major (n1 x n2)
 * view, out_rows = its (n1/world x n2) block of the result, X[k1 + n1*k2] at [k1 - rank*n1/world][k2]; steps:
 * gfa_ntt_columns, ONE all-to-all over xGMI, gfa_ntt_chunked on the receive buffer.  When RCCL's send / recv are available the
 * column pass runs in two sub-blocks and the exchange of the first (grouped ncclSend / ncclRecv on a side stream: still one logical
 * all-to-all, in two halves) overlaps the transform of the second.  The inverse consumes that row-block
 * layout and returns the column-block layout (`omega` is the FORWARD root in both calls).  n1 <= 2^10 for the inverse,
 * n2 <= 2^20; dtype = the field's native device width; input and output buffers must differ. */
int gfa_ntt_dist(gfa_field_t *f, void *nccl_comm, int rank, int world, const void *local_cols, void *out_rows, int64_t n1, int64_t n2,
                 uint64_t omega, int dtype, gfa_stream_t stream);
int gfa_intt_dist(gfa_field_t *f, void *nccl_comm, int rank, int world, const void *local_rows, void *out_cols, int64_t n1, int64_t n2,
                  uint64_t omega, int scale_by_n_inverse, int dtype, gfa_stream_t stream);

/* ---- Field linear algebra ------------------------------------------------------- *
 * gfa_matmul replaces matmul_jit.implementation `int64[:,:,:](int64[:,:,:], int64[:,:,:])` (_domains/_linalg.py:283-308)
 * and the BLAS-then-mod-p shortcut of prime fields (_lapack_linalg, :21-75): out[b] = a[b] @ b[b] for `batch` row-major
 * matrices, a: (M x K), b: (K x N), out: (M x N) contiguous.  a_batch_stride / b_batch_stride are in elements; 0
 * broadcasts one matrix over the batch (np.matmul broadcasting, _linalg.py:232-246). */
int gfa_matmul(gfa_field_t *f, const void *a, const void *b, void *out, int64_t batch, int64_t M, int64_t K, int64_t N,
               int64_t a_batch_stride, int64_t b_batch_stride, int dtype, gfa_stream_t stream);
/* row_reduce_jit.__call__ (_linalg.py:315-351): Gauss-Jordan elimination IN PLACE on `batch` contiguous (m x n)
 * matrices over the first `ncols` columns; pivot = first non-zero entry at or below the pivot row.  rank_out[b] =
 * number of pivots.  inv_jit / solve_jit / matrix_rank_jit and the row/column/null spaces are host compositions of
 * this call (:480-548, _fields/_array.py:1541-1760).  At most 4096 rows. */
int gfa_row_reduce(gfa_field_t *f, void *a, int64_t batch, int64_t m, int64_t n, int64_t ncols, int64_t *rank_out, int dtype,
                   gfa_stream_t stream);
/* lu_decompose_jit (pivoting == 0, _linalg.py:354-384) / plu_decompose_jit (pivoting != 0, :387-424) on `batch`
 * (m x n) matrices: `a` is overwritten with U; l_out (m x m) and p_out (m x m, the ROW permutation matrix -- the
 * reference returns its transpose) may be NULL.  n_permutations_out (may be NULL): row exchanges per matrix.
 * det_out (may be NULL): (-1)^exchanges * prod(diag U), the value det_jit computes (:447-477).  Without pivoting a
 * zero pivot above a non-zero entry ORs GFA_DEVERR_NO_LU into *dev_err. */
int gfa_plu_decompose(gfa_field_t *f, void *a, void *l_out, void *p_out, int64_t batch, int64_t m, int64_t n, int pivoting,
                      int64_t *n_permutations_out, void *det_out, int dtype, gfa_stream_t stream, int32_t *dev_err);

/* ---- Reed-Solomon -------------------------------------------------------------------------------- *
 * gfa_rs_create replaces the arithmetic part of ReedSolomon.__init__ (_codes/_reed_solomon.py:111-218) and
 * _poly_to_generator_matrix (_codes/_cyclic.py:198-226): roots alpha^(c..c+d-2), g(x), systematic parity matrix.
 *
 * Symbol storage (`dtype` of the encode / detect / decode / extract calls below): codes whose (syndrome) field has at most 256
 * elements take GFA_U8 symbols and run on the byte kernels.  Codes over larger fields with EXP/LOG tables (q <= 2^20:
 * RS(1023, k) over GF(2^10), BCH(1023, k) over GF(2), RS over GF(3^6), ...) take GFA_U8 / GFA_U16 / GFA_U32 symbols, whichever
 * holds a symbol of the code's symbol field, and run on the table-driven kernels (same algorithm; d - 1 <= 254 roots). */
int gfa_rs_create(gfa_field_t *f, int64_t n, int64_t k, int64_t c, uint64_t alpha, int systematic, gfa_rs_t **out);
/* BCH(n, k) code over the prime field GF(p) with syndrome arithmetic in `ext` = GF(p^m): replaces the arithmetic part of
 * BCH.__init__ after the generator polynomial is known (_codes/_bch.py:106-240) -- roots alpha^c .. alpha^(c+d-2) in
 * GF(p^m), systematic parity matrix of g(x) (_codes/_cyclic.py:198-226).  `generator_poly`: n-k+1 coefficients in
 * GF(p), highest degree first, monic (the host computes it as the product of the distinct minimal polynomials of the
 * roots, _bch.py:1178-1197); every alpha^(c+i) is verified to be one of its roots.  The handle is used with the same
 * gfa_rs_encode / gfa_rs_detect / gfa_rs_decode / gfa_rs_extract_message entry points; decoding is the reference's
 * bch_decode_jit with SUBTRACT_BASE = GF(p) subtraction (_bch.py:1310, 1573). */
int gfa_bch_create(gfa_field_t *ext, uint64_t base_p, int64_t n, int64_t k, int64_t d, int64_t c, uint64_t alpha,
                   const uint64_t *generator_poly, int systematic, gfa_rs_t **out);
void gfa_rs_destroy(gfa_rs_t *code);
/* Host copies: roots (d-1; = n-k for Reed-Solomon), generator polynomial (n-k+1, highest degree first), parity matrix P (k x (n-k)). */
int gfa_rs_describe(const gfa_rs_t *code, uint64_t *roots, uint64_t *generator_poly, uint64_t *parity_matrix);
/* _LinearCode._encode_message -> matmul_jit (_codes/_linear.py:270-284, _domains/_linalg.py:286-308).
 * msg: (batch, ks) row-major, ks <= k (shortened codes pass fewer symbols).  out: (batch, ks + n - k) codewords, or
 * (batch, n - k) parity symbols when parity_only != 0 (systematic codes).  Non-systematic codes: codeword = m(x) g(x),
 * i.e. message @ G[pad:, pad:]. */
int gfa_rs_encode(gfa_rs_t *code, const void *msg, int64_t ks, void *out, int64_t batch, int parity_only, int dtype,
                  gfa_stream_t stream);
/* _CyclicCode._convert_codeword_to_message (_codes/_cyclic.py:129-138): the first ks symbols of a systematic codeword,
 * or the quotient codeword(x) / g(x) (divmod_jit) for a non-systematic code.  cw: (batch, ns); out_msg: (batch, ks). */
int gfa_rs_extract_message(gfa_rs_t *code, const void *cw, int64_t ns, void *out_msg, int64_t batch, int dtype,
                           gfa_stream_t stream);
/* _LinearCode._detect_errors (_codes/_linear.py:286-298): detected[i] = any(syndrome_i != 0). cw: (batch, ns). */
int gfa_rs_detect(gfa_rs_t *code, const void *cw, int64_t ns, uint8_t *detected, int64_t batch, int dtype,
                  gfa_stream_t stream);
/* bch_decode_jit.implementation (_codes/_bch.py:1337-1578) via reed_solomon_decode_jit (_reed_solomon.py:1105-1113).
 * recv: (batch, ns) received words, index 0 = highest degree; erasures: (batch, ns) bytes (non-zero = erased) or NULL;
 * out_codeword: (batch, ns) corrected codewords (the received row unchanged where decoding fails); may alias recv
 * (in-place decoding);
 * out_n_errors: (batch) int64, number of corrected errors (not erasures), -1 on failure.
 * BCH codes: a MIScorrected word can leave symbols outside GF(p) (the reference then raises on its field-membership check,
 * _bch.py:1300); such symbols are stored as values >= p (saturated for the narrow storage types) so that the host can detect
 * them with one comparison.  batch == 0 is accepted by every entry point of this section (no buffers are touched). */
int gfa_rs_decode(gfa_rs_t *code, const void *recv, const uint8_t *erasures, int64_t ns, void *out_codeword,
                  int64_t *out_n_errors, int64_t batch, int dtype, gfa_stream_t stream);

/* ---- measurement support (bench.py) --------------------------------------------------------------- *
 * Times `iters` back-to-back launches of the named hot kernel on `stream` with HIP events recorded on that same
 * stream and returns the average milliseconds per launch in *ms_out: about 50 ms of the same call run first, untimed
 * (table uploads, work-buffer pools, and the clocks -- after an idle phase the part needs tens of milliseconds of load
 * before a short kernel times at its steady rate), then three groups of `iters` launches; the median group is reported. */
int gfa_time_matmul(gfa_field_t *f, const void *a, const void *b, void *out, int64_t batch, int64_t M, int64_t K, int64_t N,
                    int dtype, gfa_stream_t stream, int iters, float *ms_out);
int gfa_time_binary(gfa_field_t *f, int op, const void *a, const void *b, void *out, int64_t n, int dtype,
                    gfa_stream_t stream, int iters, float *ms_out);
int gfa_time_unary(gfa_field_t *f, int op, const void *a, void *out, int64_t n, int dtype, gfa_stream_t stream,
                   int iters, float *ms_out);
int gfa_time_ntt(gfa_field_t *f, const void *in, void *out, int64_t n, int64_t batch, uint64_t omega, int dtype,
                 gfa_stream_t stream, int iters, float *ms_out);
int gfa_time_rs_encode(gfa_rs_t *code, const void *msg, int64_t ks, void *out, int64_t batch, int dtype,
                       gfa_stream_t stream, int iters, float *ms_out);
int gfa_time_rs_decode(gfa_rs_t *code, const void *recv, int64_t ns, void *out_codeword, int64_t *out_n_errors,
                       int64_t batch, int dtype, gfa_stream_t stream, int iters, float *ms_out);
/* Tuning aid of the GF(65537) one-pass transform (tools/fermat_phases.py): when `buf` is not NULL the kernel records eight
 * 100 MHz timestamps per (workgroup, round) there; NULL switches the recording off again.  Not part of the product path. */
void gfa_debug_fermat_stamps(unsigned long long *buf);
/* tuning aid of the signed-Montgomery NTT kernels (tools/m32_tune3.py): key 0 / 1 = line lengths (log2) of the first / second pass of the
 * three-pass form (0: default), 2 = non-temporal last pass (-1 by size, 0 never, 1 always), 3 = 1024-thread workgroups in the last pass */
void gfa_debug_m32_tune(int key, int value);
/* Test-only: the hand-assembled Berlekamp-Massey loop of the Reed-Solomon wave decoder (replaces berlekamp_massey_jit,
 * _lfsr.py:1647-1702, inside bch_decode_jit) against a compiler-generated loop of the same recurrence, on `nseq` syndrome
 * sequences derived from `seed` over a GF(2^8) field; *mismatches = sequences on which the two disagree (must be 0). */
int gfa_debug_rs_bm_selftest(gfa_field_t *f, int64_t nseq, uint64_t seed, int64_t *mismatches, gfa_stream_t stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* GALOIS_AMD_H */
