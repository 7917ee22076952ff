// gfa_rs_wide.hip -- Reed-Solomon / BCH encode, detect and decode for codes whose (syndrome) field has 256 < q <= 2^20
// elements: RS(1023, k) over GF(2^10), RS over GF(3^6), BCH(511 / 1023 / 4095 / ..., k) over GF(2) or GF(p), ...
//
// Same algorithm and the same reference anchors as gfa_rs.hip (bch_decode_jit.implementation, _codes/_bch.py:1337-1578;
// _LinearCode._encode_message / _detect_errors, _codes/_linear.py:270-298; _convert_codeword_to_message,
// _codes/_cyclic.py:129-138) with two differences in the execution model:
//   * field arithmetic goes through the field's EXP / LOG / ZECH_LOG tables in global memory (they do not fit the
//     64 KiB product-table scheme of the byte codes); characteristic 2 adds with XOR, prime fields add modulo p;
//   * symbols keep the caller's storage type (uint8 for BCH over GF(2) / GF(p), uint16 / uint32 for Reed-Solomon) and the
//     received word is never staged: syndromes and corrections read the input row, per-codeword polynomials (at most
//     d + 1 <= 256 coefficients) live in a per-wave LDS scratch of 32-bit words.
// One codeword per wavefront, as in the byte kernels.  This path exists for coverage and parity; it is not tuned.
#include <algorithm>

#include "gfa_internal.h"

using namespace gfa;

namespace {

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct WideParams {
    int n, k, nroots, c, p, base_p;
    u32 qm1, log_alpha;
};

constexpr int WIDE_MAX_D = 254; // longest root list (d - 1); polynomial updates run in 4 chunks of 64 lanes

struct WideScratch {
    u32 *synd, *gamma, *sprime, *C, *B, *ltotal, *omega, *ltp, *epos, *errpos, *errloc, *corr;
    static __host__ __device__ int words(int dd) { return 14 * (dd + 2); }
    __device__ void carve(u32 *p, int dd)
    {
        const int s = dd + 2;
        synd = p; p += s; gamma = p; p += s; sprime = p; p += s; C = p; p += s; B = p; p += s;
        ltotal = p; p += 2 * s; omega = p; p += s; ltp = p; p += 2 * s; epos = p; p += s; errpos = p; p += s;
        errloc = p; p += s; corr = p;
    }
};

__device__ __forceinline__ u32 wsum(const FieldDev &fd, u32 x)
{ // field sum over the 64 lanes, result in every lane
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x = Lut::add(fd, x, (u32)__shfl_xor((int)x, off));
    return x;
}

__device__ __forceinline__ u32 horner_asc(const FieldDev &fd, const u32 *co, int len, u32 x)
{ // coefficients ascending: acc = co[len-1]; acc = acc * x + co[i]
    u32 acc = co[len - 1];
    for (int i = len - 2; i >= 0; i--) acc = Lut::add(fd, Lut::mul(fd, acc, x), co[i]);
    return acc;
}

// Fields of at most 2^13 elements: the workgroup copies EXP / LOG (/ ZECH_LOG for odd characteristic) into LDS and the field
// descriptor is re-pointed at the copies, so that the dependent table gathers of every Horner step have LDS latency instead of
// a round trip to L2 (RS(1023,1003) decoding: 9.0 -> 14.1 M codewords/s).  `lds` = 0 keeps the global tables.  Returns the first free LDS word.
__device__ __forceinline__ u32 *stage_lut(FieldDev &fd, u32 *lds, int use_lds)
{
    if (!use_lds) return lds;
    const u32 q = fd.qm1 + 1;
    u32 *e = lds, *l = lds + 2 * q, *z = l + q;
    for (u32 i = threadIdx.x; i < 2 * q; i += blockDim.x) e[i] = fd.exp_tab[i];
    for (u32 i = threadIdx.x; i < q; i += blockDim.x) l[i] = fd.log_tab[i];
    const bool zech = fd.p != 2 && fd.m > 1;
    if (zech)
        for (u32 i = threadIdx.x; i < q; i += blockDim.x) z[i] = fd.zech_tab[i];
    __syncthreads();
    fd.exp_tab = e; fd.log_tab = l;
    if (zech) fd.zech_tab = z;
    return zech ? z + q : z;
}

// systematic parity: out = message @ P[pad:, :]
template <typename TS>
__global__ __launch_bounds__(256) void wide_encode_kernel(FieldDev fd, WideParams rp, const u32 *__restrict__ Pg,
                                                          const TS *__restrict__ msg, int ks, TS *__restrict__ out, i64 batch,
                                                          int parity_only, int use_lds)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds_w[];
    (void)stage_lut(fd, lds_w, use_lds);
    const int nk = rp.n - rp.k, pad = rp.k - ks, ns = ks + nk;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const u32 *P = Pg + (size_t)pad * nk;
    for (i64 cw = (i64)blockIdx.x * nwaves + wave; cw < batch; cw += (i64)gridDim.x * nwaves) {
        const TS *m = msg + cw * ks;
        if (!parity_only)
            for (int i = lane; i < ks; i += 64) out[cw * ns + i] = m[i];
        for (int j = lane; j < nk; j += 64) {
            u32 acc = 0;
            for (int t = 0; t < ks; t++) acc = Lut::add(fd, acc, Lut::mul(fd, (u32)m[t], P[(size_t)t * nk + j]));
            if (parity_only) out[cw * nk + j] = (TS)acc;
            else out[cw * ns + ks + j] = (TS)acc;
        }
    }
}

// non-systematic: c(x) = m(x) g(x), descending coefficient order on both sides
template <typename TS>
__global__ __launch_bounds__(256) void wide_polymul_kernel(FieldDev fd, WideParams rp, const u32 *__restrict__ g,
                                                           const TS *__restrict__ msg, int ks, TS *__restrict__ out, i64 batch)
{
    const int nk = rp.n - rp.k, ns = ks + nk;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    for (i64 cw = (i64)blockIdx.x * nwaves + wave; cw < batch; cw += (i64)gridDim.x * nwaves) {
        const TS *m = msg + cw * ks;
        for (int j = lane; j < ns; j += 64) {
            const int lo = j - nk > 0 ? j - nk : 0, hi = j < ks - 1 ? j : ks - 1;
            u32 acc = 0;
            for (int i = lo; i <= hi; i++) acc = Lut::add(fd, acc, Lut::mul(fd, (u32)m[i], g[j - i]));
            out[cw * ns + j] = (TS)acc;
        }
    }
}

// non-systematic: m(x) = c(x) / g(x) by synthetic division in a per-wave row of global scratch
template <typename TS>
__global__ __launch_bounds__(256) void wide_polydiv_kernel(FieldDev fd, WideParams rp, const u32 *__restrict__ g,
                                                           const TS *__restrict__ cw_g, int ns, TS *__restrict__ out, i64 batch,
                                                           u32 *__restrict__ scratch)
{
    const int nk = rp.n - rp.k, ks = ns - nk;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    u32 *r = scratch + ((size_t)blockIdx.x * nwaves + wave) * ns;
    for (i64 cw = (i64)blockIdx.x * nwaves + wave; cw < batch; cw += (i64)gridDim.x * nwaves) {
        for (int i = lane; i < ns; i += 64) r[i] = (u32)cw_g[cw * ns + i];
        __threadfence_block();
        wave_sync();
        for (int tt = 0; tt < ks; tt++) {
            const u32 q = r[tt];
            if (q != 0)
                for (int j = lane; j < nk; j += 64) r[tt + 1 + j] = Lut::sub(fd, r[tt + 1 + j], Lut::mul(fd, q, g[1 + j]));
            __threadfence_block();
            wave_sync();
        }
        for (int i = lane; i < ks; i += 64) out[cw * ks + i] = (TS)r[i];
        __threadfence_block();
        wave_sync();
    }
}

template <typename TS, bool DETECT_ONLY>
__global__ __launch_bounds__(256) void wide_decode_kernel(FieldDev fd, WideParams rp, const u32 *__restrict__ roots_g,
                                                          const TS *__restrict__ recv_g, const uint8_t *__restrict__ eras_g, int n,
                                                          TS *__restrict__ out_g, i64 *__restrict__ nerr_g,
                                                          uint8_t *__restrict__ detected_g, i64 batch, int use_lds)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds_w[];
    const int dd = rp.nroots;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    u32 *free_w = stage_lut(fd, lds_w, use_lds);
    WideScratch ws;
    ws.carve(free_w + (size_t)wave * WideScratch::words(dd), dd);
    const unsigned long long lt_mask = ((unsigned long long)1 << lane) - 1;
    const bool bin = rp.p == 2;

    for (i64 cw = (i64)blockIdx.x * nwaves + wave; cw < batch; cw += (i64)gridDim.x * nwaves) {
        const TS *row = recv_g + cw * n;
        const uint8_t *er_row = (!DETECT_ONLY && eras_g) ? eras_g + cw * n : nullptr;
        // ---- erasure positions as degrees, ascending (_bch.py:1351-1355) ----
        int u = 0;
        if (er_row) {
            for (int base = 0; base < n; base += 64) {
                const int i = base + lane;
                const bool er = i < n && er_row[n - 1 - i] != 0;
                const unsigned long long m = __ballot(er);
                if (er) {
                    const int slot = u + __popcll(m & lt_mask);
                    if (slot < dd + 2) ws.epos[slot] = (u32)i;
                }
                u += __popcll(m);
            }
        }
        wave_sync();
        int status = 0; // 0 = corrected, 1 = no errors, -1 = failure (row returned unchanged)
        int v = 0, v_total = 0;
        if (u > dd) {
            status = -1;
        } else {
            // ---- 1. syndromes S_j = r(alpha^(c+j)), erased symbols read as zero (_bch.py:1355, 1370) ----
            bool nz = false;
            if (dd <= 32) {
                // few roots: W lanes per root set, G = 64 / W lane groups each run Horner's rule over one contiguous chunk of
                // the word; chunk g's partial sum is then shifted by x^(symbols after the chunk) and the groups are added
                int W = 1;
                while (W < dd) W <<= 1;
                const int G = 64 / W, g = lane / W, j = lane % W;
                const int L = (n + G - 1) / G;
                const int t0 = g * L < n ? g * L : n, t1 = t0 + L < n ? t0 + L : n;
                const u32 x = j < dd ? roots_g[j] : 0;
                u32 acc = 0;
                for (int t = t0; t < t1; t++) {
                    u32 s = (u32)row[t];
                    if (er_row && er_row[t]) s = 0;
                    acc = Lut::add(fd, Lut::mul(fd, x, acc), s);
                }
                if (n - t1 > 0 && acc != 0 && x != 0) acc = Lut::mul(fd, acc, Lut::pow_nz(fd, x, (i64)(n - t1)));
                for (int off = W; off < 64; off <<= 1) acc = Lut::add(fd, acc, (u32)__shfl_xor((int)acc, off));
                if (lane < dd) ws.synd[lane] = acc;
                nz = lane < dd && acc != 0;
            } else {
                for (int j = lane; j < dd; j += 64) {
                    const u32 x = roots_g[j];
                    u32 acc = 0;
                    for (int t = 0; t < n; t++) {
                        u32 s = (u32)row[t];
                        if (er_row && er_row[t]) s = 0;
                        acc = Lut::add(fd, Lut::mul(fd, x, acc), s);
                    }
                    ws.synd[j] = acc;
                    nz |= acc != 0;
                }
            }
            const bool any_nz = __any(nz);
            if constexpr (DETECT_ONLY) {
                if (lane == 0) detected_g[cw] = any_nz ? 1 : 0;
                wave_sync();
                continue;
            }
            wave_sync();
            if (!any_nz && u == 0) {
                status = 1; // _bch.py:1373-1376
            } else {
                // ---- 2. erasure locator Gamma(x) = prod (1 - Y_k x) (_bch.py:1389-1393) ----
                int glen = 1;
                if (lane == 0) ws.gamma[0] = 1;
                wave_sync();
                for (int k = 0; k < u; k++) {
                    const u32 e = ws.epos[k];
                    const u32 Yk = fd.exp_tab[(u32)(((u64)rp.log_alpha * e) % rp.qm1)];
                    const u32 nY = Lut::neg(fd, Yk);
                    u32 nv[4];
#pragma unroll
                    for (int ch = 0; ch < 4; ch++) {
                        const int i = ch * 64 + lane;
                        u32 g = 0;
                        if (i <= glen) {
                            const u32 gi = i < glen ? ws.gamma[i] : 0;
                            const u32 gm = i >= 1 ? ws.gamma[i - 1] : 0;
                            g = Lut::add(fd, gi, Lut::mul(fd, gm, nY));
                        }
                        nv[ch] = g;
                    }
                    wave_sync();
#pragma unroll
                    for (int ch = 0; ch < 4; ch++) {
                        const int i = ch * 64 + lane;
                        if (i <= glen) ws.gamma[i] = nv[ch];
                    }
                    glen++;
                    wave_sync();
                }
                // ---- 3. S'(x) = Gamma(x) S(x) mod x^(d-1) (_bch.py:1408-1409) ----
                for (int l = lane; l < dd; l += 64) {
                    u32 acc = 0;
                    const int imax = l < glen - 1 ? l : glen - 1;
                    for (int i = 0; i <= imax; i++) acc = Lut::add(fd, acc, Lut::mul(fd, ws.gamma[i], ws.synd[l - i]));
                    ws.sprime[l] = acc;
                }
                wave_sync();
                // ---- 4. Berlekamp-Massey on S'[u:] (_bch.py:1421-1428, _lfsr.py:1647-1702) ----
                int llen = 1;
                const int nsq = dd - u;
                if (nsq > 0) {
                    const u32 *S = ws.sprime + u;
                    for (int i = lane; i < nsq; i += 64) { ws.C[i] = i == 0; ws.B[i] = i == 0; }
                    wave_sync();
                    int L = 0, m = 1;
                    u32 b = 1;
                    for (int k = 0; k < nsq; k++) {
                        u32 part = 0;
                        for (int i = lane; i <= L; i += 64) part = Lut::add(fd, part, Lut::mul(fd, S[k - i], ws.C[i]));
                        const u32 dsc = wsum(fd, part);
                        if (dsc == 0) {
                            m++;
                        } else {
                            const u32 coef = Lut::mul(fd, dsc, Lut::inv(fd, b));
                            const bool grow = !(2 * L > k);
                            u32 newc[4], oldc[4];
#pragma unroll
                            for (int ch = 0; ch < 4; ch++) {
                                const int i = ch * 64 + lane;
                                u32 cv = 0, nc = 0;
                                if (i < nsq) {
                                    cv = ws.C[i];
                                    nc = i >= m ? Lut::sub(fd, cv, Lut::mul(fd, coef, ws.B[i - m])) : cv;
                                }
                                newc[ch] = nc; oldc[ch] = cv;
                            }
                            wave_sync();
#pragma unroll
                            for (int ch = 0; ch < 4; ch++) {
                                const int i = ch * 64 + lane;
                                if (i < nsq) {
                                    ws.C[i] = newc[ch];
                                    if (grow) ws.B[i] = oldc[ch];
                                }
                            }
                            if (grow) { L = k + 1 - L; b = dsc; m = 1; }
                            else m++;
                            wave_sync();
                        }
                    }
                    const int clen = L + 1 < nsq ? L + 1 : nsq; // C[:L+1], trailing zeros trimmed (_lfsr.py:1692-1700)
                    int last = 0;
                    for (int base = 0; base < clen; base += 64) {
                        const int i = base + lane;
                        const unsigned long long mk = __ballot(i < clen && ws.C[i] != 0);
                        if (mk) last = base + 63 - __clzll((long long)mk);
                    }
                    llen = last + 1;
                } else {
                    if (lane == 0) ws.C[0] = 1; // Lambda(x) = 1 (_bch.py:1426-1427)
                    wave_sync();
                }
                const u32 *lambda = ws.C;
                v = llen - 1;
                if (2 * v + u > dd) {
                    status = -1; // _bch.py:1431-1433
                } else {
                    // ---- 5. Lambda_total = Gamma * Lambda (_bch.py:1450) ----
                    const int ltlen = glen + llen - 1;
                    for (int l = lane; l < ltlen; l += 64) {
                        u32 acc = 0;
                        const int ilo = l - (llen - 1) > 0 ? l - (llen - 1) : 0;
                        const int ihi = l < glen - 1 ? l : glen - 1;
                        for (int i = ilo; i <= ihi; i++) acc = Lut::add(fd, acc, Lut::mul(fd, ws.gamma[i], lambda[l - i]));
                        ws.ltotal[l] = acc;
                    }
                    wave_sync();
                    // ---- 6. Chien search over i = 0..design_n-1 (_bch.py:1462-1481) ----
                    bool out_of_range_root = false;
                    for (int base = 0; base < rp.n; base += 64) {
                        const int i = base + lane;
                        bool root = false;
                        u32 xinv = 0;
                        if (i < rp.n) {
                            const u32 fwd = (u32)(((u64)rp.log_alpha * (u32)i) % rp.qm1);
                            xinv = fd.exp_tab[fwd == 0 ? 0 : rp.qm1 - fwd];
                            root = horner_asc(fd, ws.ltotal, ltlen, xinv) == 0;
                        }
                        if (__any(root && i >= n)) out_of_range_root = true;
                        const bool rec = root && i < n;
                        const unsigned long long mk = __ballot(rec);
                        if (rec) {
                            const int slot = v_total + __popcll(mk & lt_mask);
                            if (slot < dd + 2) { ws.errpos[slot] = (u32)i; ws.errloc[slot] = xinv; }
                        }
                        v_total += __popcll(mk);
                    }
                    wave_sync();
                    if (out_of_range_root || v_total != v + u) {
                        status = -1; // _bch.py:1469-1485
                    } else {
                        // ---- 7. Omega' = Lambda * S' mod x^(d-1) (_bch.py:1498-1499) ----
                        for (int l = lane; l < dd; l += 64) {
                            u32 acc = 0;
                            const int ihi = l < llen - 1 ? l : llen - 1;
                            for (int i = 0; i <= ihi; i++) acc = Lut::add(fd, acc, Lut::mul(fd, lambda[i], ws.sprime[l - i]));
                            ws.omega[l] = acc;
                        }
                        // ---- 8. formal derivative of Lambda_total (_bch.py:1512-1515) ----
                        const int L_total = ltlen - 1;
                        for (int j = 1 + lane; j <= L_total; j += 64) ws.ltp[j - 1] = Lut::mul(fd, (u32)(j % rp.p), ws.ltotal[j]);
                        wave_sync();
                        // ---- 9./10. Forney magnitudes and corrected symbols (_bch.py:1536-1573) ----
                        for (int kk = lane; kk < v_total; kk += 64) {
                            const u32 x = ws.errloc[kk];
                            const u32 num = horner_asc(fd, ws.omega, dd, x);
                            const u32 den = L_total > 0 ? horner_asc(fd, ws.ltp, L_total, x) : 0;
                            u32 E = den ? Lut::div_nz(fd, num, den) : 0;
                            E = Lut::mul(fd, E, Lut::pow_nz(fd, x, (i64)rp.c - 1));
                            E = Lut::neg(fd, E);
                            const int idx = n - 1 - (int)ws.errpos[kk];
                            u32 r = (u32)row[idx];
                            if (er_row && er_row[idx]) r = 0;
                            if (rp.base_p == 0) {
                                ws.corr[kk] = Lut::sub(fd, r, E);
                            } else if (bin) {
                                // GF(2) subtraction is XOR on the integers (_fields/_gf2.py); out-of-field results are kept
                                // out of the field through the narrowing store, as below
                                const u32 vv = r ^ E;
                                ws.corr[kk] = vv >= (u32)rp.base_p ? 0xffffffffu : vv;
                            } else {
                                // SUBTRACT_BASE: the prime subfield's modular subtract on the integer representations
                                // (_calculate.py:235-251); a miscorrection reproduces the reference's integer result
                                // (E lies in GF(p) whenever the word was correctable).  The reference then fails its field-membership
                                // check on such a row (_bch.py:1300); the value must therefore stay outside GF(p) after the
                                // narrowing store too -- E can exceed 255 here, so a plain truncation could wrap into range.
                                const i64 a = (i64)r, bb = (i64)E;
                                const i64 vv = a >= bb ? a - bb : (i64)rp.base_p + a - bb;
                                ws.corr[kk] = (vv < 0 || vv >= (i64)rp.base_p) ? 0xffffffffu : (u32)vv;
                            }
                        }
                        wave_sync();
                        status = 0;
                    }
                }
            }
        }
        if constexpr (!DETECT_ONLY) {
            // ---- output row: corrected codeword (erased symbols that needed no correction read 0), or the received
            //      row unchanged (_bch.py:1344, 1575-1576) ----
            TS *orow = out_g + cw * n;
            if (status == 0) {
                for (int j = lane; j < n; j += 64) orow[j] = (er_row && er_row[j]) ? (TS)0 : row[j];
                __threadfence_block();
                wave_sync();
                for (int kk = lane; kk < v_total; kk += 64) orow[n - 1 - (int)ws.errpos[kk]] = (TS)ws.corr[kk];
            } else {
                for (int j = lane; j < n; j += 64) orow[j] = row[j];
            }
            if (lane == 0) nerr_g[cw] = status < 0 ? -1 : (status == 1 ? 0 : v);
        }
        wave_sync();
    }
}

WideParams make_wide_params(const gfa_rs *code)
{
    WideParams rp;
    rp.n = (int)code->n; rp.k = (int)code->k; rp.nroots = (int)code->roots.size(); rp.c = (int)code->c;
    rp.p = (int)code->field->calc.p; rp.base_p = (int)code->base_p;
    rp.qm1 = (u32)(code->field->calc.q - 1);
    rp.log_alpha = code->field->h_log[code->alpha];
    return rp;
}

int grid_for_waves(i64 batch, int nwaves, size_t lds_bytes = 0)
{
    static int cached[64] = {0};
    int d = 0, cus = 256;
    if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) {
        if (!cached[d]) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0) cached[d] = prop.multiProcessorCount;
        }
        if (cached[d]) cus = cached[d];
    }
    // persistent workgroups: as many as fit per CU (each one stages the tables once when they live in LDS)
    const i64 per_cu = lds_bytes > 20 * 1024 ? std::max<i64>(1, (i64)(160 * 1024 / lds_bytes)) : 8;
    return (int)std::max<i64>(1, std::min<i64>((batch + nwaves - 1) / nwaves, (i64)cus * per_cu));
}

// LDS words for the staged tables, or 0 when the field is too large for them (then the kernels read global memory)
size_t lut_lds_words(const FieldDev &fd, size_t other_words)
{
    const size_t q = (size_t)fd.qm1 + 1;
    const size_t words = 3 * q + ((fd.p != 2 && fd.m > 1) ? q : 0);
    if (q > 8192 || sizeof(u32) * (words + other_words) > 160 * 1024) return 0;
    return words;
}

template <typename K>
int allow_big_lds(K kern, size_t bytes)
{
    if (bytes > 64 * 1024) GFA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return GFA_OK;
}

template <typename TS>
int encode_t(gfa_rs *code, const FieldDev &fd, gfa_rs::Dev *cd, const void *msg, i64 ks, void *out, i64 batch, int parity_only,
             hipStream_t st)
{
    const WideParams rp = make_wide_params(code);
    const int grid = grid_for_waves(batch, 4);
    if (!code->systematic)
        hipLaunchKernelGGL((wide_polymul_kernel<TS>), dim3(grid), dim3(256), 0, st, fd, rp, cd->gw, (const TS *)msg, (int)ks, (TS *)out, batch);
    else {
        // the encoder is throughput-bound over independent lanes: tables in LDS only while they leave the occupancy alone
        const size_t tw = fd.qm1 < 2048 ? lut_lds_words(fd, 0) : 0;
        int rc = allow_big_lds(wide_encode_kernel<TS>, sizeof(u32) * tw);
        if (rc) return rc;
        const int grid = grid_for_waves(batch, 4, sizeof(u32) * tw);
        hipLaunchKernelGGL((wide_encode_kernel<TS>), dim3(grid), dim3(256), sizeof(u32) * tw, st, fd, rp, cd->Pw, (const TS *)msg, (int)ks,
                           (TS *)out, batch, parity_only, tw ? 1 : 0);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

template <typename TS>
int decode_t(gfa_rs *code, const FieldDev &fd, gfa_rs::Dev *cd, const void *recv, const uint8_t *eras, i64 ns, void *out, i64 *nerr,
             uint8_t *detected, i64 batch, bool detect_only, hipStream_t st)
{
    const WideParams rp = make_wide_params(code);
    const int nwaves = 4;
    const size_t sw = (size_t)nwaves * WideScratch::words(rp.nroots);
    const size_t tw = lut_lds_words(fd, sw);
    const size_t lds = sizeof(u32) * (sw + tw);
    const int grid = grid_for_waves(batch, nwaves, lds);
    int rc;
    if (detect_only) {
        if ((rc = allow_big_lds(wide_decode_kernel<TS, true>, lds))) return rc;
        hipLaunchKernelGGL((wide_decode_kernel<TS, true>), dim3(grid), dim3(nwaves * 64), lds, st, fd, rp, cd->rootsw, (const TS *)recv,
                           nullptr, (int)ns, (TS *)nullptr, (i64 *)nullptr, detected, batch, tw ? 1 : 0);
    } else {
        if ((rc = allow_big_lds(wide_decode_kernel<TS, false>, lds))) return rc;
        hipLaunchKernelGGL((wide_decode_kernel<TS, false>), dim3(grid), dim3(nwaves * 64), lds, st, fd, rp, cd->rootsw, (const TS *)recv,
                           eras, (int)ns, (TS *)out, nerr, (uint8_t *)nullptr, batch, tw ? 1 : 0);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

template <typename TS>
int polydiv_t(gfa_rs *code, const FieldDev &fd, gfa_rs::Dev *cd, const void *cw, i64 ns, void *out, i64 batch, hipStream_t st)
{
    const WideParams rp = make_wide_params(code);
    const int nwaves = 4;
    const int grid = grid_for_waves(batch, nwaves);
    u32 *scratch = nullptr;
    GFA_HIP(gfa::scratch_alloc((void **)&scratch, sizeof(u32) * (size_t)grid * nwaves * (size_t)ns, st));
    hipLaunchKernelGGL((wide_polydiv_kernel<TS>), dim3(grid), dim3(nwaves * 64), 0, st, fd, rp, cd->gw, (const TS *)cw, (int)ns, (TS *)out,
                       batch, scratch);
    const hipError_t e = hipGetLastError();
    (void)gfa::scratch_free(scratch, st);
    GFA_HIP(e);
    return GFA_OK;
}

size_t dtype_size(int dtype) { return dtype == GFA_U8 ? 1 : dtype == GFA_U16 ? 2 : dtype == GFA_U32 ? 4 : 8; }

} // namespace

namespace gfa {

// Codes this file serves: the (syndrome) field has EXP/LOG tables but no byte product tables.
bool rs_wide_code(const gfa_rs *code) { return !code->field->has_tab8 && code->field->has_lut; }

int rs_wide_check(const gfa_rs *code, int dtype, const char *what)
{
    const u64 symbols = code->base_p ? (u64)code->base_p : code->field->calc.q;
    if ((dtype != GFA_U8 && dtype != GFA_U16 && dtype != GFA_U32) || !dtype_holds(dtype, symbols)) {
        set_error(std::string(what) + ": symbols must be stored as uint8 / uint16 / uint32 wide enough for the symbol field");
        return GFA_ERR_INVALID;
    }
    if ((i64)code->roots.size() > WIDE_MAX_D) {
        set_error(std::string(what) + ": the device path covers design distances up to 255");
        return GFA_ERR_UNSUPPORTED;
    }
    if ((double)code->k * (double)(code->n - code->k) > (double)((i64)1 << 28)) {
        set_error(std::string(what) + ": parity matrix too large for the device path");
        return GFA_ERR_UNSUPPORTED;
    }
    return GFA_OK;
}

#define GFA_WIDE_DISPATCH(FUNC, ...)                          \
    switch (dtype) {                                          \
    case GFA_U8: return FUNC<uint8_t>(__VA_ARGS__);           \
    case GFA_U16: return FUNC<uint16_t>(__VA_ARGS__);         \
    default: return FUNC<uint32_t>(__VA_ARGS__);              \
    }

int rs_wide_encode(gfa_rs *code, const void *msg, i64 ks, void *out, i64 batch, int parity_only, int dtype, hipStream_t st)
{
    int rc;
    FieldDeviceState *ds;
    gfa_rs::Dev *cd;
    if ((rc = code->field->ensure_device(nullptr, &ds))) return rc;
    if ((rc = code->ensure_device(nullptr, &cd))) return rc;
    const FieldDev fd = code->field->lut_desc(*ds);
    GFA_WIDE_DISPATCH(encode_t, code, fd, cd, msg, ks, out, batch, parity_only, st);
}

int rs_wide_decode(gfa_rs *code, const void *recv, const uint8_t *eras, i64 ns, void *out, i64 *nerr, uint8_t *detected, i64 batch,
                   bool detect_only, int dtype, hipStream_t st)
{
    int rc;
    FieldDeviceState *ds;
    gfa_rs::Dev *cd;
    if ((rc = code->field->ensure_device(nullptr, &ds))) return rc;
    if ((rc = code->ensure_device(nullptr, &cd))) return rc;
    const FieldDev fd = code->field->lut_desc(*ds);
    if (!detect_only && out == recv) {
        // the kernel reads the received row while it writes the output row: decode in place through a copy of the input
        void *tmp = nullptr;
        const size_t bytes = dtype_size(dtype) * (size_t)batch * (size_t)ns;
        GFA_HIP(gfa::scratch_alloc(&tmp, bytes, st));
        GFA_HIP(hipMemcpyAsync(tmp, recv, bytes, hipMemcpyDeviceToDevice, st));
        rc = rs_wide_decode(code, tmp, eras, ns, out, nerr, detected, batch, false, dtype, st);
        (void)gfa::scratch_free(tmp, st);
        return rc;
    }
    GFA_WIDE_DISPATCH(decode_t, code, fd, cd, recv, eras, ns, out, nerr, detected, batch, detect_only, st);
}

int rs_wide_polydiv(gfa_rs *code, const void *cw, i64 ns, void *out, i64 batch, int dtype, hipStream_t st)
{
    int rc;
    FieldDeviceState *ds;
    gfa_rs::Dev *cd;
    if ((rc = code->field->ensure_device(nullptr, &ds))) return rc;
    if ((rc = code->ensure_device(nullptr, &cd))) return rc;
    const FieldDev fd = code->field->lut_desc(*ds);
    GFA_WIDE_DISPATCH(polydiv_t, code, fd, cd, cw, ns, out, batch, st);
}

} // namespace gfa
