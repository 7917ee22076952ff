// gfa_rs.hip -- Reed-Solomon encode / detect / decode on gfx950 for codes over fields of order <= 256 (uint8); codes over
// larger fields are routed to gfa_rs_wide.hip from the entry points at the bottom of this file.
//
// Replaces, for ReedSolomon codes (reference paths relative to src/galois):
//   * encode : _LinearCode._encode_message -> matmul_jit            (_codes/_linear.py:270-284, _domains/_linalg.py:286-308)
//   * detect : _LinearCode._detect_errors                           (_codes/_linear.py:286-298)
//   * decode : bch_decode_jit.implementation, one loop iteration    (_codes/_bch.py:1337-1578) with
//              evaluate_elementwise_jit (_polys/_dense.py:432-440), convolve_jit (_domains/_function.py:141-167),
//              berlekamp_massey_jit (_lfsr.py:1647-1702)
//
// Execution model: ONE CODEWORD PER WAVEFRONT.  The 64 lanes of a wave cooperate on a codeword: lanes run over
// syndrome indices / polynomial coefficients / Chien positions / error indices, the serial recurrences
// (Berlekamp-Massey iterations) are wave-uniform, and every per-codeword polynomial lives in a small LDS scratch
// area owned by the wave, so there is no divergence between codewords with different error counts.
// Field arithmetic is one LDS gather per operation from full 64 KiB tables (index (a<<8)|b) shared by the
// workgroup; characteristic-2 fields use XOR for addition.
#include <algorithm>

#include "gfa_internal.h"
#include "gfa_rs_host.h"

using namespace gfa;


namespace {

__device__ __forceinline__ void wave_sync()
{
    // LDS ops of one wave execute in order; this stops the compiler from moving accesses across phase boundaries
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct RsTables {
    const uint8_t *mul8, *add8, *neg8, *inv8, *exp8, *log8; // global
};

struct RsParams {
    int n;        // design n
    int k;
    int nroots;   // d - 1
    int c;
    int p;        // characteristic
    int qm1;      // q - 1
    int log_alpha; // LOG[alpha]
    int base_p;    // 0 = Reed-Solomon; p = BCH over GF(p): corrections use SUBTRACT_BASE (_bch.py:1310, 1573)
    int per_block; // codewords per workgroup (host-computed: no 64-bit division in the kernel)
};

template <bool BIN>
struct Arith8 {
    const uint8_t *mul_t, *add_t, *neg_t, *inv_t, *exp_t, *log_t; // LDS
    int qm1;
    __device__ __forceinline__ u32 mul(u32 a, u32 b) const { return mul_t[(a << 8) | b]; }
    __device__ __forceinline__ u32 add(u32 a, u32 b) const
    {
        if constexpr (BIN) return a ^ b;
        else return add_t[(a << 8) | b];
    }
    __device__ __forceinline__ u32 neg(u32 a) const
    {
        if constexpr (BIN) return a;
        else return neg_t[a];
    }
    __device__ __forceinline__ u32 sub(u32 a, u32 b) const
    {
        if constexpr (BIN) return a ^ b;
        else return add_t[(a << 8) | neg_t[b]];
    }
    __device__ __forceinline__ u32 inv(u32 a) const { return inv_t[a]; }
    // x != 0, any integer e: x^e = EXP[(LOG[x] * e) mod (q-1)]  (power_ufunc.lookup, _lookup.py:247-270)
    __device__ __forceinline__ u32 pow_nz(u32 x, int e) const
    {
        int em = e % qm1;
        if (em < 0) em += qm1;
        return exp_t[((int)log_t[x] * em) % qm1];
    }
    __device__ __forceinline__ u32 wave_sum(u32 x) const
    { // field sum over the 64 lanes, result in every lane (wave-uniform)
        if constexpr (BIN) {
            // XOR: four DPP steps fold each 16-lane row, then the four row results are combined through SGPRs
            int v = (int)x;
            v ^= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
            v ^= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
            v ^= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true); // row_half_mirror
            v ^= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true); // row_mirror
            return (u32)(__builtin_amdgcn_readlane(v, 0) ^ __builtin_amdgcn_readlane(v, 16) ^
                         __builtin_amdgcn_readlane(v, 32) ^ __builtin_amdgcn_readlane(v, 48));
        } else {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) x = add(x, (u32)__shfl_xor((int)x, off));
            return x;
        }
    }
};

// Cooperative staging of the arithmetic tables into LDS.  Returns the first free LDS byte.
template <bool BIN>
__device__ __forceinline__ uint8_t *stage_tables(uint8_t *lds, const RsTables &t, Arith8<BIN> &ar, int qm1, int nthreads)
{
    uint8_t *mul_l = lds; lds += 65536;
    uint8_t *add_l = nullptr;
    if constexpr (!BIN) { add_l = lds; lds += 65536; }
    uint8_t *small = lds; lds += 512 + 256 * 3; // exp(512) log neg inv
    {
        const uint4 *s = reinterpret_cast<const uint4 *>(t.mul8);
        uint4 *d = reinterpret_cast<uint4 *>(mul_l);
        for (int i = threadIdx.x; i < 4096; i += nthreads) d[i] = s[i];
        if constexpr (!BIN) {
            const uint4 *s2 = reinterpret_cast<const uint4 *>(t.add8);
            uint4 *d2 = reinterpret_cast<uint4 *>(add_l);
            for (int i = threadIdx.x; i < 4096; i += nthreads) d2[i] = s2[i];
        }
        for (int i = threadIdx.x; i < 512; i += nthreads) small[i] = t.exp8[i];
        for (int i = threadIdx.x; i < 256; i += nthreads) {
            small[512 + i] = t.log8[i];
            small[768 + i] = t.neg8[i];
            small[1024 + i] = t.inv8[i];
        }
    }
    ar.mul_t = mul_l; ar.add_t = add_l; ar.exp_t = small; ar.log_t = small + 512; ar.neg_t = small + 768;
    ar.inv_t = small + 1024; ar.qm1 = qm1;
    return lds;
}

// ------------------------------------------------------------------------------------------------
// encode: parity = message @ P[pad:, :]   (systematic)
// ------------------------------------------------------------------------------------------------
// A wave handles GROUPS codewords at once: lanes are split into GROUPS groups of LPG = 64/GROUPS lanes, lane j of a
// group accumulates parity symbol j (and j+LPG, ... when n-k > LPG).
template <bool BIN>
__global__ __launch_bounds__(1024) void rs_encode_kernel(RsTables t, RsParams rp, const uint8_t *__restrict__ Pg,
                                                         const uint8_t *__restrict__ msg, int ks,
                                                         uint8_t *__restrict__ out, i64 batch, int parity_only,
                                                         int groups)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    Arith8<BIN> ar;
    uint8_t *free_l = stage_tables<BIN>(lds_raw, t, ar, rp.qm1, blockDim.x);
    const int nk = rp.n - rp.k;
    const int pad = rp.k - ks;
    uint8_t *P_l = free_l; // ks x nk (rows pad..k-1 of P)
    for (int i = threadIdx.x; i < ks * nk; i += blockDim.x) P_l[i] = Pg[pad * nk + i];
    free_l += ((ks * nk + 15) / 16) * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int lpg = 64 / groups;
    const int grp = lane / lpg, gl = lane % lpg;
    const int mpitch = ((ks + 15) / 16) * 16;
    uint8_t *m_l = free_l + (size_t)wave * groups * mpitch; // message rows of this wave's codewords
    __syncthreads();
    const int ns = ks + nk;
    const i64 cw_per_block = (i64)nwaves * groups;
    for (i64 base = (i64)blockIdx.x * cw_per_block; base < batch; base += (i64)gridDim.x * cw_per_block) {
        const i64 cw0 = base + (i64)wave * groups;
        // stage the message rows (and copy them to the output codewords)
        for (int g = 0; g < groups; g++) {
            const i64 cw = cw0 + g;
            if (cw < batch)
                for (int i = lane; i < ks; i += 64) {
                    uint8_t v = msg[cw * ks + i];
                    m_l[g * mpitch + i] = v;
                    if (!parity_only) out[cw * ns + i] = v;
                }
        }
        wave_sync();
        const i64 cw = cw0 + grp;
        if (cw < batch) {
            const uint8_t *mrow = m_l + grp * mpitch;
            for (int j = gl; j < nk; j += lpg) {
                u32 acc = 0;
                for (int tt = 0; tt < ks; tt++) acc = ar.add(acc, ar.mul(mrow[tt], P_l[tt * nk + j]));
                if (parity_only) out[cw * nk + j] = (uint8_t)acc;
                else out[cw * ns + ks + j] = (uint8_t)acc;
            }
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// decode / detect
// ------------------------------------------------------------------------------------------------
struct WaveScratch {
    uint8_t *recv, *synd, *gamma, *sprime, *C, *B, *ltotal, *omega, *ltp, *epos, *errpos, *errloc;
    static __host__ __device__ int bytes(int n, int dd) { return ((n + 14 * (dd + 2) + 15) / 16) * 16; }
    __device__ void carve(uint8_t *p, int n, int dd)
    {
        const int s = dd + 2;
        recv = p; p += n;
        synd = p; p += s; gamma = p; p += s; sprime = p; p += s; C = p; p += s; B = p; p += s;
        ltotal = p; p += 2 * s; omega = p; p += s; ltp = p; p += 2 * s; epos = p; p += s; errpos = p; p += s;
        errloc = p; p += s;
    }
};

// polynomial evaluation by Horner, coefficients ascending in `co[0..len)`, i.e. acc = co[len-1]; acc = acc*x + co[i]
// (evaluate_elementwise_jit with the coefficient order reversed, _polys/_dense.py:432-440)
template <bool BIN>
__device__ __forceinline__ u32 horner_asc(const Arith8<BIN> &ar, const uint8_t *co, int len, u32 x)
{
    u32 acc = co[len - 1];
    for (int i = len - 2; i >= 0; i--) acc = ar.add(ar.mul(acc, x), co[i]);
    return acc;
}

template <bool BIN, bool DETECT_ONLY>
__global__ __launch_bounds__(1024) void rs_decode_kernel(RsTables t, RsParams rp, const uint8_t *__restrict__ roots_g,
                                                         const uint8_t *__restrict__ recv_g,
                                                         const uint8_t *__restrict__ eras_g, int n,
                                                         uint8_t *__restrict__ out_g, i64 *__restrict__ nerr_g,
                                                         uint8_t *__restrict__ detected_g, i64 batch)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    Arith8<BIN> ar;
    uint8_t *free_l = stage_tables<BIN>(lds_raw, t, ar, rp.qm1, blockDim.x);
    const int dd = rp.nroots;
    uint8_t *roots_l = free_l;
    for (int i = threadIdx.x; i < dd; i += blockDim.x) roots_l[i] = roots_g[i];
    free_l += ((dd + 15) / 16) * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    WaveScratch ws;
    ws.carve(free_l + (size_t)wave * WaveScratch::bytes(n, dd), n, dd);
    __syncthreads();
    const unsigned long long lt_mask = ((unsigned long long)1 << lane) - 1;

    for (i64 cw = (i64)blockIdx.x * nwaves + wave; cw < batch; cw += (i64)gridDim.x * nwaves) {
        const uint8_t *row = recv_g + cw * n;
        // ---- received word in ascending degree order, erased symbols zeroed (_bch.py:1351-1355) ----
        int u = 0;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            bool er = false;
            if (i < n) {
                u32 r = row[n - 1 - i];
                if (!DETECT_ONLY && eras_g) er = eras_g[cw * n + (n - 1 - i)] != 0;
                ws.recv[i] = er ? 0 : (uint8_t)r;
            }
            const unsigned long long m = __ballot(er);
            if (er) ws.epos[u + __popcll(m & lt_mask)] = (uint8_t)i;
            u += __popcll(m);
        }
        wave_sync();
        int status = 0; // 0 = corrected (write recv), 1 = unchanged row / no errors, -1 = failure (unchanged row)
        int v = 0;
        if (u > dd) {
            status = -1;
        } else {
            // ---- 1. syndromes S_j = r(alpha^(c+j)) (_bch.py:1370) ----
            bool nz = false;
            for (int j = lane; j < dd; j += 64) {
                const u32 x = roots_l[j];
                u32 acc = ws.recv[n - 1];
                for (int i = n - 2; i >= 0; i--) acc = ar.add(ar.mul(x, acc), ws.recv[i]);
                ws.synd[j] = (uint8_t)acc;
                nz |= acc != 0;
            }
            const bool any_nz = __any(nz);
            if constexpr (DETECT_ONLY) {
                if (lane == 0) detected_g[cw] = any_nz ? 1 : 0;
                wave_sync();
                continue;
            }
            wave_sync();
            if (!any_nz && u == 0) {
                status = 1; // no errors (_bch.py:1373-1376)
            } else {
                // ---- 2. erasure locator Gamma(x) = prod (1 - Y_k x), Y_k = alpha^e_k (_bch.py:1389-1393) ----
                int glen = 1;
                if (lane == 0) ws.gamma[0] = 1;
                wave_sync();
                for (int k = 0; k < u; k++) {
                    const int e = ws.epos[k];
                    const u32 Yk = ar.exp_t[(rp.log_alpha * e) % rp.qm1];
                    const u32 nY = ar.neg(Yk);
                    u32 nv[4];
#pragma unroll
                    for (int ch = 0; ch < 4; ch++) {
                        const int i = ch * 64 + lane;
                        u32 g = 0;
                        if (i <= glen) {
                            const u32 gi = i < glen ? ws.gamma[i] : 0;
                            const u32 gm = i >= 1 ? ws.gamma[i - 1] : 0;
                            g = ar.add(gi, ar.mul(gm, nY));
                        }
                        nv[ch] = g;
                    }
                    wave_sync();
#pragma unroll
                    for (int ch = 0; ch < 4; ch++) {
                        const int i = ch * 64 + lane;
                        if (i <= glen) ws.gamma[i] = (uint8_t)nv[ch];
                    }
                    glen++;
                    wave_sync();
                }
                // ---- 3. modified syndromes S'(x) = Gamma(x) S(x) mod x^(d-1) (_bch.py:1408-1409) ----
                for (int l = lane; l < dd; l += 64) {
                    u32 acc = 0;
                    const int imax = l < glen - 1 ? l : glen - 1;
                    for (int i = 0; i <= imax; i++) acc = ar.add(acc, ar.mul(ws.gamma[i], ws.synd[l - i]));
                    ws.sprime[l] = (uint8_t)acc;
                }
                wave_sync();
                // ---- 4. Berlekamp-Massey on S'[u:] (_bch.py:1421-1428, _lfsr.py:1647-1702) ----
                int llen = 1;
                const int nsq = dd - u;
                if (nsq > 0) {
                    const uint8_t *S = ws.sprime + u;
                    for (int i = lane; i < nsq; i += 64) { ws.C[i] = i == 0; ws.B[i] = i == 0; }
                    wave_sync();
                    int L = 0, m = 1;
                    u32 b = 1;
                    for (int k = 0; k < nsq; k++) {
                        u32 part = 0;
                        for (int i = lane; i <= L; i += 64) part = ar.add(part, ar.mul(S[k - i], ws.C[i]));
                        const u32 dsc = ar.wave_sum(part);
                        if (dsc == 0) {
                            m++;
                        } else {
                            const u32 coef = ar.mul(dsc, ar.inv(b));
                            const bool grow = !(2 * L > k);
                            u32 newc[4], oldc[4];
#pragma unroll
                            for (int ch = 0; ch < 4; ch++) {
                                const int i = ch * 64 + lane;
                                u32 cv = 0, nc = 0;
                                if (i < nsq) {
                                    cv = ws.C[i];
                                    nc = i >= m ? ar.sub(cv, ar.mul(coef, ws.B[i - m])) : cv;
                                }
                                newc[ch] = nc; oldc[ch] = cv;
                            }
                            wave_sync();
#pragma unroll
                            for (int ch = 0; ch < 4; ch++) {
                                const int i = ch * 64 + lane;
                                if (i < nsq) {
                                    ws.C[i] = (uint8_t)newc[ch];
                                    if (grow) ws.B[i] = (uint8_t)oldc[ch];
                                }
                            }
                            if (grow) { L = k + 1 - L; b = dsc; m = 1; }
                            else m++;
                            wave_sync();
                        }
                    }
                    // C = C[:L+1], trailing zeros trimmed (_lfsr.py:1692-1700)
                    const int clen = L + 1 < nsq ? L + 1 : nsq;
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
                const uint8_t *lambda = ws.C;
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
                        for (int i = ilo; i <= ihi; i++) acc = ar.add(acc, ar.mul(ws.gamma[i], lambda[l - i]));
                        ws.ltotal[l] = (uint8_t)acc;
                    }
                    wave_sync();
                    // ---- 6. Chien search over i = 0..design_n-1 (_bch.py:1462-1481) ----
                    int v_total = 0;
                    bool out_of_range_root = false;
                    for (int base = 0; base < rp.n; base += 64) {
                        const int i = base + lane;
                        bool root = false;
                        u32 xinv = 0;
                        if (i < rp.n) {
                            int e = (-(rp.log_alpha * i)) % rp.qm1;
                            if (e < 0) e += rp.qm1;
                            xinv = ar.exp_t[e];
                            root = horner_asc<BIN>(ar, ws.ltotal, ltlen, xinv) == 0;
                        }
                        if (__any(root && i >= n)) out_of_range_root = true;
                        const bool rec = root && i < n;
                        const unsigned long long mk = __ballot(rec);
                        if (rec) {
                            const int slot = v_total + __popcll(mk & lt_mask);
                            if (slot < dd + 2) { ws.errpos[slot] = (uint8_t)i; ws.errloc[slot] = (uint8_t)xinv; }
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
                            for (int i = 0; i <= ihi; i++) acc = ar.add(acc, ar.mul(lambda[i], ws.sprime[l - i]));
                            ws.omega[l] = (uint8_t)acc;
                        }
                        // ---- 8. formal derivative of Lambda_total (_bch.py:1512-1515) ----
                        const int L_total = ltlen - 1;
                        for (int j = 1 + lane; j <= L_total; j += 64)
                            ws.ltp[j - 1] = (uint8_t)ar.mul((u32)(j % rp.p), ws.ltotal[j]);
                        wave_sync();
                        // ---- 9./10. Forney magnitudes and correction (_bch.py:1536-1573) ----
                        for (int kk = lane; kk < v_total; kk += 64) {
                            const u32 x = ws.errloc[kk];
                            const u32 num = horner_asc<BIN>(ar, ws.omega, dd, x);
                            const u32 den = horner_asc<BIN>(ar, ws.ltp, L_total, x);
                            u32 E = ar.mul(num, ar.inv(den));
                            E = ar.mul(E, ar.pow_nz(x, rp.c - 1));
                            E = ar.neg(E);
                            const int pos = ws.errpos[kk];
                            if (rp.base_p == 0 || BIN) {
                                ws.recv[pos] = (uint8_t)ar.sub(ws.recv[pos], E);
                            } else {
                                // SUBTRACT_BASE: the prime subfield's modular subtract on the integer representations
                                // (_calculate.py:235-251); E lies in GF(p) whenever the word was correctable, and for a
                                // miscorrection the reference's integer result is reproduced, wrapped to the storage dtype
                                const int a = ws.recv[pos], b = (int)E;
                                ws.recv[pos] = (uint8_t)(a >= b ? a - b : rp.base_p + a - b);
                            }
                        }
                        wave_sync();
                        status = 0;
                    }
                }
            }
        }
        if constexpr (!DETECT_ONLY) {
            // ---- output row: corrected codeword, or the received row unchanged (_bch.py:1344, 1575-1576) ----
            uint8_t *orow = out_g + cw * n;
            if (status == 0) {
                for (int j = lane; j < n; j += 64) orow[j] = ws.recv[n - 1 - j];
            } else {
                for (int j = lane; j < n; j += 64) orow[j] = row[j];
            }
            if (lane == 0) nerr_g[cw] = status < 0 ? -1 : (status == 1 ? 0 : v);
        }
        wave_sync();
    }
}


// ------------------------------------------------------------------------------------------------
// LFSR remainder kernel (characteristic 2, (n-k) % 4 == 0, n-k <= 64): ONE CODEWORD PER LANE
// ------------------------------------------------------------------------------------------------
// Division by the generator polynomial g(x) as a byte-wide LFSR whose (n-k)-byte state lives in NKW VGPRs.  One step
// per symbol: shift the state up by one byte and XOR in the table row  f * (g_{nk-1}, ..., g_0)  for the feedback
// byte f -- a single wide LDS read instead of the (n-k) byte gathers a matrix-vector product spends per symbol
// (16x fewer LDS operations for RS(255,223)).  The same loop serves
//   ENCODE:  state <- m(x) * x^(n-k) mod g(x)   = the systematic parity symbols   (_LinearCode._encode_message)
//   !ENCODE: state <- r(x) mod g(x)             = 0 iff all syndromes vanish      (_detect_errors; decoder pre-pass)
// Identical values to the reference's matrix products: parity = message @ P is by construction -(m x^(n-k) mod g).
// Measured and not kept (r03): a look-ahead for the feedback byte (next top byte = role-1 plane's top byte ^ one byte of this
// symbol's row, from a 256-byte table, so that the 16-byte row reads leave the symbol-to-symbol chain): encode 0.0389 -> 0.0421 ms.
// The kernel is bound by the LDS queue (two ds_read_b128 per lane and symbol), not by the latency of one read; a third read costs.
// REP4 (NKW = 4 or 8, one 512-thread workgroup per CU): every 16-byte table chunk is stored FOUR times, 32 bytes apart, rows 2m
// and 2m + 1 interleaved in one 128-byte block, and lane l reads copy (l >> kshift) & 3 -- the lanes the LDS serves in one cycle
// then touch disjoint bank groups wherever their feedback bytes allow it (the kernel is bound by the LDS queue: two ds_read_b128
// per symbol with random 16-byte offsets; PMC before: 59 % of the LDS cycles were bank conflicts).
template <int NKW, bool ENCODE, bool REP4 = false>
__global__ __launch_bounds__(REP4 ? 512 : 256) void rs_lfsr_kernel(const u32 *__restrict__ rowtab, const uint8_t *__restrict__ in,
                                                      const uint8_t *__restrict__ eras, int len, uint8_t *__restrict__ out,
                                                      int parity_only, uint8_t *__restrict__ rem_out,
                                                      uint8_t *__restrict__ flag_out, i64 batch, int stage_bytes, u32 div_magic, int kshift)
{
    constexpr int NK = NKW * 4;
    constexpr int TABLE_BYTES = REP4 ? (NKW / 4) * 16384 : 256 * NK;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    u32 *T = reinterpret_cast<u32 *>(lds_raw); // 256 rows x NKW words
    if constexpr (REP4) {
        // source: chunk c of row f at rowtab[c * 1024 + f * 4 ..]; copy k of it at byte c * 16384 + (f >> 1) * 128 + k * 32 + (f & 1) * 16
        for (int i = threadIdx.x; i < (NKW / 4) * 1024 * 4; i += blockDim.x) {
            const int wd = i & 3, k = (i >> 2) & 3, f = (i >> 4) & 255, c = i >> 12;
            T[c * 4096 + (f >> 1) * 32 + k * 8 + (f & 1) * 4 + wd] = rowtab[c * 1024 + f * 4 + wd];
        }
    } else {
        for (int i = threadIdx.x; i < 256 * NKW; i += blockDim.x) T[i] = rowtab[i];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const u32 kofs = (u32)((lane >> kshift) & 3) << 5;
    // Lane l divides row rowi = 4*(l % 16) + l/16 of the wave's 64 staged rows.  With the natural pitch of n = 255 bytes,
    // rows r and r+1 start 63.75 dwords apart, so four CONSECUTIVE rows share an LDS bank; the interleave leaves at most
    // two lanes of a 32-lane group on one bank for the per-symbol byte read.
    const int rowi = ((lane & 15) << 2) | (lane >> 4);
    uint8_t *stage = lds_raw + TABLE_BYTES + (size_t)wave * stage_bytes;
    __syncthreads();
    const int ns_out = len + NK;
    for (i64 cw0 = ((i64)blockIdx.x * nwaves + wave) * 64; cw0 < batch; cw0 += (i64)gridDim.x * nwaves * 64) {
        const int count = (int)(batch - cw0 < 64 ? batch - cw0 : 64);
        const uint8_t *src = in + cw0 * len;
        const int nbytes = count * len;
        // ---- stage this wave's rows (contiguous in memory) into LDS ----
        if (ENCODE && !parity_only) {
            // Build the output codewords in LDS: message symbols scattered from a flat, coalesced 16-byte read into rows
            // of pitch ks + (n-k); the parity bytes are appended after the LFSR and the rows leave as one flat copy.
            const bool al = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
            const int nv = al ? nbytes >> 4 : 0;
            const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
            for (int i = lane; i < nv; i += 64) {
                const uint4 v = s4[i];
                const u32 b0 = (u32)i << 4;
                u32 row = __umulhi(b0, div_magic); // b0 / len, exact for b0 < 2^16
                u32 col = b0 - row * (u32)len;
                const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    stage[row * ns_out + col] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
                    col++;
                    if (col == (u32)len) { col = 0; row++; }
                }
            }
            for (int b = (nv << 4) + lane; b < nbytes; b += 64) {
                const u32 row = __umulhi((u32)b, div_magic);
                stage[row * ns_out + ((u32)b - row * (u32)len)] = src[b];
            }
        } else {
            // decoder pre-pass: `out` (if given and not the input itself) receives a verbatim copy of the received rows, so
            // that the wave kernel only has to patch the corrected symbols (failed words stay "received row unchanged")
            uint8_t *cp = (!ENCODE && out && out != in) ? out + cw0 * len : nullptr;
            const bool al = ((reinterpret_cast<uintptr_t>(src) | (eras ? reinterpret_cast<uintptr_t>(eras + cw0 * len) : 0) |
                              reinterpret_cast<uintptr_t>(cp)) & 15) == 0;
            int done = 0;
            if (al) {
                const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
                const uint4 *e4 = eras ? reinterpret_cast<const uint4 *>(eras + cw0 * len) : nullptr;
                uint4 *d4 = reinterpret_cast<uint4 *>(stage);
                const int nv = nbytes >> 4;
                for (int i = lane; i < nv; i += 64) {
                    uint4 v = s4[i];
                    if (cp) reinterpret_cast<uint4 *>(cp)[i] = v;
                    if (e4) { // erased symbols are treated as zeros (_bch.py:1355): byte mask from "non-zero"
                        const uint4 e = e4[i];
                        auto mask = [](u32 w) -> u32 { return ((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) >> 7 & 0x01010101u) * 0xffu; };
                        v.x &= ~mask(e.x); v.y &= ~mask(e.y); v.z &= ~mask(e.z); v.w &= ~mask(e.w);
                    }
                    d4[i] = v;
                }
                done = nv << 4;
            }
            for (int i = done + lane; i < nbytes; i += 64) {
                uint8_t v = src[i];
                if (cp) cp[i] = v;
                if (eras && eras[cw0 * len + i]) v = 0;
                stage[i] = v;
            }
        }
        wave_sync();
        // ---- the LFSR: lane t divides row t ----
        u32 st[NKW];
#pragma unroll
        for (int d = 0; d < NKW; d++) st[d] = 0;
        if (rowi < count) {
            const uint8_t *row = stage + rowi * ((ENCODE && !parity_only) ? ns_out : len);
            if constexpr (NKW % 4 == 0) {
                // PLANAR state: plane p holds the state bytes p, p + 4, p + 8, ... (W = NKW / 4 words, first byte on top).  Moving
                // the whole state up one byte turns plane p + 1 into plane p -- a renaming, free in the unrolled loop -- and only
                // the old plane 0, shifted one byte with the new symbol at the bottom, becomes the new plane 3: W shift
                // instructions per symbol instead of NKW.  The table rows are stored in the same planar order (host side).
                constexpr int W = NKW / 4;
                u32 P[4][W];
#pragma unroll
                for (int p = 0; p < 4; p++)
#pragma unroll
                    for (int h = 0; h < W; h++) P[p][h] = 0;
                // one symbol; K = symbols taken so far mod 4: the plane in role r is P[(r + K) & 3]
                auto step = [&](auto kc, u32 sym) {
                    constexpr int K = decltype(kc)::value;
                    const u32 top = P[K & 3][0] >> 24;
                    const u32 f = ENCODE ? (sym ^ top) : top;
#pragma unroll
                    for (int h = 0; h + 1 < W; h++) P[K & 3][h] = __builtin_amdgcn_alignbit(P[K & 3][h], P[K & 3][h + 1], 24);
                    P[K & 3][W - 1] = (P[K & 3][W - 1] << 8) | (ENCODE ? 0u : sym);
#pragma unroll
                    for (int c4 = 0; c4 < NKW / 4; c4++) {
                        const uint4 rv = REP4 ? *reinterpret_cast<const uint4 *>(lds_raw + c4 * 16384 + (((f >> 1) << 7) | ((f & 1) << 4) | kofs))
                                              : *reinterpret_cast<const uint4 *>(T + c4 * 1024 + f * 4);
                        const u32 w[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int q = 4 * c4 + j;                  // planar word: role q / W, word q % W
                            P[(q / W + K + 1) & 3][q % W] ^= w[j];
                        }
                    }
                };
                int i = 0;
                for (; i + 4 <= len; i += 4) {
                    step(std::integral_constant<int, 0>{}, row[i]);
                    step(std::integral_constant<int, 1>{}, row[i + 1]);
                    step(std::integral_constant<int, 2>{}, row[i + 2]);
                    step(std::integral_constant<int, 3>{}, row[i + 3]);
                }
                const int tail = len - i; // 0 .. 3 symbols left; afterwards role r is P[(r + tail) & 3]
                if (tail > 0) step(std::integral_constant<int, 0>{}, row[i]);
                if (tail > 1) step(std::integral_constant<int, 1>{}, row[i + 1]);
                if (tail > 2) step(std::integral_constant<int, 2>{}, row[i + 2]);
                // back to consecutive bytes: a 4 x 4 byte transpose per word index (8 v_perm)
                auto unplane = [&](auto kc) {
                    constexpr int K = decltype(kc)::value;
#pragma unroll
                    for (int h = 0; h < W; h++) {
                        const u32 A = P[K & 3][h], B = P[(K + 1) & 3][h], C = P[(K + 2) & 3][h], D = P[(K + 3) & 3][h];
                        const u32 t0 = __builtin_amdgcn_perm(A, B, 0x07030602u), t1 = __builtin_amdgcn_perm(A, B, 0x05010400u);
                        const u32 u0 = __builtin_amdgcn_perm(C, D, 0x07030602u), u1 = __builtin_amdgcn_perm(C, D, 0x05010400u);
                        st[4 * h + 0] = __builtin_amdgcn_perm(t0, u0, 0x07060302u);
                        st[4 * h + 1] = __builtin_amdgcn_perm(t0, u0, 0x05040100u);
                        st[4 * h + 2] = __builtin_amdgcn_perm(t1, u1, 0x07060302u);
                        st[4 * h + 3] = __builtin_amdgcn_perm(t1, u1, 0x05040100u);
                    }
                };
                switch (tail) {
                case 0: unplane(std::integral_constant<int, 0>{}); break;
                case 1: unplane(std::integral_constant<int, 1>{}); break;
                case 2: unplane(std::integral_constant<int, 2>{}); break;
                default: unplane(std::integral_constant<int, 3>{}); break;
                }
            } else {
            for (int i = 0; i < len; i++) {
                const u32 sym = row[i];
                const u32 top = st[0] >> 24;
                const u32 f = ENCODE ? (sym ^ top) : top;
#pragma unroll
                for (int d = 0; d + 1 < NKW; d++) st[d] = __builtin_amdgcn_alignbit(st[d], st[d + 1], 24);
                st[NKW - 1] = (st[NKW - 1] << 8) | (ENCODE ? 0u : sym);
                // table rows are stored in 16-byte chunks, chunk c of row f at T[c*1024 + f*4 ..]: a ds_read_b128 then
                // spreads over 16 bank classes instead of 8 (the 32-byte row pitch made 75 % of the LDS cycles conflicts)
#pragma unroll
                for (int c4 = 0; c4 < NKW / 4; c4++) {
                    const uint4 rv = *reinterpret_cast<const uint4 *>(T + c4 * 1024 + f * 4);
                    st[4 * c4 + 0] ^= rv.x; st[4 * c4 + 1] ^= rv.y; st[4 * c4 + 2] ^= rv.z; st[4 * c4 + 3] ^= rv.w;
                }
                if constexpr (NKW % 4 != 0) {
                    const u32 *r = T + (NKW / 4) * 1024 + f * (NKW % 4);
#pragma unroll
                    for (int d = 0; d < NKW % 4; d++) st[(NKW / 4) * 4 + d] ^= r[d];
                }
            }
            }
        }
        wave_sync();
        if (ENCODE) {
            if (parity_only) {
                // parity bytes (highest degree first) -> LDS -> global, flat
                u32 *ps = reinterpret_cast<u32 *>(stage) + rowi * NKW;
                if (rowi < count) {
#pragma unroll
                    for (int d = 0; d < NKW; d++) ps[d] = __builtin_bswap32(st[d]);
                }
                wave_sync();
                u32 *dst = reinterpret_cast<u32 *>(out + cw0 * NK);
                const u32 *pw = reinterpret_cast<const u32 *>(stage);
                if ((reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
                    for (int i = lane; i < count * NKW; i += 64) dst[i] = pw[i];
                } else {
                    for (int i = lane; i < count * NK; i += 64) out[cw0 * NK + i] = stage[i];
                }
            } else {
                if (rowi < count) {
                    uint8_t *pp = stage + rowi * ns_out + len;
#pragma unroll
                    for (int d = 0; d < NKW; d++) {
                        pp[4 * d + 0] = (uint8_t)(st[d] >> 24); pp[4 * d + 1] = (uint8_t)(st[d] >> 16);
                        pp[4 * d + 2] = (uint8_t)(st[d] >> 8);  pp[4 * d + 3] = (uint8_t)st[d];
                    }
                }
                wave_sync();
                uint8_t *dst = out + cw0 * ns_out;
                const int ob = count * ns_out;
                const int nv = (reinterpret_cast<uintptr_t>(dst) & 15) == 0 ? ob >> 4 : 0;
                for (int i = lane; i < nv; i += 64) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(stage)[i];
                for (int b = (nv << 4) + lane; b < ob; b += 64) dst[b] = stage[b];
            }
        } else {
            u32 nzw = 0;
#pragma unroll
            for (int d = 0; d < NKW; d++) nzw |= st[d];
            if (rowi < count) {
                if (rem_out) {
                    u32 *dst = reinterpret_cast<u32 *>(rem_out) + (cw0 + rowi) * NKW; // rem_out is 16-byte aligned scratch
#pragma unroll
                    for (int d = 0; d < NKW; d++) dst[d] = __builtin_bswap32(st[d]);
                }
                if (flag_out) flag_out[cw0 + rowi] = nzw != 0;
            }
        }
        wave_sync();
    }
}


// ------------------------------------------------------------------------------------------------
// The same LFSR with the codeword ROW in registers (r04): n - k = 32, full-length rows of LEN = 223 / 255 symbols
// ------------------------------------------------------------------------------------------------
// rs_lfsr_kernel stages 64 rows per wave in LDS (16 KiB), which leaves a CU eight waves and a 4-copy table whose reads conflict:
// LDS array 69 % busy, 64 % of it conflicts, two waves per SIMD to hide the table-read -> xor -> next-feedback chain.  Here the row
// lives in 64 VGPRs of its lane (rows start at any byte: gfx950 serves unaligned vector accesses under HSA), and LDS holds nothing
// but the table -- sixteen private copies, one per lane of a ds_read_b128 lane group, so NO read can conflict: byte offset
// f * 512 + c * 256 + s * 16 for chunk c of row f, copy s = lane & 15.  One workgroup per CU, four waves per SIMD.  Per symbol: one SDWA
// xor (feedback = state's top byte ^ the symbol's byte of its row register), one address, two conflict-free 16-byte reads, the planar
// state update of rs_lfsr_kernel (12 vector instructions).
// MODE 0: parity appended (full codewords), 1: parity only, 2: decoder pre-pass (remainder, non-zero flag, verbatim copy of the rows).
// Measured (profiles/r04_rs_lfsr_reg.txt, GB/s of codewords at 2^17 / 2^18 / 2^20 / 2^22 words): encode 880 / 1075 / 1065 / 1219 against
// 861 / 930 / 945 / 958 for rs_lfsr_kernel; pre-pass 498 / 611 / 782 / 802 against 562 / 656 / 688 / 732.  Probes of the same kernel: WITHOUT its
// table reads it runs no faster (the LFSR loop is not what it waits for); storing only the last 64 bytes of every row it reaches
// 1.6-1.8 TB/s, parity only 1.65-2.06: what is left is the 255-byte-pitch output written in 64-byte pieces.
typedef u32 rs_u32x4 __attribute__((ext_vector_type(4)));

template <int B> // top byte of p ^ byte B of r, zero-extended: ONE instruction
__device__ __forceinline__ u32 xor_top_byte(u32 p, u32 r)
{
    u32 f;
    if constexpr (B == 0) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_0" : "=v"(f) : "v"(p), "v"(r));
    if constexpr (B == 1) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_1" : "=v"(f) : "v"(p), "v"(r));
    if constexpr (B == 2) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_2" : "=v"(f) : "v"(p), "v"(r));
    if constexpr (B == 3) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(f) : "v"(p), "v"(r));
    return f;
}

template <class F, int... Js>
__device__ __forceinline__ void for_each_index(F &&f, std::integer_sequence<int, Js...>)
{
    (f(std::integral_constant<int, Js>{}), ...);
}

constexpr size_t LFSR_REG_LDS = 256 * 512; // 128 KiB: 256 rows x 2 chunks x 16 copies x 16 bytes

// Global accesses (v2): a quad of lanes owns four consecutive rows.  Load r of a 64-byte block takes the block of row 4Q + r with
// the quad's four lanes side by side (64 contiguous bytes per quad: 16-32 cache lines per wave instruction instead of 64), and a
// 4 x 4 transpose inside the quad -- two rounds of a DPP exchange and a bit-field select per register -- hands every lane the four
// chunks of its OWN row.  Stores run the same way backwards.  LEN = 223 / 255: blocks at bytes 0, 64, 128 and LEN - 64 (the last
// one ends at the row's end and overlaps its predecessor; its registers serve the symbols from byte 192 on).
__device__ __forceinline__ void quad_transpose(u32 (&t)[4], u32 m0, u32 m1)
{ // t[r] of lane k <-> t[k] of lane r within a quad; m0 / m1: all-ones on lanes whose bit 0 / bit 1 is clear
#pragma unroll
    for (int a = 0; a < 4; a += 2) {
        const u32 x = (u32)__builtin_amdgcn_update_dpp(0, (int)t[a], 0xB1, 0xf, 0xf, false);     // quad_perm [1,0,3,2]
        const u32 y = (u32)__builtin_amdgcn_update_dpp(0, (int)t[a + 1], 0xB1, 0xf, 0xf, false);
        t[a + 1] = (m0 & x) | (~m0 & t[a + 1]);
        t[a] = (m0 & t[a]) | (~m0 & y);
    }
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const u32 x = (u32)__builtin_amdgcn_update_dpp(0, (int)t[a], 0x4E, 0xf, 0xf, false);     // quad_perm [2,3,0,1]
        const u32 y = (u32)__builtin_amdgcn_update_dpp(0, (int)t[a + 2], 0x4E, 0xf, 0xf, false);
        t[a + 2] = (m1 & x) | (~m1 & t[a + 2]);
        t[a] = (m1 & t[a]) | (~m1 & y);
    }
}

template <int LEN, int MODE>
__global__ __launch_bounds__(1024) void rs_lfsr_reg_kernel(const u32 *__restrict__ rowtab, const uint8_t *in, uint8_t *out, uint8_t *__restrict__ rem_out,
                                                           uint8_t *__restrict__ flag_out, i64 batch)
{
    static_assert(LEN == 223 || LEN == 255, "three whole 64-byte blocks and an end-aligned fourth");
    constexpr int NKW = 8, W = 2, LASTB = LEN - 64; // byte offset of the fourth block
    constexpr bool ENCODE = MODE != 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) { // 16-byte chunk c of row f: ONE load, sixteen copies
        const int c = i & 1, f = i >> 1;
        const uint4 v = reinterpret_cast<const uint4 *>(rowtab)[c * 256 + f];
        uint4 *d = reinterpret_cast<uint4 *>(lds_raw + f * 512 + c * 256);
#pragma unroll
        for (int k = 0; k < 16; k++) d[k] = v;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const u32 slot = (u32)(lane & 15) << 4;
    const u32 m0 = (lane & 1) ? 0u : ~0u, m1 = (lane & 2) ? 0u : ~0u;
    const int kq = lane & 3; // this lane's chunk inside a quad access
    __syncthreads();
    for (i64 cw0 = ((i64)blockIdx.x * nwaves + wave) * 64; cw0 < batch; cw0 += (i64)gridDim.x * nwaves * 64) {
        const i64 cw = cw0 + lane;
        // D[g][4 c + w]: word w of chunk c of block g of this lane's own row (after the transposes)
        u32 D[4][16];
        // buffer addressing: descriptors on this wave's 64 rows (input pitch LEN, output pitch LEN / 255), ONE 32-bit lane offset, row and
        // block offsets as immediates; rows past the end of the batch are out of range (loads return 0, stores are dropped)
        const i64 rows_left = batch - cw0 < 64 ? batch - cw0 : 64;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + cw0 * LEN), 0, (u32)(rows_left * LEN), 0x00020000);
        const u32 vin = (u32)(lane & ~3) * (u32)LEN + 16u * (u32)kq;
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + cw0 * 255), 0, MODE == 0 ? (u32)(rows_left * 255) : 0u, 0x00020000);
        const u32 vout = (u32)(lane & ~3) * 255u + 16u * (u32)kq;
        rs_u32x4 T[2][4]; // two blocks in flight: block g + 1 travels while block g is transposed (all four at once would not fit 128 registers)
        auto load_block = [&](auto gc, rs_u32x4 (&t)[4]) {
            constexpr int G = decltype(gc)::value;
#pragma unroll
            for (int r = 0; r < 4; r++) t[r] = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)(vin + (u32)(r * LEN + (G < 3 ? 64 * G : LASTB))), 0, 0);
        };
        load_block(std::integral_constant<int, 0>{}, T[0]);
        auto take_block = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + 1 < 4) load_block(std::integral_constant<int, g + 1>{}, T[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            rs_u32x4(&t)[4] = T[g & 1];
            if constexpr (MODE == 2) { // the received rows, verbatim, in the very form they arrived in
                if (out && out != in) {
                    const __amdgpu_buffer_rsrc_t rcp = __builtin_amdgcn_make_buffer_rsrc((void *)(out + cw0 * LEN), 0, (u32)(rows_left * LEN), 0x00020000);
#pragma unroll
                    for (int r = 0; r < 4; r++) __builtin_amdgcn_raw_buffer_store_b128(t[r], rcp, (int)(vin + (u32)(r * LEN + (g < 3 ? 64 * g : LASTB))), 0, 0);
                }
            }
            u32 tx[4] = {t[0][0], t[1][0], t[2][0], t[3][0]}, ty[4] = {t[0][1], t[1][1], t[2][1], t[3][1]};
            u32 tz[4] = {t[0][2], t[1][2], t[2][2], t[3][2]}, tw[4] = {t[0][3], t[1][3], t[2][3], t[3][3]};
            quad_transpose(tx, m0, m1);
            __builtin_amdgcn_sched_barrier(0);
            quad_transpose(ty, m0, m1);
            __builtin_amdgcn_sched_barrier(0);
            quad_transpose(tz, m0, m1);
            __builtin_amdgcn_sched_barrier(0);
            quad_transpose(tw, m0, m1);
#pragma unroll
            for (int c = 0; c < 4; c++) { D[g][4 * c + 0] = tx[c]; D[g][4 * c + 1] = ty[c]; D[g][4 * c + 2] = tz[c]; D[g][4 * c + 3] = tw[c]; }
            __builtin_amdgcn_sched_barrier(0);
        };
        for_each_index(take_block, std::make_integer_sequence<int, 4>{});
        u32 P[4][W];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int h = 0; h < W; h++) P[q][h] = 0;
        auto step = [&](auto jc) {
            constexpr int J = decltype(jc)::value, K = J & 3;
            constexpr int G = J < 192 ? J / 64 : 3, BB = J < 192 ? J % 64 : J - LASTB, BQ = BB & 3; // block, byte inside it
            const u32 rw = D[G][BB / 4];
            const u32 f = ENCODE ? xor_top_byte<BQ>(P[K][0], rw) : P[K][0] >> 24;
            P[K][0] = __builtin_amdgcn_alignbit(P[K][0], P[K][1], 24);
            if constexpr (ENCODE) P[K][1] = P[K][1] << 8;
            else P[K][1] = __builtin_amdgcn_perm(P[K][1], rw, 0x06050400u + (u32)BQ); // {P.b2, P.b1, P.b0, rw.byte BQ}
            const u32 a = (f << 9) | slot;
            const uint4 r0 = *reinterpret_cast<const uint4 *>(lds_raw + a), r1 = *reinterpret_cast<const uint4 *>(lds_raw + a + 256);
            const u32 w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int q = 0; q < 8; q++) P[(q / W + K + 1) & 3][q % W] ^= w[q]; // planar word q: role q / W, word q % W
        };
        for_each_index(step, std::make_integer_sequence<int, LEN>{});
        // back to consecutive bytes (role r is P[(r + LEN) & 3]); S[d] = the d-th remainder word in MEMORY order (highest degree first)
        u32 S[NKW];
        {
            constexpr int K = LEN & 3;
#pragma unroll
            for (int h = 0; h < W; h++) {
                const u32 A = P[K & 3][h], Bv = P[(K + 1) & 3][h], C = P[(K + 2) & 3][h], Dd = P[(K + 3) & 3][h];
                const u32 t0 = __builtin_amdgcn_perm(A, Bv, 0x07030602u), t1 = __builtin_amdgcn_perm(A, Bv, 0x05010400u);
                const u32 u0 = __builtin_amdgcn_perm(C, Dd, 0x07030602u), u1 = __builtin_amdgcn_perm(C, Dd, 0x05010400u);
                S[4 * h + 0] = __builtin_bswap32(__builtin_amdgcn_perm(t0, u0, 0x07060302u));
                S[4 * h + 1] = __builtin_bswap32(__builtin_amdgcn_perm(t0, u0, 0x05040100u));
                S[4 * h + 2] = __builtin_bswap32(__builtin_amdgcn_perm(t1, u1, 0x07060302u));
                S[4 * h + 3] = __builtin_bswap32(__builtin_amdgcn_perm(t1, u1, 0x05040100u));
            }
        }
        if constexpr (MODE == 1 || MODE == 2) { // 32 bytes per word, the wave's 64 words contiguous: plain aligned stores
            if (cw < batch) {
                uint4 *dst = reinterpret_cast<uint4 *>((MODE == 1 ? out : rem_out) + cw * 32);
                dst[0] = make_uint4(S[0], S[1], S[2], S[3]);
                dst[1] = make_uint4(S[4], S[5], S[6], S[7]);
                if (MODE == 2 && flag_out) flag_out[cw] = (S[0] | S[1] | S[2] | S[3] | S[4] | S[5] | S[6] | S[7]) != 0;
            }
        } else {
            // ALL of a row leaves at the end of the pass, within a few microseconds.  Message blocks stored as they arrive would leave every
            // 128-byte line of the output half written for the ~50 us of the LFSR loop; the L2 (4 MiB per XCD, 8 MB written per pass) evicts
            // them partial, and the kernel ran at 1.06 instead of the 1.6-1.8 TB/s it reaches without those stores.  Holding the message in
            // registers until here costs 384 bytes of scratch and is slower still (0.81 TB/s) -- so message bytes 0 .. 191 are READ AGAIN
            // (in the quad form they are stored in: no transposes; the rows are a pass old and come from L2 / the Infinity Cache) and
            // stored at once; the block at byte 191 is message bytes 191 .. 222 (words 8 .. 15 of the fourth message block, which starts
            // at byte 159) and the 32 parity bytes (profiles/r04_rs_lfsr_reg.txt).
#pragma unroll
            for (int g = 0; g < 3; g++) {
                rs_u32x4 t[4];
#pragma unroll
                for (int r = 0; r < 4; r++) t[r] = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)(vin + (u32)(r * LEN + 64 * g)), 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) __builtin_amdgcn_raw_buffer_store_b128(t[r], rout, (int)(vout + (u32)(r * 255 + 64 * g)), 0, 0);
            }
            u32 tx[4], ty[4], tz[4], tw[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c < 2) { tx[c] = D[3][8 + 4 * c]; ty[c] = D[3][9 + 4 * c]; tz[c] = D[3][10 + 4 * c]; tw[c] = D[3][11 + 4 * c]; }
                else { tx[c] = S[4 * (c - 2)]; ty[c] = S[4 * (c - 2) + 1]; tz[c] = S[4 * (c - 2) + 2]; tw[c] = S[4 * (c - 2) + 3]; }
            }
            quad_transpose(tx, m0, m1); quad_transpose(ty, m0, m1); quad_transpose(tz, m0, m1); quad_transpose(tw, m0, m1);
#pragma unroll
            for (int r = 0; r < 4; r++)
                __builtin_amdgcn_raw_buffer_store_b128(rs_u32x4{tx[r], ty[r], tz[r], tw[r]}, rout, (int)(vout + (u32)(r * 255 + 191)), 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fast decoder for characteristic-2 codes with n-k <= 60: second kernel of the two-kernel decode
// ------------------------------------------------------------------------------------------------
// rs_lfsr_kernel has already reduced every received word modulo g(x) and copied the received rows to the output.  Words
// with a zero remainder and no erasures need nothing more.  The rest are decoded ONE CODEWORD PER WAVEFRONT with the same
// mathematics and failure exits as rs_decode_kernel / bch_decode_jit.  r02's version was vector-issue bound (PMC: 89 M vector
// instructions per 2^17 words); this one halves them and keeps the LDS gathers off each other's banks:
//   * syndromes S_j = rem(root_j), Chien search Lambda_total(x) (four points per lane) and Forney's numerator / denominator are
//     Horner recurrences through the 64 KiB LDS product table: per term one byte gather and ONE vector instruction (horner_step);
//   * Berlekamp-Massey in the inversionless RiBM arrangement, in a frame that moves one lane per step (bm_run, written in
//     assembly: 4 vector instructions + 2 gathers per step, early stop on an all-zero tail);
//   * corrected symbols are patched into the output row in place (erased symbols become E, others r ^ E).
// Per-wave LDS scratch with a compile-time layout: every array is `base + constant`, so the addresses ride in the
// immediate offset field of the LDS instructions instead of ten live VGPRs (the kernel is register-bound).
// S = slots per array (>= d - 1 + 4): 40 for d - 1 <= 36, 64 otherwise.
template <int S>
struct WaveScratch2 {
    uint8_t *base;
    static constexpr int BYTES = 7 * S;
    __device__ __forceinline__ uint8_t *synd() const { return base; }
    __device__ __forceinline__ uint8_t *gamma() const { return base + S; }
    __device__ __forceinline__ uint8_t *sprime() const { return base + 2 * S; }
    __device__ __forceinline__ uint8_t *lam() const { return base + 3 * S; }
    __device__ __forceinline__ uint8_t *epos() const { return base + 4 * S; }
    __device__ __forceinline__ uint8_t *errpos() const { return base + 5 * S; }
    __device__ __forceinline__ uint8_t *errloc() const { return base + 6 * S; }
};

// LDS accesses by 32-bit LDS address (no generic-pointer arithmetic): the product table is symmetric, so the product of a
// per-lane (or wave-uniform) factor f and a running value a is the byte at row(f) + a with row(f) = table + 256 f held in a
// register, and a Horner step  a' = T[a] ^ c  followed by the next address  row + a'  is ONE v_xad_u32 ((T ^ c) + row).
typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
__device__ __forceinline__ u32 lds_addr(const uint8_t *p) { return (u32)(size_t)(lds_cu8 *)p; }
__device__ __forceinline__ u32 lds_ld8(u32 a) { return *(lds_cu8 *)(size_t)a; }
// One Horner step on an ADDRESS register a = (running value << 8) | x (x: the lane's fixed evaluation point, byte 0): with
// T = table[a] just gathered, byte 1 of a becomes (T ^ c) and byte 0 stays -- the next gather address in ONE instruction
// (SDWA destination byte select, the rest of the register preserved).  c is wave-uniform; only its low byte takes part.
__device__ __forceinline__ void horner_step(u32 &a, u32 T, u32 c)
{
    asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(a) : "v"(T), "s"(c));
}
// the same with c = byte B of a wave-uniform word (the byte select of the second source: no scalar shift)
template <int B>
__device__ __forceinline__ void horner_step_byte(u32 &a, u32 T, u32 w)
{
    if constexpr (B == 0) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0" : "+v"(a) : "v"(T), "s"(w));
    if constexpr (B == 1) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(a) : "v"(T), "s"(w));
    if constexpr (B == 2) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_2" : "+v"(a) : "v"(T), "s"(w));
    if constexpr (B == 3) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_3" : "+v"(a) : "v"(T), "s"(w));
}
template <int B0, int B1>
__device__ __forceinline__ void horner_bytes2(u32 &a1, u32 &a2, const u32 (&w)[8], u32 tbl_off)
{ // two independent chains side by side: a1 takes bytes B0 .. B1 - 1 of the eight words (memory order), a2 bytes 16 + B0 ..
    if constexpr (B0 < B1) {
        const u32 T1 = lds_ld8(a1 + tbl_off), T2 = lds_ld8(a2 + tbl_off);
        horner_step_byte<B0 & 3>(a1, T1, w[B0 >> 2]);
        horner_step_byte<B0 & 3>(a2, T2, w[4 + (B0 >> 2)]);
        horner_bytes2<B0 + 1, B1>(a1, a2, w, tbl_off);
    }
}
// Chien search: the four evaluation points of lane l are xl | CHIEN_X(s), xl = ((l & 31) << 2) | (l >> 5): the elements of the
// field enumerated by VALUE, so that the 32 lanes of a half-wave read 32 different LDS banks whatever the running values are
// (bank = bits 2..6 of the column); the position a point stands for comes from a 256-byte table built with the code.
#define CHIEN_X(s) ((((s) & 1) << 1) | (((s) >> 1) << 7))

// Berlekamp-Massey steps r .. limit - 1 in the moving frame (see the kernel), written out instruction by instruction: the
// compiler's versions of this loop carry 6-8 vector instructions per step, this one 4 (v_readfirstlane, v_add_dpp = index of
// gamma * (X shifted), v_add = index of d0 * Y, v_xor) plus two on the steps that change Y and gamma, and a zero discrepancy
// with nothing but zeros ahead (lanes 0 .. nsq - r - 1 of X: the first zero step of a word with v <= t errors) ends the run.
// X, Y: the two polynomials; G: 256 * gamma in every lane; L: the LFSR length; r: steps taken (the frame has moved r lanes).
// The product table sits at LDS offset 16 (checked at kernel entry).  Manual wait states: a DPP read needs two instructions
// after the vector write of its source (there are at least five on every path).
__device__ __forceinline__ void bm_run(u32 &X, u32 &Y, u32 &G, int &L, int &r, int limit, int nsq)
{
    u32 a1, a2, t1, t2;
    int d0, row, tmp, twoL;
    asm volatile(
        "s_cmp_ge_i32 %[r], %[limit]\n\t"
        "s_cbranch_scc1 BM_END%=\n\t"
        "s_lshl_b32 %[twoL], %[L], 1\n\t"
        "BM_TOP%=:\n\t"
        "v_readfirstlane_b32 %[d0], %[X]\n\t"
        "s_cmp_eq_u32 %[d0], 0\n\t"
        "s_cbranch_scc1 BM_ZERO%=\n\t"
        "s_lshl_b32 %[row], %[d0], 8\n\t"
        "v_add_u32_dpp %[a1], %[X], %[G] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 %[a2], %[row], %[Y]\n\t"
        "ds_read_u8 %[t1], %[a1] offset:16\n\t"
        "ds_read_u8 %[t2], %[a2] offset:16\n\t"
        "s_cmp_gt_i32 %[twoL], %[r]\n\t"
        "s_cbranch_scc1 BM_SAME%=\n\t"
        "v_mov_b32_dpp %[Y], %[X] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mov_b32_e32 %[G], %[row]\n\t"
        "s_sub_i32 %[L], %[r], %[L]\n\t"
        "s_add_i32 %[L], %[L], 1\n\t"
        "s_lshl_b32 %[twoL], %[L], 1\n\t"
        "BM_SAME%=:\n\t"
        "s_add_i32 %[r], %[r], 1\n\t"
        "s_cmp_lt_i32 %[r], %[limit]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_xor_b32_e32 %[X], %[t1], %[t2]\n\t"
        "s_cbranch_scc1 BM_TOP%=\n\t"
        "s_branch BM_END%=\n\t"
        "BM_ZERO%=:\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[X]\n\t"
        "s_sub_i32 %[tmp], %[nsq], %[r]\n\t"
        "s_sub_i32 %[tmp], 32, %[tmp]\n\t"
        "s_lshl_b32 %[tmp], vcc_lo, %[tmp]\n\t"
        "s_cmp_eq_u32 %[tmp], 0\n\t"
        "s_cbranch_scc1 BM_END%=\n\t"
        "s_add_i32 %[r], %[r], 1\n\t"
        "v_mov_b32_dpp %[X], %[X] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_cmp_lt_i32 %[r], %[limit]\n\t"
        "s_cbranch_scc1 BM_TOP%=\n\t"
        "BM_END%=:"
        : [X] "+v"(X), [Y] "+v"(Y), [G] "+v"(G), [L] "+s"(L), [r] "+s"(r), [a1] "=&v"(a1), [a2] "=&v"(a2), [t1] "=&v"(t1),
          [t2] "=&v"(t2), [d0] "=&s"(d0), [row] "=&s"(row), [tmp] "=&s"(tmp), [twoL] "=&s"(twoL)
        : [limit] "s"(limit), [nsq] "s"(nsq)
        : "vcc", "scc", "memory");
}

__device__ __forceinline__ int lane_shift_up1(int v, int)
{ // lane i <- lane i-1 across the whole wavefront, lane 0 <- 0: one DPP move (wave_shr:1, GFX9 family incl. gfx950;
  // checked on hardware by tools/ubench/wave_shr.hip).  The row_shr + 3 readlane + 3 select form it replaces cost 7.
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
}

// ---- test-only: the hand-assembled bm_run against a loop the COMPILER generates from the same recurrence ----
// One wave per sequence S[0 .. nsq - 1] (bytes of GF(2^8)); both forms start from the kernel's initial frame and run the kernel's
// two calls (31 steps, then the 32nd with the product half of Y cleared).  Every lane's X and Y, gamma's row, L and the step
// count must agree; out[seq] = number of lanes that differ (+ 64 per differing scalar).
__device__ __forceinline__ void bm_ref_run(u32 &X, u32 &Y, u32 &G, int &L, int &r, int limit, int nsq, const uint8_t *mul_t, int lane)
{
    while (r < limit) {
        const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)X);
        const u32 A = (u32)__builtin_amdgcn_update_dpp(0, (int)X, 0x130, 0xf, 0xf, true); // lane l <- lane l + 1, lane 63 <- 0
        if (d0 == 0) {
            const unsigned long long nz = __builtin_amdgcn_ballot_w64(X != 0 && lane < nsq - r);
            if (nz == 0) return; // nothing but zero discrepancies to come
            X = A;
            r++;
            continue;
        }
        const u32 Xn = (u32)mul_t[G + A] ^ (u32)mul_t[(d0 << 8) + Y];
        if (!(2 * L > r)) { Y = A; G = d0 << 8; L = r + 1 - L; }
        X = Xn;
        r++;
    }
}

__global__ __launch_bounds__(256) void rs_bm_selftest_kernel(const uint8_t *__restrict__ mul8, const uint8_t *__restrict__ seqs,
                                                             const int *__restrict__ lens, int *__restrict__ out, i64 nseq)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint8_t *mul_t = lds_raw + 16; // where bm_run's immediates expect the product table (dynamic LDS starts at 0: no static LDS here)
    {
        const uint4 *s = reinterpret_cast<const uint4 *>(mul8);
        uint4 *d = reinterpret_cast<uint4 *>(mul_t);
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) d[i] = s[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const i64 wave0 = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
    for (i64 q = wave0; q < nseq; q += nwaves) {
        const int nsq = __builtin_amdgcn_readfirstlane(lens[q]);
        const u32 x0 = lane == 63 ? 1u : (lane < nsq ? (u32)seqs[q * 32 + lane] : 0u);
        u32 Xa = x0, Ya = x0, Ga = 1u << 8, Xb = x0, Yb = x0, Gb = 1u << 8;
        int La = 0, ra = 0, Lb = 0, rb = 0;
        const int rmain = nsq < 31 ? nsq : 31;
        bm_run(Xa, Ya, Ga, La, ra, rmain, nsq);
        if (ra == 31 && nsq == 32) {
            Ya = lane < 32 ? 0u : Ya;
            bm_run(Xa, Ya, Ga, La, ra, 32, nsq);
        }
        bm_ref_run(Xb, Yb, Gb, Lb, rb, rmain, nsq, mul_t, lane);
        if (rb == 31 && nsq == 32) {
            Yb = lane < 32 ? 0u : Yb;
            bm_ref_run(Xb, Yb, Gb, Lb, rb, 32, nsq, mul_t, lane);
        }
        const unsigned long long bad = __builtin_amdgcn_ballot_w64(Xa != Xb || Ya != Yb || Ga != Gb);
        if (lane == 0) out[q] = __popcll(bad) + (La != Lb ? 64 : 0) + (ra != rb ? 64 : 0);
    }
}

// WPS = resident waves per SIMD the register budget is sized for (block = 2 * WPS waves, two blocks per CU)
template <int S, int WPS>
__global__ __launch_bounds__(128 * WPS) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void rs_decode_bin_kernel(RsTables t, RsParams rp, const uint8_t *__restrict__ eras_g,
                                                             const uint8_t *__restrict__ rem_g, const uint8_t *__restrict__ aux_g, int n,
                                                             uint8_t *__restrict__ out_g, i64 *__restrict__ nerr_g, i64 batch)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    Arith8<true> ar;
    uint8_t *free_l = stage_tables<true>(lds_raw + 16, t, ar, rp.qm1, blockDim.x); // bytes 0..15: the claim counter (ds_append wants a 16-bit address)
    const int dd = rp.nroots, qm1 = rp.qm1, la = rp.log_alpha;
    const int nk = rp.n - rp.k; // length of the remainder r(x) mod g(x); equals dd for Reed-Solomon, larger for BCH
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    WaveScratch2<S> ws;
    ws.base = free_l + (size_t)wave * WaveScratch2<S>::BYTES;
    __syncthreads();
    // The product table is used in two ways.  (a) One operand wave-uniform (Berlekamp-Massey, Omega): row = that operand, the
    // lanes read one 256-byte row, at most two dwords per bank.  (b) Horner's rule at a per-lane point x (syndromes, Chien,
    // Forney): row = running value, column = x, so the bank depends on x alone and the lane <-> point assignment decides the
    // conflicts: the Chien points are enumerated by value (no conflicts), the syndrome roots are dealt to the lanes by the host
    // so that a half-wave holds one root per bank where the roots allow it (aux: 64 x-values, 64 source lanes, 256 positions).
    // PMC history of this kernel (2^17 words, e ~ U{0..16}): r02 89 M vector instructions / 81 M LDS cycles (50 % conflicts);
    // every product as row(x) + value, one v_xad per step: 53 M / 121 M (72 % conflicts, slower); this arrangement: see DESIGN.
    // The Horner gathers add the table offset as an immediate and ds_append addresses the claim counter through M0[15:0]: the
    // dynamic LDS block must start at address 0, i.e. the kernel must own no static LDS -- checked on the host before the launch
    // (decode_lds_base_is_zero; GFA_ERR_UNSUPPORTED instead of a device trap).
    constexpr u32 TBL = 16;
    u32 synp = (u32)aux_g[lane] | ((u32)aux_g[64 + lane] << 10); // byte 0: root evaluated by this lane; bits 8..15: 4 * (lane holding S_lane)
    { // byte 2: the root's 16th power
        u32 x16 = synp & 0xffu;
        for (int i = 0; i < 4; i++) x16 = ar.mul_t[(x16 << 8) | x16];
        synp |= x16 << 16;
    }
    const u32 xl = (u32)((lane & 31) << 2) | (u32)(lane >> 5);
    u32 posp = 0; // positions of the four Chien points of this lane (255: not a position of the code)
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) posp |= (u32)aux_g[128 + (xl | CHIEN_X(s4))] << (8 * s4);
    // Codewords are CLAIMED, not dealt out: a word with v errors costs about 3 + v units, every wave is resident from the start, and
    // with a static deal the launch lasts as long as its unluckiest wave (+38 % over the mean for 16 words per wave at e ~ U{0..16}).
    // Each workgroup owns a contiguous range; a wave takes its next word with ONE ds_append on a counter at LDS offset 0 (returns the
    // counter, adds the 64 active lanes; wave-uniform result).  Two things found the hard way: the counter
    // must sit BELOW 64 KiB -- ds_append takes its address through M0[15:0], and with the counter behind the tables (offset 71296)
    // words were claimed twice or never; and the `if (lane == 0) atomicAdd(...)` + readfirstlane form hangs in this kernel (LDS or
    // global counter alike, even as a bare loop), cause not isolated.
    // Measured and not kept: syndromes as two half-wave Horner chains (0.286 vs 0.261 ms: the select between two scalar
    // broadcasts per step is a v_cndmask on VCC), leaving Berlekamp-Massey at the first all-zero tail (0.281 ms).
    unsigned int *claim_p = reinterpret_cast<unsigned int *>(lds_raw);
    if (threadIdx.x == 0) *claim_p = 0;
    __syncthreads();
    const i64 per_block = rp.per_block; // host-computed
    const i64 cw_lo = (i64)blockIdx.x * per_block;
    const i64 cw_hi_ = cw_lo + per_block < batch ? cw_lo + per_block : batch;
    const unsigned int count = (unsigned int)__builtin_amdgcn_readfirstlane((int)(cw_hi_ > cw_lo ? cw_hi_ - cw_lo : 0));
    const bool nk32 = nk == 32; // the remainder is eight aligned words: read through the scalar cache, bytes picked by s_bfe
    for (;;) {
        typedef __attribute__((address_space(3))) int lds_int;
        const unsigned int idx = (unsigned int)__builtin_amdgcn_readfirstlane((int)((unsigned int)__builtin_amdgcn_ds_append((lds_int *)claim_p) >> 6));
        if (idx >= count) break;
        const i64 cw = cw_lo + (i64)idx;
        uint8_t *orow = out_g + cw * n; // already holds the received row (copied by the pre-pass)
        // remainder, stored highest degree first: rw[] (wave-uniform words) when n - k = 32, else coefficient of x^lane per lane
        u32 rw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        u32 remc = 0;
        bool any_nz;
        if (nk32) {
            const u32 *rp32 = reinterpret_cast<const u32 *>(rem_g + cw * 32);
#pragma unroll
            for (int i = 0; i < 8; i++) rw[i] = rp32[i];
            any_nz = (rw[0] | rw[1] | rw[2] | rw[3] | rw[4] | rw[5] | rw[6] | rw[7]) != 0;
        } else {
            remc = lane < nk ? rem_g[cw * nk + (nk - 1 - lane)] : 0;
            any_nz = __builtin_amdgcn_ballot_w64(remc != 0) != 0;
        }
        if (!any_nz && !eras_g) { // clean word (_bch.py:1373-1376)
            if (lane == 0) nerr_g[cw] = 0;
            continue;
        }
        // ---- erased positions, ascending degree (_bch.py:1351-1355; the pre-pass already treated them as zeros) ----
        int u = 0;
        if (eras_g) {
            for (int base = 0; base < n; base += 64) {
                const int i = base + lane;
                const bool er = i < n && eras_g[cw * n + (n - 1 - i)] != 0;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(er);
                const int before = (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                if (er && u + before < dd + 4) ws.epos()[u + before] = (uint8_t)i;
                u += __popcll(m);
            }
        }
        int status = 0, v = 0;
        if (u > dd) {
            status = -1;
        } else if (!any_nz && u == 0) {
            status = 1;
        } else {
            // ---- 1. syndromes from the remainder: S_j = rem(root_j) by Horner's rule through the product table ----
            // (per term one LDS gather and one SDWA xor, see horner_step)
            u32 synd;
            {
                u32 a, last;
                if (nk32) {
                    // r(x) = H1(x) x^16 + H2(x): two 15-step chains side by side (half the dependent gathers), then one product
                    // by this lane's x^16 (byte 2 of synp) and one xor
                    u32 a1 = (synp & 0xffu) | ((rw[0] & 0xffu) << 8), a2 = (synp & 0xffu) | ((rw[4] & 0xffu) << 8);
                    horner_bytes2<1, 16>(a1, a2, rw, TBL);
                    // gfx940-class parts want one wait state between an SDWA write with a byte destination and a VECTOR instruction that
                    // reads the register (the compiler inserts it for its own SDWA code, not behind inline assembly); the LDS reads
                    // that follow every other horner_step are not affected
                    asm("s_nop 0" : "+v"(a1), "+v"(a2));
                    a = __builtin_amdgcn_perm(a1, synp, 0x0c0c0502u); // byte 1 <- H1, byte 0 <- x^16
                    last = a2 >> 8;
                } else {
                    a = (synp & 0xffu) | ((u32)__builtin_amdgcn_readlane((int)remc, nk - 1) << 8);
                    for (int tt = nk - 2; tt >= 1; tt--) horner_step(a, lds_ld8(a + TBL), (u32)__builtin_amdgcn_readlane((int)remc, tt));
                    last = (u32)__builtin_amdgcn_readlane((int)remc, 0);
                }
                // the lane that evaluated root j hands S_j to lane j
                synd = (u32)__builtin_amdgcn_ds_bpermute((int)((synp >> 8) & 0xffu), (int)(lds_ld8(a + TBL) ^ last));
            }
            // The word without erasures over a code with d - 1 <= 32 (the common case) keeps S' = S, Gamma = 1 and Lambda in
            // registers; everything else goes through the per-wave LDS arrays as before.
            const bool inreg = u == 0 && dd <= 32;
            int glen = 1;
            u32 sp; // S'[lane] (0 from lane d - 1 upward)
            if (inreg) {
                sp = lane < dd ? synd : 0u;
            } else {
                if (lane < dd) ws.synd()[lane] = (uint8_t)synd;
                wave_sync();
                // ---- 2. erasure locator (_bch.py:1389-1393) ----
                if (lane == 0) ws.gamma()[0] = 1;
                wave_sync();
                for (int k = 0; k < u; k++) {
                    const int e = ws.epos()[k];
                    const u32 Yk = ar.exp_t[(la * e) % qm1];
                    u32 g = 0;
                    if (lane <= glen) {
                        const u32 gi = lane < glen ? ws.gamma()[lane] : 0;
                        const u32 gm = lane >= 1 ? ws.gamma()[lane - 1] : 0;
                        g = gi ^ ar.mul(gm, Yk);
                    }
                    wave_sync();
                    if (lane <= glen) ws.gamma()[lane] = (uint8_t)g;
                    glen++;
                    wave_sync();
                }
                // ---- 3. modified syndromes S' = Gamma * S mod x^(d-1) (_bch.py:1408-1409) ----
                sp = 0;
                if (lane < dd) {
                    const int imax = lane < glen - 1 ? lane : glen - 1;
                    for (int i = 0; i <= imax; i++) sp ^= ar.mul(ws.gamma()[i], ws.synd()[lane - i]);
                    ws.sprime()[lane] = (uint8_t)sp;
                }
                wave_sync();
            }
            // ---- 4. Berlekamp-Massey on S'[u:], coefficients one per lane (_lfsr.py:1647-1702) ----
            int llen = 1;
            const int nsq = dd - u;
            u32 Creg = lane == 0 ? 1u : 0u;
            int L = 0; // LFSR length found by Berlekamp-Massey
            if (nsq > 0) {
                if (dd <= 32) {
                    // Inversionless Berlekamp-Massey without a discrepancy reduction (the RiBM arrangement of Sarwate &
                    // Shanbhag) in a frame that moves down one lane per step, so that ONE full-wave shift serves both halves:
                    // lanes 0..31 hold the coefficients r.. of Lambda*S (X) and B*S (Y), the discrepancy of step r is X[0];
                    // the locator half starts at lane 63 and after r steps Lambda_i (X) and (x^k B)_i (Y) sit at lane 63 - r + i.
                    //     A = X shifted down one lane (zero fill),   X' = gamma*A - d0*Y,   Y' = (d0 != 0 && 2L <= r) ? A : Y
                    // (in the old static frame: Lambda stays and x*B moves up, the products move down and stay).  The halves never
                    // meet (the gap between them is zero) except in the 32nd step of a 32-step run, when Lambda_0 lands on lane 31:
                    // the product half of Y, no longer needed then, is cleared first.  Lambda comes out multiplied by a non-zero
                    // constant; Omega' = Lambda*S' carries the same constant and Forney's quotient, the roots and the degree are
                    // unchanged.  Per step with a non-zero discrepancy: v_readfirstlane, v_add_dpp (index of gamma*A: row(gamma) is
                    // a register refreshed when gamma changes), v_add (index of d0*Y: row(d0) is scalar), two gathers, v_xor.
                    u32 X = lane == 63 ? 1u : (lane < nsq ? (inreg ? sp : (u32)ws.sprime()[u + lane]) : 0u);
                    u32 Y = X;
                    int moves = 0; // steps taken = lanes the frame has moved
                    {
                        u32 G = 1u << 8;
                        const int rmain = nsq < 31 ? nsq : 31;
                        bm_run(X, Y, G, L, moves, rmain, nsq);
                        if (moves == 31 && nsq == 32) { // (a run cannot stop at 31: its last test is before step 30)
                            Y = lane < 32 ? 0u : Y;
                            bm_run(X, Y, G, L, moves, 32, nsq);
                        }
                    }
                    // Lambda_i is at lane 63 - moves + i: bring it to lane i
                    const u32 moved = (u32)__builtin_amdgcn_ds_bpermute(((lane + 63 - moves) & 63) << 2, (int)X);
                    Creg = (lane < 32 && lane <= moves) ? moved : 0u;
                } else {
                    const int Sall = lane < nsq ? (int)ws.sprime()[u + lane] : 0;
                    // Bs holds x^m * B(x) / b, so the update C -= (d/b) x^m B is ONE table gather on the critical path
                    u32 Bs = (lane == 1 && nsq > 1) ? 1u : 0u; // m = 1, B = 1, b = 1
                    int Sreg = 0;                               // S[k - lane]
                    for (int k = 0; k < nsq; k++) {
                        Sreg = lane_shift_up1(Sreg, lane);
                        const int sk = __builtin_amdgcn_readlane(Sall, k);
                        if (lane == 0) Sreg = sk;
                        const u32 term = lane <= L ? ar.mul((u32)Sreg, Creg) : 0u;
                        const u32 dsc = ar.wave_sum(term);
                        u32 nextB = Bs;
                        if (dsc != 0) {
                            const u32 cnew = Creg ^ ar.mul(dsc, Bs);
                            if (!(2 * L > k)) {
                                nextB = ar.mul(ar.inv(dsc), Creg); // new B / new b = C_old / d
                                L = k + 1 - L;
                            }
                            Creg = cnew;
                        }
                        const int sh = lane_shift_up1((int)nextB, lane);
                        Bs = lane < nsq ? (u32)sh : 0u;
                    }
                }
                const int clen = L + 1 < nsq ? L + 1 : nsq;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(lane < clen && Creg != 0);
                llen = mk ? 64 - __clzll((long long)mk) : 1;
            }
            Creg = lane < llen ? Creg : 0u;
            v = llen - 1;
            if (2 * v + u > dd) {
                status = -1; // _bch.py:1431-1433
            } else {
                // ---- 5. Lambda_total = Gamma * Lambda ----
                const int ltlen = glen + llen - 1;
                u32 ltk = Creg; // Gamma = 1 without erasures
                if (u > 0) {
                    if (lane < dd + 4) ws.lam()[lane] = (uint8_t)Creg;
                    wave_sync();
                    ltk = 0;
                    if (lane < ltlen) {
                        const int ilo = lane - (llen - 1) > 0 ? lane - (llen - 1) : 0;
                        const int ihi = lane < glen - 1 ? lane : glen - 1;
#pragma unroll 4
                        for (int i = ilo; i <= ihi; i++) ltk ^= ar.mul(ws.gamma()[i], ws.lam()[lane - i]);
                    }
                }
                // ---- 6. Chien search: Lambda_total(x) by Horner's rule at the four points of this lane (see CHIEN_X)
                u32 acc[4];
                {
                    const u32 top = (u32)__builtin_amdgcn_readlane((int)ltk, ltlen - 1);
                    if (ltlen >= 2) {
                        u32 a[4];
#pragma unroll
                        for (int s4 = 0; s4 < 4; s4++) a[s4] = xl | ((top << 8) | CHIEN_X(s4));
                        for (int k = ltlen - 2; k >= 1; k--) {
                            const u32 lk = (u32)__builtin_amdgcn_readlane((int)ltk, k);
                            u32 T[4];
#pragma unroll
                            for (int s4 = 0; s4 < 4; s4++) T[s4] = lds_ld8(a[s4] + TBL);
#pragma unroll
                            for (int s4 = 0; s4 < 4; s4++) horner_step(a[s4], T[s4], lk);
                        }
                        const u32 l0 = (u32)__builtin_amdgcn_readlane((int)ltk, 0);
#pragma unroll
                        for (int s4 = 0; s4 < 4; s4++) acc[s4] = lds_ld8(a[s4] + TBL) ^ l0;
                    } else {
#pragma unroll
                        for (int s4 = 0; s4 < 4; s4++) acc[s4] = top;
                    }
                }
                int v_total = 0;
                bool out_of_range_root = false;
                u32 pp = posp;
                asm volatile("" : "+v"(pp)); // the position tests below are two v_cmp per word; hoisted out of the word loop they
                                             // become eight SGPR-pair masks that spill (the kernel has 72 SGPRs at 8 waves per SIMD)
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) {
                    const int i = (int)((pp >> (8 * s4)) & 0xffu);
                    const bool root = acc[s4] == 0 && i < rp.n;
                    const bool rec = root && i < n;
                    const unsigned long long mroot = __builtin_amdgcn_ballot_w64(root);
                    const unsigned long long mk = __builtin_amdgcn_ballot_w64(rec);
                    if (mroot != mk) out_of_range_root = true;
                    if (rec) {
                        const int slot = v_total + (int)__builtin_amdgcn_mbcnt_hi((u32)(mk >> 32), __builtin_amdgcn_mbcnt_lo((u32)mk, 0u));
                        if (slot < dd + 4) { ws.errpos()[slot] = (uint8_t)i; ws.errloc()[slot] = (uint8_t)(xl | CHIEN_X(s4)); }
                    }
                    v_total += __popcll(mk);
                }
                wave_sync();
                if (out_of_range_root || v_total != v + u) {
                    status = -1; // _bch.py:1469-1485
                } else {
                    // ---- 7. Omega' = Lambda * S' mod x^(d-1) ----
                    // Berlekamp-Massey guarantees sum_i Lambda_i S'_(k-i) = 0 for u + L <= k < d - 1, so the coefficients
                    // from u + L upward are exactly zero: not fed to Horner's rule below.  Lane k accumulates
                    // Lambda_i * S'_(k-i): Lambda_i is a scalar (its table row a scalar address), S' moves up one lane per term.
                    const int oplen = u + L < dd ? (u + L > 0 ? u + L : 1) : dd;
                    u32 om = lds_ld8(sp + ((u32)__builtin_amdgcn_readlane((int)Creg, 0) << 8) + TBL);
                    {
                        u32 cur = sp;
                        for (int i = 1; i < llen; i++) {
                            cur = (u32)lane_shift_up1((int)cur, lane);
                            om ^= lds_ld8(cur + ((u32)__builtin_amdgcn_readlane((int)Creg, i) << 8) + TBL);
                        }
                    }
                    // ---- 8./9./10. Forney, one located symbol per lane: numerator Omega'(x) and denominator
                    // Lambda_total'(x) by Horner's rule at x = X^-1.  Characteristic 2: the formal derivative keeps the
                    // odd-degree coefficients (_bch.py:1512-1515), i.e. it is a polynomial in x^2.
                    const int L_total = ltlen - 1;
                    const bool act = lane < v_total;
                    const u32 x = act ? (u32)ws.errloc()[lane] : 0u;
                    u32 num = (u32)__builtin_amdgcn_readlane((int)om, oplen - 1);
                    if (oplen >= 2) {
                        u32 a = x | (num << 8);
                        for (int tt = oplen - 2; tt >= 1; tt--) horner_step(a, lds_ld8(a + TBL), (u32)__builtin_amdgcn_readlane((int)om, tt));
                        num = lds_ld8(a + TBL) ^ (u32)__builtin_amdgcn_readlane((int)om, 0);
                    }
                    const u32 x2 = lds_ld8(x * 257u + TBL);
                    const int jtop = (L_total & 1) ? L_total : L_total - 1; // highest odd degree
                    u32 den = 0;
                    if (jtop >= 1) {
                        den = (u32)__builtin_amdgcn_readlane((int)ltk, jtop);
                        if (jtop >= 3) {
                            u32 a = x2 | (den << 8);
                            for (int j = jtop - 2; j >= 3; j -= 2) horner_step(a, lds_ld8(a + TBL), (u32)__builtin_amdgcn_readlane((int)ltk, j));
                            den = lds_ld8(a + TBL) ^ (u32)__builtin_amdgcn_readlane((int)ltk, 1);
                        }
                    }
                    if (act) {
                        // corrected = received - E; an erased symbol was taken as zero, so it becomes E itself
                        u32 E = 0;
                        if (num != 0 && den != 0) {
                            int ex = (int)ar.log_t[num] - (int)ar.log_t[den] + ((rp.c - 1) % qm1) * (int)ar.log_t[x];
                            ex %= qm1;
                            if (ex < 0) ex += qm1;
                            E = ar.exp_t[ex];
                        }
                        uint8_t *o = orow + (n - 1 - (int)ws.errpos()[lane]);
                        const bool erased = eras_g && eras_g[cw * n + (n - 1 - (int)ws.errpos()[lane])] != 0;
                        if (erased) *o = (uint8_t)E;
                        else if (E) *o = (uint8_t)(*o ^ E);
                    }
                    wave_sync();
                    status = 0;
                }
            }
        }
        if (lane == 0) nerr_g[cw] = status < 0 ? -1 : (status == 1 ? 0 : v);
        wave_sync();
    }
}


// ------------------------------------------------------------------------------------------------
// non-systematic codes: c(x) = m(x) g(x) (message @ G[pad:, pad:], _codes/_linear.py:281-282) and the inverse,
// m(x) = c(x) / g(x) (divmod_jit in _convert_codeword_to_message, _codes/_cyclic.py:129-138).  One codeword per wave.
// ------------------------------------------------------------------------------------------------
template <bool BIN>
__global__ __launch_bounds__(1024) void rs_polymul_kernel(RsTables t, RsParams rp, const uint8_t *__restrict__ gdesc,
                                                          const uint8_t *__restrict__ msg, int ks,
                                                          uint8_t *__restrict__ out, i64 batch)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    Arith8<BIN> ar;
    uint8_t *free_l = stage_tables<BIN>(lds_raw, t, ar, rp.qm1, blockDim.x);
    const int nk = rp.n - rp.k, ns = ks + nk;
    uint8_t *g_l = free_l; // generator polynomial, highest degree first, nk + 1 coefficients
    for (int i = threadIdx.x; i <= nk; i += blockDim.x) g_l[i] = gdesc[i];
    free_l += ((nk + 1 + 15) / 16) * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    uint8_t *m_l = free_l + (size_t)wave * 256;
    __syncthreads();
    for (i64 cw = (i64)blockIdx.x * nwaves + wave; cw < batch; cw += (i64)gridDim.x * nwaves) {
        for (int i = lane; i < ks; i += 64) m_l[i] = msg[cw * ks + i];
        wave_sync();
        // descending coefficient order on both sides: out[j] = sum_i m[i] * g[j - i]
        for (int j = lane; j < ns; j += 64) {
            const int lo = j - nk > 0 ? j - nk : 0, hi = j < ks - 1 ? j : ks - 1;
            u32 acc = 0;
            for (int i = lo; i <= hi; i++) acc = ar.add(acc, ar.mul(m_l[i], g_l[j - i]));
            out[cw * ns + j] = (uint8_t)acc;
        }
        wave_sync();
    }
}

template <bool BIN>
__global__ __launch_bounds__(1024) void rs_polydiv_kernel(RsTables t, RsParams rp, const uint8_t *__restrict__ gdesc,
                                                          const uint8_t *__restrict__ cw_g, int ns,
                                                          uint8_t *__restrict__ out, i64 batch)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    Arith8<BIN> ar;
    uint8_t *free_l = stage_tables<BIN>(lds_raw, t, ar, rp.qm1, blockDim.x);
    const int nk = rp.n - rp.k, ks = ns - nk;
    uint8_t *g_l = free_l;
    for (int i = threadIdx.x; i <= nk; i += blockDim.x) g_l[i] = gdesc[i];
    free_l += ((nk + 1 + 15) / 16) * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    uint8_t *r_l = free_l + (size_t)wave * 256;
    __syncthreads();
    for (i64 cw = (i64)blockIdx.x * nwaves + wave; cw < batch; cw += (i64)gridDim.x * nwaves) {
        for (int i = lane; i < ns; i += 64) r_l[i] = cw_g[cw * ns + i];
        wave_sync();
        // synthetic division by the monic g(x): quotient digit q_t = r[t]; r[t+1+j] -= q_t * g[1+j]
        for (int tt = 0; tt < ks; tt++) {
            const u32 q = r_l[tt];
            if (q != 0) {
                for (int j = lane; j < nk; j += 64) r_l[tt + 1 + j] = (uint8_t)ar.sub(r_l[tt + 1 + j], ar.mul(q, g_l[1 + j]));
            }
            wave_sync();
        }
        for (int i = lane; i < ks; i += 64) out[cw * ks + i] = r_l[i];
        wave_sync();
    }
}

size_t elem_bytes(int dtype) { return dtype == GFA_U8 ? 1 : dtype == GFA_U16 ? 2 : dtype == GFA_U32 ? 4 : 8; }

int rs_check_device_path(const gfa_rs *code, int dtype, const char *what)
{
    if (!code->field->has_tab8 || dtype != GFA_U8) {
        set_error(std::string(what) + ": the device path covers codes over fields of order <= 256 stored as uint8");
        return GFA_ERR_UNSUPPORTED;
    }
    return GFA_OK;
}

RsParams make_params(const gfa_rs *code)
{
    RsParams rp;
    rp.n = (int)code->n; rp.k = (int)code->k; rp.nroots = (int)code->roots.size(); rp.c = (int)code->c;
    rp.base_p = (int)code->base_p;
    rp.p = (int)code->field->calc.p; rp.qm1 = (int)(code->field->calc.q - 1);
    rp.log_alpha = (int)code->field->h_log[code->alpha];
    return rp;
}

RsTables make_tables(const FieldDeviceState &ds)
{
    RsTables t;
    t.mul8 = ds.mul8; t.add8 = ds.add8; t.neg8 = ds.neg8; t.inv8 = ds.inv8; t.exp8 = ds.exp8; t.log8 = ds.log8;
    return t;
}

// rs_decode_bin_kernel needs its dynamic LDS at address 0 (see the kernel): true exactly when the compiled kernel has no static LDS
template <typename K>
int decode_lds_base_is_zero(K kern)
{
    hipFuncAttributes fa;
    GFA_HIP(hipFuncGetAttributes(&fa, (const void *)kern));
    if (fa.sharedSizeBytes != 0) {
        set_error("Reed-Solomon wave decoder: the kernel image carries static LDS, its tables would not start at LDS address 0");
        return GFA_ERR_UNSUPPORTED;
    }
    return GFA_OK;
}

template <typename K>
int set_lds_limit(K kern, bool *done)
{
    if (!*done) {
        GFA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        *done = true;
    }
    return GFA_OK;
}

int cu_count()
{ // cached per device: the property query is far slower than a kernel launch
    static int cached[64] = {0};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return 256;
    if (!cached[d]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) return 256;
        cached[d] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cached[d];
}


template <bool DIV>
int launch_poly(gfa_rs *code, const void *in, int len_in, void *out, i64 batch, hipStream_t st)
{ // DIV = false: polymul (len_in = ks); DIV = true: polydiv (len_in = ns)
    int rc;
    FieldDeviceState *ds;
    gfa_rs::Dev *cd;
    if ((rc = code->field->ensure_device(nullptr, &ds))) return rc;
    if ((rc = code->ensure_device(nullptr, &cd))) return rc;
    const RsParams rp = make_params(code);
    const bool bin = rp.p == 2;
    const int nk = rp.n - rp.k;
    const int nwaves = bin ? 16 : 8;
    const size_t lds = (bin ? 65536 : 131072) + 1280 + ((nk + 1 + 15) / 16) * 16 + (size_t)nwaves * 256;
    const int grid = (int)std::max<i64>(1, std::min<i64>((batch + nwaves - 1) / nwaves, (i64)cu_count()));
    static bool a[2] = {false, false};
    if (bin) {
        auto k = DIV ? rs_polydiv_kernel<true> : rs_polymul_kernel<true>;
        if ((rc = set_lds_limit(k, &a[0]))) return rc;
        hipLaunchKernelGGL(k, dim3(grid), dim3(nwaves * 64), lds, st, make_tables(*ds), rp, cd->g8, (const uint8_t *)in, len_in,
                           (uint8_t *)out, batch);
    } else {
        auto k = DIV ? rs_polydiv_kernel<false> : rs_polymul_kernel<false>;
        if ((rc = set_lds_limit(k, &a[1]))) return rc;
        hipLaunchKernelGGL(k, dim3(grid), dim3(nwaves * 64), lds, st, make_tables(*ds), rp, cd->g8, (const uint8_t *)in, len_in,
                           (uint8_t *)out, batch);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}


bool lfsr_eligible(const gfa_rs *code)
{
    const int64_t nk = code->n - code->k;
    return code->field->has_tab8 && code->field->calc.p == 2 && code->systematic && nk >= 4 && nk <= 64 && (nk % 4) == 0;
}

template <bool ENCODE>
int launch_lfsr(gfa_rs *code, gfa_rs::Dev *cd, const uint8_t *in, const uint8_t *eras, int len, uint8_t *out,
                int parity_only, uint8_t *rem_out, uint8_t *flag_out, i64 batch, hipStream_t st)
{
    const int nk = (int)(code->n - code->k), nkw = nk / 4;
    {   // full-length rows of RS(255,223)-shaped codes: the register-resident form (rs_lfsr_reg_kernel)
        // measured against the staged kernel (profiles/r04_rs_lfsr_reg.txt): encode wins from 2^16 words (891 vs 861 GB/s at 2^17, 1081 vs 945 at
        // 2^20), the decoder's pre-pass only from 2^19 (782 vs 688 GB/s of clean words at 2^20; 498 vs 562 at 2^17)
        const bool shape = nkw == 8 && !eras && ((ENCODE && len == 223 && batch >= ((i64)1 << 16)) || (!ENCODE && len == 255 && rem_out && batch >= ((i64)1 << 19)));
        if (shape) {
            const int threads = batch >= (i64)1024 * cu_count() ? 1024 : 512; // one workgroup per CU (LDS); smaller ones cover every CU sooner
            const int grid = (int)std::max<i64>(1, std::min<i64>((batch + threads - 1) / threads, (i64)cu_count()));
#define GFA_LFSR_REG(LENV, MODEV)                                                                                        \
    do {                                                                                                                \
        static bool attr = false;                                                                                       \
        int rc = set_lds_limit(rs_lfsr_reg_kernel<LENV, MODEV>, &attr);                                                 \
        if (rc) return rc;                                                                                              \
        hipLaunchKernelGGL((rs_lfsr_reg_kernel<LENV, MODEV>), dim3(grid), dim3(threads), LFSR_REG_LDS, st, cd->lfsr, in, out, rem_out, flag_out, batch); \
    } while (0)
            if (ENCODE && parity_only) GFA_LFSR_REG(223, 1);
            else if (ENCODE) GFA_LFSR_REG(223, 0);
            else GFA_LFSR_REG(255, 2);
#undef GFA_LFSR_REG
            GFA_HIP(hipGetLastError());
            return GFA_OK;
        }
    }
    const int stage_bytes = ((64 * (ENCODE ? len + nk : std::max(len, nk)) + 15) / 16) * 16;
    const u32 div_magic = (u32)((((u64)1 << 32) + (u64)len - 1) / (u64)len); // ceil(2^32 / len): exact quotients below 2^16
    // four table copies and eight waves per workgroup when that fits the 160 KiB of a CU (RS(255,223): 32 KiB + 8 x 16320 B)
    static const int rep_env = [] { const char *e = getenv("GFA_RS_LFSR_REP4"); return e ? atoi(e) : 1; }();
    constexpr int kshift = 1; // which lanes share a table copy: pairs (measured against 0 and 2)
    const bool rep4 = rep_env && (nkw == 4 || nkw == 8) && (size_t)(nkw / 4) * 16384 + 8 * (size_t)stage_bytes <= 160 * 1024 &&
                      (batch >= 64 * 8 * (i64)cu_count() / 2 || rep_env == 2); // smaller batches: more, smaller workgroups (2: always)
    const int threads = rep4 ? 512 : 256;
    const size_t lds = (rep4 ? (size_t)(nkw / 4) * 16384 : (size_t)256 * nk) + (size_t)(threads / 64) * stage_bytes;
    const i64 blocks = (batch + threads - 1) / threads;
    const int grid = (int)std::max<i64>(1, std::min<i64>(blocks, (i64)cu_count() * (rep4 ? 1 : 2)));
#define GFA_LFSR(W)                                                                                                     \
    case W: {                                                                                                           \
        static bool attr = false;                                                                                       \
        int rc = set_lds_limit(rs_lfsr_kernel<W, ENCODE>, &attr);                                                       \
        if (rc) return rc;                                                                                              \
        hipLaunchKernelGGL((rs_lfsr_kernel<W, ENCODE>), dim3(grid), dim3(threads), lds, st, cd->lfsr, in, eras, len, out, \
                           parity_only, rem_out, flag_out, batch, stage_bytes, div_magic, kshift);                      \
        break;                                                                                                          \
    }
#define GFA_LFSR_REP(W)                                                                                                 \
    case W: {                                                                                                           \
        static bool attr = false;                                                                                       \
        int rc = set_lds_limit(rs_lfsr_kernel<W, ENCODE, true>, &attr);                                                 \
        if (rc) return rc;                                                                                              \
        hipLaunchKernelGGL((rs_lfsr_kernel<W, ENCODE, true>), dim3(grid), dim3(threads), lds, st, cd->lfsr, in, eras, len, out, \
                           parity_only, rem_out, flag_out, batch, stage_bytes, div_magic, kshift);                      \
        break;                                                                                                          \
    }
    if (rep4) {
        switch (nkw) {
            GFA_LFSR_REP(4) GFA_LFSR_REP(8)
        default: break;
        }
        GFA_HIP(hipGetLastError());
        return GFA_OK;
    }
    switch (nkw) {
        GFA_LFSR(1) GFA_LFSR(2) GFA_LFSR(3) GFA_LFSR(4) GFA_LFSR(5) GFA_LFSR(6) GFA_LFSR(7) GFA_LFSR(8)
        GFA_LFSR(9) GFA_LFSR(10) GFA_LFSR(11) GFA_LFSR(12) GFA_LFSR(13) GFA_LFSR(14) GFA_LFSR(15) GFA_LFSR(16)
    default: set_error("lfsr: unsupported parity length"); return GFA_ERR_UNSUPPORTED;
    }
#undef GFA_LFSR
#undef GFA_LFSR_REP
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

} // namespace

int gfa_rs::ensure_device(int *device_out, Dev **out)
{
    int d = 0;
    GFA_HIP(hipGetDevice(&d));
    std::lock_guard<std::mutex> lock(mu);
    if ((size_t)d >= dev.size()) dev.resize(d + 1);
    Dev &st = dev[d];
    if (!st.ready && !field->has_tab8) {
        // wide codes: 32-bit copies of P, the roots and g(x) for gfa_rs_wide.hip
        auto upload = [](const std::vector<uint64_t> &src, uint32_t **dst) -> int {
            std::vector<uint32_t> w(src.size());
            for (size_t i = 0; i < w.size(); i++) w[i] = (uint32_t)src[i];
            GFA_HIP(hipMalloc((void **)dst, std::max<size_t>(w.size() * sizeof(uint32_t), 16)));
            if (!w.empty()) GFA_HIP(hipMemcpy(*dst, w.data(), w.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            return GFA_OK;
        };
        int rc;
        if ((rc = upload(P, &st.Pw)) || (rc = upload(roots, &st.rootsw)) || (rc = upload(gpoly, &st.gw))) return rc;
        st.ready = true;
    }
    if (!st.ready) {
        const size_t nk = (size_t)(n - k);
        std::vector<uint8_t> P8((size_t)k * nk), r8(roots.size());
        for (size_t i = 0; i < P8.size(); i++) P8[i] = (uint8_t)P[i];
        for (size_t i = 0; i < r8.size(); i++) r8[i] = (uint8_t)roots[i];
        GFA_HIP(hipMalloc((void **)&st.P8, std::max<size_t>(P8.size(), 16)));
        GFA_HIP(hipMalloc((void **)&st.roots8, std::max<size_t>(r8.size(), 16)));
        if (!P8.empty()) GFA_HIP(hipMemcpy(st.P8, P8.data(), P8.size(), hipMemcpyHostToDevice));
        if (!r8.empty()) GFA_HIP(hipMemcpy(st.roots8, r8.data(), r8.size(), hipMemcpyHostToDevice));
        {
            std::vector<uint8_t> g8(gpoly.size());
            for (size_t i = 0; i < g8.size(); i++) g8[i] = (uint8_t)gpoly[i];
            GFA_HIP(hipMalloc((void **)&st.g8, std::max<size_t>(g8.size(), 16)));
            GFA_HIP(hipMemcpy(st.g8, g8.data(), g8.size(), hipMemcpyHostToDevice));
        }
        if (field->has_tab8 && field->calc.p == 2 && nk >= 4 && nk <= 64 && nk % 4 == 0) {
            // tables built by gfa_rs_host.h (checked on the host by tests/csrc/rs_host_test.cpp): the LFSR rows in the order
            // rs_lfsr_kernel reads them, and the lane tables of rs_decode_bin_kernel
            const std::vector<uint32_t> rows = rs_lfsr_rows(field->h_mul8.data(), field->calc.q, gpoly, nk);
            GFA_HIP(hipMalloc((void **)&st.lfsr, rows.size() * sizeof(uint32_t)));
            GFA_HIP(hipMemcpy(st.lfsr, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            const std::vector<uint8_t> aux = rs_decode_lane_tables(field->h_mul8.data(), field->calc.q, alpha, n, roots);
            GFA_HIP(hipMalloc((void **)&st.aux8, aux.size()));
            GFA_HIP(hipMemcpy(st.aux8, aux.data(), aux.size(), hipMemcpyHostToDevice));
        }
        st.ready = true;
    }
    if (device_out) *device_out = d;
    *out = &st;
    return GFA_OK;
}

extern "C" {

// systematic parity matrix from g(x): row i = -(x^(n-1-i) mod g(x)), built downward from row 0 by dividing by x
// (same matrix as _poly_to_generator_matrix, _cyclic.py:198-226)
static void build_parity_matrix(gfa_rs *code)
{
    const FieldDev &d = code->field->calc;
    const int64_t n = code->n, k = code->k, nk = n - k;
    code->P.assign((size_t)k * nk, 0);
    if (nk <= 0) return;
    u64 g0inv;
    HostArith::inv(d, code->gpoly[nk], &g0inv);
    for (int64_t j = 0; j < nk; j++) code->P[j] = HostArith::mul(d, code->gpoly[j], g0inv);
    for (int64_t i = 1; i < k; i++) {
        u64 *row = &code->P[(size_t)i * nk];
        const u64 *prev = &code->P[(size_t)(i - 1) * nk];
        const u64 last = prev[nk - 1];
        row[0] = 0;
        for (int64_t j = 1; j < nk; j++) row[j] = prev[j - 1];
        if (last)
            for (int64_t j = 0; j < nk; j++) row[j] = HostArith::sub(d, row[j], HostArith::mul(d, last, code->P[j]));
    }
}

int gfa_rs_create(gfa_field_t *f, int64_t n, int64_t k, int64_t c, uint64_t alpha, int systematic, gfa_rs_t **out)
{
    if (!f || !out || n < 1 || k < 1 || k > n || c < 0) { set_error("gfa_rs_create: bad arguments"); return GFA_ERR_INVALID; }
    const FieldDev &d = f->calc;
    if ((u64)n >= d.q || (d.q - 1) % (u64)n != 0) { set_error("gfa_rs_create: n must divide q - 1"); return GFA_ERR_INVALID; }
    if (alpha == 0 || alpha >= d.q) { set_error("gfa_rs_create: alpha out of range"); return GFA_ERR_INVALID; }
    gfa_rs *code = new gfa_rs();
    code->field = f; code->n = n; code->k = k; code->c = c; code->alpha = alpha; code->systematic = systematic != 0;
    const int64_t nk = n - k;
    // roots alpha^(c+i) and g(x) = prod (x - root_i)  (_reed_solomon.py:206-207)
    code->roots.resize(nk);
    std::vector<u64> g(1, 1); // ascending
    for (int64_t i = 0; i < nk; i++) {
        u64 r;
        HostArith::pow(d, alpha, (i64)(c + i), &r);
        code->roots[i] = r;
        std::vector<u64> ng(g.size() + 1, 0);
        const u64 nr = HostArith::neg(d, r);
        for (size_t j = 0; j < g.size(); j++) {
            ng[j + 1] = HostArith::add(d, ng[j + 1], g[j]);
            ng[j] = HostArith::add(d, ng[j], HostArith::mul(d, g[j], nr));
        }
        g.swap(ng);
    }
    code->gpoly.assign(g.rbegin(), g.rend()); // highest degree first
    build_parity_matrix(code);
    *out = code;
    return GFA_OK;
}

int gfa_bch_create(gfa_field_t *ext, uint64_t base_p, int64_t n, int64_t k, int64_t d_design, int64_t c, uint64_t alpha,
                   const uint64_t *generator_poly, int systematic, gfa_rs_t **out)
{
    if (!ext || !out || !generator_poly || n < 1 || k < 1 || k > n || c < 0 || d_design < 1) {
        set_error("gfa_bch_create: bad arguments");
        return GFA_ERR_INVALID;
    }
    const FieldDev &d = ext->calc;
    if (base_p != d.p) { set_error("gfa_bch_create: the symbol field must be the prime subfield of the extension field"); return GFA_ERR_INVALID; }
    if ((u64)n >= d.q || (d.q - 1) % (u64)n != 0) { set_error("gfa_bch_create: n must divide q^m - 1"); return GFA_ERR_INVALID; }
    if (alpha == 0 || alpha >= d.q) { set_error("gfa_bch_create: alpha out of range"); return GFA_ERR_INVALID; }
    const int64_t nk = n - k;
    if (generator_poly[0] != 1) { set_error("gfa_bch_create: the generator polynomial must be monic of degree n - k"); return GFA_ERR_INVALID; }
    for (int64_t i = 0; i <= nk; i++)
        if (generator_poly[i] >= base_p) { set_error("gfa_bch_create: generator polynomial coefficient outside GF(p)"); return GFA_ERR_INVALID; }
    gfa_rs *code = new gfa_rs();
    code->field = ext; code->n = n; code->k = k; code->c = c; code->alpha = alpha; code->systematic = systematic != 0;
    code->base_p = (int64_t)base_p;
    // roots alpha^c .. alpha^(c+d-2) (_bch.py:1178-1197); every one must be a root of g(x)
    code->roots.resize(d_design - 1);
    for (int64_t i = 0; i < d_design - 1; i++) {
        u64 r;
        HostArith::pow(d, alpha, (i64)(c + i), &r);
        code->roots[i] = r;
        u64 acc = 0;
        for (int64_t j = 0; j <= nk; j++) acc = HostArith::add(d, HostArith::mul(d, acc, r), generator_poly[j]);
        if (acc != 0) { delete code; set_error("gfa_bch_create: alpha^(c+i) is not a root of the generator polynomial"); return GFA_ERR_INVALID; }
    }
    code->gpoly.assign(generator_poly, generator_poly + nk + 1);
    build_parity_matrix(code);
    *out = code;
    return GFA_OK;
}

void gfa_rs_destroy(gfa_rs_t *code)
{
    if (!code) return;
    for (auto &st : code->dev)
        if (st.ready) { (void)hipFree(st.P8); (void)hipFree(st.roots8); (void)hipFree(st.lfsr); (void)hipFree(st.aux8); (void)hipFree(st.g8); (void)hipFree(st.Pw); (void)hipFree(st.rootsw); (void)hipFree(st.gw); }
    delete code;
}

int gfa_rs_describe(const gfa_rs_t *code, uint64_t *roots, uint64_t *generator_poly, uint64_t *parity_matrix)
{
    if (!code) return GFA_ERR_INVALID;
    if (roots) std::copy(code->roots.begin(), code->roots.end(), roots); // d - 1 entries
    if (generator_poly) std::copy(code->gpoly.begin(), code->gpoly.end(), generator_poly);
    if (parity_matrix) std::copy(code->P.begin(), code->P.end(), parity_matrix);
    return GFA_OK;
}

int gfa_rs_encode(gfa_rs_t *code, const void *msg, int64_t ks, void *out, int64_t batch, int parity_only, int dtype,
                  gfa_stream_t stream)
{
    if (!code || batch < 0 || ks < 1 || ks > code->k) { set_error("gfa_rs_encode: bad arguments"); return GFA_ERR_INVALID; }
    if (batch == 0) return GFA_OK; // empty batches carry no buffers
    if (!msg || !out) { set_error("gfa_rs_encode: bad arguments"); return GFA_ERR_INVALID; }
    if (rs_wide_code(code)) {
        int rcw = rs_wide_check(code, dtype, "gfa_rs_encode");
        if (rcw) return rcw;
        if (!code->systematic && parity_only) { set_error("gfa_rs_encode: parity output exists only for systematic codes"); return GFA_ERR_INVALID; }
        if (batch == 0) return GFA_OK;
        if (code->n == code->k) {
            if (!parity_only) GFA_HIP(hipMemcpyAsync(out, msg, elem_bytes(dtype) * (size_t)(batch * ks), hipMemcpyDeviceToDevice, (hipStream_t)stream));
            return GFA_OK;
        }
        return rs_wide_encode(code, msg, ks, out, batch, parity_only, dtype, (hipStream_t)stream);
    }
    int rc = rs_check_device_path(code, dtype, "gfa_rs_encode");
    if (rc) return rc;
    if (batch == 0) return GFA_OK;
    FieldDeviceState *ds;
    gfa_rs::Dev *cd;
    if ((rc = code->field->ensure_device(nullptr, &ds))) return rc;
    if ((rc = code->ensure_device(nullptr, &cd))) return rc;
    if (!code->systematic) {
        if (parity_only) { set_error("gfa_rs_encode: parity output exists only for systematic codes"); return GFA_ERR_INVALID; }
        if (code->n == code->k) {
            GFA_HIP(hipMemcpyAsync(out, msg, (size_t)(batch * ks), hipMemcpyDeviceToDevice, (hipStream_t)stream));
            return GFA_OK;
        }
        return launch_poly<false>(code, msg, (int)ks, out, batch, (hipStream_t)stream);
    }
    const RsParams rp = make_params(code);
    const bool bin = rp.p == 2;
    const int nk = rp.n - rp.k;
    if (nk == 0) { // identity code: codeword == message
        if (!parity_only) GFA_HIP(hipMemcpyAsync(out, msg, (size_t)(batch * ks), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return GFA_OK;
    }
    if (lfsr_eligible(code) && cd->lfsr && ks >= 2)
        return launch_lfsr<true>(code, cd, (const uint8_t *)msg, nullptr, (int)ks, (uint8_t *)out, parity_only, nullptr, nullptr,
                                 batch, (hipStream_t)stream);
    int groups = 1;
    while (groups < 64 && (64 / (groups * 2)) >= nk) groups *= 2;
    const size_t mpitch = (((size_t)ks + 15) / 16) * 16;
    const size_t fixed = (bin ? 65536 : 131072) + 1280 + (((size_t)ks * nk + 15) / 16) * 16;
    int nwaves = 16;
    while (nwaves > 1 && fixed + (size_t)nwaves * groups * mpitch > 160 * 1024) nwaves /= 2;
    const size_t lds = fixed + (size_t)nwaves * groups * mpitch;
    if (lds > 160 * 1024) { set_error("gfa_rs_encode: code too large for the LDS-resident encoder"); return GFA_ERR_UNSUPPORTED; }
    const int threads = nwaves * 64;
    const i64 cw_per_block = (i64)nwaves * groups;
    const int grid = (int)std::min<i64>((batch + cw_per_block - 1) / cw_per_block, (i64)cu_count());
    static bool a0 = false, a1 = false;
    if (bin) {
        if ((rc = set_lds_limit(rs_encode_kernel<true>, &a0))) return rc;
        hipLaunchKernelGGL(rs_encode_kernel<true>, dim3(grid), dim3(threads), lds, (hipStream_t)stream, make_tables(*ds), rp,
                           cd->P8, (const uint8_t *)msg, (int)ks, (uint8_t *)out, batch, parity_only, groups);
    } else {
        if ((rc = set_lds_limit(rs_encode_kernel<false>, &a1))) return rc;
        hipLaunchKernelGGL(rs_encode_kernel<false>, dim3(grid), dim3(threads), lds, (hipStream_t)stream, make_tables(*ds), rp,
                           cd->P8, (const uint8_t *)msg, (int)ks, (uint8_t *)out, batch, parity_only, groups);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

static int launch_decode(gfa_rs_t *code, const void *recv, const uint8_t *erasures, int64_t ns, void *out_codeword,
                         int64_t *out_n_errors, uint8_t *detected, int64_t batch, bool detect_only, hipStream_t st)
{
    int rc;
    FieldDeviceState *ds;
    gfa_rs::Dev *cd;
    if ((rc = code->field->ensure_device(nullptr, &ds))) return rc;
    if ((rc = code->ensure_device(nullptr, &cd))) return rc;
    const RsParams rp = make_params(code);
    const bool bin = rp.p == 2;
    const int dd = rp.nroots;
    const size_t fixed = (bin ? 65536 : 131072) + 1280 + ((dd + 15) / 16) * 16;
    const size_t per_wave = (size_t)WaveScratch::bytes((int)ns, dd);
    int nwaves = 16;
    while (nwaves > 1 && fixed + nwaves * per_wave > 160 * 1024) nwaves /= 2;
    if (fixed + nwaves * per_wave > 160 * 1024) { set_error("gfa_rs_decode: code too large for LDS"); return GFA_ERR_UNSUPPORTED; }
    const size_t lds = fixed + ((nwaves * per_wave + 15) & ~(size_t)15) + 16;
    const int threads = nwaves * 64;
    const int grid = (int)std::min<i64>((batch + nwaves - 1) / nwaves, (i64)cu_count());
    static bool a[4] = {false, false, false, false};
#define GFA_RS_LAUNCH(BINV, DET, IDX)                                                                                  \
    do {                                                                                                               \
        if ((rc = set_lds_limit(rs_decode_kernel<BINV, DET>, &a[IDX]))) return rc;                                     \
        hipLaunchKernelGGL((rs_decode_kernel<BINV, DET>), dim3(grid), dim3(threads), lds, st, make_tables(*ds), rp,    \
                           cd->roots8, (const uint8_t *)recv, erasures, (int)ns, (uint8_t *)out_codeword,              \
                           (i64 *)out_n_errors, detected, batch);                                                      \
    } while (0)
    if (bin && detect_only) GFA_RS_LAUNCH(true, true, 0);
    else if (bin) GFA_RS_LAUNCH(true, false, 1);
    else if (detect_only) GFA_RS_LAUNCH(false, true, 2);
    else GFA_RS_LAUNCH(false, false, 3);
#undef GFA_RS_LAUNCH
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int gfa_rs_detect(gfa_rs_t *code, const void *cw, int64_t ns, uint8_t *detected, int64_t batch, int dtype,
                  gfa_stream_t stream)
{
    if (!code || batch < 0 || ns < code->n - code->k + 1 || ns > code->n) {
        set_error("gfa_rs_detect: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (batch == 0) return GFA_OK;
    if (!cw || !detected) { set_error("gfa_rs_detect: bad arguments"); return GFA_ERR_INVALID; }
    int rc = rs_wide_code(code) ? rs_wide_check(code, dtype, "gfa_rs_detect") : rs_check_device_path(code, dtype, "gfa_rs_detect");
    if (rc) return rc;
    if (batch == 0) return GFA_OK;
    if (code->n == code->k) {
        GFA_HIP(hipMemsetAsync(detected, 0, (size_t)batch, (hipStream_t)stream));
        return GFA_OK;
    }
    if (rs_wide_code(code)) return rs_wide_decode(code, cw, nullptr, ns, nullptr, nullptr, detected, batch, true, dtype, (hipStream_t)stream);
    if (lfsr_eligible(code)) {
        FieldDeviceState *ds;
        gfa_rs::Dev *cd;
        if ((rc = code->field->ensure_device(nullptr, &ds))) return rc;
        if ((rc = code->ensure_device(nullptr, &cd))) return rc;
        if (cd->lfsr)
            return launch_lfsr<false>(code, cd, (const uint8_t *)cw, nullptr, (int)ns, nullptr, 0, nullptr, detected, batch,
                                      (hipStream_t)stream);
    }
    return launch_decode(code, cw, nullptr, ns, nullptr, nullptr, detected, batch, true, (hipStream_t)stream);
}

int gfa_rs_decode(gfa_rs_t *code, const void *recv, const uint8_t *erasures, int64_t ns, void *out_codeword,
                  int64_t *out_n_errors, int64_t batch, int dtype, gfa_stream_t stream)
{
    if (!code || batch < 0 || ns < code->n - code->k + 1 || ns > code->n) {
        set_error("gfa_rs_decode: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (batch == 0) return GFA_OK;
    if (!recv || !out_codeword || !out_n_errors) { set_error("gfa_rs_decode: bad arguments"); return GFA_ERR_INVALID; }
    int rc = rs_wide_code(code) ? rs_wide_check(code, dtype, "gfa_rs_decode") : rs_check_device_path(code, dtype, "gfa_rs_decode");
    if (rc) return rc;
    if (batch == 0) return GFA_OK;
    if (code->n == code->k && !erasures) { // identity code: nothing to correct (with erasures every erased word fails, below)
        if (out_codeword != recv)
            GFA_HIP(hipMemcpyAsync(out_codeword, recv, elem_bytes(dtype) * (size_t)(batch * ns), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        GFA_HIP(hipMemsetAsync(out_n_errors, 0, sizeof(int64_t) * (size_t)batch, (hipStream_t)stream));
        return GFA_OK;
    }
    if (rs_wide_code(code))
        return rs_wide_decode(code, recv, erasures, ns, out_codeword, (i64 *)out_n_errors, nullptr, batch, false, dtype, (hipStream_t)stream);
    if (lfsr_eligible(code) && code->n - code->k <= 60 && (code->base_p == 0 || code->base_p == 2) && code->roots.size() >= 1) {
        FieldDeviceState *ds;
        gfa_rs::Dev *cd;
        if ((rc = code->field->ensure_device(nullptr, &ds))) return rc;
        if ((rc = code->ensure_device(nullptr, &cd))) return rc;
        if (cd->lfsr) {
            hipStream_t st = (hipStream_t)stream;
            const int nk = (int)(code->n - code->k);
            const size_t need = (size_t)batch * nk;
            // remainder scratch of THIS call, stream-ordered (two decodes of one code on different streams, or from
            // different host threads, must not share it; no device synchronisation, so the call stays graph-capturable)
            uint8_t *rem = nullptr;
            GFA_HIP(gfa::scratch_alloc((void **)&rem, need, st));
            if ((rc = launch_lfsr<false>(code, cd, (const uint8_t *)recv, erasures, (int)ns, (uint8_t *)out_codeword, 0, rem, nullptr,
                                         batch, st))) {
                (void)gfa::scratch_free(rem, st);
                return rc;
            }
            const RsParams rp = make_params(code);
            const size_t fixed = 65536 + 1280;
            const bool small = (int)code->roots.size() + 4 <= 40;
            const size_t per_wave = small ? WaveScratch2<40>::BYTES : WaveScratch2<64>::BYTES;
            static const int wps = [] { // initialised once (thread-safe), GFA_RS_WPS: tuning override
                const char *e = getenv("GFA_RS_WPS");
                const int w = e ? atoi(e) : 8;
                return (w == 4 || w == 5 || w == 6) ? w : 8;
            }();
            const int nwaves = 2 * wps;
            const size_t lds = fixed + ((nwaves * per_wave + 15) & ~(size_t)15) + 16;
            const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
            const int grid = (int)std::max<i64>(1, std::min<i64>((batch + nwaves - 1) / nwaves, (i64)cu_count() * per_cu));
            RsParams rpk = rp;
            rpk.per_block = (int)((batch + grid - 1) / grid);
#define GFA_K2(SV, W, IDX)                                                                                              \
    do {                                                                                                                \
        static bool attr = false;                                                                                       \
        if (!attr && (rc = decode_lds_base_is_zero(rs_decode_bin_kernel<SV, W>))) {                                     \
            (void)gfa::scratch_free(rem, st);                                                                           \
            return rc;                                                                                                  \
        }                                                                                                               \
        if ((rc = set_lds_limit(rs_decode_bin_kernel<SV, W>, &attr))) return rc;                                        \
        hipLaunchKernelGGL((rs_decode_bin_kernel<SV, W>), dim3(grid), dim3(nwaves * 64), lds, st, make_tables(*ds), rpk, \
                           erasures, rem, cd->aux8, (int)ns, (uint8_t *)out_codeword, (i64 *)out_n_errors, batch);      \
    } while (0)
            if (small) {
                switch (wps) {
                case 4: GFA_K2(40, 4, 0); break;
                case 5: GFA_K2(40, 5, 1); break;
                case 6: GFA_K2(40, 6, 2); break;
                default: GFA_K2(40, 8, 3); break;
                }
            } else {
                switch (wps) {
                case 4: GFA_K2(64, 4, 4); break;
                case 5: GFA_K2(64, 5, 5); break;
                case 6: GFA_K2(64, 6, 6); break;
                default: GFA_K2(64, 8, 7); break;
                }
            }
#undef GFA_K2
            const hipError_t launch_err = hipGetLastError();
            GFA_HIP(gfa::scratch_free(rem, st));
            GFA_HIP(launch_err);
            return GFA_OK;
        }
    }
    return launch_decode(code, recv, erasures, ns, out_codeword, out_n_errors, nullptr, batch, false, (hipStream_t)stream);
}

int gfa_rs_extract_message(gfa_rs_t *code, const void *cw, int64_t ns, void *out_msg, int64_t batch, int dtype,
                           gfa_stream_t stream)
{
    if (!code || batch < 0 || ns < code->n - code->k + 1 || ns > code->n) {
        set_error("gfa_rs_extract_message: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (batch == 0) return GFA_OK;
    if (!cw || !out_msg) { set_error("gfa_rs_extract_message: bad arguments"); return GFA_ERR_INVALID; }
    int rc = rs_wide_code(code) ? rs_wide_check(code, dtype, "gfa_rs_extract_message") : rs_check_device_path(code, dtype, "gfa_rs_extract_message");
    if (rc) return rc;
    if (batch == 0) return GFA_OK;
    const int64_t ks = code->k - (code->n - ns);
    if (code->systematic || code->n == code->k) {
        const size_t eb = elem_bytes(dtype);
        GFA_HIP(hipMemcpy2DAsync(out_msg, eb * (size_t)ks, cw, eb * (size_t)ns, eb * (size_t)ks, (size_t)batch, hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
        return GFA_OK;
    }
    if (rs_wide_code(code)) return rs_wide_polydiv(code, cw, ns, out_msg, batch, dtype, (hipStream_t)stream);
    return launch_poly<true>(code, cw, (int)ns, out_msg, batch, (hipStream_t)stream);
}

int gfa_time_rs_encode(gfa_rs_t *code, const void *msg, int64_t ks, void *out, int64_t batch, int dtype,
                       gfa_stream_t stream, int iters, float *ms_out)
{
    return gfa::time_loop((hipStream_t)stream, iters, ms_out,
                          [&]() { return gfa_rs_encode(code, msg, ks, out, batch, 0, dtype, stream); });
}

int gfa_time_rs_decode(gfa_rs_t *code, const void *recv, int64_t ns, void *out_codeword, int64_t *out_n_errors,
                       int64_t batch, int dtype, gfa_stream_t stream, int iters, float *ms_out)
{
    return gfa::time_loop((hipStream_t)stream, iters, ms_out, [&]() {
        return gfa_rs_decode(code, recv, nullptr, ns, out_codeword, out_n_errors, batch, dtype, stream);
    });
}

// Test-only (tests/test_gpu_rs.py): runs rs_bm_selftest_kernel on `nseq` sequences drawn from `seed` -- a third random bytes of
// random length 1..32, a third outputs of random LFSRs of length <= 16 (what a correctable word produces: the run ends early on
// zero discrepancies), a third with leading / embedded zeros -- and returns the number of sequences on which the hand-assembled
// loop and the compiler-generated one disagree.
int gfa_debug_rs_bm_selftest(gfa_field_t *f, int64_t nseq, uint64_t seed, int64_t *mismatches, gfa_stream_t stream)
{
    if (!f || !mismatches || nseq < 1 || !f->has_tab8 || f->calc.p != 2 || gfa_field_order(f) != 256) {
        set_error("gfa_debug_rs_bm_selftest: a GF(2^8) field, nseq >= 1");
        return GFA_ERR_INVALID;
    }
    int rc, dev = 0;
    FieldDeviceState *ds = nullptr;
    if ((rc = f->ensure_device(&dev, &ds))) return rc;
    hipStream_t st = (hipStream_t)stream;
    std::vector<uint8_t> seqs((size_t)nseq * 32), mul(65536);
    std::vector<int> lens((size_t)nseq);
    for (u32 a = 0; a < 256; a++)
        for (u32 b = 0; b < 256; b++) mul[(a << 8) | b] = (uint8_t)HostArith::mul(f->calc, a, b);
    u64 sd = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (u32)(sd >> 24); };
    for (i64 q = 0; q < nseq; q++) {
        uint8_t *S = &seqs[(size_t)q * 32];
        const int nsq = (q % 4 == 3) ? 32 : 1 + (int)(rnd() % 32);
        lens[(size_t)q] = nsq;
        const int cls = (int)(q % 3);
        if (cls == 0) {
            for (int i = 0; i < 32; i++) S[i] = (uint8_t)rnd();
        } else if (cls == 1) { // S_i = sum_j c_j S_{i-j}: an LFSR of length len <= 16
            const int len = 1 + (int)(rnd() % 16);
            uint8_t c[16];
            for (int j = 0; j < len; j++) c[j] = (uint8_t)rnd();
            for (int i = 0; i < 32; i++) {
                if (i < len) { S[i] = (uint8_t)rnd(); continue; }
                u32 acc = 0;
                for (int j = 0; j < len; j++) acc ^= mul[((u32)c[j] << 8) | S[i - 1 - j]];
                S[i] = (uint8_t)acc;
            }
        } else {
            const int z = (int)(rnd() % 33);
            for (int i = 0; i < 32; i++) S[i] = (i < z || rnd() % 4 == 0) ? 0 : (uint8_t)rnd();
        }
    }
    uint8_t *d_seq = nullptr;
    int *d_len = nullptr, *d_out = nullptr;
    GFA_HIP(gfa::scratch_alloc((void **)&d_seq, seqs.size(), st));
    GFA_HIP(gfa::scratch_alloc((void **)&d_len, lens.size() * sizeof(int), st));
    GFA_HIP(gfa::scratch_alloc((void **)&d_out, lens.size() * sizeof(int), st));
    GFA_HIP(hipMemcpyAsync(d_seq, seqs.data(), seqs.size(), hipMemcpyHostToDevice, st));
    GFA_HIP(hipMemcpyAsync(d_len, lens.data(), lens.size() * sizeof(int), hipMemcpyHostToDevice, st));
    static bool attr = false;
    if ((rc = decode_lds_base_is_zero(rs_bm_selftest_kernel))) return rc;
    if ((rc = set_lds_limit(rs_bm_selftest_kernel, &attr))) return rc;
    const int grid = (int)std::max<i64>(1, std::min<i64>((nseq + 3) / 4, 4 * (i64)cu_count()));
    hipLaunchKernelGGL(rs_bm_selftest_kernel, dim3(grid), dim3(256), 65536 + 16, st, ds->mul8, d_seq, d_len, d_out, (i64)nseq);
    GFA_HIP(hipGetLastError());
    std::vector<int> out((size_t)nseq);
    GFA_HIP(hipMemcpyAsync(out.data(), d_out, out.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    GFA_HIP(hipStreamSynchronize(st));
    (void)gfa::scratch_free(d_seq, st); (void)gfa::scratch_free(d_len, st); (void)gfa::scratch_free(d_out, st);
    i64 bad = 0;
    for (int v : out) bad += v != 0;
    *mismatches = bad;
    return GFA_OK;
}

} // extern "C"
