// gfa_rs.hip -- Reed-Solomon encode / detect / decode on gfx950 for codes over fields of order <= 256 (uint8).
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
#include "gfa_internal.h"

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
                            ws.recv[pos] = (uint8_t)ar.sub(ws.recv[pos], E);
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

int rs_check_device_path(const gfa_rs *code, int dtype, const char *what)
{
    if (!code->field->has_tab8 || dtype != GFA_U8) {
        set_error(std::string(what) + ": the device path covers codes over fields of order <= 256 stored as uint8");
        return GFA_ERR_UNSUPPORTED;
    }
    if (!code->systematic) {
        set_error(std::string(what) + ": non-systematic codes have no device path");
        return GFA_ERR_UNSUPPORTED;
    }
    return GFA_OK;
}

RsParams make_params(const gfa_rs *code)
{
    RsParams rp;
    rp.n = (int)code->n; rp.k = (int)code->k; rp.nroots = (int)(code->n - code->k); rp.c = (int)code->c;
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
{
    int d = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&d) != hipSuccess || hipGetDeviceProperties(&prop, d) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

} // namespace

int gfa_rs::ensure_device(int *device_out, Dev **out)
{
    int d = 0;
    GFA_HIP(hipGetDevice(&d));
    std::lock_guard<std::mutex> lock(mu);
    if ((size_t)d >= dev.size()) dev.resize(d + 1);
    Dev &st = dev[d];
    if (!st.ready) {
        const size_t nk = (size_t)(n - k);
        std::vector<uint8_t> P8((size_t)k * nk), r8(nk);
        for (size_t i = 0; i < P8.size(); i++) P8[i] = (uint8_t)P[i];
        for (size_t i = 0; i < nk; i++) r8[i] = (uint8_t)roots[i];
        GFA_HIP(hipMalloc((void **)&st.P8, std::max<size_t>(P8.size(), 16)));
        GFA_HIP(hipMalloc((void **)&st.roots8, std::max<size_t>(r8.size(), 16)));
        if (!P8.empty()) GFA_HIP(hipMemcpy(st.P8, P8.data(), P8.size(), hipMemcpyHostToDevice));
        if (!r8.empty()) GFA_HIP(hipMemcpy(st.roots8, r8.data(), r8.size(), hipMemcpyHostToDevice));
        st.ready = true;
    }
    if (device_out) *device_out = d;
    *out = &st;
    return GFA_OK;
}

extern "C" {

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
    // systematic parity matrix: row i = -(x^(n-1-i) mod g(x)), built downward from row 0 by dividing by x
    // (same matrix as _poly_to_generator_matrix, _cyclic.py:198-226)
    code->P.assign((size_t)k * nk, 0);
    if (nk > 0) {
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
    *out = code;
    return GFA_OK;
}

void gfa_rs_destroy(gfa_rs_t *code)
{
    if (!code) return;
    for (auto &st : code->dev)
        if (st.ready) { (void)hipFree(st.P8); (void)hipFree(st.roots8); (void)hipFree(st.lfsr); }
    delete code;
}

int gfa_rs_describe(const gfa_rs_t *code, uint64_t *roots, uint64_t *generator_poly, uint64_t *parity_matrix)
{
    if (!code) return GFA_ERR_INVALID;
    if (roots) std::copy(code->roots.begin(), code->roots.end(), roots);
    if (generator_poly) std::copy(code->gpoly.begin(), code->gpoly.end(), generator_poly);
    if (parity_matrix) std::copy(code->P.begin(), code->P.end(), parity_matrix);
    return GFA_OK;
}

int gfa_rs_encode(gfa_rs_t *code, const void *msg, int64_t ks, void *out, int64_t batch, int parity_only, int dtype,
                  gfa_stream_t stream)
{
    if (!code || !msg || !out || batch < 0 || ks < 1 || ks > code->k) { set_error("gfa_rs_encode: bad arguments"); return GFA_ERR_INVALID; }
    int rc = rs_check_device_path(code, dtype, "gfa_rs_encode");
    if (rc) return rc;
    if (batch == 0) return GFA_OK;
    FieldDeviceState *ds;
    gfa_rs::Dev *cd;
    if ((rc = code->field->ensure_device(nullptr, &ds))) return rc;
    if ((rc = code->ensure_device(nullptr, &cd))) return rc;
    const RsParams rp = make_params(code);
    const bool bin = rp.p == 2;
    const int nk = rp.n - rp.k;
    if (nk == 0) { // identity code: codeword == message
        if (!parity_only) GFA_HIP(hipMemcpyAsync(out, msg, (size_t)(batch * ks), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return GFA_OK;
    }
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
    const size_t lds = fixed + nwaves * per_wave;
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
    if (!code || !cw || !detected || batch < 0 || ns < code->n - code->k + 1 || ns > code->n) {
        set_error("gfa_rs_detect: bad arguments");
        return GFA_ERR_INVALID;
    }
    int rc = rs_check_device_path(code, dtype, "gfa_rs_detect");
    if (rc) return rc;
    if (batch == 0) return GFA_OK;
    if (code->n == code->k) {
        GFA_HIP(hipMemsetAsync(detected, 0, (size_t)batch, (hipStream_t)stream));
        return GFA_OK;
    }
    return launch_decode(code, cw, nullptr, ns, nullptr, nullptr, detected, batch, true, (hipStream_t)stream);
}

int gfa_rs_decode(gfa_rs_t *code, const void *recv, const uint8_t *erasures, int64_t ns, void *out_codeword,
                  int64_t *out_n_errors, int64_t batch, int dtype, gfa_stream_t stream)
{
    if (!code || !recv || !out_codeword || !out_n_errors || batch < 0 || ns < code->n - code->k + 1 || ns > code->n) {
        set_error("gfa_rs_decode: bad arguments");
        return GFA_ERR_INVALID;
    }
    int rc = rs_check_device_path(code, dtype, "gfa_rs_decode");
    if (rc) return rc;
    if (batch == 0) return GFA_OK;
    if (code->n == code->k) {
        GFA_HIP(hipMemcpyAsync(out_codeword, recv, (size_t)(batch * ns), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        GFA_HIP(hipMemsetAsync(out_n_errors, 0, sizeof(int64_t) * (size_t)batch, (hipStream_t)stream));
        return GFA_OK;
    }
    return launch_decode(code, recv, erasures, ns, out_codeword, out_n_errors, nullptr, batch, false, (hipStream_t)stream);
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

} // extern "C"
