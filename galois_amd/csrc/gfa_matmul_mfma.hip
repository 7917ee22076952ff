// gfa_matmul_mfma.hip -- matrix products over small prime fields GF(p), p <= 256, on the matrix cores.
//
// The one place in this engine where the work IS a GEMM (SURVEY.md section 8(f) item 2 asks for exactly this evaluation;
// the reference itself multiplies prime-field matrices with an integer BLAS call and reduces afterwards,
// _domains/_linalg.py:21-75).  Elements are moved to the centred residue system a' = a - p*(a > p/2) in [-127, 127], so a
// product fits int8 x int8 and K <= 131072 products fit the int32 accumulator of v_mfma_i32_32x32x32_i8 exactly; one
// integer reduction mod p per output element recovers the field element.  Exact arithmetic throughout: bit-identical to the
// scalar kernels.
//
//   1. centre_kernel: A (M x K, any storage width) -> int8 A' (Mp x Kp, zero padded), B (K x N) -> int8 Bt' (Np x Kp), i.e.
//      transposed so that both operands are K-contiguous, which is what the MFMA operand layout wants (16 consecutive k
//      per lane).  Costs one pass over A and B; the product does M*N*K / (M*K + K*N) times more work.
//   2. gemm_i8_nt_kernel: 256 x 256 (16 waves) or 128 x 128 (4 waves) block tile, each wave 2 x 2 MFMA tiles of 32 x 32,
//      K step 64, register-prefetched global loads, LDS rows padded to 80 bytes (conflict-free ds_read_b128), epilogue
//      acc mod p.
#include <cstdlib>
#include <vector>

#include "gfa_internal.h"
#include "gfa_karatsuba.h"

using namespace gfa;

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BK = 64, PITCH = 80;
// GF(2^m) bit planes from M N K >= 2^GFA_MFMA_BITS_MIN_LOG (environment, read once; default 24 = 256^3, where the planes already win 50 us to 121 us for GF(2^8) and 86 us to 1.5 ms for GF(2^16): tools/matmul_bits_crossover.py; the tests lower it)
inline int mfma_bits_min_log()
{
    static const int v = [] { const char *e = getenv("GFA_MFMA_BITS_MIN_LOG"); const int x = e ? atoi(e) : 24; return x < 16 ? 16 : (x > 62 ? 62 : x); }();
    return v;
}

// stream-ordered work buffers of one call, released on EVERY way out of it (a failed launch in the middle used to leave them in the pool's books)
struct Scratch {
    hipStream_t st;
    void *p[4] = {nullptr, nullptr, nullptr, nullptr};
    int n = 0;
    explicit Scratch(hipStream_t s) : st(s) {}
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    hipError_t get(void **out, size_t bytes)
    {
        const hipError_t e = gfa::scratch_alloc(out, bytes, st);
        if (e == hipSuccess) p[n++] = *out;
        return e;
    }
    ~Scratch()
    {
        for (int i = 0; i < n; i++) (void)gfa::scratch_free(p[i], st);
    }
};

__device__ __forceinline__ int8_t centre(u32 a, u32 p, u32 half) { return (int8_t)(a > half ? (int)a - (int)p : (int)a); }
// operand byte: the centred residue (shift < 0, p <= 256) or the 7-bit limb of the element at bit `shift` (larger primes)
__device__ __forceinline__ int8_t operand(u64 a, u32 p, u32 half, int shift)
{
    return shift < 0 ? centre((u32)a, p, half) : (int8_t)((a >> shift) & 127u);
}

// dst[r][c] = centre(src[r][c]) for r < rows, c < cols; zero elsewhere in the (rows_p x cols_p) padded array
template <typename T>
__global__ __launch_bounds__(256) void centre_rows_kernel(const T *__restrict__ src, int8_t *__restrict__ dst, i64 rows, i64 cols,
                                                          i64 rows_p, i64 cols_p, u32 p, i64 src_bstride, i64 dst_bstride, int shift)
{
    const T *s = src + (i64)blockIdx.z * src_bstride;
    int8_t *d = dst + (i64)blockIdx.z * dst_bstride;
    const u32 half = p >> 1;
    const i64 total = rows_p * cols_p;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (i64)gridDim.x * blockDim.x) {
        const i64 r = e / cols_p, c = e - r * cols_p;
        d[e] = (r < rows && c < cols) ? operand((u64)s[r * cols + c], p, half, shift) : (int8_t)0;
    }
}

// dst[c][r] = centre(src[r][c]) (transpose), 32 x 32 tiles through LDS; dst is (cols_p x rows_p), zero padded
template <typename T>
__global__ __launch_bounds__(256) void centre_transpose_kernel(const T *__restrict__ src, int8_t *__restrict__ dst, i64 rows, i64 cols,
                                                               i64 rows_p, i64 cols_p, u32 p, i64 src_bstride, i64 dst_bstride, int shift)
{
    __shared__ int8_t tile[32][33];
    const T *s = src + (i64)blockIdx.z * src_bstride;
    int8_t *d = dst + (i64)blockIdx.z * dst_bstride;
    const u32 half = p >> 1;
    const i64 r0 = (i64)blockIdx.y * 32, c0 = (i64)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const i64 r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? operand((u64)s[r * cols + c], p, half, shift) : (int8_t)0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const i64 c = c0 + j, r = r0 + tx; // dst row = source column
        if (c < cols_p && r < rows_p) d[c * rows_p + r] = tile[tx][j];
    }
}

// C[m][n] = (sum_k A'[m][k] * Bt'[n][k]) mod p.  A': (Mp x Kp), Bt': (Np x Kp), both padded to 256.
// RAW: C is an int32 array and the exact integer sums are ADDED to it (limb products of one diagonal share a buffer).
// Block tile (64 WM) x (64 WN) with WM x WN waves of 64 x 64 each.  The kernel is bound by the L2 -> LDS operand stream
// (every block re-reads its A and B panels): 128 x 128 tiles moved 8 TB/s from L2 at 512 TMAC/s, so large products use
// 256 x 256 tiles (16 waves, one workgroup per CU) and halve that traffic.
template <typename T, bool RAW, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_i8_nt_kernel(const int8_t *__restrict__ A, const int8_t *__restrict__ Bt,
                                                                   T *__restrict__ C, int M, int N, int Kp, i64 a_bstride,
                                                                   i64 b_bstride, int p)
{
    constexpr int TBM = 64 * WM, TBN = 64 * WN, THREADS = 64 * WM * WN;
    constexpr int ACH = TBM * 4 / THREADS, BCH = TBN * 4 / THREADS; // 16-byte chunks per thread per K slab
    static_assert(ACH >= 1 && BCH >= 1 && (TBM * 4) % THREADS == 0 && (TBN * 4) % THREADS == 0, "tile / thread mismatch");
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    int8_t *As0 = smem, *Bs0 = smem + 2 * TBM * PITCH; // As[2][TBM*PITCH], Bs[2][TBN*PITCH]
    const int8_t *Ab = A + (i64)blockIdx.z * a_bstride + (i64)blockIdx.y * TBM * Kp;
    const int8_t *Bb = Bt + (i64)blockIdx.z * b_bstride + (i64)blockIdx.x * TBN * Kp;
    T *Cb = C + (i64)blockIdx.z * M * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int srow = tid >> 2, schunk = (tid & 3) * 16; // thread moves rows srow + j * THREADS/4, 16-byte chunk tid % 4
    v4i ga[ACH], gb[BCH];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < ACH; j++) ga[j] = *reinterpret_cast<const v4i *>(Ab + (i64)(srow + (THREADS / 4) * j) * Kp + k0 + schunk);
#pragma unroll
        for (int j = 0; j < BCH; j++) gb[j] = *reinterpret_cast<const v4i *>(Bb + (i64)(srow + (THREADS / 4) * j) * Kp + k0 + schunk);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < ACH; j++) *reinterpret_cast<v4i *>(As0 + buf * TBM * PITCH + (srow + (THREADS / 4) * j) * PITCH + schunk) = ga[j];
#pragma unroll
        for (int j = 0; j < BCH; j++) *reinterpret_cast<v4i *>(Bs0 + buf * TBN * PITCH + (srow + (THREADS / 4) * j) * PITCH + schunk) = gb[j];
    };
    v16i acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0;

    gload(0);
    lstore(0);
    __syncthreads();
    const int nsteps = Kp / BK;
    const int frow = lane & 31, fk = (lane >> 5) * 16; // MFMA operand: row/column `frow`, 16 consecutive k from `fk`
    for (int s = 0; s < nsteps; s++) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload((s + 1) * BK);
        const int8_t *Asb = As0 + buf * TBM * PITCH, *Bsb = Bs0 + buf * TBN * PITCH;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 32) {
            v4i a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                a[t] = *reinterpret_cast<const v4i *>(Asb + (wm * 64 + t * 32 + frow) * PITCH + kk + fk);
                b[t] = *reinterpret_cast<const v4i *>(Bsb + (wn * 64 + t * 32 + frow) * PITCH + kk + fk);
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nsteps) {
            lstore(buf ^ 1);
            __syncthreads();
        }
    }
    // epilogue: D[row][col], col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int m_base = blockIdx.y * TBM + wm * 64, n_base = blockIdx.x * TBN + wn * 64;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = n_base + j * 32 + (lane & 31);
                if (row < M && col < N) {
                    if constexpr (RAW) {
                        Cb[(i64)row * N + col] += (T)acc[i][j][r];
                    } else {
                        int v = acc[i][j][r] % p;
                        if (v < 0) v += p;
                        Cb[(i64)row * N + col] = (T)v;
                    }
                }
            }
}

// launches the tile configuration that suits the product: 256 x 256 for large M, N, else 128 x 128
template <typename T, bool RAW>
int launch_gemm(const int8_t *A, const int8_t *Bt, T *C, i64 M, i64 N, i64 Mp, i64 Np, i64 Kp, i64 a_bstride, i64 b_bstride, i64 batch,
                int p, hipStream_t st)
{
    if (M >= 1024 && N >= 1024) {
        auto k = gemm_i8_nt_kernel<T, RAW, 4, 4>;
        constexpr size_t lds = 2 * (256 + 256) * PITCH;
        static bool attr = false;
        if (!attr) { GFA_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
        hipLaunchKernelGGL(k, dim3((unsigned)(Np / 256), (unsigned)(Mp / 256), (unsigned)batch), dim3(1024), lds, st, A, Bt, C, (int)M, (int)N,
                           (int)Kp, a_bstride, b_bstride, p);
    } else {
        auto k = gemm_i8_nt_kernel<T, RAW, 2, 2>;
        constexpr size_t lds = 2 * (128 + 128) * PITCH;
        hipLaunchKernelGGL(k, dim3((unsigned)(Np / 128), (unsigned)(Mp / 128), (unsigned)batch), dim3(256), lds, st, A, Bt, C, (int)M, (int)N,
                           (int)Kp, a_bstride, b_bstride, p);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

template <typename T>
int run_mfma(const FieldDev &fd, const void *a, const void *b, void *out, i64 batch, i64 M, i64 K, i64 N, i64 a_bstride,
             i64 b_bstride, hipStream_t st)
{
    const i64 Mp = (M + 255) / 256 * 256, Np = (N + 255) / 256 * 256, Kp = (K + BK - 1) / BK * BK;
    const i64 nA = a_bstride ? batch : 1, nB = b_bstride ? batch : 1;
    int8_t *Ac = nullptr, *Bc = nullptr;
    Scratch ws(st);
    GFA_HIP(ws.get((void **)&Ac, (size_t)(nA * Mp * Kp)));
    GFA_HIP(ws.get((void **)&Bc, (size_t)(nB * Np * Kp)));
    const u32 p = (u32)fd.p;
    {
        const i64 total = Mp * Kp;
        const unsigned gx = (unsigned)std::min<i64>((total + 255) / 256, 65535);
        hipLaunchKernelGGL(centre_rows_kernel<T>, dim3(gx, 1, (unsigned)nA), dim3(256), 0, st, (const T *)a, Ac, M, K, Mp, Kp, p,
                           a_bstride, Mp * Kp, -1);
        hipLaunchKernelGGL(centre_transpose_kernel<T>, dim3((unsigned)(Np / 32), (unsigned)(Kp / 32), (unsigned)nB), dim3(256), 0, st,
                           (const T *)b, Bc, K, N, Kp, Np, p, b_bstride, Np * Kp, -1);
    }
    int rcg = launch_gemm<T, false>(Ac, Bc, (T *)out, M, N, Mp, Np, Kp, a_bstride ? Mp * Kp : 0, b_bstride ? Np * Kp : 0, batch, (int)p, st);
    if (rcg) return rcg;
    return GFA_OK;
}

// ---- primes up to 2^31: 7-bit limbs ------------------------------------------------------------------------------
// a = sum_l a_l 2^(7l), 0 <= a_l < 128, NL = ceil(bits(p) / 7) limbs (3 for p < 2^21, 5 for p < 2^35).  Every limb pair
// (i, j) is one exact int8 GEMM whose int32 sums are added to the buffer of its diagonal s = i + j (at most NL products
// of K * 127^2 each: K is limited so that the buffer cannot overflow); a last pass folds the 2NL-1 diagonals,
// out = sum_s (D_s mod p) * (2^(7s) mod p) mod p.  NL^2 matrix-core GEMMs instead of M*N*K Barrett products.
template <typename T>
__global__ __launch_bounds__(256) void fold_diagonals_kernel(const int *__restrict__ D, int ndiag, i64 plane, T *__restrict__ out,
                                                             i64 count, u64 p, u64 mu)
{
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (i64)gridDim.x * blockDim.x) {
        u64 acc = 0, w = 1; // w = 2^(7s) mod p
        for (int sdg = 0; sdg < ndiag; sdg++) {
            const u64 v = (u64)(u32)D[(i64)sdg * plane + e] % p; // sums are non-negative (unsigned limbs)
            acc = (u64)(((unsigned __int128)acc + (unsigned __int128)v * w % p) % p); // (p up to 2^64: nothing here may wrap)
            w = (u64)(((unsigned __int128)w << 7) % p);
        }
        out[e] = (T)acc;
    }
    (void)mu;
}

template <typename T>
int run_mfma_limbs(const FieldDev &fd, int nl, const void *a, const void *b, void *out, i64 batch, i64 M, i64 K, i64 N, i64 a_bstride,
                   i64 b_bstride, hipStream_t st)
{
    const i64 Mp = (M + 255) / 256 * 256, Np = (N + 255) / 256 * 256, Kp = (K + BK - 1) / BK * BK;
    const int ndiag = 2 * nl - 1;
    const i64 plane = M * N;
    int8_t *Ac = nullptr, *Bc = nullptr;
    int *D = nullptr;
    Scratch ws(st);
    GFA_HIP(ws.get((void **)&Ac, (size_t)(nl * Mp * Kp)));
    GFA_HIP(ws.get((void **)&Bc, (size_t)(nl * Np * Kp)));
    GFA_HIP(ws.get((void **)&D, sizeof(int) * (size_t)(ndiag * plane)));
    const u32 p32 = (u32)(fd.p & 0xffffffffu);
    static const int itemsize[4] = {1, 2, 4, 8};
    (void)itemsize;
    for (i64 bi = 0; bi < batch; bi++) { // one matrix pair at a time: the diagonal buffers are the large scratch
        const T *pa = (const T *)a + bi * a_bstride;
        const T *pb = (const T *)b + bi * b_bstride;
        if (bi == 0 || a_bstride)
            for (int l = 0; l < nl; l++) {
                const unsigned gx = (unsigned)std::min<i64>((Mp * Kp + 255) / 256, 65535);
                hipLaunchKernelGGL(centre_rows_kernel<T>, dim3(gx, 1, 1), dim3(256), 0, st, pa, Ac + (i64)l * Mp * Kp, M, K, Mp, Kp, p32,
                                   (i64)0, (i64)0, 7 * l);
            }
        if (bi == 0 || b_bstride)
            for (int l = 0; l < nl; l++)
                hipLaunchKernelGGL(centre_transpose_kernel<T>, dim3((unsigned)(Np / 32), (unsigned)(Kp / 32), 1), dim3(256), 0, st, pb,
                                   Bc + (i64)l * Np * Kp, K, N, Kp, Np, p32, (i64)0, (i64)0, 7 * l);
        GFA_HIP(hipMemsetAsync(D, 0, sizeof(int) * (size_t)(ndiag * plane), st));
        if (Mp == M) { // r06: the nl limb planes of A are one contiguous (nl M) x Kp operand, and limb i of the product with limb j of B belongs to
                       // diagonal i + j = nl consecutive planes of D starting at j: nl launches instead of nl^2
            for (int j = 0; j < nl; j++) {
                int rcg = launch_gemm<int, true>(Ac, Bc + (i64)j * Np * Kp, D + (i64)j * plane, (i64)nl * M, N, (i64)nl * Mp, Np, Kp, 0, 0, 1, 0, st);
                if (rcg) return rcg;
            }
        } else
        for (int i = 0; i < nl; i++)
            for (int j = 0; j < nl; j++) {
                int rcg = launch_gemm<int, true>(Ac + (i64)i * Mp * Kp, Bc + (i64)j * Np * Kp, D + (i64)(i + j) * plane, M, N, Mp, Np, Kp, 0, 0, 1, 0, st);
                if (rcg) return rcg;
            }
        const unsigned gf = (unsigned)std::min<i64>((plane + 255) / 256, 65535);
        hipLaunchKernelGGL(fold_diagonals_kernel<T>, dim3(gf), dim3(256), 0, st, D, ndiag, plane, (T *)out + bi * plane, plane, fd.p, fd.mu);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// ---- GF(2^m), 2 <= m <= 32: Karatsuba bit planes (r06) ---------------------------------------------------------------
// a = sum_i a_i x^i with a_i in {0, 1}: the product of two elements is a product in GF(2)[x] reduced mod f.  Three levels of Karatsuba over
// the eight (four levels: sixteen, five: thirty-two) bit positions turn it into 27 (81, 243) products of single BITS -- each the parity of the element under a
// mask -- and the result into sum_t P_t r_t(x) with fixed polynomials r_t (linear over GF(2): c = P_lo (1 + x^h) + P_hi (x^h + x^2h)
// + P_mid x^h, recursively).  For matrices: plane t of A is the 0/1 matrix parity(A & mask_t), likewise B; P_t = (A_t B_t) mod 2 is ONE
// exact int8 GEMM with the epilogue of GF(2) (the 27 products ride on the batch dimension of a single launch), and the fold xors
// r_t(x) mod f where P_t is set.  27 instead of 64 plane products for GF(2^8).  Same values as the reference's loops of multiply / add
// ufuncs (_domains/_linalg.py:286-308).
template <typename T>
__global__ __launch_bounds__(256) void fold_bits_kernel(const uint8_t *__restrict__ P, BinFold bf, i64 plane, T *__restrict__ out, i64 count)
{
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (i64)gridDim.x * blockDim.x) {
        u32 v = 0;
        for (int t = 0; t < bf.nt; t++) v ^= P[(i64)t * plane + e] ? (u32)bf.red[t] : 0u;
        out[e] = (T)v;
    }
}

// all nt bit planes of an operand in ONE pass over it (the per-plane launch of centre_*_kernel re-read the operand nt times: 148 us of the 470 us
// of a 1024^3 product over GF(2^16)): plane t = parity(element & mask_t)
template <typename T>
__global__ __launch_bounds__(256) void bits_rows_kernel(const T *__restrict__ src, int8_t *__restrict__ dst, i64 rows, i64 cols, i64 rows_p, i64 cols_p, int nt,
                                                        PlaneMasks pm)
{
    const i64 total = rows_p * cols_p;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (i64)gridDim.x * blockDim.x) {
        const i64 r = e / cols_p, c = e - r * cols_p;
        const u32 v = (r < rows && c < cols) ? (u32)src[r * cols + c] : 0u;
        for (int t = 0; t < nt; t++) dst[(i64)t * total + e] = (int8_t)(__popc(v & pm.m[t]) & 1);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void bits_transpose_kernel(const T *__restrict__ src, int8_t *__restrict__ dst, i64 rows, i64 cols, i64 rows_p, i64 cols_p, int nt,
                                                             PlaneMasks pm)
{
    __shared__ int8_t tile[32][33];
    const i64 r0 = (i64)blockIdx.y * 32, c0 = (i64)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8: a thread owns the four elements (ty + 8 jj, tx)
    u32 v[4];
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        const i64 r = r0 + ty + 8 * jj, c = c0 + tx;
        v[jj] = (r < rows && c < cols) ? (u32)src[r * cols + c] : 0u;
    }
    const i64 plane = rows_p * cols_p;
    for (int t = 0; t < nt; t++) {
        const u32 mask = pm.m[t];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) tile[ty + 8 * jj][tx] = (int8_t)(__popc(v[jj] & mask) & 1);
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const i64 c = c0 + j, r = r0 + tx; // dst row = source column
            if (c < cols_p && r < rows_p) dst[(i64)t * plane + c * rows_p + r] = tile[tx][j];
        }
        __syncthreads();
    }
}

template <typename T>
int run_mfma_bits(const FieldDev &fd, const void *a, const void *b, void *out, i64 batch, i64 M, i64 K, i64 N, i64 a_bstride, i64 b_bstride,
                  hipStream_t st)
{
    BinFold bf{};
    PlaneMasks pm{};
    make_bin_fold(fd, &pm, &bf);
    const int nt = bf.nt;
    const i64 Mp = (M + 255) / 256 * 256, Np = (N + 255) / 256 * 256, Kp = (K + BK - 1) / BK * BK;
    const i64 plane = M * N;
    int8_t *Ac = nullptr, *Bc = nullptr;
    uint8_t *P = nullptr;
    Scratch ws(st);
    GFA_HIP(ws.get((void **)&Ac, (size_t)(nt * Mp * Kp)));
    GFA_HIP(ws.get((void **)&Bc, (size_t)(nt * Np * Kp)));
    GFA_HIP(ws.get((void **)&P, (size_t)(nt * plane)));
    for (i64 bi = 0; bi < batch; bi++) { // one matrix pair at a time: nt planes of each operand are the large scratch
        const T *pa = (const T *)a + bi * a_bstride;
        const T *pb = (const T *)b + bi * b_bstride;
        if (bi == 0 || a_bstride) { // all nt planes of an operand in one launch: plane z = parity(element & mask_z)
            const unsigned gx = (unsigned)std::min<i64>((Mp * Kp + 255) / 256, 65535);
            hipLaunchKernelGGL(bits_rows_kernel<T>, dim3(gx), dim3(256), 0, st, pa, Ac, M, K, Mp, Kp, nt, pm);
        }
        if (bi == 0 || b_bstride)
            hipLaunchKernelGGL(bits_transpose_kernel<T>, dim3((unsigned)(Np / 32), (unsigned)(Kp / 32), 1), dim3(256), 0, st, pb, Bc, K, N, Kp, Np, nt, pm);
        int rcg = launch_gemm<uint8_t, false>(Ac, Bc, P, M, N, Mp, Np, Kp, Mp * Kp, Np * Kp, nt, 2, st); // P_t = (A_t B_t) mod 2, t on the batch dimension
        if (rcg) return rcg;
        const unsigned gf = (unsigned)std::min<i64>((plane + 255) / 256, 65535);
        hipLaunchKernelGGL(fold_bits_kernel<T>, dim3(gf), dim3(256), 0, st, P, bf, plane, (T *)out + bi * plane, plane);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

// ---- GF(p^m), odd p <= 251, 2 <= m <= 16: Karatsuba digit planes (r06) ----------------------------------------------------
// The same construction over GF(p): elements are polynomials in their base-p digits, Karatsuba over the (padded) digit positions gives
// leaves t with a SET of positions E_t and an integer weight polynomial (c = P_lo (1 - x^h) + P_hi (x^2h - x^h) + P_mid x^h, recursively);
// plane t of an operand is (sum of the digits in E_t) mod p as a centred residue, P_t = (A_t B_t) mod p one exact int8 GEMM with the prime
// field's epilogue, and -- reduction mod f being linear too -- digit k of the result is (sum_t P_t R_t[k]) mod p with R_t = weight_t mod f
// computed on the host.  Leaves whose set is empty (positions padded up to a power of two) are dropped: 8 products for degree 3, 22 for
// degree 5 (schoolbook: 9, 25).  The table kernels these fields ran on manage 0.17 TMAC/s (GF(3^5), 1024^3).
__device__ __forceinline__ int8_t digit_plane(const u32 (&d)[16], u32 set, u32 p, u32 half)
{
    u32 sum = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) sum += ((set >> i) & 1u) ? d[i] : 0u;
    return centre(sum % p, p, half);
}

// all nt planes of A in one pass: dst plane t = (rows_p x cols_p), zero padded
template <typename T>
__global__ __launch_bounds__(256) void digit_rows_kernel(const T *__restrict__ src, int8_t *__restrict__ dst, i64 rows, i64 cols, i64 rows_p, i64 cols_p, DigitFold df)
{
    const u32 half = df.p >> 1;
    const i64 total = rows_p * cols_p;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (i64)gridDim.x * blockDim.x) {
        const i64 r = e / cols_p, c = e - r * cols_p;
        const bool in = r < rows && c < cols;
        u32 d[16];
        if (in) digits_of(src[r * cols + c], df.p, df.m, d);
        for (int t = 0; t < df.nt; t++) dst[(i64)t * total + e] = in ? digit_plane(d, df.set[t], df.p, half) : (int8_t)0;
    }
}
// all nt planes of B, transposed (dst plane t = (cols_p x rows_p)), 32 x 32 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void digit_transpose_kernel(const T *__restrict__ src, int8_t *__restrict__ dst, i64 rows, i64 cols, i64 rows_p, i64 cols_p, DigitFold df)
{
    __shared__ int8_t tile[32][33];
    const u32 half = df.p >> 1;
    const i64 r0 = (i64)blockIdx.y * 32, c0 = (i64)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8: a thread owns the four elements (ty + 8 jj, tx)
    u32 d[4][16];
    bool in[4];
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        const i64 r = r0 + ty + 8 * jj, c = c0 + tx;
        in[jj] = r < rows && c < cols;
        if (in[jj]) digits_of(src[r * cols + c], df.p, df.m, d[jj]);
    }
    const i64 plane = rows_p * cols_p;
    for (int t = 0; t < df.nt; t++) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++) tile[ty + 8 * jj][tx] = in[jj] ? digit_plane(d[jj], df.set[t], df.p, half) : (int8_t)0;
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const i64 c = c0 + j, r = r0 + tx; // dst row = source column
            if (c < cols_p && r < rows_p) dst[(i64)t * plane + c * rows_p + r] = tile[tx][j];
        }
        __syncthreads();
    }
}
template <typename T>
__global__ __launch_bounds__(256) void fold_digits_kernel(const uint8_t *__restrict__ P, DigitFold df, i64 plane, T *__restrict__ out, i64 count)
{
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (i64)gridDim.x * blockDim.x) {
        u32 acc[16];
        for (int k = 0; k < df.m; k++) acc[k] = 0;
        for (int t = 0; t < df.nt; t++) {
            const u32 v = P[(i64)t * plane + e];
            if (v)
                for (int k = 0; k < df.m; k++) acc[k] += v * df.R[t][k]; // <= 81 * 250 * 250
        }
        u64 r = 0;
        for (int k = df.m - 1; k >= 0; k--) r = r * df.p + acc[k] % df.p;
        out[e] = (T)r;
    }
}

template <typename T>
int run_mfma_digits(const FieldDev &fd, const void *a, const void *b, void *out, i64 batch, i64 M, i64 K, i64 N, i64 a_bstride, i64 b_bstride,
                    hipStream_t st)
{
    DigitFold df{};
    if (!make_digit_fold(fd, &df)) return GFA_ERR_UNSUPPORTED;
    const int nt = df.nt;
    const u32 p = df.p;
    const i64 Mp = (M + 255) / 256 * 256, Np = (N + 255) / 256 * 256, Kp = (K + BK - 1) / BK * BK;
    const i64 plane = M * N;
    int8_t *Ac = nullptr, *Bc = nullptr;
    uint8_t *P = nullptr;
    Scratch ws(st);
    GFA_HIP(ws.get((void **)&Ac, (size_t)(nt * Mp * Kp)));
    GFA_HIP(ws.get((void **)&Bc, (size_t)(nt * Np * Kp)));
    GFA_HIP(ws.get((void **)&P, (size_t)(nt * plane)));
    for (i64 bi = 0; bi < batch; bi++) {
        const T *pa = (const T *)a + bi * a_bstride;
        const T *pb = (const T *)b + bi * b_bstride;
        if (bi == 0 || a_bstride) {
            const unsigned gx = (unsigned)std::min<i64>((Mp * Kp + 255) / 256, 65535);
            hipLaunchKernelGGL(digit_rows_kernel<T>, dim3(gx), dim3(256), 0, st, pa, Ac, M, K, Mp, Kp, df);
        }
        if (bi == 0 || b_bstride)
            hipLaunchKernelGGL(digit_transpose_kernel<T>, dim3((unsigned)(Np / 32), (unsigned)(Kp / 32), 1), dim3(256), 0, st, pb, Bc, K, N, Kp, Np, df);
        int rcg = launch_gemm<uint8_t, false>(Ac, Bc, P, M, N, Mp, Np, Kp, Mp * Kp, Np * Kp, nt, (int)p, st); // P_t = (A_t B_t) mod p
        if (rcg) return rcg;
        const unsigned gf = (unsigned)std::min<i64>((plane + 255) / 256, 65535);
        hipLaunchKernelGGL(fold_digits_kernel<T>, dim3(gf), dim3(256), 0, st, P, df, plane, (T *)out + bi * plane, plane);
    }
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int limbs_for(u64 p)
{
    int bits = 0;
    while (bits < 64 && (p >> bits)) bits++;
    return (bits + 6) / 7;
}

} // namespace

namespace gfa {

// true if the matrix-core path applies: prime field with p <= 256, K small enough for exact int32 accumulation, and a
// product large enough to amortise the centring pass.  Batches ride on gridDim.z (<= 65535 per call, sliced by the caller).
bool matmul_mfma_eligible(const FieldDev &fd, i64 M, i64 K, i64 N)
{
    if (fd.kind == KIND_BIN && fd.m >= 2 && fd.m <= 32) { // r06: Karatsuba bit planes: 27 (m <= 8) / 81 (m <= 16) / at most 243 GEMMs over GF(2), two staging launches, a fold
        const i64 cap = (i64)1 << (fd.m <= 16 ? 28 : 25); // planes: at most 243 x this many bytes per operand
        return M >= 128 && N >= 128 && K < ((i64)1 << 31) && M * N * K >= ((i64)1 << (mfma_bits_min_log() - (fd.m > 8 ? 2 : 0))) && M * N <= cap && M * K <= cap &&
               K * N <= cap;
    }
    if (fd.kind == KIND_EXT && (fd.p & 1) && fd.p <= 251 && fd.m >= 2 && fd.m <= 16) // r06: Karatsuba digit planes (K as for GF(p): exact int32 sums)
        return M >= 128 && N >= 128 && K <= 131072 && M * N * K >= ((i64)1 << mfma_bits_min_log()) && M * N <= ((i64)1 << 28) && M * K <= ((i64)1 << 28) &&
               K * N <= ((i64)1 << 28);
    if (fd.m != 1 || M > (1 << 24) || N > (1 << 24)) return false;
    if (fd.p <= 256) return K <= 131072 && M * N * K >= ((i64)1 << 21);
    if (fd.kind != KIND_PRIME32 && fd.kind != KIND_PRIME64 && fd.kind != KIND_GOLDILOCKS) return false;
    // 7-bit limbs: NL^2 GEMMs + NL staging passes + a fold; worth it for large products only, and NL * K * 127^2 < 2^31
    // (r06: also the 64-bit primes -- ten limbs, 100 GEMMs, 19 diagonals: Goldilocks 1024^3 0.87 -> see profiles/r06_linalg_bench.txt)
    const int nl = limbs_for(fd.p);
    // (measured: Goldilocks 1024^3 3.9 ms against 1.24 ms on the scalar kernel -- 100 launches, 19 read-modify-written diagonals and a 128-bit
    // fold -- 2048^3 6.9 against 9.9 ms: more than five limbs from 2^33 multiply-adds)
    return nl <= 10 && (i64)nl * K * 16129 < ((i64)1 << 31) && M * N * K >= ((i64)1 << (nl <= 5 ? 27 : 33)) && M >= 64 && N >= 64 &&
           M * N <= ((i64)1 << (nl <= 5 ? 27 : 26));
}

int matmul_mfma(const FieldDev &fd, int dtype, const void *a, const void *b, void *out, i64 batch, i64 M, i64 K, i64 N, i64 a_bstride,
                i64 b_bstride, hipStream_t st)
{
    if (fd.kind == KIND_BIN && fd.m >= 2) {
        switch (dtype) {
        case GFA_U8: return run_mfma_bits<uint8_t>(fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
        case GFA_U16: return run_mfma_bits<uint16_t>(fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
        case GFA_U32: return run_mfma_bits<uint32_t>(fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
        case GFA_U64: return run_mfma_bits<uint64_t>(fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
        default: set_error("gfa_matmul: bad dtype"); return GFA_ERR_INVALID;
        }
    }
    if (fd.kind == KIND_EXT) {
        switch (dtype) {
        case GFA_U8: return run_mfma_digits<uint8_t>(fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
        case GFA_U16: return run_mfma_digits<uint16_t>(fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
        case GFA_U32: return run_mfma_digits<uint32_t>(fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
        case GFA_U64: return run_mfma_digits<uint64_t>(fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
        default: set_error("gfa_matmul: bad dtype"); return GFA_ERR_INVALID;
        }
    }
    if (fd.p > 256) {
        const int nl = limbs_for(fd.p);
        switch (dtype) {
        case GFA_U32: return run_mfma_limbs<uint32_t>(fd, nl, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
        case GFA_U64: return run_mfma_limbs<uint64_t>(fd, nl, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
        case GFA_U16: return run_mfma_limbs<uint16_t>(fd, nl, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
        default: set_error("gfa_matmul: bad dtype"); return GFA_ERR_INVALID;
        }
    }
    switch (dtype) {
    case GFA_U8: return run_mfma<uint8_t>(fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
    case GFA_U16: return run_mfma<uint16_t>(fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
    case GFA_U32: return run_mfma<uint32_t>(fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
    case GFA_U64: return run_mfma<uint64_t>(fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
    default: set_error("gfa_matmul: bad dtype"); return GFA_ERR_INVALID;
    }
}

} // namespace gfa
