// gfa_linalg.hip -- field linear algebra on the device.
//
//   gfa_matmul ........ batched C = A @ B, LDS-tiled; replaces matmul_jit.implementation (_domains/_linalg.py:286-308) and
//                       the "BLAS then % p" shortcut the reference takes for prime fields (_lapack_linalg, :21-75)
//   gfa_row_reduce .... Gauss-Jordan to reduced row echelon form, pivot = first non-zero at/below the pivot row
//                       (row_reduce_jit, _linalg.py:315-351); inv / solve / rank / the four subspaces are built on it
//   gfa_plu_decompose . LU with or without row exchanges (lu_decompose_jit :354-384, plu_decompose_jit :387-424) and
//                       the determinant as (-1)^swaps * prod(diag U) (det_jit :447-477)
//
// Elimination kernels run one workgroup per matrix of the batch with the matrix left in HBM/L2 (they are latency bound,
// not bandwidth bound); the batch dimension is what fills the chip.  Everything is exact field arithmetic, so any
// correct elimination order gives the same RREF / determinant; the pivot rule above is kept so that L, U and P match
// the reference entry for entry.
#include "gfa_internal.h"

using namespace gfa;

namespace {


// ------------------------------------------------------------------------------------------------
// matmul: 64 x 64 output tile per 256-thread workgroup, 4 x 4 outputs per thread, K in slabs of 16
// ------------------------------------------------------------------------------------------------
constexpr int MM_BM = 64, MM_BN = 64, MM_BK = 16, MM_THREADS = 256;

// multiply-accumulate policy.  Default (FOLD = 0): acc = acc + a*b in the field.  Prime fields accumulate raw 64-bit
// products and reduce every FOLD steps of k: FOLD = 16 for p < 2^30 (16 (p-1)^2 + p < 2^64), FOLD = 4 for p < 2^31.
template <class F, int FOLD>
struct Mac {
    typedef typename F::elem acc_t;
    static __device__ __forceinline__ void mac(const FieldDev &fd, acc_t &acc, typename F::elem a, typename F::elem b)
    {
        acc = F::add(fd, acc, F::mul(fd, a, b));
    }
    static __device__ __forceinline__ void fold(const FieldDev &, acc_t &) {}
    static __device__ __forceinline__ typename F::elem result(const FieldDev &, acc_t acc) { return acc; }
};
template <int FOLD>
struct MacLazy32 {
    typedef u64 acc_t;
    static __device__ __forceinline__ void mac(const FieldDev &, acc_t &acc, u32 a, u32 b) { acc += (u64)a * b; }
    static __device__ __forceinline__ void fold(const FieldDev &fd, acc_t &acc) { acc = Prime32::reduce64(fd, acc); }
    static __device__ __forceinline__ u32 result(const FieldDev &, acc_t acc) { return (u32)acc; }
};
template <> struct Mac<Prime32, 16> : MacLazy32<16> {};
template <> struct Mac<Prime32, 4> : MacLazy32<4> {};

template <class F, typename T, int FOLD>
__global__ __launch_bounds__(MM_THREADS) void matmul_kernel(FieldDev fd, const T *__restrict__ A, const T *__restrict__ B,
                                                            T *__restrict__ C, int M, int K, int N, i64 a_bstride,
                                                            i64 b_bstride)
{
    typedef typename F::elem E;
    typedef Mac<F, FOLD> MAC;
    __shared__ E As[MM_BK][MM_BM + 1]; // As[k][m]
    __shared__ E Bs[MM_BK][MM_BN + 1]; // Bs[k][n]
    const int batch = blockIdx.z;
    const T *Ab = A + (i64)batch * a_bstride;
    const T *Bb = B + (i64)batch * b_bstride;
    T *Cb = C + (i64)batch * M * N;
    const int m0 = blockIdx.y * MM_BM, n0 = blockIdx.x * MM_BN;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4; // thread owns rows ty*4.., cols tx + 16*j (coalesced stores)
    typename MAC::acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0;

    for (int k0 = 0; k0 < K; k0 += MM_BK) {
        // stage A (64 x 16) and B (16 x 64): 1024 elements each, 4 per thread
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int idx = threadIdx.x + r * MM_THREADS;
            const int am = idx >> 4, ak = idx & 15; // consecutive threads walk k (row-major A: contiguous)
            const int gm = m0 + am, gk = k0 + ak;
            As[ak][am] = (gm < M && gk < K) ? (E)Ab[(i64)gm * K + gk] : (E)0;
            const int bk = idx >> 6, bn = idx & 63;
            const int gkb = k0 + bk, gn = n0 + bn;
            Bs[bk][bn] = (gkb < K && gn < N) ? (E)Bb[(i64)gkb * N + gn] : (E)0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < MM_BK; kk++) {
            E av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) av[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; j++) bv[j] = Bs[kk][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) MAC::mac(fd, acc[i][j], av[i], bv[j]);
            if (FOLD > 0 && (kk + 1) % (FOLD > 0 ? FOLD : 1) == 0) {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) MAC::fold(fd, acc[i][j]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int gm = m0 + ty * 4 + i;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int gn = n0 + tx + 16 * j;
            if (gn < N) Cb[(i64)gm * N + gn] = (T)MAC::result(fd, acc[i][j]);
        }
    }
}

// GF(2^m), q <= 256, uint8 storage: the 64 KiB product table lives in LDS, addition is XOR.
__global__ __launch_bounds__(MM_THREADS) void matmul_tab8_kernel(const uint8_t *__restrict__ table,
                                                                 const uint8_t *__restrict__ A,
                                                                 const uint8_t *__restrict__ B, uint8_t *__restrict__ C,
                                                                 int M, int K, int N, i64 a_bstride, i64 b_bstride)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *tab = lds;                                  // 65536
    uint8_t(*As)[MM_BM + 4] = reinterpret_cast<uint8_t(*)[MM_BM + 4]>(lds + 65536);
    uint8_t(*Bs)[MM_BN + 4] = reinterpret_cast<uint8_t(*)[MM_BN + 4]>(lds + 65536 + MM_BK * (MM_BM + 4));
    for (int i = threadIdx.x; i < 4096; i += MM_THREADS)
        reinterpret_cast<uint4 *>(tab)[i] = reinterpret_cast<const uint4 *>(table)[i];
    const int batch = blockIdx.z;
    const uint8_t *Ab = A + (i64)batch * a_bstride;
    const uint8_t *Bb = B + (i64)batch * b_bstride;
    uint8_t *Cb = C + (i64)batch * M * N;
    const int m0 = blockIdx.y * MM_BM, n0 = blockIdx.x * MM_BN;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    u32 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0;
    for (int k0 = 0; k0 < K; k0 += MM_BK) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int idx = threadIdx.x + r * MM_THREADS;
            const int am = idx >> 4, ak = idx & 15;
            const int gm = m0 + am, gk = k0 + ak;
            As[ak][am] = (gm < M && gk < K) ? Ab[(i64)gm * K + gk] : (uint8_t)0;
            const int bk = idx >> 6, bn = idx & 63;
            const int gkb = k0 + bk, gn = n0 + bn;
            Bs[bk][bn] = (gkb < K && gn < N) ? Bb[(i64)gkb * N + gn] : (uint8_t)0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < MM_BK; kk++) {
            u32 av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) av[i] = (u32)As[kk][ty * 4 + i] << 8;
#pragma unroll
            for (int j = 0; j < 4; j++) bv[j] = Bs[kk][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] ^= tab[av[i] | bv[j]];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int gm = m0 + ty * 4 + i;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int gn = n0 + tx + 16 * j;
            if (gn < N) Cb[(i64)gm * N + gn] = (uint8_t)acc[i][j];
        }
    }
}

template <class F, typename T>
int launch_matmul_ft(const FieldDev &fd, const void *a, const void *b, void *out, i64 batch, i64 M, i64 K, i64 N,
                     i64 a_bstride, i64 b_bstride, hipStream_t st)
{
    const dim3 grid((unsigned)((N + MM_BN - 1) / MM_BN), (unsigned)((M + MM_BM - 1) / MM_BM), (unsigned)batch);
    if constexpr (std::is_same<F, Prime32>::value) {
        if (fd.p < (1ull << 30)) {
            hipLaunchKernelGGL((matmul_kernel<F, T, 16>), grid, dim3(MM_THREADS), 0, st, fd, (const T *)a, (const T *)b,
                               (T *)out, (int)M, (int)K, (int)N, a_bstride, b_bstride);
            GFA_HIP(hipGetLastError());
            return GFA_OK;
        }
        if (fd.p < (1ull << 31)) {
            hipLaunchKernelGGL((matmul_kernel<F, T, 4>), grid, dim3(MM_THREADS), 0, st, fd, (const T *)a, (const T *)b,
                               (T *)out, (int)M, (int)K, (int)N, a_bstride, b_bstride);
            GFA_HIP(hipGetLastError());
            return GFA_OK;
        }
    }
    hipLaunchKernelGGL((matmul_kernel<F, T, 0>), grid, dim3(MM_THREADS), 0, st, fd, (const T *)a, (const T *)b,
                       (T *)out, (int)M, (int)K, (int)N, a_bstride, b_bstride);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}

int dispatch_matmul(const FieldDev &fd, int dtype, const void *a, const void *b, void *out, i64 batch, i64 M, i64 K, i64 N,
                    i64 a_bstride, i64 b_bstride, hipStream_t st)
{
    GFA_DISPATCH_FT(launch_matmul_ft, fd, dtype, fd, a, b, out, batch, M, K, N, a_bstride, b_bstride, st);
}

// ------------------------------------------------------------------------------------------------
// Gauss-Jordan / PLU: one workgroup per matrix
// ------------------------------------------------------------------------------------------------
constexpr int GJ_THREADS = 256;
constexpr int GJ_MAX_ROWS = 4096; // column factors are staged in LDS

template <class F>
__device__ __forceinline__ typename F::elem field_inv(const FieldDev &fd, typename F::elem a)
{
    return F::inv(fd, a);
}

// row_reduce_jit (_linalg.py:315-351).  A: (batch, m, n) in place.  rank_out[b] = number of pivots found.
template <class F, typename T>
__global__ __launch_bounds__(GJ_THREADS) void row_reduce_kernel(FieldDev fd, T *__restrict__ Aall, int m, int n, int ncols,
                                                                i64 *__restrict__ rank_out)
{
    typedef typename F::elem E;
    __shared__ E factor[GJ_MAX_ROWS];
    __shared__ int piv_row;
    __shared__ E piv_inv;
    T *A = Aall + (i64)blockIdx.x * m * n;
    int p = 0;
    for (int j = 0; j < ncols && p < m; j++) {
        if (threadIdx.x == 0) piv_row = m;
        __syncthreads();
        for (int i = p + threadIdx.x; i < m; i += GJ_THREADS)
            if (A[(i64)i * n + j] != 0) { atomicMin(&piv_row, i); break; } // rows ascend per thread: first hit is its minimum
        __syncthreads();
        const int pr = piv_row;
        if (pr == m) { __syncthreads(); continue; }
        if (threadIdx.x == 0) piv_inv = field_inv<F>(fd, (E)A[(i64)pr * n + j]);
        __syncthreads();
        const E inv = piv_inv;
        // swap rows p and pr, scaling the pivot row to a leading 1
        for (int c = threadIdx.x; c < n; c += GJ_THREADS) {
            const E top = (E)A[(i64)p * n + c];
            const E piv = F::mul(fd, (E)A[(i64)pr * n + c], inv);
            if (pr != p) A[(i64)pr * n + c] = (T)top;
            A[(i64)p * n + c] = (T)piv;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += GJ_THREADS) factor[i] = i == p ? (E)0 : (E)A[(i64)i * n + j];
        __syncthreads();
        // A[i, j:] -= factor[i] * A[p, j:] for every other row with a non-zero entry in column j (the pivot row is zero
        // to the left of column j).  Threads tile (rows x columns) with a power-of-two column count: no integer division.
        {
            const int width = n - j;
            int tcols = GJ_THREADS;
            while (tcols > 1 && (tcols >> 1) >= width) tcols >>= 1;
            const int lc = threadIdx.x & (tcols - 1), r0 = threadIdx.x / tcols, rstep = GJ_THREADS / tcols;
            for (int c = j + lc; c < n; c += tcols) {
                const E pv = (E)A[(i64)p * n + c];
                if (pv == 0) continue;
                for (int i = r0; i < m; i += rstep) {
                    const E f = factor[i];
                    if (f != 0) A[(i64)i * n + c] = (T)F::sub(fd, (E)A[(i64)i * n + c], F::mul(fd, f, pv));
                }
            }
        }
        __syncthreads();
        p++;
    }
    if (threadIdx.x == 0) rank_out[blockIdx.x] = p;
}

// ---- large matrices: the same elimination as two kernels per column, the update spread over the whole chip -----------
struct GjState { int p, cur, has, pad; };

template <class F, typename T>
__global__ __launch_bounds__(1024) void gj_pivot_kernel(FieldDev fd, T *__restrict__ Aall, int m, int n, int j,
                                                        GjState *__restrict__ state, u64 *__restrict__ factor_all)
{
    typedef typename F::elem E;
    __shared__ int piv_row;
    __shared__ E piv_inv;
    T *A = Aall + (i64)blockIdx.x * m * n;
    GjState *st = state + blockIdx.x;
    u64 *factor = factor_all + (i64)blockIdx.x * m;
    const int p = st->p;
    if (threadIdx.x == 0) piv_row = m;
    __syncthreads();
    if (p < m)
        for (int i = p + threadIdx.x; i < m; i += 1024)
            if (A[(i64)i * n + j] != 0) { atomicMin(&piv_row, i); break; }
    __syncthreads();
    const int pr = piv_row;
    if (pr >= m) {
        if (threadIdx.x == 0) st->has = 0;
        return;
    }
    if (threadIdx.x == 0) piv_inv = field_inv<F>(fd, (E)A[(i64)pr * n + j]);
    __syncthreads();
    const E inv = piv_inv;
    // rows p and pr are zero left of column j (earlier pivot columns were cleared, earlier pivot-free columns were
    // already zero from row p down), so the exchange and the scaling only touch columns >= j
    for (int c = j + threadIdx.x; c < n; c += 1024) {
        const E top = (E)A[(i64)p * n + c];
        const E piv = F::mul(fd, (E)A[(i64)pr * n + c], inv);
        if (pr != p) A[(i64)pr * n + c] = (T)top;
        A[(i64)p * n + c] = (T)piv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += 1024) factor[i] = i == p ? 0 : (u64)A[(i64)i * n + j];
    if (threadIdx.x == 0) { st->cur = p; st->p = p + 1; st->has = 1; }
}

// grid: (column tiles of 64 starting at column j, row tiles of 32, batch)
template <class F, typename T>
__global__ __launch_bounds__(256) void gj_eliminate_kernel(FieldDev fd, T *__restrict__ Aall, int m, int n, int j,
                                                           const GjState *__restrict__ state,
                                                           const u64 *__restrict__ factor_all)
{
    typedef typename F::elem E;
    const GjState st = state[blockIdx.z];
    if (!st.has) return;
    T *A = Aall + (i64)blockIdx.z * m * n;
    const u64 *factor = factor_all + (i64)blockIdx.z * m;
    const int c = j + blockIdx.x * 64 + (threadIdx.x & 63);
    if (c >= n) return;
    const E pv = (E)A[(i64)st.cur * n + c];
    if (pv == 0) return;
    const int r_end = min(m, (int)(blockIdx.y + 1) * 32);
    for (int i = blockIdx.y * 32 + (threadIdx.x >> 6); i < r_end; i += 4) {
        const E f = (E)factor[i];
        if (f != 0) A[(i64)i * n + c] = (T)F::sub(fd, (E)A[(i64)i * n + c], F::mul(fd, f, pv));
    }
}

__global__ void gj_finish_kernel(const GjState *__restrict__ state, i64 *__restrict__ rank_out, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) rank_out[b] = state[b].p;
}

template <class F, typename T>
int launch_row_reduce_wide_ft(const FieldDev &fd, void *a, i64 batch, i64 m, i64 n, i64 ncols, i64 *rank_out, hipStream_t st)
{
    GjState *state = nullptr;
    u64 *factor = nullptr;
    GFA_HIP(gfa::scratch_alloc((void **)&state, sizeof(GjState) * (size_t)batch, st));
    GFA_HIP(gfa::scratch_alloc((void **)&factor, sizeof(u64) * (size_t)(batch * m), st));
    GFA_HIP(hipMemsetAsync(state, 0, sizeof(GjState) * (size_t)batch, st));
    for (i64 j = 0; j < ncols; j++) {
        hipLaunchKernelGGL((gj_pivot_kernel<F, T>), dim3((unsigned)batch), dim3(1024), 0, st, fd, (T *)a, (int)m, (int)n, (int)j,
                           state, factor);
        const dim3 grid((unsigned)((n - j + 63) / 64), (unsigned)((m + 31) / 32), (unsigned)batch);
        hipLaunchKernelGGL((gj_eliminate_kernel<F, T>), grid, dim3(256), 0, st, fd, (T *)a, (int)m, (int)n, (int)j, state, factor);
    }
    hipLaunchKernelGGL(gj_finish_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st, state, rank_out, (int)batch);
    GFA_HIP(hipGetLastError());
    GFA_HIP(gfa::scratch_free(state, st));
    GFA_HIP(gfa::scratch_free(factor, st));
    return GFA_OK;
}
int dispatch_row_reduce_wide(const FieldDev &fd, int dtype, void *a, i64 batch, i64 m, i64 n, i64 ncols, i64 *rank_out,
                             hipStream_t st)
{
    GFA_DISPATCH_FT(launch_row_reduce_wide_ft, fd, dtype, fd, a, batch, m, n, ncols, rank_out, st);
}

// lu_decompose_jit / plu_decompose_jit (_linalg.py:354-424) and det_jit (:447-477).
//   A: (batch, m, n) in place -> U.   Lm: (batch, m, m) or NULL.   Pm: (batch, m, m) row-permutation matrix or NULL
//   (the reference returns its transpose).  PIVOT = false reproduces lu_decompose: a zero pivot with a non-zero entry
//   below it sets *err (the reference raises ValueError) and leaves the loop.
//   det_out (or NULL): (-1)^swaps * prod(diag U) for square matrices.
template <class F, typename T, bool PIVOT>
__global__ __launch_bounds__(GJ_THREADS) void plu_kernel(FieldDev fd, T *__restrict__ Aall, T *__restrict__ Lall,
                                                         T *__restrict__ Pall, int m, int n, i64 *__restrict__ nperm_out,
                                                         T *__restrict__ det_out, int32_t *__restrict__ err)
{
    typedef typename F::elem E;
    __shared__ E factor[GJ_MAX_ROWS];
    __shared__ int piv_row;
    __shared__ E piv_inv;
    T *A = Aall + (i64)blockIdx.x * m * n;
    T *Lm = Lall ? Lall + (i64)blockIdx.x * m * m : nullptr;
    T *Pm = Pall ? Pall + (i64)blockIdx.x * m * m : nullptr;
    // L = 0 (PLU) or I (LU); P = I
    for (i64 e = threadIdx.x; e < (i64)m * m; e += GJ_THREADS) {
        const int r = (int)(e / m), c = (int)(e % m);
        if (Lm) Lm[e] = (T)((!PIVOT && r == c) ? 1 : 0);
        if (Pm) Pm[e] = (T)(r == c ? 1 : 0);
    }
    __syncthreads();
    int nperm = 0;
    bool failed = false;
    const int steps = PIVOT ? (m < n ? m : n) : m - 1;
    for (int i = 0; i < steps; i++) {
        if (threadIdx.x == 0) piv_row = m;
        __syncthreads();
        const bool diag_zero = A[(i64)i * n + i] == 0;
        if (diag_zero) {
            for (int r = i + threadIdx.x; r < m; r += GJ_THREADS)
                if (A[(i64)r * n + i] != 0) { atomicMin(&piv_row, r); break; }
        }
        __syncthreads();
        if (diag_zero) {
            const int pr = piv_row;
            if (pr == m) {
                if (Lm && threadIdx.x == 0) Lm[(i64)i * m + i] = 1;
                __syncthreads();
                continue;
            }
            if (!PIVOT) { failed = true; break; }
            for (int c = threadIdx.x; c < n; c += GJ_THREADS) {
                const T t = A[(i64)i * n + c]; A[(i64)i * n + c] = A[(i64)pr * n + c]; A[(i64)pr * n + c] = t;
            }
            for (int c = threadIdx.x; c < m; c += GJ_THREADS) {
                if (Pm) { const T t = Pm[(i64)i * m + c]; Pm[(i64)i * m + c] = Pm[(i64)pr * m + c]; Pm[(i64)pr * m + c] = t; }
                if (Lm) { const T t = Lm[(i64)i * m + c]; Lm[(i64)i * m + c] = Lm[(i64)pr * m + c]; Lm[(i64)pr * m + c] = t; }
            }
            nperm++;
            __syncthreads();
        }
        if (threadIdx.x == 0) piv_inv = field_inv<F>(fd, (E)A[(i64)i * n + i]);
        __syncthreads();
        const E inv = piv_inv;
        for (int r = i + 1 + threadIdx.x; r < m; r += GJ_THREADS) {
            const E l = F::mul(fd, (E)A[(i64)r * n + i], inv);
            factor[r] = l;
            if (Lm) Lm[(i64)r * m + i] = (T)l;
        }
        if (Lm && threadIdx.x == 0) Lm[(i64)i * m + i] = 1;
        __syncthreads();
        const i64 total = (i64)(m - i - 1) * n;
        for (i64 e = threadIdx.x; e < total; e += GJ_THREADS) {
            const int r = i + 1 + (int)(e / n), c = (int)(e % n);
            const E f = factor[r];
            if (f != 0) A[(i64)r * n + c] = (T)F::sub(fd, (E)A[(i64)r * n + c], F::mul(fd, f, (E)A[(i64)i * n + c]));
        }
        __syncthreads();
    }
    if (failed) {
        if (threadIdx.x == 0 && err) atomicOr((int *)err, GFA_DEVERR_NO_LU);
        return;
    }
    if (threadIdx.x == 0) {
        if (Lm && PIVOT) Lm[(i64)(m - 1) * m + (m - 1)] = 1; // "set the final diagonal to 1" (_linalg.py:419)
        if (nperm_out) nperm_out[blockIdx.x] = nperm;
        if (det_out) {
            E d = F::one(fd);
            const int k = m < n ? m : n;
            for (int i = 0; i < k; i++) d = F::mul(fd, d, (E)A[(i64)i * n + i]);
            if (nperm & 1) d = F::neg(fd, d);
            det_out[blockIdx.x] = (T)d;
        }
    }
}

template <class F, typename T>
int launch_row_reduce_ft(const FieldDev &fd, void *a, i64 batch, i64 m, i64 n, i64 ncols, i64 *rank_out, hipStream_t st)
{
    hipLaunchKernelGGL((row_reduce_kernel<F, T>), dim3((unsigned)batch), dim3(GJ_THREADS), 0, st, fd, (T *)a, (int)m, (int)n,
                       (int)ncols, rank_out);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}
int dispatch_row_reduce(const FieldDev &fd, int dtype, void *a, i64 batch, i64 m, i64 n, i64 ncols, i64 *rank_out,
                        hipStream_t st)
{
    GFA_DISPATCH_FT(launch_row_reduce_ft, fd, dtype, fd, a, batch, m, n, ncols, rank_out, st);
}

template <class F, typename T>
int launch_plu_ft(const FieldDev &fd, bool pivot, void *a, void *l, void *p, i64 batch, i64 m, i64 n, i64 *nperm, void *det,
                  int32_t *err, hipStream_t st)
{
    if (pivot)
        hipLaunchKernelGGL((plu_kernel<F, T, true>), dim3((unsigned)batch), dim3(GJ_THREADS), 0, st, fd, (T *)a, (T *)l, (T *)p,
                           (int)m, (int)n, nperm, (T *)det, err);
    else
        hipLaunchKernelGGL((plu_kernel<F, T, false>), dim3((unsigned)batch), dim3(GJ_THREADS), 0, st, fd, (T *)a, (T *)l,
                           (T *)p, (int)m, (int)n, nperm, (T *)det, err);
    GFA_HIP(hipGetLastError());
    return GFA_OK;
}
int dispatch_plu(const FieldDev &fd, int dtype, bool pivot, void *a, void *l, void *p, i64 batch, i64 m, i64 n, i64 *nperm,
                 void *det, int32_t *err, hipStream_t st)
{
    GFA_DISPATCH_FT(launch_plu_ft, fd, dtype, fd, pivot, a, l, p, batch, m, n, nperm, det, err, st);
}

} // namespace

extern "C" {

int gfa_matmul(gfa_field_t *f, const void *a, const void *b, void *out, int64_t batch, int64_t M, int64_t K, int64_t N,
               int64_t a_batch_stride, int64_t b_batch_stride, int dtype, gfa_stream_t stream)
{
    if (!f || batch < 0 || M < 0 || K < 0 || N < 0 || a_batch_stride < 0 || b_batch_stride < 0) {
        set_error("gfa_matmul: bad arguments");
        return GFA_ERR_INVALID;
    }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    if (batch == 0 || M == 0 || N == 0) return GFA_OK;
    if (!out || (K > 0 && (!a || !b))) { set_error("gfa_matmul: bad arguments"); return GFA_ERR_INVALID; }
    if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30)) { set_error("gfa_matmul: dimension too large"); return GFA_ERR_UNSUPPORTED; }
    static const int itemsize[4] = {1, 2, 4, 8};
    const size_t isz = (size_t)itemsize[dtype];
    if (K == 0) { // empty sum: zeros
        GFA_HIP(hipMemsetAsync(out, 0, (size_t)(batch * M * N) * isz, (hipStream_t)stream));
        return GFA_OK;
    }
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    // the batch rides on gridDim.z (<= 65535): longer batches go out in slices
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
        const char *pa = (const char *)a + (size_t)(b0 * a_batch_stride) * isz;
        const char *pb = (const char *)b + (size_t)(b0 * b_batch_stride) * isz;
        char *po = (char *)out + (size_t)(b0 * M * N) * isz;
        if (matmul_mfma_eligible(f->calc, M, K, N)) {
            rc = matmul_mfma(f->calc, dtype, pa, pb, po, nb, M, K, N, a_batch_stride, b_batch_stride, (hipStream_t)stream);
        } else if (f->has_tab8 && f->calc.p == 2 && dtype == GFA_U8 && f->use_lookup()) {
            static bool attr = false;
            const size_t lds = 65536 + 2 * MM_BK * (MM_BM + 4);
            if (!attr) { GFA_HIP(hipFuncSetAttribute((const void *)matmul_tab8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
            const dim3 grid((unsigned)((N + MM_BN - 1) / MM_BN), (unsigned)((M + MM_BM - 1) / MM_BM), (unsigned)nb);
            hipLaunchKernelGGL(matmul_tab8_kernel, grid, dim3(MM_THREADS), lds, (hipStream_t)stream, ds->mul8, (const uint8_t *)pa,
                               (const uint8_t *)pb, (uint8_t *)po, (int)M, (int)K, (int)N, (i64)a_batch_stride, (i64)b_batch_stride);
            GFA_HIP(hipGetLastError());
        } else if (f->use_lookup() && f->calc.m > 1) { // prime fields: integer products + lazy reduction, not log/exp gathers
            rc = dispatch_matmul(f->lut_desc(*ds), dtype, pa, pb, po, nb, M, K, N, a_batch_stride, b_batch_stride, (hipStream_t)stream);
        } else {
            rc = dispatch_matmul(f->calc, dtype, pa, pb, po, nb, M, K, N, a_batch_stride, b_batch_stride, (hipStream_t)stream);
        }
        if (rc) return rc;
    }
    return GFA_OK;
}

int gfa_row_reduce(gfa_field_t *f, void *a, int64_t batch, int64_t m, int64_t n, int64_t ncols, int64_t *rank_out, int dtype,
                   gfa_stream_t stream)
{
    if (!f || batch < 0 || m < 0 || n < 0 || ncols < 0 || ncols > n) { set_error("gfa_row_reduce: bad arguments"); return GFA_ERR_INVALID; }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    if (batch == 0) return GFA_OK;
    if (!rank_out || ((m > 0 && n > 0) && !a)) { set_error("gfa_row_reduce: bad arguments"); return GFA_ERR_INVALID; }
    if ((m > GJ_MAX_ROWS && !(m * n >= 131072 && batch <= 32)) || n > (1 << 24) || m > (1 << 24)) {
        set_error("gfa_row_reduce: stacks of matrices are limited to 4096 rows each");
        return GFA_ERR_UNSUPPORTED;
    }
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    if (m == 0 || n == 0) {
        GFA_HIP(hipMemsetAsync(rank_out, 0, sizeof(int64_t) * (size_t)batch, (hipStream_t)stream));
        return GFA_OK;
    }
    const FieldDev fd = f->use_lookup() ? f->lut_desc(*ds) : f->calc;
    // few large matrices: two kernels per column with the update spread over all CUs; otherwise one workgroup per matrix
    if (m * n >= 131072 && batch <= 32 && batch <= 65535)
        return dispatch_row_reduce_wide(fd, dtype, a, batch, m, n, ncols, (i64 *)rank_out, (hipStream_t)stream);
    return dispatch_row_reduce(fd, dtype, a, batch, m, n, ncols, (i64 *)rank_out, (hipStream_t)stream);
}

int gfa_plu_decompose(gfa_field_t *f, void *a, void *l_out, void *p_out, int64_t batch, int64_t m, int64_t n, int pivoting,
                      int64_t *n_permutations_out, void *det_out, int dtype, gfa_stream_t stream, int32_t *dev_err)
{
    if (!f || !a || batch < 0 || m < 1 || n < 1) { set_error("gfa_plu_decompose: bad arguments"); return GFA_ERR_INVALID; }
    if (!dtype_holds(dtype, f->calc.q)) { set_error("dtype cannot hold the field's elements"); return GFA_ERR_INVALID; }
    if (!pivoting && m - 1 > n) { set_error("gfa_plu_decompose: LU without pivoting needs m - 1 <= n"); return GFA_ERR_INVALID; }
    if (batch == 0) return GFA_OK;
    if (m > GJ_MAX_ROWS || n > (1 << 24)) { set_error("gfa_plu_decompose: matrix too large (at most 4096 rows)"); return GFA_ERR_UNSUPPORTED; }
    FieldDeviceState *ds;
    int rc = f->ensure_device(nullptr, &ds);
    if (rc) return rc;
    if (f->use_lookup())
        return dispatch_plu(f->lut_desc(*ds), dtype, pivoting != 0, a, l_out, p_out, batch, m, n, (i64 *)n_permutations_out, det_out,
                            dev_err, (hipStream_t)stream);
    return dispatch_plu(f->calc, dtype, pivoting != 0, a, l_out, p_out, batch, m, n, (i64 *)n_permutations_out, det_out, dev_err,
                        (hipStream_t)stream);
}

int gfa_time_matmul(gfa_field_t *f, const void *a, const void *b, void *out, int64_t batch, int64_t M, int64_t K, int64_t N,
                    int dtype, gfa_stream_t stream, int iters, float *ms_out)
{
    return gfa::time_loop((hipStream_t)stream, iters, ms_out,
                          [&]() { return gfa_matmul(f, a, b, out, batch, M, K, N, M * K, K * N, dtype, stream); });
}

} // extern "C"
