// How fast does the memory system move a strided tile in PIECES of 16 / 32 / 64 / 128 bytes?  (r06)
//
// The three-pass transforms of 2^21 .. 2^28 points (and Goldilocks at every length above 2^10) are three passes because a strided
// pass wants its columns in pieces of at least 64 bytes (16 columns x 4 B, 8 x 8 B): with 2^13-point lines only 8 (4) columns fit
// one workgroup's registers, i.e. 32-byte pieces.  Whether a two-pass form could pay depends on what 32-byte pieces cost; this
// skeleton measures it: an array of ROWS x PITCH bytes is read (and written back to a second array) tile by tile, a tile = all ROWS
// rows x PIECE bytes, one persistent workgroup per CU walking over tiles; lanes cover PIECE / 4 consecutive words of 256 / PIECE
// ... 64 * 4 / PIECE consecutive rows per instruction.  No arithmetic.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/strided_pieces.hip -o _variants/strided_pieces && _variants/strided_pieces
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)

// PIECE bytes per row and tile; ROWS rows of pitch `pitch` bytes; tiles = pitch / PIECE.  Each thread moves UNR words per iteration.
template <int PIECE, int UNR>
__global__ __launch_bounds__(1024) void piece_copy(const unsigned *__restrict__ in, unsigned *__restrict__ out, long rows, long pitch_words, long tiles)
{
    constexpr int WPR = PIECE / 4;     // words (lanes) per row piece
    constexpr int RPI = 1024 / WPR;    // rows per workgroup-wide instruction
    const int tid = threadIdx.x;
    const int col = tid % WPR, row0 = tid / WPR;
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const unsigned *src = in + t * WPR + col;
        unsigned *dst = out + t * WPR + col;
        for (long r = row0; r < rows; r += (long)RPI * UNR) {
            unsigned v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) v[u] = __builtin_nontemporal_load(src + (r + (long)u * RPI) * pitch_words);
#pragma unroll
            for (int u = 0; u < UNR; u++) __builtin_nontemporal_store(v[u], dst + (r + (long)u * RPI) * pitch_words);
        }
    }
}

template <int PIECE>
static void run(const unsigned *in, unsigned *out, long rows, long pitch_words, int cus)
{
    const long tiles = pitch_words * 4 / PIECE;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((piece_copy<PIECE, 16>), dim3(cus * 2), dim3(1024), 0, 0, in, out, rows, pitch_words, tiles);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(e0, 0));
    const int it = 10;
    for (int i = 0; i < it; i++) hipLaunchKernelGGL((piece_copy<PIECE, 16>), dim3(cus * 2), dim3(1024), 0, 0, in, out, rows, pitch_words, tiles);
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= it;
    const double bytes = 2.0 * rows * pitch_words * 4;
    printf("piece %4d B: %.4f ms  %.0f GB/s moved (read + write)  %.3f of 8 TB/s\n", PIECE, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000);
}

int main(int argc, char **argv)
{
    const long rows = argc > 1 ? atol(argv[1]) : 8192;          // 2^13-point lines
    const long pitch_words = argc > 2 ? atol(argv[2]) : 8192;   // 2^13 columns of 4 bytes: a 2^26-point array, 256 MiB
    HIPCHK(hipSetDevice(0));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, 0));
    unsigned *in, *out;
    HIPCHK(hipMalloc((void **)&in, rows * pitch_words * 4));
    HIPCHK(hipMalloc((void **)&out, rows * pitch_words * 4));
    HIPCHK(hipMemset(in, 1, rows * pitch_words * 4));
    printf("array %ld rows x %ld words (%.0f MiB), %d CUs\n", rows, pitch_words, rows * pitch_words * 4 / 1048576.0, prop.multiProcessorCount);
    run<16>(in, out, rows, pitch_words, prop.multiProcessorCount);
    run<32>(in, out, rows, pitch_words, prop.multiProcessorCount);
    run<64>(in, out, rows, pitch_words, prop.multiProcessorCount);
    run<128>(in, out, rows, pitch_words, prop.multiProcessorCount);
    run<256>(in, out, rows, pitch_words, prop.multiProcessorCount);
    return 0;
}
