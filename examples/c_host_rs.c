/* A plain C99 host driving the library through include/galois_amd.h only -- no Python, no torch, no C++:
 * GF(2^8) multiply on 2^20 bytes, then RS(255,223) encode -> corrupt 16 symbols -> decode on 4096 codewords.
 * This is what a cgo / JNI / Rust-FFI binding of the reference's hot path would look like from the other side.
 *
 * Build (tests/test_host_logic.py does exactly this; the GPU suite also runs it):
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_host_rs.c \
 *       -Lgalois_amd -lgalois_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/galois_amd -Wl,-rpath,/opt/rocm/lib -o c_host_rs
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "galois_amd.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != GFA_OK) {                                                         \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, gfa_last_error());   \
            return 1;                                                                \
        }                                                                            \
    } while (0)
#define HIPCHECK(call)                                                               \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));        \
            return 1;                                                                \
        }                                                                            \
    } while (0)

/* the reference's GF(2^8) multiply (shift-and-xor, _calculate.py:288-324) for the spot check */
static uint8_t gf256_mul(uint8_t a, uint8_t b)
{
    unsigned r = 0, x = a;
    for (int i = 0; i < 8; i++) {
        if (b & (1u << i)) r ^= x;
        x <<= 1;
        if (x & 0x100) x ^= 0x11D;
    }
    return (uint8_t)r;
}

int main(void)
{
    /* x^8 + x^4 + x^3 + x^2 + 1, highest degree first; primitive element 2 (galois.ReedSolomon's default field) */
    const uint64_t irr[9] = {1, 0, 0, 0, 1, 1, 1, 0, 1};
    gfa_field_t *f = NULL;
    CHECK(gfa_field_create(2, 8, irr, 2, &f));

    const int64_t n = 1 << 20;
    uint8_t *ha = malloc(n), *hb = malloc(n), *ho = malloc(n);
    for (int64_t i = 0; i < n; i++) { ha[i] = (uint8_t)(i * 131 + 7); hb[i] = (uint8_t)(i * 29 + (i >> 8)); }
    uint8_t *da, *db, *dout;
    HIPCHECK(hipMalloc((void **)&da, n)); HIPCHECK(hipMalloc((void **)&db, n)); HIPCHECK(hipMalloc((void **)&dout, n));
    HIPCHECK(hipMemcpy(da, ha, n, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(db, hb, n, hipMemcpyHostToDevice));
    int32_t *derr;
    HIPCHECK(hipMalloc((void **)&derr, sizeof(int32_t))); HIPCHECK(hipMemset(derr, 0, sizeof(int32_t)));
    CHECK(gfa_binary(f, GFA_OP_MUL, da, 1, db, 1, dout, n, GFA_U8, NULL, derr));
    HIPCHECK(hipMemcpy(ho, dout, n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++)
        if (ho[i] != gf256_mul(ha[i], hb[i])) { fprintf(stderr, "multiply mismatch at %lld\n", (long long)i); return 1; }

    gfa_rs_t *rs = NULL;
    CHECK(gfa_rs_create(f, 255, 223, 1, 2, 1, &rs));
    const int64_t B = 4096;
    uint8_t *hm = malloc(B * 223), *hc = malloc(B * 255), *hr = malloc(B * 255), *hd = malloc(B * 255);
    int64_t *hn = malloc(B * sizeof(int64_t));
    for (int64_t i = 0; i < B * 223; i++) hm[i] = (uint8_t)(i * 2654435761u >> 13);
    uint8_t *dm, *dc, *dd;
    int64_t *dn;
    HIPCHECK(hipMalloc((void **)&dm, B * 223)); HIPCHECK(hipMalloc((void **)&dc, B * 255)); HIPCHECK(hipMalloc((void **)&dd, B * 255));
    HIPCHECK(hipMalloc((void **)&dn, B * sizeof(int64_t)));
    HIPCHECK(hipMemcpy(dm, hm, B * 223, hipMemcpyHostToDevice));
    CHECK(gfa_rs_encode(rs, dm, 223, dc, B, 0, GFA_U8, NULL));
    HIPCHECK(hipMemcpy(hc, dc, B * 255, hipMemcpyDeviceToHost));
    memcpy(hr, hc, B * 255);
    for (int64_t i = 0; i < B; i++)
        for (int e = 0; e < 16; e++) hr[i * 255 + (i + 15 * e) % 255] ^= (uint8_t)(1 + (i + e) % 255); /* 16 distinct positions */
    HIPCHECK(hipMemcpy(dc, hr, B * 255, hipMemcpyHostToDevice));
    CHECK(gfa_rs_decode(rs, dc, NULL, 255, dd, dn, B, GFA_U8, NULL));
    HIPCHECK(hipMemcpy(hd, dd, B * 255, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(hn, dn, B * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (memcmp(hd, hc, B * 255) != 0) { fprintf(stderr, "decode did not restore the codewords\n"); return 1; }
    for (int64_t i = 0; i < B; i++)
        if (hn[i] != 16) { fprintf(stderr, "codeword %lld: n_errors = %lld\n", (long long)i, (long long)hn[i]); return 1; }
    /* systematic: the message is the first 223 symbols */
    for (int64_t i = 0; i < B; i++)
        if (memcmp(hd + i * 255, hm + i * 223, 223) != 0) { fprintf(stderr, "message mismatch\n"); return 1; }

    gfa_rs_destroy(rs);
    gfa_field_destroy(f);
    printf("c_host_rs: GF(2^8) multiply of %lld bytes and RS(255,223) round trip of %lld codewords (16 errors each) OK\n",
           (long long)n, (long long)B);
    return 0;
}
