// Host model of the Berlekamp-Massey arrangement of rs_decode_bin_kernel (galois_amd/csrc/gfa_rs.hip, bm_run): the inversionless
// RiBM recurrence on 64 lanes in a frame that moves down one lane per step -- products of Lambda * S in lanes 0..31 (the discrepancy
// is lane 0), the locator starting at lane 63 -- with the two special cases the kernel has: the product half of Y cleared before the
// 32nd step of a 32-step run (Lambda_0 lands on lane 31 there), and the early stop on a zero discrepancy with nothing but zeros
// ahead.  Checked against the textbook algorithm with divisions (Massey 1969) on random syndrome sequences of every length
// 1..32: from error patterns within and beyond the correction radius, and from arbitrary random sequences.
// The model restates the assembly loop statement by statement; the loop itself is pinned by the GPU parity tests.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static uint8_t MUL[65536], INV[256];
static void build_field()
{
    for (uint32_t a = 0; a < 256; a++)
        for (uint32_t b = 0; b < 256; b++) {
            uint32_t r = 0, aa = a, bb = b;
            while (bb) { if (bb & 1) r ^= aa; bb >>= 1; aa <<= 1; if (aa & 256) aa ^= 0x11D; }
            MUL[(a << 8) | b] = (uint8_t)r;
            if (r == 1) INV[a] = (uint8_t)b;
        }
}
static inline uint32_t mul(uint32_t a, uint32_t b) { return MUL[(a << 8) | b]; }

// the kernel's arrangement: returns L, writes Lambda (a non-zero multiple of the connection polynomial) to lam[0..32]
static int bm_frame(const uint8_t *S, int nsq, uint8_t *lam, int *steps_taken)
{
    uint32_t X[64], Y[64];
    for (int l = 0; l < 64; l++) X[l] = l == 63 ? 1u : (l < nsq ? S[l] : 0u);
    memcpy(Y, X, sizeof X);
    uint32_t gamma = 1;
    int L = 0, r = 0;
    auto run = [&](int limit) {
        while (r < limit) {
            const uint32_t d0 = X[0];
            if (d0 == 0) {
                bool ahead = false;
                for (int l = 0; l < nsq - r; l++) ahead |= X[l] != 0;
                if (!ahead) return;                      // BM_END: nothing but zero discrepancies to come
                for (int l = 0; l < 63; l++) X[l] = X[l + 1];
                X[63] = 0;                               // wave_shl:1 with bound_ctrl
                r++;
                continue;
            }
            uint32_t A[64];
            for (int l = 0; l < 63; l++) A[l] = X[l + 1];
            A[63] = 0;
            uint32_t Xn[64];
            for (int l = 0; l < 64; l++) Xn[l] = mul(gamma, A[l]) ^ mul(d0, Y[l]);
            if (!(2 * L > r)) { memcpy(Y, A, sizeof A); gamma = d0; L = r + 1 - L; }
            memcpy(X, Xn, sizeof X);
            r++;
        }
    };
    run(nsq < 31 ? nsq : 31);
    if (r == 31 && nsq == 32) {
        for (int l = 0; l < 32; l++) Y[l] = 0;
        run(32);
    }
    for (int i = 0; i <= 32; i++) lam[i] = (i < 32 && i <= r && 63 - r + i <= 63) ? (uint8_t)X[63 - r + i] : 0;
    *steps_taken = r;
    return L;
}

// Massey's algorithm with divisions: C monic (C[0] = 1), returns L
static int bm_textbook(const uint8_t *S, int n, uint8_t *C)
{
    uint8_t B[64] = {0}, T[64];
    memset(C, 0, 64);
    C[0] = 1; B[0] = 1;
    int L = 0, m = 1;
    uint32_t b = 1;
    for (int k = 0; k < n; k++) {
        uint32_t d = S[k];
        for (int i = 1; i <= L; i++) d ^= mul(C[i], S[k - i]);
        if (d == 0) { m++; continue; }
        const uint32_t coef = mul(d, INV[b]);
        memcpy(T, C, 64);
        for (int i = 0; i + m < 64; i++) C[i + m] ^= (uint8_t)mul(coef, B[i]);
        if (2 * L <= k) { L = k + 1 - L; memcpy(B, T, 64); b = d; m = 1; } else m++;
    }
    return L;
}

int main()
{
    build_field();
    int fails = 0, stopped_early = 0, full32 = 0;
    uint64_t x = 4242;
    auto rnd = [&]() { x = x * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(x >> 33); };
    uint32_t alpha_pow[255];
    alpha_pow[0] = 1;
    for (int i = 1; i < 255; i++) alpha_pow[i] = mul(alpha_pow[i - 1], 2);
    for (int trial = 0; trial < 400000; trial++) {
        int nsq = 1 + (int)(rnd() % 32);
        uint8_t S[32] = {0};
        const int kind = (int)(rnd() % 8);
        if (trial < 2 * 32 * 32) { // a single non-zero syndrome (a random value, then 1) at every place of every length: the first
                                   // discrepancy arrives late, and with it at the very end of a 32-step run the product half of Y is
                                   // still the syndromes themselves (S_31 = 1 alone would cancel Lambda_0 without the clearing)
            nsq = 1 + (trial % 1024) / 32;
            if (trial % 32 < nsq) S[trial % 32] = trial < 1024 ? (uint8_t)(1 + rnd() % 255) : (uint8_t)1;
        } else if (kind == 0) {
            for (int j = 0; j < nsq; j++) S[j] = (uint8_t)rnd(); // an arbitrary sequence
        } else {
            const int nerr = kind < 6 ? (int)(rnd() % (nsq / 2 + 1)) : (int)(rnd() % (nsq + 3)); // mostly within the radius
            for (int e = 0; e < nerr; e++) {
                const uint32_t pos = rnd() % 255, val = 1 + rnd() % 255;
                for (int j = 0; j < nsq; j++) S[j] ^= (uint8_t)mul(val, alpha_pow[(pos * (uint32_t)(j + 1)) % 255]);
            }
        }
        uint8_t lam[33], C[64];
        int steps;
        const int L1 = bm_frame(S, nsq, lam, &steps), L2 = bm_textbook(S, nsq, C);
        if (steps < nsq) stopped_early++;
        if (steps == 32) full32++;
        if (L1 != L2 || lam[0] == 0) { fails++; continue; }
        const int clen = L1 + 1 < nsq ? L1 + 1 : nsq; // what the kernel reads (_lfsr.py keeps L + 1 coefficients; the kernel has nsq lanes)
        const uint32_t s = INV[lam[0]];
        for (int i = 0; i < clen; i++)
            if (mul(s, lam[i]) != C[i]) { fails++; break; }
    }
    printf("fails %d (early stops %d, 32-step runs %d)\n", fails, stopped_early, full32);
    return fails != 0 || stopped_early == 0 || full32 == 0;
}
