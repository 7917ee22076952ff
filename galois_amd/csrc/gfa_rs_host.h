// gfa_rs_host.h -- host-side table builders of the byte-field Reed-Solomon / BCH kernels (gfa_rs.hip).  Plain C++ (no HIP), so
// that tests/csrc/rs_host_test.cpp can check them without a device: the LFSR row table in the order rs_lfsr_kernel reads it and
// the lane tables of rs_decode_bin_kernel.  mul8 is the field's 256 x 256 product table (row a, column b at (a << 8) | b).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace gfa {

// State byte that byte b (0 = top) of row word d multiplies.  Consecutive order: 4d + b.  Planar order (nkw % 4 == 0, the planar
// LFSR of rs_lfsr_kernel): word d is word d % W of plane d / W (W = nkw / 4 words per plane) and plane p holds the state bytes
// p, p + 4, p + 8, ..., so its byte b is state byte p + 4 (4 (d % W) + b).
inline size_t rs_lfsr_state_byte(size_t nkw, size_t d, int b)
{
    if (nkw % 4 != 0) return 4 * d + (size_t)b;
    const size_t W = nkw / 4;
    return d / W + 4 * (4 * (d % W) + (size_t)b);
}

// 256 rows x nkw words: word d of row f packs f * g_{1 + s} for the four state bytes s of that word, first in the top byte (g =
// generator polynomial, highest degree first, monic, nk + 1 coefficients).  Chunked layout: full 4-word chunks first, chunk c of
// row f at c * 1024 + f * 4; the remaining nkw % 4 words of row f at (nkw / 4) * 1024 + f * (nkw % 4).
inline std::vector<uint32_t> rs_lfsr_rows(const uint8_t *mul8, uint64_t q, const std::vector<uint64_t> &gpoly, size_t nk)
{
    const size_t nkw = nk / 4, full = nkw / 4, tail = nkw % 4;
    std::vector<uint32_t> rows(256 * nkw);
    for (uint32_t fb = 0; fb < 256; fb++)
        for (size_t d = 0; d < nkw; d++) {
            uint32_t w = 0;
            for (int b = 0; b < 4; b++) {
                const uint64_t gc = gpoly[1 + rs_lfsr_state_byte(nkw, d, b)];
                w = (w << 8) | (fb < q && gc < q ? mul8[(fb << 8) | gc] : 0);
            }
            const size_t c4 = d / 4;
            rows[c4 < full ? c4 * 1024 + fb * 4 + (d % 4) : full * 1024 + fb * tail + (d - full * 4)] = w;
        }
    return rows;
}

// rs_decode_bin_kernel's lane tables, 384 bytes.  [128 + x]: the position i < n with alpha^-i = x (255: none).  [lane]: the
// syndrome root that lane evaluates, [64 + j]: the lane that evaluates root j (at most 64 roots; more: the tables stay zero and
// the kernel is not used).  The LDS bank of a Horner gather is bits 2..6 of the root, so the roots are dealt out one per bank and
// half-wave as far as they allow; idle lanes repeat a root of their own half-wave (same address as its owner: a broadcast).
inline std::vector<uint8_t> rs_decode_lane_tables(const uint8_t *mul8, uint64_t q, uint64_t alpha, int64_t n, const std::vector<uint64_t> &roots)
{
    std::vector<uint8_t> aux(384, 0);
    std::fill(aux.begin() + 128, aux.end(), (uint8_t)255);
    {
        uint32_t ainv = 1;
        for (uint32_t y = 1; y < q; y++)
            if (mul8[((uint32_t)alpha << 8) | y] == 1) { ainv = y; break; }
        uint32_t x = 1;
        for (int64_t i = 0; i < n && i < 255; i++) {
            if (aux[128 + x] == 255) aux[128 + x] = (uint8_t)i;
            x = mul8[(x << 8) | ainv];
        }
    }
    if (roots.size() <= 64) {
        bool used[2][32] = {};
        int filled[2] = {0, 0};
        int lane_of[64];
        std::vector<int> later;
        for (size_t j = 0; j < roots.size(); j++) {
            const int bank = (int)((roots[j] >> 2) & 31);
            const int h = !used[0][bank] && filled[0] < 32 ? 0 : (!used[1][bank] && filled[1] < 32 ? 1 : -1);
            if (h < 0) { later.push_back((int)j); continue; }
            used[h][bank] = true;
            lane_of[j] = 32 * h + filled[h]++;
        }
        for (int j : later) { // a third root on one bank: any free lane
            const int h = filled[0] <= filled[1] && filled[0] < 32 ? 0 : 1;
            lane_of[j] = 32 * h + filled[h]++;
        }
        for (size_t j = 0; j < roots.size(); j++) { aux[lane_of[j]] = (uint8_t)roots[j]; aux[64 + j] = (uint8_t)lane_of[j]; }
        for (int h = 0; h < 2; h++)
            for (int l = filled[h]; l < 32; l++) aux[32 * h + l] = filled[h] ? aux[32 * h] : (uint8_t)0;
    }
    return aux;
}

} // namespace gfa
