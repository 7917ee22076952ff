// Host checks of galois_amd/csrc/gfa_rs_host.h (no device needed):
//   * rs_lfsr_rows: a host model of rs_lfsr_kernel's state handling -- consecutive words for (n - k) % 16 != 0, the PLANAR state
//     (plane p = bytes p, p + 4, ...; per symbol one plane is shifted, the others are renamed) otherwise -- run on the table and
//     compared with the schoolbook remainder of m(x) x^(n-k) modulo g(x), for n - k = 4 .. 64;
//   * rs_decode_lane_tables: positions, the root <-> lane bijection, idle lanes, and the point of it all -- within a half-wave
//     no two DIFFERENT roots share an LDS bank (bits 2..6 of the root) unless more than two roots of the code do.
#include "gfa_rs_host.h"
#include <cstdio>
#include <set>
using namespace gfa;

static uint8_t MUL[65536];
static void build_field(uint32_t poly, int m)
{
    const uint32_t q = 1u << m;
    for (uint32_t a = 0; a < 256; a++)
        for (uint32_t b = 0; b < 256; b++) {
            uint32_t r = 0, aa = a, bb = b;
            if (a < q && b < q)
                while (bb) { if (bb & 1) r ^= aa; bb >>= 1; aa <<= 1; if (aa & q) aa ^= poly; }
            MUL[(a << 8) | b] = (uint8_t)r;
        }
}
static uint32_t powm(uint32_t a, int e) { uint32_t r = 1; while (e-- > 0) r = MUL[(r << 8) | a]; return r; }

// generator polynomial prod (x - alpha^(c+j)), j < nk, highest degree first
static std::vector<uint64_t> genpoly(uint32_t alpha, int c, int nk, std::vector<uint64_t> *roots)
{
    std::vector<uint64_t> g{1};
    for (int j = 0; j < nk; j++) {
        const uint32_t r = powm(alpha, c + j);
        if (roots) roots->push_back(r);
        std::vector<uint64_t> h(g.size() + 1, 0);
        for (size_t i = 0; i < g.size(); i++) { h[i] ^= g[i]; h[i + 1] ^= MUL[((uint32_t)g[i] << 8) | r]; }
        g = h;
    }
    return g;
}

// the kernel's arithmetic on the row table, restated: returns the nk state bytes (index s = coefficient of x^(nk-1-s)) after `len` symbols
static std::vector<uint8_t> lfsr_model(const std::vector<uint32_t> &rows, size_t nkw, const std::vector<uint8_t> &sym, bool encode)
{
    const size_t full = nkw / 4, tail = nkw % 4;
    auto rowword = [&](uint32_t f, size_t d) { const size_t c4 = d / 4; return rows[c4 < full ? c4 * 1024 + f * 4 + (d % 4) : full * 1024 + f * tail + (d - full * 4)]; };
    std::vector<uint8_t> out(nkw * 4);
    if (nkw % 4 != 0) {
        std::vector<uint32_t> st(nkw, 0);
        for (uint8_t s : sym) {
            const uint32_t top = st[0] >> 24, f = encode ? (s ^ top) : top;
            for (size_t d = 0; d + 1 < nkw; d++) st[d] = (st[d] << 8) | (st[d + 1] >> 24);
            st[nkw - 1] = (st[nkw - 1] << 8) | (encode ? 0u : s);
            for (size_t d = 0; d < nkw; d++) st[d] ^= rowword(f, d);
        }
        for (size_t d = 0; d < nkw; d++)
            for (int b = 0; b < 4; b++) out[4 * d + b] = (uint8_t)(st[d] >> (24 - 8 * b));
        return out;
    }
    const size_t W = nkw / 4;
    std::vector<std::vector<uint32_t>> P(4, std::vector<uint32_t>(W, 0));
    size_t K = 0; // symbols taken mod 4: the plane in role r is P[(r + K) & 3]
    for (uint8_t s : sym) {
        std::vector<uint32_t> &p0 = P[K & 3];
        const uint32_t top = p0[0] >> 24, f = encode ? (s ^ top) : top;
        for (size_t h = 0; h + 1 < W; h++) p0[h] = (p0[h] << 8) | (p0[h + 1] >> 24);
        p0[W - 1] = (p0[W - 1] << 8) | (encode ? 0u : s);
        for (size_t qd = 0; qd < nkw; qd++) P[(qd / W + K + 1) & 3][qd % W] ^= rowword(f, qd);
        K++;
    }
    for (size_t s = 0; s < nkw * 4; s++) { // state byte s: plane s % 4 (in role order), byte s / 4 of the plane
        const std::vector<uint32_t> &pl = P[(s % 4 + K) & 3];
        const size_t byte = s / 4;
        out[s] = (uint8_t)(pl[byte / 4] >> (24 - 8 * (byte % 4)));
    }
    return out;
}

int main()
{
    int fails = 0;
    build_field(0x11D, 8);
    uint64_t x = 99;
    auto rnd = [&]() { x = x * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(x >> 33); };
    for (int nk = 4; nk <= 64; nk += 4) {
        for (int c = 0; c < 2; c++) {
            std::vector<uint64_t> roots;
            const std::vector<uint64_t> g = genpoly(2, c, nk, &roots);
            const std::vector<uint32_t> rows = rs_lfsr_rows(MUL, 256, g, (size_t)nk);
            for (int trial = 0; trial < 20; trial++) {
                const int len = 1 + (int)(rnd() % (255 - nk));
                std::vector<uint8_t> msg(len);
                for (auto &v : msg) v = (uint8_t)rnd();
                // schoolbook: remainder of m(x) x^nk modulo g(x)
                std::vector<uint8_t> work(msg);
                work.resize(len + nk, 0);
                for (int i = 0; i < len; i++) {
                    const uint32_t f = work[i];
                    if (f)
                        for (int j = 1; j <= nk; j++) work[i + j] ^= MUL[(f << 8) | (uint32_t)g[j]];
                }
                const std::vector<uint8_t> par = lfsr_model(rows, (size_t)nk / 4, msg, true);
                for (int s = 0; s < nk; s++)
                    if (par[s] != work[len + s]) { fails++; break; }
                // decoder pre-pass form: the remainder of the whole codeword is zero, of a corrupted one it is not
                std::vector<uint8_t> cw(msg);
                cw.insert(cw.end(), work.begin() + len, work.end());
                std::vector<uint8_t> rem = lfsr_model(rows, (size_t)nk / 4, cw, false);
                for (uint8_t v : rem) if (v) { fails++; break; }
                cw[rnd() % cw.size()] ^= (uint8_t)(1 + rnd() % 255);
                rem = lfsr_model(rows, (size_t)nk / 4, cw, false);
                bool nz = false;
                for (uint8_t v : rem) nz |= v != 0;
                if (!nz) fails++;
            }
            if (nk > 60) continue;
            // lane tables
            const std::vector<uint8_t> aux = rs_decode_lane_tables(MUL, 256, 2, 255, roots);
            std::set<int> lanes;
            for (int j = 0; j < nk; j++) {
                const int l = aux[64 + j];
                if (l > 63 || aux[l] != roots[j] || !lanes.insert(l).second) fails++;
            }
            std::vector<int> per_bank(32, 0);
            for (uint64_t r : roots) per_bank[(r >> 2) & 31]++;
            for (int h = 0; h < 2; h++) {
                std::set<int> banks;
                for (int l = 32 * h; l < 32 * h + 32; l++) {
                    if (!lanes.count(l)) { // idle lane: a copy of a root of its own half-wave (or 0 when the half is empty)
                        bool ok = aux[l] == 0;
                        for (int l2 = 32 * h; l2 < 32 * h + 32; l2++) ok |= lanes.count(l2) && aux[l2] == aux[l];
                        if (!ok) fails++;
                        continue;
                    }
                    const int bank = (aux[l] >> 2) & 31;
                    if (!banks.insert(bank).second && per_bank[bank] <= 2) fails++; // two roots of one bank in one half although they could be split
                }
            }
            for (int i = 0; i < 255; i++)
                if (aux[128 + powm(powm(2, 254), i)] != i) fails++; // alpha^-i = (alpha^254)^i
            if (aux[128 + 0] != 255) fails++;
        }
    }
    // a subgroup code: n = 51 over GF(2^8), alpha = 2^5 -- only the 51 powers of alpha^-1 are positions
    {
        std::vector<uint64_t> roots;
        const uint32_t alpha = powm(2, 5);
        (void)genpoly(alpha, 1, 32, &roots);
        const std::vector<uint8_t> aux = rs_decode_lane_tables(MUL, 256, alpha, 51, roots);
        int npos = 0;
        for (int v = 0; v < 256; v++) npos += aux[128 + v] != 255;
        if (npos != 51) fails++;
        for (int i = 0; i < 51; i++)
            if (aux[128 + powm(powm(alpha, 50), i)] != i) fails++;
    }
    printf("fails %d\n", fails);
    return fails != 0;
}
