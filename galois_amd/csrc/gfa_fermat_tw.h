// gfa_fermat_tw.h -- the first inter-network twiddles w^(m k0) of the one-pass GF(65537) kernel, formed in registers (r06).
//
// Rounds 3-5 streamed them from a 256 KiB table (every workgroup re-read it for every transform: as many L2 -> CU bytes as the
// data itself, in the same in-order memory queue as the data).  Wave stamps (profiles/r06_fermat_varA_wave_stamps.txt) put
// 2-3.5 us of every ~28 us round into waiting for those loads behind the HBM traffic, so they are computed instead:
//     k0 = 8 kh + kl,   w^(m k0) = A[kh] * B[kl],   A[i] = (w^(8m))^i,  B[j] = (w^m)^j
// from two per-thread seeds w^m and w^(8m) (loaded once per kernel).  12 "tight" products build the two progressions, 49 "loose"
// ones combine them; 14 twiddles are table entries themselves.  +195 vector instructions per 64 points, no memory traffic.
//
// Ranges (replayed on a range-checking integer by tests/csrc/fermat_tw_host_test.cpp):
//   seeds: balanced residues, |s| <= 32768;
//   fm_tw_tight(x, y), |x|, |y| <= 32770:  |x y| < 2^31, fold -> [-32767, 98303], bfold -> |.| <= 32769;
//   fm_tw_loose(x, y) = bfold(x y):        |x y| <= 32769^2 < 2^31 -> |.| <= 32768 + 16385 = 49153;
//   fm_tw_apply(q, T) = fold(bfold(q) * T): |q| < 2^29 (network output) -> |bfold(q)| <= 40961, 40961 * 49153 < 2^31.
// Needs fm_mulc / fm_fold / fm_bfold (gfa_ntt_fermat.hip, or the host model of the test).
#pragma once

#ifndef FM_TW_FN
#define FM_TW_FN __device__ __forceinline__
#endif

template <typename V>
FM_TW_FN V fm_tw_tight(V x, V y) { return fm_bfold(fm_fold(fm_mulc(x, y))); }
template <typename V>
FM_TW_FN V fm_tw_loose(V x, V y) { return fm_bfold(fm_mulc(x, y)); }
template <typename V>
FM_TW_FN V fm_tw_apply(V q, V t) { return fm_fold(fm_mulc(fm_bfold(q), t)); }

// A[1..7], B[1..7] from the seeds (index 0 unused: the factor 1)
template <typename V>
FM_TW_FN void fm_tw_progressions(V seed1, V seed8, V (&A)[8], V (&B)[8])
{
    A[1] = seed8;
    B[1] = seed1;
#pragma unroll
    for (int i = 2; i < 8; i++) {
        A[i] = fm_tw_tight(A[i - 1], seed8);
        B[i] = fm_tw_tight(B[i - 1], seed1);
    }
}
// the twiddle of output k0 (1 <= k0 < 64)
template <typename V>
FM_TW_FN V fm_tw_of(const V (&A)[8], const V (&B)[8], int k0)
{
    const int kh = k0 >> 3, kl = k0 & 7;
    return kh == 0 ? B[kl] : kl == 0 ? A[kh] : fm_tw_loose(A[kh], B[kl]);
}
