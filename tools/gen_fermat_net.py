"""Generates galois_amd/csrc/gfa_fermat_nets.inc: fully unrolled in-register DFT networks over GF(65537).

In GF(2^16 + 1) the element 2 has order 32 and sqrt(2) = 2^12 - 2^4 has order 64, so every twiddle INSIDE a radix-32 /
radix-64 decimation-in-frequency network is a power of two (a shift) or 2^i * sqrt(2) (one multiplication by a small
constant); only the twiddles BETWEEN networks are general field products.  Values are kept as loose signed 32-bit
representatives; a `fold` (x mod 2^16) - (x >> 16) -- one v_sub_u32_sdwa -- brings any int32 back to about 17 bits.
This script tracks an exact interval for every register and inserts a fold only where the next operation could leave
the signed 32-bit range, so the emitted code is overflow-free by construction; `--check` replays the emitted operation
list in Python on random and extreme inputs against a direct DFT.

The networks use the CANONICAL roots z64 = 4080 (z64^2 = 2).  A transform whose root of unity is z64^u for another odd u
feeds its inputs in the order a' = u*a mod R (see gfa_ntt_fermat.hip), so one set of networks serves every omega.

Emitted per network: `__device__ __forceinline__ void NAME(int (&v)[R])`, in place, outputs in bit-reversed positions,
plus `NAME_OUT_MAX` = the largest magnitude an output can have (the caller's range contract).
"""
from __future__ import annotations

import argparse
import os
import random

P = 65537
Z64 = 4080  # sqrt(2) = 2^12 - 2^4, order 64
I32_MAX = (1 << 31) - 1
I32_MIN = -(1 << 31)


def balanced(c: int) -> int:
    c %= P
    return c - P if c > P // 2 else c


class Gen:
    def __init__(self, R: int, in_lo: int, in_hi: int, out_max: int):
        self.R = R
        self.iv = [(in_lo, in_hi)] * R  # interval of every register
        self.ops = []  # ("add"/"sub"/"shl"/"fold"/"bfold"/"mulc"/"mov", dst, a, b)
        self.out_max = out_max
        self.ntmp = 0

    # ---- interval helpers ----
    @staticmethod
    def _fits(lo, hi):
        return I32_MIN <= lo and hi <= I32_MAX

    @staticmethod
    def fold_iv(iv):
        lo, hi = iv
        return (-(hi >> 16), 65535 - (lo >> 16))

    @staticmethod
    def bfold_iv(iv):
        lo, hi = iv
        return (-32768 - ((hi + 32768) >> 16), 32767 - ((lo + 32768) >> 16))

    def mag(self, r):
        lo, hi = self.iv[r]
        return max(-lo, hi)

    # ---- emitters (registers are ints < R for v[], strings for temporaries) ----
    def fold(self, r):
        self.ops.append(("fold", r, r, None))
        self.iv[r] = self.fold_iv(self.iv[r])

    def bfold(self, r):
        self.ops.append(("bfold", r, r, None))
        self.iv[r] = self.bfold_iv(self.iv[r])

    def shl(self, r, k):
        lo, hi = self.iv[r]
        assert self._fits(lo << k, hi << k)
        self.ops.append(("shl", r, r, k))
        self.iv[r] = (lo << k, hi << k)

    def ensure_mag(self, r, limit):
        """fold r until |r| <= limit is guaranteed"""
        if self.mag(r) <= limit:
            return
        self.fold(r)
        if self.mag(r) > limit:
            self.bfold(r)
        if self.mag(r) > limit:
            self.bfold(r)
        assert self.mag(r) <= limit, (self.iv[r], limit)

    def butterfly(self, iu, ix, e):
        """(u, x) -> (u + x, (u - x) * z64^e), 0 <= e < 32"""
        # make u + x and u - x representable
        while True:
            (ul, uh), (xl, xh) = self.iv[iu], self.iv[ix]
            if self._fits(ul + xl, uh + xh) and self._fits(ul - xh, uh - xl):
                break
            big = iu if self.mag(iu) >= self.mag(ix) else ix
            self.fold(big)
        (ul, uh), (xl, xh) = self.iv[iu], self.iv[ix]
        s_iv = (ul + xl, uh + xh)
        d_iv = (ul - xh, uh - xl)
        if e == 0:
            self.ops.append(("bfly", iu, ix, None))  # u' = u + x ; x' = u - x
            self.iv[iu], self.iv[ix] = s_iv, d_iv
            return
        if e % 2 == 0:
            k = e // 2  # * 2^k, 1 <= k <= 15
            dm = max(-d_iv[0], d_iv[1])
            if dm << k > I32_MAX:
                need = I32_MAX >> k  # |d| <= need
                if need // 2 >= 32771:
                    # reduce the operands first so that the difference is small enough for the shift
                    self.ensure_mag(iu, need // 2)
                    self.ensure_mag(ix, need // 2)
                    return self.butterfly(iu, ix, e)
                # (k = 15, or k = 14 on wide operands): split the shift around a fold
                self.butterfly(iu, ix, 0)
                k1 = k // 2
                self.ensure_mag(ix, I32_MAX >> k1)
                self.shl(ix, k1)
                self.ensure_mag(ix, I32_MAX >> (k - k1))
                self.shl(ix, k - k1)
                return
            self.ops.append(("bfly_shl", iu, ix, k))
            self.iv[iu] = s_iv
            self.iv[ix] = (d_iv[0] << k, d_iv[1] << k)
            return
        # odd exponent: general constant c = z64^e (balanced), |c| <= 32768
        c = balanced(pow(Z64, e, P))
        dm = max(-d_iv[0], d_iv[1])
        if dm * abs(c) > I32_MAX:
            need = I32_MAX // abs(c)
            self.ensure_mag(iu, need // 2)
            self.ensure_mag(ix, need // 2)
            return self.butterfly(iu, ix, e)
        self.ops.append(("bfly_mulc", iu, ix, c))
        self.iv[iu] = s_iv
        lo, hi = sorted((d_iv[0] * c, d_iv[1] * c))
        self.iv[ix] = (lo, hi)

    def network(self):
        R = self.R
        L = R.bit_length() - 1
        unit = 64 // R  # exponent of z64 per unit of w_R
        self.op_layer = []  # layer (0 = first) of every op, for split()
        for s in range(L - 1, -1, -1):
            half = 1 << s
            for b in range(0, R, 2 * half):
                for j in range(half):
                    e = (j << (L - 1 - s)) * unit
                    self.butterfly(b + j, b + j + half, e)
            self.op_layer += [L - 1 - s] * (len(self.ops) - len(self.op_layer))
        for r in range(R):
            self.ensure_mag(r, self.out_max)
        self.op_layer += [L] * (len(self.ops) - len(self.op_layer))

    def split(self, layers, classes, modulus):
        """Reorders the ops into (head, tail): head = every op of the first `layers` layers whose registers all lie in residue classes
        < `classes` modulo `modulus` (those layers pair registers of ONE class, so the head is closed under dependencies and the tail
        -- everything else, in the original order -- never feeds it).  Per-register op sequences, hence the tracked intervals, are
        unchanged.  The kernel runs the head on the rows that arrived first while the last ones are still in flight."""
        head, tail = [], []
        for op, lay in zip(self.ops, self.op_layer):
            regs = [op[1]] + ([op[2]] if op[0].startswith("bfly") else [])
            (head if lay < layers and all(r % modulus < classes for r in regs) else tail).append(op)
        return head, tail

    # ---- output ----
    def emit_cpp(self, name, ops=None, header=True):
        out = ([f"// radix-{self.R} DIF network over GF(65537), canonical roots; outputs in bit-reversed positions",
                f"static constexpr int {name.upper()}_OUT_MAX = {max(self.mag(r) for r in range(self.R))};"] if header else [])
        out += [f"__device__ __forceinline__ void {name}(int (&v)[{self.R}])", "{", "    int t;"]
        for op, a, b, k in (self.ops if ops is None else ops):
            if op == "fold":
                out.append(f"    v[{a}] = fm_fold(v[{a}]);")
            elif op == "bfold":
                out.append(f"    v[{a}] = fm_bfold(v[{a}]);")
            elif op == "shl":
                out.append(f"    v[{a}] = fm_shl(v[{a}], {k});")
            elif op == "bfly":
                out.append(f"    t = fm_sub(v[{a}], v[{b}]); v[{a}] = fm_add(v[{a}], v[{b}]); v[{b}] = t;")
            elif op == "bfly_shl":
                out.append(f"    t = fm_sub(v[{a}], v[{b}]); v[{a}] = fm_add(v[{a}], v[{b}]); v[{b}] = fm_shl(t, {k});")
            elif op == "bfly_mulc":
                out.append(f"    t = fm_sub(v[{a}], v[{b}]); v[{a}] = fm_add(v[{a}], v[{b}]); v[{b}] = fm_mulc(t, {k});")
        out.append("}")
        return "\n".join(out)

    def counts(self):
        c = {}
        for op, *_ in self.ops:
            c[op] = c.get(op, 0) + 1
        return c

    # ---- replay on Python integers with overflow checks ----
    def replay(self, x):
        def chk(v):
            assert I32_MIN <= v <= I32_MAX, v
            return v

        v = list(x)
        for op, a, b, k in self.ops:
            if op == "fold":
                v[a] = (v[a] & 0xFFFF) - (v[a] >> 16)
            elif op == "bfold":
                lo = ((v[a] + 0x8000) & 0xFFFF) - 0x8000
                v[a] = lo - ((v[a] + 0x8000) >> 16)
            elif op == "shl":
                v[a] = chk(v[a] << k)
            else:
                t = chk(v[a] - v[b])
                v[a] = chk(v[a] + v[b])
                if op == "bfly_shl":
                    t = chk(t << k)
                elif op == "bfly_mulc":
                    t = chk(t * k)
                v[b] = t
        return v


def brev(x, bits):
    return int(format(x, f"0{bits}b")[::-1], 2)


def check(g: Gen, in_lo, in_hi, trials=40):
    R = g.R
    L = R.bit_length() - 1
    w = pow(Z64, 64 // R, P)
    rng = random.Random(R)
    cases = [[in_hi] * R, [in_lo] * R, [in_hi if i % 2 else in_lo for i in range(R)], [in_lo if i % 2 else in_hi for i in range(R)]]
    for m in (1, 2, 4, 8, 16, 32):
        cases.append([in_hi if (i // m) % 2 else in_lo for i in range(R)])
    for _ in range(trials):
        cases.append([rng.randint(in_lo, in_hi) for _ in range(R)])
        cases.append([rng.choice((in_lo, in_hi, 0, 65536, 1)) for _ in range(R)])
    for x in cases:
        y = g.replay(x)
        for k in range(R):
            want = sum(x[a] * pow(w, a * k, P) for a in range(R)) % P
            got = y[brev(k, L)]
            assert abs(got) <= g.out_max, (got, g.out_max)
            assert got % P == want, (k, got, want)
    return len(cases)


# networks: (name, R, input interval, output magnitude bound)
FOLD1 = (-32767, 98303)  # range of one fold of any int32
NETS = [
    ("fermat_net64_canon", 64, (0, 65536), 1 << 29),  # first network: canonical inputs from memory
    ("fermat_net32_fold", 32, FOLD1, 1 << 29),        # later networks: inputs are folded twiddle products
    ("fermat_net32_canon", 32, (0, 65536), 1 << 29),
    ("fermat_net16_fold", 16, FOLD1, 1 << 29),
    ("fermat_net16_canon", 16, (0, 65536), 1 << 29),
    ("fermat_net64_fold", 64, FOLD1, 1 << 29),
    # first networks of the grouped kernel (r06: G = 64 / R transforms of 2^16 / G points per workgroup)
    ("fermat_net8_canon", 8, (0, 65536), 1 << 29),
    ("fermat_net4_canon", 4, (0, 65536), 1 << 29),
    ("fermat_net2_canon", 2, (0, 65536), 1 << 29),
]


# networks also emitted in two parts (Gen.split): (layers, classes, modulus)
SPLITS = {"fermat_net64_canon": (3, 5, 8)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--split", action="store_true", help="also emit the networks of SPLITS in two parts (tools/ubench/fermat_r06.hip, -DDFS=1)")
    ap.add_argument("-o", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "galois_amd", "csrc",
                                               "gfa_fermat_nets.inc"))
    a = ap.parse_args()
    parts = ["// GENERATED by tools/gen_fermat_net.py -- do not edit.  Interval-checked: no operation can leave int32.",
             "// Needs fm_add / fm_sub / fm_shl / fm_mulc / fm_fold / fm_bfold from gfa_ntt_fermat.hip.", ""]
    for name, R, (lo, hi), omax in NETS:
        g = Gen(R, lo, hi, omax)
        g.network()
        c = g.counts()
        n = check(g, lo, hi) if a.check else 0
        print(f"{name}: {c}  out_max={max(g.mag(r) for r in range(R))}" + (f"  checked {n} vectors" if a.check else ""))
        parts.append(f"// {name}: {c}")
        parts.append(g.emit_cpp(name))
        parts.append("")
        if a.split and name in SPLITS:
            layers, classes, modulus = SPLITS[name]
            head, tail = g.split(layers, classes, modulus)
            if a.check:  # the reordered op list computes the same network within the same bounds
                g2 = Gen(R, lo, hi, omax)
                g2.ops = head + tail
                check(g2, lo, hi)
            parts.append(f"// {name} in two parts: _head = layers 1..{layers} on the positions = 0..{classes - 1} (mod {modulus}) ({len(head)} ops), _tail = the rest ({len(tail)} ops)")
            parts.append(g.emit_cpp(name + "_head", head, header=False))
            parts.append(g.emit_cpp(name + "_tail", tail, header=False))
            parts.append("")
    with open(a.o, "w") as f:
        f.write("\n".join(parts))
    print("wrote", a.o)


if __name__ == "__main__":
    main()
