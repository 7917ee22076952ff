"""
TEST INFRASTRUCTURE -- never imported by the product (galois_amd/): CPU restatement, on Python integers, of the reference's
scalar arithmetic for fields of order >= 2^64, which the reference itself runs as dtype=object arrays through the same
formulas (src/galois/_domains/_calculate.py):
    add / subtract / negative   add_modular :133-147, subtract_modular :235-251, negative_modular :184-199 (prime fields),
                                add_vector / subtract_vector / negative_vector :150-181, :254-285, :202-232 (GF(p^m)),
                                xor for characteristic 2 (_fields/_ufunc.py:59-61)
    multiply                    multiply_modular :327-340, multiply_binary :288-324, multiply_vector :343-383
    reciprocal                  reciprocal_modular_egcd :386-417 (value = a^-1 mod p), reciprocal_itoh_tsujii :447-489
    power                       power_square_and_multiply :558-592 (0**0 = 1, 0**negative raises)
Pinned by tests/test_oracle_golden.py against the reference's own Sage vectors for GF(2^100), GF(36893488147419103183) and
GF(109987^4) (tests/golden/sage_wide_*.npz, packed from /root/reference/tests/fields/data by generate_golden.py).
"""
from __future__ import annotations


class WideOracle:
    def __init__(self, p: int, m: int, irreducible_poly_coeffs: list[int] | None):
        self.p, self.m, self.q = p, m, p**m
        self.irr = list(irreducible_poly_coeffs) if irreducible_poly_coeffs else None  # degree m .. 0
        if m > 1 and p == 2:
            self.irr_int = sum(c << (m - i) for i, c in enumerate(self.irr))

    # ---- helpers ----
    def _vec(self, a: int) -> list[int]:
        d = []
        for _ in range(self.m):
            d.append(a % self.p)
            a //= self.p
        return d[::-1]  # most significant first (int_to_vector, _calculate.py:22-33)

    def _int(self, v: list[int]) -> int:
        a = 0
        for c in v:
            a = a * self.p + c
        return a

    def add(self, a: int, b: int) -> int:
        if self.m == 1:
            return (a + b) % self.p
        if self.p == 2:
            return a ^ b
        return self._int([(x + y) % self.p for x, y in zip(self._vec(a), self._vec(b))])

    def neg(self, a: int) -> int:
        if self.m == 1:
            return (-a) % self.p
        if self.p == 2:
            return a
        return self._int([(-x) % self.p for x in self._vec(a)])

    def sub(self, a: int, b: int) -> int:
        return self.add(a, self.neg(b))

    def mul(self, a: int, b: int) -> int:
        if self.m == 1:
            return a * b % self.p
        if self.p == 2:  # multiply_binary
            c = 0
            while b:
                if b & 1:
                    c ^= a
                b >>= 1
                a <<= 1
                if a >> self.m:
                    a ^= self.irr_int
            return c
        av, bv = self._vec(a), self._vec(b)  # multiply_vector
        m, p = self.m, self.p
        c = [0] * m
        for it in range(m):
            bl = bv[m - 1 - it]
            if bl:
                c = [(ci + bl * ai) % p for ci, ai in zip(c, av)]
            qd = av[0]
            av = av[1:] + [0]
            if qd:
                av = [(ai - qd * self.irr[1 + i]) % p for i, ai in enumerate(av)]
        return self._int(c)

    def pow(self, a: int, e: int) -> int:
        if e == 0:
            return 1
        if a == 0:
            if e < 0:
                raise ZeroDivisionError("Cannot compute the multiplicative inverse of 0 in a Galois field.")
            return 0
        if e < 0:
            a, e = self.inv(a), -e
        r = 1
        while e:
            if e & 1:
                r = self.mul(r, a)
            a = self.mul(a, a)
            e >>= 1
        return r

    def inv(self, a: int) -> int:
        if a == 0:
            raise ZeroDivisionError("Cannot compute the multiplicative inverse of 0 in a Galois field.")
        if self.m == 1:
            return pow(a, -1, self.p)
        return self.pow(a, self.q - 2)

    def div(self, a: int, b: int) -> int:
        return self.mul(a, self.inv(b))
