"""
Host-side integer and polynomial number theory used only at field / code construction time (never on the data path):
primality, factoring, primitive roots, irreducibility and primitivity of polynomials over GF(p), default polynomials.

Mirrors the *behaviour* of the reference helpers the class factory leans on (reference paths relative to
/root/reference/src/galois): `_prime.py` (is_prime :1000ff, factors :811-878), `_modular.py` (primitive_root,
is_primitive_root), `_polys/_conway.py` (conway_poly), `_polys/_primitive.py` (primitive_poly :default lexicographically
first, matlab_primitive_poly :330-433) and `_fields/_primitive_element.py` (primitive_element).  Independent
implementation: Miller-Rabin + Pollard rho instead of the reference's lookup DB / trial division chain.
"""
from __future__ import annotations

import functools
import math
import os
import random

_SMALL_PRIMES = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37]


def is_prime(n: int) -> bool:
    n = int(n)
    if n < 2:
        return False
    for p in _SMALL_PRIMES:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    # deterministic for n < 3.3e24 with the first 13 primes; beyond that add random bases
    bases = list(_SMALL_PRIMES) + [41]
    if n >= 3317044064679887385961981:
        rnd = random.Random(n)
        bases += [rnd.randrange(2, n - 1) for _ in range(24)]
    for a in bases:
        if a % n == 0:
            continue
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _pollard_rho(n: int) -> int:
    if n % 2 == 0:
        return 2
    rnd = random.Random(n)
    while True:
        c = rnd.randrange(1, n)
        y = rnd.randrange(0, n)
        m, g, r, q = 128, 1, 1, 1
        x = ys = y
        while g == 1:
            x = y
            for _ in range(r):
                y = (y * y + c) % n
            k = 0
            while k < r and g == 1:
                ys = y
                for _ in range(min(m, r - k)):
                    y = (y * y + c) % n
                    q = q * abs(x - y) % n
                g = math.gcd(q, n)
                k += m
            r *= 2
        if g == n:
            g = 1
            while g == 1:
                ys = (ys * ys + c) % n
                g = math.gcd(abs(x - ys), n)
        if g != n:
            return g


@functools.lru_cache(maxsize=4096)
def factors(n: int) -> tuple[list[int], list[int]]:
    """Prime factorisation: ([primes ascending], [multiplicities]) -- same return shape as galois.factors."""
    n = int(n)
    if n < 2:
        raise ValueError(f"Argument 'n' must be at least 2, not {n}.")
    out: dict[int, int] = {}

    def rec(v: int):
        if v == 1:
            return
        if is_prime(v):
            out[v] = out.get(v, 0) + 1
            return
        for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97):
            if v % p == 0:
                out[p] = out.get(p, 0) + 1
                rec(v // p)
                return
        d = _pollard_rho(v)
        rec(d)
        rec(v // d)

    rec(n)
    primes = sorted(out)
    return primes, [out[p] for p in primes]


def prime_power(order: int) -> tuple[int, int]:
    """order = p**m -> (p, m); raises ValueError like galois.GF for non prime powers (_fields/_factory.py:262-268)."""
    ps, ms = factors(order)
    if len(ps) != 1:
        raise ValueError(f"Argument 'order' must be a prime power, not {order} = {' * '.join(f'{p}^{e}' for p, e in zip(ps, ms))}.")
    return ps[0], ms[0]


def euler_phi_factors(p: int) -> list[int]:
    return factors(p - 1)[0] if p > 2 else []


def is_primitive_root(g: int, p: int) -> bool:
    if p == 2:
        return g % 2 == 1
    if g % p == 0:
        return False
    return all(pow(g, (p - 1) // r, p) != 1 for r in euler_phi_factors(p))


def primitive_root(p: int) -> int:
    """Smallest primitive root modulo the prime p (the reference's default primitive element of GF(p),
    _fields/_factory.py:375-376)."""
    if p == 2:
        return 1
    for g in range(2, p):
        if is_primitive_root(g, p):
            return g
    raise RuntimeError(f"no primitive root modulo {p}")


# ---------------------------------------------------------------------------------------------------------------------
# polynomials over GF(p) as coefficient lists, highest degree first
# ---------------------------------------------------------------------------------------------------------------------

def poly_from_int(value: int, p: int) -> list[int]:
    """Integer representation (base-p digits, the reference's int(Poly)) -> coefficients, highest degree first."""
    if value == 0:
        return [0]
    out = []
    while value:
        out.append(value % p)
        value //= p
    return out[::-1]


def poly_to_int(coeffs: list[int], p: int) -> int:
    v = 0
    for c in coeffs:
        v = v * p + c
    return v


def poly_str(coeffs: list[int]) -> str:
    """Same format as the reference's Poly.__str__ (e.g. 'x^8 + x^4 + x^3 + x^2 + 1')."""
    deg = len(coeffs) - 1
    terms = []
    for i, c in enumerate(coeffs):
        d = deg - i
        if c == 0:
            continue
        if d == 0:
            terms.append(f"{c}")
        else:
            xs = "x" if d == 1 else f"x^{d}"
            terms.append(xs if c == 1 else f"{c}{xs}")
    return " + ".join(terms) if terms else "0"


def _trim(a: list[int]) -> list[int]:
    i = 0
    while i < len(a) - 1 and a[i] == 0:
        i += 1
    return a[i:]


def poly_mod(a: list[int], f: list[int], p: int) -> list[int]:
    a = _trim([c % p for c in a])
    df = len(f) - 1
    inv_lead = pow(f[0], -1, p)
    a = list(a)
    while len(a) - 1 >= df and not (len(a) == 1 and a[0] == 0):
        q = a[0] * inv_lead % p
        if q:
            for i in range(df + 1):
                a[i] = (a[i] - q * f[i]) % p
        a = a[1:] if len(a) > 1 else [0]
        if len(a) - 1 < df:
            break
    return _trim(a)


def poly_mulmod(a: list[int], b: list[int], f: list[int], p: int) -> list[int]:
    prod = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                prod[i + j] = (prod[i + j] + x * y) % p
    return poly_mod(prod, f, p)


def poly_powmod(a: list[int], e: int, f: list[int], p: int) -> list[int]:
    result = [1]
    base = poly_mod(a, f, p)
    while e > 0:
        if e & 1:
            result = poly_mulmod(result, base, f, p)
        base = poly_mulmod(base, base, f, p)
        e >>= 1
    return result


def poly_gcd(a: list[int], b: list[int], p: int) -> list[int]:
    a, b = _trim(a), _trim(b)
    while not (len(b) == 1 and b[0] == 0):
        a, b = b, poly_mod(a, b, p)
    inv = pow(a[0], -1, p)
    return [c * inv % p for c in a]


def poly_sub(a: list[int], b: list[int], p: int) -> list[int]:
    n = max(len(a), len(b))
    a = [0] * (n - len(a)) + a
    b = [0] * (n - len(b)) + b
    return _trim([(x - y) % p for x, y in zip(a, b)])


def is_irreducible(f: list[int], p: int) -> bool:
    """Rabin's irreducibility test for a polynomial of degree m over GF(p)."""
    f = _trim(f)
    m = len(f) - 1
    if m < 1:
        return False
    if m == 1:
        return True
    x = [1, 0]
    # x^(p^m) == x mod f
    h = x
    for _ in range(m):
        h = poly_powmod(h, p, f, p)
    if poly_sub(h, x, p) != [0]:
        return False
    for r in factors(m)[0]:
        h = x
        for _ in range(m // r):
            h = poly_powmod(h, p, f, p)
        g = poly_gcd(f, poly_sub(h, x, p), p)
        if g != [1]:
            return False
    return True


def is_primitive_element(g: list[int], f: list[int], p: int) -> bool:
    """True if g(x) is a multiplicative generator of GF(p)[x]/(f(x))  (galois.is_primitive_element)."""
    m = len(_trim(f)) - 1
    order = p**m - 1
    g = poly_mod(g, f, p)
    if g == [0]:
        return False
    if order == 1:
        return g == [1]
    for r in factors(order)[0]:
        if poly_powmod(g, order // r, f, p) == [1]:
            return False
    return True


def is_primitive_poly(f: list[int], p: int) -> bool:
    f = _trim(f)
    if len(f) - 1 == 1:
        # f(x) = x + a is primitive iff -a is a primitive root of GF(p)
        return is_primitive_root((-f[1]) % p, p) if p > 2 else f[1] == 1
    if f[-1] == 0:
        return False
    return is_irreducible(f, p) and is_primitive_element([1, 0], f, p)


def primitive_element(f: list[int], p: int) -> int:
    """Smallest (integer representation) primitive element of GF(p)[x]/(f(x))  (galois.primitive_element, method='min';
    the search starts at the polynomial 'x', integer p -- _fields/_primitive_element.py)."""
    m = len(_trim(f)) - 1
    for v in range(p, p**m):
        if is_primitive_element(poly_from_int(v, p), f, p):
            return v
    raise RuntimeError("no primitive element found")


@functools.lru_cache(maxsize=256)
def primitive_poly(p: int, m: int) -> int:
    """Lexicographically-first monic primitive polynomial of degree m over GF(p), as an integer
    (galois.primitive_poly(p, m) with default terms=None, method='min')."""
    start = p**m
    for v in range(start + 1, 2 * start):
        f = poly_from_int(v, p)
        if f[-1] == 0:
            continue
        if is_primitive_poly(f, p):
            return v
    raise RuntimeError("no primitive polynomial found")


def matlab_primitive_poly(p: int, m: int) -> int:
    """galois.matlab_primitive_poly (_polys/_primitive.py:330-433): Matlab's default primitive polynomial = the
    lexicographically first one, except for GF(2^7), GF(2^14) and GF(2^16)."""
    if p == 2 and m == 7:
        return (1 << 7) | (1 << 3) | 1
    if p == 2 and m == 14:
        return (1 << 14) | (1 << 10) | (1 << 6) | (1 << 1) | 1
    if p == 2 and m == 16:
        return (1 << 16) | (1 << 12) | (1 << 3) | (1 << 1) | 1
    return primitive_poly(p, m)


# ---------------------------------------------------------------------------------------------------------------------
# Conway polynomials (Frank Luebeck's published table; subset shipped in data/conway_polys.txt)
# ---------------------------------------------------------------------------------------------------------------------

_CONWAY: dict[tuple[int, int], int] | None = None


def _load_conway() -> dict[tuple[int, int], int]:
    global _CONWAY
    if _CONWAY is None:
        table = {}
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "conway_polys.txt")
        with open(path) as fh:
            for line in fh:
                line = line.strip()
                if not line or line.startswith("#"):
                    continue
                p, m, v = line.split()
                table[(int(p), int(m))] = int(v)
        _CONWAY = table
    return _CONWAY


def conway_poly(p: int, m: int) -> int:
    """Conway polynomial C_{p,m} as an integer; LookupError when it is not in the shipped table, like the reference
    (_polys/_conway.py:288-299, _databases/_interface.py:140-150)."""
    if m == 1:
        return 2 * p - primitive_root(p) if p > 2 else 3  # x - g
    table = _load_conway()
    try:
        return table[(p, m)]
    except KeyError:
        raise LookupError(
            f"The shipped table of Conway polynomials (Frank Luebeck's list, subset) does not contain an entry for a "
            f"degree-{m} polynomial over GF({p}). Pass `irreducible_poly=` explicitly."
        ) from None
