"""Shared test helpers: golden-fixture loading and oracle construction (the oracle is the checker, never the product)."""
import json
import os

import numpy as np

from oracle import gf_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDILOCKS = 2**64 - 2**32 + 1

SAGE_FIELDS = sorted(f[len("sage_fields_"):-4] for f in os.listdir(GOLDEN) if f.startswith("sage_fields_"))


def load_sage_field(tag):
    d = np.load(os.path.join(GOLDEN, f"sage_fields_{tag}.npz"))
    props = json.loads(str(d["properties"]))
    return props, d


def poly_coeffs_to_int(coeffs, p):
    v = 0
    for c in coeffs:
        v = v * p + int(c)
    return v


def oracle_field_from_props(props, lookup):
    p, m = props["characteristic"], props["degree"]
    irr = poly_coeffs_to_int(props["irreducible_poly"], p) if m > 1 else None
    return O.OracleField(p, m, irr, int(props["primitive_element"]), lookup=lookup)


def reference_outputs():
    return np.load(os.path.join(GOLDEN, "reference_outputs.npz"))


def reference_table_fields():
    return np.load(os.path.join(GOLDEN, "reference_table_fields.npz"))


def sage_rs():
    d = np.load(os.path.join(GOLDEN, "sage_rs.npz"))
    names = json.loads(str(d["names"]))
    return names, d


def as_int_list(a):
    return [int(v) for v in np.asarray(a).ravel()]


def assert_equal_ints(actual, expected, msg=""):
    a, e = np.asarray(actual), np.asarray(expected)
    assert a.shape == e.shape, f"{msg}: shape {a.shape} != {e.shape}"
    if a.dtype == object or e.dtype == object or a.dtype.kind != e.dtype.kind:
        assert as_int_list(a) == as_int_list(e), msg
    else:
        assert np.array_equal(a, e), msg


def sage_bch():
    d = np.load(os.path.join(GOLDEN, "sage_bch.npz"))
    names = json.loads(str(d["names"]))
    return names, d


def reference_bch_outputs():
    return np.load(os.path.join(GOLDEN, "reference_bch_outputs.npz"))


def reference_wide_codes():
    """RS / BCH codes whose syndrome field has more than 256 elements (generate_golden.py reference_wide)."""
    return np.load(os.path.join(GOLDEN, "reference_wide_codes.npz"))


WIDE_RS_CASES = ["rs1023_1003", "rs1023_1011_short_c0", "rs728_712_gf729", "rs511_501"]
WIDE_BCH_CASES = ["bch511_493", "bch1023_1003", "bch1023_973_short", "bch728_gf3", "bch511_484_nonsys"]


def parse_sage_poly(text, p):
    """'x^18 + 2*x^14 + x + 2' -> coefficients, highest degree first (Sage's str() of a polynomial over GF(p))."""
    terms = {}
    for term in text.replace(" ", "").replace("-", "+-").split("+"):
        if not term:
            continue
        if "x" in term:
            coef, _, rest = term.partition("x")
            coef = coef.rstrip("*")
            c = 1 if coef == "" else (-1 if coef == "-" else int(coef))
            deg = int(rest[1:]) if rest.startswith("^") else 1
        else:
            c, deg = int(term), 0
        terms[deg] = (terms.get(deg, 0) + c) % p
    n = max(terms)
    return [terms.get(i, 0) for i in range(n, -1, -1)]


SAGE_LINALG = sorted(f[len("sage_linalg_"):-4] for f in os.listdir(GOLDEN) if f.startswith("sage_linalg_"))


def load_sage_linalg(tag):
    d = np.load(os.path.join(GOLDEN, f"sage_linalg_{tag}.npz"))
    props = json.loads(str(d["properties"]))
    return props, d


def linalg_cases(d, op, keys):
    """Yields tuples of arrays (one per key) for every stored case of `op`."""
    for i in range(int(d[f"{op}_count"])):
        yield tuple(d[f"{op}{i}_{k}"] for k in keys)


SAGE_POLYS = sorted(f[len("sage_polys_"):-4] for f in os.listdir(GOLDEN) if f.startswith("sage_polys_"))


def load_sage_polys(tag):
    d = np.load(os.path.join(GOLDEN, f"sage_polys_{tag}.npz"))
    props = json.loads(str(d["properties"]))
    return props, d
