"""
TEST INFRASTRUCTURE ONLY -- imports the reference package in place from /root/reference/src.

Only usable inside the build container (/root/reference does not exist on the GPU box).  It is used
to (a) validate the CPU restatement in oracle/gf_oracle.c and (b) generate the committed golden
fixtures under tests/golden/ (see tests/golden/generate_golden.py).

Two accommodations, neither of which touches the reference's arithmetic:
  1. `numba` stand-in (oracle/ref_shim/numba) put first on sys.path.
  2. The reference's prime_factors.db is absent from the snapshot; PrimeFactorsDatabase.file is
     pointed at an empty SQLite table so galois._prime.factors falls through to its own algorithms
     (reference: src/galois/_prime.py:820-828, src/galois/_databases/_interface.py:56-68).
"""
import os
import sqlite3
import sys
import tempfile

REFERENCE_SRC = "/root/reference/src"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "galois"))


def load():
    """Returns the reference `galois` module (python-calculate capable)."""
    if "galois" in sys.modules and getattr(sys.modules["galois"], "_IS_REFERENCE_SHIM", False):
        return sys.modules["galois"]
    if not available():
        raise RuntimeError("reference tree /root/reference/src is not present on this machine")
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)  # numba stand-in
    sys.path.insert(1, REFERENCE_SRC)
    import galois  # noqa: E402
    from galois._databases import _interface

    tmp = os.path.join(tempfile.gettempdir(), "galois_ref_empty_prime_factors.db")
    if not os.path.exists(tmp):
        con = sqlite3.connect(tmp)
        con.execute(
            "CREATE TABLE IF NOT EXISTS factorizations (value TEXT PRIMARY KEY, factors TEXT, "
            "multiplicities TEXT, composite TEXT)"
        )
        con.commit()
        con.close()
    _interface.PrimeFactorsDatabase.file = tmp
    galois._IS_REFERENCE_SHIM = True
    return galois


def ref_field(order, **kwargs):
    """galois.GF(order, ...) in the reference's documented no-Numba mode ("python-calculate",
    src/galois/_domains/_array.py:341-344).  The prime subfield is switched to python-calculate first because an
    extension field's array add/subtract runs through its prime subfield's ufuncs (_calculate.py:155-166)."""
    galois = load()
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p = galois.factors(order)[0][0] if not isinstance(order, tuple) else order[0]
        galois.GF(p, compile="python-calculate")
        kwargs.setdefault("compile", "python-calculate")
        return galois.GF(order, **kwargs)
