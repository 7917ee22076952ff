"""CPU-only tests of the host side: C-ABI surface, field factory, number theory, RS construction.  No compute calls."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import galois_amd as ga
from galois_amd import _lib as L
from galois_amd import _numtheory as nt
from oracle import gf_oracle as O
from tests import helpers as H


def test_library_exports_every_declared_symbol(repo_root):
    header = open(os.path.join(repo_root, "include", "galois_amd.h")).read()
    declared = set(re.findall(r"\b(gfa_[a-z0-9_]+)\s*\(", header))
    declared -= {"gfa_field_t", "gfa_rs_t", "gfa_stream_t"}
    lib = ctypes.CDLL(L.LIB_PATH)
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, f"not exported: {missing}"
    assert declared == set(L.SIGNATURES), f"binding/header mismatch: {declared ^ set(L.SIGNATURES)}"
    assert lib.gfa_abi_version() == 1


def test_library_exports_nothing_but_the_c_abi():
    """-fvisibility=hidden + the push/pop in include/galois_amd.h: no C++ internals (gfa::HostArith, kernels' host stubs, ...)
    in the dynamic symbol table."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [line.split()[-1] for line in out.splitlines() if line.strip()]
    stray = [n for n in names if not n.startswith("gfa_") and n not in ("_init", "_fini", "__bss_start", "_edata", "_end")
             and not n.startswith("__hip_")]
    assert not stray, stray[:20]
    assert "gfa_ntt" in names and "gfa_binary" in names


def test_product_never_touches_the_oracle(repo_root):
    pkg = os.path.join(repo_root, "galois_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "gf_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f
                assert "/root/reference" not in text.replace("/root/reference/src/galois", "REFDOC"), f


def test_flyweights_and_properties():
    GF = ga.GF(2**8)
    assert ga.GF(2, 8) is GF and ga.GF(256) is GF
    assert ga.GF2 is ga.GF(2)
    assert GF.name == "GF(2^8)" and GF.characteristic == 2 and GF.degree == 8 and GF.order == 256
    assert int(GF.irreducible_poly) == 285 and str(GF.irreducible_poly) == "x^8 + x^4 + x^3 + x^2 + 1"
    assert GF.is_extension_field and not GF.is_prime_field and GF.prime_subfield is ga.GF(2)
    assert GF.ufunc_modes == ["jit-lookup", "jit-calculate"] and GF.ufunc_mode == "jit-lookup"
    GF.compile("jit-calculate")
    assert GF.ufunc_mode == "jit-calculate"
    GF.compile("auto")
    assert GF.ufunc_mode == GF.default_ufunc_mode == "jit-lookup"
    with pytest.raises(ValueError):
        GF.compile("python-calculate")
    aes = ga.GF(2**8, irreducible_poly=283)
    assert aes is not GF and aes._primitive_element_int == 3 and not aes.is_primitive_poly
    assert aes is ga.GF(2**8, irreducible_poly="x^8 + x^4 + x^3 + x + 1", primitive_element=3)
    with pytest.raises(ValueError):
        ga.GF(2**8, irreducible_poly=0x11D + 1)  # reducible
    with pytest.raises(ValueError):
        ga.GF(2**8, primitive_element=1)  # not a generator
    with pytest.raises(ValueError):
        ga.GF(6)
    with pytest.raises(ValueError):
        ga.GF(31, primitive_element=5)
    with pytest.raises(TypeError):
        ga.GF(2.0)
    # orders in [2^64, 2^128] get the two-limb device representation (the reference: dtype=object); above that 4 / 8 / 16 limbs, with the
    # polynomial and the primitive element named by the caller (their default searches factor q - 1); beyond 1024 bits nothing
    big = ga.GF(36893488147419103183, primitive_element=3)
    assert big.dtypes == [np.object_] and big.order == 36893488147419103183 and big.ufunc_modes == ["jit-calculate"]
    with pytest.raises(ValueError, match="primitive_element"):
        ga.GF(2**127 - 1, 2, irreducible_poly=[1, 0, 1], verify=False)
    with pytest.raises(NotImplementedError):  # extension fields keep base-p digits of 32 bits on the device: p < 2^32
        ga.GF(2**127 - 1, 2, irreducible_poly=[1, 0, 1], primitive_element=[1, 3], verify=False)
    k4 = ga.GF(65537, 12, irreducible_poly=[1] + [0] * 10 + [1, 2], primitive_element=[1, 0], verify=False)
    assert k4._NL == 4 and k4.order == 65537**12 and k4.dtypes == [np.object_]
    assert ga.GF(2**521 - 1, primitive_element=3, verify=False)._NL == 16
    with pytest.raises(NotImplementedError):
        ga.GF(2, 1025, irreducible_poly=(1 << 1025) | 3, primitive_element=2, verify=False)


def test_dtypes_follow_the_reference_rules():
    # SURVEY.md 8(a1), probed on the reference
    names = lambda F: [np.dtype(d).name for d in F.dtypes]
    assert names(ga.GF(31)) == ["uint8", "uint16", "uint32", "int8", "int16", "int32", "int64"]
    assert names(ga.GF(2**8)) == ["uint8", "uint16", "uint32", "int16", "int32", "int64"]
    assert names(ga.GF(65537)) == ["uint32", "int32", "int64"]
    assert names(ga.GF(7340033)) == ["uint32", "int32", "int64"]
    assert names(ga.GF(2**64 - 2**32 + 1)) == ["object"]
    assert names(ga.GF(2**32)) == ["uint32", "int64"]
    assert names(ga.GF(2147483647)) == ["uint32", "int32", "int64"]
    assert ga.GF(31).ufunc_mode == "jit-lookup" and ga.GF(3**5).ufunc_mode == "jit-lookup"  # the reference defaults too
    assert ga.GF(65537).ufunc_mode == "jit-calculate" and ga.GF(2**16).ufunc_mode == "jit-calculate"


@pytest.mark.parametrize("order", [2**8, 31, 3**5, 5**3, 2**4, 65537])
def test_lookup_tables_match_oracle(order):
    GF = ga.GF(order)
    p, m = GF.characteristic, GF.degree
    F = O.OracleField(p, m, int(GF.irreducible_poly) if m > 1 else None, GF._primitive_element_int, lookup=True)
    E, Lg, Z, ze = GF._tables()
    oe, ol, oz, oze = F.tables()
    assert np.array_equal(E, oe) and np.array_equal(Lg, ol) and np.array_equal(Z, oz) and ze == oze
    if order == 2**8:  # SURVEY.md 8(c) constants
        assert list(E[:10]) == [1, 2, 4, 8, 16, 32, 64, 128, 29, 58] and list(Lg[:10]) == [0, 0, 1, 25, 2, 50, 26, 198, 3, 223]


@pytest.mark.parametrize("order", [2**8, 31, 7**3, 65537, 7340033, 2**64 - 2**32 + 1, 2**32, 2147483647, 251**3,
                                   18446744073709551557])
def test_host_scalar_arithmetic_matches_oracle(order):
    """gfa_scalar runs the same formulas as the kernels (gfa_arith.h compiled for the host)."""
    GF = ga.GF(order)
    p, m = GF.characteristic, GF.degree
    F = O.OracleField(p, m, int(GF.irreducible_poly) if m > 1 else None, GF._primitive_element_int)
    rng = np.random.default_rng(order % 1000)
    n = 300
    a = [int(rng.integers(0, 2**63)) * 2 % order for _ in range(n)]
    b = [(int(rng.integers(0, 2**63)) * 2 + 1) % order for _ in range(n)]
    a[0] = 0
    a[1] = order - 1
    b[1] = order - 1
    oa, ob = np.array(a, dtype=object), np.array(b, dtype=object)
    bnz = [v or 1 for v in b]
    assert [GF._scalar(L.OP_ADD, x, y) for x, y in zip(a, b)] == H.as_int_list(F.add(oa, ob))
    assert [GF._scalar(L.OP_SUB, x, y) for x, y in zip(a, b)] == H.as_int_list(F.sub(oa, ob))
    assert [GF._scalar(L.OP_MUL, x, y) for x, y in zip(a, b)] == H.as_int_list(F.mul(oa, ob))
    assert [GF._scalar(L.OP_NEG, x) for x in a] == H.as_int_list(F.neg(oa))
    assert [GF._scalar(L.OP_DIV, x, y) for x, y in zip(a, bnz)] == H.as_int_list(F.div(oa, np.array(bnz, dtype=object)))
    assert [GF._scalar(L.OP_RECIP, y) for y in bnz] == H.as_int_list(F.recip(np.array(bnz, dtype=object)))
    es = [int(e) for e in rng.integers(-40, 80, n)]
    anz = [v or 1 for v in a]
    assert [GF._scalar(L.OP_POW, x, e) for x, e in zip(anz, es)] == H.as_int_list(F.pow(np.array(anz, dtype=object), es))
    with pytest.raises(ZeroDivisionError):
        GF._scalar(L.OP_RECIP, 0)
    with pytest.raises(ZeroDivisionError):
        GF._scalar(L.OP_POW, 0, -3)


def test_roots_of_unity_known_values():
    # SURVEY.md 8(c)
    assert ga.GF(65537)._root_of_unity_int(2**16) == 3
    assert ga.GF(7340033)._root_of_unity_int(2**20) == 2187
    G = ga.GF(2**64 - 2**32 + 1)
    assert G._root_of_unity_int(2**20) == 3511170319078647661
    assert G._root_of_unity_int(2**26) == 17096174751763063430
    with pytest.raises(ValueError):
        ga.GF(65537).primitive_root_of_unity(2**20)  # BASELINE config 3 as written does not exist


def test_number_theory():
    assert nt.factors(2**64 - 2**32) == ([2, 3, 5, 17, 257, 65537], [32, 1, 1, 1, 1, 1])
    assert nt.primitive_root(7340033) == 3 and nt.primitive_root(2**64 - 2**32 + 1) == 7 and nt.primitive_root(31) == 3
    assert nt.matlab_primitive_poly(2, 8) == 0x11D and nt.matlab_primitive_poly(2, 4) == 0b10011
    assert nt.matlab_primitive_poly(2, 7) == 0b10001001
    assert not nt.is_prime(2**32 + 1) and nt.is_prime(2**61 - 1)
    with pytest.raises(LookupError):
        nt.conway_poly(65537, 2)


def test_reed_solomon_construction_matches_sage_fixtures():
    names, d = H.sage_rs()
    for key in names:
        meta = json.loads(str(d[f"{key}/meta"]))
        q = meta["q"]
        rs = ga.ReedSolomon(meta["n"], meta["k"], field=ga.GF(q), alpha=meta["alpha"], c=meta["c"],
                            systematic=meta["is_systematic"])
        assert (rs.n, rs.k, rs.d) == (meta["n"], meta["k"], meta["d"])
        assert rs.is_primitive == meta["is_primitive"] and rs.is_narrow_sense == meta["is_narrow_sense"]
        assert str(rs.generator_poly) == meta["generator_poly"].replace("*", ""), key  # Sage writes 11*x^4
        H.assert_equal_ints(rs.G, d[f"{key}/G"], key + " G")
        H.assert_equal_ints(rs.H, d[f"{key}/H"], key + " H")


def test_reed_solomon_255_223_constants():
    rs = ga.ReedSolomon(255, 223)
    assert int(rs.field.irreducible_poly) == 0x11D and rs.alpha == 2 and rs.t == 16 and rs.d == 33
    assert list(rs.roots[:4]) == [2, 4, 8, 16]
    assert list(rs.generator_poly.coeffs[:4]) == [1, 232, 29, 189] and list(rs.generator_poly.coeffs[-2:]) == [216, 45]
    d = H.reference_outputs()
    H.assert_equal_ints(rs.generator_poly.coeffs, d["rs/rs255_223/generator_poly"])
    assert ga.ReedSolomon(255, d=33).k == 223
    with pytest.raises(ValueError):
        ga.ReedSolomon(255, 223, 30)
    with pytest.raises(ValueError):
        ga.ReedSolomon(255)
    with pytest.raises(TypeError):
        ga.ReedSolomon(255.0, 223)


def test_ntt_argument_checks_need_no_gpu():
    with pytest.raises(ValueError):
        ga.ntt([1, 2, 3, 4], modulus=7)  # 4 does not divide 6
    with pytest.raises(ValueError):
        ga.ntt([1, 2, 3, 4], modulus=15)  # not prime
    with pytest.raises(ValueError):
        ga.ntt([1, 2, 3, 4], size=3, modulus=13)
    with pytest.raises(ValueError):
        ga.ntt([1, 2, 3, 14], modulus=13)
    with pytest.raises(TypeError):
        ga.ntt(np.float32(3.0))


def test_shard_range_partitions():
    from galois_amd import dist

    for total in (0, 1, 7, 8, 2**20, 10**8 + 3):
        for world in (1, 2, 3, 8):
            spans = [dist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- BCH construction (host logic; the library call gfa_bch_create is host-only) ----------------------------------
def test_bch_properties_against_sage_fixtures():
    """tests/codes/test_bch.py:61-95 over the 204 Sage fixtures: k, d, g(x), h(x), G, H, flags."""
    import json

    from tests import helpers as H

    names, d = H.sage_bch()
    for key in names:
        meta = json.loads(str(d[f"{key}/meta"]))
        b = ga.BCH(meta["n"], meta["k"], d=meta["d"], field=ga.GF(meta["q"]), alpha=meta["alpha"], c=meta["c"],
                   systematic=meta["is_systematic"])
        assert (b.n, b.k, b.d, b.t) == (meta["n"], meta["k"], meta["d"], (meta["d"] - 1) // 2)
        assert [int(v) for v in b.generator_poly.coeffs] == H.parse_sage_poly(meta["generator_poly"], meta["q"]), key
        assert [int(v) for v in b.parity_check_poly.coeffs] == H.parse_sage_poly(meta["parity_check_poly"], meta["q"]), key
        assert np.array_equal(b.G, d[f"{key}/G"]) and np.array_equal(b.H, d[f"{key}/H"]), key
        assert (b.is_primitive, b.is_narrow_sense, b.is_systematic) == (meta["is_primitive"], meta["is_narrow_sense"],
                                                                         meta["is_systematic"])
        assert b.extension_field.order == meta["q"] ** meta["m"]


def test_bch_valid_binary_codes():
    """(n, k, t) rows of the classical table of primitive binary BCH codes (tests/codes/test_bch.py:172-226 checks the
    same table from Lin & Costello): a sample per length, through the k-only constructor (binary search for d)."""
    for n, k, t in [(7, 4, 1), (15, 11, 1), (15, 7, 2), (15, 5, 3), (31, 26, 1), (31, 21, 2), (31, 16, 3), (31, 11, 5),
                    (31, 6, 7), (63, 57, 1), (63, 45, 3), (63, 36, 5), (63, 30, 6), (63, 18, 10), (63, 7, 15),
                    (127, 120, 1), (127, 99, 4), (127, 64, 10), (255, 247, 1), (255, 223, 4), (255, 131, 18)]:
        b = ga.BCH(n, k)
        assert (b.n, b.k, b.t) == (n, k, t)
    assert repr(ga.BCH(15, 7)) == "<BCH Code: [15, 7, 5] over GF(2)>"
    assert repr(ga.BCH(26, 14, field=ga.GF(3))) == "<BCH Code: [26, 14, 7] over GF(3)>"
    with pytest.raises(ValueError):
        ga.BCH(15, 8)
    with pytest.raises(ValueError):
        ga.BCH(15, 7, field=ga.GF(4))
    big = ga.BCH(511, d=11)  # syndrome field GF(2^9): served by the table-driven kernels (gfa_rs_wide.hip)
    assert (big.k, big.extension_field.order) == (466, 512)


def test_wide_code_construction_against_the_reference():
    """Codes whose syndrome field has more than 256 elements: k, d, roots and g(x) as the reference built them
    (tests/golden/reference_wide_codes.npz), through both BCH constructors."""
    import json

    from tests import helpers as H

    d = H.reference_wide_codes()
    for tag in H.WIDE_RS_CASES:
        meta = json.loads(str(d[f"rs/{tag}/meta"]))
        GF = ga.GF(meta["p"], meta["m"], irreducible_poly=meta["irr"], primitive_element=meta["field_alpha"])
        rs = ga.ReedSolomon(meta["n"], meta["k"], field=GF, c=meta["c"], alpha=meta["alpha"])
        assert [int(v) for v in rs.generator_poly.coeffs] == [int(v) for v in d[f"rs/{tag}/generator_poly"]], tag
        assert (rs.d, rs.t) == (meta["n"] - meta["k"] + 1, (meta["n"] - meta["k"]) // 2)
    for tag in H.WIDE_BCH_CASES:
        meta = json.loads(str(d[f"bch/{tag}/meta"]))
        p = meta["p"]
        ext = ga.GF(p, meta["ext_m"], irreducible_poly=meta["ext_irr"], primitive_element=meta["ext_alpha"])
        kw = dict(field=ga.GF(p), extension_field=ext, alpha=meta["alpha"], c=meta["c"], systematic=meta["systematic"])
        b = ga.BCH(meta["n"], meta["k"], d=meta["d"], **kw)
        assert (b.k, b.d) == (meta["k"], meta["d"]), tag
        assert [int(v) for v in b.generator_poly.coeffs] == [int(v) for v in d[f"bch/{tag}/generator_poly"]], tag
        assert [int(v) for v in b.roots] == [int(v) for v in d[f"bch/{tag}/roots"]], tag
        bd = ga.BCH(meta["n"], d=meta["d"], **kw)
        assert (bd.k, [int(v) for v in bd.generator_poly.coeffs]) == (b.k, [int(v) for v in b.generator_poly.coeffs]), tag


def _build_c_host(tmp_path):
    """gcc -std=c99 on examples/c_host_rs.c against include/galois_amd.h and the in-tree library: the boundary is usable
    from a plain C host (no C++, no Python)."""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("gcc / ROCm headers not available")
    exe = str(tmp_path / "c_host_rs")
    lib_dir = os.path.join(root, "galois_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(root, "include"), "-I/opt/rocm/include",
           os.path.join(root, "examples", "c_host_rs.c"), "-L" + lib_dir, "-lgalois_amd", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_host_example_compiles_and_links(tmp_path):
    assert os.path.exists(_build_c_host(tmp_path))


@pytest.mark.gpu
def test_c_host_example_runs(tmp_path):
    import subprocess

    r = subprocess.run([_build_c_host(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "round trip of 4096 codewords (16 errors each) OK" in r.stdout


def test_lazy_goldilocks_arithmetic_header_on_the_host(tmp_path, repo_root):
    """galois_amd/csrc/gfa_goldilocks.h compiles for the host too: its 96-bit lazy add / sub / fold / multiply formulas are
    checked against __int128 arithmetic on a million random and edge-case operands (tests/csrc/goldilocks_host_test.cpp)."""
    import subprocess

    exe = str(tmp_path / "gl_test")
    subprocess.run(["g++", "-O2", "-I", os.path.join(repo_root, "galois_amd", "csrc"), os.path.join(repo_root, "tests", "csrc", "goldilocks_host_test.cpp"),
                    "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "fails 0" in r.stdout, r.stdout + r.stderr


def test_field_arithmetic_header_on_the_host(tmp_path, repo_root):
    """galois_amd/csrc/gfa_arith.h on the host: the Goldilocks inversion chain (64 squarings + 9 products for p - 2) against binary
    square-and-multiply and a * a^-1 = 1, the 32-bit Montgomery power against the Barrett power, the carry-less products of the binary-field
    kernels (16 / 9 integer multiplies + folds, nine multiplies on 16-bit halves + byte-indexed reduction tables) against shift-and-xor
    (tests/csrc/arith_host_test.cpp)."""
    import subprocess

    exe = str(tmp_path / "arith_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(repo_root, "galois_amd", "csrc"),
                    os.path.join(repo_root, "tests", "csrc", "arith_host_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "fails 0" in r.stdout, r.stdout + r.stderr


def test_signed_montgomery_networks_on_the_host_never_leave_int32(tmp_path, repo_root):
    """galois_amd/csrc/gfa_m32_net.h on the host (r05: primes up to 2^29 on the signed-Montgomery NTT kernels): the SAME network
    templates the kernels instantiate run on a range-checking integer -- every sum, difference and product operand of every
    shape (radix 4 .. 64) and prime class (BMAX 64 / 32 / 8 / 4, largest prime of the class) is checked against the int32 range
    on worst-case inputs, the outputs against a direct DFT (tests/csrc/m32_net_host_test.cpp)."""
    import subprocess

    exe = str(tmp_path / "m32_net_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(repo_root, "galois_amd", "csrc"),
                    os.path.join(repo_root, "tests", "csrc", "m32_net_host_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "m32 networks ok" in r.stdout, r.stdout + r.stderr


def test_fermat_first_twiddles_in_registers_on_the_host(tmp_path, repo_root):
    """galois_amd/csrc/gfa_fermat_tw.h on the host (r06: the one-pass GF(65537) kernel forms w^(m k0) from two per-thread seeds
    instead of streaming a 256 KiB table): the same header over a range-checking integer for every column, every output and
    five roots of unity -- values against w^(m k0) mod 65537, every product / fold against the int32 range, the stated bounds
    (tests/csrc/fermat_tw_host_test.cpp)."""
    import subprocess

    exe = str(tmp_path / "fermat_tw_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(repo_root, "galois_amd", "csrc"),
                    os.path.join(repo_root, "tests", "csrc", "fermat_tw_host_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "range failures 0, wrong 0" in r.stdout, r.stdout + r.stderr


def test_packed_digit_sums_on_the_host(tmp_path, repo_root):
    """galois_amd/csrc/gfa_packed.h on the host (r05): packed base-p digit sums / differences / negatives of GF(p^m), p odd, against
    digit-wise arithmetic -- every element pair of the small fields, random and edge pairs of 22 field shapes up to 2^20 elements, the
    exact-quotient claim swept over the whole range, and the fields the scheme must refuse (tests/csrc/packed_host_test.cpp)."""
    import subprocess

    exe = str(tmp_path / "packed_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(repo_root, "galois_amd", "csrc"),
                    os.path.join(repo_root, "tests", "csrc", "packed_host_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "packed digits ok" in r.stdout, r.stdout + r.stderr


def test_reed_solomon_table_builders_on_the_host(tmp_path, repo_root):
    """galois_amd/csrc/gfa_rs_host.h on the host: the LFSR row table in consecutive and in planar order, run through a host model
    of rs_lfsr_kernel's state handling and compared with schoolbook division for n - k = 4 .. 64; the decoder's lane tables
    (positions by field element, root <-> lane bijection, one root per LDS bank and half-wave) -- tests/csrc/rs_host_test.cpp."""
    import subprocess

    exe = str(tmp_path / "rs_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(repo_root, "galois_amd", "csrc"),
                    os.path.join(repo_root, "tests", "csrc", "rs_host_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "fails 0" in r.stdout, r.stdout + r.stderr


def test_berlekamp_massey_arrangement_of_the_wave_kernel_on_the_host(tmp_path, repo_root):
    """tests/csrc/bm_host_test.cpp: a lane-by-lane host model of rs_decode_bin_kernel's Berlekamp-Massey loop (inversionless recurrence
    in a frame that moves one lane per step, early stop, the cleared half before the 32nd step) against Massey's algorithm with
    divisions on 400 000 syndrome sequences of every length 1..32 -- same LFSR length, same polynomial up to its leading scale."""
    import subprocess

    exe = str(tmp_path / "bm_test")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(repo_root, "tests", "csrc", "bm_host_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "fails 0" in r.stdout, r.stdout + r.stderr
