"""
Generates the committed golden fixtures under tests/golden/ (run in the build container only):

    PYTHONPATH=/root/repo python tests/golden/generate_golden.py

Two sources, both the reference's own:
  1. tests/golden/sage_fields_*.npz, sage_rs.npz -- the Sage/SymPy-generated vectors the reference's test-suite pins
     this path with (/root/reference/tests/fields/data/*/{add,subtract,multiply,divide,additive_inverse,
     multiplicative_inverse,scalar_multiply,power}.pkl, tests/codes/data/reed_solomon/*.pkl; loaders in
     tests/fields/conftest.py:160-293 and tests/codes/conftest.py:65-138), re-packed as compressed .npz because the
     pickles cannot travel to the GPU box.  Fields of order >= 2^64 are skipped (no device representation).
  2. tests/golden/reference_outputs.npz -- outputs of the reference itself, imported in place from
     /root/reference/src in its pure-Python mode (oracle/ref_shim), for the configurations that have no upstream
     fixture: NTTs over GF(65537) / GF(7340033) / Goldilocks (+ the four SymPy KATs of tests/fields/test_ntt.py:13-18),
     RS(255,223) encode/decode incl. erasures and uncorrectable words, GF(2^8)/GF(31)/Goldilocks element-wise samples.
"""
import glob
import json
import os
import pickle
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.environ.get("GOLDEN_OUT", HERE)  # tests regenerate into a scratch directory and compare
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_shim"))
REF_TESTS = "/root/reference/tests"

warnings.simplefilter("ignore")


def small(a):
    a = np.asarray(a)
    if a.dtype == object:
        a = np.array([int(v) for v in a.ravel()], dtype=np.uint64).reshape(a.shape)
        return a
    if a.dtype.kind in "iu" and a.size:
        mx, mn = int(a.max()), int(a.min())
        for dt in (np.uint8, np.uint16, np.uint32, np.int8, np.int16, np.int32):
            if np.iinfo(dt).min <= mn and mx <= np.iinfo(dt).max:
                return a.astype(dt)
    return a


def pack_sage_wide_fields():
    """The three Sage folders of order >= 2^64 (GF(2^100), GF(36893488147419103183), GF(109987^4)): every number is stored
    as a decimal string (elements, exponents and integer multiplicands exceed 64 bits, some are negative)."""
    ops = ["add", "subtract", "multiply", "divide", "additive_inverse", "multiplicative_inverse", "scalar_multiply", "power"]
    for folder in sorted(os.listdir(os.path.join(REF_TESTS, "fields", "data"))):
        path = os.path.join(REF_TESTS, "fields", "data", folder)
        props = json.load(open(os.path.join(path, "properties.json")))
        if props["order"] < 2**64:
            continue
        out = {"properties": np.array(json.dumps(props))}
        for op in ops:
            d = pickle.load(open(os.path.join(path, op + ".pkl"), "rb"))
            for k, v in d.items():
                a = np.array(v, dtype=object)
                out[f"{op}_{k}"] = np.array([str(int(t)) for t in a.ravel()]).reshape(a.shape)
        for op in ("convolve", "matrix_multiply"):  # lists of cases of different shapes: one entry per case
            d = pickle.load(open(os.path.join(path, op + ".pkl"), "rb"))
            out[f"{op}_count"] = np.array(len(d["X"]))
            for k, cases in d.items():
                for i, v in enumerate(cases):
                    a = np.array(v, dtype=object)
                    out[f"{op}_{k}_{i}"] = np.array([str(int(t)) for t in a.ravel()]).reshape(a.shape)
        name = folder.replace("(", "_").replace(")", "").replace("^", "e").replace(", ", "_")
        np.savez_compressed(os.path.join(OUT_DIR, f"sage_wide_{name}.npz"), **out)
        print("packed (wide)", folder)


def pack_sage_wide_linalg():
    """r05: the rest of the three big folders (order >= 2^64) -- row_reduce / lu / plu / inverse / determinant / solve / the four
    spaces (tests/fields/data/*/), log.pkl, and the polynomial evaluations (tests/polys/data/*/evaluate.pkl, evaluate_matrix.pkl):
    what the reference pins its dtype=object linear algebra, discrete logarithm and Horner loops with.  Decimal strings."""
    ops = ["row_reduce", "lu_decompose", "plu_decompose", "matrix_inverse", "matrix_determinant", "matrix_solve", "row_space",
           "column_space", "left_null_space", "null_space"]

    def dec(v):
        a = np.array(v, dtype=object)
        return np.array([str(int(t)) for t in a.ravel()]).reshape(a.shape)

    for folder in sorted(os.listdir(os.path.join(REF_TESTS, "fields", "data"))):
        path = os.path.join(REF_TESTS, "fields", "data", folder)
        props = json.load(open(os.path.join(path, "properties.json")))
        if props["order"] < 2**64:
            continue
        out = {"properties": np.array(json.dumps(props))}
        for op in ops:
            d = pickle.load(open(os.path.join(path, op + ".pkl"), "rb"))
            out[f"{op}_count"] = np.array(len(d["X"]))
            for k, vals in d.items():
                for i, v in enumerate(vals):
                    out[f"{op}{i}_{k}"] = dec(v if not np.isscalar(v) else int(v))
        d = pickle.load(open(os.path.join(path, "log.pkl"), "rb"))
        out["log_X"], out["log_Z"] = dec(d["X"]), dec(d["Z"])
        ppath = os.path.join(REF_TESTS, "polys", "data", folder)
        d = pickle.load(open(os.path.join(ppath, "evaluate.pkl"), "rb"))
        out["evaluate_count"] = np.array(len(d["X"]))
        out["evaluate_Y"] = dec(d["Y"])
        for i, v in enumerate(d["X"]):
            out[f"evaluate{i}_X"] = dec(v.coeffs if hasattr(v, "coeffs") else v)
            out[f"evaluate{i}_Z"] = dec(d["Z"][i])
        d = pickle.load(open(os.path.join(ppath, "evaluate_matrix.pkl"), "rb"))
        out["evaluate_matrix_count"] = np.array(len(d["X"]))
        for k, vals in d.items():
            for i, v in enumerate(vals):
                out[f"evaluate_matrix{i}_{k}"] = dec(v.coeffs if hasattr(v, "coeffs") else v)
        name = folder.replace("(", "_").replace(")", "").replace("^", "e").replace(", ", "_")
        np.savez_compressed(os.path.join(OUT_DIR, f"sage_wide_linalg_{name}.npz"), **out)
        print("packed (wide linalg)", folder)


def pack_sage_fields():
    ops = ["add", "subtract", "multiply", "divide", "additive_inverse", "multiplicative_inverse", "scalar_multiply",
           "power"]
    for folder in sorted(os.listdir(os.path.join(REF_TESTS, "fields", "data"))):
        path = os.path.join(REF_TESTS, "fields", "data", folder)
        props = json.load(open(os.path.join(path, "properties.json")))
        if props["order"] >= 2**64:
            print("skip", folder)
            continue
        out = {"properties": np.array(json.dumps(props))}
        for op in ops:
            d = pickle.load(open(os.path.join(path, op + ".pkl"), "rb"))
            for k, v in d.items():
                out[f"{op}_{k}"] = small(v)
        d = pickle.load(open(os.path.join(path, "convolve.pkl"), "rb"))  # three polynomial products per field
        for i in range(len(d["X"])):
            for k in ("X", "Y", "Z"):
                out[f"convolve{i}_{k}"] = small(d[k][i])
        name = folder.replace("(", "_").replace(")", "").replace("^", "e").replace(", ", "_")
        np.savez_compressed(os.path.join(OUT_DIR, f"sage_fields_{name}.npz"), **out)
        print("packed", folder)


def pack_sage_linalg():
    """tests/fields/data/*/{matrix_multiply,row_reduce,lu_decompose,plu_decompose,matrix_inverse,matrix_determinant,
    matrix_solve,row_space,column_space,left_null_space,null_space}.pkl (loaders tests/fields/conftest.py:296-420)."""
    ops = ["matrix_multiply", "row_reduce", "lu_decompose", "plu_decompose", "matrix_inverse", "matrix_determinant",
           "matrix_solve", "row_space", "column_space", "left_null_space", "null_space"]
    for folder in sorted(os.listdir(os.path.join(REF_TESTS, "fields", "data"))):
        path = os.path.join(REF_TESTS, "fields", "data", folder)
        props = json.load(open(os.path.join(path, "properties.json")))
        if props["order"] >= 2**64:
            continue
        out = {"properties": np.array(json.dumps(props))}
        for op in ops:
            d = pickle.load(open(os.path.join(path, op + ".pkl"), "rb"))
            out[f"{op}_count"] = np.array(len(d["X"]))
            for k, vals in d.items():
                for i, v in enumerate(vals):
                    out[f"{op}{i}_{k}"] = small(np.asarray(v if not np.isscalar(v) else int(v)))
        name = folder.replace("(", "_").replace(")", "").replace("^", "e").replace(", ", "_")
        np.savez_compressed(os.path.join(OUT_DIR, f"sage_linalg_{name}.npz"), **out)
        print("packed linalg", folder)


def pack_sage_polys():
    """tests/polys/data/*/{evaluate,evaluate_matrix,add,subtract,multiply,scalar_multiply,derivative}.pkl (loaders
    tests/polys/conftest.py) and tests/fields/data/*/log.pkl (tests/fields/conftest.py:257-262)."""
    for folder in sorted(os.listdir(os.path.join(REF_TESTS, "polys", "data"))):
        path = os.path.join(REF_TESTS, "polys", "data", folder)
        fpath = os.path.join(REF_TESTS, "fields", "data", folder)
        props = json.load(open(os.path.join(fpath, "properties.json")))
        if props["order"] >= 2**64:
            continue
        out = {"properties": np.array(json.dumps(props))}
        for op in ["add", "subtract", "multiply", "scalar_multiply", "derivative", "evaluate_matrix"]:
            d = pickle.load(open(os.path.join(path, op + ".pkl"), "rb"))
            out[f"{op}_count"] = np.array(len(d["X"]))
            for k, vals in d.items():
                for i, v in enumerate(vals):
                    out[f"{op}{i}_{k}"] = small(np.asarray(v))
        d = pickle.load(open(os.path.join(path, "evaluate.pkl"), "rb"))
        out["evaluate_count"] = np.array(len(d["X"]))
        out["evaluate_Y"] = small(np.asarray(d["Y"]))
        for i, v in enumerate(d["X"]):
            out[f"evaluate{i}_X"] = small(np.asarray(v))
            out[f"evaluate{i}_Z"] = small(np.asarray(d["Z"][i]))
        d = pickle.load(open(os.path.join(fpath, "log.pkl"), "rb"))
        out["log_X"], out["log_Z"] = small(np.asarray(d["X"])), small(np.asarray(d["Z"]))
        for op in ("field_trace", "field_norm", "additive_order", "multiplicative_order"):
            d = pickle.load(open(os.path.join(fpath, op + ".pkl"), "rb"))
            out[f"{op}_X"], out[f"{op}_Z"] = small(np.asarray(d["X"])), small(np.asarray(d["Z"]))
        name = folder.replace("(", "_").replace(")", "").replace("^", "e").replace(", ", "_")
        np.savez_compressed(os.path.join(OUT_DIR, f"sage_polys_{name}.npz"), **out)
        print("packed polys", folder)


def pack_sage_rs():
    out = {}
    names = []
    for f in sorted(glob.glob(os.path.join(REF_TESTS, "codes", "data", "reed_solomon", "*.pkl"))):
        d = pickle.load(open(f, "rb"))
        key = os.path.basename(f)[:-4]
        names.append(key)
        meta = {k: d[k] for k in ("q", "n", "k", "d", "alpha", "c", "is_systematic", "is_primitive", "is_narrow_sense",
                                  "generator_poly")}
        out[f"{key}/meta"] = np.array(json.dumps(meta))
        out[f"{key}/G"] = small(d["G"])
        out[f"{key}/H"] = small(d["H"])
        out[f"{key}/messages"] = small(d["encode"]["messages"])
        out[f"{key}/codewords"] = small(d["encode"]["codewords"])
        if d["encode_shortened"]:
            out[f"{key}/short_messages"] = small(d["encode_shortened"]["messages"])
            out[f"{key}/short_codewords"] = small(d["encode_shortened"]["codewords"])
    out["names"] = np.array(json.dumps(names))
    np.savez_compressed(os.path.join(OUT_DIR, "sage_rs.npz"), **out)
    print("packed", len(names), "RS fixtures")


def _rs_case(out, rng, galois, GFref, tag, order, n, k, c=1, N=12, field_kw=None, shorten=0):
    GF = GFref(order, **(field_kw or {}))
    rs = galois.ReedSolomon(n, k, field=GF, c=c)
    ks, ns = k - shorten, n - shorten
    t = (n - k) // 2
    M = rng.integers(0, order, (N, ks))
    C = np.asarray(rs.encode(GF(M))).astype(np.int64)
    R = C.copy()
    E = np.zeros((N, ns), dtype=bool)
    plan = [(0, 0), (t, 0), (t + 1, 0), (t // 2, 0), (1, 0), (t + 3, 0), (0, 2), (t - 1, 2), (0, n - k), (0, n - k + 1),
            (t // 2, (n - k) - 2 * (t // 2)), (1, n - k)]
    for i in range(N):
        ne, nu = plan[i % len(plan)]
        ne, nu = min(ne, ns), min(nu, ns)
        pos = rng.choice(ns, ne, replace=False)
        R[i, pos] = (R[i, pos] + rng.integers(1, order, ne)) % order
        if nu:
            rest = np.setdiff1d(np.arange(ns), pos)
            epos = rng.choice(rest, min(nu, rest.size), replace=False)
            E[i, epos] = True
            R[i, epos] = rng.integers(0, order, epos.size)
    dec, nerr = rs.decode(GF(R), erasures=E, output="codeword", errors=True)
    out[f"rs/{tag}/meta"] = np.array(json.dumps({"q": order, "n": n, "k": k, "c": c, "alpha": int(rs.alpha),
                                                 "irr": int(GF.irreducible_poly), "p": int(GF.characteristic),
                                                 "m": int(GF.degree), "field_alpha": int(GF.primitive_element)}))
    out[f"rs/{tag}/generator_poly"] = small(rs.generator_poly.coeffs)
    out[f"rs/{tag}/messages"] = small(M)
    out[f"rs/{tag}/codewords"] = small(C)
    out[f"rs/{tag}/received"] = small(R)
    out[f"rs/{tag}/erasures"] = E
    out[f"rs/{tag}/decoded"] = small(dec)
    out[f"rs/{tag}/n_errors"] = np.asarray(nerr, dtype=np.int64)
    out[f"rs/{tag}/detected"] = np.asarray(rs.detect(GF(R)))
    print("rs", tag, list(nerr))


def _bch_case(out, rng, galois, load_reference, tag, n, k=None, d=None, p=2, c=1, systematic=True, N=14, shorten=0, ext_kw=None):
    GFp = load_reference.ref_field(p)
    # the default extension field (_bch.py:207-211), built explicitly so that it is in python-calculate mode (the
    # numba stand-in cannot freeze per-field globals for "jit" Function objects)
    m = galois.ilog(n, p) + 1
    ext = load_reference.ref_field(p**m, irreducible_poly=galois.matlab_primitive_poly(p, m))
    bch = galois.BCH(n, k, d, field=GFp, extension_field=ext, c=c, systematic=systematic)
    k, dd = bch.k, bch.d
    ks, ns = k - shorten, n - shorten
    t = (dd - 1) // 2
    M = rng.integers(0, p, (N, ks))
    C = np.asarray(bch.encode(GFp(M))).astype(np.int64)
    R = C.copy()
    E = np.zeros((N, ns), dtype=bool)
    plan = [(0, 0), (t, 0), (t + 1, 0), (t // 2, 0), (1, 0), (t + 2, 0), (0, 2), (max(t - 1, 0), 2), (0, dd - 1), (0, dd),
            (t // 2, (dd - 1) - 2 * (t // 2)), (1, dd - 1), (t + 3, 0), (2 * t + 1, 0)]
    for i in range(N):
        ne, nu = plan[i % len(plan)]
        ne, nu = min(ne, ns), min(nu, ns)
        pos = rng.choice(ns, ne, replace=False)
        R[i, pos] = (R[i, pos] + rng.integers(1, p, ne)) % p
        if nu:
            rest = np.setdiff1d(np.arange(ns), pos)
            epos = rng.choice(rest, min(nu, rest.size), replace=False)
            E[i, epos] = True
            R[i, epos] = rng.integers(0, p, epos.size)
    # row by row: a miscorrection whose error values fall outside GF(p) makes the reference raise ValueError when it
    # views the decoded int64 array as the base field (_bch.py:1300 -> _fields/_array.py:177); recorded in `raises`
    dec = np.zeros((N, ns), dtype=np.int64)
    msg = np.zeros((N, ks), dtype=np.int64)
    nerr = np.zeros(N, dtype=np.int64)
    raises = np.zeros(N, dtype=bool)
    for i in range(N):
        try:
            if p**m > 256:
                # Syndrome fields above 256 elements: BCH.decode casts the codeword to the base field's uint8 before the
                # stand-in runs the kernel body in place of the compiled function, and extension-field values do not
                # fit.  Call the decoder the way the compiled branch does (bch_decode_jit.__call__, _bch.py:1280-1288:
                # int64 arrays), then apply the same post-processing (_bch.py:1299-1300, _cyclic.py:129-138).
                from galois._codes._bch import bch_decode_jit

                func = bch_decode_jit(bch.field, bch.extension_field)
                d2, n2 = func.python(R[i:i + 1].astype(np.int64), E[i:i + 1].astype(bool), bch.n, int(bch.alpha), bch.c,
                                     np.asarray(bch.roots).astype(np.int64))
                d8 = np.asarray(d2).astype(np.uint8)
                if (d8 >= p).any():
                    raise ValueError("decoded symbols outside the base field")
                dec[i], nerr[i] = d8[0], int(n2[0])
                msg[i] = np.asarray(bch._convert_codeword_to_message(GFp(d8)))[0]
                continue
            d_i, n_i = bch.decode(GFp(R[i]), erasures=E[i], output="codeword", errors=True)
            dec[i], nerr[i] = np.asarray(d_i), n_i
            msg[i] = np.asarray(bch.decode(GFp(R[i]), erasures=E[i], output="message"))
        except (ValueError, OverflowError):
            # OverflowError: in python-calculate mode the decoder works on the uint8 array directly and NumPy 2
            # refuses the negative SUBTRACT_BASE result; the jit path computes it in int64, wraps on astype and
            # then fails the same field-membership check (ValueError).  Either way: an exception.
            raises[i] = True
    ext = bch.extension_field
    out[f"bch/{tag}/meta"] = np.array(json.dumps({
        "p": p, "n": n, "k": int(k), "d": int(dd), "c": c, "alpha": int(bch.alpha), "systematic": systematic,
        "ext_order": int(ext.order), "ext_m": int(ext.degree), "ext_irr": int(ext.irreducible_poly),
        "ext_alpha": int(ext.primitive_element), "shorten": shorten}))
    out[f"bch/{tag}/generator_poly"] = small(bch.generator_poly.coeffs)
    out[f"bch/{tag}/roots"] = small(bch.roots)
    out[f"bch/{tag}/messages"] = small(M)
    out[f"bch/{tag}/codewords"] = small(C)
    out[f"bch/{tag}/received"] = small(R)
    out[f"bch/{tag}/erasures"] = E
    out[f"bch/{tag}/decoded"] = np.asarray(dec).astype(np.int64)
    out[f"bch/{tag}/decoded_message"] = np.asarray(msg).astype(np.int64)
    out[f"bch/{tag}/n_errors"] = np.asarray(nerr, dtype=np.int64)
    out[f"bch/{tag}/raises"] = raises
    out[f"bch/{tag}/detected"] = np.asarray(bch.detect(GFp(R)))
    print("bch", tag, (n, int(k), int(dd)), list(nerr), "raises", list(np.nonzero(raises)[0]))


def reference_outputs():
    import load_reference

    galois = load_reference.load()
    rng = np.random.default_rng(20260925)
    out = {}

    def GFref(order, **kw):
        return load_reference.ref_field(order, **kw)

    # ---- element-wise samples (incl. zeros) ----
    for tag, order, kw in [("gf256", 2**8, {}), ("gf31", 31, {}), ("gf65537", 65537, {}), ("gf7340033", 7340033, {}),
                           ("goldilocks", 2**64 - 2**32 + 1, {}), ("gf2e32", 2**32, {}), ("gf3e5", 3**5, {}),
                           ("gf251e3", 251**3, {})]:
        GF = GFref(order, **kw)
        n = 512
        if order < 2**63:
            a = rng.integers(0, order, n, dtype=np.uint64)
            b = rng.integers(0, order, n, dtype=np.uint64)
        else:
            a = np.array([int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2)) for _ in range(n)], dtype=object) % order
            b = np.array([int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2)) for _ in range(n)], dtype=object) % order
            a = np.array([int(v) for v in a], dtype=np.uint64)
            b = np.array([int(v) for v in b], dtype=np.uint64)
        a[:4] = 0
        b[2:6] = 0
        ga, gb = GF([int(v) for v in a]), GF([int(v) for v in b])
        bnz = np.where(b == 0, 1, b)
        gbnz = GF([int(v) for v in bnz])
        e = rng.integers(-50, 100, n)
        anz = np.where(a == 0, 1, a)
        ganz = GF([int(v) for v in anz])
        out[f"ew/{tag}/meta"] = np.array(json.dumps({"p": int(GF.characteristic), "m": int(GF.degree),
                                                     "irr": int(GF.irreducible_poly), "alpha": int(GF.primitive_element)}))
        out[f"ew/{tag}/a"], out[f"ew/{tag}/b"], out[f"ew/{tag}/e"] = a, b, e
        out[f"ew/{tag}/add"] = small(ga + gb)
        out[f"ew/{tag}/sub"] = small(ga - gb)
        out[f"ew/{tag}/mul"] = small(ga * gb)
        out[f"ew/{tag}/neg"] = small(-ga)
        out[f"ew/{tag}/div"] = small(ga / gbnz)
        out[f"ew/{tag}/recip"] = small(gbnz**-1)
        out[f"ew/{tag}/pow"] = small(ganz**e)
        out[f"ew/{tag}/smul"] = small(ga * 7)
        print("elementwise", tag)

    # ---- NTT known answers from the reference's own tests (tests/fields/test_ntt.py:13-18, SymPy) ----
    kats = [(5, [1, 2, 3, 4], None), (13, [1, 2, 3, 4], None), (17, [1, 2, 3, 4], None), (769, [1, 2, 3, 4], None)]
    for p, x, _ in kats:
        out[f"ntt/kat{p}/x"] = np.array(x)
        out[f"ntt/kat{p}/X"] = small(galois.ntt(x, modulus=p))
    # ---- reference NTT / INTT outputs ----
    for tag, order, n in [("gf65537_256", 65537, 256), ("gf65537_4096", 65537, 4096), ("gf7340033_1024", 7340033, 1024),
                          ("goldilocks_64", 2**64 - 2**32 + 1, 64), ("goldilocks_1024", 2**64 - 2**32 + 1, 1024),
                          ("gf31_30", 31, 30), ("gf31_15", 31, 15), ("gf256_255", 2**8, 255), ("gf256_85", 2**8, 85),
                          ("gf3e5_22", 3**5, 22), ("gf769_96", 769, 96)]:
        GF = GFref(order)
        if order < 2**63:
            x = rng.integers(0, order, n, dtype=np.uint64)
        else:
            x = np.array([(int(rng.integers(0, 2**63)) * 2 + 1) % order for _ in range(n)], dtype=np.uint64)
        gx = GF([int(v) for v in x])
        out[f"ntt/{tag}/order"] = np.array(order, dtype=np.uint64)
        out[f"ntt/{tag}/x"] = x
        out[f"ntt/{tag}/fft"] = small(np.fft.fft(gx))
        out[f"ntt/{tag}/ifft"] = small(np.fft.ifft(gx))
        print("ntt", tag)

    # ---- Reed-Solomon: RS(255,223) (no upstream fixture) and small codes with erasures ----
    def rs_case(tag, order, n, k, c=1, N=12, field_kw=None, shorten=0):
        _rs_case(out, rng, galois, GFref, tag, order, n, k, c, N, field_kw, shorten)

    matlab = galois.matlab_primitive_poly(2, 8)
    rs_case("rs255_223", 2**8, 255, 223, field_kw=dict(irreducible_poly=matlab))
    rs_case("rs255_223_short", 2**8, 255, 223, field_kw=dict(irreducible_poly=matlab), shorten=55)
    rs_case("rs15_9", 2**4, 15, 9)
    rs_case("rs15_11_c3", 2**4, 15, 11, c=3)
    rs_case("rs80_70_gf81", 3**4, 80, 70, c=2)
    rs_case("rs26_20_gf27", 3**3, 26, 20)
    rs_case("rs30_22_gf31", 31, 30, 22)
    rs_case("rs85_65", 2**8, 85, 65)
    # parity of message [0..222] (SURVEY.md 8(c) bootstrap KAT)
    GF = GFref(2**8, irreducible_poly=matlab)
    rs = galois.ReedSolomon(255, 223, field=GF)
    out["rs/kat_arange_parity"] = small(rs.encode(GF(np.arange(223)), output="parity"))
    np.savez_compressed(os.path.join(OUT_DIR, "reference_outputs.npz"), **out)


def pack_sage_bch():
    """tests/codes/data/bch/*.pkl (204 Sage fixtures; loader tests/codes/conftest.py:47-63)."""
    out = {}
    names = []
    for f in sorted(glob.glob(os.path.join(REF_TESTS, "codes", "data", "bch", "*.pkl"))):
        d = pickle.load(open(f, "rb"))
        key = os.path.basename(f)[:-4]
        names.append(key)
        meta = {k: d[k] for k in ("q", "m", "n", "k", "d", "d_min", "alpha", "c", "is_systematic", "is_primitive",
                                  "is_narrow_sense", "generator_poly", "parity_check_poly")}
        out[f"{key}/meta"] = np.array(json.dumps(meta))
        out[f"{key}/G"] = small(d["G"])
        out[f"{key}/H"] = small(d["H"])
        out[f"{key}/messages"] = small(d["encode"]["messages"])
        out[f"{key}/codewords"] = small(d["encode"]["codewords"])
        if d["encode_shortened"]:
            out[f"{key}/short_messages"] = small(d["encode_shortened"]["messages"])
            out[f"{key}/short_codewords"] = small(d["encode_shortened"]["codewords"])
    out["names"] = np.array(json.dumps(names))
    np.savez_compressed(os.path.join(OUT_DIR, "sage_bch.npz"), **out)
    print("packed", len(names), "BCH fixtures")


def reference_bch_outputs():
    """BCH encode / detect / decode outputs of the reference itself (incl. erasures, > t errors and miscorrections)."""
    import load_reference

    galois = load_reference.load()
    rng = np.random.default_rng(20260926)
    out = {}

    def case(tag, n, k=None, d=None, p=2, c=1, systematic=True, N=14, shorten=0, ext_kw=None):
        _bch_case(out, rng, galois, load_reference, tag, n, k, d, p, c, systematic, N, shorten, ext_kw)

    case("bch15_7", 15, 7)
    case("bch15_5_c3", 15, d=7, c=3)
    case("bch31_16", 31, 16)
    case("bch63_45", 63, 45)
    case("bch63_36_short", 63, 36, shorten=20)
    case("bch255_223", 255, 223, N=8)
    case("bch127_99_nonsys", 127, 99, systematic=False, N=8)
    case("bch13_4_gf3", 13, 4, p=3)
    case("bch26_14_gf3", 26, 14, p=3)
    case("bch26_8_gf3_c3", 26, d=9, p=3, c=3)
    case("bch80_60_gf3", 80, d=9, p=3, N=10)
    case("bch24_gf5", 24, d=5, p=5)
    case("bch26_14_gf3_nonsys_short", 26, 14, p=3, systematic=False, shorten=5)
    np.savez_compressed(os.path.join(OUT_DIR, "reference_bch_outputs.npz"), **out)


def reference_wide_codes():
    """Codes whose syndrome field has more than 256 elements (gfa_rs_wide.hip): outputs of the reference itself."""
    import load_reference

    galois = load_reference.load()
    rng = np.random.default_rng(20260927)
    out = {}

    def GFref(order, **kw):
        return load_reference.ref_field(order, **kw)

    _rs_case(out, rng, galois, GFref, "rs1023_1003", 2**10, 1023, 1003, N=8)
    _rs_case(out, rng, galois, GFref, "rs1023_1011_short_c0", 2**10, 1023, 1011, c=0, N=8, shorten=600)
    _rs_case(out, rng, galois, GFref, "rs728_712_gf729", 3**6, 728, 712, N=8)
    _rs_case(out, rng, galois, GFref, "rs511_501", 2**9, 511, 501, c=2, N=6)
    _bch_case(out, rng, galois, load_reference, "bch511_493", 511, 493, N=8)
    _bch_case(out, rng, galois, load_reference, "bch1023_1003", 1023, 1003, N=8)
    _bch_case(out, rng, galois, load_reference, "bch1023_973_short", 1023, 973, N=6, shorten=500)
    _bch_case(out, rng, galois, load_reference, "bch728_gf3", 728, d=7, p=3, N=8)
    _bch_case(out, rng, galois, load_reference, "bch511_484_nonsys", 511, 484, systematic=False, N=6)
    np.savez_compressed(os.path.join(OUT_DIR, "reference_wide_codes.npz"), **out)


TABLE_FIELDS = [("gf3e7", 3**7), ("gf2e10", 2**10), ("gf2e13", 2**13), ("gf5e5", 5**5), ("gf8191", 8191), ("gf2e14", 2**14),
                ("gf3e9", 3**9), ("gf2e15", 2**15), ("gf32749", 32749), ("gf2e16", 2**16), ("gf3e10", 3**10), ("gf65521", 65521),
                ("gf251e2", 251**2)]


def reference_table_fields():
    """Fields of 257 .. 65536 elements (the size classes of csrc/gfa_elementwise_mid.hip): element-wise outputs of the reference
    itself incl. zeros among the operands, one exponent per element and one for the whole array, and mixed-radix transforms of
    the reference's FFT benchmark sizes over such fields."""
    import load_reference

    load_reference.load()
    rng = np.random.default_rng(20260928)
    out = {}
    for tag, order in TABLE_FIELDS:
        GF = load_reference.ref_field(order)
        n = 256
        a = rng.integers(0, order, n, dtype=np.uint64)
        b = rng.integers(0, order, n, dtype=np.uint64)
        a[:4] = 0
        b[2:6] = 0
        a[6:10] = b[6:10]                      # a - a, a / a
        a[10], b[10] = order - 1, order - 1
        a[11], b[11] = 1, order - 1
        bnz, anz = np.where(b == 0, 1, b), np.where(a == 0, 1, a)
        ga, gb, gbnz, ganz = (GF([int(v) for v in x]) for x in (a, b, bnz, anz))
        e = rng.integers(-300, 300, n)
        e[:6] = [0, 1, -1, order - 1, -(order - 1), order - 2]
        out[f"ew/{tag}/meta"] = np.array(json.dumps({"p": int(GF.characteristic), "m": int(GF.degree),
                                                     "irr": int(GF.irreducible_poly), "alpha": int(GF.primitive_element)}))
        out[f"ew/{tag}/a"], out[f"ew/{tag}/b"], out[f"ew/{tag}/e"] = a, b, e
        out[f"ew/{tag}/add"] = small(ga + gb)
        out[f"ew/{tag}/sub"] = small(ga - gb)
        out[f"ew/{tag}/mul"] = small(ga * gb)
        out[f"ew/{tag}/neg"] = small(-ga)
        out[f"ew/{tag}/div"] = small(ga / gbnz)
        out[f"ew/{tag}/recip"] = small(gbnz**-1)
        out[f"ew/{tag}/pow"] = small(ganz**e)
        out[f"ew/{tag}/pow12345"] = small(ga**12345)
        out[f"ew/{tag}/pow_minus7"] = small(ganz**-7)
        print("table field", tag, flush=True)
    for tag, order, n in [("gf769_768", 769, 768), ("gf7681_1536", 7681, 1536), ("gf127e2_2304", 127**2, 2304), ("gf2e12_4095", 2**12, 4095),
                          ("gf3e7_1093", 3**7, 1093)]:
        GF = load_reference.ref_field(order)
        x = rng.integers(0, order, n, dtype=np.uint64)
        out[f"ntt/{tag}/meta"] = np.array(json.dumps({"p": int(GF.characteristic), "m": int(GF.degree),
                                                      "irr": int(GF.irreducible_poly), "alpha": int(GF.primitive_element)}))
        out[f"ntt/{tag}/x"] = x
        out[f"ntt/{tag}/fft"] = small(np.fft.fft(GF([int(v) for v in x])))
        print("mixed-radix transform", tag, flush=True)
    np.savez_compressed(os.path.join(OUT_DIR, "reference_table_fields.npz"), **out)


if __name__ == "__main__":
    what = sys.argv[1:] or ["fields", "rs", "reference", "bch", "reference_bch", "reference_wide", "reference_tables", "linalg", "polys"]
    if "fields" in what:
        pack_sage_fields()
        pack_sage_wide_fields()
    if "rs" in what:
        pack_sage_rs()
    if "reference" in what:
        reference_outputs()
    if "polys" in what:
        pack_sage_polys()
    if "linalg" in what:
        pack_sage_linalg()
        pack_sage_wide_linalg()
    if "bch" in what:
        pack_sage_bch()
    if "reference_bch" in what:
        reference_bch_outputs()
    if "reference_wide" in what:
        reference_wide_codes()
    if "reference_tables" in what:
        reference_table_fields()
    print("done")
