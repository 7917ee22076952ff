"""Regenerates galois_amd/data/conway_polys.txt from the reference's Conway-polynomial database.

Frank Luebeck's Conway polynomial list is published mathematical data; the reference ships it as an SQLite file
(/root/reference/src/galois/_databases/conway_polys.db, schema in scripts/create_conway_polys_database.py).  Only runs
in the build container.  Subset kept: p <= 1000 with p^m < 2^64 (the device limit), plus the fields the reference's
own test-suite uses.
"""
import os
import sqlite3

DB = "/root/reference/src/galois/_databases/conway_polys.db"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "galois_amd", "data", "conway_polys.txt")
EXTRA = {(109987, 4), (2, 100)}

con = sqlite3.connect(f"file:{DB}?mode=ro", uri=True)
rows = con.execute("select characteristic, degree, nonzero_degrees, nonzero_coeffs from polys").fetchall()
lines = []
for p, m, degs, coeffs in rows:
    if m == 1:
        continue
    if not ((p <= 1000 and p**m < 2**64) or (p, m) in EXTRA):
        continue
    value = sum(int(c) * p ** int(d) for d, c in zip(degs.split(","), coeffs.split(",")))
    lines.append((p, m, value))
lines.sort()
with open(OUT, "w") as fh:
    fh.write("# Conway polynomials C_{p,m} (Frank Luebeck's list): 'p m integer' with integer = sum c_i * p^i\n")
    for p, m, v in lines:
        fh.write(f"{p} {m} {v}\n")
print(len(lines), "entries ->", OUT)
