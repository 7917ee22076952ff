#!/bin/bash
# Static ISA evidence for the instruction counts DESIGN.md quotes (no GPU needed: hipcc -S for gfx950).
#   bash tools/isa_excerpts.sh r03    ->  profiles/r03_isa_*.txt
set -eu
R=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -S --cuda-device-only -I$ROOT/include"
/opt/rocm/bin/hipcc $FLAGS -o $TMP/rs.s $ROOT/galois_amd/csrc/gfa_rs.hip 2> /dev/null
/opt/rocm/bin/hipcc $FLAGS -o $TMP/ew.s $ROOT/galois_amd/csrc/gfa_elementwise.hip 2> /dev/null
kernel() { # <asm file> <mangled-name substring> -> the kernel's body on stdout
    local a b
    a=$(grep -n "^_ZN.*$2.*:" $1 | head -1 | cut -d: -f1)
    b=$(grep -n "\.amdhsa_kernel _ZN.*$2" $1 | head -1 | cut -d: -f1)
    sed -n "${a},${b}p" $1
}
kernel $TMP/rs.s "rs_decode_bin_kernelILi40ELi8" > $TMP/dec.s
{
    echo "# rs_decode_bin_kernel<40, 8>, gfx950, $(cd $ROOT && git log --oneline | head -1)"
    echo "# whole kernel: $(grep -c '^\s*v_' $TMP/dec.s) vector, $(grep -c '^\s*s_' $TMP/dec.s) scalar, $(grep -c '^\s*ds_' $TMP/dec.s) LDS instructions (static)"
    grep "rs_decode_bin_kernelILi40ELi8.*\(num_vgpr\|numbered_sgpr\)," $TMP/rs.s | sed 's/.*\.\(num_vgpr\|numbered_sgpr\), /#   \1 = /'
    echo
    echo "## Berlekamp-Massey (bm_run): the loop as assembled"
    awk '/BM_TOP[0-9]*:/{p=1} p{print} /BM_END[0-9]*:/{if(p){exit}}' $TMP/dec.s
    echo
    echo "## syndromes, n - k = 32: two chains side by side, one gather + one SDWA xor per term and chain (first four steps)"
    l=$(grep -n "src1_sel:BYTE_1" $TMP/dec.s | head -1 | cut -d: -f1)
    sed -n "$((l - 8)),$((l + 24))p" $TMP/dec.s | grep -v "ASMSTART\|ASMEND"
    echo
    echo "## Chien search: the loop over the locator's coefficients (four points per lane)"
    python3 - $TMP/dec.s <<'PY'
import re, sys
cur, name, best = [], "entry", None
for ln in open(sys.argv[1]):
    if re.match(r"^\.LBB\d+_\d+:", ln):
        if sum("src1_sel:DWORD" in x for x in cur) >= 4 and sum("ds_read_u8" in x for x in cur) >= 4 and best is None: best = (name, cur)
        cur, name = [], ln.split(":")[0]
    else:
        cur.append(ln)
print(best[0] + ":")
print("".join(x for x in best[1] if "ASMSTART" not in x and "ASMEND" not in x and not x.lstrip().startswith(";")), end="")
PY
} > $ROOT/profiles/${R}_isa_rs_decode.txt
kernel $TMP/rs.s "rs_lfsr_kernelILi8ELb1ELb1" > $TMP/lfsr.s
kernel $TMP/ew.s "bin16_holes_mul_kernel" > $TMP/holes.s
python3 - $TMP/lfsr.s $TMP/holes.s > $ROOT/profiles/${R}_isa_loops.txt <<'PY'
import re, sys, collections
def blocks(path):
    out, cur, name = [], [], "entry"
    for ln in open(path):
        if re.match(r"^\.LBB\d+_\d+:", ln):
            out.append((name, cur)); cur = []; name = ln.split(":")[0]
        else:
            cur.append(ln)
    out.append((name, cur))
    return out
def report(title, path, key, per):
    best = max(blocks(path), key=lambda b: sum(1 for x in b[1] if key in x))
    ops = [x.split()[0] for x in best[1] if re.match(r"^\s+(v_|ds_|s_|global_|buffer_)", x)]
    c = collections.Counter(ops)
    v = sum(n for o, n in c.items() if o.startswith("v_"))
    print(f"## {title}: main loop block {best[0]}: {v} vector, {sum(n for o, n in c.items() if o.startswith('ds_'))} LDS, "
          f"{sum(n for o, n in c.items() if o.startswith('s_') and not o.startswith('s_waitcnt') and not o.startswith('s_nop'))} scalar instructions {per}")
    for o, n in sorted(c.items(), key=lambda t: -t[1]):
        if o.startswith(("v_", "ds_")): print(f"    {n:4d}  {o}")
    print()
report("rs_lfsr_kernel<8, encode, four table copies> (planar state)", sys.argv[1], "ds_read_b128", "per 4 symbols and lane")
report("bin16_holes_mul_kernel", sys.argv[2], "v_mul_hi_u32", "per 8 elements (one 16-byte vector)")
PY
rm -rf $TMP
ls -la $ROOT/profiles/${R}_isa_*.txt
