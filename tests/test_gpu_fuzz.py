"""Short seeded runs of the differential fuzzers (tools/fuzz_codes.py [wide], tools/fuzz_fields.py, tools/fuzz_ntt_linalg.py, tools/fuzz_conv_ntt3.py, tools/fuzz_table_fields.py, tools/fuzz_r04.py, tools/fuzz_r05.py, tools/fuzz_r06.py): random codes / fields / shapes
against the oracle.  Longer campaigns: `python tools/fuzz_codes.py 600 <seed>`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seed,extra", [("fuzz_codes.py", 7, []), ("fuzz_fields.py", 11, []), ("fuzz_ntt_linalg.py", 13, []),
                                             ("fuzz_codes.py", 17, ["wide"]), ("fuzz_conv_ntt3.py", 19, []), ("fuzz_table_fields.py", 23, []),
                                             ("fuzz_r04.py", 41, []), ("fuzz_r05.py", 51, []), ("fuzz_r06.py", 61, [])])
def test_fuzz(tool, seed, extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "8", str(seed)] + extra, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "identical to the oracle" in r.stdout


def test_fuzz_codes_with_the_replicated_lfsr_table_forced():
    """rs_lfsr_kernel's four-copy table layout is chosen for batches of 2^16 words and more; GFA_RS_LFSR_REP4=2 forces it on the
    fuzzer's small batches (codes with n - k = 16 or 32), so that every code shape the fuzzer draws goes through it too."""
    env = dict(os.environ, GFA_RS_LFSR_REP4="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_codes.py"), "8", "29"], capture_output=True, text=True, timeout=600,
                       env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "identical to the oracle" in r.stdout
