"""
galois_amd -- an MI355X (gfx950) finite-field engine behind the galois.GF / FieldArray / ntt / ReedSolomon surface.

Scope: element-wise field ufuncs, np.fft.fft/ifft + ntt/intt, ReedSolomon encode/detect/decode.
All data-path work runs in hand-written HIP kernels reached through the C-ABI of include/galois_amd.h; there is no
CPU arithmetic path (importing the package without the built library raises ImportError).
"""
from . import _lib as _lib_module

_lib_module.lib()  # fail loudly if the HIP extension has not been built

from ._array import FieldArray  # noqa: E402
from ._factory import GF, Field  # noqa: E402
from ._ntt import ntt, intt  # noqa: E402
from ._codes import BCH, ReedSolomon  # noqa: E402
from ._poly import Poly, berlekamp_massey  # noqa: E402
from ._numtheory import (  # noqa: E402
    is_prime, factors, primitive_root, is_primitive_root, matlab_primitive_poly, conway_poly, primitive_poly,
)
from . import _dist as dist  # noqa: E402
from . import _linalg as linalg  # noqa: E402

GF2 = GF(2)

__all__ = [
    "FieldArray", "GF", "GF2", "Field", "ntt", "intt", "ReedSolomon", "BCH", "Poly", "berlekamp_massey", "is_prime", "factors", "primitive_root",
    "is_primitive_root", "matlab_primitive_poly", "conway_poly", "primitive_poly", "dist", "linalg",
]
__version__ = "0.1.0"
