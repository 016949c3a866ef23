"""Import harness for the UNMODIFIED reference package (build container only).

The reference (/root/reference, pure Python) needs two third-party packages
that are absent from this image: `numba` (only used as a JIT decorator) and
`batman` (transit light curves).  This module writes two tiny stand-in packages
into a temporary directory and puts it, together with /root/reference, on
sys.path, so that `import transitleastsquares` runs the reference's own code:

* numba: jit/njit -> identity decorators; guvectorize -> a plain-Python driver
  for the two 1-D signatures the reference uses (interpolation.py:46,55);
* batman: TransitParams/TransitModel delegating to tls_amd.transit_model (this
  repo's restatement of the published Mandel & Agol algorithm).

It is used ONLY by tools/gen_golden.py (golden-vector generation) and by
developer checks; nothing under tls_amd/, tests/, bench.py or
__graft_entry__.py imports it, and /root/reference does not exist on the GPU box.
"""
import os
import sys
import tempfile
import textwrap

REFERENCE_ROOT = "/root/reference"

_NUMBA = '''
import numpy as _np

def _identity(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    def deco(f):
        return f
    return deco

jit = njit = vectorize = _identity

class _GU(object):
    """Non-function callable: must not bind as a method (interpolation.py:55)."""
    def __init__(self, f, sig):
        self.f, self.sig = f, sig.replace(" ", "")
    def __call__(self, *a):
        if self.sig == "(m),(m),(n)->(m)":
            index, theta, y = a
            out = _np.empty(len(index), dtype=float)
            self.f(_np.asarray(index), _np.asarray(theta), _np.asarray(y, dtype=float), out)
            return out
        if self.sig == "(),(n)->(),()":
            x_new, x = a
            x_new = _np.atleast_1d(_np.asarray(x_new, dtype=float))
            idx = _np.zeros(len(x_new), dtype=_np.int64)
            th = _np.zeros(len(x_new), dtype=float)
            for k in range(len(x_new)):
                self.f(x_new[k:k + 1], _np.asarray(x, dtype=float), idx[k:k + 1], th[k:k + 1])
            return idx, th
        raise NotImplementedError(self.sig)

def guvectorize(types, sig, **kw):
    def deco(f):
        return _GU(f, sig)
    return deco
'''

_BATMAN = '''
from tls_amd.transit_model import TransitParams, TransitModel
'''


def activate():
    """Make `import transitleastsquares` resolve to the reference. Idempotent."""
    if "transitleastsquares" in sys.modules:
        return sys.modules["transitleastsquares"]
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present (this only works in the build container)")
    stub_dir = tempfile.mkdtemp(prefix="tls_ref_shim_")
    for name, body in (("numba", _NUMBA), ("batman", _BATMAN)):
        os.makedirs(os.path.join(stub_dir, name))
        with open(os.path.join(stub_dir, name, "__init__.py"), "w") as fh:
            fh.write(textwrap.dedent(body))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (repo, REFERENCE_ROOT, stub_dir):
        if p not in sys.path:
            sys.path.insert(0, p)
    import transitleastsquares  # noqa: F401  (the reference)
    return transitleastsquares
