"""Build tls_amd/data/fap_sde_milli.npy from the reference's FAP lookup table.

The reference ships an empirical table (transitleastsquares/fap.csv: false-alarm
probability vs. SDE threshold, 1252 rows).  Its FAP column is analytic,
FAP[k] = round(max(1251 - k, 1) / 12495, 9) with FAP[0] = NaN; only the SDE
thresholds are data.  We keep those (3 decimals -> int16 milli-SDE; the last row
is +inf) and rebuild the FAP column in tls_amd.stats.FAP.  This script checks the
rebuilt table against the csv, value for value.  Build container only.
"""
import os
import sys

import numpy

REF = "/root/reference/transitleastsquares/fap.csv"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "tls_amd", "data", "fap_sde_milli.npy")

data = numpy.genfromtxt(REF, dtype="f8, f8", delimiter=",", names=["FAP", "SDE"])
sde = data["SDE"]
assert numpy.isinf(sde[-1]) and numpy.isnan(data["FAP"][0])
milli = numpy.round(sde[:-1] * 1000).astype(numpy.int16)
assert numpy.array_equal(milli / 1000.0, sde[:-1])
numpy.save(OUT, milli)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd.stats import _fap_table  # noqa: E402

fap, thr = _fap_table()
assert numpy.array_equal(thr, sde)
assert numpy.array_equal(fap[1:], data["FAP"][1:]) and numpy.isnan(fap[0])
print("wrote", OUT, len(milli), "thresholds; rebuilt table identical to the reference csv")
