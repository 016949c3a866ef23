"""CPU oracle of the TLS search path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package (as the checker / the timed CPU baseline).  tls_amd never does.
See oracle/tls_oracle.c for the reference file:line map and the parity pins.
"""
from .oracle import OracleLibrary, search, build, usable_cores  # noqa: F401
