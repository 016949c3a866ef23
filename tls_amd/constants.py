"""Numeric contract of the search: defaults and physical constants.

The VALUES are part of the drop-in contract (reference: transitleastsquares/
tls_constants.py:17-148); a search with no kwargs must use exactly these.
"""
import os

VERSION = "1.0.31"  # reference version this build is a drop-in for (version.py:1-2)
BACKEND_BANNER = "Transit Least Squares TLS %s (MI355X/HIP backend)" % VERSION

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# physical constants, SI (tls_constants.py:20-25)
G = 6.673e-11
R_sun = 695508000
R_earth = 6371000
R_jup = 69911000
M_sun = 1.989 * 10 ** 30
SECONDS_PER_DAY = 86400

# search defaults (tls_constants.py:28-41)
TRANSIT_DEPTH_MIN = 10 * 10 ** -6
NUMERICAL_STABILITY_CUTOFF = 0.01 * 10 ** -6
R_STAR = 1.0
M_STAR = 1.0
OVERSAMPLING_FACTOR = 3
N_TRANSITS_MIN = 2
M_STAR_MIN = 0.1
M_STAR_MAX = 1.0
R_STAR_MIN = 0.13
R_STAR_MAX = 3.5
DURATION_GRID_STEP = 1.1

# template presets (tls_constants.py:47-66)
DEFAULT_U = [0.4804, 0.1867]
DEFAULT_LIMB_DARK = "quadratic"
DEFAULT_ECC = 0
DEFAULT_W = 90
DEFAULT_PERIOD = 12.9
DEFAULT_RP = 0.03
DEFAULT_A = 23.1
DEFAULT_INC = 89.21
GRAZING_B = 0.99
BOX_PERIOD = 29
BOX_RP = 0.1
BOX_A = 26.9
BOX_B = 0
BOX_INC = 90
BOX_U = [0]
BOX_LIMB_DARK = "linear"

SIGNAL_DEPTH = 0.5  # depth every cached template row is normalised to (tls_constants.py:71)
FRACTIONAL_TRANSIT_DURATION_MAX = 0.12  # cap of T14 (tls_constants.py:78)
SUPERSAMPLE_SIZE = 10000  # template supersampling (tls_constants.py:89)
OVERSAMPLE_MODEL_LIGHT_CURVE = 5  # tls_constants.py:90
PERIODS_SEARCH_ORDER = "shuffled"  # tls_constants.py:94 (RNG side effect kept, see api.py)
SDE_MEDIAN_KERNEL_SIZE = 30  # tls_constants.py:100
T0_FIT_MARGIN = 0.01  # tls_constants.py:109
PROGRESSBAR_THRESHOLD = 5000  # tls_constants.py:114
MINIMUM_PERIOD_GRID_SIZE = 100  # tls_constants.py:118

# kwargs power() accepts without a warning (tls_constants.py:121-148)
VALID_PARAMETERS = (
    "R_star", "R_star_min", "R_star_max", "M_star", "M_star_min", "M_star_max",
    "period_min", "period_max", "n_transits_min", "per", "rp", "a", "inc", "b",
    "ecc", "w", "u", "limb_dark", "duration_grid_step", "transit_depth_min",
    "oversampling_factor", "T0_fit_margin", "use_threads", "show_progress_bar",
    "transit_template", "verbose",
)
# extensions of this build; never collide with the reference's names
EXTRA_PARAMETERS = ("device", "context", "devices")
