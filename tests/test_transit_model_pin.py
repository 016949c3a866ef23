"""Independent pin of tls_amd/transit_model.py (the reference's template comes from
batman-package through transit.py:8-42, absent from this image).

Nothing here shares code or formulation with the module under test:

* the light curve is the DEFINITION -- one minus the stellar intensity integrated over the part of the disc the
  planet covers, divided by its integral over the whole disc -- evaluated by nested adaptive quadrature in polar
  coordinates centred on the PLANET (the module integrates rings around the stellar centre with fixed nodes, or
  uses the Mandel & Agol closed form);
* the sky-projected separation of an eccentric orbit comes from a bracketed root of Kepler's equation and the
  rotation of the orbital-plane position vector (the module: Newton iteration and the true anomaly).

The integrator is first checked on the one case with an elementary answer (uniform source: area of a lens).
Then: exact-K/E variant of the closed form <= 1e-12 (measured 1e-15), shipped Hastings-polynomial variant
<= 1e-8 absolute (measured 9.4e-9; SURVEY Appendix A: the polynomials are what reproduces the reference's known
answers), for the reference's three template presets (tls_constants.py:47-66), three eccentric orbits, and
every other limb-darkening law (<= 1e-12, measured 3e-14).
"""
import math

import numpy
import pytest
from scipy import integrate, optimize, special

from tls_amd import constants as C
from tls_amd import transit_model as tm

# (quad is asked for more digits than it can certify; what it delivers is checked against the lens area below)
pytestmark = pytest.mark.filterwarnings("ignore::scipy.integrate.IntegrationWarning")


# ---- the definition, by quadrature ---------------------------------------------------------------------------
def intensity(mu, law, u):
    """I(mu) / I(1) of the limb-darkening laws, as tabulated in Kreidberg (2015), Table 1."""
    if law == "uniform":
        return 1.0
    if law == "linear":
        return 1.0 - u[0] * (1.0 - mu)
    if law == "quadratic":
        return 1.0 - u[0] * (1.0 - mu) - u[1] * (1.0 - mu) ** 2
    if law == "squareroot":
        return 1.0 - u[0] * (1.0 - mu) - u[1] * (1.0 - math.sqrt(mu))
    if law == "logarithmic":
        return 1.0 - u[0] * (1.0 - mu) - (u[1] * mu * math.log(mu) if mu > 0.0 else 0.0)
    if law == "exponential":
        # (singular at the limb like -1/mu, integrable: r dr = -mu dmu; the clamp only guards a rounded mu = 0)
        return 1.0 - u[0] * (1.0 - mu) + u[1] / math.expm1(max(mu, 1e-14))
    if law == "power2":
        return 1.0 - u[0] * (1.0 - mu ** u[1])
    if law == "nonlinear":
        return 1.0 - sum(u[k] * (1.0 - mu ** ((k + 1) / 2.0)) for k in range(4))
    raise ValueError(law)


def disc_flux(law, u):
    """Integral of I over the unit disc: 2 pi int_0^1 I(mu) mu dmu (r dr = -mu dmu)."""
    val, _ = integrate.quad(lambda mu: intensity(mu, law, u) * mu, 0.0, 1.0, epsabs=1e-15, epsrel=1e-14, limit=400)
    return 2.0 * math.pi * val


def blocked_flux(z, p, law, u):
    """Integral of I over {points of the planet's disc (radius p, centre at distance z) that lie on the star},
    in polar coordinates (rho, phi) about the planet's centre; phi is measured from the direction AWAY from the
    star's centre, so a point is at r^2 = z^2 + rho^2 + 2 z rho cos(phi) from it."""
    if z >= 1.0 + p:
        return 0.0

    def ring(rho):
        # the part of the circle of radius rho that lies on the star: |phi| > phi_star, cos(phi_star) from r = 1
        if z + rho <= 1.0:
            phi0 = 0.0                                   # the whole circle
        elif abs(z - rho) >= 1.0:
            return 0.0                                   # none of it
        else:
            phi0 = math.acos(max(-1.0, min(1.0, (1.0 - z * z - rho * rho) / (2.0 * z * rho))))

        def along(s):
            # phi = pi - (pi - phi0) (1 - s^2): the limb (square-root behaviour of mu) sits at s = 1 ... substitute
            # so that mu is smooth in s near the limb: phi(s) = phi0 + (pi - phi0) * (1 - (1 - s)^2) would do the
            # opposite end; the limb is at phi0, so use phi = phi0 + (pi - phi0) s^2, dphi = 2 (pi - phi0) s ds
            phi = phi0 + (math.pi - phi0) * s * s
            r2 = z * z + rho * rho + 2.0 * z * rho * math.cos(phi)
            mu = math.sqrt(max(1.0 - r2, 0.0))
            return intensity(mu, law, u) * 2.0 * (math.pi - phi0) * s

        val, _ = integrate.quad(along, 0.0, 1.0, epsabs=1e-15, epsrel=1e-13, limit=200)
        return 2.0 * val * rho                           # both halves of the circle

    pts = sorted(x for x in (abs(1.0 - z), ) if 0.0 < x < p)
    val, _ = integrate.quad(ring, 0.0, p, points=pts or None, epsabs=1e-15, epsrel=1e-13, limit=400)
    return val


def quadrature_flux(z, p, law, u):
    total = disc_flux(law, u)
    return numpy.array([1.0 - blocked_flux(float(zz), p, law, u) / total for zz in z])


def lens_area(z, p):
    """Area common to the unit disc and a disc of radius p at centre distance z (elementary geometry)."""
    if z >= 1.0 + p:
        return 0.0
    if z <= abs(1.0 - p):
        return math.pi * min(1.0, p) ** 2
    a = math.acos((z * z + 1.0 - p * p) / (2.0 * z))
    b = math.acos((z * z + p * p - 1.0) / (2.0 * z * p))
    return a + p * p * b - 0.5 * math.sqrt((-z + 1 + p) * (z + 1 - p) * (z - 1 + p) * (z + 1 + p))


# ---- an eccentric orbit, from the position vector ---------------------------------------------------------------
def separation_by_vectors(t, t0, per, a, inc_deg, ecc, w_deg):
    """Sky-projected separation: solve Kepler's equation by a bracketed root finder, place the planet in its
    orbital plane (x towards periastron), rotate by the argument of periastron and tilt by the inclination.
    The observer looks down the +Z axis; the conjunction (mid-transit) is where the planet crosses X = 0 in front."""
    inc, w = math.radians(inc_deg), math.radians(w_deg)

    def position(mean_anomaly):
        M = math.fmod(mean_anomaly, 2.0 * math.pi)
        if M < 0:
            M += 2.0 * math.pi
        E = optimize.brentq(lambda x: x - ecc * math.sin(x) - M, -0.1, 2.0 * math.pi + 0.1, xtol=1e-15, rtol=1e-15)
        xo, yo = a * (math.cos(E) - ecc), a * math.sqrt(1.0 - ecc * ecc) * math.sin(E)   # orbital plane
        xr = xo * math.cos(w) - yo * math.sin(w)          # rotate by the argument of periastron
        yr = xo * math.sin(w) + yo * math.cos(w)
        return xr, yr * math.cos(inc), yr * math.sin(inc)  # X, Y on the sky, Z towards the observer

    # mean anomaly of the conjunction: the point of the orbit with xr = 0 and yr > 0, by root finding in M
    def xr_of(M):
        return position(M)[0]
    # true anomaly there is pi/2 - w; bracket its mean anomaly numerically
    grid = numpy.linspace(0.0, 2.0 * math.pi, 721)
    vals = [xr_of(m) for m in grid]
    M_conj = None
    for i in range(len(grid) - 1):
        if vals[i] == 0.0 or vals[i] * vals[i + 1] < 0.0:
            m = optimize.brentq(xr_of, grid[i], grid[i + 1], xtol=1e-15, rtol=1e-15)
            if position(m)[2] > 0.0:
                M_conj = m
                break
    assert M_conj is not None
    out = numpy.empty(len(t))
    for j, tj in enumerate(t):
        X, Y, Z = position(M_conj + 2.0 * math.pi * (tj - t0) / per)
        out[j] = math.hypot(X, Y) if Z > 0.0 else 1.0e10
    return out


# ---- the checks ----------------------------------------------------------------------------------------------
def test_the_integrator_itself_on_the_uniform_source():
    for p in (0.03, 0.1, 0.6):
        for z in (0.0, 0.3, 1.0 - p, 1.0 - 0.5 * p, 1.0, 1.0 + 0.7 * p, 1.0 + p):
            want = lens_area(z, p)
            assert abs(blocked_flux(z, p, "uniform", []) - want) <= 2e-14, (p, z)


def _presets():
    """(label, per, rp, a, inc, u, law) of the three template presets (tls_constants.py:47-66, validate.py:101-119)."""
    grazing_inc = math.degrees(math.acos(C.GRAZING_B / C.DEFAULT_A))
    return [
        ("default", C.DEFAULT_PERIOD, C.DEFAULT_RP, C.DEFAULT_A, C.DEFAULT_INC, C.DEFAULT_U, "quadratic"),
        ("grazing", C.DEFAULT_PERIOD, C.DEFAULT_RP, C.DEFAULT_A, grazing_inc, C.DEFAULT_U, "quadratic"),
        ("box", C.BOX_PERIOD, C.BOX_RP, C.BOX_A, C.BOX_INC, C.BOX_U, "linear"),
    ]


def _sample_times(per, a, rp, inc, n=41):
    """Times across the transit, contacts included, a few outside."""
    b = a * math.cos(math.radians(inc))
    chord = math.sqrt(max((1.0 + rp) ** 2 - b * b, 0.0))
    half = 1.3 * per / (2 * math.pi) * math.asin(min(1.0, chord / a))
    return numpy.linspace(-half, half, n)


@pytest.fixture
def exact_elliptic(monkeypatch):
    """The closed form with exact complete elliptic integrals instead of the Hastings polynomials."""
    monkeypatch.setattr(tm, "ellip_k", lambda k: special.ellipk(numpy.asarray(k) ** 2))
    monkeypatch.setattr(tm, "ellip_e", lambda k: special.ellipe(numpy.asarray(k) ** 2))


@pytest.mark.parametrize("preset", _presets(), ids=lambda p: p[0])
def test_closed_form_with_exact_elliptic_integrals_is_the_definition(preset, exact_elliptic):
    _, per, rp, a, inc, u, law = preset
    t = _sample_times(per, a, rp, inc)
    z = separation_by_vectors(t, 0.0, per, a, inc, 0.0, 90.0)
    uu = list(u) + [0.0] * (2 - len(u))
    want = quadrature_flux(z, rp, "quadratic", uu)
    got = tm.light_curve(t, 0.0, per, rp, a, inc, 0, 90, u, law)
    assert (want < 1.0).sum() >= 10                    # the sample does cross the transit
    numpy.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


@pytest.mark.parametrize("preset", _presets(), ids=lambda p: p[0])
def test_shipped_hastings_variant_stays_within_1e8(preset):
    _, per, rp, a, inc, u, law = preset
    t = _sample_times(per, a, rp, inc)
    z = separation_by_vectors(t, 0.0, per, a, inc, 0.0, 90.0)
    uu = list(u) + [0.0] * (2 - len(u))
    want = quadrature_flux(z, rp, "quadratic", uu)
    got = tm.light_curve(t, 0.0, per, rp, a, inc, 0, 90, u, law)
    numpy.testing.assert_allclose(got, want, rtol=0, atol=1e-8)


@pytest.mark.parametrize("ecc,w", [(0.2, 60.0), (0.45, 200.0), (0.1, 310.0)])
def test_eccentric_orbits(ecc, w, exact_elliptic):
    per, rp, a, inc, u = 7.3, 0.08, 14.0, 88.3, [0.35, 0.22]
    t = numpy.linspace(-0.25, 0.25, 31)
    z_want = separation_by_vectors(t, 0.0, per, a, inc, ecc, w)
    z_got = tm.projected_separation(t, 0.0, per, a, inc, ecc, w)
    front = z_want < 1.0e9
    assert numpy.array_equal(front, z_got < 1.0e9) and front.all()
    numpy.testing.assert_allclose(z_got, z_want, rtol=0, atol=2e-12)
    want = quadrature_flux(z_want, rp, "quadratic", u)
    assert (want < 1.0).sum() >= 5
    numpy.testing.assert_allclose(tm.light_curve(t, 0.0, per, rp, a, inc, ecc, w, u, "quadratic"), want, rtol=0, atol=1e-12)
    # the transit is centred on t0 by construction of both
    assert abs(t[numpy.argmin(z_want)]) <= t[1] - t[0]


@pytest.mark.parametrize("law,u", [
    ("nonlinear", [0.5, 0.1, 0.1, -0.1]),
    ("squareroot", [0.2, 0.4]),
    ("logarithmic", [0.6, 0.2]),
    ("exponential", [0.5, 0.05]),
    ("power2", [0.6, 0.7]),
    ("linear", [0.55]),
    ("uniform", []),
])
def test_every_other_limb_darkening_law(law, u, exact_elliptic):
    per, rp, a, inc = 9.1, 0.11, 11.0, 87.0
    t = _sample_times(per, a, rp, inc, n=25)
    z = separation_by_vectors(t, 0.0, per, a, inc, 0.0, 90.0)
    want = quadrature_flux(z, rp, law, u)
    got = tm.light_curve(t, 0.0, per, rp, a, inc, 0, 90, u, law)
    # (measured: closed form 2e-15, the module's fixed-node ring integration of the other laws <= 3e-14)
    numpy.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
