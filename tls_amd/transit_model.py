"""Limb-darkened transit light curve (Mandel & Agol 2002, quadratic law).

The reference obtains its template from the third-party package
`batman-package` (unpinned in the reference's setup.py:41; call sites
transit.py:14-25), which is not available in this image and is not part of the
reference tree.  This module restates the PUBLISHED algorithm that package
evaluates for the linear/quadratic laws:

* circular-orbit sky-projected separation z(t);
* the Mandel & Agol (2002, ApJ 580, L171) closed form for a quadratically
  limb-darkened source (their eqs. 1, 7 and Table 1: lambda_e, lambda_d, eta_d);
* complete elliptic integrals K, E from the Hastings polynomial fits
  (Abramowitz & Stegun 17.3.34 and 17.3.36, |err| ~ 2e-8) and the third kind
  Pi(n, k) from Bulirsch's (1965) iteration with a 1e-8 convergence test.

The polynomial K/E are deliberate: the reference's known-answer tests
(tests/test_synthetic.py:50 chi2_min) were produced with them, and the exact
integrals move that pin by 7e-6 relative.

Everything is vectorised numpy; one 10 000-point template costs ~1 ms.
Provided: every argument the reference hands to batman (transit.py:14-25) -- any
eccentricity in [0, 1) and argument of periastron, the closed-form laws "quadratic",
"linear" (= quadratic with u2 = 0) and "uniform", and the laws without a closed form
("nonlinear", "squareroot", "logarithmic", "exponential", "power2") by numerical
integration.  Pinned independently of this module by tests/test_transit_model_pin.py
(the definition by nested quadrature about the planet's centre, Kepler's equation by
a bracketed root: exact-K/E variant 1e-15, this Hastings variant 9.4e-9, the other
laws 3e-14).
"""
import numpy

_TOL = 1e-14
_BIG = 1.0e10  # separation reported while the planet is behind the star

# Hastings coefficients, A&S 17.3.34 (K) and 17.3.36 (E), argument m1 = 1 - k^2
_KA = (1.38629436112, 0.09666344259, 0.03590092383, 0.03742563713, 0.01451196212)
_KB = (0.5, 0.12498593597, 0.06880248576, 0.03328355346, 0.00441787012)
_EA = (0.44325141463, 0.06260601220, 0.04757383546, 0.01736506451)
_EB = (0.24998368310, 0.09200180037, 0.04069697526, 0.00526449639)


def ellip_k(k):
    """Complete elliptic integral of the first kind, polynomial fit."""
    m1 = 1.0 - k * k
    a = _KA[0] + m1 * (_KA[1] + m1 * (_KA[2] + m1 * (_KA[3] + m1 * _KA[4])))
    b = _KB[0] + m1 * (_KB[1] + m1 * (_KB[2] + m1 * (_KB[3] + m1 * _KB[4])))
    return a - b * numpy.log(m1)


def ellip_e(k):
    """Complete elliptic integral of the second kind, polynomial fit."""
    m1 = 1.0 - k * k
    a = 1.0 + m1 * (_EA[0] + m1 * (_EA[1] + m1 * (_EA[2] + m1 * _EA[3])))
    b = m1 * (_EB[0] + m1 * (_EB[1] + m1 * (_EB[2] + m1 * _EB[3])))
    return a + b * numpy.log(1.0 / m1)


def ellip_pi(n, k):
    """Complete elliptic integral of the third kind Pi(n, k), Bulirsch's
    descending iteration; elementwise over arrays, each element stops on its
    own |1 - kc/g| <= 1e-8 test."""
    n = numpy.atleast_1d(numpy.asarray(n, dtype=float))
    k = numpy.atleast_1d(numpy.asarray(k, dtype=float))
    kc = numpy.sqrt(1.0 - k * k)
    p = numpy.sqrt(n + 1.0)
    m0 = numpy.ones_like(kc)
    c = numpy.ones_like(kc)
    d = 1.0 / p
    e = kc.copy()
    out = numpy.empty_like(kc)
    live = numpy.ones(kc.shape, dtype=bool)
    for _ in range(100):
        if not live.any():
            break
        f = c
        c = d / p + c
        g = e / p
        d = 2.0 * (f * g + d)
        p = g + p
        g = m0
        m0 = kc + m0
        done = live & ~(numpy.abs(1.0 - kc / g) > 1.0e-8)
        out[done] = (0.5 * numpy.pi * (c * m0 + d) / (m0 * (m0 + p)))[done]
        live = live & ~done
        kc = numpy.where(live, 2.0 * numpy.sqrt(e), kc)
        e = numpy.where(live, kc * m0, e)
    return out


def quadratic_ld_flux(z, p, u1, u2):
    """Relative flux of a star (quadratic limb darkening u1, u2) occulted by a
    dark disc of radius p at centre separations z (both in stellar radii)."""
    z = numpy.abs(numpy.array(z, dtype=float))
    flux = numpy.ones_like(z)
    omega = 1.0 - u1 / 3.0 - u2 / 6.0
    c2 = u1 + 2.0 * u2

    # snap separations that sit on a case boundary
    z = numpy.where(numpy.abs(p - z) < _TOL, p, z)
    z = numpy.where(numpy.abs(p - 1.0 - z) < _TOL, p - 1.0, z)
    z = numpy.where(numpy.abs(1.0 - p - z) < _TOL, 1.0 - p, z)
    z = numpy.where(z < _TOL, 0.0, z)

    x1 = (p - z) ** 2
    x2 = (p + z) ** 2
    x3 = p * p - z * z

    todo = z < 1.0 + p  # everything else is unocculted
    lam_e = numpy.zeros_like(z)
    lam_d = numpy.zeros_like(z)
    eta_d = numpy.zeros_like(z)
    kap0 = numpy.zeros_like(z)
    kap1 = numpy.zeros_like(z)

    def finish(mask, add_two_thirds=True):
        ld = lam_d[mask]
        if add_two_thirds:
            ld = ld + numpy.where(p > z[mask], 2.0 / 3.0, 0.0)
        flux[mask] = 1.0 - ((1.0 - c2) * lam_e[mask] + c2 * ld + u2 * eta_d[mask]) / omega
        todo[mask] = False

    # star fully covered
    if p >= 1.0:
        m = todo & (z <= p - 1.0)
        if m.any():
            lam_e[m], lam_d[m], eta_d[m] = 1.0, 0.0, 0.5
            flux[m] = 1.0 - ((1.0 - c2) + c2 * (2.0 / 3.0) + u2 * 0.5) / omega
            todo[m] = False

    # disc crosses the stellar limb: uniform-source term and the two angles
    m = todo & (z >= abs(1.0 - p)) & (z <= 1.0 + p)
    if m.any():
        zz = z[m]
        k1 = numpy.arccos(numpy.minimum((1.0 - p * p + zz * zz) / 2.0 / zz, 1.0))
        k0 = numpy.arccos(numpy.minimum((p * p + zz * zz - 1.0) / 2.0 / p / zz, 1.0))
        kap1[m], kap0[m] = k1, k0
        le = p * p * k0 + k1
        lam_e[m] = (le - 0.5 * numpy.sqrt(numpy.maximum(
            4.0 * zz * zz - (1.0 + zz * zz - p * p) ** 2, 0.0))) / numpy.pi

    # planet edge on the stellar centre (z == p)
    m = todo & (z == p)
    if m.any():
        if p < 0.5:
            q = 2.0 * p
            lam_d[m] = 1.0 / 3.0 + 2.0 / 9.0 / numpy.pi * (
                4.0 * (2.0 * p * p - 1.0) * ellip_e(q) + (1.0 - 4.0 * p * p) * ellip_k(q))
            eta_d[m] = p * p / 2.0 * (p * p + 2.0 * z[m] * z[m])
            lam_e[m] = p * p
        elif p > 0.5:
            q = 0.5 / p
            lam_d[m] = 1.0 / 3.0 + 16.0 * p / 9.0 / numpy.pi * (2.0 * p * p - 1.0) * ellip_e(q) \
                - (32.0 * p ** 4 - 20.0 * p * p + 3.0) / 9.0 / numpy.pi / p * ellip_k(q)
            eta_d[m] = 0.5 / numpy.pi * (kap1[m] + p * p * (p * p + 2.0 * z[m] ** 2) * kap0[m]
                                         - (1.0 + 5.0 * p * p + z[m] ** 2) / 4.0
                                         * numpy.sqrt((1.0 - x1[m]) * (x2[m] - 1.0)))
        else:
            lam_d[m] = 1.0 / 3.0 - 4.0 / numpy.pi / 9.0
            eta_d[m] = 3.0 / 32.0
        finish(m, add_two_thirds=False)

    # ingress/egress: partial overlap, limb crossed
    m = todo & (((z > 0.5 + abs(p - 0.5)) & (z < 1.0 + p))
                | ((p > 0.5) & (z > abs(1.0 - p) * 1.0001) & (z < p)))
    if m.any():
        zz, a1, a2, a3 = z[m], x1[m], x2[m], x3[m]
        q = numpy.sqrt((1.0 - a1) / (a2 - a1))
        Kk, Ek = ellip_k(q), ellip_e(q)
        Pk = ellip_pi(1.0 / a1 - 1.0, q)
        lam_d[m] = 1.0 / 9.0 / numpy.pi / numpy.sqrt(p * zz) * (
            ((1.0 - a2) * (2.0 * a2 + a1 - 3.0) - 3.0 * a3 * (a2 - 2.0)) * Kk
            + 4.0 * p * zz * (zz * zz + 7.0 * p * p - 4.0) * Ek
            - 3.0 * a3 / a1 * Pk)
        eta_d[m] = 1.0 / 2.0 / numpy.pi * (
            kap1[m] + p * p * (p * p + 2.0 * zz * zz) * kap0[m]
            - (1.0 + 5.0 * p * p + zz * zz) / 4.0 * numpy.sqrt((1.0 - a1) * (a2 - 1.0)))
        finish(m)

    # planet disc entirely inside the stellar disc
    m = todo & (z <= 1.0 - p) if p <= 1.0 else numpy.zeros_like(todo)
    if m.any():
        zz, a1, a2, a3 = z[m], x1[m], x2[m], x3[m]
        eta_d[m] = p * p / 2.0 * (p * p + 2.0 * zz * zz)
        lam_e[m] = p * p
        ld = numpy.empty_like(zz)
        centre = zz == 0.0
        if centre.any():  # concentric discs: closed form, no elliptic integrals
            ld[centre] = -2.0 / 3.0 * (1.0 - p * p) ** 1.5
        rest = ~centre
        if rest.any():
            zr, b1, b2, b3 = zz[rest], a1[rest], a2[rest], a3[rest]
            q = numpy.sqrt((b2 - b1) / (1.0 - b1))
            Kk, Ek = ellip_k(q), ellip_e(q)
            Pk = ellip_pi(b2 / b1 - 1.0, q)
            v = 2.0 / 9.0 / numpy.pi / numpy.sqrt(1.0 - b1) * (
                (1.0 - 5.0 * zr * zr + p * p + b3 * b3) * Kk
                + (1.0 - b1) * (zr * zr + 7.0 * p * p - 4.0) * Ek
                - 3.0 * b3 / b1 * Pk)
            touch = numpy.abs(p + zr - 1.0) <= _TOL  # second contact exactly
            v = numpy.where(touch, 2.0 / 3.0 / numpy.pi * numpy.arccos(1.0 - 2.0 * p)
                            - 4.0 / 9.0 / numpy.pi * numpy.sqrt(p * (1.0 - p))
                            * (3.0 + 2.0 * p - 8.0 * p * p), v)
            ld[rest] = v
        lam_d[m] = ld
        # concentric case already carries its own +2/3 through the closed form
        flux_m = numpy.empty_like(zz)
        add = numpy.where(p > zz, 2.0 / 3.0, 0.0)
        flux_m = 1.0 - ((1.0 - c2) * p * p + c2 * (ld + add) + u2 * eta_d[m]) / omega
        flux[m] = flux_m
        todo[m] = False
    return flux


def _true_anomaly(t, t0, per, ecc, omega):
    """True anomaly at times t for an orbit whose inferior conjunction (mid-transit) is at t0.
    Published two-body relations (as batman's _rsky evaluates them): the conjunction has true anomaly
    pi/2 - omega; its eccentric anomaly E = 2 atan(sqrt((1-e)/(1+e)) tan(f/2)) and mean anomaly
    M = E - e sin E give the time of periastron; Kepler's equation is solved by Newton iteration."""
    f_conj = numpy.pi / 2.0 - omega
    if ecc < 1e-5:
        tp = t0 - per * f_conj / (2.0 * numpy.pi)
        x = (t - tp) / per
        return (x - numpy.trunc(x)) * 2.0 * numpy.pi
    E_conj = 2.0 * numpy.arctan(numpy.sqrt((1.0 - ecc) / (1.0 + ecc)) * numpy.tan(f_conj / 2.0))
    M_conj = E_conj - ecc * numpy.sin(E_conj)
    tp = t0 - per * M_conj / (2.0 * numpy.pi)
    x = (t - tp) / per
    M = (x - numpy.floor(x)) * 2.0 * numpy.pi
    E = M + ecc * numpy.sin(M)                       # starting value; Newton converges quadratically for e < 1
    for _ in range(60):
        step = (E - ecc * numpy.sin(E) - M) / (1.0 - ecc * numpy.cos(E))
        E = E - step
        if numpy.max(numpy.abs(step)) < 1e-15:
            break
    return 2.0 * numpy.arctan2(numpy.sqrt(1.0 + ecc) * numpy.sin(E / 2.0), numpy.sqrt(1.0 - ecc) * numpy.cos(E / 2.0))


def projected_separation(t, t0, per, a, inc, ecc, w):
    """Sky-projected star-planet separation in stellar radii; `inc`, `w` in degrees, any
    eccentricity in [0, 1).  While the planet is on the far side of the star a large sentinel
    is returned (no secondary eclipse)."""
    ecc = float(ecc)
    if not 0.0 <= ecc < 1.0:
        raise ValueError("eccentricity must be in [0, 1)")
    t = numpy.asarray(t, dtype=float)
    inc = numpy.radians(inc)
    omega = numpy.radians(w)
    f = _true_anomaly(t, t0, per, ecc, omega)
    s = numpy.sin(f + omega) * numpy.sin(inc)
    r = a * (1.0 - ecc * ecc) / (1.0 + ecc * numpy.cos(f)) if ecc >= 1e-5 else a
    z = r * numpy.sqrt(numpy.maximum(1.0 - s * s, 0.0))
    return numpy.where(s <= 0.0, _BIG, z)


# ---- limb-darkening laws without a closed form: numerical integration over the occulted part of the disc
_LAW_COEFFS = {"nonlinear": 4, "squareroot": 2, "logarithmic": 2, "exponential": 2, "power2": 2}


def _intensity(mu, law, u):
    """Stellar intensity profile I(mu), mu = sqrt(1 - r^2), of the laws batman names."""
    if law == "nonlinear":
        return 1.0 - sum(u[k] * (1.0 - mu ** ((k + 1) / 2.0)) for k in range(4))
    if law == "squareroot":
        return 1.0 - u[0] * (1.0 - mu) - u[1] * (1.0 - numpy.sqrt(mu))
    if law == "logarithmic":
        return 1.0 - u[0] * (1.0 - mu) - u[1] * mu * numpy.log(numpy.maximum(mu, 1e-300))
    if law == "exponential":
        return 1.0 - u[0] * (1.0 - mu) - u[1] / (1.0 - numpy.exp(numpy.maximum(mu, 1e-300)))
    if law == "power2":
        return 1.0 - u[0] * (1.0 - mu ** u[1])
    raise ValueError("unknown limb darkening law %r" % (law,))


_GL_NODES, _GL_WEIGHTS = numpy.polynomial.legendre.leggauss(192)


def _disc_integral(law, u):
    """Integral of I over the whole stellar disc (r dr dphi), substitution r = sin(theta)."""
    th = 0.25 * numpy.pi * (_GL_NODES + 1.0)
    r = numpy.sin(th)
    return 2.0 * numpy.pi * numpy.sum(_GL_WEIGHTS * _intensity(numpy.cos(th), law, u) * r * numpy.cos(th)) * 0.25 * numpy.pi


def numerical_ld_flux(z, p, law, u):
    """Relative flux of a star with intensity profile `law` occulted by a disc of radius p at
    separations z: 1 - (integral of I over the occulted area) / (integral over the disc), the occulted
    area integrated in rings around the stellar centre (a ring of radius r is covered over the angle
    2 acos((r^2 + z^2 - p^2) / (2 r z))) with Gauss-Legendre nodes clustered towards both ends of the
    radial range, where the integrand has square-root behaviour.  Accuracy ~1e-9 of the depth
    (checked against the quadratic closed form through the equivalent non-linear coefficients)."""
    z = numpy.asarray(z, dtype=float)
    flux = numpy.ones_like(z)
    p = abs(float(p))
    touching = z < 1.0 + p
    if not touching.any() or p == 0.0:
        return flux
    zz = z[touching][:, None]
    lo = numpy.maximum(zz - p, 0.0)
    hi = numpy.minimum(zz + p, 1.0)
    # r = lo + (hi - lo) * (1 - cos(theta)) / 2: dr = (hi - lo)/2 sin(theta) dtheta, nodes dense at both ends
    th = 0.5 * numpy.pi * (_GL_NODES[None, :] + 1.0)
    r = lo + (hi - lo) * 0.5 * (1.0 - numpy.cos(th))
    dr = (hi - lo) * 0.5 * numpy.sin(th) * 0.5 * numpy.pi
    with numpy.errstate(divide="ignore", invalid="ignore"):
        cosang = (r * r + zz * zz - p * p) / (2.0 * r * zz)
    cosang = numpy.where(numpy.isfinite(cosang), cosang, -1.0)
    # inside the planet's disc entirely (r < p - z): the whole ring; outside (r > z + p or r < z - p): none
    ang = numpy.arccos(numpy.clip(cosang, -1.0, 1.0))
    ang = numpy.where(r <= p - zz, numpy.pi, ang)
    mu = numpy.sqrt(numpy.maximum(1.0 - r * r, 0.0))
    blocked = numpy.sum(_GL_WEIGHTS[None, :] * _intensity(mu, law, u) * 2.0 * ang * r * dr, axis=1)
    flux[touching] = 1.0 - blocked / _disc_integral(law, u)
    return flux


class TransitParams(object):
    """Attribute bag with the names the reference's call sites set
    (transit.py:14-23): t0, per, rp, a, inc, ecc, w, u, limb_dark."""

    def __init__(self):
        self.t0 = self.per = self.rp = self.a = self.inc = None
        self.ecc = 0.0
        self.w = 90.0
        self.u = None
        self.limb_dark = "quadratic"


class TransitModel(object):
    """TransitModel(params, t).light_curve(params) -> relative flux."""

    def __init__(self, params, t):
        self.t = numpy.asarray(t, dtype=float)

    def light_curve(self, params):
        u = [float(v) for v in params.u] if params.u is not None else []
        law = params.limb_dark
        z = projected_separation(self.t, params.t0, params.per, params.a,
                                 params.inc, params.ecc, params.w)
        if law == "quadratic":
            return quadratic_ld_flux(z, float(params.rp), u[0], u[1])
        if law == "linear":
            return quadratic_ld_flux(z, float(params.rp), u[0], 0.0)
        if law == "uniform":
            return quadratic_ld_flux(z, float(params.rp), 0.0, 0.0)
        if law in _LAW_COEFFS:
            if len(u) != _LAW_COEFFS[law]:
                raise ValueError("limb darkening law %r takes %d coefficients" % (law, _LAW_COEFFS[law]))
            return numerical_ld_flux(z, float(params.rp), law, u)
        raise ValueError("unknown limb darkening law %r" % (law,))


def light_curve(t, t0, per, rp, a, inc, ecc, w, u, limb_dark):
    """Functional form of TransitModel(...).light_curve(...)."""
    ma = TransitParams()
    ma.t0, ma.per, ma.rp, ma.a, ma.inc, ma.ecc, ma.w = t0, per, rp, a, inc, ecc, w
    ma.u, ma.limb_dark = u, limb_dark
    return TransitModel(ma, t).light_curve(ma)
