"""World coordinates of a FITS image header: the object the reference builds with `astropy.wcs.WCS(header)` at
caesar_yolo/inference.py:473 (`SFinder.wcs`) and caesar_yolo/utils.py:236 / :411 (third return value of `read_fits` /
`read_fits_crop`).  astropy is not a dependency of this build, so the celestial part of the FITS WCS standard is restated
here (Greisen & Calabretta 2002, Calabretta & Greisen 2002: linear transform, TAN / SIN / ARC / STG zenithal, CAR / SFL / MER
cylindrical projections, spherical rotation with LONPOLE / LATPOLE defaults, CROTA2 / CDi_j / PCi_j conventions); everything
else (distortions, -TAB, spectral axes) is out of scope and raises.  Pinned against astropy 4.3.1 by tests/golden/wcs.json
(oracle/gen_golden.py: gen_wcs; tests/test_wcs_cpu.py).  No output of the detect path uses it -- as in the reference."""
import math
import numpy as np

D2R, R2D = math.pi / 180.0, 180.0 / math.pi
_ZENITHAL = ("TAN", "SIN", "ARC", "STG")
_CYLINDRICAL = ("CAR", "SFL", "MER")


class WCSError(ValueError):
    pass


def _get(h, k, default=None):
    try:
        return h[k]
    except (KeyError, IndexError):
        return default


class WCS(object):
    """Two celestial (or linear) axes of a header-like mapping (dict, or the card dict of utils.read_fits_image).

    wcs_pix2world(x, y, origin) / all_pix2world: pixel -> world (degrees for celestial axes); wcs_world2pix / all_world2pix: the
    inverse.  `origin` is 0 for numpy-style and 1 for FITS-style pixel coordinates, as in astropy."""

    def __init__(self, header=None):
        h = header if header is not None else {}
        self.naxis = 2
        self.ctype = [str(_get(h, "CTYPE%d" % i, "") or "").strip() for i in (1, 2)]
        self.crpix = [float(_get(h, "CRPIX%d" % i, 0.0)) for i in (1, 2)]
        self.crval = [float(_get(h, "CRVAL%d" % i, 0.0)) for i in (1, 2)]
        self.cunit = [str(_get(h, "CUNIT%d" % i, "") or "").strip() for i in (1, 2)]
        cdelt = [float(_get(h, "CDELT%d" % i, 1.0)) for i in (1, 2)]
        has_cd = any(_get(h, "CD%d_%d" % (i, j)) is not None for i in (1, 2) for j in (1, 2))
        has_pc = any(_get(h, "PC%d_%d" % (i, j)) is not None for i in (1, 2) for j in (1, 2))
        if has_cd and not has_pc:
            m = [[float(_get(h, "CD%d_%d" % (i, j), 0.0)) for j in (1, 2)] for i in (1, 2)]
            self.cdelt = [1.0, 1.0]
            self.pc = m
        else:
            pc = [[float(_get(h, "PC%d_%d" % (i, j), 1.0 if i == j else 0.0)) for j in (1, 2)] for i in (1, 2)]
            rot = _get(h, "CROTA2")
            if rot is not None and not has_pc and float(rot) != 0.0:      # AIPS convention -> PC (Paper II, section 6.1)
                r = float(rot) * D2R
                lam = cdelt[1] / cdelt[0]
                pc = [[math.cos(r), -lam * math.sin(r)], [math.sin(r) / lam, math.cos(r)]]
            self.cdelt, self.pc = cdelt, pc
        self.lonpole = _get(h, "LONPOLE")
        self.latpole = _get(h, "LATPOLE")
        self.radesys = str(_get(h, "RADESYS", "") or "").strip()
        self.equinox = _get(h, "EQUINOX")
        # which axis is the longitude, which projection
        self.proj, self.lng, self.lat = None, None, None
        codes = []
        for i, ct in enumerate(self.ctype):
            if len(ct) >= 8 and ct[4] == "-" or (len(ct) > 5 and "-" in ct[:5] and len(ct.split("-")[-1]) == 3):
                name, code = ct[:4].rstrip("-"), ct.split("-")[-1]
                codes.append(code)
                if name in ("RA", "GLON", "ELON", "SLON", "HLON") or name.endswith("LN"):
                    self.lng = i
                elif name in ("DEC", "GLAT", "ELAT", "SLAT", "HLAT") or name.endswith("LT"):
                    self.lat = i
        if self.lng is not None and self.lat is not None and len(set(codes)) == 1:
            self.proj = codes[0]
            if self.proj not in _ZENITHAL + _CYLINDRICAL:
                raise WCSError("projection %r is not implemented (TAN SIN ARC STG CAR SFL MER are)" % self.proj)
        elif self.lng is not None or self.lat is not None or codes:
            raise WCSError("inconsistent celestial axes: CTYPE = %r" % (self.ctype,))
        if self.proj:
            self._setup_rotation()

    # ---- linear part
    def pixel_scale_matrix(self):
        return np.array([[self.cdelt[i] * self.pc[i][j] for j in (0, 1)] for i in (0, 1)], np.float64)

    @property
    def is_celestial(self):
        return self.proj is not None

    # ---- spherical rotation (Paper II section 2.4)
    def _setup_rotation(self):
        zen = self.proj in _ZENITHAL
        self.theta0 = 90.0 if zen else 0.0
        self.phi0 = 0.0
        a0, d0 = self.crval[self.lng], self.crval[self.lat]
        lonpole = self.lonpole
        if lonpole is None:
            lonpole = 0.0 if d0 >= self.theta0 else 180.0
        self.phi_p = float(lonpole)
        if zen:
            self.alpha_p, self.delta_p = a0, d0
            return
        # non-polar reference point: celestial coordinates of the native pole (Paper II eqs. 8-10)
        latpole = 90.0 if self.latpole is None else float(self.latpole)
        phi_p, th0 = self.phi_p * D2R, self.theta0 * D2R
        d0r = d0 * D2R
        cth0, sth0 = math.cos(th0), math.sin(th0)
        cdp = math.cos(phi_p - self.phi0 * D2R)
        sdp = math.sin(phi_p - self.phi0 * D2R)
        x = cth0 * cdp
        y = sth0
        r = math.hypot(x, y)
        if r == 0.0:
            if d0r != 0.0:
                raise WCSError("invalid CRVAL / LONPOLE combination")
            delta_p = latpole * D2R
        else:
            u = math.atan2(y, x)
            s = math.sin(d0r) / r
            if abs(s) > 1.0:
                if abs(s) > 1.0 + 1e-13:
                    raise WCSError("invalid CRVAL / LONPOLE combination")
                s = math.copysign(1.0, s)
            v = math.acos(s)
            c1, c2 = u + v, u - v
            cands = [c for c in (c1, c2) if -math.pi / 2 - 1e-13 <= _wrap_pi(c) <= math.pi / 2 + 1e-13]
            cands = [_wrap_pi(c) for c in cands]
            if not cands:
                raise WCSError("no valid celestial pole for this header")
            if len(cands) == 2 and abs(cands[0] - cands[1]) > 1e-15:      # the solution closer to LATPOLE
                lp = latpole * D2R
                delta_p = cands[0] if abs(lp - cands[0]) <= abs(lp - cands[1]) else cands[1]
            else:
                delta_p = cands[0]
        delta_p = max(-math.pi / 2, min(math.pi / 2, delta_p))
        # longitude of the pole
        if abs(abs(delta_p) - math.pi / 2) < 1e-15:
            alpha_p = a0 * D2R + (phi_p - self.phi0 * D2R - math.pi if delta_p > 0 else -(phi_p - self.phi0 * D2R))
        else:
            das = sdp * cth0 / math.cos(d0r) if math.cos(d0r) != 0.0 else 0.0
            dac = (sth0 - math.sin(delta_p) * math.sin(d0r)) / (math.cos(delta_p) * math.cos(d0r)) if math.cos(d0r) != 0.0 else 1.0
            alpha_p = a0 * D2R - math.atan2(das, dac)
        self.alpha_p, self.delta_p = alpha_p * R2D, delta_p * R2D

    # ---- projections: intermediate world (deg) <-> native spherical (deg)
    def _deproject(self, x, y):
        p = self.proj
        if p in _ZENITHAL:
            r = np.hypot(x, y)
            phi = np.where(r == 0.0, 0.0, np.arctan2(x, -y) * R2D)
            rr = r * D2R
            if p == "TAN":
                theta = np.arctan2(1.0, rr) * R2D
            elif p == "SIN":
                theta = np.arccos(np.clip(rr, -1.0, 1.0)) * R2D
            elif p == "ARC":
                theta = 90.0 - r
            else:  # STG
                theta = 90.0 - 2.0 * np.arctan(rr / 2.0) * R2D
            return phi, theta
        if p == "CAR":
            return x, y
        if p == "SFL":
            return x / np.cos(y * D2R), y
        # MER
        return x, 2.0 * np.arctan(np.exp(y * D2R)) * R2D - 90.0

    def _project(self, phi, theta):
        p = self.proj
        if p in _ZENITHAL:
            th = theta * D2R
            if p == "TAN":
                r = R2D / np.tan(th)
            elif p == "SIN":
                r = R2D * np.cos(th)
            elif p == "ARC":
                r = 90.0 - theta
            else:
                r = 2.0 * R2D * np.tan((math.pi / 2 - th) / 2.0)
            return r * np.sin(phi * D2R), -r * np.cos(phi * D2R)
        if p == "CAR":
            return phi, theta
        if p == "SFL":
            return phi * np.cos(theta * D2R), theta
        return phi, R2D * np.log(np.tan((90.0 + theta) / 2.0 * D2R))

    def _native_to_celestial(self, phi, theta):
        ap, dp, pp = self.alpha_p * D2R, self.delta_p * D2R, self.phi_p * D2R
        ph, th = phi * D2R, theta * D2R
        dphi = ph - pp
        sd, cd = math.sin(dp), math.cos(dp)
        st, ct = np.sin(th), np.cos(th)
        delta = np.arcsin(np.clip(st * sd + ct * cd * np.cos(dphi), -1.0, 1.0))
        alpha = ap + np.arctan2(-ct * np.sin(dphi), st * cd - ct * sd * np.cos(dphi))
        alpha = alpha * R2D
        ref = self.crval[self.lng]
        alpha = np.where(alpha < 0.0, alpha + 360.0, alpha)
        alpha = np.where(alpha >= 360.0, alpha - 360.0, alpha)
        if ref < 0.0:                                   # (wcslib keeps the longitude in the range of CRVAL's sign)
            alpha = np.where(alpha > 0.0, alpha - 360.0, alpha)
        return alpha, delta * R2D

    def _celestial_to_native(self, alpha, delta):
        ap, dp, pp = self.alpha_p * D2R, self.delta_p * D2R, self.phi_p * D2R
        al, de = alpha * D2R, delta * D2R
        da = al - ap
        sd, cd = math.sin(dp), math.cos(dp)
        theta = np.arcsin(np.clip(np.sin(de) * sd + np.cos(de) * cd * np.cos(da), -1.0, 1.0))
        phi = pp + np.arctan2(-np.cos(de) * np.sin(da), np.sin(de) * cd - np.cos(de) * sd * np.cos(da))
        phi = phi * R2D
        phi = np.where(phi > 180.0, phi - 360.0, phi)
        phi = np.where(phi < -180.0, phi + 360.0, phi)
        return phi, theta * R2D

    # ---- public transforms
    def wcs_pix2world(self, x, y, origin):
        x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
        p = [x + (1 - origin) - self.crpix[0], y + (1 - origin) - self.crpix[1]]
        m = self.pixel_scale_matrix()
        w = [m[0, 0] * p[0] + m[0, 1] * p[1], m[1, 0] * p[0] + m[1, 1] * p[1]]
        if not self.proj:
            return w[0] + self.crval[0], w[1] + self.crval[1]
        phi, theta = self._deproject(w[self.lng], w[self.lat])
        a, d = self._native_to_celestial(phi, theta)
        out = [None, None]
        out[self.lng], out[self.lat] = a, d
        return out[0], out[1]

    all_pix2world = wcs_pix2world

    def wcs_world2pix(self, a, d, origin):
        a, d = np.asarray(a, np.float64), np.asarray(d, np.float64)
        if self.proj:
            w_in = [a, d]
            phi, theta = self._celestial_to_native(w_in[self.lng], w_in[self.lat])
            xl, yl = self._project(phi, theta)
            w = [None, None]
            w[self.lng], w[self.lat] = xl, yl
        else:
            w = [a - self.crval[0], d - self.crval[1]]
        mi = np.linalg.inv(self.pixel_scale_matrix())
        p0 = mi[0, 0] * w[0] + mi[0, 1] * w[1] + self.crpix[0] - (1 - origin)
        p1 = mi[1, 0] * w[0] + mi[1, 1] * w[1] + self.crpix[1] - (1 - origin)
        return p0, p1

    all_world2pix = wcs_world2pix

    def proj_plane_pixel_scales(self):
        m = self.pixel_scale_matrix()
        return np.sqrt((m ** 2).sum(axis=0))

    def __repr__(self):
        return "WCS(ctype=%r, crval=%r, crpix=%r, cdelt=%r)" % (self.ctype, self.crval, self.crpix, self.cdelt)


def _wrap_pi(a):
    while a > math.pi:
        a -= 2 * math.pi
    while a < -math.pi:
        a += 2 * math.pi
    return a
