"""Host-side helpers between the samplers and the kernels.

Mirrors the hot-path part of the reference's triceratops/funcs.py:
  stellar_relations (funcs.py:54-79), flux_relation (121-140), renorm_flux (164-177),
  file_to_contrast_curve (203-219), separation_at_contrast (222-238),
  trilegal_results (335-403).
color_Teff_relations (143-161) and Gauss2D (180-200) have no caller on the path and are kept
for API parity only.  The web / FITS I/O (funcs.py:241-333, 405-474) is out of scope (SURVEY.md
section 2 rows 9-10).
"""
import os

import numpy as np
from pandas import read_csv
from scipy.interpolate import InterpolatedUnivariateSpline

# mass -> (radius, Teff) nodes: Torres et al. above 0.63 M_sun, cool dwarfs below
# (values of funcs.py:19-51)
_M_HOT = np.array([0.26, 0.47, 0.59, 0.69, 0.87, 0.98, 1.085, 1.4, 1.65, 2.0, 2.5, 3.0, 4.4,
                   15.0, 40.0])
_T_HOT = np.array([3170, 3520, 3840, 4410, 5150, 5560, 5940, 6650, 7300, 8180, 9790, 11400,
                   15200, 30000, 42000])
_R_HOT = np.array([0.28, 0.47, 0.60, 0.72, 0.9, 1.05, 1.2, 1.55, 1.8, 2.1, 2.4, 2.6, 3.0, 6.2,
                   11.0])
_M_COOL = np.array([0.1, 0.135, 0.2, 0.35, 0.48, 0.58, 0.63])
_T_COOL = np.array([2800, 3000, 3200, 3400, 3600, 3800, 4000])
_R_COOL = np.array([0.12, 0.165, 0.23, 0.36, 0.48, 0.585, 0.6])

_spl = {
    "T_hot": InterpolatedUnivariateSpline(_M_HOT, _T_HOT),
    "R_hot": InterpolatedUnivariateSpline(_M_HOT, _R_HOT),
    "T_cool": InterpolatedUnivariateSpline(_M_COOL, _T_COOL),
    "R_cool": InterpolatedUnivariateSpline(_M_COOL, _R_COOL),
}

# mass -> log10 flux relative to a ~1 M_sun star, per band (values of funcs.py:81-119)
_FLUX_NODES = {
    "TESS": (np.array([0.1, 0.15, 0.23, 0.4, 0.58, 0.7, 0.9, 1.15, 1.45, 2.2, 2.8]),
             np.array([-3, -2.5, -2, -1.5, -1, -0.5, 0, 0.5, 1, 1.5, 2])),
    "J": (np.array([0.1, 0.2, 0.5, 0.75, 1.0, 1.5, 2.0, 2.5, 3]),
          np.array([-5.7, -3.8, -1.6, 0, 1.2, 2.9, 3.3, 4, 6]) / 2.5),
    "H": (np.array([0.1, 0.23, 0.5, 0.75, 1.0, 1.5, 2.0, 2.5, 3]),
          np.array([-4.9, -2.8, -0.9, 0.6, 1.5, 3, 3.3, 4, 6]) / 2.5),
    "K": (np.array([0.1, 0.2, 0.35, 0.5, 0.75, 1.0, 1.5, 2.0, 2.5, 3]),
          np.array([-4.7, -2.9, -1.7, -0.7, 0.6, 1.6, 3, 3.3, 4, 6]) / 2.5),
}
_flux_spl = {band: InterpolatedUnivariateSpline(m, f) for band, (m, f) in _FLUX_NODES.items()}
_flux_spl["Vis"] = _flux_spl["TESS"]


def stellar_relations(Masses, max_Radii, max_Teffs):
    """Radii [R_sun] and Teffs [K] of stars of the given masses, capped at the host's values
    and floored at 0.1 R_sun / 2800 K (funcs.py:54-79)."""
    Masses = np.asarray(Masses, dtype=np.float64)
    hot = Masses > 0.63
    cool = ~hot & (Masses <= 0.63)
    Radii = np.zeros(len(Masses))
    Teffs = np.zeros(len(Masses))
    Radii[hot] = _spl["R_hot"](Masses[hot])
    Teffs[hot] = _spl["T_hot"](Masses[hot])
    Radii[cool] = _spl["R_cool"](Masses[cool])
    Teffs[cool] = _spl["T_cool"](Masses[cool])
    over = Radii > max_Radii
    Radii[over] = max_Radii[over]
    over = Teffs > max_Teffs
    Teffs[over] = max_Teffs[over]
    Radii[Radii < 0.1] = 0.1
    Teffs[Teffs < 2800] = 2800
    return Radii, Teffs


def flux_relation(Masses, filt: str = "TESS"):
    """Flux of stars of the given masses relative to a ~1 M_sun star in band `filt`
    (TESS, Vis, J, H or K; funcs.py:121-140)."""
    if filt not in _flux_spl:
        raise UnboundLocalError("unknown filter %r (the reference leaves `fluxes` unbound)" % (filt,))
    return 10 ** _flux_spl[filt](Masses)


def renorm_flux(flux, flux_err, star_fluxratio: float):
    """Light curve renormalised to the flux share of one star (funcs.py:164-177)."""
    return (flux - (1 - star_fluxratio)) / star_fluxratio, flux_err / star_fluxratio


def file_to_contrast_curve(contrast_curve_file: str):
    """(separations [arcsec], |delta_mag|) from a two-column csv (funcs.py:203-219)."""
    data = np.loadtxt(contrast_curve_file, delimiter=',')
    return data.T[0], np.abs(data.T[1])


def separation_at_contrast(delta_mags, separations, contrasts):
    """Separation beyond which a companion of contrast delta_mags is ruled out
    (np.interp over the contrast curve; funcs.py:222-238)."""
    return np.interp(delta_mags, contrasts, separations)


def trilegal_results(trilegal_fname: str, Tmag: float):
    """Background-star population fainter than the target from a saved TRILEGAL table
    (funcs.py:335-403): (Tmags, Masses, loggs, Teffs, Zs, Jmags, Hmags, Kmags).
    The last two rows of the file are TRILEGAL's trailer and are dropped (:353); without a
    TESS column the T magnitudes come from the 2MASS relations of Stassun et al. 2018."""
    Tmags, cols = _trilegal_table(trilegal_fname)
    keep = Tmags >= Tmag
    return (Tmags[keep], cols["Masses"][keep], cols["loggs"][keep], cols["Teffs"][keep],
            cols["Zs"][keep], cols["Jmags"][keep], cols["Hmags"][keep], cols["Kmags"][keep])


_trilegal_cache = {}


def _trilegal_table(trilegal_fname):
    """(Tmags, columns) of the whole file.  The ~10 field-star calls of a calc_probs read the same file:
    the parsed table is kept per (path, mtime, size) -- a rewritten file is read again; the boolean
    selections above return fresh arrays, so callers never see each other's."""
    st = os.stat(trilegal_fname)
    key = (os.path.abspath(trilegal_fname), st.st_mtime_ns, st.st_size)
    hit = _trilegal_cache.get(key)
    if hit is not None:
        return hit
    df = read_csv(trilegal_fname)[:-2]
    cols = {
        "Masses": df["Mact"].values,
        "loggs": df["logg"].values,
        "Teffs": 10 ** df["logTe"].values,
        "Zs": np.array(df["[M/H]"], dtype=float),
        "Jmags": df["J"].values,
        "Hmags": df["H"].values,
        "Kmags": df["Ks"].values,
    }
    if "TESS" in list(df):
        Tmags = df["TESS"].values
    else:
        J, Ks = cols["Jmags"], cols["Kmags"]
        c = J - Ks
        Tmags = np.zeros(df.shape[0])
        blue = (-0.1 <= c) & (c <= 0.70)
        red = (0.7 < c) & (c <= 1.0)
        Tmags[blue] = (J[blue] + 1.22163 * c[blue] ** 3 - 1.74299 * c[blue] ** 2
                       + 1.89115 * c[blue] + 0.0563)
        Tmags[red] = (J[red] - 269.372 * c[red] ** 3 + 668.453 * c[red] ** 2
                      - 545.64 * c[red] + 147.811)
        Tmags[c < -0.1] = J[c < -0.1] + 0.5
        Tmags[c > 1.0] = J[c > 1.0] + 1.75
    if len(_trilegal_cache) >= 4:
        _trilegal_cache.clear()
    _trilegal_cache[key] = (Tmags, cols)
    return Tmags, cols
