"""
qingdai_amd/spectral.py -- first stage of the ecology spectral sub-step on the device:
`dual_star_insolation_to_bands` of pygcm/ecology/spectral.py:304-426, with its host-side band tables
(`make_bands` :24-56, `estimate_teff_from_LM` :237-246, `blackbody_band_weights` :249-285, Rayleigh factor :60-72,288-301).

The NB band weights are a handful of host scalars; the per-cell work -- S_b = (specA_b insA + specB_b insB) T_ray_b, its
sum over bands, I_b = S_b / sum * (insA + insB) where both are > 1e-12 -- runs on the resident ISR_A / ISR_B fields
(`qd_band_insolation`) and stays resident ([NB][n_lat][n_lon] f64) for the stages that will follow; `download=True` copies it out.
The rest of the ecology (canopy, populations, individuals) is not on the device yet.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import numpy as np

from .forcing import L_A, L_B, L_SUN, M_A, M_B, M_SUN

T_SUN = 5778.0
_H, _C, _KB = 6.62607015e-34, 2.99792458e8, 1.380649e-23


@dataclass
class SpectralBands:
    nbands: int
    lambda_edges: np.ndarray
    lambda_centers: np.ndarray
    delta_lambda: np.ndarray


def make_bands(nbands=None, lam0_nm=None, lam1_nm=None):
    if nbands is None:
        nbands = int(os.getenv("QD_ECO_SPECTRAL_BANDS", "16"))
    if lam0_nm is None or lam1_nm is None:
        try:
            lam0_nm, lam1_nm = [float(x.strip()) for x in os.getenv("QD_ECO_SPECTRAL_RANGE_NM", "380,780").split(",")]
        except Exception:
            lam0_nm, lam1_nm = 380.0, 780.0
    nbands = max(1, int(nbands))
    if lam1_nm <= lam0_nm:
        lam0_nm, lam1_nm = 380.0, 780.0
    edges = np.linspace(float(lam0_nm), float(lam1_nm), nbands + 1)
    return SpectralBands(nbands, edges, 0.5 * (edges[:-1] + edges[1:]), edges[1:] - edges[:-1])


def estimate_teff_from_LM(L_ratio, M_ratio, j=0.8, T_sun=T_SUN):
    return float(T_sun * (float(max(L_ratio, 1e-12)) ** 0.25) * (float(max(M_ratio, 1e-12)) ** (-0.5 * j)))


def blackbody_band_weights(T_eff, bands):
    lam_m = np.maximum(np.asarray(bands.lambda_centers, dtype=float) * 1e-9, 1e-20)
    x = np.clip((_H * _C) / (lam_m * _KB * max(1e-12, float(T_eff))), 1e-8, 1e3)
    B = np.clip((1.0 / (lam_m ** 5)) * (1.0 / (np.expm1(x) + 1e-30)), 0.0, np.inf)
    w = B * np.asarray(bands.delta_lambda, dtype=float)
    return w / (float(np.sum(w)) + 1e-30)


def rayleigh_band_factor(bands):
    if os.getenv("QD_ECO_TOA_TO_SURF_MODE", "simple").strip().lower() != "rayleigh":
        return np.ones(bands.nbands, dtype=float)
    t0 = float(os.getenv("QD_ECO_RAYLEIGH_T0", "0.9"))
    lref = float(os.getenv("QD_ECO_RAYLEIGH_LREF_NM", "550"))
    eta = float(os.getenv("QD_ECO_RAYLEIGH_ETA", "4.0"))
    lam = np.maximum(1e-6, bands.lambda_centers)
    return np.clip(t0 * (lam / max(1e-6, lref)) ** eta, 0.0, None)


def star_band_weights(bands, T_eff_A=None, T_eff_B=None, j_A=None, j_B=None):
    """-> (specA, specB, T_ray), each [NB]: everything of spectral.py:346-395 that does not depend on the grid."""
    j_A = float(os.getenv("QD_STAR_A_J", "0.8")) if j_A is None else float(j_A)
    j_B = float(os.getenv("QD_STAR_B_J", "0.8")) if j_B is None else float(j_B)
    if T_eff_A is None and os.getenv("QD_STAR_A_TEFF_K"):
        T_eff_A = float(os.environ["QD_STAR_A_TEFF_K"])
    if T_eff_B is None and os.getenv("QD_STAR_B_TEFF_K"):
        T_eff_B = float(os.environ["QD_STAR_B_TEFF_K"])
    if T_eff_A is None:
        T_eff_A = estimate_teff_from_LM(float(L_A / L_SUN), float(M_A / M_SUN), j=j_A)
    if T_eff_B is None:
        T_eff_B = estimate_teff_from_LM(float(L_B / L_SUN), float(M_B / M_SUN), j=j_B)
    return blackbody_band_weights(T_eff_A, bands), blackbody_band_weights(T_eff_B, bands), np.clip(rayleigh_band_factor(bands), 0.0, np.inf)


def dual_star_insolation_to_bands(dev, bands, download=True, **star_kw):
    """Band intensities [NB, n_lat, n_lon] from the device's resident ISR_A / ISR_B (set them with ThermalForcing.update_device
    or by assigning gcm.isr_A / gcm.isr_B).  Returns the host array, or None with download=False (result stays resident)."""
    specA, specB, tray = (np.ascontiguousarray(a, dtype=np.float64) for a in star_band_weights(bands, **star_kw))
    dev.flush()
    out = np.empty((bands.nbands,) + dev.shape, dtype=np.float64) if download else None
    dp = ctypes.POINTER(ctypes.c_double)
    dev._chk(dev.lib.qd_band_insolation(dev.h, int(bands.nbands), specA.ctypes.data_as(dp), specB.ctypes.data_as(dp),
                                        tray.ctypes.data_as(dp), out.ctypes.data if download else None), "qd_band_insolation")
    return out


def band_weights_from_mode(bands, mode=None):
    """Normalised reduction weights w_b (spectral.py:138-162): flat, or the Rayleigh factor, over its sum + 1e-12."""
    mode = (mode or os.getenv("QD_ECO_TOA_TO_SURF_MODE", "simple")).strip().lower()
    if mode == "rayleigh":
        t0 = float(os.getenv("QD_ECO_RAYLEIGH_T0", "0.9"))
        lref = float(os.getenv("QD_ECO_RAYLEIGH_LREF_NM", "550"))
        eta = float(os.getenv("QD_ECO_RAYLEIGH_ETA", "4.0"))
        w = np.clip(t0 * (np.maximum(1e-6, bands.lambda_centers) / max(1e-6, lref)) ** eta, 0.0, None)
    else:
        w = np.ones_like(bands.lambda_centers, dtype=float)
    return w / (float(np.sum(w)) + 1e-12)


def default_leaf_reflectance(bands):
    """Green-ish leaf template (spectral.py:70-82,165-169): 0.25 + 0.15 exp(-(lam - 550)^2 / (2 60^2)), clipped to [0, 1]."""
    lam = np.asarray(bands.lambda_centers, dtype=float)
    return np.clip(0.25 + 0.15 * np.exp(-((lam - 550.0) ** 2) / (2.0 * 60.0 ** 2)), 0.0, 1.0)


def absorbance_from_peaks(bands, peaks):
    """Band absorbance of a list of (center_nm, width_nm, height) Gaussian peaks (genes.py:95-111): peaks with a non-positive
    width or height are skipped, the sum is clipped to [0, 1]."""
    lam = np.asarray(bands.lambda_centers, dtype=float)
    A = np.zeros_like(lam)
    for (c, w, h) in peaks:
        if w <= 0 or h <= 0:
            continue
        A += h * np.exp(-((lam - c) ** 2) / (2 * (w ** 2)))
    return np.clip(A, 0.0, 1.0)
