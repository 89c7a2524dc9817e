"""
TEST INFRASTRUCTURE ONLY.  First stage of the ecology spectral sub-step:
pygcm/ecology/spectral.py:24-56 (bands), 230-285 (main-sequence T_eff, Planck band weights),
60-72 + 288-301 (Rayleigh band factor), 304-426 (dual_star_insolation_to_bands).
"""
from __future__ import annotations

import numpy as np

T_SUN = 5778.0
_H, _C, _KB = 6.62607015e-34, 2.99792458e8, 1.380649e-23
M_SUN, L_SUN = 1.989e30, 3.828e26
M_A, L_A, M_B, L_B = 0.914 * M_SUN, 0.7 * L_SUN, 0.8 * M_SUN, 0.410 * L_SUN     # pygcm/constants.py:13-24


def make_bands(nbands=16, lam0_nm=380.0, lam1_nm=780.0):
    nbands = max(1, int(nbands))
    if lam1_nm <= lam0_nm:
        lam0_nm, lam1_nm = 380.0, 780.0
    edges = np.linspace(float(lam0_nm), float(lam1_nm), nbands + 1)
    return {"nbands": nbands, "edges": edges, "centers": 0.5 * (edges[:-1] + edges[1:]), "widths": edges[1:] - edges[:-1]}


def estimate_teff_from_LM(L_ratio, M_ratio, j=0.8, T_sun=T_SUN):
    return float(T_sun * (float(max(L_ratio, 1e-12)) ** 0.25) * (float(max(M_ratio, 1e-12)) ** (-0.5 * j)))


def blackbody_band_weights(T_eff, bands):
    lam_m = np.maximum(np.asarray(bands["centers"], dtype=float) * 1e-9, 1e-20)
    x = np.clip((_H * _C) / (lam_m * _KB * max(1e-12, float(T_eff))), 1e-8, 1e3)
    B = np.clip((1.0 / (lam_m ** 5)) * (1.0 / (np.expm1(x) + 1e-30)), 0.0, np.inf)
    w = B * np.asarray(bands["widths"], dtype=float)
    return w / (float(np.sum(w)) + 1e-30)


def rayleigh_band_factor(bands, mode="simple", t0=0.9, lref_nm=550.0, eta=4.0):
    if mode != "rayleigh":
        return np.ones(bands["nbands"], dtype=float)
    lam = np.maximum(1e-6, bands["centers"])
    return np.clip(t0 * (lam / max(1e-6, lref_nm)) ** float(eta), 0.0, None)


def dual_star_insolation_to_bands(insA, insB, bands, j_A=0.8, j_B=0.8, T_eff_A=None, T_eff_B=None, rayleigh=None):
    """-> I_b [NB, nlat, nlon]; `rayleigh` = kwargs of rayleigh_band_factor (None = the default 'simple' mode)."""
    if T_eff_A is None:
        T_eff_A = estimate_teff_from_LM(float(L_A / L_SUN), float(M_A / M_SUN), j=j_A)
    if T_eff_B is None:
        T_eff_B = estimate_teff_from_LM(float(L_B / L_SUN), float(M_B / M_SUN), j=j_B)
    specA, specB = blackbody_band_weights(T_eff_A, bands), blackbody_band_weights(T_eff_B, bands)
    T_ray = np.clip(rayleigh_band_factor(bands, **(rayleigh or {})), 0.0, np.inf)
    insA, insB = np.asarray(insA, dtype=float), np.asarray(insB, dtype=float)
    NB = bands["nbands"]
    I_b = np.zeros((NB,) + insA.shape)
    I_tot = insA + insB
    for b in range(NB):
        I_b[b] = (specA[b] * insA + specB[b] * insB) * T_ray[b]
    S_sum = np.sum(I_b, axis=0)
    pos = (S_sum > 1e-12) & (I_tot > 1e-12)
    if np.any(pos):
        for b in range(NB):
            tmp = np.zeros_like(S_sum)
            tmp[pos] = (I_b[b][pos] / S_sum[pos]) * I_tot[pos]
            I_b[b] = tmp
    else:
        I_b[:] = 0.0
    return np.nan_to_num(I_b, nan=0.0, posinf=0.0, neginf=0.0)
