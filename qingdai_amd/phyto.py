"""
qingdai_amd/phyto.py -- transport of the phytoplankton tracers by the ocean currents
(pygcm/ecology/phyto.py:496-547) on the MI355X library.

`PhytoManager.advect_diffuse` reuses the ocean's two operators per species -- the semi-Lagrangian gather
and the spherical Laplacian, both on the ocean cos floor max(cos, 0.5) -- so it maps onto `qd_op_advect` and
`qd_op_laplacian` (cos kind 1) of the C-ABI; the blend, clip, land mask and the two polar-ring means are a few
NumPy lines on the [S, n_lat, n_lon] host array the ecology code owns.  The ecology itself (daily cadence,
genes, optics) stays outside this path.
"""
from __future__ import annotations

import os

import numpy as np


def advect_diffuse(dev, C_s, uo, vo, dt_seconds, land_mask, K_h=None, adv_alpha=None):
    """dev: qingdai_amd.device.Device (or `grid._ops()`); C_s: [S, n_lat, n_lon]; returns the new array."""
    C_s = np.array(C_s, dtype=np.float64, copy=True)
    if dt_seconds <= 0.0:
        return C_s
    K_h = float(os.getenv("QD_PHYTO_KH", os.getenv("QD_KH_OCEAN", "5.0e3"))) if K_h is None else float(K_h)
    adv_alpha = float(os.getenv("QD_PHYTO_ADV_ALPHA", "0.7")) if adv_alpha is None else float(adv_alpha)
    ocean = (np.asarray(land_mask) == 0)
    for s in range(C_s.shape[0]):
        C = C_s[s]
        C_adv = dev.op_advect(C, uo, vo, float(dt_seconds), ocean=True)
        C_new = (1.0 - adv_alpha) * C + adv_alpha * C_adv
        if K_h > 0.0:
            C_new = np.nan_to_num(C_new)
            C_new += float(dt_seconds) * K_h * dev.op_laplacian(C_new, ocean=True)
        C_new = np.clip(C_new, 0.0, np.inf)
        C_new[~ocean] = 0.0
        C_s[s] = C_new
    for j in (0, -1):
        row = ocean[j, :]
        if np.any(row):
            for s in range(C_s.shape[0]):
                C_s[s, j, row] = float(np.mean(C_s[s, j, :][row]))
    return C_s
