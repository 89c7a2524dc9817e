"""
qingdai_amd/energy.py -- host side of pygcm/energy.py's diagnostics + greenhouse autotuning
(energy.py:494-579), over the device-resident state.

`compute_energy_diagnostics(dev)` is one C-ABI call (qd_energy_diagnostics: ten cos-weighted global
means reduced on the device).  `autotune_greenhouse_params` is the reference's proportional controller;
it nudges the DRIVER's EnergyParams copy (QdParams.qnet_lw_eps0 / qnet_lw_kc), exactly like
run_simulation.py:2242-2246 nudges `eparams` and never `gcm.energy_params`.
"""
from __future__ import annotations

import math
import os


def compute_energy_diagnostics(dev):
    """-> {"TOA_net", "SFC_net", "ATM_net", "I_mean", "R_mean", "OLR_mean", "SW_sfc_mean", "LW_sfc_mean", "SH_mean", "LH_mean"}"""
    return dev.energy_diagnostics()


def autotune_greenhouse_params(params, diag, rate_eps=None, rate_kc=None, bounds_eps=(0.30, 0.98), bounds_kc=(0.0, 0.80),
                               verbose=None):
    """energy.py:544-579: eps0 -= rate_eps * TOA_net, kc -= rate_kc * TOA_net, clipped to the bounds.
    `params` is a QdParams; its qnet_lw_* pair starts from lw_eps0 / lw_kc when still unset."""
    err = float(diag.get("TOA_net", 0.0))
    rate_eps = float(os.getenv("QD_TUNE_RATE_EPS", "5e-5")) if rate_eps is None else float(rate_eps)
    rate_kc = float(os.getenv("QD_TUNE_RATE_KC", "2e-5")) if rate_kc is None else float(rate_kc)
    verbose = (int(os.getenv("QD_ENERGY_AUTOTUNE_DIAG", "1")) == 1) if verbose is None else bool(verbose)
    eps0 = params.lw_eps0 if math.isnan(params.qnet_lw_eps0) else params.qnet_lw_eps0
    kc = params.lw_kc if math.isnan(params.qnet_lw_kc) else params.qnet_lw_kc
    params.qnet_lw_eps0 = float(min(max(eps0 - rate_eps * err, bounds_eps[0]), bounds_eps[1]))
    params.qnet_lw_kc = float(min(max(kc - rate_kc * err, bounds_kc[0]), bounds_kc[1]))
    if verbose:
        print(f"[EnergyTune] TOA_net={err:+.3f} W/m^2 -> eps0={params.qnet_lw_eps0:.3f}, kc={params.qnet_lw_kc:.3f}")
    return params
