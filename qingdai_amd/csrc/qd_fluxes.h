// qd_fluxes.h -- column-physics scalars and the surface-flux device function shared by
// time_step's energy branch (qd_atmos.hip) and the driver's ocean coupling (qd_ocean.hip).
#pragma once
#include "qd_internal.h"

#define QD_EPSILON 0.622

struct QdColP {           // scalars of the column kernel, all derived on the host exactly as the
    double ga;            // g/1004                       reference derives them per step
    double M_col, tau_c, L_v, p0, rhoCE, s_ocean, s_land, s_ice;
    double sigma, gfs, c_sfc_safe, dt;
    double rh0, k_q, k_p;
    double sw_a0, sw_kc, lw_eps0, lw_kc, eps_clear, tau0, k_tau, hice_ref_safe;
    double eps_ocean, eps_land, eps_ice, eps_default;
    double g_lw, rhocpch;
    double t_freeze, rhoiLf, rho_i, L_f, Cs_ocean, Cs_land, Cs_ice, t_floor;
    double w_energy, h_eq_fac, tau_rad, atm_denom, atm_w;
    int couple, lw_v2, gh_lock, seaice, fix_s, fix_n, has_csmap, atm_couple;
    int write_diag = 1;   // 0: a step inside a qd_step_n span -- E, LH_release and OLR, which nothing inside a span reads, are not stored
};

__device__ __forceinline__ double qd_qsat(double T, double p0) {        // humidity.py:85-101
    const double T_c = qd_clip(T - 273.15, -80.0, 60.0);
    const double e_s = 610.94 * exp(17.625 * T_c / (T_c + 243.04));
    const double denom = qd_max(p0 - (1.0 - QD_EPSILON) * e_s, 1.0);
    return qd_clip(QD_EPSILON * e_s / denom, 0.0, 0.5);
}

// humidity column (dynamics.py:282-297, humidity.py:104-183): evaporation, condensation of the excess over saturation, the new q.
// One body for k_column (qd_atmos.hip, all phases) and for the P_cond that k_snow_albedo_forcing (qd_physics.hip) writes ahead of
// time_step's median when it stands in for phase 1.
struct QdHum { double q, E, Pc, LH, LHrel; };
__device__ __forceinline__ QdHum qd_humidity_column(const QdColP& P, double u, double v, double Ts, double q0, double qsat_air,
                                                    bool land, double hice) {
    QdHum r;
    const double fac = land ? P.s_land : ((hice > 1e-6) ? P.s_ice : P.s_ocean);
    const double V = sqrt(u * u + v * v);
    const double deficit = qd_max(0.0, qd_qsat(Ts, P.p0) - q0);
    r.E = qd_nn(P.rhoCE * V * deficit * fac);
    r.LH = P.L_v * r.E;
    const double q_evap = q0 + (r.E / P.M_col) * P.dt;
    const double excess = qd_max(0.0, q_evap - qsat_air);
    double Pc = (excess / P.tau_c) * P.M_col;
    double q_next = q_evap - (Pc / P.M_col) * P.dt;
    q_next = qd_clip(qd_nn(q_next), 0.0, 0.5);
    r.Pc = qd_nn(Pc);
    r.LHrel = P.L_v * r.Pc;
    r.q = qd_clip(qd_nn(q_next), 0.0, 0.5);
    return r;
}

// Surface fluxes shared by the energy branch of time_step and by the ocean coupling of the
// driver (run_simulation.py:2197-2249): SW (energy.py:77-98), LW v1/v2 (energy.py:101-234),
// SH (energy.py:423-442).
struct QdFlux { double SW_atm, SW_sfc, R, LW_atm, LW_sfc, OLR, SH; };

__device__ __forceinline__ QdFlux qd_surface_fluxes(const QdColP& P, double I, double albedo, double cloud_eff,
                                                    double Ts, double Ta, double u, double v, bool land, double hice) {
    QdFlux F;
    const double alpha = qd_clip(albedo, 0.0, 1.0);
    const double Ic = qd_max(0.0, I);
    F.R = Ic * alpha;
    const double A_sw = qd_clip(P.sw_a0 + P.sw_kc * qd_clip(cloud_eff, 0.0, 1.0), 0.0, 0.95);
    F.SW_atm = Ic * A_sw;
    F.SW_sfc = qd_max(0.0, Ic - F.R - F.SW_atm);
    const double s = P.sigma;
    const double Tsp = qd_max(0.0, Ts), Tap = qd_max(0.0, Ta);
    const double Ts4 = qd_pow4(Tsp), Ta4 = qd_pow4(Tap);
    if (P.lw_v2) {
        const double ce = qd_clip(cloud_eff, 0.0, 1.0);
        const double tau_cloud = P.tau0 * ce;
        const double eps_cloud = qd_clip(1.0 - exp(-P.k_tau * tau_cloud), 0.0, 1.0);
        const double eps_eff = 1.0 - (1.0 - P.eps_clear) * (1.0 - eps_cloud);
        double es;
        if (land) es = P.eps_land;
        else {
            const double ice_frac = 1.0 - exp(-qd_max(hice, 0.0) / P.hice_ref_safe);
            const double fi = qd_clip(ice_frac, 0.0, 1.0);
            es = (1.0 - fi) * P.eps_ocean + fi * P.eps_ice;
        }
        es = qd_clip(qd_nn(es), 0.0, 1.0);
        F.OLR = eps_eff * s * Ta4 + (1.0 - eps_eff) * s * es * Ts4;
        const double DLR = eps_eff * s * Ta4;
        F.LW_sfc = DLR - s * es * Ts4;
        F.LW_atm = eps_eff * (s * es * Ts4 - 2.0 * s * Ta4);
        if (P.gh_lock) {
            F.OLR = (1.0 - P.g_lw) * s * Ts4;
            const double DLRt = P.g_lw * s * Ts4;
            F.LW_sfc = DLRt - s * es * Ts4;
        }
    } else {
        const double eps = qd_clip(P.lw_eps0 + P.lw_kc * qd_clip(cloud_eff, 0.0, 1.0), 0.0, 1.0);
        F.OLR = eps * s * Ta4 + (1.0 - eps) * s * Ts4;
        const double DLR = eps * s * Ta4;
        F.LW_sfc = DLR - s * Ts4;
        F.LW_atm = eps * (s * Ts4 - 2.0 * s * Ta4);
        if (P.gh_lock) {
            F.OLR = (1.0 - P.g_lw) * s * Ts4;
            const double DLRt = P.g_lw * s * Ts4;
            F.LW_sfc = DLRt - s * Ts4;
        }
    }
    const double V = sqrt(u * u + v * v);
    F.SH = P.rhocpch * V * (Ts - Ta);
    return F;
}

