// qd_pointwise.h -- per-cell bodies of pointwise kernels that are launched on their own AND as stages of one merged launch
// (k_snow_albedo_forcing, qd_physics.hip: the tail of the driver physics + the forcing of the same step).  Each body is the former
// kernel body, unchanged; a stage hands its results to the next in registers instead of through memory.
#pragma once
#include "qd_internal.h"

// ------------------------------------------------------------------ P019 lapse + snow (run_simulation.py:1946-2019, hydrology.py:100-177)
struct QdSnowP {
    double dt, ga, rho_snow_safe, polar_lat, ice_max, elev_max, gamma, t_thresh, dT, ddf_s, tref, rate_s, swe_max, swe_ref_safe,
           gl_frac, gl_swe;
    int lapse, mode, swe;
};
struct QdSnowOut { double Pr, Sn, melt, Cs, gl; };

__device__ __forceinline__ QdSnowOut qd_snow_cell(const QdTabs& T, const QdSnowP& P, int i, bool is_land, double hval, double S0, double elev,
                                                  double Pf) {
    QdSnowOut r;
    const double T_a = 288.0 + P.ga * hval;
    const double h_snow = is_land ? qd_max(S0, 0.0) / P.rho_snow_safe : 0.0;
    const bool polar = fabs(T.lat_deg[i]) >= P.polar_lat;
    const double h_ice_eff = polar ? qd_min(h_snow, P.ice_max) : h_snow;
    const double H_eff = qd_min(elev + h_ice_eff, P.elev_max);
    const double T_hat = P.lapse ? T_a - P.gamma * (H_eff / 1000.0) : T_a;
    double f_snow = 1.0 / (1.0 + exp((T_hat - P.t_thresh) / P.dT));
    f_snow = qd_clip(f_snow, 0.0, 1.0);
    const double Ps = qd_nn(f_snow * Pf);
    const double Pr = qd_nn((1.0 - f_snow) * Pf);
    r.Pr = Pr;
    if (!P.swe) { r.Sn = S0; r.melt = 0.0; r.Cs = 0.0; r.gl = 0.0; return r; }
    const double Ps_land = Ps * (is_land ? 1.0 : 0.0);
    double melt_flux;
    if (P.mode == 0) melt_flux = P.ddf_s * qd_max(T_hat - P.tref, 0.0);
    else melt_flux = (T_hat >= P.t_thresh) ? P.rate_s : 0.0;
    const double actual = qd_min(qd_max(S0, 0.0), melt_flux * P.dt);
    double Sn = S0 + Ps_land * P.dt - actual;
    if (P.swe_max > 0.0) Sn = qd_min(Sn, P.swe_max);
    Sn = qd_max(0.0, Sn);
    const double melt_out = (P.dt > 0) ? actual / P.dt : 0.0;
    const double Cs = qd_clip(1.0 - exp(-qd_max(Sn, 0.0) / P.swe_ref_safe), 0.0, 1.0);
    Sn = qd_nn(Sn);
    const bool gl = is_land && ((Cs >= P.gl_frac) || (Sn >= P.gl_swe));
    // rain on an ice cap is deposited into the snowpack (run_simulation.py:1996-2001)
    const double Pr_gl = (Pr * (is_land ? 1.0 : 0.0)) * (gl ? 1.0 : 0.0);
    if (Pr_gl != 0.0) Sn = Sn + Pr_gl * P.dt;
    r.Sn = Sn; r.melt = qd_nn(melt_out); r.Cs = Cs; r.gl = gl ? 1.0 : 0.0;
    return r;
}

// ------------------------------------------------------------------ cloud tracer blend + dynamic albedo (physics.py:164-250)
struct QdAlbP { double alpha, hice_ref_safe, alpha_ice, alpha_cloud, alpha_water, alpha_snow, w_lai; int do_adv, use_topo, snow, eco, banded, water, eco_f32; };

// c: the cloud cover after the tracer blend (the caller stores it when P.do_adv); cs / gl: snow cover and glacier flag of the cell
__device__ __forceinline__ double qd_albedo_cell(const QdAlbP& P, size_t o, double c, const double* __restrict__ cloud_eff,
                                                 const double* __restrict__ hice, const double* __restrict__ base, int landv, double cs, double gl,
                                                 const double* __restrict__ eco_alpha, const double* __restrict__ banded,
                                                 const double* __restrict__ water) {
    const double crad = cloud_eff ? cloud_eff[o] : c;
    const double C = qd_clip(crad, 0.0, 1.0);
    const double ice_frac = 1.0 - exp(-qd_max(hice[o], 0.0) / P.hice_ref_safe);
    double fi = qd_clip(ice_frac, 0.0, 1.0);
    fi = fi * ((landv == 0) ? 1.0 : 0.0);
    double b0 = P.use_topo ? base[o] : P.alpha_water;
    if (P.eco && landv == 1 && gl == 0.0) {                  // run_simulation.py:2086-2100: ecology alpha, not on ice sheets
        const double ae = P.eco_f32 ? (double)reinterpret_cast<const float*>(eco_alpha)[o] : eco_alpha[o];
        if (fabs(ae) <= DBL_MAX) b0 = (1.0 - P.w_lai) * b0 + P.w_lai * ae;
    }
    if (P.banded && landv == 1) {                            // run_simulation.py:2107-2112: daily banded alpha
        const double ab = P.eco_f32 ? (double)reinterpret_cast<const float*>(banded)[o] : banded[o];
        if (fabs(ab) <= DBL_MAX) b0 = qd_clip(ab, 0.0, 1.0);
    }
    if (P.water && landv == 0) {                             // run_simulation.py:2121-2128: ocean colour
        const double aw = water[o];
        if (fabs(aw) <= DBL_MAX) b0 = qd_clip(aw, 0.0, 1.0);
    }
    if (P.snow && landv == 1) b0 = qd_clip((1.0 - cs) * b0 + cs * P.alpha_snow, 0.0, 1.0);      // run_simulation.py:2130-2141
    const double surf = b0 * (1.0 - fi) + P.alpha_ice * fi;
    return qd_clip(surf * (1.0 - C) + P.alpha_cloud * C, 0.0, 1.0);
}

// ------------------------------------------------------------------ two-star insolation + Teq (forcing.py:78-165)
struct QdStar { double flux, sin_d, cos_d, alpha; };
struct QdForcingP { QdStar A, B; double theta, sigma; int with_teq; };

// alb: the cell's albedo (only read when P.with_teq).  Returns {total insolation, Teq} of the cell; Teq == nullptr: not stored (the
// caller hands it on in a register)
struct QdForcingOut { double tot, teq; };
__device__ __forceinline__ QdForcingOut qd_forcing_cell(const QdTabs& T, const QdForcingP& P, int i, int j, size_t o, double alb,
                                                        double* __restrict__ isrA, double* __restrict__ isrB, double* __restrict__ isr,
                                                        double* __restrict__ Teq, double* __restrict__ eday, double eday_dt) {
    const double sl = T.sin_raw[i], cl = T.cos_raw[i], lon = T.lon_rad[j];
    const double hA = P.theta + lon - P.A.alpha;
    const double hB = P.theta + lon - P.B.alpha;
    const double czA = qd_max(0.0, sl * P.A.sin_d + cl * P.A.cos_d * cos(hA));
    const double czB = qd_max(0.0, sl * P.B.sin_d + cl * P.B.cos_d * cos(hB));
    const double a_ = P.A.flux * czA, b_ = P.B.flux * czB;
    const double tot = a_ + b_;
    if (isrA) { isrA[o] = a_; isrB[o] = b_; }                  // nullptr: a step inside a span whose per-star fluxes nobody reads (lazy diagnostics)
    isr[o] = tot;
    if (eday) eday[o] += qd_nn(tot) * eday_dt;                 // PopulationManager.step_subdaily (population.py:267-268)
    QdForcingOut r{tot, 0.0};
    if (P.with_teq) {
        double num = tot * (1 - alb);
        if (num < 0) num = 0;
        r.teq = sqrt(sqrt(num / P.sigma));         // (num / SIGMA) ** 0.25
        if (Teq) Teq[o] = r.teq;
    }
    return r;
}
