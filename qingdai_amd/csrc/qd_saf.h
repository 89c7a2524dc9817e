// qd_saf.h -- the last launch of the driver physics inside qd_step_n (whole-globe handles): snowpack -> cloud tracer blend + albedo ->
// two-star insolation + Teq (k_snow_albedo_forcing, qd_physics.hip), as ONE argument block and ONE per-cell body, so that time_step's
// column kernel can run it as its first stage when nothing separates the two launches (k_saf_column2, qd_atmos.hip).
#pragma once
#include "qd_internal.h"
#include "qd_pointwise.h"
#include "qd_fluxes.h"

struct QdSafArgs {
    QdGeom G; QdTabs T; QdSnowP S; QdAlbP A; QdForcingP Fo;
    const double *precip, *h, *S_snow, *elev; const uint8_t* land;
    double *P_rain, *S_next, *melt, *C_snow, *glacier;
    const double* adv; double* cloud; const double *cloud_eff, *hice, *base, *eco_alpha, *banded, *water;
    double *albedo, *isrA, *isrB, *isr, *Teq, *eday; double eday_dt;
    // inside a qd_step_n span: lazy diagnostic stores, and phase 1 of time_step's column riding along
    int write_diag, col1; QdColP P; const double *u, *v, *Ts, *q; double* Pcond;
};
struct QdSafOut { double cloud, albedo, isr, teq; };

// store_teq: false when the caller consumes Teq from the register (k_saf_column2) and nobody else reads the field inside the span
__device__ __forceinline__ QdSafOut qd_saf_cell(const QdSafArgs& K, int i, int j, size_t o, bool store_teq) {
    const int landv = K.land[o];
    const double hh = K.h[o];
    if (K.col1) {
        // phase 1 of time_step's column (k_column<1>, dynamics.py:282-297): this step's P_cond for the median time_step starts with
        const double hi = K.hice[o];
        const double qsat_air = qd_qsat(288.0 + K.P.ga * hh, K.P.p0);
        K.Pcond[o] = qd_humidity_column(K.P, K.u[o], K.v[o], K.Ts[o], K.q[o], qsat_air, landv == 1, hi).Pc;
    }
    const QdSnowOut r = qd_snow_cell(K.T, K.S, i, landv == 1, hh, K.S_snow[o], K.elev[o], K.precip[o]);
    if (K.write_diag) { K.P_rain[o] = r.Pr; K.S_next[o] = r.Sn; K.melt[o] = r.melt; K.C_snow[o] = r.Cs; K.glacier[o] = r.gl; }
    double c = K.cloud[o];
    if (K.A.do_adv) { c = qd_clip((1.0 - K.A.alpha) * c + K.A.alpha * K.adv[o], 0.0, 1.0); K.cloud[o] = c; }
    const double alb = qd_albedo_cell(K.A, o, c, K.cloud_eff, K.hice, K.base, landv, r.Cs, r.gl, K.eco_alpha, K.banded, K.water);
    K.albedo[o] = alb;
    const QdForcingOut f = qd_forcing_cell(K.T, K.Fo, i, j, o, alb, K.write_diag ? K.isrA : nullptr, K.isrB, K.isr, store_teq ? K.Teq : nullptr,
                                           K.eday, K.eday_dt);
    return QdSafOut{c, alb, f.tot, f.teq};
}
