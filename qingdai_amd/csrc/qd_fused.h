// qd_fused.h -- argument blocks of the fused momentum + del^4 kernels (qd_fused.hip).
#pragma once
#include "qd_internal.h"

struct QdDynArgs {
    const double *u, *v, *h, *fric, *q, *cloud;
    double *uo, *vo, *ho, *qo, *co;
    const double* k4row[5];     // u v h q cloud; nullptr -> scalar k4s
    double k4s[5];
    int skip[5];                // k4 <= 0 early-out of _hyperdiffuse: field passes through
    double g, a, dt, dlat, dlon, f_min;
    double inv_dlon, inv_2dlon, inv_dlat, inv_2dlat, pgf_y;
    int primitive;
    int fast;                   // interior tiles may take the FAST path (QD_FUSED_FAST=0 disables)
    QdTileShape ts;
};


struct QdOcnArgs {
    const double *uo, *vo, *eta, *taux, *tauy;
    const uint8_t* land;
    double *uo_out, *vo_out, *eta_out;
    const double* k4row[3];
    double k4s[3];
    int skip[3];
    double a, g, dlat, dlon, sub_dt, rhoH, r_bot;
    double inv_2dlon, inv_2dlat, inv_a, inv_rhoH;
    // deferred end of the previous sub-step (ocean.py:375,436-443): eta <- clip(nan_to_num(eta - *eta_mean), -cap, cap) applied on
    // load instead of in a pass of its own; nullptr = eta is already final
    const double* eta_mean;
    double eta_cap;
    int fast;
    QdTileShape ts;
};


QdTileShape qd_pick_tile(const QdGeom& G);
int qd_launch_dyn_hyper(qd_ctx* c, QdDynArgs& P, int margin);
int qd_launch_ocn_hyper(qd_ctx* c, QdOcnArgs& P, int margin);

// qd_stream.hip: the row-streaming form of the same two kernels (default on grids of >= 64 columns)
bool qd_stream_ok(const qd_ctx* c, int margin);
bool qd_ocn_stream_ok(const qd_ctx* c, int margin);
int qd_launch_dyn_stream(qd_ctx* c, const QdDynArgs& P, int margin);
int qd_launch_ocn_stream(qd_ctx* c, const QdOcnArgs& P, int margin);
bool qd_ocn_stream_ok_list(const qd_ctx* c, const QdSegList& S);              // explicit row segments (interior / boundary rows around a halo exchange)
int qd_launch_ocn_stream_list(qd_ctx* c, const QdOcnArgs& P, const QdSegList& S);
