// qd_ocntail.h -- argument block of k_ocn_tail (qd_ocntail.hip): the second half of an ocean sub-step in one launch
#pragma once
#include "qd_internal.h"
#include "qd_peer_dev.h"

struct QdTailArgs {
    const double *uo, *vo, *Ts, *qnet;
    const uint8_t *land, *ice;
    double *eta, *Ts_out, *uo_out, *vo_out, *partial;
    const double* tab;                                       // k_ocn_tail_fast: packed per-row coefficients [n_lat][16] (qd_ocntail.hip, k_tail_tab)
    unsigned long long* acc;                                 // fixed-point accumulator + tickets of the strip sums (qd_wave.h), or nullptr
    double* mean_out;                                        // acc != nullptr: the last workgroup writes sum / (wsum + 1e-15) here
    double wsum;
    double a, dlat, dlon, sub_dt, msdtH, alpha, K_h, rcH, ice_qfac, cap;
    double r_a, r_dlon, r_dlat, r_2dlon, r_2dlat, r_rcH;     // correctly rounded reciprocals of a, dlon, dlat, 2 dlon, 2 dlat, rcH (host)
    int use_q, has_ice, mean4, ntc, R, Rp;                   // R: strip height of the streaming forms; Rp: height of a pole strip (k_ocn_tail_fast)
    int nmid, flags;                                         // k_ocn_tail_fast: strips between the pole strips; bit0 = every wave takes the general form
    int own0, own1;                                          // rows whose eta enters the area-weighted sum (a band's owned rows; set by the launcher for whole-globe handles)
    const double* eta_in = nullptr;                          // k_ocn_fused, sequential form: eta' comes from this slab, the new eta goes to `eta` (nullptr: `eta` in place)
    // whole-globe k_ocn_tail_fast: nan_to_num and the speed cap leave all but a handful of cells of uo', vo' as they are, so uo'' / vo''
    // are NOT stored (2 x 8.3 MB per sub-step at 721 x 1440): a cell whose value changes is noted in a list -- {byte offset in the slab,
    // u bits, v bits} -- and the launch's finishing wave, which runs when every wave of every strip is done, patches uo', vo' in place
    unsigned int* fix_count = nullptr;
    unsigned long long* fix_list = nullptr;
    QdPeerFold pf;                                           // latitude bands over the peer exchange: the finishing wave all-reduces the band's share itself (qd_peer_dev.h)
};

int qd_ocn_tail_tiles(const qd_ctx* c, const QdGeom& G);
int qd_launch_ocn_tail(qd_ctx* c, const QdGeom& G, QdTailArgs& P);

// the whole sub-step in one launch, streaming form (QD_OCN_FUSED=1; qd_ocntail.hip, k_ocn_fused)
struct QdOcnArgs;
bool qd_ocn_fused_ok(const qd_ctx* c);
int qd_ocn_fused_tiles(const qd_ctx* c);
int qd_launch_ocn_fused(qd_ctx* c, const QdOcnArgs& O, QdTailArgs& P);
