// qd_atmos.hip -- SpectralModel.time_step (pygcm/dynamics.py:260-667) on gfx950.
//
// Kernel sequence of one atmosphere step (global syncs force the cuts):
//   k_column<PHASE>   humidity column (humidity.py:85-183) + Newton / explicit energy surface
//                     temperature incl. sea ice (energy.py:77-449) + radiative / energy h forcing
//                     -- pointwise, one pass over 11 inputs
//   [radix-select]    median of P_cond>0 only when QD_PCOND_REF is unset (dynamics.py:344-348)
//   k_advect          T_s, q semi-Lagrangian gather with the OLD winds, alpha=0.2 (dynamics.py:454-461)
//   k_momentum        np.gradient(h) + geostrophic relaxation | primitive update (dynamics.py:482-530)
//   k_laplacian / k_hyper_apply   del^4 on u,v,h,q,cloud batched in one launch each (dynamics.py:533-594)
//   k_shapiro_pass    every QD_SHAPIRO_EVERY steps (dynamics.py:610-626)
//   k_final           cloud gather with the NEW winds, decay, 0.998 damp, nan_to_num (dynamics.py:642-667)
#include "qd_internal.h"
#include "qd_pointwise.h"
#include "qd_device.h"
#include "qd_saf.h"

#include "qd_fluxes.h"
#include "qd_fused.h"

struct QdColPtrs {
    const double *u, *v, *Teq, *isr, *albedo, *csmap;
    double *h, *Ts, *q, *cloud, *hice;
    double *E, *Pcond, *LH, *LHrel, *olr, *cloud_eff;
    const uint8_t* land;
    const double* pref;                // device scalar: P_ref for the tanh term
    unsigned long long* npos;          // device counter of P_cond > 0 cells (phase 1)
};

// PHASE 0: whole column in one pass (P_ref known up front or not needed)
// PHASE 1: the humidity part as far as P_cond, the only thing the median between the phases reads: ONE field written
// PHASE 2: the whole column again, now with P_ref = median(P_cond > 0) on the device.  (Round 3b: phase 1 used to write q, E,
//          P_cond, LH and LH_release and phase 2 to read them back -- 33 MB more written and the same number read as it costs phase 2
//          to take q, u, v, T_s once more and redo ~100 f64 instructions; the humidity arithmetic is the same in both phases, so the
//          P_cond phase 2 writes is the one the median saw.)
// what a stage in front of the column hands it in registers instead of through memory (k_saf_column2): the cloud cover after the tracer
// blend, the insolation, the albedo and Teq of the cell
struct QdColIn { double cloud, isr, albedo, teq; };

// one cell of the column; IN = nullptr: everything from memory
template <int PHASE, bool HAS_ALB>
__device__ __forceinline__ void qd_column_cell(const QdColP& P, const QdColPtrs& A, size_t o, const QdColIn* IN, int i, int nlat) {
    const double u = A.u[o], v = A.v[o], h = A.h[o], Ts = A.Ts[o];
    const double hice = A.hice[o];
    const bool land = (A.land[o] == 1);
    const bool ocean = !land;
    const double T_a = 288.0 + P.ga * h;
    const double qsat_air = qd_qsat(T_a, P.p0);

    // ---- humidity column: dynamics.py:282-297 (qd_fluxes.h)
    const QdHum Hm = qd_humidity_column(P, u, v, Ts, A.q[o], qsat_air, land, hice);
    const double q = Hm.q, E = Hm.E, Pc = Hm.Pc, LH = Hm.LH, LHrel = Hm.LHrel;
    if (PHASE == 1) { A.Pcond[o] = Pc; return; }
    A.q[o] = q; A.LH[o] = LH;
    if (PHASE != 2) A.Pcond[o] = Pc;                         // phase 2: phase 1 has stored this very value (same arithmetic on the same inputs)
    if (P.write_diag) { A.E[o] = E; A.LHrel[o] = LHrel; }    // nobody inside a span reads these two (the hydrology commit asks for them: qd_step_n)

    // ---- Newton path: dynamics.py:304-322
    const double Teq = IN ? IN->teq : A.Teq[o];
    const double olr_old = P.sigma * qd_pow4(Ts);
    const double net_old = P.sigma * qd_pow4(Teq) + P.gfs * qd_pow4(T_a) - olr_old;
    const double Ts_newton = Ts + (net_old / P.c_sfc_safe) * P.dt;

    double Ts_new, h_new = h;
    QdFlux F;
    if (HAS_ALB) {
        // ---- cloud optical consistency: dynamics.py:329-353
        const double cloud = IN ? IN->cloud : A.cloud[o];
        double cloud_eff;
        if (P.couple) {
            const double RH = qd_clip(q / qd_max(1e-12, qsat_air), 0.0, 1.5);
            const double rh_excess = qd_max(0.0, RH - P.rh0);
            const double P_ref = *A.pref;
            const double p_term = qd_tanh(P_ref > 0 ? Pc / P_ref : 0.0);
            cloud_eff = qd_clip(cloud + P.k_q * rh_excess + P.k_p * p_term, 0.0, 1.0);
        } else cloud_eff = cloud;
        A.cloud_eff[o] = cloud_eff;
        F = qd_surface_fluxes(P, IN ? IN->isr : A.isr[o], IN ? IN->albedo : A.albedo[o], cloud_eff, Ts, T_a, u, v, land, hice);
        double Ts_energy;
        if (P.seaice) {
            // ---- energy.py:291-420
            double Q = F.SW_sfc - F.LW_sfc - F.SH - LH;
            double Tn = Ts, hi = hice;
            if (hi > 0.0 && ocean && Q > 0.0) {
                const double dh_melt = (Q * P.dt) / P.rhoiLf;
                const double dh_cap = qd_min(dh_melt, hi);
                hi -= dh_cap;
                Q = Q - (dh_cap * P.rho_i * P.L_f) / P.dt;
            }
            if (ocean && Q < 0.0 && Tn <= (P.t_freeze + 0.5)) {
                hi += (-Q * P.dt) / P.rhoiLf;
                Q = 0.0;
                Tn = qd_min(Tn, P.t_freeze);
            }
            double Cs = land ? P.Cs_land : (hi > 0.0 ? P.Cs_ice : P.Cs_ocean);
            Cs = (isfinite(Cs) && Cs > 1e3) ? Cs : 1e3;
            Tn = Tn + (Q / Cs) * P.dt;
            if ((i == 0 && P.fix_s) || (i == nlat - 1 && P.fix_n)) {
                if (ocean && Q < 0.0 && Tn > P.t_freeze) Tn = P.t_freeze;
            }
            if (hi > 0.0 && ocean) Tn = qd_min(Tn, P.t_freeze);
            Tn = qd_max(P.t_floor, Tn);
            Ts_energy = qd_nn(Tn);
            { const double hn_ = qd_nn(hi); if (__double_as_longlong(hn_) != __double_as_longlong(hice)) A.hice[o] = hn_; }    // ice-free and land cells: unchanged
        } else {
            const double net = F.SW_sfc - F.LW_sfc - F.SH - LH;
            double Cs;
            if (P.has_csmap) { const double cm = A.csmap[o]; Cs = (isfinite(cm) && cm > 1e3) ? cm : 1e3; }
            else Cs = P.c_sfc_safe;
            Ts_energy = qd_nn(qd_max(P.t_floor, Ts + (net / Cs) * P.dt));
        }
        if (P.write_diag) A.olr[o] = F.OLR;
        Ts_new = (1.0 - P.w_energy) * Ts_newton + P.w_energy * Ts_energy;
    } else {
        if (P.write_diag) A.olr[o] = olr_old;
        Ts_new = Ts_newton;
    }
    A.Ts[o] = Ts_new;

    // ---- radiative relaxation of h: dynamics.py:464-467; atmosphere energy -> h: 470-480
    const double h_eq = P.h_eq_fac * Teq;
    h_new = h + ((h_eq - h) / P.tau_rad) * P.dt;
    if (HAS_ALB && P.atm_couple) {
        const double F_atm = F.SW_atm + F.LW_atm + F.SH + LHrel;
        h_new = qd_nn(h_new + P.atm_w * (F_atm / P.atm_denom) * P.dt);
    }
    A.h[o] = h_new;
}

template <int PHASE, bool HAS_ALB>
__global__ void __launch_bounds__(QD_BLOCK)
k_column(QdGeom G, QdColP P, QdColPtrs A) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    qd_column_cell<PHASE, HAS_ALB>(P, A, (size_t)qd_lrow(G, i) * G.nlon + j, nullptr, i, G.nlat);
}

// The driver physics' last launch (snowpack -> cloud blend + albedo -> insolation + Teq: qd_saf.h) as the first STAGE of the column's
// phase 2 (whole-globe qd_step_n, when the P_cond median has run ahead of the cloud block and nothing separates the two launches any
// more): the cloud cover, the albedo, the insolation and Teq reach the column in registers -- four fields not read back, Teq not even
// stored inside a span -- and h, h_ice and the land mask are read once.  Same per-cell bodies as the two kernels: same bits.
__global__ void __launch_bounds__(QD_BLOCK)
k_saf_column2(QdSafArgs K, QdColP P, QdColPtrs A) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= K.G.nlon) return;
    const int i = K.G.row0 + tl.row;
    const size_t o = (size_t)qd_lrow(K.G, i) * K.G.nlon + j;
    const QdSafOut s = qd_saf_cell(K, i, j, o, K.write_diag != 0);
    const QdColIn in{s.cloud, s.isr, s.albedo, s.teq};
    qd_column_cell<2, true>(P, A, o, &in, i, K.G.nlat);
}

// ------------------------------------------------------------------ momentum: dynamics.py:482-530
struct QdMomP { double g, a, dt, dlat, dlon, f_min; int primitive; };

__global__ void __launch_bounds__(QD_BLOCK)
k_momentum(QdGeom G, QdTabs T, QdMomP P, const double* __restrict__ h, const double* __restrict__ fric,
           double* __restrict__ u, double* __restrict__ v) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const int n = G.nlat, m = G.nlon;
    const size_t b = (size_t)qd_lrow(G, i) * m;
    const size_t o = b + j;
    // np.gradient(h, dlon, axis=1): NOT periodic
    double dh_dlon;
    if (j == 0) dh_dlon = (h[b + 1] - h[b]) / P.dlon;
    else if (j == m - 1) dh_dlon = (h[b + m - 1] - h[b + m - 2]) / P.dlon;
    else dh_dlon = (h[b + j + 1] - h[b + j - 1]) / (2.0 * P.dlon);
    double dh_dlat;
    if (i == 0) dh_dlat = (h[(size_t)qd_lrow(G, 1) * m + j] - h[o]) / P.dlat;
    else if (i == n - 1) dh_dlat = (h[o] - h[(size_t)qd_lrow(G, n - 2) * m + j]) / P.dlat;
    else dh_dlat = (h[(size_t)qd_lrow(G, i + 1) * m + j] - h[(size_t)qd_lrow(G, i - 1) * m + j]) / (2.0 * P.dlat);
    const double cosc = T.cos6[i];
    const double f = T.fcor[i];
    const double r = fric[o];
    const double u0 = u[o], v0 = v[o];
    if (P.primitive) {
        const double PGF_x = -(P.g / (P.a * cosc)) * dh_dlon;
        const double PGF_y = -(P.g / P.a) * dh_dlat;
        const double du = (PGF_x + f * v0 - r * u0) * P.dt;
        const double dv = (PGF_y - f * u0 - r * v0) * P.dt;
        u[o] = qd_clip(u0 + du, -200.0, 200.0);
        v[o] = qd_clip(v0 + dv, -200.0, 200.0);
    } else {
        const double sgn = (f >= 0.0) ? 1.0 : -1.0;
        const double f_safe = (fabs(f) < P.f_min) ? sgn * P.f_min : f;
        const double u_g = qd_clip(-(P.g / (f_safe * P.a * cosc)) * dh_dlat, -200.0, 200.0);
        const double v_g = qd_clip((P.g / (f_safe * P.a)) * dh_dlon, -200.0, 200.0);
        double un = u0 * 0.8 + u_g * 0.2;
        double vn = v0 * 0.8 + v_g * 0.2;
        un = un + (-r * un) * P.dt;
        vn = vn + (-r * vn) * P.dt;
        u[o] = un; v[o] = vn;
    }
}

// ------------------------------------------------------------------ final: dynamics.py:642-667
// cloud <- advect(cloud) with the post-filter winds, 2-day decay, then the global damp and the
// nan_to_num scrub of every prognostic field.  Winds are read unscaled by the gather (it only
// needs the cell's own u, v) and written back scaled in the same pass.
__global__ void __launch_bounds__(QD_BLOCK)
k_final(QdGeom G, const double* __restrict__ cosl, double dt, double a, double dlat, double dlon,
        double* __restrict__ u, double* __restrict__ v, double* __restrict__ h, double* __restrict__ Ts,
        double* __restrict__ q, const double* __restrict__ cloud_in, double* __restrict__ cloud_out,
        double decay, double dfac) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
    const double uu = u[o], vv = v[o];
    const QdBilin b = qd_departure(G, i, j, uu, vv, dt, a, cosl[i], dlat, dlon);
    double c = qd_gather(cloud_in, G, b);
    c = c * decay;
    cloud_out[o] = qd_nn(c * dfac);
    u[o] = qd_nn(uu * dfac);
    v[o] = qd_nn(vv * dfac);
    h[o] = qd_nn(h[o] * dfac);
    q[o] = qd_nn(q[o] * dfac);
    Ts[o] = qd_nn(Ts[o]);
}

// ------------------------------------------------------------------ forcing.py:78-165 (body: qd_pointwise.h)
__global__ void __launch_bounds__(QD_BLOCK)
k_forcing(QdGeom G, QdTabs T, QdForcingP P, double* __restrict__ isrA, double* __restrict__ isrB, double* __restrict__ isr,
          const double* __restrict__ albedo, double* __restrict__ Teq, double* __restrict__ eday, double eday_dt) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
    qd_forcing_cell(T, P, i, j, o, P.with_teq ? albedo[o] : 0.0, isrA, isrB, isr, Teq, eday, eday_dt);
}

int qd_forcing_impl(qd_ctx* c, const double* sa, const double* sb, double theta, int with_teq) {
    QdScope sc(c, "forcing");
    QdStar A{sa[0], std::sin(sa[1]), std::cos(sa[1]), sa[2]};
    QdStar B{sb[0], std::sin(sb[1]), std::cos(sb[1]), sb[2]};
    const int m = with_teq ? qd_plan(c, {QD_IN(c->f[QD_F_ALBEDO], 0)}) : c->geo.halo;
    if (m < 0) return -1;
    const QdForcingP FP{A, B, theta, 5.670374e-8, with_teq};
    QD_ROWS(c, m, G, hipLaunchKernelGGL(k_forcing, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, c->tabs, FP,
                                        c->f[QD_F_ISR_A], c->f[QD_F_ISR_B], c->f[QD_F_ISR],
                                        c->f[QD_F_ALBEDO], c->f[QD_F_TEQ], c->eco.eday_dt > 0 ? c->f[QD_F_ECO_EDAY] : (double*)nullptr,
                                        c->eco.eday_dt));
    c->eco.eday_dt = 0;
    qd_mark(c, {c->f[QD_F_ISR_A], c->f[QD_F_ISR_B], c->f[QD_F_ISR]}, m);
    if (with_teq) qd_mark(c, {c->f[QD_F_TEQ]}, m);
    return 0;
}

__global__ void __launch_bounds__(QD_BLOCK)
k_simple_albedo(QdGeom G, const uint8_t* __restrict__ land, const double* __restrict__ base, double ocean_albedo,
                double* __restrict__ albedo) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    albedo[o] = (land[o] == 0) ? ocean_albedo : base[o];
}

int qd_simple_albedo_impl(qd_ctx* c, double ocean_albedo) {
    const int m = c->geo.full ? 0 : c->geo.halo;              // static inputs: valid on the whole slab
    QD_ROWS(c, m, G, hipLaunchKernelGGL(k_simple_albedo, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, c->land,
                                        c->f[QD_F_BASE_ALBEDO], ocean_albedo, c->f[QD_F_ALBEDO]));
    qd_mark(c, {c->f[QD_F_ALBEDO]}, m);
    return 0;
}

// ------------------------------------------------------------------ host orchestration
static inline bool qd_isset(double x) { return !(x != x); }

QdColP qd_make_colp(const qd_ctx* c, double dt) {
    const qd_params& p = c->p;
    QdColP P;
    P.ga = p.g / 1004.0;
    P.M_col = std::max(1e-6, p.rho_a * p.h_mbl);
    P.tau_c = std::max(1e-6, p.tau_cond);
    P.L_v = p.L_v; P.p0 = p.p0; P.rhoCE = p.rho_a * p.C_E;
    P.s_ocean = p.ocean_evap_scale; P.s_land = p.land_evap_scale; P.s_ice = p.ice_evap_scale;
    P.sigma = 5.670374e-8;
    P.gfs = p.greenhouse_factor * P.sigma;
    P.c_sfc_safe = std::max(1e-12, p.c_sfc);
    P.dt = dt;
    P.rh0 = p.rh0; P.k_q = p.k_q; P.k_p = p.k_p;
    P.sw_a0 = p.sw_a0; P.sw_kc = p.sw_kc; P.lw_eps0 = p.lw_eps0; P.lw_kc = p.lw_kc;
    P.eps_clear = std::min(std::max(p.lw_eps0, 0.0), 1.0);
    P.tau0 = p.lw_tau0; P.k_tau = p.lw_ktau;
    P.hice_ref_safe = std::max(1e-6, p.hice_ref);
    P.eps_ocean = p.eps_ocean; P.eps_land = p.eps_land; P.eps_ice = p.eps_ice; P.eps_default = p.eps_default;
    P.g_lw = p.gh_factor_lw;
    P.rhocpch = p.rho_a * p.cp_a * p.ch;
    P.t_freeze = p.t_freeze; P.rho_i = p.rho_i; P.L_f = p.L_f; P.rhoiLf = p.rho_i * p.L_f;
    P.Cs_ocean = p.Cs_ocean; P.Cs_land = p.Cs_land; P.Cs_ice = p.Cs_ice; P.t_floor = p.t_floor;
    P.w_energy = std::min(1.0, std::max(0.0, p.energy_w));
    P.h_eq_fac = 287.0 / p.g;
    P.tau_rad = p.tau_rad;
    const double H_atm = qd_isset(p.atm_h) ? p.atm_h : p.h_mbl;
    P.atm_denom = std::max(1e-6, p.rho_a) * std::max(1.0, H_atm) * p.g;
    P.atm_w = p.energy_w;
    P.atm_couple = (p.energy_w > 0.0) ? 1 : 0;
    P.couple = p.cloud_couple; P.lw_v2 = p.lw_v2; P.gh_lock = p.gh_lock; P.seaice = p.seaice_enabled;
    P.fix_s = p.polar_freeze_fix_s; P.fix_n = p.polar_freeze_fix_n; P.has_csmap = p.has_csmap;
    P.write_diag = c->diag_write;
    return P;
}

__global__ void k_set_scalar(double* p, double v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = v; }
__global__ void k_zero_count(unsigned long long* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = 0ull; }

static QdColPtrs qd_col_ptrs(qd_ctx* c) {
    double** F = c->f;
    QdColPtrs A;
    A.u = F[QD_F_U]; A.v = F[QD_F_V]; A.Teq = F[QD_F_TEQ]; A.isr = F[QD_F_ISR];
    A.albedo = F[QD_F_ALBEDO]; A.csmap = F[QD_F_CSMAP];
    A.h = F[QD_F_H]; A.Ts = F[QD_F_TS]; A.q = F[QD_F_Q]; A.cloud = F[QD_F_CLOUD]; A.hice = F[QD_F_HICE];
    A.E = F[QD_F_EFLUX]; A.Pcond = F[QD_F_PCOND]; A.LH = F[QD_F_LH]; A.LHrel = F[QD_F_LHREL];
    A.olr = F[QD_F_OLR]; A.cloud_eff = F[QD_F_CLOUD_EFF];
    A.land = c->land; A.pref = c->dscal + QD_S_PREF; A.npos = c->dcount;
    return A;
}

// QD_MED_SIDE (whole-globe qd_step_n): phase 1 of the column and the P_cond median read nothing the driver physics writes and are
// four launches that leave most of the chip idle (one workgroup per CU, chains of dependent round trips): they run on the side
// stream BESIDE the physics' launches.  Fork: everything queued so far (the ocean step's T_s injection is the last writer of an
// input); join: qd_atmos_step_impl, in front of the column's phase 2.
int qd_pcond_phase1(qd_ctx* c, double dt) {
    hipLaunchKernelGGL((k_column<1, true>), qd_grid2d(c->geo), dim3(QD_BLOCK), 0, c->stream, c->geo, qd_make_colp(c, dt), qd_col_ptrs(c));
    return 0;
}
int qd_pcond_median_side(qd_ctx* c, double dt) {
    QD_HIP(c, hipEventRecord(c->med_fork, c->stream));
    QD_HIP(c, hipStreamWaitEvent(c->side_stream, c->med_fork, 0));
    std::swap(c->stream, c->side_stream);
    c->med_side_active = true;
    const QdColP P = qd_make_colp(c, dt);
    const QdColPtrs A = qd_col_ptrs(c);
    hipLaunchKernelGGL((k_column<1, true>), qd_grid2d(c->geo), dim3(QD_BLOCK), 0, c->stream, c->geo, P, A);
    const int r = qd_median_positive_dev(c, c->f[QD_F_PCOND], 1e-6, QD_S_PREF, 0, 0.0, 1);
    c->med_side_active = false;
    std::swap(c->stream, c->side_stream);
    if (r) return -1;
    QD_HIP(c, hipEventRecord(c->med_done, c->side_stream));
    c->pcond_ahead = 2;
    return 0;
}

int qd_atmos_step_impl(qd_ctx* c, double dt, int has_albedo) {
    const qd_params& p = c->p;
    const QdGeom& G0 = c->geo;
    const dim3 blk(QD_BLOCK);
    double** F = c->f;
    QdColP P = qd_make_colp(c, dt);
    const int R = qd_adv_reach(c, dt, 250.0);      // lat reach of the gather for |v| <= 250 m/s

    // (a launch the driver physics left for the column kernel is flushed by every path that does not merge it)
    if (c->saf_pending && !(has_albedo && p.cloud_couple && !qd_isset(p.pcond_ref) && c->pcond_ahead == 3) && qd_saf_flush(c)) return -1;
    {
        QdScope sc(c, "column");
        // pointwise: every input at radius 0
        int m = has_albedo
            ? qd_plan(c, {QD_IN(F[QD_F_U], 0), QD_IN(F[QD_F_V], 0), QD_IN(F[QD_F_H], 0), QD_IN(F[QD_F_TS], 0), QD_IN(F[QD_F_Q], 0),
                          QD_IN(F[QD_F_CLOUD], 0), QD_IN(F[QD_F_HICE], 0), QD_IN(F[QD_F_TEQ], 0), QD_IN(F[QD_F_ISR], 0),
                          QD_IN(F[QD_F_ALBEDO], 0)})
            : qd_plan(c, {QD_IN(F[QD_F_U], 0), QD_IN(F[QD_F_V], 0), QD_IN(F[QD_F_H], 0), QD_IN(F[QD_F_TS], 0), QD_IN(F[QD_F_Q], 0),
                          QD_IN(F[QD_F_HICE], 0), QD_IN(F[QD_F_TEQ], 0)});
        if (m < 0) return -1;
        QdColPtrs A;
        A.u = F[QD_F_U]; A.v = F[QD_F_V]; A.Teq = F[QD_F_TEQ]; A.isr = F[QD_F_ISR];
        A.albedo = F[QD_F_ALBEDO]; A.csmap = F[QD_F_CSMAP];
        A.h = F[QD_F_H]; A.Ts = F[QD_F_TS]; A.q = F[QD_F_Q]; A.cloud = F[QD_F_CLOUD]; A.hice = F[QD_F_HICE];
        A.E = F[QD_F_EFLUX]; A.Pcond = F[QD_F_PCOND]; A.LH = F[QD_F_LH]; A.LHrel = F[QD_F_LHREL];
        A.olr = F[QD_F_OLR]; A.cloud_eff = F[QD_F_CLOUD_EFF];
        A.land = c->land; A.pref = c->dscal + QD_S_PREF; A.npos = c->dcount;
        if (!has_albedo) {
            QD_ROWS(c, m, G, hipLaunchKernelGGL((k_column<0, false>), qd_grid2d(G), blk, 0, c->stream, G, P, A));
        } else if (!p.cloud_couple || qd_isset(p.pcond_ref)) {
            hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, c->stream, c->dscal + QD_S_PREF,
                               qd_isset(p.pcond_ref) ? p.pcond_ref : 1e-6);
            QD_ROWS(c, m, G, hipLaunchKernelGGL((k_column<0, true>), qd_grid2d(G), blk, 0, c->stream, G, P, A));
        } else {
            // (whole-globe qd_step_n: the last launch of the driver physics has written this step's P_cond already: k_snow_albedo_forcing)
            // 1: P_cond is there; 2: its median too, on the side stream (qd_pcond_median_side); 3: its median too, on this stream (the
            // cloud block ran it together with the precipitation median: qd_median_pair_dev)
            const int ahead = c->pcond_ahead;
            c->pcond_ahead = 0;
            if (!ahead) QD_ROWS(c, m, G, hipLaunchKernelGGL((k_column<1, true>), qd_grid2d(G), blk, 0, c->stream, G, P, A));
            if (ahead == 2) QD_HIP(c, hipStreamWaitEvent(c->stream, c->med_done, 0));
            else if (ahead != 3 && qd_median_positive_dev(c, F[QD_F_PCOND], 1e-6, QD_S_PREF, 0, 0.0, 1)) return -1;
            if (c->saf_pending && ahead == 3 && c->geo.full) {
                // the driver physics left its last launch for this one (qd_physics.hip): one kernel does both
                QdSafArgs* K = (QdSafArgs*)c->saf_pending;
                c->saf_pending = nullptr;
                hipLaunchKernelGGL(k_saf_column2, qd_grid2d(K->G), blk, 0, c->stream, *K, P, A);
                delete K;
            } else {
                if (qd_saf_flush(c)) return -1;
                QD_ROWS(c, m, G, hipLaunchKernelGGL((k_column<2, true>), qd_grid2d(G), blk, 0, c->stream, G, P, A));
            }
        }
        qd_mark(c, {F[QD_F_H], F[QD_F_TS], F[QD_F_Q], F[QD_F_EFLUX], F[QD_F_PCOND], F[QD_F_LH], F[QD_F_LHREL], F[QD_F_OLR]}, m);
        if (has_albedo) { qd_mark(c, {F[QD_F_HICE], F[QD_F_CLOUD_EFF]}, m); c->cloud_eff_valid = 1; }
    }
    c->atm_counter += 1;
    const int64_t sc_ = c->atm_counter;

    // T_s, q gather with the OLD winds (dynamics.py:454-461)
    {
        QdScope sc(c, "advect_tsq");
        const int m = qd_plan(c, {QD_IN(F[QD_F_TS], R), QD_IN(F[QD_F_Q], R), QD_IN(F[QD_F_U], 0), QD_IN(F[QD_F_V], 0)});
        if (m < 0) return -1;
        double* oT = qd_scratch(c, 0); double* oq = qd_scratch(c, 1);
        qd_launch_advect(c, F[QD_F_U], F[QD_F_V], c->tabs.cos6, dt, F[QD_F_TS], oT, F[QD_F_Q], oq, 0.2, 1, m);
        qd_swap(c, QD_F_TS, 0); qd_swap(c, QD_F_Q, 1);
    }
    // momentum (dynamics.py:482-530) + del^4 (dynamics.py:533-594)
    const int ft = p.filter_type;
    const bool do_diff = p.diff_enable && (ft == 0 || ft == 1) && (sc_ % std::max(1, p.diff_every) == 0);
    const double f_min = 2.0 * p.omega * std::sin(5.0 * (M_PI / 180.0));
    if (do_diff) { int rc = qd_build_k4_tables(c, dt, false, 0.0); if (rc) return rc; }
    const double ov[5] = {p.k4_u, p.k4_v, p.k4_h, p.k4_q, p.k4_cloud};
    if (do_diff && c->use_fused && p.k4_nsub == 1) {
        // one launch: u,v,h,friction,q,cloud in -> u,v,h,q,cloud out
        const int m = qd_plan(c, {QD_IN(F[QD_F_H], 5), QD_IN(F[QD_F_U], 4), QD_IN(F[QD_F_V], 4), QD_IN(F[QD_F_FRICTION], 4),
                                  QD_IN(F[QD_F_Q], 4), QD_IN(F[QD_F_CLOUD], 4)});
        if (m < 0) return -1;
        QdDynArgs D;
        D.u = F[QD_F_U]; D.v = F[QD_F_V]; D.h = F[QD_F_H]; D.fric = F[QD_F_FRICTION];
        D.q = F[QD_F_Q]; D.cloud = F[QD_F_CLOUD];
        D.uo = qd_scratch(c, 0); D.vo = qd_scratch(c, 1); D.ho = qd_scratch(c, 2); D.qo = qd_scratch(c, 3); D.co = qd_scratch(c, 4);
        for (int f = 0; f < 5; ++f) {
            const bool sc1 = qd_isset(ov[f]);
            D.k4row[f] = sc1 ? nullptr : c->k4_atm + (size_t)f * G0.nlat;
            D.k4s[f] = sc1 ? ov[f] : 0.0;
            D.skip[f] = c->k4_atm_skip[f];
        }
        D.g = p.g; D.a = p.a; D.dt = dt; D.dlat = c->dlat; D.dlon = c->dlon; D.f_min = f_min; D.primitive = p.mom_scheme == 1;
        D.inv_dlon = 1.0 / c->dlon; D.inv_2dlon = 1.0 / (2.0 * c->dlon); D.inv_dlat = 1.0 / c->dlat; D.inv_2dlat = 1.0 / (2.0 * c->dlat);
        D.pgf_y = -(p.g / p.a);
        if (qd_launch_dyn_hyper(c, D, m)) return -1;
        qd_mark(c, {D.uo, D.vo, D.ho, D.qo, D.co}, m);
        qd_swap(c, QD_F_U, 0); qd_swap(c, QD_F_V, 1); qd_swap(c, QD_F_H, 2); qd_swap(c, QD_F_Q, 3); qd_swap(c, QD_F_CLOUD, 4);
    } else {
        {
            QdScope sc(c, "momentum");
            const int m = qd_plan(c, {QD_IN(F[QD_F_H], 1), QD_IN(F[QD_F_U], 0), QD_IN(F[QD_F_V], 0), QD_IN(F[QD_F_FRICTION], 0)});
            if (m < 0) return -1;
            QdMomP M;
            M.g = p.g; M.a = p.a; M.dt = dt; M.dlat = c->dlat; M.dlon = c->dlon; M.f_min = f_min;
            M.primitive = p.mom_scheme == 1;
            QD_ROWS(c, m, G, hipLaunchKernelGGL(k_momentum, qd_grid2d(G), blk, 0, c->stream, G, c->tabs, M, F[QD_F_H],
                                                F[QD_F_FRICTION], F[QD_F_U], F[QD_F_V]));
            qd_mark(c, {F[QD_F_U], F[QD_F_V]}, m);
        }
        if (do_diff) {
            QdScope sc(c, "hyperdiffusion");
            const int ns = std::max(1, p.k4_nsub);
            const int m3 = qd_plan(c, {QD_IN(F[QD_F_U], 4 * ns), QD_IN(F[QD_F_V], 4 * ns), QD_IN(F[QD_F_H], 4 * ns)});
            const int m2 = qd_plan(c, {QD_IN(F[QD_F_Q], 4), QD_IN(F[QD_F_CLOUD], 4)});
            if (m3 < 0 || m2 < 0) return -1;
            double* fl[5] = {F[QD_F_U], F[QD_F_V], F[QD_F_H], F[QD_F_Q], F[QD_F_CLOUD]};
            // u,v,h use QD_K4_NSUB sub-steps, q and cloud always one (dynamics.py:584-594)
            int skip3[5] = {c->k4_atm_skip[0], c->k4_atm_skip[1], c->k4_atm_skip[2], 1, 1};
            int skip2[5] = {1, 1, 1, c->k4_atm_skip[3], c->k4_atm_skip[4]};
            qd_hyperdiffuse_fields(c, fl, 5, c->k4_atm, skip3, ov, dt, ns, c->tabs.cos02, m3);
            qd_hyperdiffuse_fields(c, fl, 5, c->k4_atm, skip2, ov, dt, 1, c->tabs.cos02, m2);
            F[QD_F_U] = fl[0]; F[QD_F_V] = fl[1]; F[QD_F_H] = fl[2]; F[QD_F_Q] = fl[3]; F[QD_F_CLOUD] = fl[4];
        }
    }
    // Shapiro (dynamics.py:610-626): combo, shapiro AND hyper4 all trigger it
    if ((ft == 0 || ft == 1 || ft == 2) && p.shapiro_every > 0 && (sc_ % p.shapiro_every == 0)) {
        QdScope sc(c, "shapiro");
        const int np_ = std::max(1, p.shapiro_n);
        const int m = qd_plan(c, {QD_IN(F[QD_F_U], np_), QD_IN(F[QD_F_V], np_), QD_IN(F[QD_F_H], np_)});
        if (m < 0) return -1;
        double* fl[3] = {F[QD_F_U], F[QD_F_V], F[QD_F_H]};
        qd_shapiro_fields(c, fl, 3, np_, m);
        F[QD_F_U] = fl[0]; F[QD_F_V] = fl[1]; F[QD_F_H] = fl[2];
        const int n1 = std::max(1, p.shapiro_n - 1);
        if (p.diff_q) {
            const int mq = qd_plan(c, {QD_IN(F[QD_F_Q], n1)}); if (mq < 0) return -1;
            double* g1[1] = {F[QD_F_Q]}; qd_shapiro_fields(c, g1, 1, n1, mq); F[QD_F_Q] = g1[0];
        }
        if (p.diff_cloud) {
            const int mc = qd_plan(c, {QD_IN(F[QD_F_CLOUD], n1)}); if (mc < 0) return -1;
            double* g1[1] = {F[QD_F_CLOUD]}; qd_shapiro_fields(c, g1, 1, n1, mc); F[QD_F_CLOUD] = g1[0];
        }
    }
    // zonal spectral filter (dynamics.py:627-637): spectral and combo
    if ((ft == 0 || ft == 3) && p.spec_every > 0 && (sc_ % p.spec_every == 0)) {
        QdScope sc(c, "zonal_filter");
        const int m = qd_plan(c, {QD_IN(F[QD_F_U], 0), QD_IN(F[QD_F_V], 0), QD_IN(F[QD_F_H], 0)});
        if (m < 0) return -1;
        double* fl[3] = {F[QD_F_U], F[QD_F_V], F[QD_F_H]};
        if (qd_zonal_filter_fields(c, fl, 3, p.spec_cutoff, p.spec_damp, m)) return -1;
    }
    // cloud gather + decay + damp + scrub (dynamics.py:642-667)
    {
        QdScope sc(c, "final");
        const int m = qd_plan(c, {QD_IN(F[QD_F_CLOUD], R), QD_IN(F[QD_F_U], 0), QD_IN(F[QD_F_V], 0), QD_IN(F[QD_F_H], 0),
                                  QD_IN(F[QD_F_TS], 0), QD_IN(F[QD_F_Q], 0)});
        if (m < 0) return -1;
        const double decay = 1 - dt / (2.0 * 24 * 3600);
        if (c->defer_final && c->geo.full) {                  // qd_step_n: the ocean step's first launch does this (k_final_qnet_stress)
            c->final_pending.on = 1; c->final_pending.dt = dt; c->final_pending.decay = decay; c->final_pending.dfac = p.diff_factor;
            return 0;
        }
        double* oc = qd_scratch(c, 0);
        QD_ROWS(c, m, G, hipLaunchKernelGGL(k_final, qd_grid2d(G), blk, 0, c->stream, G, c->tabs.cos6, dt, p.a, c->dlat, c->dlon,
                                            F[QD_F_U], F[QD_F_V], F[QD_F_H], F[QD_F_TS], F[QD_F_Q], F[QD_F_CLOUD], oc, decay,
                                            p.diff_factor));
        qd_mark(c, {F[QD_F_U], F[QD_F_V], F[QD_F_H], F[QD_F_TS], F[QD_F_Q], oc}, m);
        qd_swap(c, QD_F_CLOUD, 0);
    }
    return 0;
}
