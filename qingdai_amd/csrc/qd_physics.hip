// qd_physics.hip -- driver-side per-step diagnostics on gfx950
// (pygcm/physics.py:12-354 as sequenced by scripts/run_simulation.py:1766-1934, 2063-2146):
//
//   k_precip_raw      divergence -> pos = max(0, -(div - D_crit)); F_div = clip(pos/median(pos>0), 0, 5);
//                     P_raw = max(0, P_cond) (1 + beta F_div); weighted sums for the renormalisation
//   k_gauss_axis x2   sigma = 1 blur of s * P_raw                       (physics.py:318-330)
//   k_pdyn + blur     legacy convergence precipitation for the fallback blend (physics.py:344-352)
//   k_precip_blend    (1 - a) P + a P_dyn when <P_cond> < PQ_MIN, clip >= 0
//   radix select      P_ref = median(precip > 0)                        (run_simulation.py:1866-1874)
//   k_cloud_from_p    C_max tanh(P / (P_ref + 1e-12)) -> blur -> clip   (physics.py:48-70)
//   k_cloud_source    tanh terms from T_s, relative vorticity, |T advection| -> blur -> clip (physics.py:72-114)
//   k_cloud_blend     memory / precipitation / source blend + event floor (run_simulation.py:1890-1913)
//   k_advect + blur   cloud tracer advection, cos floor 0.5, sigma = 0.2 wrap (run_simulation.py:1916-1934)
//   k_cloud_albedo    alpha-blend + ice fraction + dynamic albedo       (physics.py:164-250)
#include "qd_internal.h"
#include "qd_band.h"
#include "qd_pointwise.h"
#include "qd_device.h"
#include "qd_saf.h"
#include "qd_fluxes.h"
QdColP qd_make_colp(const qd_ctx* c, double dt);   // qd_atmos.hip

__device__ __forceinline__ double qd_wsum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}

// one workgroup per row: P_raw and the two weighted row sums (num = sum Pq w, den = sum P_raw w)
__global__ void __launch_bounds__(QD_BLOCK)
k_precip_raw(QdGeom G, QdTabs T, const double* __restrict__ u, const double* __restrict__ v,
             const double* __restrict__ pcond, double a, double dlat, double dlon, double D_crit, double beta,
             const double* __restrict__ scale_p, const unsigned long long* __restrict__ sel_state,
             double* __restrict__ praw, double* __restrict__ pos_out, double* __restrict__ partial,
             const double* __restrict__ orog) {
    __shared__ double sm[2][QD_BLOCK / 64];
    const int i = G.row0 + blockIdx.y;
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    const double w = T.warea[i];
    const bool anypos = sel_state[0] > 0;                    // np.any(pos > 0)
    const double scale = qd_max(*scale_p, 1e-12);
    double s_num = 0.0, s_den = 0.0;
    for (int j = threadIdx.x; j < G.nlon; j += QD_BLOCK) {
        const double div = qd_divvort_point(G, T, u, v, i, j, a, dlat, dlon, 0);
        const double pos = qd_max(0.0, -(div - D_crit));
        const double F_div = anypos ? qd_clip(pos / scale, 0.0, 5.0) : 0.0;
        const double Pq = qd_max(0.0, pcond[b + j]);
        const double F_orog = orog ? qd_clip(orog[b + j], 1.0, 3.0) : 1.0;     // physics.py:309-312
        const double F = (1.0 + beta * F_div) * F_orog;
        const double pr = Pq * F;
        praw[b + j] = pr;
        pos_out[b + j] = pos;
        s_num += Pq * w;
        s_den += pr * w;
    }
    s_num = qd_wsum(s_num); s_den = qd_wsum(s_den);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sm[0][wv] = s_num; sm[1][wv] = s_den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < QD_BLOCK / 64; ++k) { s_num += sm[0][k]; s_den += sm[1][k]; }
        partial[blockIdx.y] = s_num;
        partial[gridDim.y + blockIdx.y] = s_den;
    }
}

// compute_orographic_factor before its blur (physics.py:116-158): upslope wind on the elevation gradient
// (np.roll on both axes, pole rows of dH/dy zeroed)
__global__ void __launch_bounds__(QD_BLOCK)
k_orog_factor(QdGeom G, QdTabs T, const double* __restrict__ elev, const double* __restrict__ u, const double* __restrict__ v,
              double a, double dlat, double dlon, double k_orog, double cap, double* __restrict__ out) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row, n = G.nlat;
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    const int jp = qd_wrapc(j + 1, G.nlon), jm = qd_wrapc(j - 1, G.nlon);
    const double dx = a * T.cos6[i] * dlon, dy = a * dlat;
    const double dHdx = (elev[b + jp] - elev[b + jm]) / (2.0 * dx);
    double dHdy = 0.0;
    if (i != 0 && i != n - 1)
        dHdy = (elev[(size_t)qd_lrow(G, i + 1) * G.nlon + j] - elev[(size_t)qd_lrow(G, i - 1) * G.nlon + j]) / (2.0 * dy);
    const double gn = sqrt(dHdx * dHdx + dHdy * dHdy);
    const double nxh = gn > 1e-12 ? dHdx / (gn + 1e-12) : 0.0;
    const double nyh = gn > 1e-12 ? dHdy / (gn + 1e-12) : 0.0;
    out[b + j] = qd_clip(1.0 + k_orog * qd_max(0.0, u[b + j] * nxh + v[b + j] * nyh), 1.0, cap);
}

// s = num/den (renorm), Pq_mean = num / wsum; out[0] = s, out[1] = blend weight of the legacy field
__global__ void __launch_bounds__(QD_BLOCK)
k_precip_scalars(const double* __restrict__ partial, int n, double wsum, double pq_min, double p_blend, int use_fb,
                 double* __restrict__ out) {
    __shared__ double sm[2][QD_BLOCK / 64];
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < n; k += QD_BLOCK) { a += partial[k]; b += partial[n + k]; }
    a = qd_wsum(a); b = qd_wsum(b);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sm[0][wv] = a; sm[1][wv] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < QD_BLOCK / 64; ++k) { a += sm[0][k]; b += sm[1][k]; }
        const double den = b + 1e-20;
        out[0] = den > 0 ? a / den : 1.0;
        const double pq_mean = a / (wsum + 1e-15);
        out[1] = (use_fb && pq_mean < pq_min) ? p_blend : 0.0;
    }
}

__global__ void __launch_bounds__(QD_BLOCK)
k_scale_field(QdGeom G, const double* __restrict__ in, const double* __restrict__ s, double k, double* __restrict__ out) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    out[o] = s ? in[o] * (*s) : k * in[o];
}

__global__ void __launch_bounds__(QD_BLOCK)
k_precip_blend(QdGeom G, const double* __restrict__ P, const double* __restrict__ Pdyn, const double* __restrict__ sc,
               double* __restrict__ out) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    const double al = sc[1];
    double p = P[o];
    if (al != 0.0) p = (1.0 - al) * p + al * Pdyn[o];
    out[o] = qd_max(p, 0.0);                                  // np.clip(P, 0, None)
}

__global__ void __launch_bounds__(QD_BLOCK)
k_cloud_from_p(QdGeom G, const double* __restrict__ precip, const double* __restrict__ pref, double cmax,
               double* __restrict__ out) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    out[o] = cmax * qd_tanh(precip[o] / (*pref + 1e-12));
}


// np.clip(tanh(x), 0, 1) without the tanh where it cannot matter: tanh(x) <= 0 for x <= 0, so those lanes take 0 -- and a wavefront in
// which every lane does (cold rows, anticyclonic regions) skips the ~45 instructions of the f64 tanh altogether.  Same result bit for
// bit: for every other x, NaN included, the same tanh is evaluated -- np.clip(np.tanh(nan), 0, 1) is nan (physics.py:85-107), and a
// blown-up T_s / u / v has to reach the cloud source like it does in the reference (qd_tanh_nonneg and qd_clip propagate NaN).
__device__ __forceinline__ double qd_tanh01(double x) {
    double t = 0.0;
    if (!(x <= 0.0)) t = qd_tanh_nonneg(x);
    return qd_clip(t, 0.0, 1.0);
}

// physics.py:72-109 before the blur (one cell)
__device__ __forceinline__ double qd_cloud_source_cell(const QdGeom& G, const QdTabs& T, const double* __restrict__ u, const double* __restrict__ v,
                                                       const double* __restrict__ Ts, double a, double dlat, double dlon, int i, int j);

__global__ void __launch_bounds__(QD_BLOCK)
k_cloud_source(QdGeom G, QdTabs T, const double* __restrict__ u, const double* __restrict__ v,
               const double* __restrict__ Ts, double a, double dlat, double dlon, double* __restrict__ out) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    out[(size_t)qd_lrow(G, i) * G.nlon + j] = qd_cloud_source_cell(G, T, u, v, Ts, a, dlat, dlon, i, j);
}

// whole-globe handles: C_from_P (k_cloud_from_p) and the cloud source of the same cell in one launch
__global__ void __launch_bounds__(QD_BLOCK)
k_cloud_fromp_source(QdGeom G, QdTabs T, const double* __restrict__ precip, const double* __restrict__ pref, double cmax,
                     const double* __restrict__ u, const double* __restrict__ v, const double* __restrict__ Ts, double a, double dlat,
                     double dlon, double* __restrict__ cfp, double* __restrict__ src) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
    cfp[o] = cmax * qd_tanh(precip[o] / (*pref + 1e-12));
    src[o] = qd_cloud_source_cell(G, T, u, v, Ts, a, dlat, dlon, i, j);
}

__device__ __forceinline__ double qd_cloud_source_cell(const QdGeom& G, const QdTabs& T, const double* __restrict__ u, const double* __restrict__ v,
                                                       const double* __restrict__ Ts, double a, double dlat, double dlon, int i, int j) {
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    const size_t o = b + j;
    const double T0 = Ts[o];
    double src = 0.0;
    src = src + 0.5 * qd_tanh01((T0 - 285.0) / 12.0);
    const double vort = qd_divvort_point(G, T, v, u, i, j, a, dlat, dlon, 1);
    const double rel = vort / (T.fcor[i] + 1e-12);
    src = src + 0.4 * qd_tanh01((rel - 0.5) / 2.0);
    const double dx = dlon * a * T.cos6[i];
    const double dy = dlat * a;
    const int jp = qd_wrapc(j + 1, G.nlon), jm = qd_wrapc(j - 1, G.nlon);
    const double gx = (Ts[b + jp] - Ts[b + jm]) / (2 * dx);
    const double gy = (Ts[(size_t)qd_lrow(G, i + 1) * G.nlon + j] - Ts[(size_t)qd_lrow(G, i - 1) * G.nlon + j]) / (2 * dy);
    const double tadv = -(u[o] * gx + v[o] * gy);
    src = src + 0.3 * qd_tanh01(fabs(tadv) / 2e-5);
    return src;
}

struct QdBlendP { double w_mem, w_p, w_src, tend, c_floor; };

__global__ void __launch_bounds__(QD_BLOCK)
k_cloud_blend(QdGeom G, QdBlendP P, const double* __restrict__ cfp, const double* __restrict__ src,
              double* __restrict__ cloud) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    const double c0 = cloud[o], cp = cfp[o];
    const double tendency = src[o] * P.tend;
    double c = (P.w_mem * c0 + P.w_p * cp + P.w_src * qd_clip(c0 + tendency, 0.0, 1.0));
    if (P.c_floor > 0.0) c = qd_max(c, qd_clip(P.c_floor * cp, 0.0, 1.0));
    cloud[o] = qd_clip(c, 0.0, 1.0);
}

// cloud <- clip((1-a) cloud + a adv, 0, 1)  and  the dynamic albedo (physics.py:164-250); body: qd_pointwise.h
__global__ void __launch_bounds__(QD_BLOCK)
k_cloud_albedo(QdGeom G, QdAlbP P, const double* __restrict__ adv, double* __restrict__ cloud,
               const double* __restrict__ cloud_eff, const double* __restrict__ hice,
               const double* __restrict__ base, const uint8_t* __restrict__ land, const double* __restrict__ csnow,
               const double* __restrict__ eco_alpha, const double* __restrict__ glacier, const double* __restrict__ banded,
               const double* __restrict__ water, double* __restrict__ albedo) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    double c = cloud[o];
    if (P.do_adv) { c = qd_clip((1.0 - P.alpha) * c + P.alpha * adv[o], 0.0, 1.0); cloud[o] = c; }
    const int landv = land[o];
    const double cs = (P.snow && landv == 1) ? csnow[o] : 0.0;
    const double gl = (P.eco && landv == 1) ? glacier[o] : 0.0;
    albedo[o] = qd_albedo_cell(P, o, c, cloud_eff, hice, base, landv, cs, gl, eco_alpha, banded, water);
}

__global__ void k_precip_scalars_post(const double* raw, double wsum, double pq_min, double p_blend, int use_fb, double* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double a = raw[0], b = raw[1];
        const double den = b + 1e-20;
        out[0] = den > 0 ? a / den : 1.0;
        const double pq_mean = a / (wsum + 1e-15);
        out[1] = (use_fb && pq_mean < pq_min) ? p_blend : 0.0;
    }
}
// raw (num, den) sums of the owned rows -> out[0], out[1]
__global__ void __launch_bounds__(QD_BLOCK)
k_precip_rawsums(const double* __restrict__ partial, int n, double* __restrict__ out) {
    __shared__ double sm[2][QD_BLOCK / 64];
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < n; k += QD_BLOCK) { a += partial[k]; b += partial[n + k]; }
    a = qd_wsum(a); b = qd_wsum(b);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sm[0][wv] = a; sm[1][wv] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < QD_BLOCK / 64; ++k) { a += sm[0][k]; b += sm[1][k]; }
        out[0] = a; out[1] = b;
    }
}

// ------------------------------------------------------------------ P019 lapse + snow, land bucket
// run_simulation.py:1946-2019 + hydrology.py:100-177: lapse-adjusted air temperature, sigmoid rain/snow
// split, provisional snowpack (degree-day | constant melt), optical snow cover, glacier mask; body: qd_pointwise.h
__global__ void __launch_bounds__(QD_BLOCK)
k_snow_provisional(QdGeom G, QdTabs T, QdSnowP P, const double* __restrict__ precip, const double* __restrict__ h,
                   const double* __restrict__ S_snow, const double* __restrict__ elev, const uint8_t* __restrict__ land,
                   double* __restrict__ P_rain, double* __restrict__ S_next, double* __restrict__ melt,
                   double* __restrict__ C_snow, double* __restrict__ glacier) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
    const QdSnowOut r = qd_snow_cell(T, P, i, land[o] == 1, h[o], S_snow[o], elev[o], precip[o]);
    P_rain[o] = r.Pr; S_next[o] = r.Sn; melt[o] = r.melt; C_snow[o] = r.Cs; glacier[o] = r.gl;
}

// The tail of the driver physics and the forcing of the same step in ONE launch (whole-globe handles inside qd_step_n): snowpack ->
// cloud tracer blend + albedo -> two-star insolation + Teq are three pointwise kernels on the same cells (17 + 13 + 12 us as three
// launches at 721 x 1440; each re-reads what its predecessor has just written: C_snow, the glacier mask, the albedo).  Same bodies,
// same order, results handed on in registers.
__global__ void __launch_bounds__(QD_BLOCK)
k_snow_albedo_forcing(QdSafArgs K) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= K.G.nlon) return;
    const int i = K.G.row0 + tl.row;
    qd_saf_cell(K, i, j, (size_t)qd_lrow(K.G, i) * K.G.nlon + j, true);
}
// a launch that qd_step_n left for time_step's column kernel and that did not happen there after all (qd_atmos.hip)
int qd_saf_launch(qd_ctx* c, const QdSafArgs& K) {
    hipLaunchKernelGGL(k_snow_albedo_forcing, qd_grid2d(K.G), dim3(QD_BLOCK), 0, c->stream, K);
    return 0;
}
void qd_saf_drop(qd_ctx* c) { delete (QdSafArgs*)c->saf_pending; c->saf_pending = nullptr; }
int qd_saf_flush(qd_ctx* c) {
    if (!c->saf_pending) return 0;
    QdSafArgs* K = (QdSafArgs*)c->saf_pending;
    c->saf_pending = nullptr;
    const int r = qd_saf_launch(c, *K);
    delete K;
    return r;
}

struct QdBucketP { double dt, tau_s, cap; };

// run_simulation.py:2290-2339 + hydrology.py:219-260
__global__ void __launch_bounds__(QD_BLOCK)
k_hydro_commit(QdGeom G, QdBucketP P, const double* __restrict__ P_rain, const double* __restrict__ melt,
               const double* __restrict__ glacier, const double* __restrict__ E, const uint8_t* __restrict__ land,
               const double* __restrict__ S_next, double* __restrict__ S_snow, double* __restrict__ W_land,
               double* __restrict__ runoff) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    const double L = land[o] == 1 ? 1.0 : 0.0;
    const double gl = glacier[o];
    const double non_gl = (L != 0.0 && gl == 0.0) ? 1.0 : 0.0;
    const double ml = melt[o];
    const double P_in = (P_rain[o] * L + ml) * non_gl;
    const double E_l = (E[o] * L) * non_gl;
    const double W = W_land[o];
    const double R_base = W / P.tau_s;
    double Wn = qd_max(0.0, W + (P_in - E_l - R_base) * P.dt);
    double R_fast = 0.0;
    if (P.cap > 0.0) {
        const double over = qd_max(0.0, Wn - P.cap);
        Wn = Wn - over;
        R_fast = (P.dt > 0) ? over / P.dt : 0.0;
    }
    W_land[o] = qd_nn(Wn);
    runoff[o] = qd_nn(R_base + R_fast) + ml * gl;
    S_snow[o] = S_next[o];
}

int qd_hydrology_commit_impl(qd_ctx* c, double dt) {
    const qd_params& p = c->p;
    double** F = c->f;
    QdScope sc(c, "hydrology");
    const int m = qd_plan(c, {QD_IN(F[QD_F_P_RAIN], 0), QD_IN(F[QD_F_MELT], 0), QD_IN(F[QD_F_GLACIER], 0), QD_IN(F[QD_F_EFLUX], 0),
                              QD_IN(F[QD_F_S_SNOW_NEXT], 0), QD_IN(F[QD_F_W_LAND], 0)});
    if (m < 0) return -1;
    QdBucketP B{dt, std::max(1.0, p.runoff_tau_days * 86400.0), (p.wland_cap_mm == p.wland_cap_mm && p.wland_cap_mm > 0) ? p.wland_cap_mm : -1.0};
    QD_ROWS(c, m, G, hipLaunchKernelGGL(k_hydro_commit, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, B, F[QD_F_P_RAIN], F[QD_F_MELT],
                                        F[QD_F_GLACIER], F[QD_F_EFLUX], c->land, F[QD_F_S_SNOW_NEXT], F[QD_F_S_SNOW], F[QD_F_W_LAND],
                                        F[QD_F_RUNOFF]));
    qd_mark(c, {F[QD_F_S_SNOW], F[QD_F_W_LAND], F[QD_F_RUNOFF]}, m);
    return 0;
}

int qd_driver_physics_impl(qd_ctx* c, double dt, const QdForcingCall* fc, int part) {
    const qd_params& p = c->p;
    const QdGeom& G0 = c->geo;
    const dim3 blk(QD_BLOCK);
    double** F = c->f;
    const bool band = !G0.full;
    const QdGeom Gown = qd_segments(c, 0).g[0];
    auto isset = [](double x) { return !(x != x); };
    double*& praw = c->scratch[4];
    double*& pos = c->scratch[5];
    double*& tmp = c->scratch[6];
    double*& pdyn = c->scratch[7];
    const int R1 = qd_gauss_radius(1.0);
    const int Rc = qd_adv_reach(c, dt, 250.0);
    // the block's row partials: its own buffer when it may run beside the ocean step, which owns red_partial (QD_SIDE_STREAM)
    double* const rp = (c->side_stream_on && c->red_partial_b) ? c->red_partial_b : c->red_partial;
    if (part == 2 && qd_side_join(c)) return -1;              // the block ran on the side stream: its products are read from here on
    // QD_MED_SIDE: time_step's P_cond + its median start here, on the side stream, beside everything below (qd_atmos.hip)
    const bool pcond_side = part != 1 && c->want_pcond_ahead && c->med_side && c->side_stream && c->geo.full && p.cloud_couple && !isset(p.pcond_ref) &&
                            c->timing != 1;
    if (pcond_side && qd_pcond_median_side(c, dt)) return -1;
    const bool pcond_pair = part != 1 && !pcond_side && c->want_pcond_ahead && c->med_pair && c->hist_b && c->geo.full && p.cloud_couple &&
                            !isset(p.pcond_ref) && !(isset(p.pref) && p.pref != 0.0);
    if (part != 2) {
        QdScope sc(c, "phys_precip");
        // median of pos = max(0, -(div - D_crit)) over pos > 0, straight from the divergence field
        int m = qd_plan(c, {QD_IN(F[QD_F_U], 1), QD_IN(F[QD_F_V], 1), QD_IN(F[QD_F_PCOND], 0)});
        if (m < 0) return -1;
        qd_launch_divvort(c, F[QD_F_U], F[QD_F_V], tmp, 0, m);
        if (qd_median_positive_dev(c, tmp, 1e-12, QD_S_PSCALE, 1, p.D_crit, 2)) return -1;
        // orographic enhancement (run_simulation.py:1769-1775): only with QD_OROG=1 and an elevation map
        const double* orog = nullptr;
        if (p.orog_enable && c->has_elevation) {
            double*& of = c->scratch[8];
            const int mo = qd_plan(c, {QD_IN(F[QD_F_U], 0), QD_IN(F[QD_F_V], 0), QD_IN(F[QD_F_ELEVATION], 1)});
            if (mo < 0) return -1;
            QD_ROWS(c, mo, G, hipLaunchKernelGGL(k_orog_factor, qd_grid2d(G), blk, 0, c->stream, G, c->tabs, F[QD_F_ELEVATION],
                                                 F[QD_F_U], F[QD_F_V], p.a, c->dlat, c->dlon, p.orog_k, 2.0, of));
            qd_mark(c, {of}, mo);
            const int mb = qd_plan(c, {QD_IN(of, R1)});
            if (mb < 0) return -1;
            if (qd_gaussian_swap(c, of, tmp, 1.0, 0, mb)) return -1;
            orog = of;
            m = qd_plan(c, {QD_IN(F[QD_F_U], 1), QD_IN(F[QD_F_V], 1), QD_IN(F[QD_F_PCOND], 0), QD_IN(c->scratch[8], 0)});
            if (m < 0) return -1;
        }
        // P_raw / pos on the margin; the two weighted sums over owned rows only
        if (band && m > 0) {
            QdSegs S = qd_segments(c, m);
            const int own0 = c->own_row0, own1 = c->own_row0 + c->own_nrows;
            for (int k = 0; k < S.n; ++k) {
                const QdGeom& G = S.g[k];
                auto run = [&](int r0, int r1) {
                    if (r1 <= r0) return;
                    QdGeom g2 = G; g2.row0 = r0; g2.nrows = r1 - r0;
                    hipLaunchKernelGGL(k_precip_raw, dim3(1, g2.nrows), blk, 0, c->stream, g2, c->tabs, F[QD_F_U], F[QD_F_V],
                                       F[QD_F_PCOND], p.a, c->dlat, c->dlon, p.D_crit, p.p_betadiv, c->dscal + QD_S_PSCALE,
                                       c->dcount, praw, pos, rp + (size_t)2 * G0.lrows(), orog);
                };
                const int a0 = G.row0, a1 = G.row0 + G.nrows;
                if (a1 <= own0 || a0 >= own1) run(a0, a1);
                else { run(a0, std::max(a0, own0)); run(std::min(a1, own1), a1); }
            }
        }
        hipLaunchKernelGGL(k_precip_raw, dim3(1, Gown.nrows), blk, 0, c->stream, Gown, c->tabs, F[QD_F_U], F[QD_F_V],
                           F[QD_F_PCOND], p.a, c->dlat, c->dlon, p.D_crit, p.p_betadiv, c->dscal + QD_S_PSCALE, c->dcount, praw,
                           pos, rp, orog);
        qd_mark(c, {praw, pos}, m);
        if (band && qd_peer_hooks(c)) {
            // row sums -> all-reduce -> the two scalars in ONE launch (QdPeerHook: pre 2 = k_precip_rawsums, post 1 = k_precip_scalars_post)
            QdPeerHook H; H.pre = 2; H.partial = rp; H.n = Gown.nrows;
            H.post = 1; H.wsum = c->wsum_all; H.pq_min = p.pq_min; H.p_blend = p.p_blend; H.use_fb = p.p_hybrid_fallback; H.out = c->dscal + QD_S_RENORM;
            c->allreduces++;
            if (qd_peer_allreduce_hooked(c, c->dscal + QD_S_TMP0, 2, 0, H)) return -1;
        } else if (band) {
            hipLaunchKernelGGL(k_precip_rawsums, dim3(1), blk, 0, c->stream, rp, Gown.nrows, c->dscal + QD_S_TMP0);
            if (qd_allreduce_f64(c, c->dscal + QD_S_TMP0, 2, 0)) return -1;
            hipLaunchKernelGGL(k_precip_scalars_post, dim3(1), dim3(64), 0, c->stream, c->dscal + QD_S_TMP0, c->wsum_all, p.pq_min,
                               p.p_blend, p.p_hybrid_fallback, c->dscal + QD_S_RENORM);
        } else {
            hipLaunchKernelGGL(k_precip_scalars, dim3(1), blk, 0, c->stream, rp, Gown.nrows, c->wsum_all, p.pq_min,
                               p.p_blend, p.p_hybrid_fallback, c->dscal + QD_S_RENORM);
        }
        // P = gaussian(P_raw * s); P_dyn = gaussian(k_precip * pos): the two scalings ride on the blur's loads
        if (qd_gauss_pair_ok(c, 1.0)) {
            // both blurs and the blend in one launch (the convergence field is not even read while the fallback is off)
            const int mg = qd_plan(c, {QD_IN(praw, R1), QD_IN(pos, R1)});
            if (mg < 0) return -1;
            if (qd_gaussian_pair(c, praw, pos, 1.0, 0, c->dscal + QD_S_RENORM, 1.0, p.k_precip, c->dscal + QD_S_RENORM, 1, nullptr,
                                 nullptr, nullptr, F[QD_F_PRECIP], mg)) return -1;
            qd_mark(c, {F[QD_F_PRECIP]}, mg);
        } else if (qd_gauss_can_fuse(c, 1.0)) {
            const int mg = qd_plan(c, {QD_IN(praw, R1), QD_IN(pos, R1)});
            if (mg < 0) return -1;
            if (qd_gaussian_swap(c, praw, tmp, 1.0, 0, mg, 0, c->dscal + QD_S_RENORM)) return -1;
            if (qd_gaussian(c, pos, pdyn, nullptr, 1.0, 0, mg, 0, nullptr, p.k_precip)) return -1;
            QD_ROWS(c, mg, G, hipLaunchKernelGGL(k_precip_blend, qd_grid2d(G), blk, 0, c->stream, G, praw, pdyn,
                                                 c->dscal + QD_S_RENORM, F[QD_F_PRECIP]));
            qd_mark(c, {F[QD_F_PRECIP]}, mg);
        } else {
        QD_ROWS(c, m, G, hipLaunchKernelGGL(k_scale_field, qd_grid2d(G), blk, 0, c->stream, G, praw, c->dscal + QD_S_RENORM, 0.0, praw));
        QD_ROWS(c, m, G, hipLaunchKernelGGL(k_scale_field, qd_grid2d(G), blk, 0, c->stream, G, pos, (const double*)nullptr,
                                            p.k_precip, pdyn));
        qd_mark(c, {pdyn}, m);
        const int mg = qd_plan(c, {QD_IN(praw, R1), QD_IN(pdyn, R1)});
        if (mg < 0) return -1;
        if (qd_gaussian_swap(c, praw, tmp, 1.0, 0, mg)) return -1;
        if (qd_gaussian_swap(c, pdyn, tmp, 1.0, 0, mg)) return -1;
        QD_ROWS(c, mg, G, hipLaunchKernelGGL(k_precip_blend, qd_grid2d(G), blk, 0, c->stream, G, praw, pdyn,
                                             c->dscal + QD_S_RENORM, F[QD_F_PRECIP]));
        qd_mark(c, {F[QD_F_PRECIP]}, mg);
        }
    }
    if (part == 1) return 0;
    {
        QdScope sc(c, "phys_cloud");
        if (isset(p.pref) && p.pref != 0.0) {
            hipMemcpyAsync(c->dscal + QD_S_MED_OUT, &p.pref, sizeof(double), hipMemcpyHostToDevice, c->stream);
        } else if (pcond_pair) {
            // time_step's P_cond (phase 1 of its column) is computed HERE, so that its median and the precipitation median go through
            // ONE chain of three launches (qd_median_pair_dev); time_step then starts with the column's phase 2
            if (c->med_fold && qd_median_pair_ready(c, 3, 1)) {
                // ... and P_cond itself comes out of the pair's histogram pass (k_med_hist2p): no k_column<1> launch
                if (qd_median_pair_pcond_dev(c, F[QD_F_PRECIP], 1e-6, QD_S_MED_OUT, 0, 0.0, 3, qd_make_colp(c, dt), 1e-6, QD_S_PREF, 1)) return -1;
            } else {
                if (qd_pcond_phase1(c, dt)) return -1;
                if (qd_median_pair_dev(c, F[QD_F_PRECIP], 1e-6, QD_S_MED_OUT, 0, 0.0, 3, F[QD_F_PCOND], 1e-6, QD_S_PREF, 0, 0.0, 1)) return -1;
            }
            c->pcond_ahead = 3;
        } else {
            if (qd_median_positive_dev(c, F[QD_F_PRECIP], 1e-6, QD_S_MED_OUT, 0, 0.0, 3)) return -1;
        }
        double*& cfp = F[QD_F_CLOUD_FROM_P];
        double*& src = F[QD_F_CLOUD_SRC];
        int m = qd_plan(c, {QD_IN(F[QD_F_PRECIP], 0)});
        if (m < 0) return -1;
        int ms = qd_plan(c, {QD_IN(F[QD_F_U], 1), QD_IN(F[QD_F_V], 1), QD_IN(F[QD_F_TS], 1)});
        if (ms < 0) return -1;
        if (c->merge_pointwise) {
            // one launch for both fields (bands: on the rows both can be computed on -- the blur below needs both on the same rows anyway)
            m = ms = std::min(m, ms);
            QD_ROWS(c, m, G, hipLaunchKernelGGL(k_cloud_fromp_source, qd_grid2d(G), blk, 0, c->stream, G, c->tabs, F[QD_F_PRECIP],
                                                c->dscal + QD_S_MED_OUT, p.cmax, F[QD_F_U], F[QD_F_V], F[QD_F_TS], p.a, c->dlat, c->dlon, cfp, src));
        } else {
        QD_ROWS(c, m, G, hipLaunchKernelGGL(k_cloud_from_p, qd_grid2d(G), blk, 0, c->stream, G, F[QD_F_PRECIP],
                                            c->dscal + QD_S_MED_OUT, p.cmax, cfp));
        QD_ROWS(c, ms, G, hipLaunchKernelGGL(k_cloud_source, qd_grid2d(G), blk, 0, c->stream, G, c->tabs, F[QD_F_U], F[QD_F_V],
                                             F[QD_F_TS], p.a, c->dlat, c->dlon, src));
        }
        qd_mark(c, {cfp}, m);
        qd_mark(c, {src}, ms);
        const int mg = qd_plan(c, {QD_IN(cfp, R1), QD_IN(src, R1), QD_IN(F[QD_F_CLOUD], 0)});
        if (mg < 0) return -1;
        double wm = p.w_mem, wp = p.w_p, ws = p.w_src, wsum = wm + wp + ws;
        if (wsum <= 0) { wm = 0.5; wp = 0.4; ws = 0.1; wsum = 1.0; }
        wm /= wsum; wp /= wsum; ws /= wsum;
        QdBlendP B{wm, wp, ws, dt / (6 * 3600), p.cloud_from_p_floor};
        if (qd_gauss_pair_ok(c, 1.0)) {
            // blur + clip of both fields and the blend in one launch; the blurred fields land in two scratch slabs that then trade
            // places with the inputs (like qd_gaussian_swap)
            double*& oa = c->scratch[6]; double*& ob = c->scratch[7];
            const double b5[5] = {B.w_mem, B.w_p, B.w_src, B.tend, B.c_floor};
            if (qd_gaussian_pair(c, cfp, src, 1.0, 0, nullptr, 1.0, 1.0, nullptr, 2, b5, oa, ob, F[QD_F_CLOUD], mg)) return -1;
            std::swap(cfp, oa); std::swap(src, ob);
            qd_mark(c, {cfp, src, F[QD_F_CLOUD]}, mg);
        } else {
        if (qd_gaussian_swap(c, cfp, tmp, 1.0, 0, mg, 1)) return -1;      // np.clip(gaussian(...), 0, 1)
        if (qd_gaussian_swap(c, src, tmp, 1.0, 0, mg, 1)) return -1;
        QD_ROWS(c, mg, G, hipLaunchKernelGGL(k_cloud_blend, qd_grid2d(G), blk, 0, c->stream, G, B, cfp, src, F[QD_F_CLOUD]));
        qd_mark(c, {F[QD_F_CLOUD]}, mg);
        }
    }
    {
        QdScope sc(c, "phys_albedo");
        double*& adv = c->scratch[7];
        int m;
        if (p.cloud_advect) {
            const int ma = qd_plan(c, {QD_IN(F[QD_F_CLOUD], Rc), QD_IN(F[QD_F_U], 0), QD_IN(F[QD_F_V], 0)});
            if (ma < 0) return -1;
            qd_launch_advect(c, F[QD_F_U], F[QD_F_V], c->tabs.cos05, dt, F[QD_F_CLOUD], adv, nullptr, nullptr, 1.0, 0, ma);
            if (p.cloud_smooth_sigma > 0.0) {
                const int rs = qd_gauss_radius(p.cloud_smooth_sigma);
                const int mb = qd_plan(c, {QD_IN(adv, rs)});
                if (mb < 0) return -1;
                if (qd_gaussian_swap(c, adv, tmp, p.cloud_smooth_sigma, 1, mb)) return -1;
            }
            m = qd_plan(c, {QD_IN(adv, 0), QD_IN(F[QD_F_CLOUD], 0), QD_IN(F[QD_F_HICE], 0)});
        } else {
            m = qd_plan(c, {QD_IN(F[QD_F_CLOUD], 0), QD_IN(F[QD_F_HICE], 0)});
        }
        if (m < 0) return -1;
        if (c->cloud_eff_valid) { const int me = qd_plan(c, {QD_IN(F[QD_F_CLOUD_EFF], 0)}); if (me < 0) return -1; m = std::min(m, me); }
        const QdEco& E = c->eco;
        QdAlbP A{p.cloud_adv_alpha, std::max(1e-6, p.hice_ref), p.alpha_ice, p.alpha_cloud, p.alpha_water, p.snow_albedo_fresh, E.p.w_lai,
                 p.cloud_advect ? 1 : 0, p.use_topo_albedo ? 1 : 0, p.swe_enable ? 1 : 0,
                 (E.configured && E.p.albedo_couple && E.alpha_valid) ? 1 : 0, (E.configured && E.p.bands_couple && E.banded_valid) ? 1 : 0,
                 (E.configured && E.p.water_couple && E.water_valid) ? 1 : 0, (E.configured && E.p.map_f32) ? 1 : 0};
        if (A.eco) { const int me = qd_plan(c, {QD_IN(F[QD_F_ECO_ALPHA], 0)}); if (me < 0) return -1; m = std::min(m, me); }
        if (A.banded) { const int me = qd_plan(c, {QD_IN(F[QD_F_ECO_ALPHA_BANDED], 0)}); if (me < 0) return -1; m = std::min(m, me); }
        if (A.water) { const int me = qd_plan(c, {QD_IN(F[QD_F_WATER_ALPHA], 0)}); if (me < 0) return -1; m = std::min(m, me); }
        // P019 lapse + phase split + provisional snowpack (run_simulation.py:1946-2019): pointwise
        {
            const int msn = qd_plan(c, {QD_IN(F[QD_F_PRECIP], 0), QD_IN(F[QD_F_H], 0), QD_IN(F[QD_F_S_SNOW], 0), QD_IN(F[QD_F_ELEVATION], 0)});
            if (msn < 0) return -1;
            QdSnowP S;
            S.dt = dt; S.ga = 9.81 / 1004.0; S.rho_snow_safe = std::max(p.rho_snow, 1e-6); S.polar_lat = p.polar_lat_thresh;
            S.ice_max = p.polar_ice_thick_max_m; S.elev_max = p.land_elev_max_m; S.gamma = p.lapse_k_kpm; S.lapse = p.lapse_enable;
            S.t_thresh = p.snow_thresh_K; S.dT = std::max(1e-6, p.snow_t_band_K); S.mode = p.snow_melt_mode;
            S.ddf_s = p.snow_ddf_mm_per_k_day / 86400.0; S.tref = p.snow_melt_tref_K; S.rate_s = p.snow_melt_rate_mm_day / 86400.0;
            S.swe_max = (p.swe_max_mm == p.swe_max_mm && p.swe_max_mm > 0) ? p.swe_max_mm : -1.0;
            S.swe_ref_safe = std::max(1e-6, p.swe_ref_mm); S.gl_frac = p.glacier_frac; S.gl_swe = p.glacier_swe_mm; S.swe = p.swe_enable;
            // the forcing of this step rides on the same launch when the caller (qd_step_n, whole-globe handle) hands it over
            if (fc) {
                const QdForcingP Fo{QdStar{fc->sa[0], std::sin(fc->sa[1]), std::cos(fc->sa[1]), fc->sa[2]},
                                    QdStar{fc->sb[0], std::sin(fc->sb[1]), std::cos(fc->sb[1]), fc->sb[2]}, fc->theta, 5.670374e-8, 1};
                // (bands: on the rows every input is valid on -- the margin the three separate launches end up with as well, since
                //  the albedo needs the snow cover and the forcing the albedo)
                const int mm = std::min(m, msn);
                QdSafArgs K;
                K.T = c->tabs; K.S = S; K.A = A; K.Fo = Fo;
                K.precip = F[QD_F_PRECIP]; K.h = F[QD_F_H]; K.S_snow = F[QD_F_S_SNOW]; K.elev = F[QD_F_ELEVATION]; K.land = c->land;
                K.P_rain = F[QD_F_P_RAIN]; K.S_next = F[QD_F_S_SNOW_NEXT]; K.melt = F[QD_F_MELT]; K.C_snow = F[QD_F_C_SNOW]; K.glacier = F[QD_F_GLACIER];
                K.adv = adv; K.cloud = F[QD_F_CLOUD]; K.cloud_eff = c->cloud_eff_valid ? F[QD_F_CLOUD_EFF] : (const double*)nullptr;
                K.hice = F[QD_F_HICE]; K.base = F[QD_F_BASE_ALBEDO]; K.eco_alpha = F[QD_F_ECO_ALPHA]; K.banded = F[QD_F_ECO_ALPHA_BANDED];
                K.water = F[QD_F_WATER_ALPHA]; K.albedo = F[QD_F_ALBEDO]; K.isrA = F[QD_F_ISR_A]; K.isrB = F[QD_F_ISR_B]; K.isr = F[QD_F_ISR];
                K.Teq = F[QD_F_TEQ]; K.eday = c->eco.eday_dt > 0 ? F[QD_F_ECO_EDAY] : (double*)nullptr; K.eday_dt = c->eco.eday_dt;
                K.write_diag = c->diag_write;
                // time_step's column phase 1 rides along when qd_step_n says the next thing is time_step with the P_cond median
                K.col1 = (c->want_pcond_ahead && !pcond_side && !pcond_pair && c->geo.full && p.cloud_couple && !isset(p.pcond_ref)) ? 1 : 0;
                K.P = qd_make_colp(c, dt);
                K.u = F[QD_F_U]; K.v = F[QD_F_V]; K.Ts = F[QD_F_TS]; K.q = F[QD_F_Q]; K.Pcond = F[QD_F_PCOND];
                if (pcond_pair && c->merge_saf && c->pcond_ahead == 3 && c->geo.full) {
                    // the P_cond median is done already, so NOTHING separates this launch from time_step's column kernel: left for it
                    // (k_saf_column2, qd_atmos.hip: cloud, albedo, isr and Teq reach the column in registers)
                    K.G = c->geo;
                    delete (QdSafArgs*)c->saf_pending;
                    c->saf_pending = new QdSafArgs(K);
                } else
                QD_ROWS(c, mm, G, K.G = G; hipLaunchKernelGGL(k_snow_albedo_forcing, qd_grid2d(G), blk, 0, c->stream, K));
                c->eco.eday_dt = 0;
                if (K.col1) c->pcond_ahead = 1;
                qd_mark(c, {F[QD_F_P_RAIN], F[QD_F_S_SNOW_NEXT], F[QD_F_MELT], F[QD_F_C_SNOW], F[QD_F_GLACIER]}, mm);
                qd_mark(c, {F[QD_F_CLOUD], F[QD_F_ALBEDO], F[QD_F_ISR_A], F[QD_F_ISR_B], F[QD_F_ISR], F[QD_F_TEQ]}, mm);
                return 0;
            }
            QD_ROWS(c, msn, G, hipLaunchKernelGGL(k_snow_provisional, qd_grid2d(G), blk, 0, c->stream, G, c->tabs, S, F[QD_F_PRECIP],
                                                  F[QD_F_H], F[QD_F_S_SNOW], F[QD_F_ELEVATION], c->land, F[QD_F_P_RAIN],
                                                  F[QD_F_S_SNOW_NEXT], F[QD_F_MELT], F[QD_F_C_SNOW], F[QD_F_GLACIER]));
            qd_mark(c, {F[QD_F_P_RAIN], F[QD_F_S_SNOW_NEXT], F[QD_F_MELT], F[QD_F_C_SNOW], F[QD_F_GLACIER]}, msn);
            m = std::min(m, msn);
        }
        QD_ROWS(c, m, G, hipLaunchKernelGGL(k_cloud_albedo, qd_grid2d(G), blk, 0, c->stream, G, A, adv, F[QD_F_CLOUD],
                                            c->cloud_eff_valid ? F[QD_F_CLOUD_EFF] : (const double*)nullptr, F[QD_F_HICE],
                                            F[QD_F_BASE_ALBEDO], c->land, F[QD_F_C_SNOW], F[QD_F_ECO_ALPHA], F[QD_F_GLACIER],
                                            F[QD_F_ECO_ALPHA_BANDED], F[QD_F_WATER_ALPHA], F[QD_F_ALBEDO]));
        qd_mark(c, {F[QD_F_CLOUD], F[QD_F_ALBEDO]}, m);
    }
    return 0;
}


// ------------------------------------------------------------------ ecology spectral sub-step, stage 1
// dual_star_insolation_to_bands (pygcm/ecology/spectral.py:397-426): per cell S_b = (specA_b insA + specB_b insB) T_ray_b,
// S_sum = sum_b S_b (in band order, like np.sum(axis=0)), I_b = S_b / S_sum * (insA + insB) where S_sum > 1e-12 and
// insA + insB > 1e-12, else 0; non-finite -> 0.  One thread per cell, bands in registers (NB <= QD_MAXBANDS).
struct QdBandW { double a[QD_MAXBANDS], b[QD_MAXBANDS], t[QD_MAXBANDS]; int nb; };
__global__ void __launch_bounds__(QD_BLOCK)
k_band_insolation(QdGeom G, QdBandW W, const double* __restrict__ insA, const double* __restrict__ insB, double* __restrict__ out,
                  size_t plane) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    const double A = insA[o], B = insB[o];
    const double tot = A + B;
    double S[QD_MAXBANDS];
    double sum = 0.0;
#pragma unroll
    for (int b = 0; b < QD_MAXBANDS; ++b) {
        if (b < W.nb) { S[b] = (W.a[b] * A + W.b[b] * B) * W.t[b]; sum += S[b]; }
    }
    const bool pos = (sum > 1e-12) && (tot > 1e-12);
#pragma unroll
    for (int b = 0; b < QD_MAXBANDS; ++b) {
        if (b < W.nb) {
            double v = pos ? (S[b] / sum) * tot : 0.0;
            if (!(fabs(v) <= DBL_MAX)) v = 0.0;               // nan_to_num(nan=0, posinf=0, neginf=0)
            out[(size_t)b * plane + o] = v;
        }
    }
}

int qd_band_insolation_impl(qd_ctx* c, int nb, const double* specA, const double* specB, const double* tray, double* out_host) {
    if (nb < 1 || nb > QD_MAXBANDS) return qd_fail(c, "qd_band_insolation: 1 <= nb <= 32");
    const size_t plane = c->geo.cells();
    if (c->bands_nb < nb) {
        if (c->bands) hipFree(c->bands);
        c->bands = nullptr; c->bands_nb = 0;
        if (hipMalloc(&c->bands, (size_t)nb * plane * sizeof(double)) != hipSuccess) return qd_fail(c, "hipMalloc band planes");
        c->bands_nb = nb;
    }
    QdBandW W; W.nb = nb;
    for (int b = 0; b < QD_MAXBANDS; ++b) { W.a[b] = b < nb ? specA[b] : 0.0; W.b[b] = b < nb ? specB[b] : 0.0; W.t[b] = b < nb ? tray[b] : 0.0; }
    double** F = c->f;
    const int m = qd_plan(c, {QD_IN(F[QD_F_ISR_A], 0), QD_IN(F[QD_F_ISR_B], 0)});
    if (m < 0) return -1;
    QD_ROWS(c, m, G, hipLaunchKernelGGL(k_band_insolation, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, W, F[QD_F_ISR_A], F[QD_F_ISR_B],
                                        c->bands, plane));
    if (out_host) {
        const QdGeom& G = c->geo;
        const size_t rowb = (size_t)G.nlon * sizeof(double);
        for (int b = 0; b < nb; ++b)
            QD_HIP(c, hipMemcpyAsync((char*)out_host + ((size_t)b * G.nlat + G.row0) * rowb,
                                     (const char*)(c->bands + (size_t)b * plane) + (size_t)G.halo * rowb, rowb * G.nrows,
                                     hipMemcpyDeviceToHost, c->stream));
        QD_HIP(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}
