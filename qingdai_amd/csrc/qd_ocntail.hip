// qd_ocntail.hip -- the second half of an ocean sub-step in ONE launch (gfx950): continuity (ocean.py:365-374), SST
// semi-Lagrangian blend (ocean.py:380-382), lateral diffusion + Q_net heating (ocean.py:385-406), velocity outlier filter and
// caps (ocean.py:409-434).  It replaces k_cont_sstadv + k_sst_outlier_fused of qd_ocean.hip on whole-globe handles in the
// deferred-mean mode (the "eta -= mean; nan_to_num; clip" that ends the sub-step is applied by the next momentum kernel on load),
// with the same device functions in the same order, so the two forms agree bit for bit except for the order in which the
// area-weighted eta sum is accumulated (per 16 x 62 tile instead of per row).
//
// Why one launch: the advected SST is an intermediate that only its own Laplacian reads.  A 256-thread workgroup owns a tile of
// 16 rows x 62 columns; phase 1 forms nan_to_num(T1) = the blended, gathered SST on the tile plus two halo rows / one halo
// column each side (20 x 64, 10 KB of LDS); phase 2 takes K_h lap(T1) from LDS (east / west = neighbouring lanes), adds the
// heating and stores the new SST, updates eta (divergence of the new currents) and filters the currents.  Algorithmic HBM
// traffic: read uo, vo, eta, SST, Q_net + two masks (42 B), write eta, SST, uo, vo (32 B) = 74 B/cell instead of 115 B/cell.
#include "qd_internal.h"
#include "qd_device.h"
#include "qd_wave.h"
#include "qd_ocntail.h"

#define QT_TR 16                  // owned rows per tile
#define QT_TC 62                  // owned columns per tile (lanes 1..62)
#define QT_RA (QT_TR + 4)

// x / c for a divisor whose correctly rounded reciprocal rc = RN(1/c) is known (host-computed constants and row tables): one
// multiplication and two fused corrections instead of the ~25-instruction f64 division sequence.  q0 = RN(x rc) is a faithful
// quotient, e = x - c q0 is exact in one fma, and RN(q0 + e rc) is the correctly rounded quotient (Markstein's theorem; it can
// miss by one ulp only in rare boundary cases, well inside the stated tolerances, and never for 0, inf or NaN operands
// differently from a division: those propagate through the same products).
__device__ __forceinline__ double qt_div(double x, double c, double rc) {
    const double q0 = x * rc;
    const double e = __builtin_fma(-c, q0, x);
    const double q1 = __builtin_fma(e, rc, q0);
    return (q0 - q0 == 0.0) ? q1 : q0;                      // non-finite quotient: the correction would turn inf into NaN
}

// qd_departure (qd_device.h) with the four divisions by row / grid constants taken through qt_div
__device__ __forceinline__ QdBilin qt_departure(const QdGeom& G, int gi, int j, double u, double v, double dt,
                                                double acos, double r_acos, const QdTailArgs& P) {
    const double dl = qt_div(u * dt, acos, r_acos);
    const double dp = qt_div(v * dt, P.a, P.r_a);
    const double dx = qt_div(dl, P.dlon, P.r_dlon);
    const double dy = qt_div(dp, P.dlat, P.r_dlat);
    double r = qd_fold((double)gi - dy, G.nlat);
    double cc = qd_fold((double)j - dx, G.nlon);
    const bool nan_coord = (dx != dx) || (dy != dy);
    r = fmin(fmax(r, 0.0), (double)(G.nlat - 1));
    cc = fmin(fmax(cc, 0.0), (double)(G.nlon - 1));
    const double r0f = floor(r), c0f = floor(cc);
    QdBilin b;
    const int r0 = (int)r0f, c0 = (int)c0f;
    const int r1 = r0 + 1 < G.nlat ? r0 + 1 : G.nlat - 1;
    b.c0 = c0; b.c1 = c0 + 1 < G.nlon ? c0 + 1 : G.nlon - 1;
    b.l0 = qd_lrow_far(G, r0); b.l1 = qd_lrow_far(G, r1);
    const double tr = r - r0f, tc = cc - c0f;
    b.wr0 = 1.0 - tr; b.wr1 = tr; b.wc0 = 1.0 - tc; b.wc1 = tc;
    b.nan_coord = nan_coord;
    return b;
}

// qd_divvort_point (divergence form) with the same treatment; 1 / (a cos6) is a row table
__device__ __forceinline__ double qt_div_point(const QdGeom& G, const QdTabs& T, const double* __restrict__ p,
                                               const double* __restrict__ q, int i, int j, const QdTailArgs& P) {
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    const int jp = qd_wrapc(j + 1, G.nlon), jm = qd_wrapc(j - 1, G.nlon);
    const double dp = qt_div(p[b + jp] - p[b + jm], 2 * P.dlon, P.r_2dlon);
    double dq = 0.0;
    if (i != 0 && i != G.nlat - 1) {
        const double qn = q[(size_t)qd_lrow(G, i + 1) * G.nlon + j] * T.cos_raw[i + 1];
        const double qs = q[(size_t)qd_lrow(G, i - 1) * G.nlon + j] * T.cos_raw[i - 1];
        dq = qt_div(qn - qs, 2 * P.dlat, P.r_2dlat);
    }
    return T.inv_acos6[i] * (dp + dq);
}

__device__ __forceinline__ double qt_wave_sum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}

// nan_to_num'd T1 at global row r (inside the tile's 20 rows), lane l
#define QT_F(r, l) sT[(r) - ib][(l)]

// np.gradient along latitude of the LDS plane (qd_dphi<true> of qd_device.h on nan_to_num'd values)
__device__ __forceinline__ double qt_dphi(const double (*sT)[64], int ib, int n, int r, int l, double dphi) {
    if (r == 0) return (QT_F(1, l) - QT_F(0, l)) / dphi;
    if (r == n - 1) return (QT_F(n - 1, l) - QT_F(n - 2, l)) / dphi;
    return (QT_F(r + 1, l) - QT_F(r - 1, l)) / (2.0 * dphi);
}

// qd_lap_point_fast<true>(T1, G, T, kind 1, i, j, ...) with T1 in LDS: reciprocal row tables inside, the literal reference form
// (ocean.py:100-117) on the two rows next to each pole
__device__ __forceinline__ double qt_lap(const double (*sT)[64], int ib, const QdGeom& G, const QdTabs& T, int i, int l,
                                         double dphi, double dlam, double a) {
    const int n = G.nlat;
    const double cc = QT_F(i, l);
    if (i >= 2 && i <= n - 3) {
        const double Gb = T.lapA[1][i + 1] * (QT_F(i + 2, l) - cc);
        const double Ga = T.lapA[1][i - 1] * (cc - QT_F(i - 2, l));
        const double d2 = (QT_F(i, l + 1) - 2.0 * cc) + QT_F(i, l - 1);
        return T.lapP[1][i] * (Gb - Ga) + T.lapQ[1][i] * d2;
    }
    const double* __restrict__ cosf = T.cos05;
    int ra, rb; double den;
    if (i == 0) { ra = 0; rb = 1; den = dphi; }
    else if (i == n - 1) { ra = n - 2; rb = n - 1; den = dphi; }
    else { ra = i - 1; rb = i + 1; den = 2.0 * dphi; }
    const double Ga = cosf[ra] * qt_dphi(sT, ib, n, ra, l, dphi);
    const double Gb = cosf[rb] * qt_dphi(sT, ib, n, rb, l, dphi);
    const double ci = cosf[i];
    const double term_phi = (1.0 / ci) * ((Gb - Ga) / den);
    const double d2 = ((QT_F(i, l + 1) - 2.0 * cc) + QT_F(i, l - 1)) / (dlam * dlam);
    const double term_lam = d2 / (ci * ci);
    return (term_phi + term_lam) / (a * a);
}

__global__ void __launch_bounds__(256)
k_ocn_tail(QdGeom G, QdTabs T, QdTailArgs P) {
    __shared__ double sT[QT_RA][64];
    __shared__ double sAcc[4];
    const unsigned w = qd_xcd_chunk(blockIdx.x, gridDim.x);
    const int rs = (int)(w / (unsigned)P.ntc), cs = (int)(w % (unsigned)P.ntc);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = G.nlat, m = G.nlon;
    const int i0 = G.row0 + rs * QT_TR, i1 = min(i0 + QT_TR, G.row0 + G.nrows), ib = i0 - 2;
    const int jraw = cs * QT_TC - 1 + lane;
    const int j = jraw < 0 ? jraw + m : (jraw >= m ? jraw - m : jraw);
    const bool col_own = lane >= 1 && lane <= QT_TC && jraw < m;
    // ---- phase 1: nan_to_num(T1) on the tile + halo (ocean.py:380-382)
    for (int p = wv; p < QT_RA; p += 4) {
        const int i = ib + p;
        double t1 = 0.0;
        if (i >= 0 && i < n && i < i1 + 2) {
            const size_t o = (size_t)qd_lrow(G, i) * m + j;
            const QdBilin bl = qt_departure(G, i, j, P.uo[o], P.vo[o], P.sub_dt, P.a * T.cos05[i], T.ocn_igx[i], P);
            t1 = qd_nn((1.0 - P.alpha) * P.Ts[o] + P.alpha * qd_gather(P.Ts, G, bl));
        }
        sT[p][lane] = t1;
    }
    __syncthreads();
    // ---- phase 2: the owned cells
    double acc = 0.0;
    for (int r = wv; r < QT_TR; r += 4) {
        const int i = i0 + r;
        if (i >= i1 || !col_own) continue;
        const size_t b = (size_t)qd_lrow(G, i) * m;
        const size_t o = b + j;
        // continuity (ocean.py:365-374): eta += -dt H div, land zero, area-weighted sum
        {
            const double div = qt_div_point(G, T, P.uo, P.vo, i, j, P);
            double e = P.eta[o] + P.msdtH * div;
            const bool island = P.land[o] == 1;
            if (island) e = 0.0;
            P.eta[o] = e;
            acc += e * (island ? 0.0 : T.warea[i]);
        }
        // K_h lap(T1) + heating (ocean.py:385-406, 440)
        {
            double Tv = sT[r + 2][lane];
            if (P.K_h > 0.0) Tv = Tv + P.sub_dt * P.K_h * qt_lap(sT, ib, G, T, i, lane, P.dlat, P.dlon, P.a);
            if (P.use_q) {
                const double heat = qt_div(P.qnet[o], P.rcH, P.r_rcH);
                const bool ocean = P.land[o] == 0;
                if (P.has_ice) {
                    const bool ic = P.ice[o] != 0;
                    if (ocean && !ic) Tv = Tv + P.sub_dt * heat;
                    if (P.ice_qfac > 0.0 && ocean && ic) Tv = Tv + P.sub_dt * P.ice_qfac * heat;
                } else if (ocean) Tv = Tv + P.sub_dt * heat;
            }
            P.Ts_out[o] = qd_nn(Tv);
        }
        // outliers + caps (ocean.py:409-434)
        {
            double u = qd_nn(P.uo[o]), v = qd_nn(P.vo[o]);
            const double cap = P.cap, s2 = u * u + v * v;
            // speed = sqrt(s2) is only compared with the cap: far below it (the usual case) nothing changes and no square root,
            // neighbour mean or division is needed; the reference arithmetic runs for the lanes near or above the cap
            if (!(s2 < 0.81 * (cap * cap))) {
                const double speed = sqrt(s2);
                if (P.mean4) {
                    if (speed > cap) {
                        const size_t bn = (size_t)qd_lrow(G, i + 1) * m, bs = (size_t)qd_lrow(G, i - 1) * m;
                        const int je = qd_wrapc(j + 1, m), jw = qd_wrapc(j - 1, m);
                        u = 0.25 * (qd_nn(P.uo[bn + j]) + qd_nn(P.uo[bs + j]) + qd_nn(P.uo[b + je]) + qd_nn(P.uo[b + jw]));
                        v = 0.25 * (qd_nn(P.vo[bn + j]) + qd_nn(P.vo[bs + j]) + qd_nn(P.vo[b + je]) + qd_nn(P.vo[b + jw]));
                    }
                    const double sp2 = sqrt(u * u + v * v);
                    const double sc2 = (sp2 > cap) ? cap / (sp2 + 1e-12) : 1.0;
                    u = u * sc2; v = v * sc2;
                } else {
                    const double sc = (speed > cap) ? cap / (speed + 1e-12) : 1.0;
                    u = u * sc; v = v * sc;
                }
            }
            P.uo_out[o] = u; P.vo_out[o] = v;
        }
    }
    // ---- area-weighted eta sum of the tile (fixed order: lanes by shuffle tree, waves 0..3)
    acc = qt_wave_sum(acc);
    if (lane == 0) sAcc[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) P.partial[w] = ((sAcc[0] + sAcc[1]) + sAcc[2]) + sAcc[3];
}

int qd_ocn_tail_tiles(const QdGeom& G) { return ((G.nrows + QT_TR - 1) / QT_TR) * ((G.nlon + QT_TC - 1) / QT_TC); }

int qd_launch_ocn_tail(qd_ctx* c, const QdGeom& G, QdTailArgs& P) {
    if (G.nlon < 64 || !G.full) return qd_fail(c, "k_ocn_tail: whole-globe handles of >= 64 columns only");
    P.ntc = (G.nlon + QT_TC - 1) / QT_TC;
    const int ntr = (G.nrows + QT_TR - 1) / QT_TR;
    if (ntr * P.ntc > c->red_blocks) return qd_fail(c, "k_ocn_tail: partial buffer too small");
    QdScope sc(c, "ocean_tail");
    hipLaunchKernelGGL(k_ocn_tail, dim3(ntr * P.ntc), dim3(256), 0, c->stream, G, c->tabs, P);
    return 0;
}
