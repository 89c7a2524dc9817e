// qd_device.h -- device-side building blocks shared by the kernels of libqingdai_hip.so.
// Expression order follows the reference line by line (citations in qd_stencil.hip).
#pragma once
#include "qd_internal.h"
#include "qd_math.h"

// ------------------------------------------------------------------ Laplacian (O1)
template <bool SCRUB>
__device__ __forceinline__ double qd_ld(const double* __restrict__ F, const QdGeom& G, int g, int j) {
    double x = F[(size_t)qd_lrow(G, g) * G.nlon + j];
    return SCRUB ? qd_nn(x) : x;
}

// d/dphi of F at global row r (np.gradient: centred inside, one-sided at the two poles)
template <bool SCRUB>
__device__ __forceinline__ double qd_dphi(const double* __restrict__ F, const QdGeom& G, int r, int j, double dphi) {
    const int n = G.nlat;
    if (r == 0) return (qd_ld<SCRUB>(F, G, 1, j) - qd_ld<SCRUB>(F, G, 0, j)) / dphi;
    if (r == n - 1) return (qd_ld<SCRUB>(F, G, n - 1, j) - qd_ld<SCRUB>(F, G, n - 2, j)) / dphi;
    return (qd_ld<SCRUB>(F, G, r + 1, j) - qd_ld<SCRUB>(F, G, r - 1, j)) / (2.0 * dphi);
}

template <bool SCRUB>
__device__ __forceinline__ double qd_lap_point(const double* __restrict__ F, const QdGeom& G,
                                               const double* __restrict__ cosf, int i, int j,
                                               double dphi, double dlam, double a) {
    const int n = G.nlat;
    int ra, rb; double den;
    if (i == 0) { ra = 0; rb = 1; den = dphi; }
    else if (i == n - 1) { ra = n - 2; rb = n - 1; den = dphi; }
    else { ra = i - 1; rb = i + 1; den = 2.0 * dphi; }
    const double Ga = cosf[ra] * qd_dphi<SCRUB>(F, G, ra, j, dphi);
    const double Gb = cosf[rb] * qd_dphi<SCRUB>(F, G, rb, j, dphi);
    const double ci = cosf[i];
    const double term_phi = (1.0 / ci) * ((Gb - Ga) / den);
    const int jp = qd_wrapc(j + 1, G.nlon), jm = qd_wrapc(j - 1, G.nlon);
    const double c = qd_ld<SCRUB>(F, G, i, j);
    const double d2 = ((qd_ld<SCRUB>(F, G, i, jp) - 2.0 * c) + qd_ld<SCRUB>(F, G, i, jm)) / (dlam * dlam);
    const double term_lam = d2 / (ci * ci);
    return (term_phi + term_lam) / (a * a);
}


// Reciprocal-table form of the same Laplacian for interior rows (2 <= i <= n-3); the two rows next to
// each pole fall back to the literal form above.  k: cos-floor kind (0: 0.2, 1: 0.5).
template <bool SCRUB>
__device__ __forceinline__ double qd_lap_point_fast(const double* __restrict__ F, const QdGeom& G, const QdTabs& T,
                                                    int k, int i, int j, double dphi, double dlam, double a) {
    if (i < 2 || i > G.nlat - 3) return qd_lap_point<SCRUB>(F, G, k ? T.cos05 : T.cos02, i, j, dphi, dlam, a);
    const int jp = qd_wrapc(j + 1, G.nlon), jm = qd_wrapc(j - 1, G.nlon);
    const double cc = qd_ld<SCRUB>(F, G, i, j);
    const double Gb = T.lapA[k][i + 1] * (qd_ld<SCRUB>(F, G, i + 2, j) - cc);
    const double Ga = T.lapA[k][i - 1] * (cc - qd_ld<SCRUB>(F, G, i - 2, j));
    const double d2 = (qd_ld<SCRUB>(F, G, i, jp) - 2.0 * cc) + qd_ld<SCRUB>(F, G, i, jm);
    return T.lapP[k][i] * (Gb - Ga) + T.lapQ[k][i] * d2;
}

// ------------------------------------------------------------------ semi-Lagrangian gather (O5)
__device__ __forceinline__ double qd_fold(double x, int n) {
    // scipy.ndimage map_coordinates(mode='wrap'): period n-1
    const double sz = (double)(n - 1);
    if (n <= 1) return 0.0;
    if (x < 0.0) return x + sz * (trunc(-x / sz) + 1.0);
    if (x > sz) return x - sz * trunc(x / sz);
    return x;
}

struct QdBilin { int l0, l1, c0, c1; double wr0, wr1, wc0, wc1; bool nan_coord; };

__device__ __forceinline__ QdBilin qd_departure(const QdGeom& G, int gi, int j, double u, double v, double dt,
                                                double a, double cosl, double dlat, double dlon) {
    const double dl = u * dt / (a * cosl);
    const double dp = v * dt / a;
    const double dx = dl / dlon;
    const double dy = dp / dlat;
    double r = qd_fold((double)gi - dy, G.nlat);
    double cc = qd_fold((double)j - dx, G.nlon);
    // scipy treats a NaN coordinate as outside the array: constant fill 0.0 (measured)
    const bool nan_coord = (dx != dx) || (dy != dy);
    // NaN/garbage-safe clamps (never fault); finite inputs are already inside the range
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

__device__ __forceinline__ double qd_gather(const double* __restrict__ F, const QdGeom& G, const QdBilin& b) {
    const size_t r0 = (size_t)b.l0 * G.nlon, r1 = (size_t)b.l1 * G.nlon;
    double t = 0.0;                       // scipy NI_GeometricTransform corner order
    t += F[r0 + b.c0] * b.wr0 * b.wc0;
    t += F[r0 + b.c1] * b.wr0 * b.wc1;
    t += F[r1 + b.c0] * b.wr1 * b.wc0;
    t += F[r1 + b.c1] * b.wr1 * b.wc1;
    return b.nan_coord ? 0.0 : t;
}


// ------------------------------------------------------------------ divergence / vorticity (O11)
__device__ __forceinline__ double qd_divvort_point(const QdGeom& G, const QdTabs& T, const double* __restrict__ p,
                                                   const double* __restrict__ q, int i, int j, double a,
                                                   double dlat, double dlon, int vort) {
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    const int jp = qd_wrapc(j + 1, G.nlon), jm = qd_wrapc(j - 1, G.nlon);
    const double dp = (p[b + jp] - p[b + jm]) / (2 * dlon);
    double dq = 0.0;
    if (i != 0 && i != G.nlat - 1) {
        const double qn = q[(size_t)qd_lrow(G, i + 1) * G.nlon + j] * T.cos_raw[i + 1];
        const double qs = q[(size_t)qd_lrow(G, i - 1) * G.nlon + j] * T.cos_raw[i - 1];
        dq = (qn - qs) / (2 * dlat);
    }
    const double pre = 1 / (a * T.cos6[i]);
    return vort ? pre * (dp - dq) : pre * (dp + dq);
}

