// qd_fused.hip -- the fused momentum + del^4 kernels (gfx950), the hot stencil of the path.
//
//   k_dyn_hyper<TR>   atmosphere: np.gradient(h) -> geostrophic-relaxation | primitive momentum
//                     (dynamics.py:482-530) -> del^4 hyperdiffusion of u, v, h, q, cloud
//                     (dynamics.py:533-594, 144-212) in ONE launch.
//   k_ocn_hyper<TR>   ocean sub-step: grad(eta) + Coriolis + wind stress + drag + land mask + polar
//                     sponge (ocean.py:306-336) -> del^4 of uo, vo, eta (ocean.py:341-356).
//
// Algorithmic HBM traffic of k_dyn_hyper: read u,v,h,friction,q,cloud (48 B) + write u,v,h,q,cloud
// (40 B) = 88 B/cell instead of the 22 field passes of the unfused sequence;
// k_ocn_hyper: read uo,vo,eta,tau_x,tau_y (+mask) and write uo,vo,eta = 65 B/cell.
//
// Design (MI355X / CDNA4), details at the FAST / EXACT sections below and in DESIGN.md section 4:
//   * One 512-thread workgroup = 8 wavefronts owns a TR x 58 tile (RA = TR+10 plane rows x 64 columns).  A wavefront IS a
//     64-column row segment: lane l always works on global column j0-3+l, so every global access is one coalesced 512-byte
//     row segment.  Each Laplacian is a 5-row x 3-column star (its latitude part gradient(cos*gradient(F)) only touches rows
//     r-2, r, r+2), so del^4 needs F on (TR+8) x 62 and the momentum update that produces F needs h on (TR+10) x 64.  Halo
//     cells are recomputed, never exchanged.
//   * FAST path (every tile of a >= 64-column grid): wave w owns K = RA/8 consecutive plane rows and keeps its cells of every
//     plane in registers; row neighbours across a wave boundary go through LDS (4 rows per pass), column neighbours through
//     DPP wave shifts, per-row coefficients through scalar loads; no nan_to_num (one v_cmp_class per output, EXACT fallback).
//   * EXACT path (narrow grids, short row segments, non-finite values): three LDS planes A (h), B (field entering del^4),
//     D (its Laplacian), LDS-staged row tables, every cell masked, literal nan_to_num / np.clip.  Bit-identical to FAST.
//   * Divisions by per-row / constant metrics are folded into host-computed reciprocal tables (lapA/lapP/lapQ, lapPoleA for the
//     four one-sided np.gradient rows next to the poles, mom_cu/cv/px, ocn_igx).
//   * TR is a template parameter; the host picks the instantiation whose tile count fills 256 CUs x 2 resident workgroups in
//     one round (TR = 38 at 721 x 1440).  Tiles are dealt to the 8 XCDs in contiguous chunks, pole tiles first.
#include "qd_internal.h"
#include "qd_device.h"
#include "qd_fused.h"
#include "qd_wave.h"
#include <cstdlib>
#include <algorithm>

#define QD_TC 58                 // owned columns per tile (lanes 3..60)
#define QD_S 64                  // plane row stride (doubles) = one wavefront
#define QD_FBLOCK 512             // threads per fused-kernel workgroup
#define QD_NW (QD_FBLOCK / 64)   // wavefronts per workgroup

struct QdLapC {                  // scalars + reciprocal row tables of the spherical Laplacian
    double dphi, dlam, a;
    int n;                       // nlat
    const double* poleA;         // QdTabs::lapPoleA of the cos-floor kind (pole rows only)
    const double *sA, *sP, *sQ;  // LDS copies of lapA/lapP/lapQ indexed by PLANE row rho
};

// pole-row type of global row g: 0,1,2,3 for g = 0, 1, n-2, n-1; -1 otherwise
__device__ __forceinline__ int qd_pole_type(int g, int n) { return g == 0 ? 0 : (g == 1 ? 1 : (g == n - 2 ? 2 : (g == n - 1 ? 3 : -1))); }

// the two rows next to each pole: one-sided np.gradient (grid.py:41-88) in the same reciprocal form, with the
// host-computed coefficients of the row type (QdTabs::lapPoleA; lapP holds the row's P)
__device__ __forceinline__ double qd_lap_lds_pole(const double* __restrict__ p, int g, int rho, const QdLapC& C) {
    const int S = QD_S, t = qd_pole_type(g, C.n);
    const double cc = p[0];
    double ahi = cc, alo, bhi, blo = cc;
    if (t == 0) { ahi = p[S]; alo = cc; } else if (t == 1) alo = p[-S]; else alo = p[-2 * S];
    if (t == 2) bhi = p[S]; else if (t == 3) { bhi = cc; blo = p[-S]; } else bhi = p[2 * S];
    const double Gb = C.poleA[2 * t + 1] * (bhi - blo);
    const double Ga = C.poleA[2 * t] * (ahi - alo);
    const double d2 = (p[1] - 2.0 * cc) + p[-1];
    return C.sP[rho] * (Gb - Ga) + C.sQ[rho] * d2;
}

// Laplacian at plane position p (row stride QD_S) whose global row is g.  Plane values are already
// nan_to_num'd.  Interior rows use the reciprocal form; rows 0,1,n-2,n-1 the literal reference form.
__device__ __forceinline__ double qd_lap_lds(const double* __restrict__ p, int g, int rho, const QdLapC& C) {
    const int n = C.n, S = QD_S;
    const double cc = p[0];
    if (g >= 2 && g <= n - 3) {
        //   L = P_i (A_{i+1} (F_{i+2} - F_i) - A_{i-1} (F_i - F_{i-2})) + Q_i ((F_{j+1} - 2F) + F_{j-1})
        const double Gb = C.sA[rho + 1] * (p[2 * S] - cc);
        const double Ga = C.sA[rho - 1] * (cc - p[-2 * S]);
        const double d2 = (p[1] - 2.0 * cc) + p[-1];
        return C.sP[rho] * (Gb - Ga) + C.sQ[rho] * d2;
    }
    return qd_lap_lds_pole(p, g, rho, C);
}

// Plane geometry shared by both kernels.  Common origin: plane row rho <-> global row i0-5+rho,
// lane l <-> global column j0-3+l.   A: rho in [0,TR+10)   B: [1,TR+9)   D: [3,TR+7)   owned: [5,TR+5)
template <int TR> struct QdPl {
    static constexpr int RA = TR + 10;
    static constexpr int K = (RA + QD_NW - 1) / QD_NW;     // cells per thread
    static constexpr int NT = 10;                             // staged row tables
    static constexpr size_t lds_bytes = sizeof(double) * (QD_S * (size_t)(3 * RA) + (size_t)NT * RA);
};

// del^4 of the field currently in plane B (valid rows [1,TR+9), lanes 1..62), written to `out`.
// Caller has synchronised after filling B.  The routine does NOT end with a barrier (see its last comment).
template <int TR>
__device__ __forceinline__ void qd_del4_from_B(const double* __restrict__ B, double* __restrict__ D,
                                               double* __restrict__ out, const QdGeom& G, int i0, int j0,
                                               const QdLapC& C, const double* __restrict__ sK4, double dt) {
    constexpr int K = QdPl<TR>::K;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row math + table loads go scalar
    const int gend = G.row0 + G.nrows;
#pragma unroll 2
    for (int k = 0; k < K; ++k) {
        const int rho = wv + QD_NW * k;
        const int g = i0 - 5 + rho;
        if (rho >= 3 && rho < TR + 7 && lane >= 2 && lane <= 61)
            D[rho * QD_S + lane] = (g >= 0 && g < C.n) ? qd_nnf(qd_lap_lds(B + rho * QD_S + lane, g, rho, C)) : 0.0;
    }
    __syncthreads();
    const int j = j0 - 3 + lane;
#pragma unroll 2
    for (int k = 0; k < K; ++k) {
        const int rho = wv + QD_NW * k;
        const int g = i0 - 5 + rho;
        if (rho >= 5 && rho < TR + 5 && g < gend && lane >= 3 && lane <= 60 && j < G.nlon) {
            const double L2 = qd_lap_lds(D + rho * QD_S + lane, g, rho, C);
            const double k4 = sK4[rho];
            out[(size_t)qd_lrow(G, g) * G.nlon + j] = qd_nnf(B[rho * QD_S + lane] - (k4 * L2) * dt);
        }
    }
    // no barrier here: this phase reads B only at the thread's OWN cells, which are exactly the cells the
    // thread overwrites when it fills B for the next field; D is rewritten only after the barrier that
    // follows that fill, i.e. after every wave has left this phase.
}

// store owned cells straight from per-thread register values (field skipped by the k4<=0 early-out)
template <int TR>
__device__ __forceinline__ void qd_store_owned(const double (&val)[QdPl<TR>::K], double* __restrict__ out,
                                               const QdGeom& G, int i0, int j0) {
    constexpr int K = QdPl<TR>::K;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row math + table loads go scalar
    const int j = j0 - 3 + lane, gend = G.row0 + G.nrows;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int rho = wv + QD_NW * k;
        const int g = i0 - 5 + rho;
        if (rho >= 5 && rho < TR + 5 && g < gend && lane >= 3 && lane <= 60 && j < G.nlon)
            out[(size_t)qd_lrow(G, g) * G.nlon + j] = val[k];
    }
}

__device__ __forceinline__ int qd_wrapj(int j, int n) {   // periodic column; narrow grids wrap more than once
    j = j < 0 ? j + n : (j >= n ? j - n : j);
    if (j < 0 || j >= n) { j %= n; if (j < 0) j += n; }
    return j;
}

// Tiles are dealt to the 8 XCDs in contiguous chunks; the first and last tile rows (the pole tiles, which
// take the slower EXACT path) come first in the order so they never form the tail of the launch.
__device__ __forceinline__ void qd_tile_origin(int ntc, int ntr, int tr, int row0, int& i0, int& j0) {
    const unsigned nb = gridDim.x, L = blockIdx.x, per = nb >> 3, rem = nb & 7u, x = L & 7u;
    const unsigned w = x * per + (x < rem ? x : rem) + (L >> 3);      // XCD-contiguous dealing
    int ti = (int)(w / (unsigned)ntc);
    ti = ti == 0 ? 0 : (ti == 1 ? ntr - 1 : ti - 1);
    i0 = row0 + ti * tr;
    j0 = (int)(w % (unsigned)ntc) * QD_TC;
}

// =========================================================================================
// atmosphere
// =========================================================================================
// momentum update of one cell (dynamics.py:488-530); p points at the cell in the h plane.
// t8/t9: staged row coefficients (geos: mom_cu, mom_cv; primitive: mom_px, f)
__device__ __forceinline__ double qd_mom_cell(const double* __restrict__ p, int g, int j, const QdGeom& G,
                                              const QdDynArgs& P, int comp, double u0, double v0, double fr,
                                              double t8, double t9) {
    const int n = G.nlat, m = G.nlon, S = QD_S;
    if (P.primitive) {
        const double f = t9;
        if (comp == 0) {
            const double dh_dlon = (j == 0) ? (p[1] - p[0]) * P.inv_dlon : (j == m - 1) ? (p[0] - p[-1]) * P.inv_dlon
                                                                                      : (p[1] - p[-1]) * P.inv_2dlon;
            return qd_clip(u0 + (t8 * dh_dlon + f * v0 - fr * u0) * P.dt, -200.0, 200.0);
        }
        const double dh_dlat = (g == 0) ? (p[S] - p[0]) * P.inv_dlat : (g == n - 1) ? (p[0] - p[-S]) * P.inv_dlat
                                                                                    : (p[S] - p[-S]) * P.inv_2dlat;
        return qd_clip(v0 + (P.pgf_y * dh_dlat - f * u0 - fr * v0) * P.dt, -200.0, 200.0);
    }
    if (comp == 0) {
        const double dh_dlat = (g == 0) ? (p[S] - p[0]) * P.inv_dlat : (g == n - 1) ? (p[0] - p[-S]) * P.inv_dlat
                                                                                    : (p[S] - p[-S]) * P.inv_2dlat;
        const double u_g = qd_clip(t8 * dh_dlat, -200.0, 200.0);
        const double un = u0 * 0.8 + u_g * 0.2;
        return un + (-fr * un) * P.dt;
    }
    const double dh_dlon = (j == 0) ? (p[1] - p[0]) * P.inv_dlon : (j == m - 1) ? (p[0] - p[-1]) * P.inv_dlon
                                                                              : (p[1] - p[-1]) * P.inv_2dlon;
    const double v_g = qd_clip(t9 * dh_dlon, -200.0, 200.0);
    const double vn = v0 * 0.8 + v_g * 0.2;
    return vn + (-fr * vn) * P.dt;
}

// EXACT path: any tile (pole rows, slab edges, narrow grids, non-finite values): literal nan_to_num at
// every stage, one-sided np.gradient rows, every cell masked individually.
template <int TR>
__device__ __forceinline__ void qd_dyn_exact(const QdGeom& G, const QdTabs& T, const QdDynArgs& P, int i0, int j0, double* lds) {
    constexpr int RA = QdPl<TR>::RA, K = QdPl<TR>::K;
    double* A = lds;
    double* B = A + RA * QD_S;
    double* D = B + RA * QD_S;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row math + table loads go scalar
    const int j = qd_wrapj(j0 - 3 + lane, G.nlon);
    // row tables of this tile's plane rows, staged once in LDS (broadcast reads afterwards):
    // 0 lapA  1 lapP  2 lapQ  3..7 k4[u,v,h,q,cloud]  8 mom_cu | mom_px  9 mom_cv | fcor
    double* sT = D + RA * QD_S;
    if (threadIdx.x < RA) {
        const int rho = threadIdx.x, g = i0 - 5 + rho;
        const bool ok = g >= 0 && g < G.nlat;
        const int gg = ok ? g : 0;
        sT[0 * RA + rho] = ok ? T.lapA[0][gg] : 0.0;
        sT[1 * RA + rho] = ok ? T.lapP[0][gg] : 0.0;
        sT[2 * RA + rho] = ok ? T.lapQ[0][gg] : 0.0;
        for (int f = 0; f < 5; ++f) sT[(3 + f) * RA + rho] = P.k4row[f] ? (ok ? P.k4row[f][gg] : 0.0) : P.k4s[f];
        sT[8 * RA + rho] = ok ? (P.primitive ? T.mom_px[gg] : T.mom_cu[gg]) : 0.0;
        sT[9 * RA + rho] = ok ? (P.primitive ? T.fcor[gg] : T.mom_cv[gg]) : 0.0;
    }
    const QdLapC C{P.dlat, P.dlon, P.a, G.nlat, T.lapPoleA[0], sT, sT + RA, sT + 2 * RA};

    // ---- every global load of the tile, issued back to back (unconditional loads from clamped
    //      offsets, masked afterwards: no branch and no wait sits between two loads)
    double ah[K], ru[K], rv[K], rf[K], rq[K], rc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int rho = wv + QD_NW * k;
        const int g = i0 - 5 + rho;
        // rows of the last tile that fall off the slab (band handles) are masked like rows beyond a pole
        const bool ok = (rho < RA) && (g >= 0) && (g < G.nlat) && (qd_lrow(G, g) < G.lrows());
        const unsigned o = ok ? (unsigned)qd_lrow(G, g) * (unsigned)G.nlon + (unsigned)j : (unsigned)j;
        ah[k] = P.h[o]; ru[k] = P.u[o]; rv[k] = P.v[o]; rf[k] = P.fric[o]; rq[k] = P.q[o]; rc[k] = P.cloud[o];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int rho = wv + QD_NW * k;
        const int g = i0 - 5 + rho;
        const bool ok = (rho < RA) && (g >= 0) && (g < G.nlat) && (qd_lrow(G, g) < G.lrows());
        const bool inB = ok && rho >= 1 && rho < TR + 9;
        ah[k] = ok ? ah[k] : 0.0;
        ru[k] = inB ? ru[k] : 0.0; rv[k] = inB ? rv[k] : 0.0; rf[k] = inB ? rf[k] : 0.0;
        rq[k] = inB ? rq[k] : 0.0; rc[k] = inB ? rc[k] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) { const int rho = wv + QD_NW * k; if (rho < RA) A[rho * QD_S + lane] = ah[k]; }
    __syncthreads();

    // ---- u', v'
#pragma unroll
    for (int comp = 0; comp < 2; ++comp) {
        double keep[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int rho = wv + QD_NW * k;
            const int g = i0 - 5 + rho;
            double val = 0.0;
            if (rho >= 1 && rho < TR + 9 && lane >= 1 && lane <= 62 && g >= 0 && g < G.nlat)
                val = qd_mom_cell(A + rho * QD_S + lane, g, j, G, P, comp, ru[k], rv[k], rf[k], sT[8 * RA + rho], sT[9 * RA + rho]);
            keep[k] = val;
            if (rho >= 1 && rho < TR + 9) B[rho * QD_S + lane] = qd_nnf(val);     // _hyperdiffuse starts from nan_to_num(F)
        }
        double* out = comp == 0 ? P.uo : P.vo;
        if (P.skip[comp]) { qd_store_owned<TR>(keep, out, G, i0, j0); }
        else {
            __syncthreads();
            qd_del4_from_B<TR>(B, D, out, G, i0, j0, C, sT + (3 + comp) * RA, P.dt);
        }
    }
    // ---- h, q, cloud: B <- nan_to_num(register copy)
#pragma unroll
    for (int f = 2; f < 5; ++f) {
        double* out = f == 2 ? P.ho : (f == 3 ? P.qo : P.co);
        if (P.skip[f]) {
            if (f == 2) qd_store_owned<TR>(ah, out, G, i0, j0);
            else if (f == 3) qd_store_owned<TR>(rq, out, G, i0, j0);
            else qd_store_owned<TR>(rc, out, G, i0, j0);
            continue;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int rho = wv + QD_NW * k;
            if (rho >= 1 && rho < TR + 9) B[rho * QD_S + lane] = qd_nnf(f == 2 ? ah[k] : (f == 3 ? rq[k] : rc[k]));
        }
        __syncthreads();
        qd_del4_from_B<TR>(B, D, out, G, i0, j0, C, sT + (3 + f) * RA, P.dt);
    }
}

// =========================================================================================
// ocean
// =========================================================================================
template <int TR>
__device__ __forceinline__ void qd_ocn_exact(const QdGeom& G, const QdTabs& T, const QdOcnArgs& P, int i0, int j0, double* lds) {
    constexpr int RA = QdPl<TR>::RA, K = QdPl<TR>::K;
    double* A = lds;
    double* B = A + RA * QD_S;
    double* D = B + RA * QD_S;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row math + table loads go scalar
    const int j = qd_wrapj(j0 - 3 + lane, G.nlon);
    // staged row tables: 0 lapA 1 lapP 2 lapQ 3..5 k4[uo,vo,eta] 6 fcor 7 ocn_igx 8 r_extra
    double* sT = D + RA * QD_S;
    if (threadIdx.x < RA) {
        const int rho = threadIdx.x, g = i0 - 5 + rho;
        const bool ok = g >= 0 && g < G.nlat;
        const int gg = ok ? g : 0;
        sT[0 * RA + rho] = ok ? T.lapA[1][gg] : 0.0;
        sT[1 * RA + rho] = ok ? T.lapP[1][gg] : 0.0;
        sT[2 * RA + rho] = ok ? T.lapQ[1][gg] : 0.0;
        for (int f = 0; f < 3; ++f) sT[(3 + f) * RA + rho] = P.k4row[f] ? (ok ? P.k4row[f][gg] : 0.0) : P.k4s[f];
        sT[6 * RA + rho] = ok ? T.fcor[gg] : 0.0;
        sT[7 * RA + rho] = ok ? T.ocn_igx[gg] : 0.0;
        sT[8 * RA + rho] = ok ? T.r_extra[gg] : 0.0;
    }
    const QdLapC C{P.dlat, P.dlon, P.a, G.nlat, T.lapPoleA[1], sT, sT + RA, sT + 2 * RA};

    double ae[K], ru[K], rv[K], rtx[K], rty[K];
    int rl[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int rho = wv + QD_NW * k;
        const int g = i0 - 5 + rho;
        // eta rows wrap across the poles (np.roll(axis=0), ocean.py:308); band mode reads the halo
        int ge = g;
        if (ge < 0) ge += G.nlat; else if (ge >= G.nlat) ge -= G.nlat;
        const bool okA = rho < RA && qd_lrow(G, ge) < G.lrows();      // off-slab rows of a band's last tile: masked
        const bool inB = okA && rho >= 1 && rho < TR + 9 && g >= 0 && g < G.nlat;
        const unsigned oe = okA ? (unsigned)qd_lrow(G, ge) * (unsigned)G.nlon + (unsigned)j : (unsigned)j;
        const unsigned o = inB ? (unsigned)qd_lrow(G, g) * (unsigned)G.nlon + (unsigned)j : (unsigned)j;
        ae[k] = P.eta[oe]; ru[k] = P.uo[o]; rv[k] = P.vo[o]; rtx[k] = P.taux[o]; rty[k] = P.tauy[o];
        rl[k] = (int)P.land[o];
    }
    if (P.eta_mean) {                                       // deferred end of the previous sub-step (see QdOcnArgs)
        const double em = *P.eta_mean;
#pragma unroll
        for (int k = 0; k < K; ++k) ae[k] = qd_clip(qd_nn(ae[k] - em), -P.eta_cap, P.eta_cap);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int rho = wv + QD_NW * k;
        const int g = i0 - 5 + rho;
        int ge = g;
        if (ge < 0) ge += G.nlat; else if (ge >= G.nlat) ge -= G.nlat;
        const bool okA = rho < RA && qd_lrow(G, ge) < G.lrows();
        const bool inB = okA && rho >= 1 && rho < TR + 9 && g >= 0 && g < G.nlat;
        ae[k] = okA ? ae[k] : 0.0;
        ru[k] = inB ? ru[k] : 0.0; rv[k] = inB ? rv[k] : 0.0; rtx[k] = inB ? rtx[k] : 0.0; rty[k] = inB ? rty[k] : 0.0;
        rl[k] = inB ? rl[k] : 0;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) { const int rho = wv + QD_NW * k; if (rho < RA) A[rho * QD_S + lane] = ae[k]; }
    __syncthreads();

#pragma unroll
    for (int comp = 0; comp < 2; ++comp) {
        double keep[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int rho = wv + QD_NW * k;
            const int g = i0 - 5 + rho;
            double val = 0.0;
            if (rho >= 1 && rho < TR + 9 && lane >= 1 && lane <= 62 && g >= 0 && g < G.nlat) {
                // ocean.py:306-336 (both axes roll-periodic)
                const double* p = A + rho * QD_S + lane;
                const double f = sT[6 * RA + rho];
                double xn;
                if (comp == 0) {
                    const double gx = ((p[1] - p[-1]) * P.inv_2dlon) * sT[7 * RA + rho];
                    const double du = (f * rv[k] - P.g * gx + rtx[k] * P.inv_rhoH - P.r_bot * ru[k]);
                    xn = ru[k] + P.sub_dt * du;
                } else {
                    const double gy = ((p[QD_S] - p[-QD_S]) * P.inv_2dlat) * P.inv_a;
                    const double dv = (-f * ru[k] - P.g * gy + rty[k] * P.inv_rhoH - P.r_bot * rv[k]);
                    xn = rv[k] + P.sub_dt * dv;
                }
                if (rl[k] == 1) xn = 0.0;
                val = xn - P.sub_dt * sT[8 * RA + rho] * xn;
            }
            keep[k] = val;
            if (rho >= 1 && rho < TR + 9) B[rho * QD_S + lane] = qd_nnf(val);
        }
        double* out = comp == 0 ? P.uo_out : P.vo_out;
        if (P.skip[comp]) { qd_store_owned<TR>(keep, out, G, i0, j0); }
        else {
            __syncthreads();
            qd_del4_from_B<TR>(B, D, out, G, i0, j0, C, sT + (3 + comp) * RA, P.sub_dt);
        }
    }
    if (P.skip[2]) { qd_store_owned<TR>(ae, P.eta_out, G, i0, j0); return; }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int rho = wv + QD_NW * k;
        const int g = i0 - 5 + rho;
        if (rho >= 1 && rho < TR + 9) B[rho * QD_S + lane] = (g >= 0 && g < G.nlat) ? qd_nnf(ae[k]) : 0.0;
    }
    __syncthreads();
    qd_del4_from_B<TR>(B, D, P.eta_out, G, i0, j0, C, sT + 5 * RA, P.sub_dt);
}


// =========================================================================================
// FAST path
// =========================================================================================
// What limits the EXACT kernels is not HBM but the per-CU LDS pipe, the VALU instruction count and the
// latency of 11 barrier phases (profiles/: ~10 LDS reads and ~75 VALU instructions per Laplacian point).
// For finite values the same arithmetic needs far less machinery:
//   * wave w owns K = (TR+10)/8 CONSECUTIVE plane rows; a thread keeps its K cells of every plane in
//     registers, so the +-2 row neighbours of a Laplacian are registers except across a wave boundary
//     (only the first/last two rows of each wave travel through LDS: 4 writes + 4 reads per phase
//     instead of ~60 reads);
//   * east / west neighbours come from the adjacent lanes with DPP wave shifts of the register value;
//   * per-row metric tables are read with scalar loads (constant address space -> s_load into SGPRs);
//   * nan_to_num is the identity on finite values: the fast path omits it, tests every owned output (and
//     every clip input) with one v_cmp_class_f64 and, if any thread of the workgroup saw a non-finite
//     value, the whole tile is recomputed on the EXACT path (outputs go to separate buffers, so that is
//     safe).  A non-finite value anywhere in the stencil inputs of an owned cell reaches that cell.
//   * POLE = false: interior tiles, straight-line code.  POLE = true: tiles holding a pole row, rows
//     beyond a pole or rows off the slab: same code plus per-row (wave-uniform) patches -- literal
//     one-sided np.gradient expressions for rows 0,1,n-2,n-1, zero for rows outside the domain.
// Per-field launch arguments (output pointer, k4 row / scalar, skip flag) are read from the KERNARG segment at the point of use,
// through a laundered pointer, instead of through the by-value parameter: as parameters all ~75 dwords of the argument struct are
// live in SGPRs from the first instruction on, most of them get spilled into VGPR lanes, and every later use costs a v_readlane
// on the (binding) VALU pipe.  A scalar load of a few dwords right before a field's del^4 costs nothing there.
struct QdKargDyn { QdGeom G; QdTabs T; QdDynArgs P; };
struct QdKargOcn { QdGeom G; QdTabs T; QdOcnArgs P; };
template <typename A>
__device__ __forceinline__ const A QD_CONST* qd_kargs(unsigned off) {
    const char QD_CONST* p = (const char QD_CONST*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                             // opaque: the loads below cannot be hoisted to the kernel entry
    return (const A QD_CONST*)(p + off);
}
template <int TR> struct QdFast {
    static constexpr int RA = TR + 10;
    static constexpr int K = RA / QD_NW;                   // consecutive rows per wave
    static constexpr bool ok = (RA % QD_NW) == 0 && K >= 2;
};

// Tile geometry of the FAST path.  A tile owns global rows [o0, o1) (o0 = row0 + t*TR) and works on the RA = TR+10
// plane rows [p0, p0+RA).  Normally p0 = o0-5; next to a pole (or to the end of the row segment) the plane is
// shifted so that it never leaves [0, n) (or the segment +-5 rows): there are no out-of-domain rows, hence no
// masks.  An owned row needs 4 plane rows on each side except towards a pole, where the stencils stop anyway.

struct QdFastTile { int p0, o0, o1; bool fits, interior; };
template <int TR>
__device__ __forceinline__ QdFastTile qd_fast_tile(const QdGeom& G, int i0) {
    constexpr int RA = TR + 10;
    const int n = G.nlat, s0 = G.row0, s1 = G.row0 + G.nrows;
    const int lo = s0 == 0 ? 0 : s0 - 5, hi = s1 == n ? n : s1 + 5;
    QdFastTile t;
    t.o0 = i0; t.o1 = i0 + TR < s1 ? i0 + TR : s1;
    t.p0 = qd_clampi(i0 - 5, lo, hi - RA);
    t.fits = hi - lo >= RA && lo >= 0 && hi <= n && qd_lrow(G, t.p0) + RA <= G.lrows() && G.nlon >= 64;
    t.interior = t.p0 >= 2 && t.p0 + RA <= n - 2;
    return t;
}

struct QdPoleC { int n; const double* poleA; };
struct QdLapTabs { const double *A, *P, *Q; };      // lapA / lapP / lapQ of the cos-floor kind

// one spherical Laplacian over the wave's K rows.  X: the rows in registers, Xp: LDS plane for the rows that
// cross a wave boundary.
template <int TR, bool POLE>
__device__ __forceinline__ void qd_lap_rows(const double (&X)[QdFast<TR>::K], double (&L)[QdFast<TR>::K], double* __restrict__ Xp,
                                            int lane, int rho0, int g0, const QdLapTabs& LT, const QdPoleC& C) {
    constexpr int K = QdFast<TR>::K, RA = QdFast<TR>::RA, S = QD_S;
    // The row coefficients are re-read through the scalar cache in EVERY pass (three merged s_load_dwordxN, issued here so
    // the LDS exchange and the barrier hide their latency) instead of living in SGPRs across the kernel: with ~45 table
    // values plus the kernel's pointers the SGPR file overflowed and every use turned into a v_readlane from a spill VGPR
    // (measured: 4600 v_readlane in the kernel, 45 % of its VALU instructions).  The empty asm makes the row index opaque,
    // otherwise the identical loads of consecutive passes are merged again and stay live.
    int gq = g0;
    asm volatile("" : "+s"(gq));
    double sA[K + 2], sP[K], sQ[K];
#pragma unroll
    for (int k = 0; k < K + 2; ++k) sA[k] = qd_sload(LT.A, POLE ? qd_clampi(gq - 1 + k, 0, C.n - 1) : gq - 1 + k);
#pragma unroll
    for (int k = 0; k < K; ++k) { sP[k] = qd_sload(LT.P, gq + k); sQ[k] = qd_sload(LT.Q, gq + k); }
    Xp[(rho0 + 0) * S + lane] = X[0];
    Xp[(rho0 + 1) * S + lane] = X[1];
    Xp[(rho0 + K - 2) * S + lane] = X[K - 2];
    Xp[(rho0 + K - 1) * S + lane] = X[K - 1];
    __syncthreads();
    double Y[K + 4];                                          // rows rho0-2 .. rho0+K+1 (clamped at the plane edges: halo rows)
    Y[0] = Xp[qd_clampi(rho0 - 2, 0, RA - 1) * S + lane];
    Y[1] = Xp[qd_clampi(rho0 - 1, 0, RA - 1) * S + lane];
    Y[K + 2] = Xp[qd_clampi(rho0 + K, 0, RA - 1) * S + lane];
    Y[K + 3] = Xp[qd_clampi(rho0 + K + 1, 0, RA - 1) * S + lane];
#pragma unroll
    for (int k = 0; k < K; ++k) Y[k + 2] = X[k];
    const bool top = POLE && g0 == 0, bot = POLE && g0 + K == C.n;
    double pA[8];
    if (POLE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) pA[i] = qd_sload(C.poleA, i);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double c = Y[k + 2];
        double ahi = c, alo = Y[k], bhi = Y[k + 4], blo = c, Aa = sA[k], Ab = sA[k + 2];
        if (POLE) {
            // the plane of a pole tile starts at row 0 or ends at row n-1, so the pole rows can only be the first
            // two rows of wave 0 (types 0, 1) or the last two rows of the last wave (types 2, 3): wave-uniform tests
            if (k == 0 && top) { ahi = Y[k + 3]; alo = c; Aa = pA[0]; Ab = pA[1]; }
            if (k == 1 && top) { alo = Y[k + 1]; Aa = pA[2]; Ab = pA[3]; }
            if (k == K - 2 && bot) { bhi = Y[k + 3]; Aa = pA[4]; Ab = pA[5]; }
            if (k == K - 1 && bot) { bhi = c; blo = Y[k + 1]; Aa = pA[6]; Ab = pA[7]; }
        }
        const double Gb = Ab * (bhi - blo);
        const double Ga = Aa * (ahi - alo);
        // e - 2c in one fma: 2c is exact, so fma(-2, c, e) rounds exactly like (e - 2.0 * c) -- same bits, one instruction less
        const double d2 = __builtin_fma(-2.0, c, qd_east(c)) + qd_west(c);
        L[k] = sP[k] * (Gb - Ga) + sQ[k] * d2;
    }
}

// del^4 of the register field Bk (the wave's K rows) -> out.
template <int TR, bool POLE>
__device__ __forceinline__ void qd_del4_fast(const double (&Bk)[QdFast<TR>::K], double* __restrict__ Bp, double* __restrict__ Dp,
                                             double* __restrict__ out, const QdGeom& G, const QdFastTile& t, int jraw, int lane, int rho0,
                                             const QdLapTabs& LT, const double* k4row, double k4s, double dt,
                                             const QdPoleC& C, bool& bad) {
    constexpr int K = QdFast<TR>::K;
    const int g0 = t.p0 + rho0;
    double sK[K];
#pragma unroll
    for (int k = 0; k < K; ++k) sK[k] = k4row ? qd_sload(k4row, g0 + k) : k4s;
    double Dk[K], L2[K];
    qd_lap_rows<TR, POLE>(Bk, Dk, Bp, lane, rho0, g0, LT, C);
    qd_lap_rows<TR, POLE>(Dk, L2, Dp, lane, rho0, g0, LT, C);
    const bool col_ok = lane >= 3 && lane <= 60 && jraw < G.nlon;
    double* op = out + (size_t)qd_lrow(G, g0) * G.nlon + jraw;
    double val[K];
#pragma unroll
    for (int k = 0; k < K; ++k) val[k] = Bk[k] - (sK[k] * L2[k]) * dt;
    if (g0 >= t.o0 && g0 + K <= t.o1) {                       // wave-uniform: every row of this wave is owned (6 of the 8 waves)
#pragma unroll
        for (int k = 0; k < K; ++k) bad |= qd_nonfinite(val[k]);
        if (col_ok) {
#pragma unroll
            for (int k = 0; k < K; ++k) op[(size_t)k * G.nlon] = val[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int g = g0 + k;
            if (g >= t.o0 && g < t.o1) {
                bad |= qd_nonfinite(val[k]);
                if (col_ok) op[(size_t)k * G.nlon] = val[k];
            }
        }
    }
}

template <int TR>
__device__ __forceinline__ void qd_store_fast(const double (&Bk)[QdFast<TR>::K], double* __restrict__ out, const QdGeom& G,
                                              const QdFastTile& t, int jraw, int lane, int rho0) {
    constexpr int K = QdFast<TR>::K;
    const int g0 = t.p0 + rho0;
    const bool col_ok = lane >= 3 && lane <= 60 && jraw < G.nlon;
    double* op = out + (size_t)qd_lrow(G, g0) * G.nlon + jraw;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int g = g0 + k;
        if (g >= t.o0 && g < t.o1 && col_ok) op[(size_t)k * G.nlon] = Bk[k];
    }
}

// atmosphere.  Returns this thread's "saw a non-finite value" flag.
template <int TR, bool POLE>
__device__ __forceinline__ bool qd_dyn_fast(const QdGeom& G, const QdTabs& T, const QdDynArgs& P, const QdFastTile& t, int j0, double* lds) {
    constexpr int RA = QdFast<TR>::RA, K = QdFast<TR>::K, S = QD_S;
    double* Ap = lds;
    double* Bp = Ap + RA * S;
    double* Dp = Bp + RA * S;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rho0 = wv * K, g0 = t.p0 + rho0;
    const int jraw = j0 - 3 + lane, mlon = G.nlon, n = G.nlat;
    const int j = jraw < 0 ? jraw + mlon : (jraw >= mlon ? jraw - mlon : jraw);
    // ---- global loads of the momentum inputs (the wave's K rows), issued back to back; q and cloud are
    //      loaded later into the registers u' and v' free up (they are not needed before their own del^4)
    const unsigned o0 = (unsigned)qd_lrow(G, g0) * (unsigned)mlon + (unsigned)j;
    double ah[K], ru[K], rv[K], rf[K], rq[K], rc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const unsigned o = o0 + (unsigned)k * (unsigned)mlon;
        ah[k] = P.h[o]; ru[k] = P.u[o]; rv[k] = P.v[o]; rf[k] = P.fric[o];
    }
    // ---- row tables through the scalar cache
    const QdLapTabs LT{T.lapA[0], T.lapP[0], T.lapQ[0]};
    double c8[K], c9[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        c8[k] = qd_sload(P.primitive ? T.mom_px : T.mom_cu, g0 + k);
        c9[k] = qd_sload(P.primitive ? T.fcor : T.mom_cv, g0 + k);
    }
    // ---- h rows that cross a wave boundary
    Ap[rho0 * S + lane] = ah[0];
    Ap[(rho0 + K - 1) * S + lane] = ah[K - 1];
    __syncthreads();
    const double hs_edge = Ap[qd_clampi(rho0 - 1, 0, RA - 1) * S + lane];
    const double hn_edge = Ap[qd_clampi(rho0 + K, 0, RA - 1) * S + lane];
    // ---- momentum (dynamics.py:488-530); np.gradient is one-sided at both ends of both axes
    bool bad = false;
    const bool west_edge = j == 0, east_edge = j == mlon - 1;
    const double inv_lon = (west_edge || east_edge) ? P.inv_dlon : P.inv_2dlon;
#define QD_MOM_ROWS(PRIM)                                                                                   \
    _Pragma("unroll") for (int k = 0; k < K; ++k) {                                                         \
        const int g = g0 + k;                                                                               \
        const double hc = ah[k], hw = qd_west(hc), he = qd_east(hc);                                        \
        const double hs = k > 0 ? ah[k > 0 ? k - 1 : 0] : hs_edge, hn = k < K - 1 ? ah[k < K - 1 ? k + 1 : 0] : hn_edge; \
        const double dh_dlon = ((east_edge ? hc : he) - (west_edge ? hc : hw)) * inv_lon;                   \
        double dh_dlat = (hn - hs) * P.inv_2dlat;                                                           \
        if (POLE) { if (k == 0 && g == 0) dh_dlat = (hn - hc) * P.inv_dlat; if (k == K - 1 && g == n - 1) dh_dlat = (hc - hs) * P.inv_dlat; } \
        const double u0 = ru[k], v0 = rv[k], fr = rf[k];                                                    \
        if (PRIM) {                                                                                         \
            const double ux = u0 + (c8[k] * dh_dlon + c9[k] * v0 - fr * u0) * P.dt;                         \
            const double vx = v0 + (P.pgf_y * dh_dlat - c9[k] * u0 - fr * v0) * P.dt;                       \
            bad |= qd_nonfinite(ux) | qd_nonfinite(vx);                                                     \
            ru[k] = fmin(fmax(ux, -200.0), 200.0); rv[k] = fmin(fmax(vx, -200.0), 200.0);                   \
        } else {                                                                                            \
            const double ugx = c8[k] * dh_dlat, vgx = c9[k] * dh_dlon;                                      \
            bad |= qd_nonfinite(ugx) | qd_nonfinite(vgx);                                                   \
            const double u_g = fmin(fmax(ugx, -200.0), 200.0), v_g = fmin(fmax(vgx, -200.0), 200.0);        \
            const double ur = u0 * 0.8 + u_g * 0.2;                                                         \
            ru[k] = ur + (-fr * ur) * P.dt;                                                                 \
            const double vr = v0 * 0.8 + v_g * 0.2;                                                         \
            rv[k] = vr + (-fr * vr) * P.dt;                                                                 \
        }                                                                                                   \
    }
    if (P.primitive) { QD_MOM_ROWS(true) } else { QD_MOM_ROWS(false) }
#undef QD_MOM_ROWS
    // the EXACT path zeroes the momentum result in lanes 0 and 63 (no east/west neighbour): keep the planes identical
    if (lane == 0 || lane == 63) {
#pragma unroll
        for (int k = 0; k < K; ++k) { ru[k] = 0.0; rv[k] = 0.0; }
    }
    const QdPoleC C{n, T.lapPoleA[0]};
#define QD_FIELD(ARR, OUT, FI)                                                                              \
    {                                                                                                       \
        const QdDynArgs QD_CONST* Pk = qd_kargs<QdDynArgs>((unsigned)offsetof(QdKargDyn, P));               \
        double* const outp = Pk->OUT;                                                                       \
        if (Pk->skip[FI]) qd_store_fast<TR>(ARR, outp, G, t, jraw, lane, rho0);                             \
        else qd_del4_fast<TR, POLE>(ARR, Bp, Dp, outp, G, t, jraw, lane, rho0, LT, Pk->k4row[FI], Pk->k4s[FI], Pk->dt, C, bad); \
    }
    QD_FIELD(ru, uo, 0)
    {
        const double* const qp = qd_kargs<QdDynArgs>((unsigned)offsetof(QdKargDyn, P))->q;
#pragma unroll
        for (int k = 0; k < K; ++k) rq[k] = qp[o0 + (unsigned)k * (unsigned)mlon];
    }
    QD_FIELD(rv, vo, 1)
    {
        const double* const cp = qd_kargs<QdDynArgs>((unsigned)offsetof(QdKargDyn, P))->cloud;
#pragma unroll
        for (int k = 0; k < K; ++k) rc[k] = cp[o0 + (unsigned)k * (unsigned)mlon];
    }
    QD_FIELD(ah, ho, 2)
    QD_FIELD(rq, qo, 3)
    QD_FIELD(rc, co, 4)
#undef QD_FIELD
    return bad;
}

// ocean sub-step
template <int TR, bool POLE>
__device__ __forceinline__ bool qd_ocn_fast(const QdGeom& G, const QdTabs& T, const QdOcnArgs& P, const QdFastTile& t, int j0, double* lds) {
    constexpr int RA = QdFast<TR>::RA, K = QdFast<TR>::K, S = QD_S;
    double* Ap = lds;
    double* Bp = Ap + RA * S;
    double* Dp = Bp + RA * S;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rho0 = wv * K, g0 = t.p0 + rho0;
    const int jraw = j0 - 3 + lane, mlon = G.nlon, n = G.nlat;
    const int j = jraw < 0 ? jraw + mlon : (jraw >= mlon ? jraw - mlon : jraw);
    const unsigned o0 = (unsigned)qd_lrow(G, g0) * (unsigned)mlon + (unsigned)j;
    double ae[K], ru[K], rv[K], rtx[K], rty[K];
    unsigned lmask = 0u;                                    // land flags of the K cells, one bit each
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const unsigned o = o0 + (unsigned)k * (unsigned)mlon;
        ae[k] = P.eta[o]; ru[k] = P.uo[o]; rv[k] = P.vo[o]; rtx[k] = P.taux[o]; rty[k] = P.tauy[o];
        lmask |= (P.land[o] == 1 ? 1u : 0u) << k;
    }
    bool bad = false;
    if (P.eta_mean) {                                       // deferred eta -= mean; nan_to_num; clip of the previous sub-step
        const double em = *P.eta_mean, cap = P.eta_cap;
#pragma unroll
        for (int k = 0; k < K; ++k) { const double e = ae[k] - em; bad |= qd_nonfinite(e); ae[k] = fmin(fmax(e, -cap), cap); }
    }
    const QdLapTabs LT{T.lapA[1], T.lapP[1], T.lapQ[1]};
    double sF[K], sI[K], sX[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        sF[k] = qd_sload(T.fcor, g0 + k); sI[k] = qd_sload(T.ocn_igx, g0 + k); sX[k] = qd_sload(T.r_extra, g0 + k);
    }
    Ap[rho0 * S + lane] = ae[0];
    Ap[(rho0 + K - 1) * S + lane] = ae[K - 1];
    __syncthreads();
    double es_edge = Ap[qd_clampi(rho0 - 1, 0, RA - 1) * S + lane];
    double en_edge = Ap[qd_clampi(rho0 + K, 0, RA - 1) * S + lane];
    if (POLE && (g0 == 0 || g0 + K == n)) {
        // eta rows wrap across the poles (np.roll(axis=0), ocean.py:308): the row beyond a pole is the other pole's row
        // (loaded here, by the one wave that needs it, rather than carried in registers from the top of the kernel)
        double wr = P.eta[(unsigned)qd_lrow(G, g0 == 0 ? n - 1 : 0) * (unsigned)mlon + (unsigned)j];
        if (P.eta_mean) { const double e = wr - *P.eta_mean; bad |= qd_nonfinite(e); wr = fmin(fmax(e, -P.eta_cap), P.eta_cap); }
        if (g0 == 0) es_edge = wr; else en_edge = wr;
    }
    // ocean.py:306-336
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double ec = ae[k];
        const double es = k > 0 ? ae[k > 0 ? k - 1 : 0] : es_edge, en = k < K - 1 ? ae[k < K - 1 ? k + 1 : 0] : en_edge;
        const double f = sF[k];
        const double gx = ((qd_east(ec) - qd_west(ec)) * P.inv_2dlon) * sI[k];
        const double gy = ((en - es) * P.inv_2dlat) * P.inv_a;
        const double u0 = ru[k], v0 = rv[k];
        const double du = (f * v0 - P.g * gx + rtx[k] * P.inv_rhoH - P.r_bot * u0);
        const double dv = (-f * u0 - P.g * gy + rty[k] * P.inv_rhoH - P.r_bot * v0);
        double un = u0 + P.sub_dt * du, vn = v0 + P.sub_dt * dv;
        if ((lmask >> k) & 1u) { un = 0.0; vn = 0.0; }
        const double sx = P.sub_dt * sX[k];
        ru[k] = un - sx * un;
        rv[k] = vn - sx * vn;
    }
    if (lane == 0 || lane == 63) {
#pragma unroll
        for (int k = 0; k < K; ++k) { ru[k] = 0.0; rv[k] = 0.0; }
    }
    const QdPoleC C{n, T.lapPoleA[1]};
#define QD_FIELD(ARR, OUT, FI)                                                                              \
    {                                                                                                       \
        const QdOcnArgs QD_CONST* Pk = qd_kargs<QdOcnArgs>((unsigned)offsetof(QdKargOcn, P));               \
        double* const outp = Pk->OUT;                                                                       \
        if (Pk->skip[FI]) qd_store_fast<TR>(ARR, outp, G, t, jraw, lane, rho0);                             \
        else qd_del4_fast<TR, POLE>(ARR, Bp, Dp, outp, G, t, jraw, lane, rho0, LT, Pk->k4row[FI], Pk->k4s[FI], Pk->sub_dt, C, bad); \
    }
    QD_FIELD(ru, uo_out, 0)
    QD_FIELD(rv, vo_out, 1)
    QD_FIELD(ae, eta_out, 2)
#undef QD_FIELD
    return bad;
}

template <int TR>
__global__ void __launch_bounds__(QD_FBLOCK, 4)
k_dyn_hyper(QdGeom G, QdTabs T, QdDynArgs P) {
    extern __shared__ __align__(16) double lds[];
    int i0, j0;
    qd_tile_origin(P.ts.ntc, P.ts.ntr, TR, G.row0, i0, j0);
    if constexpr (QdFast<TR>::ok) {
        const QdFastTile t = qd_fast_tile<TR>(G, i0);
        if (P.fast && t.fits) {                                         // workgroup-uniform
            const bool bad = t.interior ? qd_dyn_fast<TR, false>(G, T, P, t, j0, lds) : qd_dyn_fast<TR, true>(G, T, P, t, j0, lds);
            if (!__syncthreads_or(bad)) return;
        }
    }
    qd_dyn_exact<TR>(G, T, P, i0, j0, lds);
}

template <int TR>
__global__ void __launch_bounds__(QD_FBLOCK, 4)
k_ocn_hyper(QdGeom G, QdTabs T, QdOcnArgs P) {
    extern __shared__ __align__(16) double lds[];
    int i0, j0;
    qd_tile_origin(P.ts.ntc, P.ts.ntr, TR, G.row0, i0, j0);
    if constexpr (QdFast<TR>::ok) {
        const QdFastTile t = qd_fast_tile<TR>(G, i0);
        // a pole tile also reads the other pole's eta row (np.roll): it must be on this slab
        const bool wrap_ok = t.interior || (qd_lrow(G, 0) < G.lrows() && qd_lrow(G, G.nlat - 1) < G.lrows());
        if (P.fast && t.fits && wrap_ok) {
            const bool bad = t.interior ? qd_ocn_fast<TR, false>(G, T, P, t, j0, lds) : qd_ocn_fast<TR, true>(G, T, P, t, j0, lds);
            if (!__syncthreads_or(bad)) return;
        }
    }
    qd_ocn_exact<TR>(G, T, P, i0, j0, lds);
}

// =========================================================================================
// host side
// =========================================================================================
static const int kTileRows[] = {6, 10, 14, 18, 22, 26, 30, 38};
static const int kNTileRows = sizeof(kTileRows) / sizeof(int);

static size_t qd_tile_lds_bytes(int tr) { return sizeof(double) * (QD_S * (size_t)(3 * (tr + 10)) + (size_t)10 * (tr + 10)); }

// Pick the instantiation whose tile count fills (256 CUs x resident workgroups) in the fewest
// rounds, each round weighted by the tile's work including the recomputed halo rows.
QdTileShape qd_pick_tile(const QdGeom& G) {
    const int ncu = 256;
    const size_t lds_cu = 160 * 1024;
    double best = 1e300;
    QdTileShape bs{kTileRows[0], QD_TC, 1, 1};
    const int ntc = (G.nlon + QD_TC - 1) / QD_TC;
    for (int t = 0; t < kNTileRows; ++t) {
        const int tr = kTileRows[t];
        const int slots = (int)std::min<size_t>(lds_cu / qd_tile_lds_bytes(tr), 2);   // 2 x 8 waves: the 128-VGPR cap of __launch_bounds__(512, 4)
        if (slots < 1) continue;
        const int ntr = (G.nrows + tr - 1) / tr;
        const long tiles = (long)ntr * ntc;
        const long cap = (long)ncu * slots;
        const long rounds = (tiles + cap - 1) / cap;
        // co-resident workgroups share the CU: a round costs (tile work) x (workgroups per CU in it)
        const double per_cu = std::max(1.0, std::ceil((double)tiles / (double)rounds / ncu));
        const double work = (double)(tr + 10) + 0.5 * (double)(tr + 8);
        const double cost = rounds * (per_cu * work + 12.0);
        if (cost < best) { best = cost; bs = QdTileShape{tr, QD_TC, ntr, ntc}; }
    }
    return bs;
}

static void qd_tile_init(qd_ctx* c) {
    if (c->tile.tr != 0) return;
    c->tile = qd_pick_tile(c->geo);
    if (c->tune.tile_tr > 0) {                                // QD_TILE_TR: tuning override (read at create)
        const int tr = c->tune.tile_tr;
        for (int t = 0; t < kNTileRows; ++t)
            if (kTileRows[t] == tr)
                c->tile = QdTileShape{tr, QD_TC, (c->geo.nrows + tr - 1) / tr, (c->geo.nlon + QD_TC - 1) / QD_TC};
    }
}

// one launch per row segment (a whole-globe handle has one; a polar band computing its wrap rows has two)
template <int TR> static void launch_dyn(qd_ctx* c, const QdDynArgs& P, int margin) {
    // the dynamic-LDS cap is a property of the (kernel, DEVICE) pair: one flag per device, not per process
    static bool once[QD_MAX_DEVICES] = {false};
    const int dv = c->desc.device >= 0 && c->desc.device < QD_MAX_DEVICES ? c->desc.device : 0;
    if (!once[dv]) { hipFuncSetAttribute((const void*)k_dyn_hyper<TR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)QdPl<TR>::lds_bytes); once[dv] = c->desc.device == dv; }
    QdDynArgs Q = P;
    QD_ROWS(c, margin, G, Q.ts.ntr = (G.nrows + TR - 1) / TR; hipLaunchKernelGGL(k_dyn_hyper<TR>, dim3(Q.ts.ntr * Q.ts.ntc), dim3(QD_FBLOCK),
                                             QdPl<TR>::lds_bytes, c->stream, G, c->tabs, Q));
}
template <int TR> static void launch_ocn(qd_ctx* c, const QdOcnArgs& P, int margin) {
    static bool once[QD_MAX_DEVICES] = {false};
    const int dv = c->desc.device >= 0 && c->desc.device < QD_MAX_DEVICES ? c->desc.device : 0;
    if (!once[dv]) { hipFuncSetAttribute((const void*)k_ocn_hyper<TR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)QdPl<TR>::lds_bytes); once[dv] = c->desc.device == dv; }
    QdOcnArgs Q = P;
    QD_ROWS(c, margin, G, Q.ts.ntr = (G.nrows + TR - 1) / TR; hipLaunchKernelGGL(k_ocn_hyper<TR>, dim3(Q.ts.ntr * Q.ts.ntc), dim3(QD_FBLOCK),
                                             QdPl<TR>::lds_bytes, c->stream, G, c->tabs, Q));
}

#define QD_DISPATCH_TR(tr, CALL)                                                                              \
    switch (tr) {                                                                                             \
        case 6: CALL(6); break; case 10: CALL(10); break; case 14: CALL(14); break; case 18: CALL(18); break; \
        case 22: CALL(22); break; case 26: CALL(26); break; case 30: CALL(30); break; case 38: CALL(38); break; \
        default: return qd_fail(c, "fused kernel: no instantiation for this tile height");                    \
    }

int qd_launch_dyn_hyper(qd_ctx* c, QdDynArgs& P, int margin) {
    if (qd_stream_ok(c, margin)) return qd_launch_dyn_stream(c, P, margin);
    qd_tile_init(c);
    P.ts = c->tile;
    P.fast = c->fused_fast != 0;
    QdScope sc(c, "k_dyn_hyper");
#define QD_CALL_DYN(N) launch_dyn<N>(c, P, margin)
    QD_DISPATCH_TR(P.ts.tr, QD_CALL_DYN)
    return 0;
}

int qd_launch_ocn_hyper(qd_ctx* c, QdOcnArgs& P, int margin) {
    if (qd_ocn_stream_ok(c, margin)) return qd_launch_ocn_stream(c, P, margin);
    qd_tile_init(c);
    P.ts = c->tile;
    P.fast = c->fused_fast != 0;
    QdScope sc(c, "k_ocn_hyper");
#define QD_CALL_OCN(N) launch_ocn<N>(c, P, margin)
    QD_DISPATCH_TR(P.ts.tr, QD_CALL_OCN)
    return 0;
}
