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
#include "qd_stream.h"
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <cstring>
#include <algorithm>


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

// qt_div without the guard: a non-finite quotient comes out as NaN instead of +-inf.  For the fast SST wave's departure point, where
// either sends the lane to the general form (|dx| < 1 fails for both).
__device__ __forceinline__ double qt_div_fin(double x, double c, double rc) {
    const double q0 = x * rc;
    const double e = __builtin_fma(-c, q0, x);
    return __builtin_fma(e, rc, q0);
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

// =========================================================================================
// row-streaming form of the same sub-step tail (default): no LDS, no barrier
// =========================================================================================
// Two wavefronts per strip of QS_R rows x 62 owned columns (lane l = column 62 cs - 1 + l; lanes 0 and 63 are halo columns),
// each marching down its rows like the waves of qd_stream.hip:
//   wave 0 (currents)  continuity + outlier filter: a 3-row register window of uo and vo (row g+1 arrives while row g is worked
//                      on); east / west neighbours by DPP; stores eta, uo, vo of row g; accumulates the area-weighted eta sum.
//   wave 1 (SST)       rows o0-2 .. o1+1: the departure point of row g+1 is computed and its four gather loads are issued while
//                      row g's gathered values are consumed (one row of software pipelining); nan_to_num(T1) lives in a 5-row
//                      register window; K_h lap(T1) + heating of row g-2 is stored.
// Same device expressions as k_ocn_tail above (qt_div, qt_departure's arithmetic, the row-table Laplacian): the two forms agree to
// the rounding of the eta sum.
#define QS_TC2 62
#define QS_TCF 60                 // k_ocn_tail_fast: lanes 2 .. 61 own a column (1 and 62: T1 for the Laplacian; 0 and 63: SST for their gather)

typedef unsigned int qt_u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t qt_rsrc;
__device__ __forceinline__ qt_rsrc qt_make_rsrc(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ double qt_ld(qt_rsrc r, unsigned row_elems, unsigned vo) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, row_elems * 8u, 0)); }
__device__ __forceinline__ void qt_st(qt_rsrc r, unsigned row_elems, unsigned vo, double v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(qt_u32x2, v), r, vo, row_elems * 8u, 0); }
__device__ __forceinline__ int qt_ld8(qt_rsrc r, unsigned row_elems, unsigned vo8) { return (int)__builtin_amdgcn_raw_buffer_load_b8(r, vo8, row_elems, 0); }

// uo'' / vo'' of one cell: stored -- or, with a fix list (QdTailArgs::fix_count), only noted when they differ from the uo' / vo' (uc, vc)
// the cell holds; bit patterns are compared, so a NaN that nan_to_num turned into 0 counts as changed.  The notes of a strip are staged
// in LDS by the strip's currents wave (the only wave of the workgroup that makes any: a plain counter, no atomic) and go to the list
// behind ONE returning global atomic per strip (qt_fix_flush) -- returning atomics on one word serialise at ~90 per us, and one per
// row with a hit put ten extra round trips on the critical path of exactly the strips that were the slow ones already (polar
// currents at the cap).  A strip with more notes than the stage holds appends the rest directly.
#define QT_FIX_STAGE 512                   // entries of 24 bytes
struct QtFixW { unsigned int* fixc; unsigned long long* fixl; unsigned long long* stage; unsigned int* nstage; };
// (FIX is a template parameter all the way up to the kernel: with the list as a run-time branch in the row loop the STORING form carried
//  the list's pointers through the loop -- 21 more spilled SGPRs -- and ran 8 us per step slower than before there was a list)
template <bool FIX>
__device__ __forceinline__ void qt_put_uv(const QtFixW& X, qt_rsrc UO, qt_rsrc VO, unsigned row_elems, unsigned vs,
                                          double u, double v, double uc, double vc) {
    if (!FIX) { qt_st(UO, row_elems, vs, u); qt_st(VO, row_elems, vs, v); return; }
    const unsigned long long ub = (unsigned long long)__double_as_longlong(u), vb = (unsigned long long)__double_as_longlong(v);
    const bool ch = vs != 0x80000000u && (ub != (unsigned long long)__double_as_longlong(uc) || vb != (unsigned long long)__double_as_longlong(vc));
    const unsigned long long chm = __builtin_amdgcn_ballot_w64(ch);
    if (chm != 0ull) {
        const int lane = (int)(threadIdx.x & 63u);
        const unsigned cnt = (unsigned)__popcll(chm);
        const unsigned base = __hip_atomic_load(X.nstage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);    // wave-uniform: this wave is the only writer
        const unsigned k = base + (unsigned)__popcll(chm & ((1ull << lane) - 1ull));
        const unsigned long long off = (unsigned long long)row_elems * 8ull + (unsigned long long)vs;
        if (base + cnt <= (unsigned)QT_FIX_STAGE) {
            if (ch) { X.stage[3u * k] = off; X.stage[3u * k + 1u] = ub; X.stage[3u * k + 2u] = vb; }
            if (lane == 0) __hip_atomic_store(X.nstage, base + cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
            const int leader = __ffsll((long long)chm) - 1;
            unsigned g = 0;
            if (lane == leader) g = __hip_atomic_fetch_add(X.fixc, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g = (unsigned)__shfl((int)g, leader, 64);
            if (ch) {
                unsigned long long* e = X.fixl + (size_t)3 * (g + (k - base));
                __hip_atomic_store(e, off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(e + 1, ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(e + 2, vb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
// end of the strip (all 64 lanes of the currents wave): the staged notes go to the list
__device__ __forceinline__ void qt_fix_flush(const QtFixW& X) {
    if (!X.fixc) return;
    const unsigned n = __hip_atomic_load(X.nstage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (n == 0u) return;
    const int lane = (int)(threadIdx.x & 63u);
    unsigned g = 0;
    if (lane == 0) g = __hip_atomic_fetch_add(X.fixc, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    g = (unsigned)__shfl((int)g, 0, 64);
    for (unsigned w = (unsigned)lane; w < 3u * n; w += 64u)
        __hip_atomic_store(X.fixl + (size_t)3 * g + w, X.stage[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) __hip_atomic_store(X.nstage, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

struct QtW { int n, m, lane, j, o0, o1; unsigned vo, vs, vo8, slab; bool own; int lbase, lrows, own0, own1, jbase; };      // lbase / lrows: the slab (band handles); own0 / own1: rows whose eta enters the sum

// element offset of global row g, clamped into the domain and into the slab (whole-globe handles: local row = global row; band
// handles: local row = g - lbase with the period-n wrap of their ring halo)
__device__ __forceinline__ unsigned qt_row(const QtW& W, int g) {
    int l = qd_clampi(g, 0, W.n - 1) - W.lbase;
    if (l < 0) l += W.n; else if (l >= W.n) l -= W.n;
    l = l < W.lrows ? l : W.lrows - 1;
    return (unsigned)l * (unsigned)W.m;
}
__device__ __forceinline__ unsigned qt_row_roll(const QtW& W, int g) { if (g < 0) g += W.n; else if (g >= W.n) g -= W.n; return qt_row(W, g); }

// ---- wave 0: continuity (ocean.py:365-374) + outlier filter and caps (ocean.py:409-434)
template <bool FIX = false>
__device__ __forceinline__ double qt_currents_wave(const QdTabs& T, const QdTailArgs& P, const QtW& W, const QtFixW& FX = QtFixW{nullptr, nullptr, nullptr, nullptr}) {
    const unsigned sb = W.slab;
    const qt_rsrc U = qt_make_rsrc(P.uo, sb), V = qt_make_rsrc(P.vo, sb), E = qt_make_rsrc(P.eta, sb), L = qt_make_rsrc(P.land, sb / 8u);
    const qt_rsrc EI = qt_make_rsrc(P.eta_in ? P.eta_in : P.eta, sb);       // (k_ocn_fused's sequential form reads eta' from a slab of its own)
    const qt_rsrc UO = qt_make_rsrc(P.uo_out, sb), VO = qt_make_rsrc(P.vo_out, sb);
    // np.roll rows: mean4 of the outlier filter reads row -1 as row n-1 and row n as row 0
    double us = qt_ld(U, qt_row_roll(W, W.o0 - 1), W.vo), vs = qt_ld(V, qt_row_roll(W, W.o0 - 1), W.vo);
    double uc = qt_ld(U, qt_row(W, W.o0), W.vo), vc = qt_ld(V, qt_row(W, W.o0), W.vo);
    double en = qt_ld(EI, qt_row(W, W.o0), W.vo); int ln = qt_ld8(L, qt_row(W, W.o0), W.vo8);
    double acc = 0.0;
    for (int g = W.o0; g < W.o1; ++g) {
        const unsigned rn = qt_row_roll(W, g + 1), r1 = qt_row(W, g + 1);
        const double un = qt_ld(U, rn, W.vo), vn = qt_ld(V, rn, W.vo);                 // row g+1 (in flight while row g is worked on)
        const double e0 = en; const int l0 = ln;
        en = qt_ld(EI, r1, W.vo); ln = qt_ld8(L, r1, W.vo8);
        // divergence (grid.py:41-88 through qt_div_point's expressions)
        const double dp = qt_div(qd_east(uc) - qd_west(uc), 2 * P.dlon, P.r_2dlon);
        double dq = 0.0;
        if (g != 0 && g != W.n - 1) {
            const double qn = vn * qd_sload(T.cos_raw, g + 1);
            const double qs = vs * qd_sload(T.cos_raw, g - 1);
            dq = qt_div(qn - qs, 2 * P.dlat, P.r_2dlat);
        }
        const double div = qd_sload(T.inv_acos6, g) * (dp + dq);
        double e = e0 + P.msdtH * div;
        const bool island = l0 == 1;
        if (island) e = 0.0;
        const unsigned r0 = qt_row(W, g);
        qt_st(E, r0, W.vs, e);
        acc += (W.own && g >= W.own0 && g < W.own1) ? e * (island ? 0.0 : qd_sload(T.warea, g)) : 0.0;      // a band's halo rows: not its sum
        // outliers + caps
        double u = qd_nn(uc), v = qd_nn(vc);
        const double cap = P.cap, s2 = u * u + v * v;
        // the lane neighbours are taken OUTSIDE the branch: a DPP move reads 0 from a lane that is not executing, and a spike
        // usually takes this branch alone in its wavefront
        const double ue = qd_east(uc), uw = qd_west(uc), ve = qd_east(vc), vw = qd_west(vc);
        if (!(s2 < 0.81 * (cap * cap))) {
            const double speed = sqrt(s2);
            if (P.mean4) {
                if (speed > cap) {
                    u = 0.25 * (qd_nn(un) + qd_nn(us) + qd_nn(ue) + qd_nn(uw));
                    v = 0.25 * (qd_nn(vn) + qd_nn(vs) + qd_nn(ve) + qd_nn(vw));
                }
                const double sp2 = sqrt(u * u + v * v);
                const double sc2 = (sp2 > cap) ? cap / (sp2 + 1e-12) : 1.0;
                u = u * sc2; v = v * sc2;
            } else {
                const double sc = (speed > cap) ? cap / (speed + 1e-12) : 1.0;
                u = u * sc; v = v * sc;
            }
        }
        qt_put_uv<FIX>(FX, UO, VO, r0, W.vs, u, v, uc, vc);
        us = uc; uc = un; vs = vc; vc = vn;
    }
    return acc;
}

// the departure point of (row gi, this lane's column) and the four corner loads of its bilinear gather, issued but not consumed
struct QtGather { double f00, f01, f10, f11, wr0, wr1, wc0, wc1; bool nan_coord; };
__device__ __forceinline__ QtGather qt_gather_issue(const QdGeom& G, const QdTabs& T, const QdTailArgs& P, int gi, int j, double u, double v) {
    const QdBilin b = qt_departure(G, gi, j, u, v, P.sub_dt, P.a * qd_sload(T.cos05, gi), qd_sload(T.ocn_igx, gi), P);
    const size_t r0 = (size_t)b.l0 * G.nlon, r1 = (size_t)b.l1 * G.nlon;
    QtGather q;
    q.f00 = P.Ts[r0 + b.c0]; q.f01 = P.Ts[r0 + b.c1]; q.f10 = P.Ts[r1 + b.c0]; q.f11 = P.Ts[r1 + b.c1];
    q.wr0 = b.wr0; q.wr1 = b.wr1; q.wc0 = b.wc0; q.wc1 = b.wc1; q.nan_coord = b.nan_coord;
    return q;
}
__device__ __forceinline__ double qt_gather_use(const QtGather& q) {      // qd_gather's corner order (scipy NI_GeometricTransform)
    double t = 0.0;
    t += q.f00 * q.wr0 * q.wc0;
    t += q.f01 * q.wr0 * q.wc1;
    t += q.f10 * q.wr1 * q.wc0;
    t += q.f11 * q.wr1 * q.wc1;
    return q.nan_coord ? 0.0 : t;
}

// qt_lap on a 5-row register window w[0..4] = nan_to_num(T1) rows i-2 .. i+2 (rows outside the domain hold anything)
__device__ __forceinline__ double qt_lap_win(const double (&w)[5], const QdTabs& T, int i, int n, double dphi, double dlam, double a) {
    const double cc = w[2], e = qd_east(cc), wst = qd_west(cc);
    if (i >= 2 && i <= n - 3) {
        const double Gb = qd_sload(T.lapA[1], i + 1) * (w[4] - cc);
        const double Ga = qd_sload(T.lapA[1], i - 1) * (cc - w[0]);
        const double d2 = (e - 2.0 * cc) + wst;
        return qd_sload(T.lapP[1], i) * (Gb - Ga) + qd_sload(T.lapQ[1], i) * d2;
    }
    // the two rows next to each pole: the literal reference form (ocean.py:100-117); F(r) = w[r - i + 2]
    const double* __restrict__ cosf = T.cos05;
    auto dphi_of = [&](int r) -> double {
        if (r == 0) return (w[1 - i + 2] - w[0 - i + 2]) / dphi;
        if (r == n - 1) return (w[n - 1 - i + 2] - w[n - 2 - i + 2]) / dphi;
        return (w[r + 1 - i + 2] - w[r - 1 - i + 2]) / (2.0 * dphi);
    };
    int ra, rb; double den;
    if (i == 0) { ra = 0; rb = 1; den = dphi; }
    else if (i == n - 1) { ra = n - 2; rb = n - 1; den = dphi; }
    else { ra = i - 1; rb = i + 1; den = 2.0 * dphi; }
    const double Ga = qd_sload(cosf, ra) * dphi_of(ra);
    const double Gb = qd_sload(cosf, rb) * dphi_of(rb);
    const double ci = qd_sload(cosf, i);
    const double term_phi = (1.0 / ci) * ((Gb - Ga) / den);
    const double d2 = ((e - 2.0 * cc) + wst) / (dlam * dlam);
    const double term_lam = d2 / (ci * ci);
    return (term_phi + term_lam) / (a * a);
}

// ---- wave 1: SST blend (ocean.py:380-382), K_h lap + heating (ocean.py:385-406, 440)
__device__ __forceinline__ void qt_sst_wave(const QdGeom& G, const QdTabs& T, const QdTailArgs& P, const QtW& W) {
    const unsigned sb = W.slab;
    const qt_rsrc U = qt_make_rsrc(P.uo, sb), V = qt_make_rsrc(P.vo, sb), S = qt_make_rsrc(P.Ts, sb), Q = qt_make_rsrc(P.qnet, sb);
    const qt_rsrc L = qt_make_rsrc(P.land, sb / 8u), I = qt_make_rsrc(P.ice, sb / 8u), SO = qt_make_rsrc(P.Ts_out, sb);
    const int n = W.n;
    const int t0 = W.o0 - 2 > 0 ? W.o0 - 2 : 0, t1 = W.o1 + 2 < n ? W.o1 + 2 : n;      // rows of T1 this strip needs
    double w[5] = {0, 0, 0, 0, 0};                                                       // nan_to_num(T1) rows g-4 .. g at step g
    // pipeline: at step g the gather of row g is consumed and the gather of row g+1 is issued
    double u1 = qt_ld(U, qt_row(W, t0 + 1), W.vo), v1 = qt_ld(V, qt_row(W, t0 + 1), W.vo);   // currents of row g+1
    double tc0 = qt_ld(S, qt_row(W, t0), W.vo), tc1 = qt_ld(S, qt_row(W, t0 + 1), W.vo);     // SST at the cell itself, rows g, g+1
    QtGather cur = qt_gather_issue(G, T, P, t0, W.j, qt_ld(U, qt_row(W, t0), W.vo), qt_ld(V, qt_row(W, t0), W.vo));
    for (int g = t0; g < t1 + 2; ++g) {                                                  // two drain steps: the window centre lags by 2
        double t1v = 0.0;
        if (g < t1) {
            // issue row g+1 first (its loads fly while row g is finished), rows g+2 of the plain inputs behind it
            const double un = qt_ld(U, qt_row(W, g + 2), W.vo), vn = qt_ld(V, qt_row(W, g + 2), W.vo), tn = qt_ld(S, qt_row(W, g + 2), W.vo);
            const QtGather nxt = qt_gather_issue(G, T, P, qd_clampi(g + 1, 0, n - 1), W.j, u1, v1);
            t1v = qd_nn((1.0 - P.alpha) * tc0 + P.alpha * qt_gather_use(cur));
            cur = nxt; u1 = un; v1 = vn; tc0 = tc1; tc1 = tn;
        }
        w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = w[4]; w[4] = t1v;
        const int i = g - 2;                                                             // window centre
        if (i >= W.o0 && i < W.o1) {
            const unsigned r = qt_row(W, i);
            double Tv = w[2];
            if (P.K_h > 0.0) Tv = Tv + P.sub_dt * P.K_h * qt_lap_win(w, T, i, n, P.dlat, P.dlon, P.a);
            if (P.use_q) {
                const double heat = qt_div(qt_ld(Q, r, W.vo), P.rcH, P.r_rcH);
                const bool ocean = qt_ld8(L, r, W.vo8) == 0;
                if (P.has_ice) {
                    const bool ic = qt_ld8(I, r, W.vo8) != 0;
                    if (ocean && !ic) Tv = Tv + P.sub_dt * heat;
                    if (P.ice_qfac > 0.0 && ocean && ic) Tv = Tv + P.sub_dt * P.ice_qfac * heat;
                } else if (ocean) Tv = Tv + P.sub_dt * heat;
            }
            qt_st(SO, r, W.vs, qd_nn(Tv));
        }
    }
}

// Diagnostic build only (-DQT_STAMPS): per-wave {entry, exit} s_memrealtime of the LAST launch, dumped to $QD_STAMPS_FILE
#ifdef QT_STAMPS
__device__ unsigned long long* qt_stamp_buf;
__device__ __forceinline__ void qt_stamp(int slot) {
    if ((threadIdx.x & 63) == 0) {
        const unsigned wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        qt_stamp_buf[(size_t)wid * 8 + slot] = __builtin_amdgcn_s_memrealtime();
        if (slot == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            qt_stamp_buf[(size_t)wid * 8 + 4] = hw; qt_stamp_buf[(size_t)wid * 8 + 5] = xcc;
            qt_stamp_buf[(size_t)wid * 8 + 1] = qt_stamp_buf[(size_t)wid * 8 + 0];
            qt_stamp_buf[(size_t)wid * 8 + 3] = qt_stamp_buf[(size_t)wid * 8 + 0];
        }
    }
}
#define QT_STAMP(k) qt_stamp(k)
#else
#define QT_STAMP(k) ((void)0)
#endif

// NS: SST waves per strip.  The SST wave is the long one of a strip (per-wave timelines, round 3: 22 us for its 12 rows against 12 us
// for the currents wave's 8); with NS = 2 each SST wave takes half of the strip's rows (+ its own four halo rows of T1).
template <int NS>
__global__ void __launch_bounds__(64 + 64 * NS)
k_ocn_tail_stream(QdGeom G, QdTabs T, QdTailArgs P) {
    const unsigned w = qd_xcd_chunk(blockIdx.x, gridDim.x);
    const int rs = (int)(w / (unsigned)P.ntc), cs = (int)(w % (unsigned)P.ntc);
    QtW W;
    W.n = G.nlat; W.m = G.nlon; W.lane = threadIdx.x & 63;
    const int jraw = cs * QS_TC2 - 1 + W.lane;
    W.j = jraw < 0 ? jraw + W.m : (jraw >= W.m ? jraw - W.m : jraw);
    W.own = W.lane >= 1 && W.lane <= QS_TC2 && jraw < W.m;
    W.vo = (unsigned)W.j * 8u; W.vo8 = (unsigned)W.j; W.vs = W.own ? (unsigned)jraw * 8u : 0x80000000u;
    W.slab = (unsigned)(G.lrows_ + QD_PAD_ROWS) * (unsigned)G.nlon * 8u;
    W.lbase = G.lbase; W.lrows = G.lrows_; W.own0 = P.own0; W.own1 = P.own1;
    W.o0 = G.row0 + rs * P.R;
    W.o1 = min(W.o0 + P.R, G.row0 + G.nrows);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    QT_STAMP(0);
    if (wv >= 1) {                                           // the SST waves are the long ones
        __builtin_amdgcn_s_setprio(2);
        if (NS == 2) {
            const int mid = W.o0 + (W.o1 - W.o0 + 1) / 2;
            if (wv == 1) W.o1 = mid; else W.o0 = mid;
            if (W.o0 >= W.o1) return;
        }
        qt_sst_wave(G, T, P, W);
        QT_STAMP(2);
        return;
    }
    double acc = qt_currents_wave(T, P, W);
    QT_STAMP(2);
    acc = qt_wave_sum(acc);
    if (W.lane == 0) __hip_atomic_store(P.partial + w, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // coherent: the finisher may read it
    if (P.acc) {                                             // eta mean inside this launch: the last workgroup to arrive finishes it
        const bool last = W.lane == 0 && qd_acc_arrive(P.acc, w, gridDim.x, acc);
        if (__builtin_amdgcn_ballot_w64(last) != 0ull) {
            double m = qd_acc_finish(P.acc, P.partial, (int)gridDim.x, P.wsum);
            if (P.pf.pbox) m = qp_fold_sum(P.pf, m);          // bands over the peer exchange: the sum over the ranks, here and now
            if (W.lane == 0) *P.mean_out = m;
        }
    }
}

// ---- the streaming form again, slimmed for the rows where nothing special happens (round 3b) -------------------------------------
// What the counters said about k_ocn_tail_stream (profiles/README.md, round 3b): its waves keep the SIMDs issuing ~70 % of the
// time (176 VALU + 141 SALU instructions per SST row: row clamps, pole branches, 64-bit gather addresses, spilled SGPRs coming back
// through v_readlane) AND every row of either wave waits a full memory round trip -- the currents wave loads row g+1 while it
// works on row g, the SST wave cannot issue a gather before the currents it departs with have arrived.  k_ocn_tail_fast:
//   * the rows are cut into a short strip at each pole (Rp rows: the forms above, with their clamps, folds and pole stencils) and
//     even strips in between, whose waves never see a pole: plain row offsets, the interior Laplacian, no np.roll rows;
//   * every load is issued TWO row steps before its use into one of two fixed register slots, refilled at the end of the step
//     that consumed it (qd_stream.h has the reasons for the slots, the scheduling barrier and qs_own);
//   * the SST wave's gather does not go to memory: a sub-step moves water by ~0.003 of a cell (3 m/s x 23 s against 28 km), so the
//     four corners of the bilinear gather are among the nine cells around (g, j) -- rows g-1, g, g+1, which the stream holds in
//     registers, columns by DPP lane shifts.  Same arithmetic on the same operands in the same order (the corner values are
//     selected, the weights computed as qt_departure computes them; of scipy's fold only "+- (n_lon - 1)" can happen, the clamps
//     are identities), so the result is the memory gather's bit for bit.  Waves that hold column 0 or n_lon - 1 (period
//     n_lon - 1: the fold lands two lanes away, the clamp c1 = c0 on the lane itself) pick the corners by ds_bpermute.
//   * a lane whose departure point is a cell or more away, or NaN, raises a flag; the wave finishes its strip (garbage in, garbage
//     out) and then runs the general form over the same strip, which overwrites everything it stored.
struct QtPart { int ps, pn, nmid, mid0, mid1; };
__host__ __device__ inline QtPart qt_partition(int lo, int hi, int n, int R, int Rp) {
    QtPart q;
    const int rows = hi - lo;
    q.ps = lo == 0 ? (Rp < rows ? Rp : rows) : 0;
    q.pn = hi == n ? (Rp < rows - q.ps ? Rp : rows - q.ps) : 0;
    q.mid0 = lo + q.ps; q.mid1 = hi - q.pn;
    q.nmid = (q.mid1 - q.mid0 + R - 1) / R;
    return q;
}
__host__ __device__ inline int qt_part_strips(const QtPart& q) { return (q.ps > 0) + q.nmid + (q.pn > 0); }

// Per-row coefficients of the two fast waves, packed so that a row step needs ONE row of ONE table, fetched by vector loads with a
// wave-uniform address (every lane gets the same 16 bytes) in the same two-steps-ahead slots as the fields.  As scalar loads from
// eight tables they were five or six exposed s_waitcnt lgkmcnt(0) a step (SMEM returns out of order: any use waits for all) and
// ten SGPRs of table pointers in kernels that spill SGPRs by the hundred.
//   [0] a cos05[g]   [1] 1/(a cos05[g])   [2] lapA[g-1]   [3] lapA[g-3]   [4] lapP[g-2]   [5] lapQ[g-2]      (SST wave, T1 row g / output row g-2)
//   [8] cos[g+1]     [9] cos[g-1]         [10] 1/(a cos6[g])   [11] max(cos[g], 0)                            (currents wave, row g)
__global__ void k_tail_tab(QdTabs T, int n, double a, double* __restrict__ tab) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    auto cl = [n](int r) { return r < 0 ? 0 : (r > n - 1 ? n - 1 : r); };
    double* o = tab + (size_t)g * 16;
    o[0] = a * T.cos05[g]; o[1] = T.ocn_igx[g]; o[2] = T.lapA[1][cl(g - 1)]; o[3] = T.lapA[1][cl(g - 3)];
    o[4] = T.lapP[1][cl(g - 2)]; o[5] = T.lapQ[1][cl(g - 2)]; o[6] = 0.0; o[7] = 0.0;
    o[8] = T.cos_raw[cl(g + 1)]; o[9] = T.cos_raw[cl(g - 1)]; o[10] = T.inv_acos6[g]; o[11] = T.warea[g];
    o[12] = 0.0; o[13] = 0.0; o[14] = 0.0; o[15] = 0.0;
}
typedef unsigned int qt_u32x4 __attribute__((ext_vector_type(4)));
typedef double qt_f64x2 __attribute__((ext_vector_type(2)));
// two doubles of table row `row16` (element offset of the row = 16 g), the same for every lane
__device__ __forceinline__ qt_f64x2 qt_ldk(qt_rsrc K, unsigned row16, unsigned pair) {
    return __builtin_bit_cast(qt_f64x2, __builtin_amdgcn_raw_buffer_load_b128(K, 0u, row16 * 8u + pair * 16u, 0));
}

// a wave-uniform constant parked in VGPRs: left to itself the compiler keeps such values as kernarg state and, out of SGPRs,
// re-loads the argument block inside the row loop (56 dwords of s_load per step, each an exposed lgkmcnt(0)) or spills them to
// lanes of a VGPR (v_readlane per use); VGPRs are not scarce in these kernels
__device__ __forceinline__ double qt_vreg(double x) {
    double y;
    asm("v_mov_b64 %0, %1" : "=v"(y) : "s"(x));
    return y;
}

// k_ocn_fused (below): uo', vo', eta' rows arrive through an LDS ring instead of global memory.  prog[0..2]: rows produced so far by the
// uo / vo / eta waves (= first row not yet there), prog[3], prog[4]: first row the currents / the SST wave still needs.
#define QFU_NR 16                 // rows of a ring (power of two)
struct QfuRing { double* u; double* v; double* e; int* prog; };
__device__ __forceinline__ void qfu_wait_gt(const int* p, int row) {          // until *p > row (LDS poll; all waves of the workgroup are resident)
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= row) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ double qfu_ring_ld(const double* plane, int row, int lane) { return plane[(row & (QFU_NR - 1)) * 64 + lane]; }

struct QtCurSlot { double u, v, e; int l; qt_f64x2 k0, k1; };
struct QtCurK { double dlon2, r_2dlon, dlat2, r_2dlat, msdtH, cap, cap81; int mean4; QtFixW fx = QtFixW{nullptr, nullptr, nullptr, nullptr}; };

// one row of the continuity + caps wave away from the poles; `ro`: element offset of row g in the slab
// RING: u, v of row g + 1 and eta of row g come from the LDS ring of k_ocn_fused (R), everything else as before
template <bool RING = false, bool FIX = false>
__device__ __forceinline__ void qt_cur_fast_step(const QtCurK& P, const QtW& W, qt_rsrc U, qt_rsrc V, qt_rsrc E, qt_rsrc L,
                                                 qt_rsrc UO, qt_rsrc VO, qt_rsrc KT, double& us, double& uc, double& vs, double& vc, double& acc,
                                                 QtCurSlot& sl, int g, unsigned ro, const QfuRing* R = nullptr) {
    double un, vn, e0;
    if (RING) {
        qfu_wait_gt(R->prog + 0, g + 1); qfu_wait_gt(R->prog + 1, g + 1); qfu_wait_gt(R->prog + 2, g);
        un = qfu_ring_ld(R->u, g + 1, W.lane); vn = qfu_ring_ld(R->v, g + 1, W.lane); e0 = qfu_ring_ld(R->e, g, W.lane);
    } else { un = qs_own(sl.u); vn = qs_own(sl.v); e0 = sl.e; }
    const bool island = sl.l == 1;
    const double ue = qd_east(uc), uw = qd_west(uc), ve = qd_east(vc), vw = qd_west(vc);
    const double dp = qt_div(ue - uw, P.dlon2, P.r_2dlon);
    const double qn = vn * sl.k0.x;
    const double qs = vs * sl.k0.y;
    const double dq = qt_div(qn - qs, P.dlat2, P.r_2dlat);
    const double div = sl.k1.x * (dp + dq);
    double e = e0 + P.msdtH * div;
    if (island) e = 0.0;
    qt_st(E, ro, W.vs, e);
    acc += (W.own && g >= W.own0 && g < W.own1) ? e * (island ? 0.0 : sl.k1.y) : 0.0;
    double u = qd_nn(uc), v = qd_nn(vc);
    const double cap = P.cap, s2 = u * u + v * v;
    if (!(s2 < P.cap81)) {
        const double speed = sqrt(s2);
        if (P.mean4) {
            if (speed > cap) {
                u = 0.25 * (qd_nn(un) + qd_nn(us) + qd_nn(ue) + qd_nn(uw));
                v = 0.25 * (qd_nn(vn) + qd_nn(vs) + qd_nn(ve) + qd_nn(vw));
            }
            const double sp2 = sqrt(u * u + v * v);
            const double sc2 = (sp2 > cap) ? cap / (sp2 + 1e-12) : 1.0;
            u = u * sc2; v = v * sc2;
        } else {
            const double sc = (speed > cap) ? cap / (speed + 1e-12) : 1.0;
            u = u * sc; v = v * sc;
        }
    }
    qt_put_uv<FIX>(P.fx, UO, VO, ro, W.vs, u, v, uc, vc);
    us = uc; uc = un; vs = vc; vc = vn;
    if (RING) { if (W.lane == 0) __hip_atomic_store(R->prog + 3, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }     // rows <= g are done with
    __builtin_amdgcn_sched_barrier(0);
    const unsigned r2 = ro + 2u * (unsigned)W.m;              // what step g + 2 consumes: u, v of row g + 3, eta and land of row g + 2
    if (!RING) { sl.u = qt_ld(U, r2 + (unsigned)W.m, W.vo); sl.v = qt_ld(V, r2 + (unsigned)W.m, W.vo); sl.e = qt_ld(E, r2, W.vo); }
    sl.l = qt_ld8(L, r2, W.vo8);
    sl.k0 = qt_ldk(KT, 16u * (unsigned)(g + 2), 4u); sl.k1 = qt_ldk(KT, 16u * (unsigned)(g + 2), 5u);
    __builtin_amdgcn_sched_barrier(0);                       // (else the next step's arithmetic is scheduled in front of these loads)
}

template <bool FIX = false>
__device__ __forceinline__ double qt_currents_fast(const QdTailArgs& P, const QtW& W, const QtFixW& FX = QtFixW{nullptr, nullptr, nullptr, nullptr}) {
    const unsigned sb = W.slab, m = (unsigned)W.m;
    const qt_rsrc U = qt_make_rsrc(P.uo, sb), V = qt_make_rsrc(P.vo, sb), E = qt_make_rsrc(P.eta, sb), L = qt_make_rsrc(P.land, sb / 8u);
    const qt_rsrc UO = qt_make_rsrc(P.uo_out, sb), VO = qt_make_rsrc(P.vo_out, sb), KT = qt_make_rsrc(P.tab, (unsigned)W.n * 128u);
    unsigned ro = (unsigned)(W.o0 - W.lbase) * m;
    double us = qt_ld(U, ro - m, W.vo), vs = qt_ld(V, ro - m, W.vo);
    double uc = qt_ld(U, ro, W.vo), vc = qt_ld(V, ro, W.vo);
    double acc = 0.0;
    QtCurK K;
    K.dlon2 = qt_vreg(2 * P.dlon); K.r_2dlon = qt_vreg(P.r_2dlon); K.dlat2 = qt_vreg(2 * P.dlat); K.r_2dlat = qt_vreg(P.r_2dlat);
    K.msdtH = qt_vreg(P.msdtH); K.cap = qt_vreg(P.cap); K.cap81 = qt_vreg(0.81 * (P.cap * P.cap)); K.mean4 = P.mean4;
    if (FIX) K.fx = FX;
    QtCurSlot a, b;
    a.u = qt_ld(U, ro + m, W.vo); a.v = qt_ld(V, ro + m, W.vo); a.e = qt_ld(E, ro, W.vo); a.l = qt_ld8(L, ro, W.vo8);
    a.k0 = qt_ldk(KT, 16u * (unsigned)W.o0, 4u); a.k1 = qt_ldk(KT, 16u * (unsigned)W.o0, 5u);
    b.u = qt_ld(U, ro + 2u * m, W.vo); b.v = qt_ld(V, ro + 2u * m, W.vo); b.e = qt_ld(E, ro + m, W.vo); b.l = qt_ld8(L, ro + m, W.vo8);
    b.k0 = qt_ldk(KT, 16u * (unsigned)(W.o0 + 1), 4u); b.k1 = qt_ldk(KT, 16u * (unsigned)(W.o0 + 1), 5u);
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // the loop is entered with nothing in flight (exact counts inside: qd_stream.h)
    int g = W.o0;
    for (; g + 1 < W.o1; g += 2) {
        qt_cur_fast_step<false, FIX>(K, W, U, V, E, L, UO, VO, KT, us, uc, vs, vc, acc, a, g, ro);
        qt_cur_fast_step<false, FIX>(K, W, U, V, E, L, UO, VO, KT, us, uc, vs, vc, acc, b, g + 1, ro + m);
        ro += 2u * m;
    }
    if (g < W.o1) qt_cur_fast_step<false, FIX>(K, W, U, V, E, L, UO, VO, KT, us, uc, vs, vc, acc, a, g, ro);
    return acc;
}

struct QtSstSlot { double u, v, s, q; int l, ic; qt_f64x2 k0, k1, k2; };
struct QtSstK { double a, r_a, dt, dlon, r_dlon, dlat, r_dlat, alpha, om_alpha, dtK, rcH, r_rcH, dtq; int has_ice, qfac_on; };

__device__ __forceinline__ double qt_shfl(double x, int src_lane) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(x));
    return __hiloint2double(hi, lo);
}

// One row step of the SST wave away from the poles.  g: the row whose T1 is computed; ro: element offset of row g.  OUT: row g - 2 is
// finished (K_h lap of the window + heating) and stored.  Returns through `bad` whether a lane needed the general gather.
template <bool EDGE, bool OUT, bool RING = false>
__device__ __forceinline__ void qt_sst_fast_step(const QtSstK& K, const QtW& W, qt_rsrc U, qt_rsrc V, qt_rsrc S, qt_rsrc Q,
                                                 qt_rsrc L, qt_rsrc I, qt_rsrc SO, qt_rsrc KT, double& Tm, double& T0, double& Wm, double& W0, double& Em,
                                                 double& E0, double& w0, double& w1, double& w2, double& w3, double& w4, bool& bad,
                                                 QtSstSlot& sl, int g, unsigned ro, const QfuRing* R = nullptr) {
    const double Tp = qs_own(sl.s);                           // SST row g + 1: lives on as row g of the next step
    double Wp = 0.0, Ep = 0.0;
    if (!EDGE) { Wp = qd_west(Tp); Ep = qd_east(Tp); }
    double u, v;
    if (RING) {                                               // k_ocn_fused: the currents of row g come out of the LDS ring
        qfu_wait_gt(R->prog + 0, g); qfu_wait_gt(R->prog + 1, g);
        u = qfu_ring_ld(R->u, g, W.lane); v = qfu_ring_ld(R->v, g, W.lane);
        if (W.lane == 0) __hip_atomic_store(R->prog + 4, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else { u = sl.u; v = sl.v; }
    // qt_departure's arithmetic, piece by piece
    const double acos = sl.k0.x, r_acos = sl.k0.y;
    const double dl = qt_div_fin(u * K.dt, acos, r_acos);
    const double dph = qt_div_fin(v * K.dt, K.a, K.r_a);
    const double dx = qt_div_fin(dl, K.dlon, K.r_dlon);
    const double dy = qt_div_fin(dph, K.dlat, K.r_dlat);
    const bool near = fabs(dx) < 1.0 && fabs(dy) < 1.0;       // (false for NaN: a non-finite u or v, or an overflow on the way)
    bad = bad || (W.lane >= 1 && W.lane <= 62 && !near);      // lanes 0 and 63 only lend their SST: what they compute is never read
    const double r = (double)g - dy;                          // 1 <= g <= n - 2 and |dy| < 1: inside [0, n - 1], no fold, no clamp
    double cc = (double)W.j - dx;
    if (EDGE) {                                               // qd_fold with |dx| < 1: trunc(.) is 0 below the range and 1 above it
        const double sz = (double)(W.m - 1);
        cc = cc < 0.0 ? cc + sz * 1.0 : (cc > sz ? cc - sz * 1.0 : cc);
    }
    const double r0f = floor(r), c0f = floor(cc);
    const double tr = r - r0f, tc = cc - c0f;
    const bool pr = r0f < (double)g;                          // r0 = g - 1 (else g)
    double X_m, X_0, X_p, Y_m, Y_0, Y_p;
    if (EDGE) {
        const int c0 = (int)c0f;
        const int c1 = c0 + 1 < W.m ? c0 + 1 : W.m - 1;
        int l0 = c0 - W.jbase, l1 = c1 - W.jbase;
        l0 = l0 < 0 ? l0 + W.m : (l0 >= W.m ? l0 - W.m : l0);
        l1 = l1 < 0 ? l1 + W.m : (l1 >= W.m ? l1 - W.m : l1);
        l0 &= 63; l1 &= 63;                                   // lanes 1 .. 62 stay inside the wave when |dx| < 1
        X_m = qt_shfl(Tm, l0); X_0 = qt_shfl(T0, l0); X_p = qt_shfl(Tp, l0);
        Y_m = qt_shfl(Tm, l1); Y_0 = qt_shfl(T0, l1); Y_p = qt_shfl(Tp, l1);
    } else {
        const bool pc = c0f < (double)W.j;                    // c0 = j - 1 (else j); c1 = c0 + 1: no clamp away from the two edge columns
        X_m = pc ? Wm : Tm; X_0 = pc ? W0 : T0; X_p = pc ? Wp : Tp;
        Y_m = pc ? Tm : Em; Y_0 = pc ? T0 : E0; Y_p = pc ? Tp : Ep;
    }
    QtGather q;
    q.f00 = pr ? X_m : X_0; q.f10 = pr ? X_0 : X_p; q.f01 = pr ? Y_m : Y_0; q.f11 = pr ? Y_0 : Y_p;
    q.wr0 = 1.0 - tr; q.wr1 = tr; q.wc0 = 1.0 - tc; q.wc1 = tc; q.nan_coord = false;
    const double t1v = qd_nn(K.om_alpha * T0 + K.alpha * qt_gather_use(q));
    Tm = T0; T0 = Tp;
    if (!EDGE) { Wm = W0; W0 = Wp; Em = E0; E0 = Ep; }
    w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = t1v;
    if (OUT) {
        // window centre i = g - 2: 2 <= i <= n - 3, the interior form of qt_lap_win
        const double e = qd_east(w2), wst = qd_west(w2);
        const double Gb = sl.k1.x * (w4 - w2);                // lapA[i + 1]
        const double Ga = sl.k1.y * (w2 - w0);                // lapA[i - 1]
        const double d2 = (e - 2.0 * w2) + wst;
        const double lap = sl.k2.x * (Gb - Ga) + sl.k2.y * d2;   // lapP[i], lapQ[i]
        double Tv = w2 + K.dtK * lap;
        const double heat = qt_div(sl.q, K.rcH, K.r_rcH);
        const bool ocean = sl.l == 0;
        const bool ic = K.has_ice && sl.ic != 0;
        if (ocean && !ic) Tv = Tv + K.dt * heat;
        if (K.qfac_on && ocean && ic) Tv = Tv + K.dtq * heat;
        qt_st(SO, ro - 2u * (unsigned)W.m, W.vs, qd_nn(Tv));
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned r2 = ro + 2u * (unsigned)W.m;              // what step g + 2 consumes
    if (!RING) { sl.u = qt_ld(U, r2, W.vo); sl.v = qt_ld(V, r2, W.vo); }
    sl.s = qt_ld(S, r2 + (unsigned)W.m, W.vo);
    sl.q = qt_ld(Q, ro, W.vo); sl.l = qt_ld8(L, ro, W.vo8); sl.ic = qt_ld8(I, ro, W.vo8);
    sl.k0 = qt_ldk(KT, 16u * (unsigned)(g + 2), 0u); sl.k1 = qt_ldk(KT, 16u * (unsigned)(g + 2), 1u); sl.k2 = qt_ldk(KT, 16u * (unsigned)(g + 2), 2u);
    __builtin_amdgcn_sched_barrier(0);                       // (else the next step's arithmetic is scheduled in front of these loads)
}

// true: a lane needed the general gather (the caller runs the general wave over the strip)
template <bool EDGE>
__device__ __forceinline__ bool qt_sst_fast(const QdTailArgs& P, const QtW& W) {
    const unsigned sb = W.slab, m = (unsigned)W.m;
    const qt_rsrc U = qt_make_rsrc(P.uo, sb), V = qt_make_rsrc(P.vo, sb), S = qt_make_rsrc(P.Ts, sb), Q = qt_make_rsrc(P.qnet, sb);
    const qt_rsrc L = qt_make_rsrc(P.land, sb / 8u), I = qt_make_rsrc(P.ice, sb / 8u), SO = qt_make_rsrc(P.Ts_out, sb);
    const qt_rsrc KT = qt_make_rsrc(P.tab, (unsigned)W.n * 128u);
    QtSstK K;
    K.a = qt_vreg(P.a); K.r_a = qt_vreg(P.r_a); K.dt = qt_vreg(P.sub_dt); K.dlon = qt_vreg(P.dlon); K.r_dlon = qt_vreg(P.r_dlon);
    K.dlat = qt_vreg(P.dlat); K.r_dlat = qt_vreg(P.r_dlat); K.alpha = qt_vreg(P.alpha); K.om_alpha = qt_vreg(1.0 - P.alpha);
    K.dtK = qt_vreg(P.sub_dt * P.K_h); K.rcH = qt_vreg(P.rcH); K.r_rcH = qt_vreg(P.r_rcH); K.dtq = qt_vreg(P.sub_dt * P.ice_qfac);
    K.has_ice = P.has_ice; K.qfac_on = P.ice_qfac > 0.0 ? 1 : 0;
    const int t0 = W.o0 - 2;                                 // first row of T1 this strip needs (last: o1 + 1)
    unsigned ro = (unsigned)(t0 - W.lbase) * m;
    double Tm = qt_ld(S, ro - m, W.vo), T0 = qt_ld(S, ro, W.vo);
    QtSstSlot a, b;
    a.u = qt_ld(U, ro, W.vo); a.v = qt_ld(V, ro, W.vo); a.s = qt_ld(S, ro + m, W.vo);
    a.k0 = qt_ldk(KT, 16u * (unsigned)t0, 0u); a.k1 = qt_ldk(KT, 16u * (unsigned)t0, 1u); a.k2 = qt_ldk(KT, 16u * (unsigned)t0, 2u);
    b.u = qt_ld(U, ro + m, W.vo); b.v = qt_ld(V, ro + m, W.vo); b.s = qt_ld(S, ro + 2u * m, W.vo);
    b.k0 = qt_ldk(KT, 16u * (unsigned)(t0 + 1), 0u); b.k1 = qt_ldk(KT, 16u * (unsigned)(t0 + 1), 1u); b.k2 = qt_ldk(KT, 16u * (unsigned)(t0 + 1), 2u);
    a.q = b.q = 0.0; a.l = b.l = 0; a.ic = b.ic = 0;         // (the first four steps store nothing)
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // the loop is entered with nothing in flight (exact counts inside: qd_stream.h)
    double Wm = 0.0, W0 = 0.0, Em = 0.0, E0 = 0.0;
    if (!EDGE) { Wm = qd_west(Tm); W0 = qd_west(T0); Em = qd_east(Tm); E0 = qd_east(T0); }
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0, w4 = 0.0;
    bool bad = false;
#define QT_FS(OUTF, SL, GG, RO) qt_sst_fast_step<EDGE, OUTF>(K, W, U, V, S, Q, L, I, SO, KT, Tm, T0, Wm, W0, Em, E0, w0, w1, w2, w3, w4, bad, SL, GG, RO)
    int g = t0;
    QT_FS(false, a, g, ro); QT_FS(false, b, g + 1, ro + m); QT_FS(false, a, g + 2, ro + 2u * m); QT_FS(false, b, g + 3, ro + 3u * m);
    g += 4; ro += 4u * m;
    const int gend = W.o1 + 2;
    for (; g + 1 < gend; g += 2) { QT_FS(true, a, g, ro); QT_FS(true, b, g + 1, ro + m); ro += 2u * m; }
    if (g < gend) QT_FS(true, a, g, ro);
#undef QT_FS
    return __builtin_amdgcn_ballot_w64(bad) != 0ull;
}

template <bool FIX>
__global__ void __launch_bounds__(128)
k_ocn_tail_fast(QdGeom G, QdTabs T, QdTailArgs P) {
    const unsigned w = qd_xcd_chunk(blockIdx.x, gridDim.x);
    const int rs = (int)(w / (unsigned)P.ntc), cs = (int)(w % (unsigned)P.ntc);
    QtW W;
    W.n = G.nlat; W.m = G.nlon; W.lane = threadIdx.x & 63;
    W.jbase = cs * QS_TCF - 2;
    const int jraw = W.jbase + W.lane;
    W.j = jraw < 0 ? jraw + W.m : (jraw >= W.m ? jraw - W.m : jraw);
    W.own = W.lane >= 2 && W.lane <= QS_TCF + 1 && jraw < W.m;
    W.vo = (unsigned)W.j * 8u; W.vo8 = (unsigned)W.j; W.vs = W.own ? (unsigned)jraw * 8u : 0x80000000u;
    W.slab = (unsigned)(G.lrows_ + QD_PAD_ROWS) * (unsigned)G.nlon * 8u;
    W.lbase = G.lbase; W.lrows = G.lrows_; W.own0 = P.own0; W.own1 = P.own1;
    const int lo = G.row0, hi = G.row0 + G.nrows;
    const QtPart q = qt_partition(lo, hi, G.nlat, P.R, P.Rp);
    const int k = rs - (q.ps > 0 ? 1 : 0);                   // index among the middle strips
    if (q.ps > 0 && rs == 0) { W.o0 = lo; W.o1 = lo + q.ps; }
    else if (k < q.nmid) {                                   // even cut of the middle rows
        const int M = q.mid1 - q.mid0;
        W.o0 = q.mid0 + (int)(((long long)k * M) / q.nmid);
        W.o1 = q.mid0 + (int)(((long long)(k + 1) * M) / q.nmid);
    } else { W.o0 = q.mid1; W.o1 = hi; }
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // rows a fast wave touches lie inside the slab without a wrap, and away from the poles
    const bool plain = (P.flags & 1) == 0 && W.o0 - 3 - W.lbase >= 0 && W.o1 + 2 - W.lbase <= W.lrows - 1;
    QT_STAMP(0);
    if (wv >= 1) {
        __builtin_amdgcn_s_setprio(2);
        bool general = !(plain && W.o0 >= 3 && W.o1 <= W.n - 3 && P.use_q && P.K_h > 0.0);
        if (!general) {
            const bool edge_lane = W.lane >= 1 && W.lane <= 62 && (W.j == 0 || W.j == W.m - 1);
            general = __builtin_amdgcn_ballot_w64(edge_lane) != 0ull ? qt_sst_fast<true>(P, W) : qt_sst_fast<false>(P, W);
        }
        if (general) qt_sst_wave(G, T, P, W);
        QT_STAMP(2);
        if (!FIX) return;
        __syncthreads();                                     // fix list: this strip's ticket is taken when BOTH its waves have read their uo', vo'
        return;
    }
    __shared__ unsigned long long s_fix[FIX ? 3 * QT_FIX_STAGE : 1];
    __shared__ unsigned int s_nfix;
    const QtFixW FX{P.fix_count, P.fix_list, s_fix, &s_nfix};
    if (FIX && W.lane == 0) __hip_atomic_store(&s_nfix, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);             // (this wave is the only one that touches the stage: its LDS accesses are in order)
    double acc = (plain && W.o0 >= 1 && W.o1 <= W.n - 1) ? qt_currents_fast<FIX>(P, W, FX) : qt_currents_wave<FIX>(T, P, W, FX);
    if (FIX) qt_fix_flush(FX);
    QT_STAMP(2);
    if (FIX) __syncthreads();
    acc = qt_wave_sum(acc);
    if (W.lane == 0) __hip_atomic_store(P.partial + w, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // coherent: the finisher may read it
    if (P.acc) {                                             // eta mean inside this launch: the last workgroup to arrive finishes it
        const bool last = W.lane == 0 && qd_acc_arrive(P.acc, w, gridDim.x, acc);
        if (__builtin_amdgcn_ballot_w64(last) != 0ull) {
            double mm = qd_acc_finish(P.acc, P.partial, (int)gridDim.x, P.wsum);
            if (P.pf.pbox) mm = qp_fold_sum(P.pf, mm);        // bands over the peer exchange: the sum over the ranks, here and now
            if (W.lane == 0) *P.mean_out = mm;
            if (FIX) {
                // every wave of the launch is done with uo', vo': the noted cells get their uo'' / vo'' in place (entries and count were
                // written with agent-scope atomics before their strips' tickets: qd_acc_arrive waits for them)
                const unsigned nfix = __hip_atomic_load(P.fix_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // one wave, eight entries per lane in flight: the loop is a chain of round trips, 512 entries each
                for (unsigned k0 = 0; k0 < nfix; k0 += 512u) {
                    unsigned long long off[8], ub[8], vb[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const unsigned k = k0 + (unsigned)q * 64u + (unsigned)W.lane;
                        const unsigned long long* e = P.fix_list + (size_t)3 * (k < nfix ? k : nfix - 1u);
                        off[q] = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ub[q] = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        vb[q] = __hip_atomic_load(e + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (k0 + (unsigned)q * 64u + (unsigned)W.lane >= nfix) continue;
                        *(unsigned long long*)((char*)const_cast<double*>(P.uo) + off[q]) = ub[q];
                        *(unsigned long long*)((char*)const_cast<double*>(P.vo) + off[q]) = vb[q];
                    }
                }
                if (W.lane == 0) {
                    __hip_atomic_store(P.fix_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // statistics (QD_TAIL_FIX_DEBUG prints them when the handle is destroyed): entries of all launches, launches, the longest list
                    P.fix_count[4] += nfix; P.fix_count[5] += 1u; if (nfix > P.fix_count[6]) P.fix_count[6] = nfix;
                }
            }
        }
    }
}

// =========================================================================================
// the WHOLE ocean sub-step in one launch, streaming form (round 4; QD_OCN_FUSED=1)
// =========================================================================================
// k_ocn_stream writes uo', vo', eta' (the momentum update + del^4: ocean.py:306-356) and k_ocn_tail_fast reads them back (continuity,
// SST, outlier filter: ocean.py:365-444): 25 MB out, 33 MB in again, two launches with a ramp each.  Here a 320-thread workgroup owns
// a strip of R rows x 56 columns (lanes 4 .. 59 of 64) and its five waves run CONCURRENTLY:
//   waves 0, 1, 2   the uo / vo / eta waves of qd_stream.h on rows o0 - 2 .. o1 + 1 (what the SST advection needs), their output rows
//                   going into three LDS rings of QFU_NR rows instead of global memory (QsOutRing),
//   wave 3          the currents wave of k_ocn_tail_fast (continuity, eta sum, outlier filter / caps), uo', vo', eta' from the rings,
//   wave 4          its SST wave (register gather, K_h lap, heating), uo', vo' from the rings;
// producers publish "rows so far" in LDS, consumers "first row still needed"; a producer waits when it is a ring ahead.  uo', vo',
// eta' never reach memory; the same device functions in the same order as the two launches: bit-identical fields, the eta sum in
// another strip order.
// What cannot stream takes the SEQUENTIAL form inside the same launch -- K1 of the strip into the global scratch slabs, workgroup
// barrier, the general tail waves from there --: (a) the two polar tiles of a column strip (rows 0 .. 6 and n-7 .. n-1; np.roll(axis=0)
// couples them: mean4 reads row -1 as row n-1), which ONE workgroup owns, (b) any strip whose fast waves raised `bad` (a non-finite
// value, a departure point a cell away): the strip is done again -- every output goes to a buffer nobody reads during the launch.
#define QFU_TC 56
#define QFU_PH 7
typedef double __attribute__((address_space(3)))* qfu_ldp;
typedef int __attribute__((address_space(3)))* qfu_lip;
struct QfuArgs { int ntc, nmid, R, mode; };                  // mode bit0: every strip takes the sequential form (tests)

struct QsOutRing {                // where a streaming K1 wave's rows go
    double* plane; int* prod; const int* c3; const int* c4; int row, chk, lane;
    __device__ __forceinline__ void put(double v) {
        if (row >= chk) {                                    // room for rows row .. row + 3 ?
            for (;;) {
                const int a = __hip_atomic_load(c3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int b = __hip_atomic_load(c4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (row + 3 - (a < b ? a : b) < QFU_NR) break;
                __builtin_amdgcn_s_sleep(1);
            }
            chk = row + 4;
        }
        plane[(row & (QFU_NR - 1)) * 64 + lane] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the row is in LDS before its number is
        if (lane == 0) __hip_atomic_store(prod, row + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ++row;
    }
};

// the strip as the K1 waves see it: rows [k0, k1) to produce, lanes 3 .. 60 hold valid columns (stored at the WRAPPED column: a
// duplicate of a neighbouring strip's column carries the same bits)
__device__ __forceinline__ void qfu_k1w(const QdGeom& G, int cs, int k0, int k1, QsW& W) {
    W.n = G.nlat; W.nlon = G.nlon; W.lane = threadIdx.x & 63;
    const int jraw = cs * QFU_TC - 4 + W.lane;
    W.j = jraw < 0 ? jraw + G.nlon : (jraw >= G.nlon ? jraw - G.nlon : jraw);
    W.vo = (unsigned)W.j * 8u; W.vo8 = (unsigned)W.j;
    W.vs = (W.lane >= 3 && W.lane <= 60) ? (unsigned)W.j * 8u : QS_OOB;
    W.slab_bytes = (unsigned)(G.lrows_ + QD_PAD_ROWS) * (unsigned)G.nlon * 8u;
    W.west_edge = W.j == 0; W.east_edge = W.j == G.nlon - 1;
    W.o0 = k0; W.o1 = k1;
}
__device__ __forceinline__ void qfu_tailw(const QdGeom& G, const QdTailArgs& P, int cs, int o0, int o1, QtW& W) {
    W.n = G.nlat; W.m = G.nlon; W.lane = threadIdx.x & 63;
    W.jbase = cs * QFU_TC - 4;
    const int jraw = W.jbase + W.lane;
    W.j = jraw < 0 ? jraw + W.m : (jraw >= W.m ? jraw - W.m : jraw);
    W.own = W.lane >= 4 && W.lane <= 3 + QFU_TC && jraw < W.m;
    W.vo = (unsigned)W.j * 8u; W.vo8 = (unsigned)W.j; W.vs = W.own ? (unsigned)jraw * 8u : 0x80000000u;
    W.slab = (unsigned)(G.lrows_ + QD_PAD_ROWS) * (unsigned)G.nlon * 8u;
    W.lbase = G.lbase; W.lrows = G.lrows_; W.own0 = P.own0; W.own1 = P.own1;
    W.o0 = o0; W.o1 = o1;
}

// K1 of rows [k0, k1) into the global scratch slabs (FAST, and EXACT again for a wave that saw a non-finite value: k_ocn_stream's logic)
__device__ __forceinline__ void qfu_k1_global(const QsOcnArgs& A, int cs, int k0, int k1, int wv, const QsRec QD_CONST* fp) {
    QsW W;
    qfu_k1w(A.G, cs, k0, k1, W);
    if (!A.exact) {
        QsOutGlobal out{qs_make_rsrc(fp->out, W.slab_bytes), qs_off(A.G, W.o0), W.vs, (unsigned)W.nlon};
        const bool bad = qs_ocn_wave<QS_FAST>(A, W, wv, fp, out);
        if (__builtin_amdgcn_ballot_w64(bad) == 0ull) return;
    }
    QsOutGlobal out{qs_make_rsrc(fp->out, W.slab_bytes), qs_off(A.G, W.o0), W.vs, (unsigned)W.nlon};
    qs_ocn_wave<QS_EXACT>(A, W, wv, fp, out);
}

// the ring-fed currents wave: qt_currents_fast with uo', vo', eta' out of LDS
__device__ __forceinline__ double qfu_currents(const QdTailArgs& P, const QtW& W, const QfuRing& R) {
    const unsigned sb = W.slab, m = (unsigned)W.m;
    const qt_rsrc E = qt_make_rsrc(P.eta, sb), L = qt_make_rsrc(P.land, sb / 8u);
    const qt_rsrc UO = qt_make_rsrc(P.uo_out, sb), VO = qt_make_rsrc(P.vo_out, sb), KT = qt_make_rsrc(P.tab, (unsigned)W.n * 128u);
    unsigned ro = (unsigned)(W.o0 - W.lbase) * m;
    double acc = 0.0;
    QtCurK K;
    K.dlon2 = qt_vreg(2 * P.dlon); K.r_2dlon = qt_vreg(P.r_2dlon); K.dlat2 = qt_vreg(2 * P.dlat); K.r_2dlat = qt_vreg(P.r_2dlat);
    K.msdtH = qt_vreg(P.msdtH); K.cap = qt_vreg(P.cap); K.cap81 = qt_vreg(0.81 * (P.cap * P.cap)); K.mean4 = P.mean4;
    QtCurSlot a, b;
    a.u = a.v = a.e = b.u = b.v = b.e = 0.0;
    a.l = qt_ld8(L, ro, W.vo8); a.k0 = qt_ldk(KT, 16u * (unsigned)W.o0, 4u); a.k1 = qt_ldk(KT, 16u * (unsigned)W.o0, 5u);
    b.l = qt_ld8(L, ro + m, W.vo8); b.k0 = qt_ldk(KT, 16u * (unsigned)(W.o0 + 1), 4u); b.k1 = qt_ldk(KT, 16u * (unsigned)(W.o0 + 1), 5u);
    qfu_wait_gt(R.prog + 0, W.o0); qfu_wait_gt(R.prog + 1, W.o0);
    double us = qfu_ring_ld(R.u, W.o0 - 1, W.lane), vs = qfu_ring_ld(R.v, W.o0 - 1, W.lane);
    double uc = qfu_ring_ld(R.u, W.o0, W.lane), vc = qfu_ring_ld(R.v, W.o0, W.lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int g = W.o0;
    for (; g + 1 < W.o1; g += 2) {
        qt_cur_fast_step<true>(K, W, E, E, E, L, UO, VO, KT, us, uc, vs, vc, acc, a, g, ro, &R);
        qt_cur_fast_step<true>(K, W, E, E, E, L, UO, VO, KT, us, uc, vs, vc, acc, b, g + 1, ro + m, &R);
        ro += 2u * m;
    }
    if (g < W.o1) qt_cur_fast_step<true>(K, W, E, E, E, L, UO, VO, KT, us, uc, vs, vc, acc, a, g, ro, &R);
    if (W.lane == 0) __hip_atomic_store(R.prog + 3, 0x3fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // done: never in a producer's way
    return acc;
}

// the ring-fed SST wave: qt_sst_fast with uo', vo' out of LDS; true: a lane needed the general gather
template <bool EDGE>
__device__ __forceinline__ bool qfu_sst(const QdTailArgs& P, const QtW& W, const QfuRing& R) {
    const unsigned sb = W.slab, m = (unsigned)W.m;
    const qt_rsrc S = qt_make_rsrc(P.Ts, sb), Q = qt_make_rsrc(P.qnet, sb);
    const qt_rsrc L = qt_make_rsrc(P.land, sb / 8u), I = qt_make_rsrc(P.ice, sb / 8u), SO = qt_make_rsrc(P.Ts_out, sb);
    const qt_rsrc KT = qt_make_rsrc(P.tab, (unsigned)W.n * 128u);
    QtSstK K;
    K.a = qt_vreg(P.a); K.r_a = qt_vreg(P.r_a); K.dt = qt_vreg(P.sub_dt); K.dlon = qt_vreg(P.dlon); K.r_dlon = qt_vreg(P.r_dlon);
    K.dlat = qt_vreg(P.dlat); K.r_dlat = qt_vreg(P.r_dlat); K.alpha = qt_vreg(P.alpha); K.om_alpha = qt_vreg(1.0 - P.alpha);
    K.dtK = qt_vreg(P.sub_dt * P.K_h); K.rcH = qt_vreg(P.rcH); K.r_rcH = qt_vreg(P.r_rcH); K.dtq = qt_vreg(P.sub_dt * P.ice_qfac);
    K.has_ice = P.has_ice; K.qfac_on = P.ice_qfac > 0.0 ? 1 : 0;
    const int t0 = W.o0 - 2;
    unsigned ro = (unsigned)(t0 - W.lbase) * m;
    double Tm = qt_ld(S, ro - m, W.vo), T0 = qt_ld(S, ro, W.vo);
    QtSstSlot a, b;
    a.u = a.v = b.u = b.v = 0.0;
    a.s = qt_ld(S, ro + m, W.vo);
    a.k0 = qt_ldk(KT, 16u * (unsigned)t0, 0u); a.k1 = qt_ldk(KT, 16u * (unsigned)t0, 1u); a.k2 = qt_ldk(KT, 16u * (unsigned)t0, 2u);
    b.s = qt_ld(S, ro + 2u * m, W.vo);
    b.k0 = qt_ldk(KT, 16u * (unsigned)(t0 + 1), 0u); b.k1 = qt_ldk(KT, 16u * (unsigned)(t0 + 1), 1u); b.k2 = qt_ldk(KT, 16u * (unsigned)(t0 + 1), 2u);
    a.q = b.q = 0.0; a.l = b.l = 0; a.ic = b.ic = 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    double Wm = 0.0, W0 = 0.0, Em = 0.0, E0 = 0.0;
    if (!EDGE) { Wm = qd_west(Tm); W0 = qd_west(T0); Em = qd_east(Tm); E0 = qd_east(T0); }
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0, w4 = 0.0;
    bool bad = false;
#define QFU_FS(OUTF, SL, GG, RO) qt_sst_fast_step<EDGE, OUTF, true>(K, W, S, S, S, Q, L, I, SO, KT, Tm, T0, Wm, W0, Em, E0, w0, w1, w2, w3, w4, bad, SL, GG, RO, &R)
    int g = t0;
    QFU_FS(false, a, g, ro); QFU_FS(false, b, g + 1, ro + m); QFU_FS(false, a, g + 2, ro + 2u * m); QFU_FS(false, b, g + 3, ro + 3u * m);
    g += 4; ro += 4u * m;
    const int gend = W.o1 + 2;
    for (; g + 1 < gend; g += 2) { QFU_FS(true, a, g, ro); QFU_FS(true, b, g + 1, ro + m); ro += 2u * m; }
    if (g < gend) QFU_FS(true, a, g, ro);
#undef QFU_FS
    if (W.lane == 0) __hip_atomic_store(R.prog + 4, 0x3fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_ballot_w64(bad) != 0ull;
}

__global__ void __launch_bounds__(320)
k_ocn_fused(QsOcnArgs A, QdTabs T, QdTailArgs P, QfuArgs F) {
    __shared__ double s_ring[3][QFU_NR][64];
    __shared__ int s_prog[8];
    const QdGeom& G = A.G;
    const int n = G.nlat;
    const unsigned w = qd_xcd_chunk(blockIdx.x, gridDim.x);
    // Which wave plays which role rotates with the workgroup's layer on its CU (workgroups b, b + 256, b + 512 share a CU: dispatch is
    // round-robin over 8 XCDs x 32 CUs; waves w and w + 4 of a workgroup share a SIMD): with fixed roles the SST waves of a CU, the
    // heaviest, sit on one SIMD (measured: 64.8 -> 61.5 us per launch at R = 26).  QD_FUSED_NOROT=1: fixed roles.
    const int hwv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lyr = (int)((blockIdx.x >> 8) % 3u);
    const int rot = (F.mode & 2) ? 0 : (lyr == 0 ? 0 : (lyr == 1 ? 2 : 3));
    const int wv = (hwv + 5 - rot) % 5;                       // the ROLE of this wave: 0, 1, 2 = uo / vo / eta, 3 = currents, 4 = SST
    const int lane = threadIdx.x & 63;
    const QsOcnArgs QD_CONST* Ak = (const QsOcnArgs QD_CONST*)__builtin_amdgcn_kernarg_segment_ptr();
    const QsRec QD_CONST* fp = &Ak->rec[wv < 3 ? wv : 0];
    double acc = 0.0;
    QT_STAMP(0);
    if ((int)w < F.ntc) {
        // ---- the two polar tiles of column strip w, sequential form
        const int cs = (int)w;
        if (wv < 3) { qfu_k1_global(A, cs, 0, QFU_PH + 2, wv, fp); qfu_k1_global(A, cs, n - QFU_PH - 2, n, wv, fp); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        QtW W;
        if (wv == 3) { qfu_tailw(G, P, cs, 0, QFU_PH, W); acc = qt_currents_wave(T, P, W); qfu_tailw(G, P, cs, n - QFU_PH, n, W); acc += qt_currents_wave(T, P, W); }
        if (wv == 4) { qfu_tailw(G, P, cs, 0, QFU_PH, W); qt_sst_wave(G, T, P, W); qfu_tailw(G, P, cs, n - QFU_PH, n, W); qt_sst_wave(G, T, P, W); }
    } else {
        const unsigned k = w - (unsigned)F.ntc;
        const int rs = (int)(k / (unsigned)F.ntc), cs = (int)(k % (unsigned)F.ntc);
        const int M = n - 2 * QFU_PH;
        const int o0 = QFU_PH + (int)(((long long)rs * M) / F.nmid), o1 = QFU_PH + (int)(((long long)(rs + 1) * M) / F.nmid);
        bool redo = (F.mode & 1) != 0;
        if (!redo) {
            // ---- streaming form
            if (threadIdx.x < 8) s_prog[threadIdx.x] = threadIdx.x < 3 ? o0 - 2 : (threadIdx.x == 3 ? o0 - 1 : (threadIdx.x == 4 ? o0 - 2 : 0));
            __syncthreads();
            QfuRing R{&s_ring[0][0][0], &s_ring[1][0][0], &s_ring[2][0][0], s_prog};
            bool bad = false;
            if (wv < 3) {
                __builtin_amdgcn_s_setprio(2);
                QsW W;
                qfu_k1w(G, cs, o0 - 2, o1 + 2, W);
                QsOutRing out{&s_ring[wv][0][0], s_prog + wv, s_prog + 3, s_prog + 4, o0 - 2, o0 - 2, lane};
                bad = qs_ocn_wave<QS_FAST>(A, W, wv, fp, out);
                __builtin_amdgcn_s_setprio(0);
            } else {
                QtW W;
                qfu_tailw(G, P, cs, o0, o1, W);
                if (wv == 3) acc = qfu_currents(P, W, R);
                else {
                    const bool edge_lane = W.lane >= 1 && W.lane <= 62 && (W.j == 0 || W.j == W.m - 1);
                    bad = __builtin_amdgcn_ballot_w64(edge_lane) != 0ull ? qfu_sst<true>(P, W, R) : qfu_sst<false>(P, W, R);
                }
            }
            QT_STAMP(2);
            if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) __hip_atomic_store(s_prog + 5, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();
            redo = __hip_atomic_load(s_prog + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
        }
        if (redo) {
            // ---- sequential form of this strip (test mode, or the fast waves raised `bad`): everything it stores overwrites the above
            if (wv < 3) qfu_k1_global(A, cs, o0 - 2, o1 + 2, wv, fp);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            QtW W;
            qfu_tailw(G, P, cs, o0, o1, W);
            if (wv == 3) acc = qt_currents_wave(T, P, W);
            if (wv == 4) qt_sst_wave(G, T, P, W);
        }
    }
    if (wv != 3) return;
    acc = qt_wave_sum(acc);
    if (lane == 0) __hip_atomic_store(P.partial + w, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (P.acc) {
        const bool last = lane == 0 && qd_acc_arrive(P.acc, w, gridDim.x, acc);
        if (__builtin_amdgcn_ballot_w64(last) != 0ull) {
            const double mm = qd_acc_finish(P.acc, P.partial, (int)gridDim.x, P.wsum);
            if (lane == 0) *P.mean_out = mm;
        }
    }
}

static bool qfu_shape(const qd_ctx* c, QfuArgs& F) {
    const QdGeom& G = c->geo;
    if (!G.full || G.nlon < 64 || G.nlat < 2 * QFU_PH + 24) return false;
    F.ntc = (G.nlon + QFU_TC - 1) / QFU_TC;
    const int M = G.nlat - 2 * QFU_PH;
    // strip height: QD_FUSED_R, else about three workgroups of five waves per CU resident at once (256 CUs: see qt_fast_rows for why
    // the device is not asked)
    int R = c->tune.fused_r;
    if (R <= 0) { const int nrs = std::max(1, (768 - F.ntc) / F.ntc); R = (M + nrs - 1) / nrs; }
    R = std::max(R, 12);
    F.nmid = std::max(1, M / R);
    F.R = (M + F.nmid - 1) / F.nmid;
    F.mode = (c->tune.fused_seq ? 1 : 0) | (c->tune.fused_norot ? 2 : 0);
    return true;
}
bool qd_ocn_fused_ok(const qd_ctx* c) { QfuArgs F; return qfu_shape(c, F); }
int qd_ocn_fused_tiles(const qd_ctx* c) { QfuArgs F; return qfu_shape(c, F) ? F.ntc * (1 + F.nmid) : 0; }

// O: the momentum kernel's arguments with uo_out / vo_out / eta_out = the scratch slabs of the sequential form; P: the tail's
// arguments with uo / vo = those slabs, eta = the eta_out slab (in: eta' of the sequential form, out: the new eta)
int qd_launch_ocn_fused(qd_ctx* c, const QdOcnArgs& O, QdTailArgs& P) {
    QfuArgs F;
    if (!qfu_shape(c, F)) return qd_fail(c, "k_ocn_fused: whole-globe handles of >= 64 columns and >= 38 rows only");
    QsOcnArgs A;
    if (!qd_stream_ocn_args(c, O, A)) return qd_fail(c, "k_ocn_fused: coefficient row tables");
    A.G = c->geo;
    if (F.ntc * (1 + F.nmid) > c->red_blocks) return qd_fail(c, "k_ocn_fused: partial buffer too small");
    if (!c->qt_tab || c->qt_tab_a != P.a) {
        if (!c->qt_tab) {
            if (hipMalloc(&c->qt_tab, (size_t)c->geo.nlat * 16 * sizeof(double)) != hipSuccess) return qd_fail(c, "k_ocn_fused: row table");
            c->tab_alloc.push_back(c->qt_tab);
        }
        hipLaunchKernelGGL(k_tail_tab, dim3((c->geo.nlat + 255) / 256), dim3(256), 0, c->stream, c->tabs, c->geo.nlat, P.a, c->qt_tab);
        c->qt_tab_a = P.a;
    }
    P.tab = c->qt_tab; P.own0 = 0; P.own1 = c->geo.nlat; P.flags = 0; P.R = F.R; P.Rp = QFU_PH; P.ntc = F.ntc; P.nmid = F.nmid;
    QdScope sc(c, "ocean_step", true);
#ifdef QT_STAMPS
    static unsigned long long* stamps = nullptr;
    const size_t stamp_words = (size_t)8 * 5 * 8192;
    if (!stamps) { hipMalloc(&stamps, stamp_words * 8); hipMemcpyToSymbol(HIP_SYMBOL(qt_stamp_buf), &stamps, sizeof(stamps)); }
    hipMemsetAsync(stamps, 0, stamp_words * 8, c->stream);
#endif
    QD_LAUNCH_TIMED(sc, k_ocn_fused, dim3(F.ntc * (1 + F.nmid)), dim3(320), c->stream, A, c->tabs, P, F);
#ifdef QT_STAMPS
    if (const char* f = std::getenv("QD_STAMPS_FILE")) {     // developer build (-DQT_STAMPS) only
        std::vector<unsigned long long> h(stamp_words);
        hipStreamSynchronize(c->stream);
        hipMemcpy(h.data(), stamps, stamp_words * 8, hipMemcpyDeviceToHost);
        if (FILE* fp = std::fopen(f, "wb")) { std::fwrite(h.data(), 8, stamp_words, fp); std::fclose(fp); }
    }
#endif
    return 0;
}

// Strip height of the round-3 streaming form (k_ocn_tail_stream, QD_TAIL_V=1).  Measured (rocprofv3 kernel trace, 721 x 1440), with the
// eta mean finished inside the launch: R = 6 / 7 / 8 / 9 / 10 / 12 -> 29.2 / 24.6 / 26.3 / 26.3 / 26.9 / 28.5 us: a wave is a serial chain of rows,
// shorter strips mean more of them in parallel; below 7 the SST wave's four halo rows dominate.  (Two SST waves per strip, QD_TAIL_NS=2
// of round 3, looked good in a 12-step probe and lost in the 240-step bench: retired.)  QD_TAIL_R overrides (read at create).
static int qt_rows(const qd_ctx* c) { return c->tune.tail_r > 0 ? c->tune.tail_r : 7; }

// the slim streaming form (k_ocn_tail_fast) unless QD_TAIL_V=1 asks for the round-3 kernel; its strip heights
static bool qt_use_fast(const qd_ctx* c) { return !c->tune.tail_v && c->ocn_tail == 1; }
// 240-step bench, 721 x 1440, ms per step (same box): R = 8 / 9 / 10 / 11 / 12 -> 0.926 / 0.895 / 0.894 / 0.907 / 0.896; pole strips of 3 / 4 / 5
// rows: 0.894 / 0.893 / 0.893; the round-3 kernel on that box: 0.940
// 1441 x 2880 (48 column groups): R = 10 / 14 / 20 / 28 / 40 / 56 -> 5.13 / 5.09 / 4.87 / 4.84 / 4.79 / 4.88 ms per step (round-3 kernel: 5.28).  Both optima
// are the strip height that makes ~3500 waves -- every wave of an MI355X (256 CUs) resident from the start, three to four per SIMD.
// The height is a function of the GRID ONLY (not of the device's CU count, which round 3 queried): the strip cut fixes the
// grouping of the eta partial sums, i.e. the bits of every later step, and those must not depend on which GPU or partition mode
// the run lands on.
static int qt_fast_rows(const qd_ctx* c, const QdGeom& G) {
    if (c->tune.tail_r > 0) return c->tune.tail_r;
    const int ntc = (G.nlon + QS_TCF - 1) / QS_TCF;
    const double target = 13.7 * 256;                        // waves
    const int r = (int)std::lround((double)G.nrows * ntc * 2.0 / target);
    return r < 7 ? 7 : r;
}
static int qt_fast_pole_rows(const qd_ctx* c) { return c->tune.tail_rp; }      // >= 3: a middle strip's T1 rows start at o0 - 2 >= 1

// number of eta partial sums the launch leaves in P.partial
int qd_ocn_tail_tiles(const qd_ctx* c, const QdGeom& G) {
    if (qt_use_fast(c))
        return qt_part_strips(qt_partition(G.row0, G.row0 + G.nrows, G.nlat, qt_fast_rows(c, G), qt_fast_pole_rows(c))) * ((G.nlon + QS_TCF - 1) / QS_TCF);
    const int R = qt_rows(c);
    return ((G.nrows + R - 1) / R) * ((G.nlon + QS_TC2 - 1) / QS_TC2);
}

int qd_launch_ocn_tail(qd_ctx* c, const QdGeom& G, QdTailArgs& P) {
    if (G.nlon < 64) return qd_fail(c, "k_ocn_tail: >= 64 columns");
    if (G.full) { P.own0 = 0; P.own1 = G.nlat; }
    QdScope sc(c, "ocean_tail", true);
    const bool fast = qt_use_fast(c);
    if (fast && (!c->qt_tab || c->qt_tab_a != P.a)) {        // packed row table of the fast waves (once per handle)
        if (!c->qt_tab) {
            if (hipMalloc(&c->qt_tab, (size_t)G.nlat * 16 * sizeof(double)) != hipSuccess) return qd_fail(c, "k_ocn_tail_fast: row table");
            c->tab_alloc.push_back(c->qt_tab);
        }
        hipLaunchKernelGGL(k_tail_tab, dim3((G.nlat + 255) / 256), dim3(256), 0, c->stream, c->tabs, G.nlat, P.a, c->qt_tab);
        c->qt_tab_a = P.a;
    }
    P.tab = c->qt_tab;
    P.flags = c->tune.tail_general ? 1 : 0;                  // QD_TAIL_GENERAL=1: every wave the general form (A/B runs, tests)
    P.R = fast ? qt_fast_rows(c, G) : qt_rows(c);
    P.Rp = qt_fast_pole_rows(c);
    P.ntc = fast ? (G.nlon + QS_TCF - 1) / QS_TCF : (G.nlon + QS_TC2 - 1) / QS_TC2;
    const QtPart part = qt_partition(G.row0, G.row0 + G.nrows, G.nlat, P.R, P.Rp);
    P.nmid = part.nmid;
    const int nrs = fast ? qt_part_strips(part) : (G.nrows + P.R - 1) / P.R;
    if (nrs * P.ntc > c->red_blocks) return qd_fail(c, "k_ocn_tail_stream: partial buffer too small");
#ifdef QT_STAMPS
    static unsigned long long* stamps = nullptr;
    const size_t stamp_words = (size_t)8 * 2 * 8192;
    if (!stamps) { hipMalloc(&stamps, stamp_words * 8); hipMemcpyToSymbol(HIP_SYMBOL(qt_stamp_buf), &stamps, sizeof(stamps)); }
    hipMemsetAsync(stamps, 0, stamp_words * 8, c->stream);
#endif
    if (!fast || !P.acc) { P.fix_count = nullptr; P.fix_list = nullptr; }          // the fix list is applied by k_ocn_tail_fast's finishing wave
    if (fast && P.fix_count) QD_LAUNCH_TIMED(sc, k_ocn_tail_fast<true>, dim3(nrs * P.ntc), dim3(128), c->stream, G, c->tabs, P);
    else if (fast) QD_LAUNCH_TIMED(sc, k_ocn_tail_fast<false>, dim3(nrs * P.ntc), dim3(128), c->stream, G, c->tabs, P);
    else QD_LAUNCH_TIMED(sc, k_ocn_tail_stream<1>, dim3(nrs * P.ntc), dim3(128), c->stream, G, c->tabs, P);
#ifdef QT_STAMPS
    if (const char* f = std::getenv("QD_STAMPS_FILE")) {     // developer build (-DQT_STAMPS) only
        std::vector<unsigned long long> h(stamp_words);
        hipStreamSynchronize(c->stream);
        hipMemcpy(h.data(), stamps, stamp_words * 8, hipMemcpyDeviceToHost);
        if (FILE* fp = std::fopen(f, "wb")) { std::fwrite(h.data(), 8, stamp_words, fp); std::fclose(fp); }
    }
#endif
    return 0;
}
