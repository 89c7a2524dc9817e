// qd_stream.h -- device side of the row-streaming fused momentum + del^4 kernels: row sources, the two-stage del^4 pipeline over a
// row stream, the per-wave entry points.  Shared by qd_stream.hip (k_dyn_stream, k_ocn_stream) and qd_ocntail.hip (k_ocn_step, which
// streams into LDS).  The story of the mapping is at the top of qd_stream.hip.  gfx950 only.
#pragma once
#include "qd_internal.h"
#include "qd_device.h"
#include "qd_fused.h"
#include "qd_wave.h"

#ifndef QS_STAMP
#define QS_STAMP(k) ((void)0)
#endif

#define QS_TC 58                 // owned columns per strip (lanes 3..60)

// what ONE wave needs of its field; read from the kernarg segment with the wave's (uniform) field index, so that the
// records of the other fields never occupy SGPRs.  tab: packed rows {lapA[r+1], lapP[r], lapQ[r], k4[r]} of the field.
struct QsRec { const double* in; const double* aux; double* out; const double* tab; int skip; int pad_; };

struct QsDynArgs {
    QdGeom G;
    const double *poleA, *c8, *c9;             // c8 / c9: mom_cu, mom_cv (geostrophic) | mom_px, fcor (primitive)
    const double *h, *fric;
    double dt, inv_dlon, inv_2dlon, inv_dlat, inv_2dlat, pgf_y;
    int vb, ntc, nrs, exact;
    QsRec rec[5];                              // u v h q cloud; aux = the other momentum component (primitive scheme)
};

struct QsOcnArgs {
    QdGeom G;
    const double *poleA, *fcor, *igx, *rx;
    const double *uo, *vo, *eta;
    const uint8_t* land;
    const double* eta_mean;                    // deferred end of the previous sub-step (see QdOcnArgs)
    double eta_cap, sub_dt, g, r_bot, inv_2dlon, inv_2dlat, inv_a, inv_rhoH;
    int vb, ntc, nrs, exact;
    QsRec rec[3];                              // uo vo eta; aux = taux | tauy
};

// ---------------------------------------------------------------- buffer access
typedef unsigned int qs_u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t qs_rsrc;
#define QS_RSRC_FLAGS 0x00020000               // raw buffer, 32-bit data format (the word-3 encoding of gfx90a / gfx94x / gfx950)
#define QS_OOB 0x80000000u                     // a lane offset the range check always rejects

__device__ __forceinline__ qs_rsrc qs_make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, QS_RSRC_FLAGS);
}
// row: ELEMENT offset of the row (wave-uniform, goes to the SGPR offset); vo: the lane's byte offset inside the row
__device__ __forceinline__ double qs_ld(qs_rsrc r, unsigned row, unsigned vo) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, row * 8u, 0));
}
__device__ __forceinline__ void qs_st(qs_rsrc r, unsigned row, unsigned vo, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(qs_u32x2, v), r, vo, row * 8u, 0);
}
__device__ __forceinline__ int qs_ld8(qs_rsrc r, unsigned row, unsigned vo8) {
    return (int)__builtin_amdgcn_raw_buffer_load_b8(r, vo8, row, 0);
}

// ---------------------------------------------------------------- row addressing (wave-uniform)
// Element offset of global row g.  Inside a strip local rows are consecutive, and every slab has QD_PAD_ROWS rows of slack
// behind it, so the row loop just adds the row stride; only the rows next to a pole need care:
// atmosphere rows never wrap (a row beyond a pole is never used: clamp), np.roll(axis=0) rows of the ocean's eta do
// (row -1 is row n-1, row n is row 0; band handles find them in their ring halo).
__device__ __forceinline__ unsigned qs_off(const QdGeom& G, int g) {
    g = qd_clampi(g, 0, G.nlat - 1);
    int l = qd_lrow(G, g);
    l = l < G.lrows_ ? l : G.lrows_ - 1;
    return (unsigned)l * (unsigned)G.nlon;
}
__device__ __forceinline__ unsigned qs_off_roll(const QdGeom& G, int g) {
    if (g < 0) g += G.nlat; else if (g >= G.nlat) g -= G.nlat;
    return qs_off(G, g);
}

// Rows of every input a wave keeps in flight ahead of the row it is working on.  The depth equals the unroll factor of the row
// loop, and row g of an input always lives in slot g mod 4 of a register array: a slot is read (step K of the unrolled body) and
// at once re-issued for the row four steps later, so a register that a load is still writing is never shifted or copied.  (The
// first version kept the rows in a shift chain of depth 2-3: after an unrolled body of four steps the chain is rotated against the
// registers, the allocator repairs that with v_mov copies at the loop edge, a copy has to wait for the load it copies, and the
// loop drained vmcnt once per body -- every fourth row paid a full memory round trip whatever the depth; round-3 ISA reading.)
#ifndef QS_PD_DYN
#define QS_PD_DYN 4               // atmosphere kernel
#endif
#ifndef QS_PD_OCN
#define QS_PD_OCN 2               // ocean kernel: its momentum waves hold five inputs per row -- two rows in flight measured 19.0 us against 20.2 with four (register pressure: 91 against 110 VGPRs, fewer SGPR spills in the row loop)
#endif

struct QsW {                      // what a wave knows about its strip (everything but lane / v* is wave-uniform)
    int n, nlon, lane, j, o0, o1;
    unsigned vo, vs, vo8;         // lane byte offset inside a row: loads (wrapped column) / stores (QS_OOB on halo lanes) / u8 loads
    unsigned slab_bytes;          // size of one f64 slab incl. its slack rows (the range of every buffer)
    bool west_edge, east_edge;
};

// Code variants of a wave's pass over its strip:
//   QS_FAST   finite values: no nan_to_num, non-finite values only detected (one v_cmp_class per stored value / clip input)
//   QS_EXACT  literal nan_to_num / np.clip at the reference's places (re-run of a wave that saw a non-finite value, or
//             QD_FUSED_FAST=2)
enum { QS_FAST = 0, QS_EXACT = 2 };

template <int V> __device__ __forceinline__ double qs_clip200(double x, bool& bad) {
    if (V == QS_EXACT) return qd_clip(x, -200.0, 200.0);
    bad |= qd_nonfinite(x);
    return fmin(fmax(x, -200.0), 200.0);
}

// ---------------------------------------------------------------- row sources: F[g] of one field, one row per call
// A source holds the inputs of rows g .. g+3 in four register slots (the later ones still in flight); get<K>(g) -- K = the step's
// position in the unrolled row loop = (g - first row) mod 4, a template argument so that every slot index is static -- turns row
// g into F[g]; refill<K>(), called at the END of the row step (behind a scheduling barrier: QS_REFILL), issues the loads of row
// g+4 into the slot the step has read, at the running row offset `ro` -- every instruction that reads the old contents has been
// issued by then, so the allocator can give the new row the same registers and nothing is copied at the loop edge.  get_edge<K>(g) is
// get<K>(g) for a pole row (one-sided np.gradient / np.roll across the pole).  K is never a run-time value: a select over the
// four slots is folded back into an indexed access by the compiler, and an indexed register array lives in scratch memory.

// A loaded row that lives on as it is (a plain field's F row, the h / eta rows of a three-row window) must LEAVE its slot: while
// the value sits in the slot's registers the refill needs other registers, and the slots rotate after all.  An opaque move (the
// compiler would coalesce a plain copy away) ends the slot's life at the step that reads it.
__device__ __forceinline__ double qs_own(double x) {
#if QS_OWN_ASM
    double y;
    asm("v_mov_b64 %0, %1" : "=v"(y) : "v"(x));
    return y;
#else
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xE4, 0xf, 0xf, true);      // quad_perm:[0,1,2,3]: the identity
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xE4, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
#endif
}

template <int V, int PD> struct QsSrcPlain {                  // h, q, cloud: the field itself
    qs_rsrc p; const QdGeom& G; const QsW& W; double q[PD]; unsigned ro;
    __device__ __forceinline__ void start(int g0, bool&) {
        ro = qs_off(G, g0);
#pragma unroll
        for (int k = 0; k < PD; ++k) { q[k] = qs_ld(p, ro, W.vo); ro += W.nlon; }
    }
    template <int K> __device__ __forceinline__ double get(int, bool&) { return qs_own(q[K % PD]); }
    template <int K> __device__ __forceinline__ double get_edge(int g, bool& bad) { return get<K>(g, bad); }
    template <int K> __device__ __forceinline__ void refill() { q[K % PD] = qs_ld(p, ro, W.vo); ro += W.nlon; }
};

// momentum component that needs dh/dphi: u of the geostrophic scheme, v of the primitive one (dynamics.py:488-530)
template <bool PRIM, int V, int PD> struct QsSrcLat {
    const QsDynArgs& A; const QsW& W;
    qs_rsrc H, FR, X, Y;                                              // h, friction, own component, the other one (PRIM)
    double hm, hc, hq[PD], x[PD], y[PD], fr[PD];          // h rows g-1, g (arrived); hq: h rows g+1 .. g+4
    unsigned ro;                                                      // row g+4
    __device__ __forceinline__ void start(int g0, bool&) {
        hm = qs_ld(H, qs_off(A.G, g0 - 1), W.vo);                     // row -1 does not exist: clamped, never used
        ro = qs_off(A.G, g0);
        hc = qs_ld(H, ro, W.vo);
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            x[k] = qs_ld(X, ro, W.vo); fr[k] = qs_ld(FR, ro, W.vo); y[k] = PRIM ? qs_ld(Y, ro, W.vo) : 0.0;
            ro += W.nlon;
            hq[k] = qs_ld(H, ro, W.vo);
        }
    }
    template <int EDGE> __device__ __forceinline__ double eval(int g, double hn, double x0, double y0, double f0, bool& bad) {
        double dh = (hn - hm) * A.inv_2dlat;
        if (EDGE) { if (g == 0) dh = (hn - hc) * A.inv_dlat; if (g == W.n - 1) dh = (hc - hm) * A.inv_dlat; }
        double val;
        if (PRIM) {
            const double vx = x0 + (A.pgf_y * dh - qd_sload(A.c9, g) * y0 - f0 * x0) * A.dt;
            val = qs_clip200<V>(vx, bad);
        } else {
            const double u_g = qs_clip200<V>(qd_sload(A.c8, g) * dh, bad);
            const double ur = x0 * 0.8 + u_g * 0.2;
            val = ur + (-f0 * ur) * A.dt;
        }
        // (lanes 0 and 63 have no east / west neighbour; what they hold only reaches halo columns: see QS_TC)
        hm = hc; hc = hn;                                             // arrived values: plain register moves
        return val;
    }
    template <int K> __device__ __forceinline__ double get(int g, bool& bad) { return eval<0>(g, qs_own(hq[K % PD]), x[K % PD], y[K % PD], fr[K % PD], bad); }
    template <int K> __device__ __forceinline__ double get_edge(int g, bool& bad) { return eval<1>(g, qs_own(hq[K % PD]), x[K % PD], y[K % PD], fr[K % PD], bad); }
    template <int K> __device__ __forceinline__ void refill() {
        x[K % PD] = qs_ld(X, ro, W.vo); fr[K % PD] = qs_ld(FR, ro, W.vo); if (PRIM) y[K % PD] = qs_ld(Y, ro, W.vo);
        ro += W.nlon;
        hq[K % PD] = qs_ld(H, ro, W.vo);
    }
};

// momentum component that needs dh/dlambda: v of the geostrophic scheme, u of the primitive one
// EC: the strip holds longitude 0 or n_lon - 1, where np.gradient is one-sided (the first and the last column of strips only)
template <bool PRIM, int V, int PD, bool EC> struct QsSrcLon {
    const QsDynArgs& A; const QsW& W;
    qs_rsrc H, FR, X, Y;
    double hh[PD], x[PD], y[PD], fr[PD];
    unsigned ro;
    template <int K> __device__ __forceinline__ void load() {
        hh[K % PD] = qs_ld(H, ro, W.vo); x[K % PD] = qs_ld(X, ro, W.vo); fr[K % PD] = qs_ld(FR, ro, W.vo); y[K % PD] = PRIM ? qs_ld(Y, ro, W.vo) : 0.0;
        ro += W.nlon;
    }
    __device__ __forceinline__ void start(int g0, bool&) {
        ro = qs_off(A.G, g0);
        load<0>(); if (PD > 1) load<1>(); if (PD > 2) { load<2>(); load<3>(); }
    }
    __device__ __forceinline__ double eval(int g, double hc, double x0, double y0, double f0, bool& bad) {
        const double hw = qd_west(hc), he = qd_east(hc);
        // np.gradient is one-sided at both ends of the longitude axis (not periodic: SURVEY 0.6)
        double dh;
        if (EC) {
            const double inv_lon = (W.west_edge || W.east_edge) ? A.inv_dlon : A.inv_2dlon;
            dh = ((W.east_edge ? hc : he) - (W.west_edge ? hc : hw)) * inv_lon;
        } else dh = (he - hw) * A.inv_2dlon;
        double val;
        if (PRIM) {
            const double ux = x0 + (qd_sload(A.c8, g) * dh + qd_sload(A.c9, g) * y0 - f0 * x0) * A.dt;
            val = qs_clip200<V>(ux, bad);
        } else {
            const double v_g = qs_clip200<V>(qd_sload(A.c9, g) * dh, bad);
            const double vr = x0 * 0.8 + v_g * 0.2;
            val = vr + (-f0 * vr) * A.dt;
        }
        return val;
    }
    template <int K> __device__ __forceinline__ double get(int g, bool& bad) { return eval(g, hh[K % PD], x[K % PD], y[K % PD], fr[K % PD], bad); }
    template <int K> __device__ __forceinline__ double get_edge(int g, bool& bad) { return get<K>(g, bad); }
    template <int K> __device__ __forceinline__ void refill() { load<K>(); }
};

// ocean: eta as the kernel sees it = the deferred "eta -= mean; nan_to_num; clip" of the previous sub-step applied on load.
// FAST: no branch on `defer` inside the row loops -- a handle without a deferred mean passes em = 0, cap = +inf (x - 0 and the two
// clamps are the identity on every finite x; a non-finite x raises `bad` and the strip is redone by the EXACT variant)
template <int V> __device__ __forceinline__ double qs_eta(double raw, bool defer, double em, double cap, bool& bad) {
    if (V == QS_EXACT) {
        if (!defer) return raw;
        return qd_clip(qd_nn(raw - em), -cap, cap);
    }
    const double e = raw - em;
    bad |= qd_nonfinite(e);
    return fmin(fmax(e, -cap), cap);
}

template <int V, int PD> struct QsSrcEta {
    const QsOcnArgs& A; const QsW& W; qs_rsrc E; bool defer; double em, cap; double q[PD]; unsigned ro;
    __device__ __forceinline__ void start(int g0, bool&) {
        ro = qs_off(A.G, g0);
#pragma unroll
        for (int k = 0; k < PD; ++k) { q[k] = qs_ld(E, ro, W.vo); ro += W.nlon; }
    }
    template <int K> __device__ __forceinline__ double get(int, bool& bad) { return qs_eta<V>(qs_own(q[K % PD]), defer, em, cap, bad); }
    template <int K> __device__ __forceinline__ double get_edge(int g, bool& bad) { return get<K>(g, bad); }
    template <int K> __device__ __forceinline__ void refill() { q[K % PD] = qs_ld(E, ro, W.vo); ro += W.nlon; }
};

// uo: zonal pressure gradient (ocean.py:306-336)
template <int V, int PD> struct QsSrcOcnU {
    const QsOcnArgs& A; const QsW& W; qs_rsrc E, U, Vv, T, L; bool defer; double em, cap;
    double e[PD], u0[PD], v0[PD], tx[PD]; int ld[PD]; unsigned ro;
    template <int K> __device__ __forceinline__ void load() {
        e[K % PD] = qs_ld(E, ro, W.vo); u0[K % PD] = qs_ld(U, ro, W.vo); v0[K % PD] = qs_ld(Vv, ro, W.vo); tx[K % PD] = qs_ld(T, ro, W.vo);
        ld[K % PD] = qs_ld8(L, ro, W.vo8);
        ro += W.nlon;
    }
    __device__ __forceinline__ void start(int g0, bool&) {
        ro = qs_off(A.G, g0);
        load<0>(); if (PD > 1) load<1>(); if (PD > 2) { load<2>(); load<3>(); }
    }
    __device__ __forceinline__ double eval(int g, double eraw, double u, double v, double t, int land, bool& bad) {
        const double ec = qs_eta<V>(eraw, defer, em, cap, bad);
        const double f = qd_sload(A.fcor, g);
        const double gx = ((qd_east(ec) - qd_west(ec)) * A.inv_2dlon) * qd_sload(A.igx, g);
        const double du = (f * v - A.g * gx + t * A.inv_rhoH - A.r_bot * u);
        double un = u + A.sub_dt * du;
        if (land == 1) un = 0.0;
        const double sx = A.sub_dt * qd_sload(A.rx, g);
        return un - sx * un;
    }
    template <int K> __device__ __forceinline__ double get(int g, bool& bad) { return eval(g, e[K % PD], u0[K % PD], v0[K % PD], tx[K % PD], ld[K % PD], bad); }
    template <int K> __device__ __forceinline__ double get_edge(int g, bool& bad) { return get<K>(g, bad); }
    template <int K> __device__ __forceinline__ void refill() { load<K>(); }
};

// vo: meridional pressure gradient; eta rows wrap across the poles (np.roll(axis=0), ocean.py:308)
template <int V, int PD> struct QsSrcOcnV {
    const QsOcnArgs& A; const QsW& W; qs_rsrc E, U, Vv, T, L; bool defer; double em, cap;
    double es, ec, en[PD], u0[PD], v0[PD], ty[PD]; int ld[PD];  // es, ec: eta rows g-1, g (as the kernel sees them); en: raw rows g+1 .. g+4
    unsigned ro;                                                               // row g+4
    template <int K> __device__ __forceinline__ void load() {
        u0[K % PD] = qs_ld(U, ro, W.vo); v0[K % PD] = qs_ld(Vv, ro, W.vo); ty[K % PD] = qs_ld(T, ro, W.vo);
        ld[K % PD] = qs_ld8(L, ro, W.vo8);
        ro += W.nlon;
        en[K % PD] = qs_ld(E, ro, W.vo);
    }
    __device__ __forceinline__ void start(int g0, bool& bad) {
        es = qs_eta<V>(qs_ld(E, qs_off_roll(A.G, g0 - 1), W.vo), defer, em, cap, bad);      // g0 = 0: the other pole's row
        ro = qs_off(A.G, g0);
        ec = qs_eta<V>(qs_ld(E, ro, W.vo), defer, em, cap, bad);
        load<0>(); if (PD > 1) load<1>(); if (PD > 2) { load<2>(); load<3>(); }
    }
    template <int EDGE> __device__ __forceinline__ double eval(int g, double enr, double u, double v, double t, int land, bool& bad) {
        if (EDGE) { if (g == W.n - 1) enr = qs_ld(E, qs_off_roll(A.G, g + 1), W.vo); }            // row n is row 0
        const double enc = qs_eta<V>(enr, defer, em, cap, bad);
        const double f = qd_sload(A.fcor, g);
        const double gy = ((enc - es) * A.inv_2dlat) * A.inv_a;
        const double dv = (-f * u - A.g * gy + t * A.inv_rhoH - A.r_bot * v);
        double vn = v + A.sub_dt * dv;
        if (land == 1) vn = 0.0;
        const double sx = A.sub_dt * qd_sload(A.rx, g);
        es = ec; ec = enc;
        return vn - sx * vn;
    }
    template <int K> __device__ __forceinline__ double get(int g, bool& bad) { return eval<0>(g, qs_own(en[K % PD]), u0[K % PD], v0[K % PD], ty[K % PD], ld[K % PD], bad); }
    template <int K> __device__ __forceinline__ double get_edge(int g, bool& bad) { return eval<1>(g, qs_own(en[K % PD]), u0[K % PD], v0[K % PD], ty[K % PD], ld[K % PD], bad); }
    template <int K> __device__ __forceinline__ void refill() { load<K>(); }
};

// ---------------------------------------------------------------- del^4 of a row stream
// The spherical Laplacian in the reciprocal form of the fused kernels (qd_fused.hip, QdTabs::lapA/P/Q):
//   L_r = P_r (Gb_r - Ga_r) + Q_r ((X_{j+1} - 2 X) + X_{j-1}),   Gb_r = A_{r+1} (X_{r+2} - X_r),   Ga_r = A_{r-1} (X_r - X_{r-2})
// Ga_r IS Gb_{r-2} -- the same product of the same operands -- so a row stream needs ONE new difference per row and stage:
// when row gg of F arrives the wave forms Gb(gg-2), then D[gg-2] = lap(F)[gg-2], then the same for D two rows later, and stores
// out[gg-4] = F[gg-4] - (k4 lap(D)[gg-4]) dt.  Registers carried from row to row: F rows gg-4..gg-1, D rows gg-4, gg-3, the
// last two Gb of each stage (10 doubles), and the packed coefficient rows {A[r+1], P[r], Q[r], k4[r]} of rows gg-2, gg-3,
// gg-4 in SGPRs (the row a step needs first was loaded one step earlier).
// Next to a pole np.gradient is one-sided (grid.py:41-88; coefficients QdTabs::lapPoleA, row types 0, 1, n-2, n-1):
//   Ga_0 = Ga_1 = pA0 (X_1 - X_0),   Gb_{n-2} = Gb_{n-1} = pA5 (X_{n-1} - X_{n-2}),   everything else as above
// (pA1 = A_1, pA3 = A_2, pA4 = A_{n-3}, pA6 = A_{n-2}; pA2 = pA0, pA7 = pA5), and lapP holds the pole rows' own P.
// So a strip at the south pole has its own four first steps, a strip at the north pole its own five last ones, and
// every strip runs the same row loop in between.
template <int V> __device__ __forceinline__ double qs_d2(double c) {
    // e - 2c in one fma: 2c is exact, so fma(-2, c, e) rounds exactly like (e - 2.0 * c) -- unless 2c overflows, which only
    // values that went through nan_to_num can do: the EXACT variant keeps the reference's two operations
    return (V == QS_EXACT ? (qd_east(c) - 2.0 * c) : __builtin_fma(-2.0, c, qd_east(c))) + qd_west(c);
}
template <int V> __device__ __forceinline__ double qs_nn(double x) { return V == QS_EXACT ? qd_nnf(x) : x; }

struct QsCoef { double a, p, q, k; };           // A[r+1], P[r], Q[r], k4[r]
__device__ __forceinline__ QsCoef qs_coef(const double* tab, int r, int n) {
    r = qd_clampi(r, 0, n - 1);
    const qd_cptr t = (qd_cptr)(unsigned long long)tab + 4u * (unsigned)r;
    return QsCoef{t[0], t[1], t[2], t[3]};
}

struct QsPipe {                   // what a wave carries from row to row
    double f1, f2, f3, f4;        // F rows gg-1 .. gg-4
    double gf1, gf2;              // Gb of F for rows gg-3, gg-4
    double d1, d2;                // D rows gg-3, gg-4
    double gd1, gd2;              // Gb of D for rows gg-5, gg-6
    qd_cptr kp;                   // coefficient row gg-4 of the next full step (rows gg-4 and gg-2 are in range there: no clamp, no index arithmetic)
};

// the scheduler must not lift the refill loads over the arithmetic that still reads the slot (see the row sources)
#ifndef QS_SCHED_BARRIER
#define QS_SCHED_BARRIER 1
#endif
#if QS_SCHED_BARRIER
#define QS_REFILL(S, K) do { __builtin_amdgcn_sched_barrier(0); (S).template refill<K>(); } while (0)
#else
#define QS_REFILL(S, K) (S).template refill<K>()
#endif

// Where a wave's output rows go: put() takes the rows in ascending order, starting at the first owned row of the strip.
struct QsOutGlobal {              // a slab in global memory; halo lanes: offset out of range, dropped by the hardware
    qs_rsrc r; unsigned so, vs, nlon;
    __device__ __forceinline__ void put(double v) { qs_st(r, so, vs, v); so += nlon; }
};
struct QsOutLds {                 // a plane of 64-lane rows in LDS (qd_ocntail.hip, k_ocn_step); p points at the lane's cell of the first row
    double* p;
    __device__ __forceinline__ void put(double v) { *p = v; p += 64; }
};

// one row step.  STAGE: 0 F only; 1 + Gb(F); 2 + D; 3 + Gb(D); 4 + lap(D) and the store.  K: the row's slot in the source (see QS_PD)
template <int STAGE, int K, int V, class SRC, class OUT>
__device__ __forceinline__ void qs_step(SRC& S, QsPipe& p, int gg, const QsW& W, const double* tab, double dt, OUT& out, bool& bad) {
    QsCoef k0, k2;
    if (STAGE >= 4) { k2 = QsCoef{p.kp[0], p.kp[1], p.kp[2], p.kp[3]}; k0 = QsCoef{p.kp[8], p.kp[9], p.kp[10], p.kp[11]}; p.kp += 4; }
    else { k0 = qs_coef(tab, STAGE >= 1 ? gg - 2 : 0, W.n); k2 = qs_coef(tab, STAGE >= 3 ? gg - 4 : 0, W.n); }
    const double x = qs_nn<V>(S.template get<K>(gg, bad));               // _hyperdiffuse starts from nan_to_num(F)
    double gb = 0.0, dn = 0.0, gb2 = 0.0;
    if (STAGE >= 1) {
        gb = k0.a * (x - p.f2);                                          // Gb(gg-2) = A[gg-1] (F[gg] - F[gg-2])
        if (STAGE >= 2) dn = qs_nn<V>(k0.p * (gb - p.gf2) + k0.q * qs_d2<V>(p.f2));             // D[gg-2]
    }
    if (STAGE >= 3) {
        gb2 = k2.a * (dn - p.d2);                                        // Gb2(gg-4) = A[gg-3] (D[gg-2] - D[gg-4])
        if (STAGE >= 4) {
            const double L2 = k2.p * (gb2 - p.gd2) + k2.q * qs_d2<V>(p.d2);
            double val = p.f4 - (k2.k * L2) * dt;
            if (V == QS_EXACT) val = qd_nnf(val); else bad |= qd_nonfinite(val);
            out.put(val);
        }
    }
    p.f4 = p.f3; p.f3 = p.f2; p.f2 = p.f1; p.f1 = x;
    if (STAGE >= 1) { p.gf2 = p.gf1; p.gf1 = gb; }
    if (STAGE >= 2) { p.d2 = p.d1; p.d1 = dn; }
    if (STAGE >= 3) { p.gd2 = p.gd1; p.gd1 = gb2; }
    QS_REFILL(S, K);
}

template <int V, class SRC, class OUT>
__device__ __forceinline__ bool qs_del4_stream(SRC& S, const QdGeom& G, const QsW& W, const double* poleA,
                                               const QsRec QD_CONST* fp, double dt, OUT& out) {
    const int n = W.n, o0 = W.o0, o1 = W.o1;
    bool bad = false;
    if (fp->skip) {                                          // k4 <= 0 early-out of _hyperdiffuse: the field passes through
        S.start(o0, bad);
#define QS_PASS(K) if (g + K < o1) { const int gg = g + K; \
            const double x = (gg == 0 || gg == n - 1) ? S.template get_edge<K>(gg, bad) : S.template get<K>(gg, bad); \
            out.put(x); QS_REFILL(S, K); }
        for (int g = o0; g < o1; g += 4) { QS_PASS(0) QS_PASS(1) QS_PASS(2) QS_PASS(3) }
#undef QS_PASS
        return bad;
    }
    const double* __restrict__ tab = fp->tab;
    const bool top = o0 == 0, bot = o1 == n;
    QsPipe p{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, (qd_cptr)(unsigned long long)tab + 4u * (unsigned)o0};
    if (!top) {
        // rows o0-4 .. o0+3: four rows of F, then two with Gb, two with D, two with Gb(D)
        S.start(o0 - 4, bad);
        qs_step<0, 0, V>(S, p, o0 - 4, W, tab, dt, out, bad);
        qs_step<0, 1, V>(S, p, o0 - 3, W, tab, dt, out, bad);
        qs_step<1, 2, V>(S, p, o0 - 2, W, tab, dt, out, bad);
        qs_step<1, 3, V>(S, p, o0 - 1, W, tab, dt, out, bad);
        qs_step<2, 0, V>(S, p, o0, W, tab, dt, out, bad);
        qs_step<2, 1, V>(S, p, o0 + 1, W, tab, dt, out, bad);
        qs_step<3, 2, V>(S, p, o0 + 2, W, tab, dt, out, bad);
        qs_step<3, 3, V>(S, p, o0 + 3, W, tab, dt, out, bad);
    } else {
        // south pole: rows 0 .. 3; the one-sided difference pA0 (X_1 - X_0) stands in for Ga of rows 0 and 1, in both stages
        const double pA0 = qd_sload(poleA, 0);
        S.start(0, bad);
        const double x0 = qs_nn<V>(S.template get_edge<0>(0, bad));
        QS_REFILL(S, 0);
        const double x1 = qs_nn<V>(S.template get<1>(1, bad));
        QS_REFILL(S, 1);
        p.f2 = x0; p.f1 = x1;
        p.gf1 = pA0 * (x1 - x0); p.gf2 = p.gf1;
        qs_step<2, 2, V>(S, p, 2, W, tab, dt, out, bad);    // D[0]
        const double dd0 = p.d1;
        qs_step<2, 3, V>(S, p, 3, W, tab, dt, out, bad);    // D[1]
        p.gd1 = pA0 * (p.d1 - dd0); p.gd2 = p.gd1;
    }
    QS_STAMP(1);
    // both prologues end on slot 3: the row loop starts on slot 0
    const int gEnd = bot ? n - 1 : o1 + 4;                   // north pole: the loop stops before row n-1
    int g = o0 + 4;
    // The loop is entered with nothing in flight.  The compiler's s_waitcnt at the loop head covers BOTH ways in (prologue and
    // back edge) and takes the stricter count: with the prologue's loads still pending it came out as vmcnt(0) -- a full drain
    // per four rows; after one explicit drain here the waits inside the loop are the back edge's exact counts.
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
    for (; g + 4 <= gEnd; g += 4) {                          // 4 = period of the F shift register and a multiple of the source slots; straight-line body
        qs_step<4, 0, V>(S, p, g, W, tab, dt, out, bad);
        qs_step<4, 1, V>(S, p, g + 1, W, tab, dt, out, bad);
        qs_step<4, 2, V>(S, p, g + 2, W, tab, dt, out, bad);
        qs_step<4, 3, V>(S, p, g + 3, W, tab, dt, out, bad);
    }
    const int rem = gEnd - g;                                // 0 .. 3 rows left; the source's next slot afterwards
    if (rem > 0) qs_step<4, 0, V>(S, p, g, W, tab, dt, out, bad);
    if (rem > 1) qs_step<4, 1, V>(S, p, g + 1, W, tab, dt, out, bad);
    if (rem > 2) qs_step<4, 2, V>(S, p, g + 2, W, tab, dt, out, bad);
    if (bot) {
        // north pole: rows n-1 (last row of F) .. n+3; pA5 (X_{n-1} - X_{n-2}) stands in for Gb of rows n-2 and n-1
        const double pA5 = qd_sload(poleA, 5);
        {   // gg = n-1
            const QsCoef k0 = qs_coef(tab, n - 3, n), k2 = qs_coef(tab, n - 5, n);
            double xe;                                                      // row n-1 sits in slot `rem`
            switch (rem) {
            case 0: xe = S.template get_edge<0>(n - 1, bad); break;
            case 1: xe = S.template get_edge<1>(n - 1, bad); break;
            case 2: xe = S.template get_edge<2>(n - 1, bad); break;
            default: xe = S.template get_edge<3>(n - 1, bad); break;
            }
            const double x = qs_nn<V>(xe);
            const double e1 = pA5 * (x - p.f1);
            const double gb = k0.a * (x - p.f2);
            const double dn = qs_nn<V>(k0.p * (gb - p.gf2) + k0.q * qs_d2<V>(p.f2));          // D[n-3]
            const double gb2 = k2.a * (dn - p.d2);
            const double L2 = k2.p * (gb2 - p.gd2) + k2.q * qs_d2<V>(p.d2);                   // lap(D)[n-5]
            double val = p.f4 - (k2.k * L2) * dt;
            if (V == QS_EXACT) val = qd_nnf(val); else bad |= qd_nonfinite(val);
            out.put(val);
            p.f4 = p.f3; p.f3 = p.f2; p.f2 = p.f1; p.f1 = x;
            p.gf2 = p.gf1; p.gf1 = gb; p.d2 = p.d1; p.d1 = dn; p.gd2 = p.gd1; p.gd1 = gb2;
            // ---- gg = n, n+1: D[n-2], D[n-1] with Gb = e1
            double e1d = 0.0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const QsCoef k0 = qs_coef(tab, n - 2 + t, n), k2 = qs_coef(tab, n - 4 + t, n);
                const double dq = qs_nn<V>(k0.p * (e1 - p.gf2) + k0.q * qs_d2<V>(p.f2));      // D[n-2+t]
                if (t == 1) e1d = pA5 * (dq - p.d1);                                          // pA5 (D[n-1] - D[n-2])
                const double g2 = k2.a * (dq - p.d2);
                const double M2 = k2.p * (g2 - p.gd2) + k2.q * qs_d2<V>(p.d2);                // lap(D)[n-4+t]
                double v2 = p.f4 - (k2.k * M2) * dt;
                if (V == QS_EXACT) v2 = qd_nnf(v2); else bad |= qd_nonfinite(v2);
                out.put(v2);
                p.f4 = p.f3; p.f3 = p.f2; p.f2 = p.f1; p.f1 = 0.0;
                p.gf2 = p.gf1; p.gf1 = e1; p.d2 = p.d1; p.d1 = dq; p.gd2 = p.gd1; p.gd1 = g2;
            }
            // ---- gg = n+2, n+3: lap(D)[n-2], lap(D)[n-1] with Gb(D) = e1d
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const QsCoef k2 = qs_coef(tab, n - 2 + t, n);
                const double M2 = k2.p * (e1d - p.gd2) + k2.q * qs_d2<V>(p.d2);
                double v2 = p.f4 - (k2.k * M2) * dt;
                if (V == QS_EXACT) v2 = qd_nnf(v2); else bad |= qd_nonfinite(v2);
                out.put(v2);
                p.f4 = p.f3; p.f3 = p.f2; p.f2 = p.f1; p.f1 = 0.0;
                p.d2 = p.d1; p.d1 = 0.0; p.gd2 = p.gd1; p.gd1 = e1d;
            }
        }
    }
    return bad;
}

// strip of this workgroup -> wave context.  The nrs strips of a row segment split its rows evenly (heights differ by at most one row:
// a last strip that also took the remainder ran 1.5-2x as long as the others at some grid sizes and set the launch time); strips
// are at least 12 rows high, so none ends within four rows of a pole without holding it.  `vb` (host: qs_shape) = rows by which the
// strip that ENDS at the north pole is shorter than an even share: its five epilogue steps are a serial chain behind a drained
// pipeline, and it is the last strip to be dispatched -- at an even share its waves ended 3.5 us after everybody else's
// (per-wave s_memrealtime stamps, round 3: median end 18.2 us, strip 29 of 30 at 21.7 us, launch over at 22.8 us).
__host__ __device__ __forceinline__ int qs_cut(int nrows, int nrs, int vb, int rs) {
    return rs >= nrs ? nrows : (int)(((long long)rs * (nrows + vb)) / nrs);
}
// (bid of nb: the workgroup's index among the launch's strip workgroups -- k_ocn_stream_push has others in front of them)
__device__ __forceinline__ void qs_strip(const QdGeom& G, int vb, int ntc, int nrs, QsW& W, unsigned bid, unsigned nb) {
    const unsigned w = qd_xcd_chunk(bid, nb);
    const int rs = (int)(w / (unsigned)ntc), cs = (int)(w % (unsigned)ntc);
    W.n = G.nlat; W.nlon = G.nlon;
    W.lane = threadIdx.x & 63;
    const int jraw = cs * QS_TC - 3 + W.lane;
    W.j = jraw < 0 ? jraw + G.nlon : (jraw >= G.nlon ? jraw - G.nlon : jraw);
    const bool col_ok = W.lane >= 3 && W.lane <= 60 && jraw < G.nlon;
    W.vo = (unsigned)W.j * 8u; W.vo8 = (unsigned)W.j; W.vs = col_ok ? (unsigned)jraw * 8u : QS_OOB;
#ifdef QS_NOSTORE                 // diagnostic build: every store is dropped by the range check (what do the writes cost?)
    W.vs = QS_OOB;
#endif
    W.slab_bytes = (unsigned)(G.lrows_ + QD_PAD_ROWS) * (unsigned)G.nlon * 8u;
    W.west_edge = W.j == 0; W.east_edge = W.j == G.nlon - 1;
    W.o0 = G.row0 + qs_cut(G.nrows, nrs, vb, rs);
    W.o1 = G.row0 + qs_cut(G.nrows, nrs, vb, rs + 1);
}
__device__ __forceinline__ void qs_strip(const QdGeom& G, int vb, int ntc, int nrs, QsW& W) { qs_strip(G, vb, ntc, nrs, W, blockIdx.x, gridDim.x); }

template <bool PRIM, int V>
__device__ __forceinline__ bool qs_dyn_wave(const QsDynArgs& A, const QsW& W, int wv) {
    const QsDynArgs QD_CONST* Ak = (const QsDynArgs QD_CONST*)__builtin_amdgcn_kernarg_segment_ptr();
    const QsRec QD_CONST* fp = &Ak->rec[wv];
    const unsigned sb = W.slab_bytes;
    QsOutGlobal out{qs_make_rsrc(fp->out, sb), qs_off(A.G, W.o0), W.vs, (unsigned)W.nlon};
    if (wv <= 1) {
        const qs_rsrc H = qs_make_rsrc(A.h, sb), FR = qs_make_rsrc(A.fric, sb), X = qs_make_rsrc(fp->in, sb), Y = qs_make_rsrc(fp->aux, sb);
        if ((wv == 0) == PRIM) {
            if (__builtin_amdgcn_ballot_w64(W.west_edge || W.east_edge) != 0ull) {
                QsSrcLon<PRIM, V, QS_PD_DYN, true> S{A, W, H, FR, X, Y};
                return qs_del4_stream<V>(S, A.G, W, A.poleA, fp, A.dt, out);
            }
            QsSrcLon<PRIM, V, QS_PD_DYN, false> S{A, W, H, FR, X, Y};
            return qs_del4_stream<V>(S, A.G, W, A.poleA, fp, A.dt, out);
        }
        QsSrcLat<PRIM, V, QS_PD_DYN> S{A, W, H, FR, X, Y};
        return qs_del4_stream<V>(S, A.G, W, A.poleA, fp, A.dt, out);
    }
    QsSrcPlain<V, QS_PD_DYN> S{qs_make_rsrc(fp->in, sb), A.G, W};
    return qs_del4_stream<V>(S, A.G, W, A.poleA, fp, A.dt, out);
}

// fp: the wave's field record (kernarg segment of the calling kernel); out: where its rows go
template <int V, class OUT>
__device__ __forceinline__ bool qs_ocn_wave(const QsOcnArgs& A, const QsW& W, int wv, const QsRec QD_CONST* fp, OUT& out) {
    const bool defer = A.eta_mean != nullptr;
    const double em = defer ? *A.eta_mean : 0.0;
    const double cap = (V == QS_EXACT || defer) ? A.eta_cap : __builtin_inf();
    const unsigned sb = W.slab_bytes;
    const qs_rsrc E = qs_make_rsrc(A.eta, sb);
    if (wv <= 1) {
        const qs_rsrc U = qs_make_rsrc(A.uo, sb), Vv = qs_make_rsrc(A.vo, sb), T = qs_make_rsrc(fp->aux, sb), L = qs_make_rsrc(A.land, sb / 8u);
        if (wv == 0) { QsSrcOcnU<V, QS_PD_OCN> S{A, W, E, U, Vv, T, L, defer, em, cap}; return qs_del4_stream<V>(S, A.G, W, A.poleA, fp, A.sub_dt, out); }
        QsSrcOcnV<V, QS_PD_OCN> S{A, W, E, U, Vv, T, L, defer, em, cap};
        return qs_del4_stream<V>(S, A.G, W, A.poleA, fp, A.sub_dt, out);
    }
    QsSrcEta<V, QS_PD_OCN> S{A, W, E, defer, em, cap};
    return qs_del4_stream<V>(S, A.G, W, A.poleA, fp, A.sub_dt, out);
}


// host side (qd_stream.hip)
const double* qd_stream_tables(qd_ctx* c, int kind, int nf, const double* const* k4row, const double* k4s, const int* skip);
bool qd_stream_ocn_args(qd_ctx* c, const QdOcnArgs& P, QsOcnArgs& A);
