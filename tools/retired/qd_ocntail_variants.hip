// Retired forms of the ocean sub-step tail (moved out of libqingdai_hip.so in round 4; not compiled).
// k_ocn_tail (LDS tile, QD_OCN_TAIL=2), k_ocn_tail_tile (QD_OCN_TAIL=3), k_ocn_step (the whole sub-step in one launch,
// QD_OCN_TAIL=4): all bit-compatible with the shipped k_ocn_tail_fast / k_ocn_tail_stream, all measured slower
// (profiles/README.md, DESIGN.md section 4 "Round 3").  They need qd_ocntail.hip of commit 17c5f5f around them to build.

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

// =========================================================================================
// tile form with every load of a phase in flight at once (QD_OCN_TAIL=3)
// =========================================================================================
// Per-wave timelines of k_ocn_tail_stream (s_memrealtime stamps, round 3) showed what bounds it: its SST wave lives 22 us for 12 rows
// against 12 us for the currents wave's 8 -- each row of the SST wave is a serial chain (plain loads -> departure point -> four
// corner loads -> blend) and the strip's four halo rows of T1 are half as much work again.  This form keeps the LDS tile of
// k_ocn_tail (16 x 62 cells, 20 rows of T1, 1.25x instead of 1.5x) and removes the chains instead: a wave owns FIVE consecutive
// rows of T1 and issues the loads of all five before it touches the first, then all twenty corner loads, then blends; it owns FOUR
// consecutive rows of phase 2 and issues their 28 plain loads before the barrier.  Straight-line code (no row loop), so every
// s_waitcnt is exact, and the four waves of a workgroup do the same work.  Same expressions as the two kernels above.
__global__ void __launch_bounds__(256)
k_ocn_tail_tile(QdGeom G, QdTabs T, QdTailArgs P) {
    __shared__ double sT[QT_RA][64];
    __shared__ double sAcc[4];
    const unsigned wq = qd_xcd_chunk(blockIdx.x, gridDim.x);
    const int rs = (int)(wq / (unsigned)P.ntc), cs = (int)(wq % (unsigned)P.ntc);
    QtW W;
    W.n = G.nlat; W.m = G.nlon; W.lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jraw = cs * QT_TC - 1 + W.lane;
    W.j = jraw < 0 ? jraw + W.m : (jraw >= W.m ? jraw - W.m : jraw);
    W.own = W.lane >= 1 && W.lane <= QT_TC && jraw < W.m;
    W.vo = (unsigned)W.j * 8u; W.vo8 = (unsigned)W.j; W.vs = W.own ? (unsigned)jraw * 8u : 0x80000000u;
    W.slab = (unsigned)(G.lrows_ + QD_PAD_ROWS) * (unsigned)G.nlon * 8u;
    W.lbase = G.lbase; W.lrows = G.lrows_; W.own0 = 0; W.own1 = G.nlat;
    const int n = W.n;
    const int i0 = G.row0 + rs * QT_TR, i1 = min(i0 + QT_TR, G.row0 + G.nrows), ib = i0 - 2;
    W.o0 = i0; W.o1 = i1;
    const unsigned sb = W.slab;
    const qt_rsrc U = qt_make_rsrc(P.uo, sb), V = qt_make_rsrc(P.vo, sb), S = qt_make_rsrc(P.Ts, sb);
    // ---- phase 1: nan_to_num(T1) rows ib + 5 wv .. + 4 (ocean.py:380-382)
    {
        double u[5], v[5], ts[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const unsigned ro = qt_row(W, ib + 5 * wv + k);
            u[k] = qt_ld(U, ro, W.vo); v[k] = qt_ld(V, ro, W.vo); ts[k] = qt_ld(S, ro, W.vo);
        }
        QtGather gq[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) gq[k] = qt_gather_issue(G, T, P, qd_clampi(ib + 5 * wv + k, 0, n - 1), W.j, u[k], v[k]);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = ib + 5 * wv + k;
            const bool valid = i >= 0 && i < n && i < i1 + 2;
            const double t1 = qd_nn((1.0 - P.alpha) * ts[k] + P.alpha * qt_gather_use(gq[k]));
            sT[5 * wv + k][W.lane] = valid ? t1 : 0.0;
        }
    }
    // ---- the plain inputs of this wave's four rows of phase 2, in flight across the barrier
    const qt_rsrc E = qt_make_rsrc(P.eta, sb), L = qt_make_rsrc(P.land, sb / 8u);
    const qt_rsrc Q = qt_make_rsrc(P.use_q ? P.qnet : P.Ts, sb), I = qt_make_rsrc(P.has_ice ? P.ice : P.land, sb / 8u);
    const qt_rsrc UO = qt_make_rsrc(P.uo_out, sb), VO = qt_make_rsrc(P.vo_out, sb), SO = qt_make_rsrc(P.Ts_out, sb);
    const int r0 = i0 + 4 * wv;
    double uu[6], vv[6], ee[4], qq[4]; int ll[4], ii[4];
#pragma unroll
    for (int k = 0; k < 6; ++k) {                            // np.roll rows: mean4 reads row -1 as row n-1 and row n as row 0
        const unsigned rr = qt_row_roll(W, r0 - 1 + k);
        uu[k] = qt_ld(U, rr, W.vo); vv[k] = qt_ld(V, rr, W.vo);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned rc = qt_row(W, r0 + k);
        ee[k] = qt_ld(E, rc, W.vo); ll[k] = qt_ld8(L, rc, W.vo8); qq[k] = qt_ld(Q, rc, W.vo); ii[k] = qt_ld8(I, rc, W.vo8);
    }
    __syncthreads();
    // ---- phase 2: the owned cells of rows r0 .. r0 + 3
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = r0 + k;
        const bool rowok = i < i1;                           // the last tile row of the grid may be short
        const unsigned vs = rowok ? W.vs : 0x80000000u;
        const unsigned ro = qt_row(W, i);
        const double us = uu[k], uc = uu[k + 1], un = uu[k + 2], vs_ = vv[k], vc = vv[k + 1], vn = vv[k + 2];
        // continuity (ocean.py:365-374; grid.py:41-88 through qt_div_point's expressions)
        const double dp = qt_div(qd_east(uc) - qd_west(uc), 2 * P.dlon, P.r_2dlon);
        double dq = 0.0;
        if (i != 0 && i != n - 1) {
            const double qn = vn * qd_sload(T.cos_raw, qd_clampi(i + 1, 0, n - 1));
            const double qs = vs_ * qd_sload(T.cos_raw, qd_clampi(i - 1, 0, n - 1));
            dq = qt_div(qn - qs, 2 * P.dlat, P.r_2dlat);
        }
        const int ic_ = qd_clampi(i, 0, n - 1);
        const double div = qd_sload(T.inv_acos6, ic_) * (dp + dq);
        double e = ee[k] + P.msdtH * div;
        const bool island = ll[k] == 1;
        if (island) e = 0.0;
        qt_st(E, ro, vs, e);
        acc += (W.own && rowok) ? e * (island ? 0.0 : qd_sload(T.warea, ic_)) : 0.0;
        // K_h lap(T1) + heating (ocean.py:385-406, 440)
        {
            double Tv = sT[4 * wv + k + 2][W.lane];
            if (P.K_h > 0.0 && rowok) Tv = Tv + P.sub_dt * P.K_h * qt_lap_s(sT, ib, T, i, n, W.lane, P.dlat, P.dlon, P.a);
            if (P.use_q) {
                const double heat = qt_div(qq[k], P.rcH, P.r_rcH);
                const bool ocean = ll[k] == 0;
                if (P.has_ice) {
                    const bool ic = ii[k] != 0;
                    if (ocean && !ic) Tv = Tv + P.sub_dt * heat;
                    if (P.ice_qfac > 0.0 && ocean && ic) Tv = Tv + P.sub_dt * P.ice_qfac * heat;
                } else if (ocean) Tv = Tv + P.sub_dt * heat;
            }
            qt_st(SO, ro, vs, qd_nn(Tv));
        }
        // outliers + caps (ocean.py:409-434)
        {
            double u = qd_nn(uc), v = qd_nn(vc);
            const double cap = P.cap, s2 = u * u + v * v;
            // the lane neighbours are taken OUTSIDE the branch: a DPP move reads 0 from a lane that is not executing
            const double ue = qd_east(uc), uw = qd_west(uc), ve = qd_east(vc), vw = qd_west(vc);
            if (!(s2 < 0.81 * (cap * cap))) {
                const double speed = sqrt(s2);
                if (P.mean4) {
                    if (speed > cap) {
                        u = 0.25 * (qd_nn(un) + qd_nn(us) + qd_nn(ue) + qd_nn(uw));
                        v = 0.25 * (qd_nn(vn) + qd_nn(vs_) + qd_nn(ve) + qd_nn(vw));
                    }
                    const double sp2 = sqrt(u * u + v * v);
                    const double sc2 = (sp2 > cap) ? cap / (sp2 + 1e-12) : 1.0;
                    u = u * sc2; v = v * sc2;
                } else {
                    const double sc = (speed > cap) ? cap / (speed + 1e-12) : 1.0;
                    u = u * sc; v = v * sc;
                }
            }
            qt_st(UO, ro, vs, u); qt_st(VO, ro, vs, v);
        }
    }
    // ---- area-weighted eta sum of the tile (fixed order: lanes by shuffle tree, waves 0..3)
    acc = qt_wave_sum(acc);
    if (W.lane == 0) sAcc[wv] = acc;
    __syncthreads();
    if (wv == 0) {
        const double tile = ((sAcc[0] + sAcc[1]) + sAcc[2]) + sAcc[3];
        if (W.lane == 0) __hip_atomic_store(P.partial + wq, tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (P.acc) {                                         // eta mean inside this launch: the last workgroup to arrive finishes it
            const bool last = W.lane == 0 && qd_acc_arrive(P.acc, wq, gridDim.x, tile);
            if (__builtin_amdgcn_ballot_w64(last) != 0ull) {
                const double m = qd_acc_finish(P.acc, P.partial, (int)gridDim.x, P.wsum);
                if (W.lane == 0) *P.mean_out = m;
            }
        }
    }
}

// =========================================================================================
// the WHOLE ocean sub-step in one launch (QD_OCN_TAIL=4): momentum + del^4 streamed into LDS, then the tail from LDS
// =========================================================================================
// k_ocn_stream writes uo', vo', eta' (25 MB at 721 x 1440), k_ocn_tail_stream reads them back, and each launch pays its own ramp:
// the per-wave timelines of round 3 put ~7 us of every such launch into dispatch skew (1 us), a spread of wave end times (3.5 us) and
// the launch boundary (2.4 us).  Here a 256-thread workgroup owns a tile of <= 16 rows x 56 columns (lanes 4 .. 59 of 64):
//   phase A   waves 0, 1, 2 are the uo / vo / eta waves of qd_stream.h (same row sources, same del^4 pipeline) whose output rows go to
//             LDS planes instead of global memory: uo', vo' on the tile's rows +- 2 (the rows the SST advection needs), eta' on the
//             tile's rows.  (Wave 3 has nothing to stream; its SIMD slot serves the other workgroups of the CU.)
//   phase B1  every wave forms five rows of T1 = nan_to_num((1-alpha) Ts + alpha gather(Ts)) from uo', vo' in LDS (k_ocn_tail_tile's
//             phase 1: all plain loads, then all corner loads, then the blends) -> LDS plane.
//   phase B2  every wave finishes four rows: continuity, K_h lap(T1) + heating, outlier filter and caps, the tile's eta sum.
// np.roll(axis=0) couples the two poles (mean4 of the outlier filter reads row -1 as row n-1): ONE workgroup per column strip owns
// both polar tiles (rows 0 .. 6 and n-7 .. n-1, two segments in its planes), so the wrap neighbours are in its own LDS.  Every other
// tile keeps five rows between itself and a pole, which is what the non-polar prologue of the row stream needs.
// Non-finite values: a wave of phase A that saw one raises the workgroup's flag and all three redo their strips with the EXACT
// variant (phase B is the literal arithmetic anyway).
#define QF_TC 56
#define QF_TR 16
#define QF_TA (QF_TR + 4)
#define QF_PH 7                   // rows of a polar tile

struct QfArgs { int ntc, nmid, pad0_, pad1_; };

template <int V>
__device__ __forceinline__ bool qf_phase_a(const QsOcnArgs& A, const QsOcnArgs QD_CONST* Ak, QsW& W, int wv, int r0, int r1, double* plane) {
    const QsRec QD_CONST* fp = &Ak->rec[wv];
    W.o0 = r0; W.o1 = r1;
    QsOutLds out{plane + W.lane};
    return qs_ocn_wave<V>(A, W, wv, fp, out);
}

__global__ void __launch_bounds__(256, 4)
k_ocn_step(QsOcnArgs A, QdTabs T, QdTailArgs P, QfArgs F) {
    __shared__ double sU[QF_TA][64];
    __shared__ double sV[QF_TA][64];
    __shared__ double sE[QF_TR][64];
    __shared__ double sT[QF_TA][64];
    __shared__ double sAcc[4];
    const QdGeom& G = A.G;
    const unsigned wq = qd_xcd_chunk(blockIdx.x, gridDim.x);
    const int rs = (int)(wq / (unsigned)F.ntc), cs = (int)(wq % (unsigned)F.ntc);
    const int n = G.nlat, m = G.nlon;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- the tile's row segments: owned rows [o0, o1), plane rows [t0, t1) = owned +- 2 inside the domain
    int nseg, o0[2], o1[2], t0[2], t1[2];
    if (rs == 0) { nseg = 2; o0[0] = 0; o1[0] = QF_PH; o0[1] = n - QF_PH; o1[1] = n; }
    else {
        const int mid = n - 2 * QF_PH;
        nseg = 1;
        o0[0] = QF_PH + (int)(((long long)(rs - 1) * mid) / F.nmid); o1[0] = QF_PH + (int)(((long long)rs * mid) / F.nmid);
        o0[1] = o1[1] = 0;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) { t0[k] = max(o0[k] - 2, 0); t1[k] = k < nseg ? min(o1[k] + 2, n) : t0[k]; }
    const int lenA = t1[0] - t0[0], lenT = lenA + (t1[1] - t0[1]);
    const int nA = o1[0] - o0[0], nO = nA + (o1[1] - o0[1]);
    // ---- lane <-> column
    const int jraw = cs * QF_TC - 4 + lane;
    const int j = jraw < 0 ? jraw + m : (jraw >= m ? jraw - m : jraw);
    const bool own = lane >= 4 && lane < 4 + QF_TC && jraw < m;
    const unsigned slab = (unsigned)(G.lrows_ + QD_PAD_ROWS) * (unsigned)m * 8u;
    QT_STAMP(0);
    // ---- phase A
    {
        QsW W;
        W.n = n; W.nlon = m; W.lane = lane; W.j = j; W.o0 = 0; W.o1 = 0;
        W.vo = (unsigned)j * 8u; W.vo8 = (unsigned)j; W.vs = QS_OOB; W.slab_bytes = slab; W.west_edge = false; W.east_edge = false;
        const QsOcnArgs QD_CONST* Ak = (const QsOcnArgs QD_CONST*)__builtin_amdgcn_kernarg_segment_ptr();
        // the wave's plane and its rows in the two segments (eta: owned rows; uo, vo: owned +- 2)
        double* const plane = wv == 0 ? &sU[0][0] : (wv == 1 ? &sV[0][0] : &sE[0][0]);
        const int a0 = wv == 2 ? o0[0] : t0[0], a1 = wv == 2 ? o1[0] : t1[0], b0 = wv == 2 ? o0[1] : t0[1], b1 = wv == 2 ? o1[1] : t1[1];
        bool bad = false;
        if (wv < 3 && !A.exact) {
            bad = qf_phase_a<QS_FAST>(A, Ak, W, wv, a0, a1, plane);
            if (nseg > 1) bad |= qf_phase_a<QS_FAST>(A, Ak, W, wv, b0, b1, plane + (size_t)(a1 - a0) * 64);
        }
        if (__syncthreads_or((A.exact || bad) ? 1 : 0)) {
            if (wv < 3) {
                qf_phase_a<QS_EXACT>(A, Ak, W, wv, a0, a1, plane);
                if (nseg > 1) qf_phase_a<QS_EXACT>(A, Ak, W, wv, b0, b1, plane + (size_t)(a1 - a0) * 64);
            }
            __syncthreads();
        }
    }
    QT_STAMP(1);
    QtW W;
    W.n = n; W.m = m; W.lane = lane; W.j = j; W.own = own; W.o0 = 0; W.o1 = 0;
    W.vo = (unsigned)j * 8u; W.vo8 = (unsigned)j; W.vs = own ? (unsigned)jraw * 8u : 0x80000000u; W.slab = slab;
    W.lbase = G.lbase; W.lrows = G.lrows_; W.own0 = 0; W.own1 = n;
    const unsigned sb = slab;
    // ---- phase B1: T1 rows of slots 5 wv .. 5 wv + 4 (slot = plane row)
    {
        const qt_rsrc S = qt_make_rsrc(P.Ts, sb);
        double ts[5]; int gr[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int sl = 5 * wv + k;
            gr[k] = qd_clampi(sl < lenA ? t0[0] + sl : t0[1] + (sl - lenA), 0, n - 1);
            ts[k] = qt_ld(S, qt_row(W, gr[k]), W.vo);
        }
        QtGather gq[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int sl = min(5 * wv + k, QF_TA - 1);
            gq[k] = qt_gather_issue(G, T, P, gr[k], j, sU[sl][lane], sV[sl][lane]);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int sl = 5 * wv + k;
            const double t1v = qd_nn((1.0 - P.alpha) * ts[k] + P.alpha * qt_gather_use(gq[k]));
            if (sl < QF_TA) sT[sl][lane] = sl < lenT ? t1v : 0.0;
        }
    }
    // ---- the plain inputs of this wave's four rows of phase B2 (in flight across the barrier)
    const qt_rsrc L = qt_make_rsrc(P.land, sb / 8u);
    const qt_rsrc Q = qt_make_rsrc(P.use_q ? P.qnet : P.Ts, sb), I = qt_make_rsrc(P.has_ice ? P.ice : P.land, sb / 8u);
    const qt_rsrc EO = qt_make_rsrc(P.eta, sb), UO = qt_make_rsrc(P.uo_out, sb), VO = qt_make_rsrc(P.vo_out, sb), SO = qt_make_rsrc(P.Ts_out, sb);
    double qq[4]; int ll[4], ii[4], row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sl = 4 * wv + k;
        row[k] = qd_clampi(sl < nA ? o0[0] + sl : o0[1] + (sl - nA), 0, n - 1);
        const unsigned rc = qt_row(W, row[k]);
        ll[k] = qt_ld8(L, rc, W.vo8); qq[k] = qt_ld(Q, rc, W.vo); ii[k] = qt_ld8(I, rc, W.vo8);
    }
    __syncthreads();
    QT_STAMP(3);
    // ---- phase B2: the owned cells of slots 4 wv .. 4 wv + 3
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sl = 4 * wv + k;
        const int i = row[k];
        const bool rowok = sl < nO;
        const bool inA = sl < nA;
        const unsigned vs = rowok ? W.vs : 0x80000000u;
        const unsigned ro = qt_row(W, i);
        // plane rows: inside the segment, and across the poles (np.roll) in the other segment of a polar tile
        const int base = inA ? t0[0] : t0[1] - lenA;                              // plane row of global row g in this segment: g - base
        int gs = i - 1, gn = i + 1;
        int ps = gs - base, pn = gn - base;
        if (gs < 0) ps = lenA + (n - 1 - t0[1]);                                  // row -1 is row n-1: last row of the north segment
        if (gn >= n) pn = 0 - t0[0];                                              // row n is row 0: first row of the south segment
        ps = qd_clampi(ps, 0, QF_TA - 1); pn = qd_clampi(pn, 0, QF_TA - 1);
        const int pc = qd_clampi(i - base, 0, QF_TA - 1);
        const double us = sU[ps][lane], uc = sU[pc][lane], un = sU[pn][lane];
        const double vs_ = sV[ps][lane], vc = sV[pc][lane], vn = sV[pn][lane];
        // continuity (ocean.py:365-374; grid.py:41-88 through qt_div_point's expressions)
        const double dp = qt_div(qd_east(uc) - qd_west(uc), 2 * P.dlon, P.r_2dlon);
        double dq = 0.0;
        if (i != 0 && i != n - 1) {
            const double qn = vn * qd_sload(T.cos_raw, i + 1);
            const double qs = vs_ * qd_sload(T.cos_raw, i - 1);
            dq = qt_div(qn - qs, 2 * P.dlat, P.r_2dlat);
        }
        const double div = qd_sload(T.inv_acos6, i) * (dp + dq);
        double e = sE[min(sl, QF_TR - 1)][lane] + P.msdtH * div;
        const bool island = ll[k] == 1;
        if (island) e = 0.0;
        qt_st(EO, ro, vs, e);
        acc += (own && rowok) ? e * (island ? 0.0 : qd_sload(T.warea, i)) : 0.0;
        // K_h lap(T1) + heating (ocean.py:385-406, 440)
        {
            double Tv = sT[pc][lane];
            if (P.K_h > 0.0 && rowok) Tv = Tv + P.sub_dt * P.K_h * qt_lap_s(sT, base, T, i, n, lane, P.dlat, P.dlon, P.a);
            if (P.use_q) {
                const double heat = qt_div(qq[k], P.rcH, P.r_rcH);
                const bool ocean = ll[k] == 0;
                if (P.has_ice) {
                    const bool ic = ii[k] != 0;
                    if (ocean && !ic) Tv = Tv + P.sub_dt * heat;
                    if (P.ice_qfac > 0.0 && ocean && ic) Tv = Tv + P.sub_dt * P.ice_qfac * heat;
                } else if (ocean) Tv = Tv + P.sub_dt * heat;
            }
            qt_st(SO, ro, vs, qd_nn(Tv));
        }
        // outliers + caps (ocean.py:409-434)
        {
            double u = qd_nn(uc), v = qd_nn(vc);
            const double cap = P.cap, s2 = u * u + v * v;
            // the lane neighbours are taken OUTSIDE the branch: a DPP move reads 0 from a lane that is not executing
            const double ue = qd_east(uc), uw = qd_west(uc), ve = qd_east(vc), vw = qd_west(vc);
            if (!(s2 < 0.81 * (cap * cap))) {
                const double speed = sqrt(s2);
                if (P.mean4) {
                    if (speed > cap) {
                        u = 0.25 * (qd_nn(un) + qd_nn(us) + qd_nn(ue) + qd_nn(uw));
                        v = 0.25 * (qd_nn(vn) + qd_nn(vs_) + qd_nn(ve) + qd_nn(vw));
                    }
                    const double sp2 = sqrt(u * u + v * v);
                    const double sc2 = (sp2 > cap) ? cap / (sp2 + 1e-12) : 1.0;
                    u = u * sc2; v = v * sc2;
                } else {
                    const double sc = (speed > cap) ? cap / (speed + 1e-12) : 1.0;
                    u = u * sc; v = v * sc;
                }
            }
            qt_st(UO, ro, vs, u); qt_st(VO, ro, vs, v);
        }
    }
    QT_STAMP(2);
    // ---- area-weighted eta sum of the tile (fixed order: lanes by shuffle tree, waves 0..3)
    acc = qt_wave_sum(acc);
    if (lane == 0) sAcc[wv] = acc;
    __syncthreads();
    if (wv == 0) {
        const double tile = ((sAcc[0] + sAcc[1]) + sAcc[2]) + sAcc[3];
        if (lane == 0) __hip_atomic_store(P.partial + wq, tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (P.acc) {                                         // eta mean inside this launch: the last workgroup to arrive finishes it
            const bool last = lane == 0 && qd_acc_arrive(P.acc, wq, gridDim.x, tile);
            if (__builtin_amdgcn_ballot_w64(last) != 0ull) {
                const double mm = qd_acc_finish(P.acc, P.partial, (int)gridDim.x, P.wsum);
                if (lane == 0) *P.mean_out = mm;
            }
        }
    }
}

// shape of the fused launch; false: the grid is too small for polar tiles of QF_PH rows + a middle (the caller takes the two-kernel path)
static bool qf_shape(const QdGeom& G, QfArgs& F) {
    if (!G.full || G.nlon < 64 || G.nlat < 2 * QF_PH + 12) return false;
    const int mid = G.nlat - 2 * QF_PH;
    F.ntc = (G.nlon + QF_TC - 1) / QF_TC;
    F.nmid = (mid + QF_TR - 1) / QF_TR;
    F.pad0_ = F.pad1_ = 0;
    return true;
}
bool qd_ocn_step_ok(const qd_ctx* c) {
    QfArgs F;
    return c->geo.full && c->fused_fast >= 1 && c->fused_fast <= 2 &&
           (size_t)(c->geo.lrows_ + QD_PAD_ROWS) * (size_t)c->geo.nlon * 8u < 0x7fffffffull && qf_shape(c->geo, F);
}
int qd_ocn_step_tiles(const qd_ctx* c) {
    QfArgs F;
    return qf_shape(c->geo, F) ? F.ntc * (1 + F.nmid) : 0;
}
// O: the momentum half's arguments (outputs unused), P: the tail's (P.uo / P.vo unused, P.eta = the NEW eta slab)
int qd_launch_ocn_step(qd_ctx* c, const QdOcnArgs& O, QdTailArgs& P) {
    QfArgs F;
    if (!qf_shape(c->geo, F)) return qd_fail(c, "k_ocn_step: whole-globe handles of >= 64 columns and >= 26 rows only");
    QsOcnArgs A;
    if (!qd_stream_ocn_args(c, O, A)) return qd_fail(c, "k_ocn_step: coefficient row tables");
    A.G = c->geo;
    if (F.ntc * (1 + F.nmid) > c->red_blocks) return qd_fail(c, "k_ocn_step: partial buffer too small");
    QdScope sc(c, "ocean_step", true);
#ifdef QT_STAMPS
    static unsigned long long* stamps = nullptr;
    const size_t stamp_words = (size_t)8 * 2 * 8192;
    if (!stamps) { hipMalloc(&stamps, stamp_words * 8); hipMemcpyToSymbol(HIP_SYMBOL(qt_stamp_buf), &stamps, sizeof(stamps)); }
    hipMemsetAsync(stamps, 0, stamp_words * 8, c->stream);
#endif
    QD_LAUNCH_TIMED(sc, k_ocn_step, dim3(F.ntc * (1 + F.nmid)), dim3(256), c->stream, A, c->tabs, P, F);
#ifdef QT_STAMPS
    if (const char* f = std::getenv("QD_STAMPS_FILE")) {
        std::vector<unsigned long long> h(stamp_words);
        hipStreamSynchronize(c->stream);
        hipMemcpy(h.data(), stamps, stamp_words * 8, hipMemcpyDeviceToHost);
        if (FILE* fp = std::fopen(f, "wb")) { std::fwrite(h.data(), 8, stamp_words, fp); std::fclose(fp); }
    }
#endif
    return 0;
}


// qt_lap with the row tables read through the scalar cache (a plain T.lapA[1][i] of a by-value table block is a vector-memory load
// of a wave-uniform address: it joins the vmcnt queue behind every row in flight)
__device__ __forceinline__ double qt_lap_s(const double (*sT)[64], int ib, const QdTabs& T, int i, int n, int l,
                                           double dphi, double dlam, double a) {
    const double cc = QT_F(i, l);
    if (i >= 2 && i <= n - 3) {
        const double Gb = qd_sload(T.lapA[1], i + 1) * (QT_F(i + 2, l) - cc);
        const double Ga = qd_sload(T.lapA[1], i - 1) * (cc - QT_F(i - 2, l));
        const double d2 = (QT_F(i, l + 1) - 2.0 * cc) + QT_F(i, l - 1);
        return qd_sload(T.lapP[1], i) * (Gb - Ga) + qd_sload(T.lapQ[1], i) * d2;
    }
    const double* __restrict__ cosf = T.cos05;
    int ra, rb; double den;
    if (i == 0) { ra = 0; rb = 1; den = dphi; }
    else if (i == n - 1) { ra = n - 2; rb = n - 1; den = dphi; }
    else { ra = i - 1; rb = i + 1; den = 2.0 * dphi; }
    const double Ga = qd_sload(cosf, ra) * qt_dphi(sT, ib, n, ra, l, dphi);
    const double Gb = qd_sload(cosf, rb) * qt_dphi(sT, ib, n, rb, l, dphi);
    const double ci = qd_sload(cosf, i);
    const double term_phi = (1.0 / ci) * ((Gb - Ga) / den);
    const double d2 = ((QT_F(i, l + 1) - 2.0 * cc) + QT_F(i, l - 1)) / (dlam * dlam);
    const double term_lam = d2 / (ci * ci);
    return (term_phi + term_lam) / (a * a);
}

