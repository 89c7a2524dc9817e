// qd_stream.hip -- row-streaming form of the fused momentum + del^4 kernels (gfx950): the hot stencil of the path.
//
//   k_dyn_stream   atmosphere: np.gradient(h) -> geostrophic-relaxation | primitive momentum (dynamics.py:482-530)
//                  -> del^4 hyperdiffusion of u, v, h, q, cloud (dynamics.py:533-594, 144-212), ONE launch.
//   k_ocn_stream   ocean sub-step: grad(eta) + Coriolis + wind stress + drag + land mask + polar sponge
//                  (ocean.py:306-336) -> del^4 of uo, vo, eta (ocean.py:341-356).
//
// Same arithmetic on every owned cell, bit for bit, as the LDS-tiled kernels of qd_fused.hip (which stay as the EXACT reference path and
// for grids narrower than one wavefront); different machine mapping:
//   * A WAVEFRONT owns one field of one strip: 58 owned columns (64 lanes, 3 halo lanes each side; lane l is global
//     column 58*cs-3+l, so every global access is one coalesced 512-byte row segment) x R owned rows, and marches down
//     the rows.  The latitude part of the spherical Laplacian, gradient(cos*gradient(F)), only touches rows r-2, r, r+2,
//     so del^4 is a two-stage pipeline over the row stream (see "del^4 of a row stream" below): 10 doubles of state per
//     lane, no LDS, no barrier.  Waves are independent: the loads of the next rows are in flight while a row is being
//     worked on, and the waves of a SIMD cover each other's latencies.  East / west neighbours are DPP wave shifts.
//   * Memory access is through buffer instructions: wave-uniform row offset in an SGPR, constant per-lane column offset
//     in a VGPR (no per-access address arithmetic), and the hardware range check drops the stores of the halo lanes
//     (their offset is out of range) -- no exec masking, no branch: the row loop is straight-line code, which is what
//     lets the compiler's s_waitcnt placement keep several rows of loads and stores in flight.
//   * The workgroup is the set of fields of one strip (5 waves: u, v, h, q, cloud; 3: uo, vo, eta) so that the rows of h
//     / eta the momentum waves re-read come from the CU's L1.  Halo rows (4 above, 4 below; 5 for h) are recomputed.
//   * Per-row coefficients {A[r-1], A[r+1], P[r], Q[r]} are packed so that one scalar load fetches a row's set.
//   * nan_to_num is the identity on finite values: the FAST variant omits it and tests every stored value (and every
//     clip input) with one v_cmp_class; a WAVE that saw a non-finite value re-runs its strip with the EXACT variant
//     (literal nan_to_num / np.clip at the reference's places; outputs go to separate buffers, so that is safe).
//   * Strips next to a pole take the POLE variant: the one-sided np.gradient rows are two extra differences.
#include "qd_internal.h"
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <algorithm>

// Diagnostic build only (-DQS_STAMPS, tools/build_variant.sh): every wave of k_dyn_stream writes {s_memrealtime at entry, after its
// prologue, at exit; HW_ID; XCC_ID} to a buffer the launcher dumps to $QD_STAMPS_FILE -- the per-wave timeline of one launch.
#ifdef QS_STAMPS
__device__ unsigned long long* qs_stamp_buf;
__device__ __forceinline__ void qs_stamp(int slot) {
    if ((threadIdx.x & 63) == 0) {
        const unsigned wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        qs_stamp_buf[(size_t)wid * 8 + slot] = __builtin_amdgcn_s_memrealtime();
        if (slot == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            qs_stamp_buf[(size_t)wid * 8 + 4] = hw; qs_stamp_buf[(size_t)wid * 8 + 5] = xcc;
        }
    }
}
#define QS_STAMP(k) qs_stamp(k)
#endif

#include "qd_stream.h"
#include "qd_band.h"

// FAST pass; a wave that met a non-finite value repeats its strip with the EXACT arithmetic
template <bool PRIM>
__global__ void __launch_bounds__(320)
k_dyn_stream(QsDynArgs A) {
    QsW W;
    qs_strip(A.G, A.vb, A.ntc, A.nrs, W);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv <= 1) __builtin_amdgcn_s_setprio(2);             // the momentum waves are the long ones of a strip (measured: -5 % on k_ocn_stream)
    QS_STAMP(0);
    if (!A.exact) {
        const bool bad = qs_dyn_wave<PRIM, QS_FAST>(A, W, wv);
        QS_STAMP(2);
        if (__builtin_amdgcn_ballot_w64(bad) == 0ull) return;
    }
    qs_dyn_wave<PRIM, QS_EXACT>(A, W, wv);
}

__global__ void __launch_bounds__(192)
k_ocn_stream(QsOcnArgs A) {
    QsW W;
    qs_strip(A.G, A.vb, A.ntc, A.nrs, W);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv <= 1) __builtin_amdgcn_s_setprio(2);             // the momentum waves are the long ones of a strip (measured: -5 % on k_ocn_stream)
    const QsOcnArgs QD_CONST* Ak = (const QsOcnArgs QD_CONST*)__builtin_amdgcn_kernarg_segment_ptr();
    const QsRec QD_CONST* fp = &Ak->rec[wv];
    if (!A.exact) {
        QsOutGlobal out{qs_make_rsrc(fp->out, W.slab_bytes), qs_off(A.G, W.o0), W.vs, (unsigned)W.nlon};
        const bool bad = qs_ocn_wave<QS_FAST>(A, W, wv, fp, out);
        if (__builtin_amdgcn_ballot_w64(bad) == 0ull) return;
    }
    QsOutGlobal out{qs_make_rsrc(fp->out, W.slab_bytes), qs_off(A.G, W.o0), W.vs, (unsigned)W.nlon};
    qs_ocn_wave<QS_EXACT>(A, W, wv, fp, out);
}

// =========================================================================================
// host side
// =========================================================================================
static int qs_wgs_per_cu(qd_ctx* c, int which) {
    if (c->qs_wgs_per_cu[which] > 0) return c->qs_wgs_per_cu[which];
    int nb = 0;
    hipError_t e = hipSuccess;
    if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_dyn_stream<false>, 320, 0);
    else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_dyn_stream<true>, 320, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_ocn_stream, 192, 0);
    if (e != hipSuccess || nb < 1) nb = which == 2 ? 6 : 4;
    c->qs_wgs_per_cu[which] = nb;
    return nb;
}

// Number of strips per column of strips.  Measured on MI355X (rocprofv3 kernel trace, 721 x 1440): the launch is fastest with about
// three waves per SIMD (k_dyn_stream: 30 strips of 24 rows -> 750 workgroups, 24 us; 16 rows -> 27 us; 32 -> 33 us; k_ocn_stream:
// 40 strips of 18 rows -> 1000 workgroups, 18.6 us; 1125 -> 19.2; 750 -> 21.2; 1500 -> 20.0) -- fewer waves leave the memory pipeline idle between a wave's rows, more waves
// recompute more halo rows (8 per strip) and evict each other's rows from L2.  The same workgroup counts hold at 1441 x 2880 (one
// round of taller strips: k_dyn_stream 79-89 us = 0.52-0.58 of the HBM peak with 15-20 strips of 72-96 rows, against 91-124 us for
// strip heights that leave a long last strip or fall between two rounds).  QD_STREAM_R* give a strip height instead.
struct QsShape { int R, nrs, vb; };
static QsShape qs_shape(qd_ctx* c, int nrows, int nlon, int which, bool north_pole) {
    const int ntc = (nlon + QS_TC - 1) / QS_TC;
    int R = 0;
    R = which == 2 ? c->tune.stream_r_ocn : c->tune.stream_r_dyn;      // tuning overrides (QD_STREAM_R_OCN / _DYN, QD_STREAM_R: read at create)
    if (R <= 0) R = c->stream_rows;
    int nrs;
    if (R > 0) nrs = std::max(1, nrows / std::max(R, 5));
    else {
        const long target = which == 2 ? 1000 : 760;         // workgroups in flight
        nrs = (int)std::max(1L, (target + ntc / 2) / ntc);
    }
    nrs = std::max(1, std::min(nrs, nrows / 12));            // strips of at least 12 rows
    int vb = 0;
    if (north_pole && nrs > 1) {
        vb = 6;
        if (c->tune.stream_vb >= 0) vb = c->tune.stream_vb;                                          // QD_STREAM_VB (read at create)
        while (vb > 0 && nrows - qs_cut(nrows, nrs, vb, nrs - 1) < 12) --vb;                         // the pole strip keeps >= 12 rows
    }
    return QsShape{(nrows + nrs - 1) / nrs, nrs, vb};
}

static int host_lrow(const QdGeom& G, int g) {
    int l = g - G.lbase;
    if (l < 0) l += G.nlat; else if (l >= G.nlat) l -= G.nlat;
    return l;
}

// every row segment of the launch must start / end at a pole or at least five rows away from it
static bool qs_segments_ok(const qd_ctx* c, int margin) {
    QdSegs S = qd_segments(const_cast<qd_ctx*>(c), margin);
    for (int k = 0; k < S.n; ++k) {
        const QdGeom& G = S.g[k];
        const int s0 = G.row0, s1 = G.row0 + G.nrows;
        if ((s0 > 0 && s0 < 5) || (s1 < G.nlat && s1 > G.nlat - 5) || G.nrows < 10) return false;
    }
    return true;
}

bool qd_stream_ok(const qd_ctx* c, int margin) {
    return c->fused_fast >= 1 && c->fused_fast <= 2 && c->geo.nlon >= 64 && c->geo.nlat >= 12 &&
           (size_t)(c->geo.lrows_ + QD_PAD_ROWS) * (size_t)c->geo.nlon * 8u < 0x7fffffffull &&          // buffer range: 31-bit byte counts
           qs_segments_ok(c, margin);
}

// packed row tables {lapA[r+1], lapP[r], lapQ[r], k4[r]} per field; a scalar k4 override (QD_K4_U, ...) fills the column
const double* qd_stream_tables(qd_ctx* c, int kind, int nf, const double* const* k4row, const double* k4s, const int* skip) {
    const int nlat = c->geo.nlat;
    double key[8] = {1.0, 0, 0, 0, 0, 0, 0, 0};
    for (int f = 0; f < nf; ++f) key[1 + f] = (k4row[f] || skip[f]) ? 0.0 : k4s[f];
    bool same = c->qs_tab[kind] != nullptr && c->qs_key[kind][0] == 1.0;
    for (int f = 0; same && f < nf; ++f) same = c->qs_key[kind][1 + f] == key[1 + f] || (key[1 + f] != key[1 + f] && c->qs_key[kind][1 + f] != c->qs_key[kind][1 + f]);
    if (same) return c->qs_tab[kind];
    if (c->h_lapK[kind].size() != (size_t)nlat * 4) return nullptr;
    std::vector<double> t((size_t)nf * nlat * 4);
    for (int f = 0; f < nf; ++f)
        for (int r = 0; r < nlat; ++r) {
            double* o = &t[((size_t)f * nlat + r) * 4];
            o[0] = c->h_lapK[kind][4 * r + 1]; o[1] = c->h_lapK[kind][4 * r + 2]; o[2] = c->h_lapK[kind][4 * r + 3];
            o[3] = k4row[f] ? (c->h_k4[kind].size() == (size_t)nf * nlat ? c->h_k4[kind][(size_t)f * nlat + r] : 0.0) : k4s[f];
        }
    if (!c->qs_tab[kind] && hipMalloc(&c->qs_tab[kind], (size_t)5 * nlat * 4 * sizeof(double)) != hipSuccess) return nullptr;
    if (hipMemcpyAsync(c->qs_tab[kind], t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) return nullptr;
    hipStreamSynchronize(c->stream);                         // `t` is pageable and about to go away
    for (int k = 0; k < 8; ++k) c->qs_key[kind][k] = key[k];
    return c->qs_tab[kind];
}

int qd_launch_dyn_stream(qd_ctx* c, const QdDynArgs& P, int margin) {
    QsDynArgs A;
    A.poleA = c->tabs.lapPoleA[0];
    A.c8 = P.primitive ? c->tabs.mom_px : c->tabs.mom_cu;
    A.c9 = P.primitive ? c->tabs.fcor : c->tabs.mom_cv;
    A.h = P.h; A.fric = P.fric;
    const double* tab = qd_stream_tables(c, 0, 5, P.k4row, P.k4s, P.skip);
    if (!tab) return qd_fail(c, "fused kernel: coefficient row tables");
    const double* in[5] = {P.u, P.v, P.h, P.q, P.cloud};
    const double* aux[5] = {P.v, P.u, nullptr, nullptr, nullptr};
    double* out[5] = {P.uo, P.vo, P.ho, P.qo, P.co};
    for (int f = 0; f < 5; ++f) A.rec[f] = QsRec{in[f], aux[f], out[f], tab + (size_t)f * c->geo.nlat * 4, P.skip[f], 0};
    A.dt = P.dt; A.inv_dlon = P.inv_dlon; A.inv_2dlon = P.inv_2dlon; A.inv_dlat = P.inv_dlat; A.inv_2dlat = P.inv_2dlat; A.pgf_y = P.pgf_y;
    A.exact = c->fused_fast == 2;
    QdScope sc(c, "k_dyn_hyper", c->geo.full != 0);          // whole globe: one launch, timed by the dispatch itself
#ifdef QS_STAMPS
    static unsigned long long* stamps = nullptr;
    const size_t stamp_words = (size_t)8 * 5 * 4096;
    if (!stamps) { hipMalloc(&stamps, stamp_words * 8); hipMemcpyToSymbol(HIP_SYMBOL(qs_stamp_buf), &stamps, sizeof(stamps)); }
    hipMemsetAsync(stamps, 0, stamp_words * 8, c->stream);
#endif
    QD_ROWS(c, margin, G,
            const QsShape sh = qs_shape(c, G.nrows, G.nlon, P.primitive ? 1 : 0, G.row0 + G.nrows == G.nlat);
            A.G = G; A.vb = sh.vb; A.nrs = sh.nrs; A.ntc = (G.nlon + QS_TC - 1) / QS_TC;
            if (P.primitive) QD_LAUNCH_TIMED(sc, k_dyn_stream<true>, dim3(A.nrs * A.ntc), dim3(320), c->stream, A);
            else QD_LAUNCH_TIMED(sc, k_dyn_stream<false>, dim3(A.nrs * A.ntc), dim3(320), c->stream, A));
#ifdef QS_STAMPS
    if (const char* f = std::getenv("QD_STAMPS_FILE")) {     // the LAST launch's stamps stay in the file
        std::vector<unsigned long long> h(stamp_words);
        hipStreamSynchronize(c->stream);
        hipMemcpy(h.data(), stamps, stamp_words * 8, hipMemcpyDeviceToHost);
        if (FILE* fp = std::fopen(f, "wb")) { std::fwrite(h.data(), 8, stamp_words, fp); std::fclose(fp); }
    }
#endif
    return 0;
}

// a segment that holds a pole row also reads the other pole's eta row (np.roll): it must be on this slab
bool qd_ocn_stream_ok(const qd_ctx* c, int margin) {
    if (!qd_stream_ok(c, margin)) return false;
    if (c->geo.full) return true;
    QdSegs S = qd_segments(const_cast<qd_ctx*>(c), margin);
    for (int k = 0; k < S.n; ++k) {
        const QdGeom& G = S.g[k];
        const bool pole = G.row0 == 0 || G.row0 + G.nrows == G.nlat;
        if (pole && !(host_lrow(G, 0) < G.lrows_ && host_lrow(G, G.nlat - 1) < G.lrows_)) return false;
    }
    return true;
}

// the stream form of the ocean momentum arguments (everything but the geometry / strip shape); false: no coefficient tables
bool qd_stream_ocn_args(qd_ctx* c, const QdOcnArgs& P, QsOcnArgs& A) {
    A.poleA = c->tabs.lapPoleA[1];
    A.fcor = c->tabs.fcor; A.igx = c->tabs.ocn_igx; A.rx = c->tabs.r_extra;
    A.uo = P.uo; A.vo = P.vo; A.eta = P.eta; A.land = P.land;
    const double* tab = qd_stream_tables(c, 1, 3, P.k4row, P.k4s, P.skip);
    if (!tab) return false;
    const double* in[3] = {P.uo, P.vo, P.eta};
    const double* aux[3] = {P.taux, P.tauy, nullptr};
    double* out[3] = {P.uo_out, P.vo_out, P.eta_out};
    for (int f = 0; f < 3; ++f) A.rec[f] = QsRec{in[f], aux[f], out[f], tab + (size_t)f * c->geo.nlat * 4, P.skip[f], 0};
    A.eta_mean = P.eta_mean; A.eta_cap = P.eta_cap; A.sub_dt = P.sub_dt; A.g = P.g; A.r_bot = P.r_bot;
    A.inv_2dlon = P.inv_2dlon; A.inv_2dlat = P.inv_2dlat; A.inv_a = P.inv_a; A.inv_rhoH = P.inv_rhoH;
    A.exact = c->fused_fast == 2;
    A.vb = 0; A.ntc = 0; A.nrs = 0;
    return true;
}

static bool qs_seg_ok(const QdGeom& G) {
    const int s0 = G.row0, s1 = G.row0 + G.nrows;
    return !((s0 > 0 && s0 < 5) || (s1 < G.nlat && s1 > G.nlat - 5) || G.nrows < 10);
}
// the same conditions as qd_ocn_stream_ok, for an explicit list of row segments (interior / boundary launches around a halo exchange)
bool qd_ocn_stream_ok_list(const qd_ctx* c, const QdSegList& S) {
    if (!(c->fused_fast >= 1 && c->fused_fast <= 2 && c->geo.nlon >= 64 && c->geo.nlat >= 12 &&
          (size_t)(c->geo.lrows_ + QD_PAD_ROWS) * (size_t)c->geo.nlon * 8u < 0x7fffffffull)) return false;
    for (int k = 0; k < S.n; ++k) {
        const QdGeom& G = S.g[k];
        if (!qs_seg_ok(G)) return false;
        const bool pole = G.row0 == 0 || G.row0 + G.nrows == G.nlat;
        if (!c->geo.full && pole && !(host_lrow(G, 0) < G.lrows_ && host_lrow(G, G.nlat - 1) < G.lrows_)) return false;
    }
    return S.n > 0;
}

int qd_launch_ocn_stream_list(qd_ctx* c, const QdOcnArgs& P, const QdSegList& S) {
    QsOcnArgs A;
    if (!qd_stream_ocn_args(c, P, A)) return qd_fail(c, "fused kernel: coefficient row tables");
    if (S.n == 2 && !c->tune.stream_no_pair && !qd_peer_job_waiting(c)) {                // the two boundary segments of a split launch: one launch (k_ocn_stream_pair)
        const QsShape s0 = qs_shape(c, S.g[0].nrows, S.g[0].nlon, 2, S.g[0].row0 + S.g[0].nrows == S.g[0].nlat);
        const QsShape s1 = qs_shape(c, S.g[1].nrows, S.g[1].nlon, 2, S.g[1].row0 + S.g[1].nrows == S.g[1].nlat);
        A.G = S.g[0]; A.vb = s0.vb; A.nrs = s0.nrs; A.ntc = (S.g[0].nlon + QS_TC - 1) / QS_TC;
        qd_launch_ocn_stream_pair(c, A, S.g[1], s1.vb, s1.nrs, (S.g[1].nlon + QS_TC - 1) / QS_TC);
        return 0;
    }
    for (int k = 0; k < S.n; ++k) {
        const QdGeom& G = S.g[k];
        const QsShape sh = qs_shape(c, G.nrows, G.nlon, 2, G.row0 + G.nrows == G.nlat);
        A.G = G; A.vb = sh.vb; A.nrs = sh.nrs; A.ntc = (G.nlon + QS_TC - 1) / QS_TC;
        // a halo push waits for a launch to carry it (qd_plan_begin): k_ocn_stream_push, its workgroups go first (qd_stream_push.hip)
        if (!qd_launch_ocn_stream_push(c, A)) hipLaunchKernelGGL(k_ocn_stream, dim3(A.nrs * A.ntc), dim3(192), 0, c->stream, A);
    }
    return 0;
}

int qd_launch_ocn_stream(qd_ctx* c, const QdOcnArgs& P, int margin) {
    QsOcnArgs A;
    A.poleA = c->tabs.lapPoleA[1];
    A.fcor = c->tabs.fcor; A.igx = c->tabs.ocn_igx; A.rx = c->tabs.r_extra;
    A.uo = P.uo; A.vo = P.vo; A.eta = P.eta; A.land = P.land;
    const double* tab = qd_stream_tables(c, 1, 3, P.k4row, P.k4s, P.skip);
    if (!tab) return qd_fail(c, "fused kernel: coefficient row tables");
    const double* in[3] = {P.uo, P.vo, P.eta};
    const double* aux[3] = {P.taux, P.tauy, nullptr};
    double* out[3] = {P.uo_out, P.vo_out, P.eta_out};
    for (int f = 0; f < 3; ++f) A.rec[f] = QsRec{in[f], aux[f], out[f], tab + (size_t)f * c->geo.nlat * 4, P.skip[f], 0};
    A.eta_mean = P.eta_mean; A.eta_cap = P.eta_cap; A.sub_dt = P.sub_dt; A.g = P.g; A.r_bot = P.r_bot;
    A.inv_2dlon = P.inv_2dlon; A.inv_2dlat = P.inv_2dlat; A.inv_a = P.inv_a; A.inv_rhoH = P.inv_rhoH;
    A.exact = c->fused_fast == 2;
    QdScope sc(c, "k_ocn_hyper", c->geo.full != 0);
    QD_ROWS(c, margin, G,
            const QsShape sh = qs_shape(c, G.nrows, G.nlon, 2, G.row0 + G.nrows == G.nlat);
            A.G = G; A.vb = sh.vb; A.nrs = sh.nrs; A.ntc = (G.nlon + QS_TC - 1) / QS_TC;
            QD_LAUNCH_TIMED(sc, k_ocn_stream, dim3(A.nrs * A.ntc), dim3(192), c->stream, A));
    return 0;
}
