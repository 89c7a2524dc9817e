// qd_phyto.hip -- phytoplankton tracers carried by the ocean currents, resident on the device.
//
// PhytoManager.advect_diffuse (pygcm/ecology/phyto.py:496-547), called by the driver once per step right after the SST
// write-back (scripts/run_simulation.py:2254-2258), per species s on [S, n_lat, n_lon]:
//     C_adv = semi-Lagrangian gather of C along (uo, vo) dt, cos floor max(cos, 0.5)       phyto.py:467-494,512
//     C_new = (1 - alpha) C + alpha C_adv                                                  phyto.py:516-517
//     if K_h > 0: C_new = nan_to_num(C_new); C_new += dt K_h lap(C_new)                    phyto.py:520-522
//     C_new = clip(C_new, 0, inf); C_new[land] = 0                                         phyto.py:525-526
// then the ocean cells of the two pole rows get the mean of their row                      phyto.py:531-545
// The same device functions as the SST path of the ocean step (qd_departure / qd_gather / qd_lap_point, qd_device.h); the S
// species live in one stack and ride on blockIdx.z, so a step is three launches whatever S is, on the currents the ocean step
// left in HBM -- nothing crosses PCIe.  The ecology that feeds on the tracers (daily growth, optics) stays on the host.
#include "qd_internal.h"
#include "qd_device.h"
#include <vector>

__global__ void __launch_bounds__(QD_BLOCK)
k_phyto_advect(QdGeom G, const double* __restrict__ cos05, double dt, double a, double dlat, double dlon,
               const double* __restrict__ uo, const double* __restrict__ vo, const double* __restrict__ Cst,
               double* __restrict__ Tst, size_t stride, double alpha, int scrub) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
    const double* __restrict__ C = Cst + (size_t)tl.fld * stride;
    const QdBilin b = qd_departure(G, i, j, uo[o], vo[o], dt, a, cos05[i], dlat, dlon);
    const double adv = qd_gather(C, G, b);
    const double v = (1.0 - alpha) * C[o] + alpha * adv;
    Tst[(size_t)tl.fld * stride + o] = scrub ? qd_nn(v) : v;
}

__global__ void __launch_bounds__(QD_BLOCK)
k_phyto_diffuse(QdGeom G, const double* __restrict__ cos05, double dlat, double dlon, double a, double dtK, int has_diff,
                const double* __restrict__ Tst, double* __restrict__ Cst, size_t stride, const uint8_t* __restrict__ land) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
    const double* __restrict__ T = Tst + (size_t)tl.fld * stride;
    double v = T[o];
    if (has_diff) v = v + dtK * qd_lap_point<true>(T, G, cos05, i, j, dlat, dlon, a);
    v = (v >= 0.0 || v != v) ? v : 0.0;                       // np.clip(., 0, inf): NaN stays NaN
    Cst[(size_t)tl.fld * stride + o] = land[o] != 0 ? 0.0 : v;
}

// one workgroup per (pole row, species): mean over the ocean cells of the row, fixed-order tree sum
__global__ void __launch_bounds__(QD_BLOCK)
k_phyto_polar(QdGeom G, const uint8_t* __restrict__ land, double* __restrict__ Cst, size_t stride) {
    __shared__ double sm[2][QD_BLOCK / 64];
    const int i = blockIdx.x == 1 ? G.nlat - 1 : 0;
    if (i < G.row0 || i >= G.row0 + G.nrows) return;          // this band does not own the pole
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    double* __restrict__ C = Cst + (size_t)blockIdx.y * stride;
    double cnt = 0.0, s = 0.0;
    for (int j = threadIdx.x; j < G.nlon; j += QD_BLOCK)
        if (land[b + j] == 0) { cnt += 1.0; s += C[b + j]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_down(cnt, o, 64); s += __shfl_down(s, o, 64); }
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = cnt; sm[1][threadIdx.x >> 6] = s; }
    __syncthreads();
    cnt = sm[0][0]; s = sm[1][0];
    for (int k = 1; k < QD_BLOCK / 64; ++k) { cnt += sm[0][k]; s += sm[1][k]; }
    if (cnt <= 0.0) return;
    const double m = s / cnt;
    for (int j = threadIdx.x; j < G.nlon; j += QD_BLOCK)
        if (land[b + j] == 0) C[b + j] = m;
}

static void qd_phyto_free(qd_ctx* c) {
    QdPhyto& P = c->phyto;
    for (int k = 0; k < 2; ++k) { if (P.stack[k]) hipFree(P.stack[k]); P.stack[k] = nullptr; }
    for (int s = 0; s < QD_MAX_SPECIES; ++s) P.cur[s] = P.tmp[s] = nullptr;
    P.S = 0;
}
void qd_phyto_release(qd_ctx* c) { qd_phyto_free(c); }

static void qd_phyto_slots(qd_ctx* c) {
    QdPhyto& P = c->phyto;
    for (int s = 0; s < P.S; ++s) { P.cur[s] = P.stack[0] + (size_t)s * P.stride; P.tmp[s] = P.stack[1] + (size_t)s * P.stride; }
}

extern "C" int qd_phyto_configure(qd_handle c, int n_species, double K_h, double adv_alpha) {
    if (!c) return -1;
    if (n_species < 0 || n_species > QD_MAX_SPECIES) return qd_fail(c, "qd_phyto_configure: species count out of range (0..64)");
    hipSetDevice(c->desc.device);
    QdPhyto& P = c->phyto;
    P.K_h = K_h; P.alpha = adv_alpha;
    if (n_species == P.S) return 0;
    QD_HIP(c, hipStreamSynchronize(c->stream));
    qd_phyto_free(c);
    if (n_species == 0) return 0;
    P.stride = c->geo.cells() + (size_t)QD_PAD_ROWS * c->geo.nlon;
    const size_t bytes = (size_t)n_species * P.stride * sizeof(double);
    for (int k = 0; k < 2; ++k) {
        hipError_t e = hipMalloc(&P.stack[k], bytes);
        if (e != hipSuccess) { qd_phyto_free(c); return qd_fail(c, "qd_phyto_configure: hipMalloc", e); }
        hipMemsetAsync(P.stack[k], 0, bytes, c->stream);
    }
    P.S = n_species;
    qd_phyto_slots(c);
    for (int s = 0; s < P.S; ++s) { qd_mark(c, {P.cur[s]}, c->geo.halo); qd_mark(c, {P.tmp[s]}, 0); }
    QD_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int qd_phyto_upload(qd_handle c, int s, const double* host) {
    if (!c || !host) return -1;
    if (s < 0 || s >= c->phyto.S) return qd_fail(c, "qd_phyto_upload: no such species (qd_phyto_configure first)");
    hipSetDevice(c->desc.device);
    if (qd_band_copy_in(c, c->phyto.cur[s], host, sizeof(double))) return qd_fail(c, "qd_phyto_upload: copy failed");
    qd_mark(c, {c->phyto.cur[s]}, c->geo.halo);
    QD_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int qd_phyto_download(qd_handle c, int s, double* host) {
    if (!c || !host) return -1;
    if (s < 0 || s >= c->phyto.S) return qd_fail(c, "qd_phyto_download: no such species");
    hipSetDevice(c->desc.device);
    const QdGeom& G = c->geo;
    const size_t rowb = (size_t)G.nlon * sizeof(double);
    QD_HIP(c, hipMemcpyAsync((char*)host + (size_t)G.row0 * rowb, (const char*)c->phyto.cur[s] + (size_t)G.halo * rowb,
                             rowb * G.nrows, hipMemcpyDeviceToHost, c->stream));
    QD_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int qd_phyto_step_impl(qd_ctx* c, double dt) {
    QdPhyto& P = c->phyto;
    if (P.S == 0 || dt <= 0.0) return 0;                      // phyto.py:507-508
    QdScope sc(c, "phyto");
    double** F = c->f;
    const qd_params& p = c->p;
    const dim3 blk(QD_BLOCK);
    const int has_diff = P.K_h > 0.0 ? 1 : 0;
    const int Ro = qd_adv_reach(c, dt, 4.0);                  // currents are capped at QD_OCEAN_MAX_U (3 m/s)
    std::vector<QdUse> in;
    for (int s = 0; s < P.S; ++s) in.push_back(QD_IN(P.cur[s], Ro));
    in.push_back(QD_IN(F[QD_F_UO], 0)); in.push_back(QD_IN(F[QD_F_VO], 0));
    const int m1 = qd_plan(c, in.data(), (int)in.size());
    if (m1 < 0) return -1;
    QD_ROWS(c, m1, G, hipLaunchKernelGGL(k_phyto_advect, qd_grid2d(G, P.S), blk, 0, c->stream, G, c->tabs.cos05, dt, p.a, c->dlat,
                                         c->dlon, F[QD_F_UO], F[QD_F_VO], P.stack[0], P.stack[1], P.stride, P.alpha, has_diff));
    for (int s = 0; s < P.S; ++s) qd_mark(c, {P.tmp[s]}, m1);
    in.clear();
    // qd_lap_point<true> reads rows i-2 .. i+2 (qd_dphi at rows i+-1; qd_device.h:23-55), like the SST path (qd_ocean.hip, QD_IN(T1s, 2))
    for (int s = 0; s < P.S; ++s) in.push_back(QD_IN(P.tmp[s], has_diff ? 2 : 0));
    const int m2 = qd_plan(c, in.data(), (int)in.size());
    if (m2 < 0) return -1;
    QD_ROWS(c, m2, G, hipLaunchKernelGGL(k_phyto_diffuse, qd_grid2d(G, P.S), blk, 0, c->stream, G, c->tabs.cos05, c->dlat, c->dlon,
                                         p.a, dt * P.K_h, has_diff, P.stack[1], P.stack[0], P.stride, c->land));
    // the two pole rows are owned by the first / last band; their copies in the other polar band's wrap halo go stale
    const QdGeom Gown = qd_segments(c, 0).g[0];
    hipLaunchKernelGGL(k_phyto_polar, dim3(2, P.S), blk, 0, c->stream, Gown, c->land, P.stack[0], P.stride);
    for (int s = 0; s < P.S; ++s) qd_mark(c, {P.cur[s]}, 0);
    return 0;
}

extern "C" int qd_phyto_advect_diffuse(qd_handle c, double dt) {
    if (!c) return -1;
    if (c->phyto.S == 0) return qd_fail(c, "qd_phyto_advect_diffuse: no tracers (qd_phyto_configure first)");
    hipSetDevice(c->desc.device);
    int rc = qd_phyto_step_impl(c, dt);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qd_fail(c, "qd_phyto_advect_diffuse: launch", e);
    return 0;
}
