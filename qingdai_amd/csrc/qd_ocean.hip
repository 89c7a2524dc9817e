// qd_ocean.hip -- WindDrivenSlabOcean.step (pygcm/ocean.py:265-533) and the driver's ocean
// coupling (scripts/run_simulation.py:2197-2253, benchmark_jax.py:134-158) on gfx950.
//
// Per outer step:   k_qnet (SW/LW/SH/LH -> Q_net, ice mask)      run_simulation.py:2199-2239
//                   k_stress_max (wind stress + CFL maxima)       ocean.py:283-303
//                   host reads two maxima -> n_sub                ocean.py:301-303
// Per sub-step:     k_ocean_momentum                              ocean.py:306-336
//                   k_laplacian / k_hyper_apply on uo,vo,eta      ocean.py:341-356
//                   k_continuity (+ weighted eta sum)             ocean.py:365-377
//                   k_sst_advect (eta mean removal folded in)     ocean.py:375,380-382
//                   k_sst_diffuse_heat                            ocean.py:385-406,440
//                   k_outlier                                     ocean.py:409-444
// After:            k_polar_fill (2 workgroups), clamp + SST write-back   ocean.py:519-533
#include <cstdlib>
#include <atomic>
#include <chrono>
#include "qd_internal.h"
#include "qd_device.h"
#include "qd_wave.h"

#include "qd_fluxes.h"
#include "qd_fused.h"
#include "qd_ocntail.h"
#include "qd_band.h"

QdColP qd_make_colp(const qd_ctx* c, double dt);   // qd_atmos.hip

// ------------------------------------------------------------------ Q_net for the coupling
__global__ void __launch_bounds__(QD_BLOCK)
k_qnet(QdGeom G, QdColP P, const double* __restrict__ isr, const double* __restrict__ albedo,
       const double* __restrict__ cloud, const double* __restrict__ Ts, const double* __restrict__ h,
       const double* __restrict__ u, const double* __restrict__ v, const uint8_t* __restrict__ land,
       const double* __restrict__ hice, const double* __restrict__ LH, double* __restrict__ qnet,
       uint8_t* __restrict__ icemask) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    const double hh = h[o];
    const double T_a = 288.0 + P.ga * hh;
    const double hi = hice[o];
    const QdFlux F = qd_surface_fluxes(P, isr[o], albedo[o], cloud[o], Ts[o], T_a, u[o], v[o], land[o] == 1, hi);
    qnet[o] = F.SW_sfc - F.LW_sfc - F.SH - LH[o];
    icemask[o] = (hi > 0.0) ? 1 : 0;
}

__device__ __forceinline__ double qd_wave_sum_d(double x);

// ------------------------------------------------------------------ energy budget means (energy.py:494-538)
// cos-weighted row sums of the ten budget terms, from the same flux function as k_qnet:
//   0 TOA_net = I - R - OLR   1 SFC_net = SW_sfc - LW_sfc - SH - LH   2 ATM_net = TOA - SFC
//   3 I  4 R  5 OLR  6 SW_sfc  7 LW_sfc  8 SH  9 LH
#define QD_NDIAG 10
__global__ void __launch_bounds__(QD_BLOCK)
k_energy_diag(QdGeom G, QdTabs T, QdColP P, const double* __restrict__ isr, const double* __restrict__ albedo,
              const double* __restrict__ cloud, const double* __restrict__ Ts, const double* __restrict__ h,
              const double* __restrict__ u, const double* __restrict__ v, const uint8_t* __restrict__ land,
              const double* __restrict__ hice, const double* __restrict__ LH, double* __restrict__ partial) {
    __shared__ double sm[QD_NDIAG][QD_BLOCK / 64];
    const int i = G.row0 + blockIdx.y;
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    const double w = T.warea[i];
    double acc[QD_NDIAG];
#pragma unroll
    for (int k = 0; k < QD_NDIAG; ++k) acc[k] = 0.0;
    for (int j = threadIdx.x; j < G.nlon; j += QD_BLOCK) {
        const size_t o = b + j;
        const double T_a = 288.0 + P.ga * h[o];
        const double I = isr[o], lh = LH[o];
        const QdFlux F = qd_surface_fluxes(P, I, albedo[o], cloud[o], Ts[o], T_a, u[o], v[o], land[o] == 1, hice[o]);
        const double toa = I - F.R - F.OLR, sfc = F.SW_sfc - F.LW_sfc - F.SH - lh;
        const double t[QD_NDIAG] = {toa, sfc, toa - sfc, I, F.R, F.OLR, F.SW_sfc, F.LW_sfc, F.SH, lh};
#pragma unroll
        for (int k = 0; k < QD_NDIAG; ++k) acc[k] += t[k] * w;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < QD_NDIAG; ++k) { const double s = qd_wave_sum_d(acc[k]); if (lane == 0) sm[k][wv] = s; }
    __syncthreads();
    if (threadIdx.x < QD_NDIAG) {
        double r = sm[threadIdx.x][0];
        for (int k = 1; k < QD_BLOCK / 64; ++k) r += sm[threadIdx.x][k];
        partial[(size_t)threadIdx.x * gridDim.y + blockIdx.y] = r;
    }
}
__global__ void __launch_bounds__(QD_BLOCK)
k_energy_diag_finish(const double* __restrict__ partial, int nrows, double* __restrict__ out) {
    __shared__ double sm[QD_BLOCK / 64];
    for (int q = 0; q < QD_NDIAG; ++q) {
        double a = 0.0;
        for (int k = threadIdx.x; k < nrows; k += QD_BLOCK) a += partial[(size_t)q * nrows + k];
        a = qd_wave_sum_d(a);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) { double r = sm[0]; for (int k = 1; k < QD_BLOCK / 64; ++k) r += sm[k]; out[q] = r; }
    }
}

// the driver's coupling block works on its own EnergyParams copy (the one autotune nudges)
static QdColP qd_make_colp_driver(const qd_ctx* c, double dt) {
    QdColP P = qd_make_colp(c, dt);
    if (c->p.qnet_lw_eps0 == c->p.qnet_lw_eps0) { P.lw_eps0 = c->p.qnet_lw_eps0; P.eps_clear = std::min(std::max(c->p.qnet_lw_eps0, 0.0), 1.0); }
    if (c->p.qnet_lw_kc == c->p.qnet_lw_kc) P.lw_kc = c->p.qnet_lw_kc;
    return P;
}

int qd_energy_diag_impl(qd_ctx* c, double* host_out) {
    double** F = c->f;
    QdColP P = qd_make_colp_driver(c, 0.0);
    double*& cl = c->cloud_eff_valid ? F[QD_F_CLOUD_EFF] : F[QD_F_CLOUD];
    const int m = qd_plan(c, {QD_IN(F[QD_F_ISR], 0), QD_IN(F[QD_F_ALBEDO], 0), QD_IN(cl, 0), QD_IN(F[QD_F_TS], 0), QD_IN(F[QD_F_H], 0),
                              QD_IN(F[QD_F_U], 0), QD_IN(F[QD_F_V], 0), QD_IN(F[QD_F_HICE], 0), QD_IN(F[QD_F_LH], 0)});
    if (m < 0) return -1;
    const QdGeom G = qd_segments(c, 0).g[0];                 // owned rows only
    if ((size_t)QD_NDIAG * G.nrows > (size_t)c->red_blocks) return qd_fail(c, "qd_energy_diagnostics: partial buffer too small");
    hipLaunchKernelGGL(k_energy_diag, dim3(1, G.nrows), dim3(QD_BLOCK), 0, c->stream, G, c->tabs, P, F[QD_F_ISR], F[QD_F_ALBEDO], cl,
                       F[QD_F_TS], F[QD_F_H], F[QD_F_U], F[QD_F_V], c->land, F[QD_F_HICE], F[QD_F_LH], c->red_partial);
    hipLaunchKernelGGL(k_energy_diag_finish, dim3(1), dim3(QD_BLOCK), 0, c->stream, c->red_partial, G.nrows, c->dscal + QD_S_DIAG0);
    if (qd_allreduce_f64(c, c->dscal + QD_S_DIAG0, QD_NDIAG, 0)) return -1;
    QD_HIP(c, hipMemcpyAsync(c->hpin, c->dscal + QD_S_DIAG0, QD_NDIAG * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    QD_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < QD_NDIAG; ++k) host_out[k] = c->hpin[k] / (c->wsum_all + 1e-15);
    return 0;
}

// ------------------------------------------------------------------ wind stress + CFL maxima
__device__ __forceinline__ double qd_wave_max_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double y = __shfl_down(x, o, 64); x = (y > x) ? y : x; }
    return x;
}

__global__ void __launch_bounds__(QD_BLOCK)
k_stress_max(QdGeom G, const double* __restrict__ ua, const double* __restrict__ va,
             const double* __restrict__ uo, const double* __restrict__ vo, double vcap, double rhoCD, double tau_scale,
             double* __restrict__ taux, double* __restrict__ tauy, double* __restrict__ partial, double* seq_out, double seq) {
    __shared__ double sm[2][QD_BLOCK / 64];
    const int i = G.row0 + blockIdx.y;
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    double mVa = 0.0, mUo = 0.0;
    for (int j = threadIdx.x; j < G.nlon; j += QD_BLOCK) {
        const size_t o = b + j;
        const double u_o = uo[o], v_o = vo[o];
        const double u_rel = ua[o] - u_o, v_rel = va[o] - v_o;
        const double Va = sqrt(u_rel * u_rel + v_rel * v_rel);
        const double Va_eff = qd_min(Va, vcap);
        taux[o] = tau_scale * (rhoCD * Va_eff * u_rel);
        tauy[o] = tau_scale * (rhoCD * Va_eff * v_rel);
        const double so = sqrt(u_o * u_o + v_o * v_o);
        mVa = Va > mVa ? Va : mVa;        // NaN never wins, like np.max on nan_to_num'd data
        mUo = so > mUo ? so : mUo;
    }
    mVa = qd_wave_max_d(mVa); mUo = qd_wave_max_d(mUo);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sm[0][w] = mVa; sm[1][w] = mUo; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < QD_BLOCK / 64; ++k) { mVa = sm[0][k] > mVa ? sm[0][k] : mVa; mUo = sm[1][k] > mUo ? sm[1][k] : mUo; }
        partial[blockIdx.y] = mVa;
        partial[gridDim.y + blockIdx.y] = mUo;
        // whole-globe handles: `partial` is pinned host memory and the host is waiting for it -- every row stamps its own arrival
        // (system-scope release behind its two maxima), so no k_host_flag launch has to follow (4.7 us per step)
        // (a RELEASE here writes the whole L2 back once per row: 21.6 us for the launch instead of 11.6; the two maxima and the stamp
        //  are uncached stores of ONE thread to pinned host memory -- they leave in order once the maxima have been acknowledged)
        if (seq_out) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(seq_out + blockIdx.y, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------ k_final + k_stress_max + k_qnet in one launch (whole globe, qd_step_n)
// time_step's last kernel (dynamics.py:642-667: cloud gather with the new winds, decay, damp, scrub) is pointwise apart from the
// gather from the OLD cloud slab, and so are the two kernels the coupling block starts with (run_simulation.py:2199-2239, ocean.py:
// 283-303): they read the u, v, T_s, h it has just written.  One 2-D launch does all three from registers: 183 MB instead of 233 MB
// moved.  The CFL maxima: every workgroup stores its two maxima (>= 0; NaN never wins: the running maximum is only replaced by
// something GREATER) into its own two slots of device memory and ends; one small workgroup behind it (k_max2_publish) reduces
// the 2 x 4 326 values and hands the two maxima to the host behind a stamp.  What was tried for the hand-over instead
// (profiles/README.md): one workgroup per row like k_stress_max (round 3: 51 us); an integer atomicMax per row + a ticket + the
// row's stamp from its last workgroup (42 us: 4 326 workgroups that each end in two dependent L2 round trips of one thread);
// two uncached stores per workgroup straight into self-validating host slots (64 us: 8 652 small writes over PCIe).
struct QdFqsArgs {
    const double* cosl; double dt, a, dlat, dlon, decay, dfac;
    double *u, *v, *h, *Ts, *q; const double* cloud_in; double* cloud_out;
    const double *isr, *albedo, *cloud_eff, *hice, *LH; const uint8_t* land; double* qnet; uint8_t* icemask;
    const double *uo, *vo; double vcap, rhoCD, tau_scale; double *taux, *tauy;
    double* dev_wg; int n_wg;
};
__global__ void __launch_bounds__(QD_BLOCK)
k_final_qnet_stress(QdGeom G, QdColP P, QdFqsArgs A) {
    __shared__ double sm[2][QD_BLOCK / 64];
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    const int i = G.row0 + tl.row;
    double Va = 0.0, so = 0.0;
    if (j < G.nlon) {
        const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
        // ---- k_final
        const double uu = A.u[o], vv = A.v[o];
        const QdBilin b = qd_departure(G, i, j, uu, vv, A.dt, A.a, A.cosl[i], A.dlat, A.dlon);
        double c = qd_gather(A.cloud_in, G, b);
        c = c * A.decay;
        const double cn = qd_nn(c * A.dfac), un = qd_nn(uu * A.dfac), vn = qd_nn(vv * A.dfac);
        const double t0 = A.Ts[o];
        const double hn = qd_nn(A.h[o] * A.dfac), tn = qd_nn(t0);
        A.cloud_out[o] = cn; A.u[o] = un; A.v[o] = vn; A.h[o] = hn; A.q[o] = qd_nn(A.q[o] * A.dfac);
        if (__double_as_longlong(tn) != __double_as_longlong(t0)) A.Ts[o] = tn;                  // nan_to_num of a finite T_s: the bits that are there
        // ---- k_stress_max
        const double u_o = A.uo[o], v_o = A.vo[o];
        const double u_rel = un - u_o, v_rel = vn - v_o;
        Va = sqrt(u_rel * u_rel + v_rel * v_rel);
        const double Va_eff = qd_min(Va, A.vcap);
        A.taux[o] = A.tau_scale * (A.rhoCD * Va_eff * u_rel);
        A.tauy[o] = A.tau_scale * (A.rhoCD * Va_eff * v_rel);
        so = sqrt(u_o * u_o + v_o * v_o);
        // ---- k_qnet (cloud optical field: cloud_eff_last when time_step produced one, else the cloud cover just written)
        const double T_a = 288.0 + P.ga * hn;
        const double hi = A.hice[o];
        const QdFlux F = qd_surface_fluxes(P, A.isr[o], A.albedo[o], A.cloud_eff ? A.cloud_eff[o] : cn, tn, T_a, un, vn, A.land[o] == 1, hi);
        A.qnet[o] = F.SW_sfc - F.LW_sfc - F.SH - A.LH[o];
        A.icemask[o] = (hi > 0.0) ? 1 : 0;
    }
    double mVa = Va > 0.0 ? Va : 0.0, mUo = so > 0.0 ? so : 0.0;           // NaN never wins (k_stress_max: replaced only by something greater than 0)
    mVa = qd_wave_max_d(mVa); mUo = qd_wave_max_d(mUo);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sm[0][w] = mVa; sm[1][w] = mUo; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < QD_BLOCK / 64; ++k) { mVa = sm[0][k] > mVa ? sm[0][k] : mVa; mUo = sm[1][k] > mUo ? sm[1][k] : mUo; }
        const unsigned wg = (unsigned)tl.row * gridDim.x + (unsigned)tl.seg;
        A.dev_wg[wg] = mVa; A.dev_wg[A.n_wg + wg] = mUo;                 // device memory, nothing to wait for: k_max2_publish follows
    }
}

// the two maxima over the per-workgroup maxima of k_final_qnet_stress -> two self-validating slots in pinned host memory (the host
// keeps them at -1 and polls for >= 0: qd_wait_host_nonneg; no stamp, so nothing to wait for here).  One workgroup.
__global__ void __launch_bounds__(1024)
k_max2_publish(const double* __restrict__ wgmax, int n, double* host2, unsigned int* fixstat) {
    __shared__ double sm[2][16];
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < n; k += 1024) { const double x = wgmax[k], y = wgmax[n + k]; a = x > a ? x : a; b = y > b ? y : b; }
    a = qd_wave_max_d(a); b = qd_wave_max_d(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sm[0][w] = a; sm[1][w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) { a = sm[0][k] > a ? sm[0][k] : a; b = sm[1][k] > b ? sm[1][k] : b; }
        // how long the ocean tail's fix lists were since the last step (entries per launch; no launch with a list: "no news", -2 -- the
        // host keeps the slot at -1): the host takes the storing form while they are long (qd_ocean_step_impl).  Stored FIRST: the host
        // reads it once the two maxima behind it have arrived (uncached stores of one thread leave in order)
        double avg = -2.0;
        if (fixstat) {
            const unsigned e = fixstat[4], l = fixstat[5];
            if (l) avg = (double)e / (double)l;
            fixstat[4] = 0u; fixstat[5] = 0u;
        }
        __hip_atomic_store((unsigned long long*)host2 - 1, (unsigned long long)__double_as_longlong(avg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store((unsigned long long*)host2, (unsigned long long)__double_as_longlong(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((unsigned long long*)host2 + 1, (unsigned long long)__double_as_longlong(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void __launch_bounds__(QD_BLOCK)
k_max2_finish(const double* __restrict__ partial, int n, double* __restrict__ out, int nzero) {
    __shared__ double sm[2][QD_BLOCK / 64];
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < n; k += QD_BLOCK) { a = partial[k] > a ? partial[k] : a; b = partial[n + k] > b ? partial[n + k] : b; }
    a = qd_wave_max_d(a); b = qd_wave_max_d(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sm[0][w] = a; sm[1][w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < QD_BLOCK / 64; ++k) { a = sm[0][k] > a ? sm[0][k] : a; b = sm[1][k] > b ? sm[1][k] : b; }
        out[0] = a; out[1] = b;
        for (int k = 0; k < nzero; ++k) out[2 + k] = 0.0;    // the slots of segments this band does not have (no memset launch)
    }
}

// ------------------------------------------------------------------ momentum: ocean.py:306-336
struct QdOcnP { double a, g, dlat, dlon, sub_dt, rhoH, r_bot; };

__global__ void __launch_bounds__(QD_BLOCK)
k_ocean_momentum(QdGeom G, QdTabs T, QdOcnP P, const double* __restrict__ eta, const double* __restrict__ taux,
                 const double* __restrict__ tauy, const uint8_t* __restrict__ land,
                 double* __restrict__ uo, double* __restrict__ vo) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    const size_t o = b + j;
    const int jp = qd_wrapc(j + 1, G.nlon), jm = qd_wrapc(j - 1, G.nlon);
    const double deta_dlam = (eta[b + jp] - eta[b + jm]) / (2.0 * P.dlon);
    const double deta_dphi = (eta[(size_t)qd_lrow(G, i + 1) * G.nlon + j] - eta[(size_t)qd_lrow(G, i - 1) * G.nlon + j]) / (2.0 * P.dlat);
    const double gx = deta_dlam / (P.a * T.cos05[i]);
    const double gy = deta_dphi / P.a;
    const double f = T.fcor[i];
    const double u0 = uo[o], v0 = vo[o];
    const double du = (f * v0 - P.g * gx + taux[o] / P.rhoH - P.r_bot * u0);
    const double dv = (-f * u0 - P.g * gy + tauy[o] / P.rhoH - P.r_bot * v0);
    double un = u0 + P.sub_dt * du;
    double vn = v0 + P.sub_dt * dv;
    if (land[o] == 1) { un = 0.0; vn = 0.0; }
    const double rx = T.r_extra[i];
    un = un - P.sub_dt * rx * un;
    vn = vn - P.sub_dt * rx * vn;
    uo[o] = un; vo[o] = vn;
}

// ------------------------------------------------------------------ continuity: ocean.py:365-374
__device__ __forceinline__ double qd_wave_sum_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}

__global__ void __launch_bounds__(QD_BLOCK)
k_continuity(QdGeom G, QdTabs T, double a, double dlat, double dlon, double msdtH, const double* __restrict__ uo,
             const double* __restrict__ vo, const uint8_t* __restrict__ land, double* __restrict__ eta,
             double* __restrict__ partial) {
    __shared__ double sm[QD_BLOCK / 64];
    const int i = G.row0 + blockIdx.y;
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    const double w = T.warea[i];
    double acc = 0.0;
    for (int j = threadIdx.x; j < G.nlon; j += QD_BLOCK) {
        const size_t o = b + j;
        const double div = qd_divvort_point(G, T, uo, vo, i, j, a, dlat, dlon, 0);
        double e = eta[o] + msdtH * div;
        const bool island = land[o] == 1;
        if (island) e = 0.0;
        eta[o] = e;
        acc += e * (island ? 0.0 : w);          // eta * (w * ocean_mask)
    }
    acc = qd_wave_sum_d(acc);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sm[0];
        for (int k = 1; k < QD_BLOCK / 64; ++k) r += sm[k];
        partial[blockIdx.y] = r;
    }
}

__global__ void __launch_bounds__(QD_BLOCK)
k_eta_mean(const double* __restrict__ partial, int n, double wsum, double* __restrict__ out, double seq = 0.0) {
    __shared__ double sm[QD_BLOCK / 64];
    double acc = 0.0;
    for (int k = threadIdx.x; k < n; k += QD_BLOCK) acc += partial[k];
    acc = qd_wave_sum_d(acc);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sm[0];
        for (int k = 1; k < QD_BLOCK / 64; ++k) r += sm[k];
        *out = (wsum < 0.0) ? r : r / (wsum + 1e-15);      // wsum < 0: raw sum (bands all-reduce it first)
        // seq != 0: `out` is pinned host memory and the host polls out[1] for this sequence number (system-scope release: the
        // sum is visible before the flag)
        if (seq != 0.0) __hip_atomic_store(&out[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------ continuity + SST advection, merged
// ocean.py:365-374 (eta += -dt H div, land zero, area-weighted sum over the OWNED rows) and ocean.py:380-382
// (SST semi-Lagrangian blend) read the same uo, vo: one pass.  The eta mean is removed later, in
// k_sst_outlier_fused, right before the clip (same arithmetic: clip(nan_to_num(eta - mean))).
// One workgroup per row; when `ticket` is non-null the last workgroup to finish turns the row partials into
// eta_mean = sum / (wsum + 1e-15) (whole-globe handles; bands all-reduce the raw sum instead).
__global__ void __launch_bounds__(QD_BLOCK)
k_cont_sstadv(QdGeom G, QdTabs T, double a, double dlat, double dlon, double msdtH, double sub_dt, double alpha,
              const double* __restrict__ uo, const double* __restrict__ vo, const uint8_t* __restrict__ land,
              double* __restrict__ eta, const double* __restrict__ Ts, double* __restrict__ Ts_out,
              int own0, int own1, double* __restrict__ partial, unsigned long long* ticket, double wsum,
              double* __restrict__ eta_mean) {
    __shared__ double sm[QD_BLOCK / 64];
    __shared__ int s_last;
    const int i = G.row0 + blockIdx.y;
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    const double w = (i >= own0 && i < own1) ? T.warea[i] : 0.0;
    const double cosl = T.cos05[i];
    const int pidx = blockIdx.y * gridDim.x + blockIdx.x, pcount = gridDim.x * gridDim.y;
    double acc = 0.0;
    for (int j = blockIdx.x * QD_BLOCK + threadIdx.x; j < G.nlon; j += gridDim.x * QD_BLOCK) {
        const size_t o = b + j;
        const double div = qd_divvort_point(G, T, uo, vo, i, j, a, dlat, dlon, 0);
        double e = eta[o] + msdtH * div;
        const bool island = land[o] == 1;
        if (island) e = 0.0;
        eta[o] = e;
        acc += e * (island ? 0.0 : w);
        const QdBilin bl = qd_departure(G, i, j, uo[o], vo[o], sub_dt, a, cosl, dlat, dlon);
        Ts_out[o] = (1.0 - alpha) * Ts[o] + alpha * qd_gather(Ts, G, bl);
    }
    acc = qd_wave_sum_d(acc);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sm[0];
        for (int k = 1; k < QD_BLOCK / 64; ++k) r += sm[k];
        if (!ticket) partial[pidx] = r;
        else {
            // write-through (agent-scope) store, drained, then the ticket: the last workgroup reads the
            // partials back with agent-scope loads -- no cache flush on thousands of workgroups
            __hip_atomic_store(&partial[pidx], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t = atomicAdd(ticket, 1ull);
            s_last = (t == (unsigned long long)pcount - 1ull) ? 1 : 0;
        }
    }
    if (!ticket) return;
    __syncthreads();
    if (!s_last) return;
    double a2 = 0.0;
    for (int k = threadIdx.x; k < pcount; k += QD_BLOCK)
        a2 += __hip_atomic_load(&partial[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a2 = qd_wave_sum_d(a2);
    __syncthreads();
    if (lane == 0) sm[wv] = a2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sm[0];
        for (int k = 1; k < QD_BLOCK / 64; ++k) r += sm[k];
        *eta_mean = r / (wsum + 1e-15);
        __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------ SST advection: ocean.py:375,380-382
__global__ void __launch_bounds__(QD_BLOCK)
k_sst_advect(QdGeom G, const double* __restrict__ cos05, double sub_dt, double a, double dlat, double dlon,
             const double* __restrict__ uo, const double* __restrict__ vo, const double* __restrict__ Ts,
             double* __restrict__ Ts_out, double alpha, double* __restrict__ eta, const double* __restrict__ eta_mean,
             int has_ocean) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
    if (has_ocean) eta[o] = eta[o] - *eta_mean;
    const QdBilin b = qd_departure(G, i, j, uo[o], vo[o], sub_dt, a, cos05[i], dlat, dlon);
    const double adv = qd_gather(Ts, G, b);
    Ts_out[o] = (1.0 - alpha) * Ts[o] + alpha * adv;
}

// ------------------------------------------------------------------ diffusion + heating: ocean.py:385-406,440
struct QdHeatP { double sub_dt, K_h, rcH, ice_qfac; int use_q, has_ice; };

__global__ void __launch_bounds__(QD_BLOCK)
k_sst_diffuse_heat(QdGeom G, const double* __restrict__ cos05, double dlat, double dlon, double a, QdHeatP P,
                   const double* __restrict__ Ts1, double* __restrict__ Ts_out, const double* __restrict__ qnet,
                   const uint8_t* __restrict__ land, const uint8_t* __restrict__ ice) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
    double T = Ts1[o];
    if (P.K_h > 0.0) T = qd_nn(T) + P.sub_dt * P.K_h * qd_lap_point<true>(Ts1, G, cos05, i, j, dlat, dlon, a);   // in-place scrub, ocean.py:112,386
    if (P.use_q) {
        const double heat = qnet[o] / P.rcH;
        const bool ocean = land[o] == 0;
        if (P.has_ice) {
            const bool ic = ice[o] != 0;
            if (ocean && !ic) T = T + P.sub_dt * heat;
            if (P.ice_qfac > 0.0 && ocean && ic) T = T + P.sub_dt * P.ice_qfac * heat;
        } else if (ocean) T = T + P.sub_dt * heat;
    }
    Ts_out[o] = qd_nn(T);
}

// ------------------------------------------------------------------ outliers + caps: ocean.py:409-444
__global__ void __launch_bounds__(QD_BLOCK)
k_outlier(QdGeom G, const double* __restrict__ uo, const double* __restrict__ vo, double* __restrict__ uo_out,
          double* __restrict__ vo_out, double* __restrict__ eta, double cap, double eta_cap, int mean4) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    const size_t o = b + j;
    double u = qd_nn(uo[o]), v = qd_nn(vo[o]);
    const double speed = sqrt(u * u + v * v);
    if (mean4) {
        if (speed > cap) {
            const size_t bn = (size_t)qd_lrow(G, i + 1) * G.nlon, bs = (size_t)qd_lrow(G, i - 1) * G.nlon;
            const int je = qd_wrapc(j + 1, G.nlon), jw = qd_wrapc(j - 1, G.nlon);
            u = 0.25 * (qd_nn(uo[bn + j]) + qd_nn(uo[bs + j]) + qd_nn(uo[b + je]) + qd_nn(uo[b + jw]));
            v = 0.25 * (qd_nn(vo[bn + j]) + qd_nn(vo[bs + j]) + qd_nn(vo[b + je]) + qd_nn(vo[b + jw]));
        }
        const double sp2 = sqrt(u * u + v * v);
        const double sc2 = (sp2 > cap) ? cap / (sp2 + 1e-12) : 1.0;
        u = u * sc2; v = v * sc2;
    } else {
        const double sc = (speed > cap) ? cap / (speed + 1e-12) : 1.0;
        u = u * sc; v = v * sc;
    }
    uo_out[o] = u; vo_out[o] = v;
    eta[o] = qd_clip(qd_nn(eta[o]), -eta_cap, eta_cap);
}

// ------------------------------------------------------------------ fused: SST diffusion + heating + outliers + caps
// (ocean.py:385-444 in one pass; the fast Laplacian uses the reciprocal row tables)
__global__ void __launch_bounds__(QD_BLOCK)
k_sst_outlier_fused(QdGeom G, QdTabs T, double dlat, double dlon, double a, QdHeatP P,
                    const double* __restrict__ Ts1, double* __restrict__ Ts_out, const double* __restrict__ qnet,
                    const uint8_t* __restrict__ land, const uint8_t* __restrict__ ice,
                    const double* __restrict__ uo, const double* __restrict__ vo, double* __restrict__ uo_out,
                    double* __restrict__ vo_out, double* __restrict__ eta, double cap, double eta_cap, int mean4,
                    const double* __restrict__ eta_mean, const double* __restrict__ partial, int pcount, double wsum,
                    double* __restrict__ mean_out) {
    // deferred-mean mode (mean_out != nullptr): eta is not touched here -- the next momentum kernel applies
    // eta - mean, nan_to_num and the clip on load -- and workgroup 0 turns the continuity kernel's partial sums into
    // the mean on the side (same summation order as k_eta_mean)
    if (mean_out && blockIdx.x == 0 && blockIdx.y == 0) {
        __shared__ double smm[QD_BLOCK / 64];
        double acc = 0.0;
        for (int k = threadIdx.x; k < pcount; k += QD_BLOCK) acc += partial[k];
        acc = qd_wave_sum_d(acc);
        if ((threadIdx.x & 63) == 0) smm[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double r = smm[0];
            for (int k = 1; k < QD_BLOCK / 64; ++k) r += smm[k];
            *mean_out = r / (wsum + 1e-15);
        }
    }
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    const size_t o = b + j;
    double Tv = Ts1[o];
    // ocean.py:386 `self.Ts += dt*K_h*lap(self.Ts)`: the Laplacian scrubs self.Ts IN PLACE (nan_to_num(copy=False)) before the add
    if (P.K_h > 0.0) Tv = qd_nn(Tv) + P.sub_dt * P.K_h * qd_lap_point_fast<true>(Ts1, G, T, 1, i, j, dlat, dlon, a);
    if (P.use_q) {
        const double heat = qnet[o] / P.rcH;
        const bool ocean = land[o] == 0;
        if (P.has_ice) {
            const bool ic = ice[o] != 0;
            if (ocean && !ic) Tv = Tv + P.sub_dt * heat;
            if (P.ice_qfac > 0.0 && ocean && ic) Tv = Tv + P.sub_dt * P.ice_qfac * heat;
        } else if (ocean) Tv = Tv + P.sub_dt * heat;
    }
    Ts_out[o] = qd_nn(Tv);
    double u = qd_nn(uo[o]), v = qd_nn(vo[o]);
    const double speed = sqrt(u * u + v * v);
    if (mean4) {
        if (speed > cap) {
            const size_t bn = (size_t)qd_lrow(G, i + 1) * G.nlon, bs = (size_t)qd_lrow(G, i - 1) * G.nlon;
            const int je = qd_wrapc(j + 1, G.nlon), jw = qd_wrapc(j - 1, G.nlon);
            u = 0.25 * (qd_nn(uo[bn + j]) + qd_nn(uo[bs + j]) + qd_nn(uo[b + je]) + qd_nn(uo[b + jw]));
            v = 0.25 * (qd_nn(vo[bn + j]) + qd_nn(vo[bs + j]) + qd_nn(vo[b + je]) + qd_nn(vo[b + jw]));
        }
        const double sp2 = sqrt(u * u + v * v);
        const double sc2 = (sp2 > cap) ? cap / (sp2 + 1e-12) : 1.0;
        u = u * sc2; v = v * sc2;
    } else {
        const double sc = (speed > cap) ? cap / (speed + 1e-12) : 1.0;
        u = u * sc; v = v * sc;
    }
    uo_out[o] = u; vo_out[o] = v;
    // eta -= area-weighted ocean mean (ocean.py:375), then nan_to_num + clip (ocean.py:436-443)
    if (mean_out) return;
    // pcount < 0: *eta_mean is the RAW all-reduced sum of the bands (no launch of its own for one division)
    const double e = eta_mean ? eta[o] - (pcount < 0 ? *eta_mean / (wsum + 1e-15) : *eta_mean) : eta[o];
    eta[o] = qd_clip(qd_nn(e), -eta_cap, eta_cap);
}


// ------------------------------------------------------------------ polar ring fills: ocean.py:197-262
// one workgroup per pole row; fixed-order tree sums (deterministic)
__device__ __forceinline__ double qd_block_sum_ocn(double x, double* sm) {
    x = qd_wave_sum_d(x);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wv] = x;
    __syncthreads();
    double r = sm[0];
    for (int k = 1; k < QD_BLOCK / 64; ++k) r += sm[k];
    return r;
}

// the polar fill of one pole row (ocean.py:197-262, 519-533) by ONE workgroup: ocean means of SST and of the current vector
__device__ __forceinline__ void qd_polar_fill_row(const QdGeom& G, const QdTabs& T, const uint8_t* __restrict__ land, double* __restrict__ Ts,
                                                  double* __restrict__ uo, double* __restrict__ vo, bool north, double* sm) {
    const int i = north ? G.nlat - 1 : 0;
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    double cnt = 0, sT = 0, s0 = 0, s1 = 0;
    for (int j = threadIdx.x; j < G.nlon; j += QD_BLOCK) {
        if (land[b + j] != 0) continue;
        cnt += 1.0;
        sT += Ts[b + j];
        const double sl = T.sin_lon[j], cl = T.cos_lon[j];
        const double u = uo[b + j], v = vo[b + j];
        // e_east = (-sin, cos, 0); e_north = (-cos, -sin, 0) at +90, (cos, sin, 0) at -90; z component is 0
        const double nx = north ? -cl : cl, ny = north ? -sl : sl;
        s0 += (-sl) * u + nx * v;
        s1 += cl * u + ny * v;
    }
    cnt = qd_block_sum_ocn(cnt, sm);
    sT = qd_block_sum_ocn(sT, sm);
    s0 = qd_block_sum_ocn(s0, sm);
    s1 = qd_block_sum_ocn(s1, sm);
    if (cnt <= 0.0) return;
    const double mT = sT / cnt, m0 = s0 / cnt, m1 = s1 / cnt;
    for (int j = threadIdx.x; j < G.nlon; j += QD_BLOCK) {
        if (land[b + j] != 0) continue;
        const double sl = T.sin_lon[j], cl = T.cos_lon[j];
        const double nx = north ? -cl : cl, ny = north ? -sl : sl;
        Ts[b + j] = mT;
        uo[b + j] = (-sl) * m0 + cl * m1;       // ee_all @ v3_mean (third component is zero)
        vo[b + j] = nx * m0 + ny * m1;
    }
}

__global__ void __launch_bounds__(QD_BLOCK)
k_polar_fill(QdGeom G, QdTabs T, const uint8_t* __restrict__ land, double* __restrict__ Ts, double* __restrict__ uo,
             double* __restrict__ vo) {
    __shared__ double sm[QD_BLOCK / 64];
    const bool north = blockIdx.x == 1;
    const int i = north ? G.nlat - 1 : 0;
    if (i < G.row0 || i >= G.row0 + G.nrows) return;      // this band does not own the pole
    qd_polar_fill_row(G, T, land, Ts, uo, vo, north, sm);
}

__device__ __forceinline__ void qd_sst_clamp_cell(size_t o, double* __restrict__ sst, double tmin, double tmax, int inject,
                                                  const uint8_t* __restrict__ land, const uint8_t* __restrict__ ice, int has_ice,
                                                  double* __restrict__ Ts_atm, double* __restrict__ eta, const double* __restrict__ eta_mean,
                                                  double eta_cap) {
    // the deferred eta update of the LAST sub-step (nobody loads eta through the momentum kernel afterwards)
    if (eta) eta[o] = qd_clip(qd_nn(eta[o] - *eta_mean), -eta_cap, eta_cap);
    const double t0 = sst[o];
    const double t = qd_clip(t0, tmin, tmax);
    if (__double_as_longlong(t) != __double_as_longlong(t0)) sst[o] = t;      // (a store of the bits that are there already costs what any store costs)
    // gcm.T_s = where(ocean & ~ice, ocean.Ts, gcm.T_s)    run_simulation.py:2252-2253
    if (inject && land[o] == 0 && !(has_ice && ice[o] != 0)) Ts_atm[o] = t;
}

__global__ void __launch_bounds__(QD_BLOCK)
k_sst_clamp_inject(QdGeom G, double* __restrict__ sst, double tmin, double tmax, int inject,
                   const uint8_t* __restrict__ land, const uint8_t* __restrict__ ice, int has_ice,
                   double* __restrict__ Ts_atm, double* __restrict__ eta, const double* __restrict__ eta_mean, double eta_cap) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    qd_sst_clamp_cell(o, sst, tmin, tmax, inject, land, ice, has_ice, Ts_atm, eta, eta_mean, eta_cap);
}

// whole-globe handles: the polar fill and the clamp + write-back in ONE launch.  The clamp of a cell reads nothing but the cell, so
// only the two pole rows depend on the fill: the first workgroup of a pole row does the fill of its row and then the clamp of the
// whole row, the other workgroups of that row leave at once, every other row is clamped as before.  The pole workgroup is the long
// one (a chain of row passes around four block sums: 21 us when written as fill-then-clamp over memory), so for rows of up to
// QD_PF_CH x 256 columns it loads its row ONCE (all chunks in flight together), reduces the four sums in one exchange and writes the
// filled, clamped row from registers -- same sums in the same order (thread-strided, shuffle tree, waves in order).
#define QD_PF_CH 6
__global__ void __launch_bounds__(QD_BLOCK)
k_polar_clamp_inject(QdGeom G, QdTabs T, double* __restrict__ sst, double* __restrict__ uo, double* __restrict__ vo, double tmin, double tmax,
                     int inject, const uint8_t* __restrict__ land, const uint8_t* __restrict__ ice, int has_ice,
                     double* __restrict__ Ts_atm, double* __restrict__ eta, const double* __restrict__ eta_mean, double eta_cap) {
    __shared__ double sm[4][QD_BLOCK / 64];
    const QdTile tl = qd_tile();
    const int i = G.row0 + tl.row;
    if (i == 0 || i == G.nlat - 1) {
        if (tl.seg != 0) return;
        const bool north = i != 0;
        const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
        if (G.nlon > QD_PF_CH * QD_BLOCK) {                  // wide rows: the two-pass form
            qd_polar_fill_row(G, T, land, sst, uo, vo, north, &sm[0][0]);
            __syncthreads();                                 // (every thread re-reads only the columns it wrote itself: same stride)
            for (int j = threadIdx.x; j < G.nlon; j += QD_BLOCK)
                qd_sst_clamp_cell(b + j, sst, tmin, tmax, inject, land, ice, has_ice, Ts_atm, eta, eta_mean, eta_cap);
            return;
        }
        double t_[QD_PF_CH], u_[QD_PF_CH], v_[QD_PF_CH], sl_[QD_PF_CH], cl_[QD_PF_CH]; int ld_[QD_PF_CH];
#pragma unroll
        for (int k = 0; k < QD_PF_CH; ++k) {
            const int j = threadIdx.x + k * QD_BLOCK;
            const int jj = j < G.nlon ? j : G.nlon - 1;
            ld_[k] = j < G.nlon ? (int)land[b + jj] : 1;     // columns beyond the row count as land: they contribute nothing
            t_[k] = sst[b + jj]; u_[k] = uo[b + jj]; v_[k] = vo[b + jj]; sl_[k] = T.sin_lon[jj]; cl_[k] = T.cos_lon[jj];
        }
        double cnt = 0, sT = 0, s0 = 0, s1 = 0;
#pragma unroll
        for (int k = 0; k < QD_PF_CH; ++k) {
            if (ld_[k] != 0) continue;
            cnt += 1.0;
            sT += t_[k];
            const double nx = north ? -cl_[k] : cl_[k], ny = north ? -sl_[k] : sl_[k];
            s0 += (-sl_[k]) * u_[k] + nx * v_[k];
            s1 += cl_[k] * u_[k] + ny * v_[k];
        }
        cnt = qd_wave_sum_d(cnt); sT = qd_wave_sum_d(sT); s0 = qd_wave_sum_d(s0); s1 = qd_wave_sum_d(s1);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) { sm[0][wv] = cnt; sm[1][wv] = sT; sm[2][wv] = s0; sm[3][wv] = s1; }
        __syncthreads();
        cnt = sm[0][0]; sT = sm[1][0]; s0 = sm[2][0]; s1 = sm[3][0];
        for (int k = 1; k < QD_BLOCK / 64; ++k) { cnt += sm[0][k]; sT += sm[1][k]; s0 += sm[2][k]; s1 += sm[3][k]; }
        const bool any = cnt > 0.0;
        const double mT = sT / cnt, m0 = s0 / cnt, m1 = s1 / cnt;
#pragma unroll
        for (int k = 0; k < QD_PF_CH; ++k) {
            const int j = threadIdx.x + k * QD_BLOCK;
            if (j >= G.nlon) continue;
            const size_t o = b + j;
            double tv = t_[k];
            if (any && ld_[k] == 0) {
                const double nx = north ? -cl_[k] : cl_[k], ny = north ? -sl_[k] : sl_[k];
                tv = mT;
                uo[o] = (-sl_[k]) * m0 + cl_[k] * m1;       // ee_all @ v3_mean (third component is zero)
                vo[o] = nx * m0 + ny * m1;
            }
            // qd_sst_clamp_cell on the filled value
            if (eta) eta[o] = qd_clip(qd_nn(eta[o] - *eta_mean), -eta_cap, eta_cap);
            const double t = qd_clip(tv, tmin, tmax);
            sst[o] = t;
            if (inject && ld_[k] == 0 && !(has_ice && ice[o] != 0)) Ts_atm[o] = t;
        }
        return;
    }
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    qd_sst_clamp_cell((size_t)qd_lrow(G, i) * G.nlon + j, sst, tmin, tmax, inject, land, ice, has_ice, Ts_atm, eta, eta_mean, eta_cap);
}

// the mean of the tail kernel's strip / tile sums: one workgroup, fixed order (thread-strided partial sums, shuffle tree per
// wave, waves 0..3 in order) -- a few thousand values in ~4 us
__global__ void __launch_bounds__(256) k_eta_mean_tail(const double* __restrict__ partial, int n, double wsum, double* __restrict__ out) {
    __shared__ double sm[4];
    double a = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) a += partial[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) *out = (((sm[0] + sm[1]) + sm[2]) + sm[3]) / (wsum + 1e-15);
}

// ------------------------------------------------------------------ host orchestration
// device scalars after a reduction: raw sums/maxima are all-reduced across latitude bands first
__global__ void k_eta_mean_post(double* s, double wsum) { if (threadIdx.x == 0 && blockIdx.x == 0) *s = *s / (wsum + 1e-15); }

// a later kernel of the stream publishes "everything before me has reached host memory" (system-scope release at its end)
__global__ void k_host_flag(double* flag, double seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// spin on a flag in pinned host memory: no HIP call on the wait path, the stream keeps executing what is queued behind
int qd_ocean_step_impl(qd_ctx* c, double dt, int compute_qnet, int use_ice_mask, int inject_sst) {
    const qd_params& p = c->p;
    const QdGeom& G0 = c->geo;
    const dim3 blk(QD_BLOCK);
    double** F = c->f;
    const bool band = !G0.full;
    c->ocn_counter += 1;
    const int64_t step = c->ocn_counter;
    const double H = p.H_ocean;
    const QdGeom Gown = qd_segments(c, 0).g[0];        // owned rows: reductions never count halo rows
    const dim3 rows(1, Gown.nrows);

    // whole-globe handles launch Q_net AFTER the stress kernel (the two are independent): it runs while the host waits for the
    // CFL maxima, and the first sub-step is queued before it has drained
    auto launch_qnet = [&]() -> int {
        QdScope sc(c, "ocean_qnet");
        QdColP P = qd_make_colp_driver(c, dt);
        // cloud optical field: cloud_eff_last when time_step produced one, else cloud_cover
        double*& cl = c->cloud_eff_valid ? F[QD_F_CLOUD_EFF] : F[QD_F_CLOUD];
        const int m = qd_plan(c, {QD_IN(F[QD_F_ISR], 0), QD_IN(F[QD_F_ALBEDO], 0), QD_IN(cl, 0), QD_IN(F[QD_F_TS], 0),
                                  QD_IN(F[QD_F_H], 0), QD_IN(F[QD_F_U], 0), QD_IN(F[QD_F_V], 0), QD_IN(F[QD_F_HICE], 0),
                                  QD_IN(F[QD_F_LH], 0)});
        if (m < 0) return -1;
        QD_ROWS(c, m, G, hipLaunchKernelGGL(k_qnet, qd_grid2d(G), blk, 0, c->stream, G, P, F[QD_F_ISR], F[QD_F_ALBEDO], cl,
                                            F[QD_F_TS], F[QD_F_H], F[QD_F_U], F[QD_F_V], c->land, F[QD_F_HICE], F[QD_F_LH],
                                            F[QD_F_QNET], c->icemask));
        qd_mark(c, {F[QD_F_QNET], c->icemask}, m);
        use_ice_mask = 1;
        return 0;
    };
    if (compute_qnet && band && launch_qnet()) return -1;
    double*& taux = c->scratch[14];
    double*& tauy = c->scratch[15];
    int n_sub;
    bool band_fix_stats = false;                              // a band learns how long its fix lists are only through the hooked CFL reduce
    {
        QdScope sc(c, "ocean_stress");
        // stress on the widest valid margin (sub-steps read it under the fused kernel's halo);
        // the CFL maxima come from the owned rows, then the bands agree on them
        const int m = qd_plan(c, {QD_IN(F[QD_F_U], 0), QD_IN(F[QD_F_V], 0), QD_IN(F[QD_F_UO], 0), QD_IN(F[QD_F_VO], 0)});
        if (m < 0) return -1;
        QdSegs S = qd_segments(c, m);
        double maxVa = 0.0, maxUo = 0.0;
        if (!band) {
            // whole globe: one launch; the 2 x n_lat row maxima are written straight into pinned host memory (device-visible,
            // coherent), so the wait the host needs anyway is the only cost -- no copy kernel (measured: 16 us per step)
            const QdGeom& G = S.g[0];
            c->eta_seq += 1.0;
            bool merged_final = false;
            if (c->final_pending.on) {
                // qd_step_n left time_step's last kernel for this launch: final + stress + row maxima + Q_net in one (k_final_qnet_stress)
                if (!compute_qnet) return qd_fail(c, "ocean step: a deferred k_final needs compute_qnet");
                c->final_pending.on = 0;
                QdFqsArgs A;
                A.cosl = c->tabs.cos6; A.dt = c->final_pending.dt; A.a = p.a; A.dlat = c->dlat; A.dlon = c->dlon;
                A.decay = c->final_pending.decay; A.dfac = c->final_pending.dfac;
                A.u = F[QD_F_U]; A.v = F[QD_F_V]; A.h = F[QD_F_H]; A.Ts = F[QD_F_TS]; A.q = F[QD_F_Q];
                A.cloud_in = F[QD_F_CLOUD]; A.cloud_out = qd_scratch(c, 0);
                A.isr = F[QD_F_ISR]; A.albedo = F[QD_F_ALBEDO]; A.cloud_eff = c->cloud_eff_valid ? F[QD_F_CLOUD_EFF] : nullptr;
                A.hice = F[QD_F_HICE]; A.LH = F[QD_F_LH]; A.land = c->land; A.qnet = F[QD_F_QNET]; A.icemask = c->icemask;
                A.uo = F[QD_F_UO]; A.vo = F[QD_F_VO]; A.vcap = p.vcap; A.rhoCD = p.rho_a_ocean * p.CD; A.tau_scale = p.tau_scale;
                A.taux = taux; A.tauy = tauy;
                A.dev_wg = c->wgmax; A.n_wg = c->n_wgmax;
                merged_final = true;
                hipLaunchKernelGGL(k_final_qnet_stress, qd_grid2d(G), blk, 0, c->stream, G, qd_make_colp_driver(c, dt), A);
                hipLaunchKernelGGL(k_max2_publish, dim3(1), dim3(1024), 0, c->stream, c->wgmax, c->n_wgmax, c->hpin + 58, c->fix_count);
                qd_swap(c, QD_F_CLOUD, 0);
                use_ice_mask = 1;
            } else {
            hipLaunchKernelGGL(k_stress_max, dim3(1, G.nrows), blk, 0, c->stream, G, F[QD_F_U], F[QD_F_V], F[QD_F_UO],
                               F[QD_F_VO], p.vcap, p.rho_a_ocean * p.CD, p.tau_scale, taux, tauy, c->hpin_rows,
                               c->hpin_rows + (size_t)2 * G0.lrows(), c->eta_seq);
            qd_mark(c, {taux, tauy}, m);
            if (compute_qnet && launch_qnet()) return -1;
            }
            // qd_step_n: work of the NEXT step that depends on neither the ocean nor the sub-step count (the precipitation block:
            // ~65 us of launches) goes in here, so the device has something to do while the host waits (the gap was 24 us a step)
            if (c->before_cfl_wait) {
                std::function<int()> fn = std::move(c->before_cfl_wait);
                c->before_cfl_wait = nullptr;
                if (c->side_stream_on && c->timing != 1 && !c->ocn_fused) {
                    // QD_SIDE_STREAM: the block goes BESIDE the sub-steps, not in front of them.  It reads what time_step left (fork:
                    // everything queued so far) and writes only its own slabs and scalars; its first consumer joins (qd_side_join)
                    QD_HIP(c, hipEventRecord(c->side_fork, c->stream));
                    QD_HIP(c, hipStreamWaitEvent(c->side_stream, c->side_fork, 0));
                    std::swap(c->stream, c->side_stream);
                    const int r = fn();
                    std::swap(c->stream, c->side_stream);
                    if (r) return -1;
                    QD_HIP(c, hipEventRecord(c->side_done, c->side_stream));
                    c->side_pending = true;
                } else if (fn()) return -1;
            }
            if (merged_final) {
                if (qd_wait_host_nonneg(c, c->hpin + 58, &maxVa, "ocean step: the CFL maxima never arrived") ||
                    qd_wait_host_nonneg(c, c->hpin + 59, &maxUo, "ocean step: the CFL maxima never arrived")) return -1;
                // the same launch stored the fix lists' average length in front of the maxima (written before them by the same thread)
                { const double avg = c->hpin[57]; c->hpin[57] = -1.0; if (avg >= 0.0) c->fix_avg = avg; }
            } else {
            for (int k = 0; k < G.nrows; ++k)
                if (qd_wait_host_flag(c, c->hpin_rows + (size_t)2 * G0.lrows() + k, c->eta_seq, "ocean step: the CFL maxima never arrived")) return -1;
            for (int k = 0; k < G.nrows; ++k) { maxVa = std::max(maxVa, c->hpin_rows[k]); maxUo = std::max(maxUo, c->hpin_rows[G.nrows + k]); }
            }
        } else {
        const bool hooked = qd_peer_hooks(c) && !qd_has_host_ring(c) && S.n <= 3;
        for (int k = 0; k < S.n; ++k) {
            const QdGeom& G = S.g[k];
            // partial maxima of halo segments are harmless (a max over more valid rows of the globe)
            hipLaunchKernelGGL(k_stress_max, dim3(1, G.nrows), blk, 0, c->stream, G, F[QD_F_U], F[QD_F_V], F[QD_F_UO],
                               F[QD_F_VO], p.vcap, p.rho_a_ocean * p.CD, p.tau_scale, taux, tauy,
                               c->red_partial + (size_t)k * 2 * G0.lrows(), (double*)nullptr, 0.0);
            if (!hooked)
            hipLaunchKernelGGL(k_max2_finish, dim3(1), blk, 0, c->stream, c->red_partial + (size_t)k * 2 * G0.lrows(), G.nrows,
                               c->dscal + QD_S_TMP0 + 2 * k, k == S.n - 1 ? 2 * (3 - S.n) : 0);
        }
        qd_mark(c, {taux, tauy}, m);
        if (hooked) {
            // the reducing kernel finishes the row maxima of every segment itself (QdPeerHook::pre 3 = k_max2_finish), all-reduces the
            // six maxima and hands them to the host: one launch
            QdPeerHook H; H.pre = 3; H.partial = c->red_partial; H.pstride = (size_t)2 * G0.lrows(); H.nseg = S.n;
            for (int k = 0; k < S.n; ++k) H.nsegrows[k] = S.g[k].nrows;
            c->allreduces++;
            c->pub_seq += 1.0;
            // (a seventh value rides along: the longest average fix list of any band since the last step -- every band takes the same form)
            H.fixstat = (c->tail_fix && c->fix_count) ? c->fix_count : nullptr;
            if (qd_peer_allreduce_hooked(c, c->dscal + QD_S_TMP0, H.fixstat ? 7 : 6, 1, H, c->hpin, c->pub_seq)) return -1;
            if (qd_wait_host_flag(c, c->hpin + 61, c->pub_seq, "reduced scalars never reached the host")) return -1;
            if (c->hpin[60] != 0.0) return qd_fail(c, "peer exchange: a rank did not arrive within the deadline");
            if (H.fixstat && c->hpin[6] >= 0.0) c->fix_avg = c->hpin[6];
            band_fix_stats = H.fixstat != nullptr;
        } else if (qd_has_host_ring(c)) {
            // the host waits for these six maxima anyway: reduce them across the ranks in the host ring, no RCCL launch
            QD_HIP(c, hipMemcpyAsync(c->hpin, c->dscal + QD_S_TMP0, 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            QD_HIP(c, hipStreamSynchronize(c->stream));
            if (qd_host_allreduce(c, c->hpin, 6, 1)) return -1;
        } else {
        // ONE collective of fixed size on every band (polar bands have more segments than interior ones)
        // ... stored into pinned host memory behind a stamp the host polls (no hipStreamSynchronize: its wake-up alone idled the stream
        // for ~20 us per step; over the peer exchange the reducing kernel publishes them itself)
        if (qd_allreduce_fetch(c, c->dscal + QD_S_TMP0, 6, 1, c->hpin)) return -1;
        }
        for (int k = 0; k < 3; ++k) { maxVa = std::max(maxVa, c->hpin[2 * k]); maxUo = std::max(maxUo, c->hpin[2 * k + 1]); }
        }
        // ocean.py:293-303
        const double dx_lat = p.a * c->dlat;
        const double min_cos = 0.5;                       // min of max(cos, 0.5) on a pole-to-pole grid
        const double dx_lon_min = p.a * c->dlon * std::max(1e-3, min_cos);
        const double dx_min = std::min(dx_lat, dx_lon_min);
        const double cg = std::sqrt(p.g_ocean * H);
        double uadv = std::max(maxUo, maxVa);
        const double target = std::max(1e-3, p.ocean_cfl);
        const double v = std::ceil(std::max(cg, uadv) * (dt / std::max(1e-12, dx_min)) / target);
        n_sub = (v != v) ? 1 : (v > 500.0 ? 500 : (v < 1.0 ? 1 : (int)v));
        c->last_nsub = n_sub;
    }
    // the ocean tail's fix list (QD_TAIL_FIX) pays while few cells change (a run from rest: ~40 per launch, -1.2 us per launch on the whole
    // globe, -2.5 on a 1/8 band) and costs once the polar currents sit at the cap (~700 per launch at step 240 of the benchmark: the
    // finishing wave's patch loop, +1.3 us): taken while the lists of the last step that had any averaged <= fix_dense entries, probed
    // again 32, 64, ... 256 steps later while it is off
    bool fix_now = c->tail_fix && c->fix_count && (!band || band_fix_stats);
    if (fix_now) {
        if (c->fix_avg <= c->fix_dense) c->fix_probe_every = 32;                          // short lists: the list stays on
        else if (step >= c->fix_probe_at) {                                                 // long lists: a probe is due (its result arrives with the next step's maxima)
            c->fix_probe_at = step + c->fix_probe_every;
            c->fix_probe_every = std::min(256, c->fix_probe_every * 2);
        } else fix_now = false;
    }
    const double sub_dt = dt / n_sub;
    QdOcnP OP{p.a, p.g_ocean, c->dlat, c->dlon, sub_dt, p.rho_w * H, p.r_bot};
    QdHeatP HP{sub_dt, p.K_h, p.rho_w * p.cp_w * H, p.ocean_ice_qfac, p.ocean_use_qnet ? 1 : 0, use_ice_mask ? 1 : 0};
    const bool do_diff = (p.ocean_diff_every > 0) && (step % p.ocean_diff_every == 0);
    const bool do_shap = (p.ocean_shapiro_n > 0) && (p.ocean_shapiro_every > 0) && (step % p.ocean_shapiro_every == 0);
    if (do_diff) { int rc = qd_build_k4_tables(c, dt, true, sub_dt); if (rc) return rc; }
    const double ov[3] = {p.ocean_k4_u, p.ocean_k4_v, p.ocean_k4_eta};
    const int Ro = qd_adv_reach(c, sub_dt, 4.0);           // currents are capped at QD_OCEAN_MAX_U (3 m/s)

    // whole-globe handles on the fused path defer "eta -= mean; nan_to_num; clip" of a sub-step to the load of the next
    // momentum kernel (and to k_eta_finalize after the last one): no k_eta_mean launch, no eta pass in the SST kernel
    const bool can_defer = do_diff && c->use_fused && p.ocean_k4_nsub == 1 && !do_shap && c->wsum_ocean > 0.0;
    // latitude bands with a host ring defer the same way; their mean is reduced on the HOST between the ranks (event wait on the
    // continuity kernel while the SST kernel is already queued) and handed to the next momentum kernel in pinned host memory
    const bool band_defer = band && can_defer && qd_has_host_ring(c);
    // latitude bands WITHOUT a host ring run the same two kernels per sub-step as the whole globe (round 3): the streaming tail kernel
    // on the band's segments, its in-launch sum finished as the band's SHARE of the mean (sum_b / (W + 1e-15), W the global ocean
    // weight), one all-reduce of that scalar, applied by the next momentum kernel on load.  QD_BAND_TAIL=0: the round-2 band path.
    // (the sub-step is then planned two rows wider -- eta 7, currents 6, SST Ro + 2 --: handles with a thinner halo keep the round-2 path)
    const bool band_tail = band && can_defer && !band_defer && c->ocn_tail == 1 && c->tail_acc && c->band_tail && G0.nlon >= 64 &&
                           G0.halo >= std::max(7, Ro + 2);
    const bool defer_eta = (!band && can_defer) || band_defer || band_tail;
    double* const mean_ptr = band_defer ? c->hpin + 42 : c->dscal + QD_S_ETA_MEAN;
    // latitude bands: whenever a sub-step has to exchange halos, every slab of the sub-step loop is refreshed in the same group
    struct CoRefresh { qd_ctx* c; ~CoRefresh() { c->corefresh.clear(); } } corefresh_guard{c};
    if (band) c->corefresh = {QD_IN(F[QD_F_UO], 0), QD_IN(F[QD_F_VO], 0), QD_IN(F[QD_F_ETA], 0), QD_IN(F[QD_F_SST], 0), QD_IN(taux, 0),
                              QD_IN(tauy, 0), QD_IN(F[QD_F_QNET], 0), QD_IN8(c->icemask, 0)};
    // whole globe, deferred mean, >= 64 columns: the rest of a sub-step is one launch (k_ocn_tail) + a one-wave kernel that turns
    // its per-tile eta sums into the mean.  Measured alternatives, both slower: the tail kernel's last workgroup doing it behind
    // two-level tickets (+12 us per sub-step: ticket round trips, acquire fence and read-back form a serial chain at the very end
    // of the launch); every wave of the next momentum kernel adding the ~1100 sums itself (+3.7 us per launch).
    const bool use_tail = c->use_fused && (!band || band_tail) && defer_eta && G0.nlon >= 64 && c->ocn_tail;
    // QD_TAIL_ACC (default): the streaming tail kernel reduces its own strip sums (fixed-point atomics + spread tickets, qd_wave.h)
    // and its last workgroup writes the mean -- no k_eta_mean_tail launch (4.5 us of launch floor per sub-step)
    const bool tail_acc = use_tail && c->tail_acc;
    // QD_OCN_FUSED=1: the WHOLE sub-step is one launch (k_ocn_fused, qd_ocntail.hip): the momentum + del^4 waves hand their rows to the
    // tail waves of the same strip through LDS rings; uo', vo', eta' never reach memory
    const bool use_fused1 = use_tail && !band && c->ocn_fused && c->ocn_tail == 1 && tail_acc && c->fused_fast == 1 && do_diff &&
                            p.ocean_k4_nsub == 1 && !do_shap && HP.use_q && HP.K_h > 0.0 && qd_ocn_fused_ok(c) &&
                            !c->k4_ocn_skip[0] && !c->k4_ocn_skip[1] && !c->k4_ocn_skip[2];
    for (int s = 0; s < n_sub; ++s) {
        if (use_fused1) {
            QdOcnArgs O;
            O.uo = F[QD_F_UO]; O.vo = F[QD_F_VO]; O.eta = F[QD_F_ETA]; O.taux = taux; O.tauy = tauy; O.land = c->land;
            // (scratch slabs of the sequential form: the polar tiles and any strip whose fast waves gave up)
            O.uo_out = qd_scratch(c, 0); O.vo_out = qd_scratch(c, 1); O.eta_out = qd_scratch(c, 6);
            for (int f = 0; f < 3; ++f) {
                const bool sc1 = !(ov[f] != ov[f]);
                O.k4row[f] = sc1 ? nullptr : c->k4_ocn + (size_t)f * G0.nlat;
                O.k4s[f] = sc1 ? ov[f] : 0.0;
                O.skip[f] = c->k4_ocn_skip[f];
            }
            O.a = p.a; O.g = p.g_ocean; O.dlat = c->dlat; O.dlon = c->dlon; O.sub_dt = sub_dt; O.rhoH = p.rho_w * H; O.r_bot = p.r_bot;
            O.inv_2dlon = 1.0 / (2.0 * c->dlon); O.inv_2dlat = 1.0 / (2.0 * c->dlat); O.inv_a = 1.0 / p.a; O.inv_rhoH = 1.0 / (p.rho_w * H);
            O.eta_mean = s > 0 ? mean_ptr : nullptr;
            O.eta_cap = p.eta_cap;
            QdTailArgs A;
            A.uo = O.uo_out; A.vo = O.vo_out; A.eta_in = O.eta_out; A.Ts = F[QD_F_SST]; A.qnet = F[QD_F_QNET]; A.land = c->land; A.ice = c->icemask;
            A.eta = qd_scratch(c, 2); A.Ts_out = qd_scratch(c, 3); A.uo_out = qd_scratch(c, 12); A.vo_out = qd_scratch(c, 13); A.partial = c->red_partial;
            A.a = p.a; A.dlat = c->dlat; A.dlon = c->dlon; A.sub_dt = sub_dt; A.msdtH = -sub_dt * H; A.alpha = p.ocean_adv_alpha;
            A.K_h = HP.K_h; A.rcH = HP.rcH; A.ice_qfac = HP.ice_qfac; A.cap = p.ocean_max_u;
            A.use_q = HP.use_q; A.has_ice = HP.has_ice; A.mean4 = p.ocean_outlier == 0 ? 1 : 0;
            A.r_a = 1.0 / p.a; A.r_dlon = 1.0 / c->dlon; A.r_dlat = 1.0 / c->dlat; A.r_2dlon = 1.0 / (2 * c->dlon); A.r_2dlat = 1.0 / (2 * c->dlat);
            A.r_rcH = 1.0 / HP.rcH;
            A.acc = c->eta_acc; A.mean_out = c->dscal + QD_S_ETA_MEAN; A.wsum = c->wsum_ocean;
            if (qd_launch_ocn_fused(c, O, A)) return -1;
            qd_swap(c, QD_F_UO, 12); qd_swap(c, QD_F_VO, 13); qd_swap(c, QD_F_ETA, 2); qd_swap(c, QD_F_SST, 3);
            continue;
        }
        if (do_diff && c->use_fused && p.ocean_k4_nsub == 1) {
            // band_tail: every exchange of the sub-step happens HERE, so that the previous sub-step's eta sum (still pending) can ride
            // in its group: the momentum kernel is planned two rows wider than it needs (its outputs then carry the margin the tail's
            // T1 rows want) and the tail's other inputs are checked now instead of between the two launches
            const QdUse narrow[5] = {QD_IN(F[QD_F_ETA], 5), QD_IN(F[QD_F_UO], 4), QD_IN(F[QD_F_VO], 4), QD_IN(taux, 4), QD_IN(tauy, 4)};
            bool pending = false;                             // the exchange of this sub-step has been pushed, not yet unpacked
            int m_old = 0;
            if (band_tail) {
                const QdUse wide[8] = {QD_IN(F[QD_F_ETA], 7), QD_IN(F[QD_F_UO], 6), QD_IN(F[QD_F_VO], 6), QD_IN(taux, 6), QD_IN(tauy, 6),
                                       QD_IN(F[QD_F_SST], Ro + 2), QD_IN(F[QD_F_QNET], 0), QD_IN8(c->icemask, 0)};
                if (qd_plan_begin(c, wide, 8, &pending) < 0) return -1;
                if (pending) m_old = qd_plan_peek(c, narrow, 5);      // what the momentum kernel can compute BEFORE the halos arrive
            }
            QdOcnArgs O;
            O.uo = F[QD_F_UO]; O.vo = F[QD_F_VO]; O.eta = F[QD_F_ETA]; O.taux = taux; O.tauy = tauy; O.land = c->land;
            O.uo_out = qd_scratch(c, 0); O.vo_out = qd_scratch(c, 1); O.eta_out = qd_scratch(c, 2);
            for (int f = 0; f < 3; ++f) {
                const bool sc1 = !(ov[f] != ov[f]);
                O.k4row[f] = sc1 ? nullptr : c->k4_ocn + (size_t)f * G0.nlat;
                O.k4s[f] = sc1 ? ov[f] : 0.0;
                O.skip[f] = c->k4_ocn_skip[f];
            }
            O.a = p.a; O.g = p.g_ocean; O.dlat = c->dlat; O.dlon = c->dlon; O.sub_dt = sub_dt; O.rhoH = p.rho_w * H; O.r_bot = p.r_bot;
            O.inv_2dlon = 1.0 / (2.0 * c->dlon); O.inv_2dlat = 1.0 / (2.0 * c->dlat); O.inv_a = 1.0 / p.a; O.inv_rhoH = 1.0 / (p.rho_w * H);
            O.eta_mean = (defer_eta && s > 0) ? mean_ptr : nullptr;
            O.eta_cap = p.eta_cap;
            // Exchange overlapped with interior compute (peer exchange, QD_PEER_OVERLAP): the rows whose stencils stay inside the OLD
            // margins are launched between the push and the unpack -- the halo rows travel while they run -- and only the two
            // boundary strips wait for the neighbours.  Rows do not depend on the strip they are computed in: bit-identical.
            int m_int = 0;
            bool split = false;
            if (pending) {
                // a negative old margin shrinks the interior; streaming segments must start / end at a pole or >= 5 rows from it
                m_int = m_old >= 0 ? m_old : -12;
                const QdSegList Si = qd_segments_rows(c, c->own_row0 - m_int, c->own_nrows + 2 * m_int);
                if (c->own_nrows + 2 * m_int >= 12 && qd_ocn_stream_ok_list(c, Si)) {
                    if (qd_launch_ocn_stream_list(c, O, Si)) return -1;
                    split = true;
                }
                if (qd_plan_end(c)) return -1;                // wait for both neighbours' rows, copy them into the halos
            }
            // (after the wide plan this one never exchanges: it only tells how far the momentum kernel can go -- at least two rows
            //  beyond what the tail will store)
            const int m = qd_plan(c, narrow, 5);
            if (m < 0) return -1;
            if (qd_allreduce_flush(c)) return -1;             // the previous sub-step's eta sum, unless the exchange above took it along
            if (split && m > m_int) {
                QdSegList Sb = qd_segments_rows(c, c->own_row0 - m, m - m_int);
                Sb = qd_segments_rows(c, c->own_row0 + c->own_nrows + m_int, m - m_int, Sb);
                if (!qd_ocn_stream_ok_list(c, Sb)) {          // boundary strips too thin for the streaming kernel: redo the whole launch (same values)
                    if (qd_launch_ocn_hyper(c, O, m)) return -1;
                } else if (qd_launch_ocn_stream_list(c, O, Sb)) return -1;
            } else if (!split) {
                if (qd_launch_ocn_hyper(c, O, m)) return -1;
            }
            qd_mark(c, {O.uo_out, O.vo_out, O.eta_out}, m);
            qd_swap(c, QD_F_UO, 0); qd_swap(c, QD_F_VO, 1); qd_swap(c, QD_F_ETA, 2);
        } else {
            {
                QdScope sc(c, "ocean_momentum");
                const int m = qd_plan(c, {QD_IN(F[QD_F_ETA], 1), QD_IN(F[QD_F_UO], 0), QD_IN(F[QD_F_VO], 0), QD_IN(taux, 0),
                                          QD_IN(tauy, 0)});
                if (m < 0) return -1;
                QD_ROWS(c, m, G, hipLaunchKernelGGL(k_ocean_momentum, qd_grid2d(G), blk, 0, c->stream, G, c->tabs, OP,
                                                    F[QD_F_ETA], taux, tauy, c->land, F[QD_F_UO], F[QD_F_VO]));
                qd_mark(c, {F[QD_F_UO], F[QD_F_VO]}, m);
            }
            if (do_diff) {
                QdScope sc(c, "ocean_hyperdiffusion");
                const int ns = std::max(1, p.ocean_k4_nsub);
                const int m = qd_plan(c, {QD_IN(F[QD_F_UO], 4 * ns), QD_IN(F[QD_F_VO], 4 * ns), QD_IN(F[QD_F_ETA], 4 * ns)});
                if (m < 0) return -1;
                double* fl[3] = {F[QD_F_UO], F[QD_F_VO], F[QD_F_ETA]};
                c->lap_tag = "ocean_k_laplacian"; c->hyp_tag = "ocean_k_hyper_apply";
                qd_hyperdiffuse_fields(c, fl, 3, c->k4_ocn, c->k4_ocn_skip, ov, sub_dt, ns, c->tabs.cos05, m);
                c->lap_tag = "k_laplacian"; c->hyp_tag = "k_hyper_apply";
                F[QD_F_UO] = fl[0]; F[QD_F_VO] = fl[1]; F[QD_F_ETA] = fl[2];
            }
        }
        if (do_shap) {
            const int np_ = p.ocean_shapiro_n;
            const int m = qd_plan(c, {QD_IN(F[QD_F_UO], np_), QD_IN(F[QD_F_VO], np_), QD_IN(F[QD_F_ETA], np_)});
            if (m < 0) return -1;
            double* fl[3] = {F[QD_F_UO], F[QD_F_VO], F[QD_F_ETA]};
            qd_shapiro_fields(c, fl, 3, np_, m);
            F[QD_F_UO] = fl[0]; F[QD_F_VO] = fl[1]; F[QD_F_ETA] = fl[2];
        }
        if (use_tail) {
            // whole globe, deferred mean: the rest of the sub-step is ONE launch (qd_ocntail.hip) + the mean of its tile sums
            QdTailArgs A;
            A.uo = F[QD_F_UO]; A.vo = F[QD_F_VO]; A.Ts = F[QD_F_SST]; A.qnet = F[QD_F_QNET]; A.land = c->land; A.ice = c->icemask;
            A.eta = F[QD_F_ETA]; A.Ts_out = qd_scratch(c, 1); A.uo_out = qd_scratch(c, 2); A.vo_out = qd_scratch(c, 3); A.partial = c->red_partial;
            A.a = p.a; A.dlat = c->dlat; A.dlon = c->dlon; A.sub_dt = sub_dt; A.msdtH = -sub_dt * H; A.alpha = p.ocean_adv_alpha;
            A.K_h = HP.K_h; A.rcH = HP.rcH; A.ice_qfac = HP.ice_qfac; A.cap = p.ocean_max_u;
            A.use_q = HP.use_q; A.has_ice = HP.has_ice; A.mean4 = p.ocean_outlier == 0 ? 1 : 0;
            A.r_a = 1.0 / p.a; A.r_dlon = 1.0 / c->dlon; A.r_dlat = 1.0 / c->dlat; A.r_2dlon = 1.0 / (2 * c->dlon); A.r_2dlat = 1.0 / (2 * c->dlat);
            A.r_rcH = 1.0 / HP.rcH;
            A.acc = tail_acc ? c->eta_acc : nullptr; A.mean_out = c->dscal + QD_S_ETA_MEAN; A.wsum = c->wsum_ocean;
            if (band) {
                // T1 = the blended, gathered SST is needed two rows beyond the rows a launch stores (K_h lap), uo / vo with it
                const int m = qd_plan(c, {QD_IN(F[QD_F_UO], 2), QD_IN(F[QD_F_VO], 2), QD_IN(F[QD_F_ETA], 0), QD_IN(F[QD_F_SST], Ro + 2),
                                          QD_IN(F[QD_F_QNET], 0), QD_IN8(c->icemask, 0)});
                if (m < 0) return -1;
                A.own0 = c->own_row0; A.own1 = c->own_row0 + c->own_nrows;
                QdSegs S = qd_segments(c, m);
                size_t off = 0;
                // peer exchange: the finishing wave of the owned segment's launch all-reduces the band's share itself (no launch)
                QdPeerFold pf;
                const bool folded = qd_peer_fold_begin(c, &pf);
                // one launch that covers the band and has the finishing wave: uo'' / vo'' in place through the fix list, like the whole globe
                if (fix_now && S.n == 1) { A.fix_count = c->fix_count; A.fix_list = c->fix_list; }
                for (int k = 0; k < S.n; ++k) {
                    const QdGeom& Gs = S.g[k];
                    const bool owned = Gs.row0 <= c->own_row0 && c->own_row0 < Gs.row0 + Gs.nrows;      // the segment that holds the band's own rows
                    A.acc = owned ? c->eta_acc : nullptr;
                    A.pf = owned ? pf : QdPeerFold();
                    A.partial = c->red_partial + off;
                    if (qd_launch_ocn_tail(c, Gs, A)) return -1;
                    off += (size_t)qd_ocn_tail_tiles(c, Gs);
                }
                // first read by the NEXT momentum kernel: if that kernel's inputs need a halo exchange, the sum rides in its group
                if (!folded && qd_allreduce_sum_deferred(c, c->dscal + QD_S_ETA_MEAN)) return -1;
                if (A.fix_count) qd_mark(c, {F[QD_F_ETA], A.Ts_out, F[QD_F_UO], F[QD_F_VO]}, m);
                else qd_mark(c, {F[QD_F_ETA], A.Ts_out, A.uo_out, A.vo_out}, m);
            } else {
            // uo'' / vo'' in place through the fix list (the launcher drops it when the kernel it picks has no finishing wave)
            if (fix_now) { A.fix_count = c->fix_count; A.fix_list = c->fix_list; }
            if (qd_launch_ocn_tail(c, Gown, A)) return -1;
            if (!tail_acc)
                hipLaunchKernelGGL(k_eta_mean_tail, dim3(1), dim3(256), 0, c->stream, c->red_partial, qd_ocn_tail_tiles(c, Gown), c->wsum_ocean,
                                   c->dscal + QD_S_ETA_MEAN);
            }
            qd_swap(c, QD_F_SST, 1);
            if (!A.fix_count) { qd_swap(c, QD_F_UO, 2); qd_swap(c, QD_F_VO, 3); }
        } else if (c->use_fused) {
            QdScope sc(c, "ocean_cont_sst");
            bool band_raw_mean = false;
            const int m = qd_plan(c, {QD_IN(F[QD_F_UO], 1), QD_IN(F[QD_F_VO], 1), QD_IN(F[QD_F_ETA], 0), QD_IN(F[QD_F_SST], Ro)});
            if (m < 0) return -1;
            double* T1 = qd_scratch(c, 0);
            const int own0 = c->own_row0, own1 = c->own_row0 + c->own_nrows;
            const int has_ocean = c->wsum_ocean > 0.0 ? 1 : 0;
            if (!band) {
                // (a last-workgroup ticket here costs more than the 5 us finishing kernel: thousands of workgroups
                //  contending on one counter word run at ~90 tickets/us)
                const dim3 g2 = qd_grid2d(Gown);
                hipLaunchKernelGGL(k_cont_sstadv, g2, blk, 0, c->stream, Gown, c->tabs, p.a, c->dlat, c->dlon, -sub_dt * H, sub_dt,
                                   p.ocean_adv_alpha, F[QD_F_UO], F[QD_F_VO], c->land, F[QD_F_ETA], F[QD_F_SST], T1, own0, own1,
                                   c->red_partial, (unsigned long long*)nullptr, 0.0, (double*)nullptr);
                if (!defer_eta)
                    hipLaunchKernelGGL(k_eta_mean, dim3(1), blk, 0, c->stream, c->red_partial, (int)(g2.x * g2.y), c->wsum_ocean,
                                       c->dscal + QD_S_ETA_MEAN);
            } else {
                // halo segments first (their partials are zero-weighted), the owned rows' partials are summed below
                QdSegs S = qd_segments(c, m);
                size_t off = 0;
                for (int k = 0; k < S.n; ++k) {
                    hipLaunchKernelGGL(k_cont_sstadv, qd_grid2d(S.g[k]), blk, 0, c->stream, S.g[k], c->tabs, p.a, c->dlat, c->dlon,
                                       -sub_dt * H, sub_dt, p.ocean_adv_alpha, F[QD_F_UO], F[QD_F_VO], c->land, F[QD_F_ETA],
                                       F[QD_F_SST], T1, own0, own1, c->red_partial + off, (unsigned long long*)nullptr, 0.0,
                                       (double*)nullptr);
                    off += (size_t)S.g[k].nrows * qd_grid2d(S.g[k]).x;
                }
                if (band_defer) {
                    // raw sum of the owned rows straight into pinned host memory; the host picks it up after the SST launch
                    c->eta_seq += 1.0;
                    hipLaunchKernelGGL(k_eta_mean, dim3(1), blk, 0, c->stream, c->red_partial, (int)off, -1.0, c->hpin + 40, c->eta_seq);
                } else {
                hipLaunchKernelGGL(k_eta_mean, dim3(1), blk, 0, c->stream, c->red_partial, (int)off, -1.0, c->dscal + QD_S_ETA_MEAN);
                if (qd_allreduce_f64(c, c->dscal + QD_S_ETA_MEAN, 1, 0)) return -1;
                band_raw_mean = !defer_eta;      // the SST kernel divides the raw sum itself: no k_eta_mean_post launch (4.5 us per sub-step)
                if (!band_raw_mean)
                    hipLaunchKernelGGL(k_eta_mean_post, dim3(1), dim3(64), 0, c->stream, c->dscal + QD_S_ETA_MEAN, c->wsum_ocean);
                }
            }
            qd_mark(c, {T1, F[QD_F_ETA]}, m);
            double*& T1s = c->scratch[0];
            double* T2 = qd_scratch(c, 1);
            const int m2 = qd_plan(c, {QD_IN(T1s, 2), QD_IN(F[QD_F_QNET], 0), QD_IN8(c->icemask, 0), QD_IN(F[QD_F_UO], 1),
                                       QD_IN(F[QD_F_VO], 1), QD_IN(F[QD_F_ETA], 0)});
            if (m2 < 0) return -1;
            double* u2 = qd_scratch(c, 2); double* v2 = qd_scratch(c, 3);
            QD_ROWS(c, m2, G, hipLaunchKernelGGL(k_sst_outlier_fused, qd_grid2d(G), blk, 0, c->stream, G, c->tabs, c->dlat,
                                                 c->dlon, p.a, HP, c->scratch[0], T2, F[QD_F_QNET], c->land, c->icemask,
                                                 F[QD_F_UO], F[QD_F_VO], u2, v2, F[QD_F_ETA], p.ocean_max_u, p.eta_cap,
                                                 p.ocean_outlier == 0 ? 1 : 0,
                                                 has_ocean ? c->dscal + QD_S_ETA_MEAN : (const double*)nullptr,
                                                 c->red_partial, band_defer ? 0 : (band_raw_mean ? -1 : (int)(qd_grid2d(Gown).x * qd_grid2d(Gown).y)), c->wsum_ocean,
                                                 band_defer ? c->dscal + QD_S_TMP1 : (defer_eta ? c->dscal + QD_S_ETA_MEAN : (double*)nullptr)));
            qd_mark(c, {T2, u2, v2, F[QD_F_ETA]}, m2);
            if (band_defer) {
                // the SST kernel is queued; meanwhile: wait for the continuity kernel's sum (the flag it writes into pinned host
                // memory: no HIP call on the wait path), reduce it between the ranks, publish
                if (qd_wait_host_flag(c, c->hpin + 41, c->eta_seq, "ocean sub-step: the eta sum never arrived")) return -1;
                double sum = c->hpin[40];
                if (qd_host_allreduce(c, &sum, 1, 0)) return -1;
                c->hpin[42] = sum / (c->wsum_ocean + 1e-15);
            }
            qd_swap(c, QD_F_SST, 1); qd_swap(c, QD_F_UO, 2); qd_swap(c, QD_F_VO, 3);
        } else {
        {
            QdScope sc(c, "ocean_continuity");
            // eta update on the margin; the area-weighted sum only over owned rows
            const int m = qd_plan(c, {QD_IN(F[QD_F_UO], 1), QD_IN(F[QD_F_VO], 1), QD_IN(F[QD_F_ETA], 0)});
            if (m < 0) return -1;
            if (band && m > 0) {
                // halo rows first (their partial sums are discarded), owned rows last
                QdSegs S = qd_segments(c, m);
                for (int k = 0; k < S.n; ++k) {
                    QdGeom G = S.g[k];
                    // trim the owned part out of this segment: it is done by the dedicated owned launch below
                    const int own0 = c->own_row0, own1 = c->own_row0 + c->own_nrows;
                    int a0 = G.row0, a1 = G.row0 + G.nrows;
                    auto run = [&](int r0, int r1) {
                        if (r1 <= r0) return;
                        QdGeom g2 = G; g2.row0 = r0; g2.nrows = r1 - r0;
                        hipLaunchKernelGGL(k_continuity, dim3(1, g2.nrows), blk, 0, c->stream, g2, c->tabs, p.a, c->dlat, c->dlon,
                                           -sub_dt * H, F[QD_F_UO], F[QD_F_VO], c->land, F[QD_F_ETA],
                                           c->red_partial + (size_t)G0.lrows());
                    };
                    if (a1 <= own0 || a0 >= own1) run(a0, a1);
                    else { run(a0, std::max(a0, own0)); run(std::min(a1, own1), a1); }
                }
            }
            hipLaunchKernelGGL(k_continuity, rows, blk, 0, c->stream, Gown, c->tabs, p.a, c->dlat, c->dlon, -sub_dt * H,
                               F[QD_F_UO], F[QD_F_VO], c->land, F[QD_F_ETA], c->red_partial);
            qd_mark(c, {F[QD_F_ETA]}, m);
            if (band) {
                hipLaunchKernelGGL(k_eta_mean, dim3(1), blk, 0, c->stream, c->red_partial, Gown.nrows, -1.0,
                                   c->dscal + QD_S_ETA_MEAN);      // raw sum
                if (qd_allreduce_f64(c, c->dscal + QD_S_ETA_MEAN, 1, 0)) return -1;
                hipLaunchKernelGGL(k_eta_mean_post, dim3(1), dim3(64), 0, c->stream, c->dscal + QD_S_ETA_MEAN, c->wsum_ocean);
            } else {
                hipLaunchKernelGGL(k_eta_mean, dim3(1), blk, 0, c->stream, c->red_partial, Gown.nrows, c->wsum_ocean,
                                   c->dscal + QD_S_ETA_MEAN);
            }
        }
        {
            QdScope sc(c, "ocean_sst");
            const int m1 = qd_plan(c, {QD_IN(F[QD_F_SST], Ro), QD_IN(F[QD_F_UO], 0), QD_IN(F[QD_F_VO], 0), QD_IN(F[QD_F_ETA], 0)});
            if (m1 < 0) return -1;
            double* T1 = qd_scratch(c, 0);
            QD_ROWS(c, m1, G, hipLaunchKernelGGL(k_sst_advect, qd_grid2d(G), blk, 0, c->stream, G, c->tabs.cos05, sub_dt, p.a,
                                                 c->dlat, c->dlon, F[QD_F_UO], F[QD_F_VO], F[QD_F_SST], T1, p.ocean_adv_alpha,
                                                 F[QD_F_ETA], c->dscal + QD_S_ETA_MEAN, c->wsum_ocean > 0.0 ? 1 : 0));
            qd_mark(c, {T1, F[QD_F_ETA]}, m1);
            double*& T1s = c->scratch[0];
            double* T2 = qd_scratch(c, 1);
            const int m2 = qd_plan(c, {QD_IN(T1s, 2), QD_IN(F[QD_F_QNET], 0), QD_IN8(c->icemask, 0)});
            if (m2 < 0) return -1;
            QD_ROWS(c, m2, G, hipLaunchKernelGGL(k_sst_diffuse_heat, qd_grid2d(G), blk, 0, c->stream, G, c->tabs.cos05,
                                                 c->dlat, c->dlon, p.a, HP, c->scratch[0], T2, F[QD_F_QNET], c->land,
                                                 c->icemask));
            qd_mark(c, {T2}, m2);
            qd_swap(c, QD_F_SST, 1);
        }
        }
        if (!c->use_fused) {
            QdScope sc(c, "ocean_outlier");
            const int m = qd_plan(c, {QD_IN(F[QD_F_UO], 1), QD_IN(F[QD_F_VO], 1), QD_IN(F[QD_F_ETA], 0)});
            if (m < 0) return -1;
            double* u2 = qd_scratch(c, 2); double* v2 = qd_scratch(c, 3);
            QD_ROWS(c, m, G, hipLaunchKernelGGL(k_outlier, qd_grid2d(G), blk, 0, c->stream, G, F[QD_F_UO], F[QD_F_VO], u2, v2,
                                                F[QD_F_ETA], p.ocean_max_u, p.eta_cap, p.ocean_outlier == 0 ? 1 : 0));
            qd_mark(c, {u2, v2, F[QD_F_ETA]}, m);
            qd_swap(c, QD_F_UO, 2); qd_swap(c, QD_F_VO, 3);
        }
    }
    if (qd_allreduce_flush(c)) return -1;
    {
        QdScope sc(c, "ocean_finish");
        if (!band && p.ocean_polar_fix && c->merge_pointwise) {
            hipLaunchKernelGGL(k_polar_clamp_inject, qd_grid2d(G0), blk, 0, c->stream, G0, c->tabs, F[QD_F_SST], F[QD_F_UO], F[QD_F_VO],
                               p.ts_min, p.ts_max, inject_sst, c->land, c->icemask, use_ice_mask ? 1 : 0, F[QD_F_TS],
                               (defer_eta && n_sub > 0) ? F[QD_F_ETA] : (double*)nullptr, mean_ptr, p.eta_cap);
            return 0;
        }
        if (p.ocean_polar_fix) {
            // the two pole rows are owned by the first / last band; their copies in the other polar
            // band's wrap halo go stale -> margins drop to 0 so the next stencil refreshes them
            hipLaunchKernelGGL(k_polar_fill, dim3(2), blk, 0, c->stream, Gown, c->tabs, c->land, F[QD_F_SST], F[QD_F_UO],
                               F[QD_F_VO]);
            qd_mark(c, {F[QD_F_SST], F[QD_F_UO], F[QD_F_VO]}, 0);
        }
        const int m = qd_plan(c, {QD_IN(F[QD_F_SST], 0), QD_IN(F[QD_F_TS], 0), QD_IN8(c->icemask, 0)});
        if (m < 0) return -1;
        QD_ROWS(c, m, G, hipLaunchKernelGGL(k_sst_clamp_inject, qd_grid2d(G), blk, 0, c->stream, G, F[QD_F_SST], p.ts_min, p.ts_max,
                                            inject_sst, c->land, c->icemask, use_ice_mask ? 1 : 0, F[QD_F_TS],
                                            (defer_eta && n_sub > 0) ? F[QD_F_ETA] : (double*)nullptr, mean_ptr, p.eta_cap));
        qd_mark(c, {F[QD_F_SST]}, m);
        if (inject_sst) qd_mark(c, {F[QD_F_TS]}, m);
        if (defer_eta && n_sub > 0) qd_mark(c, {F[QD_F_ETA]}, m);     // rows beyond m still hold eta before "- mean, clip"
    }
    return 0;
}
