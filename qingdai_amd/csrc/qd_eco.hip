// qd_eco.hip -- the per-physics-step part of the ecology (BASELINE config 5; SURVEY.md 8(f)3, stages 2-4), gfx950.
//
//   PopulationManager.step_subdaily / total_LAI            pygcm/ecology/population.py:252-294
//   canopy cache + recompute policy                         population.py:831-842, 895-915
//   get_surface_albedo_bands + daily reduction              population.py:856-893, scripts/run_simulation.py:1843-1844
//   EcologyAdapter.step_subdaily (land-only alpha map)      pygcm/ecology/adapter.py:140-186
//   IndividualPool.try_substep                              pygcm/ecology/individuals.py:142-191
//
// The reference re-reduces the [S][K][lat][lon] LAI stack every physics step (np.sum(axis=(0,1)) inside
// _should_recompute_canopy) although the stack only changes in the daily step.  Here the plane sum runs when the host hands
// over new layers (once per planet-day), the canopy factor and the alpha map are rebuilt only when the reference's own
// policy says their inputs changed, and the steady-state cost per step is one read-modify-write of E_day riding on the
// forcing kernel plus one extra read in the albedo kernel.  The individuals never see a [NB][lat][lon] band stack: each
// thread evaluates the band split at its own sampled cell.  All of it is HBM-bound pointwise work.
#include "qd_internal.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

// ------------------------------------------------------------------ map storage
// qd_eco_params.map_f32 (QD_ECO_F32, BASELINE configs[4] "f32 mixed precision"): the canopy maps LAI_tot, its snapshot, the canopy
// factor f and the two alpha maps are STORED as f32 in their slabs (same allocation, first half used); every kernel loads them
// into f64, computes in f64 and rounds once on the store.  E_day, the plane sum of the LAI stack and the lai-delta reduction
// stay f64.  f32: wave-uniform flag, the branch is free.
__device__ __forceinline__ double qd_mld(const double* __restrict__ p, size_t o, int f32) {
    return f32 ? (double)reinterpret_cast<const float*>(p)[o] : p[o];
}
__device__ __forceinline__ void qd_mst(double* __restrict__ p, size_t o, int f32, double v) {
    if (f32) reinterpret_cast<float*>(p)[o] = (float)v; else p[o] = v;
}

// ------------------------------------------------------------------ kernels
__global__ void __launch_bounds__(QD_BLOCK)
k_eco_lai_accum(size_t n, const double* __restrict__ plane, double* __restrict__ lai, int first) {
    const size_t o = (size_t)blockIdx.x * QD_BLOCK + threadIdx.x;
    if (o >= n) return;
    lai[o] = first ? plane[o] : lai[o] + plane[o];            // np.sum(axis=(0,1)): plane after plane, s outer, k inner
}

// storage conversion of a whole slab (upload / download seam, end of the f64 plane sum); src and dst are different slabs
__global__ void __launch_bounds__(QD_BLOCK)
k_eco_convert(size_t n, const double* __restrict__ src, int src_f32, double* __restrict__ dst, int dst_f32) {
    const size_t o = (size_t)blockIdx.x * QD_BLOCK + threadIdx.x;
    if (o < n) qd_mst(dst, o, dst_f32, qd_mld(src, o, src_f32));
}

__global__ void __launch_bounds__(QD_BLOCK)
k_eco_copy(size_t n, const double* __restrict__ a, double* __restrict__ b) {
    const size_t o = (size_t)blockIdx.x * QD_BLOCK + threadIdx.x;
    if (o < n) b[o] = a[o];
}

__global__ void __launch_bounds__(QD_BLOCK)
k_eco_eday(QdGeom G, const double* __restrict__ isr, double dt, double* __restrict__ eday) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    eday[o] += qd_nn(isr[o]) * dt;                             // population.py:267-268
}

// f = 1 - exp(-k max(LAI_tot, 0)); snapshot <- LAI_tot          (population.py:911-915, 274-277)
__global__ void __launch_bounds__(QD_BLOCK)
k_eco_canopy(size_t n, const double* __restrict__ lai, double k, double* __restrict__ f, double* __restrict__ snap, int f32) {
    const size_t o = (size_t)blockIdx.x * QD_BLOCK + threadIdx.x;
    if (o >= n) return;
    const double L = qd_mld(lai, o, f32);
    qd_mst(f, o, f32, 1.0 - exp(-k * qd_max(L, 0.0)));
    qd_mst(snap, o, f32, L);
}

// alpha = clip(leaf_s f + (1 - f) soil, 0, 1) on land, NaN elsewhere          (adapter.py:160-174)
__global__ void __launch_bounds__(QD_BLOCK)
k_eco_alpha(size_t n, const double* __restrict__ f, const uint8_t* __restrict__ land, double leaf_s, double soil,
            double* __restrict__ alpha, int f32) {
    const size_t o = (size_t)blockIdx.x * QD_BLOCK + threadIdx.x;
    if (o >= n) return;
    if (!f) { qd_mst(alpha, o, f32, (land[o] == 1) ? qd_clip(leaf_s, 0.0, 1.0) : NAN); return; }      // M1 branch: no population (adapter.py:162-166)
    const double fv = qd_mld(f, o, f32);
    qd_mst(alpha, o, f32, (land[o] == 1) ? qd_clip(leaf_s * fv + (1.0 - fv) * soil, 0.0, 1.0) : NAN);
}

struct QdBandR { double r[QD_MAXBANDS], w[QD_MAXBANDS]; int nb; };
// clip(nansum_b A_b w_b, 0, 1) with A_b = clip(R_eff[b] f + (1 - f) soil, 0, 1) on land, NaN elsewhere (nansum of NaNs = 0)
__global__ void __launch_bounds__(QD_BLOCK)
k_eco_banded(size_t n, QdBandR W, const double* __restrict__ f, const uint8_t* __restrict__ land, double soil,
             double* __restrict__ out, int f32) {
    const size_t o = (size_t)blockIdx.x * QD_BLOCK + threadIdx.x;
    if (o >= n) return;
    double acc = 0.0;
    if (land[o] == 1) {
        const double fv = qd_mld(f, o, f32);
        for (int b = 0; b < W.nb; ++b) {
            const double t = qd_clip(W.r[b] * fv + (1.0 - fv) * soil, 0.0, 1.0) * W.w[b];
            const double tt = (t != t) ? 0.0 : t;              // nansum
            acc = (b == 0) ? tt : acc + tt;
        }
    }
    qd_mst(out, o, f32, qd_clip(acc, 0.0, 1.0));
}

__device__ __forceinline__ double qd_eco_wsum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}
// four row partials of lai_delta_ratio (population.py:903-907): sum |now - snap| and count over non-NaN, sum max(snap, 1e-6)
// and count over non-NaN
__global__ void __launch_bounds__(QD_BLOCK)
k_eco_ratio_rows(QdGeom G, const double* __restrict__ lai, const double* __restrict__ snap, double* __restrict__ partial, int f32) {
    __shared__ double sm[4][QD_BLOCK / 64];
    const size_t b = (size_t)qd_lrow(G, G.row0 + blockIdx.x) * G.nlon;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int j = threadIdx.x; j < G.nlon; j += QD_BLOCK) {
        const double s = qd_mld(snap, b + j, f32);
        const double d = fabs(qd_mld(lai, b + j, f32) - s);
        if (d == d) { a0 += d; a1 += 1.0; }
        const double m = qd_max(s, 1e-6);
        if (m == m) { a2 += m; a3 += 1.0; }
    }
    a0 = qd_eco_wsum(a0); a1 = qd_eco_wsum(a1); a2 = qd_eco_wsum(a2); a3 = qd_eco_wsum(a3);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sm[0][wv] = a0; sm[1][wv] = a1; sm[2][wv] = a2; sm[3][wv] = a3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double r = sm[threadIdx.x][0];
        for (int k = 1; k < QD_BLOCK / 64; ++k) r += sm[threadIdx.x][k];
        partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = r;
    }
}
__global__ void __launch_bounds__(QD_BLOCK)
k_eco_ratio_finish(const double* __restrict__ partial, int nrows, double* __restrict__ out) {
    __shared__ double sm[QD_BLOCK / 64];
    for (int q = 0; q < 4; ++q) {
        double a = 0.0;
        for (int k = threadIdx.x; k < nrows; k += QD_BLOCK) a += partial[(size_t)q * nrows + k];
        a = qd_eco_wsum(a);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) { double r = sm[0]; for (int k = 1; k < QD_BLOCK / 64; ++k) r += sm[k]; out[q] = r; }
    }
}

struct QdIndivW { double a[QD_MAXBANDS], b[QD_MAXBANDS], t[QD_MAXBANDS]; int nb; };
// individuals.py:164-191 for individual i: band split of its cell's two-star insolation (spectral.py:397-426), energy
// increment max(0, sum_b Ab[i][b] I_b * period), water-stress days where the soil index is below its tolerance
template <typename TA>
__global__ void __launch_bounds__(QD_BLOCK)
k_indiv_substep(QdGeom G, QdIndivW W, int n_indiv, const int32_t* __restrict__ cell, const int32_t* __restrict__ sj,
                const int32_t* __restrict__ si, const TA* __restrict__ Ab, const double* __restrict__ tol,
                const double* __restrict__ insA, const double* __restrict__ insB, const double* __restrict__ wland,
                double soil_cap_safe, double period, double stress_inc, double* __restrict__ E, double* __restrict__ stress) {
    const int i = blockIdx.x * QD_BLOCK + threadIdx.x;
    if (i >= n_indiv) return;
    const int c = cell[i];
    const int gj = sj[c], gi = si[c];
    if (gj < G.row0 || gj >= G.row0 + G.nrows) return;         // another band owns this cell
    const size_t o = (size_t)qd_lrow(G, gj) * G.nlon + gi;
    const double A = insA[o], B = insB[o];
    const double tot = A + B;
    double S[QD_MAXBANDS];
    double sum = 0.0;
#pragma unroll
    for (int b = 0; b < QD_MAXBANDS; ++b)
        if (b < W.nb) { S[b] = (W.a[b] * A + W.b[b] * B) * W.t[b]; sum += S[b]; }
    const bool pos = (sum > 1e-12) && (tot > 1e-12);
    double dE = 0.0;
#pragma unroll
    for (int b = 0; b < QD_MAXBANDS; ++b)
        if (b < W.nb) {
            double v = pos ? (S[b] / sum) * tot : 0.0;
            if (!(fabs(v) <= DBL_MAX)) v = 0.0;
            dE += (double)Ab[(size_t)b * n_indiv + i] * v;
        }
    E[i] += qd_max(0.0, dE * period);
    const double soil = qd_clip(wland[o] / soil_cap_safe, 0.0, 1.0);
    if (soil < tol[i]) stress[i] += stress_inc;
}

// ------------------------------------------------------------------ host side
static inline dim3 flat_grid(size_t n) { return dim3((unsigned)((n + QD_BLOCK - 1) / QD_BLOCK)); }

void qd_eco_free(qd_ctx* c) {
    QdEco& E = c->eco;
    void* p[] = {E.sample_j, E.sample_i, E.cell, E.Ab, E.Ab32, E.tol, E.E_day, E.stress};
    for (void* q : p) if (q) hipFree(q);
    E.sample_j = E.sample_i = E.cell = nullptr; E.Ab = E.tol = E.E_day = E.stress = nullptr; E.Ab32 = nullptr;
    E.n_indiv = E.n_cells = 0;
}

// fields whose slab holds f32 when map_f32 is set
bool qd_eco_is_f32(const qd_ctx* c, int field) {
    return c->eco.p.map_f32 && (field == QD_F_ECO_LAI || field == QD_F_ECO_LAI_SNAP || field == QD_F_ECO_F || field == QD_F_ECO_ALPHA ||
                                field == QD_F_ECO_ALPHA_BANDED);
}
void qd_eco_convert_slab(qd_ctx* c, const double* src, int src_f32, double* dst, int dst_f32) {
    const size_t n = c->geo.cells();
    hipLaunchKernelGGL(k_eco_convert, flat_grid(n), dim3(QD_BLOCK), 0, c->stream, n, src, src_f32, dst, dst_f32);
}

extern "C" int qd_eco_configure(qd_handle c, const qd_eco_params* p, size_t sz) {
    if (!c || !p) return -1;
    if (sz != sizeof(qd_eco_params)) return qd_fail(c, "qd_eco_configure: struct size mismatch (ABI)");
    QdEco& E = c->eco;
    const bool first = !E.configured;
    if (!first && (p->leaf_scalar != E.p.leaf_scalar || p->soil_ref != E.p.soil_ref || p->use_lai != E.p.use_lai)) E.alpha_dirty = 1;
    if (!first && (p->map_f32 != 0) != (E.p.map_f32 != 0) && (E.have_lai || E.alpha_valid || E.banded_valid))
        return qd_fail(c, "qd_eco_configure: map_f32 cannot change once the canopy maps hold data");
    E.p = *p;
    if (first) E.next_h = p->light_update_hours;               // population.py:72
    E.configured = 1;
    return 0;
}

extern "C" int qd_eco_set_lai_layers(qd_handle c, const double* layers, int n_planes, int init) {
    if (!c || !layers || n_planes < 1) return -1;
    hipSetDevice(c->desc.device);
    const size_t n = c->geo.cells(), plane = (size_t)c->geo.nlat * c->geo.nlon;
    double* stage = c->scratch[10];
    double* lai = c->f[QD_F_ECO_LAI];
    const int f32 = c->eco.p.map_f32 ? 1 : 0;
    double* acc = f32 ? c->scratch[11] : lai;                   // the plane sum is a reduction: f64 whatever the storage
    for (int k = 0; k < n_planes; ++k) {
        if (qd_band_copy_in(c, stage, layers + (size_t)k * plane, sizeof(double))) return qd_fail(c, "qd_eco_set_lai_layers: copy failed");
        hipLaunchKernelGGL(k_eco_lai_accum, flat_grid(n), dim3(QD_BLOCK), 0, c->stream, n, stage, acc, k == 0 ? 1 : 0);
        QD_HIP(c, hipStreamSynchronize(c->stream));             // the staging slab is reused; the host buffer is borrowed
    }
    if (f32) hipLaunchKernelGGL(k_eco_convert, flat_grid(n), dim3(QD_BLOCK), 0, c->stream, n, acc, 0, lai, 1);
    QdEco& E = c->eco;
    E.have_lai = 1; E.lai_version++;
    if (init) {
        hipLaunchKernelGGL(k_eco_copy, flat_grid(n), dim3(QD_BLOCK), 0, c->stream, n, lai, c->f[QD_F_ECO_LAI_SNAP]);
        E.snap_version = E.lai_version;
    }
    qd_mark(c, {lai, c->f[QD_F_ECO_LAI_SNAP]}, c->geo.halo);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qd_fail(c, "qd_eco_set_lai_layers: launch", e);
    return 0;
}

// lai_delta_ratio (population.py:903-907); synchronises -- only called on steps after the layers changed
static int eco_ratio(qd_ctx* c, double* ratio) {
    const QdGeom G = qd_segments(c, 0).g[0];
    hipLaunchKernelGGL(k_eco_ratio_rows, dim3(G.nrows), dim3(QD_BLOCK), 0, c->stream, G, c->f[QD_F_ECO_LAI], c->f[QD_F_ECO_LAI_SNAP],
                       c->red_partial, c->eco.p.map_f32 ? 1 : 0);
    hipLaunchKernelGGL(k_eco_ratio_finish, dim3(1), dim3(QD_BLOCK), 0, c->stream, c->red_partial, G.nrows, c->dscal + QD_S_DIAG0);
    if (qd_allreduce_f64(c, c->dscal + QD_S_DIAG0, 4, 0)) return -1;
    QD_HIP(c, hipMemcpyAsync(c->hpin, c->dscal + QD_S_DIAG0, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    QD_HIP(c, hipStreamSynchronize(c->stream));
    const double delta = c->hpin[1] > 0 ? c->hpin[0] / c->hpin[1] : NAN;       // nanmean of an all-NaN field is NaN
    const double base = c->hpin[3] > 0 ? c->hpin[2] / c->hpin[3] : NAN;
    *ratio = (base > 0) ? delta / base : delta;
    return 0;
}

int qd_eco_canopy_impl(qd_ctx* c, double dt) {
    QdEco& E = c->eco;
    if (!E.configured) return qd_fail(c, "ecology sub-step: qd_eco_configure has not been called");
    const size_t n = c->geo.cells();
    if (!E.p.use_lai) {                                         // adapter.py:146-166 without a population
        E.count++;
        if (E.count % std::max(1, (int)E.p.substep_every_nphys) == 0 && (E.alpha_dirty || !E.alpha_valid)) {
            hipLaunchKernelGGL(k_eco_alpha, flat_grid(n), dim3(QD_BLOCK), 0, c->stream, n, (const double*)nullptr, c->land,
                               E.p.leaf_scalar, E.p.soil_ref, c->f[QD_F_ECO_ALPHA], E.p.map_f32 ? 1 : 0);
            qd_mark(c, {c->f[QD_F_ECO_ALPHA]}, c->geo.halo);
            E.alpha_valid = 1; E.alpha_dirty = 0;
        }
        return 0;
    }
    if (!E.have_lai) return qd_fail(c, "ecology sub-step: no LAI layers (qd_eco_set_lai_layers)");
    QdScope sc(c, "eco_canopy");
    E.count++;
    E.hours += dt / 3600.0;                                     // population.py:271
    bool rec = !E.f_valid || E.hours >= E.next_h;               // population.py:897-901
    if (!rec) {
        double ratio = 0.0;                                     // untouched layers: |now - snapshot| is zero everywhere
        if (E.lai_version != E.snap_version && eco_ratio(c, &ratio)) return -1;
        rec = ratio >= E.p.recompute_lai_delta;
    }
    if (rec) {
        hipLaunchKernelGGL(k_eco_canopy, flat_grid(n), dim3(QD_BLOCK), 0, c->stream, n, c->f[QD_F_ECO_LAI], E.p.k_canopy,
                           c->f[QD_F_ECO_F], c->f[QD_F_ECO_LAI_SNAP], E.p.map_f32 ? 1 : 0);
        qd_mark(c, {c->f[QD_F_ECO_F], c->f[QD_F_ECO_LAI_SNAP]}, c->geo.halo);
        E.f_valid = 1; E.alpha_dirty = 1; E.snap_version = E.lai_version; E.n_recompute++;
        E.next_h = E.hours + E.p.light_update_hours;
    }
    const int every = std::max(1, (int)E.p.substep_every_nphys);
    if (E.count % every == 0 && (E.alpha_dirty || !E.alpha_valid)) {
        hipLaunchKernelGGL(k_eco_alpha, flat_grid(n), dim3(QD_BLOCK), 0, c->stream, n, c->f[QD_F_ECO_F], c->land, E.p.leaf_scalar,
                           E.p.soil_ref, c->f[QD_F_ECO_ALPHA], E.p.map_f32 ? 1 : 0);
        qd_mark(c, {c->f[QD_F_ECO_ALPHA]}, c->geo.halo);
        E.alpha_valid = 1; E.alpha_dirty = 0;
    }
    return 0;
}

int qd_eco_eday_impl(qd_ctx* c, double dt) {
    QdScope sc(c, "eco_eday");
    const int m = qd_plan(c, {QD_IN(c->f[QD_F_ISR], 0)});
    if (m < 0) return -1;
    QD_ROWS(c, 0, G, hipLaunchKernelGGL(k_eco_eday, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, c->f[QD_F_ISR], dt, c->f[QD_F_ECO_EDAY]));
    return 0;
}

extern "C" int qd_eco_substep(qd_handle c, double dt) {
    if (!c) return -1;
    hipSetDevice(c->desc.device);
    if (!c->eco.configured) return qd_fail(c, "qd_eco_substep: qd_eco_configure has not been called");
    if (c->eco.p.use_lai && qd_eco_eday_impl(c, dt)) return -1; // pop.step_subdaily first (adapter.py:151-155)
    if (qd_eco_canopy_impl(c, dt)) return -1;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qd_fail(c, "qd_eco_substep: launch", e);
    return 0;
}

extern "C" int qd_eco_banded_alpha(qd_handle c, int nb, const double* r_eff, const double* w_b) {
    if (!c || !r_eff || !w_b) return -1;
    hipSetDevice(c->desc.device);
    QdEco& E = c->eco;
    if (nb < 1 || nb > QD_MAXBANDS) return qd_fail(c, "qd_eco_banded_alpha: 1 <= nb <= 32");
    if (!E.configured || !E.have_lai) return qd_fail(c, "qd_eco_banded_alpha: ecology not configured / no LAI layers");
    const size_t n = c->geo.cells();
    if (!E.f_valid) {                                           // canopy_reflectance_factor builds the cache on demand (population.py:837-838)
        hipLaunchKernelGGL(k_eco_canopy, flat_grid(n), dim3(QD_BLOCK), 0, c->stream, n, c->f[QD_F_ECO_LAI], E.p.k_canopy,
                           c->f[QD_F_ECO_F], c->scratch[10], E.p.map_f32 ? 1 : 0);
        E.f_valid = 1; E.alpha_dirty = 1; E.n_recompute++;
    }
    QdBandR W; W.nb = nb;
    for (int b = 0; b < QD_MAXBANDS; ++b) { W.r[b] = b < nb ? r_eff[b] : 0.0; W.w[b] = b < nb ? w_b[b] : 0.0; }
    hipLaunchKernelGGL(k_eco_banded, flat_grid(n), dim3(QD_BLOCK), 0, c->stream, n, W, c->f[QD_F_ECO_F], c->land, E.p.soil_ref,
                       c->f[QD_F_ECO_ALPHA_BANDED], E.p.map_f32 ? 1 : 0);
    qd_mark(c, {c->f[QD_F_ECO_ALPHA_BANDED], c->f[QD_F_ECO_F]}, c->geo.halo);
    E.banded_valid = 1;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qd_fail(c, "qd_eco_banded_alpha: launch", e);
    return 0;
}

extern "C" int qd_eco_get_state(qd_handle c, double* out) {
    if (!c || !out) return -1;
    const QdEco& E = c->eco;
    out[0] = E.hours; out[1] = E.next_h; out[2] = (double)E.count; out[3] = (double)E.n_recompute; out[4] = E.alpha_valid ? 1.0 : 0.0;
    return 0;
}
extern "C" int qd_eco_set_state(qd_handle c, const double* in) {
    if (!c || !in) return -1;
    c->eco.hours = in[0]; c->eco.next_h = in[1]; c->eco.count = (int64_t)in[2];
    return 0;
}

// ------------------------------------------------------------------ individuals
template <typename T>
static int up(qd_ctx* c, T** dst, const T* src, size_t n) {
    if (hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return -1;
    if (src) return hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
    return hipMemset(*dst, 0, n * sizeof(T)) == hipSuccess ? 0 : -1;
}

extern "C" int qd_indiv_configure(qd_handle c, int n_cells, const int32_t* sample_j, const int32_t* sample_i, int n_indiv,
                                  const int32_t* cell_index, const double* Ab, const double* tol, int nb, const double* specA,
                                  const double* specB, const double* tray, int substeps_per_day, double day_seconds, double soil_cap,
                                  int ab_f32) {
    if (!c || !sample_j || !sample_i || !cell_index || !Ab || !tol || !specA || !specB || !tray) return -1;
    hipSetDevice(c->desc.device);
    if (nb < 1 || nb > QD_MAXBANDS) return qd_fail(c, "qd_indiv_configure: 1 <= nb <= 32");
    if (n_cells < 1 || n_indiv < 1 || !(day_seconds > 0)) return qd_fail(c, "qd_indiv_configure: empty pool or bad day length");
    // the kernel indexes the grid with these: reject anything outside it before it reaches the device
    for (int k = 0; k < n_cells; ++k)
        if (sample_j[k] < 0 || sample_j[k] >= c->geo.nlat || sample_i[k] < 0 || sample_i[k] >= c->geo.nlon)
            return qd_fail(c, "qd_indiv_configure: sampled cell outside the grid");
    for (int i = 0; i < n_indiv; ++i)
        if (cell_index[i] < 0 || cell_index[i] >= n_cells) return qd_fail(c, "qd_indiv_configure: cell index outside the sample");
    QD_HIP(c, hipStreamSynchronize(c->stream));
    qd_eco_free(c);
    QdEco& E = c->eco;
    std::vector<double> abt((size_t)nb * n_indiv);              // [n_indiv][nb] -> [nb][n_indiv]
    for (int i = 0; i < n_indiv; ++i)
        for (int b = 0; b < nb; ++b) abt[(size_t)b * n_indiv + i] = Ab[(size_t)i * nb + b];
    std::vector<float> abt32;
    if (ab_f32) { abt32.assign(abt.begin(), abt.end()); }       // storage only: the dot product still accumulates in f64
    if (up(c, &E.sample_j, sample_j, n_cells) || up(c, &E.sample_i, sample_i, n_cells) || up(c, &E.cell, cell_index, n_indiv) ||
        (ab_f32 ? up(c, &E.Ab32, abt32.data(), abt32.size()) : up(c, &E.Ab, abt.data(), abt.size())) || up(c, &E.tol, tol, n_indiv) || up<double>(c, &E.E_day, nullptr, n_indiv) ||
        up<double>(c, &E.stress, nullptr, n_indiv)) { qd_eco_free(c); return qd_fail(c, "qd_indiv_configure: device allocation / copy failed"); }
    E.n_cells = n_cells; E.n_indiv = n_indiv; E.nb = nb; E.k_per_day = std::max(1, substeps_per_day);
    for (int b = 0; b < QD_MAXBANDS; ++b) { E.specA[b] = b < nb ? specA[b] : 0.0; E.specB[b] = b < nb ? specB[b] : 0.0; E.tray[b] = b < nb ? tray[b] : 0.0; }
    E.day_seconds = day_seconds; E.soil_cap = soil_cap; E.period = -1.0; E.accum = 0.0; E.n_fired = 0;
    return 0;
}

int qd_indiv_substep_impl(qd_ctx* c, double dt, int* fired) {
    QdEco& E = c->eco;
    if (fired) *fired = 0;
    if (E.n_indiv <= 0) return qd_fail(c, "qd_indiv_substep: no pool configured (qd_indiv_configure)");
    if (E.period < 0) { E.period = E.day_seconds / (double)E.k_per_day; E.accum = 0.0; }      // individuals.py:155-157
    E.accum += dt;
    if (E.accum < E.period) return 0;
    E.accum -= E.period;
    QdScope sc(c, "eco_indiv");
    double** F = c->f;
    const int m = qd_plan(c, {QD_IN(F[QD_F_ISR_A], 0), QD_IN(F[QD_F_ISR_B], 0), QD_IN(F[QD_F_W_LAND], 0)});
    if (m < 0) return -1;
    QdIndivW W; W.nb = E.nb;
    for (int b = 0; b < QD_MAXBANDS; ++b) { W.a[b] = E.specA[b]; W.b[b] = E.specB[b]; W.t[b] = E.tray[b]; }
    const QdGeom G = qd_segments(c, 0).g[0];
    const dim3 grid((E.n_indiv + QD_BLOCK - 1) / QD_BLOCK);
    if (E.Ab32)
        hipLaunchKernelGGL(k_indiv_substep<float>, grid, dim3(QD_BLOCK), 0, c->stream, G, W, E.n_indiv, E.cell, E.sample_j, E.sample_i,
                           (const float*)E.Ab32, E.tol, F[QD_F_ISR_A], F[QD_F_ISR_B], F[QD_F_W_LAND], std::max(1e-6, E.soil_cap),
                           E.period, E.period / E.day_seconds, E.E_day, E.stress);
    else
        hipLaunchKernelGGL(k_indiv_substep<double>, grid, dim3(QD_BLOCK), 0, c->stream, G, W, E.n_indiv, E.cell, E.sample_j, E.sample_i,
                           (const double*)E.Ab, E.tol, F[QD_F_ISR_A], F[QD_F_ISR_B], F[QD_F_W_LAND], std::max(1e-6, E.soil_cap),
                           E.period, E.period / E.day_seconds, E.E_day, E.stress);
    E.n_fired++;
    if (fired) *fired = 1;
    return 0;
}

extern "C" int qd_indiv_substep(qd_handle c, double dt, int* fired) {
    if (!c) return -1;
    hipSetDevice(c->desc.device);
    if (qd_indiv_substep_impl(c, dt, fired)) return -1;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qd_fail(c, "qd_indiv_substep: launch", e);
    return 0;
}

extern "C" int qd_indiv_download(qd_handle c, double* E_day, double* stress) {
    if (!c) return -1;
    hipSetDevice(c->desc.device);
    QdEco& E = c->eco;
    if (E.n_indiv <= 0) return qd_fail(c, "qd_indiv_download: no pool configured");
    if (E_day) QD_HIP(c, hipMemcpyAsync(E_day, E.E_day, (size_t)E.n_indiv * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (stress) QD_HIP(c, hipMemcpyAsync(stress, E.stress, (size_t)E.n_indiv * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    QD_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int qd_indiv_upload(qd_handle c, const double* E_day, const double* stress) {
    if (!c) return -1;
    hipSetDevice(c->desc.device);
    QdEco& E = c->eco;
    if (E.n_indiv <= 0) return qd_fail(c, "qd_indiv_upload: no pool configured");
    if (E_day) QD_HIP(c, hipMemcpyAsync(E.E_day, E_day, (size_t)E.n_indiv * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (stress) QD_HIP(c, hipMemcpyAsync(E.stress, stress, (size_t)E.n_indiv * sizeof(double), hipMemcpyHostToDevice, c->stream));
    QD_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}
