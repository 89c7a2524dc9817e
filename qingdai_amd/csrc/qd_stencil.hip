// qd_stencil.hip -- lat-lon stencil / gather operators (gfx950).
//
//   O1 spherical Laplacian          dynamics.py:144-173, ocean.py:100-117, jax_compat.py:111-132
//   O2 del^4 hyperdiffusion         dynamics.py:175-212, ocean.py:119-152, jax_compat.py:135-187
//   O3 Shapiro 1-2-1                dynamics.py:215-231, ocean.py:154-164
//   O5 semi-Lagrangian gather       dynamics.py:90-118, ocean.py:166-194, run_simulation.py:1131-1158
//   O11 divergence / vorticity      grid.py:41-88
//   O12 Gaussian blur               physics.py:44,69,111,159,330; run_simulation.py:1931
//
// Launch shape: one workgroup row-segment per (blockIdx.x, row = blockIdx.y, field = blockIdx.z):
// lanes run along longitude so every load is a coalesced f64 row segment and the per-row
// metrics (cos floors, k4, pole one-sidedness) are wave-uniform scalars.
#include "qd_internal.h"
#include "qd_device.h"
#include "qd_wave.h"
#include <cstdlib>

// ------------------------------------------------------------------ Laplacian (device)
__global__ void __launch_bounds__(QD_BLOCK)
k_laplacian(QdGeom G, QdFieldList fl, const double* __restrict__ cosf, double dphi, double dlam, double a) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const int f = tl.fld;
    const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
    fl.out[f][o] = qd_lap_point<true>(fl.in[f], G, cosf, i, j, dphi, dlam, a);
}

// second Laplacian + explicit update: out = nan_to_num( nan_to_num(F) - k4 * L(L1) * sub_dt )
__global__ void __launch_bounds__(QD_BLOCK)
k_hyper_apply(QdGeom G, QdFieldList fl, const double* __restrict__ cosf, double dphi, double dlam, double a,
              double sub_dt) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const int f = tl.fld;
    const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
    const double L2 = qd_lap_point<true>(fl.in[f], G, cosf, i, j, dphi, dlam, a);
    const double k4 = fl.k4row[f] ? fl.k4row[f][i] : fl.k4s[f];
    const double F0 = qd_nn(fl.aux[f][o]);
    fl.out[f][o] = qd_nn(F0 - (k4 * L2) * sub_dt);
}

void qd_launch_laplacian(qd_ctx* c, const QdFieldList& fl, const double* coslat, int m) {
    QdScope sc(c, c->lap_tag);
    QD_ROWS(c, m, G, hipLaunchKernelGGL(k_laplacian, qd_grid2d(G, fl.n), dim3(QD_BLOCK), 0, c->stream,
                                        G, fl, coslat, c->dlat, c->dlon, c->p.a));
}
void qd_launch_hyper_apply(qd_ctx* c, const QdFieldList& fl, const double* coslat, double sub_dt, int m) {
    QdScope sc(c, c->hyp_tag);
    QD_ROWS(c, m, G, hipLaunchKernelGGL(k_hyper_apply, qd_grid2d(G, fl.n), dim3(QD_BLOCK), 0, c->stream,
                                        G, fl, coslat, c->dlat, c->dlon, c->p.a, sub_dt));
}

// lat reach (rows) of the semi-Lagrangian gather for winds up to vmax, + the bilinear neighbour
int qd_adv_reach(const qd_ctx* c, double dt, double vmax) {
    return (int)std::ceil(vmax * dt / (c->p.a * c->dlat)) + 1;
}
int qd_gauss_radius(double sigma) { return sigma > 1e-15 ? (int)(4.0 * sigma + 0.5) : 0; }

// _hyperdiffuse on up to QD_MAXF resident fields, in place (pointer swap with scratch).
// k4tab: [n][nlat] per-row maps; skip[f] != 0 -> field untouched (k4 <= 0 early-out);
// k4s_override[f] not NaN -> scalar coefficient instead of the row map.
int qd_hyperdiffuse_fields(qd_ctx* c, double** fields, int n, const double* k4tab, const int* skip,
                           const double* k4s_override, double dt, int nsub, const double* coslat, int m_out) {
    if (dt <= 0.0) return 0;
    QdFieldList a; a.n = 0;
    int idx[QD_MAXF];
    for (int f = 0; f < n; ++f) {
        if (skip && skip[f]) continue;
        idx[a.n] = f;
        a.in[a.n] = fields[f];
        a.out[a.n] = qd_scratch(c, a.n);                 // L1
        a.aux[a.n] = fields[f];
        bool sc = k4s_override && !(k4s_override[f] != k4s_override[f]);
        a.k4row[a.n] = sc ? nullptr : k4tab + (size_t)f * c->geo.nlat;
        a.k4s[a.n] = sc ? k4s_override[f] : 0.0;
        a.n++;
    }
    if (a.n == 0) return 0;
    const int ns = nsub < 1 ? 1 : nsub;
    const double sub_dt = dt / ns;
    for (int s = 0; s < ns; ++s) {
        const int m_apply = m_out + 4 * (ns - 1 - s);          // each sub-step eats 2 + 2 rows of margin
        qd_launch_laplacian(c, a, coslat, m_apply + 2);
        QdFieldList b = a;
        for (int k = 0; k < a.n; ++k) {
            b.in[k] = a.out[k];                           // L1
            b.out[k] = qd_scratch(c, QD_MAXF + k);        // new F
            b.aux[k] = a.in[k];
        }
        qd_launch_hyper_apply(c, b, coslat, sub_dt, m_apply);
        for (int k = 0; k < a.n; ++k) qd_mark(c, {b.out[k]}, m_apply);
        // new F becomes the field: copy pointer roles (swap field storage with scratch slot)
        for (int k = 0; k < a.n; ++k) {
            double* newF = b.out[k];
            double* oldF = fields[idx[k]];
            fields[idx[k]] = newF;
            c->scratch[QD_MAXF + k] = oldF;
            a.in[k] = newF; a.aux[k] = newF;
        }
    }
    return 0;
}

// ------------------------------------------------------------------ Shapiro
__global__ void __launch_bounds__(QD_BLOCK)
k_shapiro_pass(QdGeom G, QdFieldList fl, int scrub) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const int f = tl.fld;
    const double* __restrict__ F = fl.in[f];
    const int jm = qd_wrapc(j - 1, G.nlon), jp = qd_wrapc(j + 1, G.nlon);
    const int rm = i > 0 ? i - 1 : 0, rp = i < G.nlat - 1 ? i + 1 : G.nlat - 1;   // mode='nearest'
    double lc[3];
    const int rows[3] = {rm, i, rp};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const size_t b = (size_t)qd_lrow(G, rows[k]) * G.nlon;
        double xm = F[b + jm], x0 = F[b + j], xp = F[b + jp];
        if (scrub) { xm = qd_nn(xm); x0 = qd_nn(x0); xp = qd_nn(xp); }
        lc[k] = (xm * 0.25 + x0 * 0.5) + xp * 0.25;          // lon pass, mode='wrap'
    }
    fl.out[f][(size_t)qd_lrow(G, i) * G.nlon + j] = (lc[0] * 0.25 + lc[1] * 0.5) + lc[2] * 0.25;
}

void qd_launch_shapiro_pass(qd_ctx* c, const QdFieldList& fl, int scrub, int m) {
    QD_ROWS(c, m, G, hipLaunchKernelGGL(k_shapiro_pass, qd_grid2d(G, fl.n), dim3(QD_BLOCK), 0, c->stream, G, fl, scrub));
}

// All NP passes in ONE launch, rows streamed through registers.  One wave owns a strip of 64 - 2 NP columns x R rows of one field:
// lanes are columns (the longitude taps are DPP lane shifts; every pass spoils one more lane at either end), rows arrive in
// order and run through a cascade of NP stages, each holding the two previous longitude-filtered rows of its input.
//   stage p receives row r of pass p-1's output, l_r = (x[j-1]/4 + x[j]/2) + x[j+1]/4, and emits row r-1 = (l_{r-2}/4 + l_{r-1}/2) + l_r/4
// scipy's mode='nearest' at the poles (dynamics.py:228) is l_{-1} := l_0 and l_n := l_{n-1} of the SAME pass: a stage primes both
// history registers with the first row it sees, and gets one virtual row behind the north pole.  Same operations in the same
// order as NP launches of k_shapiro_pass, so the results are bit-identical; HBM traffic is one read and one write per field
// (+ 2 NP halo rows per strip) instead of NP of each.
#define QD_SH_PF 8                              // rows in flight ahead of the cascade
#define QD_SH_WAVES 1                            // waves per workgroup (measured: 4 is 5 % slower)
template <int NP> __global__ void __launch_bounds__(64 * QD_SH_WAVES)
k_shapiro_stream(QdGeom G, QdFieldList fl, int scrub, int R, int ntc, int nstrips) {
    constexpr int W = 64 - 2 * NP;
    const unsigned w = qd_xcd_chunk(blockIdx.x, gridDim.x) * QD_SH_WAVES + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= (unsigned)nstrips) return;
    const int rs = (int)(w / (unsigned)ntc), cs = (int)(w % (unsigned)ntc);
    const int lane = threadIdx.x & 63, n = G.nlat;
    const int jraw = cs * W - NP + lane;
    const int j = jraw < 0 ? jraw + G.nlon : (jraw >= G.nlon ? jraw - G.nlon : jraw);
    const bool own = lane >= NP && lane < 64 - NP && jraw < G.nlon;
    const unsigned vo = (unsigned)j * 8u, vs = own ? (unsigned)jraw * 8u : QD_BUF_OOB;
    const unsigned slab = (unsigned)(G.lrows_ + QD_PAD_ROWS) * (unsigned)G.nlon * 8u;
    const qd_rsrc src = qd_buf(fl.in[blockIdx.y], slab), dst = qd_buf(fl.out[blockIdx.y], slab);
    int a[NP + 1], b[NP + 1];                   // rows stage p has to produce (stage 0: rows to read)
    a[NP] = G.row0 + rs * R; b[NP] = min(a[NP] + R, G.row0 + G.nrows) - 1;
#pragma unroll
    for (int p = NP; p > 0; --p) { a[p - 1] = max(0, a[p] - 1); b[p - 1] = min(n - 1, b[p] + 1); }
    const int t0 = a[0], t1 = b[NP] + NP;       // the last owned row leaves the cascade NP ticks after it was read
    // the prefetch runs up to 15 rows past the strip: clamp into the domain AND into the slab (a band's slab ends lrows_ rows in;
    // the row offset sits in the SGPR offset of the buffer access, which the range check of the resource does not see)
    auto row_off = [&](int g) { return (unsigned)min(qd_lrow(G, min(g, n - 1)), G.lrows_ - 1) * (unsigned)G.nlon; };
    double pf[QD_SH_PF], h1[NP], h2[NP];
#pragma unroll
    for (int k = 0; k < QD_SH_PF; ++k) pf[k] = qd_buf_ld(src, row_off(t0 + k), vo);
#pragma unroll
    for (int p = 0; p < NP; ++p) h1[p] = h2[p] = 0.0;
    // ticks on which every stage takes a real row, emits a wanted row and is past its priming row: [t_lo, t_hi].  They run as
    // straight-line blocks of four (no branch inside: a wave-uniform branch in the unrolled body splits its basic block, and the
    // loads in flight are then drained by a full s_waitcnt at every block edge); the pipeline fill, the drain and the rows next to a
    // pole go through the general tick.
    int t_lo = t0, t_hi = t1;
#pragma unroll
    for (int p = 1; p <= NP; ++p) { t_lo = max(t_lo, a[p] + p); t_hi = min(t_hi, min(min(b[p] + 1, b[p - 1]), n - 1) + (p - 1)); }
    auto tick = [&](int t, double cur) {                        // the general tick
        if (t > t1) return;
        if (scrub) cur = qd_nn(cur);
        bool live = false;                                      // did the last stage emit a row on this tick
#pragma unroll
        for (int p = 1; p <= NP; ++p) {
            const int rp = t - (p - 1);                         // the row arriving at stage p on this tick
            live = false;
            if (rp < a[p - 1] || rp > b[p - 1] + (b[p - 1] == n - 1 ? 1 : 0)) continue;    // filling / drained
            const double l = rp <= n - 1 ? (qd_west(cur) * 0.25 + cur * 0.5) + qd_east(cur) * 0.25 : h1[p - 1];
            if (rp == a[p - 1]) { h1[p - 1] = l; h2[p - 1] = l; continue; }
            const double o = (h2[p - 1] * 0.25 + h1[p - 1] * 0.5) + l * 0.25;       // row rp - 1 of pass p
            h2[p - 1] = h1[p - 1]; h1[p - 1] = l;
            live = rp - 1 >= a[p] && rp - 1 <= b[p];
            if (live) cur = o;
        }
        if (live) qd_buf_st(dst, row_off(t - NP), vs, cur);
    };
    auto tick_all = [&](int t, double cur) {                    // every stage live, no conditions
        cur = scrub ? qd_nn(cur) : cur;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const double l = (qd_west(cur) * 0.25 + cur * 0.5) + qd_east(cur) * 0.25;
            const double o = (h2[p] * 0.25 + h1[p] * 0.5) + l * 0.25;
            h2[p] = h1[p]; h1[p] = l;
            cur = o;
        }
        qd_buf_st(dst, row_off(t - NP), vs, cur);
    };
    static_assert(QD_SH_PF == 8, "the tick loop is written as two half-rings of four");
    for (int tb = t0; tb <= t1; tb += QD_SH_PF) {
        if (tb >= t_lo && tb + 3 <= t_hi) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const double cur = pf[k]; pf[k] = qd_buf_ld(src, row_off(tb + k + QD_SH_PF), vo); tick_all(tb + k, cur); }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const double cur = pf[k]; pf[k] = qd_buf_ld(src, row_off(tb + k + QD_SH_PF), vo); tick(tb + k, cur); }
        }
        if (tb + 4 >= t_lo && tb + 7 <= t_hi) {
#pragma unroll
            for (int k = 4; k < 8; ++k) { const double cur = pf[k]; pf[k] = qd_buf_ld(src, row_off(tb + k + QD_SH_PF), vo); tick_all(tb + k, cur); }
        } else {
#pragma unroll
            for (int k = 4; k < 8; ++k) { const double cur = pf[k]; pf[k] = qd_buf_ld(src, row_off(tb + k + QD_SH_PF), vo); tick(tb + k, cur); }
        }
    }
}

// strip height of the one-launch Shapiro filter (QD_SHAPIRO_R: tuning override).  rocprofv3 kernel trace, u v h at 721 x 1440, two
// passes, R = 8 / 12 / 16 / 24 / 32: 14.4 / 14.1 / 14.4 / 14.9 / 15.2 us with the straight-line steady-state blocks (15.8 / 16.1 / 17.1 /
// 18.9 / 19.7 with the general tick everywhere; two k_shapiro_pass launches: 42 us); 4 instead of 8 rows in flight: +1 us.
static int qd_shapiro_rows(const qd_ctx* c) { return c->tune.shapiro_r > 0 ? c->tune.shapiro_r : 12; }

int qd_shapiro_fields(qd_ctx* c, double** fields, int n, int npass, int m_out) {
    if (npass < 1) npass = 1;
    QdFieldList a; a.n = n;
    if (npass <= 3 && c->geo.nlon >= 64 && c->shapiro_stream) {
        for (int k = 0; k < n; ++k) { a.in[k] = fields[k]; a.out[k] = qd_scratch(c, k); a.aux[k] = nullptr; a.k4row[k] = nullptr; a.k4s[k] = 0; }
        const int R = qd_shapiro_rows(c), W = 64 - 2 * npass, ntc = (c->geo.nlon + W - 1) / W;
        QD_ROWS(c, m_out, G, {
            const int ns = ((G.nrows + R - 1) / R) * ntc;
            const dim3 grid((ns + QD_SH_WAVES - 1) / QD_SH_WAVES, n), blk(64 * QD_SH_WAVES);
            if (npass == 1) hipLaunchKernelGGL(k_shapiro_stream<1>, grid, blk, 0, c->stream, G, a, 1, R, ntc, ns);
            else if (npass == 2) hipLaunchKernelGGL(k_shapiro_stream<2>, grid, blk, 0, c->stream, G, a, 1, R, ntc, ns);
            else hipLaunchKernelGGL(k_shapiro_stream<3>, grid, blk, 0, c->stream, G, a, 1, R, ntc, ns);
        });
        for (int k = 0; k < n; ++k) { double* t = fields[k]; fields[k] = c->scratch[k]; c->scratch[k] = t; qd_mark(c, {fields[k]}, m_out); }
        return 0;
    }
    for (int p = 0; p < npass; ++p) {
        for (int k = 0; k < n; ++k) { a.in[k] = fields[k]; a.out[k] = qd_scratch(c, k); a.aux[k] = nullptr; a.k4row[k] = nullptr; a.k4s[k] = 0; }
        const int m = m_out + (npass - 1 - p);
        qd_launch_shapiro_pass(c, a, p == 0, m);
        for (int k = 0; k < n; ++k) { double* t = fields[k]; fields[k] = c->scratch[k]; c->scratch[k] = t; qd_mark(c, {fields[k]}, m); }
    }
    return 0;
}

// ------------------------------------------------------------------ semi-Lagrangian gather
// out = (1-alpha)*F + alpha*advect(F)   (alpha == 1: plain replace); second field optional;
// clipq: second field gets clip(nan_to_num(.), 0, 0.5) (dynamics.py:461)
__global__ void __launch_bounds__(QD_BLOCK)
k_advect(QdGeom G, const double* __restrict__ u, const double* __restrict__ v, const double* __restrict__ cosl,
         double dt, double a, double dlat, double dlon,
         const double* __restrict__ f0, double* __restrict__ o0,
         const double* __restrict__ f1, double* __restrict__ o1, double alpha, int clipq) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
    const QdBilin b = qd_departure(G, i, j, u[o], v[o], dt, a, cosl[i], dlat, dlon);
    const bool blend = (alpha != 1.0);
    {
        const double adv = qd_gather(f0, G, b);
        o0[o] = blend ? (1.0 - alpha) * f0[o] + alpha * adv : adv;
    }
    if (f1) {
        const double adv = qd_gather(f1, G, b);
        double x = blend ? (1.0 - alpha) * f1[o] + alpha * adv : adv;
        if (clipq) x = qd_clip(qd_nn(x), 0.0, 0.5);
        o1[o] = x;
    }
}

void qd_launch_advect(qd_ctx* c, const double* u, const double* v, const double* coslat, double dt,
                      const double* f0, double* o0, const double* f1, double* o1, double alpha, int clipq, int m) {
    QD_ROWS(c, m, G, hipLaunchKernelGGL(k_advect, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, u, v, coslat, dt,
                                        c->p.a, c->dlat, c->dlon, f0, o0, f1, o1, alpha, clipq));
    qd_mark(c, {o0, o1}, m);
}

// ------------------------------------------------------------------ divergence / vorticity
// div  = 1/(a cos6) * [ d(u)/dlam + d(v cos)/dphi ]        grid.py:41-68
// vort = 1/(a cos6) * [ d(v)/dlam - d(u cos)/dphi ]        grid.py:70-88
// both axes roll-periodic, lat derivative zeroed on the two pole rows
__global__ void __launch_bounds__(QD_BLOCK)
k_divvort(QdGeom G, QdTabs T, const double* __restrict__ u, const double* __restrict__ v, double* __restrict__ out,
          double a, double dlat, double dlon, int vort) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    out[(size_t)qd_lrow(G, i) * G.nlon + j] =
        vort ? qd_divvort_point(G, T, v, u, i, j, a, dlat, dlon, 1) : qd_divvort_point(G, T, u, v, i, j, a, dlat, dlon, 0);
}

void qd_launch_divvort(qd_ctx* c, const double* u, const double* v, double* out, int vort, int m) {
    QD_ROWS(c, m, G, hipLaunchKernelGGL(k_divvort, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, c->tabs, u, v, out,
                                        c->p.a, c->dlat, c->dlon, vort));
    qd_mark(c, {out}, m);
}

// ------------------------------------------------------------------ Gaussian blur (separable)
#define QD_GAUSS_MAXR 24
struct QdGaussW { double w[QD_GAUSS_MAXR + 1]; int r; };   // w[k] = weight at offset k (symmetric)

__device__ __forceinline__ int qd_ext(int i, int n, int mode_wrap) {
    if (mode_wrap) { i %= n; return i < 0 ? i + n : i; }
    // 'reflect': d c b a | a b c d | d c b a
    const int p = 2 * n;
    i %= p; if (i < 0) i += p;
    return i >= n ? p - 1 - i : i;
}

// scipy correlate1d symmetric branch: tmp = x0*w0; for k = r..1: tmp += (x[-k] + x[+k]) * w[k]
__global__ void __launch_bounds__(QD_BLOCK)
k_gauss_axis(QdGeom G, const double* __restrict__ in, double* __restrict__ out, QdGaussW W, int axis, int mode_wrap) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int i = G.row0 + tl.row;
    const size_t o = (size_t)qd_lrow(G, i) * G.nlon + j;
    double tmp = in[o] * W.w[0];
    if (axis == 0) {
        for (int k = W.r; k >= 1; --k) {
            const double lo = in[(size_t)qd_lrow(G, qd_ext(i - k, G.nlat, mode_wrap)) * G.nlon + j];
            const double hi = in[(size_t)qd_lrow(G, qd_ext(i + k, G.nlat, mode_wrap)) * G.nlon + j];
            tmp += (lo + hi) * W.w[k];
        }
    } else {
        const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
        for (int k = W.r; k >= 1; --k) {
            const double lo = in[b + qd_ext(j - k, G.nlon, mode_wrap)];
            const double hi = in[b + qd_ext(j + k, G.nlon, mode_wrap)];
            tmp += (lo + hi) * W.w[k];
        }
    }
    out[o] = tmp;
}

__global__ void __launch_bounds__(QD_BLOCK) k_clip01_field(QdGeom G, double* __restrict__ x) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    x[o] = qd_clip(x[o], 0.0, 1.0);
}

// both axes in one launch: the axis-0 result of a row segment (+- r halo columns, recomputed, never exchanged)
// goes to LDS and the axis-1 pass reads it from there -- same arithmetic in the same order as the two-kernel form,
// without the round trip of the intermediate field through memory.  out must not alias in.
__global__ void __launch_bounds__(QD_BLOCK)
k_gauss_fused(QdGeom G, const double* __restrict__ in, double* __restrict__ out, QdGaussW W, int mode_wrap, int clip01,
              const double* __restrict__ scale_p, double scale_k) {
    __shared__ double sm[QD_BLOCK + 2 * QD_GAUSS_MAXR];
    const QdTile tl = qd_tile();
    const int i = G.row0 + tl.row;
    const int jbase = tl.seg * QD_BLOCK;
    const int r = W.r;
    for (int s = threadIdx.x; s < QD_BLOCK + 2 * r; s += QD_BLOCK) {
        const int jj = jbase - r + s;
        if (jj >= G.nlon + r) break;
        const int j = qd_ext(jj, G.nlon, mode_wrap);
        // optional pre-scaling of the input (the caller's `field * s` pass folded in; one rounding per element, as there)
        const double sc = scale_p ? *scale_p : scale_k;
        double tmp = (in[(size_t)qd_lrow(G, i) * G.nlon + j] * sc) * W.w[0];
        for (int k = r; k >= 1; --k) {
            const double lo = in[(size_t)qd_lrow(G, qd_ext(i - k, G.nlat, mode_wrap)) * G.nlon + j] * sc;
            const double hi = in[(size_t)qd_lrow(G, qd_ext(i + k, G.nlat, mode_wrap)) * G.nlon + j] * sc;
            tmp += (lo + hi) * W.w[k];
        }
        sm[s] = tmp;
    }
    __syncthreads();
    const int j = jbase + threadIdx.x;
    if (j >= G.nlon) return;
    const int c0 = threadIdx.x + r;
    double acc = sm[c0] * W.w[0];
    for (int k = r; k >= 1; --k) acc += (sm[c0 - k] + sm[c0 + k]) * W.w[k];
    if (clip01) acc = qd_clip(acc, 0.0, 1.0);              // np.clip(blurred, 0, 1) of the caller, folded in
    out[(size_t)qd_lrow(G, i) * G.nlon + j] = acc;
}

// Row-blocked form of the same blur for small compile-time radii: a workgroup owns QD_GB consecutive rows of a
// 256-column segment; a thread loads its column's QD_GB + 2R input rows ONCE into registers (instead of 2R+1 rows
// per output row), forms the axis-0 result of all QD_GB rows from them, and the axis-1 pass reads those from LDS.
// Identical arithmetic and tap order to k_gauss_fused / the two-kernel form.
#define QD_GB 8
// A workgroup owns QD_BLOCK - 2 RR columns and loads QD_BLOCK (its own + RR halo columns per side): one column per thread, one
// pass -- with QD_BLOCK owned columns the 2 RR halo columns were a second, serialised pass of a few threads (10.3 us per launch).
template <int RR>
__global__ void __launch_bounds__(QD_BLOCK)
k_gauss_rows(QdGeom G, const double* __restrict__ in, double* __restrict__ out, QdGaussW W, int mode_wrap, int clip01,
             const double* __restrict__ scale_p, double scale_k) {
    constexpr int OWN = QD_BLOCK - 2 * RR;
    __shared__ double sm[QD_GB][QD_BLOCK];
    const int i0 = G.row0 + (int)blockIdx.y * QD_GB;
    const int nvalid = min(QD_GB, G.row0 + G.nrows - i0);                  // rows of this block inside the launch
    const int jbase = (int)blockIdx.x * OWN;
    const double sc = scale_p ? *scale_p : scale_k;
    double w[RR + 1];
#pragma unroll
    for (int k = 0; k <= RR; ++k) w[k] = W.w[k];
    {
        const int s = threadIdx.x;
        const int jj = jbase - RR + s;
        if (jj < G.nlon + RR) {
            const int j = qd_ext(jj, G.nlon, mode_wrap);
            double x[QD_GB + 2 * RR];
#pragma unroll
            for (int q = 0; q < QD_GB + 2 * RR; ++q) {
                const int qq = q < nvalid + 2 * RR ? q : nvalid + 2 * RR - 1;   // never beyond the rows the valid outputs need
                x[q] = in[(size_t)qd_lrow(G, qd_ext(i0 - RR + qq, G.nlat, mode_wrap)) * G.nlon + j] * sc;
            }
#pragma unroll
            for (int k = 0; k < QD_GB; ++k) {
                double tmp = x[k + RR] * w[0];
#pragma unroll
                for (int q = RR; q >= 1; --q) tmp += (x[k + RR - q] + x[k + RR + q]) * w[q];
                sm[k][s] = tmp;
            }
        }
    }
    __syncthreads();
    const int c0 = threadIdx.x;
    const int j = jbase + c0 - RR;
    if (c0 < RR || c0 >= QD_BLOCK - RR || j >= G.nlon) return;
#pragma unroll
    for (int k = 0; k < QD_GB; ++k) {
        if (k >= nvalid) break;
        double acc = sm[k][c0] * w[0];
#pragma unroll
        for (int q = RR; q >= 1; --q) acc += (sm[k][c0 - q] + sm[k][c0 + q]) * w[q];
        if (clip01) acc = qd_clip(acc, 0.0, 1.0);
        out[(size_t)qd_lrow(G, i0 + k) * G.nlon + j] = acc;
    }
}

template <int RR>
static void qd_launch_gauss_rows(qd_ctx* c, const double* in, double* out, const QdGaussW& W, int mode_wrap, int clip01,
                                 const double* scale_p, double scale_k, int m_out) {
    constexpr int OWN = QD_BLOCK - 2 * RR;
    QD_ROWS(c, m_out, G, hipLaunchKernelGGL(k_gauss_rows<RR>, dim3((G.nlon + OWN - 1) / OWN, (G.nrows + QD_GB - 1) / QD_GB),
                                            dim3(QD_BLOCK), 0, c->stream, G, in, out, W, mode_wrap, clip01, scale_p, scale_k));
}

// scipy _gaussian_kernel1d weights (radius int(4 sigma + 0.5)); false: radius beyond QD_GAUSS_MAXR
static bool qd_gauss_weights(double sigma, QdGaussW& W) {
    const int r = (int)(4.0 * sigma + 0.5);
    if (r > QD_GAUSS_MAXR) return false;
    W.r = r;
    // scipy _gaussian_kernel1d: exp(-0.5/sigma^2 * x^2) / sum, summed in array order -r..r
    std::vector<double> phi(2 * r + 1);
    const double s2 = sigma * sigma;
    for (int x = -r; x <= r; ++x) phi[x + r] = std::exp(-0.5 / s2 * (double)(x * x));
    // numpy pairwise sum degenerates to a plain left-to-right loop for < 8 elements and an
    // 8-accumulator form above; mirror both.
    double tot;
    const int m = 2 * r + 1;
    if (m < 8) { tot = 0.0; for (int k = 0; k < m; ++k) tot += phi[k]; }
    else {
        double acc[8];
        for (int k = 0; k < 8; ++k) acc[k] = phi[k];
        int k = 8;
        for (; k < m - (m % 8); k += 8) for (int t = 0; t < 8; ++t) acc[t] += phi[k + t];
        tot = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        for (; k < m; ++k) tot += phi[k];
    }
    for (int k = 0; k <= r; ++k) W.w[k] = phi[r + k] / tot;
    return true;
}

// Two blurs of the same sigma and the pointwise blend of their results in ONE launch (whole-globe handles inside the driver
// physics: precipitation = blend(blur(s P_raw), blur(k pos)); cloud = blend(cloud, clip(blur(C_from_P)), clip(blur(source)))).
// A workgroup runs k_gauss_rows' two passes for field A, keeps its eight results per thread in registers, does the same for
// field B through the same LDS rows, and blends -- same taps in the same order, two launches and three field round trips fewer.
//   op 1 (precipitation, k_precip_blend): out0 = max((al != 0 ? (1 - al) a + al b : a), 0), al = sc[1]; field B is not even read
//                                         when al == 0 (the fallback blend of physics.py:344-352 is off)
//   op 2 (cloud, k_cloud_blend):          outA = a, outB = b (both clipped to [0, 1]), out0 = the blended cloud cover
struct QdGaussPairP { int op; double w_mem, w_p, w_src, tend, c_floor; };
template <int RR>
__device__ __forceinline__ void qd_gauss_rows_pass(const QdGeom& G, const double* __restrict__ in, double sc, const double (&w)[RR + 1],
                                                   int mode_wrap, int i0, int nvalid, int jbase, double (*sm)[QD_BLOCK], double (&acc)[QD_GB]) {
    {
        const int s = threadIdx.x;
        const int jj = jbase - RR + s;
        if (jj < G.nlon + RR) {
            const int j = qd_ext(jj, G.nlon, mode_wrap);
            double x[QD_GB + 2 * RR];
#pragma unroll
            for (int q = 0; q < QD_GB + 2 * RR; ++q) {
                const int qq = q < nvalid + 2 * RR ? q : nvalid + 2 * RR - 1;
                x[q] = in[(size_t)qd_lrow(G, qd_ext(i0 - RR + qq, G.nlat, mode_wrap)) * G.nlon + j] * sc;
            }
#pragma unroll
            for (int k = 0; k < QD_GB; ++k) {
                double tmp = x[k + RR] * w[0];
#pragma unroll
                for (int q = RR; q >= 1; --q) tmp += (x[k + RR - q] + x[k + RR + q]) * w[q];
                sm[k][s] = tmp;
            }
        }
    }
    __syncthreads();
    const int c0 = threadIdx.x;
    if (c0 >= RR && c0 < QD_BLOCK - RR) {
#pragma unroll
        for (int k = 0; k < QD_GB; ++k) {
            double a = sm[k][c0] * w[0];
#pragma unroll
            for (int q = RR; q >= 1; --q) a += (sm[k][c0 - q] + sm[k][c0 + q]) * w[q];
            acc[k] = a;
        }
    }
    __syncthreads();
}

template <int RR>
__global__ void __launch_bounds__(QD_BLOCK)
k_gauss_pair(QdGeom G, const double* __restrict__ inA, const double* __restrict__ inB, QdGaussW W, int mode_wrap,
             const double* __restrict__ scA_p, double scA_k, double scB_k, const double* __restrict__ sc, QdGaussPairP P,
             double* __restrict__ outA, double* __restrict__ outB, double* __restrict__ out0) {
    constexpr int OWN = QD_BLOCK - 2 * RR;
    __shared__ double sm[QD_GB][QD_BLOCK];
    const int i0 = G.row0 + (int)blockIdx.y * QD_GB;
    const int nvalid = min(QD_GB, G.row0 + G.nrows - i0);
    const int jbase = (int)blockIdx.x * OWN;
    double w[RR + 1];
#pragma unroll
    for (int k = 0; k <= RR; ++k) w[k] = W.w[k];
    const double al = (P.op == 1) ? sc[1] : 0.0;
    double a[QD_GB], b[QD_GB];
#pragma unroll
    for (int k = 0; k < QD_GB; ++k) { a[k] = 0.0; b[k] = 0.0; }
    qd_gauss_rows_pass<RR>(G, inA, scA_p ? *scA_p : scA_k, w, mode_wrap, i0, nvalid, jbase, sm, a);
    if (P.op != 1 || al != 0.0) qd_gauss_rows_pass<RR>(G, inB, scB_k, w, mode_wrap, i0, nvalid, jbase, sm, b);
    const int c0 = threadIdx.x;
    const int j = jbase + c0 - RR;
    if (c0 < RR || c0 >= QD_BLOCK - RR || j >= G.nlon) return;
#pragma unroll
    for (int k = 0; k < QD_GB; ++k) {
        if (k >= nvalid) break;
        const size_t o = (size_t)qd_lrow(G, i0 + k) * G.nlon + j;
        if (P.op == 1) {
            double p = a[k];
            if (al != 0.0) p = (1.0 - al) * p + al * b[k];
            out0[o] = qd_max(p, 0.0);                                      // np.clip(P, 0, None)
        } else {
            const double cp = qd_clip(a[k], 0.0, 1.0), sr = qd_clip(b[k], 0.0, 1.0);    // np.clip(gaussian(...), 0, 1)
            outA[o] = cp; outB[o] = sr;
            const double c0v = out0[o];
            const double tendency = sr * P.tend;
            double c = (P.w_mem * c0v + P.w_p * cp + P.w_src * qd_clip(c0v + tendency, 0.0, 1.0));
            if (P.c_floor > 0.0) c = qd_max(c, qd_clip(P.c_floor * cp, 0.0, 1.0));
            out0[o] = qd_clip(c, 0.0, 1.0);
        }
    }
}

// the caller checks qd_gauss_pair_ok and (band handles) plans both inputs with the blur's radius; m_out: margin of the outputs
bool qd_gauss_pair_ok(const qd_ctx* c, double sigma) {
    const int r = qd_gauss_radius(sigma);
    return c->use_fused && c->merge_pointwise && sigma > 1e-15 && c->geo.nlon > 2 * r && (r == 1 || r == 2 || r == 4);
}
int qd_gaussian_pair(qd_ctx* c, const double* inA, const double* inB, double sigma, int mode_wrap, const double* scA_p, double scA_k,
                     double scB_k, const double* sc, int op, const double* blend5, double* outA, double* outB, double* out0, int m_out) {
    QdGaussW W;
    if (!qd_gauss_weights(sigma, W)) return qd_fail(c, "gaussian radius too large");
    QdGaussPairP P{op, 0, 0, 0, 0, 0};
    if (blend5) { P.w_mem = blend5[0]; P.w_p = blend5[1]; P.w_src = blend5[2]; P.tend = blend5[3]; P.c_floor = blend5[4]; }
    const int r = W.r;
    QD_ROWS(c, m_out, G,
        const dim3 grid((G.nlon + (QD_BLOCK - 2 * r) - 1) / (QD_BLOCK - 2 * r), (G.nrows + QD_GB - 1) / QD_GB);
        if (r == 1) hipLaunchKernelGGL(k_gauss_pair<1>, grid, dim3(QD_BLOCK), 0, c->stream, G, inA, inB, W, mode_wrap, scA_p, scA_k, scB_k, sc, P, outA, outB, out0);
        else if (r == 2) hipLaunchKernelGGL(k_gauss_pair<2>, grid, dim3(QD_BLOCK), 0, c->stream, G, inA, inB, W, mode_wrap, scA_p, scA_k, scB_k, sc, P, outA, outB, out0);
        else hipLaunchKernelGGL(k_gauss_pair<4>, grid, dim3(QD_BLOCK), 0, c->stream, G, inA, inB, W, mode_wrap, scA_p, scA_k, scB_k, sc, P, outA, outB, out0));
    return 0;
}

// gaussian_filter(in, sigma, mode): axis 0 then axis 1.  out may alias in; tmp is a distinct slab.
int qd_gaussian(qd_ctx* c, const double* in, double* out, double* tmp, double sigma, int mode_wrap, int m_out, int clip01,
                const double* scale_p, double scale_k) {
    const bool scaled = scale_p != nullptr || scale_k != 1.0;
    if (scaled && !(c->use_fused && out != in && sigma > 1e-15 && c->geo.nlon > 2 * qd_gauss_radius(sigma)))
        return qd_fail(c, "qd_gaussian: input scaling needs the fused blur");
    if (!(sigma > 1e-15)) {
        if (out != in) hipMemcpyAsync(out, in, c->geo.cells() * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
        qd_mark(c, {out}, m_out);
        return 0;
    }
    QdGaussW W;
    if (!qd_gauss_weights(sigma, W)) return qd_fail(c, "gaussian radius too large");
    const int r = W.r;
    // axis 0 reaches r rows; axis 1 is row-local.  Both passes run on the output margin.
    if (c->use_fused && out != in && c->geo.nlon > 2 * r && (r == 1 || r == 2 || r == 4)) {
        if (r == 1) qd_launch_gauss_rows<1>(c, in, out, W, mode_wrap, clip01, scale_p, scale_k, m_out);
        else if (r == 2) qd_launch_gauss_rows<2>(c, in, out, W, mode_wrap, clip01, scale_p, scale_k, m_out);
        else qd_launch_gauss_rows<4>(c, in, out, W, mode_wrap, clip01, scale_p, scale_k, m_out);
        qd_mark(c, {out}, m_out);
        return 0;
    }
    if (c->use_fused && out != in && c->geo.nlon > 2 * r) {
        QD_ROWS(c, m_out, G, hipLaunchKernelGGL(k_gauss_fused, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, in, out, W, mode_wrap, clip01, scale_p, scale_k));
        qd_mark(c, {out}, m_out);
        return 0;
    }
    QD_ROWS(c, m_out, G, hipLaunchKernelGGL(k_gauss_axis, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, in, tmp, W, 0, mode_wrap));
    QD_ROWS(c, m_out, G, hipLaunchKernelGGL(k_gauss_axis, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, tmp, out, W, 1, mode_wrap));
    qd_mark(c, {tmp, out}, m_out);
    if (clip01) {
        QD_ROWS(c, m_out, G, hipLaunchKernelGGL(k_clip01_field, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, out));
    }
    return 0;
}

bool qd_gauss_can_fuse(const qd_ctx* c, double sigma) { return c->use_fused && sigma > 1e-15 && c->geo.nlon > 2 * qd_gauss_radius(sigma); }

// in-place form for fields held in context slots: blur `field` into `tmp`, then exchange the two slots
int qd_gaussian_swap(qd_ctx* c, double*& field, double*& tmp, double sigma, int mode_wrap, int m_out, int clip01,
                     const double* scale_p, double scale_k) {
    if (!(sigma > 1e-15)) {
        if (clip01) QD_ROWS(c, m_out, G, hipLaunchKernelGGL(k_clip01_field, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, field));
        return 0;
    }
    if (c->use_fused && c->geo.nlon > 2 * qd_gauss_radius(sigma)) {
        if (qd_gaussian(c, field, tmp, nullptr, sigma, mode_wrap, m_out, clip01, scale_p, scale_k)) return -1;
        std::swap(field, tmp);
        return 0;
    }
    if (scale_p != nullptr || scale_k != 1.0) return qd_fail(c, "qd_gaussian_swap: input scaling needs the fused blur");
    return qd_gaussian(c, field, field, tmp, sigma, mode_wrap, m_out, clip01);
}


__global__ void __launch_bounds__(QD_BLOCK) k_scrub_field(QdGeom G, double* __restrict__ x) {
    const QdTile tl = qd_tile();
    const int j = tl.seg * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const size_t o = (size_t)qd_lrow(G, G.row0 + tl.row) * G.nlon + j;
    x[o] = qd_nn(x[o]);
}

// ------------------------------------------------------------------ zonal spectral filter (O4): dynamics.py:233-258
// F' = nan_to_num(irfft(rfft(nan_to_num(F)) * factor)), factor = s = max(0, 1 - min(1, damp)) for bins >= kcut, else 1.
// The filter is linear and row-local:  F' = F - (1 - s) * high,  high = irfft(X restricted to bins kcut..kN).
// Only the damped bins (a quarter of the spectrum at the default cutoff 0.75) are transformed, with a direct DFT
// per row: one workgroup per (row, field), the row and the twiddle table cos/sin(2 pi m / n) staged in LDS,
// phase 1 one damped bin per thread, phase 2 one output column per thread.  Differs from pocketfft only by
// summation order (rounding, ~1e-15 of the row's max-norm).
__global__ void __launch_bounds__(QD_BLOCK)
k_zonal_filter(QdGeom G, QdFieldList fl, const double* __restrict__ tw, int kcut, int kN, double one_minus_s) {
    extern __shared__ double zs[];
    const int n = G.nlon, nb = kN - kcut + 1;
    double* x = zs;                 // [n]   nan_to_num(row)
    double* tc = x + n;             // [n]   cos(2 pi m / n)
    double* ts = tc + n;            // [n]   sin(2 pi m / n)
    double* Xr = ts + n;            // [nb]
    double* Xi = Xr + nb;           // [nb]
    const int i = G.row0 + blockIdx.y, f = blockIdx.z;
    double* Frow = fl.out[f] + (size_t)qd_lrow(G, i) * n;
    for (int j = threadIdx.x; j < n; j += QD_BLOCK) { x[j] = qd_nn(Frow[j]); tc[j] = tw[j]; ts[j] = tw[n + j]; }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += QD_BLOCK) {
        const int k = kcut + b;
        double ar = 0.0, ai = 0.0;
        int idx = 0;
        for (int j = 0; j < n; ++j) {
            const double v = x[j];
            ar += v * tc[idx];
            ai -= v * ts[idx];
            idx += k; if (idx >= n) idx -= n;
        }
        Xr[b] = ar; Xi[b] = ai;
    }
    __syncthreads();
    const bool even = (n & 1) == 0;
    const double inv_n = 1.0 / (double)n;
    for (int j = threadIdx.x; j < n; j += QD_BLOCK) {
        double hp = 0.0;
        int idx = (int)(((long long)j * kcut) % n);
        for (int b = 0; b < nb; ++b) {
            const bool nyq = even && (kcut + b == kN);            // irfft uses only the real part of the Nyquist bin, weight 1
            const double t = Xr[b] * tc[idx] - (nyq ? 0.0 : Xi[b] * ts[idx]);
            hp += nyq ? t : 2.0 * t;
            idx += j; if (idx >= n) idx -= n;
        }
        Frow[j] = qd_nn(x[j] - one_minus_s * (hp * inv_n));
    }
}

int qd_zonal_filter_fields(qd_ctx* c, double** fields, int nf, double cutoff, double damp, int m) {
    const int n = c->geo.nlon;
    if (!(damp > 0.0) || !(cutoff > 0.0) || n / 2 + 1 <= 1) {
        // the reference still scrubs: nan_to_num(F)
        for (int f = 0; f < nf; ++f)
            QD_ROWS(c, m, G, hipLaunchKernelGGL(k_scrub_field, qd_grid2d(G), dim3(QD_BLOCK), 0, c->stream, G, fields[f]));
        return 0;
    }
    const int kN = n / 2;                                         // rfft bins - 1
    const int kcut = std::max(1, std::min(kN, (int)(cutoff * (double)kN)));
    const double s = std::max(0.0, 1.0 - std::min(1.0, damp));
    if (!c->zonal_tw) {
        std::vector<double> tw(2 * (size_t)n);
        for (int k = 0; k < n; ++k) { const double a = 2.0 * M_PI * (double)k / (double)n; tw[k] = std::cos(a); tw[n + k] = std::sin(a); }
        if (hipMalloc(&c->zonal_tw, tw.size() * sizeof(double)) != hipSuccess) return qd_fail(c, "hipMalloc zonal twiddles");
        QD_HIP(c, hipMemcpy(c->zonal_tw, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const int nb = kN - kcut + 1;
    const size_t lds = sizeof(double) * (3 * (size_t)n + 2 * (size_t)nb);
    if (lds > 150 * 1024) return qd_fail(c, "zonal filter: row too long for the LDS-staged DFT");
    static bool once[QD_MAX_DEVICES] = {false};                // per (kernel, device)
    const int dv = c->desc.device >= 0 && c->desc.device < QD_MAX_DEVICES ? c->desc.device : 0;
    if (!once[dv]) { hipFuncSetAttribute((const void*)k_zonal_filter, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); once[dv] = c->desc.device == dv; }
    QdFieldList fl{}; fl.n = nf;
    for (int f = 0; f < nf; ++f) { fl.in[f] = fields[f]; fl.out[f] = fields[f]; }
    QD_ROWS(c, m, G, hipLaunchKernelGGL(k_zonal_filter, dim3(1, G.nrows, nf), dim3(QD_BLOCK), lds, c->stream, G, fl, c->zonal_tw,
                                        kcut, kN, 1.0 - s));
    qd_mark(c, {fields[0], nf > 1 ? fields[1] : fields[0], nf > 2 ? fields[2] : fields[0]}, m);
    return 0;
}
