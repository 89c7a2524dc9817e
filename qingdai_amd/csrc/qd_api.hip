// qd_api.hip -- the C-ABI of include/qingdai_hip.h: context, memory, tables, operator seam.
#include "qd_internal.h"
#include <chrono>
#include <atomic>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

thread_local std::string g_qd_create_err;

int qd_fail(qd_ctx* c, const char* what, hipError_t e) {
    std::string m = what ? what : "error";
    if (e != hipSuccess) { m += ": "; m += hipGetErrorString(e); }
    if (c) c->err = m; else g_qd_create_err = m;
    return -1;
}

static hipEvent_t qd_get_event(qd_ctx* c) {
    if (!c->ev_free.empty()) { hipEvent_t e = c->ev_free.back(); c->ev_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}
QdScope::QdScope(qd_ctx* c_, const char* n, bool attach_) : c(c_), name(n), attach(attach_) {
    on = c->timing == 1;
    if (c->timing == 2) {                       // selection: "name[:stride]" or a comma-separated list of them
        const std::string& sel = c->timing_sel;
        const size_t ln = std::strlen(n);
        for (size_t pos = sel.find(n); pos != std::string::npos && !on; pos = sel.find(n, pos + 1)) {
            const size_t e = pos + ln;
            if (!(pos == 0 || sel[pos - 1] == ',') || !(e == sel.size() || sel[e] == ',' || sel[e] == ':')) continue;
            // an event pair costs the stream ~5 us of idle time per bracket (measured in the kernel trace): with a stride only
            // every stride-th launch of the group is bracketed, so the timed loop is barely disturbed by its own measurement
            const long stride = (e < sel.size() && sel[e] == ':') ? std::max(1L, std::atol(sel.c_str() + e + 1)) : 1L;
            on = (c->timing_seen[n]++ % stride) == 0;
        }
    }
    if (on) { e0 = qd_get_event(c); e1 = qd_get_event(c); if (!attach) hipEventRecord(e0, c->stream); }
}
QdScope::~QdScope() {
    if (!on) return;
    if (attach && !used) { c->ev_free.push_back(e0); c->ev_free.push_back(e1); return; }   // never recorded: nothing to resolve
    if (!attach) hipEventRecord(e1, c->stream);
    c->pending.push_back({e0, e1, name});
}
static void qd_resolve_timers(qd_ctx* c) {
    if (c->pending.empty()) return;
    hipStreamSynchronize(c->stream);
    for (auto& p : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) { QdTimer& t = c->timers[p.name]; t.total_ms += ms; t.n += 1; }
        c->ev_free.push_back(p.e0); c->ev_free.push_back(p.e1);
    }
    c->pending.clear();
}

double* qd_scratch(qd_ctx* c, int i) { return c->scratch[i]; }
void qd_swap(qd_ctx* c, int field, int si) { std::swap(c->f[field], c->scratch[si]); }

// ------------------------------------------------------------------ tables
static double* dev_table(qd_ctx* c, const std::vector<double>& v) {
    double* d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(double)) != hipSuccess) return nullptr;
    hipMemcpy(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice);
    c->tab_alloc.push_back(d);
    return d;
}

static std::vector<double> linspace(double a, double b, int n) {
    // numpy.linspace: start + arange(n) * step, last sample forced to `stop`
    std::vector<double> v(n);
    const double step = (b - a) / (double)(n - 1);
    for (int i = 0; i < n; ++i) v[i] = a + (double)i * step;
    if (n > 1) v[n - 1] = b;
    return v;
}

static int build_tables(qd_ctx* c) {
    const int nlat = c->geo.nlat, nlon = c->geo.nlon;
    const double d2r = M_PI / 180.0;
    std::vector<double> lat = linspace(-90.0, 90.0, nlat), lon = linspace(0.0, 360.0, nlon);
    c->dlat = (lat[1] - lat[0]) * d2r;        // np.deg2rad(lat[1]-lat[0])   grid.py:37-38
    c->dlon = (lon[1] - lon[0]) * d2r;
    std::vector<double> cr(nlat), sr(nlat), c6(nlat), c3(nlat), c02(nlat), c05(nlat), fc(nlat), wa(nlat), rx(nlat);
    const qd_params& p = c->p;
    for (int i = 0; i < nlat; ++i) {
        const double phi = lat[i] * d2r;
        cr[i] = std::cos(phi); sr[i] = std::sin(phi);
        c6[i] = std::max(cr[i], 1e-6); c3[i] = std::max(cr[i], 1e-3);
        c02[i] = std::max(cr[i], 0.2); c05[i] = std::max(cr[i], 0.5);
        fc[i] = 2 * p.omega * sr[i];
        wa[i] = std::max(cr[i], 0.0);
        const double lat_deg = std::fabs(phi * (180.0 / M_PI));          // ocean.py:332
        double s = (lat_deg - p.polar_sponge_lat) / std::max(1e-6, 90.0 - p.polar_sponge_lat);
        s = std::min(std::max(s, 0.0), 1.0);
        rx[i] = p.polar_sponge_gain * (s * s);
    }
    {
        const double a = p.a, dphi = c->dlat, dlam = c->dlon;
        const double f_min = 2.0 * p.omega * std::sin(5.0 * (M_PI / 180.0));
        for (int k = 0; k < 2; ++k) {
            const std::vector<double>& ck = k == 0 ? c02 : c05;
            std::vector<double> A(nlat), P(nlat), Q(nlat);
            for (int i = 0; i < nlat; ++i) {
                A[i] = ck[i] / (2.0 * dphi);
                P[i] = 1.0 / ((a * a) * ck[i] * (2.0 * dphi));
                Q[i] = 1.0 / ((a * a) * (dlam * dlam) * (ck[i] * ck[i]));
            }
            // rows 0,1,n-2,n-1 (one-sided np.gradient, grid.py:41-88) in the same reciprocal form:
            //   L = P[g] * (Ab * dFb - Aa * dFa) + Q[g] * d2,  (Aa, Ab) per pole-row type t = 0..3 (g = 0, 1, n-2, n-1)
            std::vector<double> PA(8, 0.0);
            if (nlat >= 5) {
                const int n = nlat;
                PA[0] = ck[0] / dphi;               PA[1] = ck[1] / (2.0 * dphi);      P[0] = 1.0 / ((a * a) * ck[0] * dphi);
                PA[2] = ck[0] / dphi;               PA[3] = ck[2] / (2.0 * dphi);      P[1] = 1.0 / ((a * a) * ck[1] * (2.0 * dphi));
                PA[4] = ck[n - 3] / (2.0 * dphi);   PA[5] = ck[n - 1] / dphi;          P[n - 2] = 1.0 / ((a * a) * ck[n - 2] * (2.0 * dphi));
                PA[6] = ck[n - 2] / (2.0 * dphi);   PA[7] = ck[n - 1] / dphi;          P[n - 1] = 1.0 / ((a * a) * ck[n - 1] * dphi);
            }
            c->tabs.lapA[k] = dev_table(c, A); c->tabs.lapP[k] = dev_table(c, P); c->tabs.lapQ[k] = dev_table(c, Q);
            c->tabs.lapPoleA[k] = dev_table(c, PA);
            std::vector<double> K4((size_t)nlat * 4);
            for (int i = 0; i < nlat; ++i) {
                K4[4 * i + 0] = A[i > 0 ? i - 1 : 0]; K4[4 * i + 1] = A[i < nlat - 1 ? i + 1 : nlat - 1];
                K4[4 * i + 2] = P[i]; K4[4 * i + 3] = Q[i];
            }
            c->tabs.lapK[k] = dev_table(c, K4);
            c->h_lapK[k] = K4;
        }
        std::vector<double> cu(nlat), cv(nlat), px(nlat), igx(nlat), ia6(nlat);
        for (int i = 0; i < nlat; ++i) {
            const double f = fc[i];
            const double sgn = f >= 0.0 ? 1.0 : -1.0;
            const double f_safe = std::fabs(f) < f_min ? sgn * f_min : f;
            cu[i] = -(p.g / (f_safe * a * c6[i]));
            cv[i] = p.g / (f_safe * a);
            px[i] = -(p.g / (a * c6[i]));
            igx[i] = 1.0 / (a * c05[i]);
            ia6[i] = 1 / (a * c6[i]);
        }
        c->tabs.mom_cu = dev_table(c, cu); c->tabs.mom_cv = dev_table(c, cv); c->tabs.mom_px = dev_table(c, px);
        c->tabs.ocn_igx = dev_table(c, igx);
        c->tabs.inv_acos6 = dev_table(c, ia6);
    }
    std::vector<double> lr(nlon), sl(nlon), cl(nlon);
    for (int j = 0; j < nlon; ++j) { lr[j] = lon[j] * d2r; sl[j] = std::sin(lr[j]); cl[j] = std::cos(lr[j]); }
    QdTabs& T = c->tabs;
    T.cos_raw = dev_table(c, cr); T.sin_raw = dev_table(c, sr); T.cos6 = dev_table(c, c6); T.cos3 = dev_table(c, c3);
    T.cos02 = dev_table(c, c02); T.cos05 = dev_table(c, c05); T.fcor = dev_table(c, fc); T.warea = dev_table(c, wa);
    T.r_extra = dev_table(c, rx);
    T.lat_deg = dev_table(c, lat);
    T.lon_rad = dev_table(c, lr); T.sin_lon = dev_table(c, sl); T.cos_lon = dev_table(c, cl);
    if (!T.cos_raw || !T.lon_rad || !T.cos_lon) return -1;
    double ws = 0.0;
    for (int i = 0; i < nlat; ++i) ws += wa[i] * nlon;
    c->wsum_all = ws;
    return 0;
}

// k4 row maps: sigma4 * min(a dphi, a dlam cos)^4 / max(1e-12, dt)   (dynamics.py:557-570, ocean.py:343-352)
int qd_build_k4_tables(qd_ctx* c, double dt, bool ocean, double sub_dt) {
    const int nlat = c->geo.nlat;
    const qd_params& p = c->p;
    const double d2r = M_PI / 180.0;
    std::vector<double> lat = linspace(-90.0, 90.0, nlat);
    auto isset = [](double x) { return !(x != x); };
    if (!ocean) {
        if (c->k4_atm_dt == dt) return 0;
        std::vector<double> tab((size_t)5 * nlat);
        const double scale[5] = {1.0, 1.0, 0.5, 0.5, 0.25};
        const double ov[5] = {p.k4_u, p.k4_v, p.k4_h, p.k4_q, p.k4_cloud};
        bool anypos[5] = {false, false, false, false, false};
        for (int i = 0; i < nlat; ++i) {
            const double cs = std::max(std::cos(lat[i] * d2r), 1e-3);
            const double dxm = std::min(p.a * c->dlat, p.a * c->dlon * cs);
            const double base = p.sigma4 * std::pow(dxm, 4.0) / std::max(1e-12, dt);
            for (int f = 0; f < 5; ++f) {
                const double k = (scale[f] == 1.0) ? base : scale[f] * base;
                tab[(size_t)f * nlat + i] = k;
                if (k > 0.0) anypos[f] = true;
            }
        }
        for (int f = 0; f < 5; ++f) {
            const bool pos = isset(ov[f]) ? (ov[f] > 0.0) : anypos[f];
            bool apply = pos;                                    // _hyperdiffuse early-outs when k4 <= 0
            // q / cloud are only *called* when k4>0 or QD_DIFF_Q/CLOUD=1 (dynamics.py:589-594); the call
            // itself still early-outs for k4<=0, so `pos` decides in every case.
            c->k4_atm_skip[f] = apply ? 0 : 1;
        }
        QD_HIP(c, hipMemcpy(c->k4_atm, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
        c->h_k4[0] = tab; c->qs_key[0][0] = NAN;
        c->k4_atm_dt = dt;
    } else {
        if (c->k4_ocn_dt == sub_dt) return 0;
        std::vector<double> tab((size_t)3 * nlat);
        const double ov[3] = {p.ocean_k4_u, p.ocean_k4_v, p.ocean_k4_eta};
        bool anypos[3] = {false, false, false};
        for (int i = 0; i < nlat; ++i) {
            const double cs = std::max(std::cos(lat[i] * d2r), 0.5);
            const double dxm = std::min(p.a * c->dlat, p.a * c->dlon * cs);
            const double base = p.sigma4_ocean * std::pow(dxm, 4.0) / std::max(1e-12, sub_dt);
            tab[i] = base; tab[(size_t)nlat + i] = base; tab[(size_t)2 * nlat + i] = 0.5 * base;
            if (base > 0) { anypos[0] = anypos[1] = true; }
            if (0.5 * base > 0) anypos[2] = true;
        }
        for (int f = 0; f < 3; ++f) c->k4_ocn_skip[f] = (isset(ov[f]) ? (ov[f] > 0.0) : anypos[f]) ? 0 : 1;
        QD_HIP(c, hipMemcpy(c->k4_ocn, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
        c->h_k4[1] = tab; c->qs_key[1][0] = NAN;
        c->k4_ocn_dt = sub_dt;
    }
    return 0;
}

// ------------------------------------------------------------------ init kernels
__global__ void __launch_bounds__(QD_BLOCK)
k_init_state(QdGeom G, QdTabs T, double H, double q0, double* __restrict__ h, double* __restrict__ Ts,
             double* __restrict__ q, double* __restrict__ sst) {
    const int j = blockIdx.x * QD_BLOCK + threadIdx.x;
    if (j >= G.nlon) return;
    const int l = blockIdx.y;                 // every local row incl. halo
    int g = l + G.row0 - G.halo;
    if (g < 0) g += G.nlat; if (g >= G.nlat) g -= G.nlat;
    const size_t o = (size_t)l * G.nlon + j;
    const double s = T.sin_raw[g];
    h[o] = H + 300 * (s * s);                 // dynamics.py:60-62
    Ts[o] = 288.0; q[o] = q0; sst[o] = 288.0;
}

static double host_qsat(double T, double p0) {
    double T_c = std::min(std::max(T - 273.15, -80.0), 60.0);
    double e_s = 610.94 * std::exp(17.625 * T_c / (T_c + 243.04));
    double denom = std::max(p0 - (1.0 - 0.622) * e_s, 1.0);
    return std::min(std::max(0.622 * e_s / denom, 0.0), 0.5);
}


// ------------------------------------------------------------------ small device -> host hand-overs without a stream sync
int qd_wait_host_flag(qd_ctx* c, volatile double* flag, double seq, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (*flag != seq) {
        __builtin_ia32_pause();
        if ((++spins & 0x3FFFu) == 0) {
            // a stream that is no longer "not ready" cannot deliver the flag any more: either it drained without writing it, or a
            // kernel faulted (sticky error) -- report that error instead of spinning out the timeout
            const hipError_t q = hipStreamQuery(c->stream);
            if (q != hipErrorNotReady && *flag != seq) return qd_fail(c, what, q);
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) return qd_fail(c, what);
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
}


// a self-validating slot in pinned host memory: the host keeps it at -1, the device stores a value >= 0 into it exactly once (one
// uncached 8-byte store, no stamp and nothing to wait for on the device).  Takes the value and puts the slot back to -1.
int qd_wait_host_nonneg(qd_ctx* c, volatile double* slot, double* out, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    double v;
    while (!((v = *slot) >= 0.0)) {
        __builtin_ia32_pause();
        if ((++spins & 0x3FFFu) == 0) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q != hipErrorNotReady && !(*slot >= 0.0)) return qd_fail(c, what, q);
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) return qd_fail(c, what);
        }
    }
    *slot = -1.0;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    *out = v;
    return 0;
}

// n <= 32 device doubles into pinned host memory behind a stamp (system-scope stores; the stamp leaves after the values have been
// acknowledged): what the host needs to decide its next launches (CFL maxima -> n_sub, the miss flag of a band's median).
__global__ void k_publish_host(const double* __restrict__ src, int n, double* dst, double* stamp, double seq) {
    if ((int)threadIdx.x < n)
        __hip_atomic_store((unsigned long long*)dst + threadIdx.x, (unsigned long long)__double_as_longlong(src[threadIdx.x]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0)
        __hip_atomic_store((unsigned long long*)stamp, (unsigned long long)__double_as_longlong(seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int qd_fetch_scalars(qd_ctx* c, const double* dsrc, int n, double* hdst) {
    if (n < 1 || n > 32) return qd_fail(c, "qd_fetch_scalars: bad count");
    c->pub_seq += 1.0;
    hipLaunchKernelGGL(k_publish_host, dim3(1), dim3(64), 0, c->stream, dsrc, n, hdst, c->hpin + 61, c->pub_seq);
    if (qd_wait_host_flag(c, c->hpin + 61, c->pub_seq, "device scalars never reached the host")) return -1;
    if (c->hpin[60] != 0.0) return qd_fail(c, "peer exchange: a rank did not arrive within the deadline");
    return 0;
}

// ------------------------------------------------------------------ lifetime
extern "C" int qd_abi_version(void) { return QD_ABI_VERSION; }
extern "C" int qd_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

// launcher tuning switches (QdTune): read at qd_create and, for the developer scripts that sweep them on one handle, by qd_tune_reload
static void qd_read_tuning(qd_ctx* c) {
    c->tune = QdTune();
    auto geti = [](const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; };
    QdTune& t = c->tune;
    c->stream_rows = std::max(0, geti("QD_STREAM_R", 0));
    { const int r = geti("QD_TAIL_R", 0); if (r > 0) t.tail_r = r; }
    { const int r = geti("QD_TAIL_RP", 0); if (r >= 3) t.tail_rp = r; }
    t.tail_v = geti("QD_TAIL_V", 0) == 1 ? 1 : 0;
    t.tail_general = geti("QD_TAIL_GENERAL", 0) == 1 ? 1 : 0;
    t.stream_r_dyn = std::max(0, geti("QD_STREAM_R_DYN", 0));
    t.stream_r_ocn = std::max(0, geti("QD_STREAM_R_OCN", 0));
    t.stream_vb = geti("QD_STREAM_VB", -1);
    { const int b = geti("QD_MED_BLOCKS", 0); if (b >= 16) t.med_blocks = b; }
    t.shapiro_r = std::max(0, geti("QD_SHAPIRO_R", 0));
    t.tile_tr = std::max(0, geti("QD_TILE_TR", 0));
    t.fused_r = std::max(0, geti("QD_FUSED_R", 0));
    t.fused_seq = geti("QD_FUSED_SEQ", 0) == 1 ? 1 : 0;
    t.fused_norot = geti("QD_FUSED_NOROT", 0) == 1 ? 1 : 0;
    t.stream_no_pair = geti("QD_STREAM_NO_PAIR", 0) == 1 ? 1 : 0;
}
extern "C" int qd_tune_reload(qd_handle c) { if (!c) return -1; qd_read_tuning(c); c->tile = QdTileShape{0, 0, 0, 0}; return 0; }

extern "C" int qd_create(const qd_grid_desc* d, const qd_params* params, double q_init_rh, qd_handle* out) {
    if (!d || !params || !out) return qd_fail(nullptr, "qd_create: null argument");
    if (d->n_lat < 5 || d->n_lon < 4) return qd_fail(nullptr, "qd_create: grid too small (need n_lat>=5, n_lon>=4)");
    if (d->row0 < 0 || d->n_rows < 1 || d->row0 + d->n_rows > d->n_lat || d->halo < 0)
        return qd_fail(nullptr, "qd_create: bad latitude band");
    const bool full = (d->row0 == 0 && d->n_rows == d->n_lat);
    if (full && d->halo != 0) return qd_fail(nullptr, "qd_create: a whole-globe handle takes halo = 0");
    // (a band must be shorter than the globe by at least its two halos: halo rows alias owned rows with period n_lat otherwise)
    if (!full && d->n_rows + 2 * d->halo > d->n_lat) return qd_fail(nullptr, "qd_create: band + 2 halos taller than the grid");
    if (!full && d->halo < 5) return qd_fail(nullptr, "qd_create: band handles need halo >= 5 rows");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return qd_fail(nullptr, "qd_create: no HIP device visible");
    if (d->device < 0 || d->device >= ndev) return qd_fail(nullptr, "qd_create: bad device ordinal");
    qd_ctx* c = new qd_ctx();
    c->desc = *d; c->p = *params;
    { const char* ef = std::getenv("QD_FUSED"); if (ef && ef[0] == '0') c->use_fused = 0; }
    { const char* ef = std::getenv("QD_FUSED_FAST"); if (ef) c->fused_fast = std::atoi(ef); }
    { const char* ef = std::getenv("QD_TAIL_ACC"); if (ef) c->tail_acc = std::atoi(ef); }
    { const char* ef = std::getenv("QD_SHAPIRO_STREAM"); if (ef) c->shapiro_stream = std::atoi(ef); }
    { const char* ef = std::getenv("QD_OCN_TAIL"); if (ef) c->ocn_tail = std::atoi(ef); }   // 0: two launches (k_cont_sstadv, k_sst_outlier_fused); anything else: k_ocn_tail_fast (QD_TAIL_V=1: k_ocn_tail_stream).  The LDS-tile forms and the one-launch sub-step of round 3 (2, 3, 4) are retired: tools/retired/
    if (c->ocn_tail != 0) c->ocn_tail = 1;
    { const char* ef = std::getenv("QD_OCN_FUSED"); if (ef) c->ocn_fused = ef[0] == '1' ? 1 : 0; }
    { const char* ef = std::getenv("QD_BAND_TAIL"); if (ef) c->band_tail = ef[0] == '0' ? 0 : 1; }
    qd_read_tuning(c);
    { const char* ef = std::getenv("QD_GROUP_SUMS"); if (ef) c->group_sums = ef[0] == '0' ? 0 : 1; }
    c->geo = QdGeom{d->n_lat, d->n_lon, d->row0, d->n_rows, d->halo, full ? 1 : 0, d->row0 - d->halo, d->n_rows + 2 * d->halo};
    c->own_row0 = d->row0; c->own_nrows = d->n_rows;
    auto bail = [&](const char* w, hipError_t e) { qd_fail(nullptr, w, e); qd_destroy(c); return -1; };
    hipError_t e;
    if ((e = hipSetDevice(d->device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreate(&c->stream)) != hipSuccess) return bail("hipStreamCreate", e);
    { int ncu = 0; c->n_cu = (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, d->device) == hipSuccess && ncu > 0) ? ncu : 256; }
    // every slab is allocated with QD_PAD_ROWS rows of slack behind its last row: the row-streaming kernels prefetch a few rows
    // ahead of the row they work on without clamping (what they read there is never used)
    const size_t cells = c->geo.cells() + (size_t)QD_PAD_ROWS * c->geo.nlon;
    for (int f = 0; f < QD_F_COUNT_F64; ++f) {
        if ((e = hipMalloc(&c->f[f], cells * sizeof(double))) != hipSuccess) return bail("hipMalloc field", e);
        hipMemsetAsync(c->f[f], 0, cells * sizeof(double), c->stream);
    }
    for (int s = 0; s < QD_NSCRATCH; ++s) {
        if ((e = hipMalloc(&c->scratch[s], cells * sizeof(double))) != hipSuccess) return bail("hipMalloc scratch", e);
        hipMemsetAsync(c->scratch[s], 0, cells * sizeof(double), c->stream);
    }
    if ((e = hipMalloc(&c->land, cells)) != hipSuccess) return bail("hipMalloc mask", e);
    if ((e = hipMalloc(&c->icemask, cells)) != hipSuccess) return bail("hipMalloc mask", e);
    hipMemsetAsync(c->land, 0, cells, c->stream); hipMemsetAsync(c->icemask, 0, cells, c->stream);
    if (build_tables(c)) return bail("table allocation", hipErrorOutOfMemory);
    if ((e = hipMalloc(&c->k4_atm, (size_t)5 * d->n_lat * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&c->k4_ocn, (size_t)3 * d->n_lat * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
    c->red_blocks = (12 + (d->n_lon + QD_BLOCK - 1) / QD_BLOCK) * c->geo.lrows() + 64;   // >= 10 x rows: qd_energy_diagnostics
    if ((e = hipMalloc(&c->red_partial, (size_t)c->red_blocks * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&c->dscal, QD_S_COUNT * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&c->dcount, 32 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&c->eta_acc, 512 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    hipMemsetAsync(c->eta_acc, 0, 512 * sizeof(unsigned long long), c->stream);
    if ((e = hipMalloc(&c->hist, 2 * QD_HIST_BINS * sizeof(unsigned int))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&c->sel_state, 8 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    hipMemsetAsync(c->dscal, 0, QD_S_COUNT * sizeof(double), c->stream);
    hipMemsetAsync(c->dcount, 0, 32 * sizeof(unsigned long long), c->stream);
    hipMemsetAsync(c->hist, 0, 2 * QD_HIST_BINS * sizeof(unsigned int), c->stream);
    hipMemsetAsync(c->sel_state, 0, 8 * sizeof(unsigned long long), c->stream);
    { const char* ef = std::getenv("QD_TAIL_FIX"); if (ef) c->tail_fix = ef[0] == '0' ? 0 : 1; }
    c->fix_dense = 256.0 * (double)c->geo.cells() / (721.0 * 1440.0);        // measured at 721 x 1440; the patch loop's share of a launch goes with entries per cell
    { const char* ef = std::getenv("QD_TAIL_FIX_DENSE"); if (ef) c->fix_dense = std::atof(ef); }
    { const char* ef = std::getenv("QD_LAZY_DIAG"); if (ef) c->lazy_diag = ef[0] == '0' ? 0 : 1; }
    if (c->tail_fix) {                                        // list of the cells whose currents the ocean tail kernel changes (qd_ocntail.h)
        if ((e = hipMalloc(&c->fix_count, 64)) != hipSuccess) return bail("hipMalloc", e);
        hipMemsetAsync(c->fix_count, 0, 64, c->stream);
        if ((e = hipMalloc(&c->fix_list, c->geo.cells() * 3 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    }
    if (!full) {
        if ((e = hipMalloc(&c->med_pred, 64 * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
        hipMemsetAsync(c->med_pred, 0, 64 * sizeof(double), c->stream);
        if ((e = hipMalloc(&c->med_gather, (size_t)std::max(1, d->world) * 4096 * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
        if ((e = hipMalloc(&c->sel_ccount, 2 * sizeof(unsigned int))) != hipSuccess) return bail("hipMalloc", e);
        hipMemsetAsync(c->sel_ccount, 0, 2 * sizeof(unsigned int), c->stream);
        { const char* ef = std::getenv("QD_MEDIAN_PREDICT"); if (ef && ef[0] == '0') c->med_predict = 0; }
    }
    if (full) {
        if ((e = hipMalloc(&c->sel_cand, 2 * cells * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
        if ((e = hipMalloc(&c->sel_ccount, 2 * sizeof(unsigned int))) != hipSuccess) return bail("hipMalloc", e);
        hipMemsetAsync(c->sel_ccount, 0, 2 * sizeof(unsigned int), c->stream);
        if ((e = hipMalloc(&c->med_pred, 64 * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
        hipMemsetAsync(c->med_pred, 0, 64 * sizeof(double), c->stream);
        { const char* ef = std::getenv("QD_MEDIAN_PREDICT"); if (ef && ef[0] == '0') c->med_predict = 0; }
        { const char* ef = std::getenv("QD_MERGE_POINTWISE"); if (ef) c->merge_pointwise = ef[0] == '0' ? 0 : 1; }
        { const char* ef = std::getenv("QD_HOIST_PRECIP"); if (ef) c->hoist_precip = ef[0] == '0' ? 0 : 1; }
        { const char* ef = std::getenv("QD_SIDE_STREAM"); if (ef) c->side_stream_on = ef[0] == '0' ? 0 : 1; }
        { const char* ef = std::getenv("QD_MERGE_FINAL"); if (ef) c->merge_final = ef[0] == '0' ? 0 : 1; }
        { const char* ef = std::getenv("QD_MERGE_PCOND"); if (ef) c->merge_pcond = ef[0] == '0' ? 0 : 1; }


        // per-workgroup CFL maxima of k_final_qnet_stress: 2 x (segments x rows) doubles
        c->n_wgmax = (int)(qd_grid2d(c->geo).x * (unsigned)c->geo.nrows);
        if ((e = hipMalloc(&c->wgmax, (size_t)2 * c->n_wgmax * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
        { const char* ef = std::getenv("QD_MED_SIDE"); if (ef) c->med_side = ef[0] == '0' ? 0 : 1; }
        { const char* ef = std::getenv("QD_MED_PAIR"); if (ef) c->med_pair = ef[0] == '0' ? 0 : 1; }
        { const char* ef = std::getenv("QD_MED_FOLD"); if (ef) c->med_fold = ef[0] == '0' ? 0 : 1; }
        { const char* ef = std::getenv("QD_MERGE_SAF"); if (ef) c->merge_saf = ef[0] == '0' ? 0 : 1; }
        if (c->med_side) {
            if ((e = hipEventCreateWithFlags(&c->med_fork, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
            if ((e = hipEventCreateWithFlags(&c->med_done, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        }
        if (c->med_side || c->med_pair) {                    // a second set of median buffers
            if ((e = hipMalloc(&c->hist_b, 2 * QD_HIST_BINS * sizeof(unsigned int))) != hipSuccess) return bail("hipMalloc", e);
            if ((e = hipMalloc(&c->sel_state_b, 8 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
            if ((e = hipMalloc(&c->sel_cand_b, 2 * cells * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
            if ((e = hipMalloc(&c->sel_ccount_b, 2 * sizeof(unsigned int))) != hipSuccess) return bail("hipMalloc", e);
            hipMemsetAsync(c->hist_b, 0, 2 * QD_HIST_BINS * sizeof(unsigned int), c->stream);
            hipMemsetAsync(c->sel_state_b, 0, 8 * sizeof(unsigned long long), c->stream);
            hipMemsetAsync(c->sel_ccount_b, 0, 2 * sizeof(unsigned int), c->stream);
        }
        if ((c->side_stream_on || c->med_side) && (e = hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
        if (c->side_stream_on) {
            if ((e = hipEventCreateWithFlags(&c->side_fork, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
            if ((e = hipEventCreateWithFlags(&c->side_done, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
            if ((e = hipMalloc(&c->red_partial_b, (size_t)c->red_blocks * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
        }
    }
    if ((e = hipHostMalloc((void**)&c->hpin, 64 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail("hipHostMalloc", e);
    if ((e = hipHostMalloc((void**)&c->hpin_rows, (size_t)3 * c->geo.lrows() * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail("hipHostMalloc", e);
    std::memset(c->hpin, 0, 64 * sizeof(double));
    c->hpin[57] = -1.0; c->hpin[58] = -1.0; c->hpin[59] = -1.0;                  // self-validating slots of k_max2_publish (qd_wait_host_nonneg)
    std::memset(c->hpin_rows, 0, (size_t)3 * c->geo.lrows() * sizeof(double));      // [2 n .. 3 n): per-row arrival stamps of k_stress_max
    // reference initial state
    const double q0 = std::min(std::max(q_init_rh, 0.0), 1.0) * host_qsat(288.0, params->p0);
    hipLaunchKernelGGL(k_init_state, dim3((d->n_lon + QD_BLOCK - 1) / QD_BLOCK, c->geo.lrows()), dim3(QD_BLOCK), 0,
                       c->stream, c->geo, c->tabs, params->H, q0, c->f[QD_F_H], c->f[QD_F_TS], c->f[QD_F_Q], c->f[QD_F_SST]);
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail("init", e);
    if (!full) {
        // a fresh handle is consistent everywhere: every field slab (and the masks) is valid on its whole halo
        for (int f = 0; f < QD_F_COUNT_F64; ++f) c->vm[c->f[f]] = d->halo;
        c->vm[c->land] = d->halo; c->vm[c->icemask] = d->halo;
    }
    *out = c;
    return 0;
}

extern "C" int qd_destroy(qd_handle c) {
    if (!c) return 0;
    hipSetDevice(c->desc.device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->med_pred && std::getenv("QD_MEDIAN_DEBUG")) {
        double h[64];
        if (hipMemcpy(h, c->med_pred, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
            for (int s = 0; s < 4; ++s)
                fprintf(stderr, "[median site %d] last median %.6g bracket [%.6g, %.6g] hits %.0f misses %.0f last list %.0f of %.0f; "
                        "last miss at call %.0f: centre %.6g -> median %.6g\n", s, h[16 * s], h[16 * s + 8], h[16 * s + 9], h[16 * s + 4],
                        h[16 * s + 5], h[16 * s + 6], h[16 * s + 7], h[16 * s + 13], h[16 * s + 14], h[16 * s + 15]);
    }
    qd_phyto_release(c);
    for (int f = 0; f < QD_F_COUNT_F64; ++f) if (c->f[f]) hipFree(c->f[f]);
    for (int s = 0; s < QD_NSCRATCH; ++s) if (c->scratch[s]) hipFree(c->scratch[s]);
    for (double* t : c->tab_alloc) hipFree(t);
    if (c->land) hipFree(c->land); if (c->icemask) hipFree(c->icemask);
    if (c->k4_atm) hipFree(c->k4_atm); if (c->k4_ocn) hipFree(c->k4_ocn); for (int k = 0; k < 2; ++k) if (c->qs_tab[k]) hipFree(c->qs_tab[k]);
    if (c->red_partial) hipFree(c->red_partial); if (c->dscal) hipFree(c->dscal);
    if (c->red_partial_b) hipFree(c->red_partial_b);
    if (c->wgmax) hipFree(c->wgmax);
    qd_saf_drop(c);
    if (c->fix_count && std::getenv("QD_TAIL_FIX_DEBUG")) {
        unsigned int h[8] = {0};
        if (hipMemcpy(h, c->fix_count, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[tail fix list] %u entries in %u launches (%.1f per launch), longest list %u of %zu cells\n", h[4], h[5],
                    h[5] ? (double)h[4] / h[5] : 0.0, h[6], c->geo.cells());
    }
    if (c->fix_count) hipFree(c->fix_count); if (c->fix_list) hipFree(c->fix_list);
    if (c->side_stream) { hipStreamSynchronize(c->side_stream); hipStreamDestroy(c->side_stream); }
    if (c->side_fork) hipEventDestroy(c->side_fork); if (c->side_done) hipEventDestroy(c->side_done);
    if (c->med_fork) hipEventDestroy(c->med_fork); if (c->med_done) hipEventDestroy(c->med_done);
    if (c->hist_b) hipFree(c->hist_b); if (c->sel_state_b) hipFree(c->sel_state_b); if (c->sel_cand_b) hipFree(c->sel_cand_b); if (c->sel_ccount_b) hipFree(c->sel_ccount_b);
    if (c->eta_acc) hipFree(c->eta_acc);
    if (c->dcount) hipFree(c->dcount); if (c->hist) hipFree(c->hist); if (c->sel_state) hipFree(c->sel_state);
    if (c->zonal_tw) hipFree(c->zonal_tw);
    if (c->bands) hipFree(c->bands);
    qd_eco_free(c);
    if (c->sel_cand) hipFree(c->sel_cand); if (c->sel_ccount) hipFree(c->sel_ccount);
    if (c->med_pred) hipFree(c->med_pred);
    if (c->med_gather) hipFree(c->med_gather);
    qd_comm_release(c);
    if (c->hpin) hipHostFree(c->hpin);
    if (c->hpin_rows) hipHostFree(c->hpin_rows);
    if (c->stage) hipHostFree(c->stage);
    qd_resolve_timers(c);
    for (hipEvent_t e : c->ev_free) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

extern "C" const char* qd_last_error(qd_handle c) { return c ? c->err.c_str() : g_qd_create_err.c_str(); }

// ------------------------------------------------------------------ upload / download
// Host arrays are GLOBAL [n_lat][n_lon]; the handle copies its band (+ halo rows, period-n_lat
// at the poles) in, and its owned rows out.
int qd_band_copy_in(qd_ctx* c, void* dst, const void* host, size_t esz) {
    const QdGeom& G = c->geo;
    const size_t rowb = (size_t)G.nlon * esz;
    if (G.full) return hipMemcpyAsync(dst, host, rowb * G.nlat, hipMemcpyHostToDevice, c->stream) == hipSuccess ? 0 : -1;
    for (int l = 0; l < G.lrows(); ++l) {
        int g = l + G.row0 - G.halo;
        if (g < 0) g += G.nlat; if (g >= G.nlat) g -= G.nlat;
        if (hipMemcpyAsync((char*)dst + (size_t)l * rowb, (const char*)host + (size_t)g * rowb, rowb,
                           hipMemcpyHostToDevice, c->stream) != hipSuccess) return -1;
    }
    return 0;
}

extern "C" int qd_upload(qd_handle c, int field, const void* host, size_t bytes) {
    if (!c || !host) return -1;
    hipSetDevice(c->desc.device);
    const size_t n = (size_t)c->geo.nlat * c->geo.nlon;
    if (field == QD_F_LAND_MASK || field == QD_F_ICE_MASK) {
        if (bytes != n) return qd_fail(c, "qd_upload: mask size mismatch");
        uint8_t* dst = field == QD_F_LAND_MASK ? c->land : c->icemask;
        if (qd_band_copy_in(c, dst, host, 1)) return qd_fail(c, "qd_upload: copy failed");
        qd_mark(c, {dst}, c->geo.halo);
    } else {
        if (field < 0 || field >= QD_F_COUNT_F64) return qd_fail(c, "qd_upload: unknown field");
        if (bytes != n * sizeof(double)) return qd_fail(c, "qd_upload: size mismatch (expect n_lat*n_lon float64)");
        if (qd_eco_is_f32(c, field)) {                         // the slab stores f32: stage the f64 host map, round once
            if (qd_band_copy_in(c, c->scratch[10], host, sizeof(double))) return qd_fail(c, "qd_upload: copy failed");
            qd_eco_convert_slab(c, c->scratch[10], 0, c->f[field], 1);
        } else if (qd_band_copy_in(c, c->f[field], host, sizeof(double))) return qd_fail(c, "qd_upload: copy failed");
        qd_mark(c, {c->f[field]}, c->geo.halo);
        if (field == QD_F_CLOUD_EFF) c->cloud_eff_valid = 1;
        if (field == QD_F_ELEVATION) c->has_elevation = 1;
        if (field == QD_F_ECO_ALPHA) c->eco.alpha_valid = 1;          // a map computed elsewhere (restart, host ecology)
        if (field == QD_F_ECO_ALPHA_BANDED) c->eco.banded_valid = 1;
        if (field == QD_F_WATER_ALPHA) c->eco.water_valid = 1;
    }
    QD_HIP(c, hipStreamSynchronize(c->stream));    // host buffer is only borrowed for the call
    if (field == QD_F_LAND_MASK) {
        // area-weighted ocean weight sum for the eta mean removal (ocean.py:372-374)
        const uint8_t* m = (const uint8_t*)host;
        const int nlat = c->geo.nlat, nlon = c->geo.nlon;
        double ws = 0.0;
        for (int i = 0; i < nlat; ++i) {
            const double w = std::max(std::cos((-90.0 + 180.0 * i / (double)(nlat - 1)) * (M_PI / 180.0)), 0.0);
            int cnt = 0;
            for (int j = 0; j < nlon; ++j) cnt += (m[(size_t)i * nlon + j] == 0);
            ws += w * cnt;
        }
        c->wsum_ocean = ws;
    }
    return 0;
}

extern "C" int qd_download(qd_handle c, int field, void* host, size_t bytes) {
    if (!c || !host) return -1;
    hipSetDevice(c->desc.device);
    const QdGeom& G = c->geo;
    const size_t n = (size_t)G.nlat * G.nlon;
    const void* src; size_t esz;
    if (field == QD_F_LAND_MASK || field == QD_F_ICE_MASK) { src = field == QD_F_LAND_MASK ? c->land : c->icemask; esz = 1; }
    else {
        if (field < 0 || field >= QD_F_COUNT_F64) return qd_fail(c, "qd_download: unknown field");
        src = c->f[field]; esz = sizeof(double);
        if (qd_eco_is_f32(c, field)) { qd_eco_convert_slab(c, c->f[field], 1, c->scratch[10], 0); src = c->scratch[10]; }
    }
    if (bytes != n * esz) return qd_fail(c, "qd_download: size mismatch");
    const size_t rowb = (size_t)G.nlon * esz;
    QD_HIP(c, hipMemcpyAsync((char*)host + (size_t)G.row0 * rowb, (const char*)src + (size_t)G.halo * rowb,
                             rowb * G.nrows, hipMemcpyDeviceToHost, c->stream));
    QD_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int qd_set_params(qd_handle c, const qd_params* p, size_t sz) {
    if (!c || !p) return -1;
    if (sz != sizeof(qd_params)) return qd_fail(c, "qd_set_params: struct size mismatch (ABI drift)");
    const bool tabs_stale = (p->omega != c->p.omega) || (p->g != c->p.g) || (p->a != c->p.a) || (p->polar_sponge_lat != c->p.polar_sponge_lat) ||
                            (p->polar_sponge_gain != c->p.polar_sponge_gain);
    c->p = *p;
    c->k4_atm_dt = -1; c->k4_ocn_dt = -1;
    if (tabs_stale) {
        hipSetDevice(c->desc.device);
        hipStreamSynchronize(c->stream);
        for (double* t : c->tab_alloc) hipFree(t);
        c->tab_alloc.clear();
        if (build_tables(c)) return qd_fail(c, "qd_set_params: table rebuild failed");
    }
    return 0;
}

extern "C" int qd_get_step_counter(qd_handle c, int64_t* a, int64_t* o) {
    if (!c) return -1; if (a) *a = c->atm_counter; if (o) *o = c->ocn_counter; return 0;
}
extern "C" int qd_set_step_counter(qd_handle c, int64_t a, int64_t o) {
    if (!c) return -1; c->atm_counter = a; c->ocn_counter = o; return 0;
}
extern "C" int qd_sync(qd_handle c) {
    if (!c) return -1;
    hipSetDevice(c->desc.device);
    QD_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int qd_last_ocean_nsub(qd_handle c, int* n) { if (!c || !n) return -1; *n = c->last_nsub; return 0; }

// ------------------------------------------------------------------ the path
extern "C" int qd_forcing(qd_handle c, const double sa[3], const double sb[3], double theta, int with_teq) {
    if (!c || !sa || !sb) return -1;
    hipSetDevice(c->desc.device);
    return qd_forcing_impl(c, sa, sb, theta, with_teq);
}
extern "C" int qd_simple_albedo(qd_handle c, double ocean_albedo) {
    if (!c) return -1;
    hipSetDevice(c->desc.device);
    return qd_simple_albedo_impl(c, ocean_albedo);
}
extern "C" int qd_atmos_step(qd_handle c, double dt, int has_albedo) {
    if (!c) return -1;
    hipSetDevice(c->desc.device);
    int rc = qd_atmos_step_impl(c, dt, has_albedo);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qd_fail(c, "qd_atmos_step: launch", e);
    return 0;
}
extern "C" int qd_ocean_step(qd_handle c, double dt, int compute_qnet, int use_ice_mask, int inject_sst) {
    if (!c) return -1;
    hipSetDevice(c->desc.device);
    int rc = qd_ocean_step_impl(c, dt, compute_qnet, use_ice_mask, inject_sst);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qd_fail(c, "qd_ocean_step: launch", e);
    return 0;
}
extern "C" int qd_driver_physics(qd_handle c, double dt) {
    if (!c) return -1;
    hipSetDevice(c->desc.device);
    int rc = qd_driver_physics_impl(c, dt);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qd_fail(c, "qd_driver_physics: launch", e);
    return 0;
}

extern "C" int qd_hydrology_commit(qd_handle c, double dt) {
    if (!c) return -1;
    hipSetDevice(c->desc.device);
    int rc = qd_hydrology_commit_impl(c, dt);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qd_fail(c, "qd_hydrology_commit: launch", e);
    return 0;
}

int qd_side_join(qd_ctx* c) {
    if (!c->side_pending) return 0;
    c->side_pending = false;
    QD_HIP(c, hipStreamWaitEvent(c->stream, c->side_done, 0));
    return 0;
}

extern "C" int qd_step_n(qd_handle c, int n, double dt, int flags, const double* stars) {
    if (!c || !stars) return -1;
    hipSetDevice(c->desc.device);
    const int with_ocean = flags & 1, with_phys = flags & 2, pass_alb = flags & 4, with_hydro = flags & 8, want_diag = flags & 16;
    const int with_eco = flags & 32, with_phyto = flags & 64;
    if (with_phyto && !with_ocean) return qd_fail(c, "qd_step_n: the tracer transport (bit6) needs the ocean step (bit0)");
    if (with_phyto && c->phyto.S == 0) return qd_fail(c, "qd_step_n: bit6 set but qd_phyto_configure has not been called");
    if (with_eco && !with_phys) return qd_fail(c, "qd_step_n: the ecology sub-step (bit5) needs the driver physics (bit1)");
    if (with_eco && !c->eco.configured) return qd_fail(c, "qd_step_n: bit5 set but qd_eco_configure has not been called");
    // whatever way this call ends, the per-span switches are back to what a stand-alone qd_* call expects
    struct SpanGuard { qd_ctx* c; ~SpanGuard() { c->diag_write = 1; c->want_pcond_ahead = 0; c->pcond_ahead = 0; c->defer_final = 0; c->final_pending.on = 0; qd_saf_drop(c); } } span_guard{c};
    for (int s = 0; s < n; ++s) {
        const double* st = stars + (size_t)7 * s;
        int rc;
        // EcologyAdapter.step_subdaily sits between the glacier mask and the base-albedo blend (run_simulation.py:2075-2104):
        // its clock / canopy / alpha part runs before the albedo kernel, its E_day += isr dt rides on this step's forcing launch
        if (with_eco && c->eco.p.albedo_couple) { if ((rc = qd_eco_canopy_impl(c, dt))) return rc; c->eco.eday_dt = c->eco.p.use_lai ? dt : 0.0; }
        // whole-globe handles: the forcing rides on the last launch of the driver physics (k_snow_albedo_forcing)
        const bool merged = with_phys && c->merge_pointwise;
        const QdForcingCall fc{st, st + 3, st[6]};
        // lazy diagnostics: inside a span only the last step stores what nothing inside a span reads -- unless a reader comes with the flags
        c->diag_write = (!c->lazy_diag || s == n - 1 || with_hydro || with_eco || with_phyto || want_diag) ? 1 : 0;
        if (with_phys) {
            const int part = c->precip_done ? 2 : 0;         // the precipitation block may have run inside the previous ocean step
            c->precip_done = 0;
            // pass_alb: time_step follows with its P_cond median -- the last physics launch writes that P_cond (k_column<1> merged in)
            c->want_pcond_ahead = (merged && pass_alb && c->merge_pcond) ? 1 : 0;
            rc = qd_driver_physics_impl(c, dt, merged ? &fc : nullptr, part);
            c->want_pcond_ahead = 0;
            if (rc) return rc;
        }
        else if ((rc = qd_simple_albedo_impl(c, 0.08))) return rc;
        if (!merged && (rc = qd_forcing_impl(c, st, st + 3, st[6], 1))) return rc;
        // the ocean step follows at once and nothing in between reads the post-final fields: k_final rides on its first launch
        c->defer_final = (with_ocean && c->geo.full && c->merge_final && c->wgmax && !(want_diag && s == 0)) ? 1 : 0;
        rc = qd_atmos_step_impl(c, dt, pass_alb ? 1 : 0);
        c->defer_final = 0;
        if (rc) return rc;
        if (c->saf_pending && (rc = qd_saf_flush(c))) return rc;          // (never: the column block takes or flushes it)
        // bit4: energy-budget means of the FIRST step, taken where the reference driver takes them -- after time_step, on the
        // fluxes of the coupling block (run_simulation.py:2199-2246) -- and kept for qd_energy_diagnostics_last
        if (with_ocean && want_diag && s == 0 && (rc = qd_energy_diag_impl(c, c->last_diag))) return rc;
        // The precipitation block of step s + 1 (run_simulation.py:1740-1790) reads u, v and P_cond as time_step left them and
        // nothing the ocean step, the tracers, the individuals or the bucket touch, and writes only what the rest of step s + 1's
        // driver physics reads: it is queued inside the ocean step, between the stress kernel and the host's wait for the CFL maxima.
        if (with_ocean && with_phys && c->geo.full && c->hoist_precip && s + 1 < n)
            c->before_cfl_wait = [c, dt]() { const int r = qd_driver_physics_impl(c, dt, nullptr, 1); if (!r) c->precip_done = 1; return r; };
        if (with_ocean) { rc = qd_ocean_step_impl(c, dt, 1, 1, 1); c->before_cfl_wait = nullptr; if (rc) return rc; }
        if (with_phyto && (rc = qd_phyto_step_impl(c, dt))) return rc;      // run_simulation.py:2254-2258
        // IndividualPool.try_substep reads this step's isr_A / isr_B and W_land before the bucket update (run_simulation.py:2021-2046)
        if (with_eco && c->eco.n_indiv > 0 && (rc = qd_indiv_substep_impl(c, dt, nullptr))) return rc;
        if (with_hydro && (rc = qd_hydrology_commit_impl(c, dt))) return rc;
    }
    c->diag_write = 1;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qd_fail(c, "qd_step_n: launch", e);
    return 0;
}

// ------------------------------------------------------------------ operator seam
static int seam_in(qd_ctx* c, double* dst, const double* host) {
    if (qd_band_copy_in(c, dst, host, sizeof(double))) return qd_fail(c, "operator seam: upload failed");
    return 0;
}
static int seam_out(qd_ctx* c, const double* src, double* host) {
    const QdGeom& G = c->geo;
    const size_t rowb = (size_t)G.nlon * sizeof(double);
    QD_HIP(c, hipMemcpyAsync((char*)host + (size_t)G.row0 * rowb, (const char*)src + (size_t)G.halo * rowb,
                             rowb * G.nrows, hipMemcpyDeviceToHost, c->stream));
    QD_HIP(c, hipStreamSynchronize(c->stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qd_fail(c, "operator seam: kernel", e);
    return 0;
}

extern "C" int qd_op_laplacian(qd_handle c, const double* F, int cos_kind, double* out) {
    if (!c || !F || !out) return -1;
    hipSetDevice(c->desc.device);
    double* in = c->scratch[10];
    if (seam_in(c, in, F)) return -1;
    QdFieldList fl; fl.n = 1; fl.in[0] = in; fl.out[0] = c->scratch[11]; fl.aux[0] = nullptr; fl.k4row[0] = nullptr; fl.k4s[0] = 0;
    qd_launch_laplacian(c, fl, cos_kind ? c->tabs.cos05 : c->tabs.cos02, 0);
    return seam_out(c, c->scratch[11], out);
}

extern "C" int qd_op_hyperdiffuse(qd_handle c, const double* F, const double* k4_row, double k4_scalar, double dt,
                                  int n_substeps, int cos_kind, double* out) {
    if (!c || !F || !out) return -1;
    hipSetDevice(c->desc.device);
    double* in = c->scratch[10];
    if (seam_in(c, in, F)) return -1;
    double* tab = c->scratch[12];           // borrow a slab for the row map
    int skip = 0;
    double ov = NAN;
    if (k4_row) {
        bool anypos = false;
        for (int i = 0; i < c->geo.nlat; ++i) anypos |= (k4_row[i] > 0.0);
        skip = anypos ? 0 : 1;
        QD_HIP(c, hipMemcpyAsync(tab, k4_row, c->geo.nlat * sizeof(double), hipMemcpyHostToDevice, c->stream));
        QD_HIP(c, hipStreamSynchronize(c->stream));
    } else { ov = k4_scalar; skip = (k4_scalar > 0.0) ? 0 : 1; }
    double* fl[1] = {in};
    double* save10 = c->scratch[10];
    qd_hyperdiffuse_fields(c, fl, 1, tab, &skip, &ov, dt, n_substeps, cos_kind ? c->tabs.cos05 : c->tabs.cos02, 0);
    int rc = seam_out(c, fl[0], out);
    // restore scratch bookkeeping: slot 10 must keep owning a distinct slab
    if (fl[0] != save10) { for (int s = 0; s < QD_NSCRATCH; ++s) if (c->scratch[s] == save10 && s != 10) { c->scratch[s] = fl[0]; break; } c->scratch[10] = save10; }
    return rc;
}

extern "C" int qd_op_advect(qd_handle c, const double* field, const double* u, const double* v, double dt,
                            int cos_kind, double* out) {
    if (!c || !field || !u || !v || !out) return -1;
    hipSetDevice(c->desc.device);
    if (seam_in(c, c->scratch[10], field) || seam_in(c, c->scratch[11], u) || seam_in(c, c->scratch[12], v)) return -1;
    qd_launch_advect(c, c->scratch[11], c->scratch[12], cos_kind ? c->tabs.cos05 : c->tabs.cos6, dt,
                     c->scratch[10], c->scratch[13], nullptr, nullptr, 1.0, 0, 0);
    return seam_out(c, c->scratch[13], out);
}

extern "C" int qd_op_shapiro(qd_handle c, const double* F, int n, double* out) {
    if (!c || !F || !out) return -1;
    hipSetDevice(c->desc.device);
    if (seam_in(c, c->scratch[10], F)) return -1;
    double* fl[1] = {c->scratch[10]};
    double* save10 = c->scratch[10];
    qd_shapiro_fields(c, fl, 1, n, 0);
    int rc = seam_out(c, fl[0], out);
    if (fl[0] != save10) { for (int s = 0; s < QD_NSCRATCH; ++s) if (c->scratch[s] == save10 && s != 10) { c->scratch[s] = fl[0]; break; } c->scratch[10] = save10; }
    return rc;
}

static int seam_divvort(qd_ctx* c, const double* u, const double* v, double* out, int vort) {
    if (!c || !u || !v || !out) return -1;
    hipSetDevice(c->desc.device);
    if (seam_in(c, c->scratch[10], u) || seam_in(c, c->scratch[11], v)) return -1;
    qd_launch_divvort(c, c->scratch[10], c->scratch[11], c->scratch[12], vort, 0);
    return seam_out(c, c->scratch[12], out);
}
extern "C" int qd_op_zonal_filter(qd_handle c, const double* F, double cutoff, double damp, double* out) {
    if (!c || !F || !out) return -1;
    hipSetDevice(c->desc.device);
    if (seam_in(c, c->scratch[10], F)) return -1;
    double* fl[1] = {c->scratch[10]};
    if (qd_zonal_filter_fields(c, fl, 1, cutoff, damp, 0)) return -1;
    return seam_out(c, c->scratch[10], out);
}
extern "C" int qd_op_divergence(qd_handle c, const double* u, const double* v, double* out) { return seam_divvort(c, u, v, out, 0); }
extern "C" int qd_op_vorticity(qd_handle c, const double* u, const double* v, double* out) { return seam_divvort(c, u, v, out, 1); }

extern "C" int qd_op_gaussian(qd_handle c, const double* F, double sigma, int mode_wrap, double* out) {
    if (!c || !F || !out) return -1;
    hipSetDevice(c->desc.device);
    if (seam_in(c, c->scratch[10], F)) return -1;
    if (qd_gaussian(c, c->scratch[10], c->scratch[11], c->scratch[12], sigma, mode_wrap, 0)) return -1;
    return seam_out(c, c->scratch[11], out);
}

extern "C" int qd_op_median_positive(qd_handle c, const double* x, double dflt, double* out) {
    if (!c || !x || !out) return -1;
    hipSetDevice(c->desc.device);
    if (seam_in(c, c->scratch[10], x)) return -1;
    qd_median_positive_dev(c, c->scratch[10], dflt, QD_S_MED_OUT, 0, 0.0, 0);
    QD_HIP(c, hipMemcpyAsync(c->hpin, c->dscal + QD_S_MED_OUT, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    QD_HIP(c, hipStreamSynchronize(c->stream));
    *out = c->hpin[0];
    return 0;
}

extern "C" int qd_median_state(qd_handle c, double* out64) {
    if (!c || !out64) return -1;
    hipSetDevice(c->desc.device);
    for (int k = 0; k < 16 * 4; ++k) out64[k] = 0.0;
    if (!c->med_pred) return 0;
    QD_HIP(c, hipStreamSynchronize(c->stream));
    QD_HIP(c, hipMemcpy(out64, c->med_pred, sizeof(double) * 16 * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int qd_reduce(qd_handle c, int field, int op, double* out) {
    if (!c || !out) return -1;
    if (field < 0 || field >= QD_F_COUNT_F64) return qd_fail(c, "qd_reduce: unknown field");
    hipSetDevice(c->desc.device);
    double v = 0;
    const int op1 = (op == QD_R_COSWEIGHTED_MEAN) ? 1 : (op == QD_R_SUM ? 0 : (op == QD_R_MAX ? 2 : (op == QD_R_MIN ? 3 : 4)));
    if (qd_reduce_field(c, c->f[field], op1, &v)) return -1;
    if (op == QD_R_COSWEIGHTED_MEAN) v = v / (c->wsum_all + 1e-15);
    *out = v;
    return 0;
}

extern "C" int qd_band_insolation(qd_handle c, int nb, const double* specA, const double* specB, const double* tray, double* out_host) {
    if (!c || !specA || !specB || !tray) return -1;
    hipSetDevice(c->desc.device);
    return qd_band_insolation_impl(c, nb, specA, specB, tray, out_host);
}
extern "C" int qd_energy_diagnostics_last(qd_handle c, double* out) {
    if (!c || !out) return -1;
    for (int k = 0; k < 10; ++k) out[k] = c->last_diag[k];
    return 0;
}
extern "C" int qd_energy_diagnostics(qd_handle c, double* out) {
    if (!c || !out) return -1;
    hipSetDevice(c->desc.device);
    return qd_energy_diag_impl(c, out);
}

extern "C" int qd_copy_ceiling(qd_handle c, size_t bytes, int reps, double* gbs) {
    if (!c || !gbs || bytes == 0 || reps < 1) return -1;
    hipSetDevice(c->desc.device);
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { if (a) hipFree(a); return qd_fail(c, "qd_copy_ceiling: hipMalloc"); }
    hipMemsetAsync(a, 1, bytes, c->stream); hipMemsetAsync(b, 2, bytes, c->stream);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream);            // warm-up
    hipEventRecord(e0, c->stream);
    for (int r = 0; r < reps; ++r) hipMemcpyAsync((r & 1) ? a : b, (r & 1) ? b : a, bytes, hipMemcpyDeviceToDevice, c->stream);
    hipEventRecord(e1, c->stream);
    hipStreamSynchronize(c->stream);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(a); hipFree(b);
    *gbs = ms > 0.f ? (2.0 * (double)bytes * reps / 1e9) / ((double)ms / 1e3) : 0.0;
    return 0;
}

// ------------------------------------------------------------------ timing
extern "C" int qd_timing_enable(qd_handle c, int on) { if (!c) return -1; c->timing = on ? 1 : 0; return 0; }
extern "C" int qd_timing_select(qd_handle c, const char* name) {
    if (!c || !name) return -1; c->timing = 2; c->timing_sel = name; return 0;
}
extern "C" int qd_timing_reset(qd_handle c) { if (!c) return -1; qd_resolve_timers(c); c->timers.clear(); c->timing_seen.clear(); return 0; }
extern "C" int qd_timing_get(qd_handle c, const char* name, double* mean_ms, int64_t* launches) {
    if (!c || !name) return -1;
    hipSetDevice(c->desc.device);
    qd_resolve_timers(c);
    auto it = c->timers.find(name);
    if (it == c->timers.end() || it->second.n == 0) { if (mean_ms) *mean_ms = 0; if (launches) *launches = 0; return 0; }
    if (mean_ms) *mean_ms = it->second.total_ms / (double)it->second.n;
    if (launches) *launches = it->second.n;
    return 0;
}
