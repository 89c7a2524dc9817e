// qd_internal.h -- shared definitions for libqingdai_hip.so (gfx950 only).
//
// Data layout in HBM: every field is a C-order float64 slab [n_rows + 2*halo][n_lon]
// holding this handle's latitude band (global rows row0 .. row0+n_rows-1) plus `halo`
// rows on each side.  Local row l <-> global row g = row0 - halo + l.  Single GPU:
// row0 = 0, n_rows = n_lat, halo = 0 and the slab is the whole grid.
// Per-row metrics (cos floors, Coriolis, k4 maps) are 1-D tables indexed by GLOBAL row.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <functional>
#include <map>
#include <unordered_map>
#include <initializer_list>
#include <climits>
#include <cfloat>
#include <cmath>

#include "../../include/qingdai_hip.h"

#define QD_NSCRATCH 16
#define QD_MAX_DEVICES 64     // per-device one-time kernel attributes (hipFuncSetAttribute is per device)
#define QD_PAD_ROWS 8        // rows of slack behind every slab (see qd_create)
#define QD_BLOCK 256
#define QD_MAXF 5            // fields per batched stencil launch
#define QD_HIST_BINS 2048    // 11-bit radix-select digit

struct QdGeom {
    int nlat, nlon;          // global grid
    int row0, nrows, halo;   // rows [row0, row0+nrows) are the rows a launch computes (a band or a segment of it)
    int full;                // 1: band == whole globe (row wrap/fold resolved by index arithmetic)
    int lbase;               // global row held by local row 0 (= owned row0 - halo; may be negative: wraps)
    int lrows_;              // local rows of the slab (owned + 2*halo)
    __host__ __device__ inline int lrows() const { return lrows_; }
    __host__ __device__ inline size_t cells() const { return (size_t)lrows() * (size_t)nlon; }
};

// 1-D metric tables (device pointers).  All indexed by global row, except lon_*.
struct QdTabs {
    const double* cos_raw;   // cos(deg2rad(lat))
    const double* sin_raw;   // sin(deg2rad(lat))
    const double* cos6;      // max(cos, 1e-6)   dynamics.py:104,490  grid.py:50
    const double* cos3;      // max(cos, 1e-3)   dynamics.py:559
    const double* cos02;     // max(cos, 0.2)    dynamics.py:164
    const double* cos05;     // max(cos, 0.5)    ocean.py:82, run_simulation.py:1145
    const double* fcor;      // 2 Omega sin(lat) grid.py:90-96
    const double* warea;     // max(cos,0)       energy.py:520, ocean.py:372
    const double* r_extra;   // polar sponge gain * s^2   ocean.py:332-334
    // reciprocal-form coefficients of the fused (fast) kernels, per cos-floor kind k = 0 (0.2), 1 (0.5):
    //   lapA[k][r] = cos_k[r]/(2 dphi)   lapP[k][i] = 1/(a^2 cos_k[i] 2 dphi)   lapQ[k][i] = 1/(a^2 dlam^2 cos_k[i]^2)
    const double* lapA[2];
    const double* lapP[2];
    const double* lapQ[2];
    const double* lapPoleA[2];   // [8]: (Aa, Ab) of the pole-row types g = 0, 1, n-2, n-1; lapP holds their P
    const double* lapK[2];       // [nlat][4] packed per row: lapA[r-1], lapA[r+1], lapP[r], lapQ[r] (one s_load_dwordx8; qd_stream.hip)
    const double* mom_cu;    // -(g / (f_safe a cos6))   geostrophic u_g coefficient
    const double* mom_cv;    //   g / (f_safe a)         geostrophic v_g coefficient
    const double* mom_px;    // -(g / (a cos6))          primitive PGF_x coefficient
    const double* ocn_igx;   // 1 / (a cos05)
    const double* inv_acos6; // 1 / (a cos6)   (the `pre` factor of grid.py:41-88 divergence / vorticity)
    const double* lat_deg;   // np.linspace(-90, 90, n_lat)
    const double* lon_rad;   // deg2rad(lon) [n_lon]
    const double* sin_lon;   // [n_lon]
    const double* cos_lon;   // [n_lon]
};

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ double qd_nn(double x) {          // np.nan_to_num
    if (x != x) return 0.0;
    if (x > DBL_MAX) return DBL_MAX;
    if (x < -DBL_MAX) return -DBL_MAX;
    return x;
}
__device__ __forceinline__ double qd_max(double a, double b) {   // np.maximum (NaN propagates)
    return (a >= b || a != a) ? a : b;
}
__device__ __forceinline__ double qd_min(double a, double b) {   // np.minimum
    return (a <= b || a != a) ? a : b;
}
__device__ __forceinline__ double qd_clip(double x, double lo, double hi) {  // np.clip
    return qd_min(qd_max(x, lo), hi);
}
__device__ __forceinline__ double qd_pow4(double x) { double x2 = x * x; return x2 * x2; }

// local row of global row g; in `full` mode rows outside [0,nlat) wrap with period nlat
// (np.roll(axis=0) semantics); in band mode they address the halo.
__device__ __forceinline__ int qd_lrow(const QdGeom& G, int g) {
    int l = g - G.lbase;                       // full mode: lbase = 0
    if (l < 0) l += G.nlat; else if (l >= G.nlat) l -= G.nlat;
    return l;
}
// local row of a global row that is already inside [0, nlat) (advection departure rows):
// band mode lets the polar bands reach the opposite pole through their period-nlat halo.
__device__ __forceinline__ int qd_lrow_far(const QdGeom& G, int g) {
    int l = g - G.lbase;
    if (l < 0) l += G.nlat; else if (l >= G.nlat) l -= G.nlat;
    if (l >= G.lrows()) l = G.lrows() - 1;     // outside the halo: never fault (band runs size the halo for the reach)
    return l;
}
__device__ __forceinline__ int qd_wrapc(int j, int n) {        // periodic column
    return j < 0 ? j + n : (j >= n ? j - n : j);
}

// XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs (block b and b+8
// share an L2); a row-segment stencil wants the rows it re-reads (i-2..i+2) in the SAME L2, so
// the linear block id is remapped such that every XCD walks one contiguous chunk of
// (field, row, segment) work items.  Pure performance: any placement gives the same result.
struct QdTile { int seg, row, fld; };
__device__ __forceinline__ QdTile qd_tile() {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned nb = gx * gy * gridDim.z;
    const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned per = nb >> 3, rem = nb & 7u;
    const unsigned x = L & 7u, s = L >> 3;
    const unsigned w = x * per + (x < rem ? x : rem) + s;
    QdTile t;
    t.seg = (int)(w % gx);
    const unsigned r = w / gx;
    t.row = (int)(r % gy);
    t.fld = (int)(r / gy);
    return t;
}

// ---------------------------------------------------------------- host context
struct QdUse { void** slot; int radius; int u8; };
// ecology sub-step: clocks and cache flags of PopulationManager / EcologyAdapter / IndividualPool (qd_eco.hip)
#define QD_MAXBANDS 32
// phytoplankton tracer stack (qd_phyto.hip): stack[0] holds the S species, stack[1] the advected intermediate
// (cur / tmp are members, not heap storage: the in-process band transport finds a peer's slab at the same offset in ITS context)
#define QD_MAX_SPECIES 64
struct QdPhyto { int S = 0; double* stack[2] = {nullptr, nullptr}; size_t stride = 0; double* cur[QD_MAX_SPECIES] = {nullptr}; double* tmp[QD_MAX_SPECIES] = {nullptr}; double K_h = 5.0e3, alpha = 0.7; };

struct QdEco {
    qd_eco_params p{0.5, 0.3, 0.20, 1.0, 6.0, 0.05, 1, 0, 0, 0, 1, 0};
    int configured = 0;
    double hours = 0, next_h = 6.0;
    int64_t count = 0;
    int have_lai = 0, f_valid = 0, alpha_valid = 0, alpha_dirty = 1, banded_valid = 0, water_valid = 0;
    int lai_version = 0, snap_version = 0, n_recompute = 0;
    double eday_dt = 0;              // > 0: the next forcing launch adds nan_to_num(isr) * eday_dt to ECO_EDAY
    // individuals
    int n_cells = 0, n_indiv = 0, nb = 0, k_per_day = 10;
    int32_t *sample_j = nullptr, *sample_i = nullptr, *cell = nullptr;
    double *Ab = nullptr;            // [nb][n_indiv] (band-major: consecutive individuals are consecutive in memory)
    float *Ab32 = nullptr;           // the same table stored as f32 (qd_indiv_configure ab_f32: BASELINE configs[4] "f32 mixed")
    double *tol = nullptr, *E_day = nullptr, *stress = nullptr;
    double specA[QD_MAXBANDS], specB[QD_MAXBANDS], tray[QD_MAXBANDS];
    double day_seconds = 0, soil_cap = 50.0, period = -1.0, accum = 0;
    int64_t n_fired = 0;
};

struct QdTimer { double total_ms = 0; int64_t n = 0; };
// Tuning switches of the launchers, read from the environment ONCE in qd_create (qd_read_tuning): nothing that qd_step_n reaches
// calls getenv.  0 / -1 = "pick per launch from the grid".
struct QdTune {
    int tail_r = 0;           // QD_TAIL_R: strip height of the ocean tail kernel
    int tail_rp = 3;          // QD_TAIL_RP: rows of its pole strips (>= 3)
    int tail_v = 0;           // QD_TAIL_V=1: k_ocn_tail_stream (the round-3 general form) instead of k_ocn_tail_fast
    int tail_general = 0;     // QD_TAIL_GENERAL=1: every wave of k_ocn_tail_fast takes the general form (A/B runs, tests)
    int stream_r_dyn = 0, stream_r_ocn = 0;   // QD_STREAM_R_DYN / _OCN (QD_STREAM_R for both is qd_ctx::stream_rows)
    int stream_vb = -1;       // QD_STREAM_VB: rows the north-pole strip is shorter by
    int med_blocks = 256;     // QD_MED_BLOCKS: fat workgroups of the median passes
    int shapiro_r = 0;        // QD_SHAPIRO_R: strip height of k_shapiro_stream
    int tile_tr = 0;          // QD_TILE_TR: tile height of the LDS fallback kernels
    int fused_r = 0;          // QD_FUSED_R: strip height of k_ocn_fused
    int fused_norot = 0;      // QD_FUSED_NOROT=1: no rotation of the wave roles (A/B)
    int stream_no_pair = 0;   // QD_STREAM_NO_PAIR=1: the two boundary segments of a split ocean momentum launch as two launches (A/B)
    int fused_seq = 0;        // QD_FUSED_SEQ=1: every strip of k_ocn_fused takes its sequential form (tests)
};
struct QdTileShape { int tr, tc, ntr, ntc; };

struct qd_ctx {
    qd_grid_desc desc;
    QdGeom geo;
    qd_params p;
    hipStream_t stream = nullptr;
    double* f[QD_F_COUNT_F64] = {nullptr};
    uint8_t* land = nullptr;
    uint8_t* icemask = nullptr;
    double* scratch[QD_NSCRATCH] = {nullptr};
    // metric tables
    std::vector<double*> tab_alloc;
    QdTabs tabs;
    double dlat = 0, dlon = 0;
    double* k4_atm = nullptr;      // [5][nlat] rows: u,v,h,q,cloud
    double* k4_ocn = nullptr;      // [3][nlat] rows: uo,vo,eta
    double k4_atm_dt = -1, k4_ocn_dt = -1;
    int k4_atm_skip[5] = {0}, k4_ocn_skip[3] = {0};
    // reductions / scalars
    double* red_partial = nullptr; // [4][max_blocks]
    int red_blocks = 0;
    double* dscal = nullptr;       // device scalars (see QD_S_*)
    unsigned long long* dcount = nullptr;  // device counters
    unsigned int* hist = nullptr;  // [2][QD_HIST_BINS]
    unsigned long long* sel_state = nullptr; // radix-select state
    double* bands = nullptr;         // [bands_nb][cells] band insolation planes (qd_band_insolation), allocated on first use
    int bands_nb = 0;
    double last_diag[10] = {0};      // energy-budget means taken inside qd_step_n (flags bit 4)
    int has_elevation = 0;           // an ELEVATION map has been uploaded (orographic factor needs one)
    QdPhyto phyto;                   // resident tracers of PhytoManager.advect_diffuse
    QdEco eco;                       // ecology sub-step state
    double* zonal_tw = nullptr;      // [2][nlon] cos / sin(2 pi m / nlon) of the zonal spectral filter
    double* sel_cand = nullptr;      // [2][cells] candidates of the two middle ranks after two radix passes (whole-globe handles)
    unsigned int* sel_ccount = nullptr; // [2] candidate counts
    double* med_pred = nullptr;      // [4 sites][16]: last median, valid flag, statistics, published bracket (see qd_reduce.hip): predicted median brackets (qd_reduce.hip)
    int med_predict = 1;             // QD_MEDIAN_PREDICT=0: always the two-histogram-pass select
    int med_seen[4] = {0, 0, 0, 0};  // call sites that have a window centre on the device
    double* pending_sum = nullptr;   // a deferred one-double all-reduce (qd_allreduce_sum_deferred)
    int group_sums = 1;              // QD_GROUP_SUMS=0: issue every eta sum as its own collective
    long grouped_sums = 0;           // deferred sums that went out inside a halo exchange's group
    int band_tail = 1;               // QD_BAND_TAIL=0: latitude bands keep the round-2 sub-step (k_cont_sstadv + k_eta_mean + k_sst_outlier_fused)
    double* qt_tab = nullptr;        // packed row table of k_ocn_tail_fast (built at its first launch for qt_tab_a = params.a)
    double qt_tab_a = 0.0;
    int hoist_precip = 1;            // QD_HOIST_PRECIP=0: qd_step_n keeps the next step's precipitation block behind the ocean step
    int precip_done = 0;             // the precipitation block of the next driver-physics call has already run (qd_step_n)
    // QD_SIDE_STREAM=1 (whole-globe handles): that hoisted block runs on a SECOND stream, concurrently with the ocean sub-steps, instead
    // of in front of them (qd_ocean.hip: fork after the stress kernel; qd_side_join: before its first consumer)
    // whole-globe qd_step_n: time_step's last kernel (k_final: cloud gather + damp + scrub) is not launched by qd_atmos_step_impl but
    // left here for the ocean step, whose first launch does it together with the wind stress, the CFL row maxima and Q_net
    // (k_final_qnet_stress: the three were 17 + 12 + 15 us and read each other's u, v, T_s, h again)
    // lazy diagnostics (QD_LAZY_DIAG, default 1): inside a qd_step_n span only the LAST step stores the fields that nothing inside a span
    // reads (P_rain, S_next, melt, C_snow, glacier, isr_A, isr_B, E, LH_release, OLR: 83 MB per step at 721 x 1440) -- unless the span's
    // flags bring a reader (hydrology commit, ecology, tracers).  diag_write is what the kernels see.
    int lazy_diag = 1, diag_write = 1;
    // whole-globe qd_step_n: k_snow_albedo_forcing (the driver physics' last launch) also computes P_cond = phase 1 of the column
    // (it reads h, h_ice, land anyway; k_column<1> was a launch of 14 us that read them again), time_step then starts with the median
    int merge_pcond = 1;             // QD_MERGE_PCOND=0: k_column<1> stays a launch of its own
    // QD_MED_SIDE=1: k_column<1> + the P_cond median (four launches of one workgroup per CU or less: latency chains, not bandwidth) run on
    // the SIDE stream beside the driver physics' launches instead of between them and time_step's column (pcond_ahead = 2: the median
    // is done too; qd_atmos_step_impl joins).  Its median has buffers of its own (qd_reduce.hip).
    int med_pair = 1;                // QD_MED_PAIR=0: the precipitation median and the P_cond median as two chains of three launches instead of one
                                     // (whole-globe qd_step_n: k_column<1> moves in front of the cloud block, pcond_ahead = 3)
    int med_fold = 0;                // QD_MED_FOLD=1 (experiment, slower): no k_column<1> launch -- the pair's histogram pass computes, stores and bins P_cond (k_med_hist2p)
    int med_side = 0;
    bool med_side_active = false;    // set around the side median's qd_median_positive_dev call
    hipEvent_t med_fork = nullptr, med_done = nullptr;
    unsigned int* hist_b = nullptr; unsigned long long* sel_state_b = nullptr; double* sel_cand_b = nullptr; unsigned int* sel_ccount_b = nullptr;
    int want_pcond_ahead = 0;        // set by qd_step_n around the driver physics
    int pcond_ahead = 0;             // the physics launch did write it: qd_atmos_step_impl skips k_column<1>
    // whole-globe k_ocn_tail_fast does not store uo'' / vo'': changed cells go through a list and are patched in place (qd_ocntail.h)
    int tail_fix = 1;                // QD_TAIL_FIX=0: uo'' / vo'' stored to slabs of their own as before
    unsigned int* fix_count = nullptr;           // [0] entries of the launch in flight, [4] entries / [5] launches since the last publish, [6] longest list
    unsigned long long* fix_list = nullptr;      // [cells][3]
    double fix_avg = 0.0;                        // entries per launch in the last step that used the list (k_max2_publish / the bands' CFL reduce)
    double fix_dense = 256.0;                    // QD_TAIL_FIX_DENSE: longer lists on average -> the storing form
    int64_t fix_probe_at = 0;                    // while the storing form is on: the ocean step that tries the list again ...
    int fix_probe_every = 32;                    // ... 32 steps after the last try, doubling up to 256 while the tries keep finding long lists
    // whole-globe qd_step_n with the P_cond median ahead of the cloud block (pcond_ahead == 3): the driver physics' last launch
    // (k_snow_albedo_forcing) and time_step's column kernel are back to back -- the first becomes the first STAGE of the second
    // (k_saf_column2, qd_atmos.hip: cloud, albedo, isr, Teq in registers, h / h_ice / land read once).  saf_pending: its argument block
    // (a QdSafArgs, qd_saf.h) between qd_driver_physics_impl and qd_atmos_step_impl
    int merge_saf = 1;               // QD_MERGE_SAF=0: two launches
    void* saf_pending = nullptr;
    int defer_final = 0;             // set by qd_step_n around qd_atmos_step_impl
    int merge_final = 1;             // QD_MERGE_FINAL=0: keep the three launches
    struct { int on = 0; double dt = 0, decay = 0, dfac = 0; } final_pending;
    double* wgmax = nullptr;         // [2][n_wgmax]: per-workgroup CFL maxima of k_final_qnet_stress (k_max2_publish reduces them)
    int n_wgmax = 0;
    int side_stream_on = 0;
    hipStream_t side_stream = nullptr;
    hipEvent_t side_fork = nullptr, side_done = nullptr;
    bool side_pending = false;       // the side stream holds work the main stream has not waited for yet
    double* red_partial_b = nullptr; // the block's own row partials while it runs beside the ocean step (which owns red_partial)
    std::function<int()> before_cfl_wait;   // whole-globe ocean step: queued after the stress kernel, before the host waits for the CFL maxima
    int merge_pointwise = 1;         // QD_MERGE_POINTWISE=0: every pointwise stage of qd_step_n as a launch of its own
    double* med_gather = nullptr;    // band handles: [world][4 + 4092] gathered candidate segments of the windowed median
    double* hpin = nullptr;        // pinned host scalars
    double* hpin_rows = nullptr;   // pinned, 2 x slab rows: per-row partial maxima read back in one copy
    double wsum_ocean = 0, wsum_all = 0;
    int64_t atm_counter = 0, ocn_counter = 0;
    QdTileShape tile{0, 0, 0, 0};   // fused-kernel tile (qd_pick_tile)
    int use_fused = 1;              // QD_FUSED=0 selects the unfused reference-order kernels
    int fused_fast = 1;             // QD_FUSED_FAST: 1 row-streaming kernels (qd_stream.hip), FAST variant with per-wave EXACT fallback; 2 the same kernels,
                                    // every wave EXACT; 0 LDS-tiled kernels (qd_fused.hip), every tile EXACT; 3 LDS-tiled kernels with their FAST path
    int tail_acc = 1;               // QD_TAIL_ACC=0: the eta mean of a sub-step from k_eta_mean_tail instead of the tail kernel's own last workgroup
    unsigned long long* eta_acc = nullptr;   // accumulator + tickets of the tail kernel's strip sums (qd_wave.h: QD_ACC_WORDS)
    int shapiro_stream = 1;         // QD_SHAPIRO_STREAM=0: one k_shapiro_pass launch per pass
    int ocn_fused = 0;              // QD_OCN_FUSED=1: the whole ocean sub-step as ONE launch (k_ocn_fused, qd_ocntail.hip)
    int ocn_tail = 1;               // QD_OCN_TAIL=0: continuity + SST + outlier filter as the two launches of qd_ocean.hip
    int stream_rows = 0;            // QD_STREAM_R: strip height of the row-streaming kernels (0 = pick per launch)
    // row-streaming kernels (qd_stream.hip): per-field packed row tables {lapA[r+1], lapP[r], lapQ[r], k4[r]}, [0] atmosphere
    // (5 fields), [1] ocean (3); rebuilt when the k4 tables or the scalar overrides change (qs_key remembers what they hold)
    double* qs_tab[2] = {nullptr, nullptr};
    double qs_key[2][8] = {{NAN, 0, 0, 0, 0, 0, 0, 0}, {NAN, 0, 0, 0, 0, 0, 0, 0}};
    std::vector<double> h_lapK[2];  // host copies of QdTabs::lapK
    std::vector<double> h_k4[2];    // host copies of k4_atm / k4_ocn
    QdTune tune;                     // launcher tuning switches (environment, read once at create)
    int n_cu = 0;                    // compute units of the device (hipDeviceAttributeMultiprocessorCount, qd_create)
    int qs_wgs_per_cu[3] = {0, 0, 0};   // resident workgroups per CU of k_dyn_stream<false>, <true>, k_ocn_stream (occupancy query, cached)
    int cloud_eff_valid = 0;
    int last_nsub = 0;
    // host staging
    void* stage = nullptr; size_t stage_bytes = 0;
    // latitude bands: validity margin (rows beyond the owned band that hold current data) per slab
    std::unordered_map<const void*, int> vm;
    int own_row0 = 0, own_nrows = 0;
    // comm
    void* comm = nullptr;          // RCCL communicator (one process per GPU)
    struct QdLocalGroup* lgroup = nullptr;   // in-process peers on one device (tests of the band logic)
    struct QdPeer* peer = nullptr;           // device-side exchange over the peer mapping (qd_peer.hip, QD_PEER_EXCHANGE)
    std::vector<struct QdUse> split_pending; // slabs of a pushed, not yet unpacked exchange (qd_plan_begin / qd_plan_end)
    std::vector<struct QdUse> corefresh;   // slabs refreshed along with any halo exchange that happens anyway (set around loops)
    int exchanges = 0;             // statistics
    int allreduces = 0;
    double pub_seq = 0.0;          // stamp of the last k_publish_host hand-over (qd_fetch_scalars)
    double eta_seq = 0.0;          // sequence number of the eta sum the host is waiting for (bands with a host ring)
    void* hring = nullptr;         // host ring (shared-memory scalar all-reduce), see qd_band.hip
    void* plansim = nullptr;       // planner simulation (qd_plansim_*): a handle without a device, exchanges are logged
    int host_allreduces = 0;
    // timing
    const char* lap_tag = "k_laplacian";       // timing-group names of the two del^4 kernels
    const char* hyp_tag = "k_hyper_apply";     // (the ocean switches them to ocean_* around its calls)
    int timing = 0;                 // 0 off, 1 all groups, 2 only `timing_sel`
    std::string timing_sel;
    std::map<std::string, long> timing_seen;   // launches seen per selected group (stride sampling)
    std::map<std::string, QdTimer> timers;
    struct Pending { hipEvent_t e0, e1; std::string name; };
    std::vector<Pending> pending;   // recorded, not yet resolved (never synchronises inside a step)
    std::vector<hipEvent_t> ev_free;
    std::string err;
};

// device scalar slots
enum { QD_S_PREF = 0, QD_S_ETA_MEAN, QD_S_MED_OUT, QD_S_PSCALE, QD_S_RENORM, QD_S_PQMEAN, QD_S_TMP0, QD_S_TMP1,
       QD_S_DIAG0 = 16,          // ten energy-budget means (qd_energy_diagnostics)
       QD_S_COUNT = 32 };

extern thread_local std::string g_qd_create_err;

int qd_fail(qd_ctx* c, const char* what, hipError_t e = hipSuccess);
#define QD_HIP(c, call) do { hipError_t _e = (call); if (_e != hipSuccess) return qd_fail((c), #call, _e); } while (0)

struct QdScope {               // optional per-kernel-group timing with hipEvents on the handle's stream
    qd_ctx* c; const char* name; hipEvent_t e0 = nullptr, e1 = nullptr; bool on = false, attach = false;
    bool used = false;          // attach: a launch has taken the events (an early return between the scope and its launch leaves them unrecorded)
    // attach_ = true: the group is ONE launch and the caller hands e0 / e1 to hipExtLaunchKernelGGL (QD_LAUNCH_TIMED), which takes
    // the start / stop time from the dispatch itself (what the rocprofv3 kernel trace reports); otherwise the events are recorded
    // around the group, and the pair also brackets the dispatch gaps on either side (~3.5 us per bracket on this stack)
    QdScope(qd_ctx* c_, const char* n, bool attach_ = false);
    ~QdScope();
};
#define QD_LAUNCH_TIMED(sc, kernel, grid, block, stream, ...) do { \
    if ((sc).on && (sc).attach) { (sc).used = true; hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, (sc).e0, (sc).e1, 0, __VA_ARGS__); } \
    else hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__); } while (0)

// ---- latitude-band planning (qd_band.hip) ----------------------------------------------------
#define QD_IN(ptr, r) QdUse{(void**)&(ptr), (r), 0}
#define QD_IN8(ptr, r) QdUse{(void**)&(ptr), (r), 1}
// makes sure every input slab is valid `radius` rows beyond what the launch will compute; exchanges
// halos when one is not; returns the margin (rows beyond the owned band) the outputs can be computed on
int qd_allreduce_sum_deferred(qd_ctx* c, double* dptr);
int qd_allreduce_flush(qd_ctx* c);
int qd_plan(qd_ctx* c, const QdUse* in, int n, int want = INT_MAX);
inline int qd_plan(qd_ctx* c, std::initializer_list<QdUse> in, int want = INT_MAX) { return qd_plan(c, in.begin(), (int)in.size(), want); }
void qd_mark(qd_ctx* c, std::initializer_list<const void*> out, int margin);
int qd_vm_get(qd_ctx* c, const void* slab);
struct QdSegs { QdGeom g[3]; int n; };
QdSegs qd_segments(qd_ctx* c, int margin);
int qd_exchange(qd_ctx* c, const QdUse* slots, int n);
// exchange overlapped with interior rows (peer exchange only): qd_plan_begin is qd_plan, except that a due exchange is only PUSHED
// (*pending = true) when the transport can finish it later; the caller launches the rows it can compute from the OLD margins
// (qd_plan_peek), calls qd_plan_end (wait + unpack, margins back to the halo width) and launches the remaining boundary rows
struct QdSegList { QdGeom g[6]; int n; };
QdSegList qd_segments_rows(qd_ctx* c, int vr0, int cnt, QdSegList S = QdSegList{{}, 0});   // rows [vr0, vr0 + cnt) of the ring (period n_lat), appended to S
int qd_side_join(qd_ctx* c);
bool qd_median_pair_ready(const qd_ctx* c, int site0, int site1);
struct QdColP;
int qd_median_pair_pcond_dev(qd_ctx* c, const double* x0, double dflt0, int slot0, int tr0, double tp0, int site0,
                             const QdColP& P, double dflt1, int slot1, int site1);      // ... with time_step's P_cond produced by the histogram pass
struct QdSafArgs;
int qd_saf_launch(qd_ctx* c, const QdSafArgs& K);  // qd_physics.hip: k_snow_albedo_forcing on the handle's stream
void qd_saf_drop(qd_ctx* c);                       // qd_physics.hip: forget a pending launch (error paths, destroy)
int qd_saf_flush(qd_ctx* c);                       // qd_physics.hip: a pending launch of it that nobody merged: now (no-op when none)
int qd_pcond_phase1(qd_ctx* c, double dt);        // qd_atmos.hip: k_column<1> on the handle's stream (whole globe)
int qd_pcond_median_side(qd_ctx* c, double dt);   // qd_atmos.hip: k_column<1> + the P_cond median on the side stream (fork here, join in qd_atmos_step_impl)         // the main stream waits for what the side stream holds (no-op when nothing is pending)
int qd_plan_begin(qd_ctx* c, const QdUse* in, int n, bool* pending);
int qd_plan_end(qd_ctx* c);
int qd_plan_peek(qd_ctx* c, const QdUse* in, int n);
int qd_allreduce_f64(qd_ctx* c, double* dptr, int n, int op);          // op 0 sum, 1 max (device scalars)
int qd_allreduce_u32(qd_ctx* c, unsigned int* dptr, int n);            // sum
int qd_allreduce_fetch(qd_ctx* c, double* dptr, int n, int op, double* hdst);   // all-reduce + hand-over to pinned host memory (the call returns when the host can read hdst)
int qd_allgather_f64(qd_ctx* c, double* buf, int n_per_rank);          // buf[world][n_per_rank]: own segment in, every segment out
bool qd_peer_on(const qd_ctx* c);                                      // qd_peer.hip: the device-side exchange carries this handle's traffic
int qd_host_allreduce(qd_ctx* c, double* host_vals, int n, int op);    // op 0 sum, 1 max; HOST scalars through the host ring
bool qd_has_host_ring(const qd_ctx* c);
extern "C" int qd_hostring_close(void* ring);
void qd_comm_release(qd_ctx* c);                         // qd_band.hip: ncclCommDestroy, host ring, in-process group
#define QD_ROWS(c, margin, G, ...) do { QdSegs _sg = qd_segments((c), (margin)); for (int _k = 0; _k < _sg.n; ++_k) { const QdGeom& G = _sg.g[_k]; __VA_ARGS__; } } while (0)

static inline dim3 qd_grid2d(const QdGeom& G, int fields = 1) {
    return dim3((G.nlon + QD_BLOCK - 1) / QD_BLOCK, G.nrows, fields);
}

// ---- module entry points (host side launchers) ------------------------------------
struct QdFieldList {
    const double* in[QD_MAXF];
    double* out[QD_MAXF];
    const double* aux[QD_MAXF];   // e.g. original F for the hyper-apply pass
    const double* k4row[QD_MAXF]; // per-row k4 (global row index) or nullptr
    double k4s[QD_MAXF];
    int n;
};

// qd_stencil.hip
// `m` / `m_out`: margin (rows beyond the owned band) the launch / the final result is computed on;
// 0 for whole-globe handles.  Callers plan the inputs with qd_plan first.
void qd_launch_laplacian(qd_ctx* c, const QdFieldList& fl, const double* coslat, int m);
void qd_launch_hyper_apply(qd_ctx* c, const QdFieldList& fl, const double* coslat, double sub_dt, int m);
int  qd_hyperdiffuse_fields(qd_ctx* c, double** fields, int n, const double* k4tab, const int* skip,
                            const double* k4s_override, double dt, int nsub, const double* coslat, int m_out);
void qd_launch_shapiro_pass(qd_ctx* c, const QdFieldList& fl, int scrub, int m);
int  qd_shapiro_fields(qd_ctx* c, double** fields, int n, int npass, int m_out);
void qd_launch_advect(qd_ctx* c, const double* u, const double* v, const double* coslat, double dt,
                      const double* f0, double* o0, const double* f1, double* o1, double alpha, int clipq, int m);
void qd_launch_divvort(qd_ctx* c, const double* u, const double* v, double* out, int vort, int m);
int  qd_gaussian(qd_ctx* c, const double* in, double* out, double* tmp, double sigma, int mode_wrap, int m_out, int clip01 = 0,
                 const double* scale_p = nullptr, double scale_k = 1.0);
bool qd_gauss_pair_ok(const qd_ctx* c, double sigma);
int  qd_gaussian_pair(qd_ctx* c, const double* inA, const double* inB, double sigma, int mode_wrap, const double* scA_p, double scA_k,
                      double scB_k, const double* sc, int op, const double* blend5, double* outA, double* outB, double* out0, int m_out = 0);
int  qd_gaussian_swap(qd_ctx* c, double*& field, double*& tmp, double sigma, int mode_wrap, int m_out, int clip01 = 0,
                      const double* scale_p = nullptr, double scale_k = 1.0);
bool qd_gauss_can_fuse(const qd_ctx* c, double sigma);
int  qd_gauss_radius(double sigma);
int  qd_energy_diag_impl(qd_ctx* c, double* host_out);    // qd_ocean.hip
int  qd_band_insolation_impl(qd_ctx* c, int nb, const double* specA, const double* specB, const double* tray, double* out_host);   // qd_physics.hip
int  qd_zonal_filter_fields(qd_ctx* c, double** fields, int nf, double cutoff, double damp, int m);
int  qd_adv_reach(const qd_ctx* c, double dt, double vmax);

// qd_eco.hip
int  qd_eco_canopy_impl(qd_ctx* c, double dt);            // clock + recompute policy + alpha map (no E_day)
int  qd_eco_eday_impl(qd_ctx* c, double dt);              // E_day += nan_to_num(ISR) dt as its own launch
int  qd_indiv_substep_impl(qd_ctx* c, double dt, int* fired);
void qd_eco_free(qd_ctx* c);
int  qd_band_copy_in(qd_ctx* c, void* dst, const void* host, size_t esz);   // qd_api.hip
int  qd_phyto_step_impl(qd_ctx* c, double dt);                              // qd_phyto.hip
void qd_phyto_release(qd_ctx* c);
bool qd_eco_is_f32(const qd_ctx* c, int field);                             // qd_eco.hip: slab stored as f32 (qd_eco_params.map_f32)
void qd_eco_convert_slab(qd_ctx* c, const double* src, int src_f32, double* dst, int dst_f32);

// qd_reduce.hip
int qd_reduce_field(qd_ctx* c, const double* x, int op, double* host_out);
// site: call-site id 0..3 (keeps a predicted bracket per site on whole-globe handles), -1 = no prediction
int qd_median_positive_dev(qd_ctx* c, const double* x, double dflt, int slot, int transform, double tparam, int site = -1);
int qd_median_pair_dev(qd_ctx* c, const double* x0, double dflt0, int slot0, int tr0, double tp0, int site0,
                       const double* x1, double dflt1, int slot1, int tr1, double tp1, int site1);      // two medians in one set of three launches

// qd_atmos.hip
int qd_atmos_step_impl(qd_ctx* c, double dt, int has_albedo);
int qd_forcing_impl(qd_ctx* c, const double* sa, const double* sb, double theta, int with_teq);
int qd_simple_albedo_impl(qd_ctx* c, double ocean_albedo);

// qd_ocean.hip
int qd_ocean_step_impl(qd_ctx* c, double dt, int compute_qnet, int use_ice_mask, int inject_sst);

// qd_physics.hip
// fc != nullptr (whole-globe handles): the forcing of the same step (stars, rotation angle) rides on the last launch of the physics
struct QdForcingCall { const double* sa; const double* sb; double theta; };
// part 0: everything; 1: only the precipitation block (divergence median, P_raw, the two blurs, the blend); 2: everything after it
int qd_driver_physics_impl(qd_ctx* c, double dt, const QdForcingCall* fc = nullptr, int part = 0);
int qd_hydrology_commit_impl(qd_ctx* c, double dt);

// qd_api.hip
int qd_wait_host_nonneg(qd_ctx* c, volatile double* slot, double* out, const char* what);
int qd_wait_host_flag(qd_ctx* c, volatile double* flag, double seq, const char* what);   // spin on a stamp in pinned host memory
int qd_fetch_scalars(qd_ctx* c, const double* dsrc, int n, double* hdst);                // n <= 32 device doubles -> pinned host memory, no stream sync
int qd_build_k4_tables(qd_ctx* c, double dt, bool ocean, double sub_dt);
double* qd_scratch(qd_ctx* c, int i);
void qd_swap(qd_ctx* c, int field, int scratch_idx);
