/*
 * include/qingdai_hip.h -- C-ABI of libqingdai_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the per-timestep lat-lon grid update of PyGCM-for-Qingdai.
 * The reference has no FFI for this path: its seam is Python level
 *  --
 *   (1) the class surface  pygcm/dynamics.py:22-23,260  SpectralModel(...).time_step(Teq, dt, albedo=None)
 *                          pygcm/ocean.py:28-34,265     WindDrivenSlabOcean(...).step(dt, u, v, Q_net, ice_mask)
 *                          + the free functions of physics/energy/humidity/forcing the driver calls
 *                            (scripts/run_simulation.py:25,36), and
 *   (2) the operator seam  pygcm/jax_compat.py:111,135,190
 *                          laplacian_sphere / hyperdiffuse / advect_semilag gated by is_enabled().
 * Every entry point below names the reference interface it replaces.  Plain C:
 * handles, pointers, sizes; no C++ or torch types.  All arrays are C-order
 * float64 [n_lat][n_lon] (uint8 for masks) in HOST memory, borrowed for the call.
 * The library owns all device memory.  Return 0 = OK, negative = error (see
 * qd_last_error); nothing throws or aborts.  One HIP stream per handle; a handle
 * is not thread-safe.  Steps are asynchronous: qd_download / qd_sync / qd_reduce
 * synchronise.
 */
#ifndef QINGDAI_HIP_H
#define QINGDAI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QD_ABI_VERSION 1

typedef struct qd_ctx* qd_handle;

/* Grid + latitude-band descriptor.  Single GPU: row0 = 0, n_rows = n_lat, halo = 0.
 * Multi GPU (one process per GPU): rows [row0, row0 + n_rows) of the global grid are
 * owned; `halo` extra rows are kept on each side and refreshed by qd_halo_* (RCCL). */
typedef struct qd_grid_desc {
    int32_t n_lat, n_lon;
    int32_t row0, n_rows, halo;
    int32_t device;            /* HIP device ordinal */
    int32_t rank, world;       /* band index / number of bands */
} qd_grid_desc;

/* Field ids: one per array attribute of the reference classes (+ static maps). */
enum qd_field {
    /* SpectralModel prognostic state (dynamics.py:56-66,84) */
    QD_F_U = 0, QD_F_V, QD_F_H, QD_F_TS, QD_F_Q, QD_F_CLOUD, QD_F_HICE,
    /* per-step inputs assigned by the driver (run_simulation.py:1943-1944,2191) */
    QD_F_ISR, QD_F_ISR_A, QD_F_ISR_B, QD_F_TEQ, QD_F_ALBEDO,
    /* diagnostics written by time_step (dynamics.py:294-297,353,411) */
    QD_F_OLR, QD_F_EFLUX, QD_F_PCOND, QD_F_LH, QD_F_LHREL, QD_F_CLOUD_EFF,
    /* static maps (dynamics.py:25-31; topography.py:295-346) */
    QD_F_FRICTION, QD_F_CSMAP, QD_F_BASE_ALBEDO, QD_F_ELEVATION,
    /* WindDrivenSlabOcean state + forcing (ocean.py:86-94,265-270) */
    QD_F_UO, QD_F_VO, QD_F_ETA, QD_F_SST, QD_F_QNET,
    /* driver-side diagnostics (run_simulation.py:1778,1876,1884,2144) */
    QD_F_PRECIP, QD_F_CLOUD_FROM_P, QD_F_CLOUD_SRC,
    /* land hydrology reservoirs (run_simulation.py:1289-1290; hydrology.py) */
    QD_F_W_LAND, QD_F_S_SNOW, QD_F_C_SNOW,
    /* P019 provisional snow / land bucket working fields (run_simulation.py:1946-2019, 2290-2339) */
    QD_F_S_SNOW_NEXT, QD_F_MELT, QD_F_P_RAIN, QD_F_GLACIER, QD_F_RUNOFF,
    /* ecology, per physics step (population.py:252-294,831-915; adapter.py:140-186; run_simulation.py:2075-2128):
     * total LAI, its snapshot at the last canopy recompute, cached canopy factor f, daily energy buffer, land-only
     * ecology alpha (NaN elsewhere), daily banded alpha, ocean-colour alpha of the phytoplankton coupling */
    QD_F_ECO_LAI, QD_F_ECO_LAI_SNAP, QD_F_ECO_F, QD_F_ECO_EDAY, QD_F_ECO_ALPHA, QD_F_ECO_ALPHA_BANDED, QD_F_WATER_ALPHA,
    QD_F_COUNT_F64,
    /* uint8 masks */
    QD_F_LAND_MASK = 100, QD_F_ICE_MASK = 101
};

/* Every env-derived scalar the reference reads inside the step,
 * as one POD.  NaN in a double = "environment variable unset".  Field names equal
 * qingdai_amd.params.QdParams and oracle/qd_oracle/params.py. */
typedef struct qd_params {
    /* SpectralModel ctor: dynamics.py:22-41 */
    double g, H, tau_rad, greenhouse_factor, a, omega;
    double t_freeze, rho_i, L_f, Cs_ocean, Cs_land, Cs_ice;
    /* humidity.py:58-82 */
    double C_E, rho_a, h_mbl, L_v, p0, ocean_evap_scale, land_evap_scale, ice_evap_scale, tau_cond;
    /* energy.py:55-74; dynamics.py:316-386 */
    double sw_a0, sw_kc, lw_eps0, lw_kc, t_floor, c_sfc;
    double energy_w, rh0, k_q, k_p, pcond_ref, hice_ref, eps_default, ch, cp_a;
    double atm_h, gh_factor_lw, eps_ocean, eps_land, eps_ice, lw_tau0, lw_ktau;
    /* dynamics.py:534-658 */
    double sigma4, k4_u, k4_v, k4_h, k4_q, k4_cloud, spec_cutoff, spec_damp, diff_factor;
    /* ocean.py:49-75,380-443,519-533 */
    double H_ocean, rho_w, cp_w, g_ocean, CD, r_bot, rho_a_ocean, vcap, tau_scale;
    double polar_sponge_lat, polar_sponge_gain, K_h, sigma4_ocean, ocean_cfl, ocean_max_u;
    double ocean_k4_u, ocean_k4_v, ocean_k4_eta, ocean_adv_alpha, ocean_ice_qfac, eta_cap, ts_min, ts_max;
    /* driver physics: run_simulation.py:1605-1613,1777,1866-1934 */
    double D_crit, k_precip, alpha_water, alpha_ice, alpha_cloud, p_betadiv, pq_min, p_blend;
    double pref, cmax, w_mem, w_p, w_src, cloud_from_p_floor, cloud_adv_alpha, cloud_smooth_sigma;
    /* land hydrology + P019 lapse / snow: hydrology.py:27-80, run_simulation.py:1616-1627 */
    double runoff_tau_days, wland_cap_mm, snow_thresh_K, snow_melt_rate_mm_day, snow_t_band_K;
    double snow_ddf_mm_per_k_day, snow_melt_tref_K, swe_ref_mm, swe_max_mm, snow_albedo_fresh;
    double lapse_k_kpm, land_elev_max_m, polar_ice_thick_max_m, polar_lat_thresh, rho_snow, glacier_frac, glacier_swe_mm;
    /* orographic precipitation factor: physics.py:116-161, run_simulation.py:1612-1613,1769-1775 */
    double orog_k;
    /* the driver's own EnergyParams copy, nudged by autotune_greenhouse_params (energy.py:544-579,
     * run_simulation.py:1245-1257,2242-2246); NaN = same as lw_eps0 / lw_kc.  Only the coupling block (Q_net,
     * energy diagnostics) reads them, never time_step. */
    double qnet_lw_eps0, qnet_lw_kc;
    /* integer switches */
    int32_t seaice_enabled, cloud_couple, lw_v2, gh_lock, polar_freeze_fix_s, polar_freeze_fix_n;
    int32_t mom_scheme;        /* 0 geos, 1 primitive (QD_MOM_SCHEME) */
    int32_t diff_enable, filter_type; /* 0 combo, 1 hyper4, 2 shapiro, 3 spectral, 4 other */
    int32_t diff_every, k4_nsub, diff_q, diff_cloud, shapiro_every, shapiro_n, spec_every;
    int32_t ocean_k4_nsub, ocean_diff_every, ocean_shapiro_n, ocean_shapiro_every;
    int32_t ocean_outlier;     /* 0 mean4, 1 clamp */
    int32_t ocean_use_qnet, ocean_polar_fix;
    int32_t p_hybrid_fallback, cloud_advect, use_topo_albedo, has_csmap;
    int32_t snow_melt_mode;    /* 0 degree_day, 1 constant (QD_SNOW_MELT_MODE) */
    int32_t swe_enable, lapse_enable;
    int32_t orog_enable;       /* QD_OROG; takes effect once an ELEVATION field has been uploaded */
} qd_params;

/* ---- lifetime ------------------------------------------------------------------ */
int qd_abi_version(void);
int qd_device_count(void);   /* visible HIP devices (0 when none): what jax_compat.is_enabled() asks of its backend */
/* SpectralModel.__init__ / WindDrivenSlabOcean.__init__ (dynamics.py:22-88, ocean.py:28-97):
 * allocates every field on `desc->device`, fills the reference's initial state
 * (u=v=0, h=H+300 sin^2, T_s=288, q=RH0*q_sat(T_s), ocean at rest, SST=288). */
int qd_create(const qd_grid_desc* desc, const qd_params* params, double q_init_rh, qd_handle* out);
int qd_destroy(qd_handle h);
const char* qd_last_error(qd_handle h);   /* h may be NULL: error of the last failed qd_create */

/* ---- attribute surface: `gcm.u = arr` / `arr = gcm.u` (run_simulation.py:1441-1447,1900,2253) */
int qd_upload(qd_handle h, int field, const void* host_global, size_t bytes);
int qd_download(qd_handle h, int field, void* host_global, size_t bytes);
/* the reference re-reads its env every step (dynamics.py:330-650); callers re-send on change */
int qd_set_params(qd_handle h, const qd_params* params, size_t sizeof_params);
int qd_get_step_counter(qd_handle h, int64_t* atmos_counter, int64_t* ocean_counter);
int qd_set_step_counter(qd_handle h, int64_t atmos_counter, int64_t ocean_counter);

/* ---- the path ------------------------------------------------------------------- */
/* forcing.py:78-103,138-165: isr_A, isr_B, isr (and Teq from the ALBEDO field when with_teq)
 * from ten host scalars: per star (flux, declination, right ascension) + theta (+ sigma). */
int qd_forcing(qd_handle h, const double star_a[3], const double star_b[3], double theta, int with_teq);
/* benchmark_jax.py:129: albedo = where(land == 0, ocean_albedo, base_albedo) */
int qd_simple_albedo(qd_handle h, double ocean_albedo);
/* SpectralModel.time_step(Teq, dt, albedo) (dynamics.py:260-667); Teq/ISR/ALBEDO fields must be current */
int qd_atmos_step(qd_handle h, double dt, int has_albedo);
/* run_simulation.py:2197-2253 + ocean.py:265-533: Q_net from SW/LW/SH/LH, ice mask from h_ice,
 * WindDrivenSlabOcean.step, then T_s <- SST over open ocean.  compute_qnet=0 uses the QNET field
 * and ICE_MASK as uploaded; inject_sst=0 skips the write-back. */
int qd_ocean_step(qd_handle h, double dt, int compute_qnet, int use_ice_mask, int inject_sst);
/* run_simulation.py:1766-1934,2063-2146: hybrid precipitation, cloud-from-precip, cloud source,
 * cloud blend + advection, dynamic albedo (physics.py:12-354). */
int qd_driver_physics(qd_handle h, double dt);
/* run_simulation.py:2290-2339: commit the provisional snowpack, update the land bucket (hydrology.py:219-260) */
int qd_hydrology_commit(qd_handle h, double dt);
/* benchmark_jax.py:124-158 as one resident loop of n steps: forcing -> albedo -> time_step [-> ocean
 * coupling] [-> hydrology commit].  flags bit0 = with_ocean, bit1 = with_driver_physics (else the simple
 * ocean/land albedo of benchmark_jax.py:129), bit2 = pass albedo to time_step, bit3 = hydrology commit, bit4 = energy diagnostics on the first step (qd_energy_diagnostics_last), bit5 = ecology sub-step (qd_eco_substep, and qd_indiv_substep when a pool is configured; needs bit1), bit6 = tracer transport after the ocean step (qd_phyto_advect_diffuse; needs bit0 and qd_phyto_configure).  `stars` holds n rows of 7 host scalars
 * (flux_A, decl_A, ra_A, flux_B, decl_B, ra_B, theta), evaluated by the caller as forcing.py:85-125 does. */
int qd_step_n(qd_handle h, int n, double dt, int flags, const double* stars);
int qd_last_ocean_nsub(qd_handle h, int* n_sub);
int qd_sync(qd_handle h);

/* ---- operator seam: jax_compat.py:111-216 (host in, host out; global arrays) ------ */
/* cos floor kinds: 0 = max(cos,0.2) atmosphere, 1 = max(cos,0.5) ocean */
int qd_op_laplacian(qd_handle h, const double* F, int cos_kind, double* out);
/* k4_row: n_lat per-row coefficients, or NULL with scalar k4 */
int qd_op_hyperdiffuse(qd_handle h, const double* F, const double* k4_row, double k4_scalar, double dt,
                       int n_substeps, int cos_kind, double* out);
/* cos floor kinds: 0 = max(1e-6,cos) atmosphere, 1 = max(cos,0.5) ocean / driver cloud */
int qd_op_advect(qd_handle h, const double* field, const double* u, const double* v, double dt,
                 int cos_kind, double* out);
int qd_op_shapiro(qd_handle h, const double* F, int n, double* out);                /* dynamics.py:215-231 */
int qd_op_zonal_filter(qd_handle h, const double* F, double cutoff, double damp, double* out); /* dynamics.py:233-258 */
int qd_op_divergence(qd_handle h, const double* u, const double* v, double* out);   /* grid.py:41-68 */
int qd_op_vorticity(qd_handle h, const double* u, const double* v, double* out);    /* grid.py:70-88 */
int qd_op_gaussian(qd_handle h, const double* F, double sigma, int mode_wrap, double* out); /* physics.py:44 */
int qd_op_median_positive(qd_handle h, const double* x, double dflt, double* out);  /* dynamics.py:344-348 */
/* (new, diagnostics) the per-call-site state of the three in-step medians (dynamics.py:344-348, run_simulation.py:1740-1747,
 * 1866-1874): 4 sites x 16 doubles {last median, -, -, valid, hits, misses, candidates of the last call, positives of the last
 * call, bracket lo, hi, -, ok, calls, last miss: call, centre, result}; site 1 = P_cond, 2 = convergence, 3 = precipitation */
int qd_median_state(qd_handle h, double* out64);

/* ---- ecology spectral sub-step, first stage (pygcm/ecology/spectral.py:304-426) --------------------
 * dual_star_insolation_to_bands on the resident ISR_A / ISR_B: specA/specB/tray are the NB host-computed band weights of the
 * two stars and the Rayleigh factor; the result [nb][n_lat][n_lon] f64 stays resident and is also copied to `out_host_or_null`. */
int qd_band_insolation(qd_handle h, int nb, const double* specA, const double* specB, const double* tray, double* out_host_or_null);

/* ---- ecology, the per-physics-step part (BASELINE config 5; SURVEY.md 8(f)3 stages 2-4) --------------------------------
 * The daily population dynamics (population.py:389-830, individuals.py:193-361) stay host code that runs once per
 * planet-day; it hands the device the LAI layers and takes the energy buffers back. */
typedef struct qd_eco_params {
    double k_canopy;             /* QD_ECO_LAI_K (0.5): f = 1 - exp(-k max(LAI_tot, 0)), population.py:911-915 */
    double leaf_scalar;          /* adapter.py:61: sum_b R_leaf[b] w_b */
    double soil_ref;             /* QD_ECO_SOIL_REFLECT (0.20), adapter.py:170 */
    double w_lai;                /* QD_ECO_LAI_ALBEDO_WEIGHT (1.0), run_simulation.py:2089-2100 */
    double light_update_hours;   /* QD_ECO_LIGHT_UPDATE_EVERY_HOURS (6), population.py:62-65,900 */
    double recompute_lai_delta;  /* QD_ECO_LIGHT_RECOMPUTE_LAI_DELTA (0.05), population.py:66-69,903-907 */
    int32_t substep_every_nphys; /* QD_ECO_SUBSTEP_EVERY_NPHYS (1), adapter.py:156 */
    int32_t albedo_couple;       /* QD_ECO_SUBDAILY_ENABLE && QD_ECO_ALBEDO_COUPLE: run_simulation.py:2075 */
    int32_t bands_couple;        /* QD_ECO_BANDS_COUPLE: land base albedo <- clip(ECO_ALPHA_BANDED), run_simulation.py:2107-2112 */
    int32_t water_couple;        /* QD_PHYTO_ENABLE && QD_PHYTO_ALBEDO_COUPLE: ocean base albedo <- clip(WATER_ALPHA), :2121-2128 */
    int32_t use_lai;             /* QD_ECO_USE_LAI (1).  0 = the adapter's M1 branch (adapter.py:162-166): no population, no E_day,
                                    alpha = clip(leaf_scalar) on land */
    int32_t map_f32;             /* QD_ECO_F32 (0): store the canopy maps ECO_LAI, ECO_LAI_SNAP, ECO_F, ECO_ALPHA, ECO_ALPHA_BANDED as f32
                                    (arithmetic, the LAI plane sum, the lai-delta reduction and E_day stay f64); qd_upload /
                                    qd_download of these fields still exchange f64 with the host.  Fixed once the maps hold data. */
} qd_eco_params;
int qd_eco_configure(qd_handle h, const qd_eco_params* p, size_t sizeof_params);
/* PopulationManager.total_LAI (population.py:288-294): layers = [n_planes][n_lat][n_lon] host f64 (the flattened
 * [S][K] planes of LAI_layers_SK), summed plane after plane like np.sum(axis=(0,1)) into the resident ECO_LAI.
 * init != 0 also takes the constructor's snapshot (population.py:71). */
int qd_eco_set_lai_layers(qd_handle h, const double* layers, int n_planes, int init);
/* EcologyAdapter.step_subdaily (adapter.py:140-186) on the resident ISR: E_day += nan_to_num(isr) dt, canopy clock and
 * recompute policy (population.py:252-280,895-915), land-only alpha map on sub-step boundaries.  qd_step_n with flags
 * bit5 runs the same inside its loop, where the driver does (run_simulation.py:2075-2104). */
int qd_eco_substep(qd_handle h, double dt);
/* population.get_surface_albedo_bands + the driver's daily reduction (population.py:875-893, run_simulation.py:1843-1844):
 * ECO_ALPHA_BANDED <- clip(nansum_b clip(R_eff[b] f + (1 - f) soil, 0, 1) w_b, 0, 1); nb <= 32 */
int qd_eco_banded_alpha(qd_handle h, int nb, const double* r_eff, const double* w_b);
/* clock state for restarts / inspection: out[0] hours accumulated, [1] next time-based recompute, [2] step count,
 * [3] canopy recomputes so far, [4] 1 when an alpha map is cached */
int qd_eco_get_state(qd_handle h, double out[5]);
int qd_eco_set_state(qd_handle h, const double in[3]);        /* hours, next recompute, step count */

/* IndividualPool (individuals.py:37-191): n_cells sampled land cells (row j, column i), n_indiv individuals each bound to
 * one sampled cell, with a per-band coefficient row Ab[n_indiv][nb] and a drought tolerance; band tables as in
 * qd_band_insolation.  State (E_day, water-stress days) stays resident.  ab_f32 != 0 keeps the coefficient table -- the
 * only large array of the sub-step -- as f32 in HBM (the "f32 mixed precision" of BASELINE configs[4]); all arithmetic and
 * the state stay f64, so results move by the f32 rounding of Ab (<= 6e-8 relative). */
int qd_indiv_configure(qd_handle h, int n_cells, const int32_t* sample_j, const int32_t* sample_i, int n_indiv,
                       const int32_t* cell_index, const double* Ab, const double* tol, int nb, const double* specA,
                       const double* specB, const double* tray, int substeps_per_day, double day_seconds, double soil_cap,
                       int ab_f32);
/* IndividualPool.try_substep (individuals.py:142-191) on the resident ISR_A / ISR_B and the soil index
 * clip(W_LAND / max(1e-6, soil_cap), 0, 1) of run_simulation.py:2025-2033; *fired = 1 when a sub-step was consumed.
 * The band intensities are evaluated per sampled cell and never materialised as [NB][n_lat][n_lon]. */
int qd_indiv_substep(qd_handle h, double dt, int* fired_or_null);
int qd_indiv_download(qd_handle h, double* E_day, double* stress_days);   /* each [n_indiv]; band handles: own cells, 0 elsewhere */
int qd_indiv_upload(qd_handle h, const double* E_day, const double* stress_days);   /* daily reset / restart */

/* ---- phytoplankton tracers carried by the ocean currents (pygcm/ecology/phyto.py:496-547) ------------
 * PhytoManager.advect_diffuse on resident state: C_phyto_s [S][n_lat][n_lon] lives on the device next to the ocean's
 * uo / vo; the driver calls it once per step after the SST write-back (scripts/run_simulation.py:2254-2258).
 * K_h = QD_PHYTO_KH (phyto.py:123), adv_alpha = QD_PHYTO_ADV_ALPHA (phyto.py:517).  n_species = 0 frees the stack. */
int qd_phyto_configure(qd_handle h, int n_species, double K_h, double adv_alpha);
int qd_phyto_upload(qd_handle h, int species, const double* host);      /* [n_lat][n_lon] f64 */
int qd_phyto_download(qd_handle h, int species, double* host);
int qd_phyto_advect_diffuse(qd_handle h, double dt_seconds);             /* all species, three launches */

/* ---- reductions for diagnostics (energy.py:494-538, ocean.py:535-561) -------------- */
/* compute_energy_diagnostics (energy.py:494-538) from the resident state, with the flux formulas of the driver's
 * coupling block (run_simulation.py:2199-2239): out[10] = cos-weighted global means of
 * TOA_net, SFC_net, ATM_net, I, R, OLR, SW_sfc, LW_sfc, SH, LH. */
int qd_energy_diagnostics(qd_handle h, double out[10]);
/* the same ten means as taken INSIDE the last qd_step_n call that had flags bit4 set: on its first step, after time_step and
 * before the ocean step, i.e. exactly where run_simulation.py:2242-2246 evaluates them for the autotuner */
int qd_energy_diagnostics_last(qd_handle h, double out[10]);
enum qd_reduce_op { QD_R_SUM = 0, QD_R_COSWEIGHTED_MEAN = 1, QD_R_MAX = 2, QD_R_MIN = 3, QD_R_MAXABS = 4 };
int qd_reduce(qd_handle h, int field, int op, double* out);

/* ---- multi-GPU: latitude bands, RCCL halo exchange ------------------ */
int qd_comm_unique_id(void* id128, size_t bytes);                 /* rank 0: ncclGetUniqueId */
int qd_comm_init(qd_handle h, const void* id128, size_t bytes);   /* all ranks: ncclCommInitRank */
/* in-process group of band handles on ONE device, one host thread per handle: exercises the band
 * logic (margins, ring halos, reductions) without RCCL; handles[] ordered by rank */
int qd_comm_init_local(qd_handle* handles, int n);
int qd_comm_stats(qd_handle h, int* halo_exchanges);
int qd_comm_allreduce_count(qd_handle h, int* allreduces);        /* all-reduce collectives issued so far (statistics) */
/* of those, the eta sums of ocean sub-steps that went out INSIDE the ncclGroup of the following halo exchange instead of as a
 * collective launch of their own (round 3; WindDrivenSlabOcean.step removes the global mean of eta once per sub-step,
 * pygcm/ocean.py:369-377 -- the reference has no counterpart for the transport) */
int qd_comm_grouped_sum_count(qd_handle h, int* grouped);
/* host ring: the ranks of ONE node all-reduce the few host-visible scalars of a step (eta sum per ocean sub-step, CFL maxima)
 * through a POSIX shared-memory segment instead of an RCCL launch each; every rank passes the same `name` (unique per launch).
 * Optional: without it those scalars go through RCCL like everything else. */
int qd_comm_init_shm(qd_handle h, const char* name);
int qd_comm_host_allreduce_count(qd_handle h, int* n);
/* the ring by itself (no handle, no GPU): what tests/test_bands_cpu.py drives from several processes */
int qd_hostring_open(const char* name, int rank, int world, void** ring_out);
int qd_hostring_allreduce(void* ring, double* vals, int n, int op_max);   /* n <= 8; op_max 0 = sum in rank order, 1 = max */
int qd_hostring_close(void* ring);

/* Planner simulation (host only, no device is touched): the latitude-band planner of SURVEY.md 8(e) -- validity margins, launch
 * segments, the decision WHEN to exchange halos and WHICH rows move where -- on a handle that owns no memory.  Exchanges are
 * logged instead of performed; tests/test_bands_cpu.py performs them with torch.distributed (gloo) on NumPy slabs.
 * The reference has no counterpart (single process, np.roll on whole arrays: pygcm/ocean.py:306-310, dynamics.py:144-173). */
int qd_plansim_create(const qd_grid_desc* desc, qd_handle* out);
int qd_plansim_destroy(qd_handle h);
int qd_plansim_plan(qd_handle h, const int* fields, const int* radii, int n, int want);      /* -> margin of the outputs */
int qd_plansim_mark(qd_handle h, const int* fields, int n, int margin);
int qd_plansim_margin(qd_handle h, int field);
int qd_plansim_segments(qd_handle h, int margin, int* row0_nrows_pairs);                    /* -> number of segments (<= 3) */
int qd_plansim_pop_exchange(qd_handle h, int* fields_out, int max_fields, int* geom4);      /* geom4 = {H, owned rows, up, dn} */
int qd_plansim_segments_rows(qd_handle h, int vr0, int cnt, int* row0_nrows_pairs);          /* ring rows [vr0, vr0 + cnt) -> segments (<= 6) */
int qd_comm_barrier(qd_handle h);
/* Device-side exchange over the peer mapping (round 4; QD_PEER_EXCHANGE=1): halo rows and global sums are STORED into the
 * neighbours' memory by small kernels on the handle's own stream and polled there -- no collective launch, no host.  Every rank
 * exports the IPC handle of its mailbox (64 bytes), the host side hands every rank all handles in rank order, qd_peer_connect
 * maps them; from then on qd_comm_init is not needed.  In-process groups (qd_comm_init_local) switch to it through the environment
 * variable.  Replaces nothing in the reference (single process: np.roll on whole arrays, pygcm/ocean.py:306-310,369-377). */
int qd_peer_export(qd_handle h, void* handle64, size_t bytes);
int qd_peer_connect(qd_handle h, const void* handles, size_t bytes_each, int world);
int qd_comm_peer_stats(qd_handle h, int* halo_exchanges, int* reductions);   /* operations that went through the mailboxes */
int qd_comm_peer_carried(qd_handle h);   /* of those halo exchanges: pushes that went out INSIDE a compute launch (QD_PEER_OVERLAP=2); < 0: bad handle */
/* self-test of a freshly connected transport: `iters` ring exchanges of rows whose values encode (sender, iteration, position) and
 * all-reduces of rank-dependent numbers with known results; *wrong = values this rank found wrong (stale data would show here, not as
 * an error).  qd_peer_disable: leave the mailboxes (all ranks together, when any rank's self-test failed) -- the host then calls
 * qd_comm_init. */
int qd_peer_selftest(qd_handle h, int iters, long long* wrong);
int qd_peer_disable(qd_handle h);
int qd_comm_allreduce_max(qd_handle h, double* inout, int n);     /* bench timing: max over ranks */

/* ---- profiling hooks ----------------------------------------------------------------- */
/* mean device time (ms) of the kernels tagged `name` since the last reset, measured with
 * hipEvents on the handle's stream when timing is enabled. */
/* measured streaming ceiling of the device ("also report a measured device-copy ceiling"): a device-to-device copy of
 * `bytes` (choose > 256 MiB to get past the Infinity Cache), `reps` times; *gbs = (read + written bytes) / time. */
int qd_copy_ceiling(qd_handle h, size_t bytes, int reps, double* gbs);
/* Launcher tuning switches (QD_STREAM_R*, QD_TAIL_R / _RP / _V / _GENERAL, QD_MED_BLOCKS, QD_SHAPIRO_R, QD_TILE_TR: README.md) are
 * read from the environment once, in qd_create; the developer scripts that sweep one of them on a live handle call this to have
 * them read again.  Nothing the reference has (it re-reads its own QD_* variables every step: pygcm/dynamics.py:330-348). */
int qd_tune_reload(qd_handle h);
int qd_timing_enable(qd_handle h, int on);           /* 0 off, 1 every kernel group */
int qd_timing_select(qd_handle h, const char* name); /* time only the groups "name[:stride],..." (implies on); with a stride
                                                       * only every stride-th launch of the group is bracketed */
int qd_timing_get(qd_handle h, const char* name, double* mean_ms, int64_t* launches);
int qd_timing_reset(qd_handle h);

#ifdef __cplusplus
}
#endif
#endif /* QINGDAI_HIP_H */
