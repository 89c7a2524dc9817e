#!/usr/bin/env python3
"""
bench.py -- headline benchmark: simulated planet-days per wall-second at 721x1440 f64.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          (no launcher: the script starts its N rank processes itself, before any HIP call)

A "step" is one pass of the per-timestep grid update over the synthetic seed-42 planet:
forcing (two-star insolation + Teq) -> SpectralModel.time_step(Teq, dt, albedo) with the explicit
energy budget (QD_ENERGY_W=1, QD_CLOUD_COUPLE=1) -> ocean coupling (Q_net, WindDrivenSlabOcean.step,
SST write-back), i.e. the loop of the reference's own harness scripts/benchmark_jax.py:124-158
(--with-ocean) at BASELINE.json configs[2].  State is resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0).  No PyTorch anywhere: ranks rendezvous through the filesystem
(single node) and synchronise / reduce through RCCL inside libqingdai_hip.so.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# RCCL between processes needs dmabuf IPC on this stack (the image exports it already; make sure ranks started some other way have it too)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PLANET_DAY_S = 2 * np.pi / 8.726646259971648e-5      # 72 000 s (constants.py:34)
HBM_PEAK_GBS = 8000.0                                # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# algorithmic bytes per cell: the del^4 pair on u,v,h,q,cloud reads 5 + writes 5
# fields in the first Laplacian pass (80 B) and reads L1 + F and writes F' in the second (120 B).
BYTES_PER_CELL = {"k_laplacian": 80.0, "k_hyper_apply": 120.0, "k_dyn_hyper": 88.0, "k_ocn_hyper": 65.0}


def build_case(nlat, nlon, with_ocean, device=0, band=None, rank=0, world=1):
    import qingdai_amd as qa
    from qingdai_amd.device import Device
    from qingdai_amd.topography import create_land_sea_mask, generate_base_properties
    grid = qa.SphericalGrid(nlat, nlon)
    mask = create_land_sea_mask(grid)
    base_albedo, friction = generate_base_properties(mask)
    p = qa.QdParams(energy_w=1.0, cloud_couple=1)
    Cs_ocean = 1000.0 * 4200.0 * 50.0
    if band is not None:
        Device(grid, p, device=device, row0=band[0], n_rows=band[1], halo=band[2], rank=rank, world=world)
    m = qa.SpectralModel(grid, friction, H=8000, tau_rad=10 * 24 * 3600, greenhouse_factor=0.40,
                         C_s_map=np.where(mask == 1, 3e6, Cs_ocean).astype(float), land_mask=mask,
                         Cs_ocean=Cs_ocean, Cs_land=3e6, Cs_ice=5e6, params=p, device=device)
    oc = None
    if with_ocean:
        oc = qa.WindDrivenSlabOcean(grid, mask, 50.0, init_Ts=np.full((nlat, nlon), 288.0))
    m._dev.upload_now("BASE_ALBEDO", base_albedo)
    forcing = qa.ThermalForcing(grid, qa.OrbitalSystem())
    return grid, m, oc, forcing, mask, base_albedo, friction


def cpu_baseline(nlat, nlon, with_ocean, with_phys=True, budget_s=20.0, state=None, t0_s=0.0):
    """The oracle (NumPy restatement, proven equal to the reference in the authoring container, speed ratio to the real reference
    in BASELINE.md) timed on this box's host cores on the SAME loop as the GPU leg -- driver physics (precipitation / cloud /
    albedo diagnostics) -> forcing -> time_step(Teq, dt, albedo) -> ocean coupling: 1 warm-up step, then as many steps as fit in
    ~budget_s (at least 2).  `state`: the prognostic fields the GPU leg ended on (downloaded after its timed region): the oracle
    continues from THEM, so its ocean runs the same number of sub-steps per step as the GPU leg's did (a cold start runs fewer:
    the round-3 baseline was a lower bound for that reason)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import qd_oracle as qo
    from qd_oracle.driver import DriverOracle
    from qingdai_amd.topography import create_land_sea_mask, generate_base_properties
    g = qo.Grid(nlat, nlon)
    mask = create_land_sea_mask(g)
    alb, fric = generate_base_properties(mask)
    P = qo.defaults(energy_w=1.0)
    m = qo.AtmosOracle(g, fric, mask, P, C_s_map=np.where(mask == 1, 3e6, P.Cs_ocean).astype(float))
    oc = qo.OceanOracle(g, mask, P, init_Ts=np.full((nlat, nlon), 288.0)) if with_ocean else None
    f = qo.Forcing(g)
    dt = 300.0
    spun_up = False
    if state is not None:
        for k_dev, k_or in (("U", "u"), ("V", "v"), ("H", "h"), ("TS", "T_s"), ("Q", "q"), ("CLOUD", "cloud_cover"), ("HICE", "h_ice")):
            if k_dev in state:
                setattr(m, k_or, np.array(state[k_dev], dtype=float))
        if oc is not None and all(k in state for k in ("UO", "VO", "ETA", "SST")):
            oc.uo, oc.vo, oc.eta, oc.Ts = (np.array(state[k], dtype=float) for k in ("UO", "VO", "ETA", "SST"))
        spun_up = True
    if with_phys:
        d = DriverOracle(g, m, oc, f, mask, alb, P)

        def one(i):
            d.step(t0_s + i * dt, dt, pass_albedo=True, commit=False)
    else:
        from qd_oracle import column as col
        albedo = np.where(mask == 0, 0.08, alb)

        def one(i):
            t = t0_s + i * dt
            a_, b_ = f.insolation_components(t)
            m.isr_A, m.isr_B, m.isr = a_, b_, a_ + b_
            Teq = f.equilibrium_temp(t, albedo)
            m.time_step(Teq, dt, albedo=albedo)
            if oc is not None:
                T_a = 288.0 + (9.81 / 1004.0) * m.h
                _, SW_sfc, _ = col.shortwave(m.isr, albedo, m.cloud_eff_last, P)
                ice_frac = 1.0 - np.exp(-np.maximum(m.h_ice, 0.0) / 0.5)
                _, LW_sfc, _, _, _ = col.longwave_v2(m.T_s, T_a, m.cloud_eff_last, col.surface_emissivity_map(mask, ice_frac, P), P)
                SH = col.sensible_heat(m.T_s, T_a, m.u, m.v, P)
                ice = m.h_ice > 0.0
                oc.step(dt, m.u, m.v, Q_net=SW_sfc - LW_sfc - SH - m.LH_last, ice_mask=ice)
                m.T_s = np.where((mask == 0) & (~ice), oc.Ts, m.T_s)
    one(0)
    t0 = time.perf_counter()
    n = 0
    while True:
        one(n + 1)
        n += 1
        el = time.perf_counter() - t0
        if n >= 2 and el >= budget_s or n >= 200:
            break
    per = el / n
    return {"value": dt / per / PLANET_DAY_S, "unit": "planet-days/s", "cores": 1, "kind": "port",
            "sample": f"{n} steps of the same {nlat}x{nlon} loop as the GPU leg (driver physics {'on' if with_phys else 'off'}, "
                      f"ocean {'on' if with_ocean else 'off'}) " + ("continuing from the state the GPU leg ended on" if spun_up else "from a cold start") +
                      f" after 1 warm-up ({per * 1e3:.1f} ms/step, {oc.last_n_sub if oc is not None else 0} ocean sub-steps per step); "
                      f"NumPy is single-threaded, {os.cpu_count()} host cores available"}


def ecology_leg(dev, grid, mask, forcing, dt, W, K):
    """ms/step of the driver loop (physics + ocean + hydrology) without and with the ecology sub-step (qd_step_n bit5):
    default population (QD_ECO_NS=20 species), 16 bands, 2 % of the land cells x 150 sampled individuals."""
    from qingdai_amd.ecology import EcologyAdapter, IndividualPool
    stars = forcing.star_table([i * dt for i in range(W + K)])

    def run(eco_on, n):
        dev.sync()
        t0 = time.perf_counter()
        dev.step_n(stars[:n], dt, with_ocean=True, with_physics=True, pass_albedo=False, with_hydrology=True, ecology=eco_on)
        dev.sync()
        return (time.perf_counter() - t0) / n * 1e3, dev.last_ocean_nsub()
    eco = EcologyAdapter(grid, mask, dev=dev, albedo_couple=True)
    pool = IndividualPool(grid, mask, eco, sample_frac=0.02, per_cell=150, substeps_per_day=10)
    t0 = time.perf_counter()
    eco.pop.push_layers()                                   # what the daily step costs the device side: the LAI stack upload + sum
    dev.sync()
    up_ms = (time.perf_counter() - t0) * 1e3
    run(True, W)
    # the ocean's sub-step count drifts with the state, so the two loops alternate (A B A B) and each is averaged
    h = max(1, K // 2)
    legs = [run(False, h), run(True, h), run(False, h), run(True, h)]
    # compare a neighbouring pair that ran with the same sub-step count (else the first pair)
    pair = next(((a, b) for a, b in ((legs[0], legs[1]), (legs[2], legs[1]), (legs[2], legs[3])) if a[1] == b[1]), (legs[0], legs[1]))
    base, with_eco = pair[0][0], pair[1][0]
    dev.timing(select="eco_indiv,eco_canopy")
    run(True, min(240, W + K))                              # one planet-day: 10 individual sub-steps at the default K = 10
    ind_ms, ind_n = dev.timing_get("eco_indiv")
    can_ms, can_n = dev.timing_get("eco_canopy")
    pool32 = IndividualPool(grid, mask, eco, sample_frac=0.02, per_cell=150, substeps_per_day=10, f32_storage=True)
    dev.timing(select="eco_indiv")
    run(True, min(240, W + K))
    ind32_ms, _ = dev.timing_get("eco_indiv")
    dev.timing(on=False)
    st = eco.pop.state()
    # f32 STORAGE of the canopy maps (LAI_tot, snapshot, f, alpha, banded alpha; QD_ECO_F32, BASELINE configs[4] "f32 mixed
    # precision") on a second handle of the same case: the same loop, and the deviation of its maps from this handle's f64 maps
    f32 = None
    try:
        for k in ("ECO_ALPHA", "ECO_F"):
            dev._host.pop(k, None)
        a64, f64_ = dev.get("ECO_ALPHA").copy(), dev.get("ECO_F").copy()
        grid2, m2, _, forcing2, _, _, _ = build_case(grid.n_lat, grid.n_lon, True)
        dev2 = m2._dev
        eco2 = EcologyAdapter(grid2, mask, dev=dev2, albedo_couple=True, f32_maps=True)
        IndividualPool(grid2, mask, eco2, sample_frac=0.02, per_cell=150, substeps_per_day=10, f32_storage=True)
        eco2.pop.push_layers()

        def run2(eco_on, n):
            dev2.sync()
            t0 = time.perf_counter()
            dev2.step_n(stars[:n], dt, with_ocean=True, with_physics=True, pass_albedo=False, with_hydrology=True, ecology=eco_on)
            dev2.sync()
            return (time.perf_counter() - t0) / n * 1e3, dev2.last_ocean_nsub()
        run2(True, W)
        legs2 = [run2(False, h), run2(True, h), run2(False, h), run2(True, h)]     # a younger state than the f64 handle's: fewer ocean sub-steps
        pair2 = next(((a, b) for a, b in ((legs2[0], legs2[1]), (legs2[2], legs2[1]), (legs2[2], legs2[3])) if a[1] == b[1]), (legs2[0], legs2[1]))
        ms32, nsub32 = pair2[1]
        a32, f32m = dev2.get("ECO_ALPHA"), dev2.get("ECO_F")
        f32 = {"ms_per_step": ms32, "ms_per_step_without_ecology": pair2[0][0], "ocean_n_sub_of_the_pair": nsub32,
               "legs_ms_nsub_A_B_A_B": [[round(a, 4), b] for a, b in legs2], "stored_as_f32": ["ECO_LAI", "ECO_LAI_SNAP", "ECO_F", "ECO_ALPHA", "ECO_ALPHA_BANDED",
                                                                           "individual coefficient table"],
               "kept_f64": ["E_day", "LAI plane sum", "lai-delta reduction", "all arithmetic"],
               "alpha_max_abs_dev_vs_f64_maps": float(np.nanmax(np.abs(a32 - a64))), "canopy_factor_max_abs_dev_vs_f64_maps": float(np.max(np.abs(f32m - f64_))),
               "dev_vs_reference_fixture": "tests/test_gpu_ecology.py::test_canopy_alpha_sequence_vs_reference[f32_maps] (eco_19x36*.npz): < 5e-7"}
        dev2.close()
    except Exception as e:       # noqa: BLE001
        f32 = {"error": str(e)}
    return {"ms_per_step": with_eco, "f32_maps": f32, "individual_substep_kernel_ms": ind_ms, "individual_substeps_timed": ind_n,
            "individual_substep_kernel_ms_f32_table": ind32_ms, "individual_table_bytes_f64": int(pool32.n_indiv) * int(eco.bands.nbands) * 8,
            "canopy_policy_launch_ms": can_ms, "canopy_policy_launches_timed": can_n, "ms_per_step_without_ecology": base, "bands": int(eco.bands.nbands),
            "species_planes": int(eco.pop.LAI_layers_SK.shape[0] * eco.pop.LAI_layers_SK.shape[1]),
            "individuals": int(pool.n_indiv), "lai_stack_upload_ms_per_day": up_ms, "canopy_recomputes": st["n_recompute"],
            "ocean_n_sub_of_the_pair": pair[0][1], "legs_ms_nsub_A_B_A_B": [[round(a, 4), b] for a, b in legs],
            "dtype": "f64", "note": "loop = driver iteration incl. hydrology commit; daily population dynamics are host code, not timed"}


def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher: start the N rank processes as fresh children (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, like torch.distributed.run sets them) and exit with the worst of their codes.  This process has
    not loaded libqingdai_hip.so or touched HIP at this point, and it never does."""
    import socket
    import subprocess
    with socket.socket() as sk:                               # a free port for the file rendezvous key (bands.exchange_unique_id)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll: the first rank that fails takes the others with it (they would block in the rendezvous or a collective for ever),
    # and the whole launch has a deadline (QD_BENCH_SPAWN_TIMEOUT seconds, default 3000)
    deadline = time.time() + float(os.environ.get("QD_BENCH_SPAWN_TIMEOUT", "3000"))
    rc, live = 0, list(procs)
    while live:
        for pr in list(live):
            r = pr.poll()
            if r is None:
                continue
            live.remove(pr)
            if r != 0:
                rc = max(rc, abs(r))
        if rc or time.time() > deadline:
            rc = rc or 124
            for pr in live:
                pr.terminate()
            t_kill = time.time() + 10.0
            for pr in live:
                try:
                    pr.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    pr.kill()
            break
        time.sleep(0.05)
    raise SystemExit(rc)


def find_profile(nlat, nlon):
    """The newest committed rocprofv3 summary (profiles/rNN*_fused_kernels.json, scripts/profile_round.sh + assemble_profiles.py) whose
    grid is this run's; None when there is none."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fused_kernels.json"))):
        try:
            with open(f) as fh:
                pj = json.load(fh)
        except Exception:
            continue
        if pj.get("grid") == [nlat, nlon]:
            best = (f, pj)                                    # sorted by name: r03 after r02
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--nlat", type=int, default=721)
    ap.add_argument("--nlon", type=int, default=1440)
    ap.add_argument("--no-ocean", action="store_true")
    ap.add_argument("--no-driver-physics", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-ecology-leg", action="store_true",
                    help="skip the supplementary BASELINE configs[4] measurement (16-band ecology sub-step in the loop)")
    ap.add_argument("--timing-stride", type=int, default=-1,
                    help="bracket every n-th launch of the profiled kernels with HIP events (default: automatic, see below; "
                         "1 = every launch, 1000000 = practically none)")
    ap.add_argument("--spawn-check", action="store_true",
                    help="(test hook) every rank prints its launcher environment and exits before loading the library")
    ap.add_argument("--profile-kernel", default="k_dyn_hyper",
                    help="kernel whose HIP-event time feeds `roofline` (k_dyn_hyper: the fused dynamics + del^4 kernel)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("QD_BENCH_ONE_DEVICE") == "1":     # rehearsal of the multi-rank flow on a 1-GPU box
        local_rank = 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # under rocprofv3 the profiler's preloaded library has already initialised the GPU in THIS process: forking rank processes
        # from here is the exec-after-GPU-init this pool forbids -- profile one rank (or the self-ring) instead
        if any(k.startswith("ROCP") or k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
            raise SystemExit("bench.py --gpus N cannot start its own rank processes under rocprofv3 (the profiler has initialised the "
                             "GPU in this process): use a launcher, or profile N=1 / QD_BENCH_SELF_RING")
        spawn_ranks(args.gpus)                                # never returns
    if args.spawn_check:
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}), flush=True)
        # (test hooks of tests/test_bands_cpu.py: one rank dies, another would block for ever -- the spawner has to end both)
        if os.environ.get("QD_BENCH_TEST_FAIL_RANK") == str(rank):
            raise SystemExit(3)
        if os.environ.get("QD_BENCH_TEST_HANG_RANK") == str(rank):
            time.sleep(300)
        return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run / torchrun "
                         f"(--nproc-per-node {args.gpus}) or without any launcher; only the launcher's env vars are used, not torch")

    with_ocean = not args.no_ocean
    with_phys = not args.no_driver_physics
    dt = 300.0
    band = None
    # QD_BENCH_SELF_RING=N (developer rehearsal on ONE GPU, never a reported number): run the multi-rank code path of this script as
    # the middle band of an N-band decomposition over a one-rank RCCL communicator whose ring neighbours are the rank itself
    ring = int(os.environ.get("QD_BENCH_SELF_RING", "0")) if world == 1 else 0
    banded = world > 1 or ring > 1
    if world > 1:
        from qingdai_amd.bands import band_ranges, preferred_halo
        r0, n = band_ranges(args.nlat, world)[rank]
        band = (r0, n, preferred_halo(args.nlat, world, dt))
    elif ring > 1:
        from qingdai_amd.bands import band_ranges, preferred_halo
        r0, n = band_ranges(args.nlat, ring)[ring // 2]
        band = (r0, n, preferred_halo(args.nlat, ring, dt))
    grid, m, oc, forcing, mask, base_albedo, friction = build_case(args.nlat, args.nlon, with_ocean, device=local_rank,
                                                                   band=band, rank=rank, world=world)
    dev = m._dev
    transport = None
    if banded:
        from qingdai_amd.bands import init_comm
        transport = init_comm(dev, rank, world)
    K, W = args.steps, args.warmup
    stars_w = forcing.star_table([i * dt for i in range(W)])
    stars_k = forcing.star_table([(W + i) * dt for i in range(K)])

    def barrier():
        dev.sync()
        if banded:
            dev._chk(dev.lib.qd_comm_barrier(dev.h), "qd_comm_barrier")

    if W > 0:
        dev.step_n(stars_w, dt, with_ocean=with_ocean, with_physics=with_phys, pass_albedo=True)
    barrier()
    comm0 = (0, 0, 0)
    if banded:
        import ctypes as _ct
        ne, na, nh = _ct.c_int(0), _ct.c_int(0), _ct.c_int(0)
        dev.lib.qd_comm_stats(dev.h, _ct.byref(ne)); dev.lib.qd_comm_allreduce_count(dev.h, _ct.byref(na))
        dev.lib.qd_comm_host_allreduce_count(dev.h, _ct.byref(nh))
        comm0 = (ne.value, na.value, nh.value)
    also = "k_ocn_hyper" if (with_ocean and args.profile_kernel != "k_ocn_hyper") else None
    # HIP events inside the timed region, on the handle's stream, SAMPLED.  The two profiled kernels are single launches whose start /
    # stop events ride on the dispatch itself (hipExtLaunchKernelGGL: the timestamps of the dispatch's completion signal, the interval
    # the rocprofv3 kernel trace reports); a timed dispatch still idles the stream for a few us (kernel trace, profiles/README.md: 14
    # timed launches a step were 7 % of the step), so the per-step kernel is timed on ~16 steps of the K and the per-sub-step kernel
    # on every 16th launch (about one a step).
    s_main = max(1, K // 16) if args.timing_stride < 0 else max(1, args.timing_stride)
    s_also = 16 if args.timing_stride < 0 else max(1, args.timing_stride)
    dev.timing(select=f"{args.profile_kernel}:{s_main}" + (f",{also}:{s_also}" if also else ""))
    t0 = time.perf_counter()
    dev.step_n(stars_k, dt, with_ocean=with_ocean, with_physics=with_phys, pass_albedo=True)
    barrier()
    el = time.perf_counter() - t0
    if banded:                                     # MAX over ranks
        import ctypes
        v = (ctypes.c_double * 1)(el)
        dev._chk(dev.lib.qd_comm_allreduce_max(dev.h, v, 1), "qd_comm_allreduce_max")
        el = v[0]
    comm = None
    if banded:                                     # collectives this rank issued inside the timed region, per step
        import ctypes as _ct
        ne, na, nh = _ct.c_int(0), _ct.c_int(0), _ct.c_int(0)
        dev.lib.qd_comm_stats(dev.h, _ct.byref(ne)); dev.lib.qd_comm_allreduce_count(dev.h, _ct.byref(na))
        dev.lib.qd_comm_host_allreduce_count(dev.h, _ct.byref(nh))
        comm = {"band_rows": band[1], "halo_rows": band[2], "halo_exchanges_per_step": (ne.value - comm0[0]) / K,
                "rccl_allreduces_per_step": (na.value - comm0[1]) / K, "host_ring_allreduces_per_step": (nh.value - comm0[2]) / K}
        ng = _ct.c_int(0)
        dev.lib.qd_comm_grouped_sum_count(dev.h, _ct.byref(ng))
        # of the all-reduces, those that went out inside a halo exchange's ncclGroup (counted over the whole run incl. warm-up)
        comm["allreduces_grouped_into_an_exchange_per_step"] = ng.value / max(1, K + W)
        comm["transport"] = transport
        if transport == "peer":
            # every halo exchange and every reduction went through the device-side mailboxes (qd_peer.hip): small kernels on the
            # handle's own stream, no collective launch at all
            nph, npr = _ct.c_int(0), _ct.c_int(0)
            dev.lib.qd_comm_peer_stats(dev.h, _ct.byref(nph), _ct.byref(npr))
            comm["reductions_per_step"] = comm.pop("rccl_allreduces_per_step")
            comm["peer_ops_whole_run"] = {"halo_exchanges": nph.value, "reductions": npr.value,
                                          # pushes that were the first workgroups of the interior momentum launch (QD_PEER_OVERLAP=2)
                                          "pushes_carried_by_a_compute_launch": dev.lib.qd_comm_peer_carried(dev.h)}
            comm["collective_launches_per_step"] = 0.0
        else:
            comm["collective_launches_per_step"] = comm["halo_exchanges_per_step"] + comm["rccl_allreduces_per_step"] - ng.value / max(1, K + W)
    end_state = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        # what the CPU baseline continues from (downloaded AFTER the timed region; the ecology leg below runs on)
        names = ["U", "V", "H", "TS", "Q", "CLOUD", "HICE"] + (["UO", "VO", "ETA", "SST"] if with_ocean else [])
        end_state = {}
        for k in names:
            dev._host.pop(k, None)
            end_state[k] = np.array(dev.get(k), dtype=float)
            dev._host.pop(k, None)
    kern_ms, kern_n = dev.timing_get(args.profile_kernel)
    also_ms, also_n = dev.timing_get(also) if also else (0.0, 0)
    dev.timing(on=False)

    ms_per_step = el / K * 1e3
    value = (K * dt / PLANET_DAY_S) / el
    cells = (band[1] if band else args.nlat) * args.nlon     # cells one launch of this rank's kernel covers
    bpc = BYTES_PER_CELL.get(args.profile_kernel, 0.0)
    achieved = (bpc * cells / 1e9) / (kern_ms / 1e3) if kern_ms > 0 else 0.0
    out = {
        "metric": f"simulated planet-days/sec at {args.nlat}x{args.nlon} f64", "value": value, "unit": "planet-days/s",
        "n_gpus": args.gpus, "steps": K, "warmup": W, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.nlat}x{args.nlon} seed-42 planet, dt=300 s: forcing + time_step(Teq, dt, albedo) "
                               f"with QD_ENERGY_W=1 QD_CLOUD_COUPLE=1" + (" + slab-ocean coupling" if with_ocean else "") +
                               (" + driver-side precipitation/cloud/albedo diagnostics (run_simulation.py:1766-1934,2063-2146)" if with_phys else "") +
                               " (BASELINE configs[2]; loop order of benchmark_jax.py:124-158)",
                   "grid": [args.nlat, args.nlon], "dt_s": dt, "ocean_n_sub": dev.last_ocean_nsub() if with_ocean else 0,
                   "parallelism": f"lat-bands x{args.gpus}"},
        "roofline": {"bound": "hbm", "kernel": args.profile_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "bytes_per_cell": bpc, "cells": cells, "avg_kernel_ms": kern_ms, "launches": kern_n},
    }
    if comm:
        out["config"]["collectives"] = comm
    if ring > 1:
        out["rehearsal"] = f"self-ring: ONE band of {ring} on one GPU, halos refreshed from the band itself -- not a benchmark result"
    if also and also_ms > 0:     # the kernel with the largest share of the step (ocean sub-steps), same accounting
        a2 = (BYTES_PER_CELL[also] * cells / 1e9) / (also_ms / 1e3)
        out["roofline_ocean_substep"] = {"bound": "hbm", "kernel": also, "achieved": a2, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": a2 / HBM_PEAK_GBS, "traffic": None, "bytes_per_cell": BYTES_PER_CELL[also],
                                         "cells": cells, "avg_kernel_ms": also_ms, "launches": also_n}
    if rank == 0 and args.gpus == 1:      # measured streaming ceiling next to the 8 TB/s spec peak (outside the timed region)
        try:
            out["roofline"]["measured_copy_ceiling_gbs"] = dev.copy_ceiling()
        except Exception:
            pass
    # What cannot be measured from inside this process comes from the committed rocprofv3 summaries of this same command
    # (profiles/, one pass per counter as MI355X_MICROARCH.md prescribes), when they cover this kernel and grid -- else null:
    #   traffic                 FETCH_SIZE (x2: gfx950 correction) + WRITE_SIZE per launch, bytes
    #   avg_kernel_ms_rocprof   the kernel-trace duration of the committed profile (the events above time the same interval live)
    try:
        found = find_profile(args.nlat, args.nlon)
        if args.gpus == 1 and found:
            prof, pj = found
            for kname, key in ((args.profile_kernel, "roofline"), (also, "roofline_ocean_substep")):
                if kname and key in out and kname in pj["kernels"]:
                    e = pj["kernels"][kname]
                    out[key]["traffic"] = e.get("traffic_bytes")
                    out[key]["avg_kernel_ms_rocprof"] = e.get("avg_kernel_ms_rocprof")
                    if e.get("avg_kernel_ms_rocprof"):
                        out[key]["frac_rocprof"] = (out[key]["bytes_per_cell"] * cells / 1e9) / (e["avg_kernel_ms_rocprof"] / 1e3) / HBM_PEAK_GBS
                    out[key]["profile_source"] = "profiles/" + os.path.basename(prof) + " (" + pj.get("source", "") + ")"
                    # the committed profile carries the hash of the DEVICE code it was measured on (qingdai_amd/_codehash.py: the
                    # .hip_fatbin of the profiled translation units; host-only edits do not touch it): say so when that has changed
                    from qingdai_amd import _codehash
                    if pj.get("device_code_sha256_16"):
                        out[key]["profile_stale"] = bool(_codehash.stale_against(pj["device_code_sha256_16"]))
                    else:                                     # profiles of rounds 1-3: source-file hashes
                        import hashlib
                        cs = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qingdai_amd", "csrc")
                        stale = [f for f, h in pj.get("kernel_sources_sha256_16", {}).items()
                                 if not os.path.exists(os.path.join(cs, f)) or hashlib.sha256(open(os.path.join(cs, f), "rb").read()).hexdigest()[:16] != h]
                        out[key]["profile_stale"] = bool(stale)
    except Exception:
        pass
    # supplementary, outside the timed region: the same grid with the driver's full iteration (+ hydrology commit) and the
    # 16-band ecology sub-step inside the resident loop (BASELINE configs[4], arithmetic in f64)
    if rank == 0 and args.gpus == 1 and with_phys and with_ocean and not args.no_ecology_leg:
        try:
            out["ecology_config5"] = ecology_leg(dev, grid, mask, forcing, dt, W, K)
        except Exception as e:       # noqa: BLE001  (never lose the main line to the supplementary leg)
            out["ecology_config5"] = {"error": str(e)}
    if not args.no_cpu_baseline and rank == 0 and args.gpus == 1:
        out["cpu_baseline"] = cpu_baseline(args.nlat, args.nlon, with_ocean, with_phys, args.cpu_budget, state=end_state, t0_s=(W + K) * dt)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
