"""GPU (-m gpu): latitude-band decomposition on ONE device.  N band handles (ring halos, validity
margins, deep-halo recompute, band-wise reductions / histogram all-reduce) driven by N host threads
must reproduce the whole-globe handle: bit-for-bit for the atmosphere (no global sums on its path
except the exact median), to rounding of the eta-mean / weighted sums for the coupled run."""
import numpy as np
import pytest

from util import STATE, surface, relerr

pytestmark = pytest.mark.gpu


def _setup(nlat, nlon, over):
    import qingdai_amd as qa
    _, mask, alb, fric = surface(nlat, nlon)
    grid = qa.SphericalGrid(nlat, nlon)
    p = qa.QdParams(**over)
    p.has_csmap = 0
    return qa, grid, mask, alb, fric, p


def _seed_state(nlat, nlon, seed):
    r = np.random.default_rng(seed)
    lat = np.linspace(-np.pi / 2, np.pi / 2, nlat)[:, None]
    lon = np.linspace(0, 2 * np.pi, nlon)[None, :]
    st = {"U": 25.0 * np.cos(lat) * np.sin(2 * lon) + r.normal(0, 3.0, (nlat, nlon)),
          "V": 8.0 * np.sin(2 * lat) * np.cos(3 * lon) + r.normal(0, 2.0, (nlat, nlon)),
          "H": 8000.0 + 300 * np.sin(lat) ** 2 + 40.0 * np.cos(lat) * np.cos(2 * lon) + r.normal(0, 2.0, (nlat, nlon)),
          "TS": 262.0 + 38.0 * np.cos(lat) ** 2 + r.normal(0, 1.0, (nlat, nlon)),
          "Q": np.clip(0.006 + 0.004 * np.cos(lat) ** 2 + r.normal(0, 5e-4, (nlat, nlon)), 0, 0.5),
          "CLOUD": np.clip(0.3 + 0.3 * np.sin(3 * lon) * np.cos(lat) + r.normal(0, 0.05, (nlat, nlon)), 0, 1),
          "HICE": np.where(np.abs(lat) > 1.1, 0.4 + 0.3 * r.random((nlat, nlon)), 0.0)}
    return st


def _run(world, nlat, nlon, nsteps, over, with_ocean, with_phys, seed=3, mutate=None):
    from qingdai_amd.bands import BandGroup
    from qingdai_amd.device import Device
    qa, grid, mask, alb, fric, p = _setup(nlat, nlon, over)
    forcing = qa.ThermalForcing(qa.SphericalGrid(nlat, nlon), qa.OrbitalSystem())
    stars = forcing.star_table([i * 300.0 for i in range(nsteps)])
    st = _seed_state(nlat, nlon, seed)
    if mutate is not None:
        mutate(st)
    static = {"LAND_MASK": mask, "FRICTION": fric, "BASE_ALBEDO": alb}
    names = ["U", "V", "H", "TS", "Q", "CLOUD", "HICE"] + (["UO", "VO", "ETA", "SST"] if with_ocean else [])
    if world == 1:
        dev = Device(grid, p)
        for k, v in {**static, **st}.items():
            dev.upload_now(k, v)
        dev.step_n(stars, 300.0, with_ocean=with_ocean, with_physics=with_phys, pass_albedo=True)
        out = {k: dev.get(k).copy() for k in names}
        dev.close()
        return out, None
    grp = BandGroup(grid, world, p)
    for k, v in {**static, **st}.items():
        grp.set(k, v)
    grp.run(lambda d, r: d.step_n(stars, 300.0, with_ocean=with_ocean, with_physics=with_phys, pass_albedo=True))
    out = {k: grp.get(k) for k in names}
    ex = grp.exchanges()
    grp.close()
    return out, ex


@pytest.mark.parametrize("world", [2, 3])
def test_bands_atmosphere_bit_identical(gpu, world):
    ref, _ = _run(1, 61, 96, 7, dict(energy_w=1.0), False, False)
    got, ex = _run(world, 61, 96, 7, dict(energy_w=1.0), False, False)
    print("halo exchanges per band:", ex)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), (k, relerr(got[k], ref[k]))


@pytest.mark.parametrize("scalars", ["host_ring", "collective", "collective_round2"])
def test_bands_full_step_with_ocean_and_physics(gpu, scalars, monkeypatch):
    """4 latitude bands against the whole globe, full coupled step.  host_ring: the eta sums travel through the in-process host ring
    (deferred mean, round-2 kernels); collective: no ring -- the path a multi-process run takes by default: k_ocn_stream +
    k_ocn_tail_stream on the band's segments, the in-launch sum all-reduced as the band's share of the mean (round 3);
    collective_round2: the same with QD_BAND_TAIL=0 (k_cont_sstadv + k_eta_mean + k_sst_outlier_fused)."""
    if scalars != "host_ring":
        monkeypatch.setenv("QD_NO_HOST_RING", "1")
    if scalars == "collective_round2":
        monkeypatch.setenv("QD_BAND_TAIL", "0")
    ref, _ = _run(1, 91, 144, 4, dict(energy_w=1.0, ocean_cfl=0.05), True, True)
    got, ex = _run(4, 91, 144, 4, dict(energy_w=1.0, ocean_cfl=0.05), True, True)
    print("halo exchanges per band:", ex)
    for k in ref:
        e = relerr(got[k], ref[k])
        assert e < 1e-12, (scalars, k, e)          # only the band-wise order of the global sums differs


@pytest.mark.parametrize("shape", [(181, 360), (91, 144)])
def test_fused_kernel_paths_agree(gpu, shape, monkeypatch):
    """The fused momentum + del^4 kernels have a FAST path (register rows + DPP + scalar-loaded tables, no
    nan_to_num) for interior and pole tiles and an EXACT path (every cell masked, literal nan_to_num) they fall
    back to.  Same arithmetic in the same order: the two must agree bit for bit; both must agree with the
    unfused reference-order kernels (QD_FUSED=0, true divisions instead of reciprocal tables) to rounding."""
    nlat, nlon = shape
    over = dict(energy_w=1.0, ocean_cfl=0.05)
    monkeypatch.setenv("QD_FUSED_FAST", "0")
    exact, _ = _run(1, nlat, nlon, 3, over, True, True)
    monkeypatch.setenv("QD_FUSED_FAST", "1")
    fast, _ = _run(1, nlat, nlon, 3, over, True, True)
    monkeypatch.setenv("QD_FUSED", "0")
    unfused, _ = _run(1, nlat, nlon, 3, over, True, True)
    for k in exact:
        assert np.array_equal(fast[k], exact[k]), (k, relerr(fast[k], exact[k]))
    errs = {k: relerr(fast[k], unfused[k]) for k in fast}
    print(errs)
    for k, e in errs.items():
        assert e < (1e-7 if k in ("UO", "VO", "ETA") else 1e-9), (k, e)


# forms of the one-launch ocean tail: "fast" = k_ocn_tail_fast (the shipped form), "stream" = k_ocn_tail_stream (QD_TAIL_V=1: the
# general waves on every strip, the reference form of the slim ones); the LDS-tile forms and the one-launch sub-step of round 3
# are retired (tools/retired/)
# "fused" = k_ocn_fused (QD_OCN_FUSED=1): the WHOLE sub-step in one launch, the momentum waves handing their rows to the tail waves
# through LDS rings; "fused_seq": the same launch with every strip on its sequential form (what the polar tiles and a strip with a
# non-finite value or a far departure point take)
TAIL_FORMS = ("fast", "stream", "fused", "fused_seq")


def _set_tail(monkeypatch, tail):
    monkeypatch.setenv("QD_OCN_TAIL", "0" if tail == "0" else "1")
    if tail == "stream":
        monkeypatch.setenv("QD_TAIL_V", "1")
    else:
        monkeypatch.delenv("QD_TAIL_V", raising=False)
    monkeypatch.setenv("QD_OCN_FUSED", "1" if tail.startswith("fused") else "0")
    monkeypatch.setenv("QD_FUSED_SEQ", "1" if tail == "fused_seq" else "0")
# bound on the agreement with the two-launch form: the forms differ in the ORDER of the area-weighted eta sum only (per row / per
# strip / per tile, f64 tree or fixed-point slots); over 2-3 coupled steps that rounding difference grows to ~2e-12 on the currents
TAIL_TOL = 1e-11


@pytest.mark.parametrize("tail", TAIL_FORMS)
@pytest.mark.parametrize("shape", [(181, 360), (91, 144)])
def test_ocean_tail_kernel_matches_the_two_launch_form(gpu, shape, tail, monkeypatch):
    """k_ocn_tail (continuity + SST blend / diffusion / heating + outlier filter of an ocean sub-step in one launch, the advected
    SST staged in LDS) against k_cont_sstadv + k_sst_outlier_fused (QD_OCN_TAIL=0): same device functions in the same order;
    only the order of the area-weighted eta sum differs (per 16 x 62 tile instead of per row)."""
    nlat, nlon = shape
    over = dict(energy_w=1.0, ocean_cfl=0.05)
    _set_tail(monkeypatch, "0")
    two, _ = _run(1, nlat, nlon, 3, over, True, True)
    _set_tail(monkeypatch, tail)
    one, _ = _run(1, nlat, nlon, 3, over, True, True)
    errs = {k: relerr(one[k], two[k]) for k in one}
    print(errs)
    for k, e in errs.items():
        assert e < TAIL_TOL, (k, e)


@pytest.mark.parametrize("shape", [(181, 360), (64, 97)])
def test_currents_patched_in_place_equal_the_stored_ones(gpu, shape, monkeypatch):
    """k_ocn_tail_fast on a whole-globe handle does not store uo'' / vo'' (ocean.py:409-434: nan_to_num + outlier filter + speed cap leave
    almost every cell as the momentum kernel wrote it): cells that do change are noted in a list and patched in place by the launch's
    finishing wave (QD_TAIL_FIX, default on).  Bit for bit the slabs of the storing form (QD_TAIL_FIX=0) -- with calm currents (an empty
    list), with isolated spikes and NaN (a few entries, the 4-neighbour mean reading the UNPATCHED neighbours), and with every
    twentieth ocean cell far above the cap (thousands of entries per launch, every strip contributing)."""
    nlat, nlon = shape

    def spikes(st):
        st["UO"] = np.zeros((nlat, nlon)); st["VO"] = np.zeros((nlat, nlon))
        st["UO"][nlat // 2, 0] = 1e6; st["VO"][nlat // 3, nlon - 1] = -2e6; st["UO"][nlat // 2 + 9, nlon // 2] = np.nan

    def storm(st):
        r = np.random.default_rng(11)
        hit = r.random((nlat, nlon)) < 0.05
        st["UO"] = np.where(hit, r.normal(0, 40.0, (nlat, nlon)), 0.01 * r.normal(0, 1, (nlat, nlon)))
        st["VO"] = np.where(hit, r.normal(0, 40.0, (nlat, nlon)), 0.01 * r.normal(0, 1, (nlat, nlon)))
    over = dict(energy_w=1.0, ocean_cfl=0.05)
    for mutate in (None, spikes, storm):
        monkeypatch.setenv("QD_TAIL_FIX", "0")
        stored, _ = _run(1, nlat, nlon, 2, over, True, True, mutate=mutate)
        monkeypatch.setenv("QD_TAIL_FIX", "1")
        patched, _ = _run(1, nlat, nlon, 2, over, True, True, mutate=mutate)
        for k in stored:
            assert np.array_equal(patched[k], stored[k], equal_nan=True), (k, mutate.__name__ if mutate else None)


@pytest.mark.parametrize("shape", [(181, 360), (91, 144), (64, 97), (121, 240)])
def test_fast_tail_waves_equal_the_general_ones_bit_for_bit(gpu, shape, monkeypatch):
    """k_ocn_tail_fast: strips away from the poles run slim waves (two-slot streams, the SST gather taken from the nine cells
    around the cell -- three streamed rows, DPP lane shifts, ds_bpermute in the waves that hold column 0 / n_lon - 1); a lane
    whose departure point is a cell or more away or NaN sends its wave back to the general form.  Same arithmetic on the same
    operands: every field must equal the all-general run (QD_TAIL_GENERAL=1) bit for bit -- with calm currents, with a current
    of 1e6 m/s and a NaN injected next to the seam and in the interior (the fallback), on grids whose last column group is
    partial (97, 144 columns) and whose strips do not divide evenly."""
    nlat, nlon = shape

    def spike(st):
        st["UO"] = np.zeros((nlat, nlon)); st["VO"] = np.zeros((nlat, nlon))
        st["UO"][nlat // 2, 0] = 1e6; st["VO"][nlat // 3, nlon - 1] = -2e6; st["UO"][nlat // 2 + 9, nlon // 2] = np.nan
    over = dict(energy_w=1.0, ocean_cfl=0.05)
    for mutate in (None, spike):
        monkeypatch.setenv("QD_TAIL_GENERAL", "1")
        gen, _ = _run(1, nlat, nlon, 3, over, True, True, mutate=mutate)
        monkeypatch.setenv("QD_TAIL_GENERAL", "0")
        fast, _ = _run(1, nlat, nlon, 3, over, True, True, mutate=mutate)
        for k in fast:
            assert np.array_equal(fast[k], gen[k], equal_nan=True), (k, mutate is not None, relerr(fast[k], gen[k]))
        monkeypatch.setenv("QD_TAIL_V", "1")                     # the round-3 kernel: another strip cut, so another order of the eta sum
        old, _ = _run(1, nlat, nlon, 3, over, True, True, mutate=mutate)
        monkeypatch.delenv("QD_TAIL_V")
        if mutate is None:
            for k in fast:
                assert relerr(fast[k], old[k]) < TAIL_TOL, (k, relerr(fast[k], old[k]))


def test_nonfinite_values_fall_back_to_the_exact_path(gpu, monkeypatch):
    """The FAST path of the fused kernels omits nan_to_num and instead detects non-finite values (one v_cmp_class per
    owned output / clip input); a workgroup that sees one recomputes its tile on the EXACT path.  Poison interior, pole and
    ocean cells with NaN / +-inf: the result must equal the EXACT-only run bit for bit (NaN-aware) and the unfused
    reference-order kernels (the reference's nan_to_num placement) to rounding."""
    nlat, nlon = 181, 360

    def poison(st):
        st["U"][90, 100] = np.nan
        st["V"][37, 11] = np.inf
        st["Q"][120, 359] = -np.inf
        st["CLOUD"][1, 5] = np.nan            # next to the south pole
        st["H"][179, 200] = np.inf            # next to the north pole
        st["H"][60, 0] = np.nan               # on the longitude seam
        st["UO"] = np.zeros((nlat, nlon)); st["UO"][100, 200] = np.nan           # ocean fused kernel
        st["ETA"] = np.zeros((nlat, nlon)); st["ETA"][50, 300] = np.inf; st["ETA"][0, 7] = np.nan
    over = dict(energy_w=1.0, ocean_cfl=0.05)
    monkeypatch.setenv("QD_FUSED_FAST", "0")
    exact, _ = _run(1, nlat, nlon, 2, over, True, False, mutate=poison)
    monkeypatch.setenv("QD_FUSED_FAST", "1")
    fast, _ = _run(1, nlat, nlon, 2, over, True, False, mutate=poison)
    monkeypatch.setenv("QD_FUSED", "0")
    unfused, _ = _run(1, nlat, nlon, 2, over, True, False, mutate=poison)
    for k in exact:
        assert np.array_equal(fast[k], exact[k], equal_nan=True), k
        assert np.array_equal(np.isfinite(fast[k]), np.isfinite(unfused[k])), k
        # nan_to_num turns an infinity into +-1.8e308; within reach of such a cell (two steps of a 4-cell stencil + gather)
        # everything saturates and the sign of an overflowed intermediate decides between the two clamps -- the reciprocal-table
        # and literal-division forms need not agree there.  Away from the poisoned neighbourhoods they must.
        far = np.ones((nlat, nlon), dtype=bool)
        for (pi, pj) in ((90, 100), (37, 11), (120, 359), (1, 5), (179, 200), (60, 0), (100, 200), (50, 300), (0, 7)):
            ii = np.arange(max(0, pi - 14), min(nlat, pi + 15))
            jj = np.arange(pj - 14, pj + 15) % nlon
            far[np.ix_(ii, jj)] = False
        bad = ~np.isclose(fast[k], unfused[k], rtol=1e-7, atol=0.0, equal_nan=True) & far
        assert not bad.any(), (k, np.argwhere(bad)[:5], fast[k][bad][:5], unfused[k][bad][:5])


@pytest.mark.parametrize("transport", ["host", "peer"])
def test_bands_ecology_substep(gpu, transport, monkeypatch):
    """The ecology sub-step (qd_step_n bit5) on 3 latitude bands against the whole-globe handle: the LAI-change ratio is a
    band-wise sum (all-reduced), the alpha blend and E_day are pointwise, each sampled individual is advanced by the band
    that owns its cell (no exchange); the per-band energy arrays add up to the whole-globe one."""
    if transport == "peer":                                  # the lai-delta sums and the halo rows through the mailboxes (qd_peer.hip)
        monkeypatch.setenv("QD_PEER_EXCHANGE", "1")
    import os
    from qingdai_amd.bands import BandGroup
    from qingdai_amd.device import Device
    from qingdai_amd.ecology import EcologyAdapter, IndividualPool
    for k in list(os.environ):
        if k.startswith("QD_ECO_"):
            monkeypatch.delenv(k)
    for k, v in {"QD_ECO_NS": "3", "QD_ECO_SUBSTEP_EVERY_NPHYS": "2", "QD_ECO_INDIV_SAMPLE_FRAC": "0.1", "QD_ECO_INDIV_PER_CELL": "4",
                 "QD_ECO_INDIV_SUBSTEPS_PER_DAY": "144", "QD_ECO_LAI_ALBEDO_WEIGHT": "0.7"}.items():
        monkeypatch.setenv(k, v)
    nlat, nlon, nsteps = 61, 96, 5
    qa, grid, mask, alb, fric, p = _setup(nlat, nlon, dict(energy_w=1.0))
    forcing = qa.ThermalForcing(qa.SphericalGrid(nlat, nlon), qa.OrbitalSystem())
    stars = forcing.star_table([i * 300.0 for i in range(nsteps)])
    st = _seed_state(nlat, nlon, 9)
    r = np.random.default_rng(11)
    land = (mask == 1)
    st["W_LAND"] = np.where(land, 45.0 * r.random((nlat, nlon)), 0.0)
    L0 = np.abs(r.normal(0.5, 0.4, (3, 1, nlat, nlon))) * land
    L1 = L0 * (1.0 + 0.5 * r.random(L0.shape))
    static = {"LAND_MASK": mask, "FRICTION": fric, "BASE_ALBEDO": alb}
    names = ["ALBEDO", "ECO_EDAY", "ECO_ALPHA", "TS", "W_LAND"]

    def attach(dev):
        eco = EcologyAdapter(grid, mask, dev=dev, albedo_couple=True)
        eco.pop.push_layers(L0, init=True)
        return eco, IndividualPool(grid, mask, eco)

    def go(dev, eco, n0, n1):
        dev.step_n(stars[n0:n1], 300.0, with_ocean=True, with_physics=True, pass_albedo=False, with_hydrology=True, ecology=True)

    dev = Device(grid, p)
    for k, v in {**static, **st}.items():
        dev.upload_now(k, v)
    eco, pool = attach(dev)
    go(dev, eco, 0, 2); eco.pop.push_layers(L1); go(dev, eco, 2, nsteps)
    ref = {k: dev.get(k).copy() for k in names}
    ref_E, ref_state = pool.indiv_E_day, eco.pop.state()
    dev.close()
    grp = BandGroup(grid, 3, p)
    for k, v in {**static, **st}.items():
        grp.set(k, v)
    att = [attach(d) for d in grp.devs]
    grp.run(lambda d, k: go(d, att[k][0], 0, 2))
    for e_, _ in att:
        e_.pop.push_layers(L1)
    grp.run(lambda d, k: go(d, att[k][0], 2, nsteps))
    got = {k: grp.get(k) for k in names}
    got_E = sum(pl.indiv_E_day for _, pl in att)
    states = [e_.pop.state() for e_, _ in att]
    grp.close()
    print(ref_state, "individual energy max", ref_E.max())
    assert all(s == ref_state for s in states) and ref_state["n_recompute"] == 2
    assert ref_E.max() > 0 and np.array_equal(got_E, ref_E)
    for k in names:
        e = relerr(np.nan_to_num(got[k]), np.nan_to_num(ref[k]))
        assert e < 1e-12, (k, e)
    assert np.array_equal(np.isnan(got["ECO_ALPHA"]), np.isnan(ref["ECO_ALPHA"]))


def test_rccl_transport_equals_in_process_transport(gpu, monkeypatch):
    """The RCCL transport itself, on ONE GPU: a communicator of one rank whose ring neighbours are the rank itself
    (grouped ncclSend / ncclRecv to self, ncclAllReduce of f64 scalars, of the 2050-counter histogram and of the 65 k-word gathered
    median segments) must move exactly the bytes the in-process transport moves.  The band covers 41 of 91 rows, so its halos are
    refreshed with its OWN edge rows -- not a physical configuration, but every collective of the band code runs (halo exchanges,
    eta sums, CFL maxima, precipitation sums, both median paths), and both transports must agree bit for bit on every field."""
    import ctypes
    from qingdai_amd.bands import init_rccl
    from qingdai_amd.device import Device
    monkeypatch.setenv("MASTER_PORT", "29731")
    nlat, nlon, nsteps = 91, 144, 5
    qa, grid, mask, alb, fric, p = _setup(nlat, nlon, dict(energy_w=1.0))
    forcing = qa.ThermalForcing(qa.SphericalGrid(nlat, nlon), qa.OrbitalSystem())
    stars = forcing.star_table([i * 300.0 for i in range(nsteps)])
    st = _seed_state(nlat, nlon, 21)
    static = {"LAND_MASK": mask, "FRICTION": fric, "BASE_ALBEDO": alb}
    names = ["U", "V", "H", "TS", "Q", "CLOUD", "UO", "VO", "ETA", "SST", "ALBEDO", "PRECIP"]
    out, counts = {}, {}
    # scalars on RCCL with the round-3 band sub-step (k_ocn_tail_stream on the segments) / on RCCL with the round-2 sub-step /
    # through the shared-memory host ring
    for ring in (False, "round2", True):
        for transport in ("local", "rccl"):
            monkeypatch.delenv("QD_NO_HOST_RING", raising=False); monkeypatch.delenv("QD_HOST_RING", raising=False)
            monkeypatch.delenv("QD_BAND_TAIL", raising=False)
            if ring is True:
                monkeypatch.setenv("QD_HOST_RING", "1")
            else:
                monkeypatch.setenv("QD_NO_HOST_RING", "1")
            if ring == "round2":
                monkeypatch.setenv("QD_BAND_TAIL", "0")
            dev = Device(qa.SphericalGrid(nlat, nlon), p, row0=25, n_rows=41, halo=12, rank=0, world=1)
            if transport == "local":
                arr = (ctypes.c_void_p * 1)(dev.h)
                assert dev.lib.qd_comm_init_local(arr, 1) == 0
            else:
                init_rccl(dev, 0, 1, tag=f"self{ring}")
            for k, v in {**static, **st}.items():
                dev.upload_now(k, v)
            na0 = ctypes.c_int(0); dev.lib.qd_comm_allreduce_count(dev.h, ctypes.byref(na0))   # init_rccl ends with a barrier
            dev.step_n(stars, 300.0, with_ocean=True, with_physics=True, pass_albedo=True)
            # what bench.py does around its timed region on every rank: barrier, max over ranks
            assert dev.lib.qd_comm_barrier(dev.h) == 0
            v = (ctypes.c_double * 2)(1.5, -2.0)
            assert dev.lib.qd_comm_allreduce_max(dev.h, v, 2) == 0 and list(v) == [1.5, -2.0]
            key = (transport, ring)
            out[key] = {k: dev.get(k)[25:66].copy() for k in names}
            ne, na, nh = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
            dev.lib.qd_comm_stats(dev.h, ctypes.byref(ne)); dev.lib.qd_comm_allreduce_count(dev.h, ctypes.byref(na))
            dev.lib.qd_comm_host_allreduce_count(dev.h, ctypes.byref(nh))
            counts[key] = (ne.value, na.value - na0.value, nh.value)
            dev.close()
    print("halo exchanges, RCCL all-reduces, host-ring all-reduces:", counts)
    for ring in (False, "round2", True):
        assert counts[("local", ring)] == counts[("rccl", ring)], ring
        for k in names:
            assert np.array_equal(out[("local", ring)][k], out[("rccl", ring)][k], equal_nan=True), (k, ring)
            assert np.isfinite(out[("rccl", ring)][k]).all(), k
    assert counts[("rccl", False)][0] > 10 and counts[("rccl", False)][1] > 20 and counts[("rccl", False)][2] == 0
    assert counts[("rccl", True)][2] >= 5 and counts[("rccl", True)][1] < counts[("rccl", False)][1]
    for k in names:                                              # the ring changes where the eta mean is reduced, not its value
        assert relerr(out[("rccl", True)][k], out[("rccl", "round2")][k]) < 1e-12, k
    # (the round-3 sub-step is NOT compared with the other two here: this band's halos hold copies of its own edge rows, so a row
    #  recomputed on the halo differs from the same row after an exchange, and the two sub-steps exchange at different moments --
    #  against the whole globe, where halos are real neighbours, all three agree: test_bands_full_step_with_ocean_and_physics)


@pytest.mark.parametrize("shape", [(64, 64), (91, 144), (181, 360), (33, 130)])
def test_one_launch_shapiro_equals_one_launch_per_pass(gpu, shape, monkeypatch):
    """k_shapiro_stream (all passes in one launch: lon taps by DPP lane shifts, lat taps from registers, rows streamed) against
    k_shapiro_pass once per pass (QD_SHAPIRO_STREAM=0) and against the oracle's scipy-convolve form (dynamics.py:215-231): bit for
    bit, 1-3 passes, with NaN / +-inf cells (scrubbed by the first pass only) in the interior, on the poles and on the seam; the
    strip heights cover one strip per globe, strips that end on a pole row and one-row strips."""
    import qingdai_amd as qa
    from qingdai_amd.device import Device
    from qd_oracle import numerics as onx
    nlat, nlon = shape
    r = np.random.default_rng(nlat * 1000 + nlon)
    F = r.normal(0.0, 10.0, (nlat, nlon))
    F[0, 3] = np.nan; F[nlat - 1, nlon - 1] = np.inf; F[nlat // 2, 0] = -np.inf; F[1, nlon // 2] = np.nan; F[nlat - 2, 7] = 1e308
    monkeypatch.setenv("QD_SHAPIRO_STREAM", "0")
    ref_dev = Device(qa.SphericalGrid(nlat, nlon))
    want = {n: ref_dev.op_shapiro(F, n) for n in (1, 2, 3)}
    ref_dev.close()
    for n in (1, 2, 3):
        assert np.array_equal(want[n], onx.shapiro(F, n)), n
    monkeypatch.setenv("QD_SHAPIRO_STREAM", "1")
    for R in ("16", "1", "7", str(nlat), str(nlat - 1)):
        monkeypatch.setenv("QD_SHAPIRO_R", R)
        dev = Device(qa.SphericalGrid(nlat, nlon))
        for n in (1, 2, 3):
            got = dev.op_shapiro(F, n)
            assert np.array_equal(got, want[n]), (R, n, np.argwhere(got != want[n])[:5])
        dev.close()


@pytest.mark.parametrize("tail", TAIL_FORMS)
@pytest.mark.parametrize("outlier", ["mean4", "clamp"])
def test_ocean_tail_outlier_filter_with_isolated_spikes(gpu, monkeypatch, outlier, tail):
    """The velocity outlier filter of the ocean sub-step (ocean.py:409-434) with ISOLATED spikes: a cell faster than QD_OCEAN_MAX_U
    whose four neighbours are slow takes the `mean4` branch alone in its wavefront, so its east / west neighbours must not come
    from lanes that skipped the branch.  Streaming tail kernel (lane neighbours) against the two-launch form (neighbours from
    memory) and against the oracle, spikes in the interior, on the seam, next to a pole and in adjacent columns."""
    import qd_oracle as qo
    from qingdai_amd.device import Device
    nlat, nlon = 91, 144
    qa, grid, mask, alb, fric, p = _setup(nlat, nlon, dict(energy_w=1.0, ocean_cfl=0.05, ocean_outlier=outlier))
    r = np.random.default_rng(8)
    ocean = mask == 0
    uo = r.normal(0, 0.2, (nlat, nlon)) * ocean
    vo = r.normal(0, 0.2, (nlat, nlon)) * ocean
    cells = [(i, j) for i, j in [(45, 70), (45, 71), (30, 0), (30, nlon - 1), (1, 40), (nlat - 2, 100), (60, 63), (60, 64), (20, 20)] if ocean[i, j]]
    assert len(cells) >= 5
    for k, (i, j) in enumerate(cells):
        uo[i, j] = 9.0 * (-1) ** k
        vo[i, j] = 7.0
    st = _seed_state(nlat, nlon, 9)
    lat = np.linspace(-np.pi / 2, np.pi / 2, nlat)[:, None]
    sst = 288.0 + 8.0 * np.cos(lat) ** 2 + r.normal(0, 0.3, (nlat, nlon))
    out = {}
    for mode in ("0", tail):
        _set_tail(monkeypatch, mode)
        oc = qa.WindDrivenSlabOcean(qa.SphericalGrid(nlat, nlon), mask, 50.0, init_Ts=sst, params=p)
        oc.uo, oc.vo, oc.eta = uo, vo, np.zeros((nlat, nlon))
        oc.step(300.0, st["U"], st["V"])
        assert oc.last_n_sub >= 2
        out[mode] = {"UO": oc.uo.copy(), "VO": oc.vo.copy(), "ETA": oc.eta.copy(), "SST": oc.Ts.copy()}
        oc._dev.close()
    # third party: the oracle's ocean step from the same state and winds -- the bug this test was written for lived in one of the
    # two DEVICE forms
    oo = qo.OceanOracle(qo.Grid(nlat, nlon), mask, qo.defaults(energy_w=1.0, ocean_cfl=0.05, ocean_outlier=outlier), init_Ts=sst.copy())
    oo.uo, oo.vo, oo.eta = uo.copy(), vo.copy(), np.zeros((nlat, nlon))
    oo.step(300.0, st["U"], st["V"])
    assert oo.last_n_sub >= 2
    want = {"UO": oo.uo, "VO": oo.vo, "ETA": oo.eta, "SST": oo.Ts}
    assert np.max(np.hypot(want["UO"], want["VO"])) <= 3.0 + 1e-9             # nothing is left above QD_OCEAN_MAX_U
    for k in want:
        for mode in ("0", tail):
            e = relerr(out[mode][k], want[k])
            assert e < 1e-10, ("vs oracle", outlier, mode, k, e)
    for k in out["0"]:
        e = relerr(out[tail][k], out["0"][k])
        bad = np.argwhere(np.abs(out[tail][k] - out["0"][k]) > 1e-9)
        assert e < TAIL_TOL, (outlier, k, e, sorted(set(int(b[0]) for b in bad))[:12], [tuple(b) for b in bad if 0 < b[0] < nlat - 1][:12])


@pytest.mark.parametrize("over", [dict(K_h=0.0), dict(ocean_use_qnet=0), dict(ocean_ice_qfac=0.0), dict(ocean_adv_alpha=1.0),
                                  dict(ocean_adv_alpha=0.0, K_h=2.0e4), dict(eta_cap=0.05), dict(ocean_cfl=0.9)],
                         ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
@pytest.mark.parametrize("tail", TAIL_FORMS)
def test_ocean_tail_kernel_parameter_branches(gpu, monkeypatch, over, tail):
    """The wave-uniform switches of the streaming tail kernel (no diffusion, no Q_net heating, no heating under ice, pure advection /
    no advection, a tight eta clip, one sub-step per step) against the two-launch form, 2 coupled steps at 91 x 144."""
    base = dict(energy_w=1.0, ocean_cfl=0.05)
    base.update(over)
    _set_tail(monkeypatch, "0")
    two, _ = _run(1, 91, 144, 2, base, True, True)
    _set_tail(monkeypatch, tail)
    one, _ = _run(1, 91, 144, 2, base, True, True)
    for k in one:
        e = relerr(one[k], two[k])
        assert e < TAIL_TOL, (over, k, e)


@pytest.mark.parametrize("shape", [(25, 64), (37, 130), (50, 200), (97, 257), (13, 70)])
def test_streaming_kernels_at_awkward_grid_sizes(gpu, shape, monkeypatch):
    """Strip geometry of the row-streaming kernels where nothing divides evenly: a single strip per column of strips, strips of
    12-13 rows, a last column strip of a few columns, an odd number of longitudes, the smallest grid the streaming path accepts.
    Streaming momentum + del^4 kernels against the LDS-tile kernels' EXACT path bit for bit; streaming ocean tail (with the eta mean
    finished inside the launch) against the two-launch form to the rounding of the eta sum; 3 coupled steps with driver physics."""
    nlat, nlon = shape
    over = dict(energy_w=1.0, ocean_cfl=0.05)
    monkeypatch.setenv("QD_OCN_TAIL", "0")
    monkeypatch.setenv("QD_FUSED_FAST", "0")
    lds, _ = _run(1, nlat, nlon, 3, over, True, True)
    monkeypatch.setenv("QD_FUSED_FAST", "1")
    stream, _ = _run(1, nlat, nlon, 3, over, True, True)
    for k in lds:
        assert np.array_equal(stream[k], lds[k]), (shape, k, relerr(stream[k], lds[k]))
    monkeypatch.setenv("QD_OCN_TAIL", "1")
    tail, _ = _run(1, nlat, nlon, 3, over, True, True)
    for k in lds:
        e = relerr(tail[k], lds[k])
        assert e < 1e-11, (shape, k, e)


@pytest.mark.parametrize("halo", [5, 6, 7, 9])
def test_ocean_step_on_thin_halo_bands(gpu, halo, monkeypatch):
    """WindDrivenSlabOcean.step (pygcm/ocean.py:265-533) on 3 latitude bands whose halo is at or just above the thinnest width
    qd_create accepts: with fewer than max(7, Ro + 2) halo rows the band sub-step falls back to the round-2 kernels (the round-3
    form plans eta 7 rows out) instead of failing in the planner; every width must reproduce the whole globe."""
    from qingdai_amd.bands import BandGroup
    from qingdai_amd.device import Device
    monkeypatch.setenv("QD_NO_HOST_RING", "1")
    nlat, nlon = 91, 144
    qa, grid, mask, alb, fric, p = _setup(nlat, nlon, dict(energy_w=1.0, ocean_cfl=0.05))
    st = _seed_state(nlat, nlon, 5)
    r = np.random.default_rng(9)
    lat = np.linspace(-np.pi / 2, np.pi / 2, nlat)[:, None]
    oc = {"UO": 0.3 * np.cos(lat) * r.normal(0, 1, (nlat, nlon)), "VO": 0.2 * r.normal(0, 1, (nlat, nlon)),
          "ETA": 0.5 * r.normal(0, 1, (nlat, nlon)), "SST": 285.0 + 10 * np.cos(lat) + r.normal(0, 0.5, (nlat, nlon)),
          "QNET": r.normal(0, 50.0, (nlat, nlon))}
    static = {"LAND_MASK": mask, "FRICTION": fric, "BASE_ALBEDO": alb}
    names = ["UO", "VO", "ETA", "SST", "TS"]

    def run(dev):
        for _ in range(2):
            dev.ocean_step(300.0, False, False, True)
    ref_dev = Device(grid, p)
    for k, v in {**static, **st, **oc}.items():
        ref_dev.upload_now(k, v)
    run(ref_dev)
    ref = {k: ref_dev.get(k).copy() for k in names}
    nsub = ref_dev.last_ocean_nsub()
    ref_dev.close()
    grp = BandGroup(grid, 3, p, halo=halo)
    for k, v in {**static, **st, **oc}.items():
        grp.set(k, v)
    grp.run(lambda d, rk: run(d))
    got = {k: grp.get(k) for k in names}
    grp.close()
    assert nsub >= 2
    for k in names:
        e = relerr(got[k], ref[k])
        assert e < 1e-12, (halo, k, e)
