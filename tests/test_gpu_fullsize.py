"""GPU (-m gpu): BASELINE's full size (721 x 1440) through size-independent properties -- the oracle needs ~2 s per
step there, so nothing here steps it.  Operators: exact constants, linearity, projection property of the zonal
filter, exact medians against numpy; the fused kernels' two code paths bit for bit at the tile shape the benchmark
really uses (TR = 38: interior tiles AND shifted-plane pole tiles); 8 latitude bands against the whole globe."""
import numpy as np
import pytest

from util import relerr
from test_gpu_bands import _run

pytestmark = pytest.mark.gpu
NLAT, NLON = 721, 1440


@pytest.fixture(scope="module")
def ops(gpu):
    import qingdai_amd as qa
    return qa.SphericalGrid(NLAT, NLON)._ops()


def _fields(seed):
    r = np.random.default_rng(seed)
    lat = np.linspace(-np.pi / 2, np.pi / 2, NLAT)[:, None]
    lon = np.linspace(0, 2 * np.pi, NLON)[None, :]
    F = 8000.0 + 300.0 * np.sin(lat) ** 2 + 40.0 * np.cos(lat) * np.cos(3 * lon) + r.normal(0, 2.0, (NLAT, NLON))
    Gf = 280.0 + 30.0 * np.cos(lat) ** 2 + r.normal(0, 1.0, (NLAT, NLON))
    u = 25.0 * np.cos(lat) * np.sin(2 * lon) + r.normal(0, 3.0, (NLAT, NLON))
    v = 8.0 * np.sin(2 * lat) * np.cos(3 * lon) + r.normal(0, 2.0, (NLAT, NLON))
    return F, Gf, u, v


def test_constants_are_fixed_points(ops):
    c = np.full((NLAT, NLON), 273.15)
    _, _, u, v = _fields(1)
    assert np.all(ops.op_laplacian(c) == 0.0) and np.all(ops.op_laplacian(c, ocean=True) == 0.0)
    assert np.array_equal(ops.op_hyperdiffuse(c, 1.0e14, 300.0, 2), c)
    assert np.array_equal(ops.op_shapiro(c, 2), c)
    assert relerr(ops.op_advect(c, u, v, 300.0), c) < 1e-15            # bilinear weights sum to 1 up to rounding
    assert relerr(ops.op_gaussian(c, 1.0), c) < 1e-15
    assert relerr(ops.op_zonal_filter(c, 0.75, 0.5), c) < 1e-13         # direct DFT of 1440 terms: ~n eps of the row maximum
    assert np.all(ops.op_divvort(np.zeros_like(c), np.zeros_like(c)) == 0.0)


def test_operators_are_linear(ops):
    F, Gf, u, v = _fields(2)
    a, b = 0.75, -1.25
    for name, op in (("lap", lambda X: ops.op_laplacian(X)), ("lap_ocn", lambda X: ops.op_laplacian(X, ocean=True)),
                     ("shapiro", lambda X: ops.op_shapiro(X, 2)), ("gauss", lambda X: ops.op_gaussian(X, 1.0)),
                     ("advect", lambda X: ops.op_advect(X, u, v, 300.0)), ("zonal", lambda X: ops.op_zonal_filter(X, 0.75, 0.5)),
                     ("hyper", lambda X: ops.op_hyperdiffuse(X, 1.0e14, 300.0, 1))):
        lhs = op(a * F + b * Gf)
        rhs = a * op(F) + b * op(Gf)
        assert relerr(lhs, rhs) < 2e-12, name
    d = ops.op_divvort(a * u + b * v, a * v - b * u) - (a * ops.op_divvort(u, v) + b * ops.op_divvort(v, -u))
    assert np.max(np.abs(d)) < 1e-12 * np.max(np.abs(ops.op_divvort(u, v)))


def test_zonal_filter_is_a_projection_and_keeps_zonal_means(ops):
    F, _, _, _ = _fields(3)
    once = ops.op_zonal_filter(F, 0.6, 1.0)                            # damp = 1: the high bins are removed entirely
    twice = ops.op_zonal_filter(once, 0.6, 1.0)
    assert relerr(twice, once) < 1e-13
    assert np.max(np.abs(once.mean(axis=1) - F.mean(axis=1))) < 1e-12 * np.abs(F).max()
    spec = np.abs(np.fft.rfft(once, axis=1))
    kcut = int(0.6 * (NLON // 2))
    assert spec[:, kcut:].max() < 1e-9 * spec[:, 0].max()              # nothing left above the cutoff


def test_median_exact_at_full_size(ops):
    r = np.random.default_rng(4)
    x = np.exp(r.normal(-11, 2.5, (NLAT, NLON))) * (r.random((NLAT, NLON)) < 0.7)
    x[r.random(x.shape) < 0.02] *= -1.0
    assert ops.op_median_positive(x, 1e-6) == float(np.median(x[x > 0]))
    x = np.round(x, 6)                                                  # heavy ties
    assert ops.op_median_positive(x, 1e-6) == float(np.median(x[x > 0]))


def test_fused_paths_agree_at_benchmark_tile_shape(gpu, monkeypatch):
    over = dict(energy_w=1.0)
    monkeypatch.setenv("QD_FUSED_FAST", "0")
    exact, _ = _run(1, NLAT, NLON, 2, over, True, True)
    monkeypatch.setenv("QD_FUSED_FAST", "1")
    fast, _ = _run(1, NLAT, NLON, 2, over, True, True)
    for k in exact:
        assert np.array_equal(fast[k], exact[k]), (k, relerr(fast[k], exact[k]))
    monkeypatch.setenv("QD_FUSED", "0")
    unfused, _ = _run(1, NLAT, NLON, 2, over, True, True)
    for k in fast:
        assert relerr(fast[k], unfused[k]) < (1e-7 if k in ("UO", "VO", "ETA") else 1e-9), k


@pytest.mark.parametrize("transport", ["host", "peer"])
def test_eight_bands_match_the_whole_globe(gpu, transport, monkeypatch):
    """721 x 1440 in 8 latitude bands against the whole globe.  host: the in-process transport (device-to-device copies, host ring);
    peer: the device-side exchange over the peer mapping (qd_peer.hip: halo rows and global sums through the mailboxes, the ocean
    momentum kernel split into interior / boundary rows around every exchange) -- the transport a multi-GPU run takes."""
    if transport == "peer":
        monkeypatch.setenv("QD_PEER_EXCHANGE", "1")
        monkeypatch.setenv("QD_PEER_OVERLAP", "2")
    ref, _ = _run(1, NLAT, NLON, 2, dict(energy_w=1.0), True, True)
    got, ex = _run(8, NLAT, NLON, 2, dict(energy_w=1.0), True, True)
    print("halo exchanges per band:", ex)
    for k in ("U", "V", "H", "TS", "Q", "CLOUD"):
        assert np.array_equal(got[k], ref[k]), (k, relerr(got[k], ref[k]))
    for k in ("UO", "VO", "ETA", "SST"):
        assert relerr(got[k], ref[k]) < 1e-13, k                       # band-wise order of the eta sum


def test_two_rank_processes_at_the_benchmark_size(gpu):
    """`bench.py --gpus 2`'s decomposition as two rank PROCESSES on device 0 (IPC-mapped mailboxes, preferred halo, exchange overlapped
    with the interior rows: scripts/peer_ranks.py) against the whole globe at 721 x 1440, coupled: atmosphere bit for bit, ocean to the
    band-wise order of the eta sums."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "peer_ranks.py"), "--world", "2", "--nlat", str(NLAT), "--nlon", str(NLON),
                        "--steps", "2", "--ocean", "--preferred-halo", "--tol", "1e-12"], env=dict(os.environ, QD_PEER_EXCHANGE="1"),
                       capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    res = json.loads(line[-1])
    print({k: res[k] for k in ("ok", "atmosphere_bitwise", "max_rel_err")}, [m["halo_exchanges"] for m in res["ranks"]])
    assert res["ok"] and res["atmosphere_bitwise"] and all(m["transport"] == "peer" for m in res["ranks"])


def test_ecology_substep_full_size(gpu, monkeypatch):
    """BASELINE configs[4] at 721 x 1440: 20 species planes, 16 bands, ~6 k sampled cells x 150 individuals inside the resident
    loop.  Properties: E_day is the step-by-step sum of isr dt (same star row every step: exact repeated addition); the alpha map
    is NaN off land and inside [min(leaf, soil), max(leaf, soil)] on it, and equals the closed form of the uploaded LAI; the
    albedo changes only where the blend may act (land without ice sheet); one individual sub-step equals the oracle's
    gather + einsum on the sampled cells."""
    import os
    import qingdai_amd as qa
    import qd_oracle as qo
    from qd_oracle import ecology as oeco, spectral as osp
    from qingdai_amd.driver import Simulation
    for k in list(os.environ):
        if k.startswith("QD_ECO_"):
            monkeypatch.delenv(k)
    monkeypatch.setenv("QD_ECO_INDIV_SUBSTEPS_PER_DAY", "80")          # period = 900 s: the third 300 s step fires
    sim = Simulation(NLAT, NLON, params=qa.QdParams(), use_ocean=True, quiet=True)
    land = (sim.land_mask == 1)
    assert sim.indiv.n_indiv == int(0.02 * land.sum()) * 150 and sim.eco.pop.LAI_layers_SK.shape[:2] == (20, 1)
    r = np.random.default_rng(3)
    L = np.abs(r.normal(0.3, 0.3, (20, 1, NLAT, NLON))) * land
    sim.eco.pop.push_layers(L, init=True)
    Ltot = np.sum(L, axis=(0, 1))
    assert np.array_equal(sim.eco.pop.total_LAI(), Ltot)
    W0 = np.where(land, 50.0 * r.random((NLAT, NLON)), 0.0)
    sim.dev.set("W_LAND", W0)
    # same star row for all steps -> the resident ISR is the same field every step
    star = sim.forcing.star_table([1.0e5])
    n = 4
    sim.dev.step_n(np.repeat(star, n, axis=0), 300.0, with_ocean=True, with_physics=True, pass_albedo=False, with_hydrology=False,
                   ecology=True)
    isr = sim.dev.get("ISR").copy()
    acc = np.zeros_like(isr)
    for _ in range(n):
        acc += isr * 300.0
    assert np.array_equal(sim.eco.pop.E_day, acc) and acc.max() > 0
    alpha = sim.dev.get("ECO_ALPHA")
    leaf, soil = sim.eco.params.leaf_scalar, sim.eco.params.soil_ref
    assert np.all(np.isnan(alpha[~land])) and np.all(np.isfinite(alpha[land]))
    assert alpha[land].min() >= min(leaf, soil) - 1e-15 and alpha[land].max() <= max(leaf, soil) + 1e-15
    f = 1.0 - np.exp(-0.5 * np.maximum(Ltot, 0.0))
    assert np.max(np.abs(alpha[land] - np.clip(leaf * f + (1.0 - f) * soil, 0, 1)[land])) < 1e-15
    st = sim.eco.pop.state()
    assert st["step_count"] == n and st["n_recompute"] == 1            # untouched layers: one canopy build, then cached
    # individuals: exactly one sub-step fired (accum 300, 600, 900 >= 900), on this ISR and the initial W_land
    pool = sim.indiv
    ob = osp.make_bands(16, 380.0, 780.0)
    oi = oeco.IndividualSubstep(pool.sample_j, pool.sample_i, pool.indiv_cell_index, pool.indiv_Ab, pool.indiv_tol, 80)
    oi.period, oi.accum = sim.day_seconds / 80.0, sim.day_seconds / 80.0 - 300.0
    assert oi.try_substep(sim.dev.get("ISR_A"), sim.dev.get("ISR_B"), ob, np.clip(W0 / 50.0, 0, 1), 300.0, sim.day_seconds)
    E, S = pool.indiv_E_day, pool.indiv_water_stress_days
    assert E.max() > 0 and relerr(E, oi.E_day) < 1e-14 and np.array_equal(S, oi.stress_days)
    # the blend only acts on land that is not an ice sheet
    alb_eco = sim.dev.get("ALBEDO").copy()
    glacier = sim.dev.get("GLACIER") != 0.0
    sim2 = Simulation(NLAT, NLON, params=qa.QdParams(), use_ocean=True, quiet=True, ecology=False)
    sim2.dev.set("W_LAND", W0)
    sim2.dev.step_n(star, 300.0, with_ocean=True, with_physics=True, pass_albedo=False, with_hydrology=False)
    sim3 = Simulation(NLAT, NLON, params=qa.QdParams(), use_ocean=True, quiet=True, individuals=False)
    sim3.eco.pop.push_layers(L, init=True)
    sim3.dev.step_n(star, 300.0, with_ocean=True, with_physics=True, pass_albedo=False, with_hydrology=False, ecology=True)
    a2, a3 = sim2.dev.get("ALBEDO"), sim3.dev.get("ALBEDO")
    gl3 = sim3.dev.get("GLACIER") != 0.0
    assert np.array_equal(a2[~land | gl3], a3[~land | gl3])
    assert np.abs(a2 - a3)[land & ~gl3].max() > 0.01
    for s_ in (sim, sim2, sim3):
        s_.dev.close()
