"""CPU: where the coupled loop at the benchmark size is chaotic IN THE REFERENCE ARITHMETIC ITSELF.

tests/test_gpu_atsize.py compares the device with the oracle after one and two coupled steps at 721 x 1440, but not on the rows next to
the poles.  This test is the measurement behind that window: the oracle against the oracle with the initial zonal wind scaled by
(1 + 1e-15) -- one ulp-level perturbation, one coupled step (12 ocean sub-steps).  The two polar ocean rows sit at the +-5 m clip of
eta and flip sign there, so the two runs differ by O(1) next to the poles; the stencils carry that at most a few rows per sub-step.
Asserted: (1) the O(1) differences exist (the window is not an excuse), (2) they are confined to rows 0 .. POLAR_ROWS-1 and
n-POLAR_ROWS .. n-1, (3) every other row agrees to the bounds the GPU test uses, eta to 1e-6 (every sub-step subtracts the global
mean, which carries the polar noise into every cell as a uniform shift)."""
import numpy as np

from util import relerr

POLAR_ROWS = 16          # rows next to each pole that tests/test_gpu_atsize.py leaves out after the first coupled step


def build_oracle(scale):
    """The oracle half of tests/test_gpu_atsize.py::_build (same planet, same spun-up wind)."""
    import qd_oracle as qo
    from qd_oracle.driver import DriverOracle
    from qingdai_amd.grid import SphericalGrid
    from qingdai_amd.topography import create_land_sea_mask, generate_base_properties
    nlat, nlon = 721, 1440
    over = dict(energy_w=1.0, cloud_couple=1)
    grid = SphericalGrid(nlat, nlon)
    mask = create_land_sea_mask(grid)
    base_albedo, friction = generate_base_properties(mask)
    Cs_ocean = 1000.0 * 4200.0 * 50.0
    csmap = np.where(mask == 1, 3e6, Cs_ocean).astype(float)
    lat = np.deg2rad(grid.lat_mesh); lon = np.deg2rad(grid.lon_mesh)
    u0, v0 = atsize_wind(lat, lon)
    g = qo.Grid(nlat, nlon)
    P = qo.defaults(**over)
    om = qo.AtmosOracle(g, friction, mask, P, C_s_map=csmap)
    om.u, om.v = u0 * scale, v0.copy()
    oo = qo.OceanOracle(g, mask, P, init_Ts=np.full((nlat, nlon), 288.0))
    return om, oo, DriverOracle(g, om, oo, qo.Forcing(g), mask, base_albedo, P)


def atsize_wind(lat, lon):
    """A spun-up wind field (the bench reaches it after its warm-up steps): after the first time_step the components sit at the
    +-200 m/s clip over wide areas, which puts the slab ocean at the 12-13 sub-steps per step the benchmark runs with."""
    return 340.0 * np.cos(lat) * (1.0 + 0.08 * np.sin(3 * lon)), 170.0 * np.sin(2 * lat) * np.cos(2 * lon)


def test_polar_rows_are_chaotic_in_the_reference_arithmetic_and_nothing_else_is():
    runs = []
    for scale in (1.0, 1.0 + 1e-15):
        om, oo, d = build_oracle(scale)
        d.step(0.0, 300.0, pass_albedo=True, commit=False)
        runs.append({"u": om.u, "v": om.v, "h": om.h, "T_s": om.T_s, "q": om.q, "cloud": om.cloud_cover, "uo": oo.uo, "vo": oo.vo,
                     "eta": oo.eta, "SST": oo.Ts, "n_sub": oo.last_n_sub})
    a, b = runs
    assert a["n_sub"] == b["n_sub"] and a["n_sub"] >= 12, (a["n_sub"], b["n_sub"])
    n = a["u"].shape[0]
    inner = slice(POLAR_ROWS, n - POLAR_ROWS)
    caps = np.r_[0:POLAR_ROWS, n - POLAR_ROWS:n]
    errs_in = {k: relerr(a[k][inner], b[k][inner]) for k in a if k != "n_sub"}
    errs_cap = {k: relerr(a[k][caps], b[k][caps]) for k in ("uo", "vo", "eta")}
    print("n_sub", a["n_sub"], "inner", errs_in, "caps", errs_cap)
    # which rows differ by more than 1e-6 of the field's max-norm at all?
    for k in ("uo", "vo", "eta"):
        d = np.max(np.abs(a[k] - b[k]), axis=1) / max(np.max(np.abs(a[k])), 1e-300)
        rows = np.nonzero(d > 1e-6)[0]
        print(k, "rows differing by > 1e-6:", rows.tolist())
        assert rows.size == 0 or (rows.max() >= n - POLAR_ROWS or rows.min() < POLAR_ROWS)
        assert all(r < POLAR_ROWS or r >= n - POLAR_ROWS for r in rows), (k, rows)
    assert max(errs_cap.values()) > 1e-3, errs_cap                     # (1) the polar rows really are chaotic
    for k, e in errs_in.items():                                       # (3) and nothing else is
        assert e < (1e-6 if k == "eta" else 1e-9), (k, e)
