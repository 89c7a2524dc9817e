"""CPU: restart NetCDF writer/reader of the driver (run_simulation.py:63-183) round-trips through the
classic-NetCDF fallback, with the reference's variable names and f4 storage."""
import numpy as np


class _FakeDev:
    def __init__(self, shape):
        r = np.random.default_rng(0)
        self.data = {}
        self.shape = shape
        self.r = r

    def get(self, name):
        if name not in self.data:
            self.data[name] = self.r.normal(0, 1, self.shape)
        return self.data[name]


def test_restart_roundtrip(tmp_path):
    from qingdai_amd.driver import save_restart, load_restart, RESTART_VARS
    from qingdai_amd.grid import SphericalGrid
    g = SphericalGrid(19, 36)
    dev = _FakeDev((19, 36))
    mask = (np.random.default_rng(1).random((19, 36)) > 0.7).astype(np.uint8)
    path = str(tmp_path / "restart.nc")
    save_restart(path, g, dev, 12345.5, mask)
    rst = load_restart(path)
    assert rst["t_seconds"] == 12345.5
    assert np.array_equal(rst["land_mask"].astype(np.uint8), mask)
    assert rst["land_mask"].dtype == np.float32                    # the reference writes the mask through its f4 helper too (:111)
    for name, fid in RESTART_VARS.items():
        assert rst[name].dtype == np.float32                       # the reference stores f4
        assert np.array_equal(rst[name], dev.get(fid).astype(np.float32)), name
    # no ocean -> no ocean variables in the file (run_simulation.py:99-103)
    path2 = str(tmp_path / "restart_noocean.nc")
    save_restart(path2, g, dev, 1.0, mask, with_ocean=False)
    rst2 = load_restart(path2)
    assert not any(k in rst2 for k in ("uo", "vo", "eta", "Ts")) and "u" in rst2 and "W_land" in rst2


def test_periodic_autosave_fires_on_schedule_whatever_the_chunking():
    """`done % autosave_steps == 0` with chunks of min(200, autosave_steps) fired every 7200 steps for a 288-step
    interval.  The loop now runs exactly up to each time threshold (run_simulation.py:1751-1764: QD_ECO_AUTOSAVE_EVERY_HOURS
    planetary hours, default 6 = a quarter planet-day = 60 steps of 300 s)."""
    from qingdai_amd.driver import chunk_until
    dt, day = 300.0, 72000.0
    for hours in (6.0, 28.8, 0.05):
        every = hours * day / 24.0
        t, nxt, done, saves, total = 0.0, every, 0, [], 1500
        while done < total:
            n = chunk_until(t, dt, nxt, total - done)
            assert 1 <= n <= 200
            t += n * dt
            done += n
            if t >= nxt - 1e-9 * dt and done < total:
                saves.append(done)
                while nxt <= t + 1e-9 * dt:
                    nxt += every
        want = []
        k = 1
        while True:                                                    # first step count whose end time reaches k * every
            s_ = int(np.ceil(k * every / dt - 1e-9))
            if s_ >= total:
                break
            if not want or s_ > want[-1]:
                want.append(s_)
            k += 1
        assert saves == want, (hours, saves[:5], want[:5])
    assert chunk_until(0.0, dt, None, 1000) == 200                     # no periodic autosave: plain chunks


def test_ocean_file_roundtrip(tmp_path):
    """data/ocean.nc (run_simulation.py:185-246): four f4 fields + the `day` attribute; loaders never raise."""
    from qingdai_amd.driver import save_ocean, load_ocean
    from qingdai_amd.grid import SphericalGrid
    g = SphericalGrid(19, 36)
    dev = _FakeDev((19, 36))
    path = str(tmp_path / "data" / "ocean.nc")
    assert save_ocean(path, g, dev, day_value=12.25)
    oc = load_ocean(path)
    assert oc["day"] == 12.25
    for name, fid in (("uo", "UO"), ("vo", "VO"), ("eta", "ETA"), ("Ts", "SST")):
        assert oc[name].dtype == np.float32 and np.array_equal(oc[name], dev.get(fid).astype(np.float32))
    missing = load_ocean(str(tmp_path / "nope.nc"))
    assert all(v is None for v in missing.values())


def test_topography_file_roundtrip_and_regrid(tmp_path):
    """data/topography.nc (run_simulation.py:126-159, pygcm/topography.py:349-575): exact-grid read returns the
    stored maps; a coarser source is regridded bilinearly (nearest for the mask), cyclic in longitude; a source
    with longitudes in [-180, 180) and descending latitudes is normalised first."""
    from qingdai_amd import topography as topo
    from qingdai_amd.grid import SphericalGrid
    g = SphericalGrid(19, 36)
    mask, elev = topo.create_land_sea_mask(g, return_elevation=True)
    alb, fric = topo.generate_base_properties(mask)
    p = str(tmp_path / "topography.nc")
    topo.export_topography_to_netcdf(p, g, mask, alb, fric, elevation=elev)
    e2, m2, a2, f2 = topo.load_topography_from_netcdf(p, g, quiet=True)
    # the model grid's longitudes run 0..360 inclusive (SphericalGrid: linspace(0, 360, n_lon)): like the reference
    # loader, the duplicated seam column is dropped and the file is regridded -- nodes coincide (to the f4 rounding
    # of the stored coordinates), the last column becomes the cyclic image of the first
    assert m2.dtype == np.uint8 and np.array_equal(m2[:, :-1], mask[:, :-1]) and np.array_equal(m2[:, -1], m2[:, 0])
    assert np.allclose(e2[:, :-1], elev[:, :-1], rtol=1e-4, atol=1e-3 * np.abs(elev).max())
    assert np.allclose(a2[:, :-1], alb[:, :-1], atol=1e-5) and np.allclose(f2[:, :-1], fric[:, :-1], rtol=1e-4)
    # no elevation variable -> None (the procedural planet of the reference driver has none)
    p2 = str(tmp_path / "noelev.nc")
    topo.export_topography_to_netcdf(p2, g, mask, alb, fric)
    assert topo.load_topography_from_netcdf(p2, g, quiet=True)[0] is None
    # regrid: a smooth analytic field on a different grid comes back within the bilinear error
    src = SphericalGrid(37, 72)
    f = 1000.0 * np.cos(np.deg2rad(src.lat_mesh)) * (1.0 + 0.3 * np.sin(np.deg2rad(src.lon_mesh)))
    msk = (src.lat_mesh > 10).astype(np.uint8)
    p3 = str(tmp_path / "coarse.nc")
    topo.export_topography_to_netcdf(p3, src, msk, np.full(src.lat_mesh.shape, 0.2), np.full(src.lat_mesh.shape, 1e-5), elevation=f)
    e3, m3, a3, _ = topo.load_topography_from_netcdf(p3, g, quiet=True)
    want = 1000.0 * np.cos(np.deg2rad(g.lat_mesh)) * (1.0 + 0.3 * np.sin(np.deg2rad(g.lon_mesh)))
    assert np.max(np.abs(e3 - want)) < 5.0 and np.allclose(a3, 0.2, atol=1e-6)
    assert np.array_equal(m3, (g.lat_mesh > 10).astype(np.uint8))
    with np.testing.assert_raises(ValueError):
        topo.load_topography_from_netcdf(p3, g, regrid="never", quiet=True)
    # [-180, 180) longitudes + descending latitudes
    from qingdai_amd import ncio
    lat_d = np.linspace(90.0, -90.0, 37)
    lon_w = np.linspace(-180.0, 175.0, 72)
    LON, LAT = np.meshgrid(np.mod(lon_w, 360.0), lat_d)
    fw = 1000.0 * np.cos(np.deg2rad(LAT)) * (1.0 + 0.3 * np.sin(np.deg2rad(LON)))
    ncio.write_nc(str(tmp_path / "w.nc"), {"lat": 37, "lon": 72},
                  {"lat": ("f4", ("lat",), lat_d.astype(np.float32)), "lon": ("f4", ("lon",), lon_w.astype(np.float32)),
                   "land_mask": ("u1", ("lat", "lon"), (LAT > 10).astype(np.uint8)),
                   "base_albedo": ("f4", ("lat", "lon"), np.full((37, 72), 0.2)),
                   "friction": ("f4", ("lat", "lon"), np.full((37, 72), 1e-5)), "elevation": ("f4", ("lat", "lon"), fw)})
    e4, m4, _, _ = topo.load_topography_from_netcdf(str(tmp_path / "w.nc"), g, quiet=True)
    assert np.max(np.abs(e4 - want)) < 5.0 and np.array_equal(m4, (g.lat_mesh > 10).astype(np.uint8))


def test_greenhouse_autotune_matches_oracle():
    """energy.autotune_greenhouse_params (energy.py:544-579): proportional nudge, bounds, and that it moves the DRIVER's
    copy (qnet_lw_*) and leaves the model's lw_eps0 / lw_kc alone (run_simulation.py:2242-2246)."""
    import math
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from qd_oracle import column as col
    import qingdai_amd as qa
    from qingdai_amd import energy
    p = qa.QdParams()
    e0, k0 = p.lw_eps0, p.lw_kc
    assert math.isnan(p.qnet_lw_eps0) and math.isnan(p.qnet_lw_kc)
    eps, kc = e0, k0
    for toa in (12.5, -3.0, 4000.0, -9000.0, -9000.0, 0.0):
        energy.autotune_greenhouse_params(p, {"TOA_net": toa}, verbose=False)
        eps, kc = col.autotune_greenhouse(eps, kc, {"TOA_net": toa})
        assert (p.qnet_lw_eps0, p.qnet_lw_kc) == (eps, kc)
        assert 0.30 <= eps <= 0.98 and 0.0 <= kc <= 0.80
    assert (p.lw_eps0, p.lw_kc) == (e0, k0)


def test_phyto_tracers_initial_state_and_plankton_nc(tmp_path, monkeypatch):
    """PhytoTracers without a device: the initial state of phyto.py:253-270 (fractions of chl0 over the ocean, 0 on land), the
    QD_PHYTO_* knobs, and the C_phyto_s round trip through data/plankton.nc (f4, dims species/lat/lon like phyto.py:752-765)."""
    import qingdai_amd as qa
    from qingdai_amd.phyto import PhytoTracers
    from qingdai_amd.topography import create_land_sea_mask
    from qingdai_amd import ncio
    grid = qa.SphericalGrid(19, 36)
    mask = create_land_sea_mask(grid)
    monkeypatch.setenv("QD_PHYTO_NSPECIES", "3")
    monkeypatch.setenv("QD_PHYTO_INIT_FRAC", "2,1,1")
    monkeypatch.setenv("QD_PHYTO_CHL0", "0.08")
    ph = PhytoTracers(grid, mask)
    assert ph.S == 3 and ph.K_h == 5.0e3 and ph.adv_alpha == 0.7
    C = ph.C_phyto_s
    assert C.shape == (3, 19, 36) and np.all(C[:, mask == 1] == 0.0)
    assert np.allclose(C[0][mask == 0], 0.5 * 0.08) and np.allclose(C[2][mask == 0], 0.25 * 0.08)
    path = str(tmp_path / "plankton.nc")
    ph.C_phyto_s = C * np.linspace(0.5, 1.5, 36)[None, None, :]
    assert ph.save_distribution_nc(path, day_value=3.5)
    v, attrs = ncio.read_nc(path, ["C_phyto_s"])
    assert v["C_phyto_s"].dtype == np.float32 and float(attrs["day"]) == 3.5 and int(attrs["S"]) == 3
    ph2 = PhytoTracers(grid, mask)
    assert ph2.load_distribution_nc(path)
    assert np.allclose(ph2.C_phyto_s, ph.C_phyto_s, rtol=1e-6)
    # a plankton.nc written by the REFERENCE (phyto.py:737-802) also holds what its daily host code maintains -- the prognostic
    # nutrient pool N, Kd_490, the water albedo maps, the band axis: an autosave of this build must not drop them
    nb = 4
    ref_vars = {"lat": ("f4", ("lat",), np.asarray(grid.lat, np.float32)), "lon": ("f4", ("lon",), np.asarray(grid.lon, np.float32)),
                "C_phyto_s": ("f4", ("species", "lat", "lon"), np.zeros((3, 19, 36), np.float32)),
                "alpha_water_bands": ("f4", ("band", "lat", "lon"), np.full((nb, 19, 36), 0.06, np.float32)),
                "alpha_water_scalar": ("f4", ("lat", "lon"), np.full((19, 36), 0.07, np.float32)),
                "Kd_490": ("f4", ("lat", "lon"), np.full((19, 36), 0.11, np.float32)),
                "N": ("f4", ("lat", "lon"), np.linspace(0, 1, 19 * 36, dtype=np.float32).reshape(19, 36)),
                "bands_lambda_centers": ("f4", ("band",), np.array([400, 500, 600, 700], np.float32))}
    shared = str(tmp_path / "shared" / "plankton.nc")
    ncio.write_nc(shared, {"lat": 19, "lon": 36, "species": 3, "band": nb}, ref_vars,
                  {"title": "Qingdai Phytoplankton Distributions", "H_mld_m": 50.0, "S": 3, "NB": nb, "day": 1.0})
    assert ph.save_distribution_nc(shared, day_value=4.25)
    dims, got, attrs = ncio.read_nc_full(shared)
    assert dims == {"lat": 19, "lon": 36, "species": 3, "band": nb}
    for name in ("alpha_water_bands", "alpha_water_scalar", "Kd_490", "N", "bands_lambda_centers"):
        assert got[name][1] == ref_vars[name][1] and np.array_equal(got[name][2], ref_vars[name][2]), name
    assert np.allclose(got["C_phyto_s"][2], ph.C_phyto_s, rtol=1e-6)        # ... while the tracers are the new ones
    assert float(attrs["day"]) == 4.25 and float(attrs["H_mld_m"]) == 50.0 and int(attrs["NB"]) == nb
    monkeypatch.setenv("QD_PHYTO_NSPECIES", "2")
    assert not PhytoTracers(grid, mask).load_distribution_nc(path)          # species count mismatch: keep the current state
    assert PhytoTracers(grid, mask).save_distribution_nc(shared, day_value=5.0)      # a file of another species count is replaced
    assert ncio.read_nc_full(shared)[0] == {"lat": 19, "lon": 36, "species": 2}
