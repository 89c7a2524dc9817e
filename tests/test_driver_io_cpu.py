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
    for name, fid in RESTART_VARS.items():
        assert rst[name].dtype == np.float32                       # the reference stores f4
        assert np.array_equal(rst[name], dev.get(fid).astype(np.float32)), name
