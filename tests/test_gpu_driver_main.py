"""GPU (-m gpu): the drop-in driver entry point (qingdai_amd.driver.main <-> scripts/run_simulation.py main()) end to end on
a small grid: periodic autosave on the reference's schedule, the autosave-load fallback of a rerun (QD_AUTOSAVE_LOAD=1, the
default: run_simulation.py:1513-1560), the data/ocean.nc override (QD_LOAD_OCEAN=1) and the restart file layout."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_driver_main_autosave_cycle(gpu, tmp_path, monkeypatch, capsys):
    from qingdai_amd import driver, ncio
    for k in list(os.environ):
        if k.startswith("QD_"):
            monkeypatch.delenv(k)
    data = tmp_path / "data"
    env = {"QD_N_LAT": "37", "QD_N_LON": "72", "QD_SIM_DAYS": "0.55", "QD_ECO_ENABLE": "0", "QD_DATA_DIR": str(data),
           "QD_DYN_DIAG_PRINT": "0", "QD_USE_OCEAN": "1"}
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.chdir(tmp_path)
    assert driver.main() == 0
    out1 = capsys.readouterr().out
    # 0.55 planet-days = 132 steps; the default interval is 6 planetary hours = 60 steps: periodic saves after steps 60 and 120
    assert out1.count("[Autosave] (periodic)") == 2, out1
    assert "[Autosave] loaded" not in out1
    v, _ = ncio.read_nc(str(data / "atmosphere.nc"), ["u", "land_mask", "uo", "t_seconds"])
    assert v["land_mask"].dtype == np.float32 and v["u"].dtype == np.float32 and "uo" in v
    t_end = float(v["t_seconds"])
    assert abs(t_end - 132 * 300.0) < 1e-6 * t_end                       # the exit-time autosave holds the final epoch
    assert os.path.exists(data / "ocean.nc") and os.path.exists(data / "topography.nc")
    # a rerun without QD_RESTART_IN starts from the checkpoint, and data/ocean.nc overrides the ocean fields
    monkeypatch.setenv("QD_SIM_DAYS", "0.05")
    assert driver.main() == 0
    out2 = capsys.readouterr().out
    assert "[Autosave] loaded" in out2 and "Ocean state overridden" in out2, out2
    v2, _ = ncio.read_nc(str(data / "atmosphere.nc"), ["t_seconds"])
    assert float(v2["t_seconds"]) > t_end                                # it continued from the loaded epoch
    # QD_AUTOSAVE_LOAD=0: cold start
    monkeypatch.setenv("QD_AUTOSAVE_LOAD", "0")
    assert driver.main() == 0
    assert "[Autosave] loaded" not in capsys.readouterr().out
