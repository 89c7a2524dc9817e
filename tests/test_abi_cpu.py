"""CPU: the C-ABI library loads and exports every symbol include/qingdai_hip.h declares; the ctypes
parameter struct has the header's layout; product and oracle defaults agree; no compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    return open(os.path.join(ROOT, "include", "qingdai_hip.h")).read()


def test_library_exports_every_declared_symbol():
    from qingdai_amd import _lib
    lib = _lib.load()
    h = re.sub(r"/\*.*?\*/", "", _header(), flags=re.S)
    declared = set(re.findall(r"\b(qd_[a-z0-9_]+)\s*\(", h)) - {"qd_star_cb"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.qd_abi_version() == 1


def test_param_struct_layout_matches_header():
    from qingdai_amd.params import qd_params
    h = _header()
    body = h[h.index("typedef struct qd_params {"):h.index("} qd_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for line in body.split("\n")[1:]:
        m = re.match(r"\s*(double|int32_t)\s+(.*);", line)
        if m:
            names += [(m.group(1), n.strip()) for n in m.group(2).split(",")]
    py = [("double" if t is ctypes.c_double else "int32_t", n) for n, t in qd_params._fields_]
    assert names == py


def test_field_ids_match_header():
    from qingdai_amd import _lib
    h = _header()
    body = h[h.index("enum qd_field {"):h.index("QD_F_COUNT_F64")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    ids = re.findall(r"QD_F_([A-Z0-9_]+)", body)
    assert ids == _lib.FIELDS


def test_product_defaults_equal_oracle_defaults():
    import qd_oracle as qo
    from qingdai_amd import QdParams
    o = qo.defaults()
    p = QdParams().as_oracle_kwargs()
    for k, v in vars(o).items():
        assert k in p, k
        pv = p[k]
        if isinstance(v, float) and v != v:
            assert pv != pv, k
        else:
            assert pv == v, (k, pv, v)


def test_env_parsing(monkeypatch):
    from qingdai_amd import QdParams
    monkeypatch.setenv("QD_MOM_SCHEME", "primitive")
    monkeypatch.setenv("QD_FILTER_TYPE", "hyper4")
    monkeypatch.setenv("QD_ENERGY_W", "1")
    monkeypatch.setenv("QD_OCEAN_OUTLIER", "clamp")
    monkeypatch.setenv("QD_K4_U", "1e14")
    p = QdParams.from_env()
    assert (p.mom_scheme, p.filter_type, p.energy_w, p.ocean_outlier, p.k4_u) == (1, 1, 1.0, 1, 1e14)
    assert p.k4_v != p.k4_v and p.pcond_ref != p.pcond_ref      # unset -> NaN


def test_no_gpu_fails_loudly():
    """Without a GPU the product must raise, never fall back."""
    from conftest import _gpu_visible
    if _gpu_visible():
        pytest.skip("GPU present")
    import qingdai_amd as qa
    from qingdai_amd._lib import QdError
    g = qa.SphericalGrid(19, 36)
    with pytest.raises(QdError):
        qa.SpectralModel(g, np.zeros((19, 36)), land_mask=np.zeros((19, 36), dtype=np.uint8))


def test_topography_seed42_fingerprint():
    import hashlib
    import qd_oracle as qo
    from qingdai_amd.topography import create_land_sea_mask
    m = create_land_sea_mask(qo.Grid(181, 360))
    assert int(m.sum()) == 16242
    assert hashlib.sha1(m.tobytes()).hexdigest().startswith("17de315c9bb5")


def test_eco_param_struct_layout_matches_header():
    from qingdai_amd._lib import qd_eco_params
    h = _header()
    body = h[h.index("typedef struct qd_eco_params {"):h.index("} qd_eco_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for line in body.split("\n")[1:]:
        m = re.match(r"\s*(double|int32_t)\s+(.*);", line)
        if m:
            names += [(m.group(1), n.strip()) for n in m.group(2).split(",")]
    py = [("double" if t is ctypes.c_double else "int32_t", n) for n, t in qd_eco_params._fields_]
    assert names == py and ctypes.sizeof(qd_eco_params) == 6 * 8 + 6 * 4
