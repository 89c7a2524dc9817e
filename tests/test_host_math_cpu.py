"""qd_math.h (the device tanh for the cloud-source and P_cond terms) compiled for the host with g++ and compared with an 80-bit tanh:
the same text the HIP build includes, so what is measured here is what the kernels evaluate (fma() is an instruction on both)."""
import ctypes, os, subprocess, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "qd_math.h"
extern "C" void qd_tanh_array(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = qd_tanh(x[i]); }
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("qdmath")
    src = d / "m.cpp"
    src.write_text(SRC)
    so = d / "libqdmath.so"
    flags = ["-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "qingdai_amd", "csrc")]
    if "fma" in open("/proc/cpuinfo").read():
        flags.append("-mfma")
    subprocess.run(["g++", *flags, str(src), "-o", str(so)], check=True)
    L = ctypes.CDLL(str(so))
    L.qd_tanh_array.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    return L


def ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


def test_tanh_within_two_ulp_of_extended_precision(lib):
    r = np.random.default_rng(5)
    x = np.concatenate([r.random(400000) * 0.5, r.random(400000) * 40.0, 10.0 ** (-12 * r.random(200000)), 0.2 + 0.1 * r.random(200000),
                        np.array([0.0, 1e-300, 0.25, np.nextafter(0.25, 0), 19.1, 19.2, 700.0, np.inf])])
    x = np.concatenate([x, -x])
    y = np.empty_like(x)
    lib.qd_tanh_array(x.ctypes.data, y.ctypes.data, x.size)
    ref = np.tanh(x.astype(np.longdouble)).astype(np.float64)
    assert ulps(y, ref).max() <= 2
    assert np.all(np.abs(y) <= 1.0) and np.all(np.sign(y) == np.sign(x))
    z = np.array([np.nan]); w = np.empty(1)
    lib.qd_tanh_array(z.ctypes.data, w.ctypes.data, 1)
    assert np.isnan(w[0])
    # what the reference evaluates (np.tanh, float64): the two agree to the few ulp either may be off
    assert ulps(y, np.tanh(x)).max() <= 4
