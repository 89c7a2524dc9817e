"""GPU (-m gpu): the device-side exchange over the peer mapping (qd_peer.hip, QD_PEER_EXCHANGE=1).

Halo rows and global sums of the latitude bands are stored into the neighbours' mailboxes by small kernels on the handle's own
stream and polled there: no collective launch, no host.  Checked three ways on ONE device:
  * in-process groups of 2 / 4 / 8 band handles (plain device pointers as the peer mapping) against the whole globe,
  * a one-rank communicator whose ring neighbours are the rank itself (the fused one-launch form of every operation) against the
    in-process host transport,
  * N rank PROCESSES with IPC-mapped mailboxes (scripts/peer_ranks.py: the rendezvous, hipIpcOpenMemHandle and the cross-process
    polling an 8-GPU run uses; RCCL cannot put two ranks on one device) against the whole globe.
The reference has no counterpart (single process, np.roll on whole arrays: pygcm/ocean.py:306-310, 369-377)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_bands import _run, _setup, _seed_state
from util import relerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_peer_bands_atmosphere_bit_identical(gpu, world, monkeypatch):
    monkeypatch.setenv("QD_PEER_EXCHANGE", "1")
    nlat = 61 if world < 8 else 181
    nlon = 96 if world < 8 else 128
    ref, _ = _run(1, nlat, nlon, 7, dict(energy_w=1.0), False, False)
    got, ex = _run(world, nlat, nlon, 7, dict(energy_w=1.0), False, False)
    print("halo exchanges per band:", ex)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), (k, relerr(got[k], ref[k]))


@pytest.mark.parametrize("overlap", ["2", "1", "0"])
@pytest.mark.parametrize("world", [2, 4])
def test_peer_bands_full_step_with_ocean_and_physics(gpu, world, overlap, monkeypatch):
    """The coupled step: eta sums, CFL maxima, precipitation sums, both median paths (histogram all-reduce + gathered candidate
    segments) and the halo exchanges of the sub-step loop all go through the mailboxes."""
    monkeypatch.setenv("QD_PEER_EXCHANGE", "1")
    monkeypatch.setenv("QD_PEER_OVERLAP", overlap)
    ref, _ = _run(1, 91, 144, 4, dict(energy_w=1.0, ocean_cfl=0.05), True, True)
    got, ex = _run(world, 91, 144, 4, dict(energy_w=1.0, ocean_cfl=0.05), True, True)
    print("halo exchanges per band:", ex)
    for k in ref:
        e = relerr(got[k], ref[k])
        assert e < 1e-12, (k, e)          # only the band-wise order of the global sums differs


@pytest.mark.parametrize("overlap", ["0", "1", "2"])
def test_peer_self_ring_equals_in_process_transport(gpu, overlap, monkeypatch):
    """One rank whose ring neighbours are the rank itself: every operation in its fused one-launch form (push + poll + unpack,
    deposit + poll + reduce in one kernel), mailboxes mapped through qd_peer_export / qd_peer_connect.  Must move exactly the
    bytes the in-process host transport moves (cf. test_rccl_transport_equals_in_process_transport)."""
    from qingdai_amd.bands import init_peer
    from qingdai_amd.device import Device
    monkeypatch.setenv("MASTER_PORT", "29741")
    monkeypatch.setenv("QD_NO_HOST_RING", "1")
    # overlap >= 1: the ocean momentum kernel of a sub-step that exchanges runs its interior rows between push and unpack, the two
    # boundary strips after the unpack; 2 (the default of a multi-rank run; a one-rank ring has to ask for it): the push is the
    # first workgroups of the interior launch itself (k_ocn_stream_push), so that the rows are in flight WHILE it computes
    monkeypatch.setenv("QD_PEER_OVERLAP", overlap)
    nlat, nlon, nsteps = 91, 144, 5
    qa, grid, mask, alb, fric, p = _setup(nlat, nlon, dict(energy_w=1.0))
    forcing = qa.ThermalForcing(qa.SphericalGrid(nlat, nlon), qa.OrbitalSystem())
    stars = forcing.star_table([i * 300.0 for i in range(nsteps)])
    st = _seed_state(nlat, nlon, 21)
    static = {"LAND_MASK": mask, "FRICTION": fric, "BASE_ALBEDO": alb}
    names = ["U", "V", "H", "TS", "Q", "CLOUD", "UO", "VO", "ETA", "SST", "ALBEDO", "PRECIP"]
    out, counts = {}, {}
    for transport in ("local", "peer"):
        monkeypatch.delenv("QD_PEER_EXCHANGE", raising=False)
        dev = Device(qa.SphericalGrid(nlat, nlon), p, row0=25, n_rows=41, halo=12, rank=0, world=1)
        if transport == "local":
            arr = (ctypes.c_void_p * 1)(dev.h)
            assert dev.lib.qd_comm_init_local(arr, 1) == 0
        else:
            assert init_peer(dev, 0, 1, tag="selfpeer")          # includes qd_peer_selftest (self-describing rows, known sums)
            wrong = ctypes.c_longlong(-1)
            assert dev.lib.qd_peer_selftest(dev.h, 9, ctypes.byref(wrong)) == 0 and wrong.value == 0
        for k, v in {**static, **st}.items():
            dev.upload_now(k, v)
        dev.step_n(stars, 300.0, with_ocean=True, with_physics=True, pass_albedo=True)
        assert dev.lib.qd_comm_barrier(dev.h) == 0
        v = (ctypes.c_double * 2)(1.5, -2.0)
        assert dev.lib.qd_comm_allreduce_max(dev.h, v, 2) == 0 and list(v) == [1.5, -2.0]
        out[transport] = {k: dev.get(k)[25:66].copy() for k in names}
        ne, nh, nr = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        dev.lib.qd_comm_stats(dev.h, ctypes.byref(ne)); dev.lib.qd_comm_peer_stats(dev.h, ctypes.byref(nh), ctypes.byref(nr))
        counts[transport] = (ne.value, nh.value, nr.value)
        carried = dev.lib.qd_comm_peer_carried(dev.h)
        assert (carried > 0) == (transport == "peer" and overlap == "2"), carried
        dev.close()
    print("halo exchanges, of them through mailboxes, reductions through mailboxes:", counts)
    assert counts["local"][0] == counts["peer"][0] and counts["local"][1:] == (0, 0)
    assert counts["peer"][1] >= counts["peer"][0] > 10 and counts["peer"][2] > 20
    for k in names:
        assert np.array_equal(out["local"][k], out["peer"][k], equal_nan=True), k
        assert np.isfinite(out["peer"][k]).all(), k


@pytest.mark.parametrize("case", ["atmosphere_x2", "coupled_x3"])
def test_peer_rank_processes_match_the_whole_globe(gpu, case):
    """N rank processes on device 0, mailboxes mapped through hipIpc*: the multi-process path of bench.py --gpus N."""
    args = {"atmosphere_x2": ["--world", "2", "--nlat", "61", "--nlon", "96", "--steps", "6"],
            "coupled_x3": ["--world", "3", "--nlat", "91", "--nlon", "144", "--steps", "4", "--ocean"]}[case]
    env = dict(os.environ, QD_PEER_EXCHANGE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "peer_ranks.py")] + args, env=env, capture_output=True, text=True,
                       timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    res = json.loads(line[-1])
    print(res)
    assert res["ok"] and all(m["transport"] == "peer" for m in res["ranks"])
    if case == "atmosphere_x2":
        assert res["atmosphere_bitwise"]
