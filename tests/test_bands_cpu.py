"""CPU (world_size 2, gloo): the latitude-band host logic -- band ranges, ring neighbours (period
n_lat at the poles), halo sizing -- exercised with the ORACLE's operators on NumPy slabs.  Each rank
owns a band + halo, refreshes the halo with a ring send/recv exactly like qd_exchange does, applies
the global-operator on the rows it can see and must reproduce the whole-globe result on its band."""
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, nlat, nlon, H, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import qd_oracle as qo
    from qd_oracle import atmos as oat
    from qingdai_amd.bands import band_ranges
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = qo.Grid(nlat, nlon)
    r = np.random.default_rng(11)
    F = 8000.0 + r.normal(0, 5.0, (nlat, nlon))
    u = r.normal(0, 30.0, (nlat, nlon))
    v = r.normal(0, 20.0, (nlat, nlon))
    r0, n = band_ranges(nlat, world)[rank]
    up, dn = (rank + 1) % world, (rank - 1) % world
    # local slab: owned rows + H halo rows each side, halos initially garbage
    slab = np.full((n + 2 * H, nlon), np.nan)
    slab[H:H + n] = F[r0:r0 + n]
    # ring exchange (same four messages as qd_exchange): top rows -> up's south halo, bottom rows -> dn's north halo
    top = torch.from_numpy(slab[n:n + H].copy())
    bot = torch.from_numpy(slab[H:2 * H].copy())
    south = torch.empty_like(top)
    north = torch.empty_like(bot)
    reqs = [dist.isend(top, up), dist.irecv(south, dn), dist.isend(bot, dn), dist.irecv(north, up)]
    for rq in reqs:
        rq.wait()
    slab[:H] = south.numpy()
    slab[n + H:] = north.numpy()
    # every slab row must now equal the global row (r0 - H + l) mod nlat
    rows = (r0 - H + np.arange(n + 2 * H)) % nlat
    ok_halo = bool(np.array_equal(slab, F[rows]))
    # embed into a NaN globe and apply the oracle's del^4 (reach 4) and gather (reach R): owned rows must match
    globe = np.full((nlat, nlon), np.nan)
    globe[rows] = slab
    cos02 = np.maximum(np.cos(np.deg2rad(g.lat_mesh)), 0.2)
    k4 = 1e14
    ref = oat.hyperdiffuse(F, k4, 300.0, 1, g.dlat_rad, g.dlon_rad, cos02, 6.371e6)
    with np.errstate(all="ignore"):
        Fn = np.where(np.isnan(globe), 1e300, globe)       # poison instead of NaN (nan_to_num would hide NaN)
        got = oat.hyperdiffuse(Fn, k4, 300.0, 1, g.dlat_rad, g.dlon_rad, cos02, 6.371e6)
    ok_h4 = bool(np.array_equal(got[r0:r0 + n], ref[r0:r0 + n]))
    cos6 = np.maximum(1e-6, np.cos(np.deg2rad(g.lat_mesh)))
    refa = oat.advect_semilag(F, u, v, 300.0, 6.371e6, g.dlat_rad, g.dlon_rad, cos6)
    gota = oat.advect_semilag(Fn, u, v, 300.0, 6.371e6, g.dlat_rad, g.dlon_rad, cos6)
    ok_adv = bool(np.array_equal(gota[r0:r0 + n], refa[r0:r0 + n]))
    q.put((rank, ok_halo, ok_h4, ok_adv))
    dist.barrier()
    dist.destroy_process_group()


def test_band_ranges_and_halo_sizing():
    from qingdai_amd.bands import band_ranges, required_halo, adv_reach
    for n, w in ((721, 8), (1441, 8), (37, 3), (181, 2)):
        rr = band_ranges(n, w)
        assert rr[0][0] == 0 and sum(k for _, k in rr) == n
        assert all(rr[i][0] + rr[i][1] == rr[i + 1][0] for i in range(w - 1))
        assert max(k for _, k in rr) - min(k for _, k in rr) <= 1
    assert adv_reach(1441, 300.0) == 7 and required_halo(1441) == 22
    assert required_halo(721) >= 2 * adv_reach(721, 300.0) + 8


def test_preferred_halo_keeps_bands_valid(monkeypatch):
    """The halo bench.py gives its ranks: never below the one-exchange-per-atmosphere-step minimum, never taller than a band,
    never so tall that band + 2 halos exceed the globe (qd_create refuses that: halo rows would alias owned rows)."""
    from qingdai_amd.bands import band_ranges, preferred_halo, required_halo
    monkeypatch.delenv("QD_BAND_HALO", raising=False)
    for nlat, world in ((721, 2), (721, 4), (721, 8), (1441, 8), (181, 2), (91, 4), (61, 2), (61, 3)):
        h = preferred_halo(nlat, world)
        assert h >= required_halo(nlat)
        for _, n in band_ranges(nlat, world):
            assert n >= h and n + 2 * h <= nlat, (nlat, world, h, n)
    assert preferred_halo(721, 8) == 32 and preferred_halo(61, 2) == required_halo(61) + 3
    monkeypatch.setenv("QD_BAND_HALO", "40")
    assert preferred_halo(721, 8) == 40
    monkeypatch.setenv("QD_BAND_HALO", "4")
    assert preferred_halo(721, 8) == required_halo(721)


@pytest.mark.timeout(240)
def test_ring_halo_exchange_two_ranks_gloo(tmp_path):
    # Each rank is its own interpreter (torch + gloo live only there): the pytest process itself never
    # imports torch, whose bundled HIP runtime must not share a process with libqingdai_hip.so's.
    import json
    import subprocess
    world, nlat, nlon, H = 2, 37, 48, 6
    port = 29650 + (os.getpid() % 200)
    code = (
        "import sys, json, queue\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "import test_bands_cpu as t\n"
        "class Q:\n"
        "    def put(self, x): print('RESULT ' + json.dumps(x), flush=True)\n"
        f"t._worker(int(sys.argv[1]), {world}, {port}, {nlat}, {nlon}, {H}, Q())\n"
    )
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    res = []
    for p in procs:
        out, err = p.communicate(timeout=200)
        assert p.returncode == 0, err[-2000:]
        line = [ln for ln in out.splitlines() if ln.startswith("RESULT ")][-1]
        res.append(json.loads(line[len("RESULT "):]))
    for rank, ok_halo, ok_h4, ok_adv in res:
        assert ok_halo, f"rank {rank}: ring halo rows wrong"
        assert ok_h4, f"rank {rank}: del^4 on the band differs from the globe"
        assert ok_adv, f"rank {rank}: gather on the band differs from the globe"


def _ring_worker(rank, world, name, iters, q):
    import ctypes
    sys.path.insert(0, ROOT)
    from qingdai_amd import _lib
    lib = _lib.load()
    ring = ctypes.c_void_p()
    assert lib.qd_hostring_open(name.encode(), rank, world, ctypes.byref(ring)) == 0
    bad = 0
    for it in range(iters):
        # every rank can rebuild everybody's contribution: the expected reduction is computed independently, in rank order
        contrib = [np.random.default_rng(1000 * it + r).normal(0, 10.0 ** (it % 7), 3) for r in range(world)]
        v = (ctypes.c_double * 3)(*contrib[rank])
        op = it % 2
        assert lib.qd_hostring_allreduce(ring, v, 3, op) == 0
        want = contrib[0].copy()
        for r in range(1, world):
            want = np.maximum(want, contrib[r]) if op else want + contrib[r]
        bad += int(not np.array_equal(np.array(list(v)), want))
    lib.qd_hostring_close(ring)
    q.put((rank, bad))


@pytest.mark.timeout(240)
def test_host_ring_allreduce_across_processes():
    """The shared-memory scalar all-reduce of the band transport (qd_hostring_*), 4 processes x 1500 calls, sums and maxima
    alternating: every rank gets the rank-ordered reduction bit for bit, and nobody overtakes (two buffers by sequence parity)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    world, iters = 4, 1500
    name = f"/qd_test_ring_{os.getpid()}"
    q = ctx.Queue()
    ps = [ctx.Process(target=_ring_worker, args=(r, world, name, iters, q)) for r in range(world)]
    for p_ in ps:
        p_.start()
    res = sorted(q.get(timeout=200) for _ in ps)
    for p_ in ps:
        p_.join(30)
    assert res == [(r, 0) for r in range(world)]
    assert not os.path.exists("/dev/shm" + name)                    # rank 0 unlinked the segment


def _rdzv_worker(rank, world, key, root, q):
    sys.path.insert(0, ROOT)
    from qingdai_amd.bands import exchange_unique_id
    uid = exchange_unique_id(rank, world, lambda: b"FRESH-ID-" + bytes(119), key, timeout_s=60.0, root=root)
    q.put((rank, bytes(uid[:9])))


@pytest.mark.timeout(120)
def test_rendezvous_ignores_what_an_earlier_run_left_behind(tmp_path):
    """the RCCL id rendezvous used to accept any existing file under a key that a later launch can reuse.  Now a rank
    accepts an id file only if it carries the nonce that rank posted for THIS launch: stale id / request files are ignored (and
    removed by rank 0), whatever order the ranks start in."""
    import multiprocessing as mp
    from qingdai_amd.bands import finish_rendezvous
    world, key, root = 3, "29500_none_0_4242_3_id", str(tmp_path)
    path = os.path.join(root, f"qd_rdzv_{key}")
    with open(path, "wb") as fh:                                       # a complete id file of an earlier run
        fh.write(b"QDRZ" + bytes(16 * world) + b"STALE-ID-" + bytes(119))
    for k in (1, 2):
        with open(f"{path}.req.{k}", "wb") as fh:                      # and its request files
            fh.write(b"s" * 16)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rdzv_worker, args=(r, world, key, root, q)) for r in (2, 1, 0)]   # rank 0 starts last
    for p_ in ps:
        p_.start()
        time.sleep(0.3)
    res = sorted(q.get(timeout=90) for _ in ps)
    for p_ in ps:
        p_.join(30)
    assert res == [(r, b"FRESH-ID-") for r in range(world)]
    finish_rendezvous(0, key, root=root)
    assert not [f for f in os.listdir(root) if f.startswith("qd_rdzv_")]


@pytest.mark.timeout(240)
def test_host_ring_survives_a_stale_segment():
    """A crashed run leaves its shared-memory segment behind with non-zero sequence counters; rank 0 of the next run replaces it
    (unlink + exclusive create) and the others only map the new one: the reductions are right from the first call."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    world, iters = 3, 200
    name = f"/qd_test_stale_{os.getpid()}"
    with open("/dev/shm" + name, "wb") as fh:
        fh.write(b"\xff" * 20000)                                      # garbage counters and values, larger than the real segment
    q = ctx.Queue()
    ps = [ctx.Process(target=_ring_worker, args=(r, world, name, iters, q)) for r in range(world)]
    ps[0].start()                                                      # rank 0 first: the others must find ITS segment
    time.sleep(1.0)
    for p_ in ps[1:]:
        p_.start()
    res = sorted(q.get(timeout=200) for _ in ps)
    for p_ in ps:
        p_.join(30)
    assert res == [(r, 0) for r in range(world)]
    assert not os.path.exists("/dev/shm" + name)


# ------------------------------------------------------------------------------------------------------------------------
# The library's OWN planner (qd_plan / qd_mark / qd_segments / the bookkeeping of qd_exchange, qd_band.hip) driven without a
# GPU: qd_plansim_* handles own no memory and log the halo exchanges they decide on; each gloo rank performs the logged
# exchanges on NumPy slabs and runs a chain of ring-periodic row stencils on exactly the rows the planner says are valid.
def _plan_worker(rank, world, port, nlat, nlon, H, q):
    sys.path.insert(0, ROOT)
    import ctypes
    import torch
    import torch.distributed as dist
    from qingdai_amd import _lib
    from qingdai_amd.bands import band_ranges
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _lib.load()
    r0, n = band_ranges(nlat, world)[rank]
    desc = _lib.qd_grid_desc(nlat, nlon, r0, n, H, 0, rank, world)
    h = ctypes.c_void_p()
    assert lib.qd_plansim_create(ctypes.byref(desc), ctypes.byref(h)) == 0
    I = ctypes.c_int32
    rng = np.random.default_rng(5)
    NF = 3
    truth = [rng.integers(-50, 50, (nlat, nlon)).astype(float) for _ in range(NF)]       # small integers: every sum is exact
    rows_of_slab = (r0 - H + np.arange(n + 2 * H)) % nlat
    slabs = []
    for f in range(NF):
        s = np.full((n + 2 * H, nlon), np.nan)
        s[H:H + n] = truth[f][r0:r0 + n]
        slabs.append(s)
    n_exch, bad = 0, []

    def do_exchanges():
        nonlocal n_exch
        fl, geo = (I * 16)(), (I * 4)()
        while True:
            k = lib.qd_plansim_pop_exchange(h, fl, 16, geo)
            assert k >= 0
            if k == 0:
                return
            n_exch += 1
            Hh, nown, up, dn = geo[0], geo[1], geo[2], geo[3]
            assert (Hh, nown, up, dn) == (H, n, (rank + 1) % world, (rank - 1) % world)
            for f in list(fl)[:k]:
                s = slabs[f]
                top = torch.from_numpy(s[nown:nown + Hh].copy()); bot = torch.from_numpy(s[Hh:2 * Hh].copy())
                south = torch.empty_like(top); north = torch.empty_like(bot)
                # tag 2f: northward traffic (my top rows -> up's south halo), tag 2f+1: southward -- world 2 has up == dn
                reqs = [dist.isend(top, up, tag=2 * f), dist.irecv(south, dn, tag=2 * f),
                        dist.isend(bot, dn, tag=2 * f + 1), dist.irecv(north, up, tag=2 * f + 1)]
                for rq in reqs:
                    rq.wait()
                s[:Hh] = south.numpy(); s[Hh + nown:] = north.numpy()

    # (inputs, radii, output): a chain shaped like a step: reach-4/5 stencils (del^4, momentum), reach-1/2 ones, a wide gather
    ops = [([0], [4], 1), ([1], [4], 2), ([2, 0], [1, 5], 1), ([1], [2], 0), ([0], [4], 2), ([2], [4], 1), ([1, 2], [H - 2, 1], 0),
           ([0], [1], 1), ([1], [5], 2), ([2], [4], 0), ([0, 1], [4, 4], 2), ([2], [H, ], 1)]
    for it, (ins, radii, out) in enumerate(ops):
        m = lib.qd_plansim_plan(h, (I * len(ins))(*ins), (I * len(ins))(*radii), len(ins), -1)
        assert m >= 0, (it, m)
        do_exchanges()
        for f, r in zip(ins, radii):
            assert lib.qd_plansim_margin(h, f) >= m + r                     # what the planner promised
        # launch segments for margin m: together exactly the rows own +- m, none wrapping inside itself
        seg = (I * 6)()
        ns = lib.qd_plansim_segments(h, m, seg)
        cover = np.concatenate([np.arange(seg[2 * k], seg[2 * k] + seg[2 * k + 1]) for k in range(ns)])
        want_rows = (r0 - m + np.arange(n + 2 * m)) % nlat
        if not (np.array_equal(cover, want_rows) and all(0 <= seg[2 * k] and seg[2 * k] + seg[2 * k + 1] <= nlat for k in range(ns))):
            bad.append(("segments", it))
        # out = sum over inputs of (x[i-r] + 2 x[i] + x[i+r]), rows periodic with period nlat (np.roll(axis=0) semantics)
        lo, hi = H - m, H + n + m
        new = np.zeros((hi - lo, nlon))
        tnew = np.zeros((nlat, nlon))
        for f, r in zip(ins, radii):
            s = slabs[f]
            new += s[lo - r:hi - r] + 2.0 * s[lo:hi] + s[lo + r:hi + r]
            tnew += np.roll(truth[f], r, 0) + 2.0 * truth[f] + np.roll(truth[f], -r, 0)
        tnew = np.mod(tnew + 50.0, 101.0) - 50.0                            # keep the integers small
        new = np.mod(new + 50.0, 101.0) - 50.0
        if not np.array_equal(new, tnew[rows_of_slab[lo:hi]]):
            bad.append(("values", it, m))
        truth[out] = tnew
        slabs[out][:] = np.nan
        slabs[out][lo:hi] = new
        assert lib.qd_plansim_mark(h, (I * 1)(out), 1, m) == 0
    lib.qd_plansim_destroy(h)
    q.put((rank, bad, n_exch))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_library_planner_drives_gloo_halo_exchanges(world):
    """world_size 2 (and 3) over gloo: the planner decides, the test moves the rows the planner's exchange log names, and
    every stencil result on own +- margin equals the whole-globe one -- margins, segment lists (incl. the pole wrap of the
    first / last band), ring neighbours and row offsets of qd_exchange are all exercised by real data movement."""
    import json
    import subprocess
    nlat, nlon, H = 61, 8, 9
    port = 29850 + (os.getpid() % 100) + 7 * world
    code = (
        "import sys, json\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "import test_bands_cpu as t\n"
        "class Q:\n"
        "    def put(self, x): print('RESULT ' + json.dumps(x), flush=True)\n"
        f"t._plan_worker(int(sys.argv[1]), {world}, {port}, {nlat}, {nlon}, {H}, Q())\n"
    )
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    res = []
    for p in procs:
        out, err = p.communicate(timeout=250)
        assert p.returncode == 0, err[-3000:]
        res.append(json.loads([ln for ln in out.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):]))
    for rank, bad, n_exch in res:
        assert bad == [], (rank, bad)
        assert 1 <= n_exch < 12, (rank, n_exch)                              # deep halos: fewer exchanges than launches
    assert len({r[2] for r in res}) == 1                                     # every rank decided on the same exchanges


def test_bench_spawns_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment starts N fresh rank processes itself (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT like torch.distributed.run sets them) before anything touches HIP; with a launcher's
    environment it is a rank.  --spawn-check makes every rank print that environment and stop before the library is loaded."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--spawn-check"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    recs = sorted((json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")), key=lambda r: int(r["RANK"]))
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2"]
    assert {r["WORLD_SIZE"] for r in recs} == {"3"} and {r["MASTER_ADDR"] for r in recs} == {"127.0.0.1"}
    assert len({r["MASTER_PORT"] for r in recs}) == 1
    # under a launcher: one process, its own rank
    env2 = dict(env, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--spawn-check"], env=env2, capture_output=True,
                          text=True, timeout=120)
    assert out2.returncode == 0 and json.loads(out2.stdout.strip())["RANK"] == "1"


def test_interior_and_boundary_segments_partition_a_launch():
    """Exchange overlapped with interior compute (qd_plan_begin / qd_plan_end, qd_ocean.hip): a consumer launch of margin m around an
    exchange is split into its INTERIOR rows [own0 - m_int, own1 + m_int) -- what the old margins allow, launched while the halos
    travel -- and two BOUNDARY strips.  Whatever the band (polar bands wrap around the ring), the three pieces must cover exactly the
    rows of the unsplit launch, each once, and no piece may leave the slab."""
    import ctypes
    from qingdai_amd import _lib
    from qingdai_amd.bands import band_ranges
    lib = _lib.load()
    I = ctypes.c_int32
    nlat, nlon, H = 181, 96, 16
    for world in (2, 3, 8):
        for rank, (r0, n) in enumerate(band_ranges(nlat, world)):
            desc = _lib.qd_grid_desc(nlat, nlon, r0, n, H, 0, rank, world)
            h = ctypes.c_void_p()
            assert lib.qd_plansim_create(ctypes.byref(desc), ctypes.byref(h)) == 0

            def rows_of(vr0, cnt):
                seg = (I * 12)()
                ns = lib.qd_plansim_segments_rows(h, vr0, cnt, seg)
                assert 0 <= ns <= 6
                out = []
                for k in range(ns):
                    g0, ln = seg[2 * k], seg[2 * k + 1]
                    assert 0 <= g0 and g0 + ln <= nlat and ln > 0
                    out += list(range(g0, g0 + ln))
                return out
            for m in (0, 3, 9, H - 5):
                seg = (I * 6)()
                ns = lib.qd_plansim_segments(h, m, seg)
                whole = sorted(sum((list(range(seg[2 * k], seg[2 * k] + seg[2 * k + 1])) for k in range(ns)), []))
                assert whole == sorted((r0 - m + k) % nlat for k in range(n + 2 * m))
                for m_int in (-12, -1, 0, 2, m):
                    if m_int > m or n + 2 * m_int < 1:
                        continue
                    interior = rows_of(r0 - m_int, n + 2 * m_int)
                    south = rows_of(r0 - m, m - m_int)
                    north = rows_of(r0 + n + m_int, m - m_int)
                    pieces = interior + south + north
                    assert len(pieces) == len(set(pieces)) == len(whole), (world, rank, m, m_int)
                    assert sorted(pieces) == whole
                    # every row is on this band's slab (owned rows + H halo rows either side, ring-periodic)
                    slab = set((r0 - H + k) % nlat for k in range(n + 2 * H))
                    assert set(pieces) <= slab
            assert lib.qd_plansim_destroy(h) == 0


def test_spawner_ends_the_other_ranks_when_one_dies():
    """bench.py --gpus N without a launcher: when a rank process fails, the ranks that would now block in the rendezvous or in a
    collective for ever are terminated, and the spawner exits with the failing rank's code -- promptly."""
    import subprocess
    import time
    root = ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(QD_BENCH_TEST_FAIL_RANK="1", QD_BENCH_TEST_HANG_RANK="2")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--spawn-check"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 3, (out.returncode, out.stderr[-500:])
    assert time.time() - t0 < 60.0                           # not the hanging rank's 300 s


def test_device_code_stamp_reads_the_fatbin_section():
    """qingdai_amd/_codehash.py: the profile stamp is the hash of the .hip_fatbin section (device code only) of the built library /
    objects; a hand-made ELF shows the section walk picks the right bytes and that bytes OUTSIDE the section do not matter."""
    import hashlib
    import struct
    from qingdai_amd import _codehash as ch

    def elf(payload, junk):
        names = b"\0.shstrtab\0.hip_fatbin\0"
        hdr = 64
        off_junk, off_pay = hdr, hdr + len(junk)
        off_str = off_pay + len(payload)
        shoff = (off_str + len(names) + 7) & ~7
        e = bytearray(b"\x7fELF" + bytes([2, 1, 1]) + bytes(9))
        e += struct.pack("<HHIQQQIHHHHHH", 3, 62, 1, 0, 0, shoff, 0, 64, 0, 0, 64, 3, 1)
        body = bytes(e) + junk + payload + names
        body += bytes(shoff - len(body))
        sh0 = bytes(64)
        sh1 = struct.pack("<IIQQQQIIQQ", 1, 3, 0, 0, off_str, len(names), 0, 0, 1, 0)            # .shstrtab
        sh2 = struct.pack("<IIQQQQIIQQ", 11, 1, 2, 0, off_pay, len(payload), 0, 0, 4096, 0)      # .hip_fatbin
        return body + sh0 + sh1 + sh2
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "a.o"), os.path.join(d, "b.o")
        open(a, "wb").write(elf(b"DEVICE-CODE" * 50, b"host text 1"))
        open(b, "wb").write(elf(b"DEVICE-CODE" * 50, b"host text 2, edited"))
        assert ch.elf_section(a, ".hip_fatbin") == b"DEVICE-CODE" * 50
        assert ch.fatbin_sha(a) == ch.fatbin_sha(b) == hashlib.sha256(b"DEVICE-CODE" * 50).hexdigest()[:16]
        open(b, "wb").write(elf(b"DEVICE-CODE" * 49 + b"DEVICE-C0DE", b"host text 1"))
        assert ch.fatbin_sha(a) != ch.fatbin_sha(b)
    lib = os.path.join(ROOT, "qingdai_amd", "libqingdai_hip.so")
    if os.path.exists(lib):
        st = ch.device_code_stamp()
        assert "libqingdai_hip.so" in st and not ch.stale_against(st)
        assert ch.stale_against({k: "0" * 16 for k in st})
