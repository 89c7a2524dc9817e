#!/usr/bin/env python3
"""
scripts/peer_ranks.py -- N latitude-band RANK PROCESSES on ONE GPU against the whole globe (developer rehearsal + GPU test).

What an 8-GPU run does between processes -- IPC handles of the mailboxes through the file rendezvous, hipIpcOpenMemHandle, halo
rows and global sums stored into the neighbours' mailboxes and polled there (qd_peer.hip) -- runs here with every rank on device
0: the peer mapping is then a same-device IPC mapping instead of an xGMI one, everything else (processes, rendezvous, kernels,
sequence counters, double buffering) is the real path.  RCCL cannot do this (it refuses two ranks on one device).

The parent never touches HIP: it starts the N rank processes and one whole-globe reference process as fresh children, waits,
assembles the bands' owned rows and compares.  Prints one JSON line; exit code 0 when the atmosphere fields agree bit for bit
(run without ocean) / every field within --tol (coupled run: only the band-wise order of the global sums differs).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def seed_state(nlat, nlon, seed):
    r = np.random.default_rng(seed)
    lat = np.linspace(-np.pi / 2, np.pi / 2, nlat)[:, None]
    lon = np.linspace(0, 2 * np.pi, nlon)[None, :]
    return {"U": 25.0 * np.cos(lat) * np.sin(2 * lon) + r.normal(0, 3.0, (nlat, nlon)),
            "V": 8.0 * np.sin(2 * lat) * np.cos(3 * lon) + r.normal(0, 2.0, (nlat, nlon)),
            "H": 8000.0 + 300 * np.sin(lat) ** 2 + 40.0 * np.cos(lat) * np.cos(2 * lon) + r.normal(0, 2.0, (nlat, nlon)),
            "TS": 262.0 + 38.0 * np.cos(lat) ** 2 + r.normal(0, 1.0, (nlat, nlon)),
            "Q": np.clip(0.006 + 0.004 * np.cos(lat) ** 2 + r.normal(0, 5e-4, (nlat, nlon)), 0, 0.5),
            "CLOUD": np.clip(0.3 + 0.3 * np.sin(3 * lon) * np.cos(lat) + r.normal(0, 0.05, (nlat, nlon)), 0, 1),
            "HICE": np.where(np.abs(lat) > 1.1, 0.4 + 0.3 * r.random((nlat, nlon)), 0.0)}


def names_of(ocean):
    return ["U", "V", "H", "TS", "Q", "CLOUD", "HICE"] + (["UO", "VO", "ETA", "SST"] if ocean else [])


def child(args):
    import ctypes
    import qingdai_amd as qa
    from qingdai_amd.bands import band_ranges, init_comm, preferred_halo, required_halo
    from qingdai_amd.device import Device
    from qingdai_amd.topography import create_land_sea_mask, generate_base_properties
    nlat, nlon = args.nlat, args.nlon
    grid = qa.SphericalGrid(nlat, nlon)
    mask = create_land_sea_mask(grid)
    alb, fric = generate_base_properties(mask)
    p = qa.QdParams(energy_w=1.0, ocean_cfl=args.ocean_cfl)
    p.has_csmap = 0
    forcing = qa.ThermalForcing(qa.SphericalGrid(nlat, nlon), qa.OrbitalSystem())
    stars = forcing.star_table([i * 300.0 for i in range(args.steps)])
    st = seed_state(nlat, nlon, args.seed)
    static = {"LAND_MASK": mask, "FRICTION": fric, "BASE_ALBEDO": alb}
    names = names_of(args.ocean)
    if args.role == "ref":
        dev = Device(grid, p)
        r0, n = 0, nlat
        transport = "none"
    else:
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        r0, n = band_ranges(nlat, world)[rank]
        halo = args.halo if args.halo > 0 else (preferred_halo(nlat, world) if args.preferred_halo else required_halo(nlat))
        dev = Device(grid, p, device=0, row0=r0, n_rows=n, halo=halo, rank=rank, world=world)
        transport = init_comm(dev, rank, world)
    for k, v in {**static, **st}.items():
        dev.upload_now(k, v)
    dev.sync()
    if args.role != "ref":
        dev._chk(dev.lib.qd_comm_barrier(dev.h), "qd_comm_barrier")
    t0 = time.perf_counter()
    dev.step_n(stars, 300.0, with_ocean=args.ocean, with_physics=args.ocean, pass_albedo=True)
    dev.sync()
    if args.role != "ref":
        dev._chk(dev.lib.qd_comm_barrier(dev.h), "qd_comm_barrier")
    el = time.perf_counter() - t0
    out = {k: dev.get(k)[r0:r0 + n].copy() for k in names}
    stats = {"transport": transport, "ms_per_step": el / args.steps * 1e3, "row0": r0, "rows": n, "n_sub": dev.last_ocean_nsub() if args.ocean else 0}
    if args.role != "ref":
        ne, na, nh, nr = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        dev.lib.qd_comm_stats(dev.h, ctypes.byref(ne)); dev.lib.qd_comm_allreduce_count(dev.h, ctypes.byref(na))
        dev.lib.qd_comm_peer_stats(dev.h, ctypes.byref(nh), ctypes.byref(nr))
        stats.update(halo_exchanges=ne.value, reductions=na.value, peer_halo=nh.value, peer_reductions=nr.value)
        # the last communication of this rank was a barrier every rank has passed: nobody stores into its mailbox any more
    np.savez(os.path.join(args.out, f"{args.role}{os.environ.get('RANK', '')}.npz"), meta=json.dumps(stats), **out)
    dev.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--nlat", type=int, default=61)
    ap.add_argument("--nlon", type=int, default=96)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--halo", type=int, default=0)
    ap.add_argument("--preferred-halo", action="store_true")
    ap.add_argument("--ocean", action="store_true")
    ap.add_argument("--ocean-cfl", type=float, default=0.05)
    ap.add_argument("--tol", type=float, default=1e-12)
    ap.add_argument("--timeout", type=float, default=600.0)
    ap.add_argument("--role", default="parent")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    if args.role != "parent":
        return child(args)

    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tmp = tempfile.mkdtemp(prefix="qd_peer_ranks_")
    base = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:]] + ["--out", tmp]
    procs = []
    for r in range(args.world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(args.world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(base + ["--role", "rank"], env=env))
    procs.append(subprocess.Popen(base + ["--role", "ref"], env=dict(os.environ)))
    deadline = time.time() + args.timeout
    rc, live = 0, list(procs)
    while live:
        for pr in list(live):
            r = pr.poll()
            if r is None:
                continue
            live.remove(pr)
            rc = max(rc, abs(r))
        if rc or time.time() > deadline:
            rc = rc or 124
            for pr in live:
                pr.terminate()
            for pr in live:
                try:
                    pr.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    pr.kill()
            break
        time.sleep(0.05)
    if rc:
        print(json.dumps({"ok": False, "error": f"a rank process failed or timed out (code {rc})"}))
        raise SystemExit(rc)
    ref = np.load(os.path.join(tmp, "ref.npz"))
    names = names_of(args.ocean)
    got = {k: np.zeros_like(ref[k]) for k in names}
    metas = []
    for r in range(args.world):
        d = np.load(os.path.join(tmp, f"rank{r}.npz"))
        m = json.loads(str(d["meta"]))
        metas.append(m)
        for k in names:
            got[k][m["row0"]:m["row0"] + m["rows"]] = d[k]
    errs = {k: float(np.max(np.abs(got[k] - ref[k]))) / max(float(np.max(np.abs(ref[k]))), 1e-300) for k in names}
    atm = ["U", "V", "H", "TS", "Q", "CLOUD", "HICE"]
    bitwise = all(np.array_equal(got[k], ref[k]) for k in atm)
    ok = (bitwise if not args.ocean else True) and all(e <= args.tol for e in errs.values())
    print(json.dumps({"ok": bool(ok), "world": args.world, "grid": [args.nlat, args.nlon], "steps": args.steps, "ocean": args.ocean,
                      "atmosphere_bitwise": bool(bitwise), "max_rel_err": errs, "ranks": metas,
                      "whole_globe": json.loads(str(ref["meta"]))}))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    raise SystemExit(0 if ok else 1)


if __name__ == "__main__":
    main()
