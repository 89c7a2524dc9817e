"""
qingdai_amd/bands.py -- latitude-band decomposition on the host side.

`band_ranges` splits the rows; `required_halo` sizes the ring halo from the stencil reaches of one
step; `BandGroup` drives several band handles that live in ONE process on ONE device (one host
thread each, device-to-device halo copies) -- the way the band logic is tested on a single GPU;
`init_rccl` wires a one-process-per-GPU handle to its RCCL communicator (file rendezvous on a
single node, no network service and no PyTorch).
"""
from __future__ import annotations

import ctypes
import math
import os
import threading
import time

import numpy as np

from . import _lib
from .device import Device
from .params import QdParams


def band_ranges(n_lat, world):
    """[(row0, n_rows)] for each band: contiguous, as even as possible, south to north."""
    base, rem = divmod(int(n_lat), int(world))
    out, r = [], 0
    for k in range(world):
        n = base + (1 if k < rem else 0)
        out.append((r, n))
        r += n
    return out


def adv_reach(n_lat, dt, vmax=250.0, a=6.371e6):
    return int(math.ceil(vmax * dt / (a * (math.pi / (n_lat - 1))))) + 1


def required_halo(n_lat, dt=300.0):
    """Rows of halo that let one atmosphere step run on a single exchange: column (0) -> T_s/q gather
    (R) -> fused momentum + del^4 (5 beyond the gathered q) -> Shapiro n=2 (2) -> cloud gather (R)."""
    R = adv_reach(n_lat, dt)
    return max(12, 2 * R + 4 + 2 + 2)


def preferred_halo(n_lat, world, dt=300.0):
    """Halo used by multi-rank runs: at least `required_halo`, and deep enough (32 rows where the bands are tall enough) that
    the ocean sub-step loop exchanges once per ~3 sub-steps instead of every sub-step -- a halo exchange costs a collective's
    latency (~30 us) whatever its size.  Measured on the RCCL self-ring, 721x1440: a 1/8 band runs at 1.08 / 1.03 / 0.96 ms per
    step with 16 / 32 / 48 halo rows; the self-ring moves halos at HBM speed, over xGMI (~50 GB/s per direction) the 4.4 MB that
    a 48-row exchange of the eight ocean slabs carries would cost more than the latency it saves, hence 32.  QD_BAND_HALO
    overrides."""
    import os
    env = os.environ.get("QD_BAND_HALO")
    req = required_halo(n_lat, dt)
    if env:
        return max(req, int(env))
    rows = min(n for _, n in band_ranges(n_lat, world))
    deep = min(32, rows, (int(n_lat) - rows - 1) // 2)     # a band is at least as tall as its halo; slab rows stay <= n_lat
    return max(req, deep)


class BandGroup:
    """N band handles of one grid in one process (threads); test vehicle for the band logic."""

    def __init__(self, grid, world, params: QdParams | None = None, halo=None, device=0):
        self.grid = grid
        self.world = int(world)
        self.ranges = band_ranges(grid.n_lat, world)
        self.halo = int(halo if halo is not None else required_halo(grid.n_lat))
        p = params or QdParams.from_env()
        self.devs = []
        saved = getattr(grid, "_device", None)
        for rank, (r0, n) in enumerate(self.ranges):
            self.devs.append(Device(grid, p, device=device, row0=r0, n_rows=n, halo=self.halo, rank=rank, world=world))
        grid._device = saved
        lib = _lib.load()
        arr = (ctypes.c_void_p * world)(*[d.h for d in self.devs])
        if lib.qd_comm_init_local(arr, world) != 0:
            raise _lib.QdError("qd_comm_init_local failed")

    def set(self, name, arr):
        for d in self.devs:
            d.upload_now(name, arr)

    def get(self, name):
        out = np.zeros(self.devs[0].shape, dtype=np.uint8 if name.endswith("MASK") else np.float64)
        for d in self.devs:
            d._host.pop(name, None)
            part = d.get(name)            # qd_download writes only the owned rows of the global array
            r0, n = self.ranges[d_rank(d, self)]
            out[r0:r0 + n] = part[r0:r0 + n]
            d._host.pop(name, None)
        return out

    def run(self, fn):
        """fn(dev, rank) in one thread per band (the C side rendezvous at every exchange)."""
        errs = [None] * self.world

        def work(k):
            try:
                fn(self.devs[k], k)
            except Exception as e:       # noqa: BLE001
                errs[k] = e
        ths = [threading.Thread(target=work, args=(k,)) for k in range(self.world)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for e in errs:
            if e is not None:
                raise e

    def exchanges(self):
        out = []
        for d in self.devs:
            n = ctypes.c_int(0)
            d.lib.qd_comm_stats(d.h, ctypes.byref(n))
            out.append(n.value)
        return out

    def allreduces(self):
        out = []
        for d in self.devs:
            n = ctypes.c_int(0)
            d.lib.qd_comm_allreduce_count(d.h, ctypes.byref(n))
            out.append(n.value)
        return out

    def close(self):
        for d in self.devs:
            d.close()


def d_rank(dev, group):
    return group.devs.index(dev)


def init_rccl(dev: Device, rank, world, tag="id", timeout_s=180.0):
    """Rank 0 creates the ncclUniqueId and publishes it in a file keyed by the launcher's
    MASTER_PORT / run id; everyone calls ncclCommInitRank through the library."""
    lib = dev.lib
    # unique per launch: all ranks of one torchrun share the agent as parent process
    key = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{world}_{tag}"
    path = os.path.join("/tmp", f"qd_rdzv_{key}")
    buf = (ctypes.c_char * 128)()
    if rank == 0:
        if lib.qd_comm_unique_id(buf, 128) != 0:
            raise _lib.QdError("qd_comm_unique_id failed")
        with open(path + ".tmp", "wb") as fh:
            fh.write(bytes(buf))
        os.replace(path + ".tmp", path)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout_s:
                raise _lib.QdError("RCCL rendezvous timeout")
            time.sleep(0.01)
        with open(path, "rb") as fh:
            data = fh.read()
        ctypes.memmove(buf, data, min(128, len(data)))
    if lib.qd_comm_init(dev.h, buf, 128) != 0:
        raise _lib.QdError("qd_comm_init failed: " + (lib.qd_last_error(dev.h) or b"?").decode())
    # QD_HOST_RING=1: the ranks of one node also meet in a shared-memory ring for the few host-visible scalars of a step (eta
    # sums, CFL maxima) instead of one RCCL all-reduce each.  Opt-in: on the one-rank self-ring a host round trip per ocean
    # sub-step (~33 us) is slower than a queued one-rank ncclAllReduce (~23 us); it can only win where a real N-rank all-reduce
    # costs more than that, which this pool's 1-GPU boxes cannot tell.
    if os.environ.get("QD_HOST_RING") == "1":
        if lib.qd_comm_init_shm(dev.h, f"/qd_ring_{key}".encode()) != 0:
            raise _lib.QdError("qd_comm_init_shm failed: " + (lib.qd_last_error(dev.h) or b"?").decode())
