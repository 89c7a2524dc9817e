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


def _rendezvous_key(world, tag):
    """Unique per launch AND per restart of a launch: MASTER_PORT, the elastic run id and restart count when a launcher sets
    them, the launcher's pid (all ranks of one torchrun share the agent as parent process), the world size."""
    env = os.environ
    return "_".join(str(x) for x in (env.get("MASTER_PORT", "0"), env.get("TORCHELASTIC_RUN_ID", "none"),
                                     env.get("TORCHELASTIC_RESTART_COUNT", "0"), os.getppid(), world, tag))


def _write_atomic(path, data):
    with open(path + f".tmp{os.getpid()}", "wb") as fh:
        fh.write(data)
    os.replace(path + f".tmp{os.getpid()}", path)


def exchange_unique_id(rank, world, make_id, key, before_publish=None, timeout_s=180.0, root="/tmp"):
    """File rendezvous that cannot be fooled by what an earlier run left behind under the same key.

    Every rank k > 0 draws a fresh 16-byte nonce and posts it as <path>.req.k (re-posting it whenever the file disappears);
    rank 0 first deletes every file of the key (stale id, stale requests), then waits for the world-1 requests, then publishes
    ONE id file = all nonces + the id.  Rank k accepts an id file only if slot k holds ITS nonce -- a file from an earlier run
    cannot.  `before_publish()` runs on rank 0 after the id exists and before it is published (creates what the other ranks will
    open next, e.g. the shared-memory ring).  Returns the id bytes; finish_rendezvous() removes the files after a barrier."""
    import glob
    path = os.path.join(root, f"qd_rdzv_{key}")
    if rank == 0:
        for f in glob.glob(path + "*"):
            try:
                os.remove(f)
            except OSError:
                pass
        uid = bytes(make_id())
        nonces = [b"\0" * 16] * world
        t0 = time.time()
        for k in range(1, world):
            while True:
                try:
                    with open(f"{path}.req.{k}", "rb") as fh:
                        d = fh.read()
                    if len(d) == 16:
                        nonces[k] = d
                        break
                except OSError:
                    pass
                if time.time() - t0 > timeout_s:
                    raise _lib.QdError(f"rendezvous timeout: rank {k} never posted its request")
                time.sleep(0.005)
        if before_publish is not None:
            before_publish()
        _write_atomic(path, b"QDRZ" + b"".join(nonces) + uid)
        return uid
    nonce = os.urandom(16)
    req = f"{path}.req.{rank}"
    t0 = time.time()
    while True:
        if not os.path.exists(req):
            _write_atomic(req, nonce)                      # (re-)post: rank 0 wipes the key's files when it starts
        try:
            with open(path, "rb") as fh:
                d = fh.read()
            if len(d) > 4 + 16 * world and d[:4] == b"QDRZ" and d[4 + 16 * rank: 20 + 16 * rank] == nonce:
                return d[4 + 16 * world:]
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise _lib.QdError("rendezvous timeout: no id file carrying this rank's nonce")
        time.sleep(0.005)


def finish_rendezvous(rank, key, root="/tmp"):
    """After a barrier that every rank has passed: rank 0 removes the key's files."""
    import glob
    if rank == 0:
        for f in glob.glob(os.path.join(root, f"qd_rdzv_{key}") + "*"):
            try:
                os.remove(f)
            except OSError:
                pass


def peer_exchange_wanted():
    """QD_PEER_EXCHANGE: "1" (default) = halos and global sums through device-side mailboxes over the peer mapping (qd_peer.hip),
    "0" = RCCL collectives."""
    return os.environ.get("QD_PEER_EXCHANGE", "1") != "0"


def init_peer(dev: Device, rank, world, tag="peer", timeout_s=180.0):
    """One process per GPU, no communication library: every rank allocates its mailbox and exports its IPC handle
    (qd_peer_export), the handles travel through the filesystem under a session token that rank 0 draws (exchange_unique_id: a
    file of an earlier run cannot carry it), every rank maps all mailboxes (qd_peer_connect).  Returns True when the transport is
    up on EVERY rank; False (on every rank alike) when any rank could not export or map -- the caller falls back to RCCL."""
    lib = dev.lib
    key = _rendezvous_key(world, tag)
    token = exchange_unique_id(rank, world, lambda: os.urandom(32), key, timeout_s=timeout_s)[:16].hex()
    base = os.path.join("/tmp", f"qd_rdzv_{key}.{token}")
    buf = (ctypes.c_char * 64)()
    ok = lib.qd_peer_export(dev.h, buf, 64) == 0
    _write_atomic(f"{base}.ipc.{rank}", (b"\1" if ok else b"\0") + bytes(buf))

    def gather(suffix, size):
        out, t0 = [], time.time()
        for k in range(world):
            while True:
                try:
                    with open(f"{base}.{suffix}.{k}", "rb") as fh:
                        d = fh.read()
                    if len(d) == size:
                        out.append(d)
                        break
                except OSError:
                    pass
                if time.time() - t0 > timeout_s:
                    raise _lib.QdError(f"peer rendezvous timeout: rank {k} never posted .{suffix}")
                time.sleep(0.002)
        return out
    recs = gather("ipc", 65)
    good = all(r[:1] == b"\1" for r in recs)
    if good:
        blob = b"".join(r[1:] for r in recs)
        arr = (ctypes.c_char * len(blob)).from_buffer_copy(blob)
        good = lib.qd_peer_connect(dev.h, arr, 64, world) == 0
    # second round: did EVERY rank map every mailbox?  (nobody may store into a mailbox before all ranks agree)
    _write_atomic(f"{base}.map.{rank}", b"\1" if good else b"\0")
    good = all(r == b"\1" for r in gather("map", 1))
    if good:
        # third round: does the transport move the RIGHT bytes on this machine?  (ring exchanges of self-describing rows and
        # all-reduces with known results, both buffer parities: stale data over a mapping nobody has run this on would show here)
        wrong = ctypes.c_longlong(-1)
        ok = lib.qd_peer_selftest(dev.h, int(os.environ.get("QD_PEER_SELFTEST", "6")), ctypes.byref(wrong)) == 0 and wrong.value == 0
        _write_atomic(f"{base}.test.{rank}", b"\1" if ok else b"\0")
        good = all(r == b"\1" for r in gather("test", 1))
        if not good:
            lib.qd_peer_disable(dev.h)
    if good:
        dev._chk(lib.qd_comm_barrier(dev.h), "qd_comm_barrier")
    _write_atomic(f"{base}.done.{rank}", b"\1")
    if rank == 0:
        gather("done", 1)
        import glob
        for f in glob.glob(os.path.join("/tmp", f"qd_rdzv_{key}") + "*"):
            try:
                os.remove(f)
            except OSError:
                pass
    return good


def init_comm(dev: Device, rank, world, timeout_s=180.0):
    """Wire a one-process-per-GPU band handle to its peers: the device-side peer exchange (default), RCCL when QD_PEER_EXCHANGE=0
    or when the mailboxes cannot be mapped on every rank.  Returns the transport's name."""
    if peer_exchange_wanted():
        if init_peer(dev, rank, world, timeout_s=timeout_s):
            return "peer"
        if os.environ.get("QD_PEER_EXCHANGE") == "1":
            raise _lib.QdError("QD_PEER_EXCHANGE=1 but the peer mailboxes could not be mapped: "
                               + (dev.lib.qd_last_error(dev.h) or b"?").decode())
    init_rccl(dev, rank, world, timeout_s=timeout_s)
    return "rccl"


def init_rccl(dev: Device, rank, world, tag="id", timeout_s=180.0):
    """One process per GPU: rank 0 creates the ncclUniqueId and hands it to the others through exchange_unique_id(); everyone
    calls ncclCommInitRank through the library; a barrier; rank 0 removes the rendezvous files."""
    lib = dev.lib
    key = _rendezvous_key(world, tag)
    # QD_HOST_RING=1: the ranks of one node also meet in a shared-memory ring for the few host-visible scalars of a step (eta
    # sums, CFL maxima) instead of one RCCL all-reduce each.  Opt-in: on the one-rank self-ring a host round trip per ocean
    # sub-step (~33 us) is slower than a queued one-rank ncclAllReduce (~23 us); it can only win where a real N-rank all-reduce
    # costs more than that, which this pool's 1-GPU boxes cannot tell.
    ring = os.environ.get("QD_HOST_RING") == "1"
    ring_name = f"/qd_ring_{key}".encode()

    def make_id():
        buf = (ctypes.c_char * 128)()
        if lib.qd_comm_unique_id(buf, 128) != 0:
            raise _lib.QdError("qd_comm_unique_id failed")
        return bytes(buf)

    def open_ring():
        if lib.qd_comm_init_shm(dev.h, ring_name) != 0:
            raise _lib.QdError("qd_comm_init_shm failed: " + (lib.qd_last_error(dev.h) or b"?").decode())

    # rank 0 creates the ring segment (unlinking a stale one) BEFORE the id is published; the others open it after reading the id
    uid = exchange_unique_id(rank, world, make_id, key, before_publish=open_ring if ring else None, timeout_s=timeout_s)
    buf = (ctypes.c_char * 128)()
    ctypes.memmove(buf, uid, min(128, len(uid)))
    if ring and rank != 0:
        open_ring()
    if lib.qd_comm_init(dev.h, buf, 128) != 0:
        raise _lib.QdError("qd_comm_init failed: " + (lib.qd_last_error(dev.h) or b"?").decode())
    dev._chk(lib.qd_comm_barrier(dev.h), "qd_comm_barrier")
    finish_rendezvous(rank, key)
