// qd_band.hip -- latitude-band decomposition: validity margins, launch segments,
// halo exchange and small all-reduces.
//
// A band handle owns global rows [row0, row0+n) and keeps `halo` extra rows on each side.  Halos
// form a RING in latitude: the south halo of band 0 holds the last rows of the last band (period
// n_lat), which is exactly what the reference's np.roll(axis=0) stencils and the period-(n-1)
// map_coordinates fold read at the poles.  Every slab carries a validity margin vm = number of rows
// beyond the owned band that hold current data.  A stencil of reach r run on inputs with margin m
// produces outputs with margin m - r ("deep halo": halo rows are recomputed redundantly instead of
// being exchanged after every kernel); when an input's margin is too small the planner exchanges
// that slab's halos (margin back to `halo`).  Whole-globe handles skip all of this.
//
// Transports: RCCL (one process per GPU, grouped ncclSend/ncclRecv on the handle's stream) or an
// in-process group of handles on one device (device-to-device copies + a pthread barrier), used to
// test the band logic on a single GPU.
#include "qd_internal.h"
#include "qd_band.h"
#include <rccl/rccl.h>
#include <pthread.h>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

// ------------------------------------------------------------------ host ring: scalar all-reduce in shared memory
// The eta sum between two ocean sub-steps and the CFL maxima are a handful of doubles that the host of every rank can read
// (one event wait that the next launch needs anyway).  An RCCL all-reduce costs ~30 us for them, and the sub-step loop
// needs one per sub-step; ranks of one node instead meet in a POSIX shared-memory segment (threads of an in-process group:
// in plain memory): every rank publishes its values under a sequence number, waits until all ranks have published that
// sequence, and reduces the slots in rank order -- the same sum on every rank, bit for bit, in ~1-2 us.  Two buffers by
// sequence parity: a rank can only be one call ahead of the slowest one, because call s + 1 needs everybody's s + 1.
static_assert(std::atomic<unsigned long long>::is_always_lock_free, "the ring needs lock-free 64-bit atomics");

static int ring_allreduce(QdHostRing* r, double* v, int n, int op, double timeout_s) {
    if (!r || !r->seg || n < 1 || n > QD_RING_MAXVALS) return -1;
    QdRingSeg* g = r->seg;
    const unsigned long long s = ++r->my_seq;
    const int buf = (int)(s & 1ull);
    for (int k = 0; k < n; ++k) g->vals[buf][r->rank][k] = v[k];
    g->seq[r->rank].store(s, std::memory_order_release);
    const auto t0 = std::chrono::steady_clock::now();
    for (int q = 0; q < r->world; ++q) {
        unsigned spins = 0;
        while (g->seq[q].load(std::memory_order_acquire) < s) {
            if ((++spins & 0x3FFu) == 0) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return -2;
                sched_yield();
            }
        }
    }
    for (int k = 0; k < n; ++k) {
        double a = g->vals[buf][0][k];
        for (int q = 1; q < r->world; ++q) { const double b = g->vals[buf][q][k]; a = op ? (b > a ? b : a) : a + b; }
        v[k] = a;
    }
    return 0;
}

// handle-free entry points (tests/test_bands_cpu.py drives them from several processes without a GPU)
// Rank 0 OWNS the segment: it removes whatever a crashed earlier run left under the name, creates it exclusively and sizes it
// (new pages read as zero: sequence 0); the other ranks only ever open an existing, fully sized segment and wait for it to
// appear.  Callers that can reuse a name across runs must not let a non-zero rank in before rank 0 has created the segment of
// THIS run (qingdai_amd.bands.init_rccl publishes the RCCL id only afterwards).
extern "C" int qd_hostring_open(const char* name, int rank, int world, void** out) {
    if (!name || !out || world < 1 || world > QD_RING_MAXRANKS || rank < 0 || rank >= world) return -1;
    QdHostRing* r = new QdHostRing();
    r->rank = rank; r->world = world; r->name = name;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name);                                    // stale segment of a crashed run: its counters are not zero
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) { delete r; return -2; }
        if (ftruncate(fd, sizeof(QdRingSeg)) != 0) { close(fd); shm_unlink(name); delete r; return -3; }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            if (fd >= 0) {
                struct stat st;
                if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(QdRingSeg)) break;
                close(fd); fd = -1;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) { delete r; return -2; }
            usleep(2000);
        }
    }
    void* m = mmap(nullptr, sizeof(QdRingSeg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { if (rank == 0) shm_unlink(name); delete r; return -4; }
    r->seg = (QdRingSeg*)m; r->mapped = true; r->owner = (rank == 0);
    *out = r;
    return 0;
}
extern "C" int qd_hostring_allreduce(void* ring, double* vals, int n, int op_max) {
    return ring_allreduce((QdHostRing*)ring, vals, n, op_max, 60.0);
}
extern "C" int qd_hostring_close(void* ring) {
    QdHostRing* r = (QdHostRing*)ring;
    if (!r) return 0;
    if (r->mapped && r->seg) munmap(r->seg, sizeof(QdRingSeg));
    if (r->mapped && r->owner) shm_unlink(r->name.c_str());
    delete r;
    return 0;
}

// planner simulation: a band handle WITHOUT a device -- only geometry, validity margins and the exchange log.  qd_plan,
// qd_mark, qd_segments and the bookkeeping of qd_exchange are pure host code; these hooks let a CPU-only test drive them and
// perform the recorded exchanges itself (torch.distributed / gloo) on real arrays.
struct QdPlanSim { std::vector<std::vector<int>> log; };


// all-reduce of a few HOST doubles across the ranks of this handle's communicator; -1 when no host ring is attached
int qd_host_allreduce(qd_ctx* c, double* v, int n, int op) {
    if (c->geo.full) return 0;
    QdHostRing* r = (QdHostRing*)c->hring;
    if (!r) return -1;
    c->host_allreduces++;
    const int rc = ring_allreduce(r, v, n, op, 120.0);
    if (rc == -2) return qd_fail(c, "host ring: a rank did not arrive within 120 s");
    return rc ? qd_fail(c, "host ring: bad call") : 0;
}
bool qd_has_host_ring(const qd_ctx* c) { return c->hring != nullptr; }

int qd_vm_get(qd_ctx* c, const void* slab) {
    if (c->geo.full) return INT_MAX / 2;
    auto it = c->vm.find(slab);
    return it == c->vm.end() ? 0 : it->second;
}

void qd_mark(qd_ctx* c, std::initializer_list<const void*> out, int margin) {
    if (c->geo.full) return;
    for (const void* s : out) if (s) c->vm[s] = margin;
}

QdSegs qd_segments(qd_ctx* c, int margin) {
    QdSegs S; S.n = 0;
    const QdGeom& G0 = c->geo;
    if (G0.full) { S.g[0] = G0; S.n = 1; return S; }
    const int n = G0.nlat;
    int vr = c->own_row0 - margin, cnt = c->own_nrows + 2 * margin;
    while (cnt > 0) {
        int g0, len;
        if (vr < 0) { g0 = vr + n; len = std::min(cnt, -vr); }
        else if (vr >= n) { g0 = vr - n; len = cnt; }
        else { g0 = vr; len = std::min(cnt, n - vr); }
        QdGeom g = G0; g.row0 = g0; g.nrows = len;
        S.g[S.n++] = g;
        vr += len; cnt -= len;
        if (S.n == 3) break;
    }
    return S;
}

QdSegList qd_segments_rows(qd_ctx* c, int vr, int cnt, QdSegList S) {
    const QdGeom& G0 = c->geo;
    const int n = G0.nlat;
    while (cnt > 0 && S.n < 6) {
        int g0, len;
        if (vr < 0) { g0 = vr + n; len = std::min(cnt, -vr); }
        else if (vr >= n) { g0 = vr - n; len = cnt; }
        else { g0 = vr; len = std::min(cnt, n - vr); }
        QdGeom g = G0; g.row0 = g0; g.nrows = len;
        S.g[S.n++] = g;
        vr += len; cnt -= len;
    }
    return S;
}

int qd_plan_peek(qd_ctx* c, const QdUse* in, int n) {
    if (c->geo.full) return 0;
    int out = INT_MAX;
    for (int k = 0; k < n; ++k) if (*in[k].slot) out = std::min(out, qd_vm_get(c, *in[k].slot) - in[k].radius);
    return out == INT_MAX ? c->geo.halo : out;
}

static int qd_plan_impl(qd_ctx* c, const QdUse* in_, int n_in, int want, bool* pending);
int qd_plan(qd_ctx* c, const QdUse* in_, int n_in, int want) { return qd_plan_impl(c, in_, n_in, want, nullptr); }
int qd_plan_begin(qd_ctx* c, const QdUse* in, int n, bool* pending) { *pending = false; return qd_plan_impl(c, in, n, INT_MAX, pending); }
int qd_plan_end(qd_ctx* c) {
    if (c->split_pending.empty()) return 0;
    if (qd_peer_halo_end(c)) return -1;
    for (const QdUse& u : c->split_pending) c->vm[*u.slot] = c->geo.halo;
    c->split_pending.clear();
    return 0;
}

static int qd_plan_impl(qd_ctx* c, const QdUse* in_, int n_in, int want, bool* pending) {
    struct { const QdUse* p; int n; const QdUse* begin() const { return p; } const QdUse* end() const { return p + n; } } in{in_, n_in};
    if (c->geo.full) return 0;
    const int H = c->geo.halo;
    bool need = false;
    for (const QdUse& u : in) {
        if (!*u.slot) continue;
        if (u.radius > H) return qd_fail(c, "band halo narrower than a stencil reach (create the handle with a larger halo)") , -1;
        if (qd_vm_get(c, *u.slot) < u.radius) need = true;
    }
    if (need) {
        // refresh every listed slab that is not already at full margin in the same grouped exchange
        std::vector<QdUse> ex;
        for (const QdUse& u : in) if (*u.slot && qd_vm_get(c, *u.slot) < H) ex.push_back(u);
        // slabs of the enclosing loop that will run out of margin soon ride on the same grouped exchange: one collective
        // instead of several out-of-phase ones (an exchange is latency, not bandwidth)
        for (const QdUse& u : c->corefresh) {
            if (!*u.slot || qd_vm_get(c, *u.slot) >= H) continue;
            bool dup = false;
            for (const QdUse& e : ex) dup |= (*e.slot == *u.slot);
            if (!dup) ex.push_back(u);
        }
        if (pending && qd_peer_overlap(c) && (int)ex.size() <= qd_peer_max_slabs() && c->own_nrows >= H) {
            // only the push: the caller computes what the old margins allow, then qd_plan_end()
            c->exchanges += 1;
            if (qd_peer_halo_begin(c, ex.data(), (int)ex.size(), qd_peer_overlap(c) >= 2)) return -1;
            c->split_pending = ex;
            *pending = true;
            return 0;
        }
        if (qd_exchange(c, ex.data(), (int)ex.size())) return -1;
    }
    int out = INT_MAX;
    for (const QdUse& u : in) if (*u.slot) out = std::min(out, qd_vm_get(c, *u.slot) - u.radius);
    if (out == INT_MAX) out = H;
    out = std::min(out, want);
    return out < 0 ? 0 : out;
}

int qd_exchange(qd_ctx* c, const QdUse* slots, int n) {
    if (c->geo.full || n == 0) return 0;
    const QdGeom& G = c->geo;
    const int H = G.halo, nown = c->own_nrows, world = c->desc.world, rank = c->desc.rank;
    if (nown < H) return qd_fail(c, "latitude band thinner than its halo");
    const int up = (rank + 1) % world, dn = (rank - 1 + world) % world;
    c->exchanges += 1;
    if (qd_peer_on(c)) {
        // device-side exchange (qd_peer.hip): stores into the neighbours' mailboxes, no collective launch
        if (qd_peer_halo(c, slots, n)) return -1;
    } else if (c->comm) {
        ncclComm_t comm = (ncclComm_t)c->comm;
        ncclGroupStart();
        if (c->pending_sum) {                                 // a scalar sum nobody has needed yet rides in this group (qd_allreduce_sum_deferred)
            ncclAllReduce(c->pending_sum, c->pending_sum, 1, ncclDouble, ncclSum, comm, c->stream);
            c->pending_sum = nullptr; c->grouped_sums++;
        }
        for (int k = 0; k < n; ++k) {
            const size_t esz = slots[k].u8 ? 1 : sizeof(double);
            const size_t cnt = (size_t)H * G.nlon;
            char* base = (char*)*slots[k].slot;
            const ncclDataType_t ty = slots[k].u8 ? ncclUint8 : ncclDouble;
            ncclSend(base + (size_t)(H + nown - H) * G.nlon * esz, cnt, ty, up, comm, c->stream);   // my top rows -> up's south halo
            ncclRecv(base, cnt, ty, dn, comm, c->stream);                                            // my south halo <- dn's top rows
            ncclSend(base + (size_t)H * G.nlon * esz, cnt, ty, dn, comm, c->stream);                 // my bottom rows -> dn's north halo
            ncclRecv(base + (size_t)(H + nown) * G.nlon * esz, cnt, ty, up, comm, c->stream);        // my north halo <- up's bottom rows
        }
        ncclResult_t r = ncclGroupEnd();
        if (r != ncclSuccess) { c->err = std::string("halo exchange: ") + ncclGetErrorString(r); return -1; }
    } else if (c->lgroup) {
        QdLocalGroup* g = c->lgroup;
        QD_HIP(c, hipStreamSynchronize(c->stream));
        pthread_barrier_wait(&g->bar);
        qd_ctx* pu = g->peers[up];
        qd_ctx* pd = g->peers[dn];
        for (int k = 0; k < n; ++k) {
            const size_t esz = slots[k].u8 ? 1 : sizeof(double);
            const size_t bytes = (size_t)H * G.nlon * esz;
            const size_t off = (char*)slots[k].slot - (char*)c;               // same logical slot in the peer context
            char* mine = (char*)*slots[k].slot;
            const char* dnb = (const char*)*(void**)((char*)pd + off);
            const char* upb = (const char*)*(void**)((char*)pu + off);
            QD_HIP(c, hipMemcpyAsync(mine, dnb + (size_t)pd->own_nrows * G.nlon * esz, bytes, hipMemcpyDeviceToDevice, c->stream));
            QD_HIP(c, hipMemcpyAsync(mine + (size_t)(H + nown) * G.nlon * esz, upb + (size_t)H * G.nlon * esz, bytes,
                                     hipMemcpyDeviceToDevice, c->stream));
        }
        QD_HIP(c, hipStreamSynchronize(c->stream));
        pthread_barrier_wait(&g->bar);
    } else if (c->plansim) {
        // planner simulation (no device, no transport): record WHAT would move -- the caller moves it (tests/test_bands_cpu.py)
        QdPlanSim* ps = (QdPlanSim*)c->plansim;
        std::vector<int> rec;
        for (int k = 0; k < n; ++k) rec.push_back((int)((double**)slots[k].slot - c->f));
        rec.push_back(H); rec.push_back(nown); rec.push_back(up); rec.push_back(dn);
        ps->log.push_back(rec);
    } else {
        return qd_fail(c, "band handle without a communicator (qd_comm_init / qd_comm_init_local)");
    }
    for (int k = 0; k < n; ++k) c->vm[*slots[k].slot] = H;
    return 0;
}

// A one-double sum whose result is first read by a LATER kernel (the eta mean of an ocean sub-step: the next momentum kernel
// applies it on load): over RCCL it is not issued here but remembered, and goes out inside the ncclGroup of the next halo exchange
// if one comes before qd_allreduce_flush -- one group launch instead of two.  Other transports reduce at once.
int qd_allreduce_sum_deferred(qd_ctx* c, double* dptr) {
    if (c->geo.full) return 0;
    if (!c->comm || !c->group_sums) return qd_allreduce_f64(c, dptr, 1, 0);
    if (c->pending_sum && qd_allreduce_flush(c)) return -1;
    c->allreduces++;
    c->pending_sum = dptr;
    return 0;
}
int qd_allreduce_flush(qd_ctx* c) {
    if (!c->pending_sum) return 0;
    double* p = c->pending_sum;
    c->pending_sum = nullptr;
    ncclResult_t r = ncclAllReduce(p, p, 1, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream);
    if (r != ncclSuccess) { c->err = std::string("allreduce: ") + ncclGetErrorString(r); return -1; }
    return 0;
}

int qd_allreduce_f64(qd_ctx* c, double* dptr, int n, int op) {
    if (c->geo.full) return 0;
    c->allreduces++;
    if (qd_peer_on(c)) return qd_peer_allreduce(c, dptr, n, op ? 1 : 0);
    if (c->comm) {
        ncclResult_t r = ncclAllReduce(dptr, dptr, n, ncclDouble, op ? ncclMax : ncclSum, (ncclComm_t)c->comm, c->stream);
        if (r != ncclSuccess) { c->err = std::string("allreduce: ") + ncclGetErrorString(r); return -1; }
        return 0;
    }
    if (c->lgroup) {
        QdLocalGroup* g = c->lgroup;
        const int world = c->desc.world, rank = c->desc.rank;
        if (n > 64) return qd_fail(c, "qd_allreduce_f64: n > 64");
        QD_HIP(c, hipMemcpyAsync(&g->stage_d[(size_t)rank * 64], dptr, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        QD_HIP(c, hipStreamSynchronize(c->stream));
        pthread_barrier_wait(&g->bar);
        double acc[64];
        for (int k = 0; k < n; ++k) {
            double a = g->stage_d[k];
            for (int r = 1; r < world; ++r) { const double b = g->stage_d[(size_t)r * 64 + k]; a = op ? (b > a ? b : a) : a + b; }
            acc[k] = a;
        }
        pthread_barrier_wait(&g->bar);
        QD_HIP(c, hipMemcpyAsync(dptr, acc, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        QD_HIP(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    return qd_fail(c, "band handle without a communicator");
}

// all-reduce of a few device doubles that the host needs next (CFL maxima): result in dptr AND in pinned host memory hdst, the call
// returns when the host can read it.  Peer exchange: the reduction kernel publishes it itself; other transports: reduce, then
// qd_fetch_scalars.
int qd_allreduce_fetch(qd_ctx* c, double* dptr, int n, int op, double* hdst) {
    if (c->geo.full) return qd_fetch_scalars(c, dptr, n, hdst);
    if (qd_peer_on(c) && n <= 256) {
        c->allreduces++;
        c->pub_seq += 1.0;
        if (qd_peer_allreduce_publish(c, dptr, n, op, hdst, c->pub_seq)) return -1;
        if (qd_wait_host_flag(c, c->hpin + 61, c->pub_seq, "reduced scalars never reached the host")) return -1;
        if (c->hpin[60] != 0.0) return qd_fail(c, "peer exchange: a rank did not arrive within the deadline");
        return 0;
    }
    if (qd_allreduce_f64(c, dptr, n, op)) return -1;
    return qd_fetch_scalars(c, dptr, n, hdst);
}

#define QD_LOCAL_U32_MAX (64 * 2 * 4096)        // the gathered median segments of up to 64 in-process bands
// buf[world][n_per_rank] with this rank's segment filled in -> every segment everywhere.  Transports without a gather of their own
// all-reduce the zero-padded buffer as integers (x + 0 + ... + 0 is exact): the caller clears the other segments first.
int qd_allgather_f64(qd_ctx* c, double* buf, int n_per_rank) {
    if (c->geo.full) return 0;
    if (qd_peer_on(c)) { c->allreduces++; return qd_peer_allgather(c, buf, n_per_rank); }
    return qd_allreduce_u32(c, (unsigned int*)buf, 2 * c->desc.world * n_per_rank);
}

int qd_allreduce_u32(qd_ctx* c, unsigned int* dptr, int n) {
    if (c->geo.full) return 0;
    c->allreduces++;
    if (qd_peer_on(c)) return qd_peer_allreduce(c, dptr, n, 2);
    if (c->comm) {
        ncclResult_t r = ncclAllReduce(dptr, dptr, n, ncclUint32, ncclSum, (ncclComm_t)c->comm, c->stream);
        if (r != ncclSuccess) { c->err = std::string("allreduce: ") + ncclGetErrorString(r); return -1; }
        return 0;
    }
    if (c->lgroup) {
        QdLocalGroup* g = c->lgroup;
        const int world = c->desc.world, rank = c->desc.rank;
        if (n > QD_LOCAL_U32_MAX) return qd_fail(c, "qd_allreduce_u32: n too large for the in-process transport");
        QD_HIP(c, hipMemcpyAsync(&g->stage_u[(size_t)rank * QD_LOCAL_U32_MAX], dptr, n * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        QD_HIP(c, hipStreamSynchronize(c->stream));
        pthread_barrier_wait(&g->bar);
        std::vector<unsigned int> acc(n);
        for (int k = 0; k < n; ++k) { unsigned a = 0; for (int r = 0; r < world; ++r) a += g->stage_u[(size_t)r * QD_LOCAL_U32_MAX + k]; acc[k] = a; }
        pthread_barrier_wait(&g->bar);
        QD_HIP(c, hipMemcpyAsync(dptr, acc.data(), n * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
        QD_HIP(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    return qd_fail(c, "band handle without a communicator");
}

// ---- C-ABI ---------------------------------------------------------------------------------
extern "C" int qd_comm_unique_id(void* id128, size_t bytes) {
    if (!id128 || bytes < sizeof(ncclUniqueId)) return -1;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int qd_comm_init(qd_handle c, const void* id128, size_t bytes) {
    if (!c || !id128 || bytes < sizeof(ncclUniqueId)) return -1;
    hipSetDevice(c->desc.device);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t comm;
    ncclResult_t r = ncclCommInitRank(&comm, c->desc.world, id, c->desc.rank);
    if (r != ncclSuccess) { c->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); return -1; }
    c->comm = (void*)comm;
    return 0;
}

// handles[] ordered by rank, all on one device, each driven by its own host thread
extern "C" int qd_comm_init_local(qd_handle* handles, int n) {
    if (!handles || n < 1) return -1;
    QdLocalGroup* g = new QdLocalGroup();
    g->peers.assign(handles, handles + n);
    pthread_barrier_init(&g->bar, nullptr, (unsigned)n);
    g->stage_d.assign((size_t)n * 64, 0.0);
    g->stage_u.assign((size_t)n * QD_LOCAL_U32_MAX, 0u);
    for (int k = 0; k < n; ++k) {
        if (!handles[k] || handles[k]->desc.rank != k || handles[k]->desc.world != n) { delete g; return -1; }
        handles[k]->lgroup = g;
    }
    for (int q = 0; q < QD_RING_MAXRANKS; ++q) g->ring.seq[q].store(0ull);
    // QD_PEER_EXCHANGE=1: halos and reductions of the group go through the device-side mailboxes (qd_peer.hip); no host ring then
    // (the ring would take the eta sums and CFL maxima away from the path under test)
    const char* pe = std::getenv("QD_PEER_EXCHANGE");
    if (pe && pe[0] == '1') return qd_peer_init_group(g);
    if (!std::getenv("QD_NO_HOST_RING"))
        for (int k = 0; k < n; ++k) {
            QdHostRing* r = new QdHostRing();
            r->seg = &g->ring; r->rank = k; r->world = n;
            handles[k]->hring = r;
        }
    return 0;
}

// one process per GPU: the ranks of a node meet in the POSIX shared-memory segment `name` (qingdai_amd.bands.init_rccl picks a
// name unique to the launch; rank 0 calls this -- and thereby creates the segment -- BEFORE it publishes the RCCL id, the
// other ranks after they have read the id, so nobody can map a segment of an earlier run)
extern "C" int qd_comm_init_shm(qd_handle c, const char* name) {
    if (!c || !name) return -1;
    if (c->geo.full) return 0;
    void* r = nullptr;
    const int rc = qd_hostring_open(name, c->desc.rank, c->desc.world, &r);
    if (rc) return qd_fail(c, "qd_comm_init_shm: shm_open / mmap failed");
    if (c->hring) qd_hostring_close(c->hring);
    c->hring = r;
    return 0;
}
// called by qd_destroy: communicator, host ring and (by the last peer) the in-process group
void qd_comm_release(qd_ctx* c) {
    qd_peer_release(c);
    if (c->comm) { ncclCommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
    if (c->hring) {
        QdHostRing* r = (QdHostRing*)c->hring;
        if (r->mapped) qd_hostring_close(r); else delete r;          // rings of an in-process group point into the group
        c->hring = nullptr;
    }
    if (c->lgroup) {
        QdLocalGroup* g = c->lgroup;
        c->lgroup = nullptr;
        bool last = true;
        for (qd_ctx*& p : g->peers) { if (p == c) p = nullptr; else if (p) last = false; }
        if (last) { pthread_barrier_destroy(&g->bar); delete g; }
    }
}

extern "C" int qd_comm_host_allreduce_count(qd_handle c, int* n) { if (!c || !n) return -1; *n = c->host_allreduces; return 0; }

extern "C" int qd_comm_allreduce_max(qd_handle c, double* inout, int n) {
    if (!c || !inout || n < 1 || n > 8) return -1;
    if (c->geo.full || (!c->comm && !c->lgroup && !qd_peer_on(c))) return 0;     // single process: identity
    hipSetDevice(c->desc.device);
    double* d = c->dscal + QD_S_TMP0 + 2;
    QD_HIP(c, hipMemcpyAsync(d, inout, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (qd_allreduce_f64(c, d, n, 1)) return -1;
    QD_HIP(c, hipMemcpyAsync(inout, d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    QD_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int qd_comm_barrier(qd_handle c) {
    if (!c) return -1;
    double z = 0.0;
    return qd_comm_allreduce_max(c, &z, 1);
}

extern "C" int qd_comm_stats(qd_handle c, int* exchanges) { if (!c || !exchanges) return -1; *exchanges = c->exchanges; return 0; }
extern "C" int qd_comm_allreduce_count(qd_handle c, int* allreduces) { if (!c || !allreduces) return -1; *allreduces = c->allreduces; return 0; }
extern "C" int qd_comm_grouped_sum_count(qd_handle c, int* grouped) { if (!c || !grouped) return -1; *grouped = (int)c->grouped_sums; return 0; }

// ---- planner simulation (host only; tests) --------------------------------------------------
extern "C" int qd_plansim_create(const qd_grid_desc* d, qd_handle* out) {
    if (!d || !out || d->n_lat < 4 || d->n_lon < 1 || d->n_rows < 1 || d->halo < 0 || d->world < 1) return -1;
    qd_ctx* c = new qd_ctx();
    c->desc = *d;
    const bool full = (d->row0 == 0 && d->n_rows == d->n_lat && d->world == 1);
    c->geo = QdGeom{d->n_lat, d->n_lon, d->row0, d->n_rows, d->halo, full ? 1 : 0, d->row0 - d->halo, d->n_rows + 2 * d->halo};
    c->own_row0 = d->row0; c->own_nrows = d->n_rows;
    c->plansim = new QdPlanSim();
    for (int f = 0; f < QD_F_COUNT_F64; ++f) c->f[f] = (double*)(uintptr_t)(0x10000u * (unsigned)(f + 1));     // identities, never dereferenced
    *out = c;
    return 0;
}
extern "C" int qd_plansim_destroy(qd_handle c) {
    if (!c || !c->plansim) return -1;
    delete (QdPlanSim*)c->plansim;
    delete c;
    return 0;
}
// margin the outputs of a launch can be computed on when it reads field[k] with stencil reach radius[k]; exchanges are logged
extern "C" int qd_plansim_plan(qd_handle c, const int* fields, const int* radii, int n, int want) {
    if (!c || !c->plansim || !fields || !radii || n < 1 || n > 16) return -1;
    QdUse u[16];
    for (int k = 0; k < n; ++k) { if (fields[k] < 0 || fields[k] >= QD_F_COUNT_F64) return -1; u[k] = QdUse{(void**)&c->f[fields[k]], radii[k], 0}; }
    // qd_plan takes an initializer_list: build the call for the common small counts
    switch (n) {
        case 1: return qd_plan(c, {u[0]}, want < 0 ? INT_MAX : want);
        case 2: return qd_plan(c, {u[0], u[1]}, want < 0 ? INT_MAX : want);
        case 3: return qd_plan(c, {u[0], u[1], u[2]}, want < 0 ? INT_MAX : want);
        case 4: return qd_plan(c, {u[0], u[1], u[2], u[3]}, want < 0 ? INT_MAX : want);
        case 5: return qd_plan(c, {u[0], u[1], u[2], u[3], u[4]}, want < 0 ? INT_MAX : want);
        case 6: return qd_plan(c, {u[0], u[1], u[2], u[3], u[4], u[5]}, want < 0 ? INT_MAX : want);
        default: return -1;
    }
}
extern "C" int qd_plansim_mark(qd_handle c, const int* fields, int n, int margin) {
    if (!c || !c->plansim || !fields) return -1;
    for (int k = 0; k < n; ++k) { if (fields[k] < 0 || fields[k] >= QD_F_COUNT_F64) return -1; qd_mark(c, {(const void*)c->f[fields[k]]}, margin); }
    return 0;
}
extern "C" int qd_plansim_margin(qd_handle c, int field) {
    if (!c || !c->plansim || field < 0 || field >= QD_F_COUNT_F64) return -1;
    return qd_vm_get(c, c->f[field]);
}
// launch segments for a margin: out[2k] = first global row, out[2k+1] = row count; returns the number of segments (<= 3)
extern "C" int qd_plansim_segments(qd_handle c, int margin, int* out) {
    if (!c || !c->plansim || !out) return -1;
    const QdSegs S = qd_segments(c, margin);
    for (int k = 0; k < S.n; ++k) { out[2 * k] = S.g[k].row0; out[2 * k + 1] = S.g[k].nrows; }
    return S.n;
}
// launch segments of an arbitrary run of ring rows [vr0, vr0 + cnt) (vr0 may be negative or beyond n_lat: the ring wraps): what the
// interior / boundary launches around an overlapped halo exchange use (qd_segments_rows); returns the number of segments (<= 6)
extern "C" int qd_plansim_segments_rows(qd_handle c, int vr0, int cnt, int* out) {
    if (!c || !c->plansim || !out) return -1;
    const QdSegList S = qd_segments_rows(c, vr0, cnt);
    for (int k = 0; k < S.n; ++k) { out[2 * k] = S.g[k].row0; out[2 * k + 1] = S.g[k].nrows; }
    return S.n;
}
// oldest logged exchange: fields_out[0..n) and geom = {halo rows H, owned rows, rank that receives my top rows (up), rank that
// receives my bottom rows (dn)}.  Slab layout: local rows [0,H) south halo <- dn's top rows [nown, nown+H) (local numbering);
// [H, H+nown) owned; [H+nown, 2H+nown) north halo <- up's bottom rows [H, 2H).  Returns n, 0 when the log is empty.
extern "C" int qd_plansim_pop_exchange(qd_handle c, int* fields_out, int max_fields, int* geom4) {
    if (!c || !c->plansim || !fields_out || !geom4) return -1;
    QdPlanSim* ps = (QdPlanSim*)c->plansim;
    if (ps->log.empty()) return 0;
    const std::vector<int> rec = ps->log.front();
    ps->log.erase(ps->log.begin());
    const int n = (int)rec.size() - 4;
    if (n > max_fields) return -1;
    for (int k = 0; k < n; ++k) fields_out[k] = rec[k];
    for (int k = 0; k < 4; ++k) geom4[k] = rec[n + k];
    return n;
}
