// qd_peer.hip -- device-side exchange over the peer mapping (QD_PEER_EXCHANGE).
//
// What the latitude bands exchange per step -- ring halos of the stencilled slabs (np.roll(axis=0) stencils pygcm/ocean.py:306-310,
// 416, dynamics.py:144-173, grid.py:56-64) and a few global scalars (the eta mean of every ocean sub-step ocean.py:369-377, CFL
// maxima, the medians' histograms) -- costs a collective LAUNCH each over RCCL (~30-37 us whatever the size, 25-38 of them per
// step).  Here nothing is launched that belongs to a communication library: every rank owns a MAILBOX in its own HBM (fine-grained
// device memory), every other rank of the node maps it (hipIpc* between processes -- over xGMI on a real node --, plain pointers
// between the band handles of one process), and small kernels on the handle's own stream STORE into the neighbours' mailboxes and
// POLL their own:
//
//   halo exchange   k_halo_push copies my H top rows of every slab of the exchange into `up`'s mailbox (staging area "from the
//                   south") and my H bottom rows into `dn`'s ("from the north"), then bumps an arrival counter there;
//                   k_halo_unpack waits until both of MY counters have reached the exchange's count and copies the staged rows
//                   into my slabs' halo rows.  Whatever the stream runs between the two launches overlaps the transfer
//                   (qd_band.hip launches the consumer's INTERIOR rows there).
//   scalar / histogram all-reduce, all-gather   k_peer_reduce: every rank deposits its n values in slot [rank] of EVERY mailbox,
//                   bumps the counter [rank] there, waits until all `world` counters of its own mailbox have arrived and reduces
//                   the slots in rank order -- the same bits on every rank, no host, no collective.
//
// No acknowledgements are needed: both staging areas and the reduction slots are double-buffered by sequence parity, and a rank
// can only be ONE exchange ahead of a neighbour -- its exchange s + 1 completes only when the neighbour's push s + 1 has arrived,
// which that neighbour issued (stream order) after its own unpack s had finished reading buffer s % 2; so nobody overwrites
// buffer s % 2 with exchange s + 2 before it has been consumed.  (Every exchange is the symmetric ring exchange, every reduction
// involves all ranks, and all ranks issue the same sequence of them: they run the same program on the same global scalars.)
//
// Memory model: data stores, then __threadfence_system() (release: L2 write-back of what is not already written through),
// workgroup barrier, ONE system-scope atomic add on the consumer's counter; the consumer polls with system-scope atomic loads, then
// fences (acquire) before it reads.  The mailbox is allocated fine-grained (hipDeviceMallocFinegrained: stores write through, loads
// do not linger in L2), so the fences have nothing to flush.  Every poll loop has a deadline (QP_TIMEOUT_S of s_memrealtime): a
// rank that never arrives turns into an error word in pinned host memory and a failed qd_* call, never into a hung GPU.
//
// In-process groups (N band handles on one device, one host thread each: the test vehicle) run the same kernels in two launches
// per operation -- deposit, pthread barrier, collect: with all deposits queued before any collect, no kernel ever polls for work
// that sits BEHIND it in a shared hardware queue (HIP multiplexes streams onto a few of those).
#include "qd_internal.h"
#include "qd_band.h"
#include <cstring>
#include <algorithm>

#define QP_MAXSLABS 16
#define QP_RV 4104                      // 8-byte units per rank slot of a reduction (median segment: 4096 + 4)
#define QP_HDR 4096                     // header: arrival counters
#define QP_OFF_HCNT 0                   // u64[2]: blocks of pushes arrived for my south halo (from dn) / my north halo (from up)
#define QP_OFF_RCNT 512                 // u64[world]: blocks of deposits arrived from rank q
#define QP_PUSH_BLOCKS 64
#define QP_TIMEOUT_S 20.0

struct QdPeerHalo {
    void* slab[QP_MAXSLABS];
    unsigned char u8[QP_MAXSLABS];
    int n, H, nown, nlon;
};

struct QdPeer {
    int on = 0, local = 0, world = 1, rank = 0, up = 0, dn = 0;
    char* box = nullptr;                          // my mailbox
    size_t box_bytes = 0, off_rv = 0, rv_stride = 0, off_stage = 0, slab_stride = 0, dir_stride = 0, par_stride = 0;
    char* pbox[QD_RING_MAXRANKS] = {nullptr};     // every rank's mailbox as THIS process maps it ([rank] == box)
    bool opened[QD_RING_MAXRANKS] = {false};      // mapped through hipIpcOpenMemHandle (to be closed)
    char** d_pbox = nullptr;                      // the same table on the device
    unsigned long long hseq = 0, hexp = 0;        // halo exchanges so far, arrival count every exchange so far adds up to
    unsigned long long rseq = 0, rexp = 0;        // reductions so far, deposits per source rank so far
    double* herr = nullptr;                       // pinned host word: a poll loop ran into its deadline
    long n_halo = 0, n_reduce = 0;
    bool pushed = false;                          // a push is out whose unpack has not been launched yet
    QdPeerHalo pend;                              // its slabs
};

bool qd_peer_on(const qd_ctx* c) { return c->peer && c->peer->on; }

// ------------------------------------------------------------------ device side
__device__ __forceinline__ bool qp_wait(const unsigned long long* p, unsigned long long expect, double* herr) {
    const unsigned long long t0 = wall_clock64();                       // s_memrealtime: 100 MHz
    const unsigned long long limit = (unsigned long long)(QP_TIMEOUT_S * 1.0e8);
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < expect) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > limit) { *(volatile double*)herr = 1.0; return false; }
    }
    return true;
}

// copy of H rows per slab and direction: f64 slabs as doubles, u8 slabs as bytes
__device__ __forceinline__ void qp_copy(char* dst, const char* src, size_t n_el, int u8, size_t tid, size_t nth) {
    if (u8) { for (size_t i = tid; i < n_el; i += nth) dst[i] = src[i]; }
    else { double* d = (double*)dst; const double* s = (const double*)src; for (size_t i = tid; i < n_el; i += nth) d[i] = s[i]; }
}

__global__ void __launch_bounds__(256)
k_halo_push(QdPeerHalo A, char* up_south, char* dn_north, size_t slab_stride, unsigned long long* up_cnt, unsigned long long* dn_cnt) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    const size_t n_el = (size_t)A.H * A.nlon;
    for (int k = 0; k < A.n; ++k) {
        const size_t esz = A.u8[k] ? 1 : sizeof(double);
        const char* base = (const char*)A.slab[k];
        qp_copy(up_south + k * slab_stride, base + (size_t)A.nown * A.nlon * esz, n_el, A.u8[k], tid, nth);    // my top rows -> up's south halo
        qp_copy(dn_north + k * slab_stride, base + (size_t)A.H * A.nlon * esz, n_el, A.u8[k], tid, nth);       // my bottom rows -> dn's north halo
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(up_cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_add(dn_cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void __launch_bounds__(256)
k_halo_unpack(QdPeerHalo A, const char* my_south, const char* my_north, size_t slab_stride, const unsigned long long* cnt,
              unsigned long long expect, double* herr) {
    if (threadIdx.x == 0) { qp_wait(cnt, expect, herr); qp_wait(cnt + 1, expect, herr); }
    __syncthreads();
    __threadfence_system();
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    const size_t n_el = (size_t)A.H * A.nlon;
    for (int k = 0; k < A.n; ++k) {
        const size_t esz = A.u8[k] ? 1 : sizeof(double);
        char* base = (char*)A.slab[k];
        qp_copy(base, my_south + k * slab_stride, n_el, A.u8[k], tid, nth);                                       // my south halo <- dn's top rows
        qp_copy(base + (size_t)(A.H + A.nown) * A.nlon * esz, my_north + k * slab_stride, n_el, A.u8[k], tid, nth);   // my north halo <- up's bottom rows
    }
}

// OP 0: f64 sum in rank order, 1: f64 max, 2: u32 sum (two per 8-byte unit), 3: gather (data[q][n8] <- slot q).
// phase bit 0: deposit my chunk in every mailbox; bit 1: wait for every rank's deposit in mine, reduce my chunk into `data`.
// Block b owns units [b * per, (b + 1) * per) of the vector in both phases, so the in-place result never races with a deposit.
template <int OP>
__global__ void __launch_bounds__(256)
k_peer_reduce(char* const* __restrict__ pbox, int world, int rank, size_t off_rv, size_t rv_stride, int parity, unsigned long long* data,
              int n8, int per, unsigned long long expect, int phase, double* herr) {
    const int i0 = blockIdx.x * per, i1 = min(n8, i0 + per);
    if (phase & 1) {
        const unsigned long long* src = OP == 3 ? data + (size_t)rank * n8 : data;
        for (int q = 0; q < world; ++q) {
            unsigned long long* dst = (unsigned long long*)(pbox[q] + off_rv + ((size_t)parity * world + rank) * rv_stride);
            for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) dst[i] = src[i];
        }
        __threadfence_system();
        __syncthreads();
        if ((int)threadIdx.x < world)
            __hip_atomic_fetch_add((unsigned long long*)(pbox[threadIdx.x] + QP_OFF_RCNT) + rank, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (phase & 2) {
        if ((int)threadIdx.x < world) qp_wait((const unsigned long long*)(pbox[rank] + QP_OFF_RCNT) + threadIdx.x, expect, herr);
        __syncthreads();
        __threadfence_system();
        const char* base = pbox[rank] + off_rv + (size_t)parity * world * rv_stride;
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            if (OP == 3) {
                for (int q = 0; q < world; ++q) data[(size_t)q * n8 + i] = ((const unsigned long long*)(base + q * rv_stride))[i];
            } else if (OP == 2) {
                unsigned long long a = ((const unsigned long long*)base)[i];
                unsigned int lo = (unsigned int)a, hi = (unsigned int)(a >> 32);
                for (int q = 1; q < world; ++q) {
                    const unsigned long long b = ((const unsigned long long*)(base + q * rv_stride))[i];
                    lo += (unsigned int)b; hi += (unsigned int)(b >> 32);
                }
                data[i] = (unsigned long long)lo | ((unsigned long long)hi << 32);
            } else {
                double a = ((const double*)base)[i];
                for (int q = 1; q < world; ++q) {
                    const double b = ((const double*)(base + q * rv_stride))[i];
                    a = OP == 1 ? (b > a ? b : a) : a + b;
                }
                ((double*)data)[i] = a;
            }
        }
    }
}

// ------------------------------------------------------------------ host side
static int qp_fail(qd_ctx* c, const char* what) { return qd_fail(c, what); }

static int qp_check(qd_ctx* c) {
    QdPeer* P = c->peer;
    if (P && P->herr && *(volatile double*)P->herr != 0.0)
        return qp_fail(c, "peer exchange: a rank did not arrive within the deadline (QP_TIMEOUT_S)");
    return 0;
}

static int qp_alloc(qd_ctx* c) {
    if (c->peer) return 0;
    if (c->geo.full) return qp_fail(c, "peer exchange: whole-globe handles have nothing to exchange");
    QdPeer* P = new QdPeer();
    const int world = c->desc.world;
    if (world < 1 || world > QD_RING_MAXRANKS) { delete P; return qp_fail(c, "peer exchange: world size out of range"); }
    P->world = world; P->rank = c->desc.rank;
    P->up = (P->rank + 1) % world; P->dn = (P->rank - 1 + world) % world;
    P->rv_stride = (size_t)QP_RV * 8;
    P->off_rv = QP_HDR;
    P->off_stage = P->off_rv + (size_t)2 * world * P->rv_stride;
    P->off_stage = (P->off_stage + 255) & ~(size_t)255;
    P->slab_stride = (((size_t)c->geo.halo * c->geo.nlon * sizeof(double)) + 255) & ~(size_t)255;
    P->dir_stride = P->slab_stride * QP_MAXSLABS;
    P->par_stride = P->dir_stride * 2;
    P->box_bytes = P->off_stage + 2 * P->par_stride;
    hipSetDevice(c->desc.device);
    // fine-grained device memory: remote stores write through, polls see them without a cache flush.  QD_PEER_COARSE=1 (developer
    // switch) takes ordinary device memory instead; the fences in the kernels keep that correct, just slower.
    const char* co = std::getenv("QD_PEER_COARSE");
    hipError_t e = (co && co[0] == '1') ? hipMalloc((void**)&P->box, P->box_bytes)
                                        : hipExtMallocWithFlags((void**)&P->box, P->box_bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { delete P; return qd_fail(c, "peer exchange: mailbox allocation", e); }
    if ((e = hipMemset(P->box, 0, P->box_bytes)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
        hipFree(P->box); delete P; return qd_fail(c, "peer exchange: mailbox clear", e);
    }
    if ((e = hipMalloc((void**)&P->d_pbox, sizeof(char*) * QD_RING_MAXRANKS)) != hipSuccess) {
        hipFree(P->box); delete P; return qd_fail(c, "peer exchange: table allocation", e);
    }
    P->pbox[P->rank] = P->box;
    P->herr = c->hpin + 60;
    *P->herr = 0.0;
    c->peer = P;
    return 0;
}

static int qp_publish_table(qd_ctx* c) {
    QdPeer* P = c->peer;
    for (int q = 0; q < P->world; ++q) if (!P->pbox[q]) return qp_fail(c, "peer exchange: a rank's mailbox is not mapped");
    hipSetDevice(c->desc.device);
    QD_HIP(c, hipMemcpy(P->d_pbox, P->pbox, sizeof(char*) * P->world, hipMemcpyHostToDevice));
    P->on = 1;
    return 0;
}

void qd_peer_release(qd_ctx* c) {
    QdPeer* P = c->peer;
    if (!P) return;
    c->peer = nullptr;
    hipSetDevice(c->desc.device);
    for (int q = 0; q < P->world; ++q) if (P->opened[q] && P->pbox[q]) hipIpcCloseMemHandle(P->pbox[q]);
    if (P->d_pbox) hipFree(P->d_pbox);
    if (P->box) hipFree(P->box);
    delete P;
}

int qd_peer_init_group(QdLocalGroup* g) {
    const int n = (int)g->peers.size();
    for (int k = 0; k < n; ++k) if (qp_alloc(g->peers[k])) return -1;
    for (int k = 0; k < n; ++k) {
        QdPeer* P = g->peers[k]->peer;
        P->local = n > 1 ? 1 : 0;
        for (int q = 0; q < n; ++q) P->pbox[q] = g->peers[q]->peer->box;
        if (qp_publish_table(g->peers[k])) return -1;
    }
    return 0;
}

// ---- halo exchange: push now, unpack when the caller says so (qd_peer_halo = both at once)
static int qp_halo_push(qd_ctx* c, const QdUse* slots, int n) {
    QdPeer* P = c->peer;
    if (P->pushed) return qp_fail(c, "peer exchange: a halo push is still waiting for its unpack");
    if (n < 1 || n > QP_MAXSLABS) return qp_fail(c, "peer exchange: bad slab count");
    QdPeerHalo& A = P->pend;
    A.n = n; A.H = c->geo.halo; A.nown = c->own_nrows; A.nlon = c->geo.nlon;
    for (int k = 0; k < n; ++k) { A.slab[k] = *slots[k].slot; A.u8[k] = slots[k].u8 ? 1 : 0; }
    const int par = (int)(P->hseq & 1ull);
    P->hseq += 1; P->hexp += QP_PUSH_BLOCKS; P->n_halo += 1;
    char* up_south = P->pbox[P->up] + P->off_stage + par * P->par_stride;                      // dir 0: "from the south"
    char* dn_north = P->pbox[P->dn] + P->off_stage + par * P->par_stride + P->dir_stride;      // dir 1: "from the north"
    hipLaunchKernelGGL(k_halo_push, dim3(QP_PUSH_BLOCKS), dim3(256), 0, c->stream, A, up_south, dn_north, P->slab_stride,
                       (unsigned long long*)(P->pbox[P->up] + QP_OFF_HCNT), (unsigned long long*)(P->pbox[P->dn] + QP_OFF_HCNT) + 1);
    P->pushed = true;
    return 0;
}

static int qp_halo_unpack(qd_ctx* c) {
    QdPeer* P = c->peer;
    if (!P->pushed) return 0;
    if (P->local) pthread_barrier_wait(&c->lgroup->bar);          // every push of the group is queued before any unpack polls
    const int par = (int)((P->hseq - 1) & 1ull);
    const char* my_south = P->box + P->off_stage + par * P->par_stride;
    const char* my_north = my_south + P->dir_stride;
    hipLaunchKernelGGL(k_halo_unpack, dim3(QP_PUSH_BLOCKS), dim3(256), 0, c->stream, P->pend, my_south, my_north, P->slab_stride,
                       (const unsigned long long*)(P->box + QP_OFF_HCNT), P->hexp, P->herr);
    P->pushed = false;
    return qp_check(c);
}

int qd_peer_halo(qd_ctx* c, const QdUse* slots, int n) {
    for (int k0 = 0; k0 < n; k0 += QP_MAXSLABS) {
        const int m = std::min(QP_MAXSLABS, n - k0);
        if (qp_halo_push(c, slots + k0, m) || qp_halo_unpack(c)) return -1;
    }
    return 0;
}
int qd_peer_halo_begin(qd_ctx* c, const QdUse* slots, int n) {      // n <= QP_MAXSLABS; qd_peer_halo_end() must follow
    return qp_halo_push(c, slots, n);
}
int qd_peer_halo_end(qd_ctx* c) { return qp_halo_unpack(c); }

// ---- reductions
template <int OP>
static void qp_launch_reduce(qd_ctx* c, unsigned long long* data, int n8, int nb, int per, int par, int phase) {
    QdPeer* P = c->peer;
    hipLaunchKernelGGL(k_peer_reduce<OP>, dim3(nb), dim3(256), 0, c->stream, (char* const*)P->d_pbox, P->world, P->rank, P->off_rv,
                       P->rv_stride, par, data, n8, per, P->rexp, phase, P->herr);
}

static int qp_reduce(qd_ctx* c, unsigned long long* data, int n8, int op) {
    QdPeer* P = c->peer;
    if (n8 < 1 || n8 > QP_RV) return qp_fail(c, "peer exchange: reduction longer than a mailbox slot");
    const int per = 512;                                          // 8-byte units per workgroup
    const int nb = (n8 + per - 1) / per;
    const int par = (int)(P->rseq & 1ull);
    P->rseq += 1; P->rexp += (unsigned long long)nb; P->n_reduce += 1;
    auto launch = [&](int phase) {
        switch (op) {
            case 0: qp_launch_reduce<0>(c, data, n8, nb, per, par, phase); break;
            case 1: qp_launch_reduce<1>(c, data, n8, nb, per, par, phase); break;
            case 2: qp_launch_reduce<2>(c, data, n8, nb, per, par, phase); break;
            default: qp_launch_reduce<3>(c, data, n8, nb, per, par, phase); break;
        }
    };
    if (P->local) { launch(1); pthread_barrier_wait(&c->lgroup->bar); launch(2); }
    else launch(3);
    return qp_check(c);
}

int qd_peer_allreduce(qd_ctx* c, void* dptr, int n, int kind) {
    if (kind == 2) {
        // u32 counters travel in pairs: callers' vectors are even-sized or padded (c->hist has 2 * QD_HIST_BINS words)
        return qp_reduce(c, (unsigned long long*)dptr, (n + 1) / 2, 2);
    }
    return qp_reduce(c, (unsigned long long*)dptr, n, kind ? 1 : 0);
}

int qd_peer_allgather(qd_ctx* c, double* buf, int n_per_rank) {
    return qp_reduce(c, (unsigned long long*)buf, n_per_rank, 3);
}

// ---- C-ABI: one process per GPU.  Every rank exports the IPC handle of its mailbox, the host side (qingdai_amd/bands.py) hands
// every rank the handles of all ranks, qd_peer_connect maps them.
extern "C" int qd_peer_export(qd_handle c, void* handle64, size_t bytes) {
    if (!c || !handle64 || bytes < sizeof(hipIpcMemHandle_t)) return -1;
    if (qp_alloc(c)) return -1;
    hipIpcMemHandle_t h;
    hipSetDevice(c->desc.device);
    QD_HIP(c, hipIpcGetMemHandle(&h, c->peer->box));
    std::memcpy(handle64, &h, sizeof(h));
    return 0;
}

extern "C" int qd_peer_connect(qd_handle c, const void* handles, size_t bytes_each, int world) {
    if (!c || !c->peer || (world > 1 && !handles) || bytes_each < sizeof(hipIpcMemHandle_t) || world != c->peer->world) return -1;
    QdPeer* P = c->peer;
    hipSetDevice(c->desc.device);
    for (int q = 0; q < world; ++q) {
        if (q == P->rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, (const char*)handles + (size_t)q * bytes_each, sizeof(h));
        void* p = nullptr;
        QD_HIP(c, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        P->pbox[q] = (char*)p; P->opened[q] = true;
    }
    return qp_publish_table(c);
}

extern "C" int qd_comm_peer_stats(qd_handle c, int* halo_exchanges, int* reductions) {
    if (!c || !halo_exchanges || !reductions) return -1;
    *halo_exchanges = c->peer ? (int)c->peer->n_halo : 0;
    *reductions = c->peer ? (int)c->peer->n_reduce : 0;
    return 0;
}
