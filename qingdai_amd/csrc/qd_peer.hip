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
//                   south") and my H bottom rows into `dn`'s ("from the north"), then publishes the exchange's number there;
//                   k_halo_unpack waits until both of MY flags carry the exchange's number and copies the staged rows
//                   into my slabs' halo rows.  What the stream runs between the two launches overlaps the NEIGHBOURS' transfers;
//                   to overlap MY OWN too, the push must not be a kernel of its own (it ends when its stores are acknowledged,
//                   and the stream starts nothing before that): qd_band.hip books the push as a job and the consumer's INTERIOR
//                   rows carry it as the first workgroups of their own launch (qd_peer_dev.h qp_push_block, qd_stream_push.hip).
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
// Memory model: EVERY access to a mailbox is a system-scope atomic (relaxed) load or store of 8 bytes -- such accesses go to the point of
// coherence whatever memory type the mapping has (the mailbox is allocated fine-grained, but the mapping hipIpcOpenMemHandle hands to
// another process behaves like ordinary L2-cached memory: found the hard way, see qp_copy_t).  Producer: data stores, a wait for their
// acknowledgement (qd_peer_dev.h: qp_release -- not a fence: a release fence writes back the whole L2), workgroup barrier, ONE store
// or add on the consumer's flag / counter; consumer: polls the flag, then loads.  Every poll loop has a deadline (QP_TIMEOUT_S of
// s_memrealtime): a rank that never arrives turns into an error word in pinned host memory and a failed qd_* call, never into a hung
// GPU.  A freshly connected transport is self-tested before anybody relies on it (qd_peer_selftest).
//
// In-process groups (N band handles on one device, one host thread each: the test vehicle) run the same kernels in two launches
// per operation -- deposit, pthread barrier, collect: with all deposits queued before any collect, no kernel ever polls for work
// that sits BEHIND it in a shared hardware queue (HIP multiplexes streams onto a few of those).
#include "qd_internal.h"
#include "qd_band.h"
#include "qd_peer_dev.h"
#include <cstring>
#include <algorithm>

#define QP_RV 4104                      // 8-byte units per rank slot of a reduction (median segment: 4096 + 4)
#define QP_PART_BYTES 16384             // bytes of a halo segment one workgroup copies
#define QP_MAX_PARTS 32

struct QdPeer {
    QdPeerPush job;                      // a booked push nobody has launched yet (qd_peer_halo_begin(defer) .. qd_peer_take_job / unpack)
    bool job_waiting = false;
    int n_carried = 0;                   // pushes that went out inside a compute kernel's launch
    int on = 0, local = 0, world = 1, rank = 0, up = 0, dn = 0;
    char* box = nullptr;                          // my mailbox
    size_t box_bytes = 0, off_rv = 0, rv_stride = 0, off_stage = 0, slab_stride = 0, dir_stride = 0, par_stride = 0;
    char* pbox[QD_RING_MAXRANKS] = {nullptr};     // every rank's mailbox as THIS process maps it ([rank] == box)
    bool opened[QD_RING_MAXRANKS] = {false};      // mapped through hipIpcOpenMemHandle (to be closed)
    char** d_pbox = nullptr;                      // the same table on the device
    unsigned long long hseq = 0;                  // halo exchanges so far
    unsigned int* tick = nullptr;                 // ticket word of the push launches (ordinary device memory)
    unsigned long long rseq = 0, rexp = 0;        // reductions so far, deposits per source rank so far
    double* herr = nullptr;                       // pinned host word: a poll loop ran into its deadline
    long n_halo = 0, n_reduce = 0;
    int coarse = 0;                               // QD_PEER_COARSE=1: mailbox in ordinary device memory, full fences in the kernels
    int overlap = 0;                              // QD_PEER_OVERLAP: consumers of an exchange run their interior rows between push and unpack: 2 (default, world > 1) =
                                                  // the push rides in the interior launch itself, 1 = as a kernel of its own before it, 0 = no split
    int one_launch = 1;                           // QD_PEER_ONE_LAUNCH=0: an exchange that is not split around a compute kernel as k_halo_push + k_halo_unpack instead of k_halo_exchange
    int hooks = 1;                                // QD_PEER_HOOKS=0: the small kernels around a reduction (k_med_pack, k_precip_rawsums, k_precip_scalars_post, k_max2_finish) stay launches of their own
    int fold = 1;                                 // QD_PEER_FOLD=0: the eta sum of a sub-step as a k_peer_reduce launch of its own
    bool pushed = false;                          // a push is out whose unpack has not been launched yet
    QdPeerHalo pend;                              // its slabs
};

bool qd_peer_on(const qd_ctx* c) { return c->peer && c->peer->on; }
int qd_peer_overlap(const qd_ctx* c) { return c->peer && c->peer->on ? c->peer->overlap : 0; }
int qd_peer_max_slabs() { return QP_MAXSLABS; }

// ------------------------------------------------------------------ device side
__global__ void __launch_bounds__(256) k_halo_push(QdPeerPush J) { qp_push_block(J, (int)blockIdx.x, (int)blockIdx.y); }

__global__ void __launch_bounds__(256)
k_halo_unpack(QdPeerHalo A, const char* my_south, const char* my_north, size_t slab_stride, const unsigned long long* flag,
              unsigned long long expect, double* herr, int coarse) {
    const int k = blockIdx.y >> 1, dir = blockIdx.y & 1;
    if (threadIdx.x == 0) qp_wait(flag + dir, expect, herr);
    __syncthreads();
    qp_acquire(coarse);
    const size_t esz = A.u8[k] ? 1 : sizeof(double);
    const size_t bytes = (size_t)A.H * A.nlon * esz;
    char* base = (char*)A.slab[k];
    if (dir == 0) qp_copy<false>(base, my_south + k * slab_stride, bytes, blockIdx.x, gridDim.x);                                     // my south halo <- dn's top rows
    else qp_copy<false>(base + (size_t)(A.H + A.nown) * A.nlon * esz, my_north + k * slab_stride, bytes, blockIdx.x, gridDim.x);     // my north halo <- up's bottom rows
}

// push + wait + unpack in ONE launch (rank processes / GPUs; not the in-process groups, whose kernels share hardware queues): every
// workgroup pushes its part of its (slab, direction) segment, the last one to take a ticket publishes the exchange's number in both
// neighbours' mailboxes, and every workgroup then waits for the neighbour's number in MY mailbox and copies the same part of the
// staged rows into my halo.  All workgroups of the launch are resident at once (<= 32 x 32 of 256 threads), so the one that
// publishes is never queued behind the ones that poll; the neighbours' launches do not depend on mine.
__global__ void __launch_bounds__(256)
k_halo_exchange(QdPeerPush J, const char* my_south, const char* my_north, const unsigned long long* flag, double* herr) {
    qp_push_block(J, (int)blockIdx.x, (int)blockIdx.y);
    const QdPeerHalo& A = J.A;
    const int k = blockIdx.y >> 1, dir = blockIdx.y & 1;
    if (threadIdx.x == 0) qp_wait(flag + dir, J.seq, herr);
    __syncthreads();
    qp_acquire(J.coarse);
    const size_t esz = A.u8[k] ? 1 : sizeof(double);
    const size_t bytes = (size_t)A.H * A.nlon * esz;
    char* base = (char*)A.slab[k];
    if (dir == 0) qp_copy<false>(base, my_south + k * J.slab_stride, bytes, blockIdx.x, gridDim.x);
    else qp_copy<false>(base + (size_t)(A.H + A.nown) * A.nlon * esz, my_north + k * J.slab_stride, bytes, blockIdx.x, gridDim.x);
}

// OP 0: f64 sum in rank order, 1: f64 max, 2: u32 sum (two per 8-byte unit), 3: gather (data[q][n8] <- slot q).
// phase bit 0: deposit my chunk in every mailbox; bit 1: wait for every rank's deposit in mine, reduce my chunk into `data`.
// Block b owns units [b * per, (b + 1) * per) of the vector in both phases, so the in-place result never races with a deposit.
// the producer of a small vector, run by workgroup 0 in front of its deposit (QdPeerHook::pre; same arithmetic in the same order as
// the kernels it stands for: k_med_pack qd_reduce.hip, k_precip_rawsums qd_physics.hip, k_max2_finish qd_ocean.hip)
__device__ __forceinline__ void qp_hook_pre(const QdPeerHook& H, unsigned long long* data, int rank, int n8) {
    __shared__ double sm[2][4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (H.pre == 1) {
        if (t == 0) {
            double* seg = (double*)(data + (size_t)rank * n8);
            seg[0] = (double)H.st[0]; seg[1] = (double)H.st[1]; seg[2] = (double)H.cc[0]; seg[3] = 0.0;
            H.st[0] = 0ull; H.st[1] = 0ull; H.cc[0] = 0u;
        }
    } else if (H.pre == 2) {
        double a = 0.0, b = 0.0;
        for (int k = t; k < H.n; k += 256) { a += H.partial[k]; b += H.partial[H.n + k]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
        if (lane == 0) { sm[0][wv] = a; sm[1][wv] = b; }
        __syncthreads();
        if (t == 0) {
            for (int k = 1; k < 4; ++k) { a += sm[0][k]; b += sm[1][k]; }
            ((double*)data)[0] = a; ((double*)data)[1] = b;
        }
    } else if (H.pre == 3) {
        for (int s = 0; s < 3; ++s) {
            double a = 0.0, b = 0.0;
            if (s < H.nseg) {
                const double* p = H.partial + (size_t)s * H.pstride;
                const int n = H.nsegrows[s];
                for (int k = t; k < n; k += 256) { a = p[k] > a ? p[k] : a; b = p[n + k] > b ? p[n + k] : b; }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_down(a, o, 64), z = __shfl_down(b, o, 64); a = y > a ? y : a; b = z > b ? z : b; }
            }
            __syncthreads();                                      // sm is reused per segment
            if (lane == 0) { sm[0][wv] = a; sm[1][wv] = b; }
            __syncthreads();
            if (t == 0) {
                for (int k = 1; k < 4; ++k) { a = sm[0][k] > a ? sm[0][k] : a; b = sm[1][k] > b ? sm[1][k] : b; }
                ((double*)data)[2 * s] = a; ((double*)data)[2 * s + 1] = b;
            }
        }
        if (t == 0 && H.fixstat) {
            const unsigned e = H.fixstat[4], l = H.fixstat[5];
            ((double*)data)[6] = l ? (double)e / (double)l : -2.0;
            H.fixstat[4] = 0u; H.fixstat[5] = 0u;
        }
    }
    __syncthreads();                                              // workgroup 0 deposits what thread 0 has just written
}

template <int OP>
__global__ void __launch_bounds__(256)
k_peer_reduce(char* const* __restrict__ pbox, int world, int rank, size_t off_rv, size_t rv_stride, int parity, unsigned long long* data,
              int n8, int per, unsigned long long expect, int phase, double* herr, int coarse, double* hdst, double* hstamp, double hseq,
              QdPeerHook H) {
    const int i0 = blockIdx.x * per, i1 = min(n8, i0 + per);
    if ((phase & 1) && H.pre && blockIdx.x == 0) qp_hook_pre(H, data, rank, n8);
    if (phase & 1) {
        const unsigned long long* src = OP == 3 ? data + (size_t)rank * n8 : data;
        for (int q = 0; q < world; ++q) {
            unsigned long long* dst = (unsigned long long*)(pbox[q] + off_rv + ((size_t)parity * world + rank) * rv_stride);
            for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        qp_release(coarse);
        __syncthreads();
        if ((int)threadIdx.x < world)
            __hip_atomic_fetch_add((unsigned long long*)(pbox[threadIdx.x] + QP_OFF_RCNT) + rank, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (phase & 2) {
        if ((int)threadIdx.x < world) qp_wait((const unsigned long long*)(pbox[rank] + QP_OFF_RCNT) + threadIdx.x, expect, herr);
        __syncthreads();
        qp_acquire(coarse);
        const char* base = pbox[rank] + off_rv + (size_t)parity * world * rv_stride;
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            if (OP == 3) {
                for (int q = 0; q < world; ++q) data[(size_t)q * n8 + i] = qp_ld((const unsigned long long*)(base + q * rv_stride) + i);
            } else if (OP == 2) {
                unsigned long long a = qp_ld((const unsigned long long*)base + i);
                unsigned int lo = (unsigned int)a, hi = (unsigned int)(a >> 32);
                for (int q = 1; q < world; ++q) {
                    const unsigned long long b = qp_ld((const unsigned long long*)(base + q * rv_stride) + i);
                    lo += (unsigned int)b; hi += (unsigned int)(b >> 32);
                }
                data[i] = (unsigned long long)lo | ((unsigned long long)hi << 32);
            } else {
                double a = __longlong_as_double((long long)qp_ld((const unsigned long long*)base + i));
                for (int q = 1; q < world; ++q) {
                    const double b = __longlong_as_double((long long)qp_ld((const unsigned long long*)(base + q * rv_stride) + i));
                    a = OP == 1 ? (b > a ? b : a) : a + b;
                }
                ((double*)data)[i] = a;
                // the host is waiting for these few numbers (CFL maxima -> n_sub): they leave for pinned host memory from here
                if (hdst) __hip_atomic_store((unsigned long long*)hdst + i, (unsigned long long)__double_as_longlong(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (hdst) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_store((unsigned long long*)hstamp, (unsigned long long)__double_as_longlong(hseq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (H.post == 1 && blockIdx.x == 0) {                     // k_precip_scalars_post (qd_physics.hip) on the reduced (num, den)
            __syncthreads();
            if (threadIdx.x == 0) {
                const double a = ((const double*)data)[0], b = ((const double*)data)[1];
                const double den = b + 1e-20;
                H.out[0] = den > 0 ? a / den : 1.0;
                const double pq_mean = a / (H.wsum + 1e-15);
                H.out[1] = (H.use_fb && pq_mean < H.pq_min) ? H.p_blend : 0.0;
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
    P->coarse = (co && co[0] == '1') ? 1 : 0;
    hipError_t e = P->coarse ? hipMalloc((void**)&P->box, P->box_bytes)
                                        : hipExtMallocWithFlags((void**)&P->box, P->box_bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { delete P; return qd_fail(c, "peer exchange: mailbox allocation", e); }
    if ((e = hipMemset(P->box, 0, P->box_bytes)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
        hipFree(P->box); delete P; return qd_fail(c, "peer exchange: mailbox clear", e);
    }
    if ((e = hipMalloc((void**)&P->d_pbox, sizeof(char*) * QD_RING_MAXRANKS)) != hipSuccess) {
        hipFree(P->box); delete P; return qd_fail(c, "peer exchange: table allocation", e);
    }
    if ((e = hipMalloc((void**)&P->tick, 64)) != hipSuccess || (e = hipMemset(P->tick, 0, 64)) != hipSuccess) {
        hipFree(P->d_pbox); hipFree(P->box); delete P; return qd_fail(c, "peer exchange: ticket allocation", e);
    }
    { const char* ef = std::getenv("QD_PEER_FOLD"); if (ef && ef[0] == '0') P->fold = 0; }
    { const char* ef = std::getenv("QD_PEER_HOOKS"); if (ef && ef[0] == '0') P->hooks = 0; }
    { const char* ef = std::getenv("QD_PEER_ONE_LAUNCH"); if (ef && ef[0] == '0') P->one_launch = 0; }
    P->overlap = world > 1 ? 2 : 0;
    { const char* ef = std::getenv("QD_PEER_OVERLAP"); if (ef) P->overlap = ef[0] == '0' ? 0 : (ef[0] == '1' ? 1 : 2); }
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
    if (P->tick) hipFree(P->tick);
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
static dim3 qp_halo_grid(const qd_ctx* c, int n) {
    const size_t seg = (size_t)c->geo.halo * c->geo.nlon * sizeof(double);
    const int nbx = (int)std::min<size_t>(QP_MAX_PARTS, std::max<size_t>(1, seg / QP_PART_BYTES));
    return dim3(nbx, 2 * n);
}
// books the exchange (sequence number, buffer parity) and describes its push as a job; defer: the job waits in the handle for a kernel
// that carries it in its own launch (qd_peer_take_job) -- qp_halo_unpack launches whatever nobody took
static int qp_halo_push(qd_ctx* c, const QdUse* slots, int n, bool defer) {
    QdPeer* P = c->peer;
    if (P->pushed) return qp_fail(c, "peer exchange: a halo push is still waiting for its unpack");
    if (n < 1 || n > QP_MAXSLABS) return qp_fail(c, "peer exchange: bad slab count");
    QdPeerHalo& A = P->pend;
    A.n = n; A.H = c->geo.halo; A.nown = c->own_nrows; A.nlon = c->geo.nlon;
    for (int k = 0; k < n; ++k) { A.slab[k] = *slots[k].slot; A.u8[k] = slots[k].u8 ? 1 : 0; }
    const int par = (int)(P->hseq & 1ull);
    P->hseq += 1; P->n_halo += 1;
    const dim3 g = qp_halo_grid(c, n);
    QdPeerPush J;
    J.A = A;
    J.up_south = P->pbox[P->up] + P->off_stage + par * P->par_stride;                      // dir 0: "from the south"
    J.dn_north = P->pbox[P->dn] + P->off_stage + par * P->par_stride + P->dir_stride;      // dir 1: "from the north"
    J.slab_stride = P->slab_stride;
    J.up_flag = (unsigned long long*)(P->pbox[P->up] + QP_OFF_HCNT);
    J.dn_flag = (unsigned long long*)(P->pbox[P->dn] + QP_OFF_HCNT) + 1;
    J.seq = P->hseq; J.tick = P->tick; J.coarse = P->coarse; J.nbx = (int)g.x; J.nby = (int)g.y;
    if (defer) { P->job = J; P->job_waiting = true; }
    else hipLaunchKernelGGL(k_halo_push, g, dim3(256), 0, c->stream, J);
    P->pushed = true;
    return 0;
}

static int qp_halo_unpack(qd_ctx* c) {
    QdPeer* P = c->peer;
    if (!P->pushed) return 0;
    if (P->job_waiting) {
        hipLaunchKernelGGL(k_halo_push, dim3(P->job.nbx, P->job.nby), dim3(256), 0, c->stream, P->job);
        P->job_waiting = false;
    }
    if (P->local) pthread_barrier_wait(&c->lgroup->bar);          // every push of the group is queued before any unpack polls
    const int par = (int)((P->hseq - 1) & 1ull);
    const char* my_south = P->box + P->off_stage + par * P->par_stride;
    const char* my_north = my_south + P->dir_stride;
    hipLaunchKernelGGL(k_halo_unpack, qp_halo_grid(c, P->pend.n), dim3(256), 0, c->stream, P->pend, my_south, my_north, P->slab_stride,
                       (const unsigned long long*)(P->box + QP_OFF_HCNT), P->hseq, P->herr, P->coarse);
    P->pushed = false;
    return qp_check(c);
}

int qd_peer_halo(qd_ctx* c, const QdUse* slots, int n) {
    QdPeer* P = c->peer;
    for (int k0 = 0; k0 < n; k0 += QP_MAXSLABS) {
        const int m = std::min(QP_MAXSLABS, n - k0);
        if (!P->local && P->one_launch) {
            // book the exchange as a deferred push, then launch it together with its own unpack (k_halo_exchange)
            if (qp_halo_push(c, slots + k0, m, true)) return -1;
            const QdPeerPush& J = P->job;
            const int par = (int)((P->hseq - 1) & 1ull);
            const char* my_south = P->box + P->off_stage + par * P->par_stride;
            hipLaunchKernelGGL(k_halo_exchange, dim3(J.nbx, J.nby), dim3(256), 0, c->stream, J, my_south, my_south + P->dir_stride,
                               (const unsigned long long*)(P->box + QP_OFF_HCNT), P->herr);
            P->job_waiting = false; P->pushed = false;
            if (qp_check(c)) return -1;
            continue;
        }
        if (qp_halo_push(c, slots + k0, m, false) || qp_halo_unpack(c)) return -1;
    }
    return 0;
}
// n <= QP_MAXSLABS; qd_peer_halo_end() must follow.  defer: the push is not launched but left for a compute kernel to carry in its
// own launch (qd_peer_take_job; k_ocn_stream_push) -- a push kernel of its own ends only when everything it sent has been
// acknowledged over the links, and the stream starts nothing before that: the interior rows would not overlap the transfer at all
int qd_peer_halo_begin(qd_ctx* c, const QdUse* slots, int n, bool defer) { return qp_halo_push(c, slots, n, defer); }
bool qd_peer_job_waiting(const qd_ctx* c) { return c->peer && c->peer->job_waiting; }
bool qd_peer_take_job(qd_ctx* c, QdPeerPush* J) {
    QdPeer* P = c->peer;
    if (!P || !P->job_waiting) return false;
    *J = P->job; P->job_waiting = false; P->n_carried += 1;
    return true;
}
int qd_peer_halo_end(qd_ctx* c) { return qp_halo_unpack(c); }

// ---- reductions
template <int OP>
static void qp_launch_reduce(qd_ctx* c, unsigned long long* data, int n8, int nb, int per, int par, int phase, double* hdst, double hseq,
                             const QdPeerHook& H) {
    QdPeer* P = c->peer;
    hipLaunchKernelGGL(k_peer_reduce<OP>, dim3(nb), dim3(256), 0, c->stream, (char* const*)P->d_pbox, P->world, P->rank, P->off_rv,
                       P->rv_stride, par, data, n8, per, P->rexp, phase, P->herr, P->coarse, hdst, c->hpin + 61, hseq, H);
}

static int qp_reduce(qd_ctx* c, unsigned long long* data, int n8, int op, double* hdst = nullptr, double hseq = 0.0,
                     const QdPeerHook& H = QdPeerHook()) {
    QdPeer* P = c->peer;
    if (n8 < 1 || n8 > QP_RV) return qp_fail(c, "peer exchange: reduction longer than a mailbox slot");
    const int per = 512;                                          // 8-byte units per workgroup
    const int nb = (n8 + per - 1) / per;
    const int par = (int)(P->rseq & 1ull);
    P->rseq += 1; P->rexp += (unsigned long long)nb; P->n_reduce += 1;
    auto launch = [&](int phase) {
        switch (op) {
            case 0: qp_launch_reduce<0>(c, data, n8, nb, per, par, phase, (phase & 2) ? hdst : nullptr, hseq, H); break;
            case 1: qp_launch_reduce<1>(c, data, n8, nb, per, par, phase, (phase & 2) ? hdst : nullptr, hseq, H); break;
            case 2: qp_launch_reduce<2>(c, data, n8, nb, per, par, phase, nullptr, 0.0, H); break;
            default: qp_launch_reduce<3>(c, data, n8, nb, per, par, phase, nullptr, 0.0, H); break;
        }
    };
    if (P->local) { launch(1); pthread_barrier_wait(&c->lgroup->bar); launch(2); }
    else launch(3);
    return qp_check(c);
}

// A one-double sum whose producer can finish it inside its own launch (qd_peer_dev.h: qp_fold_sum): hands out the argument block
// and books the reduction; false (and an empty block) when the caller has to issue qd_allreduce_* itself -- in-process groups keep
// the two-launch form (a wave that polls inside a compute kernel could wait for a deposit queued BEHIND it in a shared hardware queue).
bool qd_peer_fold_begin(qd_ctx* c, QdPeerFold* F) {
    *F = QdPeerFold();
    QdPeer* P = c->peer;
    if (!P || !P->on || P->local || !P->fold) return false;
    F->pbox = (char* const*)P->d_pbox; F->herr = P->herr;
    F->off_rv = (unsigned int)P->off_rv; F->rv_stride = (unsigned int)P->rv_stride;
    F->world = P->world; F->rank = P->rank; F->parity = (int)(P->rseq & 1ull); F->coarse = P->coarse;
    P->rseq += 1; P->rexp += 1ull; P->n_reduce += 1;
    F->expect = P->rexp;
    c->allreduces++;
    return true;
}

int qd_peer_allreduce(qd_ctx* c, void* dptr, int n, int kind) {
    if (kind == 2) {
        // u32 counters travel in pairs: callers' vectors are even-sized or padded (c->hist has 2 * QD_HIST_BINS words)
        return qp_reduce(c, (unsigned long long*)dptr, (n + 1) / 2, 2);
    }
    return qp_reduce(c, (unsigned long long*)dptr, n, kind ? 1 : 0);
}

// all-reduce of n <= 256 doubles whose result the HOST needs at once: the collecting workgroup also stores it into pinned host
// memory (hdst) and stamps c->hpin[61] with hseq (the caller polls the stamp: qd_wait_host_flag)
int qd_peer_allreduce_publish(qd_ctx* c, double* dptr, int n, int op_max, double* hdst, double hseq) {
    if (n > 256) return qp_fail(c, "peer exchange: published reduction longer than 256 values");
    return qp_reduce(c, (unsigned long long*)dptr, n, op_max ? 1 : 0, hdst, hseq);
}

int qd_peer_allgather(qd_ctx* c, double* buf, int n_per_rank) {
    return qp_reduce(c, (unsigned long long*)buf, n_per_rank, 3);
}
bool qd_peer_hooks(const qd_ctx* c) { return c->peer && c->peer->on && c->peer->hooks; }
int qd_peer_allreduce_hooked(qd_ctx* c, double* dptr, int n, int op_max, const QdPeerHook& H, double* hdst, double hseq) {
    if (n > 256) return qp_fail(c, "peer exchange: hooked reduction longer than 256 values");
    return qp_reduce(c, (unsigned long long*)dptr, n, op_max ? 1 : 0, hdst, hseq, H);
}
int qd_peer_allgather_hooked(qd_ctx* c, double* buf, int n_per_rank, const QdPeerHook& H) {
    return qp_reduce(c, (unsigned long long*)buf, n_per_rank, 3, nullptr, 0.0, H);
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

// ---- self-test of a freshly connected transport (qingdai_amd.bands.init_peer runs it before anybody relies on the mailboxes)
// A memory-model surprise on a mapping this code has never run over (xGMI between real GPUs) would show up as STALE data, not as an
// error: so every rank exchanges rows whose values encode (sender, iteration, position) and checks what arrived, and all-reduces
// rank-dependent numbers whose sums it knows, `iters` times (both buffer parities, back to back).  Returns the number of wrong values
// seen by THIS rank (0 = clean), -1 on a launch / deadline failure.
__global__ void k_peer_test_fill(double* slab, int H, int nown, int nlon, int rank, int it) {
    const size_t n = (size_t)H * nlon;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        slab[(size_t)nown * nlon + i] = 1.0e6 * (rank + 1) + 1.0e3 * it + (double)(i % 997) + 0.25;        // my top rows -> up's south halo
        slab[(size_t)H * nlon + i] = -1.0e6 * (rank + 1) - 1.0e3 * it - (double)(i % 991) - 0.5;           // my bottom rows -> dn's north halo
    }
}
__global__ void k_peer_test_check(const double* slab, int H, int nown, int nlon, int up, int dn, int it, unsigned long long* bad) {
    const size_t n = (size_t)H * nlon;
    unsigned long long b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        b += slab[i] != 1.0e6 * (dn + 1) + 1.0e3 * it + (double)(i % 997) + 0.25;                            // south halo <- dn's top rows
        b += slab[(size_t)(H + nown) * nlon + i] != -1.0e6 * (up + 1) - 1.0e3 * it - (double)(i % 991) - 0.5; // north halo <- up's bottom rows
    }
    if (b) atomicAdd(bad, b);
}
extern "C" int qd_peer_selftest(qd_handle c, int iters, long long* wrong) {
    if (!c || !wrong || !qd_peer_on(c) || iters < 1) return -1;
    QdPeer* P = c->peer;
    hipSetDevice(c->desc.device);
    const int H = c->geo.halo, nown = c->own_nrows, nlon = c->geo.nlon;
    double*& slab = c->scratch[QD_NSCRATCH - 3];              // a scratch slab nobody holds between calls
    unsigned long long* bad = c->dcount + 24;
    QD_HIP(c, hipMemsetAsync(bad, 0, sizeof(unsigned long long), c->stream));
    long long wrong_sums = 0;
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(k_peer_test_fill, dim3(64), dim3(256), 0, c->stream, slab, H, nown, nlon, P->rank, it);
        QdUse u{(void**)&slab, 0, 0};
        if (qd_peer_halo(c, &u, 1)) return -1;
        hipLaunchKernelGGL(k_peer_test_check, dim3(64), dim3(256), 0, c->stream, slab, H, nown, nlon, P->up, P->dn, it, bad);
        // sums every rank can predict: sum_r (r + 1) * (it + 1), max_r (r * 7 - it), and a vector long enough for several workgroups
        double h[4] = {(double)(P->rank + 1) * (it + 1), 0.0, 0.0, 0.0};
        double* d = c->dscal + QD_S_TMP0;
        QD_HIP(c, hipMemcpyAsync(d, h, sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (qd_peer_allreduce(c, d, 1, 0)) return -1;
        QD_HIP(c, hipMemcpyAsync(h + 1, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        h[2] = (double)(P->rank * 7 - it);
        QD_HIP(c, hipMemcpyAsync(d + 1, h + 2, sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (qd_peer_allreduce(c, d + 1, 1, 1)) return -1;
        QD_HIP(c, hipMemcpyAsync(h + 3, d + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        QD_HIP(c, hipStreamSynchronize(c->stream));
        if (qp_check(c)) return -1;
        const double want_sum = 0.5 * P->world * (P->world + 1) * (it + 1), want_max = (double)((P->world - 1) * 7 - it);
        wrong_sums += (h[1] != want_sum) + (h[3] != want_max);
    }
    unsigned long long hb = 0;
    QD_HIP(c, hipMemcpy(&hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
    *wrong = (long long)hb + wrong_sums;
    c->vm[slab] = 0;                                          // its halo rows hold test patterns
    return 0;
}
// back to the other transports (the self-test failed somewhere: every rank leaves the mailboxes together)
extern "C" int qd_peer_disable(qd_handle c) { if (!c) return -1; if (c->peer) c->peer->on = 0; return 0; }

extern "C" int qd_comm_peer_carried(qd_handle c) { return !c ? -1 : (c->peer ? c->peer->n_carried : 0); }
extern "C" int qd_comm_peer_stats(qd_handle c, int* halo_exchanges, int* reductions) {
    if (!c || !halo_exchanges || !reductions) return -1;
    *halo_exchanges = c->peer ? (int)c->peer->n_halo : 0;
    *reductions = c->peer ? (int)c->peer->n_reduce : 0;
    return 0;
}
