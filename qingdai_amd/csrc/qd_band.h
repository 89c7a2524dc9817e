// qd_band.h -- structures shared by the transports of the latitude-band decomposition (qd_band.hip: RCCL, in-process group,
// host ring; qd_peer.hip: device-side exchange over the peer mapping).
#pragma once
#include "qd_internal.h"
#include <pthread.h>
#include <atomic>

#define QD_RING_MAXRANKS 64
#define QD_RING_MAXVALS 8
struct QdRingSeg {
    std::atomic<unsigned long long> seq[QD_RING_MAXRANKS];
    double vals[2][QD_RING_MAXRANKS][QD_RING_MAXVALS];
};
struct QdHostRing {
    QdRingSeg* seg = nullptr;
    int rank = 0, world = 1;
    unsigned long long my_seq = 0;
    bool mapped = false, owner = false;
    std::string name;
};

struct QdLocalGroup {
    std::vector<qd_ctx*> peers;
    pthread_barrier_t bar;
    std::vector<double> stage_d;        // [world][64]
    std::vector<unsigned int> stage_u;  // [world][4096]
    QdRingSeg ring;                     // the host ring of an in-process group lives in ordinary memory
};

// ---- qd_peer.hip: device-side exchange over the peer mapping (QD_PEER_EXCHANGE) -------------------------------------------
struct QdPeer;
bool qd_peer_on(const qd_ctx* c);
int  qd_peer_halo(qd_ctx* c, const QdUse* slots, int n);                 // the ring halo exchange of qd_exchange
int  qd_peer_allreduce(qd_ctx* c, void* dptr, int n, int kind);          // kind 0: f64 sum in rank order, 1: f64 max, 2: u32 sum
int  qd_peer_allreduce_publish(qd_ctx* c, double* dptr, int n, int op_max, double* hdst, double hseq);   // ... and straight on to pinned host memory
int  qd_peer_allgather(qd_ctx* c, double* buf, int n_per_rank);          // buf[world][n_per_rank], own segment filled in
// What a band launches right in front of / behind a small reduction, done by the reduction's own workgroup 0 instead (one launch of
// ~4.7 us each on a 1/8 band): the producer of the vector (pre) and the consumer of the result (post).
struct QdPeerHook {
    int pre = 0;               // 1: k_med_pack (header of my gathered median segment from the select state; OP gather)
                               // 2: k_precip_rawsums (data[0], data[1] = the two row sums of partial[0..n) and partial[n..2n))
                               // 3: k_max2_finish per row segment (data[2k], data[2k+1] = maxima of partial[k * pstride ..], k < nseg; the rest of data[0..6) = 0)
    const double* partial = nullptr; int n = 0, nseg = 0, nsegrows[3] = {0, 0, 0}; size_t pstride = 0;
    unsigned long long* st = nullptr; unsigned int* cc = nullptr;
    unsigned int* fixstat = nullptr;   // pre 3: data[6] = this band's average fix-list length since the last step, -2 = no news (qd_ocean.hip: the ocean tail's fix list)
    int post = 0;              // 1: k_precip_scalars_post on the reduced (num, den)
    double wsum = 0, pq_min = 0, p_blend = 0; int use_fb = 0; double* out = nullptr;
};
int  qd_peer_allreduce_hooked(qd_ctx* c, double* dptr, int n, int op_max, const QdPeerHook& H, double* hdst = nullptr, double hseq = 0.0);
int  qd_peer_allgather_hooked(qd_ctx* c, double* buf, int n_per_rank, const QdPeerHook& H);
bool qd_peer_hooks(const qd_ctx* c);                                     // the hooked forms are available (peer exchange on, QD_PEER_HOOKS != 0)
int  qd_peer_halo_begin(qd_ctx* c, const QdUse* slots, int n, bool defer = false);           // push only (n <= qd_peer_max_slabs()) ...
struct QdPeerPush;
struct QsOcnArgs;
void qd_launch_ocn_stream_pair(qd_ctx* c, const QsOcnArgs& A, const QdGeom& G2, int vb2, int nrs2, int ntc2);   // qd_stream_push.hip: two row segments, one launch
bool qd_launch_ocn_stream_push(qd_ctx* c, const QsOcnArgs& A);           // qd_stream_push.hip: the strips of A behind a waiting push (false: none waiting)
bool qd_peer_job_waiting(const qd_ctx* c);                               // a deferred push is waiting for a launch to carry it
bool qd_peer_take_job(qd_ctx* c, QdPeerPush* J);                         // a deferred push, for the launch that will carry it
int  qd_peer_halo_end(qd_ctx* c);                                        // ... wait + unpack
int  qd_peer_max_slabs();
int  qd_peer_overlap(const qd_ctx* c);                                   // QD_PEER_OVERLAP: 0 no split; consumers split into interior / boundary rows around an exchange: 1 push as its own kernel, 2 carried by the interior launch
struct QdPeerFold;
bool qd_peer_fold_begin(qd_ctx* c, QdPeerFold* F);                       // a one-double sum finished inside its producer's launch (qd_peer_dev.h)
int  qd_peer_init_group(QdLocalGroup* g);                                // in-process group: one mailbox per handle, pointers shared
void qd_peer_release(qd_ctx* c);
