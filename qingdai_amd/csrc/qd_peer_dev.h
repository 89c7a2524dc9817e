// qd_peer_dev.h -- device-side pieces of the peer exchange (qd_peer.hip) that other kernels fold into their own launches.
#pragma once
#include "qd_internal.h"

#define QP_HDR 4096                     // mailbox header: arrival flags and counters
#define QP_OFF_HCNT 0                   // u64[2]: sequence number of the last push that arrived for my south halo (from dn) / my north halo (from up)
#define QP_OFF_RCNT 512                 // u64[world]: workgroups of deposits arrived from rank q
#define QP_TIMEOUT_S 20.0

// Release / acquire around the mailbox.  Every access to a mailbox is a system-scope atomic load or store (they go to the point of
// coherence whatever memory type the mapping has), so "release" only has to WAIT until this wave's stores have been acknowledged
// (s_waitcnt vmcnt(0)) and "acquire" only has to keep the compiler from moving loads above the poll.  The full system-scope fence
// also writes back every dirty line of the L2 -- and the kernel that ran just before an exchange has left the whole band there
// (measured: 28 us for a push of 2 x 5.9 MB with a fence per workgroup).  coarse != 0 (QD_PEER_COARSE=1: mailbox in ordinary device
// memory, developer switch) adds the full fences.
__device__ __forceinline__ void qp_release(int coarse) {
    if (coarse) __threadfence_system();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void qp_acquire(int coarse) {
    if (coarse) __threadfence_system();
    else asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned long long qp_ld(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// poll a word of this rank's mailbox until it reaches `expect`; a deadline instead of a hang (the error word is pinned host memory)
__device__ __forceinline__ bool qp_wait(const unsigned long long* p, unsigned long long expect, double* herr) {
    const unsigned long long t0 = wall_clock64();                       // s_memrealtime: 100 MHz
    const unsigned long long limit = (unsigned long long)(QP_TIMEOUT_S * 1.0e8);
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < expect) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > limit) { *(volatile double*)herr = 1.0; return false; }
    }
    return true;
}

// A one-double all-reduce (sum in rank order) folded into the LAST wave of the launch that produces the value: the eta mean of an
// ocean sub-step (pygcm/ocean.py:369-377) is finished by the tail kernel's last workgroup anyway (qd_wave.h: qd_acc_finish); with
// this block in its argument list that wave deposits the band's share in every mailbox, polls its own and returns the global sum --
// the same deposit / poll / rank-order sum as k_peer_reduce<0> with n = 1, bit for bit, without the launch (4.6 us each, 12-25 per step).
struct QdPeerFold {
    char* const* pbox = nullptr;        // device table of every rank's mailbox (nullptr: nothing to fold)
    unsigned long long expect = 0;      // deposits per source rank once this reduction is complete
    double* herr = nullptr;
    unsigned int off_rv = 0, rv_stride = 0;
    int world = 0, rank = 0, parity = 0, coarse = 0;
};
// all 64 lanes of ONE wave; v is wave-uniform
__device__ __forceinline__ double qp_fold_sum(const QdPeerFold& F, double v) {
    const int lane = threadIdx.x & 63;
    unsigned long long r = 0ull;
    if (lane < F.world) {
        char* pb = F.pbox[lane];
        unsigned long long* dst = (unsigned long long*)(pb + F.off_rv + ((size_t)F.parity * F.world + F.rank) * F.rv_stride);
        __hip_atomic_store(dst, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        qp_release(F.coarse);
        __hip_atomic_fetch_add((unsigned long long*)(pb + QP_OFF_RCNT) + F.rank, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const char* mine = F.pbox[F.rank];
        qp_wait((const unsigned long long*)(mine + QP_OFF_RCNT) + lane, F.expect, F.herr);
        qp_acquire(F.coarse);
        r = __hip_atomic_load((const unsigned long long*)(mine + F.off_rv + ((size_t)F.parity * F.world + lane) * F.rv_stride),
                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    double a = __longlong_as_double((long long)__shfl(r, 0, 64));
    for (int q = 1; q < F.world; ++q) a += __longlong_as_double((long long)__shfl(r, q, 64));
    return a;
}

// ---- a halo push as a JOB that another kernel's spare workgroups can carry (k_ocn_stream_push: the interior rows of the ocean momentum
// kernel and the push of the exchange it straddles in ONE launch -- a push kernel of its own has to finish, i.e. wait for the
// acknowledgements of everything it sent over the links, before the stream starts the interior launch: nothing would overlap)
#define QP_MAXSLABS 16
struct QdPeerHalo {
    void* slab[QP_MAXSLABS];
    unsigned char u8[QP_MAXSLABS];
    int n, H, nown, nlon;
};
struct QdPeerPush {
    QdPeerHalo A;
    char* up_south; char* dn_north; size_t slab_stride;
    unsigned long long* up_flag; unsigned long long* dn_flag; unsigned long long seq;
    unsigned int* tick;
    int coarse, nbx, nby;                                     // nbx parts x nby (= 2 n) segments = nbx * nby workgroups
};

// One (slab, direction) segment of an exchange is H rows; part `part` of `nparts` contiguous parts, four accesses in flight per lane.
// Every access to a MAILBOX is a system-scope atomic (relaxed) load or store of 8 bytes (single bytes for u8 slabs whose rows are
// not 8-byte aligned): such accesses go to the point of coherence whatever memory type the mapping has.  That matters between
// processes: a mailbox is allocated fine-grained, but the mapping hipIpcOpenMemHandle hands to ANOTHER process behaves like ordinary
// (coarse-grained, L2-cached) memory -- with plain stores and a wait for their acknowledgement, rank processes on one GPU read stale
// halo rows (the in-process groups, which share the owner's own pointer, did not); and a full release fence per workgroup writes back
// the whole L2, which the kernel before an exchange has just filled with the band (28 us for a push of 2 x 5.9 MB, 7 us like this).
template <typename T, bool PUT>
__device__ __forceinline__ void qp_copy_t(char* dst, const char* src, size_t bytes, int part, int nparts) {
    const size_t n = bytes / sizeof(T);
    const size_t per = (n + nparts - 1) / nparts;
    const size_t i0 = (size_t)part * per, i1 = i0 + per < n ? i0 + per : n;
    T* d = (T*)dst; const T* q = (const T*)src;
    auto ld = [&](size_t i) -> T { return PUT ? q[i] : __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    auto st = [&](size_t i, T v) { if (PUT) __hip_atomic_store(d + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); else d[i] = v; };
    size_t i = i0 + threadIdx.x;
    for (; i + 3 * blockDim.x < i1; i += 4 * blockDim.x) {
        const T a = ld(i), b = ld(i + blockDim.x), c = ld(i + 2 * blockDim.x), e = ld(i + 3 * blockDim.x);
        st(i, a); st(i + blockDim.x, b); st(i + 2 * blockDim.x, c); st(i + 3 * blockDim.x, e);
    }
    for (; i < i1; i += blockDim.x) st(i, ld(i));
}
// PUT: plain loads from a slab, atomic stores into a mailbox; !PUT: atomic loads from my mailbox, plain stores into a slab
template <bool PUT>
__device__ __forceinline__ void qp_copy(char* dst, const char* src, size_t bytes, int part, int nparts) {
    const unsigned long long al = (unsigned long long)dst | (unsigned long long)src | (unsigned long long)bytes;
    if ((al & 7ull) == 0) qp_copy_t<unsigned long long, PUT>(dst, src, bytes, part, nparts);
    else qp_copy_t<unsigned char, PUT>(dst, src, bytes, part, nparts);
}

// workgroup (bx, by) of a push: by = 2 k + direction.  `tick`: a word in this device's ordinary memory; the workgroup that takes the
// last ticket of the job knows that every other workgroup's stores are out (each waited for its acknowledgements before it took its
// ticket) and publishes the exchange's sequence number in both neighbours' mailboxes -- ONE remote store per direction.
__device__ __forceinline__ void qp_push_block(const QdPeerPush& J, int bx, int by) {
    const QdPeerHalo& A = J.A;
    const int k = by >> 1, dir = by & 1;
    const size_t esz = A.u8[k] ? 1 : sizeof(double);
    const size_t bytes = (size_t)A.H * A.nlon * esz;
    const char* base = (const char*)A.slab[k];
    if (dir == 0) qp_copy<true>(J.up_south + k * J.slab_stride, base + (size_t)A.nown * A.nlon * esz, bytes, bx, J.nbx);   // my top rows -> up's south halo
    else qp_copy<true>(J.dn_north + k * J.slab_stride, base + (size_t)A.H * A.nlon * esz, bytes, bx, J.nbx);               // my bottom rows -> dn's north halo
    qp_release(J.coarse);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int nb = (unsigned int)(J.nbx * J.nby);
        if (__hip_atomic_fetch_add(J.tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1u) {
            __hip_atomic_store(J.tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(J.up_flag, J.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(J.dn_flag, J.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
