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
