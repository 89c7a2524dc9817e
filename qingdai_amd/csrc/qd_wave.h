// qd_wave.h -- wavefront-level building blocks shared by the fused kernels (qd_fused.hip, qd_stream.hip, qd_ocnstep.hip):
// scalar-cache loads of per-row tables, DPP lane shifts for the longitude neighbours, nan_to_num in 5 VALU ops,
// non-finite detection with one v_cmp_class.  gfx950 only.
#pragma once
#include "qd_internal.h"

// np.nan_to_num in 5 VALU ops instead of 12: clamp with max/min (which also map NaN to a bound), then
// send NaN to 0 with one compare + select
__device__ __forceinline__ double qd_nnf(double x) {
    const double c = fmin(fmax(x, -DBL_MAX), DBL_MAX);
    return (x == x) ? c : 0.0;
}

#define QD_CONST __attribute__((address_space(4)))
typedef const double __attribute__((address_space(4)))* qd_cptr;
// wave-uniform table read through the scalar cache (constant address space -> s_load into SGPRs)
__device__ __forceinline__ double qd_sload(const double* p, int idx) { return ((qd_cptr)(unsigned long long)p)[idx]; }
__device__ __forceinline__ bool qd_nonfinite(double x) { return __builtin_amdgcn_class(x, 0x207); }   // sNaN|qNaN|-inf|+inf

// value held by lane+1 / lane-1.  bound_ctrl:1 (the lane without a neighbour reads 0) lets the move stand alone: with
// bound_ctrl:0 the destination must first be initialised with the old value, one extra v_mov per DPP move.  Lanes 0 and
// 63 are halo columns whose results never reach an owned cell.
__device__ __forceinline__ double qd_east(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xf, 0xf, true);     // wave_shl:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double qd_west(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, true);     // wave_shr:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int qd_clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// buffer access: the range check of the resource drops a store (returns 0 for a load) whose lane offset is QD_BUF_OOB, so a
// masked store needs no exec branch (a branch around a store costs the row loops a full s_waitcnt vmcnt(0))
typedef unsigned int qd_u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t qd_rsrc;
#define QD_BUF_FLAGS 0x00020000                // raw buffer, 32-bit data format (word 3 of gfx90a / gfx94x / gfx950)
#define QD_BUF_OOB 0x80000000u
__device__ __forceinline__ qd_rsrc qd_buf(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, QD_BUF_FLAGS);
}
// row: ELEMENT offset of the row (wave-uniform -> SGPR offset); vo: the lane's byte offset inside the row
__device__ __forceinline__ double qd_buf_ld(qd_rsrc r, unsigned row, unsigned vo) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, row * 8u, 0));
}
__device__ __forceinline__ void qd_buf_st(qd_rsrc r, unsigned row, unsigned vo, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(qd_u32x2, v), r, vo, row * 8u, 0);
}

// XCD-contiguous dealing of a 1-D grid: workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an L2),
// so the linear id is remapped such that every XCD walks one contiguous chunk of work items (rows re-read by vertically
// adjacent strips then come from the same L2).  Pure performance: any placement gives the same result.
__device__ __forceinline__ unsigned qd_xcd_chunk(unsigned L, unsigned nb) {
    const unsigned per = nb >> 3, rem = nb & 7u, x = L & 7u;
    return x * per + (x < rem ? x : rem) + (L >> 3);
}

// ---- sum of a few thousand per-workgroup partial sums inside the launch that produces them ---------------------------------------
// Each workgroup adds its partial sum as a FIXED-POINT number (integer part + fraction scaled by 2^50, two int64) to one of 64 slots
// with agent-scope atomics; integer addition commutes, so the total does not depend on the order the workgroups finish in.  Who is
// last is decided by tickets that are spread the same way: a workgroup takes a ticket of its slot, the last one of a slot takes a
// ticket of the launch, and the last one of those (qd_acc_arrive returns true for exactly one workgroup) reads the 64 slots, converts
// once and clears everything for the next launch.  Resolution 2^-50 per partial (finer than the rounding of an f64 tree over the same
// values).  A partial that is not finite or not below 2^62 raises a flag instead; the finisher then adds the stored partials itself
// (coherent loads), so NaN / inf propagate as they would through the f64 sum.
// Measured (k_ocn_tail_stream, 2184 workgroups, 721 x 1440): the two adds on one slot 23.9 -> 71.6 us (atomics on one word
// serialise at ~90 per us -- that is also what a single ticket word costs); 64 slots of 16 B 27.1; of 32 B 24.2; of 64 B 24.1.
#define QD_ACC_SLOTS 64
#define QD_ACC_STRIDE 4                         // u64 per slot: integer part, fraction * 2^50, tickets, flag
#define QD_ACC_WORDS (QD_ACC_SLOTS * QD_ACC_STRIDE + 4)     // + the launch's ticket word
// lane 0 of one wave per workgroup calls this with the workgroup's partial sum; true = every other workgroup has arrived
__device__ __forceinline__ bool qd_acc_arrive(unsigned long long* acc, unsigned w, unsigned nwg, double p) {
    const unsigned k = w & (QD_ACC_SLOTS - 1);
    unsigned long long* a = acc + QD_ACC_STRIDE * k;
    if (fabs(p) < 0x1p62) {                     // false for NaN
        const double fl = floor(p);
        __hip_atomic_fetch_add(a, (unsigned long long)(long long)fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a + 1, (unsigned long long)(long long)((p - fl) * 0x1p50), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_fetch_or(a + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned in_slot = (nwg - 1u - k) / QD_ACC_SLOTS + 1u;            // workgroups w' < nwg with w' % 64 == k
    const unsigned slots = nwg < QD_ACC_SLOTS ? nwg : QD_ACC_SLOTS;
    // No fences: every word involved is only ever touched by agent-scope atomics, which execute at the point of coherence; the adds
    // above only have to be COMPLETE before the ticket is taken (s_waitcnt vmcnt(0): stores and atomics without return count in
    // vmcnt on gfx9), and the finisher's loads are issued after its tickets have returned.  (With acquire / release tickets every
    // workgroup writes back and invalidates its L2: 23.9 -> 92.6 us for the launch.)
    __builtin_amdgcn_s_waitcnt(0x0F70);                                     // vmcnt(0)
    if (__hip_atomic_fetch_add(a + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != in_slot - 1u) return false;
    return __hip_atomic_fetch_add(acc + QD_ACC_SLOTS * QD_ACC_STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == slots - 1u;
}
// the finishing wave (all 64 lanes): total / (wsum + 1e-15) -- wsum < 0: the raw total (latitude bands all-reduce it first) -- and
// everything cleared for the next launch
__device__ __forceinline__ double qd_acc_finish(unsigned long long* acc, const double* partial, int n, double wsum) {
    const int lane = threadIdx.x & 63;
    unsigned long long* s = acc + QD_ACC_STRIDE * lane;
    long long hi = (long long)__hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long lo = (long long)__hip_atomic_load(s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long fl = __hip_atomic_load(s + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < QD_ACC_STRIDE; ++k) __hip_atomic_store(s + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) __hip_atomic_store(acc + QD_ACC_SLOTS * QD_ACC_STRIDE, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { hi += __shfl_down(hi, o, 64); lo += __shfl_down(lo, o, 64); fl |= __shfl_down(fl, o, 64); }
    hi = __shfl(hi, 0, 64); lo = __shfl(lo, 0, 64); fl = __shfl(fl, 0, 64);
    if (fl) {                                   // some partial was not representable: the f64 sum of the stored partials (coherent loads)
        double a = 0.0;
        for (int k = lane; k < n; k += 64) a += __hip_atomic_load(partial + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        const double tot = __shfl(a, 0, 64);
        return wsum < 0.0 ? tot : tot / (wsum + 1e-15);
    }
    const double tot = (double)hi + (double)lo * 0x1p-50;
    return wsum < 0.0 ? tot : tot / (wsum + 1e-15);
}

