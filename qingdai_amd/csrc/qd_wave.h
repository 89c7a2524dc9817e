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

// area-weighted mean of eta from the tile sums of k_ocn_tail: every wave that needs the mean adds the (~1000) partial sums itself,
// in one fixed order (lane-strided, then a shuffle tree), so all waves of all consumers get the same bits -- cheaper than a
// launch of its own between two sub-steps for one scalar
__device__ __forceinline__ double qd_partial_mean(const double* __restrict__ p, int n, double wsum) {
    const int lane = threadIdx.x & 63;
    double a = 0.0;
    for (int k = lane; k < n; k += 64) a += p[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    a = __shfl(a, 0, 64);
    return a / (wsum + 1e-15);
}
