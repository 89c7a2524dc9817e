// qd_reduce.hip -- deterministic global reductions and exact median-of-positives (gfx950).
//
//   O13: np.median(x[x>0])  dynamics.py:344-348, physics.py:298-301, run_simulation.py:1866-1874
//        area-weighted sums  physics.py:320-323,345-347; ocean.py:372-375; energy.py:520-525
//        global max          ocean.py:298-299
//
// Sums use a fixed two-level tree (wave shuffles -> LDS -> one finishing workgroup) so the
// result is bit-reproducible run to run; it is NOT numpy's pairwise order (stated tolerance
// in tests).  The median is exact: an MSB-first radix select over the IEEE-754 bit patterns
// of the positive entries (order-isomorphic to their values), 11 bits per pass, tracking the
// two middle ranks at once for even counts.
#include "qd_internal.h"
#include <algorithm>
#include <cstdlib>

__device__ __forceinline__ double qd_wave_sum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}
__device__ __forceinline__ double qd_wave_max(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double y = __shfl_down(x, o, 64); x = (y > x) ? y : x; }
    return x;
}

// block-level reduce of (sum | max); result valid in thread 0
template <int OP>
__device__ __forceinline__ double qd_block_reduce(double x) {
    __shared__ double sm[QD_BLOCK / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    x = (OP == 0) ? qd_wave_sum(x) : qd_wave_max(x);
    __syncthreads();
    if (lane == 0) sm[w] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sm[0];
        for (int k = 1; k < QD_BLOCK / 64; ++k) r = (OP == 0) ? r + sm[k] : (sm[k] > r ? sm[k] : r);
        x = r;
    }
    return x;
}

// op: 0 sum, 1 cos-weighted sum (x * warea[row]), 2 max, 3 min (as max of -x), 4 max|x|
__global__ void __launch_bounds__(QD_BLOCK)
k_reduce_stage1(QdGeom G, const double* __restrict__ x, const double* __restrict__ warea, int op,
                double* __restrict__ partial) {
    const int i = G.row0 + blockIdx.y;
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    double acc = (op >= 2) ? -DBL_MAX : 0.0;
    for (int j = blockIdx.x * QD_BLOCK + threadIdx.x; j < G.nlon; j += gridDim.x * QD_BLOCK) {
        const double v = x[b + j];
        if (op == 0) acc += v;
        else if (op == 1) acc += v * warea[i];
        else if (op == 2) acc = v > acc ? v : acc;
        else if (op == 3) acc = -v > acc ? -v : acc;
        else { const double av = fabs(v); acc = av > acc ? av : acc; }
    }
    double r = (op >= 2) ? qd_block_reduce<1>(acc) : qd_block_reduce<0>(acc);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = r;
}

__global__ void __launch_bounds__(QD_BLOCK)
k_reduce_stage2(const double* __restrict__ partial, int n, int ismax, double* __restrict__ out) {
    double acc = ismax ? -DBL_MAX : 0.0;
    for (int k = threadIdx.x; k < n; k += QD_BLOCK) {
        const double v = partial[k];
        acc = ismax ? (v > acc ? v : acc) : acc + v;
    }
    double r = ismax ? qd_block_reduce<1>(acc) : qd_block_reduce<0>(acc);
    if (threadIdx.x == 0) *out = r;
}

// reduce a resident field into device scalar slot; optionally copy to host (synchronises)
int qd_reduce_field(qd_ctx* c, const double* x, int op, double* host_out) {
    const QdGeom G = qd_segments(c, 0).g[0];                 // owned rows only: a sum must not count halo rows
    dim3 grid(1, G.nrows);
    hipLaunchKernelGGL(k_reduce_stage1, grid, dim3(QD_BLOCK), 0, c->stream, G, x, c->tabs.warea, op, c->red_partial);
    hipLaunchKernelGGL(k_reduce_stage2, dim3(1), dim3(QD_BLOCK), 0, c->stream, c->red_partial, G.nrows,
                       op >= 2 ? 1 : 0, c->dscal + QD_S_TMP0);
    if (qd_allreduce_f64(c, c->dscal + QD_S_TMP0, 1, op >= 2 ? 1 : 0)) return -1;
    if (host_out) {
        QD_HIP(c, hipMemcpyAsync(c->hpin, c->dscal + QD_S_TMP0, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        QD_HIP(c, hipStreamSynchronize(c->stream));
        double v = c->hpin[0];
        if (op == 3) v = -v;
        *host_out = v;
    }
    return 0;
}

// ------------------------------------------------------------------ exact median of positives
// sel_state layout (unsigned long long): [0] count of positives, [1] prefix_lo, [2] rank_lo,
// [3] prefix_hi, [4] rank_hi, [5] bits resolved so far.
//
// `transform` lets the caller take the median of a derived field without materialising it:
//   0: x itself
//   1: max(0, -(x - tparam))       physics.py:296 (pos = max(0, -(div - D_crit)))
__device__ __forceinline__ double qd_med_value(double x, int transform, double tparam) {
    if (transform == 1) return qd_max(0.0, -(x - tparam));
    return x;
}

// One pass = histogram of the next digit over the still-matching positive entries + (last workgroup
// to finish, found with a ticket) the scan that extends the two prefixes.  Histogram counts reach
// global memory through device-scope atomics; the last workgroup reads them back with agent-scope
// atomic loads behind a fence, so no stale L1/L2 line can be observed (MI355X: per-XCD L2s are not
// coherent for plain loads).  State and histogram are left zeroed for the next pass / next call.
// sel_state: [0] count of positives, [1] prefix_lo, [2] rank_lo, [3] prefix_hi, [4] rank_hi, [6] ticket
__global__ void __launch_bounds__(QD_BLOCK)
k_sel_pass(QdGeom G, const double* __restrict__ x, int transform, double tparam, unsigned long long* st,
           unsigned int* hist, int shift, int width, int first, int mode) {
    // mode 0: histogram + scan by the last workgroup (single GPU)
    // mode 1: histogram only   mode 2: scan only (one workgroup) -- latitude bands all-reduce the
    //         histogram between the two
    __shared__ unsigned int sh[2 * QD_HIST_BINS];
    __shared__ unsigned long long s_st[8];
    __shared__ int s_last;
    const int t = threadIdx.x;
    const unsigned long long n0 = __hip_atomic_load(&st[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!first && n0 == 0ull) return;                        // no positive entry: nothing to refine
    if (mode != 2) {
    for (int k = t; k < 2 * QD_HIST_BINS; k += QD_BLOCK) sh[k] = 0u;
    const unsigned long long plo = __hip_atomic_load(&st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long phi = __hip_atomic_load(&st[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int up = shift + width;                            // bits above the current digit
    const int jstep = gridDim.x * QD_BLOCK;
    // few, fat workgroups: each walks several rows so its LDS histogram is dense (fewer global atomics,
    // fewer tickets on the single counter word)
    for (int i = G.row0 + (int)blockIdx.y; i < G.row0 + G.nrows; i += (int)gridDim.y) {
    const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
    for (int jb = blockIdx.x * QD_BLOCK; jb < G.nlon; jb += 8 * jstep) {
    // issue up to 8 independent loads per thread before touching the (wave-synchronising) histogram code:
    // one exposed memory latency per batch instead of one per element
    double vbuf[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = jb + q * jstep + t;
        vbuf[q] = x[b + (j < G.nlon ? j : G.nlon - 1)];     // unconditional load from a clamped index (no branch)
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {                                            // wave-uniform trip count (ballots inside)
        const int j0 = jb + q * jstep;
        if (j0 >= G.nlon) break;
        const int j = j0 + t;
        const double v = (j < G.nlon) ? qd_med_value(vbuf[q], transform, tparam) : 0.0;
        const bool pos = (v > 0.0);
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
        const unsigned long long hi_bits = up >= 64 ? 0ull : (bits >> up);
        const unsigned int digit = (unsigned int)((bits >> shift) & ((1u << width) - 1u));
        const bool in_lo = pos && (first || hi_bits == (plo >> up));
        const bool in_hi = pos && !first && plo != phi && hi_bits == (phi >> up);
        // (wave-level pre-aggregation of equal digits was measured slower than plain LDS atomics here: 41 vs 25 us)
        const bool todo = in_lo;
        if (todo) atomicAdd(&sh[digit], 1u);
        if (in_hi) atomicAdd(&sh[QD_HIST_BINS + digit], 1u);
    }
    }
    }
    __syncthreads();
    // fire-and-forget device-scope atomics, all in flight at once; vmcnt(0) drains them (a write leaves the
    // counter when it has reached L2), so the ticket below is ordered after this workgroup's counts without
    // a cache flush and without paying one round trip per bin
    for (int k = t; k < 2 * QD_HIST_BINS; k += QD_BLOCK) if (sh[k]) atomicAdd(&hist[k], sh[k]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (mode == 1) return;
    __syncthreads();
    if (t == 0) {
        const unsigned long long ticket = atomicAdd(&st[6], 1ull);
        s_last = (ticket == (unsigned long long)(gridDim.x * gridDim.y) - 1ull) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    }   // mode != 2
    // ---- last workgroup: scan
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int per = QD_HIST_BINS / QD_BLOCK;
    for (int k = t; k < 2 * QD_HIST_BINS; k += QD_BLOCK) sh[k] = __hip_atomic_load(&hist[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < 8) s_st[t] = __hip_atomic_load(&st[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // chunk sums (8 bins per thread) and their exclusive prefix over the workgroup, both histograms at once:
    // wave-level shuffle scan + one LDS hop across the 4 waves -- no serial walk over 256 chunks
    __shared__ unsigned int wtot[2][QD_BLOCK / 64];
    unsigned int mysum[2], excl[2];
    const int lane = t & 63, wv = t >> 6;
    for (int hsel = 0; hsel < 2; ++hsel) {
        unsigned int sacc = 0;
        for (int k = 0; k < per; ++k) sacc += sh[hsel * QD_HIST_BINS + t * per + k];
        mysum[hsel] = sacc;
        unsigned int inc = sacc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
        if (lane == 63) wtot[hsel][wv] = inc;
        excl[hsel] = inc - sacc;
    }
    __syncthreads();
    unsigned long long total0 = 0;
    for (int hsel = 0; hsel < 2; ++hsel) {
        unsigned int base = 0;
        for (int k = 0; k < wv; ++k) base += wtot[hsel][k];
        excl[hsel] += base;
    }
    for (int k = 0; k < QD_BLOCK / 64; ++k) total0 += wtot[0][k];
    if (t == 0 && first) {
        const unsigned long long n = total0;
        s_st[0] = n;
        s_st[2] = n ? (n - 1) / 2 : 0;   // lower middle rank (0-based)
        s_st[4] = n / 2;                 // upper middle rank
        s_st[1] = 0; s_st[3] = 0;
    }
    __syncthreads();
    const bool same = (s_st[1] == s_st[3]);
    if (s_st[0] > 0) {
        // the thread whose chunk [excl, excl + mysum) holds the target rank walks its 8 bins
        for (int which = 0; which < 2; ++which) {
            const int hsel = (which == 1 && !same) ? 1 : 0;
            const unsigned long long r = s_st[which == 0 ? 2 : 4];
            if (mysum[hsel] > 0 && r >= excl[hsel] && r < (unsigned long long)excl[hsel] + mysum[hsel]) {
                unsigned long long cum = excl[hsel];
                int d = t * per;
                for (; d < t * per + per; ++d) { const unsigned int hv = sh[hsel * QD_HIST_BINS + d]; if (cum + hv > r) break; cum += hv; }
                if (d >= t * per + per) d = t * per + per - 1;
                __hip_atomic_store(&st[which == 0 ? 2 : 4], r - cum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&st[which == 0 ? 1 : 3], s_st[which == 0 ? 1 : 3] | ((unsigned long long)d << shift),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (t == 0) {
        if (first) __hip_atomic_store(&st[0], s_st[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st[6], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int k = t; k < 2 * QD_HIST_BINS; k += QD_BLOCK) __hip_atomic_store(&hist[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// After two histogram passes (22 bits: sign, exponent, 10 mantissa bits) the bucket of each middle rank holds
// ~N/1000 entries of a continuous field.  Instead of four more passes over the whole field, ONE pass copies
// the entries of the two buckets to a candidate list (wave-aggregated append), and one workgroup finishes the
// radix select on the list.  The result does not depend on the order of the list.  Capacity = the whole field
// (a constant field puts every entry in one bucket: slow, still exact).
__global__ void __launch_bounds__(QD_BLOCK)
k_sel_collect(QdGeom G, const double* __restrict__ x, int transform, double tparam, const unsigned long long* __restrict__ st,
              double* __restrict__ cand, unsigned int* __restrict__ ccount, unsigned long long cap, int bits_done) {
    const unsigned long long n0 = __hip_atomic_load(&st[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n0 == 0ull) return;
    const int up = 64 - bits_done;
    const unsigned long long plo = __hip_atomic_load(&st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> up;
    const unsigned long long phi = __hip_atomic_load(&st[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> up;
    const int t = threadIdx.x, lane = t & 63;
    const int jstep = gridDim.x * QD_BLOCK;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int i = G.row0 + (int)blockIdx.y; i < G.row0 + G.nrows; i += (int)gridDim.y) {
        const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
        for (int jb = blockIdx.x * QD_BLOCK; jb < G.nlon; jb += 8 * jstep) {
            double vbuf[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = jb + q * jstep + t;
                vbuf[q] = x[b + (j < G.nlon ? j : G.nlon - 1)];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j0 = jb + q * jstep;
                if (j0 >= G.nlon) break;                                  // wave-uniform
                const int j = j0 + t;
                const double v = (j < G.nlon) ? qd_med_value(vbuf[q], transform, tparam) : 0.0;
                const unsigned long long hb = (unsigned long long)__double_as_longlong(v) >> up;
                const bool pos = v > 0.0;
                const bool in_lo = pos && hb == plo, in_hi = pos && plo != phi && hb == phi;
                const unsigned long long mlo = __ballot(in_lo), mhi = __ballot(in_hi);
                if (mlo) {
                    unsigned int base = 0;
                    const int leader = __ffsll((long long)mlo) - 1;
                    if (lane == leader) base = atomicAdd(&ccount[0], (unsigned int)__popcll(mlo));
                    base = (unsigned int)__shfl((int)base, leader, 64);
                    if (in_lo) cand[base + (unsigned int)__popcll(mlo & lt)] = v;
                }
                if (mhi) {
                    unsigned int base = 0;
                    const int leader = __ffsll((long long)mhi) - 1;
                    if (lane == leader) base = atomicAdd(&ccount[1], (unsigned int)__popcll(mhi));
                    base = (unsigned int)__shfl((int)base, leader, 64);
                    if (in_hi) cand[cap + base + (unsigned int)__popcll(mhi & lt)] = v;
                }
            }
        }
    }
}

#define QD_FIN_BLOCK 1024
// one workgroup: remaining digits of both middle ranks on the candidate lists, the median, and the reset of all
// select state (the job of k_sel_finish on the six-pass path)
__global__ void __launch_bounds__(QD_FIN_BLOCK)
k_sel_final(unsigned long long* st, const double* __restrict__ cand, unsigned int* ccount, unsigned long long cap, int bits_done,
            double dflt, double* out, unsigned long long* count_out) {
    __shared__ unsigned int sh[QD_HIST_BINS];
    __shared__ unsigned int wtot[QD_FIN_BLOCK / 64];
    __shared__ unsigned long long s_prefix, s_rank;
    __shared__ unsigned long long s_res[2];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const unsigned long long n = st[0];
    if (n > 0) {
        const int up0 = 64 - bits_done;
        const bool same = (st[1] >> up0) == (st[3] >> up0);
        const int shifts[6] = {53, 42, 31, 20, 10, 0};
        const int widths[6] = {11, 11, 11, 11, 10, 10};
        for (int which = 0; which < 2; ++which) {
            const int li = (which == 1 && !same) ? 1 : 0;
            const double* list = cand + (size_t)li * cap;
            const unsigned int M = ccount[li];
            if (t == 0) { s_prefix = st[which == 0 ? 1 : 3]; s_rank = st[which == 0 ? 2 : 4]; }
            __syncthreads();
            int done = 0;
            for (int p = 0; p < 6; ++p) {
                done += widths[p];
                if (done <= bits_done) continue;
                const int shift = shifts[p], width = widths[p], up = shift + width;
                for (int k = t; k < QD_HIST_BINS; k += QD_FIN_BLOCK) sh[k] = 0u;
                __syncthreads();
                const unsigned long long pre = s_prefix >> up, r = s_rank;
                for (unsigned int k = t; k < M; k += QD_FIN_BLOCK) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(list[k]);
                    if ((bits >> up) == pre) atomicAdd(&sh[(unsigned int)((bits >> shift) & ((1u << width) - 1u))], 1u);
                }
                __syncthreads();
                // exclusive scan of the 2048 bins, 2 per thread
                const unsigned int h0 = sh[2 * t], h1 = sh[2 * t + 1];
                const unsigned int mine = h0 + h1;
                unsigned int inc = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
                if (lane == 63) wtot[wv] = inc;
                __syncthreads();
                unsigned int base = 0;
                for (int k = 0; k < wv; ++k) base += wtot[k];
                const unsigned long long excl = (unsigned long long)base + inc - mine;
                if (mine > 0 && r >= excl && r < excl + mine) {
                    const int d = (r < excl + h0) ? 2 * t : 2 * t + 1;
                    s_prefix = s_prefix | ((unsigned long long)d << shift);
                    s_rank = r - (d == 2 * t ? excl : excl + h0);
                }
                __syncthreads();
            }
            if (t == 0) s_res[which] = s_prefix;
            __syncthreads();
        }
    }
    if (t == 0) {
        if (n == 0) *out = dflt;
        else {
            const double lo = __longlong_as_double((long long)s_res[0]);
            const double hi = __longlong_as_double((long long)s_res[1]);
            *out = (n & 1ull) ? lo : (lo + hi) / 2.0;     // np.median: mean of the two middles
        }
        if (count_out) *count_out = n;
        for (int k = 0; k < 8; ++k) st[k] = 0ull;
        ccount[0] = 0u; ccount[1] = 0u;
    }
}

// result + reset of the select state for the next call; `count_out` (optional) keeps the count
__global__ void k_sel_finish(unsigned long long* st, double dflt, double* out, unsigned long long* count_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long n = st[0];
        if (n == 0) *out = dflt;
        else {
            const double lo = __longlong_as_double((long long)st[1]);
            const double hi = __longlong_as_double((long long)st[3]);
            *out = (n & 1ull) ? lo : (lo + hi) / 2.0;     // np.median: mean of the two middles
        }
        if (count_out) *count_out = n;
        for (int k = 0; k < 8; ++k) st[k] = 0ull;
    }
}

// median of the positive entries of x (after `transform`) -> device scalar slot; `dflt` if none
int qd_median_positive_dev(qd_ctx* c, const double* x, double dflt, int slot, int transform, double tparam) {
    const QdGeom G = qd_segments(c, 0).g[0];                 // owned rows only
    // digits from the top: bits 63..53, 52..42, 41..31, 30..20 (11 wide), 19..10, 9..0 (10 wide)
    const int shifts[6] = {53, 42, 31, 20, 10, 0};
    const int widths[6] = {11, 11, 11, 11, 10, 10};
    const int nblk = 256;         // few, fat workgroups (ms/step at 721x1440 with 64/128/256/512/721: 1.371/1.324/1.307/1.323/1.345)
    dim3 grid(1, std::min(G.nrows, nblk));
    if (c->geo.full && c->sel_cand) {
        // two histogram passes, one collecting pass, one finishing workgroup
        for (int p = 0; p < 2; ++p)
            hipLaunchKernelGGL(k_sel_pass, grid, dim3(QD_BLOCK), 0, c->stream, G, x, transform, tparam, c->sel_state,
                               c->hist, shifts[p], widths[p], p == 0 ? 1 : 0, 0);
        hipLaunchKernelGGL(k_sel_collect, grid, dim3(QD_BLOCK), 0, c->stream, G, x, transform, tparam, c->sel_state,
                           c->sel_cand, c->sel_ccount, (unsigned long long)c->geo.cells(), 22);
        hipLaunchKernelGGL(k_sel_final, dim3(1), dim3(QD_FIN_BLOCK), 0, c->stream, c->sel_state, c->sel_cand, c->sel_ccount,
                           (unsigned long long)c->geo.cells(), 22, dflt, c->dscal + slot, c->dcount);
        return 0;
    }
    for (int p = 0; p < 6; ++p) {
        if (c->geo.full) {
            hipLaunchKernelGGL(k_sel_pass, grid, dim3(QD_BLOCK), 0, c->stream, G, x, transform, tparam, c->sel_state,
                               c->hist, shifts[p], widths[p], p == 0 ? 1 : 0, 0);
        } else {
            hipLaunchKernelGGL(k_sel_pass, grid, dim3(QD_BLOCK), 0, c->stream, G, x, transform, tparam, c->sel_state,
                               c->hist, shifts[p], widths[p], p == 0 ? 1 : 0, 1);
            if (qd_allreduce_u32(c, c->hist, 2 * QD_HIST_BINS)) return -1;
            hipLaunchKernelGGL(k_sel_pass, dim3(1, 1), dim3(QD_BLOCK), 0, c->stream, G, x, transform, tparam, c->sel_state,
                               c->hist, shifts[p], widths[p], p == 0 ? 1 : 0, 2);
        }
    }
    hipLaunchKernelGGL(k_sel_finish, dim3(1), dim3(64), 0, c->stream, c->sel_state, dflt, c->dscal + slot, c->dcount);
    return 0;
}
