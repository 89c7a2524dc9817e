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
#include "qd_band.h"
#include "qd_fluxes.h"
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


// ------------------------------------------------------------------ windowed-histogram median (whole-globe handles)
// The medians of a step are medians of fields whose scale does not jump between steps.  Each call site keeps its last
// median on the device.  Pass 1 (k_med_hist) histograms the positives over a WINDOW of bit patterns centred on it -- a factor
// 16 either way, 2048 bins of 2^44 ulps (0.2 % wide) -- and counts the positives and those below the window.  Pass 2
// (k_med_scan_bracket; bands: k_med_bracket behind an all-reduce and a one-workgroup scan) locates the bins of the two middle
// ranks, i.e. their value range [lo, hi], copies the positives inside it to a list (a few hundred values) and recounts; one
// workgroup (k_med_final) selects the ranks from the list, held in LDS.  That is three launches and two passes over the field
// where the digit-by-digit select needs four and three.  The window has to be that wide: in the benchmark's first hundred steps
// the three medians move by 5-60 % from one step to the next (scripts/median_drift.py), so a bracket guessed from the last
// median alone would miss most of the time.
// Exact whatever the window was: if a rank falls outside it (first use after an upload, a field that became all-zero) the
// finishing workgroup runs the radix select over the whole field itself -- slow, rare, and it re-centres the window.
// pred[site]: {last median, -, -, valid, hits, misses, last list length, last count | lo, hi, -, bracket ok}
#define QD_MED_SITES 4
#define QD_MED_BAND_CAP 4092u      // candidates per band in the gathered segments (4096 doubles each)
#define QD_MED_WSHIFT 44
#define QD_MED_WG_STAGE 512        // candidates a workgroup of the collecting pass stages in LDS (4 KB)
#define QD_MED_LDS_LIST 4096       // candidates the finishing workgroup keeps in LDS (32 KB)
// window of bit patterns [base, base + 2048 << 44) around a site's last median: centre / 16 .. centre * 16 (8 binades)
__device__ __forceinline__ unsigned long long qd_med_window_base(bool valid, double centre) {
    const unsigned long long cbits = (unsigned long long)__double_as_longlong(valid ? centre : 1.0);
    const unsigned long long four = 4ull << 52;
    return cbits > four ? cbits - four : 0ull;
}

struct QdMedBracket { unsigned long long lo_b, hi_b; bool ok, none; };

// One workgroup of QD_BLOCK threads, the summed histogram in sh[0 .. QD_HIST_BINS) and {positives, positives below the window} in
// cnt[0..1] (LDS, already visible to every thread): the bit patterns [lo_b, hi_b] that hold both middle ranks.  Every thread gets
// the same answer.  No usable window (first call, a regime change of the field): the side the ranks went to, or every positive
// value -- the finisher then selects from the field itself.
__device__ __forceinline__ QdMedBracket qd_med_scan(const unsigned int* sh, const unsigned long long* cnt, bool valid,
                                                    unsigned long long base) {
    __shared__ unsigned int wtot[QD_BLOCK / 64];
    __shared__ int s_bin[2];
    const int t = threadIdx.x, lane = t & 63;
    const int per = QD_HIST_BINS / QD_BLOCK;
    unsigned int sacc = 0;
    for (int k = 0; k < per; ++k) sacc += sh[t * per + k];
    unsigned int inc = sacc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
    if (lane == 63) wtot[t >> 6] = inc;
    if (t < 2) s_bin[t] = -1;
    __syncthreads();
    unsigned int wbase = 0, total = 0;
    for (int k = 0; k < (t >> 6); ++k) wbase += wtot[k];
    for (int k = 0; k < QD_BLOCK / 64; ++k) total += wtot[k];
    const unsigned long long excl = (unsigned long long)wbase + inc - sacc;
    const unsigned long long m = cnt[0], below = cnt[1];
    const unsigned long long k1 = m ? (m - 1ull) / 2ull : 0ull, k2 = m / 2ull;
    const bool inside = valid && m > 0ull && k1 >= below && k2 < below + (unsigned long long)total;
    if (inside && sacc > 0) {
        for (int which = 0; which < 2; ++which) {
            const unsigned long long r = (which == 0 ? k1 : k2) - below;
            if (r >= excl && r < excl + sacc) {
                unsigned long long cum = excl;
                int d = t * per;
                for (; d < t * per + per; ++d) { const unsigned int hv = sh[d]; if (cum + hv > r) break; cum += hv; }
                s_bin[which] = d;
            }
        }
    }
    __syncthreads();
    QdMedBracket B;
    B.ok = inside && s_bin[0] >= 0 && s_bin[1] >= s_bin[0];
    B.none = m == 0ull;
    const unsigned long long top = base + ((unsigned long long)QD_HIST_BINS << QD_MED_WSHIFT);   // first pattern above the window
    const unsigned long long maxb = 0x7FEFFFFFFFFFFFFFull;                                        // DBL_MAX
    B.lo_b = 1ull; B.hi_b = maxb;                             // no usable window: every positive value is a candidate
    if (B.ok) {
        B.lo_b = base + ((unsigned long long)s_bin[0] << QD_MED_WSHIFT);
        B.hi_b = base + ((unsigned long long)(s_bin[1] + 1) << QD_MED_WSHIFT) - 1ull;
    } else if (valid && m > 0ull) {
        // the ranks left the window (a regime change of the field): hand the finisher the side they went to
        if (k2 < below) B.hi_b = base - 1ull;
        else if (k1 >= below + (unsigned long long)total) B.lo_b = top;
    }
    if (B.lo_b < 1ull) B.lo_b = 1ull;
    if (B.hi_b > maxb) B.hi_b = maxb;
    __syncthreads();                                          // s_bin / wtot may be reused by a second call
    return B;
}

// mode 1: histogram only (every workgroup adds its LDS histogram to hist[]; nothing is scanned or reset here)
// mode 2: scan only, one workgroup -- latitude bands all-reduce the histogram in between; publishes pred[8..11], resets hist[]
// (whole-globe handles: mode 1, then every workgroup of k_med_scan_bracket scans for itself)
__device__ __forceinline__ void qd_med_hist_body(const QdGeom& G, const double* __restrict__ x, int transform, double tparam, double* pred,
                                                 unsigned int* hist, int mode, int nblk_x, int nblk_y, int bx, int by) {
    __shared__ unsigned int sh[QD_HIST_BINS + 2];             // + count of positives, + count below the window
    __shared__ unsigned long long s_st[2];
    const int t = threadIdx.x, lane = t & 63;
    const bool valid = pred[3] != 0.0;
    const unsigned long long base = qd_med_window_base(valid, pred[0]);
    if (mode != 2) {
        const int jstep = nblk_x * QD_BLOCK;
        unsigned int n_pos = 0, n_below = 0;
        const int i_end = G.row0 + G.nrows;
        // the NEXT batch (the next column block of the row, or the next row of this workgroup) is in flight while this one is binned: a
        // workgroup has three rows at 721 x 1440, and three dependent round trips were a quarter of the launch
        double vbuf[8];
        {
            const int i0 = G.row0 + by;
            const size_t b0 = (size_t)qd_lrow(G, i0 < i_end ? i0 : i_end - 1) * G.nlon;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int j = bx * QD_BLOCK + q * jstep + t; vbuf[q] = x[b0 + (j < G.nlon ? j : G.nlon - 1)]; }
        }
        for (int k = t; k < QD_HIST_BINS + 2; k += QD_BLOCK) sh[k] = 0u;      // (behind the first loads: they fly while the bins are cleared)
        __syncthreads();
        for (int i = G.row0 + by; i < i_end; i += nblk_y) {
            for (int jb = bx * QD_BLOCK; jb < G.nlon; jb += 8 * jstep) {
                double cur[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) cur[q] = vbuf[q];
                {
                    const bool more_cols = jb + 8 * jstep < G.nlon;
                    const int in = more_cols ? i : i + nblk_y;
                    const int jn = more_cols ? jb + 8 * jstep : bx * QD_BLOCK;
                    if (in < i_end) {
                        const size_t bn = (size_t)qd_lrow(G, in) * G.nlon;
#pragma unroll
                        for (int q = 0; q < 8; ++q) { const int j = jn + q * jstep + t; vbuf[q] = x[bn + (j < G.nlon ? j : G.nlon - 1)]; }
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j0 = jb + q * jstep;
                    if (j0 >= G.nlon) break;
                    const int j = j0 + t;
                    const double v = (j < G.nlon) ? qd_med_value(cur[q], transform, tparam) : 0.0;
                    const bool pos = v > 0.0;
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                    n_pos += pos ? 1u : 0u;
                    const bool below = pos && bits < base;
                    n_below += below ? 1u : 0u;
                    const unsigned long long idx = (bits - base) >> QD_MED_WSHIFT;
                    if (pos && !below && idx < (unsigned long long)QD_HIST_BINS) atomicAdd(&sh[(unsigned int)idx], 1u);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { n_pos += __shfl_down(n_pos, o, 64); n_below += __shfl_down(n_below, o, 64); }
        if (lane == 0) {                                      // one global atomic per workgroup and counter, like any other bin
            if (n_pos) atomicAdd(&sh[QD_HIST_BINS], n_pos);
            if (n_below) atomicAdd(&sh[QD_HIST_BINS + 1], n_below);
        }
        __syncthreads();
        for (int k = t; k < QD_HIST_BINS + 2; k += QD_BLOCK) if (sh[k]) atomicAdd(&hist[k], sh[k]);
        return;
    }
    // ---- mode 2 (bands): the all-reduced histogram -> bracket
    for (int k = t; k < QD_HIST_BINS; k += QD_BLOCK) sh[k] = hist[k];
    if (t < 2) s_st[t] = (unsigned long long)hist[QD_HIST_BINS + t];
    __syncthreads();
    const QdMedBracket B = qd_med_scan(sh, s_st, valid, base);
    if (t == 0) {
        pred[8] = __longlong_as_double((long long)B.lo_b);
        pred[9] = __longlong_as_double((long long)B.hi_b);
        pred[10] = 0.0;
        pred[11] = 1.0;
    }
    for (int k = t; k < QD_HIST_BINS + 2; k += QD_BLOCK) hist[k] = 0u;
}

__global__ void __launch_bounds__(QD_BLOCK)
k_med_hist(QdGeom G, const double* __restrict__ x, int transform, double tparam, double* pred, unsigned int* hist, int mode) {
    qd_med_hist_body(G, x, transform, tparam, pred, hist, mode, (int)gridDim.x, (int)gridDim.y, (int)blockIdx.x, (int)blockIdx.y);
}

__global__ void k_med_seed(double* pred, const double* out, const unsigned long long* count) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { pred[0] = *out; pred[3] = (*count > 0ull) ? 1.0 : 0.0; }
}

// one value of the collecting pass: counts, and a slot in the candidate list when it lies inside [lo, hi] (one returning atomic per
// wavefront that holds any)
__device__ __forceinline__ void qd_med_collect(double v, double lo, double hi, int lane, unsigned long long lt, unsigned int& n_pos,
                                               unsigned int& n_below, double* __restrict__ cand, unsigned int* __restrict__ ccount,
                                               unsigned int cap) {
    const bool pos = v > 0.0;
    n_pos += pos ? 1u : 0u;
    n_below += (pos && v < lo) ? 1u : 0u;
    const bool in = pos && v >= lo && v <= hi;
    const unsigned long long m = __ballot(in);
    if (m) {
        unsigned int base = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) base = atomicAdd(&ccount[0], (unsigned int)__popcll(m));
        base = (unsigned int)__shfl((int)base, leader, 64);
        const unsigned int idx = base + (unsigned int)__popcll(m & lt);
        if (in && idx < cap) cand[idx] = v;                  // the count keeps running past the capacity: overflow is visible
    }
}

// lo > hi: empty bracket (the finisher falls back to the field itself)
__device__ __forceinline__ void qd_med_bracket_body(const QdGeom& G, const double* __restrict__ x, int transform, double tparam, double lo, double hi,
                                                    unsigned long long* st, double* __restrict__ cand, unsigned int* __restrict__ ccount,
                                                    unsigned int cap, int nblk_x, int nblk_y, int bx, int by) {
    const int t = threadIdx.x, lane = t & 63;
    const int jstep = nblk_x * QD_BLOCK;
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned int n_pos = 0, n_below = 0;
    for (int i = G.row0 + by; i < G.row0 + G.nrows; i += nblk_y) {
        const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
        for (int jb = bx * QD_BLOCK; jb < G.nlon; jb += 8 * jstep) {
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
                qd_med_collect(v, lo, hi, lane, lt, n_pos, n_below, cand, ccount, cap);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n_pos += __shfl_down(n_pos, o, 64); n_below += __shfl_down(n_below, o, 64); }
    __shared__ unsigned int s_c[2][QD_BLOCK / 64];
    if (lane == 0) { s_c[0][t >> 6] = n_pos; s_c[1][t >> 6] = n_below; }
    __syncthreads();
    if (t < 2) {                                              // one global atomic per workgroup and counter
        unsigned int a = 0;
        for (int k = 0; k < QD_BLOCK / 64; ++k) a += s_c[t][k];
        if (a) atomicAdd(&st[t], (unsigned long long)a);
    }
}

__global__ void __launch_bounds__(QD_BLOCK)
k_med_bracket(QdGeom G, const double* __restrict__ x, int transform, double tparam, const double* __restrict__ pred,
              unsigned long long* st, double* __restrict__ cand, unsigned int* __restrict__ ccount, unsigned int cap) {
    if (pred[2] != 0.0) return;                                               // k_med_hist counted no positive entry (its pred[10]: this kernel gets pred + 8)
    const bool valid = pred[3] != 0.0;
    const double lo = valid ? pred[0] : 0.0, hi = valid ? pred[1] : -1.0;     // invalid: empty bracket, the finisher falls back
    qd_med_bracket_body(G, x, transform, tparam, lo, hi, st, cand, ccount, cap, (int)gridDim.x, (int)gridDim.y, (int)blockIdx.x, (int)blockIdx.y);
}

// Whole-globe handles: pass 2 with the scan in front of it.  EVERY workgroup reads the summed histogram (8 KB, written by the
// previous launch's atomics) and finds the bracket for itself -- 256 redundant scans of a microsecond each instead of one scan
// behind a ticket, a fence and a launch boundary (round 3: k_med_hist was 17 us of which the ticket -> acquire -> 2050 loads -> scan
// -> publish -> reset tail of its last workgroup was 6) -- while the first row of the field is already on its way.  Workgroup
// (0, 0) also publishes the bracket for the finisher (pred[8..11]).  k_med_final resets the histogram.
__device__ __forceinline__ void qd_med_scan_bracket_body(const QdGeom& G, const double* __restrict__ x, int transform, double tparam, double* pred,
                                                         const unsigned int* __restrict__ hist, unsigned long long* st, double* __restrict__ cand,
                                                         unsigned int* __restrict__ ccount, unsigned int cap) {
    __shared__ unsigned int sh[QD_HIST_BINS];
    __shared__ unsigned long long s_cnt[2];
    __shared__ unsigned int s_c[2][QD_BLOCK / 64];
    // candidates of this workgroup are staged in LDS and appended to the list with ONE returning global atomic per workgroup: one per
    // wavefront with a hit was ~1 000 atomics on one word per launch, which serialise at ~90 per us -- the collecting pass of a site
    // with 950 candidates took 13.6 us, of one with 240 candidates 9.7
    __shared__ double s_cand[QD_MED_WG_STAGE];
    __shared__ unsigned int s_n, s_base;
    const int t = threadIdx.x, lane = t & 63;
    if (t == 0) s_n = 0u;
    const int nby = (int)gridDim.y, by = (int)blockIdx.y;
    const bool valid = pred[3] != 0.0;
    const unsigned long long base = qd_med_window_base(valid, pred[0]);
    constexpr int PER = QD_HIST_BINS / QD_BLOCK;
    unsigned int hreg[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) hreg[k] = hist[t + k * QD_BLOCK];
    unsigned int creg = 0;
    if (t < 2) creg = hist[QD_HIST_BINS + t];
    // the first row of this workgroup: in flight during the scan
    const int i_end = G.row0 + G.nrows;
    int i = G.row0 + by;
    double vbuf[8];
    {
        const size_t b = (size_t)qd_lrow(G, i < i_end ? i : i_end - 1) * G.nlon;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int j = q * QD_BLOCK + t; vbuf[q] = x[b + (j < G.nlon ? j : G.nlon - 1)]; }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) sh[t + k * QD_BLOCK] = hreg[k];
    if (t < 2) s_cnt[t] = (unsigned long long)creg;
    __syncthreads();
    const QdMedBracket B = qd_med_scan(sh, s_cnt, valid, base);
    if (t == 0 && by == 0 && blockIdx.x == 0) {
        pred[8] = __longlong_as_double((long long)B.lo_b);
        pred[9] = __longlong_as_double((long long)B.hi_b);
        pred[10] = B.none ? 1.0 : 0.0;
        pred[11] = 1.0;
    }
    if (B.none) return;                                       // no positive entry: the finisher finds its counts at zero and writes the default
    const double lo = __longlong_as_double((long long)B.lo_b), hi = __longlong_as_double((long long)B.hi_b);
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned int n_pos = 0, n_below = 0;
    for (; i < i_end; i += nby) {
        for (int jb = 0; jb < G.nlon; jb += 8 * QD_BLOCK) {
            double cur[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) cur[q] = vbuf[q];
            // the next batch (the next column block of this row, or the next row) while this one is filtered
            {
                const bool more_cols = jb + 8 * QD_BLOCK < G.nlon;
                const int in = more_cols ? i : i + nby;
                const int jn = more_cols ? jb + 8 * QD_BLOCK : 0;
                if (in < i_end) {
                    const size_t b = (size_t)qd_lrow(G, in) * G.nlon;
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const int j = jn + q * QD_BLOCK + t; vbuf[q] = x[b + (j < G.nlon ? j : G.nlon - 1)]; }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j0 = jb + q * QD_BLOCK;
                if (j0 >= G.nlon) break;                                  // wave-uniform
                const int j = j0 + t;
                const double v = (j < G.nlon) ? qd_med_value(cur[q], transform, tparam) : 0.0;
                const bool pos = v > 0.0;
                n_pos += pos ? 1u : 0u;
                n_below += (pos && v < lo) ? 1u : 0u;
                const bool in = pos && v >= lo && v <= hi;
                const unsigned long long m = __ballot(in);
                if (m) {
                    unsigned int base = 0;
                    const int leader = __ffsll((long long)m) - 1;
                    if (lane == leader) base = atomicAdd(&s_n, (unsigned int)__popcll(m));       // LDS
                    base = (unsigned int)__shfl((int)base, leader, 64);
                    const unsigned int idx = base + (unsigned int)__popcll(m & lt);
                    if (in) {
                        if (idx < (unsigned int)QD_MED_WG_STAGE) s_cand[idx] = v;
                        else { const unsigned int g = atomicAdd(&ccount[0], 1u); if (g < cap) cand[g] = v; }     // a crowded bracket: straight to the list
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n_pos += __shfl_down(n_pos, o, 64); n_below += __shfl_down(n_below, o, 64); }
    if (lane == 0) { s_c[0][t >> 6] = n_pos; s_c[1][t >> 6] = n_below; }
    __syncthreads();
    {
        const unsigned int ns = s_n < (unsigned int)QD_MED_WG_STAGE ? s_n : (unsigned int)QD_MED_WG_STAGE;
        if (t == 0 && ns) s_base = atomicAdd(&ccount[0], ns);               // the count keeps running past the capacity: overflow is visible
        __syncthreads();
        for (unsigned int k = t; k < ns; k += QD_BLOCK) { const unsigned int g = s_base + k; if (g < cap) cand[g] = s_cand[k]; }
    }
    if (t < 2) {                                              // one global atomic per workgroup and counter
        unsigned int a = 0;
        for (int k = 0; k < QD_BLOCK / 64; ++k) a += s_c[t][k];
        if (a) atomicAdd(&st[t], (unsigned long long)a);
    }
}

__global__ void __launch_bounds__(QD_BLOCK)
k_med_scan_bracket(QdGeom G, const double* __restrict__ x, int transform, double tparam, double* pred, const unsigned int* __restrict__ hist,
                   unsigned long long* st, double* __restrict__ cand, unsigned int* __restrict__ ccount, unsigned int cap) {
    qd_med_scan_bracket_body(G, x, transform, tparam, pred, hist, st, cand, ccount, cap);
}

// ---- two medians in ONE set of three launches (whole-globe handles): the median of a field is a chain of dependent round trips that
// leaves most of the chip idle (256 workgroups, one finishing workgroup), so two fields that are both there -- the precipitation
// field and time_step's P_cond at the start of the cloud block -- go through the chain side by side: blockIdx.z picks the job, every
// job has a histogram, select state and candidate list of its own.  Same device functions as the single kernels: same bits.
struct QdMedJob {
    const double* x; int transform; double tparam, dflt; double* pred; unsigned int* hist; unsigned long long* st; double* cand;
    unsigned int* ccount; double* out; unsigned long long* count_out;
};
__global__ void __launch_bounds__(QD_BLOCK)
k_med_hist2(QdGeom G, QdMedJob J0, QdMedJob J1) {
    const QdMedJob& J = blockIdx.z ? J1 : J0;
    qd_med_hist_body(G, J.x, J.transform, J.tparam, J.pred, J.hist, 1, (int)gridDim.x, (int)gridDim.y, (int)blockIdx.x, (int)blockIdx.y);
}
// k_med_hist2 whose second job PRODUCES its field: time_step's P_cond (phase 1 of the column: dynamics.py:282-297, the humidity column of
// qd_fluxes.h -- the same device function k_column runs, so the same bits) is computed, stored and binned in one go; k_column<1> was a
// launch of 13 us in front of the pair.  Workgroups [0, nb0) are job 0's fat workgroups (qd_med_hist_body), [nb0, 2 nb0) job 1's.
struct QdPcondSrc { QdColP P; const double *u, *v, *h, *Ts, *q, *hice; const uint8_t* land; double* Pcond; };
__global__ void __launch_bounds__(QD_BLOCK)
k_med_hist2p(QdGeom G, QdMedJob J0, QdMedJob J1, QdPcondSrc S, int nb0) {
    if ((int)blockIdx.y < nb0) {
        qd_med_hist_body(G, J0.x, J0.transform, J0.tparam, J0.pred, J0.hist, 1, 1, nb0, 0, (int)blockIdx.y);
        return;
    }
    // job 1: as few, fat workgroups as job 0; six cells per thread in flight.  MEASURED, and off by default (QD_MED_FOLD): 30.0 us for this
    // launch against 12.9 (k_med_hist2) + 13.4 (k_column<1>) -- a cell of the humidity column is two f64 exponentials, and a histogram
    // pass wants few workgroups (every one flushes its non-zero bins with global atomics: one workgroup per row, 721 flushes, 32.8 us)
    // while the exponentials want four waves per SIMD to hide their dependent latencies, not one
    __shared__ unsigned int sh[QD_HIST_BINS + 2];
    const int t = threadIdx.x, lane = t & 63;
    const int nb1 = (int)gridDim.y - nb0, by = (int)blockIdx.y - nb0;
    const bool valid = J1.pred[3] != 0.0;
    const unsigned long long base = qd_med_window_base(valid, J1.pred[0]);
    for (int k = t; k < QD_HIST_BINS + 2; k += QD_BLOCK) sh[k] = 0u;
    __syncthreads();
    unsigned int n_pos = 0, n_below = 0;
    constexpr int NC = 6;
    for (int i = G.row0 + by; i < G.row0 + G.nrows; i += nb1) {
        const size_t b = (size_t)qd_lrow(G, i) * G.nlon;
        for (int jb = 0; jb < G.nlon; jb += NC * QD_BLOCK) {
            double u_[NC], v_[NC], h_[NC], t_[NC], q_[NC], hi_[NC]; int l_[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int j = jb + k * QD_BLOCK + t;
                const size_t o = b + (j < G.nlon ? j : G.nlon - 1);
                u_[k] = S.u[o]; v_[k] = S.v[o]; h_[k] = S.h[o]; t_[k] = S.Ts[o]; q_[k] = S.q[o]; hi_[k] = S.hice[o]; l_[k] = (int)S.land[o];
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int j = jb + k * QD_BLOCK + t;
                if (j >= G.nlon) continue;
                const double qsat_air = qd_qsat(288.0 + S.P.ga * h_[k], S.P.p0);
                const double v = qd_humidity_column(S.P, u_[k], v_[k], t_[k], q_[k], qsat_air, l_[k] == 1, hi_[k]).Pc;
                S.Pcond[b + j] = v;
                const bool pos = v > 0.0;
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                n_pos += pos ? 1u : 0u;
                const bool below = pos && bits < base;
                n_below += below ? 1u : 0u;
                const unsigned long long idx = (bits - base) >> QD_MED_WSHIFT;
                if (pos && !below && idx < (unsigned long long)QD_HIST_BINS) atomicAdd(&sh[(unsigned int)idx], 1u);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n_pos += __shfl_down(n_pos, o, 64); n_below += __shfl_down(n_below, o, 64); }
    if (lane == 0) {
        if (n_pos) atomicAdd(&sh[QD_HIST_BINS], n_pos);
        if (n_below) atomicAdd(&sh[QD_HIST_BINS + 1], n_below);
    }
    __syncthreads();
    for (int k = t; k < QD_HIST_BINS + 2; k += QD_BLOCK) if (sh[k]) atomicAdd(&J1.hist[k], sh[k]);
}

__global__ void __launch_bounds__(QD_BLOCK)
k_med_scan_bracket2(QdGeom G, QdMedJob J0, QdMedJob J1, unsigned int cap) {
    const QdMedJob& J = blockIdx.z ? J1 : J0;
    qd_med_scan_bracket_body(G, J.x, J.transform, J.tparam, J.pred, J.hist, J.st, J.cand, J.ccount, cap);
}

// value of element k of the select source: the candidate list, or (fallback) the transformed field; non-positive = skip
// src 0: candidate list; 1: the field; 2: the gathered per-band segments [world][4 + cap] = {m_r, c_lo_r, M_r, -, candidates}
__device__ __forceinline__ double qd_med_src(const double* __restrict__ list, const double* __restrict__ field, int src,
                                             size_t k, int transform, double tparam, unsigned int cap) {
    if (src == 1) return qd_med_value(field[k], transform, tparam);
    if (src == 0) return list[k];
    const size_t r = k / cap, i = k - r * cap;
    const double* seg = list + r * (size_t)(cap + 4u);
    return ((double)i < seg[2]) ? seg[4 + i] : 0.0;
}

// band handles: counts of the bracket pass into the header of this band's segment, select state reset
__global__ void k_med_pack(unsigned long long* st, unsigned int* ccount, double* seg) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        seg[0] = (double)st[0]; seg[1] = (double)st[1]; seg[2] = (double)ccount[0]; seg[3] = 0.0;
        st[0] = 0ull; st[1] = 0ull; ccount[0] = 0u;
    }
}

template <int NT>
__device__ __forceinline__ void qd_med_final_body(unsigned long long* st, const double* __restrict__ cand, unsigned int* ccount, double* pred,
                                                  const double* __restrict__ field, unsigned long long n_field, int transform, double tparam,
                                                  double dflt, double* out, unsigned long long* count_out, int world, unsigned int cap,
                                                  double* miss_flag, unsigned int* hist_reset) {
    // world > 0: latitude bands -- `cand` holds the all-gathered segments; counts are the sums of their headers; when the
    // ranks are not inside the gathered lists (a band overflowed its capacity, or the window missed) nothing is written but
    // *miss_flag = 1 and the host falls back to the digit-by-digit select on every band
    __shared__ unsigned int sh[QD_HIST_BINS];
    __shared__ unsigned int wtot[NT / 64];
    __shared__ unsigned long long s_prefix, s_rank;
    __shared__ double s_min[NT / 64];
    __shared__ unsigned int s_cnt[NT / 64];
    __shared__ double s_list[QD_MED_LDS_LIST];                          // the candidate list, when it fits: six passes from LDS, not from L2
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (hist_reset) for (int k = t; k < QD_HIST_BINS + 2; k += NT) hist_reset[k] = 0u;      // whole globe: k_med_scan_bracket has read it
    unsigned long long m, c_lo;
    unsigned int M;
    bool overflow = false;
    if (world > 0) {
        m = 0ull; c_lo = 0ull; M = 0u;
        for (int r = 0; r < world; ++r) {
            const double* seg = cand + (size_t)r * (cap + 4u);
            m += (unsigned long long)seg[0]; c_lo += (unsigned long long)seg[1];
            overflow |= seg[2] > (double)cap;
            M += (unsigned int)(seg[2] > (double)cap ? (double)cap : seg[2]);
        }
    } else {
        m = __hip_atomic_load(&st[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c_lo = __hip_atomic_load(&st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        M = __hip_atomic_load(&ccount[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool valid = pred[11] != 0.0;
    const double plo = pred[8], phi = pred[9];
    __syncthreads();                                                     // every thread has read the state before thread 0 resets it
    if (m == 0ull) {
        if (t == 0) {
            *out = dflt; if (count_out) *count_out = 0ull; pred[3] = 0.0;
            if (world > 0) *miss_flag = 0.0; else { st[0] = 0ull; st[1] = 0ull; ccount[0] = 0u; }
        }
        return;
    }
    const unsigned long long k1 = (m - 1ull) / 2ull, k2 = m / 2ull;
    const bool hit = valid && !overflow && c_lo <= k1 && k2 < c_lo + (unsigned long long)M;
    if (world > 0 && !hit) { if (t == 0) *miss_flag = 1.0; return; }
    // latitude bands: the gathered segments hold M candidates in world x cap slots; when they fit, they are compacted into the LDS
    // list first and the passes below run as on a whole-globe handle (six passes over world x cap slots of L2, a division per
    // slot: 18.5 us for the finisher of a 1/8 band against 11 on the whole globe)
    const bool compact = world > 0 && M <= (unsigned int)QD_MED_LDS_LIST;
    const int src = compact ? 0 : ((world > 0) ? 2 : (hit ? 0 : 1));
    const size_t N = (src == 2) ? (size_t)world * cap : (src == 1 ? (size_t)n_field : (size_t)M);
    // leading bits shared by every value inside [lo, hi] (positive doubles order like their bit patterns)
    int common = 0;
    unsigned long long pre0 = 0ull;
    if (hit) {
        const unsigned long long bl = (unsigned long long)__double_as_longlong(plo), bh = (unsigned long long)__double_as_longlong(phi);
        common = (bl == bh) ? 64 : __clzll((long long)(bl ^ bh));
        pre0 = bl;
    }
    if (t == 0) { s_prefix = 0ull; s_rank = hit ? (k1 - c_lo) : k1; }
    if (compact) {
        unsigned int off = 0;
        for (int r = 0; r < world; ++r) {
            const double* seg = cand + (size_t)r * (cap + 4u);
            const unsigned int cnt = (unsigned int)seg[2];               // <= cap: an overflowing segment is a miss (above)
            for (unsigned int k = t; k < cnt; k += NT) s_list[off + k] = seg[4 + k];
            off += cnt;
        }
        cand = s_list;
    } else if (src == 0 && N <= (size_t)QD_MED_LDS_LIST) {
        for (size_t k = t; k < N; k += NT) s_list[k] = cand[k];
        cand = s_list;
    }
    __syncthreads();
    const int shifts[6] = {53, 42, 31, 20, 10, 0};
    const int widths[6] = {11, 11, 11, 11, 10, 10};
    int done = 0;
    // The usual case -- the ranks inside the bracket, the list in LDS -- takes digits that START at the first bit the bracket leaves
    // open (a bracket one or two bins of 2^44 patterns wide fixes the top 19-20 bits: with the fixed digit boundaries the pass for
    // bits 52..42 told 2-4 values apart and four more passes followed), and stops as soon as ONE candidate is left under the prefix:
    // ~1 000 candidates spread over 2^44 patterns are alone in their 11-bit bin after the first pass, the value is then simply
    // looked up.  Ties and crowded bins take further digits; a median of everything else takes the fixed digits below.
    const bool lds_fast = hit && cand == s_list;
    __shared__ unsigned int s_sel;
    if (lds_fast) {
        int rem = 64 - common;                                              // bits still open below the common prefix
        if (t == 0) s_prefix = rem >= 64 ? 0ull : (rem == 0 ? pre0 : ((pre0 >> rem) << rem));
        __syncthreads();
        while (rem > 0) {
            const int width = rem < 11 ? rem : 11, shift = rem - width, up = rem;
            for (int k = t; k < QD_HIST_BINS; k += NT) sh[k] = 0u;
            __syncthreads();
            const unsigned long long pre = up >= 64 ? 0ull : (s_prefix >> up), r = s_rank;
            for (size_t k = t; k < N; k += NT) {
                const double v = cand[k];
                if (!(v > 0.0)) continue;
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                if (up >= 64 || (bits >> up) == pre) atomicAdd(&sh[(unsigned int)((bits >> shift) & ((1u << width) - 1u))], 1u);
            }
            __syncthreads();
            constexpr int PER = QD_HIST_BINS / NT;
            unsigned int mine = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) mine += sh[PER * t + q];
            unsigned int inc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
            if (lane == 63) wtot[wv] = inc;
            __syncthreads();
            unsigned int base = 0;
            for (int k = 0; k < wv; ++k) base += wtot[k];
            const unsigned long long excl = (unsigned long long)base + inc - mine;
            if (mine > 0 && r >= excl && r < excl + mine) {
                unsigned long long cum = excl;
                int d = PER * t;
                for (; d < PER * t + PER - 1; ++d) { const unsigned int hv = sh[d]; if (cum + hv > r) break; cum += hv; }
                s_prefix = s_prefix | ((unsigned long long)d << shift);
                s_rank = r - cum;
                s_sel = sh[d];
            }
            __syncthreads();
            rem = shift;
            if (s_sel == 1u && rem > 0) {                                    // alone under its prefix: look the value up
                const unsigned long long want = s_prefix >> rem;
                __syncthreads();                                             // everybody has read s_prefix before its owner rewrites it
                for (size_t k = t; k < N; k += NT) {
                    const double v = cand[k];
                    if (!(v > 0.0)) continue;
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                    if ((bits >> rem) == want) { s_prefix = bits; s_rank = 0ull; }
                }
                __syncthreads();
                break;
            }
        }
    }
    for (int p = 0; p < 6 && !lds_fast; ++p) {
        const int shift = shifts[p], width = widths[p], up = shift + width;
        done += width;
        if (done <= common) {                                            // digit fixed by the bracket: take it from lo
            if (t == 0) s_prefix |= pre0 & (((1ull << width) - 1ull) << shift);
            __syncthreads();
            continue;
        }
        for (int k = t; k < QD_HIST_BINS; k += NT) sh[k] = 0u;
        __syncthreads();
        const unsigned long long pre = up >= 64 ? 0ull : (s_prefix >> up), r = s_rank;
        for (size_t k = t; k < N; k += NT) {
            const double v = qd_med_src(cand, field, src, k, transform, tparam, cap);
            if (!(v > 0.0)) continue;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            if (up >= 64 || (bits >> up) == pre) atomicAdd(&sh[(unsigned int)((bits >> shift) & ((1u << width) - 1u))], 1u);
        }
        __syncthreads();
        constexpr int PER = QD_HIST_BINS / NT;                            // bins per thread
        unsigned int mine = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) mine += sh[PER * t + q];
        unsigned int inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
        if (lane == 63) wtot[wv] = inc;
        __syncthreads();
        unsigned int base = 0;
        for (int k = 0; k < wv; ++k) base += wtot[k];
        const unsigned long long excl = (unsigned long long)base + inc - mine;
        if (mine > 0 && r >= excl && r < excl + mine) {
            unsigned long long cum = excl;
            int d = PER * t;
            for (; d < PER * t + PER - 1; ++d) { const unsigned int hv = sh[d]; if (cum + hv > r) break; cum += hv; }
            s_prefix = s_prefix | ((unsigned long long)d << shift);
            s_rank = r - cum;
        }
        __syncthreads();
    }
    const double v1 = __longlong_as_double((long long)s_prefix);
    double v2 = v1;
    if (!(m & 1ull)) {
        // rank k1 + 1: v1 again when it is repeated beyond rank k1, else the smallest value above it.  s_rank is now the rank
        // of the target inside its run of equal values (0-based), so the run must be longer than s_rank + 1.
        unsigned int c_eq = 0;
        double mn = DBL_MAX;
        for (size_t k = t; k < N; k += NT) {
            const double v = qd_med_src(cand, field, src, k, transform, tparam, cap);
            if (!(v > 0.0)) continue;
            c_eq += (v == v1) ? 1u : 0u;
            if (v > v1 && v < mn) mn = v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { c_eq += __shfl_down(c_eq, o, 64); const double y = __shfl_down(mn, o, 64); mn = y < mn ? y : mn; }
        if (lane == 0) { s_cnt[wv] = c_eq; s_min[wv] = mn; }
        __syncthreads();
        if (t == 0) {
            unsigned int ce = 0; double mm = DBL_MAX;
            for (int k = 0; k < NT / 64; ++k) { ce += s_cnt[k]; mm = s_min[k] < mm ? s_min[k] : mm; }
            v2 = ((unsigned long long)ce > s_rank + 1ull) ? v1 : mm;
        }
    }
    if (t == 0) {
        const double med = (m & 1ull) ? v1 : (v1 + v2) / 2.0;           // np.median: mean of the two middles
        *out = med;
        if (count_out) *count_out = m;
        pred[12] += 1.0;
        if (!hit) { pred[13] = pred[12]; pred[14] = pred[0]; pred[15] = med; }  // last miss: call number, window centre, result
        pred[0] = med; pred[3] = 1.0;                                     // centre of the next call's window
        pred[hit ? 4 : 5] += 1.0; pred[6] = (double)M; pred[7] = (double)m;      // statistics (QD_MEDIAN_DEBUG)
        if (world > 0) *miss_flag = 0.0; else { st[0] = 0ull; st[1] = 0ull; ccount[0] = 0u; }
    }
}

__global__ void __launch_bounds__(QD_FIN_BLOCK)
k_med_final(unsigned long long* st, const double* __restrict__ cand, unsigned int* ccount, double* pred,
            const double* __restrict__ field, unsigned long long n_field, int transform, double tparam, double dflt, double* out,
            unsigned long long* count_out, int world, unsigned int cap, double* miss_flag, unsigned int* hist_reset,
            double* host_flag, double* host_stamp, double seq) {
    qd_med_final_body<QD_FIN_BLOCK>(st, cand, ccount, pred, field, n_field, transform, tparam, dflt, out, count_out, world, cap, miss_flag,
                                    hist_reset);
    if (host_flag) {
        // latitude bands: the host decides on the miss flag whether the six-pass select has to run -- it goes out from here, behind a
        // stamp the host polls (no k_publish_host launch, no stream synchronisation)
        __syncthreads();
        if (threadIdx.x == 0) {
            const double f = __hip_atomic_load(miss_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((unsigned long long*)host_flag, (unsigned long long)__double_as_longlong(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store((unsigned long long*)host_stamp, (unsigned long long)__double_as_longlong(seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void __launch_bounds__(QD_FIN_BLOCK)
k_med_final2(QdMedJob J0, QdMedJob J1, unsigned long long n_field) {
    const QdMedJob& J = blockIdx.x ? J1 : J0;
    qd_med_final_body<QD_FIN_BLOCK>(J.st, J.cand, J.ccount, J.pred, J.x, n_field, J.transform, J.tparam, J.dflt, J.out, J.count_out, 0, 0u,
                                    (double*)nullptr, J.hist);
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

// Two medians at once (see k_med_hist2): job 0 on the handle's first set of median buffers, job 1 on the second (hist_b ...).  Falls back to
// two calls in a row when a site has no window yet, or the handle has no second set.
// both sites have a window and the handle has the second set of buffers: the pair runs as ONE chain (else: two calls in a row)
bool qd_median_pair_ready(const qd_ctx* c, int site0, int site1) {
    return c->geo.full && c->sel_cand && c->med_pred && c->med_predict && c->hist_b && site0 >= 0 && site0 < QD_MED_SITES &&
           site1 >= 0 && site1 < QD_MED_SITES && site0 != site1 && c->med_seen[site0] && c->med_seen[site1];
}
// the pair with job 1 = time_step's P_cond, produced by the histogram pass itself (k_med_hist2p); only when qd_median_pair_ready
int qd_median_pair_pcond_dev(qd_ctx* c, const double* x0, double dflt0, int slot0, int tr0, double tp0, int site0,
                             const QdColP& P, double dflt1, int slot1, int site1) {
    if (!qd_median_pair_ready(c, site0, site1)) return qd_fail(c, "qd_median_pair_pcond_dev: the pair is not ready");
    const QdGeom G = qd_segments(c, 0).g[0];
    const int nb0 = std::min(G.nrows, c->tune.med_blocks);
    double** F = c->f;
    const QdMedJob J0{x0, tr0, tp0, dflt0, c->med_pred + 16 * site0, c->hist, c->sel_state, c->sel_cand, c->sel_ccount, c->dscal + slot0, c->dcount};
    const QdMedJob J1{F[QD_F_PCOND], 0, 0.0, dflt1, c->med_pred + 16 * site1, c->hist_b, c->sel_state_b, c->sel_cand_b, c->sel_ccount_b,
                      c->dscal + slot1, c->dcount + 8};
    const QdPcondSrc S{P, F[QD_F_U], F[QD_F_V], F[QD_F_H], F[QD_F_TS], F[QD_F_Q], F[QD_F_HICE], c->land, F[QD_F_PCOND]};
    hipLaunchKernelGGL(k_med_hist2p, dim3(1, 2 * nb0), dim3(QD_BLOCK), 0, c->stream, G, J0, J1, S, nb0);
    hipLaunchKernelGGL(k_med_scan_bracket2, dim3(1, nb0, 2), dim3(QD_BLOCK), 0, c->stream, G, J0, J1, (unsigned int)c->geo.cells());
    hipLaunchKernelGGL(k_med_final2, dim3(2), dim3(QD_FIN_BLOCK), 0, c->stream, J0, J1, (unsigned long long)c->geo.cells());
    return 0;
}

int qd_median_pair_dev(qd_ctx* c, const double* x0, double dflt0, int slot0, int tr0, double tp0, int site0,
                       const double* x1, double dflt1, int slot1, int tr1, double tp1, int site1) {
    const bool ok = c->geo.full && c->sel_cand && c->med_pred && c->med_predict && c->hist_b && site0 >= 0 && site0 < QD_MED_SITES &&
                    site1 >= 0 && site1 < QD_MED_SITES && site0 != site1 && c->med_seen[site0] && c->med_seen[site1];
    if (!ok) {
        if (qd_median_positive_dev(c, x0, dflt0, slot0, tr0, tp0, site0)) return -1;
        return qd_median_positive_dev(c, x1, dflt1, slot1, tr1, tp1, site1);
    }
    const QdGeom G = qd_segments(c, 0).g[0];
    const dim3 grid(1, std::min(G.nrows, c->tune.med_blocks), 2);
    const QdMedJob J0{x0, tr0, tp0, dflt0, c->med_pred + 16 * site0, c->hist, c->sel_state, c->sel_cand, c->sel_ccount, c->dscal + slot0, c->dcount};
    const QdMedJob J1{x1, tr1, tp1, dflt1, c->med_pred + 16 * site1, c->hist_b, c->sel_state_b, c->sel_cand_b, c->sel_ccount_b, c->dscal + slot1,
                      c->dcount + 8};
    hipLaunchKernelGGL(k_med_hist2, grid, dim3(QD_BLOCK), 0, c->stream, G, J0, J1);
    hipLaunchKernelGGL(k_med_scan_bracket2, grid, dim3(QD_BLOCK), 0, c->stream, G, J0, J1, (unsigned int)c->geo.cells());
    hipLaunchKernelGGL(k_med_final2, dim3(2), dim3(QD_FIN_BLOCK), 0, c->stream, J0, J1, (unsigned long long)c->geo.cells());
    return 0;
}

// median of the positive entries of x (after `transform`) -> device scalar slot; `dflt` if none
int qd_median_positive_dev(qd_ctx* c, const double* x, double dflt, int slot, int transform, double tparam, int site) {
    const QdGeom G = qd_segments(c, 0).g[0];                 // owned rows only
    // digits from the top: bits 63..53, 52..42, 41..31, 30..20 (11 wide), 19..10, 9..0 (10 wide)
    const int shifts[6] = {53, 42, 31, 20, 10, 0};
    const int widths[6] = {11, 11, 11, 11, 10, 10};
    const int nblk = c->tune.med_blocks;               // few, fat workgroups; round 3b with the finisher-less histogram pass: 128 / 192 / 256 / 384 / 512 / 721 -> 0.897 / 0.895 / 0.893 / 0.899 / 0.913 / 0.921 ms/step; (ms/step at 721x1440 with 64/128/256/512/721: 1.371/1.324/1.307/1.323/1.345)
    dim3 grid(1, std::min(G.nrows, nblk));
    if (c->geo.full && c->sel_cand && c->med_pred && c->med_predict && site >= 0 && site < QD_MED_SITES) {
        // windowed histogram around the site's last median, one collecting pass, one finishing workgroup
        double* pred = c->med_pred + 16 * site;
        // a median that runs on the side stream BESIDE the main stream's medians (qd_pcond_median_side, qd_atmos.hip) has a select
        // state, histogram, candidate list and counter of its own
        const bool sb = c->med_side_active && c->hist_b;
        unsigned int* const hist = sb ? c->hist_b : c->hist;
        unsigned long long* const sel_state = sb ? c->sel_state_b : c->sel_state;
        double* const sel_cand = sb ? c->sel_cand_b : c->sel_cand;
        unsigned int* const sel_ccount = sb ? c->sel_ccount_b : c->sel_ccount;
        unsigned long long* const dcount = sb ? c->dcount + 8 : c->dcount;
        if (c->med_seen[site]) {
            hipLaunchKernelGGL(k_med_hist, grid, dim3(QD_BLOCK), 0, c->stream, G, x, transform, tparam, pred, hist, 1);
            hipLaunchKernelGGL(k_med_scan_bracket, grid, dim3(QD_BLOCK), 0, c->stream, G, x, transform, tparam, pred, hist, sel_state,
                               sel_cand, sel_ccount, (unsigned int)c->geo.cells());
            hipLaunchKernelGGL(k_med_final, dim3(1), dim3(QD_FIN_BLOCK), 0, c->stream, sel_state, sel_cand, sel_ccount, pred, x,
                               (unsigned long long)c->geo.cells(), transform, tparam, dflt, c->dscal + slot, dcount, 0, 0u,
                               (double*)nullptr, hist, (double*)nullptr, (double*)nullptr, 0.0);
            return 0;
        }
        c->med_seen[site] = 1;                                // first use: the digit-by-digit select below, then seed the window
        for (int p = 0; p < 2; ++p)
            hipLaunchKernelGGL(k_sel_pass, grid, dim3(QD_BLOCK), 0, c->stream, G, x, transform, tparam, sel_state,
                               hist, shifts[p], widths[p], p == 0 ? 1 : 0, 0);
        hipLaunchKernelGGL(k_sel_collect, grid, dim3(QD_BLOCK), 0, c->stream, G, x, transform, tparam, sel_state,
                           sel_cand, sel_ccount, (unsigned long long)c->geo.cells(), 22);
        hipLaunchKernelGGL(k_sel_final, dim3(1), dim3(QD_FIN_BLOCK), 0, c->stream, sel_state, sel_cand, sel_ccount,
                           (unsigned long long)c->geo.cells(), 22, dflt, c->dscal + slot, dcount);
        hipLaunchKernelGGL(k_med_seed, dim3(1), dim3(64), 0, c->stream, pred, c->dscal + slot, dcount);
        return 0;
    }
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
    const bool band_window = !c->geo.full && c->med_pred && c->med_gather && c->med_predict && site >= 0 && site < QD_MED_SITES;
    double* bpred = band_window ? c->med_pred + 16 * site : nullptr;
    if (band_window && c->med_seen[site]) {
        // latitude bands, 2 collectives instead of 6: all-reduce the windowed histogram, every band publishes the same bracket,
        // collects its own candidates into its segment, one all-reduce of the zero-padded segments acts as an all-gather, every
        // band selects from the same gathered list.  An overflowing segment or a missed window shows up in the flag the host
        // reads back (bands synchronise with the host once per step anyway) and falls through to the six-pass select below.
        const int world = c->desc.world, rank = c->desc.rank;
        const unsigned int cap = QD_MED_BAND_CAP;
        const size_t segd = (size_t)cap + 4u;
        hipLaunchKernelGGL(k_med_hist, grid, dim3(QD_BLOCK), 0, c->stream, G, x, transform, tparam, bpred, c->hist, 1);
        if (qd_allreduce_u32(c, c->hist, QD_HIST_BINS + 2)) return -1;
        if (!qd_peer_on(c)) QD_HIP(c, hipMemsetAsync(c->med_gather, 0, (size_t)world * segd * sizeof(double), c->stream));
        double* seg = c->med_gather + (size_t)rank * segd;
        // every workgroup of the collecting pass scans the all-reduced histogram for itself (the whole globe's k_med_scan_bracket: no
        // one-workgroup scan launch in between); k_med_final resets the histogram and hands the miss flag to the host
        hipLaunchKernelGGL(k_med_scan_bracket, grid, dim3(QD_BLOCK), 0, c->stream, G, x, transform, tparam, bpred, c->hist, c->sel_state,
                           seg + 4, c->sel_ccount, cap);
        if (qd_peer_hooks(c)) {
            // the header of my segment is written by the gathering kernel itself (QdPeerHook::pre 1 = k_med_pack)
            QdPeerHook H; H.pre = 1; H.st = c->sel_state; H.cc = c->sel_ccount;
            c->allreduces++;
            if (qd_peer_allgather_hooked(c, c->med_gather, (int)segd, H)) return -1;
        } else {
        hipLaunchKernelGGL(k_med_pack, dim3(1), dim3(64), 0, c->stream, c->sel_state, c->sel_ccount, seg);
        if (qd_allgather_f64(c, c->med_gather, (int)segd)) return -1;
        }
        c->pub_seq += 1.0;
        hipLaunchKernelGGL(k_med_final, dim3(1), dim3(QD_FIN_BLOCK), 0, c->stream, c->sel_state, c->med_gather, c->sel_ccount, bpred, x,
                           0ull, transform, tparam, dflt, c->dscal + slot, c->dcount, world, cap, c->dscal + QD_S_TMP1, c->hist,
                           c->hpin + 32, c->hpin + 61, c->pub_seq);
        if (qd_wait_host_flag(c, c->hpin + 61, c->pub_seq, "band median: the miss flag never reached the host")) return -1;
        if (c->hpin[60] != 0.0) return qd_fail(c, "peer exchange: a rank did not arrive within the deadline");
        if (c->hpin[32] == 0.0) return 0;
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
    if (band_window) {                                        // (re-)centre the window of this call site on the exact result
        hipLaunchKernelGGL(k_med_seed, dim3(1), dim3(64), 0, c->stream, bpred, c->dscal + slot, c->dcount);
        c->med_seen[site] = 1;
    }
    return 0;
}
