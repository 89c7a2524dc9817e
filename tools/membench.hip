// tools/membench.hip -- developer micro-benchmark: what does the memory system allow for the access pattern of the fused
// dynamics + del^4 kernel (6 f64 fields in, 5 out, 721 x 1440) ?   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
struct P { const double* in[6]; double* out[5]; int nlat, nlon, R, ntc, nrs, own, lead, mode; };

// A: flat double2 copy, 5 fields + one extra read
__global__ void __launch_bounds__(256) k_flat(P p, size_t n2) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += st) {
        double2 a[6];
#pragma unroll
        for (int f = 0; f < 6; ++f) a[f] = ((const double2*)p.in[f])[i];
        a[0].x += a[5].x * 1e-300; a[0].y += a[5].y * 1e-300;
#pragma unroll
        for (int f = 0; f < 5; ++f) ((double2*)p.out[f])[i] = a[f];
    }
}

// B: row strips, one field per wave (5 waves per workgroup), `own` owned columns starting at lane `lead`, PD rows in flight
template <int PD>
__global__ void __launch_bounds__(320) k_strip(P p) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nb = gridDim.x, L = blockIdx.x, per = nb >> 3, rem = nb & 7u, x = L & 7u;
    unsigned w = x * per + (x < rem ? x : rem) + (L >> 3);
    if (p.mode & 1) w = L;                                   // no XCD remap
    int rs = w / p.ntc, cs = w % p.ntc;
    if (p.mode & 8) { cs = w / p.nrs; rs = w % p.nrs; }      // row strips fastest
    const int jraw = cs * p.own - p.lead + lane;
    const int j = jraw < 0 ? jraw + p.nlon : (jraw >= p.nlon ? jraw - p.nlon : jraw);
    const bool ok = lane >= p.lead && lane < p.lead + p.own && jraw < p.nlon;
    const unsigned bytes = (unsigned)(p.nlat + 8) * p.nlon * 8u;
    const rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)p.in[wv], 0, bytes, 0x00020000);
    const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.out[wv], 0, bytes, 0x00020000);
    const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.in[5], 0, bytes, 0x00020000);
    const unsigned vo = j * 8u, vs = ok ? jraw * 8u : 0x80000000u;
    const int o0 = rs * p.R, o1 = rs == p.nrs - 1 ? p.nlat : o0 + p.R;
    const int g0 = o0 - 4 > 0 ? o0 - 4 : 0, g1 = o1 + 4 < p.nlat ? o1 + 4 : p.nlat;
    unsigned ro_ = g0 * p.nlon * 8u, so = o0 * p.nlon * 8u;
    const unsigned stride = p.nlon * 8u;
    double q[PD], e[PD];
#pragma unroll
    for (int k = 0; k < PD; ++k) { q[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ri, vo, ro_, 0));
        e[k] = wv < 2 ? __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, vo, ro_, 0)) : 0.0; ro_ += stride; }
    for (int g = g0; g < g1; ++g) {
        double c = q[0] + e[0] * 1e-300;
#pragma unroll
        for (int k = 0; k + 1 < PD; ++k) { q[k] = q[k + 1]; e[k] = e[k + 1]; }
        if (!(p.mode & 4)) { q[PD - 1] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ri, vo, ro_, 0));
        if (wv < 2) e[PD - 1] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, vo, ro_, 0)); }
        ro_ += stride;
        if (g >= o0 && g < o1 && (!(p.mode & 2) || c == 12345.0)) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, c), ro, vs, so, 0); so += stride; }
    }
}

// C: B plus optional costs of the real kernel: EXTRA extra loads for waves 0,1 (h / friction), SL scalar-table loads per row,
// NV dependent f64 operations per row, LDSB bytes of LDS per workgroup (occupancy limiter)
template <int PD, int EXTRA, int SL, int NV, int LDSB>
__global__ void __launch_bounds__(320) k_strip2(P p, const double* tab) {
    __shared__ char lds_[LDSB > 0 ? LDSB : 4];
    if (LDSB > 0 && threadIdx.x == 9999) lds_[0] = 1;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nb = gridDim.x, L = blockIdx.x, per = nb >> 3, rem = nb & 7u, x = L & 7u;
    const unsigned w = x * per + (x < rem ? x : rem) + (L >> 3);
    const int rs = w / p.ntc, cs = w % p.ntc;
    const int jraw = cs * p.own - p.lead + lane;
    const int j = jraw < 0 ? jraw + p.nlon : (jraw >= p.nlon ? jraw - p.nlon : jraw);
    const bool ok = lane >= p.lead && lane < p.lead + p.own && jraw < p.nlon;
    const unsigned bytes = (unsigned)(p.nlat + 8) * p.nlon * 8u;
    const rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)p.in[wv], 0, bytes, 0x00020000);
    const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.out[wv], 0, bytes, 0x00020000);
    const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.in[5], 0, bytes, 0x00020000);
    const rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.in[2], 0, bytes, 0x00020000);
    const unsigned vo = j * 8u, vs = ok ? jraw * 8u : 0x80000000u;
    const int o0 = rs * p.R, o1 = rs == p.nrs - 1 ? p.nlat : o0 + p.R;
    const int g0 = o0 - 4 > 0 ? o0 - 4 : 0, g1 = o1 + 4 < p.nlat ? o1 + 4 : p.nlat;
    // mode & 16: "head to head" -- odd row strips stream their rows in DESCENDING order, so that both neighbours of every strip
    // boundary touch the shared halo rows at the same time (both at their start, or both at their end): the second reader finds them in L2
    const bool desc = (p.mode & 16) && (rs & 1);
    unsigned ro_ = (desc ? g1 - 1 : g0) * p.nlon * 8u, so = (desc ? o1 - 1 : o0) * p.nlon * 8u;
    const unsigned stride = desc ? 0u - p.nlon * 8u : p.nlon * 8u;
    double q[PD], e[PD], e2[PD];
    auto ld = [&](int k) {
        q[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ri, vo, ro_, 0));
        e[k] = (EXTRA >= 1 && wv < 2) ? __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, vo, ro_, 0)) : 0.0;
        e2[k] = (EXTRA >= 2 && wv < 2) ? __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ry, vo, ro_, 0)) : 0.0;
        ro_ += stride;
    };
#pragma unroll
    for (int k = 0; k < PD; ++k) ld(k);
    typedef const double __attribute__((address_space(4)))* cptr;
    double acc = 0.0;
    for (int t = 0; t < g1 - g0; ++t) {
        const int g = desc ? g1 - 1 - t : g0 + t;
        double c = q[0] + (e[0] + e2[0]) * 1e-300;
#pragma unroll
        for (int k = 0; k + 1 < PD; ++k) { q[k] = q[k + 1]; e[k] = e[k + 1]; e2[k] = e2[k + 1]; }
        ld(PD - 1);
        if (SL > 0) {
            cptr t = (cptr)(unsigned long long)tab + 4u * (unsigned)g;
#pragma unroll
            for (int k = 0; k < SL; ++k) c += t[k] * 1e-300;
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) acc = acc * 0.999 + c;
        c += acc * 1e-300;
        if (g >= o0 && g < o1) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, c), ro, vs, so, 0); so += stride; }
    }
}

// D: two columns per lane (16-byte accesses), 120 owned columns per strip (lanes 2..61)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int PD, int EXTRA, int SL, int NV>
__global__ void __launch_bounds__(320) k_strip16(P p, const double* tab) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nb = gridDim.x, L = blockIdx.x, per = nb >> 3, rem = nb & 7u, x = L & 7u;
    const unsigned w = x * per + (x < rem ? x : rem) + (L >> 3);
    const int rs = w / p.ntc, cs = w % p.ntc;
    const int jraw = cs * 120 - 4 + 2 * lane;
    const int j = jraw < 0 ? jraw + p.nlon : (jraw >= p.nlon ? jraw - p.nlon : jraw);
    const bool ok = lane >= 2 && lane < 62 && jraw < p.nlon;
    const unsigned bytes = (unsigned)(p.nlat + 8) * p.nlon * 8u;
    const rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)p.in[wv], 0, bytes, 0x00020000);
    const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.out[wv], 0, bytes, 0x00020000);
    const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.in[5], 0, bytes, 0x00020000);
    const rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.in[2], 0, bytes, 0x00020000);
    const unsigned vo = j * 8u, vs = ok ? jraw * 8u : 0x80000000u;
    const int o0 = rs * p.R, o1 = rs == p.nrs - 1 ? p.nlat : o0 + p.R;
    const int g0 = o0 - 4 > 0 ? o0 - 4 : 0, g1 = o1 + 4 < p.nlat ? o1 + 4 : p.nlat;
    unsigned ro_ = g0 * p.nlon * 8u, so = o0 * p.nlon * 8u;
    const unsigned stride = p.nlon * 8u;
    double2 q[PD], e[PD], e2[PD];
    auto ld = [&](int k) {
        q[k] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(ri, vo, ro_, 0));
        e[k] = (EXTRA >= 1 && wv < 2) ? __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, ro_, 0)) : double2{0, 0};
        e2[k] = (EXTRA >= 2 && wv < 2) ? __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(ry, vo, ro_, 0)) : double2{0, 0};
        ro_ += stride;
    };
#pragma unroll
    for (int k = 0; k < PD; ++k) ld(k);
    typedef const double __attribute__((address_space(4)))* cptr;
    double a0 = 0.0, a1 = 0.0;
    for (int g = g0; g < g1; ++g) {
        double c0 = q[0].x + (e[0].x + e2[0].x) * 1e-300, c1 = q[0].y + (e[0].y + e2[0].y) * 1e-300;
#pragma unroll
        for (int k = 0; k + 1 < PD; ++k) { q[k] = q[k + 1]; e[k] = e[k + 1]; e2[k] = e2[k + 1]; }
        ld(PD - 1);
        if (SL > 0) {
            cptr t = (cptr)(unsigned long long)tab + 4u * (unsigned)g;
#pragma unroll
            for (int k = 0; k < SL; ++k) { c0 += t[k] * 1e-300; }
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) { a0 = a0 * 0.999 + c0; a1 = a1 * 0.999 + c1; }
        c0 += a0 * 1e-300; c1 += a1 * 1e-300;
        if (g >= o0 && g < o1) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, double2{c0, c1}), ro, vs, so, 0); so += stride; }
    }
}

int main(int argc, char** argv) {
    const int nlat = 721, nlon = 1440;
    const size_t cells = (size_t)(nlat + 8) * nlon;
    P p; p.nlat = nlat; p.nlon = nlon; p.mode = 0;
    std::vector<double> h(cells, 1.0);
    for (int f = 0; f < 6; ++f) { double* d; CK(hipMalloc(&d, cells * 8)); CK(hipMemcpy(d, h.data(), cells * 8, hipMemcpyHostToDevice)); p.in[f] = d; }
    for (int f = 0; f < 5; ++f) { double* d; CK(hipMalloc(&d, cells * 8)); CK(hipMemset(d, 0, cells * 8)); p.out[f] = d; }
    // ~120 MB of other traffic between launches, as in the real step
    double* junk; CK(hipMalloc(&junk, 128u << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        float best = 1e9f, sum = 0;
        for (int it = 0; it < 12; ++it) {
            if (argc > 1) CK(hipMemsetAsync(junk, it, 128u << 20, 0));
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%-44s mean %7.2f us   best %7.2f us   (%.2f TB/s on 91.4 MB)\n", name, sum / 10 * 1e3, best * 1e3, 91.4e6 / (sum / 10 * 1e-3) / 1e12);
    };
    const size_t n2 = (size_t)nlat * nlon / 2;
    for (int nb : {1024, 2048, 4096}) { char nm[64]; snprintf(nm, 64, "flat double2 copy, %d x 256", nb); time(nm, [&] { hipLaunchKernelGGL(k_flat, dim3(nb), dim3(256), 0, 0, p, n2); }); }
    for (int own : {58, 64}) for (int R : {12, 16, 24, 32}) {
        p.own = own; p.lead = own == 64 ? 0 : 3; p.R = R; p.ntc = (nlon + own - 1) / own; p.nrs = nlat / R;
        char nm[80];
        snprintf(nm, 80, "strips own=%d R=%d PD=2 (%d wgs)", own, R, p.ntc * p.nrs); time(nm, [&] { hipLaunchKernelGGL(k_strip<2>, dim3(p.ntc * p.nrs), dim3(320), 0, 0, p); });
        snprintf(nm, 80, "strips own=%d R=%d PD=4 (%d wgs)", own, R, p.ntc * p.nrs); time(nm, [&] { hipLaunchKernelGGL(k_strip<4>, dim3(p.ntc * p.nrs), dim3(320), 0, 0, p); });
    }
    for (int mode : {0, 1, 8, 9, 2, 4}) for (int R : {16, 24}) {
        p.own = 58; p.lead = 3; p.R = R; p.ntc = (nlon + 57) / 58; p.nrs = nlat / R; p.mode = mode;
        char nm[96]; snprintf(nm, 96, "strips R=%d PD=2 mode=%d (1 noremap 8 rowfast 2 loadonly 4 storeonly)", R, mode);
        time(nm, [&] { hipLaunchKernelGGL(k_strip<2>, dim3(p.ntc * p.nrs), dim3(320), 0, 0, p); });
    }
    p.mode = 0;
    double* tab; CK(hipMalloc(&tab, (nlat + 16) * 4 * 8)); CK(hipMemset(tab, 0, (nlat + 16) * 4 * 8));
    for (int R : {16, 24}) {
        p.own = 58; p.lead = 3; p.R = R; p.ntc = (nlon + 57) / 58; p.nrs = nlat / R;
        char nm[96];
#define RUN(PD, EX, SL, NV, LB) snprintf(nm, 96, "R=%d PD=%d extra=%d sload=%d valu=%d lds=%d", R, PD, EX, SL, NV, LB); \
        time(nm, [&] { hipLaunchKernelGGL((k_strip2<PD, EX, SL, NV, LB>), dim3(p.ntc * p.nrs), dim3(320), 0, 0, p, tab); });
        RUN(2, 0, 0, 0, 0) RUN(2, 1, 0, 0, 0) RUN(2, 2, 0, 0, 0) RUN(2, 2, 4, 0, 0) RUN(2, 2, 0, 30, 0) RUN(2, 2, 4, 30, 0) RUN(2, 2, 4, 60, 0)
        RUN(2, 2, 4, 30, 32768) RUN(2, 2, 4, 30, 40000) RUN(3, 2, 4, 30, 0)
    }
    for (int R : {16, 24, 32}) for (int mode : {0, 16}) {
        p.own = 58; p.lead = 3; p.R = R; p.ntc = (nlon + 57) / 58; p.nrs = nlat / R; p.mode = mode;
        char nm[96];
        snprintf(nm, 96, "HEAD-TO-HEAD mode=%d R=%d PD=2 extra=2 sload=4 valu=30", mode, R);
        time(nm, [&] { hipLaunchKernelGGL((k_strip2<2, 2, 4, 30, 0>), dim3(p.ntc * p.nrs), dim3(320), 0, 0, p, tab); });
        snprintf(nm, 96, "HEAD-TO-HEAD mode=%d R=%d PD=3 extra=2 sload=4 valu=30", mode, R);
        time(nm, [&] { hipLaunchKernelGGL((k_strip2<3, 2, 4, 30, 0>), dim3(p.ntc * p.nrs), dim3(320), 0, 0, p, tab); });
    }
    p.mode = 0;
    for (int R : {8, 12, 16, 24}) {
        p.R = R; p.ntc = 12; p.nrs = nlat / R;
        char nm[96];
#define RUN16(PD, EX, SL, NV) snprintf(nm, 96, "16B R=%d PD=%d extra=%d sload=%d valu=%d (%d wgs)", R, PD, EX, SL, NV, p.ntc * p.nrs); \
        time(nm, [&] { hipLaunchKernelGGL((k_strip16<PD, EX, SL, NV>), dim3(p.ntc * p.nrs), dim3(320), 0, 0, p, tab); });
        RUN16(2, 0, 0, 0) RUN16(2, 2, 4, 0) RUN16(2, 2, 4, 12) RUN16(2, 2, 4, 24) RUN16(3, 2, 4, 12)
    }
    return 0;
}
