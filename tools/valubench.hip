// Developer micro-benchmark: issue cost of the f64 VALU / DPP instructions the row-streaming kernels are made of (gfx950).
// One workgroup per CU, W waves per SIMD; every wave runs N independent-chain instructions of one kind between two s_memtime
// stamps; prints cycles per instruction per wave and per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/valubench.hip -o tools/valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define OUTER 64
template <int KIND> __global__ void k(double* out, unsigned long long* cyc, double a, double b) {
    double x0 = a + threadIdx.x, x1 = a * 2 + threadIdx.x, x2 = a * 3, x3 = a * 4 + threadIdx.x, x4 = b, x5 = b * 2, x6 = b * 3, x7 = b * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int o = 0; o < OUTER; ++o) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (KIND == 0) { x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b); x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b); }
            if (KIND == 1) { x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a; }
            if (KIND == 2) { x0 = x0 + a; x1 = x1 + a; x2 = x2 + a; x3 = x3 + a; x4 = x4 + a; x5 = x5 + a; x6 = x6 + a; x7 = x7 + a; }
            if (KIND == 3) {        // wave_shl:1 DPP move of a double = 2 v_mov_b32_dpp
#define SH(x) { int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xf, 0xf, true); int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xf, 0xf, true); x = __hiloint2double(hi, lo); }
                SH(x0) SH(x1) SH(x2) SH(x3) SH(x4) SH(x5) SH(x6) SH(x7)
            }
            if (KIND == 4) {        // row_shl:1 DPP move (inside a row of 16 lanes)
#define SR(x) { int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x101, 0xf, 0xf, true); int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x101, 0xf, 0xf, true); x = __hiloint2double(hi, lo); }
                SR(x0) SR(x1) SR(x2) SR(x3) SR(x4) SR(x5) SR(x6) SR(x7)
            }
            if (KIND == 5) {        // dependent chain: one accumulator
                x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b);
                x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b);
            }
            if (KIND == 6) {        // f32 fma for reference
                float f0 = (float)x0, f1 = (float)x1; 
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(f0), "+v"(f1) : "v"((float)a), "v"((float)b));
                x0 = f0; x1 = f1;
            }
            if (KIND == 7) {        // v_mov_b64
                asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int KIND> void run(const char* name, int waves_per_simd, int ninst_per_rep) {
    const int threads = 256 * waves_per_simd, blocks = 256;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * threads * blocks); hipMalloc(&cyc, 8 * blocks * threads / 64);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    const double per_wave = s / h.size() / ((double)OUTER * REP / 8 * ninst_per_rep);
    std::printf("%-28s %d wave(s)/SIMD: %6.2f memtime ticks per instr per wave  (%5.2f per instr per SIMD)\n", name, waves_per_simd, per_wave, per_wave / waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_fma_f64", 1, 8); run<1>("v_mul_f64", 1, 8); run<2>("v_add_f64", 1, 8); run<3>("dpp wave_shl (2 x b32)", 1, 16); run<4>("dpp row_shl (2 x b32)", 1, 16); run<5>("v_fma_f64 dependent", 1, 8); run<6>("v_fma_f32 (2 chains)", 1, 8); run<7>("v_mov_b64", 1, 8); }
        if (w == 2) { run<0>("v_fma_f64", 2, 8); run<1>("v_mul_f64", 2, 8); run<2>("v_add_f64", 2, 8); run<3>("dpp wave_shl (2 x b32)", 2, 16); run<4>("dpp row_shl (2 x b32)", 2, 16); run<5>("v_fma_f64 dependent", 2, 8); run<6>("v_fma_f32 (2 chains)", 2, 8); run<7>("v_mov_b64", 2, 8); }
        if (w == 4) { run<0>("v_fma_f64", 4, 8); run<1>("v_mul_f64", 4, 8); run<2>("v_add_f64", 4, 8); run<3>("dpp wave_shl (2 x b32)", 4, 16); run<4>("dpp row_shl (2 x b32)", 4, 16); run<5>("v_fma_f64 dependent", 4, 8); run<6>("v_fma_f32 (2 chains)", 4, 8); run<7>("v_mov_b64", 4, 8); }
    }
    return 0;
}
