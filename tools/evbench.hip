// tools/evbench.hip -- developer micro-benchmark: what does a cross-stream dependency cost on this stack?
// Per iteration: kernel A on stream 1, kernel B on stream 2 that depends on A, the next A depends on B -- against the same two
// kernels back to back on one stream.   hipcc --offload-arch=gfx950 -O3 tools/evbench.hip -o tools/evbench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_work(double* p, int n, int reps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = p[i];
    for (int r = 0; r < reps; ++r) x = x * 1.0000001 + 1e-9;
    p[i] = x;
}
int main() {
    const int n = 1 << 20;
    double *a, *b; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    const int N = 2000;
    for (int timing = 0; timing < 2; ++timing) {
        hipEvent_t e1[2], e2[2];
        for (int k = 0; k < 2; ++k) { CK(hipEventCreateWithFlags(&e1[k], timing ? hipEventDefault : hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2[k], timing ? hipEventDefault : hipEventDisableTiming)); }
        for (int reps : {1, 200}) {
            // one stream
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < N; ++it) {
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s1, a, n, reps);
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s1, b, n, reps);
            }
            CK(hipDeviceSynchronize());
            const double one = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / N * 1e6;
            // two streams, chained A -> B -> A ...
            t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < N; ++it) {
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s1, a, n, reps);
                CK(hipEventRecord(e1[it & 1], s1)); CK(hipStreamWaitEvent(s2, e1[it & 1], 0));
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s2, b, n, reps);
                CK(hipEventRecord(e2[it & 1], s2)); CK(hipStreamWaitEvent(s1, e2[it & 1], 0));
            }
            CK(hipDeviceSynchronize());
            const double two = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / N * 1e6;
            // two streams, B overlaps the NEXT A (B depends on A; the A after next depends on B)
            t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < N; ++it) {
                if (it >= 2) CK(hipStreamWaitEvent(s1, e2[it & 1], 0));
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s1, a, n, reps);
                CK(hipEventRecord(e1[it & 1], s1)); CK(hipStreamWaitEvent(s2, e1[it & 1], 0));
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s2, b, n, reps);
                CK(hipEventRecord(e2[it & 1], s2));
            }
            CK(hipDeviceSynchronize());
            const double ovl = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / N * 1e6;
            printf("events %-14s kernel reps %3d: one stream %6.2f us/iter   chained across two streams %6.2f   overlapped across two streams %6.2f\n",
                   timing ? "default" : "disable-timing", reps, one, two, ovl);
        }
    }
    // the overlapped pattern captured into a hipGraph (20 iterations per graph) and replayed
    {
        hipEvent_t e1[2], e2[2], ej;
        for (int k = 0; k < 2; ++k) { CK(hipEventCreateWithFlags(&e1[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2[k], hipEventDisableTiming)); }
        CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        for (int reps : {1, 200}) {
            const int PER = 20;
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
            for (int it = 0; it < PER; ++it) {
                if (it >= 2) CK(hipStreamWaitEvent(s1, e2[it & 1], 0));
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s1, a, n, reps);
                CK(hipEventRecord(e1[it & 1], s1)); CK(hipStreamWaitEvent(s2, e1[it & 1], 0));
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s2, b, n, reps);
                CK(hipEventRecord(e2[it & 1], s2));
            }
            CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));      // join the side stream before the capture ends
            CK(hipStreamEndCapture(s1, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s1)); CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < N / PER; ++it) CK(hipGraphLaunch(ge, s1));
            CK(hipDeviceSynchronize());
            const double gr = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / N * 1e6;
            // the same kernels on ONE stream in a graph
            hipGraph_t g1; hipGraphExec_t ge1;
            CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
            for (int it = 0; it < PER; ++it) {
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s1, a, n, reps);
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s1, b, n, reps);
            }
            CK(hipStreamEndCapture(s1, &g1));
            CK(hipGraphInstantiate(&ge1, g1, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge1, s1)); CK(hipDeviceSynchronize());
            t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < N / PER; ++it) CK(hipGraphLaunch(ge1, s1));
            CK(hipDeviceSynchronize());
            const double gr1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / N * 1e6;
            printf("hipGraph            kernel reps %3d: one-stream graph %6.2f us/iter   overlapped two-stream graph %6.2f\n", reps, gr1, gr);
        }
    }
    return 0;
}
