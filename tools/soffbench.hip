// (Result on MI355X: the SGPR offset IS part of the range check; with a 4 GB - 1 descriptor a lane offset of 0xFFFFFFF8 is rejected
// once any SGPR offset >= 8 is added -- the sum does not wrap -- but NOT with an SGPR offset of 0: qd_stream_base keeps a page of
// distance between the descriptor's base and the lowest array.)
// Developer probe: is the SGPR offset of a raw buffer access part of the range check?  One descriptor over array A (num_records =
// bytes of A); array B lives `delta` bytes behind A in the same allocation.  Loads of B through A's descriptor with soffset = delta:
// if the hardware checked soffset they would return 0.   hipcc --offload-arch=gfx950 -O3 tools/soffbench.hip -o tools/soffbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void k(const double* a, unsigned bytesA, unsigned delta, double* out) {
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a), 0, (int)bytesA, 0x00020000);
    const unsigned vo = threadIdx.x * 8u;
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const double x = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, delta, 0));            // B[lane]
    const double y = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, 0x80000000u, delta, 0));   // out of range on purpose
    const double z = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo + bytesA, 0, 0));       // voffset beyond num_records
    // the same with a descriptor that spans 4 GB - 1: the far array is in range; does a lane offset close to 2^32 stay rejected once the
    // SGPR offset is added (no 32-bit wrap)?
    rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a), 0, (int)0xFFFFFFFFu, 0x00020000);
    const double x2 = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rw, vo, delta, 0));
    const double y2 = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rw, 0xFFFFFFF8u, delta, 0));
    const double y3 = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rw, 0xFFFFFFF8u, 16u, 0));
    out[threadIdx.x] = x; out[64 + threadIdx.x] = y; out[128 + threadIdx.x] = z;
    out[192 + threadIdx.x] = x2; out[256 + threadIdx.x] = y2; out[320 + threadIdx.x] = y3;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, 900.0 + threadIdx.x), rw, 0xFFFFFFF8u, delta + 1024u, 0);   // must be dropped
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, 500.0 + threadIdx.x), rw, vo, delta + 1024u, 0);          // B[128 + lane]
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, 7.0 + threadIdx.x), r, vo, delta + 512u, 0);     // B[64 + lane] through A's descriptor
}
int main() {
    const size_t nA = 1 << 20, gap = (size_t)3 << 27;        // A: 8 MB; B: 3 GB behind A's start (beyond 2^31)
    char* base; if (hipMalloc(&base, gap * 8 + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    double* A = (double*)base; double* B = (double*)(base + gap * 8 - 0);   // delta = gap*8 bytes = 3 GB
    std::vector<double> hb(256); for (int i = 0; i < 256; ++i) hb[i] = 100.0 + i;
    hipMemcpy(B, hb.data(), 256 * 8, hipMemcpyHostToDevice);
    double* out; hipMalloc(&out, 384 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, (unsigned)(nA * 8), (unsigned)(gap * 8), out);
    std::vector<double> ho(384); hipMemcpy(ho.data(), out, 384 * 8, hipMemcpyDeviceToHost);
    std::vector<double> hb2(256); hipMemcpy(hb2.data(), B, 256 * 8, hipMemcpyDeviceToHost);
    printf("load via soffset=3GB: B[0]=%g B[63]=%g (want 100, 163)\n", ho[0], ho[63]);
    printf("voffset 0x80000000: %g (want 0)   voffset beyond num_records: %g (want 0)\n", ho[64], ho[128]);
    printf("store via soffset: B[64]=%g B[127]=%g (want 7, 70)\n", hb2[64], hb2[127]);
    printf("4 GB descriptor: load via soffset=3GB: %g %g (want 100, 163); lane offset 0xFFFFFFF8 + 3 GB: %g, + 16: %g (want 0, 0)\n", ho[192], ho[255], ho[256], ho[320]);
    printf("4 GB descriptor: store B[128]=%g B[191]=%g (want 500, 563); first bytes of A untouched? A[0..2] = ", hb2[128], hb2[191]);
    std::vector<double> ha(4); hipMemcpy(ha.data(), A, 32, hipMemcpyDeviceToHost); printf("%g %g %g\n", ha[0], ha[1], ha[2]);
    return 0;
}
