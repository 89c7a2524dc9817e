// qd_stream_push.hip -- the two launches of the ocean momentum kernel of a latitude band AROUND a split halo exchange (peer exchange,
// QD_PEER_OVERLAP=2): k_ocn_stream_push (the interior rows with the push in front) and k_ocn_stream_pair (the two boundary segments).
//
// A sub-step whose halos are due splits its momentum launch into the interior rows (computable from the margins the slabs still
// have) and two boundary strips (after the unpack; qd_ocean.hip).  For the interior to overlap the transfer, the transfer must be
// IN FLIGHT while it runs: a push kernel of its own ends only when every store it sent over the links has been acknowledged
// (s_waitcnt vmcnt(0) before the ticket), and the stream starts the next kernel after that -- on one device that is 7 us, over
// xGMI it is the whole transfer (2 x 0.2-0.9 MB per neighbour at 1/8 of 1441 x 2880 against ~50 GB/s per direction and link).
// So the push is not a kernel: it is the first J.nbx * J.nby workgroups of the interior launch.  They are dispatched first, issue
// their stores and sit in s_waitcnt while the ~900 strip workgroups behind them fill the other wave slots.
// Same strip code as k_ocn_stream (qd_stream.h); a translation unit of its own so that the whole-globe kernels are not rebuilt
// around an argument block and a branch they do not use.
#include "qd_internal.h"
#include "qd_stream.h"
#include "qd_band.h"
#include "qd_peer_dev.h"

__global__ void __launch_bounds__(192)
k_ocn_stream_push(QsOcnArgs A, QdPeerPush J) {
    const unsigned npush = (unsigned)(J.nbx * J.nby);
    if (blockIdx.x < npush) { qp_push_block(J, (int)(blockIdx.x % (unsigned)J.nbx), (int)(blockIdx.x / (unsigned)J.nbx)); return; }
    QsW W;
    qs_strip(A.G, A.vb, A.ntc, A.nrs, W, blockIdx.x - npush, gridDim.x - npush);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv <= 1) __builtin_amdgcn_s_setprio(2);
    const QsOcnArgs QD_CONST* Ak = (const QsOcnArgs QD_CONST*)__builtin_amdgcn_kernarg_segment_ptr();      // A is the FIRST argument
    const QsRec QD_CONST* fp = &Ak->rec[wv];
    if (!A.exact) {
        QsOutGlobal out{qs_make_rsrc(fp->out, W.slab_bytes), qs_off(A.G, W.o0), W.vs, (unsigned)W.nlon};
        const bool bad = qs_ocn_wave<QS_FAST>(A, W, wv, fp, out);
        if (__builtin_amdgcn_ballot_w64(bad) == 0ull) return;
    }
    QsOutGlobal out{qs_make_rsrc(fp->out, W.slab_bytes), qs_off(A.G, W.o0), W.vs, (unsigned)W.nlon};
    qs_ocn_wave<QS_EXACT>(A, W, wv, fp, out);
}

// the two boundary segments of a split launch (the rows on either side of the interior, computed after the unpack) in ONE launch: they
// are two thin sets of strips (12-20 rows each), and each was a launch of its own that could not fill its ramp
struct QsSeg2 { QdGeom G; int vb, nrs, ntc; };
__global__ void __launch_bounds__(192)
k_ocn_stream_pair(QsOcnArgs A, QsSeg2 B) {
    const unsigned n1 = (unsigned)(A.nrs * A.ntc);
    const bool second = blockIdx.x >= n1;
    const QdGeom& G = second ? B.G : A.G;
    QsW W;
    if (second) qs_strip(B.G, B.vb, B.ntc, B.nrs, W, blockIdx.x - n1, gridDim.x - n1);
    else qs_strip(A.G, A.vb, A.ntc, A.nrs, W, blockIdx.x, n1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv <= 1) __builtin_amdgcn_s_setprio(2);
    const QsOcnArgs QD_CONST* Ak = (const QsOcnArgs QD_CONST*)__builtin_amdgcn_kernarg_segment_ptr();      // A is the FIRST argument
    const QsRec QD_CONST* fp = &Ak->rec[wv];
    QsOcnArgs A2 = A;                                      // (the wave functions take the geometry from the argument block)
    if (second) A2.G = B.G;
    if (!A.exact) {
        QsOutGlobal out{qs_make_rsrc(fp->out, W.slab_bytes), qs_off(G, W.o0), W.vs, (unsigned)W.nlon};
        const bool bad = qs_ocn_wave<QS_FAST>(A2, W, wv, fp, out);
        if (__builtin_amdgcn_ballot_w64(bad) == 0ull) return;
    }
    QsOutGlobal out{qs_make_rsrc(fp->out, W.slab_bytes), qs_off(G, W.o0), W.vs, (unsigned)W.nlon};
    qs_ocn_wave<QS_EXACT>(A2, W, wv, fp, out);
}
// A: the first segment's argument block; G2 / sh2: the second segment.  One launch for both.
void qd_launch_ocn_stream_pair(qd_ctx* c, const QsOcnArgs& A, const QdGeom& G2, int vb2, int nrs2, int ntc2) {
    QsSeg2 B{G2, vb2, nrs2, ntc2};
    hipLaunchKernelGGL(k_ocn_stream_pair, dim3(A.nrs * A.ntc + nrs2 * ntc2), dim3(192), 0, c->stream, A, B);
}

// true: a deferred push was waiting and has gone out in front of the strips of A (A.nrs x A.ntc of them)
bool qd_launch_ocn_stream_push(qd_ctx* c, const QsOcnArgs& A) {
    QdPeerPush J;
    if (!qd_peer_take_job(c, &J)) return false;
    hipLaunchKernelGGL(k_ocn_stream_push, dim3(J.nbx * J.nby + A.nrs * A.ntc), dim3(192), 0, c->stream, A, J);
    return true;
}
