// qd_comm.hip -- RCCL over xGMI for latitude-band decomposition.
// One process per GPU; neighbours exchange halo rows with grouped ncclSend/ncclRecv on the
// handle's stream; global scalars (CFL max, weighted sums) use tiny all-reduces.
#include "qd_internal.h"
#include <rccl/rccl.h>
#include <cstring>

extern "C" int qd_comm_unique_id(void* id128, size_t bytes) {
    if (!id128 || bytes < sizeof(ncclUniqueId)) return -1;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int qd_comm_init(qd_handle c, const void* id128, size_t bytes) {
    if (!c || !id128 || bytes < sizeof(ncclUniqueId)) return -1;
    hipSetDevice(c->desc.device);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t comm;
    ncclResult_t r = ncclCommInitRank(&comm, c->desc.world, id, c->desc.rank);
    if (r != ncclSuccess) { c->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); return -1; }
    c->comm = (void*)comm;
    return 0;
}

extern "C" int qd_comm_allreduce_max(qd_handle c, double* inout, int n) {
    if (!c || !inout || n < 1 || n > 32) return -1;
    if (!c->comm) return 0;                 // single process: identity
    hipSetDevice(c->desc.device);
    double* d = c->dscal + QD_S_TMP0;       // 2 slots; use a dedicated region for n > 2
    if (n > 2) return qd_fail(c, "qd_comm_allreduce_max: n <= 2");
    QD_HIP(c, hipMemcpyAsync(d, inout, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    ncclResult_t r = ncclAllReduce(d, d, n, ncclDouble, ncclMax, (ncclComm_t)c->comm, c->stream);
    if (r != ncclSuccess) { c->err = std::string("ncclAllReduce: ") + ncclGetErrorString(r); return -1; }
    QD_HIP(c, hipMemcpyAsync(inout, d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    QD_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int qd_comm_barrier(qd_handle c) {
    if (!c) return -1;
    double z = 0.0;
    return qd_comm_allreduce_max(c, &z, 1);
}
