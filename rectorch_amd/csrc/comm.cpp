// comm.cpp -- RCCL hooks of the C ABI (include/rectorch_hip.h: rtx_comm_*), so that a host that is NOT Python can run the
// data-parallel step: rtx_engine_loss_grads -> rtx_comm_allreduce / rtx_comm_reduce_scatter on the gradient buffers ->
// rtx_engine_apply_adam[_rows] -> rtx_comm_allgather of the compute copies.  (The Python mirror reaches RCCL through
// torch.distributed, rectorch_amd/parallel.py; the reference has no collective code at all.)
//
// RCCL is bound at RUN time (dlopen of librccl.so, the copy a hosting process -- e.g. PyTorch-ROCm -- has already loaded
// wins): librectorch_hip.so itself has no link-time dependency on it and single-GPU users never touch it.
#include "../../include/rectorch_hip.h"
#include "rtx_common.h"

#include <dlfcn.h>
#include <string.h>

namespace {

// the slice of the NCCL API used here (rccl.h: ncclResult_t is an int enum, 0 = success; ncclDataType_t: 7 = float32,
// 9 = bfloat16, 0 = int8; ncclRedOp_t: 0 = sum)
struct NcclId { char internal[128]; };
typedef void* NcclComm;
struct Api {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Api g_api;

int load_api()
{
    if (g_api.lib) return RTX_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    RTX_CHECK(h, RTX_ESTATE, "rtx_comm: librccl.so could not be loaded (%s)", dlerror());
#define RTX_SYM(field, name)                                                               \
    g_api.field = (decltype(g_api.field))dlsym(h, name);                                   \
    RTX_CHECK(g_api.field, RTX_ESTATE, "rtx_comm: librccl.so has no symbol %s", name);
    RTX_SYM(GetUniqueId, "ncclGetUniqueId")
    RTX_SYM(CommInitRank, "ncclCommInitRank")
    RTX_SYM(CommDestroy, "ncclCommDestroy")
    RTX_SYM(AllReduce, "ncclAllReduce")
    RTX_SYM(ReduceScatter, "ncclReduceScatter")
    RTX_SYM(AllGather, "ncclAllGather")
    RTX_SYM(GroupStart, "ncclGroupStart")
    RTX_SYM(GroupEnd, "ncclGroupEnd")
    RTX_SYM(GetErrorString, "ncclGetErrorString")
#undef RTX_SYM
    g_api.lib = h;
    return RTX_OK;
}

#define RTX_NCCL(expr)                                                                                \
    do {                                                                                              \
        int _r = (expr);                                                                              \
        if (_r != 0) {                                                                                \
            rtx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_api.GetErrorString(_r));    \
            return RTX_EHIP;                                                                          \
        }                                                                                             \
    } while (0)

int nccl_dtype(int dtype, size_t* bytes)
{
    if (dtype == RTX_FP32) { *bytes = 4; return 7; }   // ncclFloat32
    *bytes = 2;
    return 9;                                          // ncclBfloat16
}

}  // namespace

struct rtx_comm {
    NcclComm comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int rtx_comm_unique_id(uint8_t* id_out)
{
    RTX_CHECK(id_out, RTX_EINVAL, "comm_unique_id: NULL argument");
    RTX_TRY(load_api());
    NcclId id;
    RTX_NCCL(g_api.GetUniqueId(&id));
    memcpy(id_out, id.internal, RTX_COMM_ID_BYTES);
    return RTX_OK;
}

int rtx_comm_init(const uint8_t* id, int32_t rank, int32_t world, rtx_comm** out)
{
    RTX_CHECK(id && out && world >= 1 && rank >= 0 && rank < world, RTX_EINVAL, "comm_init: bad arguments (rank %d of %d)", rank, world);
    RTX_TRY(load_api());
    NcclId nid;
    memcpy(nid.internal, id, RTX_COMM_ID_BYTES);
    rtx_comm* c = new rtx_comm();
    c->rank = rank;
    c->world = world;
    int r = g_api.CommInitRank(&c->comm, world, nid, rank);
    if (r != 0) {
        rtx_set_error("comm_init: ncclCommInitRank -> %s", g_api.GetErrorString(r));
        delete c;
        return RTX_EHIP;
    }
    *out = c;
    return RTX_OK;
}

int rtx_comm_destroy(rtx_comm* c)
{
    if (!c) return RTX_OK;
    if (c->comm) (void)g_api.CommDestroy(c->comm);
    delete c;
    return RTX_OK;
}

int rtx_comm_allreduce(rtx_comm* c, void* buf, int64_t n, int32_t dtype, void* stream)
{
    RTX_CHECK(c && buf && n >= 0, RTX_EINVAL, "comm_allreduce: bad arguments");
    RTX_CHECK(dtype == RTX_FP32 || dtype == RTX_BF16, RTX_EINVAL, "comm_allreduce: dtype must be RTX_FP32 or RTX_BF16");
    size_t eb;
    const int dt = nccl_dtype(dtype, &eb);
    if (n == 0) return RTX_OK;
    RTX_NCCL(g_api.AllReduce(buf, buf, (size_t)n, dt, 0, c->comm, (hipStream_t)stream));
    return RTX_OK;
}

int rtx_comm_reduce_scatter(rtx_comm* c, void* buf, int64_t n_total, int32_t dtype, void* stream)
{
    RTX_CHECK(c && buf && n_total >= 0, RTX_EINVAL, "comm_reduce_scatter: bad arguments");
    RTX_CHECK(dtype == RTX_FP32 || dtype == RTX_BF16, RTX_EINVAL, "comm_reduce_scatter: dtype must be RTX_FP32 or RTX_BF16");
    RTX_CHECK(n_total % c->world == 0, RTX_EINVAL, "comm_reduce_scatter: %lld elements do not split over %d ranks", (long long)n_total, c->world);
    size_t eb;
    const int dt = nccl_dtype(dtype, &eb);
    const size_t per = (size_t)n_total / c->world;
    if (per == 0) return RTX_OK;
    // in place: block `rank` of the buffer receives the sum (recvbuff == sendbuff + rank * recvcount)
    RTX_NCCL(g_api.ReduceScatter(buf, (char*)buf + (size_t)c->rank * per * eb, per, dt, 0, c->comm, (hipStream_t)stream));
    return RTX_OK;
}

int rtx_comm_allgather(rtx_comm* c, void* buf, int64_t bytes_total, void* stream)
{
    RTX_CHECK(c && buf && bytes_total >= 0, RTX_EINVAL, "comm_allgather: bad arguments");
    RTX_CHECK(bytes_total % c->world == 0, RTX_EINVAL, "comm_allgather: %lld bytes do not split over %d ranks", (long long)bytes_total, c->world);
    const size_t per = (size_t)bytes_total / c->world;
    if (per == 0) return RTX_OK;
    // in place: block `rank` is this rank's contribution (sendbuff == recvbuff + rank * sendcount); bytes travel as int8
    RTX_NCCL(g_api.AllGather((char*)buf + (size_t)c->rank * per, buf, per, 0, c->comm, (hipStream_t)stream));
    return RTX_OK;
}

int rtx_comm_allreduce_many(rtx_comm* c, void* const* bufs, const int64_t* counts, int32_t n_bufs, int32_t dtype, void* stream)
{
    RTX_CHECK(c && bufs && counts && n_bufs >= 0, RTX_EINVAL, "comm_allreduce_many: bad arguments");
    RTX_CHECK(dtype == RTX_FP32 || dtype == RTX_BF16, RTX_EINVAL, "comm_allreduce_many: dtype must be RTX_FP32 or RTX_BF16");
    size_t eb;
    const int dt = nccl_dtype(dtype, &eb);
    RTX_NCCL(g_api.GroupStart());
    for (int i = 0; i < n_bufs; ++i)
        if (counts[i] > 0) {
            int r = g_api.AllReduce(bufs[i], bufs[i], (size_t)counts[i], dt, 0, c->comm, (hipStream_t)stream);
            if (r != 0) {
                (void)g_api.GroupEnd();
                rtx_set_error("comm_allreduce_many: ncclAllReduce -> %s", g_api.GetErrorString(r));
                return RTX_EHIP;
            }
        }
    RTX_NCCL(g_api.GroupEnd());
    return RTX_OK;
}

int rtx_comm_group_start(rtx_comm* c)
{
    RTX_CHECK(c, RTX_EINVAL, "comm is NULL");
    RTX_NCCL(g_api.GroupStart());
    return RTX_OK;
}

int rtx_comm_group_end(rtx_comm* c)
{
    RTX_CHECK(c, RTX_EINVAL, "comm is NULL");
    RTX_NCCL(g_api.GroupEnd());
    return RTX_OK;
}

int rtx_comm_rank(const rtx_comm* c, int32_t* rank, int32_t* world)
{
    RTX_CHECK(c, RTX_EINVAL, "comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return RTX_OK;
}

}  // extern "C"
