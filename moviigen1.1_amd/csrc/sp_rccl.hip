// Sequence-parallel / block-shard collectives on an RCCL communicator, behind the C-ABI (SURVEY.md §8(b):
// `sp_all_to_all_4d(comm, ...)`, `sp_all_gather`, `shard_all_gather` on an ncclComm_t).
//
// The reference reaches these through Python libraries that wrap NCCL: xfuser's xFuserLongContextAttention /
// get_sp_group().all_gather (wan/distributed/xdit_context_parallel.py:148,185-190), FastVideo's all_to_all_4D /
// all_gather (scripts/train/model/model_seq.py:232-234,256,780) and torch FSDP's parameter all-gather
// (wan/distributed/fsdp.py:20-31).  Here they are plain C entry points that enqueue on the caller's HIP stream:
// the pack / unpack kernels of sp_exchange.hip around grouped ncclSend/ncclRecv pairs (on the xGMI mesh every
// peer pair has its own link, so an all-to-all is P-1 concurrent point-to-point transfers), and ncclAllGather.
//
// librccl is bound at RUN time (dlopen, preferring a copy the process has already loaded — PyTorch ships its own —
// so two RCCL instances never coexist); libmoviigen_hip.so itself has no link-time dependency on it and every
// other entry point works without it.  When RCCL cannot be found these functions return MG_ERR_UNAVAILABLE.
#include <dlfcn.h>
#include <string.h>

#include "common.h"
#include "../../include/moviigen_hip.h"

namespace {
struct uid128 { char b[128]; };     // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
struct rccl_api {
    void* lib = nullptr;
    int (*GetUniqueId)(void* uid) = nullptr;
    int (*CommInitRank)(void** comm, int nranks, uid128 id, int rank) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*Send)(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t st) = nullptr;
    int (*Recv)(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t st) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st) = nullptr;
    int (*CommCount)(void* comm, int* n) = nullptr;
    int (*CommUserRank)(void* comm, int* r) = nullptr;
    bool ok = false;
};
constexpr int kNcclInt8 = 0;   // ncclInt8 / ncclChar: payloads are moved as bytes

rccl_api& api() {
    static rccl_api a;
    static bool tried = false;
    if (tried) return a;
    tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)          // a copy that is already mapped wins (torch's bundled librccl.so)
        if ((a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
    if (!a.lib)
        for (const char* n : names)
            if ((a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!a.lib) return a;
#define BIND(field, sym) *(void**)(&a.field) = dlsym(a.lib, sym)
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(Send, "ncclSend");
    BIND(Recv, "ncclRecv");
    BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd");
    BIND(AllGather, "ncclAllGather");
    BIND(CommCount, "ncclCommCount");
    BIND(CommUserRank, "ncclCommUserRank");
#undef BIND
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.Send && a.Recv && a.GroupStart && a.GroupEnd &&
           a.AllGather && a.CommCount && a.CommUserRank;
    return a;
}

int comm_shape(void* comm, int& P, int& rank) {
    rccl_api& a = api();
    if (!a.ok) return MG_ERR_UNAVAILABLE;
    if (a.CommCount(comm, &P) || a.CommUserRank(comm, &rank) || P <= 0) return MG_ERR_COMM;
    return MG_OK;
}

// all-to-all of equal byte chunks: chunk p of `send` goes to rank p, chunk p of `recv` comes from rank p
int a2a_bytes(void* comm, const char* send, char* recv, size_t bytes, int P, hipStream_t st) {
    rccl_api& a = api();
    if (a.GroupStart()) return MG_ERR_COMM;
    int rc = 0;
    for (int p = 0; p < P; ++p) {
        rc |= a.Send(send + (size_t)p * bytes, bytes, kNcclInt8, p, comm, st);
        rc |= a.Recv(recv + (size_t)p * bytes, bytes, kNcclInt8, p, comm, st);
    }
    rc |= a.GroupEnd();
    return rc ? MG_ERR_COMM : MG_OK;
}
}  // namespace

extern "C" int mg_comm_unique_id(void* id128) {
    if (!id128) return MG_ERR_ARG;
    rccl_api& a = api();
    if (!a.ok) return MG_ERR_UNAVAILABLE;
    return a.GetUniqueId(id128) ? MG_ERR_COMM : MG_OK;
}

extern "C" int mg_comm_create(const void* id128, int nranks, int rank, void** comm) {
    if (!id128 || !comm) return MG_ERR_ARG;
    if (nranks <= 0 || rank < 0 || rank >= nranks) return MG_ERR_SHAPE;
    rccl_api& a = api();
    if (!a.ok) return MG_ERR_UNAVAILABLE;
    uid128 id;
    memcpy(id.b, id128, sizeof id.b);
    return a.CommInitRank(comm, nranks, id, rank) ? MG_ERR_COMM : MG_OK;
}

extern "C" int mg_comm_destroy(void* comm) {
    if (!comm) return MG_ERR_ARG;
    rccl_api& a = api();
    if (!a.ok) return MG_ERR_UNAVAILABLE;
    return a.CommDestroy(comm) ? MG_ERR_COMM : MG_OK;
}

extern "C" int mg_sp_all_to_all(void* comm, const void* send, void* recv, int64_t bytes_per_peer, void* stream) {
    if (!comm || !send || !recv) return MG_ERR_ARG;
    if (bytes_per_peer < 0) return MG_ERR_SHAPE;
    int P, rank;
    if (int rc = comm_shape(comm, P, rank)) return rc;
    if (bytes_per_peer == 0) return MG_OK;
    return a2a_bytes(comm, (const char*)send, (char*)recv, (size_t)bytes_per_peer, P, (hipStream_t)stream);
}

extern "C" int mg_sp_all_to_all_4d_bf16(void* comm, const uint16_t* x, int64_t ldx, int64_t rows, int heads, int head_dim,
                                        int seq_to_head, uint16_t* out, int64_t ldo, uint16_t* workspace, void* stream) {
    if (!comm || !x || !out || !workspace) return MG_ERR_ARG;
    int P, rank;
    if (int rc = comm_shape(comm, P, rank)) return rc;
    if (heads <= 0 || head_dim <= 0 || heads % P || rows < 0 || (seq_to_head ? 0 : rows % P)) return MG_ERR_SHAPE;
    const int nl = (heads / P) * head_dim;               // columns of one rank's head slice
    hipStream_t st = (hipStream_t)stream;
    if (seq_to_head) {
        // x [Lloc = rows][heads*hd] -> out [P*Lloc][nl]: pack block p = columns [p*nl, (p+1)*nl) as [p][Lloc][nl], exchange
        const int64_t Lloc = rows;
        if (ldo != nl) return MG_ERR_SHAPE;              // the receive side must be the contiguous [P][Lloc][nl] image
        int rc = mg_sp_copy_blocks_bf16(x, nl, ldx, workspace, Lloc * nl, nl, P, Lloc, nl, stream);
        if (rc) return rc;
        return a2a_bytes(comm, (const char*)workspace, (char*)out, (size_t)Lloc * nl * 2, P, st);
    }
    // x [P*Lloc = rows][nl] (contiguous) -> out [Lloc][heads*hd]: exchange row blocks, scatter block p to columns p*nl..
    const int64_t Lloc = rows / P;
    if (ldx != nl) return MG_ERR_SHAPE;
    int rc = a2a_bytes(comm, (const char*)x, (char*)workspace, (size_t)Lloc * nl * 2, P, st);
    if (rc) return rc;
    return mg_sp_copy_blocks_bf16(workspace, Lloc * nl, nl, out, nl, ldo, P, Lloc, nl, stream);
}

extern "C" int mg_sp_all_gather(void* comm, const void* send, void* recv, int64_t bytes, void* stream) {
    if (!comm || !send || !recv) return MG_ERR_ARG;
    if (bytes < 0) return MG_ERR_SHAPE;
    rccl_api& a = api();
    if (!a.ok) return MG_ERR_UNAVAILABLE;
    if (bytes == 0) return MG_OK;
    return a.AllGather(send, recv, (size_t)bytes, kNcclInt8, comm, (hipStream_t)stream) ? MG_ERR_COMM : MG_OK;
}

extern "C" int mg_shard_all_gather(void* comm, const void* shard, void* full, int64_t shard_bytes, void* stream) {
    return mg_sp_all_gather(comm, shard, full, shard_bytes, stream);
}
