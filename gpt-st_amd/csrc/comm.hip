// Communication entry points of the C ABI: a thin wrapper over RCCL (the ROCm collective library, xGMI on an MI355X
// node) with an explicit stream, so that a collective can sit INSIDE a captured hipGraph next to the kernels — torch.distributed's
// all_reduce cannot be called from a C-ABI consumer.  The reference has no distributed code (GPTST.py hard-codes 'cuda:0'); these are
// what the data-parallel gradient exchange (one all-reduce of [flat gradient | statistics], dist.py) and the node-sharded cluster
// aggregations (shard.py: R+2 all-reduces of (B,T,HS,C) per cap forward, one per backward) map to.
// RCCL is bound at run time (dlopen of the copy the process already holds — torch ships one — or librccl.so.1): the library has no
// link-time dependency on it, and a box without RCCL only loses these four entry points (GPTST_ECOMM).
#include "common.h"
#include <dlfcn.h>

#define GPTST_ECOMM (-4)

namespace {
struct Uid { char internal[128]; };                        // == ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
typedef int (*GetUniqueId_t)(Uid*);
typedef int (*CommInitRank_t)(void**, int, Uid, int);
typedef int (*AllReduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroy_t)(void*);
typedef int (*CommCount_t)(void*, int*);
typedef int (*AllGather_t)(const void*, void*, size_t, int, void*, hipStream_t);
struct Api { void* lib; GetUniqueId_t uid; CommInitRank_t init; AllReduce_t allreduce; CommDestroy_t destroy; CommCount_t count; AllGather_t allgather; };
Api g_api = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
struct Comm { void* nccl; int rank, world; };               // what a gptst communicator handle points to

bool bind() {
    if (g_api.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }      // the copy already in the process
    if (!h) for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return false;
    Api a{h, (GetUniqueId_t)dlsym(h, "ncclGetUniqueId"), (CommInitRank_t)dlsym(h, "ncclCommInitRank"),
          (AllReduce_t)dlsym(h, "ncclAllReduce"), (CommDestroy_t)dlsym(h, "ncclCommDestroy"), (CommCount_t)dlsym(h, "ncclCommCount"),
          (AllGather_t)dlsym(h, "ncclAllGather")};
    if (!a.uid || !a.init || !a.allreduce || !a.destroy) return false;
    g_api = a;
    return true;
}
}  // namespace

// GPTST_OK when RCCL can be bound in this process (library found, the four mandatory symbols present), GPTST_ECOMM otherwise.  No side effect
// beyond the dlopen: the probe every rank runs before a communicator is formed (ncclGetUniqueId starts a bootstrap root — a listening socket
// and a thread — per call, which a mere probe must not leave behind: ADVICE r04).
extern "C" int gptst_comm_available(void) { return bind() ? GPTST_OK : GPTST_ECOMM; }

// out: 128 bytes (ncclUniqueId) created on ONE rank and handed to every rank of the job by the caller (file, socket, torch.distributed)
extern "C" int gptst_comm_unique_id(void* out) {
    if (!out) return GPTST_EARG;
    if (!bind()) return GPTST_ECOMM;
    const int rc = g_api.uid((Uid*)out);
    return rc == 0 ? GPTST_OK : 1000 + rc;
}

// A communicator of `world` ranks in which this process (its current device: hipSetDevice first) is `rank`; unique_id: the 128 bytes of
// gptst_comm_unique_id, created by one rank of THIS communicator.  *comm_out receives the handle every collective takes.  A process may hold
// several (r04: e.g. the row and the column of a data-parallel x node-shard mesh, SURVEY 8(e) "Combination"); RCCL serialises collectives of
// different communicators that share a stream in enqueue order.
extern "C" int gptst_comm_init(int rank, int world, const void* unique_id, void** comm_out) {
    if (!unique_id || !comm_out || world <= 0 || rank < 0 || rank >= world) return GPTST_EARG;
    if (!bind()) return GPTST_ECOMM;
    Uid u = *(const Uid*)unique_id;
    void* nc = nullptr;
    const int rc = g_api.init(&nc, world, u, rank);
    if (rc != 0) return 1000 + rc;
    *comm_out = new Comm{nc, rank, world};
    return GPTST_OK;
}

// in-place sum over the ranks of buf[0..n) (fp32), enqueued on `stream` (capturable); dtype 7 = ncclFloat32, op 0 = ncclSum
extern "C" int gptst_allreduce_f32(void* comm, float* buf, long n, void* stream) {
    if (!buf || n <= 0) return GPTST_EARG;
    if (!comm) return GPTST_ECOMM;
    const int rc = g_api.allreduce(buf, buf, (size_t)n, 7, 0, ((Comm*)comm)->nccl, (hipStream_t)stream);
    return rc == 0 ? GPTST_OK : 1000 + rc;
}

// recv[r*n .. (r+1)*n) <- rank r's send[0..n) (32-bit words: labels, indices), enqueued on `stream` (capturable); dtype 2 = ncclInt32.
// send may be recv + rank*n (in place).  GPTST_ECOMM when the communicator or ncclAllGather is missing (callers fall back to an all-reduce).
extern "C" int gptst_allgather_i32(void* comm, const int* send, int* recv, long n, void* stream) {
    if (!send || !recv || n <= 0) return GPTST_EARG;
    if (!comm || !g_api.allgather) return GPTST_ECOMM;
    const int rc = g_api.allgather(send, recv, (size_t)n, 2, ((Comm*)comm)->nccl, (hipStream_t)stream);
    return rc == 0 ? GPTST_OK : 1000 + rc;
}

// number of ranks RCCL itself reports for the communicator (ncclCommCount) -> *out; lets a launcher verify that the job's collectives really
// span the ranks it started (bench.py prints it as rccl_ranks)
extern "C" int gptst_comm_count(void* comm, int* out) {
    if (!out) return GPTST_EARG;
    if (!comm || !g_api.count) return GPTST_ECOMM;
    const int rc = g_api.count(((Comm*)comm)->nccl, out);
    return rc == 0 ? GPTST_OK : 1000 + rc;
}

extern "C" int gptst_comm_destroy(void* comm) {
    if (!comm) return GPTST_OK;
    Comm* c = (Comm*)comm;
    const int rc = g_api.destroy(c->nccl);
    delete c;
    return rc == 0 ? GPTST_OK : 1000 + rc;
}
