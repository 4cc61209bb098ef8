// Data-parallel gradient exchange behind the C ABI (mapdit_allreduce_bucket; section 8e: reduce-scatter /
// all-gather of flat gradient buckets over xGMI).  For a host program that is not PyTorch: the Python integration keeps using
// torch.distributed (backend "nccl" = RCCL), which owns its own communicator.
//
// RCCL is bound at run time: dlopen of the copy already in the process if there is one (PyTorch ships its own librccl.so), else of
// the ROCm installation's - the library itself has no link-time dependency on RCCL and loads on machines without it.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {

typedef struct { char internal[128]; } nccl_unique_id;      // NCCL_UNIQUE_ID_BYTES
typedef void* nccl_comm;
enum { NCCL_SUM = 0, NCCL_FLOAT32 = 7, NCCL_BF16 = 9 };

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    char why[256] = "?";                                    // dlerror() of the last failing dlopen, captured right after it
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((x.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;         // the copy the process already holds
        if (!x.h)
            for (const char* n : names) {
                if ((x.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
                const char* e = dlerror();                                // (null once read, and after the RTLD_NOLOAD probes above)
                if (e) { strncpy(x.why, e, sizeof(x.why) - 1); x.why[sizeof(x.why) - 1] = 0; }
            }
        if (!x.h) return x;
#define BIND(field, sym) *(void**)(&x.field) = dlsym(x.h, sym)
        BIND(GetUniqueId, "ncclGetUniqueId");
        BIND(CommInitRank, "ncclCommInitRank");
        BIND(CommDestroy, "ncclCommDestroy");
        BIND(AllReduce, "ncclAllReduce");
        BIND(ReduceScatter, "ncclReduceScatter");
        BIND(AllGather, "ncclAllGather");
        BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.ReduceScatter && x.AllGather;
        return x;
    }();
    return r;
}

int rccl_ready() {
    if (!rccl().ok) {
        mapdit_set_error("comm: RCCL (librccl.so) is not available in this process: %s", rccl().h ? "symbols missing" : rccl().why);
        return MAPDIT_ERR_HIP;
    }
    return MAPDIT_OK;
}

#define RCCL_CHECK(call, what)                                                                             \
    do {                                                                                                   \
        int rc_ = (call);                                                                                  \
        if (rc_ != 0) {                                                                                    \
            mapdit_set_error("comm: %s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc_) : "?"); \
            return MAPDIT_ERR_HIP;                                                                         \
        }                                                                                                  \
    } while (0)

}  // namespace

// A communicator belongs to the HIP device that was current in mapdit_comm_create (ncclCommInitRank binds it); the collectives
// check that the calling thread is still on that device - a buffer or stream of another device would fail inside RCCL, or hang.
struct mapdit_comm {
    nccl_comm comm = nullptr;
    int rank = 0, world = 1, device = -1;
};
static int comm_device_ok(const mapdit_comm* c, const char* what) {
    int dev = -1;
    MD_CHECK(hipGetDevice(&dev) == hipSuccess, "%s: hipGetDevice failed", what);
    MD_CHECK(dev == c->device, "%s: the communicator was created on HIP device %d, the calling thread is on device %d", what, c->device, dev);
    return MAPDIT_OK;
}

extern "C" int mapdit_comm_unique_id(void* id128) {
    MD_CHECK(id128, "comm_unique_id: null argument");
    int rc = rccl_ready();
    if (rc != MAPDIT_OK) return rc;
    nccl_unique_id id;
    RCCL_CHECK(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id128, &id, sizeof(id));
    return MAPDIT_OK;
}

extern "C" int mapdit_comm_create(const void* id128, int rank, int world, mapdit_comm_t** out) {
    MD_CHECK(id128 && out && world >= 1 && rank >= 0 && rank < world, "comm_create: bad argument (rank %d of %d)", rank, world);
    int rc = rccl_ready();
    if (rc != MAPDIT_OK) return rc;
    nccl_unique_id id;
    memcpy(&id, id128, sizeof(id));
    mapdit_comm* c = new mapdit_comm();
    c->rank = rank;
    c->world = world;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; mapdit_set_error("comm_create: hipGetDevice failed"); return MAPDIT_ERR_HIP; }
    int e = rccl().CommInitRank(&c->comm, world, id, rank);
    if (e != 0) {
        delete c;
        mapdit_set_error("comm_create: ncclCommInitRank failed: %s", rccl().GetErrorString ? rccl().GetErrorString(e) : "?");
        return MAPDIT_ERR_HIP;
    }
    *out = c;
    return MAPDIT_OK;
}

extern "C" void mapdit_comm_destroy(mapdit_comm_t* c) {
    if (!c) return;
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    delete c;
}

extern "C" int mapdit_allreduce_bucket(mapdit_comm_t* c, float* buf, long count, void* stream) {
    MD_CHECK(c && buf && count > 0, "allreduce_bucket: null/empty argument");
    if (int rc = comm_device_ok(c, "allreduce_bucket")) return rc;
    RCCL_CHECK(rccl().AllReduce(buf, buf, (size_t)count, NCCL_FLOAT32, NCCL_SUM, c->comm, (hipStream_t)stream), "ncclAllReduce");
    return MAPDIT_OK;
}

extern "C" int mapdit_reduce_scatter_bucket(mapdit_comm_t* c, float* buf, long count, void* stream) {
    MD_CHECK(c && buf && count > 0 && count % c->world == 0, "reduce_scatter_bucket: count=%ld must be a positive multiple of the world size", count);
    if (int rc = comm_device_ok(c, "reduce_scatter_bucket")) return rc;
    const size_t part = (size_t)count / c->world;                       // in place: rank r receives the sum of part r at buf + r*part
    RCCL_CHECK(rccl().ReduceScatter(buf, buf + (size_t)c->rank * part, part, NCCL_FLOAT32, NCCL_SUM, c->comm, (hipStream_t)stream),
               "ncclReduceScatter");
    return MAPDIT_OK;
}

extern "C" int mapdit_allgather_bucket(mapdit_comm_t* c, float* buf, long count, void* stream) {
    MD_CHECK(c && buf && count > 0 && count % c->world == 0, "allgather_bucket: count=%ld must be a positive multiple of the world size", count);
    if (int rc = comm_device_ok(c, "allgather_bucket")) return rc;
    const size_t part = (size_t)count / c->world;                       // in place: rank r contributes buf + r*part
    RCCL_CHECK(rccl().AllGather(buf + (size_t)c->rank * part, buf, part, NCCL_FLOAT32, c->comm, (hipStream_t)stream), "ncclAllGather");
    return MAPDIT_OK;
}
