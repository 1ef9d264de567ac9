// comm.hip -- RCCL over xGMI, one process per GPU.  The data matrix is row-sharded;
// only the small k x p / k x k Gram products and packed scalars are all-reduced
// (sum or max), in place, on the context's stream.  The reference has no
// communication layer at all (single process).
//
// librccl is loaded lazily with dlopen so the single-GPU path has no RCCL dependency.
#include <dlfcn.h>

#include "aa_internal.h"

namespace aa {

typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { kNcclSum = 0, kNcclMax = 2, kNcclFloat64 = 8 };

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;

static int rccl_load()
{
    if (g_rccl.handle) return AA_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    AA_REQUIRE(h != nullptr, AA_ERR_COMM, "cannot load librccl: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    AA_REQUIRE(g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce,
               AA_ERR_COMM, "librccl lacks a required symbol");
    g_rccl.handle = h;
    return AA_OK;
}

struct Comm {
    ncclComm_t comm = nullptr;
};

#define AA_CHECK_NCCL(expr)                                                              \
    do {                                                                                 \
        ncclResult_t r__ = (expr);                                                       \
        if (r__ != 0) {                                                                  \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                      \
                      g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "rccl error"); \
            return AA_ERR_COMM;                                                          \
        }                                                                                \
    } while (0)

int comm_unique_id(void *id128)
{
    AA_CHECK(rccl_load());
    ncclUniqueId id;
    AA_CHECK_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return AA_OK;
}

int comm_init(Ctx *c, const void *id128, int rank, int world)
{
    AA_REQUIRE(world >= 1 && rank >= 0 && rank < world, AA_ERR_ARG, "bad rank/world %d/%d", rank, world);
    c->rank = rank;
    c->world = world;
    // AA_FORCE_RCCL=1: build a 1-rank communicator too (exercises the RCCL path on a
    // single-GPU box; every all-reduce then really goes through ncclAllReduce)
    const char *force = getenv("AA_FORCE_RCCL");
    c->force_comm = force && force[0] == '1';
    if (world == 1 && !c->force_comm) return AA_OK;
    AA_CHECK(rccl_load());
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    c->comm = new Comm();
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_CHECK_NCCL(g_rccl.CommInitRank(&c->comm->comm, world, id, rank));
    return AA_OK;
}

void comm_destroy(Ctx *c)
{
    if (c->comm) {
        if (c->comm->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm->comm);
        delete c->comm;
        c->comm = nullptr;
    }
}

int comm_allreduce(Ctx *c, double *dev, long count, int op)
{
    if ((c->world <= 1 && !c->force_comm) || !c->comm) return AA_OK;
    AA_CHECK_NCCL(g_rccl.AllReduce(dev, dev, (size_t)count, kNcclFloat64, op ? kNcclMax : kNcclSum,
                                   c->comm->comm, c->stream));
    return AA_OK;
}

}  // namespace aa
