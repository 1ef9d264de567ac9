// comm.hip -- RCCL over xGMI, one process per GPU.  The data matrix is row-sharded;
// only the small k x p / k x k Gram products and packed scalars are all-reduced
// (sum or max), in place, on the context's stream.  The reference has no
// communication layer at all (single process).
//
// librccl is loaded lazily with dlopen so the single-GPU path has no RCCL dependency.
//
// Second transport (round 4, SURVEY.md section 5, aa_ctx_p2p_*): a ONE-SHOT PEER-TO-PEER all-reduce
// for the messages of this path, which are at most 1 MiB and latency-bound.  Every rank owns a
// receive buffer that its peers have mapped (hipIpcGetMemHandle / hipIpcOpenMemHandle); an
// all-reduce is ONE kernel per rank: each block stores its chunk of the local vector into its slot
// of every peer's buffer (stores over xGMI), fences, raises one flag per (rank, block) on every
// peer, waits for the flags of all ranks for ITS chunk, and reduces the world slots IN RANK ORDER
// -- identical bits on every rank, the property the control decisions of the solver rely on --
// no ring, no intermediate hops, no second launch.  Two banks alternate: a rank enters all-reduce
// e + 2 only after it has seen every peer's flag of e + 1, which a peer raises at the start of its
// kernel e + 1, i.e. after its kernel e -- the last reader of bank (e mod 2) there -- has finished.
// Needs no RCCL at all (ranks may even share one GPU, which RCCL refuses: the functional test of
// tests/test_gpu_configs.py runs two ranks on the one GPU of the development box).
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


// ------------------------------------------------------------------ one-shot peer-to-peer all-reduce
#define P2P_MAX_WORLD 8
#define P2P_MAX_BLOCKS 64
#define P2P_SLOT_BYTES ((1u << 20) + (1u << 16))    // per rank and bank: k x p = 32 x 4096 float64 and the riders in its tail (pack_comm)
#define P2P_FLAG_WORDS (2 * P2P_MAX_WORLD * P2P_MAX_BLOCKS)

struct P2P {
    int rank = 0, world = 1;
    unsigned char *mine = nullptr;                 // [2 banks][world][P2P_SLOT_BYTES] | flags
    unsigned char *peer[P2P_MAX_WORLD] = {nullptr};
    bool opened[P2P_MAX_WORLD] = {false};
    unsigned epoch = 0;
    int *err_host = nullptr;                       // mapped host word: a kernel gave up waiting
    bool ready = false;
};

struct P2PArgs {
    unsigned char *peer[P2P_MAX_WORLD];
    int rank, world;
    unsigned epoch;
    int op;
};

static inline size_t p2p_bytes(int world)
{
    return (size_t)2 * world * P2P_SLOT_BYTES + (size_t)P2P_FLAG_WORDS * sizeof(unsigned);
}

__device__ __forceinline__ unsigned *p2p_flags(unsigned char *base, int world)
{
    return reinterpret_cast<unsigned *>(base + (size_t)2 * world * P2P_SLOT_BYTES);
}

// one launch per all-reduce of `count` doubles (count * 8 <= P2P_SLOT_BYTES), in place on `data`
__global__ __launch_bounds__(256) void k_p2p_allreduce(P2PArgs a, double *__restrict__ data, long count,
                                                       int *__restrict__ err)
{
    const int t = threadIdx.x, b = blockIdx.x, nb = gridDim.x;
    const int bank = (int)(a.epoch & 1u);
    // chunk of this block, in pairs of doubles (16-byte stores)
    const long pairs = (count + 1) / 2, ppb = (pairs + nb - 1) / nb;
    long lo = 2 * ppb * b, hi = lo + 2 * ppb;
    if (lo > count) lo = count;
    if (hi > count) hi = count;
    // 1. my chunk into my slot of every rank's buffer (my own included: one code path, one order)
    for (int q = 0; q < a.world; ++q) {
        double *dst = reinterpret_cast<double *>(a.peer[q] + ((size_t)bank * a.world + a.rank) * P2P_SLOT_BYTES);
        for (long i = lo + t; i < hi; i += 256) __builtin_nontemporal_store(data[i], &dst[i]);
    }
    __threadfence_system();
    __syncthreads();
    if (t < a.world) {
        unsigned *f = p2p_flags(a.peer[t], a.world) + ((size_t)bank * P2P_MAX_WORLD + a.rank) * P2P_MAX_BLOCKS + b;
        __hip_atomic_store(f, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. wait for every rank's chunk b
    if (t < a.world) {
        const unsigned *f = p2p_flags(a.peer[a.rank], a.world) + ((size_t)bank * P2P_MAX_WORLD + t) * P2P_MAX_BLOCKS + b;
        const unsigned long long t0 = wall_clock64();                 // 100 MHz
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 1000000000ull) {                // 10 s: a peer is gone
                if (err) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    __threadfence_system();
    // 3. the world slots in rank order
    const double *mine = reinterpret_cast<const double *>(a.peer[a.rank] + (size_t)bank * a.world * P2P_SLOT_BYTES);
    const size_t slot = P2P_SLOT_BYTES / sizeof(double);
    for (long i = lo + t; i < hi; i += 256) {
        double s = __builtin_nontemporal_load(&mine[i]);
        for (int r = 1; r < a.world; ++r) {
            const double v = __builtin_nontemporal_load(&mine[(size_t)r * slot + i]);
            s = a.op ? fmax(s, v) : s + v;
        }
        data[i] = s;
    }
}

int p2p_export(Ctx *c, int world, void *handle64)
{
    AA_REQUIRE(world >= 1 && world <= P2P_MAX_WORLD, AA_ERR_ARG, "p2p: world %d out of range (1..%d)", world, P2P_MAX_WORLD);
    AA_CHECK_HIP(hipSetDevice(c->device));
    if (!c->p2p) c->p2p = new P2P();
    P2P *p = c->p2p;
    if (!p->mine) {
        void *buf = nullptr;
        // uncached device memory: peers' stores and the flag polls must not meet a stale L2 line
        hipError_t e = hipExtMallocWithFlags(&buf, p2p_bytes(world), hipDeviceMallocUncached);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            AA_CHECK_HIP(hipMalloc(&buf, p2p_bytes(world)));
        }
        AA_CHECK_HIP(hipMemsetAsync(buf, 0, p2p_bytes(world), nullptr));
        AA_CHECK_HIP(hipStreamSynchronize(nullptr));
        p->mine = reinterpret_cast<unsigned char *>(buf);
        AA_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&p->err_host), sizeof(int), hipHostMallocMapped));
        *p->err_host = 0;
    }
    p->world = world;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipIpcMemHandle_t h;
    AA_CHECK_HIP(hipIpcGetMemHandle(&h, p->mine));
    memcpy(handle64, &h, 64);
    return AA_OK;
}

int p2p_init(Ctx *c, const void *handles, int rank, int world)
{
    AA_REQUIRE(c->p2p && c->p2p->mine && c->p2p->world == world, AA_ERR_STATE, "aa_ctx_p2p_export first");
    AA_REQUIRE(rank >= 0 && rank < world, AA_ERR_ARG, "bad rank/world %d/%d", rank, world);
    AA_CHECK_HIP(hipSetDevice(c->device));
    P2P *p = c->p2p;
    p->rank = rank;
    for (int q = 0; q < world; ++q) {
        if (q == rank) {
            p->peer[q] = p->mine;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, reinterpret_cast<const unsigned char *>(handles) + (size_t)64 * q, 64);
        void *ptr = nullptr;
        AA_CHECK_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        p->peer[q] = reinterpret_cast<unsigned char *>(ptr);
        p->opened[q] = true;
    }
    p->ready = true;
    c->rank = rank;
    c->world = world;
    c->force_comm = world == 1;                    // a one-rank run still goes through the multi-rank code path
    return AA_OK;
}

static void p2p_destroy(Ctx *c)
{
    P2P *p = c->p2p;
    if (!p) return;
    for (int q = 0; q < P2P_MAX_WORLD; ++q)
        if (p->opened[q] && p->peer[q]) (void)hipIpcCloseMemHandle(p->peer[q]);
    if (p->mine) (void)hipFree(p->mine);
    if (p->err_host) (void)hipHostFree(p->err_host);
    delete p;
    c->p2p = nullptr;
}

int p2p_check(Ctx *c)
{
    if (c->p2p && c->p2p->err_host && *c->p2p->err_host) {
        set_error("peer-to-peer all-reduce: a rank waited 10 s for a peer's data (a peer has stopped?)");
        return AA_ERR_COMM;
    }
    return AA_OK;
}

static int p2p_allreduce(Ctx *c, double *dev, long count, int op)
{
    P2P *p = c->p2p;
    const long per = (long)(P2P_SLOT_BYTES / sizeof(double));
    for (long off = 0; off < count; off += per) {
        const long cnt = count - off < per ? count - off : per;
        P2PArgs a;
        for (int q = 0; q < P2P_MAX_WORLD; ++q) a.peer[q] = q < p->world ? p->peer[q] : nullptr;
        a.rank = p->rank;
        a.world = p->world;
        a.epoch = ++p->epoch;
        a.op = op;
        long nb = (cnt * (long)sizeof(double) + 16383) / 16384;      // >= 16 KiB per block
        if (nb < 1) nb = 1;
        if (nb > P2P_MAX_BLOCKS) nb = P2P_MAX_BLOCKS;
        int *err_dev = nullptr;
        AA_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&err_dev), p->err_host, 0));
        hipLaunchKernelGGL(k_p2p_allreduce, dim3((unsigned)nb), dim3(256), 0, c->stream, a, dev + off, cnt, err_dev);
        AA_CHECK_HIP(hipGetLastError());
    }
    return AA_OK;
}

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
    p2p_destroy(c);
    if (c->comm) {
        if (c->comm->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm->comm);
        delete c->comm;
        c->comm = nullptr;
    }
}

int comm_allreduce(Ctx *c, double *dev, long count, int op)
{
    if (c->p2p && c->p2p->ready) return p2p_allreduce(c, dev, count, op);
    if ((c->world <= 1 && !c->force_comm) || !c->comm) return AA_OK;
    AA_CHECK_NCCL(g_rccl.AllReduce(dev, dev, (size_t)count, kNcclFloat64, op ? kNcclMax : kNcclSum,
                                   c->comm->comm, c->stream));
    return AA_OK;
}

}  // namespace aa
