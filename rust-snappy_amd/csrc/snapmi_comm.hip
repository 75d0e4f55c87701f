// snapmi_comm.hip -- the one exchange step of the path: the framed
// parts that the ranks of a multi-GPU job produced are gathered on one rank,
// over RCCL (xGMI between the GPUs of a node), from the C ABI - a host
// language needs no RCCL binding of its own to assemble one framed stream
// (reference src/write.rs:165-192 writes it chunk after chunk; here rank r
// frames a contiguous range of chunks and the parts are concatenated).
//
// RCCL has no gatherv: sizes first (one all-gather of a u64 per rank), then
// ONE group of point-to-point operations straight into the root's buffer at
// every rank's prefix offset (ncclGroupStart ... ncclSend / ncclRecv ...
// ncclGroupEnd).  xGMI is point-to-point: the root's 7 inbound links receive
// in parallel, which a ring (per-link bound) would not.
//
// librccl is opened at run time (dlopen): libsnapmi.so has no link-time
// dependency on it, and a process that already holds an RCCL (PyTorch ships
// its own) gets that one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "snapmi.h"
#include "snapmi_ctx.hpp"

namespace {

// the few declarations of rccl.h this file needs (ABI of RCCL 2.x)
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;     // 0 = ncclSuccess
constexpr int kNcclUint8 = 1; // ncclUint8 (= ncclChar + 1)
constexpr int kNcclUint64 = 5;

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t,
                         hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t,
                         hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so",
                               "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) // one the process holds already, first
            if ((r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)))
                break;
        for (size_t i = 0; !r.h && i < sizeof names / sizeof *names; i++)
            r.h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!r.h) {
            r.why = std::string("librccl not found: ") + dlerror();
            return;
        }
#define SNAPMI_SYM(f)                                                         \
    r.f = (decltype(r.f))dlsym(r.h, "nccl" #f);                               \
    if (!r.f)                                                                 \
        r.why = "librccl lacks nccl" #f;
        SNAPMI_SYM(GetUniqueId)
        SNAPMI_SYM(CommInitRank)
        SNAPMI_SYM(CommDestroy)
        SNAPMI_SYM(AllGather)
        SNAPMI_SYM(Send)
        SNAPMI_SYM(Recv)
        SNAPMI_SYM(GroupStart)
        SNAPMI_SYM(GroupEnd)
        SNAPMI_SYM(GetErrorString)
#undef SNAPMI_SYM
    });
    return r;
}

} // namespace

struct snapmi_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool owned = false;        // created here (destroyed here)
    uint64_t *d_sizes = nullptr; // [2 * world + 2] device scratch
};

#define NCCL_TRY(ctx, expr)                                                   \
    do {                                                                      \
        ncclResult_t _r = (expr);                                             \
        if (_r != 0)                                                          \
            return snapmi::fail_ctx((ctx), SNAPMI_E_DEVICE, "%s failed: %s",  \
                                    #expr, rccl().GetErrorString(_r));        \
    } while (0)

extern "C" {

int snapmi_comm_unique_id(uint8_t id_out[SNAPMI_COMM_ID_BYTES])
{
    if (!id_out)
        return SNAPMI_E_ARGUMENT;
    Rccl &r = rccl();
    if (!r.h || !r.why.empty()) {
        fprintf(stderr, "snapmi: %s\n", r.why.c_str());
        return SNAPMI_E_DEVICE;
    }
    ncclUniqueId id;
    static_assert(sizeof id == SNAPMI_COMM_ID_BYTES, "ncclUniqueId");
    if (r.GetUniqueId(&id) != 0)
        return SNAPMI_E_DEVICE;
    memcpy(id_out, &id, sizeof id);
    return SNAPMI_OK;
}

static int comm_scratch(snapmi_ctx *ctx, snapmi_comm *c)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc((void **)&c->d_sizes,
                           (size_t)(2 * c->world + 2) * sizeof(uint64_t)));
    return SNAPMI_OK;
}

int snapmi_comm_init(snapmi_ctx *ctx, const uint8_t id[SNAPMI_COMM_ID_BYTES],
                     int rank, int world, snapmi_comm **out)
{
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world)
        return SNAPMI_E_ARGUMENT;
    Rccl &r = rccl();
    if (!r.h || !r.why.empty())
        return snapmi::fail_ctx(ctx, SNAPMI_E_DEVICE, "%s", r.why.c_str());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    snapmi_comm *c = new snapmi_comm;
    c->rank = rank;
    c->world = world;
    c->owned = true;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t rc = r.CommInitRank(&c->comm, world, uid, rank);
    if (rc != 0) {
        delete c;
        return snapmi::fail_ctx(ctx, SNAPMI_E_DEVICE,
                                "ncclCommInitRank failed: %s",
                                r.GetErrorString(rc));
    }
    if (int e = comm_scratch(ctx, c)) {
        r.CommDestroy(c->comm);
        delete c;
        return e;
    }
    *out = c;
    return SNAPMI_OK;
}

int snapmi_comm_wrap(snapmi_ctx *ctx, void *nccl_comm, int rank, int world,
                     snapmi_comm **out)
{
    if (!ctx || !nccl_comm || !out || world < 1 || rank < 0 || rank >= world)
        return SNAPMI_E_ARGUMENT;
    Rccl &r = rccl();
    if (!r.h || !r.why.empty())
        return snapmi::fail_ctx(ctx, SNAPMI_E_DEVICE, "%s", r.why.c_str());
    snapmi_comm *c = new snapmi_comm;
    c->comm = (ncclComm_t)nccl_comm;
    c->rank = rank;
    c->world = world;
    if (int e = comm_scratch(ctx, c)) {
        delete c;
        return e;
    }
    *out = c;
    return SNAPMI_OK;
}

void snapmi_comm_destroy(snapmi_comm *c)
{
    if (!c)
        return;
    if (c->d_sizes)
        (void)hipFree(c->d_sizes);
    if (c->owned && c->comm)
        rccl().CommDestroy(c->comm);
    delete c;
}

int snapmi_gatherv(snapmi_ctx *ctx, snapmi_comm *c, int root,
                   const void *d_send, uint64_t send_bytes, void *d_recv,
                   uint64_t recv_cap, uint64_t *h_sizes, uint64_t *total)
{
    if (!ctx || !c || root < 0 || root >= c->world || (send_bytes && !d_send))
        return SNAPMI_E_ARGUMENT;
    Rccl &r = rccl();
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int W = c->world;
    // 1. sizes (and the root's capacity): two u64 per rank, everywhere
    // (a root without a receive buffer advertises capacity 0: then "the
    // root's buffer is too small" is every rank's error below, and no rank
    // enters the exchange to wait for a receive that is never posted)
    const uint64_t mine[2] = {send_bytes,
                              c->rank == root && !d_recv ? 0 : recv_cap};
    uint64_t *d_mine = c->d_sizes + 2 * (size_t)W;
    HIP_TRY(ctx, hipMemcpyAsync(d_mine, mine, sizeof mine,
                                hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipStreamSynchronize(s)); // `mine` is a stack variable
    NCCL_TRY(ctx, r.AllGather(d_mine, c->d_sizes, 2, kNcclUint64, c->comm, s));
    std::vector<uint64_t> all(2 * (size_t)W), sizes(W);
    HIP_TRY(ctx, hipMemcpyAsync(all.data(), c->d_sizes, all.size() * 8,
                                hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    uint64_t sum = 0;
    for (int k = 0; k < W; k++) {
        sizes[k] = all[2 * (size_t)k];
        sum += sizes[k];
    }
    if (h_sizes)
        memcpy(h_sizes, sizes.data(), (size_t)W * 8);
    if (total)
        *total = sum;
    // every rank knows all sizes and the root's capacity: an undersized
    // root buffer is everyone's error and nobody enters the exchange (no
    // rank is left waiting)
    const uint64_t cap = all[2 * (size_t)root + 1];
    if (sum > cap)
        return snapmi::fail_ctx(ctx, SNAPMI_E_ARGUMENT,
                                "gatherv: %llu bytes in all, the root's "
                                "buffer holds %llu",
                                (unsigned long long)sum,
                                (unsigned long long)cap);
    // 2. one group of point-to-point operations
    NCCL_TRY(ctx, r.GroupStart());
    ncclResult_t bad = 0;
    if (c->rank == root) {
        uint64_t off = 0;
        for (int k = 0; k < W; k++) {
            if (k != root && sizes[k] && !bad)
                bad = r.Recv((uint8_t *)d_recv + off, sizes[k], kNcclUint8, k,
                             c->comm, s);
            off += sizes[k];
        }
    } else if (send_bytes) {
        bad = r.Send(d_send, send_bytes, kNcclUint8, root, c->comm, s);
    }
    ncclResult_t ge = r.GroupEnd();
    if (bad || ge)
        return snapmi::fail_ctx(ctx, SNAPMI_E_DEVICE, "gatherv: %s",
                                r.GetErrorString(bad ? bad : ge));
    if (c->rank == root && send_bytes) {
        uint64_t off = 0;
        for (int k = 0; k < root; k++)
            off += sizes[k];
        HIP_TRY(ctx, hipMemcpyAsync((uint8_t *)d_recv + off, d_send,
                                    send_bytes, hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return SNAPMI_OK;
}

} // extern "C"
