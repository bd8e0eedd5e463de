// svo_comm.cpp -- the frame-end exchange of the tile-sharded multi-GPU frame behind the C ABI (include/svo_hip.h,
// svo_comm_* / svo_gather_frame*): RCCL over xGMI, one ncclGather of hit records to the root rank per frame, enqueued on
// the context's HIP stream.  No reference counterpart: the reference drives one device from one thread
// (/root/reference/src/main.rs:40-88); SURVEY.md 8b "Threading" / 8e ask for the communicator behind the boundary so that
// a Rust host bound per INTEGRATION.md can shard a frame without linking RCCL itself.
//
// librccl.so.1 is opened on first use instead of being linked: a single-GPU user needs no RCCL at all, and a process
// that already carries one (PyTorch-ROCm ships its own copy under the same soname) keeps exactly that instance.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "svo_ctx.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char *e = dlerror();
            r.error = std::string("librccl.so.1 not found: ") + (e ? e : "?");
            return;
        }
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p) {
                ok = false;
                r.error = std::string("librccl lacks ") + name;
            }
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Gather = reinterpret_cast<decltype(r.Gather)>(sym("ncclGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) {
            dlclose(r.handle);
            r.handle = nullptr;
        }
    });
    return r.handle ? &r : nullptr;
}

int no_rccl(svo_ctx *ctx) {
    rccl();  // (fills the error text)
    return svo_fail(ctx, SVO_ERR_COMM, "RCCL unavailable: librccl.so.1 could not be loaded");
}

int fail_nccl(svo_ctx *ctx, Rccl *r, ncclResult_t e, const char *what) {
    if (ctx) ctx->err = std::string(what) + ": " + r->GetErrorString(e);
    return SVO_ERR_COMM;
}

static_assert(sizeof(ncclUniqueId) == SVO_COMM_ID_BYTES, "SVO_COMM_ID_BYTES must be the size of ncclUniqueId");

int drop_comm(svo_ctx *ctx, Rccl *r) {
    if (!ctx->comm) return SVO_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);  // collectives in flight finish first
    ncclResult_t e = r->CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = ctx->comm_rank = 0;
    if (ctx->comm_ready) (void)hipEventDestroy(ctx->comm_ready);
    if (ctx->comm_done) (void)hipEventDestroy(ctx->comm_done);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    ctx->comm_ready = ctx->comm_done = nullptr;
    ctx->comm_stream = nullptr;
    ctx->gathers_issued = false;
    return e == ncclSuccess ? SVO_OK : fail_nccl(ctx, r, e, "ncclCommDestroy");
}

// The exchange runs on a stream of its own, so that the context's stream can go on with the next frame while records
// travel: ordered behind everything enqueued on the context's stream so far; svo_gather_wait orders the other way.
int comm_stream_ready(svo_ctx *ctx) {
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess && !ctx->comm_stream) e = hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess && !ctx->comm_ready) e = hipEventCreateWithFlags(&ctx->comm_ready, hipEventDisableTiming);
    if (e == hipSuccess && !ctx->comm_done) e = hipEventCreateWithFlags(&ctx->comm_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ctx->comm_ready, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ready, 0);
    return e == hipSuccess ? SVO_OK : svo_fail_hip(ctx, e, "communication stream set-up");
}

int comm_stream_done(svo_ctx *ctx) {
    hipError_t e = hipEventRecord(ctx->comm_done, ctx->comm_stream);
    ctx->gathers_issued = true;
    return e == hipSuccess ? SVO_OK : svo_fail_hip(ctx, e, "hipEventRecord");
}

}  // namespace

extern "C" {

int svo_comm_unique_id(uint8_t id_out[SVO_COMM_ID_BYTES]) {
    if (!id_out) return SVO_ERR_ARG;
    Rccl *r = rccl();
    if (!r) return SVO_ERR_COMM;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return SVO_ERR_COMM;
    memcpy(id_out, &id, SVO_COMM_ID_BYTES);
    return SVO_OK;
}

int svo_comm_init_rank(svo_ctx *ctx, const uint8_t id[SVO_COMM_ID_BYTES], int world, int rank) {
    if (!ctx || !id) return SVO_ERR_ARG;
    if (world < 1 || rank < 0 || rank >= world) return svo_fail(ctx, SVO_ERR_ARG, "need 0 <= rank < world");
    Rccl *r = rccl();
    if (!r) return no_rccl(ctx);
    int rc = drop_comm(ctx, r);
    if (rc) return rc;
    hipError_t he = hipSetDevice(ctx->device);  // the communicator binds to the current device
    if (he != hipSuccess) return svo_fail_hip(ctx, he, "hipSetDevice");
    ncclUniqueId uid;
    memcpy(&uid, id, SVO_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    ncclResult_t e = r->CommInitRank(&comm, world, uid, rank);
    if (e != ncclSuccess) return fail_nccl(ctx, r, e, "ncclCommInitRank");
    ctx->comm = comm;
    ctx->comm_world = world;
    ctx->comm_rank = rank;
    return SVO_OK;
}

int svo_comm_init_all(int n, svo_ctx *const *ctxs) {
    if (n < 1 || !ctxs) return SVO_ERR_ARG;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return SVO_ERR_ARG;
        for (int j = 0; j < i; j++)
            if (ctxs[j]->device == ctxs[i]->device) return svo_fail(ctxs[0], SVO_ERR_ARG, "svo_comm_init_all: one context per device");
    }
    Rccl *r = rccl();
    if (!r) return no_rccl(ctxs[0]);
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) {
        int rc = drop_comm(ctxs[i], r);
        if (rc) return rc;
        devs[i] = ctxs[i]->device;
    }
    std::vector<ncclComm_t> comms(n, nullptr);
    ncclResult_t e = r->CommInitAll(comms.data(), n, devs.data());
    if (e != ncclSuccess) return fail_nccl(ctxs[0], r, e, "ncclCommInitAll");
    for (int i = 0; i < n; i++) {
        ctxs[i]->comm = comms[i];
        ctxs[i]->comm_world = n;
        ctxs[i]->comm_rank = i;
    }
    return SVO_OK;
}

int svo_comm_destroy(svo_ctx *ctx) {
    if (!ctx) return SVO_ERR_ARG;
    if (!ctx->comm) return SVO_OK;
    Rccl *r = rccl();
    if (!r) return no_rccl(ctx);
    return drop_comm(ctx, r);
}

static int gather_args_ok(svo_ctx *ctx, const void *send, size_t bytes, void *recv_on_root, int root) {
    if (!ctx->comm) return svo_fail(ctx, SVO_ERR_STATE, "no communicator (svo_comm_init_rank / svo_comm_init_all)");
    if (root < 0 || root >= ctx->comm_world) return svo_fail(ctx, SVO_ERR_ARG, "root is not a rank of the communicator");
    if (bytes & 3u) return svo_fail(ctx, SVO_ERR_ARG, "bytes must be a multiple of 4");
    if (bytes && !send) return svo_fail(ctx, SVO_ERR_ARG, "send is NULL");
    if (bytes && ctx->comm_rank == root && !recv_on_root) return svo_fail(ctx, SVO_ERR_ARG, "recv_on_root is NULL on the root rank");
    return SVO_OK;
}

int svo_gather_frame(svo_ctx *ctx, const void *send, size_t bytes, void *recv_on_root, int root) {
    if (!ctx) return SVO_ERR_ARG;
    int rc = gather_args_ok(ctx, send, bytes, recv_on_root, root);
    if (rc) return rc;
    Rccl *r = rccl();
    if (!r) return no_rccl(ctx);
    rc = comm_stream_ready(ctx);
    if (rc) return rc;
    // hit records are words: ncclUint32 elements keep every count far below 2^31
    ncclResult_t e = r->Gather(send, ctx->comm_rank == root ? recv_on_root : nullptr, bytes / 4, ncclUint32, root, (ncclComm_t)ctx->comm,
                               ctx->comm_stream);
    if (e != ncclSuccess) return fail_nccl(ctx, r, e, "ncclGather");
    return comm_stream_done(ctx);
}

int svo_gather_frame_all(int n, svo_ctx *const *ctxs, const void *const *send, size_t bytes, void *recv_on_root, int root) {
    if (n < 1 || !ctxs || !send) return SVO_ERR_ARG;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return SVO_ERR_ARG;
        if (ctxs[i]->comm_world != n || ctxs[i]->comm_rank != i)
            return svo_fail(ctxs[i], SVO_ERR_STATE, "svo_gather_frame_all needs the contexts of one svo_comm_init_all, in rank order");
        int rc = gather_args_ok(ctxs[i], send[i], bytes, recv_on_root, root);
        if (rc) return rc;
    }
    Rccl *r = rccl();
    if (!r) return no_rccl(ctxs[0]);
    for (int i = 0; i < n; i++) {
        int rc = comm_stream_ready(ctxs[i]);
        if (rc) return rc;
    }
    // one thread issues the collective for every rank: the calls must sit inside one group or the first would wait for its peers
    ncclResult_t e = r->GroupStart();
    if (e != ncclSuccess) return fail_nccl(ctxs[0], r, e, "ncclGroupStart");
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < n && first == ncclSuccess; i++) {
        (void)hipSetDevice(ctxs[i]->device);
        first = r->Gather(send[i], i == root ? recv_on_root : nullptr, bytes / 4, ncclUint32, root, (ncclComm_t)ctxs[i]->comm,
                          ctxs[i]->comm_stream);
    }
    e = r->GroupEnd();
    if (first != ncclSuccess) return fail_nccl(ctxs[0], r, first, "ncclGather");
    if (e != ncclSuccess) return fail_nccl(ctxs[0], r, e, "ncclGroupEnd");
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(ctxs[i]->device);
        int rc = comm_stream_done(ctxs[i]);
        if (rc) return rc;
    }
    return SVO_OK;
}

int svo_gather_wait(svo_ctx *ctx) {
    if (!ctx) return SVO_ERR_ARG;
    if (!ctx->gathers_issued) return SVO_OK;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->comm_done, 0);
    return e == hipSuccess ? SVO_OK : svo_fail_hip(ctx, e, "hipStreamWaitEvent");
}

}  // extern "C"

// called by svo_ctx_destroy (svo_abi.cpp)
void svo_comm_release(svo_ctx *ctx) {
    if (!ctx || !ctx->comm) return;
    if (Rccl *r = rccl()) (void)drop_comm(ctx, r);
}
