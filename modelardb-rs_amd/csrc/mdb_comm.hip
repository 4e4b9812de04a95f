// mdb_comm.hip - the one exchange step of the path: merging the partial aggregate states of the GPUs
// of a node over RCCL / xGMI.
//
// Series shard embarrassingly over the GPUs (one process and one mdb_ctx per GPU): fit, grid and the
// per-segment aggregates never exchange anything. What remains is the final state of the
// Model{Count,Min,Max,Sum,Avg}Accumulators - {f64 sum, i64 count, f32 min, f32 max}
// (crates/modelardb_storage/src/optimizer/model_simple_aggregates.rs:362-378, 517-534, 591-612 hand
// that state to DataFusion's final aggregate; with one partition per GPU this call is that merge).
// It is done as ONE ncclAllGather of 32 bytes per rank and a fold in rank order with the
// accumulators' own update rules, not as an ncclAllReduce: an all-reduce leaves the order of the f64
// additions to the ring, the fold makes SUM reproducible run to run and identical on every rank.
// 32 bytes over xGMI: latency only.
//
// librccl is bound at FIRST USE (dlopen), not as a DT_NEEDED entry of libmdb_hip.so: a process that
// also hosts PyTorch (bench.py, the tests) must end up with ONE librccl - torch ships its own with
// the same SONAME - and mapping ROCm's copy ahead of torch's import aborts that process at exit
// (glibc "double free or corruption", reproduced with nothing but ctypes.CDLL("librccl.so.1")
// followed by `import torch`). Bound lazily, the library that is already mapped is reused; a host
// without torch (the Rust server) gets ROCm's librccl.so.1 through the normal search path.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "mdb_common.hpp"

namespace mdb {

struct WireState { // what travels: the state plus who sent it (checked on arrival)
    double sum;
    long long count;
    float min;
    float max;
    int32_t rank;
    uint32_t magic;
};
static_assert(sizeof(WireState) == 32, "one all-gather slot is 32 bytes");
constexpr uint32_t WIRE_MAGIC = 0x4D444241u; // "MDBA"

struct Rccl {
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) get_error_string = nullptr;
    std::string error;
};

// nullptr + g_last_error when librccl cannot be bound.
static const Rccl *rccl() {
    static const Rccl *bound = []() {
        Rccl *r = new Rccl();
        void *handle = nullptr;
        for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) {
            const char *why = dlerror(); // (may be NULL)
            r->error = std::string("librccl.so.1 cannot be loaded: ") + (why ? why : "unknown reason");
            return r;
        }
        auto bind = [&](auto &slot, const char *symbol) {
            slot = reinterpret_cast<std::remove_reference_t<decltype(slot)>>(dlsym(handle, symbol));
            if (!slot && r->error.empty()) r->error = std::string("librccl.so.1 lacks ") + symbol;
        };
        bind(r->get_unique_id, "ncclGetUniqueId");
        bind(r->comm_init_rank, "ncclCommInitRank");
        bind(r->comm_destroy, "ncclCommDestroy");
        bind(r->all_gather, "ncclAllGather");
        bind(r->get_error_string, "ncclGetErrorString");
        return r;
    }();
    if (!bound->error.empty()) {
        fail(bound->error);
        return nullptr;
    }
    return bound;
}

#define MDB_NCCL_CHECK(expr)                                                                       \
    do {                                                                                           \
        ncclResult_t mdb_nccl_ = (expr);                                                           \
        if (mdb_nccl_ != ncclSuccess)                                                              \
            return ::mdb::fail(std::string(#expr) + ": " + nccl->get_error_string(mdb_nccl_));    \
    } while (0)

// Fold `from` into `into` exactly as the accumulators fold a batch into their state
// (model_simple_aggregates.rs:355 count, :398-401 min, :441-444 max, :501-510 sum): NaN partial
// extrema are skipped the way f32::min / f32::max skip them.
void merge_agg_state(mdb_agg_state *into, const mdb_agg_state &from) {
    into->sum += from.sum;
    into->count += from.count;
    if (!(from.min != from.min) && (into->min != into->min || from.min < into->min)) into->min = from.min;
    if (!(from.max != from.max) && (into->max != into->max || from.max > into->max)) into->max = from.max;
}

} // namespace mdb

using namespace mdb;

extern "C" {

int mdb_comm_unique_id(void *id_out) {
    if (!id_out) return fail("id_out must not be NULL.");
    static_assert(MDB_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "MDB_COMM_ID_BYTES mirrors ncclUniqueId");
    const Rccl *nccl = rccl();
    if (!nccl) return 1;
    ncclUniqueId id;
    MDB_NCCL_CHECK(nccl->get_unique_id(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

int mdb_comm_init(mdb_ctx *ctx, int32_t rank, int32_t world, const void *unique_id) {
    if (!ctx || !unique_id) return fail("ctx and unique_id must not be NULL.");
    if (world < 1 || rank < 0 || rank >= world) return fail("rank must be in [0, world).");
    mdb::CallGuard lock(ctx);
    if (ctx->comm) return fail("The context already has a communicator.");
    const Rccl *nccl = rccl();
    if (!nccl) return 1;
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    MDB_NCCL_CHECK(nccl->comm_init_rank(&comm, world, id, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return 0;
}

int mdb_comm_close(mdb_ctx *ctx) {
    if (!ctx) return fail("ctx must not be NULL.");
    mdb::CallGuard lock(ctx);
    if (!ctx->comm) return 0;
    const Rccl *nccl = rccl();
    if (!nccl) return 1;
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 0;
    MDB_NCCL_CHECK(nccl->comm_destroy(comm));
    return 0;
}

int mdb_agg_all_reduce(mdb_ctx *ctx, mdb_agg_state *inout, int32_t *ranks_seen) {
    if (!ctx || !inout) return fail("ctx and inout must not be NULL.");
    mdb::CallGuard lock(ctx);
    if (!ctx->comm) return fail("mdb_comm_init has not been called on this context.");
    const Rccl *nccl = rccl();
    if (!nccl) return 1;
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    const int world = ctx->comm_world;
    void *p = nullptr;
    if (scratch_reserve(ctx, SCRATCH_COMM, sizeof(WireState) * (uint64_t)(world + 1), &p)) return 1;
    WireState *send = static_cast<WireState *>(p);
    WireState *recv = send + 1;
    void *h = nullptr;
    if (pinned_reserve(ctx, sizeof(WireState) * (uint64_t)(world + 1), &h)) return 1;
    WireState *host = static_cast<WireState *>(h);
    host[0] = {inout->sum, (long long)inout->count, inout->min, inout->max, ctx->comm_rank, WIRE_MAGIC};
    MDB_HIP_CHECK(hipMemcpyAsync(send, host, sizeof(WireState), hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP_CHECK(hipMemsetAsync(recv, 0, sizeof(WireState) * (size_t)world, ctx->stream));
    {
        LaunchTimer timer(ctx, "rccl_all_gather");
        MDB_NCCL_CHECK(nccl->all_gather(send, recv, sizeof(WireState), ncclChar, static_cast<ncclComm_t>(ctx->comm),
                                     ctx->stream));
    }
    MDB_HIP_CHECK(hipMemcpyAsync(host + 1, recv, sizeof(WireState) * (size_t)world, hipMemcpyDeviceToHost,
                                 ctx->stream));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    mdb_agg_state merged = {0.0, 0, 3.4028234663852886e38f, -3.4028234663852886e38f};
    int32_t seen = 0;
    for (int r = 0; r < world; r++) {
        const WireState &w = host[1 + r];
        if (w.magic != WIRE_MAGIC || w.rank != r)
            return fail("The all-gather returned no state for rank " + std::to_string(r) + ".");
        seen += 1;
        merge_agg_state(&merged, mdb_agg_state{w.sum, (int64_t)w.count, w.min, w.max});
    }
    *inout = merged;
    if (ranks_seen) *ranks_seen = seen;
    return 0;
}

int mdb_agg_merge(mdb_agg_state *into, const mdb_agg_state *from) {
    if (!into || !from) return fail("into and from must not be NULL.");
    merge_agg_state(into, *from);
    return 0;
}

} // extern "C"
