// mdb_ctx.hip - context lifetime, device memory, segment upload/download and launch profiling of
// libmdb_hip.so (see include/mdb.h for the contract of every entry point).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <set>
#include <string>

#include "mdb_common.hpp"

namespace mdb {

thread_local std::string g_last_error;

int scratch_reserve(mdb_ctx *ctx, ScratchSlot slot, uint64_t bytes, void **out) {
    if (bytes == 0) bytes = 256;
    if (ctx->scratch_bytes[slot] < bytes) {
        if (ctx->scratch[slot]) {
            MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            MDB_HIP_CHECK(hipFree(ctx->scratch[slot]));
            ctx->scratch[slot] = nullptr;
            ctx->scratch_bytes[slot] = 0;
        }
        uint64_t grown = align_up(bytes + bytes / 8, 1 << 12);
        MDB_HIP_CHECK(hipMalloc(&ctx->scratch[slot], grown));
        ctx->scratch_bytes[slot] = grown;
    }
    *out = ctx->scratch[slot];
    return 0;
}

// (made by the first small copy of the context: a context that only ever grids never pays for it)
static void mail_make(mdb_ctx *ctx) {
    if (ctx->mail || ctx->mail_failed) return;
    void *block = nullptr;
    if (hipHostMalloc(&block, MAIL_BYTES, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        ctx->mail_failed = true;
        return;
    }
    ctx->mail = static_cast<unsigned char *>(block);
}

hipError_t mail_read(mdb_ctx *ctx, void *host_to, const void *dev_from, uint64_t bytes) {
    if (bytes == 0) return hipSuccess;
    mail_make(ctx);
    const uint64_t at = (ctx->mail_used + 15u) & ~15ull;
    if (!ctx->mail || bytes > MAIL_COPY_LIMIT || at + bytes > MAIL_BYTES)
        return hipMemcpyAsync(host_to, dev_from, bytes, hipMemcpyDeviceToHost, ctx->stream);
    ctx->mail_used = at + bytes;
    ctx->mail_reads.push_back({host_to, at, bytes});
    return hipMemcpyAsync(ctx->mail + at, dev_from, bytes, hipMemcpyDeviceToHost, ctx->stream);
}

hipError_t mail_write(mdb_ctx *ctx, void *dev_to, const void *host_from, uint64_t bytes) {
    if (bytes == 0) return hipSuccess;
    mail_make(ctx);
    const uint64_t at = (ctx->mail_used + 15u) & ~15ull;
    if (!ctx->mail || bytes > MAIL_COPY_LIMIT || at + bytes > MAIL_BYTES) {
        // Too large for the mailbox (or none): straight from the caller's pageable memory - and waited for, because
        // the contract is "`host_from` may be changed or freed at once" (mdb_common.hpp) whichever way the bytes go.
        const hipError_t status = hipMemcpyAsync(dev_to, host_from, bytes, hipMemcpyHostToDevice, ctx->stream);
        return status != hipSuccess ? status : hipStreamSynchronize(ctx->stream);
    }
    ctx->mail_used = at + bytes;
    std::memcpy(ctx->mail + at, host_from, bytes);
    return hipMemcpyAsync(dev_to, ctx->mail + at, bytes, hipMemcpyHostToDevice, ctx->stream);
}

hipError_t mail_sync(mdb_ctx *ctx) {
    const hipError_t status = hipStreamSynchronize(ctx->stream);
    if (status == hipSuccess)
        for (const mdb_ctx::MailRead &read : ctx->mail_reads) std::memcpy(read.to, ctx->mail + read.at, read.bytes);
    ctx->mail_reads.clear();
    ctx->mail_used = 0; // (everything that was written from here has been read by the copy engine)
    return status;
}

void mail_drop(mdb_ctx *ctx) {
    (void)hipStreamSynchronize(ctx->stream); // (copies into the mailbox may still be under way)
    ctx->mail_reads.clear();
    ctx->mail_used = 0;
}

int pinned_reserve(mdb_ctx *ctx, uint64_t bytes, void **out) {
    if (bytes == 0) bytes = 256;
    if (ctx->pinned_bytes < bytes) {
        if (ctx->pinned) {
            MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            MDB_HIP_CHECK(hipHostFree(ctx->pinned));
            ctx->pinned = nullptr;
            ctx->pinned_bytes = 0;
        }
        uint64_t grown = align_up(bytes + bytes / 8, 1 << 12);
        MDB_HIP_CHECK(hipHostMalloc(&ctx->pinned, grown, hipHostMallocDefault));
        ctx->pinned_bytes = grown;
    }
    *out = ctx->pinned;
    return 0;
}

void scratch_enforce_limit(mdb_ctx *ctx) {
    uint64_t total = 0;
    for (int i = 0; i < SCRATCH_SLOT_COUNT; i++) total += ctx->scratch[i] ? ctx->scratch_bytes[i] : 0;
    if (total <= ctx->scratch_limit) return;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return;
    while (total > ctx->scratch_limit) {
        int largest = -1;
        for (int i = 0; i < SCRATCH_SLOT_COUNT; i++)
            if (ctx->scratch[i] && (largest < 0 || ctx->scratch_bytes[i] > ctx->scratch_bytes[largest])) largest = i;
        if (largest < 0) break;
        (void)hipFree(ctx->scratch[largest]);
        total -= ctx->scratch_bytes[largest];
        ctx->scratch[largest] = nullptr;
        ctx->scratch_bytes[largest] = 0;
    }
}

int PinnedPool::take(uint64_t bytes, void **out, uint64_t *capacity) {
    if (bytes == 0) bytes = 256;
    {
        // Best fit among the recycled blocks that are not wastefully large. The bounds are loose on
        // purpose: the batches of one query differ in size by the data they decode to, and a block that
        // does not fit costs a hipHostMalloc (13 ms for 70 MB, ten batches' worth of copying).
        std::lock_guard<std::mutex> lock(mutex);
        int best = -1;
        for (size_t i = 0; i < blocks.size(); i++) {
            uint64_t cap = blocks[i].second;
            if (cap >= bytes && cap <= 4 * bytes + (16 << 20) && (best < 0 || cap < blocks[(size_t)best].second))
                best = (int)i;
        }
        if (best >= 0) {
            *out = blocks[(size_t)best].first;
            *capacity = blocks[(size_t)best].second;
            blocks.erase(blocks.begin() + best);
            return 0;
        }
    }
    // Size classes a quarter of a power of two apart, so that blocks made for one batch fit the next.
    uint64_t grown = 1 << 16;
    while (grown < bytes) grown <<= 1;
    for (uint64_t quarter = grown >> 3, smaller = grown - quarter; quarter >= (1 << 14) && smaller >= bytes;
         smaller -= quarter)
        grown = smaller;
    MDB_HIP_CHECK(hipHostMalloc(out, grown, hipHostMallocDefault));
    *capacity = grown;
    return 0;
}

void PinnedPool::give(void *block, uint64_t capacity) {
    std::lock_guard<std::mutex> lock(mutex);
    if (closed) {
        (void)hipHostFree(block);
        return;
    }
    if (blocks.size() >= 8) { // keep the pool small: drop the smallest block
        size_t smallest = 0;
        for (size_t i = 1; i < blocks.size(); i++)
            if (blocks[i].second < blocks[smallest].second) smallest = i;
        if (blocks[smallest].second < capacity) {
            (void)hipHostFree(blocks[smallest].first);
            blocks[smallest] = {block, capacity};
        } else {
            (void)hipHostFree(block);
        }
        return;
    }
    blocks.push_back({block, capacity});
}

void PinnedPool::trim() {
    std::lock_guard<std::mutex> lock(mutex);
    for (auto &block : blocks) (void)hipHostFree(block.first);
    blocks.clear();
}

void PinnedPool::close() {
    std::lock_guard<std::mutex> lock(mutex);
    closed = true;
    for (auto &block : blocks) (void)hipHostFree(block.first);
    blocks.clear();
}

// (under a lock of their own: the context's call lock is held by a running job from its upload to its last copy,
// and a submit that waited for it could never put a second job next to the first)
namespace {
struct OwnedRegistry {
    std::mutex mutex;
    std::map<const void *, OwnedSegments *> by_values_views;
    static OwnedRegistry &instance() {
        static OwnedRegistry *registry = new OwnedRegistry();
        return *registry;
    }
};
} // namespace

void owned_segments_register(OwnedSegments *owned) {
    if (!owned->c.on_device || owned->c.seg.n == 0 || !owned->c.seg.values.views) return;
    owned->mv_index = std::make_shared<MvIndex>();
    owned->mv_index->device = owned->device;
    OwnedRegistry &registry = OwnedRegistry::instance();
    std::lock_guard<std::mutex> lock(registry.mutex);
    registry.by_values_views[owned->c.seg.values.views] = owned;
}

void owned_segments_forget(OwnedSegments *owned) {
    if (!owned->mv_index) return;
    OwnedRegistry &registry = OwnedRegistry::instance();
    std::lock_guard<std::mutex> lock(registry.mutex);
    auto found = registry.by_values_views.find(owned->c.seg.values.views);
    if (found != registry.by_values_views.end() && found->second == owned) registry.by_values_views.erase(found);
}

std::shared_ptr<MvIndex> owned_segments_index(const mdb_segments *in) {
    if (!in || in->n == 0) return nullptr;
    OwnedRegistry &registry = OwnedRegistry::instance();
    std::lock_guard<std::mutex> lock(registry.mutex);
    auto found = registry.by_values_views.find(in->values.views);
    if (found == registry.by_values_views.end()) return nullptr;
    const mdb_segments &own = found->second->c.seg;
    // (the same batch, not a slice or a foreign struct that happens to share a column)
    if (own.n != in->n || own.model_type_id != in->model_type_id || own.residuals.views != in->residuals.views ||
        own.timestamps.views != in->timestamps.views)
        return nullptr;
    return found->second->mv_index;
}

GridPipeline *ctx_pipeline(mdb_ctx *ctx) {
    std::lock_guard<std::mutex> lock(ctx->pipeline_mutex);
    return ctx->pipeline;
}

GridPipeline *ctx_pipeline_install(mdb_ctx *ctx, GridPipeline *fresh) {
    GridPipeline *installed = nullptr;
    {
        std::lock_guard<std::mutex> lock(ctx->pipeline_mutex);
        if (!ctx->pipeline) ctx->pipeline = fresh;
        installed = ctx->pipeline;
    }
    if (installed == fresh) { // (no job has been handed to it yet) its clones are profiled if the context is
        bool profiling = false;
        {
            CallGuard lock(ctx);
            profiling = ctx->profiling;
        }
        mdb_ctx *clones[8];
        const int n = pipeline_clones(ctx, clones, 8);
        for (int k = 0; k < n; k++) {
            CallGuard lock(clones[k]);
            clones[k]->profiling = profiling;
        }
    }
    return installed;
}

GridPipeline *ctx_pipeline_detach(mdb_ctx *ctx) {
    std::lock_guard<std::mutex> lock(ctx->pipeline_mutex);
    GridPipeline *pipeline = ctx->pipeline;
    ctx->pipeline = nullptr;
    return pipeline;
}

LaunchTimer::LaunchTimer(mdb_ctx *c, const char *n) : ctx(c), name(n) {
    if (!ctx->profiling) return;
    auto take = [&]() {
        hipEvent_t e = nullptr;
        if (!ctx->event_pool.empty()) {
            e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else if (hipEventCreate(&e) != hipSuccess) {
            e = nullptr;
        }
        return e;
    };
    start = take();
    stop = take();
    if (start) (void)hipEventRecord(start, ctx->stream);
}

LaunchTimer::~LaunchTimer() {
    if (!ctx->profiling || !start || !stop) return;
    (void)hipEventRecord(stop, ctx->stream);
    ctx->pending_events.push_back({name, start, stop});
}

int profile_collect(mdb_ctx *ctx) {
    if (ctx->pending_events.empty()) return 0;
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (auto &p : ctx->pending_events) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            KernelTime &k = ctx->kernel_times[p.name];
            k.launches += 1;
            k.total_ms += ms;
        }
        ctx->event_pool.push_back(p.start);
        ctx->event_pool.push_back(p.stop);
    }
    ctx->pending_events.clear();
    return 0;
}

// Every out-of-line view must point into one of the column's data buffers: the kernels follow
// buffer_index and offset without looking (view_data, mdb_common.hpp), so a view that does not is a
// wild device read. The reference cannot build such a column (arrow validates it); a foreign or
// corrupted batch gets an error here instead of a GPU memory fault.
int validate_views_host(const mdb_binview_col &col, uint64_t n) {
    for (int32_t b = 0; b < col.n_buffers; b++)
        if (col.buffer_sizes[b] < 0) return fail("Malformed BinaryView: negative buffer size.");
    for (uint64_t i = 0; i < n; i++) {
        const mdb_view16 &view = col.views[i];
        if (view.length < 0) return fail("Malformed BinaryView: negative length.");
        if (view.length <= 12) continue;
        const int32_t buffer = view.u.ref.buffer_index;
        const int64_t offset = view.u.ref.offset;
        if (buffer < 0 || buffer >= col.n_buffers || offset < 0 ||
            offset + (int64_t)view.length > col.buffer_sizes[buffer])
            return fail("Malformed BinaryView: row " + std::to_string(i) +
                        " points outside the column's data buffers.");
    }
    return 0;
}

// The same check for a batch whose views are already in device memory (buffer_sizes stays a host
// array, as everywhere in mdb_segments): flag[0] counts the offending views.
__global__ __launch_bounds__(256) void k_validate_views(const uint4 *__restrict__ views, uint64_t n,
                                                        const long long *__restrict__ buffer_sizes,
                                                        int32_t n_buffers, unsigned int *__restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 view = views[i];
    const int32_t length = (int32_t)view.x;
    bool bad = length < 0;
    if (!bad && length > 12) {
        const int32_t buffer = (int32_t)view.z;
        const int64_t offset = (int32_t)view.w;
        bad = buffer < 0 || buffer >= n_buffers || offset < 0 || offset + length > buffer_sizes[buffer];
    }
    if (bad) atomicAdd(flag, 1u);
}

} // namespace mdb

using namespace mdb;

extern "C" {

const char *mdb_last_error(void) { return g_last_error.c_str(); }

const char *mdb_version(void) { return "libmdb_hip 0.1.0 gfx950"; }

int mdb_init(int device, mdb_ctx **out) {
    if (!out) return fail("ctx must not be NULL.");
    *out = nullptr;
    int count = 0;
    MDB_HIP_CHECK(hipGetDeviceCount(&count));
    if (device < 0 || device >= count)
        return fail("No such HIP device: " + std::to_string(device) + " of " + std::to_string(count));
    MDB_HIP_CHECK(hipSetDevice(device));
    mdb_ctx *ctx = new mdb_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->compute_units = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return fail("hipStreamCreate failed.");
    }
    ctx->own_stream = true;
    // One pool of page-locked result blocks per device for the whole process: an operator makes its contexts per
    // query, and a pool that died with them would pay hipHostMalloc (13 ms for 70 MB) for every block of every
    // query again. At most eight blocks are kept (mdb_trim gives them back); never freed at process exit,
    // when the HIP runtime may be gone already.
    {
        static std::mutex pools_mutex;
        static std::map<int, std::shared_ptr<PinnedPool>> *pools = new std::map<int, std::shared_ptr<PinnedPool>>();
        std::lock_guard<std::mutex> lock(pools_mutex);
        std::shared_ptr<PinnedPool> &pool = (*pools)[device];
        if (!pool) pool = std::make_shared<PinnedPool>();
        ctx->pinned_pool = pool;
        ctx->owns_pinned_pool = false;
    }
    *out = ctx;
    return 0;
}

int mdb_clone(mdb_ctx *ctx, mdb_ctx **out) {
    if (!ctx || !out) return fail("ctx and out must not be NULL.");
    {
        mdb::CallGuard lock(ctx);
        if (!ctx->clones) ctx->clones = std::make_shared<CloneCache>();
    }
    {
        std::lock_guard<std::mutex> lock(ctx->clones->mutex);
        if (!ctx->clones->idle.empty()) { // a clone closed earlier: its stream and scratch are still there
            *out = ctx->clones->idle.back();
            ctx->clones->idle.pop_back();
            (*out)->scratch_limit = ctx->scratch_limit;
            return 0;
        }
    }
    if (mdb_init(ctx->device, out)) return 1;
    (*out)->clones = ctx->clones;
    (*out)->is_clone = true;
    (*out)->scratch_limit = ctx->scratch_limit;
    return 0;
}

int mdb_close(mdb_ctx *ctx) {
    if (!ctx) return 0;
    pipeline_close(ctx); // the workers of mdb_grid_submit finish what is queued; their second context is closed
    if (ctx->clones) {
        std::vector<mdb_ctx *> orphans;
        {
            std::unique_lock<std::mutex> lock(ctx->clones->mutex);
            // Kept for the next mdb_clone of the context it was made from - but only as what a fresh clone is:
            // its own stream (a caller's stream set with mdb_set_stream may be destroyed behind our back), no
            // communicator (the next owner's mdb_comm_init would find one), no timings of the previous owner.
            if (ctx->is_clone && !ctx->clones->origin_closed && ctx->clones->idle.size() < 4 && ctx->own_stream) {
                lock.unlock();
                (void)mdb_comm_close(ctx);
                (void)hipSetDevice(ctx->device);
                (void)hipStreamSynchronize(ctx->stream);
                for (auto &p : ctx->pending_events) {
                    ctx->event_pool.push_back(p.start);
                    ctx->event_pool.push_back(p.stop);
                }
                ctx->pending_events.clear();
                ctx->kernel_times.clear();
                ctx->profiling = false;
                lock.lock();
                if (!ctx->clones->origin_closed && ctx->clones->idle.size() < 4) {
                    ctx->clones->idle.push_back(ctx);
                    return 0;
                }
            }
            if (!ctx->is_clone) {
                ctx->clones->origin_closed = true;
                orphans.swap(ctx->clones->idle);
            }
        }
        for (mdb_ctx *orphan : orphans) {
            orphan->clones.reset();
            (void)mdb_close(orphan);
        }
    }
    (void)mdb_comm_close(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->pending_events) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < SCRATCH_SLOT_COUNT; i++)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->mail) (void)hipHostFree(ctx->mail);
    if (ctx->owns_pinned_pool) ctx->pinned_pool->close();
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int mdb_set_scratch_limit(mdb_ctx *ctx, uint64_t bytes) {
    if (!ctx) return fail("ctx must not be NULL.");
    mdb::CallGuard lock(ctx);
    ctx->scratch_limit = bytes;
    return 0;
}

int mdb_trim(mdb_ctx *ctx, uint64_t *released_bytes) {
    if (!ctx) return fail("ctx must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    uint64_t released = 0;
    for (int i = 0; i < SCRATCH_SLOT_COUNT; i++) {
        if (!ctx->scratch[i]) continue;
        MDB_HIP_CHECK(hipFree(ctx->scratch[i]));
        released += ctx->scratch_bytes[i];
        ctx->scratch[i] = nullptr;
        ctx->scratch_bytes[i] = 0;
    }
    if (ctx->pinned) {
        MDB_HIP_CHECK(hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
    }
    ctx->pinned_pool->trim();
    if (released_bytes) *released_bytes = released;
    return 0;
}

int mdb_set_stream(mdb_ctx *ctx, void *hip_stream) {
    if (!ctx) return fail("ctx must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    return 0;
}

int mdb_device_info(mdb_ctx *ctx, char *name, uint64_t name_cap, int32_t *compute_units,
                    uint64_t *hbm_bytes) {
    if (!ctx) return fail("ctx must not be NULL.");
    hipDeviceProp_t prop;
    MDB_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_cap) {
        std::string full = std::string(prop.name) + " " + prop.gcnArchName;
        std::strncpy(name, full.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return 0;
}

int mdb_dev_alloc(mdb_ctx *ctx, uint64_t bytes, void **dev_ptr) {
    if (!ctx || !dev_ptr) return fail("ctx and dev_ptr must not be NULL.");
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    MDB_HIP_CHECK(hipMalloc(dev_ptr, bytes ? bytes : 256));
    return 0;
}

int mdb_dev_free(mdb_ctx *ctx, void *dev_ptr) {
    if (!ctx) return fail("ctx must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MDB_HIP_CHECK(hipFree(dev_ptr));
    return 0;
}

int mdb_dev_upload(mdb_ctx *ctx, void *dev_dst, const void *host_src, uint64_t bytes) {
    if (!ctx) return fail("ctx must not be NULL.");
    if (bytes == 0) return 0;
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    MDB_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int mdb_dev_download(mdb_ctx *ctx, void *host_dst, const void *dev_src, uint64_t bytes) {
    if (!ctx) return fail("ctx must not be NULL.");
    if (bytes == 0) return 0;
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    MDB_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int mdb_dev_sync(mdb_ctx *ctx) {
    if (!ctx) return fail("ctx must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// Layout of an uploaded batch: one device blob holding, 256-byte aligned each, the five primitive
// columns, the three view arrays, every variadic data buffer, and three pointer tables.
int mdb_segments_upload(mdb_ctx *ctx, const mdb_segments *host, mdb_segments_owned **out) {
    if (!ctx || !host || !out) return fail("ctx, host and out must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    return mdb::upload_segments_locked(ctx, host, false, out);
}

} // extern "C"

// transient: the blob is the context's SCRATCH_UPLOAD slot instead of an allocation of its own, valid
// until the next transient upload on this context (the caller holds ctx->mutex from the upload to the
// end of the launch that reads it). hipMalloc + hipFree per batch would cost more than the kernels of a
// small batch, and hipFree waits for the WHOLE device: with it, two contexts could never overlap the copy
// of one batch with the kernels of the next (36 GB/s instead of the 56 GB/s the link gives).
int mdb::upload_segments_locked(mdb_ctx *ctx, const mdb_segments *host, bool transient, mdb_segments_owned **out) {
    return upload_segment_list_locked(ctx, &host, 1, transient, out);
}

// Several host batches as ONE device batch, rows in the order of the list (what a GridStream that has been
// handed several 8 192-row RecordBatches by its input gives to one launch): the primitive
// columns and the views are laid end to end, every batch's data buffers follow those of the batches before it,
// and the out-of-line views of the later batches are moved onto their buffers' new indices while they sit in
// the staging block.
int mdb::upload_segment_list_locked(mdb_ctx *ctx, const mdb_segments *const *hosts, uint32_t n_hosts, bool transient,
                                    mdb_segments_owned **out) {
    uint64_t n = 0;
    int64_t n_buffers_total[3] = {0, 0, 0};
    for (uint32_t h = 0; h < n_hosts; h++) {
        const mdb_segments *host = hosts[h];
        if (!host) return fail("A batch of the list is NULL.");
        const mdb_binview_col *cols[3] = {&host->timestamps, &host->values, &host->residuals};
        for (int c = 0; c < 3; c++) {
            if (cols[c]->n_buffers < 0) return fail("n_buffers must not be negative.");
            if (cols[c]->n_buffers > 0 && (!cols[c]->buffers || !cols[c]->buffer_sizes))
                return fail("buffers and buffer_sizes must be given when n_buffers > 0.");
            if (host->n > 0 && !cols[c]->views) return fail("views must not be NULL.");
            if (validate_views_host(*cols[c], host->n)) return 1;
            n_buffers_total[c] += cols[c]->n_buffers;
        }
        if (host->n > 0 && (!host->model_type_id || !host->start_time || !host->end_time || !host->min_value ||
                            !host->max_value))
            return fail("The primitive columns must not be NULL.");
        n += host->n;
    }
    for (int c = 0; c < 3; c++)
        if (n_buffers_total[c] > 0x7fffffff) return fail("Too many data buffers in one column.");

    struct Piece {
        const void *src;
        uint64_t bytes;
        uint64_t offset;
    };
    std::vector<Piece> pieces;
    uint64_t cursor = 0;
    // One column of all batches end to end: `width` bytes per row.
    auto add_column = [&](auto member, uint64_t width) {
        const uint64_t offset = cursor;
        uint64_t at = cursor;
        for (uint32_t h = 0; h < n_hosts; h++) {
            pieces.push_back({member(hosts[h]), width * hosts[h]->n, at});
            at += width * hosts[h]->n;
        }
        cursor = align_up(at, 256);
        return offset;
    };
    auto add = [&](const void *src, uint64_t bytes) {
        uint64_t offset = cursor;
        pieces.push_back({src, bytes, offset});
        cursor = align_up(cursor + bytes, 256);
        return offset;
    };
    auto column_of = [](const mdb_segments *host, int c) {
        return c == 0 ? &host->timestamps : (c == 1 ? &host->values : &host->residuals);
    };
    uint64_t off_type = add_column([](const mdb_segments *s) { return (const void *)s->model_type_id; }, 1);
    uint64_t off_start = add_column([](const mdb_segments *s) { return (const void *)s->start_time; }, 8);
    uint64_t off_end = add_column([](const mdb_segments *s) { return (const void *)s->end_time; }, 8);
    uint64_t off_min = add_column([](const mdb_segments *s) { return (const void *)s->min_value; }, 4);
    uint64_t off_max = add_column([](const mdb_segments *s) { return (const void *)s->max_value; }, 4);
    uint64_t off_views[3];
    std::vector<uint64_t> off_buffers[3];
    std::vector<int64_t> all_sizes[3];
    std::vector<int64_t> span_starts[3]; // what of every data buffer travels: [start, start + size) of the original
    uint64_t off_tables[3];
    for (int c = 0; c < 3; c++) {
        off_views[c] = add_column([&](const mdb_segments *s) { return (const void *)column_of(s, c)->views; }, 16);
        for (uint32_t h = 0; h < n_hosts; h++) {
            // Only the bytes the batch's views point at: a batch that is a slice of a larger array (arrow's
            // RecordBatch::slice, the batches a Parquet page is cut into) shares that array's data buffers,
            // hundreds of megabytes of which belong to other rows.
            const mdb_binview_col *col = column_of(hosts[h], c);
            std::vector<int64_t> low((size_t)col->n_buffers, INT64_MAX), high((size_t)col->n_buffers, 0);
            for (uint64_t i = 0; i < hosts[h]->n; i++) {
                const mdb_view16 &view = col->views[i];
                if (view.length <= 12) continue;
                const size_t b = (size_t)view.u.ref.buffer_index; // (validate_views_host has been through them)
                low[b] = std::min<int64_t>(low[b], view.u.ref.offset);
                high[b] = std::max<int64_t>(high[b], (int64_t)view.u.ref.offset + view.length);
            }
            for (int b = 0; b < col->n_buffers; b++) {
                const int64_t start = high[(size_t)b] > 0 ? (low[(size_t)b] & ~int64_t(15)) : 0; // (alignment kept)
                const int64_t size = high[(size_t)b] > 0 ? high[(size_t)b] - start : 0;
                off_buffers[c].push_back(add(col->buffers[b] + start, (uint64_t)size));
                all_sizes[c].push_back(size);
                span_starts[c].push_back(start);
            }
        }
    }
    for (int c = 0; c < 3; c++) off_tables[c] = add(nullptr, 8 * (uint64_t)(n_buffers_total[c] + 1));
    uint64_t total = cursor ? cursor : 256;

    OwnedSegments *owned = new OwnedSegments();
    void *blob = nullptr;
    if (transient) {
        if (scratch_reserve(ctx, SCRATCH_UPLOAD, total, &blob)) {
            delete owned;
            return 1;
        }
    } else {
        if (hipMalloc(&blob, total) != hipSuccess) {
            delete owned;
            return fail("hipMalloc of " + std::to_string(total) + " bytes failed.");
        }
        owned->device_allocs.push_back(blob);
    }
    owned->device = ctx->device;
    uint8_t *dev = static_cast<uint8_t *>(blob);

    void *stage_v = nullptr;
    if (pinned_reserve(ctx, total, &stage_v)) {
        if (!transient) (void)hipFree(blob);
        delete owned;
        return 1;
    }
    uint8_t *stage = static_cast<uint8_t *>(stage_v);
    // (a batch with megabytes of out-of-line payloads - MacaqueV streams - is copied by the host threads in shares of a
    // megabyte: one thread moves 37 MB in 4 ms, as long as the batch's points then take across PCIe)
    struct StagingCopy {
        std::vector<Piece> shares;
        uint8_t *stage;
        std::atomic<size_t> next{0};
    } copy;
    copy.stage = stage;
    uint64_t copied_bytes = 0;
    for (auto &p : pieces) {
        if (!p.src || !p.bytes) continue;
        copied_bytes += p.bytes;
        for (uint64_t at = 0; at < p.bytes; at += 1u << 20)
            copy.shares.push_back({static_cast<const uint8_t *>(p.src) + at, std::min<uint64_t>(1u << 20, p.bytes - at), p.offset + at});
    }
    auto copy_share = [](unsigned, void *arg) {
        StagingCopy &job = *static_cast<StagingCopy *>(arg);
        for (size_t k = job.next.fetch_add(1); k < job.shares.size(); k = job.next.fetch_add(1))
            std::memcpy(job.stage + job.shares[k].offset, job.shares[k].src, job.shares[k].bytes);
    };
    if (copied_bytes >= (8u << 20)) host_parallel(std::min<unsigned>(host_parallel_width(), 8), copy_share, &copy);
    else copy_share(0, &copy);
    for (int c = 0; c < 3; c++) {
        uint64_t *table = reinterpret_cast<uint64_t *>(stage + off_tables[c]);
        for (int64_t b = 0; b < n_buffers_total[c]; b++)
            table[b] = reinterpret_cast<uint64_t>(dev + off_buffers[c][(size_t)b]);
        table[n_buffers_total[c]] = 0;
        // the views onto their buffers' places in the joint table and onto the part of each buffer that travels
        int32_t first_buffer = 0;
        uint64_t first_row = 0;
        for (uint32_t h = 0; h < n_hosts; h++) {
            const int32_t n_buffers_here = column_of(hosts[h], c)->n_buffers;
            bool moved = first_buffer > 0;
            for (int32_t b = 0; b < n_buffers_here; b++) moved = moved || span_starts[c][(size_t)(first_buffer + b)] != 0;
            if (moved) {
                mdb_view16 *views = reinterpret_cast<mdb_view16 *>(stage + off_views[c]) + first_row;
                for (uint64_t i = 0; i < hosts[h]->n; i++) {
                    if (views[i].length <= 12) continue;
                    const int32_t joint = views[i].u.ref.buffer_index + first_buffer;
                    views[i].u.ref.offset -= (int32_t)span_starts[c][(size_t)joint];
                    views[i].u.ref.buffer_index = joint;
                }
            }
            first_buffer += n_buffers_here;
            first_row += hosts[h]->n;
        }
    }
    if (hipMemcpyAsync(dev, stage, total, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        if (!transient) (void)hipFree(blob);
        delete owned;
        return fail("hipMemcpy host to device failed.");
    }

    // Host-side copy of the buffer sizes so a later download knows how much to copy.
    owned->host_allocs.resize(3);
    mdb_segments &s = owned->c.seg;
    s.n = n;
    s.model_type_id = reinterpret_cast<const int8_t *>(dev + off_type);
    s.start_time = reinterpret_cast<const int64_t *>(dev + off_start);
    s.end_time = reinterpret_cast<const int64_t *>(dev + off_end);
    s.min_value = reinterpret_cast<const float *>(dev + off_min);
    s.max_value = reinterpret_cast<const float *>(dev + off_max);
    mdb_binview_col *out_cols[3] = {&s.timestamps, &s.values, &s.residuals};
    for (int c = 0; c < 3; c++) {
        owned->host_allocs[c].resize(8 * (size_t)(n_buffers_total[c] + 1));
        int64_t *sizes = reinterpret_cast<int64_t *>(owned->host_allocs[c].data());
        for (int64_t b = 0; b < n_buffers_total[c]; b++) sizes[b] = all_sizes[c][(size_t)b];
        out_cols[c]->views = reinterpret_cast<const mdb_view16 *>(dev + off_views[c]);
        out_cols[c]->buffers = reinterpret_cast<const uint8_t *const *>(dev + off_tables[c]);
        out_cols[c]->buffer_sizes = sizes;
        out_cols[c]->n_buffers = (int32_t)n_buffers_total[c];
    }
    owned->c.error = nullptr;
    owned->c.chunk_index = nullptr;
    owned->c.on_device = 1;
    owned->c.priv_ = owned;
    if (!transient) owned_segments_register(owned);
    *out = &owned->c;
    return 0;
}

// The profile of a context covers the clones its grid pipeline runs jobs on (mdb_grid_submit): they are switched
// and cleared with it and their launches are counted as its own. (Lock order: the context, then one clone at a
// time; a pipeline worker holds the lock of its own context only.)
namespace {
template <typename F> int with_context_and_pipeline_clones(mdb_ctx *ctx, F per_context) {
    mdb_ctx *all[1 + 8] = {ctx};
    const int n = 1 + (ctx->is_clone ? 0 : mdb::pipeline_clones(ctx, all + 1, 8));
    for (int k = 0; k < n; k++) {
        mdb::CallGuard lock(all[k]);
        if (k > 0) MDB_HIP_CHECK(hipSetDevice(all[k]->device));
        if (profile_collect(all[k])) return 1;
        per_context(all[k]);
    }
    return 0;
}
} // namespace

extern "C" {

int mdb_segments_download(mdb_ctx *ctx, const mdb_segments_owned *dev, mdb_segments_owned **out) {
    if (!ctx || !dev || !out) return fail("ctx, dev and out must not be NULL.");
    if (!dev->on_device) return fail("The batch is already in host memory.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    const mdb_segments &d = dev->seg;
    const uint64_t n = d.n;
    OwnedSegments *owned = new OwnedSegments();
    // 0 type, 1 start, 2 end, 3 min, 4 max, 5-7 views, 8-10 data, 11 error, 12 chunk_index
    owned->host_allocs.resize(13);
    auto fetch = [&](int slot, const void *src, uint64_t bytes) -> int {
        owned->host_allocs[slot].resize(bytes);
        if (bytes == 0 || !src) return 0;
        MDB_HIP_CHECK(mail_read(ctx, owned->host_allocs[slot].data(), src, bytes));
        return 0;
    };
    int rc = 0;
    rc |= fetch(0, d.model_type_id, n);
    rc |= fetch(1, d.start_time, 8 * n);
    rc |= fetch(2, d.end_time, 8 * n);
    rc |= fetch(3, d.min_value, 4 * n);
    rc |= fetch(4, d.max_value, 4 * n);
    const mdb_binview_col *cols[3] = {&d.timestamps, &d.values, &d.residuals};
    // Pointer tables live on the device; read them back to find each data buffer.
    std::vector<uint64_t> tables[3];
    for (int c = 0; c < 3 && !rc; c++) {
        rc |= fetch(5 + c, cols[c]->views, 16 * n);
        tables[c].resize((size_t)cols[c]->n_buffers);
        if (cols[c]->n_buffers > 0 &&
            mail_read(ctx, tables[c].data(), cols[c]->buffers, 8 * (size_t)cols[c]->n_buffers) != hipSuccess)
            rc = fail("hipMemcpy of the buffer table failed.");
    }
    if (!rc && mail_sync(ctx) != hipSuccess) rc = fail("stream sync failed.");
    // The data buffers of each column: one host buffer with the views rebased onto it while everything
    // fits below 2 GiB (what most consumers prefer), else buffer by buffer as they are on the device.
    // host_allocs[8 + c] is the single buffer, or the first of several; further ones go to the end.
    std::vector<size_t> buffer_slots[3];
    // (MDB_SEGMENTS_MERGE_LIMIT, read once per download: tests of the several-buffers path without 2 GiB of payloads)
    uint64_t merge_limit = 0x7fffffffull;
    if (const char *text = option_text("MDB_SEGMENTS_MERGE_LIMIT"))
        merge_limit = std::min<uint64_t>(merge_limit, (uint64_t)std::max(0ll, std::atoll(text)));
    for (int c = 0; c < 3 && !rc; c++) {
        uint64_t total = 0;
        std::vector<uint64_t> base((size_t)cols[c]->n_buffers);
        for (int b = 0; b < cols[c]->n_buffers; b++) {
            base[b] = total;
            total += (uint64_t)cols[c]->buffer_sizes[b];
        }
        // (MDB_SEGMENTS_MERGE_LIMIT: tests of the several-buffers path without 2 GiB of payloads)
        const bool merged = total <= merge_limit;
        if (merged) {
            owned->host_allocs[8 + c].resize(total);
            buffer_slots[c].push_back(8 + (size_t)c);
        }
        for (int b = 0; b < cols[c]->n_buffers && !rc; b++) {
            const uint64_t bytes = (uint64_t)cols[c]->buffer_sizes[b];
            uint8_t *to;
            if (merged) {
                to = owned->host_allocs[8 + c].data() + base[b];
            } else {
                size_t slot = 8 + (size_t)c;
                if (b > 0) {
                    owned->host_allocs.emplace_back();
                    slot = owned->host_allocs.size() - 1;
                }
                owned->host_allocs[slot].resize(bytes);
                buffer_slots[c].push_back(slot);
                to = owned->host_allocs[slot].data();
            }
            if (bytes && mail_read(ctx, to, reinterpret_cast<const void *>(tables[c][b]), bytes) != hipSuccess)
                rc = fail("hipMemcpy of a data buffer failed.");
        }
        if (!rc) {
            mdb_view16 *views = reinterpret_cast<mdb_view16 *>(owned->host_allocs[5 + c].data());
            for (uint64_t i = 0; i < n; i++) {
                if (views[i].length > 12) {
                    const int32_t buffer = views[i].u.ref.buffer_index;
                    if (buffer < 0 || buffer >= cols[c]->n_buffers) {
                        rc = fail("Malformed BinaryView: buffer index out of range.");
                        break;
                    }
                    if (merged && cols[c]->n_buffers > 1) {
                        // (total <= 2 GiB here, so the rebased offset fits)
                        views[i].u.ref.offset = (int32_t)((int64_t)views[i].u.ref.offset + (int64_t)base[(size_t)buffer]);
                        views[i].u.ref.buffer_index = 0;
                    }
                }
            }
        }
    }
    if (!rc && dev->error) rc |= fetch(11, dev->error, 4 * n);
    if (!rc && dev->chunk_index) rc |= fetch(12, dev->chunk_index, 4 * n);
    if (!rc && mail_sync(ctx) != hipSuccess) rc = fail("stream sync failed.");
    if (rc) {
        mail_drop(ctx); // (small columns may be on their way into this frame's vectors)
        delete owned;
        return 1;
    }
    mdb_segments &s = owned->c.seg;
    s.n = n;
    s.model_type_id = reinterpret_cast<const int8_t *>(owned->host_allocs[0].data());
    s.start_time = reinterpret_cast<const int64_t *>(owned->host_allocs[1].data());
    s.end_time = reinterpret_cast<const int64_t *>(owned->host_allocs[2].data());
    s.min_value = reinterpret_cast<const float *>(owned->host_allocs[3].data());
    s.max_value = reinterpret_cast<const float *>(owned->host_allocs[4].data());
    mdb_binview_col *out_cols[3] = {&s.timestamps, &s.values, &s.residuals};
    for (int c = 0; c < 3; c++) {
        for (size_t slot : buffer_slots[c]) { // (host_allocs no longer grows: the pointers stay)
            owned->buffer_ptrs[c].push_back(owned->host_allocs[slot].data());
            owned->buffer_sizes[c].push_back((int64_t)owned->host_allocs[slot].size());
        }
        out_cols[c]->views = reinterpret_cast<const mdb_view16 *>(owned->host_allocs[5 + c].data());
        out_cols[c]->buffers = owned->buffer_ptrs[c].data();
        out_cols[c]->buffer_sizes = owned->buffer_sizes[c].data();
        out_cols[c]->n_buffers = (int32_t)owned->buffer_ptrs[c].size();
    }
    owned->c.error = dev->error ? reinterpret_cast<const float *>(owned->host_allocs[11].data()) : nullptr;
    owned->c.chunk_index =
        dev->chunk_index ? reinterpret_cast<const uint32_t *>(owned->host_allocs[12].data()) : nullptr;
    owned->c.on_device = 0;
    owned->c.priv_ = owned;
    *out = &owned->c;
    return 0;
}

int mdb_segments_validate_dev(mdb_ctx *ctx, const mdb_segments *dev) {
    if (!ctx || !dev) return fail("ctx and dev must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    const mdb_binview_col *cols[3] = {&dev->timestamps, &dev->values, &dev->residuals};
    if (dev->n == 0) return 0;
    void *p = nullptr;
    if (scratch_reserve(ctx, SCRATCH_HEADER, sizeof(unsigned int) * 64, &p)) return 1;
    unsigned int *flag = static_cast<unsigned int *>(p);
    MDB_HIP_CHECK(hipMemsetAsync(flag, 0, 4, ctx->stream));
    for (int c = 0; c < 3; c++) {
        const int32_t n_buffers = cols[c]->n_buffers;
        if (n_buffers < 0) return fail("n_buffers must not be negative.");
        if (n_buffers > 0 && !cols[c]->buffer_sizes) return fail("buffer_sizes must be given when n_buffers > 0.");
        if (!cols[c]->views) return fail("views must not be NULL.");
        for (int32_t b = 0; b < n_buffers; b++)
            if (cols[c]->buffer_sizes[b] < 0) return fail("Malformed BinaryView: negative buffer size.");
        if (scratch_reserve(ctx, SCRATCH_STAGE_DEV, 8 * (uint64_t)(n_buffers + 1), &p)) return 1;
        long long *sizes = static_cast<long long *>(p);
        if (n_buffers > 0)
            MDB_HIP_CHECK(hipMemcpyAsync(sizes, cols[c]->buffer_sizes, 8 * (size_t)n_buffers, hipMemcpyHostToDevice,
                                         ctx->stream));
        hipLaunchKernelGGL(k_validate_views, dim3((uint32_t)((dev->n + 255) / 256)), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const uint4 *>(cols[c]->views), dev->n, sizes, n_buffers, flag);
        // (the pageable copy above has completed on return; the kernel reads `sizes` before the next
        // column overwrites it because everything is in stream order)
    }
    unsigned int bad = 0;
    MDB_HIP_CHECK(hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MDB_HIP_CHECK(hipGetLastError());
    if (bad) return fail("Malformed BinaryView: " + std::to_string(bad) + " views point outside the data buffers.");
    return 0;
}

void mdb_segments_free(mdb_segments_owned *segments) {
    if (!segments) return;
    OwnedSegments *owned = static_cast<OwnedSegments *>(segments->priv_);
    if (!owned) return;
    owned_segments_forget(owned);
    if (owned->device >= 0 && !owned->device_allocs.empty()) {
        (void)hipSetDevice(owned->device);
        (void)hipDeviceSynchronize();
        for (void *p : owned->device_allocs) (void)hipFree(p);
    }
    delete owned;
}

int mdb_is_value_within_error_bound(mdb_error_bound eb, float real_value, float approximate_value,
                                    int32_t *within) {
    if (!within) return fail("within must not be NULL.");
    // models/mod.rs:53-77, the arithmetic of within_error_bound in mdb_fit.hip on the host: equal (or
    // both NaN) first, then the f32 comparison of the bound's kind.
    const double real = (double)real_value, approximate = (double)approximate_value;
    bool result;
    if (real == approximate || (real != real && approximate != approximate)) {
        result = true;
    } else if (eb.kind == MDB_EB_ABSOLUTE) {
        result = fabsf(real_value - approximate_value) <= eb.value;
    } else if (eb.kind == MDB_EB_RELATIVE) {
        const float difference = real_value - approximate_value;
        const float ratio = fabsf(difference / real_value);
        result = (ratio * 100.0f) <= eb.value;
    } else if (eb.kind == MDB_EB_LOSSLESS) {
        result = false;
    } else {
        return fail("Invalid error bound.");
    }
    *within = result ? 1 : 0;
    return 0;
}

int mdb_are_compressed_timestamps_regular(const uint8_t *compressed_timestamps, uint64_t n_bytes,
                                          int32_t *regular) {
    if (!regular || (n_bytes > 0 && !compressed_timestamps))
        return fail("regular and compressed_timestamps must not be NULL.");
    // timestamps.rs:199-202
    *regular = (n_bytes == 0 || (compressed_timestamps[0] & 128u) == 0) ? 1 : 0;
    return 0;
}

int mdb_profile_enable(mdb_ctx *ctx, int enabled) {
    if (!ctx) return fail("ctx must not be NULL.");
    return with_context_and_pipeline_clones(ctx, [&](mdb_ctx *c) { c->profiling = enabled != 0; });
}

int mdb_profile_reset(mdb_ctx *ctx) {
    if (!ctx) return fail("ctx must not be NULL.");
    return with_context_and_pipeline_clones(ctx, [&](mdb_ctx *c) { c->kernel_times.clear(); });
}

int mdb_profile_get(mdb_ctx *ctx, const char *name, uint64_t *launches, double *total_ms) {
    if (!ctx || !name) return fail("ctx and name must not be NULL.");
    uint64_t n = 0;
    double ms = 0.0;
    if (with_context_and_pipeline_clones(ctx, [&](mdb_ctx *c) {
            auto it = c->kernel_times.find(name);
            if (it == c->kernel_times.end()) return;
            n += it->second.launches;
            ms += it->second.total_ms;
        }))
        return 1;
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    return 0;
}

int mdb_profile_names(mdb_ctx *ctx, char *out, uint64_t cap) {
    if (!ctx || !out || cap == 0) return fail("ctx and out must not be NULL.");
    std::set<std::string> names;
    if (with_context_and_pipeline_clones(ctx, [&](mdb_ctx *c) {
            for (auto &kv : c->kernel_times) names.insert(kv.first);
        }))
        return 1;
    std::string joined;
    for (const std::string &name : names) {
        if (!joined.empty()) joined += "\n";
        joined += name;
    }
    std::strncpy(out, joined.c_str(), cap - 1);
    out[cap - 1] = 0;
    return 0;
}

} // extern "C"
