// mdb_common.hpp - context, error handling, launch/profiling helpers and device-side primitives
// shared by the HIP translation units of libmdb_hip.so. gfx950 (MI355X) only: wave64, no
// compatibility paths. Everything on the value path is built with -ffp-contract=off because the
// reference (Rust) never fuses a*b+c.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mdb.h"
#include "mdb_host_side.hpp"

namespace mdb {

#define MDB_HIP_CHECK(expr)                                                                        \
    do {                                                                                           \
        hipError_t mdb_err_ = (expr);                                                              \
        if (mdb_err_ != hipSuccess)                                                                \
            return ::mdb::fail(std::string(#expr) + ": " + hipGetErrorString(mdb_err_));          \
    } while (0)

// ---- context -----------------------------------------------------------------------------------

struct KernelTime {
    uint64_t launches = 0;
    double total_ms = 0.0;
};

struct PendingEvent {
    std::string name;
    hipEvent_t start;
    hipEvent_t stop;
};

enum ScratchSlot {
    SCRATCH_DESC = 0,
    SCRATCH_COUNTS,
    SCRATCH_OFFSETS,
    SCRATCH_BLOCK_SUMS,
    SCRATCH_SERIAL_IDS,
    SCRATCH_TILE_MAP,
    SCRATCH_HEADER,
    SCRATCH_AGG_PARTIALS,
    SCRATCH_FIT_A,
    SCRATCH_FIT_B,
    SCRATCH_FIT_C,
    SCRATCH_FIT_D,
    SCRATCH_FIT_E,
    SCRATCH_FIT_F,
    SCRATCH_FIT_SPLIT,
    SCRATCH_FIT_GAP,
    SCRATCH_FIT_REGULAR,
    SCRATCH_FIT_TS,
    SCRATCH_MV,
    SCRATCH_AGG_MV,
    SCRATCH_STAGE_DEV,
    SCRATCH_COMM,
    SCRATCH_TS_BASE,
    SCRATCH_TS_SLOTS,
    SCRATCH_FIT_BASES,
    SCRATCH_UPLOAD,
    SCRATCH_PENDING,
    SCRATCH_TS_LEFT,
    SCRATCH_FIT_IN_TS,
    SCRATCH_FIT_IN_VALUES,
    SCRATCH_FIT_IN_OFFSETS,
    SCRATCH_FIT_WAVE,
    SCRATCH_FIT_REJECTS,
    SCRATCH_MV_HOST_INDEX,
    SCRATCH_FIT_SMALL,
    SCRATCH_AGG_CHAIN_LIST,
    SCRATCH_FIT_LONG_IDS,
    SCRATCH_FIT_LONG,
    SCRATCH_FIT_ROTATION,
    SCRATCH_FIT_GAP_STAGE_OFFSETS,
    SCRATCH_FIT_GAP_STAGE,
    SCRATCH_SLOT_COUNT
};

struct CloneCache {
    std::mutex mutex;
    std::vector<mdb_ctx *> idle;
    bool origin_closed = false;
};


} // namespace mdb

struct mdb_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::mutex mutex;
    int compute_units = 0;

    void *scratch[mdb::SCRATCH_SLOT_COUNT] = {};
    uint64_t scratch_bytes[mdb::SCRATCH_SLOT_COUNT] = {};
    void *pinned = nullptr; // pinned host staging
    uint64_t pinned_bytes = 0;
    // A few hundred KB of page-locked memory for the SMALL copies of a call (mail_read / mail_write / mail_sync): a
    // copy of eight bytes to or from pageable memory is 80 microseconds of the runtime's staging, thirty of them were
    // 2.3 of the general fit driver's 2.5 ms for one chunk.
    unsigned char *mail = nullptr;
    bool mail_failed = false;
    uint64_t mail_used = 0;
    struct MailRead { void *to; uint64_t at, bytes; };
    std::vector<MailRead> mail_reads;

    std::shared_ptr<mdb::PinnedPool> pinned_pool; // the device's pool of page-locked result blocks (mdb_init)
    bool owns_pinned_pool = false;
    uint64_t scratch_limit = 0;   // mdb_set_scratch_limit: device scratch kept between calls (0: all of it)
    // mdb_clone / mdb_close of a clone: closed clones wait here for the next mdb_clone of the same context (a
    // stream costs 2-6 ms to make and as much to destroy, an operator asks for its second context per query).
    std::shared_ptr<mdb::CloneCache> clones; // shared by a context and its clones
    bool is_clone = false;
    mdb::GridPipeline *pipeline = nullptr; // made by the first mdb_grid_submit, ended by mdb_close
    std::mutex pipeline_mutex;             // guards `pipeline` (not `mutex`: a running job holds that one)

    // RCCL communicator of mdb_comm_init (an ncclComm_t; rccl.h stays out of this header).
    void *comm = nullptr;
    int comm_rank = 0;
    int comm_world = 0;

    bool profiling = false;
    std::map<std::string, mdb::KernelTime> kernel_times;
    std::vector<mdb::PendingEvent> pending_events;
    std::vector<hipEvent_t> event_pool;
};

namespace mdb {

int validate_views_host(const mdb_binview_col &col, uint64_t n);
void merge_agg_state(mdb_agg_state *into, const mdb_agg_state &from);

// Grow-only device scratch, one allocation per slot.
int scratch_reserve(mdb_ctx *ctx, ScratchSlot slot, uint64_t bytes, void **out);
int pinned_reserve(mdb_ctx *ctx, uint64_t bytes, void **out);
// Small copies through the context's page-locked mailbox, in the order of ctx->stream:
//   mail_read(ctx, host, dev, n): `host` holds the bytes after the next mail_sync(ctx) (not before!);
//   mail_write(ctx, dev, host, n): `host` may be changed or freed at once;
//   mail_sync(ctx): hipStreamSynchronize(ctx->stream), then the reads are delivered.
// Copies that do not fit (or a context without a mailbox) go the plain way, which mail_sync covers as well. A function
// that uses mail_read must not leave pending reads behind: every path out of it goes through mail_sync or mail_drop.
constexpr uint64_t MAIL_BYTES = 512u << 10, MAIL_COPY_LIMIT = 64u << 10;
hipError_t mail_read(mdb_ctx *ctx, void *host_to, const void *dev_from, uint64_t bytes);
hipError_t mail_write(mdb_ctx *ctx, void *dev_to, const void *host_from, uint64_t bytes);
hipError_t mail_sync(mdb_ctx *ctx);
void mail_drop(mdb_ctx *ctx); // (an error path: forget what is pending)
// The lock every entry point takes on its context. When the call is over it gives back device scratch beyond
// the context's limit (mdb_set_scratch_limit), largest allocations first.
void scratch_enforce_limit(mdb_ctx *ctx);
struct CallGuard {
    mdb_ctx *ctx;
    explicit CallGuard(mdb_ctx *c) : ctx(c) { ctx->mutex.lock(); }
    ~CallGuard() {
        if (ctx->scratch_limit) scratch_enforce_limit(ctx);
        ctx->mutex.unlock();
    }
    CallGuard(const CallGuard &) = delete;
    CallGuard &operator=(const CallGuard &) = delete;
};
// mdb_segments_upload with ctx->mutex held; transient = into the context's upload scratch (mdb_ctx.hip).
int upload_segments_locked(mdb_ctx *ctx, const mdb_segments *host, bool transient, mdb_segments_owned **out);
// The same for several host batches that become one device batch, rows in the order of the list.
int upload_segment_list_locked(mdb_ctx *ctx, const mdb_segments *const *hosts, uint32_t n_hosts, bool transient,
                               mdb_segments_owned **out);

// Brackets a launch with events when profiling is on.
struct LaunchTimer {
    mdb_ctx *ctx;
    const char *name;
    hipEvent_t start = nullptr;
    hipEvent_t stop = nullptr;
    LaunchTimer(mdb_ctx *c, const char *n);
    ~LaunchTimer();
};

// Random access into the MacaqueV streams of a batch that stays on the device (mdb_grid.hip: mv_index_*): one
// 32-byte cursor in front of every 64th value of every stream - where the code begins, the window, the XOR of all
// deltas so far - so that every piece is decoded by a lane of its own. Not an Arrow column: it belongs to the
// mdb_segments_owned the library made (mdb_segments_upload, mdb_compress_chunks_dev), is built by the first
// grid / aggregate call that could use it and dies with the batch.
struct MvIndex {
    std::mutex mutex;      // building
    bool built = false;
    bool usable = false;   // false: a stream of the batch is malformed, or there is nothing to index
    int device = 0;
    unsigned long long n_pieces = 0;
    void *cursors = nullptr;    // MvCursor[n_pieces]
    void *piece_base = nullptr; // unsigned long long[n + 1]: first piece of every segment
    unsigned long long stream_values = 0; // values behind the cursors
    // The same for the batch's delta-of-delta TIMESTAMP streams: what the counting walk of a grid call without a
    // time range leaves behind (k_grid_ts_count: a cursor per 256 bits of stream, the jump lists, the list of
    // pieces still to decode, every stream's number of points), kept from the first such call for the later ones.
    bool ts_built = false;
    bool ts_jumps = false;          // (built with jump lists: MDB_GRID_TS_JUMPS)
    unsigned long long ts_n_pieces = 0, ts_live_pieces = 0;
    void *ts_piece_base = nullptr;  // unsigned long long[n + 1]
    void *ts_slots = nullptr;       // the block TsCheckpoints points into
    void *ts_totals = nullptr;      // uint32_t[n + 4]
    // And for the aggregates without a time range (agg_run): every irregular segment's number of points and, once a
    // call has asked for sums, the sum of every Swing segment among them (ts_walk_for_aggregates), kept likewise.
    bool agg_walk_built = false, agg_walk_with_sums = false;
    void *agg_walk_totals = nullptr; // uint32_t[n]
    void *agg_walk_sums = nullptr;   // double[n]
    // And for the aggregates UNDER a time range: what ALL points of every PMC-Mean / Swing segment with irregular
    // timestamps and no residuals add up to (k_grid_ts_count<WALK_RANGE> over the whole time axis, once) with every
    // irregular segment's number of points: a query then walks only the segments its range cuts
    // (ts_range_from_kept, mdb_grid.hip).
    // (24 B + 4 B per segment here, 24 B more for range_acc below: 52 B per segment for the life of the batch from its
    // first aggregate under a range on - 440 MB for 8.5 M segments; MDB_GRID_TS_CACHE=0 keeps none of it. If the memory
    // is not there the query walks what it needs every time: range_whole_failed.)
    bool range_whole_built = false;
    bool range_whole_failed = false;    // a malformed stream or no memory: not tried again
    void *range_whole = nullptr;        // TsWalkRange[n]
    void *range_whole_totals = nullptr; // uint32_t[n]
    // ... and what k_agg_range makes of ALL points of a segment (count < 0: leaves to the decoders), for the ranges
    // that contain it; `range_acc_key`: the line to the decoders it was made under (0: not made).
    // A query holds the array it reads (a copy of the pointer, taken under the mutex) for as long as its kernels run:
    // a rebuild under another key on another context makes a NEW array and leaves this one to its readers.
    // ([0]: calls on the resident batch, [1]: calls with cursors of their own - one array each, so they do not evict each other)
    std::shared_ptr<void> range_acc[2]; // TsWalkRange[n] (freed by its deleter)
    uint64_t range_acc_key[2] = {0, 0};
    bool range_acc_failed[2] = {false, false}; // no memory for it / a fault in the streams: not tried again
    // And for the sums of the MacaqueV streams (mv_index_stream_sums): where every wave of pieces lists the streams of
    // two pieces or more that begin in it (an exclusive scan over the waves, the long and the short kind packed into
    // one word, the totals behind the last wave) - a function of the cursors alone, counted by the first call that
    // asks and kept (8 B per 64 pieces).
    bool chains_built = false;
    void *chain_offsets = nullptr;          // unsigned long long[waves of pieces + 1]
    unsigned long long chains_listed = 0;   // the totals: long << 32 | short
    // The index of ONE call over host batches (mv_host_index, mdb_grid.hip): made by host threads while the batches
    // are on their way, for the long streams only - a segment without pieces is the serial kernel's - and living in
    // the context's scratch.
    bool of_one_call = false;
    ~MvIndex() {
        if (of_one_call) return;
        if (cursors || piece_base || ts_piece_base || ts_slots || ts_totals || agg_walk_totals || agg_walk_sums || range_whole ||
            range_whole_totals || chain_offsets) {
            (void)hipSetDevice(device);
            for (void *allocation : {cursors, piece_base, ts_piece_base, ts_slots, ts_totals, agg_walk_totals, agg_walk_sums,
                                     range_whole, range_whole_totals, chain_offsets})
                if (allocation) (void)hipFree(allocation);
        }
    }
};

// Owner bookkeeping behind mdb_segments_owned::priv_.
struct OwnedSegments {
    mdb_segments_owned c;
    int device = -1;                   // -1: host memory
    std::vector<void *> device_allocs; // hipFree'd on release
    std::vector<std::vector<uint8_t>> host_allocs;
    // host batches: per column, the data buffers (pointers into host_allocs) and their sizes
    std::vector<const uint8_t *> buffer_ptrs[3];
    std::vector<int64_t> buffer_sizes[3];
    std::shared_ptr<MvIndex> mv_index; // device batches that stay (not the transient uploads of one call)
};

// The device batches this library made and still owns, by the address of their `values` views: a "_dev" entry
// point is handed a plain mdb_segments (possibly a copy of the owned struct's member) and finds the batch's
// index through this. (mdb_ctx.hip)
void owned_segments_register(OwnedSegments *owned);
void owned_segments_forget(OwnedSegments *owned);
std::shared_ptr<MvIndex> owned_segments_index(const mdb_segments *in);

inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// ---- device primitives ---------------------------------------------------------------------------

#define MDB_WAVE 64

// The three 32-bit payload words of a BinaryView are [1..3] of the 16-byte view loaded as uint4.
__device__ __forceinline__ uint32_t view_inline_byte(const uint4 &view, uint32_t k) {
    // (two shifts and a select: a chain of selects between the three words is turned into an array in scratch
    // memory indexed by k / 4 - a store of the view and a dependent load per call, with a wait for EVERY load in flight)
    const uint64_t low = ((uint64_t)view.z << 32) | (uint64_t)view.y; // bytes 0..7
    const uint32_t from_low = (uint32_t)(low >> (8u * (k & 7u)));
    const uint32_t from_high = view.w >> (8u * (k & 3u));             // bytes 8..11
    return (k < 8 ? from_low : from_high) & 0xffu;
}

// A pointer the compiler has lost track of - read from the table of data buffers, or put together from
// an aligned address - is a "flat" one to it, and every load through a flat pointer makes the wave
// wait for ALL its outstanding memory operations (flat loads may come back out of order), which turns
// "one chunk loaded ahead" into a full trip to memory per chunk. Everything these readers touch is
// device (global) memory: saying so gives global_load instructions and counted waits.
template <typename T> __device__ __forceinline__ T load_global(const T *pointer) {
    typedef const __attribute__((address_space(1))) T *global_pointer;
    return *(global_pointer)(uintptr_t)pointer;
}
__device__ __forceinline__ uint4 load_global(const uint4 *pointer) { // (uint4 is a class: load the plain vector)
    typedef unsigned int plain4 __attribute__((ext_vector_type(4)));
    const plain4 v = load_global(reinterpret_cast<const plain4 *>(pointer));
    return make_uint4(v.x, v.y, v.z, v.w);
}

struct DevCol {
    const uint4 *views;
    const uint8_t *const *buffers;
    uint32_t n_buffers;
};

// Global-memory pointer to the bytes of row `row` (inline bytes live inside the view itself).
__device__ __forceinline__ const uint8_t *view_data(const DevCol &col, uint64_t row,
                                                    const uint4 &view) {
    int32_t length = (int32_t)view.x;
    if (length <= 12) return reinterpret_cast<const uint8_t *>(col.views + row) + 4;
    return col.buffers[(int32_t)view.z] + (int32_t)view.w;
}

// The same with the column's first data buffer already in hand (first_buffer(): a uniform load at the top of a kernel
// that waits for nothing): most columns have ONE buffer, and looking its address up in the table only once the view is
// known is a trip to memory in the middle of a chain of trips (cursor -> view -> table -> stream bytes).
__device__ __forceinline__ const uint8_t *first_buffer(const DevCol &col) { return col.n_buffers > 0 ? col.buffers[0] : nullptr; }
__device__ __forceinline__ const uint8_t *view_data(const DevCol &col, uint64_t row, const uint4 &view, const uint8_t *first) {
    int32_t length = (int32_t)view.x;
    if (length <= 12) return reinterpret_cast<const uint8_t *>(col.views + row) + 4;
    if ((int32_t)view.z == 0) return first + (int32_t)view.w;
    return col.buffers[(int32_t)view.z] + (int32_t)view.w;
}

struct DevSegments {
    uint64_t n;
    const int8_t *model_type_id;
    const int64_t *start_time;
    const int64_t *end_time;
    DevCol timestamps;
    const float *min_value;
    const float *max_value;
    DevCol values;
    DevCol residuals;
};

// mdb_grid.hip, for mdb_agg.hip: the MacaqueV decoders as a service to the aggregates (see there).
struct DeferredTotals {
    double sum = 0.0;
    long long count = 0;
    float min = 0.0f;
    float max = 0.0f;
};
uint32_t macaque_parallel_min_values(bool *forced);
struct DevSegments;
// (with a cursor index into the batch's MacaqueV streams, MvIndex: their f32 sums, 2 per segment)
int mv_index_ensure(mdb_ctx *ctx, const mdb_segments *in);
int mv_index_for_range(mdb_ctx *ctx, const mdb_segments *in, std::shared_ptr<MvIndex> *index, const unsigned long long **piece_base);
int mv_index_stream_sums(mdb_ctx *ctx, const mdb_segments *in, const DevSegments &s, const uint32_t *known_totals,
                         const float **stream_sums, const unsigned long long **only_with_pieces);

inline DevSegments to_dev(const mdb_segments *s) {
    DevSegments d;
    d.n = s->n;
    d.model_type_id = s->model_type_id;
    d.start_time = s->start_time;
    d.end_time = s->end_time;
    d.timestamps = {reinterpret_cast<const uint4 *>(s->timestamps.views), s->timestamps.buffers,
                    (uint32_t)std::max(s->timestamps.n_buffers, 0)};
    d.min_value = s->min_value;
    d.max_value = s->max_value;
    d.values = {reinterpret_cast<const uint4 *>(s->values.views), s->values.buffers, (uint32_t)std::max(s->values.n_buffers, 0)};
    d.residuals = {reinterpret_cast<const uint4 *>(s->residuals.views), s->residuals.buffers,
                   (uint32_t)std::max(s->residuals.n_buffers, 0)};
    return d;
}

// MSB-first bit reader over global memory (models/bits.rs:25-83 semantics). Loads aligned 16-byte
// chunks; a chunk that contains at least one payload byte never crosses a page, so touching the
// slack bytes of the first/last chunk is safe.
struct BitReaderDev {
    uint32_t next_word; // words consumed so far (of n_words)
    uint32_t n_words;
    uint64_t buffer; // MSB aligned
    int32_t available;
    uint64_t used_bits;
    uint64_t total_bits;
    // The payload is fetched 16 bytes at a time, one such chunk ahead of use. With one lane per stream
    // thousands of streams are open at once, far more than the caches hold a line for: a 4-byte load
    // per word would pull the same 128-byte line from memory again and again.
    const uint4 *chunks; // 16-byte aligned base
    uint32_t n_chunks;
    uint32_t next_chunk; // next one to load into `ahead`
    uint32_t in_chunk;   // next word of `current` to hand out
    uint4 current, ahead;

    __device__ __forceinline__ uint4 load_chunk(uint32_t index) const {
        return index < n_chunks ? load_global(chunks + index) : make_uint4(0u, 0u, 0u, 0u);
    }

    __device__ __forceinline__ uint32_t take_word() {
        const uint32_t w = in_chunk == 0 ? current.x : (in_chunk == 1 ? current.y : (in_chunk == 2 ? current.z : current.w));
        if (++in_chunk == 4) {
            current = ahead;
            ahead = load_chunk(next_chunk++);
            in_chunk = 0;
        }
        return w;
    }

    // Positions the reader `start_bit` bits into the payload (init() = seek(bytes, nbytes, 0)).
    __device__ __forceinline__ void seek(const uint8_t *bytes, uint64_t nbytes, uint64_t start_bit) {
        const uint64_t skip_bytes = (start_bit >> 3) & ~(uint64_t)3u; // whole words that are skipped
        init(bytes + skip_bytes, nbytes - skip_bytes);
        // init() measured everything from the word it started at; put the bookkeeping back on the
        // payload's own scale and drop the bits in front of start_bit.
        total_bits = nbytes * 8u;
        used_bits = skip_bytes * 8u;
        uint32_t drop = (uint32_t)(start_bit - used_bits); // < 64
        while (drop > 0) {
            const uint32_t step = drop > 32u ? 32u : drop;
            (void)get(step);
            drop -= step;
        }
    }

    __device__ __forceinline__ void init(const uint8_t *bytes, uint64_t nbytes) {
        const uintptr_t address = reinterpret_cast<uintptr_t>(bytes);
        const uint32_t misalign = (uint32_t)(address & 3u);
        const uint32_t skipped_words = (uint32_t)((address & 15u) >> 2);
        chunks = reinterpret_cast<const uint4 *>(address & ~(uintptr_t)15u);
        n_words = (uint32_t)((nbytes + misalign + 3u) >> 2);
        n_chunks = (skipped_words + n_words + 3u) >> 2;
        current = load_chunk(0);
        ahead = load_chunk(1);
        next_chunk = 2;
        in_chunk = skipped_words;
        next_word = 0;
        buffer = 0;
        available = 0;
        used_bits = 0;
        total_bits = nbytes * 8u;
        if (misalign) {
            refill();
            buffer <<= 8u * misalign;
            available -= 8 * (int32_t)misalign;
        }
    }

    __device__ __forceinline__ void refill() {
        while (available <= 32 && next_word < n_words) {
            uint32_t w = __builtin_bswap32(take_word());
            next_word++;
            buffer |= (uint64_t)w << (32 - available);
            available += 32;
        }
    }

    __device__ __forceinline__ uint64_t remaining() const { return total_bits - used_bits; }
    __device__ __forceinline__ bool exhausted() const { return used_bits >= total_bits; }

    // count in [0, 32]. Reads past the end return zeros and are flagged by overrun().
    __device__ __forceinline__ uint32_t get(uint32_t count) {
        if (count == 0) return 0;
        refill();
        uint32_t value = (uint32_t)(buffer >> (64u - count));
        buffer <<= count;
        available -= (int32_t)count;
        used_bits += count;
        return value;
    }

    __device__ __forceinline__ uint64_t get64(uint32_t count) {
        if (count <= 32) return get(count);
        uint64_t high = get(count - 32);
        return (high << 32) | get(32);
    }

    __device__ __forceinline__ bool overrun() const { return used_bits > total_bits; }
};

// A leaner reader for loops that take SHORT fields (<= 32 bits) off a long stream, one lane per
// stream: a 128-bit window (two 64-bit registers) over 8-byte loads, one load ahead. No per-word
// queue and no nested conditions - 64 lanes parsing 64 different streams execute every branch any
// of them takes. The caller stops using it `slack` bits before the end (see far_from_end) and hands
// over to BitReaderDev::seek for the tail, so it never has to think about the end of the payload.
struct WindowReaderDev {
    const uint64_t *words; // 8-byte aligned base
    uint32_t n_words;
    uint32_t next_word; // next one to load into `ahead`
    uint64_t high, low; // bits [position, position + available) of the stream, MSB first
    uint64_t ahead;     // the word after them, not byte swapped yet
    int32_t available;  // valid bits in (high, low): 65..128 whenever the caller looks
    uint64_t position;  // bits of the payload consumed
    uint64_t total_bits;

    // As stored (little endian): the byte swap happens where the word is used, a few codes later, so
    // that nothing has to wait for the load right behind it.
    // (always a load, see LeanReaderDev::load: behind the end the last word repeats, and the caller does
    // not look at those bits)
    __device__ __forceinline__ uint64_t load(uint32_t index) const {
        return load_global(words + min(index, n_words > 0 ? n_words - 1 : 0u));
    }
    // Any start_bit: whole 8-byte words in front of it are skipped, not read.
    __device__ __forceinline__ void open(const uint8_t *bytes, uint64_t nbytes, uint32_t start_bit) {
        const uintptr_t address = reinterpret_cast<uintptr_t>(bytes);
        const uint32_t misalign = (uint32_t)(address & 7u);
        const uint64_t first_bit = 8ull * misalign + start_bit; // counted from the aligned base
        const uint64_t skipped = first_bit >> 6;
        const uint64_t all_words = (nbytes + misalign + 7u) >> 3;
        words = reinterpret_cast<const uint64_t *>(address - misalign) + skipped;
        n_words = (uint32_t)(all_words > skipped ? all_words - skipped : 0u);
        high = __builtin_bswap64(load(0));
        low = __builtin_bswap64(load(1));
        ahead = load(2);
        next_word = 3;
        available = 128;
        total_bits = nbytes * 8u;
        uint32_t drop = (uint32_t)(first_bit & 63u);
        position = (uint64_t)start_bit - drop; // consume() adds it back
        while (drop > 0) {
            const uint32_t step = drop > 32u ? 32u : drop;
            consume(step);
            drop -= step;
        }
    }
    __device__ __forceinline__ bool far_from_end(uint32_t slack) const { return position + slack <= total_bits; }
    // The next 32 bits.
    __device__ __forceinline__ uint32_t top() const { return (uint32_t)(high >> 32); }
    // count in [1, 32]
    __device__ __forceinline__ void consume(uint32_t count) {
        high = (high << count) | (low >> (64u - count));
        low <<= count;
        available -= (int32_t)count;
        position += count;
        if (available <= 64) { // `low` is empty: the word loaded ahead becomes the low half
            const uint32_t fill = (uint32_t)available; // 33..64 valid bits in `high`
            const uint64_t next = __builtin_bswap64(ahead);
            high |= fill < 64u ? next >> fill : 0ull;
            low = fill < 64u ? next << (64u - fill) : next;
            ahead = load(next_word++);
            available += 64;
        }
    }
};

// The leanest of the readers, for the delta-of-delta timestamp codes (1 to 16 bits almost always): a
// 64-bit buffer refilled 32 bits at a time, without a branch except around the fetch of the word.
// Words come out of 16-byte chunks, one chunk loaded AHEAD of the one in use: a lane that walks a
// stream of its own needs a word every three codes or so, in lockstep with 63 others that is a load in
// nearly every step of the wave, and a load that is needed in the next step costs the whole wave a
// trip to memory - a chunk ahead is some ten steps of lead. (A chunk that holds at least one byte of
// the payload never crosses a page, so the slack bytes of the first and last chunk are safe to touch.)
// The caller stops `slack` bits before the end (far_from_end) and hands over to BitReaderDev::seek.
struct LeanReaderDev {
    const uint4 *chunks; // 16-byte aligned base (of the first chunk that is read)
    uint32_t last_chunk; // index of the last chunk that holds payload
    uint32_t next_chunk; // next one to load into `ahead`
    uint32_t left;       // words of `current` not handed out yet (1..4)
    uint4 current, ahead; // current.x is the next word
    uint64_t buffer;     // bits [position, position + available) of the stream, MSB first
    int32_t available;
    uint64_t position;   // bits of the payload consumed
    uint64_t total_bits;

    // Always a load of a chunk of the stream, never a choice between a load and a constant: a value
    // that is "the loaded chunk or zeros" has to be put together right behind the load, and the wave
    // would wait for the memory there instead of a chunk later. Behind the end the last chunk repeats;
    // the caller does not look at those bits (far_from_end).
    __device__ __forceinline__ uint4 load(uint32_t index) const { return load_global(chunks + min(index, last_chunk)); }
    __device__ __forceinline__ uint32_t take_word() {
        const uint32_t w = current.x;
        current.x = current.y;
        current.y = current.z;
        current.z = current.w;
        if (--left == 0) {
            current = ahead;
            ahead = load(next_chunk++);
            left = 4;
        }
        return __builtin_bswap32(w);
    }
    // nbytes > 0
    __device__ __forceinline__ void open(const uint8_t *bytes, uint64_t nbytes, uint64_t start_bit) {
        const uintptr_t address = reinterpret_cast<uintptr_t>(bytes);
        const uint32_t misalign = (uint32_t)(address & 15u);
        const uint64_t first_bit = 8ull * misalign + start_bit; // counted from the aligned base
        const uint64_t skipped = first_bit >> 7;                // whole chunks in front of it
        const uint64_t all_chunks = (nbytes + misalign + 15u) >> 4;
        chunks = reinterpret_cast<const uint4 *>(address - misalign) + skipped;
        last_chunk = (uint32_t)(all_chunks > skipped ? all_chunks - skipped - 1 : 0u);
        current = load(0);
        ahead = load(1);
        next_chunk = 2;
        left = 4;
        for (uint32_t k = (uint32_t)((first_bit >> 5) & 3u); k > 0; k--) (void)take_word();
        const uint32_t drop = (uint32_t)(first_bit & 31u);
        const uint64_t high = take_word();
        buffer = ((high << 32) | take_word()) << drop;
        available = 64 - (int32_t)drop;
        position = start_bit;
        total_bits = nbytes * 8u;
    }
    __device__ __forceinline__ bool far_from_end(uint32_t slack) const { return position + slack <= total_bits; }
    // At least 33 valid bits afterwards.
    __device__ __forceinline__ void refill() {
        if (available <= 32) {
            buffer |= (uint64_t)take_word() << (32 - available);
            available += 32;
        }
    }
    __device__ __forceinline__ uint32_t top() const { return (uint32_t)(buffer >> 32); }
    // count in [0, 32], after refill()
    __device__ __forceinline__ void consume(uint32_t count) {
        buffer <<= count;
        available -= (int32_t)count;
        position += count;
    }
};

// Rust f32::min / f32::max as the oracle defines them: minNum / maxNum, first operand kept on ties.
__device__ __forceinline__ float min_num(float a, float b) {
    if (a != a) return b;
    return (b < a) ? b : a;
}
__device__ __forceinline__ float max_num(float a, float b) {
    if (a != a) return b;
    return (b > a) ? b : a;
}
__device__ __forceinline__ double min_num(double a, double b) {
    if (a != a) return b;
    return (b < a) ? b : a;
}
__device__ __forceinline__ double max_num(double a, double b) {
    if (a != a) return b;
    return (b > a) ? b : a;
}

__device__ __forceinline__ bool equal_or_nan(double a, double b) {
    return a == b || (a != a && b != b);
}

struct LineDev {
    double slope;
    double intercept;
};

// models/swing.rs:323-340
__device__ __forceinline__ LineDev line_through(int64_t t0, double v0, int64_t t1, double v1) {
    if (equal_or_nan(v0, v1)) return {0.0, v0};
    double slope = (v1 - v0) / (double)(t1 - t0);
    double intercept = v0 - slope * (double)t0;
    return {slope, intercept};
}

// Wave-wide and block-wide exclusive scans of 64-bit values (the "wavefront prefix-sum").
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int delta) {
    uint32_t lo = __shfl_up((uint32_t)v, delta, MDB_WAVE);
    uint32_t hi = __shfl_up((uint32_t)(v >> 32), delta, MDB_WAVE);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_inclusive_scan_u64(uint64_t v) {
    int lane = threadIdx.x & (MDB_WAVE - 1);
#pragma unroll
    for (int delta = 1; delta < MDB_WAVE; delta <<= 1) {
        uint64_t up = shfl_up_u64(v, delta);
        if (lane >= delta) v += up;
    }
    return v;
}

// Exclusive scan over the block (blockDim.x multiple of 64, <= 1024). `total` gets the block sum.
// `lds` needs 17 uint64_t.
__device__ __forceinline__ uint64_t block_exclusive_scan_u64(uint64_t v, uint64_t *lds,
                                                             uint64_t *total) {
    int lane = threadIdx.x & (MDB_WAVE - 1);
    int wave = threadIdx.x / MDB_WAVE;
    int n_waves = (blockDim.x + MDB_WAVE - 1) / MDB_WAVE;
    uint64_t inclusive = wave_inclusive_scan_u64(v);
    if (lane == MDB_WAVE - 1) lds[wave] = inclusive;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t running = 0;
        for (int w = 0; w < n_waves; w++) {
            uint64_t t = lds[w];
            lds[w] = running;
            running += t;
        }
        lds[16] = running;
    }
    __syncthreads();
    uint64_t result = lds[wave] + inclusive - v;
    *total = lds[16];
    __syncthreads();
    return result;
}

} // namespace mdb
