// mdb_host_side.hpp - the part of libmdb_hip.so's internals that has no device code and no HIP type in it:
// errors, the pools of host blocks, the bookkeeping behind mdb_grid_result, and what mdb_pipeline.cpp (the
// threads behind mdb_grid_submit) needs from the rest of the library. mdb_pipeline.cpp includes only this, so
// it also builds with g++ under the CPU sanitizers against a stand-in for the kernels (tests/stub).
#pragma once

#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mdb.h"

namespace mdb {

// ---- errors (capi.rs:58-80 convention: 0 ok, 1 failure + thread-local message) -----------------

extern thread_local std::string g_last_error;

inline int fail(const std::string &message) {
    g_last_error = message;
    return 1;
}

// Recycled page-locked blocks behind mdb_grid_result (hipHostMalloc is slow). Shared between the
// context and the results it handed out, so a result may be freed after mdb_close().
struct PinnedPool {
    std::mutex mutex;
    bool closed = false;
    std::vector<std::pair<void *, uint64_t>> blocks;
    int take(uint64_t bytes, void **out, uint64_t *capacity);
    void give(void *block, uint64_t capacity);
    void trim();  // frees the recycled blocks
    void close();
};

struct GridPipeline; // the workers behind mdb_grid_submit (mdb_pipeline.cpp)
// The pipeline of a context, read and installed under the context's lock (mdb_ctx.hip; tests/stub).
// install: makes `fresh` the context's pipeline unless it has one already; returns the one it has afterwards.
GridPipeline *ctx_pipeline(mdb_ctx *ctx);
GridPipeline *ctx_pipeline_install(mdb_ctx *ctx, GridPipeline *fresh);
GridPipeline *ctx_pipeline_detach(mdb_ctx *ctx);
// The clones the context's pipeline runs jobs on beside the context itself (mdb_profile_*: their launches count as
// the context's); they live until the context is closed.
int pipeline_clones(mdb_ctx *ctx, mdb_ctx **out, int capacity);

// Owner bookkeeping behind mdb_grid_result::priv_.
struct OwnedGridResult {
    mdb_grid_result c;
    std::shared_ptr<PinnedPool> pool;
    void *block = nullptr;
    uint64_t capacity = 0;
    // replicated tag views of mdb_grid_submit, one block per tag column: (block, capacity in bytes)
    std::vector<std::pair<void *, uint64_t>> tag_blocks;
    std::vector<mdb_view16 *> tag_views; // per column: the view of the first reconstructed row
};
struct TimeRangeArg { // (TimeRange of mdb_segment_dev.hpp, for translation units that have no device code)
    int64_t lo;
    int64_t hi;
    int32_t enabled;
};
// mdb_grid.hip: one or several host batches through one launch into a page-locked block.
int grid_batch_owned_list(mdb_ctx *ctx, const mdb_segments *const *ins, uint32_t n_ins, TimeRangeArg range,
                          bool values_only, uint64_t reserve_front, mdb_grid_result **out);
// mdb_pipeline.hip: recycled ordinary host blocks (64-byte aligned) for the replicated tag views - a fresh 90 MB
// allocation per batch and tag column is 22 000 page faults - and the pool of host threads that fills them.
int host_block_take(uint64_t bytes, void **out, uint64_t *capacity);
void host_block_give(void *block, uint64_t capacity);
// The library's switches (the MDB_* names of INTEGRATION.md: A/B timings, the scheduling modes the tests force). The
// environment is read ONCE - by the first look-up of the process, or again by mdb_reload_options() - into a table that
// every call looks its switches up in: no getenv() inside a call (getenv races a host's setenv), and a host sets a
// switch without touching its environment: mdb_set_option(). nullptr: not set. The text stays valid until the switch is
// set again or the table reloaded (a test's business, not done while calls run). (mdb_pipeline.cpp)
const char *option_text(const char *name);

void host_parallel(unsigned n_shares, void (*share)(unsigned index, void *arg), void *arg);
unsigned host_parallel_width();
// memcpy with streaming stores: for blocks a core will not read again (staging for the copy engine). (mdb_pipeline.cpp)
void host_copy_streaming(void *to, const void *from, size_t n_bytes);
constexpr uint32_t MV_PIECE_VALUES = 64;
constexpr uint32_t MV_WINDOW_RESIDUAL = 1u << 16; // the piece belongs to the residual tail
constexpr uint32_t MV_WINDOW_RAW = 1u << 17;      // its first value is the stream's raw first value

struct MvCursor { // 32 bytes
    uint32_t bit_position; // of the piece's first code in its stream
    uint32_t xor_bits;     // XOR of all deltas of the chain before it
    uint32_t segment;
    uint32_t point_index;  // of the piece's first value among the segment's data points
    uint32_t n_values;     // 1..64
    uint32_t window;       // leading | trailing << 8 | MV_WINDOW_*
    uint32_t chain_seed;   // residual tail of a MacaqueV segment: the bits of its last model value (else 0)
    uint32_t pad;          // MV_CURSOR_LAST_OF_STREAM (cursors left by host threads only)
};
constexpr uint32_t MV_CURSOR_LAST_OF_STREAM = 1u;
static_assert(sizeof(MvCursor) == 32, "cursors are loaded as two uint4");

// The cursors the host threads of ONE call leave in the long MacaqueV streams of a host batch (mdb_grid.hip,
// mv_host_index): built before the batch is uploaded, used (uploaded, found by the kernels' launchers through the
// calling thread) around the call, done after it.
struct MvCallIndex {
    std::vector<unsigned long long> piece_base;
    std::vector<MvCursor> cursors;
};
// mdb_mv_host_index.cpp: piece_base (rows + 1) and cursors of the long MacaqueV streams of a list of host batches.
struct MvHostRange { // (optional) only the segments with a point in [lo, hi], by their start and end time
    int64_t lo, hi;
};
void mv_host_index(const mdb_segments *const *ins, uint32_t n_ins, std::vector<unsigned long long> *piece_base,
                   std::vector<MvCursor> *cursors, const MvHostRange *range = nullptr);
void mv_call_index_build(const mdb_segments *in, MvCallIndex *out, const MvHostRange *range = nullptr);
// Is there a row the walk could be for: a MacaqueV segment whose values payload can hold a long stream? Looks at the
// type column and the lengths in the views only (a batch without either: no): what a caller asks before it starts a
// thread for the walk.
bool mv_host_index_worthwhile(const mdb_segments *const *ins, uint32_t n_ins);
int mv_call_index_use(mdb_ctx *ctx, const mdb_segments &uploaded, const MvCallIndex &index);
void mv_call_index_done();

void pipeline_close(mdb_ctx *ctx);
int profile_collect(mdb_ctx *ctx);

} // namespace mdb
