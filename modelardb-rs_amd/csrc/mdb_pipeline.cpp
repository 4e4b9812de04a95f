// mdb_pipeline.cpp - the host side of libmdb_hip.so that keeps the GPU and PCIe busy for a POLLED operator:
// mdb_grid_submit / mdb_grid_wait (two batches in flight on two contexts, several input RecordBatches per
// launch), the replication of tag views per reconstructed row (grid_exec.rs:339-346) by a pool of host threads
// with streaming stores, and the recycled host blocks the replicated views live in. No kernels and no HIP call
// here (plain C++: the CPU sanitizers run this file against a stand-in for the kernels, tests/stub).
//
// Why this is behind the C ABI and not in the host language: the reference's GridStream is Rust, and a Rust
// shim that called mdb_grid_batch_owned synchronously per 8 192-row batch and appended one StringView per row
// on the polling thread measured 28 GB/s and 0.27 x 10^9 values/s with a tag column; the same three
// techniques in the library give every host language 50 GB/s and 3 x 10^9 (host path).
#include <emmintrin.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <thread>

#include <map>
#include <set>
#include <mutex>
#include <string>

#include "mdb_host_side.hpp"

extern char **environ;

namespace mdb {

// ---- the library's switches -----------------------------------------------------------------------------------
//
// One table per process: the MDB_* variables of the environment as they were when the table was first asked (or last
// reloaded), and what mdb_set_option() has put on top. A text, once handed out, is never changed or freed: every
// distinct value is kept once in `texts` for the life of the process (a switch has a handful of values) and an entry
// only points at one, so a call that is still reading the text a look-up gave it is not disturbed by mdb_set_option /
// mdb_reload_options on another thread (pipeline workers and the host walk ask while they run).
namespace {

struct Options {
    struct Entry {
        const std::string *text = nullptr; // into `texts`
        bool set = false;
    };
    std::mutex mutex;
    std::map<std::string, Entry> entries;
    std::set<std::string> texts; // (node based: the strings never move)
    bool loaded = false;
    const std::string *keep(const char *value) { return &*texts.insert(value).first; }
    void load_locked() {
        for (auto &entry : entries) entry.second.set = false;
        for (char **variable = environ; variable && *variable; variable++) {
            if (std::strncmp(*variable, "MDB_", 4) != 0) continue;
            const char *equals = std::strchr(*variable, '=');
            if (!equals) continue;
            Entry &entry = entries[std::string(*variable, (size_t)(equals - *variable))];
            entry.text = keep(equals + 1);
            entry.set = true;
        }
        loaded = true;
    }
};

Options &options() {
    static Options *table = new Options(); // (never destroyed: worker threads may ask while the process exits)
    return *table;
}

} // namespace

const char *option_text(const char *name) {
    Options &table = options();
    std::lock_guard<std::mutex> lock(table.mutex);
    if (!table.loaded) table.load_locked();
    const auto found = table.entries.find(name);
    return found != table.entries.end() && found->second.set ? found->second.text->c_str() : nullptr;
}

// ---- host threads -------------------------------------------------------------------------------------------

namespace {

// A fixed pool (up to 16 threads, made on first use, never joined: the process may exit while they wait).
// run() may be called from several threads at once (two grid jobs finishing together): shares are queued,
// the caller works through its own call's shares too and returns when the last of them is done.
class HostPool {
  public:
    static HostPool &instance() {
        static HostPool *pool = new HostPool();
        return *pool;
    }
    unsigned width() const { return width_; }

    struct Call {
        void (*share)(unsigned, void *);
        void *arg;
        unsigned n_shares;
        unsigned next = 0;   // next share to hand out (under the pool's mutex)
        unsigned done = 0;   // shares finished
    };

    void run(unsigned n_shares, void (*share)(unsigned, void *), void *arg) {
        if (n_shares <= 1 || width_ <= 1) {
            for (unsigned k = 0; k < n_shares; k++) share(k, arg);
            return;
        }
        Call call{share, arg, n_shares};
        std::unique_lock<std::mutex> lock(mutex_);
        calls_.push_back(&call);
        wake_.notify_all();
        // The caller takes shares of its own call like any worker.
        while (call.next < call.n_shares) {
            const unsigned k = call.next++;
            if (call.next == call.n_shares) unlist(&call);
            lock.unlock();
            share(k, arg);
            lock.lock();
            call.done++;
        }
        finished_.wait(lock, [&] { return call.done == call.n_shares; });
    }

  private:
    HostPool() {
        const unsigned hardware = std::max(1u, std::thread::hardware_concurrency());
        width_ = std::min(16u, hardware);
        for (unsigned w = 1; w < width_; w++) std::thread([this] { loop(); }).detach();
    }
    void unlist(Call *call) {
        for (auto it = calls_.begin(); it != calls_.end(); ++it)
            if (*it == call) {
                calls_.erase(it);
                return;
            }
    }
    void loop() {
        std::unique_lock<std::mutex> lock(mutex_);
        for (;;) {
            wake_.wait(lock, [&] { return !calls_.empty(); });
            Call *call = calls_.front();
            const unsigned k = call->next++;
            if (call->next == call->n_shares) unlist(call);
            lock.unlock();
            call->share(k, call->arg);
            lock.lock();
            if (++call->done == call->n_shares) finished_.notify_all();
        }
    }

    std::mutex mutex_;
    std::condition_variable wake_, finished_;
    std::deque<Call *> calls_;
    unsigned width_ = 1;
};

// Recycled host blocks, process wide, at most 16 kept (never freed at exit).
struct HostBlocks {
    std::mutex mutex;
    std::vector<std::pair<void *, uint64_t>> blocks;
    static HostBlocks &instance() {
        static HostBlocks *pool = new HostBlocks();
        return *pool;
    }
};

} // namespace

unsigned host_parallel_width() { return HostPool::instance().width(); }

void host_parallel(unsigned n_shares, void (*share)(unsigned, void *), void *arg) {
    HostPool::instance().run(n_shares, share, arg);
}

// A copy that leaves the caches alone (the staging block is read next by the copy engine, not by a core; and the lines
// of a plain store would be read from memory first): 16-byte streaming stores, what every x86-64 has.
void host_copy_streaming(void *to_bytes, const void *from_bytes, size_t n_bytes) {
#if defined(__SSE2__)
    unsigned char *to = static_cast<unsigned char *>(to_bytes);
    const unsigned char *from = static_cast<const unsigned char *>(from_bytes);
    size_t head = (16 - (reinterpret_cast<uintptr_t>(to) & 15)) & 15;
    if (head > n_bytes) head = n_bytes;
    std::memcpy(to, from, head);
    size_t i = head;
    for (; i + 64 <= n_bytes; i += 64) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(from + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(from + i + 16));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(from + i + 32));
        const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(from + i + 48));
        _mm_stream_si128(reinterpret_cast<__m128i *>(to + i), a);
        _mm_stream_si128(reinterpret_cast<__m128i *>(to + i + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i *>(to + i + 32), c);
        _mm_stream_si128(reinterpret_cast<__m128i *>(to + i + 48), d);
    }
    std::memcpy(to + i, from + i, n_bytes - i);
    _mm_sfence();
#else // (a host without SSE2: the plain copy - the streaming stores are a speed-up, not a requirement)
    std::memcpy(to_bytes, from_bytes, n_bytes);
#endif
}

int host_block_take(uint64_t bytes, void **out, uint64_t *capacity) {
    if (bytes == 0) bytes = 64;
    HostBlocks &pool = HostBlocks::instance();
    {
        std::lock_guard<std::mutex> lock(pool.mutex);
        int best = -1;
        for (size_t i = 0; i < pool.blocks.size(); i++) {
            const uint64_t cap = pool.blocks[i].second;
            if (cap >= bytes && cap <= 4 * bytes + (16 << 20) && (best < 0 || cap < pool.blocks[(size_t)best].second))
                best = (int)i;
        }
        if (best >= 0) {
            *out = pool.blocks[(size_t)best].first;
            *capacity = pool.blocks[(size_t)best].second;
            pool.blocks.erase(pool.blocks.begin() + best);
            return 0;
        }
    }
    const uint64_t grown = (bytes + bytes / 4 + 0xffffu) & ~(uint64_t)0xffffu;
    void *block = nullptr;
    if (posix_memalign(&block, 64, grown) != 0) return fail("Out of host memory for " + std::to_string(grown) + " bytes.");
    *out = block;
    *capacity = grown;
    return 0;
}

void host_block_give(void *block, uint64_t capacity) {
    if (!block) return;
    HostBlocks &pool = HostBlocks::instance();
    std::lock_guard<std::mutex> lock(pool.mutex);
    if (pool.blocks.size() >= 16) {
        size_t smallest = 0;
        for (size_t i = 1; i < pool.blocks.size(); i++)
            if (pool.blocks[i].second < pool.blocks[smallest].second) smallest = i;
        if (pool.blocks[smallest].second < capacity) {
            std::free(pool.blocks[smallest].first);
            pool.blocks[smallest] = {block, capacity};
        } else {
            std::free(block);
        }
        return;
    }
    pool.blocks.push_back({block, capacity});
}

// ---- tag views, once per reconstructed row -----------------------------------------------------------------

namespace {

// n_rows segment rows of one source array, their output starting at `to`. Written past the cache:
// nobody reads the views before the consumer does, and reading the lines first to own them would double the
// memory traffic of what is a plain fill.
void replicate_rows(const mdb_view16 *views, const uint32_t *rows_per_segment, uint64_t n_rows, int32_t buffer_shift,
                    mdb_view16 *to) {
    for (uint64_t row = 0; row < n_rows; row++) {
        mdb_view16 tag = views[row];
        if (tag.length > 12) tag.u.ref.buffer_index += buffer_shift;
        const uint32_t n = rows_per_segment[row];
        if ((reinterpret_cast<uintptr_t>(to) & 15u) == 0) {
            __m128i bits;
            std::memcpy(&bits, &tag, 16);
            for (uint32_t k = 0; k < n; k++) _mm_stream_si128(reinterpret_cast<__m128i *>(to + k), bits);
        } else {
            for (uint32_t k = 0; k < n; k++) to[k] = tag;
        }
        to += n;
    }
}

// One fill: several source arrays laid end to end (the inputs of a submit), one output per tag column.
struct Source {
    uint64_t first_row; // of this source among all segment rows
    uint64_t n;
};
struct Fill {
    const Source *sources;
    uint32_t n_sources;
    uint32_t n_columns;
    const mdb_view16 *const *views; // [source * n_columns + column]
    const int32_t *shifts;          // [source * n_columns + column]
    const uint32_t *rows_per_segment;
    mdb_view16 *const *outs;        // [column], the view of output row 0
    struct Share {
        uint64_t first_row, last_row, at;
    };
    std::vector<Share> shares;
};

void fill_share(unsigned index, void *arg) {
    const Fill &fill = *static_cast<const Fill *>(arg);
    const Fill::Share &share = fill.shares[index];
    for (uint32_t column = 0; column < fill.n_columns; column++) {
        uint64_t at = share.at;
        for (uint32_t k = 0; k < fill.n_sources; k++) {
            const Source &source = fill.sources[k];
            const uint64_t first = std::max(share.first_row, source.first_row);
            const uint64_t last = std::min(share.last_row, source.first_row + source.n);
            if (first >= last) continue;
            const mdb_view16 *views = fill.views[(size_t)k * fill.n_columns + column];
            replicate_rows(views + (first - source.first_row), fill.rows_per_segment + first, last - first,
                           fill.shifts ? fill.shifts[(size_t)k * fill.n_columns + column] : 0, fill.outs[column] + at);
            for (uint64_t row = first; row < last; row++) at += fill.rows_per_segment[row];
        }
    }
    _mm_sfence();
}

const uint64_t PARALLEL_MIN_VIEWS = 1u << 16;

void run_fill(Fill &fill, uint64_t n_segments, uint64_t n_rows_out) {
    unsigned n_shares = 1;
    if (n_rows_out * fill.n_columns >= PARALLEL_MIN_VIEWS)
        n_shares = (unsigned)std::min<uint64_t>(host_parallel_width(), std::max<uint64_t>(1, n_segments));
    // equal shares of the OUTPUT rows, cut at segment rows
    uint64_t row = 0, at = 0;
    for (unsigned w = 0; w < n_shares; w++) {
        const uint64_t target = n_rows_out * (w + 1) / n_shares;
        Fill::Share share{row, row, at};
        while (row < n_segments && (at < target || w + 1 == n_shares)) at += fill.rows_per_segment[row++];
        share.last_row = row;
        fill.shares.push_back(share);
    }
    host_parallel((unsigned)fill.shares.size(), fill_share, &fill);
}

} // namespace

// ---- two batches in flight ----------------------------------------------------------------------------------

} // namespace mdb

struct mdb_grid_ticket {
    std::mutex mutex;
    std::condition_variable finished;
    bool done = false;
    int rc = 0;
    std::string error;
    mdb_grid_result *result = nullptr;

    mdb_grid_request request;
    std::vector<mdb_segments> segments;          // per input, their tables pointing into the two vectors below
    std::vector<const uint8_t *> buffer_pointers;
    std::vector<int64_t> buffer_sizes;
    std::vector<const mdb_view16 *> tag_views;   // [input * n_tag_columns + column]
    std::vector<int32_t> tag_shifts;
};

namespace mdb {

// The context the first submit came in on, its clones (a stream and buffers each), a worker per context. A job goes
// to the idle worker of the lowest number (a lone job runs on the caller's own context) or, when all are busy, into
// ONE queue that whichever worker is free next takes from; results are waited for per ticket, in any order.
constexpr int PIPELINE_MAX_CONTEXTS = 4;
constexpr int PIPELINE_DEFAULT_CONTEXTS = 2;

struct GridPipeline {
    mdb_ctx *contexts[PIPELINE_MAX_CONTEXTS] = {};
    std::thread workers[PIPELINE_MAX_CONTEXTS];
    std::mutex mutex;
    std::condition_variable wake;
    std::deque<mdb_grid_ticket *> queue;
    mdb_grid_ticket *handed[PIPELINE_MAX_CONTEXTS] = {};
    bool idle[PIPELINE_MAX_CONTEXTS] = {};
    bool stop = false;
};

namespace {

void run_ticket(mdb_ctx *ctx, mdb_grid_ticket *ticket) {
    const mdb_grid_request &request = ticket->request;
    std::vector<const mdb_segments *> ins;
    for (const mdb_segments &s : ticket->segments) ins.push_back(&s);
    mdb_grid_result *result = nullptr;
    int rc = grid_batch_owned_list(ctx, ins.data(), (uint32_t)ins.size(),
                                   TimeRangeArg{request.t_lo, request.t_hi, (request.flags & MDB_GRID_HAS_RANGE) ? 1 : 0},
                                   (request.flags & MDB_GRID_VALUES_ONLY) != 0, request.reserve_front, &result);
    if (!rc && request.n_tag_columns > 0) {
        OwnedGridResult *owned = static_cast<OwnedGridResult *>(result->priv_);
        const uint64_t front = result->reserved_front;
        std::vector<mdb_view16 *> outs;
        for (uint32_t t = 0; t < request.n_tag_columns && !rc; t++) {
            void *block = nullptr;
            uint64_t capacity = 0;
            rc = host_block_take((front + result->n) * sizeof(mdb_view16), &block, &capacity);
            if (rc) break;
            owned->tag_blocks.push_back({block, capacity});
            owned->tag_views.push_back(static_cast<mdb_view16 *>(block) + front);
            outs.push_back(static_cast<mdb_view16 *>(block) + front);
        }
        if (!rc && result->n > 0) {
            std::vector<Source> sources;
            uint64_t first_row = 0;
            for (const mdb_segments &s : ticket->segments) {
                sources.push_back({first_row, s.n});
                first_row += s.n;
            }
            Fill fill{sources.data(), (uint32_t)sources.size(), request.n_tag_columns, ticket->tag_views.data(),
                      ticket->tag_shifts.data(), result->rows_per_segment, outs.data(), {}};
            run_fill(fill, result->n_segments, result->n);
        }
        if (rc) {
            mdb_grid_result_free(result);
            result = nullptr;
        }
    }
    std::lock_guard<std::mutex> lock(ticket->mutex);
    ticket->rc = rc;
    if (rc) ticket->error = g_last_error;
    ticket->result = result;
    ticket->done = true;
    ticket->finished.notify_all();
}

void pipeline_worker(GridPipeline *pipeline, int which) {
    for (;;) {
        mdb_grid_ticket *ticket = nullptr;
        {
            std::unique_lock<std::mutex> lock(pipeline->mutex);
            pipeline->idle[which] = true;
            pipeline->wake.wait(lock, [&] { return pipeline->stop || pipeline->handed[which] || !pipeline->queue.empty(); });
            pipeline->idle[which] = false;
            if (pipeline->handed[which]) {
                ticket = pipeline->handed[which];
                pipeline->handed[which] = nullptr;
            } else if (!pipeline->queue.empty()) {
                ticket = pipeline->queue.front();
                pipeline->queue.pop_front();
            } else {
                return; // (stop, and nothing left to do)
            }
        }
        run_ticket(pipeline->contexts[which], ticket);
    }
}

} // namespace

int pipeline_clones(mdb_ctx *ctx, mdb_ctx **out, int capacity) {
    GridPipeline *pipeline = ctx_pipeline(ctx);
    int n = 0;
    if (pipeline)
        for (int w = 1; w < PIPELINE_MAX_CONTEXTS && n < capacity; w++)
            if (pipeline->contexts[w]) out[n++] = pipeline->contexts[w];
    return n;
}

void pipeline_close(mdb_ctx *ctx) {
    GridPipeline *pipeline = ctx_pipeline_detach(ctx);
    if (!pipeline) return;
    {
        std::lock_guard<std::mutex> lock(pipeline->mutex);
        pipeline->stop = true;
    }
    pipeline->wake.notify_all();
    for (std::thread &worker : pipeline->workers)
        if (worker.joinable()) worker.join(); // (they finish what is queued: tickets stay valid for their owners)
    for (int w = 1; w < PIPELINE_MAX_CONTEXTS; w++)
        if (pipeline->contexts[w]) (void)mdb_close(pipeline->contexts[w]);
    delete pipeline;
}

} // namespace mdb

using namespace mdb;

extern "C" {

int mdb_set_option(const char *name, const char *value) {
    if (!name || std::strncmp(name, "MDB_", 4) != 0) return fail("The name of a switch begins with MDB_.");
    Options &table = options();
    std::lock_guard<std::mutex> lock(table.mutex);
    if (!table.loaded) table.load_locked();
    Options::Entry &entry = table.entries[name];
    entry.set = value != nullptr;
    if (value) entry.text = table.keep(value);
    return 0;
}

const char *mdb_option(const char *name) { return name ? option_text(name) : nullptr; }

int mdb_reload_options(void) {
    Options &table = options();
    std::lock_guard<std::mutex> lock(table.mutex);
    table.load_locked();
    return 0;
}

int mdb_replicate_views(const mdb_view16 *views, const uint32_t *rows_per_segment, uint64_t n_segments,
                        int32_t buffer_shift, mdb_view16 *out, uint64_t out_cap) {
    if (n_segments > 0 && (!views || !rows_per_segment)) return fail("views and rows_per_segment must not be NULL.");
    uint64_t total = 0;
    for (uint64_t i = 0; i < n_segments; i++) total += rows_per_segment[i];
    if (total > out_cap)
        return fail("Output views too small: " + std::to_string(total) + " rows but capacity " + std::to_string(out_cap) + ".");
    if (total == 0) return 0;
    if (!out) return fail("out must not be NULL.");
    const Source source{0, n_segments};
    const mdb_view16 *const sources_views[1] = {views};
    mdb_view16 *const outs[1] = {out};
    Fill fill{&source, 1, 1, sources_views, &buffer_shift, rows_per_segment, outs, {}};
    run_fill(fill, n_segments, total);
    return 0;
}

int mdb_grid_submit(mdb_ctx *ctx, const mdb_grid_input *inputs, uint32_t n_inputs, const mdb_grid_request *request,
                    mdb_grid_ticket **out) {
    if (!ctx || !inputs || !request || !out || n_inputs == 0)
        return fail("ctx, inputs, request and ticket must not be NULL, and there must be an input.");
    *out = nullptr;
    std::unique_ptr<mdb_grid_ticket> ticket(new mdb_grid_ticket());
    ticket->request = *request;
    const uint32_t n_tags = request->n_tag_columns;
    size_t n_buffers = 0;
    for (uint32_t k = 0; k < n_inputs; k++) {
        const mdb_binview_col *cols[3] = {&inputs[k].segments.timestamps, &inputs[k].segments.values,
                                          &inputs[k].segments.residuals};
        for (int c = 0; c < 3; c++) {
            if (cols[c]->n_buffers < 0) return fail("n_buffers must not be negative.");
            if (cols[c]->n_buffers > 0 && (!cols[c]->buffers || !cols[c]->buffer_sizes))
                return fail("buffers and buffer_sizes must be given when n_buffers > 0.");
            n_buffers += (size_t)cols[c]->n_buffers;
        }
        if (n_tags > 0 && inputs[k].segments.n > 0 && !inputs[k].tag_views) return fail("tag_views must not be NULL.");
    }
    ticket->buffer_pointers.resize(n_buffers + 1);
    ticket->buffer_sizes.resize(n_buffers + 1);
    size_t at = 0;
    for (uint32_t k = 0; k < n_inputs; k++) {
        mdb_segments copy = inputs[k].segments;
        mdb_binview_col *cols[3] = {&copy.timestamps, &copy.values, &copy.residuals};
        for (int c = 0; c < 3; c++) {
            const int32_t n = cols[c]->n_buffers;
            for (int32_t b = 0; b < n; b++) {
                ticket->buffer_pointers[at + (size_t)b] = cols[c]->buffers[b];
                ticket->buffer_sizes[at + (size_t)b] = cols[c]->buffer_sizes[b];
            }
            cols[c]->buffers = ticket->buffer_pointers.data() + at;
            cols[c]->buffer_sizes = ticket->buffer_sizes.data() + at;
            at += (size_t)n;
        }
        ticket->segments.push_back(copy);
        for (uint32_t t = 0; t < n_tags; t++) {
            const mdb_view16 *views = inputs[k].tag_views ? inputs[k].tag_views[t] : nullptr;
            if (inputs[k].segments.n > 0 && !views) return fail("tag_views[column] must not be NULL.");
            ticket->tag_views.push_back(views);
            ticket->tag_shifts.push_back(inputs[k].tag_buffer_shift ? inputs[k].tag_buffer_shift[t] : 0);
        }
    }
    GridPipeline *pipeline = ctx_pipeline(ctx);
    if (!pipeline) { // the first submit on this context: its clones and a worker for each
        std::unique_ptr<GridPipeline> fresh(new GridPipeline());
        fresh->contexts[0] = ctx;
        // MDB_GRID_PIPELINE_CONTEXTS=n, 1..4: 1 = every job on the context itself (A/B: what the clones buy)
        const char *setting = option_text("MDB_GRID_PIPELINE_CONTEXTS");
        const int n_contexts = setting ? std::min(std::max(std::atoi(setting), 1), PIPELINE_MAX_CONTEXTS) : PIPELINE_DEFAULT_CONTEXTS;
        auto close_clones = [&fresh]() {
            for (int w = 1; w < PIPELINE_MAX_CONTEXTS; w++)
                if (fresh->contexts[w]) (void)mdb_close(fresh->contexts[w]);
        };
        for (int w = 1; w < n_contexts; w++)
            if (mdb_clone(ctx, &fresh->contexts[w])) {
                close_clones();
                return 1;
            }
        for (int w = 0; w < n_contexts; w++) fresh->workers[w] = std::thread(pipeline_worker, fresh.get(), w);
        pipeline = ctx_pipeline_install(ctx, fresh.get());
        if (pipeline == fresh.get()) {
            (void)fresh.release();
        } else { // another thread's first submit came first: this one's workers and context go again
            {
                std::lock_guard<std::mutex> lock(fresh->mutex);
                fresh->stop = true;
            }
            fresh->wake.notify_all();
            for (std::thread &worker : fresh->workers)
                if (worker.joinable()) worker.join();
            close_clones();
        }
    }
    {
        std::lock_guard<std::mutex> lock(pipeline->mutex);
        int taker = -1;
        for (int w = 0; w < PIPELINE_MAX_CONTEXTS && taker < 0; w++)
            if (pipeline->contexts[w] && pipeline->idle[w] && !pipeline->handed[w] && pipeline->queue.empty()) taker = w;
        if (taker >= 0) pipeline->handed[taker] = ticket.get();
        else pipeline->queue.push_back(ticket.get());
    }
    pipeline->wake.notify_all();
    *out = ticket.release();
    return 0;
}

int mdb_grid_wait(mdb_grid_ticket *ticket, mdb_grid_result **out) {
    if (!ticket || !out) return fail("ticket and out must not be NULL.");
    *out = nullptr;
    {
        std::unique_lock<std::mutex> lock(ticket->mutex);
        ticket->finished.wait(lock, [&] { return ticket->done; });
    }
    const int rc = ticket->rc;
    if (rc) g_last_error = ticket->error;
    else *out = ticket->result;
    delete ticket;
    return rc ? 1 : 0;
}

void mdb_grid_cancel(mdb_grid_ticket *ticket) {
    if (!ticket) return;
    mdb_grid_result *result = nullptr;
    if (mdb_grid_wait(ticket, &result) == 0) mdb_grid_result_free(result);
}

mdb_view16 *mdb_grid_result_tag_views(const mdb_grid_result *result, uint32_t column) {
    if (!result || !result->priv_) return nullptr;
    const OwnedGridResult *owned = static_cast<const OwnedGridResult *>(result->priv_);
    return column < owned->tag_views.size() ? owned->tag_views[column] : nullptr;
}

} // extern "C"
