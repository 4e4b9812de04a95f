// mdb_stub.cpp - TEST INFRASTRUCTURE, never part of the product: a stand-in for the KERNELS behind the entry
// points of include/mdb.h that libmdb_host calls (the threads of mdb_grid_submit / mdb_grid_wait and the tag
// replication are the product's own mdb_pipeline.cpp, built into the same object), so the host operators (mdb_host.cpp: GridStream with its
// worker threads and aliasing of library-owned blocks, SortedJoinStream, the accumulators, the
// uncompressed data manager) can be built with -fsanitize=address,undefined and -fsanitize=thread and
// driven on a machine without a GPU (GPU AddressSanitizer is not available on the MI355X pool).
//
// It computes nothing. Every call is answered from tests/golden/host_stub_fixtures.bin: the inputs
// of the call are hashed (FNV-1a over the resolved segment rows / the raw series and the scalar
// arguments) and the record with that key is replayed into freshly malloc'ed memory of exactly the
// size the header promises (so the sanitizers see every byte the host library touches beyond it).
// A call without a record is an error return that names the key: the test fails, nothing falls back.
//
// The fixture file is written by tests/golden/make_host_stub_fixtures.py, which runs the same tests
// against this file compiled with -DMDB_STUB_RECORD: only then it links the CPU oracle
// (oracle/libmdb_oracle.so), asks it for the answer of every call and appends (key, answer).
#include "../../include/mdb.h"
#include "../../modelardb-rs_amd/csrc/mdb_host_side.hpp"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#ifdef MDB_STUB_RECORD
#include "../../oracle/mdb_oracle.h"
#endif

struct mdb_ctx {
    uint32_t magic;
    std::mutex mutex;
    mdb::GridPipeline *pipeline = nullptr;
};

thread_local std::string mdb::g_last_error;

namespace {

constexpr uint32_t CTX_MAGIC = 0x6d646273; // "mdbs"
constexpr uint32_t RECORD_MAGIC = 0x5342444d;
enum Kind : uint32_t { KIND_GRID = 1, KIND_AGG = 2, KIND_FIT = 3 };

using mdb::fail;

struct Hasher {
    uint64_t state = 1469598103934665603ull;
    void bytes(const void *data, size_t n) {
        const uint8_t *p = static_cast<const uint8_t *>(data);
        for (size_t i = 0; i < n; i++) {
            state ^= p[i];
            state *= 1099511628211ull;
        }
    }
    template <typename T> void value(T v) { bytes(&v, sizeof v); }
};

// The payload a view names; false if it points outside the column's buffers.
bool resolve(const mdb_binview_col &col, uint64_t i, const uint8_t **data, uint32_t *length) {
    const mdb_view16 &view = col.views[i];
    if (view.length < 0) return false;
    *length = static_cast<uint32_t>(view.length);
    if (view.length <= 12) {
        *data = view.u.inlined;
        return true;
    }
    const int32_t b = view.u.ref.buffer_index, offset = view.u.ref.offset;
    if (b < 0 || b >= col.n_buffers || offset < 0 || !col.buffers[b] ||
        static_cast<int64_t>(offset) + view.length > col.buffer_sizes[b])
        return false;
    *data = col.buffers[b] + offset;
    return true;
}

bool hash_segments(const mdb_segments &s, Hasher &h) {
    h.value<uint64_t>(s.n);
    for (uint64_t i = 0; i < s.n; i++) {
        h.value(s.model_type_id[i]);
        h.value(s.start_time[i]);
        h.value(s.end_time[i]);
        h.value(s.min_value[i]);
        h.value(s.max_value[i]);
        for (const mdb_binview_col *col : {&s.timestamps, &s.values, &s.residuals}) {
            const uint8_t *data;
            uint32_t length;
            if (!resolve(*col, i, &data, &length)) return false;
            h.value(length);
            h.bytes(data, length);
        }
    }
    return true;
}

// ---- the fixture file: records of {magic, kind, key, payload bytes} ----------------------------------
struct Fixtures {
    std::unordered_map<uint64_t, std::vector<uint8_t>> records;
    std::string path;
    std::mutex mutex;
};

uint64_t record_key(uint32_t kind, uint64_t key) { return key ^ (0x9e3779b97f4a7c15ull * kind); }

Fixtures &fixtures() {
    static Fixtures instance;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *path = std::getenv("MDB_STUB_FIXTURES");
        instance.path = path ? path : "tests/golden/host_stub_fixtures.bin";
        FILE *f = std::fopen(instance.path.c_str(), "rb");
        if (!f) return;
        for (;;) {
            uint32_t head[2];
            uint64_t key, size;
            if (std::fread(head, 4, 2, f) != 2 || head[0] != RECORD_MAGIC) break;
            if (std::fread(&key, 8, 1, f) != 1 || std::fread(&size, 8, 1, f) != 1) break;
            std::vector<uint8_t> payload(size);
            if (size && std::fread(payload.data(), 1, size, f) != size) break;
            instance.records[record_key(head[1], key)] = std::move(payload);
        }
        std::fclose(f);
    });
    return instance;
}

const std::vector<uint8_t> *find_record(uint32_t kind, uint64_t key) {
    Fixtures &all = fixtures();
    std::lock_guard<std::mutex> lock(all.mutex);
    auto found = all.records.find(record_key(kind, key));
    return found == all.records.end() ? nullptr : &found->second;
}

int no_record(const char *what, uint64_t key) {
    char text[200];
    std::snprintf(text, sizeof text,
                  "mdb_stub: no canned result for this %s call (key %016llx) in %s: regenerate the fixtures with "
                  "tests/golden/make_host_stub_fixtures.py",
                  what, static_cast<unsigned long long>(key), fixtures().path.c_str());
    return fail(text);
}

#ifdef MDB_STUB_RECORD
void add_record(uint32_t kind, uint64_t key, const std::vector<uint8_t> &payload) {
    Fixtures &all = fixtures();
    std::lock_guard<std::mutex> lock(all.mutex);
    if (!all.records.emplace(record_key(kind, key), payload).second) return;
    FILE *f = std::fopen(all.path.c_str(), "ab");
    if (!f) return;
    const uint32_t head[2] = {RECORD_MAGIC, kind};
    const uint64_t size = payload.size();
    std::fwrite(head, 4, 2, f);
    std::fwrite(&key, 8, 1, f);
    std::fwrite(&size, 8, 1, f);
    if (size) std::fwrite(payload.data(), 1, size, f);
    std::fclose(f);
}
#endif

struct Writer {
    std::vector<uint8_t> out;
    void bytes(const void *data, size_t n) {
        const uint8_t *p = static_cast<const uint8_t *>(data);
        out.insert(out.end(), p, p + n);
    }
    template <typename T> void value(T v) { bytes(&v, sizeof v); }
};

struct Reader {
    const uint8_t *at, *end;
    bool ok = true;
    explicit Reader(const std::vector<uint8_t> &payload) : at(payload.data()), end(payload.data() + payload.size()) {}
    void bytes(void *into, size_t n) {
        if (static_cast<size_t>(end - at) < n) {
            ok = false;
            return;
        }
        if (n) std::memcpy(into, at, n);
        at += n;
    }
    template <typename T> T value() {
        T v{};
        bytes(&v, sizeof v);
        return v;
    }
};

// ---- segments made by the "compressor": one malloc per array, views rebuilt the way arrow's builder does -----
struct OwnedSegments {
    mdb_segments_owned owned{};
    std::vector<void *> blocks;
    const uint8_t *buffer_pointers[3] = {nullptr, nullptr, nullptr};
    int64_t buffer_sizes[3] = {0, 0, 0};
    ~OwnedSegments() {
        for (void *block : blocks) std::free(block);
    }
    template <typename T> T *array(uint64_t n) {
        T *p = static_cast<T *>(std::malloc(n ? n * sizeof(T) : 1));
        blocks.push_back(p);
        return p;
    }
};

mdb_segments_owned *segments_from_payload(const std::vector<uint8_t> &payload) {
    Reader in(payload);
    const uint64_t n = in.value<uint64_t>();
    auto *result = new OwnedSegments();
    auto *type = result->array<int8_t>(n);
    auto *start = result->array<int64_t>(n), *end = result->array<int64_t>(n);
    auto *mn = result->array<float>(n), *mx = result->array<float>(n), *error = result->array<float>(n);
    auto *chunk = result->array<uint32_t>(n);
    in.bytes(type, n);
    in.bytes(start, 8 * n);
    in.bytes(end, 8 * n);
    in.bytes(mn, 4 * n);
    in.bytes(mx, 4 * n);
    in.bytes(chunk, 4 * n);
    for (uint64_t i = 0; i < n; i++) error[i] = std::numeric_limits<float>::quiet_NaN();
    mdb_segments &s = result->owned.seg;
    s.n = n;
    s.model_type_id = type;
    s.start_time = start;
    s.end_time = end;
    s.min_value = mn;
    s.max_value = mx;
    mdb_binview_col *columns[3] = {&s.timestamps, &s.values, &s.residuals};
    for (int c = 0; c < 3; c++) {
        std::vector<uint32_t> lengths(n);
        in.bytes(lengths.data(), 4 * n);
        uint64_t out_of_line = 0;
        for (uint32_t length : lengths) out_of_line += length > 12 ? length : 0;
        auto *views = result->array<mdb_view16>(n);
        auto *data = result->array<uint8_t>(out_of_line);
        uint64_t offset = 0;
        for (uint64_t i = 0; i < n; i++) {
            std::memset(&views[i], 0, sizeof(mdb_view16));
            views[i].length = static_cast<int32_t>(lengths[i]);
            if (lengths[i] <= 12) {
                in.bytes(views[i].u.inlined, lengths[i]);
            } else {
                in.bytes(data + offset, lengths[i]);
                std::memcpy(views[i].u.ref.prefix, data + offset, 4);
                views[i].u.ref.buffer_index = 0;
                views[i].u.ref.offset = static_cast<int32_t>(offset);
                offset += lengths[i];
            }
        }
        result->buffer_pointers[c] = data;
        result->buffer_sizes[c] = static_cast<int64_t>(out_of_line);
        columns[c]->views = views;
        columns[c]->buffers = &result->buffer_pointers[c];
        columns[c]->buffer_sizes = &result->buffer_sizes[c];
        columns[c]->n_buffers = out_of_line ? 1 : 0;
    }
    result->owned.error = error;
    result->owned.chunk_index = chunk;
    result->owned.on_device = 0;
    result->owned.priv_ = result;
    if (!in.ok) {
        delete result;
        return nullptr;
    }
    return &result->owned;
}

#ifdef MDB_STUB_RECORD
std::vector<uint8_t> payload_from_segments(const mdb_segments_owned &made) {
    const mdb_segments &s = made.seg;
    Writer w;
    w.value<uint64_t>(s.n);
    w.bytes(s.model_type_id, s.n);
    w.bytes(s.start_time, 8 * s.n);
    w.bytes(s.end_time, 8 * s.n);
    w.bytes(s.min_value, 4 * s.n);
    w.bytes(s.max_value, 4 * s.n);
    if (made.chunk_index) {
        w.bytes(made.chunk_index, 4 * s.n);
    } else {
        for (uint64_t i = 0; i < s.n; i++) w.value<uint32_t>(0);
    }
    for (const mdb_binview_col *col : {&s.timestamps, &s.values, &s.residuals}) {
        for (uint64_t i = 0; i < s.n; i++) w.value<uint32_t>(static_cast<uint32_t>(col->views[i].length));
        for (uint64_t i = 0; i < s.n; i++) {
            const uint8_t *data;
            uint32_t length;
            resolve(*col, i, &data, &length);
            w.bytes(data, length);
        }
    }
    return w.out;
}
#endif

struct OwnedGrid : mdb::OwnedGridResult { // (the pipeline attaches the replicated tag views to the base)
    void *ts_block = nullptr, *value_block = nullptr, *rows_block = nullptr;
};

uint64_t fit_key(const int64_t *ts, const float *values, const uint64_t *offsets, uint64_t n_chunks,
                 mdb_error_bound eb) {
    Hasher h;
    h.value(eb.kind);
    h.value(eb.value);
    h.value(n_chunks);
    h.bytes(offsets, 8 * (n_chunks + 1));
    const uint64_t first = offsets[0], n = offsets[n_chunks] - first;
    h.bytes(ts + first, 8 * n);
    h.bytes(values + first, 4 * n);
    return h.state;
}

bool valid(const mdb_ctx *ctx) { return ctx && ctx->magic == CTX_MAGIC; }

// MDB_STUB_CALL_LOG=<file>: one line per data call, in the order the kernels' side sees them (the tests of the
// call sequence: a GridStream keeps one submit ahead and gathers small input batches into one launch).
void log_call(const char *what, uint64_t a = 0, uint64_t b = 0, uint64_t c = 0) {
    static const char *path = std::getenv("MDB_STUB_CALL_LOG");
    if (!path) return;
    static std::mutex mutex;
    std::lock_guard<std::mutex> lock(mutex);
    if (FILE *f = std::fopen(path, "a")) {
        std::fprintf(f, "%s %llu %llu %llu\n", what, (unsigned long long)a, (unsigned long long)b, (unsigned long long)c);
        std::fclose(f);
    }
}

// Several batches as one: primitive columns copied end to end, views copied with buffer_index moved onto a joint
// buffer table (what upload_segment_list_locked does on its way to the device in the product).
struct JoinedSegments {
    mdb_segments seg{};
    bool ok = true;
    std::vector<int8_t> type;
    std::vector<int64_t> start, end;
    std::vector<float> mn, mx;
    std::vector<mdb_view16> views[3];
    std::vector<const uint8_t *> pointers[3];
    std::vector<int64_t> sizes[3];
    JoinedSegments(const mdb_segments *const *ins, uint32_t n_ins) {
        for (uint32_t k = 0; k < n_ins; k++) {
            const mdb_segments &s = *ins[k];
            type.insert(type.end(), s.model_type_id, s.model_type_id + s.n);
            start.insert(start.end(), s.start_time, s.start_time + s.n);
            end.insert(end.end(), s.end_time, s.end_time + s.n);
            mn.insert(mn.end(), s.min_value, s.min_value + s.n);
            mx.insert(mx.end(), s.max_value, s.max_value + s.n);
            const mdb_binview_col *cols[3] = {&s.timestamps, &s.values, &s.residuals};
            for (int c = 0; c < 3; c++) {
                const int32_t first_buffer = static_cast<int32_t>(pointers[c].size());
                for (uint64_t i = 0; i < s.n; i++) {
                    mdb_view16 view = cols[c]->views[i];
                    if (view.length > 12) {
                        if (view.u.ref.buffer_index < 0 || view.u.ref.buffer_index >= cols[c]->n_buffers) ok = false;
                        view.u.ref.buffer_index += first_buffer;
                    }
                    views[c].push_back(view);
                }
                for (int32_t b = 0; b < cols[c]->n_buffers; b++) {
                    pointers[c].push_back(cols[c]->buffers[b]);
                    sizes[c].push_back(cols[c]->buffer_sizes[b]);
                }
            }
        }
        seg.n = type.size();
        seg.model_type_id = type.data();
        seg.start_time = start.data();
        seg.end_time = end.data();
        seg.min_value = mn.data();
        seg.max_value = mx.data();
        mdb_binview_col *out[3] = {&seg.timestamps, &seg.values, &seg.residuals};
        for (int c = 0; c < 3; c++) {
            out[c]->views = views[c].data();
            out[c]->buffers = pointers[c].data();
            out[c]->buffer_sizes = sizes[c].data();
            out[c]->n_buffers = static_cast<int32_t>(pointers[c].size());
        }
    }
};

} // namespace

mdb::GridPipeline *mdb::ctx_pipeline(mdb_ctx *ctx) {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    return ctx->pipeline;
}
mdb::GridPipeline *mdb::ctx_pipeline_install(mdb_ctx *ctx, GridPipeline *fresh) {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    if (!ctx->pipeline) ctx->pipeline = fresh;
    return ctx->pipeline;
}
mdb::GridPipeline *mdb::ctx_pipeline_detach(mdb_ctx *ctx) {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    GridPipeline *pipeline = ctx->pipeline;
    ctx->pipeline = nullptr;
    return pipeline;
}

extern "C" {

int mdb_init(int device, mdb_ctx **ctx) {
    if (!ctx || device < 0) return fail("mdb_init: NULL out pointer or negative device");
    *ctx = new mdb_ctx();
    (*ctx)->magic = CTX_MAGIC;
    return 0;
}

int mdb_clone(mdb_ctx *ctx, mdb_ctx **out) {
    if (!valid(ctx) || !out) return fail("mdb_clone: NULL or closed context");
    return mdb_init(0, out);
}

int mdb_close(mdb_ctx *ctx) {
    if (!ctx) return 0;
    if (!valid(ctx)) return fail("mdb_close: not a context (closed twice?)");
    mdb::pipeline_close(ctx);
    ctx->magic = 0;
    delete ctx;
    return 0;
}

const char *mdb_last_error(void) { return mdb::g_last_error.c_str(); }

void mdb_segments_free(mdb_segments_owned *segments) {
    if (segments) delete static_cast<OwnedSegments *>(segments->priv_);
}

void mdb_grid_result_free(mdb_grid_result *result) {
    if (!result) return;
    auto *owned = static_cast<OwnedGrid *>(static_cast<mdb::OwnedGridResult *>(result->priv_));
    for (auto &tags : owned->tag_blocks) mdb::host_block_give(tags.first, tags.second);
    std::free(owned->ts_block);
    std::free(owned->value_block);
    std::free(owned->rows_block);
    delete owned;
}

} // extern "C"

// The kernels' side of mdb_grid_batch_owned and of the jobs of mdb_grid_submit (mdb_grid.hip in the product):
// several input batches are answered as the one batch they make up.
int mdb::grid_batch_owned_list(mdb_ctx *ctx, const mdb_segments *const *ins, uint32_t n_ins, TimeRangeArg range,
                               bool values_only, uint64_t reserve_front, mdb_grid_result **out) {
    if (!valid(ctx) || !ins || !out || n_ins == 0) return fail("mdb_grid_batch_owned: NULL argument or closed context");
    const bool ranged = range.enabled != 0;
    const int64_t t_lo = range.lo, t_hi = range.hi;
    {
        uint64_t segments = 0;
        for (uint32_t k = 0; k < n_ins; k++) segments += ins[k]->n;
        log_call("grid", n_ins, segments, reserve_front);
        if (ranged) log_call("grid_range", static_cast<uint64_t>(t_lo), static_cast<uint64_t>(t_hi));
    }
    // (one batch holding the rows of all inputs: the views keep pointing into their own batch's buffers)
    JoinedSegments joined(ins, n_ins);
    const mdb_segments *in = &joined.seg;
    Hasher h;
    if (!joined.ok || !hash_segments(*in, h)) return fail("mdb_grid_batch_owned: a view points outside its data buffers");
    h.value<uint32_t>(ranged ? 1 : 0);
    if (ranged) {
        h.value(t_lo);
        h.value(t_hi);
    }
#ifdef MDB_STUB_RECORD
    {
        uint64_t n = 0;
        if (ora_grid_count(in, &n)) return fail(ora_last_error());
        std::vector<int64_t> ts(n);
        std::vector<float> values(n);
        std::vector<uint32_t> rows(in->n);
        mdb_grid_metrics metrics{};
        if (ora_grid_batch(in, ts.data(), values.data(), rows.data(), n, &n, &metrics)) return fail(ora_last_error());
        if (ranged) { // what the product does: only the points inside [t_lo, t_hi] are created
            uint64_t kept = 0, at = 0;
            for (int k = 0; k < MDB_MODEL_TYPE_COUNT; k++) metrics.rows_created_by_model_type[k] = 0;
            for (uint64_t i = 0; i < in->n; i++) {
                uint32_t visible = 0;
                for (uint32_t k = 0; k < rows[i]; k++, at++) {
                    if (ts[at] < t_lo || ts[at] > t_hi) continue;
                    ts[kept] = ts[at];
                    values[kept++] = values[at];
                    visible++;
                }
                rows[i] = visible;
                metrics.rows_created_by_model_type[in->model_type_id[i]] += visible;
            }
            n = kept;
            metrics.rows_created = kept;
        }
        Writer w;
        w.value<uint64_t>(n);
        w.value<uint64_t>(in->n);
        w.value(metrics);
        w.bytes(ts.data(), 8 * n);
        w.bytes(values.data(), 4 * n);
        w.bytes(rows.data(), 4 * in->n);
        add_record(KIND_GRID, h.state, w.out);
    }
#endif
    const std::vector<uint8_t> *payload = find_record(KIND_GRID, h.state);
    if (!payload) return no_record("mdb_grid_batch_owned", h.state);
    Reader reader(*payload);
    auto *owned = new OwnedGrid();
    mdb_grid_result &r = owned->c;
    std::memset(&r, 0, sizeof r);
    r.n = reader.value<uint64_t>();
    r.n_segments = reader.value<uint64_t>();
    r.metrics = reader.value<mdb_grid_metrics>();
    r.reserved_front = reserve_front;
    // Exactly reserve_front + n elements: one row further and the sanitizer reports it.
    owned->value_block = std::malloc((reserve_front + r.n) * 4 + 1);
    owned->rows_block = std::malloc(r.n_segments * 4 + 1);
    r.values = static_cast<float *>(owned->value_block) + reserve_front;
    r.rows_per_segment = static_cast<uint32_t *>(owned->rows_block);
    if (values_only) {
        reader.at += 8 * r.n;
    } else {
        owned->ts_block = std::malloc((reserve_front + r.n) * 8 + 1);
        r.timestamps = static_cast<int64_t *>(owned->ts_block) + reserve_front;
        reader.bytes(r.timestamps, 8 * r.n);
    }
    reader.bytes(r.values, 4 * r.n);
    reader.bytes(r.rows_per_segment, 4 * r.n_segments);
    r.priv_ = static_cast<mdb::OwnedGridResult *>(owned);
    if (!reader.ok || r.n_segments != in->n) {
        mdb_grid_result_free(&r);
        return fail("mdb_stub: damaged grid record");
    }
    *out = &r;
    return 0;
}

extern "C" {

int mdb_grid_batch_owned(mdb_ctx *ctx, const mdb_segments *in, uint32_t flags, int64_t t_lo, int64_t t_hi,
                         uint64_t reserve_front, mdb_grid_result **out) {
    if (!in) return fail("mdb_grid_batch_owned: NULL argument or closed context");
    return mdb::grid_batch_owned_list(ctx, &in, 1, mdb::TimeRangeArg{t_lo, t_hi, (flags & MDB_GRID_HAS_RANGE) ? 1 : 0},
                                      (flags & MDB_GRID_VALUES_ONLY) != 0, reserve_front, out);
}

int mdb_agg_batch(mdb_ctx *ctx, const mdb_segments *in, uint32_t which_mask, mdb_agg_state *inout) {
    if (!valid(ctx) || !in || !inout) return fail("mdb_agg_batch: NULL argument or closed context");
    Hasher h;
    if (!hash_segments(*in, h)) return fail("mdb_agg_batch: a view points outside its data buffers");
    h.value(which_mask);
#ifdef MDB_STUB_RECORD
    {
        mdb_agg_state fresh = {0.0, 0, FLT_MAX, -FLT_MAX};
        if (ora_agg_batch(in, which_mask, &fresh)) return fail(ora_last_error());
        Writer w;
        w.value(fresh);
        add_record(KIND_AGG, h.state, w.out);
    }
#endif
    const std::vector<uint8_t> *payload = find_record(KIND_AGG, h.state);
    if (!payload) return no_record("mdb_agg_batch", h.state);
    Reader reader(*payload);
    const mdb_agg_state batch = reader.value<mdb_agg_state>();
    if (!reader.ok) return fail("mdb_stub: damaged aggregate record");
    inout->sum += batch.sum;
    inout->count += batch.count;
    inout->min = std::fmin(inout->min, batch.min);
    inout->max = std::fmax(inout->max, batch.max);
    return 0;
}

// (the canned answers are per batch: the list is the sum of its batches' - what the library computes in one launch)
int mdb_agg_batch_list(mdb_ctx *ctx, const mdb_segments *const *inputs, uint32_t n_inputs, uint32_t which_mask,
                       mdb_agg_state *inout) {
    if (!valid(ctx) || !inputs || !inout) return fail("mdb_agg_batch_list: NULL argument or closed context");
    log_call("agg_list", n_inputs, which_mask);
    for (uint32_t k = 0; k < n_inputs; k++)
        if (mdb_agg_batch(ctx, inputs[k], which_mask, inout)) return 1;
    return 0;
}

// (under a time range, again per batch: COUNT / MIN / MAX / SUM of the data points inside [t_lo, t_hi])
int mdb_agg_batch_range_list(mdb_ctx *ctx, const mdb_segments *const *inputs, uint32_t n_inputs, int64_t t_lo, int64_t t_hi,
                             uint32_t which_mask, mdb_agg_state *inout) {
    if (!valid(ctx) || !inputs || !inout) return fail("mdb_agg_batch_range_list: NULL argument or closed context");
    log_call("agg_range_list", n_inputs, which_mask);
    log_call("agg_range", static_cast<uint64_t>(t_lo), static_cast<uint64_t>(t_hi));
    for (uint32_t k = 0; k < n_inputs; k++) {
        const mdb_segments *in = inputs[k];
        if (!in) return fail("mdb_agg_batch_range_list: a batch of the list is NULL");
        Hasher h;
        if (!hash_segments(*in, h)) return fail("mdb_agg_batch_range_list: a view points outside its data buffers");
        h.value(which_mask);
        h.value<uint32_t>(0x72616e67); // "rang"
        h.value(t_lo);
        h.value(t_hi);
#ifdef MDB_STUB_RECORD
        {
            mdb_agg_state fresh = {0.0, 0, FLT_MAX, -FLT_MAX};
            if (ora_agg_batch_range(in, t_lo, t_hi, which_mask, &fresh)) return fail(ora_last_error());
            Writer w;
            w.value(fresh);
            add_record(KIND_AGG, h.state, w.out);
        }
#endif
        const std::vector<uint8_t> *payload = find_record(KIND_AGG, h.state);
        if (!payload) return no_record("mdb_agg_batch_range_list", h.state);
        Reader reader(*payload);
        const mdb_agg_state batch = reader.value<mdb_agg_state>();
        if (!reader.ok) return fail("mdb_stub: damaged aggregate record");
        inout->sum += batch.sum;
        inout->count += batch.count;
        inout->min = std::fmin(inout->min, batch.min);
        inout->max = std::fmax(inout->max, batch.max);
    }
    return 0;
}

int mdb_compress_chunks(mdb_ctx *ctx, const int64_t *ts, const float *values, const uint64_t *chunk_offsets,
                        uint64_t n_chunks, mdb_error_bound error_bound, mdb_segments_owned **out) {
    if (!valid(ctx) || !chunk_offsets || !out || ((!ts || !values) && chunk_offsets[n_chunks] > chunk_offsets[0]))
        return fail("mdb_compress_chunks: NULL argument or closed context");
    const uint64_t key = fit_key(ts, values, chunk_offsets, n_chunks, error_bound);
#ifdef MDB_STUB_RECORD
    {
        mdb_segments_owned *made = nullptr;
        if (ora_compress_chunks(ts, values, chunk_offsets, n_chunks, error_bound, 1, &made)) return fail(ora_last_error());
        add_record(KIND_FIT, key, payload_from_segments(*made));
        ora_segments_free(made);
    }
#endif
    const std::vector<uint8_t> *payload = find_record(KIND_FIT, key);
    if (!payload) return no_record("mdb_compress_chunks", key);
    *out = segments_from_payload(*payload);
    return *out ? 0 : fail("mdb_stub: damaged fit record");
}

int mdb_compress_chunk_list(mdb_ctx *ctx, const mdb_chunk *chunks, uint64_t n_chunks, mdb_error_bound error_bound,
                            mdb_segments_owned **out) {
    if (!valid(ctx) || !out || (n_chunks > 0 && !chunks)) return fail("mdb_compress_chunk_list: NULL argument or closed context");
    log_call("compress_chunk_list", n_chunks, static_cast<uint64_t>(error_bound.kind));
    std::vector<int64_t> ts;
    std::vector<float> values;
    std::vector<uint64_t> offsets = {0};
    for (uint64_t c = 0; c < n_chunks; c++) {
        if (chunks[c].n > 0 && (!chunks[c].ts || !chunks[c].values)) return fail("mdb_compress_chunk_list: NULL chunk");
        ts.insert(ts.end(), chunks[c].ts, chunks[c].ts + chunks[c].n);
        values.insert(values.end(), chunks[c].values, chunks[c].values + chunks[c].n);
        offsets.push_back(ts.size());
    }
    ts.push_back(0); // (data() of an empty vector may be NULL)
    values.push_back(0.0f);
    return mdb_compress_chunks(ctx, ts.data(), values.data(), offsets.data(), n_chunks, error_bound, out);
}

int mdb_compress_series(mdb_ctx *ctx, const int64_t *ts, const float *values, uint64_t n,
                        mdb_error_bound error_bound, mdb_segments_owned **out) {
    const uint64_t offsets[2] = {0, n};
    return mdb_compress_chunks(ctx, ts, values, offsets, n ? 1 : 0, error_bound, out);
}

} // extern "C"
