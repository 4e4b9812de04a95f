// mdb_host.cpp - implementation of the host-side operators declared in mdb_host.hpp and of the C
// surface (mdbh_*) the tests drive them through. No model arithmetic happens here: grid, aggregate
// and fit go to libmdb_hip.so through the C ABI of include/mdb.h.
#include "mdb_host.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <thread>

#include <condition_variable>
#include <functional>
#include <mutex>


namespace mdbhost {

namespace {

void check(int code) {
    if (code != 0) throw Error(mdb_last_error());
}

size_t width_of(Type type) {
    switch (type) {
    case Type::Int8: return 1;
    case Type::Int16: return 2;
    case Type::Float32: return 4;
    case Type::Int64:
    case Type::Timestamp:
    case Type::Float64:
    case Type::UInt64: return 8;
    case Type::BinaryView:
    case Type::Utf8View: return 16;
    }
    return 0;
}

bool is_view(Type type) { return type == Type::BinaryView || type == Type::Utf8View; }

const char *format_of(Type type) {
    switch (type) {
    case Type::Int8: return "c";
    case Type::Int16: return "s";
    case Type::Int64: return "l";
    case Type::UInt64: return "L";
    case Type::Float32: return "f";
    case Type::Float64: return "g";
    case Type::Timestamp: return "tsu:";
    case Type::BinaryView: return "vz";
    case Type::Utf8View: return "vu";
    }
    return "";
}

Type type_of(const char *format) {
    std::string f(format);
    if (f == "c") return Type::Int8;
    if (f == "s") return Type::Int16;
    if (f == "l") return Type::Int64;
    if (f == "L") return Type::UInt64;
    if (f == "f") return Type::Float32;
    if (f == "g") return Type::Float64;
    if (f.rfind("tsu:", 0) == 0) return Type::Timestamp;
    if (f == "vz") return Type::BinaryView;
    if (f == "vu") return Type::Utf8View;
    throw Error("Unsupported Arrow format: " + f);
}

// Keeps an imported ArrowArray / ArrowSchema alive until the last column referencing it dies.
struct ImportHolder {
    ArrowArray array{};
    ArrowSchema schema{};
    ~ImportHolder() {
        if (array.release) array.release(&array);
        if (schema.release) schema.release(&schema);
    }
};

ColumnPtr column_from_c(const ArrowArray *array, const ArrowSchema *schema, std::shared_ptr<void> holder) {
    auto column = std::make_shared<Column>();
    column->type = type_of(schema->format);
    column->length = array->length;
    column->keep_alive = std::move(holder);
    if (array->null_count > 0) throw Error("Nullable columns are not supported on this path.");
    const size_t width = width_of(column->type);
    if (array->n_buffers < 2) throw Error("Malformed Arrow array: too few buffers.");
    const uint8_t *base = static_cast<const uint8_t *>(array->buffers[1]);
    column->values = base ? base + static_cast<size_t>(array->offset) * width : nullptr;
    if (is_view(column->type)) {
        // buffers: validity, views, variadic data buffers..., variadic buffer sizes (int64)
        const int64_t n_variadic = array->n_buffers - 3;
        if (n_variadic < 0) throw Error("Malformed Arrow view array.");
        const int64_t *sizes = static_cast<const int64_t *>(array->buffers[array->n_buffers - 1]);
        for (int64_t b = 0; b < n_variadic; b++) {
            column->buffer_ptrs.push_back(static_cast<const uint8_t *>(array->buffers[2 + b]));
            column->buffer_sizes.push_back(sizes ? sizes[b] : 0);
        }
    }
    return column;
}

// ---- export ---------------------------------------------------------------------------------------------

struct ExportedSchema {
    std::string name;
    std::vector<ArrowSchema> children;
    std::vector<ArrowSchema *> child_ptrs;
};

void release_schema(ArrowSchema *schema) {
    if (!schema || !schema->release) return;
    ExportedSchema *state = static_cast<ExportedSchema *>(schema->private_data);
    for (auto &child : state->children)
        if (child.release) child.release(&child); // a consumer that moved a child cleared its release
    delete state;
    schema->release = nullptr;
}

void fill_schema(ArrowSchema *out, const char *format, const std::string &name, ExportedSchema *state) {
    state->name = name;
    out->format = format;
    out->name = state->name.c_str();
    out->metadata = nullptr;
    out->flags = 0;
    out->n_children = 0;
    out->children = nullptr;
    out->dictionary = nullptr;
    out->release = release_schema;
    out->private_data = state;
}

struct ExportedArray {
    ColumnPtr column;
    std::vector<const void *> buffers;
    std::vector<int64_t> variadic_sizes;
    std::vector<ArrowArray> children;
    std::vector<ArrowArray *> child_ptrs;
};

void release_array(ArrowArray *array) {
    if (!array || !array->release) return;
    ExportedArray *state = static_cast<ExportedArray *>(array->private_data);
    for (auto &child : state->children)
        if (child.release) child.release(&child);
    delete state;
    array->release = nullptr;
}

void export_column(const ColumnPtr &column, ArrowArray *out) {
    auto *state = new ExportedArray();
    state->column = column;
    state->buffers.push_back(nullptr);
    state->buffers.push_back(column->values);
    if (is_view(column->type)) {
        for (size_t b = 0; b < column->buffer_ptrs.size(); b++) {
            state->buffers.push_back(column->buffer_ptrs[b]);
            state->variadic_sizes.push_back(column->buffer_sizes[b]);
        }
        if (state->variadic_sizes.empty()) state->variadic_sizes.push_back(0); // never a null pointer
        state->buffers.push_back(state->variadic_sizes.data());
    }
    out->length = column->length;
    out->null_count = 0;
    out->offset = 0;
    out->n_buffers = static_cast<int64_t>(state->buffers.size());
    out->n_children = 0;
    out->buffers = state->buffers.data();
    out->children = nullptr;
    out->dictionary = nullptr;
    out->release = release_array;
    out->private_data = state;
}

template <typename T> ColumnPtr owned_column(Type type, const std::vector<T> &values) {
    return make_primitive_column(type, values.data(), static_cast<int64_t>(values.size()));
}

struct ViewWord {
    int32_t length;
    uint8_t rest[12];
};
static_assert(sizeof(ViewWord) == 16, "Arrow views are 16 bytes");

} // namespace

// ---- Column / RecordBatch ---------------------------------------------------------------------------------

std::string_view Column::view_value(int64_t i) const {
    const mdb_view16 &view = as<mdb_view16>()[i];
    if (view.length <= 12)
        return {reinterpret_cast<const char *>(view.u.inlined), static_cast<size_t>(view.length)};
    return {reinterpret_cast<const char *>(buffer_ptrs[view.u.ref.buffer_index] + view.u.ref.offset),
            static_cast<size_t>(view.length)};
}

ColumnPtr make_primitive_column(Type type, const void *values, int64_t length) {
    auto column = std::make_shared<Column>();
    column->type = type;
    column->length = length;
    column->data.resize(std::max<size_t>(static_cast<size_t>(length) * width_of(type), 1));
    if (length > 0) std::memcpy(column->data.data(), values, static_cast<size_t>(length) * width_of(type));
    column->values = column->data.data();
    return column;
}

ColumnPtr make_view_column(Type type, const std::vector<std::string_view> &rows) {
    auto column = std::make_shared<Column>();
    column->type = type;
    column->length = static_cast<int64_t>(rows.size());
    column->data.assign(std::max<size_t>(rows.size() * 16, 16), 0);
    column->owned_buffers.emplace_back();
    std::vector<uint8_t> &payload = column->owned_buffers[0];
    mdb_view16 *views = reinterpret_cast<mdb_view16 *>(column->data.data());
    for (size_t i = 0; i < rows.size(); i++) {
        views[i].length = static_cast<int32_t>(rows[i].size());
        if (rows[i].size() <= 12) {
            std::memcpy(views[i].u.inlined, rows[i].data(), rows[i].size());
        } else {
            std::memcpy(views[i].u.ref.prefix, rows[i].data(), 4);
            views[i].u.ref.buffer_index = 0;
            views[i].u.ref.offset = static_cast<int32_t>(payload.size());
            payload.insert(payload.end(), rows[i].begin(), rows[i].end());
        }
    }
    column->values = column->data.data();
    if (!payload.empty()) {
        column->buffer_ptrs.push_back(payload.data());
        column->buffer_sizes.push_back(static_cast<int64_t>(payload.size()));
    }
    return column;
}

RecordBatch RecordBatch::new_empty(const std::vector<Field> &schema) {
    RecordBatch batch;
    batch.schema = schema;
    for (const Field &field : schema) {
        auto column = std::make_shared<Column>();
        column->type = field.type;
        column->data.assign(16, 0);
        column->values = column->data.data();
        batch.columns.push_back(column);
    }
    return batch;
}

RecordBatch RecordBatch::slice(int64_t offset, int64_t length) const {
    RecordBatch out;
    out.schema = schema;
    out.num_rows = length;
    for (const ColumnPtr &parent : columns) {
        auto column = std::make_shared<Column>();
        column->type = parent->type;
        column->length = length;
        column->values = static_cast<const uint8_t *>(parent->values) + static_cast<size_t>(offset) * width_of(parent->type);
        column->buffer_ptrs = parent->buffer_ptrs;
        column->buffer_sizes = parent->buffer_sizes;
        column->keep_alive = parent; // zero-copy: the slice shares the parent's storage
        out.columns.push_back(column);
    }
    return out;
}

ColumnPtr import_array(ArrowArray *array, ArrowSchema *schema) {
    auto holder = std::make_shared<ImportHolder>();
    holder->array = *array;
    holder->schema = *schema;
    array->release = nullptr;
    schema->release = nullptr;
    return column_from_c(&holder->array, &holder->schema, holder);
}

RecordBatch import_record_batch(ArrowArray *array, ArrowSchema *schema) {
    auto holder = std::make_shared<ImportHolder>();
    holder->array = *array;
    holder->schema = *schema;
    array->release = nullptr;
    schema->release = nullptr;
    if (std::string(holder->schema.format) != "+s") throw Error("A RecordBatch must be exported as a struct array.");
    if (holder->array.n_children != holder->schema.n_children) throw Error("Schema and array disagree.");
    RecordBatch batch;
    batch.num_rows = holder->array.length;
    for (int64_t c = 0; c < holder->array.n_children; c++) {
        const ArrowSchema *child_schema = holder->schema.children[c];
        ColumnPtr column = column_from_c(holder->array.children[c], child_schema, holder);
        if (holder->array.offset != 0) throw Error("Sliced struct arrays are not supported; slice the columns.");
        batch.schema.push_back({child_schema->name ? child_schema->name : "", column->type});
        batch.columns.push_back(column);
    }
    return batch;
}

void export_record_batch(const RecordBatch &batch, ArrowArray *out_array, ArrowSchema *out_schema) {
    auto *schema_state = new ExportedSchema();
    fill_schema(out_schema, "+s", "", schema_state);
    const size_t n = batch.columns.size();
    schema_state->children.resize(n);
    schema_state->child_ptrs.resize(n);
    for (size_t c = 0; c < n; c++) {
        // Each child owns its own state; the parent's release releases the children.
        fill_schema(&schema_state->children[c], format_of(batch.schema[c].type), batch.schema[c].name,
                    new ExportedSchema());
        schema_state->child_ptrs[c] = &schema_state->children[c];
    }
    out_schema->n_children = static_cast<int64_t>(n);
    out_schema->children = schema_state->child_ptrs.data();

    auto *array_state = new ExportedArray();
    array_state->buffers.push_back(nullptr);
    array_state->children.resize(n);
    array_state->child_ptrs.resize(n);
    for (size_t c = 0; c < n; c++) {
        export_column(batch.columns[c], &array_state->children[c]);
        array_state->child_ptrs[c] = &array_state->children[c];
    }
    out_array->length = batch.num_rows;
    out_array->null_count = 0;
    out_array->offset = 0;
    out_array->n_buffers = 1;
    out_array->n_children = static_cast<int64_t>(n);
    out_array->buffers = array_state->buffers.data();
    out_array->children = array_state->child_ptrs.data();
    out_array->dictionary = nullptr;
    out_array->release = release_array;
    out_array->private_data = array_state;
}

// ---- schemas ------------------------------------------------------------------------------------------------

std::vector<Field> query_compressed_schema() {
    return {{"model_type_id", Type::Int8},   {"start_time", Type::Timestamp}, {"end_time", Type::Timestamp},
            {"timestamps", Type::BinaryView}, {"min_value", Type::Float32},   {"max_value", Type::Float32},
            {"values", Type::BinaryView},     {"residuals", Type::BinaryView}, {"error", Type::Float32}};
}

std::vector<Field> compressed_schema(const std::vector<std::string> &tag_names) {
    std::vector<Field> schema = query_compressed_schema();
    schema.push_back({"field_column", Type::Int16});
    for (const std::string &tag : tag_names) schema.push_back({tag, Type::Utf8View});
    return schema;
}

std::vector<Field> grid_schema(const std::vector<std::string> &tag_names) {
    std::vector<Field> schema = {{"timestamp", Type::Timestamp}, {"value", Type::Float32}};
    for (const std::string &tag : tag_names) schema.push_back({tag, Type::Utf8View});
    return schema;
}

ErrorBound ErrorBound::try_new_absolute(float value) {
    if (!std::isfinite(value) || value <= 0.0f)
        throw Error("An absolute error bound must be a positive finite value.");
    ErrorBound eb;
    eb.c = {MDB_EB_ABSOLUTE, value};
    return eb;
}

ErrorBound ErrorBound::try_new_relative(float percentage) {
    if (!(0.0f < percentage && percentage <= 100.0f))
        throw Error("A relative error bound must be a positive value that is at most 100.0%.");
    ErrorBound eb;
    eb.c = {MDB_EB_RELATIVE, percentage};
    return eb;
}

// ---- QueueExec ----------------------------------------------------------------------------------------------

namespace {
struct QueueStream : SegmentStream {
    std::shared_ptr<QueueExec::State> state;
    PollState poll_next(RecordBatch *out) override {
        if (!state->queue.empty()) {
            *out = std::move(state->queue.front());
            state->queue.pop_front();
            return PollState::ReadySome;
        }
        return state->finished ? PollState::ReadyNone : PollState::Pending;
    }
};

// The mdb_segments view of the first eight columns of a segment batch.
struct SegmentsView {
    mdb_segments seg{};
    std::vector<const uint8_t *> pointers[3];
    std::vector<int64_t> sizes[3];
};

void fill_segments_view(const std::vector<ColumnPtr> &columns, SegmentsView *view) {
    if (columns.size() < 8) throw Error("A segment batch needs the columns of QUERY_COMPRESSED_SCHEMA.");
    const Type expected[8] = {Type::Int8, Type::Timestamp, Type::Timestamp, Type::BinaryView,
                              Type::Float32, Type::Float32, Type::BinaryView, Type::BinaryView};
    for (int c = 0; c < 8; c++) {
        Type got = columns[c]->type;
        if (got == Type::Int64 && expected[c] == Type::Timestamp) continue;
        if (got != expected[c]) throw Error("Segment column " + std::to_string(c) + " has the wrong type.");
    }
    mdb_segments &s = view->seg;
    s.n = static_cast<uint64_t>(columns[0]->length);
    s.model_type_id = columns[0]->as<int8_t>();
    s.start_time = columns[1]->as<int64_t>();
    s.end_time = columns[2]->as<int64_t>();
    s.min_value = columns[4]->as<float>();
    s.max_value = columns[5]->as<float>();
    const int view_columns[3] = {3, 6, 7};
    mdb_binview_col *out[3] = {&s.timestamps, &s.values, &s.residuals};
    for (int k = 0; k < 3; k++) {
        const Column &column = *columns[view_columns[k]];
        view->pointers[k] = column.buffer_ptrs;
        view->sizes[k] = column.buffer_sizes;
        out[k]->views = column.as<mdb_view16>();
        out[k]->buffers = view->pointers[k].data();
        out[k]->buffer_sizes = view->sizes[k].data();
        out[k]->n_buffers = static_cast<int32_t>(view->pointers[k].size());
    }
}
} // namespace

std::unique_ptr<SegmentStream> QueueExec::execute_segments(size_t) {
    auto stream = std::make_unique<QueueStream>();
    stream->state = state_;
    return stream;
}

// ---- GridExec / GridStream ----------------------------------------------------------------------------------

void GridStreamMetrics::add(const mdb_grid_metrics &m) {
    rows_created += m.rows_created;
    segments_with_residuals += m.segments_with_residuals;
    segments_regular += m.segments_regular;
    segments_irregular += m.segments_irregular;
    for (int k = 0; k < MDB_MODEL_TYPE_COUNT; k++) {
        rows_created_by_model_type[k] += m.rows_created_by_model_type[k];
        segments_with_model_type[k] += m.segments_with_model_type[k];
    }
}

std::shared_ptr<GridExec> GridExec::make(mdb_ctx *ctx, std::vector<Field> schema, ExprPtr maybe_predicate,
                                         std::optional<size_t> limit, std::shared_ptr<ExecutionPlan> input) {
    auto exec = std::make_shared<GridExec>();
    exec->ctx_ = ctx;
    exec->schema_ = std::move(schema);
    exec->maybe_predicate_ = std::move(maybe_predicate);
    exec->limit_ = limit;
    exec->input_ = std::move(input);
    return exec;
}

std::shared_ptr<GridExec> GridExec::with_new_children(std::vector<std::shared_ptr<ExecutionPlan>> children) const {
    if (children.size() != 1) throw Error("Exactly one child must be provided GridExec.");
    return GridExec::make(ctx_, schema_, maybe_predicate_, limit_, children[0]);
}

std::unique_ptr<GridStream> GridExec::execute(size_t partition, size_t batch_size) {
    return std::make_unique<GridStream>(ctx_, schema_, maybe_predicate_, limit_,
                                        input_->execute_segments(partition), batch_size, metrics_, values_only_);
}

std::unique_ptr<SegmentStream> GridExec::execute_stream(size_t partition, size_t batch_size) {
    return execute(partition, batch_size);
}

std::shared_ptr<GridExec> GridExec::with_values_only() const {
    auto exec = GridExec::make(ctx_, schema_, maybe_predicate_, limit_, input_);
    exec->metrics_ = metrics_;
    exec->values_only_ = true;
    return exec;
}

std::string GridExec::fmt_as() const {
    return std::string("GridExec: limit=") + (limit_ ? "Some(" + std::to_string(*limit_) + ")" : "None");
}

GridStream::GridStream(mdb_ctx *ctx, std::vector<Field> schema, ExprPtr maybe_predicate,
                       std::optional<size_t> limit, std::unique_ptr<SegmentStream> input, size_t batch_size,
                       std::shared_ptr<GridStreamMetrics> metrics, bool values_only)
    : ctx_(ctx), schema_(std::move(schema)), maybe_predicate_(std::move(maybe_predicate)), input_(std::move(input)),
      batch_size_(limit ? std::min(*limit, batch_size) : batch_size), // grid_exec.rs:239-246
      metrics_(std::move(metrics)), values_only_(values_only) {
    if (schema_.size() < 2) throw Error("GridStream should use a static schema.");
    // (rust/patches/0001: GridStream::new works the range out of maybe_predicate once)
    if (maybe_predicate_) pushed_range_ = time_range_of_predicate(*maybe_predicate_, schema_[0].name);
    if (values_only_ && maybe_predicate_ && !(pushed_range_ && pushed_range_->exact))
        throw Error("A values-only GridStream cannot evaluate a predicate that is not a time range.");
    if (values_only_) schema_ = {schema_[1]};
    current_batch_ = RecordBatch::new_empty(schema_);
    // MDB_HOST_GRID_PREFETCH=0: grid a batch when it is polled for and not before (A/B, tests); n: n submits ahead.
    if (const char *setting = std::getenv("MDB_HOST_GRID_PREFETCH"))
        prefetch_depth_ = static_cast<size_t>(std::min(std::max(std::atoi(setting), 0), 8));
    if (const char *text = std::getenv("MDB_HOST_GRID_COALESCE_SEGMENTS"))
        coalesce_segments_ = static_cast<size_t>(std::max(0ll, std::atoll(text)));
}

GridStream::~GridStream() {
    for (Ticket &ticket : ahead_) mdb_grid_cancel(ticket.raw); // (waits for the job and frees what it made)
}

// Data points a submit should decompress to, and the segments it may hold at most.
constexpr uint64_t GRID_SUBMIT_TARGET_POINTS = 16u << 20;
constexpr uint64_t GRID_SUBMIT_MAX_SEGMENTS = 1u << 20;

PollState GridStream::poll_input_and_submit(std::optional<Ticket> *out) {
    out->reset();
    if (input_finished_) return PollState::ReadyNone;
    // How many segments to gather: the batches of the input hold batch_size segments (8 192) whether they
    // decompress to 65 536 points or to 500 M; the first submit takes one batch, later ones what the results so
    // far say about 16 M points are.
    uint64_t wanted = 1;
    if (coalesce_segments_ > 0) {
        wanted = coalesce_segments_;
    } else if (seen_segments_ > 0) {
        const uint64_t points_per_segment = std::max<uint64_t>(1, seen_points_ / seen_segments_);
        wanted = std::min<uint64_t>(GRID_SUBMIT_TARGET_POINTS / points_per_segment, GRID_SUBMIT_MAX_SEGMENTS);
    }
    Ticket ticket;
    uint64_t gathered = 0;
    PollState state = PollState::ReadySome;
    while (gathered < std::max<uint64_t>(wanted, 1)) {
        RecordBatch batch;
        state = input_->poll_next(&batch);
        if (state != PollState::ReadySome) break;
        gathered += static_cast<uint64_t>(batch.num_rows);
        ticket.batches.push_back(std::make_shared<RecordBatch>(std::move(batch)));
    }
    if (state == PollState::ReadyNone) input_finished_ = true;
    if (ticket.batches.empty()) return state; // ReadyNone or Pending
    // One library call replaces the per-row loop of grid_exec.rs:323-356 for all gathered batches. The range the
    // predicate puts on the timestamps is pushed down so out-of-range points are neither reconstructed nor copied
    // over PCIe (the leftovers were filtered when they were created); if the predicate IS that range the filter
    // step of grid_exec.rs:366-387 has nothing left to remove and is skipped, otherwise it runs on what came back. The points arrive in page-locked memory owned by the library with room in front for the
    // leftovers (fewer than batch_size of them); the columns alias it. Tag views are repeated per created row
    // by the library (grid_exec.rs:339-346); the strings stay where they are: an output tag column lists one
    // buffer of its own (long leftover strings) and then the data buffers of every gathered batch.
    const size_t n_fixed = query_compressed_schema().size();
    const size_t n_tags = values_only_ ? 0 : ticket.batches[0]->columns.size() - n_fixed;
    std::vector<SegmentsView> views(ticket.batches.size());
    std::vector<mdb_grid_input> inputs(ticket.batches.size());
    std::vector<const mdb_view16 *> tag_views(ticket.batches.size() * n_tags);
    std::vector<int32_t> tag_shifts(ticket.batches.size() * n_tags);
    std::vector<int32_t> next_buffer(n_tags, 1);
    for (size_t b = 0; b < ticket.batches.size(); b++) {
        const RecordBatch &batch = *ticket.batches[b];
        if (!values_only_ && batch.columns.size() != n_fixed + n_tags) throw Error("GridStream should use a static schema.");
        fill_segments_view(batch.columns, &views[b]);
        inputs[b].segments = views[b].seg;
        for (size_t t = 0; t < n_tags; t++) {
            const Column &tags = *batch.columns[n_fixed + t];
            tag_views[b * n_tags + t] = tags.as<mdb_view16>();
            tag_shifts[b * n_tags + t] = next_buffer[t];
            next_buffer[t] += static_cast<int32_t>(tags.buffer_ptrs.size());
        }
        inputs[b].tag_views = n_tags ? tag_views.data() + b * n_tags : nullptr;
        inputs[b].tag_buffer_shift = n_tags ? tag_shifts.data() + b * n_tags : nullptr;
    }
    const bool pushdown = pushed_range_.has_value();
    mdb_grid_request request{};
    request.flags = (pushdown ? MDB_GRID_HAS_RANGE : 0u) | (values_only_ ? MDB_GRID_VALUES_ONLY : 0u);
    request.n_tag_columns = static_cast<uint32_t>(n_tags);
    request.t_lo = pushdown ? pushed_range_->range.lo : INT64_MIN;
    request.t_hi = pushdown ? pushed_range_->range.hi : INT64_MAX;
    request.reserve_front = batch_size_;
    check(mdb_grid_submit(ctx_, inputs.data(), static_cast<uint32_t>(inputs.size()), &request, &ticket.raw));
    *out = std::move(ticket);
    return PollState::ReadySome;
}

void GridStream::wait_and_append_to_leftovers_in_current_batch(Ticket ticket) {
    const auto started = std::chrono::steady_clock::now();
    const size_t n_fixed = query_compressed_schema().size();
    const size_t n_tags = values_only_ ? 0 : ticket.batches[0]->columns.size() - n_fixed;
    if (!values_only_ && schema_.size() != 2 + n_tags) {
        mdb_grid_cancel(ticket.raw);
        throw Error("GridStream should use a static schema.");
    }
    const int64_t leftovers = current_batch_.num_rows - current_batch_offset_;
    mdb_grid_result *raw = nullptr;
    check(mdb_grid_wait(ticket.raw, &raw)); // (the ticket is gone either way)
    std::shared_ptr<mdb_grid_result> result(raw, [](mdb_grid_result *r) { mdb_grid_result_free(r); });
    const int64_t total = leftovers + static_cast<int64_t>(result->n);
    int64_t *timestamps = values_only_ ? nullptr : result->timestamps - leftovers;
    float *values = result->values - leftovers;
    if (leftovers > 0) { // keep the batch sorted: leftovers first (grid_exec.rs:302-320)
        const size_t value_column = values_only_ ? 0 : 1;
        if (!values_only_)
            std::memcpy(timestamps, current_batch_.columns[0]->as<int64_t>() + current_batch_offset_, 8 * leftovers);
        std::memcpy(values, current_batch_.columns[value_column]->as<float>() + current_batch_offset_, 4 * leftovers);
    }
    metrics_->add(result->metrics);
    seen_segments_ += result->n_segments;
    seen_points_ += result->n;

    // Tag columns: the library has written every segment's view once per created row behind the room for the
    // leftovers (mdb_grid_result_tag_views); the leftovers' own views go in front. Long strings stay in the
    // input's data buffers (shared, not copied); those of the leftovers are copied so old inputs can be dropped.
    std::vector<ColumnPtr> tag_columns;
    for (size_t t = 0; t < n_tags; t++) {
        auto column = std::make_shared<Column>();
        column->type = Type::Utf8View;
        column->length = total;
        mdb_view16 *views = mdb_grid_result_tag_views(result.get(), static_cast<uint32_t>(t)) - leftovers;
        column->owned_buffers.emplace_back();
        std::vector<uint8_t> &leftover_payload = column->owned_buffers[0];
        if (leftovers > 0) {
            const Column &previous = *current_batch_.columns[2 + t];
            for (int64_t i = 0; i < leftovers; i++) {
                std::string_view value = previous.view_value(current_batch_offset_ + i);
                mdb_view16 view{};
                view.length = static_cast<int32_t>(value.size());
                if (value.size() <= 12) {
                    std::memcpy(view.u.inlined, value.data(), value.size());
                } else {
                    std::memcpy(view.u.ref.prefix, value.data(), 4);
                    view.u.ref.buffer_index = 0;
                    view.u.ref.offset = static_cast<int32_t>(leftover_payload.size());
                    leftover_payload.insert(leftover_payload.end(), value.begin(), value.end());
                }
                views[i] = view;
            }
        }
        column->values = views;
        column->buffer_ptrs.push_back(leftover_payload.empty() ? reinterpret_cast<const uint8_t *>(views)
                                                               : leftover_payload.data());
        column->buffer_sizes.push_back(static_cast<int64_t>(leftover_payload.size()));
        // (the same order poll_input_and_submit numbered them in: tag_buffer_shift)
        struct TagStorage {
            std::shared_ptr<mdb_grid_result> result;
            std::vector<ColumnPtr> inputs;
        };
        auto storage = std::make_shared<TagStorage>();
        storage->result = result;
        for (const auto &batch : ticket.batches) {
            const ColumnPtr &input_tags = batch->columns[n_fixed + t];
            for (size_t b = 0; b < input_tags->buffer_ptrs.size(); b++) {
                column->buffer_ptrs.push_back(input_tags->buffer_ptrs[b]);
                column->buffer_sizes.push_back(input_tags->buffer_sizes[b]);
            }
            storage->inputs.push_back(input_tags);
        }
        column->keep_alive = storage;
        tag_columns.push_back(column);
    }

    auto aliased = [&](Type type, const void *data) {
        auto column = std::make_shared<Column>();
        column->type = type;
        column->length = total;
        column->values = data;
        column->keep_alive = result;
        return column;
    };
    RecordBatch current;
    current.schema = schema_;
    current.num_rows = total;
    if (!values_only_) current.columns.push_back(aliased(Type::Timestamp, timestamps));
    current.columns.push_back(aliased(Type::Float32, values));
    for (ColumnPtr &column : tag_columns) current.columns.push_back(column);
    // grid_exec.rs:366-387: "all data points are reconstructed and then pruned by time" - here only what the pushed
    // range did not already decide. The leftovers in front passed the predicate when they were created.
    if (maybe_predicate_ && !(pushed_range_ && pushed_range_->exact)) {
        std::vector<uint8_t> mask = evaluate_predicate(*maybe_predicate_, current);
        std::fill(mask.begin(), mask.begin() + leftovers, 1);
        current = filter_record_batch(current, mask);
    }
    current_batch_ = std::move(current);
    current_batch_offset_ = 0; // grid_exec.rs:389-390
    metrics_->elapsed_compute_ns += static_cast<uint64_t>(
        std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - started).count());
}

PollState GridStream::poll_next(RecordBatch *out) {
    // grid_exec.rs:402-429
    if (static_cast<size_t>(current_batch_.num_rows - current_batch_offset_) < batch_size_) {
        std::optional<Ticket> ticket;
        if (!ahead_.empty()) {
            ticket = std::move(ahead_.front());
            ahead_.pop_front();
        }
        PollState state = PollState::ReadySome;
        if (!ticket) state = poll_input_and_submit(&ticket);
        if (ticket) {
            // The submits after this one start now: their cursors are walked and their segments go up while this
            // one's points come down (the library runs them on contexts of their own, mdb_grid_submit).
            try {
                while (ahead_.size() < prefetch_depth_) {
                    std::optional<Ticket> next;
                    (void)poll_input_and_submit(&next);
                    if (!next) break;
                    ahead_.push_back(std::move(*next));
                }
            } catch (...) {
                mdb_grid_cancel(ticket->raw);
                throw;
            }
            wait_and_append_to_leftovers_in_current_batch(std::move(*ticket));
        } else if (state == PollState::ReadyNone && current_batch_offset_ < current_batch_.num_rows) {
            // Ignore Ready(None): there are data points left in the current batch.
        } else {
            return state;
        }
    }
    const int64_t remaining = current_batch_.num_rows - current_batch_offset_;
    const int64_t length = std::min<int64_t>(static_cast<int64_t>(batch_size_), remaining);
    *out = current_batch_.slice(current_batch_offset_, length);
    current_batch_offset_ += length;
    metrics_->output_rows += static_cast<uint64_t>(length);
    return PollState::ReadySome;
}

// ---- SortedJoinExec / SortedJoinStream ------------------------------------------------------------------------

std::shared_ptr<SortedJoinExec> SortedJoinExec::make(std::vector<Field> schema,
                                                     std::vector<SortedJoinColumnType> return_order,
                                                     std::vector<std::shared_ptr<ExecutionPlan>> inputs) {
    if (inputs.empty()) throw Error("At least one child must be provided SortedJoinExec.");
    auto exec = std::make_shared<SortedJoinExec>();
    exec->schema_ = std::move(schema);
    exec->return_order_ = std::move(return_order);
    for (size_t i = 1; i < inputs.size(); i++)
        if (auto grid = std::dynamic_pointer_cast<GridExec>(inputs[i]))
            if (!grid->values_only()) inputs[i] = grid->with_values_only();
    exec->inputs_ = std::move(inputs);
    return exec;
}

std::shared_ptr<SortedJoinExec>
SortedJoinExec::with_new_children(std::vector<std::shared_ptr<ExecutionPlan>> children) const {
    if (children.empty()) throw Error("At least one child must be provided SortedJoinExec.");
    return SortedJoinExec::make(schema_, return_order_, std::move(children));
}

std::unique_ptr<SortedJoinStream> SortedJoinExec::execute(size_t partition, size_t batch_size) {
    std::vector<std::unique_ptr<SegmentStream>> streams;
    for (auto &input : inputs_) streams.push_back(input->execute_stream(partition, batch_size));
    return std::make_unique<SortedJoinStream>(schema_, return_order_, std::move(streams), output_rows_);
}

std::unique_ptr<SegmentStream> SortedJoinExec::execute_stream(size_t partition, size_t batch_size) {
    return execute(partition, batch_size);
}

SortedJoinStream::SortedJoinStream(std::vector<Field> schema, std::vector<SortedJoinColumnType> return_order,
                                   std::vector<std::unique_ptr<SegmentStream>> inputs,
                                   std::shared_ptr<uint64_t> output_rows)
    : schema_(std::move(schema)), return_order_(std::move(return_order)), inputs_(std::move(inputs)),
      batches_(inputs_.size()), output_rows_(std::move(output_rows)) {
    const char *setting = std::getenv("MDB_HOST_SORTED_JOIN_CARRY_OVER");
    carry_over_ = setting && std::string(setting) == "1";
}

std::optional<PollState> SortedJoinStream::poll_all_pending_inputs() {
    // Every input without a batch is polled; the last poll that was not Ready(Some) is the reason
    // returned, like the reference (a finished input ends the join, a pending one suspends it).
    std::optional<PollState> reason_for_not_ok;
    for (size_t index = 0; index < batches_.size(); index++) {
        if (batches_[index]) continue;
        RecordBatch batch;
        PollState poll = inputs_[index]->poll_next(&batch);
        if (poll == PollState::ReadySome)
            batches_[index] = std::move(batch);
        else
            reason_for_not_ok = poll;
    }
    return reason_for_not_ok;
}

void SortedJoinStream::set_batch_num_rows_to_smallest() {
    // Inputs can differ in length (sorted_join_exec.rs:248-272): compressed segments are not transferred
    // atomically, and a GridStream emits a short batch whenever a predicate leaves it with fewer than batch_size
    // points (grid_exec.rs:419-423). The reference cuts every batch to the smallest and DROPS the surplus - and
    // so does this stream by default: results identical to the reference's. (Dropping also shifts every later row
    // of that input against the others; MDB_HOST_SORTED_JOIN_CARRY_OVER=1 keeps the surplus for the next poll
    // instead, which leaves the rows aligned. tests/test_host_ops_cpu.py shows both on the same inputs.)
    int64_t smallest = INT64_MAX;
    for (const auto &batch : batches_) smallest = std::min(smallest, batch->num_rows);
    surplus_.assign(batches_.size(), std::nullopt);
    for (size_t index = 0; index < batches_.size(); index++) {
        if (batches_[index]->num_rows == smallest) continue;
        if (carry_over_) surplus_[index] = batches_[index]->slice(smallest, batches_[index]->num_rows - smallest);
        batches_[index] = batches_[index]->slice(0, smallest);
    }
}

RecordBatch SortedJoinStream::sorted_join() const {
    const RecordBatch &first = *batches_[0];
    RecordBatch out;
    out.schema = schema_;
    out.num_rows = first.num_rows;
    size_t field_index = 0;
    for (const SortedJoinColumnType &element : return_order_) {
        switch (element.kind) {
        case SortedJoinColumnType::Kind::Timestamp:
            out.columns.push_back(first.columns.at(0));
            break;
        case SortedJoinColumnType::Kind::Field: {
            if (field_index >= batches_.size()) throw Error("SortedJoinStream has fewer inputs than field columns.");
            const RecordBatch &batch = *batches_[field_index++];
            // A values-only input has the value as its single column; a full one at index 1.
            out.columns.push_back(batch.columns.at(batch.columns.size() == 1 ? 0 : 1));
            break;
        }
        case SortedJoinColumnType::Kind::Tag: {
            ColumnPtr found;
            for (size_t c = 0; c < first.schema.size(); c++)
                if (first.schema[c].name == element.tag_name) found = first.columns[c];
            if (!found) throw Error("All tag columns should be in the schema.");
            out.columns.push_back(found);
            break;
        }
        }
    }
    if (out.columns.size() != schema_.size())
        throw Error("SortedJoinStream should have ordered columns to match the schema.");
    return out;
}

PollState SortedJoinStream::poll_next(RecordBatch *out) {
    if (auto reason_for_not_ok = poll_all_pending_inputs()) return *reason_for_not_ok;
    set_batch_num_rows_to_smallest();
    *out = sorted_join();
    for (size_t index = 0; index < batches_.size(); index++) batches_[index] = std::move(surplus_[index]);
    *output_rows_ += static_cast<uint64_t>(out->num_rows);
    return PollState::ReadySome;
}

// ---- accumulators ---------------------------------------------------------------------------------------------

namespace {

// Segments an accumulator collects before they are folded into its state with one call: DataFusion hands update_batch
// 8 192 at a time, a call costs the same for 8 192 and for 262 144 (rust/patches/0002: PENDING_SEGMENTS_PER_CALL).
constexpr size_t PENDING_SEGMENTS_PER_CALL = 262144;

class ModelAccumulator : public Accumulator {
  public:
    // Under a range COUNT is always asked for as well: a range without a data point gives NULL, not the identity.
    ModelAccumulator(mdb_ctx *ctx, uint32_t mask, std::optional<TimeRange> range)
        : ctx_(ctx), mask_(range ? (mask | MDB_AGG_COUNT) : mask), range_(range) {
        reset();
    }
    // (PendingSegments::push of the patched accumulators: the columns are kept, not copied, until enough segments are
    // pending or the state is read - nobody can observe it in between)
    void update_batch(const std::vector<ColumnPtr> &arrays) override {
        pending_segments_ += arrays.empty() ? 0 : static_cast<size_t>(arrays[0]->length);
        pending_.push_back(arrays);
        if (pending_segments_ >= PENDING_SEGMENTS_PER_CALL) fold_pending();
    }
    size_t size() const override { return sizeof(*this) + pending_segments_ * 73; }

  protected:
    // PendingSegments::fold_into: ALL pending batches through one mdb_agg_batch_list.
    void fold_pending() {
        if (pending_.empty()) return;
        std::vector<SegmentsView> views(pending_.size());
        std::vector<const mdb_segments *> inputs(pending_.size());
        for (size_t k = 0; k < pending_.size(); k++) {
            fill_segments_view(pending_[k], &views[k]);
            inputs[k] = &views[k].seg;
        }
        // (PendingSegments::fold_into of rust/patches/0002: aggregate_list, or aggregate_range_list under a range)
        const int code =
            range_ ? mdb_agg_batch_range_list(ctx_, inputs.data(), static_cast<uint32_t>(inputs.size()), range_->lo,
                                              range_->hi, mask_, &state_)
                   : mdb_agg_batch_list(ctx_, inputs.data(), static_cast<uint32_t>(inputs.size()), mask_, &state_);
        pending_.clear();
        pending_segments_ = 0;
        check(code);
    }
    void reset() {
        state_.sum = 0.0;
        state_.count = 0;
        state_.min = std::numeric_limits<float>::max();    // f32::MAX (:413)
        state_.max = std::numeric_limits<float>::lowest(); // f32::MIN (:456)
    }
    // MIN / MAX / SUM over no data point at all: NULL under a range (see make_model_*_accumulator), the identity
    // element without one (the reference's own behaviour for an empty table, :410-415, 453-458, 523-528).
    bool nothing_in_range() const { return range_ && state_.count == 0; }
    mdb_ctx *ctx_;
    uint32_t mask_;
    std::optional<TimeRange> range_;
    mdb_agg_state state_;
    std::vector<std::vector<ColumnPtr>> pending_;
    size_t pending_segments_ = 0;
};

struct ModelCountAccumulator : ModelAccumulator {
    ModelCountAccumulator(mdb_ctx *ctx, std::optional<TimeRange> range) : ModelAccumulator(ctx, MDB_AGG_COUNT, range) {}
    std::vector<ScalarValue> state() override { // :367-372
        fold_pending();
        ScalarValue v{ScalarValue::Kind::Int64};
        v.i64 = state_.count;
        reset();
        return {v};
    }
};

struct ModelMinAccumulator : ModelAccumulator {
    ModelMinAccumulator(mdb_ctx *ctx, std::optional<TimeRange> range) : ModelAccumulator(ctx, MDB_AGG_MIN, range) {}
    std::vector<ScalarValue> state() override { // :410-415
        fold_pending();
        ScalarValue v{ScalarValue::Kind::Float32};
        v.f32 = state_.min;
        v.null = nothing_in_range();
        reset();
        return {v};
    }
};

struct ModelMaxAccumulator : ModelAccumulator {
    ModelMaxAccumulator(mdb_ctx *ctx, std::optional<TimeRange> range) : ModelAccumulator(ctx, MDB_AGG_MAX, range) {}
    std::vector<ScalarValue> state() override { // :453-458
        fold_pending();
        ScalarValue v{ScalarValue::Kind::Float32};
        v.f32 = state_.max;
        v.null = nothing_in_range();
        reset();
        return {v};
    }
};

struct ModelSumAccumulator : ModelAccumulator {
    ModelSumAccumulator(mdb_ctx *ctx, std::optional<TimeRange> range) : ModelAccumulator(ctx, MDB_AGG_SUM, range) {}
    std::vector<ScalarValue> state() override { // :523-528
        fold_pending();
        ScalarValue v{ScalarValue::Kind::Float64};
        v.f64 = state_.sum;
        v.null = nothing_in_range();
        reset();
        return {v};
    }
};

struct ModelAvgAccumulator : ModelAccumulator {
    ModelAvgAccumulator(mdb_ctx *ctx, std::optional<TimeRange> range) : ModelAccumulator(ctx, MDB_AGG_AVG, range) {}
    std::vector<ScalarValue> state() override { // :597-606: [UInt64 count, Float64 sum]
        fold_pending();
        ScalarValue count{ScalarValue::Kind::UInt64};
        count.u64 = static_cast<uint64_t>(state_.count);
        ScalarValue sum{ScalarValue::Kind::Float64};
        sum.f64 = state_.sum;
        reset();
        return {count, sum};
    }
};

} // namespace

std::unique_ptr<Accumulator> make_model_count_accumulator(mdb_ctx *ctx, std::optional<TimeRange> range) {
    return std::make_unique<ModelCountAccumulator>(ctx, range);
}
std::unique_ptr<Accumulator> make_model_min_accumulator(mdb_ctx *ctx, std::optional<TimeRange> range) {
    return std::make_unique<ModelMinAccumulator>(ctx, range);
}
std::unique_ptr<Accumulator> make_model_max_accumulator(mdb_ctx *ctx, std::optional<TimeRange> range) {
    return std::make_unique<ModelMaxAccumulator>(ctx, range);
}
std::unique_ptr<Accumulator> make_model_sum_accumulator(mdb_ctx *ctx, std::optional<TimeRange> range) {
    return std::make_unique<ModelSumAccumulator>(ctx, range);
}
std::unique_ptr<Accumulator> make_model_avg_accumulator(mdb_ctx *ctx, std::optional<TimeRange> range) {
    return std::make_unique<ModelAvgAccumulator>(ctx, range);
}

// ---- compression ------------------------------------------------------------------------------------------------

namespace {

// Rows [first, last) of a host-resident owned batch as a RecordBatch with the compressed schema
// (CompressedSegmentBatchBuilder::finish, crates/modelardb_compression/src/types.rs:492-516).
RecordBatch record_batch_from_owned(const mdb_segments_owned *owned, uint64_t first, uint64_t last,
                                    const std::vector<Field> &schema, const std::vector<std::string> &tag_values,
                                    int16_t field_column_index) {
    const uint64_t n = last - first;
    const size_t n_fixed = query_compressed_schema().size() + 1;
    if (schema.size() != n_fixed + tag_values.size())
        throw Error("compressed_schema does not match the number of tag values.");
    const mdb_segments &s = owned->seg;
    RecordBatch batch;
    batch.schema = schema;
    batch.num_rows = static_cast<int64_t>(n);
    auto view_column = [&](const mdb_binview_col &col) {
        // Re-pack the rows' payloads so the batch owns exactly what it references.
        std::vector<std::string_view> rows;
        rows.reserve(n);
        for (uint64_t i = first; i < last; i++) {
            const mdb_view16 &view = col.views[i];
            const uint8_t *bytes = view.length <= 12 ? view.u.inlined
                                                     : col.buffers[view.u.ref.buffer_index] + view.u.ref.offset;
            rows.emplace_back(reinterpret_cast<const char *>(bytes), static_cast<size_t>(view.length));
        }
        return make_view_column(Type::BinaryView, rows);
    };
    batch.columns.push_back(make_primitive_column(Type::Int8, s.model_type_id + first, n));
    batch.columns.push_back(make_primitive_column(Type::Timestamp, s.start_time + first, n));
    batch.columns.push_back(make_primitive_column(Type::Timestamp, s.end_time + first, n));
    batch.columns.push_back(view_column(s.timestamps));
    batch.columns.push_back(make_primitive_column(Type::Float32, s.min_value + first, n));
    batch.columns.push_back(make_primitive_column(Type::Float32, s.max_value + first, n));
    batch.columns.push_back(view_column(s.values));
    batch.columns.push_back(view_column(s.residuals));
    std::vector<float> error(n, std::numeric_limits<float>::quiet_NaN());
    batch.columns.push_back(owned_column(Type::Float32, error));
    std::vector<int16_t> field_column(n, field_column_index);
    batch.columns.push_back(owned_column(Type::Int16, field_column));
    for (const std::string &tag : tag_values) {
        std::vector<std::string_view> rows(n, std::string_view(tag));
        batch.columns.push_back(make_view_column(Type::Utf8View, rows));
    }
    return batch;
}

struct OwnedGuard {
    mdb_segments_owned *owned = nullptr;
    ~OwnedGuard() { mdb_segments_free(owned); }
};

// One series x field to compress: where its sorted data points lie, and what the resulting batch is labelled with.
// (rust/modelardb_hip: SeriesChunk.)
struct SeriesChunk {
    const int64_t *timestamps;
    const float *values;
    uint64_t n;
    mdb_error_bound error_bound;
    const std::vector<std::string> *tag_values;
    int16_t field_column_index;
};

bool same_bound(const mdb_error_bound &a, const mdb_error_bound &b) {
    return a.kind == b.kind && (a.kind == MDB_EB_LOSSLESS || a.value == b.value);
}

// All chunks through ONE mdb_compress_chunk_list per distinct error bound (an error bound is an argument of the
// launch), the resulting batches in the order of the chunks. rust/modelardb_hip: Context::compress_chunks - the
// call the patched try_compress_multivariate_time_series and process_compressor_messages make.
std::vector<RecordBatch> compress_chunks(mdb_ctx *ctx, const std::vector<SeriesChunk> &chunks,
                                         const std::vector<Field> &compressed_schema) {
    std::vector<RecordBatch> result(chunks.size());
    std::vector<bool> done(chunks.size(), false);
    for (size_t first = 0; first < chunks.size(); first++) {
        if (done[first]) continue;
        std::vector<size_t> group;
        std::vector<mdb_chunk> list;
        for (size_t c = first; c < chunks.size(); c++) {
            if (done[c] || !same_bound(chunks[c].error_bound, chunks[first].error_bound)) continue;
            done[c] = true;
            group.push_back(c);
            list.push_back({chunks[c].timestamps, chunks[c].values, chunks[c].n});
        }
        OwnedGuard guard;
        check(mdb_compress_chunk_list(ctx, list.data(), list.size(), chunks[first].error_bound, &guard.owned));
        // Segments come back grouped by chunk, in chunk order.
        uint64_t row = 0;
        const uint64_t total = guard.owned->seg.n;
        for (size_t k = 0; k < group.size(); k++) {
            const uint64_t begin = row;
            while (row < total && guard.owned->chunk_index[row] == k) row++;
            const SeriesChunk &chunk = chunks[group[k]];
            result[group[k]] = record_batch_from_owned(guard.owned, begin, row, compressed_schema, *chunk.tag_values,
                                                       chunk.field_column_index);
        }
    }
    return result;
}

} // namespace

RecordBatch try_compress_univariate_time_series(mdb_ctx *ctx, const Column &uncompressed_timestamps,
                                                const Column &uncompressed_values, ErrorBound error_bound,
                                                const std::vector<Field> &compressed_schema,
                                                const std::vector<std::string> &tag_values,
                                                int16_t field_column_index) {
    if (uncompressed_timestamps.length != uncompressed_values.length) // compression.rs:202-206
        throw Error("Invalid Argument Error: Uncompressed timestamps and uncompressed values have different lengths.");
    if (uncompressed_timestamps.length == 0) return RecordBatch::new_empty(compressed_schema); // :208-211
    OwnedGuard guard;
    check(mdb_compress_series(ctx, uncompressed_timestamps.as<int64_t>(), uncompressed_values.as<float>(),
                              static_cast<uint64_t>(uncompressed_values.length), error_bound.c, &guard.owned));
    return record_batch_from_owned(guard.owned, 0, guard.owned->seg.n, compressed_schema, tag_values,
                                   field_column_index);
}

std::vector<RecordBatch> try_compress_multivariate_time_series(mdb_ctx *ctx,
                                                               const TimeSeriesTableMetadata &metadata,
                                                               const RecordBatch &batch) {
    const int64_t n = batch.num_rows;
    if (n == 0) throw Error("The uncompressed time series must contain at least one data point.");
    const Column &ts_column = *batch.columns[metadata.timestamp_column_index];
    std::vector<const Column *> tags;
    for (size_t index : metadata.tag_column_indices) tags.push_back(batch.columns[index].get());

    // sort_time_series_by_tags_and_time (compression.rs:111-141): lexsort(tags..., timestamp).
    std::vector<int64_t> order(static_cast<size_t>(n));
    std::iota(order.begin(), order.end(), 0);
    const int64_t *ts = ts_column.as<int64_t>();
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        for (const Column *tag : tags) {
            int cmp = tag->view_value(a).compare(tag->view_value(b));
            if (cmp != 0) return cmp < 0;
        }
        return ts[a] < ts[b];
    });

    // Split on tag change (compression.rs:64-104).
    struct Series {
        int64_t first, last; // [first, last) in `order`
        std::vector<std::string> tag_values;
    };
    std::vector<Series> series;
    auto tags_of = [&](int64_t row) {
        std::vector<std::string> values;
        for (const Column *tag : tags) values.emplace_back(tag->view_value(row));
        return values;
    };
    int64_t start = 0;
    std::vector<std::string> current_tags = tags_of(order[0]);
    for (int64_t i = 1; i <= n; i++) {
        bool boundary = i == n;
        if (!boundary) {
            for (size_t t = 0; t < tags.size(); t++)
                if (tags[t]->view_value(order[static_cast<size_t>(i)]) != current_tags[t]) boundary = true;
        }
        if (boundary) {
            series.push_back({start, i, current_tags});
            if (i < n) {
                start = i;
                current_tags = tags_of(order[static_cast<size_t>(i)]);
            }
        }
    }

    // The sorted columns (take_arrays, compression.rs:138), then every series x field as one chunk of ONE launch
    // per error bound (the reference: one try_compress_univariate_time_series per series and field, :159-176).
    const size_t n_fields = metadata.field_column_indices.size();
    std::vector<int64_t> sorted_ts(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; i++) sorted_ts[static_cast<size_t>(i)] = ts[order[static_cast<size_t>(i)]];
    std::vector<std::vector<float>> sorted_values(n_fields, std::vector<float>(static_cast<size_t>(n)));
    for (size_t f = 0; f < n_fields; f++) {
        const float *values = batch.columns[metadata.field_column_indices[f]]->as<float>();
        for (int64_t i = 0; i < n; i++) sorted_values[f][static_cast<size_t>(i)] = values[order[static_cast<size_t>(i)]];
    }
    std::vector<SeriesChunk> chunks;
    chunks.reserve(series.size() * n_fields);
    for (const Series &s : series) {
        for (size_t f = 0; f < n_fields; f++) {
            const size_t field_index = metadata.field_column_indices[f];
            chunks.push_back({sorted_ts.data() + s.first, sorted_values[f].data() + s.first,
                              static_cast<uint64_t>(s.last - s.first), metadata.error_bounds[field_index].c,
                              &s.tag_values, static_cast<int16_t>(field_index)});
        }
    }
    return compress_chunks(ctx, chunks, metadata.compressed_schema);
}

// ---- UncompressedDataManager --------------------------------------------------------------------------------

UncompressedDataManager::UncompressedDataManager(mdb_ctx *ctx, TimeSeriesTableMetadata metadata,
                                                 size_t buffer_capacity)
    : ctx_(ctx), metadata_(std::move(metadata)), capacity_(buffer_capacity) {
    if (capacity_ == 0) throw Error("The buffer capacity must be positive.");
}

void UncompressedDataManager::insert_data_points(const RecordBatch &data_points) {
    const Column &ts_column = *data_points.columns[metadata_.timestamp_column_index];
    std::vector<const Column *> tags, fields;
    for (size_t index : metadata_.tag_column_indices) tags.push_back(data_points.columns[index].get());
    for (size_t index : metadata_.field_column_indices) fields.push_back(data_points.columns[index].get());
    const uint64_t batch_index = current_batch_index_;
    // calculate_tag_hash stands in: the tag values identify the series. Rows of several series usually arrive
    // interleaved in a fixed order, so the buffer behind the one the previous row went to is tried first.
    auto is_series_of = [&](const Buffer &buffer, int64_t row) {
        for (size_t t = 0; t < tags.size(); t++)
            if (tags[t]->view_value(row) != buffer.tag_values[t]) return false;
        return true;
    };
    size_t previous = 0;
    for (int64_t row = 0; row < data_points.num_rows; row++) {
        auto it = active_.end();
        if (!active_.empty()) {
            const size_t guess = (previous + 1) % active_.size();
            if (is_series_of(active_[guess].second, row)) it = active_.begin() + static_cast<std::ptrdiff_t>(guess);
            else if (is_series_of(active_[previous % active_.size()].second, row))
                it = active_.begin() + static_cast<std::ptrdiff_t>(previous % active_.size());
            else
                it = std::find_if(active_.begin(), active_.end(), [&](const auto &kv) { return is_series_of(kv.second, row); });
        }
        if (it == active_.end()) {
            std::string key;
            Buffer buffer;
            for (const Column *tag : tags) {
                buffer.tag_values.emplace_back(tag->view_value(row));
                key.append(tag->view_value(row));
                key.push_back('\x1f');
            }
            buffer.values.resize(fields.size());
            active_.emplace_back(key, std::move(buffer));
            it = active_.end() - 1;
        }
        previous = static_cast<size_t>(it - active_.begin());
        Buffer &buffer = it->second;
        buffer.updated_by_batch_index = batch_index; // uncompressed_data_buffer.rs:141-158
        buffer.timestamps.push_back(ts_column.as<int64_t>()[row]);
        for (size_t f = 0; f < fields.size(); f++) buffer.values[f].push_back(fields[f]->as<float>()[row]);
        if (buffer.timestamps.size() == capacity_) { // is_full(): transfer to the compressor (:301-318)
            finished_.push_back(std::move(buffer));
            active_.erase(it);
            previous = previous > 0 ? previous - 1 : (active_.empty() ? 0 : active_.size() - 1); // (the next series has moved into its place)
        }
    }
    // Unused buffers are only finished at the end so buffers needed by this batch survive (:170-174).
    // fetch_add(1) hands finish_unused_buffers the index of the batch just ingested (:175-176).
    current_batch_index_ += 1;
    finish_unused_buffers(batch_index);
}

void UncompressedDataManager::finish_unused_buffers(uint64_t current_batch_index) {
    // is_unused(): updated_by_batch_index + RECORD_BATCH_OFFSET_REQUIRED_FOR_UNUSED <= current (:135-137)
    for (size_t i = 0; i < active_.size();) {
        if (active_[i].second.updated_by_batch_index + 1 <= current_batch_index) {
            finished_.push_back(std::move(active_[i].second));
            active_.erase(active_.begin() + static_cast<std::ptrdiff_t>(i));
        } else {
            i++;
        }
    }
}

void UncompressedDataManager::flush() {
    for (auto &kv : active_) finished_.push_back(std::move(kv.second));
    active_.clear();
}

std::vector<RecordBatch> UncompressedDataManager::compress_finished_buffers() {
    // What the patched process_compressor_messages does with the buffers it drained from its channel
    // (rust/patches/0004-uncompressed_data_manager.patch): every finished buffer x field column is one chunk, all of
    // them go through ONE launch per error bound, and the batches come back in the order buffer by buffer, field
    // by field - where the reference compresses one buffer at a time on its single compression thread
    // (uncompressed_data_manager.rs:505-596).
    const size_t n_fields = metadata_.field_column_indices.size();
    // record_batch(): sort each buffer by time (sort_to_indices + take, uncompressed_data_buffer.rs:181-201).
    // A buffer that is in time order already - a sensor sends its points that way - is used where it lies.
    struct Sorted {
        std::vector<int64_t> timestamps;
        std::vector<std::vector<float>> values;
    };
    std::vector<std::unique_ptr<Sorted>> sorted(finished_.size());
    for (size_t b = 0; b < finished_.size(); b++) {
        const std::vector<int64_t> &ts = finished_[b].timestamps;
        if (std::is_sorted(ts.begin(), ts.end())) continue;
        std::vector<int64_t> order(ts.size());
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return ts[x] < ts[y]; });
        sorted[b] = std::make_unique<Sorted>();
        sorted[b]->timestamps.resize(ts.size());
        sorted[b]->values.assign(n_fields, std::vector<float>(ts.size()));
        for (size_t j = 0; j < ts.size(); j++) {
            const size_t i = static_cast<size_t>(order[j]);
            sorted[b]->timestamps[j] = ts[i];
            for (size_t f = 0; f < n_fields; f++) sorted[b]->values[f][j] = finished_[b].values[f][i];
        }
    }
    std::vector<SeriesChunk> chunks;
    chunks.reserve(finished_.size() * n_fields);
    for (size_t b = 0; b < finished_.size(); b++) {
        for (size_t f = 0; f < n_fields; f++) {
            const size_t field_index = metadata_.field_column_indices[f];
            const std::vector<int64_t> &ts = sorted[b] ? sorted[b]->timestamps : finished_[b].timestamps;
            const std::vector<float> &values = sorted[b] ? sorted[b]->values[f] : finished_[b].values[f];
            chunks.push_back({ts.data(), values.data(), ts.size(), metadata_.error_bounds[field_index].c,
                              &finished_[b].tag_values, static_cast<int16_t>(field_index)});
        }
    }
    std::vector<RecordBatch> result = compress_chunks(ctx_, chunks, metadata_.compressed_schema);
    finished_.clear();
    return result;
}

} // namespace mdbhost

// =================================================================================================
// C surface for tests and non-C++ callers. 0 = ok, 1 = error (message via mdbh_last_error()).
// =================================================================================================

namespace {
thread_local std::string g_host_error;

template <typename F> int guarded(F &&body) {
    try {
        body();
        return 0;
    } catch (const std::exception &e) {
        g_host_error = e.what();
        return 1;
    }
}

struct GridHandle {
    std::shared_ptr<mdbhost::QueueExec> input;
    std::shared_ptr<mdbhost::GridExec> exec;
    std::unique_ptr<mdbhost::GridStream> stream;
};

struct JoinHandle {
    std::vector<std::shared_ptr<mdbhost::QueueExec>> inputs;
    std::shared_ptr<mdbhost::SortedJoinExec> exec;
    std::unique_ptr<mdbhost::SortedJoinStream> stream;
};
} // namespace

extern "C" {

const char *mdbh_last_error(void) { return g_host_error.c_str(); }
/* (for the other translation units of the library: sets the calling thread's error text) */
int mdbh_fail(const char *message) {
    g_host_error = message ? message : "";
    return 1;
}

int mdbh_grid_exec_create(mdb_ctx *ctx, const char *const *tag_names, int32_t n_tags, int64_t limit,
                          int32_t has_lower, int64_t lower, int32_t has_upper, int64_t upper,
                          uint64_t batch_size, void **out) {
    return guarded([&] {
        using namespace mdbhost;
        std::vector<std::string> tags(tag_names, tag_names + n_tags);
        std::vector<Field> input_schema = query_compressed_schema();
        for (const std::string &tag : tags) input_schema.push_back({tag, Type::Utf8View});
        auto handle = std::make_unique<GridHandle>();
        handle->input = std::make_shared<QueueExec>(input_schema);
        ExprPtr predicate = timestamp_range_predicate(has_lower ? std::optional<int64_t>(lower) : std::nullopt,
                                                      has_upper ? std::optional<int64_t>(upper) : std::nullopt);
        handle->exec = GridExec::make(ctx, grid_schema(tags), predicate,
                                      limit >= 0 ? std::optional<size_t>(static_cast<size_t>(limit)) : std::nullopt,
                                      handle->input);
        handle->stream = handle->exec->execute(0, batch_size);
        *out = handle.release();
    });
}

/* The same with the predicate GridExec is handed as an expression over (timestamp, value, tags...), in the text form
 * of mdbhost::parse_expr (NULL or "": none). */
int mdbh_grid_exec_create_expr(mdb_ctx *ctx, const char *const *tag_names, int32_t n_tags, int64_t limit,
                               const char *predicate, uint64_t batch_size, void **out) {
    return guarded([&] {
        using namespace mdbhost;
        std::vector<std::string> tags(tag_names, tag_names + n_tags);
        std::vector<Field> input_schema = query_compressed_schema();
        for (const std::string &tag : tags) input_schema.push_back({tag, Type::Utf8View});
        auto handle = std::make_unique<GridHandle>();
        handle->input = std::make_shared<QueueExec>(input_schema);
        ExprPtr expr = predicate && *predicate ? parse_expr(predicate) : nullptr;
        handle->exec = GridExec::make(ctx, grid_schema(tags), expr,
                                      limit >= 0 ? std::optional<size_t>(static_cast<size_t>(limit)) : std::nullopt,
                                      handle->input);
        handle->stream = handle->exec->execute(0, batch_size);
        *out = handle.release();
    });
}

int mdbh_grid_stream_push(void *handle, ArrowArray *array, ArrowSchema *schema) {
    return guarded([&] { static_cast<GridHandle *>(handle)->input->push(mdbhost::import_record_batch(array, schema)); });
}

int mdbh_grid_stream_finish_input(void *handle) {
    return guarded([&] { static_cast<GridHandle *>(handle)->input->finish(); });
}

/* state: 0 = Ready(Some(batch)) (out filled), 1 = Ready(None), 2 = Pending */
int mdbh_grid_stream_poll_next(void *handle, ArrowArray *out_array, ArrowSchema *out_schema, int32_t *state) {
    return guarded([&] {
        mdbhost::RecordBatch batch;
        mdbhost::PollState poll = static_cast<GridHandle *>(handle)->stream->poll_next(&batch);
        *state = poll == mdbhost::PollState::ReadySome ? 0 : (poll == mdbhost::PollState::ReadyNone ? 1 : 2);
        if (poll == mdbhost::PollState::ReadySome) mdbhost::export_record_batch(batch, out_array, out_schema);
    });
}

/* out[12]: rows_created, by type x3, segments_with_residuals, segments by type x3, regular,
 * irregular, output_rows, elapsed_compute_ns */
int mdbh_grid_stream_metrics(void *handle, uint64_t *out) {
    return guarded([&] {
        const mdbhost::GridStreamMetrics &m = *static_cast<GridHandle *>(handle)->exec->metrics();
        out[0] = m.rows_created;
        for (int k = 0; k < 3; k++) out[1 + k] = m.rows_created_by_model_type[k];
        out[4] = m.segments_with_residuals;
        for (int k = 0; k < 3; k++) out[5 + k] = m.segments_with_model_type[k];
        out[8] = m.segments_regular;
        out[9] = m.segments_irregular;
        out[10] = m.output_rows;
        out[11] = m.elapsed_compute_ns;
    });
}

int mdbh_grid_exec_describe(void *handle, char *out, uint64_t cap) {
    return guarded([&] {
        GridHandle *h = static_cast<GridHandle *>(handle);
        std::string text = std::string(h->exec->name()) + "|" + h->exec->fmt_as() + "|children=" +
                           std::to_string(h->exec->children().size()) + "|batch_size=" +
                           std::to_string(h->stream->batch_size()) + "|distribution=" +
                           (h->exec->required_input_distribution()[0] == mdbhost::Distribution::SinglePartition
                                ? "SinglePartition" : "Unspecified");
        try {
            h->exec->with_new_children({});
        } catch (const mdbhost::Error &e) {
            text += std::string("|with_new_children([])=Err(") + e.what() + ")";
        }
        std::strncpy(out, text.c_str(), cap - 1);
        out[cap - 1] = 0;
    });
}

/* Polls the stream to its end without exporting the batches (the measurement of bench.py's host_path:
 * the operator itself, not the cost of moving every 8 192-row batch into Python). rows / batches:
 * what came out; checksum: the sum of the first timestamp (or value bits) of every batch. */
int mdbh_grid_stream_drain(void *handle, uint64_t *rows, uint64_t *batches, uint64_t *checksum) {
    return guarded([&] {
        mdbhost::GridStream &stream = *static_cast<GridHandle *>(handle)->stream;
        uint64_t n_rows = 0, n_batches = 0, sum = 0;
        while (true) {
            mdbhost::RecordBatch batch;
            const mdbhost::PollState state = stream.poll_next(&batch);
            if (state != mdbhost::PollState::ReadySome) break;
            n_rows += static_cast<uint64_t>(batch.num_rows);
            n_batches += 1;
            if (batch.num_rows > 0 && !batch.columns.empty()) {
                uint64_t first = 0;
                std::memcpy(&first, batch.columns[0]->values, batch.columns[0]->type == mdbhost::Type::Float32 ? 4 : 8);
                sum += first;
            }
        }
        *rows = n_rows;
        *batches = n_batches;
        *checksum = sum;
    });
}

void mdbh_grid_stream_free(void *handle) { delete static_cast<GridHandle *>(handle); }

/* SortedJoinExec over one input per field column. use_grid != 0: every input is a GridExec over a
 * hand-fed queue of segment batches (needs the GPU); use_grid == 0: the inputs are hand-fed queues of
 * data point batches (timestamp, value, tags...), which exercises the join logic alone.
 * return_kinds: 0 timestamp, 1 field, 2 tag (name in return_tag_names at the same position). */
int mdbh_sorted_join_create(mdb_ctx *ctx, int32_t n_fields, const char *const *tag_names, int32_t n_tags,
                            const int32_t *return_kinds, const char *const *return_tag_names, int32_t n_return,
                            int32_t use_grid, int64_t limit, int32_t has_lower, int64_t lower, int32_t has_upper,
                            int64_t upper, uint64_t batch_size, void **out) {
    return guarded([&] {
        using namespace mdbhost;
        std::vector<std::string> tags(tag_names, tag_names + n_tags);
        std::vector<Field> segment_schema = query_compressed_schema();
        for (const std::string &tag : tags) segment_schema.push_back({tag, Type::Utf8View});
        ExprPtr predicate = timestamp_range_predicate(has_lower ? std::optional<int64_t>(lower) : std::nullopt,
                                                      has_upper ? std::optional<int64_t>(upper) : std::nullopt);
        auto handle = std::make_unique<JoinHandle>();
        std::vector<std::shared_ptr<ExecutionPlan>> inputs;
        for (int32_t f = 0; f < n_fields; f++) {
            auto queue = std::make_shared<QueueExec>(use_grid ? segment_schema : grid_schema(tags));
            handle->inputs.push_back(queue);
            if (use_grid)
                inputs.push_back(GridExec::make(
                    ctx, grid_schema(tags), predicate,
                    limit >= 0 ? std::optional<size_t>(static_cast<size_t>(limit)) : std::nullopt, queue));
            else
                inputs.push_back(queue);
        }
        std::vector<Field> schema;
        std::vector<SortedJoinColumnType> return_order;
        int32_t field_index = 0;
        for (int32_t r = 0; r < n_return; r++) {
            if (return_kinds[r] == 0) {
                schema.push_back({"timestamp", Type::Timestamp});
                return_order.push_back(SortedJoinColumnType::timestamp());
            } else if (return_kinds[r] == 1) {
                schema.push_back({"field_" + std::to_string(field_index++), Type::Float32});
                return_order.push_back(SortedJoinColumnType::field());
            } else {
                schema.push_back({return_tag_names[r], Type::Utf8View});
                return_order.push_back(SortedJoinColumnType::tag(return_tag_names[r]));
            }
        }
        handle->exec = SortedJoinExec::make(schema, return_order, inputs);
        handle->stream = handle->exec->execute(0, batch_size);
        *out = handle.release();
    });
}

int mdbh_sorted_join_push(void *handle, int32_t input, ArrowArray *array, ArrowSchema *schema) {
    return guarded([&] {
        static_cast<JoinHandle *>(handle)->inputs.at(static_cast<size_t>(input))->push(
            mdbhost::import_record_batch(array, schema));
    });
}

int mdbh_sorted_join_finish_input(void *handle, int32_t input) {
    return guarded([&] { static_cast<JoinHandle *>(handle)->inputs.at(static_cast<size_t>(input))->finish(); });
}

int mdbh_sorted_join_poll_next(void *handle, ArrowArray *out_array, ArrowSchema *out_schema, int32_t *state) {
    return guarded([&] {
        mdbhost::RecordBatch batch;
        mdbhost::PollState poll = static_cast<JoinHandle *>(handle)->stream->poll_next(&batch);
        *state = poll == mdbhost::PollState::ReadySome ? 0 : (poll == mdbhost::PollState::ReadyNone ? 1 : 2);
        if (poll == mdbhost::PollState::ReadySome) mdbhost::export_record_batch(batch, out_array, out_schema);
    });
}

/* Polls the join to its end inside the library (what a consumer written in a compiled language costs):
 * rows and batches returned, for measurements. */
int mdbh_sorted_join_drain(void *handle, uint64_t *rows, uint64_t *batches) {
    return guarded([&] {
        mdbhost::SortedJoinStream &stream = *static_cast<JoinHandle *>(handle)->stream;
        uint64_t n_rows = 0, n_batches = 0;
        while (true) {
            mdbhost::RecordBatch batch;
            if (stream.poll_next(&batch) != mdbhost::PollState::ReadySome) break;
            n_rows += static_cast<uint64_t>(batch.num_rows);
            n_batches += 1;
        }
        *rows = n_rows;
        *batches = n_batches;
    });
}

int mdbh_sorted_join_describe(void *handle, char *out, uint64_t cap) {
    return guarded([&] {
        JoinHandle *h = static_cast<JoinHandle *>(handle);
        std::string text = std::string(h->exec->name()) + "|" + h->exec->fmt_as() + "|children=" +
                           std::to_string(h->exec->children().size()) + "|distribution=";
        for (mdbhost::Distribution d : h->exec->required_input_distribution())
            text += d == mdbhost::Distribution::SinglePartition ? "SinglePartition," : "Unspecified,";
        text += "|values_only=";
        for (auto &child : h->exec->children()) {
            auto grid = std::dynamic_pointer_cast<mdbhost::GridExec>(child);
            text += grid ? (grid->values_only() ? "1" : "0") : "-";
        }
        text += "|output_rows=" + std::to_string(h->exec->output_rows());
        try {
            h->exec->with_new_children({});
        } catch (const mdbhost::Error &e) {
            text += std::string("|with_new_children([])=Err(") + e.what() + ")";
        }
        std::strncpy(out, text.c_str(), cap - 1);
        out[cap - 1] = 0;
    });
}

void mdbh_sorted_join_free(void *handle) { delete static_cast<JoinHandle *>(handle); }

/* kind: 0 count, 1 min, 2 max, 3 sum, 4 avg; has_range: only the data points with t_lo <= timestamp <= t_hi */
int mdbh_accumulator_create_range(mdb_ctx *ctx, int32_t kind, int32_t has_range, int64_t t_lo, int64_t t_hi, void **out) {
    return guarded([&] {
        std::unique_ptr<mdbhost::Accumulator> acc;
        std::optional<mdbhost::TimeRange> range;
        if (has_range) range = mdbhost::TimeRange{t_lo, t_hi};
        switch (kind) {
        case 0: acc = mdbhost::make_model_count_accumulator(ctx, range); break;
        case 1: acc = mdbhost::make_model_min_accumulator(ctx, range); break;
        case 2: acc = mdbhost::make_model_max_accumulator(ctx, range); break;
        case 3: acc = mdbhost::make_model_sum_accumulator(ctx, range); break;
        case 4: acc = mdbhost::make_model_avg_accumulator(ctx, range); break;
        default: throw mdbhost::Error("Aggregate expression is not supported.");
        }
        *out = acc.release();
    });
}

int mdbh_accumulator_create(mdb_ctx *ctx, int32_t kind, void **out) {
    return mdbh_accumulator_create_range(ctx, kind, 0, 0, 0, out);
}

int mdbh_accumulator_update_batch(void *acc, ArrowArray *array, ArrowSchema *schema) {
    return guarded([&] {
        mdbhost::RecordBatch batch = mdbhost::import_record_batch(array, schema);
        static_cast<mdbhost::Accumulator *>(acc)->update_batch(batch.columns);
    });
}

/* Writes up to two state values: kinds[i] (0 i64, 1 u64, 2 f32, 3 f64), the value as f64 /
 * i64 bit patterns in values_f64 / values_i64 and, if nulls is not NULL, whether it is NULL. Returns how many via
 * n_values. */
int mdbh_accumulator_state(void *acc, int32_t *kinds, double *values_f64, int64_t *values_i64, int32_t *n_values,
                           int32_t *nulls) {
    return guarded([&] {
        auto state = static_cast<mdbhost::Accumulator *>(acc)->state();
        *n_values = static_cast<int32_t>(state.size());
        for (size_t i = 0; i < state.size(); i++) {
            kinds[i] = static_cast<int32_t>(state[i].kind);
            if (nulls) nulls[i] = state[i].null ? 1 : 0;
            switch (state[i].kind) {
            case mdbhost::ScalarValue::Kind::Int64: values_i64[i] = state[i].i64; values_f64[i] = 0; break;
            case mdbhost::ScalarValue::Kind::UInt64: values_i64[i] = static_cast<int64_t>(state[i].u64); values_f64[i] = 0; break;
            case mdbhost::ScalarValue::Kind::Float32: values_f64[i] = state[i].f32; values_i64[i] = 0; break;
            case mdbhost::ScalarValue::Kind::Float64: values_f64[i] = state[i].f64; values_i64[i] = 0; break;
            }
        }
    });
}

int mdbh_accumulator_unreachable(void *acc, int32_t which) {
    return guarded([&] {
        if (which == 0) static_cast<mdbhost::Accumulator *>(acc)->merge_batch({});
        else static_cast<mdbhost::Accumulator *>(acc)->evaluate();
    });
}

uint64_t mdbh_accumulator_size(void *acc) { return static_cast<mdbhost::Accumulator *>(acc)->size(); }
void mdbh_accumulator_free(void *acc) { delete static_cast<mdbhost::Accumulator *>(acc); }

int mdbh_try_compress_univariate_time_series(mdb_ctx *ctx, ArrowArray *ts_array, ArrowSchema *ts_schema,
                                             ArrowArray *values_array, ArrowSchema *values_schema,
                                             mdb_error_bound error_bound, const char *const *tag_names,
                                             const char *const *tag_values, int32_t n_tags,
                                             int16_t field_column_index, ArrowArray *out_array,
                                             ArrowSchema *out_schema) {
    return guarded([&] {
        using namespace mdbhost;
        ColumnPtr ts = import_array(ts_array, ts_schema);
        ColumnPtr values = import_array(values_array, values_schema);
        std::vector<std::string> names(tag_names, tag_names + n_tags), tag_vals(tag_values, tag_values + n_tags);
        ErrorBound eb;
        if (error_bound.kind == MDB_EB_ABSOLUTE) eb = ErrorBound::try_new_absolute(error_bound.value);
        else if (error_bound.kind == MDB_EB_RELATIVE) eb = ErrorBound::try_new_relative(error_bound.value);
        RecordBatch batch = try_compress_univariate_time_series(ctx, *ts, *values, eb, compressed_schema(names),
                                                                tag_vals, field_column_index);
        export_record_batch(batch, out_array, out_schema);
    });
}

/* Compresses a multivariate batch; the resulting RecordBatches are kept in a handle and fetched
 * one by one. error_bounds has one entry per COLUMN of the input batch (ignored for non-fields). */
int mdbh_try_compress_multivariate_time_series(mdb_ctx *ctx, ArrowArray *array, ArrowSchema *schema,
                                               int32_t timestamp_column, const int32_t *field_columns,
                                               int32_t n_fields, const int32_t *tag_columns, int32_t n_tags,
                                               const mdb_error_bound *error_bounds, void **out_handle,
                                               int32_t *n_batches) {
    return guarded([&] {
        using namespace mdbhost;
        RecordBatch batch = import_record_batch(array, schema);
        TimeSeriesTableMetadata metadata;
        metadata.timestamp_column_index = static_cast<size_t>(timestamp_column);
        std::vector<std::string> tag_names;
        for (int32_t i = 0; i < n_fields; i++) metadata.field_column_indices.push_back(static_cast<size_t>(field_columns[i]));
        for (int32_t i = 0; i < n_tags; i++) {
            metadata.tag_column_indices.push_back(static_cast<size_t>(tag_columns[i]));
            tag_names.push_back(batch.schema[static_cast<size_t>(tag_columns[i])].name);
        }
        for (size_t c = 0; c < batch.columns.size(); c++) {
            ErrorBound eb;
            eb.c = error_bounds[c];
            metadata.error_bounds.push_back(eb);
        }
        metadata.compressed_schema = compressed_schema(tag_names);
        auto *result = new std::vector<RecordBatch>(try_compress_multivariate_time_series(ctx, metadata, batch));
        *out_handle = result;
        *n_batches = static_cast<int32_t>(result->size());
    });
}

int mdbh_batches_get(void *handle, int32_t index, ArrowArray *out_array, ArrowSchema *out_schema) {
    return guarded([&] {
        auto *batches = static_cast<std::vector<mdbhost::RecordBatch> *>(handle);
        mdbhost::export_record_batch(batches->at(static_cast<size_t>(index)), out_array, out_schema);
    });
}

void mdbh_batches_free(void *handle) { delete static_cast<std::vector<mdbhost::RecordBatch> *>(handle); }

/* ---- UncompressedDataManager ---- */
int mdbh_udm_create(mdb_ctx *ctx, int32_t timestamp_column, const int32_t *field_columns, int32_t n_fields,
                    const int32_t *tag_columns, const char *const *tag_names, int32_t n_tags,
                    const mdb_error_bound *error_bounds, int32_t n_columns, uint64_t capacity, void **out) {
    return guarded([&] {
        using namespace mdbhost;
        TimeSeriesTableMetadata metadata;
        metadata.timestamp_column_index = static_cast<size_t>(timestamp_column);
        for (int32_t i = 0; i < n_fields; i++) metadata.field_column_indices.push_back(static_cast<size_t>(field_columns[i]));
        std::vector<std::string> names;
        for (int32_t i = 0; i < n_tags; i++) {
            metadata.tag_column_indices.push_back(static_cast<size_t>(tag_columns[i]));
            names.emplace_back(tag_names[i]);
        }
        for (int32_t c = 0; c < n_columns; c++) {
            ErrorBound eb;
            eb.c = error_bounds[c];
            metadata.error_bounds.push_back(eb);
        }
        metadata.compressed_schema = compressed_schema(names);
        *out = new UncompressedDataManager(ctx, metadata, capacity);
    });
}

int mdbh_udm_insert_data_points(void *handle, ArrowArray *array, ArrowSchema *schema) {
    return guarded([&] {
        static_cast<mdbhost::UncompressedDataManager *>(handle)->insert_data_points(
            mdbhost::import_record_batch(array, schema));
    });
}

int mdbh_udm_flush(void *handle) {
    return guarded([&] { static_cast<mdbhost::UncompressedDataManager *>(handle)->flush(); });
}

int mdbh_udm_counts(void *handle, uint64_t *active, uint64_t *finished) {
    return guarded([&] {
        auto *manager = static_cast<mdbhost::UncompressedDataManager *>(handle);
        *active = manager->active_buffer_count();
        *finished = manager->finished_buffer_count();
    });
}

int mdbh_udm_compress_finished_buffers(void *handle, void **out_batches, int32_t *n_batches) {
    return guarded([&] {
        auto *result = new std::vector<mdbhost::RecordBatch>(
            static_cast<mdbhost::UncompressedDataManager *>(handle)->compress_finished_buffers());
        *out_batches = result;
        *n_batches = static_cast<int32_t>(result->size());
    });
}

void mdbh_udm_free(void *handle) { delete static_cast<mdbhost::UncompressedDataManager *>(handle); }

} // extern "C"
