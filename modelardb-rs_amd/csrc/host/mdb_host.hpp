// mdb_host.hpp - host-side operators above the C ABI, written in C++ because the reference's host
// code is compiled Rust and no Rust toolchain exists in this image. They mirror the reference's
// operator interface for the hot path - same names, argument meaning and error behaviour - so a
// maintainer can map each class 1:1 onto the Rust type it stands in for:
//
//   GridExec / GridStream / GridStreamMetrics   crates/modelardb_storage/src/query/grid_exec.rs:56-519
//   Model{Count,Min,Max,Sum,Avg}Accumulator     crates/modelardb_storage/src/optimizer/model_simple_aggregates.rs:336-618
//   try_compress_univariate_time_series         crates/modelardb_compression/src/compression.rs:191-275
//   try_compress_multivariate_time_series       crates/modelardb_compression/src/compression.rs:42-179
//
// RecordBatches are exchanged through the Arrow C Data Interface. All model arithmetic happens in
// libmdb_hip.so; this layer only moves columns, replicates tags, filters and re-batches.
#pragma once

#include <cstdint>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mdb.h"
#include "arrow_c_abi.h"

namespace mdbhost {

// Error type: what the Rust code would return as Err(...) (DataFusionError / ModelarDbCompressionError).
struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---- columns and batches ---------------------------------------------------------------------------------

enum class Type { Int8, Int16, Int64, Timestamp, Float32, Float64, UInt64, BinaryView, Utf8View };

struct Field {
    std::string name;
    Type type;
};

// One column. Primitive data lives in `data`; view types use `data` for the 16-byte views and
// `buffers` for the variadic data buffers. Imported columns may instead point into foreign memory
// kept alive by `keep_alive`.
struct Column {
    Type type = Type::Int8;
    int64_t length = 0;
    const void *values = nullptr;                 // primitive values or views
    std::vector<const uint8_t *> buffer_ptrs;     // view types: variadic buffers
    std::vector<int64_t> buffer_sizes;
    std::vector<uint8_t> data;                    // owned storage (may be empty when imported)
    std::vector<std::vector<uint8_t>> owned_buffers;
    std::shared_ptr<void> keep_alive;

    template <typename T> const T *as() const { return static_cast<const T *>(values); }
    std::string_view view_value(int64_t i) const; // bytes of row i of a view column
};
using ColumnPtr = std::shared_ptr<Column>;

struct RecordBatch {
    std::vector<Field> schema;
    std::vector<ColumnPtr> columns;
    int64_t num_rows = 0;

    static RecordBatch new_empty(const std::vector<Field> &schema);
    RecordBatch slice(int64_t offset, int64_t length) const;
};

// Arrow C Data Interface. import takes ownership of *array and *schema (moves them).
RecordBatch import_record_batch(ArrowArray *array, ArrowSchema *schema);
ColumnPtr import_array(ArrowArray *array, ArrowSchema *schema);
void export_record_batch(const RecordBatch &batch, ArrowArray *out_array, ArrowSchema *out_schema);

ColumnPtr make_primitive_column(Type type, const void *values, int64_t length);
ColumnPtr make_view_column(Type type, const std::vector<std::string_view> &rows);

// ---- schemas (crates/modelardb_types/src/schemas.rs:31-72) -------------------------------------------------

std::vector<Field> query_compressed_schema();                                       // 9 columns
std::vector<Field> compressed_schema(const std::vector<std::string> &tag_names);    // + field_column + tags
std::vector<Field> grid_schema(const std::vector<std::string> &tag_names);          // timestamp, value, tags

// ErrorBound (crates/modelardb_types/src/types.rs:299-335). The constructors throw Error like
// try_new_absolute / try_new_relative return Err.
struct ErrorBound {
    mdb_error_bound c{MDB_EB_LOSSLESS, 0.0f};
    static ErrorBound lossless() { return {}; }
    static ErrorBound try_new_absolute(float value);
    static ErrorBound try_new_relative(float percentage);
};

// ---- the input of GridExec: any stream of segment batches ---------------------------------------------------

enum class PollState { ReadySome, ReadyNone, Pending };

struct SegmentStream { // SendableRecordBatchStream of the child plan
    virtual ~SegmentStream() = default;
    virtual PollState poll_next(RecordBatch *out) = 0;
};

struct ExecutionPlan { // the slice of datafusion's ExecutionPlan that GridExec touches
    virtual ~ExecutionPlan() = default;
    virtual const char *name() const = 0;
    virtual std::vector<Field> schema() const = 0;
    virtual std::vector<std::shared_ptr<ExecutionPlan>> children() const = 0;
    virtual std::unique_ptr<SegmentStream> execute_segments(size_t partition) = 0;
    // ExecutionPlan::execute as a parent operator sees it: a stream of RecordBatches (SegmentStream
    // is just that). Plans that need the session's batch_size override it.
    virtual std::unique_ptr<SegmentStream> execute_stream(size_t partition, size_t /*batch_size*/) {
        return execute_segments(partition);
    }
    // ExecutionPlan::with_new_children as the optimizer rules use it (a copy of this node over other inputs).
    virtual std::shared_ptr<ExecutionPlan> with_new_children_dyn(std::vector<std::shared_ptr<ExecutionPlan>>) const {
        throw Error(std::string(name()) + " does not support new children.");
    }
};

// A child plan fed by hand: batches are pushed, then the input is finished (used by tests and by
// the C API; in the reference the child is DataSourceExec over Parquet).
class QueueExec : public ExecutionPlan {
  public:
    explicit QueueExec(std::vector<Field> schema) : schema_(std::move(schema)) {}
    const char *name() const override { return "QueueExec"; }
    std::vector<Field> schema() const override { return schema_; }
    std::vector<std::shared_ptr<ExecutionPlan>> children() const override { return {}; }
    std::unique_ptr<SegmentStream> execute_segments(size_t partition) override;
    void push(RecordBatch batch) { state_->queue.push_back(std::move(batch)); }
    void finish() { state_->finished = true; }

    struct State {
        std::deque<RecordBatch> queue;
        bool finished = false;
    };

  private:
    std::vector<Field> schema_;
    std::shared_ptr<State> state_ = std::make_shared<State>();
};

// ---- expressions: the slice of datafusion's Expr / PhysicalExpr the path's predicates are made of ------------------

enum class Operator { Lt, LtEq, Gt, GtEq, Eq, NotEq, And, Or };
const char *operator_text(Operator op); // "<", "<=", ..., "AND", "OR"

struct Scalar { // ScalarValue of a literal
    enum class Kind { TimestampMicrosecond, Int64, Float32, Utf8 } kind = Kind::Int64;
    int64_t i64 = 0;
    float f32 = 0.0f;
    std::string utf8;
};

struct Expr;
using ExprPtr = std::shared_ptr<const Expr>;
struct Expr { // Expr::Column | Expr::Literal | Expr::BinaryExpr
    enum class Kind { Column, Literal, BinaryExpr } kind = Kind::Column;
    std::string column; // Column
    Scalar literal;     // Literal
    Operator op = Operator::Eq;
    ExprPtr left, right; // BinaryExpr
    static ExprPtr col(std::string name);
    static ExprPtr lit(Scalar value);
    static ExprPtr lit_timestamp(int64_t microseconds);
    static ExprPtr binary(ExprPtr left, Operator op, ExprPtr right);
    std::string to_string() const; // "timestamp >= TimestampMicrosecond(1000)"
};
// "(and (>= timestamp ts:1000) (< timestamp ts:5000))": columns by name, literals as ts:<i64>, i64:<i64>,
// f32:<float>, str:<text>; operators < <= > >= = != and or. What the C surface and the tests write predicates in.
ExprPtr parse_expr(const std::string &text);
// utils::conjunction: None for no expressions, otherwise e0 AND e1 AND ...
ExprPtr conjunction(const std::vector<ExprPtr> &exprs);
// One byte per row: does the row satisfy `predicate` (columns looked up by name in batch.schema).
std::vector<uint8_t> evaluate_predicate(const Expr &predicate, const struct RecordBatch &batch);
// arrow::compute::filter_record_batch: the rows whose mask byte is set, in order. All set: the batch itself.
RecordBatch filter_record_batch(const RecordBatch &batch, const std::vector<uint8_t> &mask);

// The inclusive range of timestamps a predicate lets through, as mdb_grid_submit / mdb_agg_batch_range_list take it.
struct TimeRange {
    int64_t lo = INT64_MIN, hi = INT64_MAX;
    bool empty() const { return lo > hi; }
};
struct PredicateRange {
    TimeRange range;
    // true: the predicate IS the range (a conjunction of comparisons of the timestamp column with timestamp
    // literals - the only shape rewrite_filter emits for GridExec, query/time_series_table.rs:290-373), so rows the
    // library returns under the range need no filter behind them. false: the range is only implied by the predicate
    // (some conjuncts are something else) and the predicate is still evaluated on what comes back.
    bool exact = true;
};
// None if no conjunct of `predicate` is a comparison (<, <=, >, >=, =) of column `timestamp_column` with a
// TimestampMicrosecond literal (either side); anything that is not such a comparison or an AND of them makes the
// result inexact, an OR / != anywhere above a comparison keeps it out of the range.
std::optional<PredicateRange> time_range_of_predicate(const Expr &predicate, const std::string &timestamp_column);

// rewrite_filter / rewrite_and_combine_filters (query/time_series_table.rs:269-373): the filters of a query, written
// over the table's query schema, as (a filter over the segments' start_time / end_time for the Parquet scan, a filter
// over `timestamp` for GridExec); filters on anything but the timestamp column are not rewritten (None).
struct RewrittenFilters {
    ExprPtr parquet; // may be null
    ExprPtr grid;    // may be null
};
std::optional<std::pair<ExprPtr, ExprPtr>> rewrite_filter(const std::vector<Field> &query_schema, const Expr &filter);
RewrittenFilters rewrite_and_combine_filters(const std::vector<Field> &query_schema, const std::vector<ExprPtr> &filters);

// ---- GridExec ------------------------------------------------------------------------------------------------

// The predicate TimeSeriesTable::scan hands GridExec (`maybe_predicate`, grid_exec.rs:60, 370-387) for the two-sided
// range lower <= timestamp <= upper, either side optional: what the older C surface passes as two integers.
ExprPtr timestamp_range_predicate(std::optional<int64_t> lower, std::optional<int64_t> upper);

struct GridStreamMetrics { // grid_exec.rs:441-518 (+ BaselineMetrics' output_rows / elapsed_compute)
    uint64_t rows_created = 0;
    uint64_t rows_created_by_model_type[MDB_MODEL_TYPE_COUNT] = {0, 0, 0};
    uint64_t segments_with_residuals = 0;
    uint64_t segments_with_model_type[MDB_MODEL_TYPE_COUNT] = {0, 0, 0};
    uint64_t segments_regular = 0;
    uint64_t segments_irregular = 0;
    uint64_t output_rows = 0;
    uint64_t elapsed_compute_ns = 0;
    void add(const mdb_grid_metrics &m);
};

enum class Distribution { UnspecifiedDistribution, SinglePartition };

class GridStream;

class GridExec : public ExecutionPlan, public std::enable_shared_from_this<GridExec> {
  public:
    // GridExec::new (grid_exec.rs:76-109). `schema` is the grid schema (timestamp, value, tags...).
    static std::shared_ptr<GridExec> make(mdb_ctx *ctx, std::vector<Field> schema, ExprPtr maybe_predicate,
                                          std::optional<size_t> limit,
                                          std::shared_ptr<ExecutionPlan> input);
    const char *name() const override { return "GridExec"; }
    std::vector<Field> schema() const override { return schema_; }
    std::vector<std::shared_ptr<ExecutionPlan>> children() const override { return {input_}; }
    // Err(Plan("Exactly one child must be provided")) unless children.size() == 1 (:142-160).
    std::shared_ptr<GridExec> with_new_children(std::vector<std::shared_ptr<ExecutionPlan>> children) const;
    std::shared_ptr<ExecutionPlan> with_new_children_dyn(std::vector<std::shared_ptr<ExecutionPlan>> children) const override {
        return with_new_children(std::move(children));
    }
    ExprPtr maybe_predicate() const { return maybe_predicate_; }
    // execute(partition, task_context): batch_size comes from the session config (:165-182).
    std::unique_ptr<GridStream> execute(size_t partition, size_t batch_size);
    std::unique_ptr<SegmentStream> execute_segments(size_t) override {
        throw Error("GridExec produces data points, not segments.");
    }
    std::unique_ptr<SegmentStream> execute_stream(size_t partition, size_t batch_size) override;
    // A copy of this plan whose stream emits only the `value` column (same metrics object). Used by
    // SortedJoinExec for every input but the first: only their values are read
    // (sorted_join_exec.rs:268-275), so their timestamps and tags are never materialised.
    std::shared_ptr<GridExec> with_values_only() const;
    bool values_only() const { return values_only_; }
    std::vector<Distribution> required_input_distribution() const { return {Distribution::SinglePartition}; }
    std::string fmt_as() const; // "GridExec: limit=Some(5)" (:203-209)
    std::shared_ptr<GridStreamMetrics> metrics() const { return metrics_; }
    std::optional<size_t> limit() const { return limit_; }

  private:
    mdb_ctx *ctx_ = nullptr;
    std::vector<Field> schema_;
    ExprPtr maybe_predicate_;
    std::optional<size_t> limit_;
    std::shared_ptr<ExecutionPlan> input_;
    std::shared_ptr<GridStreamMetrics> metrics_ = std::make_shared<GridStreamMetrics>();
    bool values_only_ = false;
};

class GridStream : public SegmentStream { // grid_exec.rs:213-437
  public:
    GridStream(mdb_ctx *ctx, std::vector<Field> schema, ExprPtr maybe_predicate, std::optional<size_t> limit, std::unique_ptr<SegmentStream> input, size_t batch_size,
               std::shared_ptr<GridStreamMetrics> metrics, bool values_only = false);
    // Stream::poll_next (:402-429)
    PollState poll_next(RecordBatch *out) override;
    std::vector<Field> schema() const { return schema_; }
    size_t batch_size() const { return batch_size_; }

    ~GridStream();

  private:
    // One submit to the library on its way through the GPU (mdb_grid_submit): the input batches it was made from
    // (one, or several that the input had ready - they become one launch) and the ticket to wait on. The library
    // reconstructs it on a worker thread and on one of two contexts, so that the copy of one submit's data points
    // to the host overlaps the upload and the kernels of the next (the reference grids a batch when it is polled
    // for, grid_exec.rs:402-412; the stream returns the same rows in the same order). This is, call for call,
    // what rust/patches/0001-grid_exec.patch does with rust/modelardb_hip's GridTicket.
    static constexpr size_t GRID_STREAM_SUBMITS_AHEAD = 1;
    struct Ticket {
        std::vector<std::shared_ptr<RecordBatch>> batches;
        mdb_grid_ticket *raw = nullptr;
    };
    // Polls the input while it has batches ready and the submit is below the stream's target size, and submits
    // what it got: ReadySome with a ticket, ReadyNone when the input has ended, Pending when it has nothing yet.
    PollState poll_input_and_submit(std::optional<Ticket> *out);
    void wait_and_append_to_leftovers_in_current_batch(Ticket ticket); // :261-391
    mdb_ctx *ctx_;
    std::deque<Ticket> ahead_;     // submitted and not yet waited for, oldest first
    bool input_finished_ = false;
    size_t prefetch_depth_ = GRID_STREAM_SUBMITS_AHEAD; // (MDB_HOST_GRID_PREFETCH: 0 = submit when polled, n = that many ahead)
    // Segments per submit: the input hands over 8 192-row batches whatever they decompress to; the stream asks
    // for about 16 M data points per launch and learns the points per segment from the results it has seen.
    uint64_t seen_segments_ = 0, seen_points_ = 0;
    size_t coalesce_segments_ = 0; // (MDB_HOST_GRID_COALESCE_SEGMENTS: a fixed number, 0: learned)
    std::vector<Field> schema_;
    ExprPtr maybe_predicate_;
    // What maybe_predicate_ says about the timestamps (time_range_of_predicate, worked out once when the stream is
    // made): handed to every mdb_grid_submit, so points outside it are neither reconstructed nor copied; if it is
    // not the whole predicate the predicate is evaluated on what comes back (grid_exec.rs:366-387).
    std::optional<PredicateRange> pushed_range_;
    std::unique_ptr<SegmentStream> input_;
    size_t batch_size_;
    RecordBatch current_batch_;
    int64_t current_batch_offset_ = 0;
    std::shared_ptr<GridStreamMetrics> metrics_;
    bool values_only_ = false;
};

// ---- SortedJoinExec (crates/modelardb_storage/src/query/sorted_join_exec.rs) ------------------------------------

// Order of the columns SortedJoinStream returns (sorted_join_exec.rs:46-52).
struct SortedJoinColumnType {
    enum class Kind { Timestamp, Field, Tag } kind;
    std::string tag_name; // Kind::Tag only
    static SortedJoinColumnType timestamp() { return {Kind::Timestamp, {}}; }
    static SortedJoinColumnType field() { return {Kind::Field, {}}; }
    static SortedJoinColumnType tag(std::string name) { return {Kind::Tag, std::move(name)}; }
};

class SortedJoinStream;

// Joins the sorted data points of one GridExec per field column into (timestamp, fields..., tags...)
// rows. Timestamps and tags are taken from the first input only, so every later GridExec input is
// switched to values-only: 4 instead of 12 bytes per point and field cross PCIe.
class SortedJoinExec : public ExecutionPlan {
  public:
    // SortedJoinExec::new (sorted_join_exec.rs:75-101)
    static std::shared_ptr<SortedJoinExec> make(std::vector<Field> schema,
                                                std::vector<SortedJoinColumnType> return_order,
                                                std::vector<std::shared_ptr<ExecutionPlan>> inputs);
    const char *name() const override { return "SortedJoinExec"; }
    std::vector<Field> schema() const override { return schema_; }
    std::vector<std::shared_ptr<ExecutionPlan>> children() const override { return inputs_; }
    // Err(Plan("At least one child must be provided ...")) if children is empty (:131-147).
    std::shared_ptr<SortedJoinExec> with_new_children(std::vector<std::shared_ptr<ExecutionPlan>> children) const;
    std::shared_ptr<ExecutionPlan> with_new_children_dyn(std::vector<std::shared_ptr<ExecutionPlan>> children) const override {
        return with_new_children(std::move(children));
    }
    std::unique_ptr<SortedJoinStream> execute(size_t partition, size_t batch_size); // :151-168
    std::unique_ptr<SegmentStream> execute_segments(size_t) override {
        throw Error("SortedJoinExec produces data points, not segments.");
    }
    std::unique_ptr<SegmentStream> execute_stream(size_t partition, size_t batch_size) override;
    std::vector<Distribution> required_input_distribution() const { // :173-175
        return std::vector<Distribution>(inputs_.size(), Distribution::SinglePartition);
    }
    std::string fmt_as() const { return name(); } // :192-198
    uint64_t output_rows() const { return *output_rows_; }

  private:
    std::vector<Field> schema_;
    std::vector<SortedJoinColumnType> return_order_;
    std::vector<std::shared_ptr<ExecutionPlan>> inputs_;
    std::shared_ptr<uint64_t> output_rows_ = std::make_shared<uint64_t>(0); // BaselineMetrics
};

class SortedJoinStream : public SegmentStream { // sorted_join_exec.rs:200-336
  public:
    SortedJoinStream(std::vector<Field> schema, std::vector<SortedJoinColumnType> return_order,
                     std::vector<std::unique_ptr<SegmentStream>> inputs, std::shared_ptr<uint64_t> output_rows);
    PollState poll_next(RecordBatch *out) override; // :306-327

  private:
    std::optional<PollState> poll_all_pending_inputs(); // :228-246
    void set_batch_num_rows_to_smallest();              // :251-272
    RecordBatch sorted_join() const;                    // :277-311
    std::vector<Field> schema_;
    std::vector<SortedJoinColumnType> return_order_;
    std::vector<std::unique_ptr<SegmentStream>> inputs_;
    std::vector<std::optional<RecordBatch>> batches_;
    std::vector<std::optional<RecordBatch>> surplus_; // rows beyond the smallest batch, if they are carried over
    bool carry_over_ = false; // MDB_HOST_SORTED_JOIN_CARRY_OVER=1 (default: dropped, as sorted_join_exec.rs:248-272 does)
    std::shared_ptr<uint64_t> output_rows_;
};

// ---- accumulators (model_simple_aggregates.rs:336-618) -------------------------------------------------------

// ScalarValue of the accumulator state handed to DataFusion's Final aggregate.
struct ScalarValue {
    enum class Kind { Int64, UInt64, Float32, Float64 } kind;
    int64_t i64 = 0;
    uint64_t u64 = 0;
    float f32 = 0.0f;
    double f64 = 0.0;
    bool null = false; // ScalarValue::Float32(None), ...
};

class Accumulator {
  public:
    virtual ~Accumulator() = default;
    // `arrays` are the columns of a segment batch in QUERY_COMPRESSED_SCHEMA order (at least 0..=7).
    virtual void update_batch(const std::vector<ColumnPtr> &arrays) = 0;
    virtual std::vector<ScalarValue> state() = 0; // also resets, like the reference
    virtual size_t size() const = 0;
    // The model-based accumulators only ever run in the Partial aggregate (unreachable!() in the reference,
    // model_simple_aggregates.rs:374-384); DataFusion's own accumulators (query plans below) merge and evaluate.
    virtual void merge_batch(const std::vector<ScalarValue> &) { throw std::logic_error("unreachable"); }
    virtual ScalarValue evaluate() { throw std::logic_error("unreachable"); }
};

// `range`: the accumulators of a query the patched rule rewrote although it has a range on the timestamp (SURVEY
// 8(f) N1): only the data points inside it count, folded by mdb_agg_batch_range_list; MIN, MAX and SUM of a range
// without a data point are NULL, as DataFusion's are over the empty input the reference's plan would give them.
std::unique_ptr<Accumulator> make_model_count_accumulator(mdb_ctx *ctx, std::optional<TimeRange> range = std::nullopt);
std::unique_ptr<Accumulator> make_model_min_accumulator(mdb_ctx *ctx, std::optional<TimeRange> range = std::nullopt);
std::unique_ptr<Accumulator> make_model_max_accumulator(mdb_ctx *ctx, std::optional<TimeRange> range = std::nullopt);
std::unique_ptr<Accumulator> make_model_sum_accumulator(mdb_ctx *ctx, std::optional<TimeRange> range = std::nullopt);
std::unique_ptr<Accumulator> make_model_avg_accumulator(mdb_ctx *ctx, std::optional<TimeRange> range = std::nullopt);

// ---- query plans: what stands around GridExec in a plan, as far as the ModelSimpleAggregates rule looks ------------

// DataSourceExec over the Parquet files of one field column (new_data_source_exec, query/time_series_table.rs:412-451)
// with the batches handed over by the caller instead of read from files. `filter`: the ParquetSource's predicate
// (pushdown_filters = true: applied row by row to the segments).
class DataSourceExec : public ExecutionPlan {
  public:
    struct Source { // the "files" of one field column: shared by every plan that scans it
        std::vector<RecordBatch> batches;
    };
    DataSourceExec(std::vector<Field> schema, std::shared_ptr<Source> source, ExprPtr filter, std::optional<size_t> limit)
        : schema_(std::move(schema)), source_(std::move(source)), filter_(std::move(filter)), limit_(limit) {}
    const char *name() const override { return "DataSourceExec"; }
    std::vector<Field> schema() const override { return schema_; }
    std::vector<std::shared_ptr<ExecutionPlan>> children() const override { return {}; }
    std::unique_ptr<SegmentStream> execute_segments(size_t partition) override;
    ExprPtr filter() const { return filter_; } // parquet_source.filter()

  private:
    std::vector<Field> schema_;
    std::shared_ptr<Source> source_;
    ExprPtr filter_;
    std::optional<size_t> limit_;
};

// RepartitionExec / CoalescePartitionsExec / CoalesceBatchesExec: one partition here, so they pass batches through;
// they are in the plans because the rule has to look past them.
class PassThroughExec : public ExecutionPlan {
  public:
    PassThroughExec(const char *name, std::shared_ptr<ExecutionPlan> input) : name_(name), input_(std::move(input)) {}
    const char *name() const override { return name_; }
    std::vector<Field> schema() const override { return input_->schema(); }
    std::vector<std::shared_ptr<ExecutionPlan>> children() const override { return {input_}; }
    std::unique_ptr<SegmentStream> execute_segments(size_t partition) override { return input_->execute_segments(partition); }
    std::unique_ptr<SegmentStream> execute_stream(size_t partition, size_t batch_size) override {
        return input_->execute_stream(partition, batch_size);
    }
    std::shared_ptr<ExecutionPlan> with_new_children_dyn(std::vector<std::shared_ptr<ExecutionPlan>> children) const override {
        if (children.size() != 1) throw Error(std::string(name_) + " needs exactly one child.");
        return std::make_shared<PassThroughExec>(name_, children[0]);
    }

  private:
    const char *name_;
    std::shared_ptr<ExecutionPlan> input_;
};

// FilterExec: predicate over the input's schema, then an optional projection (indices into the input's schema).
class FilterExec : public ExecutionPlan {
  public:
    FilterExec(ExprPtr predicate, std::shared_ptr<ExecutionPlan> input, std::optional<std::vector<size_t>> projection)
        : predicate_(std::move(predicate)), input_(std::move(input)), projection_(std::move(projection)) {}
    const char *name() const override { return "FilterExec"; }
    std::vector<Field> schema() const override;
    std::vector<std::shared_ptr<ExecutionPlan>> children() const override { return {input_}; }
    std::unique_ptr<SegmentStream> execute_segments(size_t) override { throw Error("FilterExec produces data points."); }
    std::unique_ptr<SegmentStream> execute_stream(size_t partition, size_t batch_size) override;
    std::shared_ptr<ExecutionPlan> with_new_children_dyn(std::vector<std::shared_ptr<ExecutionPlan>> children) const override {
        if (children.size() != 1) throw Error("FilterExec needs exactly one child.");
        return std::make_shared<FilterExec>(predicate_, children[0], projection_);
    }
    ExprPtr predicate() const { return predicate_; }
    const std::optional<std::vector<size_t>> &projection() const { return projection_; }

  private:
    ExprPtr predicate_;
    std::shared_ptr<ExecutionPlan> input_;
    std::optional<std::vector<size_t>> projection_;
};

// One aggregate of an AggregateExec: name() is "count(field_1)" for DataFusion's own functions (the rule takes the
// function from the text before the parenthesis, model_simple_aggregates.rs:311-331) and "model_count" etc. for the
// model-based ones; `column` is the input column a DataFusion function reads (the model-based ones read the segment
// columns); `create` makes its accumulator.
struct AggregateFunctionExpr {
    std::string name;
    size_t column = 0;
    std::function<std::unique_ptr<Accumulator>()> create;
    std::optional<TimeRange> range; // model-based under a time range (what fmt shows)
};
AggregateFunctionExpr datafusion_aggregate(const std::string &function, const std::string &column_name, size_t column);
AggregateFunctionExpr model_aggregate(mdb_ctx *ctx, const std::string &function, std::optional<TimeRange> range);

enum class AggregateMode { Partial, Final };

class AggregateExec : public ExecutionPlan {
  public:
    AggregateExec(AggregateMode mode, std::vector<AggregateFunctionExpr> aggr_expr, std::shared_ptr<ExecutionPlan> input,
                  std::vector<Field> input_schema)
        : mode_(mode), aggr_expr_(std::move(aggr_expr)), input_(std::move(input)), input_schema_(std::move(input_schema)) {}
    const char *name() const override { return "AggregateExec"; }
    std::vector<Field> schema() const override; // Partial: the state fields; Final: one column per aggregate
    std::vector<std::shared_ptr<ExecutionPlan>> children() const override { return {input_}; }
    std::unique_ptr<SegmentStream> execute_segments(size_t) override { throw Error("AggregateExec produces aggregates."); }
    std::unique_ptr<SegmentStream> execute_stream(size_t partition, size_t batch_size) override;
    std::shared_ptr<ExecutionPlan> with_new_children_dyn(std::vector<std::shared_ptr<ExecutionPlan>> children) const override {
        if (children.size() != 1) throw Error("AggregateExec needs exactly one child.");
        return std::make_shared<AggregateExec>(mode_, aggr_expr_, children[0], input_schema_);
    }
    AggregateMode mode() const { return mode_; }
    const std::vector<AggregateFunctionExpr> &aggr_expr() const { return aggr_expr_; }
    const std::vector<Field> &input_schema() const { return input_schema_; } // of the ORIGINAL input (:253-255)
    // The results of a Final aggregate polled to its end: one ScalarValue per aggregate.
    static std::vector<ScalarValue> collect(ExecutionPlan &final_aggregate, size_t batch_size);

  private:
    AggregateMode mode_;
    std::vector<AggregateFunctionExpr> aggr_expr_;
    std::shared_ptr<ExecutionPlan> input_;
    std::vector<Field> input_schema_;
};

// TimeSeriesTable as a TableProvider, as far as scan() goes (query/time_series_table.rs:494-671): a timestamp
// column, n field columns (f32) and tag columns; per field column the segments the Delta table holds for it.
class TimeSeriesTable {
  public:
    TimeSeriesTable(mdb_ctx *ctx, size_t n_fields, std::vector<std::string> tag_names);
    std::vector<Field> query_schema() const { return query_schema_; } // timestamp, field_1.., tags..
    void push_segments(size_t field, RecordBatch batch);              // QUERY_COMPRESSED_SCHEMA + tags
    // scan(state, projection, filters, limit): DataSourceExec -> GridExec per stored field column in the projection
    // (the first field column if none), zipped by a SortedJoinExec; the filters rewritten and pushed both ways.
    std::shared_ptr<ExecutionPlan> scan(const std::vector<size_t> &projection, const std::vector<ExprPtr> &filters,
                                        std::optional<size_t> limit) const;

  private:
    mdb_ctx *ctx_;
    std::vector<std::string> tag_names_;
    std::vector<Field> query_schema_;
    std::vector<std::shared_ptr<DataSourceExec::Source>> sources_;
};

// The physical plan DataFusion makes of SELECT agg(field), ... FROM table [WHERE filters] (the shapes the reference's
// tests assert, model_simple_aggregates.rs:637-719): AggregateExec(Final) <- CoalescePartitionsExec <-
// AggregateExec(Partial) <- [FilterExec <-] RepartitionExec <- scan(). `aggregates`: (function, field column index).
std::shared_ptr<ExecutionPlan> plan_aggregate_query(const TimeSeriesTable &table,
                                                    const std::vector<std::pair<std::string, size_t>> &aggregates,
                                                    const std::vector<ExprPtr> &filters);

// ModelSimpleAggregates (optimizer/model_simple_aggregates.rs:176-302) with the extension of SURVEY 8(f) N1, as
// rust/patches/0002 makes it: besides AggregateExec <- [RepartitionExec] <- SortedJoinExec <- GridExec <-
// DataSourceExec(no filter) it accepts a FilterExec between the aggregate and the join whose predicate IS a time range
// (time_range_of_predicate: exact) over a DataSourceExec whose filter only reads start_time / end_time, and gives
// the range to the model-based accumulators it puts in.
struct ModelSimpleAggregates {
    mdb_ctx *ctx;
    std::shared_ptr<ExecutionPlan> optimize(std::shared_ptr<ExecutionPlan> execution_plan) const;
    const char *name() const { return "model_simple_aggregates"; }
    bool schema_check() const { return true; }
};
// The plan level by level, names separated by commas within a level (assert_eq_physical_plan_expected, :765-790).
std::string plan_levels(const std::shared_ptr<ExecutionPlan> &plan);

// ---- compression (compression.rs:42-275) -----------------------------------------------------------------------

RecordBatch try_compress_univariate_time_series(mdb_ctx *ctx, const Column &uncompressed_timestamps,
                                                const Column &uncompressed_values, ErrorBound error_bound,
                                                const std::vector<Field> &compressed_schema,
                                                const std::vector<std::string> &tag_values,
                                                int16_t field_column_index);

struct TimeSeriesTableMetadata { // the fields of types.rs:76-239 that compression reads
    size_t timestamp_column_index = 0;
    std::vector<size_t> field_column_indices;
    std::vector<size_t> tag_column_indices;
    std::vector<ErrorBound> error_bounds; // indexed by column index, like the reference
    std::vector<Field> compressed_schema;
};

// Sorts by (tags..., timestamp), splits into series and compresses every field column of every
// series - all series x fields of the batch go to the GPU as ONE mdb_compress_chunks call. Returns
// one RecordBatch per series x field in the reference's order.
std::vector<RecordBatch> try_compress_multivariate_time_series(mdb_ctx *ctx,
                                                               const TimeSeriesTableMetadata &metadata,
                                                               const RecordBatch &uncompressed_time_series);

// ---- ingest-side batching ---------------------------------------------------------------------

// The compress side of UncompressedDataManager (crates/modelardb_server/src/storage/
// uncompressed_data_manager.rs:130-189, 197-322, 405-451, 530-596) with one change: finished buffers
// are not compressed one at a time on one thread but collected and handed to the GPU together, all
// series x fields as the chunks of ONE mdb_compress_chunks launch per error bound. Memory pool,
// spilling to Parquet, WAL batch ids and channels are out of scope (durability / host orchestration).
class UncompressedDataManager {
  public:
    // UNCOMPRESSED_DATA_BUFFER_CAPACITY = 64 * 1024 (storage/mod.rs:58).
    UncompressedDataManager(mdb_ctx *ctx, TimeSeriesTableMetadata metadata,
                            size_t buffer_capacity = MDB_UNCOMPRESSED_DATA_BUFFER_CAPACITY);
    // One ingested batch: every row goes to the buffer of its tag values; a buffer that becomes full
    // is finished; afterwards buffers not touched by this batch are finished
    // (RECORD_BATCH_OFFSET_REQUIRED_FOR_UNUSED = 1, storage/uncompressed_data_buffer.rs:42,135-137).
    void insert_data_points(const RecordBatch &data_points);
    void flush(); // StorageEngine::flush: finish every active buffer
    size_t active_buffer_count() const { return active_.size(); }
    size_t finished_buffer_count() const { return finished_.size(); }
    // Compress every finished buffer (sorted by time first, uncompressed_data_buffer.rs:175-209).
    // Returns, per finished buffer in finish order, one RecordBatch per field column.
    std::vector<RecordBatch> compress_finished_buffers();

  private:
    struct Buffer {
        std::vector<std::string> tag_values;
        std::vector<int64_t> timestamps;
        std::vector<std::vector<float>> values; // per field
        uint64_t updated_by_batch_index = 0;
    };
    void finish_unused_buffers(uint64_t current_batch_index);
    mdb_ctx *ctx_;
    TimeSeriesTableMetadata metadata_;
    size_t capacity_;
    uint64_t current_batch_index_ = 0;
    std::vector<std::pair<std::string, Buffer>> active_; // keyed by the joined tag values
    std::vector<Buffer> finished_;
};

} // namespace mdbhost
