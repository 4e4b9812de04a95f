// mdb_host_query.cpp - the query side of the host operators: the predicate expressions a query carries, the filter
// rewrite of TimeSeriesTable::scan (crates/modelardb_storage/src/query/time_series_table.rs:269-373, 494-671), the
// plan nodes that stand around GridExec in the plans the reference's tests assert, and the ModelSimpleAggregates rule
// (crates/modelardb_storage/src/optimizer/model_simple_aggregates.rs:176-302) with the extension of SURVEY 8(f) N1:
// aggregates under a range on the timestamp computed from the segments (mdb_agg_batch_range_list) instead of from
// reconstructed data points. The C++ twin of what rust/patches/0001 (time_range_of_predicate) and 0002 (the rule,
// the accumulators' range) add on the Rust side. No model arithmetic happens here.
#include "mdb_host.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <limits>
#include <sstream>

namespace mdbhost {

// ---- expressions ---------------------------------------------------------------------------------------------------

const char *operator_text(Operator op) {
    switch (op) {
    case Operator::Lt: return "<";
    case Operator::LtEq: return "<=";
    case Operator::Gt: return ">";
    case Operator::GtEq: return ">=";
    case Operator::Eq: return "=";
    case Operator::NotEq: return "!=";
    case Operator::And: return "AND";
    case Operator::Or: return "OR";
    }
    return "?";
}

ExprPtr Expr::col(std::string name) {
    auto expr = std::make_shared<Expr>();
    expr->kind = Kind::Column;
    expr->column = std::move(name);
    return expr;
}

ExprPtr Expr::lit(Scalar value) {
    auto expr = std::make_shared<Expr>();
    expr->kind = Kind::Literal;
    expr->literal = std::move(value);
    return expr;
}

ExprPtr Expr::lit_timestamp(int64_t microseconds) {
    Scalar value;
    value.kind = Scalar::Kind::TimestampMicrosecond;
    value.i64 = microseconds;
    return lit(value);
}

ExprPtr Expr::binary(ExprPtr left, Operator op, ExprPtr right) {
    auto expr = std::make_shared<Expr>();
    expr->kind = Kind::BinaryExpr;
    expr->op = op;
    expr->left = std::move(left);
    expr->right = std::move(right);
    return expr;
}

std::string Expr::to_string() const {
    switch (kind) {
    case Kind::Column: return column;
    case Kind::Literal:
        switch (literal.kind) {
        case Scalar::Kind::TimestampMicrosecond: return "TimestampMicrosecond(" + std::to_string(literal.i64) + ")";
        case Scalar::Kind::Int64: return "Int64(" + std::to_string(literal.i64) + ")";
        case Scalar::Kind::Float32: {
            std::ostringstream text;
            text << "Float32(" << literal.f32 << ")";
            return text.str();
        }
        case Scalar::Kind::Utf8: return "Utf8(\"" + literal.utf8 + "\")";
        }
        return "?";
    case Kind::BinaryExpr: {
        const bool nested = op == Operator::And || op == Operator::Or;
        auto side = [&](const ExprPtr &e) {
            const bool wrap = nested && e->kind == Kind::BinaryExpr && (e->op == Operator::And || e->op == Operator::Or) && e->op != op;
            return wrap ? "(" + e->to_string() + ")" : e->to_string();
        };
        return side(left) + " " + operator_text(op) + " " + side(right);
    }
    }
    return "?";
}

namespace {

struct Parser {
    const std::string &text;
    size_t at = 0;
    void skip() {
        while (at < text.size() && std::isspace(static_cast<unsigned char>(text[at]))) at++;
    }
    std::string token() {
        skip();
        const size_t first = at;
        while (at < text.size() && !std::isspace(static_cast<unsigned char>(text[at])) && text[at] != '(' && text[at] != ')') at++;
        if (first == at) throw Error("Malformed expression: a token was expected at offset " + std::to_string(first) + ".");
        return text.substr(first, at - first);
    }
    static Operator op_of(const std::string &word) {
        if (word == "<") return Operator::Lt;
        if (word == "<=") return Operator::LtEq;
        if (word == ">") return Operator::Gt;
        if (word == ">=") return Operator::GtEq;
        if (word == "=") return Operator::Eq;
        if (word == "!=") return Operator::NotEq;
        if (word == "and") return Operator::And;
        if (word == "or") return Operator::Or;
        throw Error("Malformed expression: unknown operator " + word + ".");
    }
    ExprPtr expr() {
        skip();
        if (at < text.size() && text[at] == '(') {
            at++;
            const Operator op = op_of(token());
            ExprPtr left = expr(), right = expr();
            skip();
            if (at >= text.size() || text[at] != ')') throw Error("Malformed expression: ')' was expected.");
            at++;
            return Expr::binary(left, op, right);
        }
        const std::string word = token();
        const size_t colon = word.find(':');
        if (colon == std::string::npos) return Expr::col(word);
        const std::string kind = word.substr(0, colon), value = word.substr(colon + 1);
        Scalar scalar;
        try {
            if (kind == "ts") {
                scalar.kind = Scalar::Kind::TimestampMicrosecond;
                scalar.i64 = std::stoll(value);
            } else if (kind == "i64") {
                scalar.kind = Scalar::Kind::Int64;
                scalar.i64 = std::stoll(value);
            } else if (kind == "f32") {
                scalar.kind = Scalar::Kind::Float32;
                scalar.f32 = std::stof(value);
            } else if (kind == "str") {
                scalar.kind = Scalar::Kind::Utf8;
                scalar.utf8 = value;
            } else {
                throw Error("Malformed expression: unknown literal kind " + kind + ".");
            }
        } catch (const std::logic_error &) { // std::stoll / std::stof
            throw Error("Malformed expression: " + word + " is not a literal.");
        }
        return Expr::lit(scalar);
    }
};

} // namespace

ExprPtr parse_expr(const std::string &text) {
    Parser parser{text};
    ExprPtr expr = parser.expr();
    parser.skip();
    if (parser.at != text.size()) throw Error("Malformed expression: text after its end.");
    return expr;
}

ExprPtr conjunction(const std::vector<ExprPtr> &exprs) {
    ExprPtr all;
    for (const ExprPtr &expr : exprs) all = all ? Expr::binary(all, Operator::And, expr) : expr;
    return all;
}

ExprPtr timestamp_range_predicate(std::optional<int64_t> lower, std::optional<int64_t> upper) {
    std::vector<ExprPtr> parts;
    if (lower) parts.push_back(Expr::binary(Expr::col("timestamp"), Operator::GtEq, Expr::lit_timestamp(*lower)));
    if (upper) parts.push_back(Expr::binary(Expr::col("timestamp"), Operator::LtEq, Expr::lit_timestamp(*upper)));
    return conjunction(parts);
}

namespace {

template <typename T> bool compare(Operator op, T a, T b) {
    switch (op) {
    case Operator::Lt: return a < b;
    case Operator::LtEq: return a <= b;
    case Operator::Gt: return a > b;
    case Operator::GtEq: return a >= b;
    case Operator::Eq: return a == b;
    case Operator::NotEq: return a != b;
    default: return false;
    }
}

// The comparison with its sides swapped: a op b  <=>  b flip(op) a.
Operator flip(Operator op) {
    switch (op) {
    case Operator::Lt: return Operator::Gt;
    case Operator::LtEq: return Operator::GtEq;
    case Operator::Gt: return Operator::Lt;
    case Operator::GtEq: return Operator::LtEq;
    default: return op;
    }
}

const Column &column_named(const RecordBatch &batch, const std::string &name) {
    for (size_t c = 0; c < batch.schema.size(); c++)
        if (batch.schema[c].name == name) return *batch.columns[c];
    throw Error("The columns in the filter should exist in the schema: " + name + ".");
}

} // namespace

std::vector<uint8_t> evaluate_predicate(const Expr &predicate, const RecordBatch &batch) {
    const size_t rows = static_cast<size_t>(batch.num_rows);
    if (predicate.kind != Expr::Kind::BinaryExpr) throw Error("A predicate must be a comparison or AND / OR of comparisons.");
    if (predicate.op == Operator::And || predicate.op == Operator::Or) {
        std::vector<uint8_t> left = evaluate_predicate(*predicate.left, batch);
        const std::vector<uint8_t> right = evaluate_predicate(*predicate.right, batch);
        for (size_t i = 0; i < rows; i++)
            left[i] = predicate.op == Operator::And ? (left[i] & right[i]) : (left[i] | right[i]);
        return left;
    }
    // column op literal, or literal op column
    const Expr *column = predicate.left.get(), *literal = predicate.right.get();
    Operator op = predicate.op;
    if (column->kind != Expr::Kind::Column) {
        std::swap(column, literal);
        op = flip(op);
    }
    if (column->kind != Expr::Kind::Column || literal->kind != Expr::Kind::Literal)
        throw Error("A comparison must be between a column and a literal.");
    const Column &values = column_named(batch, column->column);
    const Scalar &scalar = literal->literal;
    std::vector<uint8_t> mask(rows);
    switch (values.type) {
    case Type::Timestamp:
    case Type::Int64:
        if (scalar.kind != Scalar::Kind::TimestampMicrosecond && scalar.kind != Scalar::Kind::Int64)
            throw Error("A timestamp column can only be compared with a timestamp.");
        for (size_t i = 0; i < rows; i++) mask[i] = compare<int64_t>(op, values.as<int64_t>()[i], scalar.i64);
        break;
    case Type::Float32:
        if (scalar.kind != Scalar::Kind::Float32) throw Error("A value column can only be compared with a Float32.");
        for (size_t i = 0; i < rows; i++) mask[i] = compare<float>(op, values.as<float>()[i], scalar.f32);
        break;
    case Type::Utf8View:
        if (scalar.kind != Scalar::Kind::Utf8) throw Error("A tag column can only be compared with a string.");
        for (size_t i = 0; i < rows; i++)
            mask[i] = compare<std::string_view>(op, values.view_value(static_cast<int64_t>(i)), std::string_view(scalar.utf8));
        break;
    default: throw Error("Columns of this type cannot be filtered.");
    }
    return mask;
}

RecordBatch filter_record_batch(const RecordBatch &batch, const std::vector<uint8_t> &mask) {
    if (mask.size() != static_cast<size_t>(batch.num_rows)) throw Error("The filter's mask does not match the batch.");
    size_t kept = 0;
    for (uint8_t bit : mask) kept += bit != 0;
    if (kept == mask.size()) return batch; // (IterationStrategy::All: the columns themselves)
    RecordBatch out;
    out.schema = batch.schema;
    out.num_rows = static_cast<int64_t>(kept);
    for (const ColumnPtr &parent : batch.columns) {
        size_t width = 0;
        switch (parent->type) {
        case Type::Int8: width = 1; break;
        case Type::Int16: width = 2; break;
        case Type::Float32: width = 4; break;
        case Type::BinaryView:
        case Type::Utf8View: width = 16; break;
        default: width = 8;
        }
        auto column = std::make_shared<Column>();
        column->type = parent->type;
        column->length = static_cast<int64_t>(kept);
        column->data.resize(std::max<size_t>(kept * width, 16));
        const uint8_t *from = static_cast<const uint8_t *>(parent->values);
        uint8_t *to = column->data.data();
        for (size_t i = 0; i < mask.size(); i++) {
            if (!mask[i]) continue;
            std::memcpy(to, from + i * width, width);
            to += width;
        }
        column->values = column->data.data();
        column->buffer_ptrs = parent->buffer_ptrs; // (views keep pointing into the parent's data buffers)
        column->buffer_sizes = parent->buffer_sizes;
        column->keep_alive = parent;
        out.columns.push_back(column);
    }
    return out;
}

namespace {

// Narrow `range` by `timestamp op value`; false if op is not one of the five comparisons.
bool narrow(TimeRange *range, Operator op, int64_t value) {
    switch (op) {
    case Operator::GtEq: range->lo = std::max(range->lo, value); return true;
    case Operator::Gt: // timestamp > i64::MAX holds for nothing
        if (value == INT64_MAX) *range = TimeRange{INT64_MAX, INT64_MIN};
        else range->lo = std::max(range->lo, value + 1);
        return true;
    case Operator::LtEq: range->hi = std::min(range->hi, value); return true;
    case Operator::Lt:
        if (value == INT64_MIN) *range = TimeRange{INT64_MAX, INT64_MIN};
        else range->hi = std::min(range->hi, value - 1);
        return true;
    case Operator::Eq:
        range->lo = std::max(range->lo, value);
        range->hi = std::min(range->hi, value);
        return true;
    default: return false;
    }
}

// One conjunct at a time: true if it was a comparison of the timestamp column with a timestamp literal.
void range_of_conjuncts(const Expr &expr, const std::string &timestamp_column, TimeRange *range, bool *narrowed, bool *exact) {
    if (expr.kind == Expr::Kind::BinaryExpr && expr.op == Operator::And) {
        range_of_conjuncts(*expr.left, timestamp_column, range, narrowed, exact);
        range_of_conjuncts(*expr.right, timestamp_column, range, narrowed, exact);
        return;
    }
    if (expr.kind == Expr::Kind::BinaryExpr) {
        const Expr *column = expr.left.get(), *literal = expr.right.get();
        Operator op = expr.op;
        if (column->kind != Expr::Kind::Column) {
            std::swap(column, literal);
            op = flip(op);
        }
        if (column->kind == Expr::Kind::Column && column->column == timestamp_column && literal->kind == Expr::Kind::Literal &&
            literal->literal.kind == Scalar::Kind::TimestampMicrosecond && narrow(range, op, literal->literal.i64)) {
            *narrowed = true;
            return;
        }
    }
    *exact = false; // an OR, a !=, another column, another kind of literal: the filter decides
}

} // namespace

std::optional<PredicateRange> time_range_of_predicate(const Expr &predicate, const std::string &timestamp_column) {
    PredicateRange found;
    bool narrowed = false;
    range_of_conjuncts(predicate, timestamp_column, &found.range, &narrowed, &found.exact);
    if (!narrowed) return std::nullopt;
    if (found.range.empty()) found.range = TimeRange{INT64_MAX, INT64_MIN}; // (one spelling of "nothing")
    return found;
}

// ---- rewrite_filter (query/time_series_table.rs:290-373) -----------------------------------------------------------------

std::optional<std::pair<ExprPtr, ExprPtr>> rewrite_filter(const std::vector<Field> &query_schema, const Expr &filter) {
    if (filter.kind != Expr::Kind::BinaryExpr) return std::nullopt;
    auto type_of_column = [&](const std::string &name) {
        for (const Field &field : query_schema)
            if (field.name == name) return field.type;
        throw Error("The columns in the filter should exist in the query schema.");
    };
    const ExprPtr &left = filter.left, &right = filter.right;
    const Operator op = filter.op;
    if (left->kind == Expr::Kind::Column) {
        if (type_of_column(left->column) != Type::Timestamp) return std::nullopt;
        switch (op) {
        case Operator::Gt:
        case Operator::GtEq:
            return std::make_pair(Expr::binary(Expr::col("end_time"), op, right), Expr::binary(Expr::col("timestamp"), op, right));
        case Operator::Lt:
        case Operator::LtEq:
            return std::make_pair(Expr::binary(Expr::col("start_time"), op, right), Expr::binary(Expr::col("timestamp"), op, right));
        case Operator::Eq:
            return std::make_pair(Expr::binary(Expr::binary(Expr::col("start_time"), Operator::LtEq, right), Operator::And,
                                               Expr::binary(Expr::col("end_time"), Operator::GtEq, right)),
                                  Expr::binary(Expr::col("timestamp"), op, right));
        default: return std::nullopt;
        }
    }
    if (right->kind == Expr::Kind::Column) {
        if (type_of_column(right->column) != Type::Timestamp) return std::nullopt;
        switch (op) {
        case Operator::Gt:
        case Operator::GtEq:
            return std::make_pair(Expr::binary(left, op, Expr::col("start_time")), Expr::binary(left, op, Expr::col("timestamp")));
        case Operator::Lt:
        case Operator::LtEq:
            return std::make_pair(Expr::binary(left, op, Expr::col("end_time")), Expr::binary(left, op, Expr::col("timestamp")));
        case Operator::Eq:
            // (:351-366 as it stands: `left <= start_time AND left >= end_time` - which only a segment of one data
            // point satisfies - and the query's own column in place of the literal on the grid side. Followed, not
            // fixed: DataFusion's simplifier puts the column of a comparison on the left before scan() sees it, so
            // this arm is not reached from SQL.)
            return std::make_pair(Expr::binary(Expr::binary(left, Operator::LtEq, Expr::col("start_time")), Operator::And,
                                               Expr::binary(left, Operator::GtEq, Expr::col("end_time"))),
                                  Expr::binary(right, op, Expr::col("timestamp")));
        default: return std::nullopt;
        }
    }
    return std::nullopt;
}

RewrittenFilters rewrite_and_combine_filters(const std::vector<Field> &query_schema, const std::vector<ExprPtr> &filters) {
    std::vector<ExprPtr> parquet, grid;
    for (const ExprPtr &filter : filters) {
        if (auto rewritten = rewrite_filter(query_schema, *filter)) {
            parquet.push_back(rewritten->first);
            grid.push_back(rewritten->second);
        }
    }
    return {conjunction(parquet), conjunction(grid)};
}

// ---- plan nodes ---------------------------------------------------------------------------------------------------------

namespace {

struct DataSourceStream : SegmentStream {
    std::shared_ptr<DataSourceExec::Source> source;
    ExprPtr filter;
    std::optional<size_t> remaining;
    size_t next = 0;
    PollState poll_next(RecordBatch *out) override {
        while (next < source->batches.size()) {
            if (remaining && *remaining == 0) return PollState::ReadyNone;
            RecordBatch batch = source->batches[next++];
            if (filter) batch = filter_record_batch(batch, evaluate_predicate(*filter, batch));
            if (remaining && static_cast<size_t>(batch.num_rows) > *remaining) batch = batch.slice(0, static_cast<int64_t>(*remaining));
            if (batch.num_rows == 0) continue;
            if (remaining) *remaining -= static_cast<size_t>(batch.num_rows);
            *out = std::move(batch);
            return PollState::ReadySome;
        }
        return PollState::ReadyNone;
    }
};

struct FilterStream : SegmentStream {
    ExprPtr predicate;
    std::optional<std::vector<size_t>> projection;
    std::unique_ptr<SegmentStream> input;
    PollState poll_next(RecordBatch *out) override {
        while (true) {
            RecordBatch batch;
            const PollState state = input->poll_next(&batch);
            if (state != PollState::ReadySome) return state;
            batch = filter_record_batch(batch, evaluate_predicate(*predicate, batch));
            if (batch.num_rows == 0) continue;
            if (projection) {
                RecordBatch projected;
                projected.num_rows = batch.num_rows;
                for (size_t index : *projection) {
                    projected.schema.push_back(batch.schema.at(index));
                    projected.columns.push_back(batch.columns.at(index));
                }
                batch = std::move(projected);
            }
            *out = std::move(batch);
            return PollState::ReadySome;
        }
    }
};

} // namespace

std::unique_ptr<SegmentStream> DataSourceExec::execute_segments(size_t) {
    auto stream = std::make_unique<DataSourceStream>();
    stream->source = source_;
    stream->filter = filter_;
    stream->remaining = limit_;
    return stream;
}

std::vector<Field> FilterExec::schema() const {
    const std::vector<Field> input = input_->schema();
    if (!projection_) return input;
    std::vector<Field> out;
    for (size_t index : *projection_) out.push_back(input.at(index));
    return out;
}

std::unique_ptr<SegmentStream> FilterExec::execute_stream(size_t partition, size_t batch_size) {
    auto stream = std::make_unique<FilterStream>();
    stream->predicate = predicate_;
    stream->projection = projection_;
    stream->input = input_->execute_stream(partition, batch_size);
    return stream;
}

// ---- DataFusion's own count / min / max / sum / avg over one Float32 column, as far as these plans need them -------------

namespace {

struct ValueAccumulator : Accumulator {
    enum class Function { Count, Min, Max, Sum, Avg } function;
    size_t column;
    int64_t count = 0;
    float min = 0.0f, max = 0.0f;
    bool seen = false;
    double sum = 0.0;
    ValueAccumulator(Function f, size_t c) : function(f), column(c) {}
    void update_batch(const std::vector<ColumnPtr> &arrays) override {
        const Column &values = *arrays.at(column);
        if (values.type != Type::Float32) throw Error("Aggregates are computed over a Float32 column.");
        const float *v = values.as<float>();
        for (int64_t i = 0; i < values.length; i++) {
            // (MIN / MAX: DataFusion's min_max over floats orders NaN above everything; the data here has none)
            min = seen ? std::min(min, v[i]) : v[i];
            max = seen ? std::max(max, v[i]) : v[i];
            seen = true;
            sum += static_cast<double>(v[i]);
        }
        count += values.length;
    }
    std::vector<ScalarValue> state() override {
        ScalarValue n{ScalarValue::Kind::Int64}, lo{ScalarValue::Kind::Float32}, hi{ScalarValue::Kind::Float32},
            total{ScalarValue::Kind::Float64}, un{ScalarValue::Kind::UInt64};
        n.i64 = count;
        un.u64 = static_cast<uint64_t>(count);
        lo.f32 = min, hi.f32 = max, total.f64 = sum;
        lo.null = hi.null = total.null = !seen;
        switch (function) {
        case Function::Count: return {n};
        case Function::Min: return {lo};
        case Function::Max: return {hi};
        case Function::Sum: return {total};
        case Function::Avg: return {un, total};
        }
        return {};
    }
    // The Final aggregate: the states of the Partial one, whichever accumulator made them (nulls are skipped).
    void merge_batch(const std::vector<ScalarValue> &states) override {
        switch (function) {
        case Function::Count: count += states.at(0).i64; break;
        case Function::Min:
            if (!states.at(0).null) min = seen ? std::min(min, states[0].f32) : states[0].f32, seen = true;
            break;
        case Function::Max:
            if (!states.at(0).null) max = seen ? std::max(max, states[0].f32) : states[0].f32, seen = true;
            break;
        case Function::Sum:
            if (!states.at(0).null) sum += states[0].f64, seen = true;
            break;
        case Function::Avg:
            count += static_cast<int64_t>(states.at(0).u64);
            if (!states.at(1).null) sum += states[1].f64;
            break;
        }
    }
    ScalarValue evaluate() override {
        ScalarValue out{ScalarValue::Kind::Float64};
        switch (function) {
        case Function::Count:
            out.kind = ScalarValue::Kind::Int64;
            out.i64 = count;
            break;
        case Function::Min:
            out.kind = ScalarValue::Kind::Float32;
            out.f32 = min, out.null = !seen;
            break;
        case Function::Max:
            out.kind = ScalarValue::Kind::Float32;
            out.f32 = max, out.null = !seen;
            break;
        case Function::Sum: out.f64 = sum, out.null = !seen; break;
        case Function::Avg: out.f64 = count ? sum / static_cast<double>(count) : 0.0, out.null = count == 0; break;
        }
        return out;
    }
    size_t size() const override { return sizeof(*this); }
};

ValueAccumulator::Function function_named(const std::string &name) {
    if (name == "count") return ValueAccumulator::Function::Count;
    if (name == "min") return ValueAccumulator::Function::Min;
    if (name == "max") return ValueAccumulator::Function::Max;
    if (name == "sum") return ValueAccumulator::Function::Sum;
    if (name == "avg") return ValueAccumulator::Function::Avg;
    throw Error("Aggregate expression " + name + " is not supported.");
}

std::string function_of(const AggregateFunctionExpr &expr) { // model_simple_aggregates.rs:311-314
    std::string name = expr.name.substr(0, expr.name.find('('));
    if (name.rfind("model_", 0) == 0) name = name.substr(6);
    return name;
}

struct AggregateStream : SegmentStream {
    AggregateMode mode;
    std::vector<AggregateFunctionExpr> aggr_expr;
    std::vector<std::unique_ptr<Accumulator>> accumulators;
    std::unique_ptr<SegmentStream> input;
    std::vector<Field> schema;
    bool done = false;
    PollState poll_next(RecordBatch *out) override {
        if (done) return PollState::ReadyNone;
        while (true) {
            RecordBatch batch;
            const PollState state = input->poll_next(&batch);
            if (state == PollState::Pending) return state;
            if (state == PollState::ReadyNone) break;
            if (mode == AggregateMode::Partial) {
                for (auto &accumulator : accumulators) accumulator->update_batch(batch.columns);
            } else { // one row of states per input batch
                size_t at = 0;
                for (size_t k = 0; k < accumulators.size(); k++) {
                    const size_t n_states = function_of(aggr_expr[k]) == "avg" ? 2 : 1;
                    std::vector<ScalarValue> states;
                    for (size_t s = 0; s < n_states; s++, at++) {
                        const Column &column = *batch.columns.at(at);
                        ScalarValue value{ScalarValue::Kind::Int64};
                        switch (column.type) {
                        case Type::Int64: value.i64 = column.as<int64_t>()[0]; break;
                        case Type::UInt64: value.kind = ScalarValue::Kind::UInt64, value.u64 = column.as<uint64_t>()[0]; break;
                        case Type::Float32: value.kind = ScalarValue::Kind::Float32, value.f32 = column.as<float>()[0]; break;
                        case Type::Float64: value.kind = ScalarValue::Kind::Float64, value.f64 = column.as<double>()[0]; break;
                        default: throw Error("An aggregate state of this type cannot be merged.");
                        }
                        value.null = !column.data.empty() && column.data.size() > 16 && column.data[16] != 0;
                        states.push_back(value);
                    }
                    accumulators[k]->merge_batch(states);
                }
            }
        }
        done = true;
        // One row: the states (Partial) or the results (Final). A null is marked in a byte behind the value
        // (these batches never leave the library: the C surface reports results as value + null flag).
        RecordBatch result;
        result.schema = schema;
        result.num_rows = 1;
        auto column_of = [](const ScalarValue &value) {
            auto column = std::make_shared<Column>();
            column->length = 1;
            column->data.assign(17, 0);
            switch (value.kind) {
            case ScalarValue::Kind::Int64: column->type = Type::Int64, std::memcpy(column->data.data(), &value.i64, 8); break;
            case ScalarValue::Kind::UInt64: column->type = Type::UInt64, std::memcpy(column->data.data(), &value.u64, 8); break;
            case ScalarValue::Kind::Float32: column->type = Type::Float32, std::memcpy(column->data.data(), &value.f32, 4); break;
            case ScalarValue::Kind::Float64: column->type = Type::Float64, std::memcpy(column->data.data(), &value.f64, 8); break;
            }
            column->data[16] = value.null ? 1 : 0;
            column->values = column->data.data();
            return column;
        };
        for (auto &accumulator : accumulators) {
            if (mode == AggregateMode::Partial) {
                for (const ScalarValue &value : accumulator->state()) result.columns.push_back(column_of(value));
            } else {
                result.columns.push_back(column_of(accumulator->evaluate()));
            }
        }
        *out = std::move(result);
        return PollState::ReadySome;
    }
};

} // namespace

AggregateFunctionExpr datafusion_aggregate(const std::string &function, const std::string &column_name, size_t column) {
    const ValueAccumulator::Function f = function_named(function);
    AggregateFunctionExpr expr;
    expr.name = function + "(" + column_name + ")";
    expr.column = column;
    expr.create = [f, column] { return std::make_unique<ValueAccumulator>(f, column); };
    return expr;
}

AggregateFunctionExpr model_aggregate(mdb_ctx *ctx, const std::string &function, std::optional<TimeRange> range) {
    AggregateFunctionExpr expr;
    expr.name = "model_" + function;
    expr.range = range;
    if (function == "count") expr.create = [ctx, range] { return make_model_count_accumulator(ctx, range); };
    else if (function == "min") expr.create = [ctx, range] { return make_model_min_accumulator(ctx, range); };
    else if (function == "max") expr.create = [ctx, range] { return make_model_max_accumulator(ctx, range); };
    else if (function == "sum") expr.create = [ctx, range] { return make_model_sum_accumulator(ctx, range); };
    else if (function == "avg") expr.create = [ctx, range] { return make_model_avg_accumulator(ctx, range); };
    else throw Error("Aggregate expression " + function + " is not supported."); // :322-326
    return expr;
}

std::vector<Field> AggregateExec::schema() const {
    std::vector<Field> out;
    for (const AggregateFunctionExpr &expr : aggr_expr_) {
        const std::string function = function_of(expr);
        if (mode_ == AggregateMode::Final) {
            out.push_back({expr.name, function == "count" ? Type::Int64 : (function == "min" || function == "max" ? Type::Float32 : Type::Float64)});
        } else if (function == "count") {
            out.push_back({expr.name + "[count]", Type::Int64});
        } else if (function == "min" || function == "max") {
            out.push_back({expr.name + "[" + function + "]", Type::Float32});
        } else if (function == "sum") {
            out.push_back({expr.name + "[sum]", Type::Float64});
        } else {
            out.push_back({expr.name + "[count]", Type::UInt64});
            out.push_back({expr.name + "[sum]", Type::Float64});
        }
    }
    return out;
}

std::unique_ptr<SegmentStream> AggregateExec::execute_stream(size_t partition, size_t batch_size) {
    auto stream = std::make_unique<AggregateStream>();
    stream->mode = mode_;
    stream->aggr_expr = aggr_expr_;
    for (const AggregateFunctionExpr &expr : aggr_expr_) {
        // The Final aggregate merges with DataFusion's own accumulator whatever made the states.
        if (mode_ == AggregateMode::Final)
            stream->accumulators.push_back(std::make_unique<ValueAccumulator>(function_named(function_of(expr)), 0));
        else
            stream->accumulators.push_back(expr.create());
    }
    stream->input = input_->execute_stream(partition, batch_size);
    stream->schema = schema();
    return stream;
}

std::vector<ScalarValue> AggregateExec::collect(ExecutionPlan &final_aggregate, size_t batch_size) {
    std::unique_ptr<SegmentStream> stream = final_aggregate.execute_stream(0, batch_size);
    RecordBatch batch;
    if (stream->poll_next(&batch) != PollState::ReadySome) throw Error("The aggregate returned no row.");
    std::vector<ScalarValue> results;
    for (const ColumnPtr &column : batch.columns) {
        ScalarValue value{ScalarValue::Kind::Int64};
        switch (column->type) {
        case Type::Int64: value.i64 = column->as<int64_t>()[0]; break;
        case Type::Float32: value.kind = ScalarValue::Kind::Float32, value.f32 = column->as<float>()[0]; break;
        case Type::Float64: value.kind = ScalarValue::Kind::Float64, value.f64 = column->as<double>()[0]; break;
        default: throw Error("An aggregate of this type cannot be returned.");
        }
        value.null = column->data.size() > 16 && column->data[16] != 0;
        results.push_back(value);
    }
    return results;
}

// ---- TimeSeriesTable::scan ---------------------------------------------------------------------------------------------

TimeSeriesTable::TimeSeriesTable(mdb_ctx *ctx, size_t n_fields, std::vector<std::string> tag_names)
    : ctx_(ctx), tag_names_(std::move(tag_names)) {
    if (n_fields == 0) throw Error("A time series table has at least one field column.");
    query_schema_.push_back({"timestamp", Type::Timestamp});
    for (size_t f = 0; f < n_fields; f++) {
        query_schema_.push_back({"field_" + std::to_string(f + 1), Type::Float32});
        sources_.push_back(std::make_shared<DataSourceExec::Source>());
    }
    for (const std::string &tag : tag_names_) query_schema_.push_back({tag, Type::Utf8View});
}

void TimeSeriesTable::push_segments(size_t field, RecordBatch batch) { sources_.at(field)->batches.push_back(std::move(batch)); }

std::shared_ptr<ExecutionPlan> TimeSeriesTable::scan(const std::vector<size_t> &projection, const std::vector<ExprPtr> &filters,
                                                     std::optional<size_t> limit) const {
    std::vector<Field> schema_after_projection;
    std::vector<SortedJoinColumnType> stored_columns_in_projection;
    std::vector<size_t> stored_field_columns_in_projection;
    for (size_t query_schema_index : projection) {
        const Field &field = query_schema_.at(query_schema_index);
        schema_after_projection.push_back(field);
        if (field.type == Type::Timestamp) {
            stored_columns_in_projection.push_back(SortedJoinColumnType::timestamp());
        } else if (field.type == Type::Utf8View) {
            stored_columns_in_projection.push_back(SortedJoinColumnType::tag(field.name));
        } else {
            stored_field_columns_in_projection.push_back(query_schema_index - 1);
            stored_columns_in_projection.push_back(SortedJoinColumnType::field());
        }
    }
    const RewrittenFilters rewritten = rewrite_and_combine_filters(query_schema_, filters);
    if (stored_field_columns_in_projection.empty()) stored_field_columns_in_projection.push_back(0); // fallback_field_column
    std::vector<Field> segment_schema = query_compressed_schema();
    for (const std::string &tag : tag_names_) segment_schema.push_back({tag, Type::Utf8View});
    std::vector<std::shared_ptr<ExecutionPlan>> field_column_execution_plans;
    for (size_t field_column_index : stored_field_columns_in_projection) {
        auto data_source_exec = std::make_shared<DataSourceExec>(segment_schema, sources_.at(field_column_index), rewritten.parquet, limit);
        field_column_execution_plans.push_back(GridExec::make(ctx_, grid_schema(tag_names_), rewritten.grid, limit, data_source_exec));
    }
    return SortedJoinExec::make(schema_after_projection, stored_columns_in_projection, field_column_execution_plans);
}

namespace {
void columns_of(const Expr &expr, std::vector<std::string> *out) {
    if (expr.kind == Expr::Kind::Column) out->push_back(expr.column);
    if (expr.kind == Expr::Kind::BinaryExpr) {
        columns_of(*expr.left, out);
        columns_of(*expr.right, out);
    }
}
} // namespace

std::shared_ptr<ExecutionPlan> plan_aggregate_query(const TimeSeriesTable &table,
                                                    const std::vector<std::pair<std::string, size_t>> &aggregates,
                                                    const std::vector<ExprPtr> &filters) {
    const std::vector<Field> query_schema = table.query_schema();
    // The columns the query reads: the aggregated field columns, then what only the filters name.
    std::vector<size_t> projection;
    auto need = [&](size_t query_schema_index) {
        if (std::find(projection.begin(), projection.end(), query_schema_index) == projection.end()) projection.push_back(query_schema_index);
    };
    for (const auto &aggregate : aggregates) need(1 + aggregate.second);
    const size_t n_aggregated = projection.size();
    for (const ExprPtr &filter : filters) {
        std::vector<std::string> names;
        columns_of(*filter, &names);
        for (const std::string &name : names) {
            size_t index = 0;
            while (index < query_schema.size() && query_schema[index].name != name) index++;
            if (index == query_schema.size()) throw Error("The columns in the filter should exist in the query schema.");
            need(index);
        }
    }
    // supports_filters_pushdown is Inexact for every filter (:674-682): scan() gets them AND a FilterExec stays.
    std::shared_ptr<ExecutionPlan> input = std::make_shared<PassThroughExec>("RepartitionExec", table.scan(projection, filters, std::nullopt));
    std::vector<Field> aggregate_input_schema;
    for (size_t k = 0; k < n_aggregated; k++) aggregate_input_schema.push_back(query_schema[projection[k]]);
    if (!filters.empty()) {
        std::vector<size_t> keep(n_aggregated);
        for (size_t k = 0; k < n_aggregated; k++) keep[k] = k;
        input = std::make_shared<FilterExec>(conjunction(filters), input, keep);
    }
    std::vector<AggregateFunctionExpr> aggr_expr;
    for (const auto &aggregate : aggregates) {
        const size_t column = static_cast<size_t>(std::find(projection.begin(), projection.end(), 1 + aggregate.second) - projection.begin());
        aggr_expr.push_back(datafusion_aggregate(aggregate.first, query_schema[1 + aggregate.second].name, column));
    }
    auto partial = std::make_shared<AggregateExec>(AggregateMode::Partial, aggr_expr, input, aggregate_input_schema);
    auto coalesce = std::make_shared<PassThroughExec>("CoalescePartitionsExec", partial);
    return std::make_shared<AggregateExec>(AggregateMode::Final, aggr_expr, coalesce, aggregate_input_schema);
}

// ---- ModelSimpleAggregates ----------------------------------------------------------------------------------------------

namespace {

bool named(const std::shared_ptr<ExecutionPlan> &plan, const char *name) { return std::strcmp(plan->name(), name) == 0; }

// can_rewrite_aggregate (:284-302), extended: a filter on the DataSourceExec is accepted if the query's predicate is
// a time range (then it only keeps segments out that have no point in the range, which the accumulators would not
// count either) and the filter reads nothing but start_time / end_time.
void can_rewrite_aggregate(const std::shared_ptr<ExecutionPlan> &grid_exec_child, const std::optional<TimeRange> &range) {
    auto data_source_exec = std::dynamic_pointer_cast<DataSourceExec>(grid_exec_child);
    if (data_source_exec) {
        const ExprPtr filter = data_source_exec->filter();
        if (!filter) return;
        if (range) {
            std::vector<std::string> names;
            columns_of(*filter, &names);
            if (std::all_of(names.begin(), names.end(), [](const std::string &n) { return n == "start_time" || n == "end_time"; })) return;
        }
    }
    throw Error("The input to GridExec must be a DataSourceExec without predicates or with a time range only.");
}

// try_new_aggregate_exec (:253-280)
std::shared_ptr<ExecutionPlan> try_new_aggregate_exec(mdb_ctx *ctx, const AggregateExec &aggregate_exec,
                                                      const std::vector<std::shared_ptr<ExecutionPlan>> &grid_execs,
                                                      const std::optional<TimeRange> &range) {
    if (grid_execs.size() > 1) throw Error("All aggregates must be for the same FIELD column.");
    const std::shared_ptr<ExecutionPlan> grid_exec_child = grid_execs[0]->children().at(0);
    can_rewrite_aggregate(grid_exec_child, range);
    std::vector<AggregateFunctionExpr> model_based; // try_rewrite_aggregate_exprs (:304-334)
    for (const AggregateFunctionExpr &expr : aggregate_exec.aggr_expr()) model_based.push_back(model_aggregate(ctx, function_of(expr), range));
    return std::make_shared<AggregateExec>(aggregate_exec.mode(), model_based, grid_exec_child, aggregate_exec.input_schema());
}

// rewrite_aggregates_to_use_segments (:209-251)
std::shared_ptr<ExecutionPlan> rewrite_aggregates_to_use_segments(mdb_ctx *ctx, const std::shared_ptr<ExecutionPlan> &execution_plan,
                                                                  bool *transformed) {
    const auto children = execution_plan->children();
    auto aggregate_exec = children.size() == 1 ? std::dynamic_pointer_cast<AggregateExec>(children[0]) : nullptr;
    if (!aggregate_exec) return execution_plan;
    if (aggregate_exec->input_schema().size() != 1 || aggregate_exec->input_schema()[0].type != Type::Float32) return execution_plan;
    // Look past RepartitionExec / CoalesceBatchesExec and - the extension - ONE FilterExec whose predicate is a
    // time range and nothing else.
    std::shared_ptr<ExecutionPlan> input = aggregate_exec->children()[0];
    std::optional<TimeRange> range;
    while (true) {
        if (named(input, "RepartitionExec") || named(input, "CoalesceBatchesExec")) {
            input = input->children()[0];
        } else if (auto filter_exec = std::dynamic_pointer_cast<FilterExec>(input)) {
            if (range) return execution_plan;
            const std::vector<Field> filtered = filter_exec->children()[0]->schema();
            std::string timestamp_column;
            for (const Field &field : filtered)
                if (field.type == Type::Timestamp) timestamp_column = field.name;
            const auto found = timestamp_column.empty() ? std::nullopt : time_range_of_predicate(*filter_exec->predicate(), timestamp_column);
            if (!found || !found->exact) return execution_plan;
            range = found->range;
            input = filter_exec->children()[0];
        } else {
            break;
        }
    }
    auto sorted_join_exec = std::dynamic_pointer_cast<SortedJoinExec>(input);
    if (!sorted_join_exec) return execution_plan;
    try {
        std::shared_ptr<ExecutionPlan> replacement = try_new_aggregate_exec(ctx, *aggregate_exec, sorted_join_exec->children(), range);
        *transformed = true;
        return execution_plan->with_new_children_dyn({replacement});
    } catch (const Error &) {
        return execution_plan;
    }
}

// transform_down: the node first, then the children of what it became.
std::shared_ptr<ExecutionPlan> transform_down(mdb_ctx *ctx, std::shared_ptr<ExecutionPlan> plan) {
    bool transformed = false;
    plan = rewrite_aggregates_to_use_segments(ctx, plan, &transformed);
    std::vector<std::shared_ptr<ExecutionPlan>> children = plan->children();
    bool changed = false;
    for (auto &child : children) {
        auto rewritten = transform_down(ctx, child);
        changed |= rewritten != child;
        child = rewritten;
    }
    return changed ? plan->with_new_children_dyn(children) : plan;
}

} // namespace

std::shared_ptr<ExecutionPlan> ModelSimpleAggregates::optimize(std::shared_ptr<ExecutionPlan> execution_plan) const {
    return transform_down(ctx, std::move(execution_plan));
}

std::string plan_levels(const std::shared_ptr<ExecutionPlan> &plan) {
    std::string text;
    std::vector<std::shared_ptr<ExecutionPlan>> level = {plan};
    while (!level.empty()) {
        std::vector<std::shared_ptr<ExecutionPlan>> next;
        for (size_t k = 0; k < level.size(); k++) {
            text += (k ? "," : "") + std::string(level[k]->name());
            for (auto &child : level[k]->children()) next.push_back(child);
        }
        text += "\n";
        level = std::move(next);
    }
    return text;
}

} // namespace mdbhost

// =================================================================================================
// C surface (the tests' and bench.py's way in). 0 = ok, 1 = error (message via mdbh_last_error()).
// =================================================================================================

namespace {

struct QueryHandle {
    std::shared_ptr<mdbhost::TimeSeriesTable> table;
    std::shared_ptr<mdbhost::ExecutionPlan> plan;
};

std::vector<std::string> split(const std::string &text, char separator) {
    std::vector<std::string> parts;
    std::string part;
    std::istringstream stream(text);
    while (std::getline(stream, part, separator))
        if (!part.empty()) parts.push_back(part);
    return parts;
}

void copy_text(const std::string &text, char *out, uint64_t cap) {
    if (cap == 0) return;
    std::strncpy(out, text.c_str(), cap - 1);
    out[cap - 1] = 0;
}

} // namespace

extern "C" {

int mdbh_fail(const char *message); // (mdb_host.cpp: sets the thread's error text, returns 1)

#define MDBH_GUARDED(...)                                                                                                      \
    try {                                                                                                                      \
        __VA_ARGS__;                                                                                                           \
        return 0;                                                                                                              \
    } catch (const std::exception &e) {                                                                                       \
        return mdbh_fail(e.what());                                                                                            \
    }

/* rewrite_and_combine_filters: `filters` are expressions over (timestamp, field_1.., tags) separated by ';'. The
 * rewritten filters come back as text ("" for None). */
int mdbh_rewrite_filters(int32_t n_fields, const char *filters, char *parquet_out, char *grid_out, uint64_t cap) {
    MDBH_GUARDED({
        mdbhost::TimeSeriesTable table(nullptr, static_cast<size_t>(n_fields), {"tag"});
        std::vector<mdbhost::ExprPtr> parsed;
        for (const std::string &text : split(filters ? filters : "", ';')) parsed.push_back(mdbhost::parse_expr(text));
        const mdbhost::RewrittenFilters rewritten = mdbhost::rewrite_and_combine_filters(table.query_schema(), parsed);
        copy_text(rewritten.parquet ? rewritten.parquet->to_string() : "", parquet_out, cap);
        copy_text(rewritten.grid ? rewritten.grid->to_string() : "", grid_out, cap);
    })
}

/* time_range_of_predicate over the column "timestamp": found = 0 / 1, exact = 0 / 1, [lo, hi]. */
int mdbh_time_range_of_predicate(const char *predicate, int32_t *found, int32_t *exact, int64_t *lo, int64_t *hi) {
    MDBH_GUARDED({
        const auto range = mdbhost::time_range_of_predicate(*mdbhost::parse_expr(predicate), "timestamp");
        *found = range ? 1 : 0;
        *exact = range && range->exact ? 1 : 0;
        *lo = range ? range->range.lo : INT64_MIN;
        *hi = range ? range->range.hi : INT64_MAX;
    })
}

int mdbh_query_table_create(mdb_ctx *ctx, int32_t n_fields, const char *const *tag_names, int32_t n_tags, void **out) {
    MDBH_GUARDED({
        auto handle = std::make_unique<QueryHandle>();
        handle->table = std::make_shared<mdbhost::TimeSeriesTable>(ctx, static_cast<size_t>(n_fields),
                                                                   std::vector<std::string>(tag_names, tag_names + n_tags));
        *out = handle.release();
    })
}

/* A batch of segments (QUERY_COMPRESSED_SCHEMA + the tag columns) of field column `field` (0-based). */
int mdbh_query_table_push(void *handle, int32_t field, ArrowArray *array, ArrowSchema *schema) {
    MDBH_GUARDED(static_cast<QueryHandle *>(handle)->table->push_segments(static_cast<size_t>(field),
                                                                          mdbhost::import_record_batch(array, schema)))
}

/* Plans SELECT <aggregates> FROM table WHERE <filters>: aggregates as "count:0,sum:0" (function:field column),
 * filters separated by ';'. optimize != 0: the ModelSimpleAggregates rule is applied to the plan. */
int mdbh_query_plan(void *handle, mdb_ctx *ctx, const char *aggregates, const char *filters, int32_t optimize) {
    MDBH_GUARDED({
        QueryHandle *query = static_cast<QueryHandle *>(handle);
        std::vector<std::pair<std::string, size_t>> wanted;
        for (const std::string &text : split(aggregates ? aggregates : "", ',')) {
            const size_t colon = text.find(':');
            if (colon == std::string::npos) throw mdbhost::Error("An aggregate is written function:field.");
            wanted.emplace_back(text.substr(0, colon), static_cast<size_t>(std::stoul(text.substr(colon + 1))));
        }
        std::vector<mdbhost::ExprPtr> parsed;
        for (const std::string &text : split(filters ? filters : "", ';')) parsed.push_back(mdbhost::parse_expr(text));
        query->plan = mdbhost::plan_aggregate_query(*query->table, wanted, parsed);
        if (optimize) query->plan = mdbhost::ModelSimpleAggregates{ctx}.optimize(query->plan);
    })
}

/* The plan level by level (one line per level, names separated by commas), then one line per aggregate of the
 * Partial AggregateExec: its name and, for a model-based one under a range, "[lo,hi]". */
int mdbh_query_describe(void *handle, char *out, uint64_t cap) {
    MDBH_GUARDED({
        QueryHandle *query = static_cast<QueryHandle *>(handle);
        if (!query->plan) throw mdbhost::Error("The query has not been planned.");
        std::string text = mdbhost::plan_levels(query->plan);
        std::shared_ptr<mdbhost::ExecutionPlan> node = query->plan;
        std::shared_ptr<mdbhost::AggregateExec> partial;
        while (node) {
            auto aggregate = std::dynamic_pointer_cast<mdbhost::AggregateExec>(node);
            if (aggregate && aggregate->mode() == mdbhost::AggregateMode::Partial) partial = aggregate;
            node = node->children().empty() ? nullptr : node->children()[0];
        }
        if (partial) {
            for (const mdbhost::AggregateFunctionExpr &expr : partial->aggr_expr()) {
                text += "aggregate " + expr.name;
                if (expr.range) text += "[" + std::to_string(expr.range->lo) + "," + std::to_string(expr.range->hi) + "]";
                text += "\n";
            }
        }
        copy_text(text, out, cap);
    })
}

/* Runs the planned query: per aggregate its value (as f64) and whether it is NULL. */
int mdbh_query_execute(void *handle, uint64_t batch_size, double *values, int32_t *nulls, int32_t *n_values) {
    MDBH_GUARDED({
        QueryHandle *query = static_cast<QueryHandle *>(handle);
        if (!query->plan) throw mdbhost::Error("The query has not been planned.");
        const std::vector<mdbhost::ScalarValue> results = mdbhost::AggregateExec::collect(*query->plan, batch_size);
        *n_values = static_cast<int32_t>(results.size());
        for (size_t k = 0; k < results.size(); k++) {
            nulls[k] = results[k].null ? 1 : 0;
            switch (results[k].kind) {
            case mdbhost::ScalarValue::Kind::Int64: values[k] = static_cast<double>(results[k].i64); break;
            case mdbhost::ScalarValue::Kind::UInt64: values[k] = static_cast<double>(results[k].u64); break;
            case mdbhost::ScalarValue::Kind::Float32: values[k] = results[k].f32; break;
            case mdbhost::ScalarValue::Kind::Float64: values[k] = results[k].f64; break;
            }
        }
    })
}

void mdbh_query_free(void *handle) { delete static_cast<QueryHandle *>(handle); }

} // extern "C"
