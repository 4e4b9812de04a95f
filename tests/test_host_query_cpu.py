"""The query side of the host operators without a GPU: the filter rewrite of TimeSeriesTable::scan, the time range a
GridStream / the ModelSimpleAggregates rule take from a predicate, and the plans the rule makes - written after the
reference's own tests: crates/modelardb_storage/src/query/time_series_table.rs:714-836 (rewrite_and_combine_filters)
and crates/modelardb_storage/src/optimizer/model_simple_aggregates.rs:637-719 (plan shapes), plus the shapes of
SURVEY 8(f) N1 (a range on the timestamp) that rust/patches/0002 adds. Nothing here computes: plans are built,
rewritten and described."""

import pytest

from modelardb_rs_amd import host

TIMESTAMP_PREDICATE_VALUE = 37   # time_series_table.rs:712
I64_MIN, I64_MAX = -(1 << 63), (1 << 63) - 1


# ---- rewrite_and_combine_filters (time_series_table.rs:714-836) -----------------------------------------------------------

def test_rewrite_empty_vec():
    assert host.rewrite_filters([]) == (None, None)


@pytest.mark.parametrize("operator, text, segment_column", [
    (">", ">", "end_time"), (">=", ">=", "end_time"), ("<", "<", "start_time"), ("<=", "<=", "start_time")])
def test_rewrite_comparison_of_the_timestamp(operator, text, segment_column):
    parquet, grid = host.rewrite_filters([f"({operator} timestamp ts:{TIMESTAMP_PREDICATE_VALUE})"])
    assert parquet == f"{segment_column} {text} TimestampMicrosecond(37)"
    assert grid == f"timestamp {text} TimestampMicrosecond(37)"


def test_rewrite_equal_timestamp():
    parquet, grid = host.rewrite_filters(["(= timestamp ts:37)"])
    assert parquet == "start_time <= TimestampMicrosecond(37) AND end_time >= TimestampMicrosecond(37)"
    assert grid == "timestamp = TimestampMicrosecond(37)"


def test_rewrite_with_the_column_on_the_right():
    # time_series_table.rs:334-349: `literal op column` keeps its sides
    assert host.rewrite_filters(["(> ts:37 timestamp)"]) == (
        "TimestampMicrosecond(37) > start_time", "TimestampMicrosecond(37) > timestamp")
    assert host.rewrite_filters(["(<= ts:37 timestamp)"]) == (
        "TimestampMicrosecond(37) <= end_time", "TimestampMicrosecond(37) <= timestamp")


def test_filters_on_other_columns_are_not_rewritten_and_the_rest_is_combined():
    # rewrite_filter returns None for anything but the timestamp column (:297-331); utils::conjunction of the rest
    assert host.rewrite_filters(["(= field_1 f32:37.0)"]) == (None, None)
    assert host.rewrite_filters(["(!= timestamp ts:37)"]) == (None, None)
    parquet, grid = host.rewrite_filters(["(>= timestamp ts:100)", "(= tag str:a)", "(< timestamp ts:900)"])
    assert parquet == "end_time >= TimestampMicrosecond(100) AND start_time < TimestampMicrosecond(900)"
    assert grid == "timestamp >= TimestampMicrosecond(100) AND timestamp < TimestampMicrosecond(900)"


def test_a_filter_on_an_unknown_column_is_an_error():
    with pytest.raises(host.HostError, match="should exist in the query schema"):
        host.rewrite_filters(["(> no_such_column ts:1)"])


# ---- the time range of a predicate (rust/patches/0001: time_range_of_predicate) --------------------------------------------

@pytest.mark.parametrize("predicate, expected", [
    ("(>= timestamp ts:100)", (100, I64_MAX, True)),
    ("(> timestamp ts:100)", (101, I64_MAX, True)),
    ("(<= timestamp ts:900)", (I64_MIN, 900, True)),
    ("(< timestamp ts:900)", (I64_MIN, 899, True)),
    ("(= timestamp ts:500)", (500, 500, True)),
    ("(and (>= timestamp ts:100) (<= timestamp ts:900))", (100, 900, True)),
    ("(and (and (> timestamp ts:100) (< timestamp ts:900)) (>= timestamp ts:300))", (300, 899, True)),
    ("(> ts:900 timestamp)", (I64_MIN, 899, True)),                       # literal on the left: the sides swap
    ("(<= ts:100 timestamp)", (100, I64_MAX, True)),
    ("(and (>= timestamp ts:100) (= value f32:1.0))", (100, I64_MAX, False)),   # narrowed, but the filter still decides
    ("(and (>= timestamp ts:100) (or (< timestamp ts:50) (> timestamp ts:70)))", (100, I64_MAX, False)),
    ("(and (>= timestamp ts:900) (<= timestamp ts:100))", (I64_MAX, I64_MIN, True)),   # nothing
    (f"(> timestamp ts:{I64_MAX})", (I64_MAX, I64_MIN, True)),                # no overflow: nothing is later
    (f"(< timestamp ts:{I64_MIN})", (I64_MAX, I64_MIN, True)),
    ("(or (< timestamp ts:50) (> timestamp ts:70))", None),
    ("(!= timestamp ts:50)", None),
    ("(>= timestamp i64:100)", None),                                           # not a timestamp literal: left to the filter
    ("(= timestamp timestamp)", None),
    ("(= value f32:1.0)", None),
])
def test_time_range_of_predicate(predicate, expected):
    assert host.time_range_of_predicate(predicate) == expected


def test_malformed_expressions_are_errors():
    for text in ("(>= timestamp", "(like timestamp ts:1)", "(>= timestamp ts:abc)", "(>= timestamp ts:1) trailing", ""):
        with pytest.raises(host.HostError, match="Malformed expression"):
            host.time_range_of_predicate(text)


# ---- ModelSimpleAggregates: plan shapes (model_simple_aggregates.rs:637-719) ------------------------------------------------

REWRITTEN = [["AggregateExec"], ["CoalescePartitionsExec"], ["AggregateExec"], ["DataSourceExec"]]


def _plan(aggregates, filters=(), n_fields=2, optimize=True):
    return host.AggregateQuery(None, n_fields=n_fields, tag_names=("tag",)).plan(aggregates, filters, optimize)


def test_rewrite_aggregate_on_one_column_without_predicates():
    assert _plan([("count", 0)]).levels() == REWRITTEN
    assert _plan([("count", 0)]).aggregates() == ["model_count"]


def test_rewrite_aggregates_on_one_column_without_predicates():
    query = _plan([("count", 0), ("min", 0), ("max", 0), ("sum", 0)])
    assert query.levels() == REWRITTEN
    assert query.aggregates() == ["model_count", "model_min", "model_max", "model_sum"]
    assert _plan([("avg", 0)]).levels() == REWRITTEN


def test_the_unoptimized_plan_is_what_the_reference_asserts_for_plans_it_leaves_alone():
    assert _plan([("count", 0)], optimize=False).levels() == [
        ["AggregateExec"], ["CoalescePartitionsExec"], ["AggregateExec"], ["RepartitionExec"], ["SortedJoinExec"],
        ["GridExec"], ["DataSourceExec"]]


def test_do_not_rewrite_aggregate_on_one_column_with_predicates():
    query = _plan([("count", 0)], ["(= field_1 f32:37.0)"])
    assert query.levels() == [
        ["AggregateExec"], ["CoalescePartitionsExec"], ["AggregateExec"], ["FilterExec"], ["RepartitionExec"],
        ["SortedJoinExec"], ["GridExec"], ["DataSourceExec"]]
    assert query.aggregates() == ["count(field_1)"]


def test_do_not_rewrite_aggregate_on_multiple_columns_without_predicates():
    assert _plan([("count", 0), ("count", 1)]).levels() == [
        ["AggregateExec"], ["CoalescePartitionsExec"], ["AggregateExec"], ["RepartitionExec"], ["SortedJoinExec"],
        ["GridExec", "GridExec"], ["DataSourceExec", "DataSourceExec"]]


def test_an_unsupported_aggregate_is_not_rewritten_and_not_planned():
    with pytest.raises(host.HostError, match="not supported"):
        _plan([("median", 0)])


# ---- the extension: a range on the timestamp (BASELINE configs 3 and 5) ----------------------------------------

def test_rewrite_aggregates_under_a_time_range():
    query = _plan([("count", 0), ("min", 0), ("max", 0), ("sum", 0)],
                  ["(>= timestamp ts:100)", "(< timestamp ts:900)"])
    assert query.levels() == REWRITTEN
    assert query.aggregates() == ["model_count[100,899]", "model_min[100,899]", "model_max[100,899]", "model_sum[100,899]"]
    assert _plan([("avg", 0)], ["(= timestamp ts:500)"]).aggregates() == ["model_avg[500,500]"]


def test_a_one_sided_range_is_rewritten_too():
    assert _plan([("sum", 0)], ["(> timestamp ts:100)"]).aggregates() == [f"model_sum[101,{I64_MAX}]"]
    assert _plan([("sum", 0)], ["(<= timestamp ts:900)"]).aggregates() == [f"model_sum[{I64_MIN},900]"]


@pytest.mark.parametrize("filters", [
    ["(>= timestamp ts:100)", "(= field_1 f32:37.0)"],                     # a value predicate next to the range
    ["(or (< timestamp ts:50) (> timestamp ts:70))"],                       # not a conjunction
    ["(!= timestamp ts:50)"],
    ["(>= timestamp ts:100)", "(= tag str:a)"],                             # a tag predicate next to the range
])
def test_do_not_rewrite_under_anything_but_a_pure_time_range(filters):
    query = _plan([("sum", 0)], filters)
    assert query.levels() == [
        ["AggregateExec"], ["CoalescePartitionsExec"], ["AggregateExec"], ["FilterExec"], ["RepartitionExec"],
        ["SortedJoinExec"], ["GridExec"], ["DataSourceExec"]]
    assert query.aggregates() == ["sum(field_1)"]


def test_do_not_rewrite_multiple_columns_under_a_time_range():
    query = _plan([("count", 0), ("count", 1)], ["(>= timestamp ts:100)"])
    assert query.levels()[3:] == [["FilterExec"], ["RepartitionExec"], ["SortedJoinExec"], ["GridExec", "GridExec"],
                                  ["DataSourceExec", "DataSourceExec"]]
