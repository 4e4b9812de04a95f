"""CPU-side checks of the C++ host operators that need no GPU: the library loads, GridStream's
poll protocol (Pending / Ready(None)) and GridExec's plan-facing surface."""

import ctypes as C

import modelardb_rs_amd as mdb
from modelardb_rs_amd import host


class _NullContext:
    handle = C.c_void_p()


def test_host_library_exports_the_operator_surface():
    library = host.lib()
    for name in ("mdbh_grid_exec_create", "mdbh_grid_stream_push", "mdbh_grid_stream_poll_next",
                 "mdbh_grid_stream_metrics", "mdbh_accumulator_create", "mdbh_accumulator_update_batch",
                 "mdbh_accumulator_state", "mdbh_try_compress_univariate_time_series",
                 "mdbh_try_compress_multivariate_time_series", "mdbh_last_error"):
        assert hasattr(library, name)


def test_grid_stream_poll_protocol_without_input():
    stream = host.GridStream(_NullContext(), tag_names=("tag",), limit=None, batch_size=8192)
    assert stream.poll_next() == (host.GridStream.PENDING, None)      # the child is not ready
    stream.finish_input()
    assert stream.poll_next() == (host.GridStream.READY_NONE, None)   # grid_exec.rs:412-416
    stream.close()


def test_grid_exec_plan_surface():
    # name(), DisplayAs, children(), required_input_distribution(), with_new_children()
    # (crates/modelardb_storage/src/query/grid_exec.rs:112-209).
    stream = host.GridStream(_NullContext(), tag_names=(), limit=5, batch_size=8192)
    description = stream.describe()
    assert description.startswith("GridExec|GridExec: limit=Some(5)|children=1|batch_size=5|")
    assert "distribution=SinglePartition" in description
    assert "with_new_children([])=Err(Exactly one child must be provided" in description
    unlimited = host.GridStream(_NullContext(), tag_names=(), limit=None, batch_size=100)
    assert "limit=None" in unlimited.describe() and "batch_size=100" in unlimited.describe()


def test_unknown_aggregate_is_not_supported():
    import pytest
    handle = C.c_void_p()
    assert host.lib().mdbh_accumulator_create(None, C.c_int32(9), C.byref(handle)) == 1
    assert b"not supported" in host.lib().mdbh_last_error()


def test_uncompressed_data_manager_buffers_by_series_and_finishes_unused():
    # storage/uncompressed_data_manager.rs:130-189, 405-451 and uncompressed_data_buffer.rs:135-137:
    # a buffer is finished when full or when an ingested batch does not touch it.
    import pyarrow as pa
    schema_names = ["timestamp", "field_1", "tag"]

    def batch(rows):
        return pa.RecordBatch.from_arrays([
            pa.array([r[0] for r in rows], type=pa.int64()).cast(pa.timestamp("us")),
            pa.array([r[1] for r in rows], type=pa.float32()),
            pa.array([r[2] for r in rows], type=pa.string_view())], names=schema_names)

    first = batch([(100, 1.0, "A"), (100, 2.0, "B"), (200, 1.5, "A")])
    manager = host.UncompressedDataManager(_NullContext(), first.schema, 0, [1], [2], {},
                                           buffer_capacity=4)
    manager.insert_data_points(first)
    assert manager.counts() == (2, 0)                      # A and B active
    manager.insert_data_points(batch([(300, 1.7, "A")]))
    assert manager.counts() == (1, 1)                      # B was not touched -> finished
    manager.insert_data_points(batch([(400, 1.9, "A"), (500, 2.0, "A")]))
    assert manager.counts() == (1, 2)                      # A reached capacity 4 -> finished; new A buffer
    manager.flush()
    assert manager.counts() == (0, 3)


def _points(timestamps, values, tag):
    import pyarrow as pa
    return pa.RecordBatch.from_arrays([
        pa.array(timestamps, type=pa.int64()).cast(pa.timestamp("us")),
        pa.array(values, type=pa.float32()),
        pa.array([tag] * len(timestamps), type=pa.string_view())], names=["timestamp", "value", "tag"])


def test_sorted_join_stream_zips_fields_in_return_order():
    # sorted_join_exec.rs:277-311: timestamp and tags from input 0, one value column per input, in the
    # order return_order asks for.
    order = [("tag", "tag"), "field", "timestamp", "field"]
    join = host.SortedJoinStream(None, 2, order, tag_names=("tag",), use_grid=False)
    join.push(0, _points([100, 200, 300], [1.0, 2.0, 3.0], "A"))
    join.push(1, _points([100, 200, 300], [10.0, 20.0, 30.0], "A"))
    state, batch = join.poll_next()
    assert state == host.SortedJoinStream.READY_SOME
    assert batch.schema.names == ["tag", "field_0", "timestamp", "field_1"]
    assert batch.column(0).to_pylist() == ["A", "A", "A"]
    assert batch.column(1).to_pylist() == [1.0, 2.0, 3.0]
    assert batch.column(2).cast("int64").to_pylist() == [100, 200, 300]
    assert batch.column(3).to_pylist() == [10.0, 20.0, 30.0]
    assert "output_rows=3" in join.describe()


def test_sorted_join_stream_poll_protocol_and_smallest_batch(monkeypatch):
    import pytest
    for carry_over in (False, True):
        if carry_over:
            monkeypatch.setenv("MDB_HOST_SORTED_JOIN_CARRY_OVER", "1")
        else:
            monkeypatch.delenv("MDB_HOST_SORTED_JOIN_CARRY_OVER", raising=False)
        join = host.SortedJoinStream(None, 2, ["timestamp", "field", "field"], tag_names=("tag",), use_grid=False)
        assert join.poll_next() == (host.SortedJoinStream.PENDING, None)
        join.push(0, _points([1, 2, 3, 4], [1.0, 2.0, 3.0, 4.0], "A"))
        assert join.poll_next() == (host.SortedJoinStream.PENDING, None)   # input 1 has nothing yet; batch 0 kept
        join.push(1, _points([1, 2], [5.0, 6.0], "A"))
        state, batch = join.poll_next()
        # Inputs of different length are cut to the smallest (sorted_join_exec.rs:248-272) ...
        assert state == host.SortedJoinStream.READY_SOME and batch.num_rows == 2
        assert batch.column(0).cast("int64").to_pylist() == [1, 2]
        assert batch.column(2).to_pylist() == [5.0, 6.0]
        join.push(1, _points([3], [7.0], "A"))
        if carry_over:
            # ... and with the switch the surplus of the longer one waits for the next poll: the rows stay aligned.
            state, batch = join.poll_next()
            assert state == host.SortedJoinStream.READY_SOME
            assert batch.column(0).cast("int64").to_pylist() == [3] and batch.column(2).to_pylist() == [7.0]
        else:
            # ... and by default the surplus is DROPPED, as the reference drops it (rows 3 and 4 of input 0 are
            # gone): the join waits for a new batch of input 0, and pairs whatever comes with input 1's next rows.
            assert join.poll_next() == (host.SortedJoinStream.PENDING, None)
            join.push(0, _points([5, 6], [5.5, 6.5], "A"))
            state, batch = join.poll_next()
            assert state == host.SortedJoinStream.READY_SOME
            assert batch.column(0).cast("int64").to_pylist() == [5] and batch.column(2).to_pylist() == [7.0]
        join.finish_input(1)
        if carry_over:
            assert join.poll_next() == (host.SortedJoinStream.READY_NONE, None)  # a finished input ends the join
        join.close()


def test_sorted_join_exec_plan_surface():
    # sorted_join_exec.rs:104-198. GridExec inputs after the first only reconstruct values.
    join = host.SortedJoinStream(_NullContext(), 3, ["timestamp", "field", "field", "field"], use_grid=True)
    description = join.describe()
    assert description.startswith("SortedJoinExec|SortedJoinExec|children=3|")
    assert "distribution=SinglePartition,SinglePartition,SinglePartition," in description
    assert "values_only=011" in description
    assert "with_new_children([])=Err(At least one child must be provided" in description
    assert join.n_inputs() == 3
