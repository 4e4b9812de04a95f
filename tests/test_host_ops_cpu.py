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
