"""The launches a GridStream makes, seen from the kernels' side: tests/stub's stand-in for the kernels logs every
grid launch (input batches, segments, rows reserved in front) and every mdb_compress_chunk_list, and
tests/test_gpu_host_ops.py::test_grid_stream_gathers_input_batches_and_keeps_one_submit_ahead asserts the sequence
when MDB_STUB_CALL_LOG is set. This is the call sequence rust/patches/0001-grid_exec.patch makes through
rust/modelardb_hip (Context::grid_submit / GridTicket::wait); no Rust toolchain exists here, so the C++ twin of
the patched stream (modelardb-rs_amd/csrc/host/mdb_host.cpp: poll_input_and_submit,
wait_and_append_to_leftovers_in_current_batch) is what runs."""

import os
import subprocess
import sys

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(REPO_ROOT, "tests", "stub")


def test_launches_of_a_grid_stream_over_canned_answers(tmp_path):
    done = subprocess.run(["make", "-C", STUB, "_build/libmdb_host_stub.so"], capture_output=True, text=True)
    assert done.returncode == 0, done.stdout + done.stderr
    env = dict(os.environ,
               MDB_HOST_LIBRARY_UNDER_TEST=os.path.join(STUB, "_build", "libmdb_host_stub.so"),
               MDB_STUB_FIXTURES=os.path.join(REPO_ROOT, "tests", "golden", "host_stub_fixtures.bin"),
               MDB_STUB_CALL_LOG=str(tmp_path / "calls.log"))
    done = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                           "tests/test_gpu_host_ops.py", "-k", "gathers_input_batches or one_launch"],
                          cwd=REPO_ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, (done.stdout + done.stderr)[-4000:]
    assert " passed" in done.stdout


def test_rust_patch_and_cpp_twin_name_the_same_steps():
    """The patched Rust stream and its C++ twin are kept side by side: same function names, same calls."""
    patch = open(os.path.join(REPO_ROOT, "rust", "patches", "0001-grid_exec.patch")).read()
    twin = open(os.path.join(REPO_ROOT, "modelardb-rs_amd", "csrc", "host", "mdb_host.cpp")).read()
    binding = open(os.path.join(REPO_ROOT, "rust", "modelardb_hip", "src", "lib.rs")).read()
    for name in ("poll_input_and_submit", "wait_and_append_to_leftovers_in_current_batch"):
        assert name in patch and name in twin
    for call in ("mdb_grid_submit", "mdb_grid_wait", "mdb_grid_cancel", "mdb_grid_result_tag_views"):
        assert f"sys::{call}(" in binding and f"{call}(" in twin
    for call in ("mdb_compress_chunk_list", "mdb_agg_batch_list"):
        assert f"sys::{call}(" in binding and f"{call}(" in twin
    # the accumulators: batches gathered until 262 144 segments are pending or the state is read, then ONE call
    aggregates = open(os.path.join(REPO_ROOT, "rust", "patches", "0002-model_simple_aggregates.patch")).read()
    assert "PENDING_SEGMENTS_PER_CALL: usize = 262_144" in aggregates and "aggregate_list(" in aggregates
    assert "PENDING_SEGMENTS_PER_CALL = 262144" in twin and "fold_pending()" in twin and "fold_pending()" in aggregates
    for constant, value in (("SUBMIT_TARGET_ROWS", "16 * 1024 * 1024"), ("SUBMIT_MAX_SEGMENTS", "1024 * 1024")):
        assert f"{constant}: u64 = {value}" in patch
    assert "GRID_SUBMIT_TARGET_POINTS = 16u << 20" in twin and "GRID_SUBMIT_MAX_SEGMENTS = 1u << 20" in twin


def test_rust_patches_and_cpp_twin_push_the_same_time_range(tmp_path):
    """SURVEY 8(f) N1 on both host sides: the range a predicate puts on the timestamps is worked out by a function of
    the same name and handed to the same two library calls - every grid launch of the stream, and ONE
    mdb_agg_batch_range_list per accumulator for all the batches it gathered."""
    grid = open(os.path.join(REPO_ROOT, "rust", "patches", "0001-grid_exec.patch")).read()
    aggregates = open(os.path.join(REPO_ROOT, "rust", "patches", "0002-model_simple_aggregates.patch")).read()
    binding = open(os.path.join(REPO_ROOT, "rust", "modelardb_hip", "src", "lib.rs")).read()
    host = os.path.join(REPO_ROOT, "modelardb-rs_amd", "csrc", "host")
    twin = open(os.path.join(host, "mdb_host.cpp")).read() + open(os.path.join(host, "mdb_host_query.cpp")).read()
    # the stream: the range is found once and goes into every submit; only a predicate that is more than the range
    # is evaluated behind the library
    assert "+pub(crate) fn time_range_of_predicate(" in grid and "time_range_of_predicate(" in twin
    assert "+            .grid_submit(batches, self.batch_size, maybe_time_range)" in grid
    assert "grid_submit(batches, self.batch_size, None)" not in grid
    assert "&& !predicate_is_exact_range" in grid and "pushed_range_->exact" in twin
    assert "request.t_lo = pushdown ? pushed_range_->range.lo" in twin
    # the rule and the accumulators
    assert "+use crate::query::grid_exec::time_range_of_predicate;" in aggregates
    assert "fn time_range_of_filter_exec(" in aggregates and "std::dynamic_pointer_cast<FilterExec>(input)" in twin
    assert "hip.aggregate_range_list(&segments, start_time, end_time, which_mask, state)" in aggregates
    assert "sys::mdb_agg_batch_range_list(" in binding and "mdb_agg_batch_range_list(ctx_" in twin
    for both in (aggregates, twin):   # COUNT is asked for under a range: a range without a point is NULL
        assert "MDB_AGG_COUNT" in both
    # and what the kernels' side sees when the twin runs a ranged stream and ranged accumulators
    done = subprocess.run(["make", "-C", STUB, "_build/libmdb_host_stub.so"], capture_output=True, text=True)
    assert done.returncode == 0, done.stdout + done.stderr
    env = dict(os.environ,
               MDB_HOST_LIBRARY_UNDER_TEST=os.path.join(STUB, "_build", "libmdb_host_stub.so"),
               MDB_STUB_FIXTURES=os.path.join(REPO_ROOT, "tests", "golden", "host_stub_fixtures.bin"),
               MDB_STUB_CALL_LOG=str(tmp_path / "calls.log"))
    done = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                           "tests/test_gpu_host_ops.py", "-k",
                           "pushes_the_range or accumulators_under_a_time_range or under_a_time_range_through_the_rule"],
                          cwd=REPO_ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, (done.stdout + done.stderr)[-4000:]
    assert " passed" in done.stdout
