"""CPU-side checks of the drop-in boundary: the built library exports every symbol include/mdb.h
declares, the headers agree with the ctypes mirror, and the product path refuses to run without a
GPU instead of falling back to anything."""

import ctypes
import os
import re

import numpy as np
import pytest

import modelardb_rs_amd as mdb
from modelardb_rs_amd import _abi

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_in_header():
    text = open(os.path.join(REPO_ROOT, "include", "mdb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    library = ctypes.CDLL(_abi.HIP_LIBRARY_PATH)
    declared = _declared_in_header()
    assert declared, "no declarations found in include/mdb.h"
    for name in declared:
        assert hasattr(library, name), f"libmdb_hip.so does not export {name}"


def test_ctypes_mirror_matches_header():
    assert sorted(_abi.hip_symbol_names()) == _declared_in_header()


def test_struct_layouts():
    assert ctypes.sizeof(_abi.ErrorBoundC) == 8
    assert ctypes.sizeof(_abi.BinViewColC) == 32
    assert ctypes.sizeof(_abi.SegmentsC) == 8 + 3 * 8 + 32 + 2 * 8 + 2 * 32
    assert ctypes.sizeof(_abi.GridMetricsC) == 10 * 8
    assert ctypes.sizeof(_abi.AggStateC) == 24
    assert ctypes.sizeof(_abi.SegmentsOwnedC) == ctypes.sizeof(_abi.SegmentsC) + 8 + 8 + 8 + 8
    assert ctypes.sizeof(_abi.GridResultC) == 3 * 8 + 3 * 8 + ctypes.sizeof(_abi.GridMetricsC) + 8


def test_struct_layouts_match_the_pinned_numbers_of_the_header():
    """include/mdb_format.h pins sizeof/offsetof of every ABI struct with static assertions; the
    same numbers are asserted in rust/modelardb_hip/src/sys.rs. The ctypes mirror must agree."""
    text = open(os.path.join(REPO_ROOT, "include", "mdb_format.h")).read()
    mirror = {"mdb_error_bound": _abi.ErrorBoundC, "mdb_binview_col": _abi.BinViewColC,
              "mdb_segments": _abi.SegmentsC, "mdb_grid_metrics": _abi.GridMetricsC,
              "mdb_agg_state": _abi.AggStateC, "mdb_segments_owned": _abi.SegmentsOwnedC,
              "mdb_grid_result": _abi.GridResultC, "mdb_grid_input": _abi.GridInputC,
              "mdb_grid_request": _abi.GridRequestC, "mdb_chunk": _abi.ChunkC}
    sizes = dict(re.findall(r"MDB_LAYOUT_ASSERT\(sizeof\((\w+)\) == (\d+)\)", text))
    offsets = re.findall(r"MDB_LAYOUT_ASSERT\(offsetof\((\w+), (\w+)\) == (\d+)\)", text)
    assert set(mirror) <= set(sizes) and len(offsets) >= 30
    for name, struct in mirror.items():
        assert ctypes.sizeof(struct) == int(sizes[name]), name
    for name, field, offset in offsets:
        if name in mirror:
            assert getattr(mirror[name], field).offset == int(offset), (name, field)
    rust = open(os.path.join(REPO_ROOT, "rust", "modelardb_hip", "src", "sys.rs")).read()
    for name, size in sizes.items():
        assert re.search(rf"size_of::<{name}>\(\) == {size}\b", rust), f"sys.rs does not pin sizeof({name})"
    for name, field, offset in offsets:
        assert re.search(rf"offset_of!\({name}, {field}\) == {offset}\b", rust), (name, field)


def test_crate_helpers_through_the_c_abi_match_the_oracle():
    """is_value_within_error_bound / are_compressed_timestamps_regular (lib.rs:30-33) are plain host
    arithmetic behind the C ABI; the reference's own cases (models/mod.rs:298-405,
    timestamps.rs:457-478) and a random sweep against the oracle."""
    import oracle_lib as ora
    absolute, relative, lossless = (mdb.error_bound("absolute", 1.0), mdb.error_bound("relative", 10.0),
                                    mdb.error_bound("lossless"))
    assert mdb.is_value_within_error_bound(absolute, 10.0, 11.0)
    assert mdb.is_value_within_error_bound(relative, 10.0, 11.0)
    assert not mdb.is_value_within_error_bound(lossless, 10.0, 11.0)
    nan, inf = float("nan"), float("inf")
    for eb in (absolute, relative, lossless):
        assert mdb.is_value_within_error_bound(eb, nan, nan)
        assert mdb.is_value_within_error_bound(eb, inf, inf)
        for a, b in ((inf, -inf), (nan, 1.0), (1.0, nan), (inf, 1.0), (1.0, -inf)):
            assert not mdb.is_value_within_error_bound(eb, a, b)
    rng = np.random.default_rng(5)
    for _ in range(2000):
        eb = (mdb.error_bound("absolute", float(10.0 ** rng.uniform(-6, 3))) if rng.random() < 0.5
              else mdb.error_bound("relative", float(rng.uniform(1e-3, 100.0))))
        real = float(np.float32(rng.normal() * 10.0 ** rng.integers(-3, 6)))
        approximate = float(np.float32(real * (1.0 + rng.normal() * 0.05)))
        assert mdb.is_value_within_error_bound(eb, real, approximate) == \
            bool(ora.is_value_within_error_bound(eb, real, approximate))
    assert mdb.are_compressed_timestamps_regular(b"")
    assert mdb.are_compressed_timestamps_regular(bytes([0x05]))
    assert mdb.are_compressed_timestamps_regular(bytes([0x00, 0x80]))
    assert not mdb.are_compressed_timestamps_regular(bytes([0x80, 0x01]))
    for data in (ora.compress_residual_timestamps(np.arange(5) * 100),
                 ora.compress_residual_timestamps(np.array([0, 100, 250, 300, 470, 600]))):
        assert mdb.are_compressed_timestamps_regular(data) == bool(ora.are_compressed_timestamps_regular(data))


def test_rust_binding_declares_every_symbol():
    rust = open(os.path.join(REPO_ROOT, "rust", "modelardb_hip", "src", "sys.rs")).read()
    for name in _declared_in_header():
        assert re.search(rf"pub fn {name}\(", rust), f"sys.rs does not declare {name}"


def test_rust_patches_apply_to_the_reference():
    """The three call-site patches apply cleanly to the reference tree (present in the authoring
    container only; nothing at run time reads it)."""
    import glob
    import subprocess
    reference = "/root/reference"
    if not os.path.isdir(os.path.join(reference, "crates")):
        pytest.skip("the reference tree is not present")
    patches = sorted(glob.glob(os.path.join(REPO_ROOT, "rust", "patches", "*.patch")))
    assert len(patches) == 4
    for patch in patches:
        with open(patch) as f:
            done = subprocess.run(["patch", "-p1", "--dry-run", "--force", "-d", reference], stdin=f,
                                  capture_output=True, text=True)
        assert done.returncode == 0, done.stdout + done.stderr
        assert "FAILED" not in done.stdout and "fuzz" not in done.stdout, done.stdout


def test_patched_crates_depend_on_the_binding_they_use(tmp_path):
    """The four patches applied IN SEQUENCE to a copy of the crates they touch (no toolchain can compile the result
    here, so what can be checked is checked): every patch applies on top of the ones before it without fuzz, every crate
    whose patched sources name `modelardb_hip::` lists the crate in its Cargo.toml, and every function of the binding
    the patched sources call exists in rust/modelardb_hip/src/lib.rs."""
    import glob
    import shutil
    import subprocess
    reference = "/root/reference"
    if not os.path.isdir(os.path.join(reference, "crates")):
        pytest.skip("the reference tree is not present")
    crates = ("modelardb_storage", "modelardb_compression", "modelardb_server")
    for crate in crates:
        shutil.copytree(os.path.join(reference, "crates", crate), tmp_path / "crates" / crate,
                        ignore=shutil.ignore_patterns("target"))
    for patch in sorted(glob.glob(os.path.join(REPO_ROOT, "rust", "patches", "*.patch"))):
        with open(patch) as f:
            done = subprocess.run(["patch", "-p1", "--force", "-d", str(tmp_path)], stdin=f, capture_output=True, text=True)
        assert done.returncode == 0, done.stdout + done.stderr
        assert "FAILED" not in done.stdout and "fuzz" not in done.stdout, done.stdout
    binding = open(os.path.join(REPO_ROOT, "rust", "modelardb_hip", "src", "lib.rs")).read()
    users = 0
    for crate in crates:
        sources = []
        for directory, _, files in os.walk(tmp_path / "crates" / crate / "src"):
            sources += [open(os.path.join(directory, name)).read() for name in files if name.endswith(".rs")]
        uses = [source for source in sources if "modelardb_hip::" in source]
        if not uses:
            continue
        users += 1
        manifest = open(tmp_path / "crates" / crate / "Cargo.toml").read()
        assert re.search(r'^modelardb_hip = \{ path = "\.\./modelardb_hip" \}$', manifest, re.M), \
            f"{crate} uses modelardb_hip:: but its Cargo.toml does not depend on it"
        for source in uses:   # free functions and associated items of the crate the sources name
            for item in set(re.findall(r"modelardb_hip::([a-z_]+)\(", source)):
                assert re.search(rf"pub fn {item}\(", binding), f"{crate} calls modelardb_hip::{item}, which the binding lacks"
            for method in ("grid_submit", "aggregate_list", "aggregate_range_list", "compress_chunks", "split_and_compress",
                           "compress_univariate"):
                if f".{method}(" in source:
                    assert re.search(rf"pub fn {method}\(", binding), f"{crate} calls .{method}(), which the binding lacks"
    assert users == 3


def test_version_string():
    assert b"gfx950" in mdb.load_hip_library().mdb_version()


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(mdb.HipError):
        mdb.Context(0)


def test_null_arguments_are_errors_not_crashes():
    library = mdb.load_hip_library()
    assert library.mdb_grid_count(None, None, None) == 1
    assert b"NULL" in library.mdb_last_error()
    assert library.mdb_init(0, None) == 1
    assert library.mdb_trim(None, None) == 1
    assert library.mdb_close(None) == 0
    assert library.mdb_comm_init(None, 0, 1, None) == 1
    assert library.mdb_agg_all_reduce(None, None, None) == 1
    assert library.mdb_comm_unique_id(None) == 1
    assert library.mdb_agg_merge(None, None) == 1
    assert library.mdb_is_value_within_error_bound(mdb.error_bound("lossless"), 1.0, 1.0, None) == 1
    assert library.mdb_are_compressed_timestamps_regular(None, 4, None) == 1
    assert library.mdb_split_and_compress_univariate(None, None, None, None, 0, 0, None) == 1
    assert library.mdb_segments_validate_dev(None, None) == 1
    assert library.mdb_grid_submit(None, None, 0, None, None) == 1
    assert library.mdb_grid_wait(None, None) == 1
    library.mdb_grid_cancel(None)
    assert library.mdb_grid_result_tag_views(None, 0) is None
    assert library.mdb_compress_chunk_list(None, None, 0, mdb.error_bound("lossless"), None) == 1
    assert library.mdb_replicate_views(None, None, 3, 0, None, 0) == 1


def test_replicate_views_is_host_arithmetic():
    """mdb_replicate_views needs no context and no GPU: every segment's 16-byte view once per reconstructed row,
    buffer_index of the long ones moved (grid_exec.rs:339-346), also when the fill is split over threads."""
    library = mdb.load_hip_library()
    rng = np.random.default_rng(11)
    for n_segments, most in ((0, 1), (5, 4), (3000, 90), (70_000, 4)):
        views = rng.integers(0, 256, (n_segments, 16), dtype=np.uint8)
        lengths = rng.integers(0, 40, n_segments).astype("<i4")
        views[:, 0:4] = lengths.view(np.uint8).reshape(-1, 4)
        views[:, 8:12] = rng.integers(0, 5, n_segments).astype("<i4").view(np.uint8).reshape(-1, 4)
        rows = rng.integers(0, most, n_segments).astype(np.uint32)
        expected = np.repeat(views, rows, axis=0)
        long_rows = np.repeat(lengths > 12, rows)
        index = expected[:, 8:12].copy().view("<i4").reshape(-1)
        index[long_rows] += 3
        expected[:, 8:12] = index.view(np.uint8).reshape(-1, 4)
        for misalign in (0, 8):  # (an output that is not 16-byte aligned takes the plain stores)
            backing = np.zeros(16 * len(expected) + 64, dtype=np.uint8)
            start = (-backing.ctypes.data) % 16 + misalign
            out = backing[start:start + 16 * len(expected)]
            assert library.mdb_replicate_views(views.ctypes.data, rows.ctypes.data, n_segments, 3, out.ctypes.data,
                                               len(expected)) == 0, library.mdb_last_error()
            assert np.array_equal(out.reshape(-1, 16), expected)
    assert library.mdb_replicate_views(views.ctypes.data, rows.ctypes.data, n_segments, 0, out.ctypes.data, 1) == 1
    assert b"capacity" in library.mdb_last_error()


def test_segment_batch_arrow_round_trip():
    rows = [
        (0, 100, 500, bytes([5]), 1.5, 1.5, b"", b""),
        (1, 600, 1000, bytes([5]), 1.0, 9.0, bytes([0]), bytes(range(20)) + bytes([2])),
        (2, 1100, 1100, b"", 3.0, 3.0, bytes(range(40)), b""),
    ]
    batch = mdb.SegmentBatch.from_rows(rows)
    arrow = batch.to_arrow()
    assert arrow.schema.names[:4] == ["model_type_id", "start_time", "end_time", "timestamps"]
    assert str(arrow.schema.field("timestamps").type) == "binary_view"
    back = mdb.SegmentBatch.from_arrow(arrow)
    for got, expected in zip(back.rows(), rows):
        assert got[:4] == expected[:4] and got[6:] == expected[6:]
        assert np.float32(got[4]) == np.float32(expected[4])
    assert np.isnan(back.error).all()


def test_switches_are_read_from_the_environment_once_and_set_without_it(monkeypatch):
    """mdb_option / mdb_set_option / mdb_reload_options (include/mdb.h): the table of MDB_* switches is a snapshot of
    the environment - a later setenv is not seen until the table is reloaded - and a host sets a switch without
    touching its environment."""
    library = ctypes.CDLL(_abi.HIP_LIBRARY_PATH)
    library.mdb_option.restype = ctypes.c_char_p
    library.mdb_option.argtypes = [ctypes.c_char_p]
    library.mdb_set_option.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    monkeypatch.setenv("MDB_TEST_SWITCH", "first")
    assert library.mdb_reload_options() == 0
    assert library.mdb_option(b"MDB_TEST_SWITCH") == b"first"
    monkeypatch.setenv("MDB_TEST_SWITCH", "second")
    assert library.mdb_option(b"MDB_TEST_SWITCH") == b"first"          # (a snapshot: no getenv per call)
    assert library.mdb_reload_options() == 0
    assert library.mdb_option(b"MDB_TEST_SWITCH") == b"second"
    assert library.mdb_set_option(b"MDB_TEST_SWITCH", b"third") == 0
    assert library.mdb_option(b"MDB_TEST_SWITCH") == b"third"
    assert library.mdb_set_option(b"MDB_TEST_SWITCH", None) == 0
    assert library.mdb_option(b"MDB_TEST_SWITCH") is None
    assert library.mdb_option(b"MDB_NEVER_SET") is None
    assert library.mdb_set_option(b"PATH", b"x") == 1                   # (only the library's own names)
    monkeypatch.delenv("MDB_TEST_SWITCH")
    assert library.mdb_reload_options() == 0
    assert library.mdb_option(b"MDB_TEST_SWITCH") is None
