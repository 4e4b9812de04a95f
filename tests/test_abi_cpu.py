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
